"""Progressive accumulation with checkpoint / resume.

The reference renders all samples in one go and writes the PNG once at the end (main.cpp:235-237);
a 5000-spp run that dies loses everything.  Here a frame is rendered in passes of `pass_spp`
samples (SrtRenderParams.sampleFirst selects the absolute sample range, so the passes add up to
exactly the samples a single render would draw); after every pass the float accumulators and the
next sample index can be written to an .npz and picked up again later.  A checkpoint records the
fingerprint of the scene and camera it was rendered with; resuming on a context that holds anything else
is refused."""
import os

import numpy as np


class ProgressiveRender:
    def __init__(self, ctx, abi, width, height, max_bounce, seed=1, background=(0.53, 0.81, 0.92)):
        self.ctx, self.abi = ctx, abi
        self.width, self.height, self.max_bounce, self.seed, self.background = width, height, max_bounce, seed, background
        self.accum = np.zeros((height, width, 4), np.float32)  # rgb sums + sample count
        self.next_sample = 0

    def render_pass(self, pass_spp, spp_chunks=1):
        p = self.abi.default_render_params(self.width, self.height, pass_spp, self.max_bounce, seed=self.seed,
                                           background=self.background, spp_chunks=spp_chunks,
                                           sample_first=self.next_sample)
        acc, _ = self.ctx.render_image(p, want_rgba=False)
        self.accum += acc  # pass sums added in pass order
        self.next_sample += pass_spp
        return self.next_sample

    def save(self, path):
        tmp = path + ".tmp.npz"
        np.savez(tmp, accum=self.accum, next_sample=self.next_sample, width=self.width, height=self.height,
                 max_bounce=self.max_bounce, seed=self.seed, background=np.asarray(self.background, np.float32),
                 fingerprint=np.array(self.ctx.fingerprint()))
        os.replace(tmp, path)

    @classmethod
    def resume(cls, ctx, abi, path):
        z = np.load(path)
        if "fingerprint" in z.files and str(z["fingerprint"]) != ctx.fingerprint():
            raise ValueError("checkpoint %s was rendered with a different scene or camera than the context holds" % path)
        self = cls(ctx, abi, int(z["width"]), int(z["height"]), int(z["max_bounce"]), int(z["seed"]),
                   tuple(float(x) for x in z["background"]))
        self.accum = z["accum"].copy()
        self.next_sample = int(z["next_sample"])
        return self

    def image_rgba8(self):
        """writeColorTarget (color.h:25-41) on the running sums."""
        n = max(1, self.next_sample)
        c = np.sqrt(self.accum[..., :3] * np.float32(1.0 / n))
        c = np.where(np.isnan(c), 0.0, np.clip(c, 0.0, 0.999)) * 256.0
        out = np.empty((self.height, self.width, 4), np.uint8)
        out[..., :3] = c.astype(np.uint8)
        out[..., 3] = 255
        return out
