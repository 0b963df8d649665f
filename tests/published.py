"""Pooled-residual comparison of a render of the main.cpp scene with the reference's published renders
(images/test-1kx240p.png, images/test-5kx720p.png), through the fixture tests/golden/published_blocks.npz
(tests/make_golden.py): the published image as 4x4-pixel block means, a per-pixel class map of what a
pinhole primary ray hits, and its pure-black (NaN) pixels.

What a systematic error would do: a wrong uv formula, a flipped normal-map axis, a wrong v flip or a wrong
Fresnel constant shifts the MEAN of the mesh or ground blocks (signed bias), and leaves a residual that does
not shrink when blocks are pooled; pure Monte-Carlo noise of two independent renders has zero mean and its
mean absolute difference halves every time the pool side doubles."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLASSES = {"sky": 0, "mesh": 1, "ground": 2, "light": 3, "iron": 4, "metal": 5}


def _dilate(mask, r):
    out = mask.copy()
    for _ in range(r):  # r rounds of a 3x3 cross: a diamond of radius r
        out = out | np.roll(out, 1, 0) | np.roll(out, -1, 0) | np.roll(out, 1, 1) | np.roll(out, -1, 1)
    return out


def compare(rgb, key):
    """rgb: (H, W, 3) float array in 8-bit levels (0..255), as writeColorTarget would store it.
    Returns {"bias": {class: signed mean difference mine - published over the blocks lying wholly in that
    class}, "mad": {pool side in pixels: mean |difference| of the pooled block means}, "blocks": n}.
    The iron sphere (procedural stand-in textures, its reference blobs are missing) is masked out with a
    margin for the depth-of-field blur."""
    z = np.load(os.path.join(GOLD, "published_blocks.npz"))
    pub = z[key + "_block4_x64"].astype(np.float64) / 64.0
    cls = z[key + "_class"]
    H, W = cls.shape
    assert rgb.shape[:2] == (H, W)
    hb, wb = H // 4, W // 4
    mine = rgb[:hb * 4, :wb * 4, :3].astype(np.float64).reshape(hb, 4, wb, 4, 3).mean((1, 3))
    excl = _dilate(cls == CLASSES["iron"], 12 if H >= 700 else 4)
    cb = cls[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4)
    valid = ~excl[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4).any((1, 3))
    diff = mine - pub
    out = {"bias": {}, "mad": {}, "blocks": int(valid.sum())}
    for name, c in CLASSES.items():
        pure = valid & (cb == c).all((1, 3))
        if pure.sum() >= 50:
            out["bias"][name] = float(diff[pure].mean())
    for pool in (1, 2, 4, 8):
        h2, w2 = hb // pool * pool, wb // pool * pool
        v = valid[:h2, :w2].reshape(h2 // pool, pool, w2 // pool, pool).all((1, 3))
        d = diff[:h2, :w2].reshape(h2 // pool, pool, w2 // pool, pool, 3).mean((1, 3))
        out["mad"][4 * pool] = float(np.abs(d[v]).mean())
    return out


def published_black(key):
    return np.load(os.path.join(GOLD, "published_blocks.npz"))[key + "_black_yx"]


def class_bias_linear(accum, spp, key):
    """Low-spp twin of compare(): per class, the relative difference of the summed LINEAR radiance
    (accum / spp, pooled over every block lying wholly in the class) against the published image's
    linearised blocks.  Pooling before tone mapping avoids the sqrt / clamp bias of noisy pixels."""
    z = np.load(os.path.join(GOLD, "published_blocks.npz"))
    pub = z[key + "_block4_linear_x65535"].astype(np.float64) / 65535.0
    cls = z[key + "_class"]
    H, W = cls.shape
    hb, wb = H // 4, W // 4
    mine = (accum[:hb * 4, :wb * 4, :3].astype(np.float64) / spp).reshape(hb, 4, wb, 4, 3).mean((1, 3))
    excl = _dilate(cls == CLASSES["iron"], 12 if H >= 700 else 4)
    cb = cls[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4)
    valid = ~excl[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4).any((1, 3))
    out = {}
    for name, c in CLASSES.items():
        pure = valid & (cb == c).all((1, 3))
        if pure.sum() >= 50:
            out[name] = float(np.nansum(mine[pure]) / pub[pure].sum() - 1.0)
    return out
