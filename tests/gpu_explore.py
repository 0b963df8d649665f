"""Exploration run on the GPU box: parity numbers + timings printed, nothing asserted.
Usage: python tests/gpu_explore.py [scene] [W H spp bounces]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.oracle_py as O  # noqa: E402

srt = importlib.import_module("sexy-raytracer_amd")
abi = srt.abi
dev = srt.device()


def primary_rays(cam, W, H, n_per_pixel=1, seed=3):
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    u = ((xs + rng.random((H, W))) / (W - 1)).astype(np.float32).ravel()
    v = (((H - ys) + rng.random((H, W))) / (H - 1)).astype(np.float32).ravel()
    o = np.array(cam.origin[:], np.float32)
    ll, hz, vt = (np.array(a[:], np.float32) for a in (cam.lleft, cam.horizontal, cam.vertical))
    d = ll[None] + u[:, None] * hz[None] + v[:, None] * vt[None] - o[None]
    rays = np.zeros(len(u), abi.RAY_DTYPE)
    rays["o"] = o
    rays["d"] = d.astype(np.float32)
    rays["time"] = rng.random(len(u)).astype(np.float32)
    rays["tMin"] = 0.001
    rays["tMax"] = np.inf
    return rays


def compare_hits(a, b, label):
    same_prim = a["prim"] == b["prim"]
    hit = a["prim"] >= 0
    print("[%s] rays %d, hits %d, prim mismatches %d" % (label, len(a), hit.sum(), (~same_prim).sum()))
    m = same_prim & hit
    for f in ("t", "p", "normal", "tangent", "bitangent", "uv"):
        x, y = a[f][m], b[f][m]
        bitexact = (x.view(np.uint32) == y.view(np.uint32))
        if bitexact.ndim > 1:
            bitexact = bitexact.all(axis=1)
        diff = np.abs(x.astype(np.float64) - y.astype(np.float64))
        print("   %-9s bit-exact %8d / %d   max abs diff %.3e" % (f, bitexact.sum(), m.sum(), np.nanmax(diff) if diff.size else 0))
    for f in ("frontFace", "material", "nodeVisits", "boxPasses", "triTests", "sphereTests"):
        print("   %-11s mismatches %d" % (f, (a[f] != b[f]).sum()))


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "masterchief"
    W, H, spp, mb = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (426, 240, 16, 4)
    sb = srt.scenes.SCENES[scene]()
    ctx = dev.Context(0)
    print(ctx.device_info())
    t = time.time()
    ctx.upload_scene(sb)
    print("upload %.3fs, bvh depth %d" % (time.time() - t, ctx.bvh_depth()))
    osc = O.OracleScene(sb)
    on, _ = osc.bvh(0)
    print("bvh identical to oracle:", ctx.bvh(0).tobytes() == on.tobytes())
    cam = dev.make_camera(abi.default_camera_params())
    ctx.set_camera(cam)

    rays = primary_rays(cam, 426, 240)
    hg = ctx.trace(rays)
    ho = osc.trace(rays)
    compare_hits(hg, ho, "primary faithful")
    # secondary rays: from hit points in random directions
    rng = np.random.default_rng(5)
    m = ho["prim"] >= 0
    sec = np.zeros(m.sum(), abi.RAY_DTYPE)
    sec["o"] = ho["p"][m]
    dirs = rng.normal(size=(m.sum(), 3)).astype(np.float32)
    sec["d"] = dirs
    sec["time"] = rays["time"][m]
    sec["tMin"] = 0.001
    sec["tMax"] = np.inf
    compare_hits(ctx.trace(sec), osc.trace(sec), "secondary faithful")
    compare_hits(ctx.trace(rays, abi.SRT_TRAVERSE_CLOSEST), osc.trace(rays, abi.SRT_TRAVERSE_CLOSEST), "primary closest (oracle brute force)")

    p = abi.default_render_params(W, H, spp, mb, seed=11, count_stats=1)
    t = time.time()
    acc_g, rgba_g = ctx.render_image(p)
    print("gpu render (counting) wall %.3fs kernel %.3f ms" % (time.time() - t, ctx.last_kernel_ms()))
    st = ctx.stats()
    print("gpu stats", st)
    p.countStats = 0
    for _ in range(2):
        t = time.time()
        acc_g, rgba_g = ctx.render_image(p)
        ms = ctx.last_kernel_ms()
        print("gpu render wall %.3fs kernel %.3f ms  -> %.1f Msamples/s" % (time.time() - t, ms, W * H * spp / ms / 1e3))
    t = time.time()
    acc_o, rgba_o, so = osc.render(cam, p, O.RNG_COUNTER, threads=os.cpu_count() or 8)
    dt = time.time() - t
    print("oracle render %.2fs (%d threads) -> %.2f Msamples/s" % (dt, os.cpu_count() or 8, W * H * spp / dt / 1e6))
    print("oracle stats", so)
    for k in ("samples", "rays", "nodeVisits", "boxPasses", "triTests", "sphereTests", "shadedTriHits", "texelFetches"):
        print("   counter %-14s gpu %12d oracle %12d %s" % (k, st[k], so[k], "OK" if st[k] == so[k] else "DIFF"))
    a, b = acc_g[..., :3], acc_o[..., :3]
    bit = (a.view(np.uint32) == b.view(np.uint32)).all(axis=2)
    both_nan = np.isnan(a) & np.isnan(b)
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
    rel[both_nan] = 0
    print("pixels bit-exact %d / %d (%.4f%%)" % (bit.sum(), bit.size, 100.0 * bit.mean()))
    print("max rel err %.3e, pixels with rel err > 1e-5: %d, > 1e-3: %d, nan pixels gpu %d oracle %d" % (
        np.nanmax(rel), (rel.max(axis=2) > 1e-5).sum(), (rel.max(axis=2) > 1e-3).sum(),
        np.isnan(a).any(axis=2).sum(), np.isnan(b).any(axis=2).sum()))
    print("rgba mismatching bytes %d / %d, max byte diff %d" % ((rgba_g != rgba_o).sum(), rgba_g.size,
                                                                 np.abs(rgba_g.astype(int) - rgba_o.astype(int)).max()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    from PIL import Image
    Image.fromarray(rgba_g).save(os.path.join(ROOT, "gpurun_out", "explore_%s_gpu.png" % scene))
    Image.fromarray(rgba_o).save(os.path.join(ROOT, "gpurun_out", "explore_%s_oracle.png" % scene))


if __name__ == "__main__":
    main()
