"""The path-pool kernel's hybrid form (srt_wavefront.hip HYBRID: the tree's top in LDS, the rest read from global
memory) against the 256-thread step-scheduler kernel on trees that do not fit a CU's LDS, and against the full
path-pool kernel on one that does (resident nodes capped): identical accumulators, kernel times side by side.
usage: python tools/hybrid_check.py [spp]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
ctx = dev.Context(0)
cam = dev.make_camera(abi.default_camera_params())
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
CASES = [  # scene, builder, [(label, {tunable: value}) ...]: the first entry is the baseline
    ("masterchief", lambda: srt.scenes.scene_masterchief(),
     [("path pool, whole tree in LDS", {}), ("hybrid, 24 resident", {"wf_resident_max": 24}), ("hybrid, 500 resident", {"wf_resident_max": 500}),
      ("hybrid, 2000 resident", {"wf_resident_max": 2000})]),
    ("army", lambda: srt.scenes.scene_masterchief_army(),
     [("256-thread step scheduler", {"wf_hybrid": 0})] + [("hybrid, %d visits per round" % r, {"wf_far_rounds": r}) for r in (1, 2, 3)] +
     [("hybrid, 1000 resident", {"wf_resident_max": 1000})]),
] + [
    ("soup_%dk" % (n // 1000), (lambda n=n: srt.scenes.scene_soup(n)),
     [("256-thread step scheduler", {"wf_hybrid": 0})] + [("hybrid, %d visits per round" % r, {"wf_far_rounds": r}) for r in (1, 2, 3)])
    for n in (16000, 50000, 200000, 1000000)
]
W, H = 1280, 720
for name, build, variants in CASES:
    if only and name not in only:
        continue
    sb = build()
    base = None
    for label, tun in variants:
        saved = {k: ctx.get_tunable(k) for k in tun}
        for k, v in tun.items():
            ctx.set_tunable(k, v)
        ctx.upload_scene(sb)  # wf_hybrid / wf_resident_max are read here
        ctx.set_camera(cam)
        local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
        p = abi.default_render_params(W, H, spp, 4, seed=3, spp_chunks=0)
        ms = []
        for _ in range(3):
            local.zero_()
            ctx.render_tiles(p, local.data_ptr(), None)
            torch.cuda.synchronize()
            ms.append(ctx.last_kernel_ms())
        got = local.cpu().numpy()
        info = ctx.launch_info()
        for k, v in saved.items():
            ctx.set_tunable(k, v)
        if base is None:
            base = (got, min(ms))
        same = (got.view(np.uint32) == base[0].view(np.uint32)).all(axis=-1).mean()
        print("%-12s %-32s %9.3f ms  %8.1f Msamples/s  x%.2f  identical pixels %.6f  mode %d lds %d" % (
            name, label, min(ms), W * H * spp / min(ms) / 1e3, base[1] / min(ms), same, info["lds_tree_mode"], info["lds_bytes"]), flush=True)
ctx.close()
