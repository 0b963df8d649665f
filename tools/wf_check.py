"""First-light check of the path-pool kernel (srt_wavefront.hip) against the step-scheduler kernel over the same tree:
identical accumulators on growing frames, kernel times side by side.  usage: python tools/wf_check.py [scene ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
ctx = dev.Context(0)
cam = dev.make_camera(abi.default_camera_params())
scenes = sys.argv[1:] or ["spheres", "masterchief", "iron"]
for name in scenes:
    ctx.upload_scene(srt.scenes.SCENES[name]())
    ctx.set_camera(cam)
    mb = 8 if name in ("spheres", "sphere_field") else 4
    for (W, H, spp, chunks) in ((64, 36, 2, 1), (160, 90, 8, 0), (426, 240, 16, 0), (1280, 720, 64, 0)):
        out = {}
        for wf in (0, 1):
            ctx.set_tunable("wavefront", wf)  # 1: every tree that fits, however small
            local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
            p = abi.default_render_params(W, H, spp, mb, seed=3, spp_chunks=chunks)
            ctx.render_tiles(p, local.data_ptr(), None)
            torch.cuda.synchronize()
            ms = ctx.last_kernel_ms()
            out[wf] = (local.cpu().numpy(), ms, ctx.launch_info())
        a, b = out[0][0], out[1][0]
        same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1).mean()
        print("%-12s %4dx%-4d %3d spp chunks %d: step-scheduler %8.3f ms, path-pool %8.3f ms (x%.2f), identical pixels %.6f, mode %s, lds %d" % (
            name, W, H, spp, chunks, out[0][1], out[1][1], out[0][1] / out[1][1], same, out[1][2]["lds_tree_mode"], out[1][2]["lds_bytes"]), flush=True)
        if same < 1.0:
            d = np.argwhere(~(a.view(np.uint32) == b.view(np.uint32)).all(axis=-1))[:5]
            for t, l in d:
                print("   tile %d lane %d: %s vs %s" % (t, l, a[t, l], b[t, l]))
ctx.close()
