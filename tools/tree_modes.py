"""Headline scene under the four tree / traversal combinations (tools, not a parity path)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
W, H, spp, mb = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 4
ctx = dev.Context(0)
local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
for name, builder, trav in (("reference tree, FAITHFUL", abi.SRT_BUILDER_REFERENCE, abi.SRT_TRAVERSE_FAITHFUL),
                            ("reference tree, CLOSEST", abi.SRT_BUILDER_REFERENCE, abi.SRT_TRAVERSE_CLOSEST),
                            ("LBVH, CLOSEST", abi.SRT_BUILDER_LBVH, abi.SRT_TRAVERSE_CLOSEST),
                            ("PLOC, CLOSEST", abi.SRT_BUILDER_PLOC, abi.SRT_TRAVERSE_CLOSEST)):
    sb = srt.scenes.scene_masterchief()
    sb.world[-1].builder = builder
    ctx.upload_scene(sb)
    ctx.set_camera(dev.make_camera(abi.default_camera_params()))
    p = abi.default_render_params(W, H, 8, mb, seed=1, spp_chunks=0, count_stats=1, traversal=trav)
    ctx.render_tiles(p, local.data_ptr(), None); torch.cuda.synchronize(); st = ctx.stats()
    p = abi.default_render_params(W, H, spp, mb, seed=1, spp_chunks=0, traversal=trav)
    best = 1e9
    for _ in range(2):
        ctx.render_tiles(p, local.data_ptr(), None); best = min(best, ctx.last_kernel_ms())
    print("%-26s %8.1f Msamples/s  node visits/ray %.1f  prim tests/ray %.2f" % (name, W*H*spp/best/1e3, st["nodeVisits"]/st["rays"], (st["triTests"]+st["sphereTests"])/st["rays"]), flush=True)
