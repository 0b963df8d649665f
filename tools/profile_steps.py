"""Scheduler profile of the counting variant: time share, executions and lane fill per step kind."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
scene = sys.argv[1] if len(sys.argv) > 1 else "masterchief"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H, mb = 1280, 720, 4
ctx = dev.Context(0)
ctx.upload_scene(srt.scenes.SCENES[scene]())
ctx.set_camera(dev.make_camera(abi.default_camera_params()))
local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
p = abi.default_render_params(W, H, spp, mb, seed=1, spp_chunks=0, count_stats=1)
ctx.render_tiles(p, local.data_ptr(), None)
ms = ctx.last_kernel_ms()
st = ctx.stats()
tot = st["cyclesTotal"]
print("kernel %.1f ms (counting variant), %.1f Msamples/s" % (ms, W * H * spp / ms / 1e3))
for k, n in (("Node", "nodeVisits"), ("Prim", None), ("Shade", "rays")):
    cyc, steps, lanes = st["cycles" + k], st["steps" + k], st["lanes" + k]
    print("%-5s: %5.1f%% of wave time, %12d step executions, mean fill %5.1f lanes, %7.1f clocks/execution" % (
        k, 100.0 * cyc / tot, steps, lanes / max(1, steps), cyc / max(1, steps)))
sched = tot - st["cyclesNode"] - st["cyclesPrim"] - st["cyclesShade"]
print("sched: %5.1f%% of wave time" % (100.0 * sched / tot))
print({k: st[k] for k in ("samples", "rays", "nodeVisits", "triTests", "sphereTests")})
if hasattr(dev.lib, "srtGetShadeProfile"):
    sp = ctx.shade_profile()
    hit_tot = sum(sp[0:4])
    print("hit step   : %d executions, mean fill %.1f lanes, %.0f clocks each: record %.0f%%, textures %.0f%%, direction draw %.0f%%, BRDF+rest %.0f%%" % (
        sp[5], sp[6] / max(1, sp[5]), hit_tot / max(1, sp[5]), *(100.0 * x / max(1, hit_tot) for x in sp[0:4])))
    print("restart    : %d executions, mean fill %.1f lanes, %.0f clocks each" % (sp[7], sp[8] / max(1, sp[7]), sp[4] / max(1, sp[7])))
