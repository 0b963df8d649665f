"""Step profile of the path-pool kernel (profiling variant, tunable wf_profile = 1): share of wave time, executions,
mean lane fill and clocks per execution per step kind.  usage: python tools/wf_profile.py [scene [spp [W H]]]
env: SRT_WF_POOL, SRT_WF_SWAP_MIN, SRT_WF_SWAP_BIG (read at srtCreate)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
scene = sys.argv[1] if len(sys.argv) > 1 else "masterchief"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1280, 720)
mb = 8 if scene in ("spheres", "sphere_field") else 4
ctx = dev.Context(0)
ctx.upload_scene(srt.scenes.scene_soup(int(scene[5:])) if scene.startswith("soup:") else srt.scenes.SCENES[scene]())
ctx.set_camera(dev.make_camera(abi.default_camera_params()))
local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
p = abi.default_render_params(W, H, spp, mb, seed=1, spp_chunks=0)
for prof in (0, 1):
    ctx.set_tunable("wf_profile", prof)
    ctx.render_tiles(abi.default_render_params(W, H, 1, mb, seed=1), local.data_ptr(), None)  # warm-up: the first launch of a kernel pays for its loading
    ctx.render_tiles(p, local.data_ptr(), None)
    torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()
    print("%s %dx%d %d spp, %s variant: %.3f ms, %.1f Msamples/s, launch %s" % (scene, W, H, spp, "profiling" if prof else "production", ms,
                                                                           W * H * spp / ms / 1e3, ctx.launch_info()))
pr = ctx.wf_profile()
tot = pr["total_clocks"]
samples = W * H * spp
for k in ("node", "far_node", "prim", "swap", "hit0", "hit1", "hit2", "restart", "new_item", "idle", "lost_claim"):
    v = pr[k]
    print("%-10s %5.1f%% of wave time, %11d executions, mean fill %5.1f lanes, %8.1f clocks/execution, %.3f executions/sample" % (
        k, 100.0 * v["clocks"] / tot, v["runs"], v["lanes"] / max(1, v["runs"]), v["clocks"] / max(1, v["runs"]), v["runs"] / samples))
if pr["far_node"]["runs"]:
    print("hybrid form: the closing visit of a round served %.1f lanes outside LDS and %.1f lanes in LDS on average" % (
        pr["far_node"]["lanes"] / pr["far_node"]["runs"], pr["unused"]["lanes"] / pr["far_node"]["runs"]))
print("slab certificate: %.4f%% of lane visits undecided, %.2f%% of wave visits ran the IEEE test" % (
    100.0 * pr["lost_claim"]["lanes"] / max(1, pr["node"]["lanes"]), 100.0 * pr["idle"]["lanes"] / max(1, pr["node"]["runs"])))
print("%-10s %5.1f%% of wave time, %d decisions (%.2f per sample)" % ("scheduling", 100.0 * pr["sched_clocks"] / tot, pr["decisions"], pr["decisions"] / samples))
sw = max(1, pr["swap"]["runs"])
ms_ = pr["mean_seen"]
h1n = max(1.0, ms_["idle"] * pr["decisions"])
print("hit step of class 1 (%d timed): claim + context %.0f clocks, hit record %.0f, shade %.0f (the rest: stores, fence, hand-over)" % (
    h1n, ms_["at_node"] * pr["decisions"] / h1n, ms_["at_prim"] * pr["decisions"] / h1n, ms_["finished"] * pr["decisions"] / h1n))
print("swap step: claim %.0f clocks, hand-over %.0f, arrival + set-up %.0f" % tuple(ms_[k] * pr["decisions"] / sw for k in ("ready_fill", "fullest_ring", "restart_fill")))
ctx.close()
