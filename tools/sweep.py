"""In-process parameter sweep of the render kernel's scheduler thresholds / chunking (GPU box)."""
import importlib, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "masterchief"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    W, H, mb = 1280, 720, 4
    ctx = dev.Context(0)
    ctx.upload_scene(srt.scenes.SCENES[scene]())
    ctx.set_camera(dev.make_camera(abi.default_camera_params()))
    nloc = dev.num_local_tiles(W, H, 1)
    local = torch.zeros((nloc, 64, 4), dtype=torch.float32, device="cuda")
    def run(chunks, reps=2):
        p = abi.default_render_params(W, H, spp, mb, seed=1, spp_chunks=chunks)
        best = 1e9
        for _ in range(reps):
            ctx.render_tiles(p, local.data_ptr(), None)
            best = min(best, ctx.last_kernel_ms())
        return W * H * spp / best / 1e3
    run(8)
    grid = {"chunks": [int(x) for x in os.environ.get("SWEEP_CHUNKS", "8,16,32").split(",")],
            "shade": [int(x) for x in os.environ.get("SWEEP_SHADE", "16,24,32,40").split(",")],
            "prim": [int(x) for x in os.environ.get("SWEEP_PRIM", "12,20,28").split(",")],
            "burst": [int(x) for x in os.environ.get("SWEEP_BURST", "16").split(",")],
            "hit": [int(x) for x in os.environ.get("SWEEP_HIT", "24").split(",")],
            "fuse": [int(x) for x in os.environ.get("SWEEP_FUSE", "32").split(",")],
            "again": [int(x) for x in os.environ.get("SWEEP_AGAIN", "4").split(",")],
            "keep": [int(x) for x in os.environ.get("SWEEP_KEEP", "6").split(",")]}
    for c, sm, pm, nb, hm, fu, ag, kp in itertools.product(grid["chunks"], grid["shade"], grid["prim"], grid["burst"], grid["hit"], grid["fuse"],
                                                           grid["again"], grid["keep"]):
        for k, v in (("shade_min", sm), ("prim_min", pm), ("node_burst", nb), ("hit_min", hm), ("fuse_min", fu), ("prim_again_min", ag), ("keep_eighths", kp)):
            ctx.set_tunable(k, v)
        print("chunks %3d shadeMin %2d primMin %2d burst %2d hitMin %2d fuseMin %2d againMin %2d keepEighths %d : %8.1f Msamples/s" % (
            c, sm, pm, nb, hm, fu, ag, kp, run(c)), flush=True)

main()
