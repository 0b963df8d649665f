"""Load balance of the interleaved tile partition, measured on ONE GPU: each rank's share of the
headline frame is rendered by itself (tileFirst = r, tileStride = N) and the kernel times compared.
mean/max of the per-rank times is the strong-scaling efficiency the partition alone allows at N ranks.
usage: python tools/partition_balance.py [workload] [spp]  -> one JSON line per N"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()


ctx = None
TUN = {}  # diagnostic tunables applied to the context (srtSetTunable)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "masterchief"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    W, H, mb = (int(x) for x in os.environ.get("FRAME", "1280,720,4").split(","))
    global ctx
    if ctx is None:
        ctx = dev.Context(0)
        ctx.upload_scene(srt.scenes.SCENES[scene]())
        ctx.set_camera(dev.make_camera(abi.default_camera_params()))
    for k, v in TUN.items():
        ctx.set_tunable(k, v)
    ranks = [int(x) for x in os.environ.get("RANKS", "1,2,4,8").split(",")]
    one = None
    for n in ranks:
        nloc = dev.num_local_tiles(W, H, n)
        local = torch.zeros((nloc, 64, 4), dtype=torch.float32, device="cuda")
        ms = []
        for r in range(n):
            p = abi.default_render_params(W, H, spp, mb, seed=1, tile_first=r, tile_stride=n, spp_chunks=int(os.environ.get("CHUNKS", "0")))
            best = 1e9
            for _ in range(2):
                for _ in range(int(os.environ.get("BACK2BACK", "1"))):  # >1: time a launch queued behind another
                    ctx.render_tiles(p, local.data_ptr(), None)
                best = min(best, ctx.last_kernel_ms())
            ms.append(round(best, 3))
        print(json.dumps({"unit_tiles": TUN.get("unit_tiles", "default"), "queues": TUN.get("queues", "default"), "tile_block": TUN.get("tile_block", "default"), "ranks": n, "frame": [W, H, spp], "kernel_ms": ms, "max_ms": max(ms),
                          "partition_efficiency": round(sum(ms) / n / max(ms), 4),
                          "vs_one_rank": None if n == 1 or one is None else round(one / (n * max(ms)), 4)}), flush=True)
        if n == 1:
            one = ms[0]


for nq in os.environ.get("QUEUES", "").split(","):
    if nq:
        TUN["queues"] = int(nq)
    for blk in os.environ.get("BLOCKS", "").split(","):
        if blk:
            TUN["tile_block"] = int(blk)
        for ut in os.environ.get("UNITS", "").split(","):
            if ut:
                TUN["unit_tiles"] = int(ut)
            main()
