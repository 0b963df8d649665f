"""Synthetic seeded triangle soups (SURVEY 8d 'Synthetic'): working sets from cache-resident to
well past the 256 MiB Infinity Cache, to get HBM-bound points next to the cache-resident configs.
usage: python tools/roofline_soup.py <num_triangles> [spp]      (prints one JSON line)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()
sys.path.insert(0, ROOT)
from bench import algorithmic_bytes

n = int(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
# optional 3rd argument "lbvh" / "ploc": device-built tree + closest-hit traversal (not the parity path)
lbvh = len(sys.argv) > 3 and sys.argv[3] in ("lbvh", "ploc")
ploc = len(sys.argv) > 3 and sys.argv[3] == "ploc"
ref_closest = len(sys.argv) > 3 and sys.argv[3] == "closest"  # reference tree, closest-hit (ordered) traversal
W, H, mb = 1280, 720, 4
t = time.time()
sb = srt.scenes.scene_soup(n, seed=7, extent=6.0, size=max(0.01, 0.08 * (100000.0 / n) ** (1.0 / 3.0)),
                           builder=abi.SRT_BUILDER_PLOC if ploc else (abi.SRT_BUILDER_LBVH if lbvh else abi.SRT_BUILDER_REFERENCE))
trav = abi.SRT_TRAVERSE_CLOSEST if (lbvh or ref_closest) else abi.SRT_TRAVERSE_FAITHFUL
ctx = dev.Context(0)
ctx.upload_scene(sb)
build_s = time.time() - t
ctx.set_camera(dev.make_camera(abi.default_camera_params()))
local = torch.zeros((dev.num_local_tiles(W, H, 1), 64, 4), dtype=torch.float32, device="cuda")
p = abi.default_render_params(W, H, min(spp, 4), mb, seed=1, spp_chunks=0, count_stats=1, traversal=trav)
ctx.render_tiles(p, local.data_ptr(), None)
torch.cuda.synchronize()
st = ctx.stats()
bps = (algorithmic_bytes(st, W, H) - 16 * W * H) / st["samples"]
p = abi.default_render_params(W, H, spp, mb, seed=1, spp_chunks=int(os.environ.get("CHUNKS", "0")), traversal=trav)
best = 1e30
for _ in range(3):
    ctx.render_tiles(p, local.data_ptr(), None)
    best = min(best, ctx.last_kernel_ms())
ms = best
samples = W * H * spp
print(json.dumps({"tree": "device PLOC + closest hit" if ploc else "device LBVH + closest hit" if lbvh else ("reference bvh.h + closest hit" if ref_closest else "reference bvh.h + faithful"), "triangles": n, "nodes_MB": round((2 * n) * 32 / 1e6, 1), "tri_records_MB": round(n * 112 / 1e6, 1),
                  "bvh_depth": ctx.bvh_depth(), "build_upload_s": round(build_s, 2), "spp": spp,
                  "Msamples_per_s": round(samples / ms / 1e3, 2), "kernel_ms": round(ms, 3),
                  "rays_per_sample": round(st["rays"] / st["samples"], 3),
                  "node_visits_per_ray": round(st["nodeVisits"] / st["rays"], 2),
                  "tri_tests_per_ray": round(st["triTests"] / st["rays"], 2),
                  "algorithmic_bytes_per_sample": round(bps, 1),
                  "algorithmic_GBps": round(bps * samples / (ms * 1e-3) / 1e9, 1)}))
