"""Design probe: rays/s of a traversal-only kernel (srtTraverseBench) on a ray mix like the path
tracer's (primaries + two generations of diffuse bounces), next to the megakernel's effective rate."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srt = importlib.import_module("sexy-raytracer_amd")
abi, dev = srt.abi, srt.device()

ctx = dev.Context(0)
ctx.upload_scene(srt.scenes.scene_masterchief())
cam = dev.make_camera(abi.default_camera_params())
W, H = 1280, 720
rng = np.random.default_rng(1)
# primaries in 8x8-tile order, like the render kernel's work items
ty, tx, ly, lx = np.meshgrid(np.arange(H // 8), np.arange(W // 8), np.arange(8), np.arange(8), indexing="ij")
ys, xs = (ty * 8 + ly).ravel(), (tx * 8 + lx).ravel()
u = ((xs + rng.random(len(xs))) / (W - 1)).astype(np.float32)
v = (((H - ys) + rng.random(len(xs))) / (H - 1)).astype(np.float32)
o = np.array(cam.origin[:], np.float32)
ll, hz, vt = (np.array(a[:], np.float32) for a in (cam.lleft, cam.horizontal, cam.vertical))
gen = [np.zeros(len(xs), abi.RAY_DTYPE)]
gen[0]["o"] = o
gen[0]["d"] = (ll[None] + u[:, None] * hz[None] + v[:, None] * vt[None] - o[None]).astype(np.float32)
gen[0]["tMin"], gen[0]["tMax"] = 0.001, np.inf
for g in range(2):
    hits = ctx.trace(gen[-1])
    m = hits["prim"] >= 0
    nxt = np.zeros(int(m.sum()), abi.RAY_DTYPE)
    nxt["o"] = hits["p"][m]
    rnd = rng.normal(size=(len(nxt), 3)).astype(np.float32)
    rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
    d = hits["normal"][m] + rnd
    nxt["d"] = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    nxt["tMin"], nxt["tMax"] = 0.001, np.inf
    gen.append(nxt)
rays = np.concatenate(gen)
print("rays: primaries %d, bounce1 %d, bounce2 %d" % tuple(len(g) for g in gen))
want = ctx.trace(rays)
os.environ["SRT_TB_PROFILE"] = "1"
os.environ["SRT_TB_BLOCKS"], os.environ["SRT_FETCH_MIN"] = "8", "16"
ctx.traverse_bench(rays, reps=256)
os.environ["SRT_TB_PROFILE"] = "0"
for blocks in (5, 8):
    os.environ["SRT_TB_BLOCKS"] = str(blocks)
    for fetch_min in (16, 24, 32):
        os.environ["SRT_FETCH_MIN"] = str(fetch_min)
        ms, t, ref = ctx.traverse_bench(rays, reps=256)
        ok = np.array_equal(t[want["prim"] >= 0].view(np.uint32), want["t"][want["prim"] >= 0].view(np.uint32))
        print("blocks/CU %d fetchMin %2d: %.3f ms for %d rays -> %.2f Grays/s  (t bit-exact vs trace: %s)" % (
            blocks, fetch_min, ms, 256 * len(rays), 256 * len(rays) / ms / 1e6, ok), flush=True)
print("node visits/ray of this mix: %.1f, prim tests/ray %.2f" % (want["nodeVisits"].mean(), (want["triTests"] + want["sphereTests"]).mean()))
