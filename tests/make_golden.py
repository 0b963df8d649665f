"""Generates the fixtures under tests/golden/.  Run in the build container (needs
/root/reference for the published images); the GPU box only ever sees the outputs.

  published_blocks.npz    the same two renders as 4x4-pixel block means (fixed point, 1/64 level) + a per-pixel
                          class map (what a pinhole primary ray hits: sky / mesh / ground / light / iron /
                          metal, traced by the oracle) + the coordinates of their pure-black pixels: ~40 000
                          numbers per image for the pooled-residual parity checks instead of five.
  published_regions.json  region means of the reference's two published renders
                          (images/test-1kx240p.png, images/test-5kx720p.png): the only
                          outputs the reference ships; statistical parity anchors.
  rng_kat.json            libstdc++ mt19937 known answers.
  trace_<scene>.npz       fixed ray set -> oracle hit records (prim, t bits, ...).
  bvh_topology.json       pre-order node arrays of the config scenes (count, depth, sha1, first nodes).
  render_<scene>.npz      tiny full renders by the oracle in COUNTER and MT mode
                          (float accumulators + RGBA8).
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.oracle_py as O  # noqa: E402

srt = importlib.import_module("sexy-raytracer_amd")
abi = srt.abi
GOLD = os.path.join(ROOT, "tests", "golden")

# rows y0:y1, cols x0:x1 of the 426x240 image; scaled x3 for 720p
REGIONS = {"sky": (5, 60, 5, 150), "ground": (200, 238, 150, 300), "far_ground": (130, 145, 20, 120),
           "chief": (60, 150, 195, 230), "metal_sphere": (150, 190, 290, 340), "iron_sphere": (150, 195, 85, 135)}


def fixed_rays(cam, n_side_x=64, n_side_y=36, seed=3):
    """pinhole-ish primaries on a grid (through the thin lens centre) + seeded secondaries."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:n_side_y, 0:n_side_x]
    u = ((xs + rng.random(xs.shape)) / (n_side_x - 1)).astype(np.float32).ravel()
    v = (((n_side_y - ys) + rng.random(xs.shape)) / (n_side_y - 1)).astype(np.float32).ravel()
    o = np.array(cam.origin[:], np.float32)
    ll, hz, vt = (np.array(a[:], np.float32) for a in (cam.lleft, cam.horizontal, cam.vertical))
    d = (ll[None] + u[:, None] * hz[None] + v[:, None] * vt[None] - o[None]).astype(np.float32)
    rays = np.zeros(len(u), abi.RAY_DTYPE)
    rays["o"], rays["d"] = o, d
    rays["time"] = rng.random(len(u)).astype(np.float32)
    rays["tMin"], rays["tMax"] = 0.001, np.inf
    return rays


def secondary_rays(hits, primaries, seed=5):
    rng = np.random.default_rng(seed)
    m = hits["prim"] >= 0
    sec = np.zeros(int(m.sum()), abi.RAY_DTYPE)
    sec["o"] = hits["p"][m]
    sec["d"] = rng.normal(size=(len(sec), 3)).astype(np.float32)
    sec["time"] = primaries["time"][m]
    sec["tMin"], sec["tMax"] = 0.001, np.inf
    return sec


def primary_classes(W, H):
    """Per pixel: what a pinhole ray through the pixel centre hits in the main.cpp scene
    (0 sky, 1 mesh, 2 ground, 3 light sphere, 4 iron sphere, 5 metal sphere)."""
    sb = srt.scenes.scene_masterchief()
    osc = O.OracleScene(sb)
    cam = O.make_camera(abi.default_camera_params())
    ys, xs = np.mgrid[0:H, 0:W]
    u = ((xs + 0.5) / (W - 1)).astype(np.float32).ravel()            # main.cpp:210
    v = (((H - ys) + 0.5) / (H - 1)).astype(np.float32).ravel()      # main.cpp:211
    o = np.array(cam.origin[:], np.float32)
    ll, hz, vt = (np.array(a[:], np.float32) for a in (cam.lleft, cam.horizontal, cam.vertical))
    rays = np.zeros(W * H, abi.RAY_DTYPE)
    rays["o"] = o
    rays["d"] = (ll[None] + u[:, None] * hz[None] + v[:, None] * vt[None] - o[None]).astype(np.float32)
    rays["tMin"], rays["tMax"] = 0.001, np.inf
    prim = osc.trace(rays)["prim"]
    ntri = int(sb.desc().numTriangles)
    cls = np.zeros(W * H, np.uint8)
    cls[(prim >= 0) & (prim < ntri)] = 1
    for k, c in enumerate((2, 3, 4, 5)):  # scene_masterchief appends ground, light, iron, metal after the mesh
        cls[prim == ntri + k] = c
    return cls.reshape(H, W)


def main():
    os.makedirs(GOLD, exist_ok=True)
    from PIL import Image
    ref = "/root/reference/images"
    if os.path.isdir(ref):
        blocks = {}
        for key, name in (("240p", "test-1kx240p.png"), ("720p", "test-5kx720p.png")):
            im = np.asarray(Image.open(os.path.join(ref, name)).convert("RGB")).astype(np.float64)
            H, W = im.shape[:2]
            hb, wb = H // 4, W // 4
            b = im[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4, 3).mean((1, 3))
            blocks[key + "_block4_x64"] = np.round(b * 64).astype(np.uint16)
            # the same blocks in linear radiance, ((q + 0.5) / 256)^2 per pixel (color.h:33-40 inverted at the
            # centre of the quantisation step), for comparisons against low-spp renders pooled before tone mapping
            lin = (((im + 0.5) / 256.0) ** 2)[:hb * 4, :wb * 4].reshape(hb, 4, wb, 4, 3).mean((1, 3))
            blocks[key + "_block4_linear_x65535"] = np.round(lin * 65535).astype(np.uint16)
            blocks[key + "_class"] = primary_classes(W, H)
            ys, xs = np.nonzero(im.sum(2) == 0)
            blocks[key + "_black_yx"] = np.stack([ys, xs], 1).astype(np.int16)
        np.savez_compressed(os.path.join(GOLD, "published_blocks.npz"), **blocks)
        out = {}
        for name, scale in (("test-1kx240p.png", 1), ("test-5kx720p.png", 3)):
            im = np.asarray(Image.open(os.path.join(ref, name)).convert("RGB")).astype(np.float64)
            out[name] = {"size": [im.shape[1], im.shape[0]],
                         "regions": {k: {"rows_cols": [y0 * scale, y1 * scale, x0 * scale, x1 * scale],
                                         "mean_rgb": im[y0 * scale:y1 * scale, x0 * scale:x1 * scale].mean((0, 1)).round(3).tolist()}
                                     for k, (y0, y1, x0, x1) in REGIONS.items()},
                         "black_pixels": int((im.sum(2) == 0).sum())}
        json.dump(out, open(os.path.join(GOLD, "published_regions.json"), "w"), indent=1)

    kat = {"mt19937_default_seed_uniform_float": [float(x) for x in O.rng_kat(16, libstdcxx=True)],
           "survey_first_six": [0.81472367, 0.135477006, 0.905791938, 0.835008562, 0.126986817, 0.968867779],
           "survey_randomVec3f_m1_1": [0.811584, -0.729046, 0.629447],
           "counter_seed1_pixel2_sample3": [float(x) for x in O.rng_counter(1, 2, 3, 8)]}
    json.dump(kat, open(os.path.join(GOLD, "rng_kat.json"), "w"), indent=1)

    cam = O.make_camera(abi.default_camera_params())
    for name in ("spheres", "iron", "masterchief"):
        sb = srt.scenes.SCENES[name]()
        osc = O.OracleScene(sb)
        rays = fixed_rays(cam)
        h1 = osc.trace(rays)
        sec = secondary_rays(h1, rays)
        h2 = osc.trace(sec)
        allrays = np.concatenate([rays, sec])
        hits = np.concatenate([h1, h2])
        closest = osc.trace(allrays, abi.SRT_TRAVERSE_CLOSEST)
        np.savez_compressed(os.path.join(GOLD, "trace_%s.npz" % name), rays=allrays, hits=hits,
                            closest_prim=closest["prim"], closest_t=closest["t"])
        W, H, spp, mb = 64, 36, 4, (8 if name == "spheres" else 4)
        p = abi.default_render_params(W, H, spp, mb, seed=7)
        acc_c, rgba_c, st_c = osc.render(cam, p, O.RNG_COUNTER, threads=4)
        osc2 = O.OracleScene(sb)  # fresh generator: BVH build then render, like a new process
        acc_m, rgba_m, st_m = osc2.render(cam, p, O.RNG_MT, threads=1)
        np.savez_compressed(os.path.join(GOLD, "render_%s.npz" % name), width=W, height=H, spp=spp, max_bounce=mb,
                            seed=7, accum_counter=acc_c, rgba_counter=rgba_c, accum_mt=acc_m, rgba_mt=rgba_m,
                            stats_counter=json.dumps(st_c), stats_mt=json.dumps(st_m))
        nodes, depth = osc.bvh(0)
        import hashlib
        topo = json.load(open(os.path.join(GOLD, "bvh_topology.json"))) if os.path.exists(os.path.join(GOLD, "bvh_topology.json")) else {}
        topo[name] = {"nodes": len(nodes), "depth_root1": depth, "sha1_nodes": hashlib.sha1(nodes.tobytes()).hexdigest(),
                      "first_nodes": [{"left": int(n["left"]), "right": int(n["right"]), "bmin": [float(x) for x in n["bmin"]],
                                       "bmax": [float(x) for x in n["bmax"]]} for n in nodes[:4]]}
        json.dump(topo, open(os.path.join(GOLD, "bvh_topology.json"), "w"), indent=1)
        print(name, "rays", len(allrays), "hits", int((hits["prim"] >= 0).sum()),
              "faithful!=closest", int((hits["prim"] != closest["prim"]).sum() + ((hits["prim"] == closest["prim"]) & (hits["t"] != closest["t"])).sum()))


if __name__ == "__main__":
    main()
