"""Pins the CPU oracle (oracle/oracle.cpp) against everything the reference offers for this
path: libstdc++ RNG known answers, the BVH statistics and traversal counters recorded in
SURVEY.md A.5, and region means of the two published renders (tests/golden/published_regions.json,
made by tests/make_golden.py from /root/reference/images).  The reference ships no tests."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD


def test_rng_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    got = oracle.rng_kat(16)
    # closed form float(u32)/2^32 == libstdc++'s uniform_real_distribution<float> over mt19937
    assert got.tolist() == oracle.rng_kat(16, libstdcxx=True).tolist()
    assert got.tolist() == kat["mt19937_default_seed_uniform_float"]
    np.testing.assert_allclose(got[:6], kat["survey_first_six"], rtol=0, atol=5e-9)
    big = oracle.rng_kat(200000)
    assert big.tobytes() == oracle.rng_kat(200000, libstdcxx=True).tobytes()
    assert 0.0 <= big.min() and big.max() < 1.0


def test_random_vec3_gcc_argument_order(oracle):
    v = np.zeros(3, np.float32)
    oracle.lib().orc_random_vec3_kat(v.ctypes.data)
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    np.testing.assert_allclose(v, kat["survey_randomVec3f_m1_1"], atol=1e-6)  # z drawn first


def test_counter_rng_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    assert oracle.rng_counter(1, 2, 3, 8).tolist() == kat["counter_seed1_pixel2_sample3"]
    a, b = oracle.rng_counter(1, 2, 3, 64), oracle.rng_counter(1, 2, 4, 64)
    assert not np.array_equal(a, b)


def test_bvh_anchors_main_scene(oracle, srt):
    """3046 primitives -> 4043 bvhNodes, 998 single-object leaves,
    1024 two-object leaves, depth 11 (root = 0), first randomInt(0,2) of a fresh process = 2."""
    sb = srt.scenes.scene_masterchief()
    assert sb.num_prims == 3046
    sc = oracle.OracleScene(sb)
    nodes, depth = sc.bvh(0)
    assert len(nodes) == 4043 and sc.build_draws() == 4043
    single = nodes["left"] == nodes["right"]
    double = (nodes["left"] < 0) & (nodes["right"] < 0) & ~single
    assert single.sum() == 998 and double.sum() == 1024
    assert depth - 1 == 11
    assert int(3.0 * oracle.rng_kat(1)[0]) == 2
    # every primitive is referenced by exactly one leaf slot pair
    refs = np.concatenate([nodes["left"][nodes["left"] < 0], nodes["right"][(nodes["right"] < 0) & ~single]])
    assert sorted((~refs).tolist()) == list(range(3046))


def test_traversal_counters_main_scene(oracle, srt, abi):
    """SURVEY.md 3.3/A.5 (240p, 4 bounces): 1.74 rays/sample, 55.4 node visits/ray,
    29.6 box passes/ray, 4.74 primitive tests/ray."""
    sb = srt.scenes.scene_masterchief()
    sc = oracle.OracleScene(sb)
    cam = oracle.make_camera(abi.default_camera_params())
    p = abi.default_render_params(426, 240, 2, 4, seed=1)
    _, _, st = sc.render(cam, p, oracle.RNG_COUNTER, threads=8)
    rays = st["rays"]
    assert abs(rays / st["samples"] - 1.74) < 0.02
    assert abs(st["nodeVisits"] / rays - 55.4) < 0.6
    assert abs(st["boxPasses"] / rays - 29.6) < 0.4
    assert abs((st["triCalls"] + st["sphereCalls"]) / rays - 4.74) < 0.08


def test_faithful_vs_closest_f4(oracle, srt, abi):
    """triangle::hit never tests tMax, so reference traversal returns a farther
    triangle on a tiny fraction of rays; never a hit/miss disagreement."""
    g = np.load(os.path.join(GOLD, "trace_masterchief.npz"))
    hits, cp, ct = g["hits"], g["closest_prim"], g["closest_t"]
    assert ((hits["prim"] >= 0) == (cp >= 0)).all()
    differ = (hits["prim"] != cp) | ((hits["prim"] >= 0) & (hits["t"] != ct))
    assert 0 < differ.sum() < 0.005 * len(hits)
    assert (hits["t"][differ] >= ct[differ]).all()  # always farther


def test_published_image_region_means(oracle, srt, abi):
    """Statistical pin: 426x240 render of the main.cpp scene vs images/test-1kx240p.png.
    Excludes the iron sphere (its four texture files are missing blobs in the reference; the
    scene uses seeded procedural stand-ins)."""
    pub = json.load(open(os.path.join(GOLD, "published_regions.json")))["test-1kx240p.png"]
    sb = srt.scenes.scene_masterchief()
    sc = oracle.OracleScene(sb)
    cam = oracle.make_camera(abi.default_camera_params())
    p = abi.default_render_params(426, 240, 160, 4, seed=1)
    _, rgba, _ = sc.render(cam, p, oracle.RNG_COUNTER, threads=8, want_stats=False)
    img = rgba[..., :3].astype(np.float64)
    tol = {"sky": 0.51, "far_ground": 3.0, "metal_sphere": 3.0, "ground": 6.0, "chief": 8.0}
    for name, t in tol.items():
        y0, y1, x0, x1 = pub["regions"][name]["rows_cols"]
        mean = img[y0:y1, x0:x1].mean((0, 1))
        assert np.abs(mean - pub["regions"][name]["mean_rgb"]).max() <= t, (name, mean, pub["regions"][name]["mean_rgb"])
    # sky is exact: background (0.53,0.81,0.92) -> sqrt gamma -> 186,230,245
    assert (rgba[5:60, 5:150, :3] == np.array([186, 230, 245])).all()


def test_published_image_class_biases(oracle, srt, abi):
    """The oracle against the published 240p render block by block (tests/published.py): summed linear
    radiance of every 4x4 block lying wholly on the mesh / the ground / the sky vs the published image's
    linearised blocks (~2 500 ground, ~350 mesh and ~2 700 sky blocks).  64 spp: the statistical error of
    a class total is 1-2 % (the light sphere's rare 250x samples dominate it); a wrong uv rule, v flip,
    normal-map axis or Fresnel constant moves a class by 10 % and more.  The GPU twin at the published
    sample counts is test_gpu_properties.py::test_headline_frame_pooled_residuals_vs_published."""
    import published
    sb = srt.scenes.scene_masterchief()
    cam = oracle.make_camera(abi.default_camera_params())
    p = abi.default_render_params(426, 240, 64, 4, seed=1)
    acc, _, _ = oracle.OracleScene(sb).render(cam, p, oracle.RNG_COUNTER, threads=8, want_rgba=False)
    bias = published.class_bias_linear(acc, 64, "240p")
    assert abs(bias["sky"]) <= 0.002, bias      # the background colour itself (quantisation only)
    assert abs(bias["ground"]) <= 0.03, bias
    assert abs(bias["mesh"]) <= 0.05, bias


def test_golden_renders_reproduce(oracle, srt, abi):
    """The committed tiny renders are reproduced bit for bit in both RNG modes (MT mode: a
    fresh generator feeds the BVH build and then the render, like a new process)."""
    cam = oracle.make_camera(abi.default_camera_params())
    for name in ("spheres", "iron", "masterchief"):
        g = np.load(os.path.join(GOLD, "render_%s.npz" % name))
        sb = srt.scenes.SCENES[name]()
        p = abi.default_render_params(int(g["width"]), int(g["height"]), int(g["spp"]), int(g["max_bounce"]), seed=int(g["seed"]))
        acc, rgba, _ = oracle.OracleScene(sb).render(cam, p, oracle.RNG_COUNTER, threads=3)
        assert acc.tobytes() == g["accum_counter"].tobytes() and rgba.tobytes() == g["rgba_counter"].tobytes()
        acc, rgba, st = oracle.OracleScene(sb).render(cam, p, oracle.RNG_MT, threads=1)
        assert acc.tobytes() == g["accum_mt"].tobytes() and rgba.tobytes() == g["rgba_mt"].tobytes()
        assert st["rngDraws"] == json.loads(str(g["stats_mt"]))["rngDraws"]


def test_golden_traces_reproduce(oracle, srt, abi):
    for name in ("spheres", "iron", "masterchief"):
        g = np.load(os.path.join(GOLD, "trace_%s.npz" % name))
        sc = oracle.OracleScene(srt.scenes.SCENES[name]())
        assert sc.trace(g["rays"]).tobytes() == g["hits"].tobytes()
