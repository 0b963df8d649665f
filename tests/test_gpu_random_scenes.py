"""Randomised scenes on the GPU through the C-ABI against the oracle: every primitive and material
kind mixed, degenerate triangles (zero area, repeated vertices, axis-aligned slivers that get the
+-1e-4 box padding of model.h:199-204), tiny and huge spheres, moving spheres, 1- and 3-byte image
textures, several world roots.  Fixed ray sets must match bit for bit; tiny renders to the usual
render tolerance."""
import numpy as np
import pytest

from conftest import render_counted

pytestmark = pytest.mark.gpu


def random_scene(abi, seed):
    rng = np.random.default_rng(seed)
    sb = abi.SceneBuilder()
    texs = [sb.solid(*rng.uniform(20, 240, 3)), sb.checker(tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3))),
            sb.image(rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3), dtype=np.uint8), 3),
            sb.image(rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 1), dtype=np.uint8), 1),
            sb.image(None, 3)]
    mats = [sb.pbr(albedo_tex=texs[0], metalness=float(rng.uniform()), roughness=float(rng.uniform(0.05, 1))),
            sb.pbr(albedo_tex=texs[1]),
            sb.pbr(albedo_tex=texs[2], normal_tex=texs[2], metallic_tex=texs[3], roughness_tex=texs[3],
                   albedo=tuple(rng.uniform(0.2, 1, 4))),
            sb.pbr(albedo_tex=texs[4], roughness=0.3),
            sb.pbr(albedo_tex=-1, albedo=(0.8, 0.5, 0.3, 1.0), metalness=0.5, roughness=0.5),
            sb.metal(tuple(rng.uniform(0.3, 1, 3)), float(rng.uniform(0, 1.5))),
            sb.dielectric(float(rng.uniform(1.1, 2.4))),
            sb.light(tuple(rng.uniform(1, 20, 3)))]
    n_tri = int(rng.integers(1, 60))
    pos = (rng.uniform(-3, 3, (n_tri * 3, 3)) + np.array([0, 3, -1])).astype(np.float32)
    # degenerate ones
    for k in range(0, n_tri, 7):
        pos[3 * k + 2] = pos[3 * k + 1]                       # repeated vertex: zero area
    for k in range(3, n_tri, 11):
        pos[3 * k:3 * k + 3, int(rng.integers(0, 3))] = 1.25  # axis-aligned: flat box axis gets padded
    uv = rng.uniform(-0.2, 1.2, (n_tri * 3, 2)).astype(np.float32)
    idx = np.arange(n_tri * 3).reshape(-1, 3)
    sb.add_triangles(pos, uv, idx, mats[int(rng.integers(0, 5))])
    first_sphere = sb.num_prims
    for _ in range(int(rng.integers(1, 12))):
        c = rng.uniform(-3, 3, 3) + np.array([0, 3, -1])
        r = float(rng.choice([0.0, 1e-3, 0.3, 0.8, 50.0], p=[0.05, 0.1, 0.5, 0.3, 0.05]))
        moving = rng.uniform() < 0.4
        sb.add_sphere(tuple(c), r, mats[int(rng.integers(0, len(mats)))],
                      center1=tuple(c + rng.uniform(-0.5, 0.5, 3)) if moving else None, time0=0.0, time1=1.0)
    sb.add_sphere((0.0, -1000.0, 0.0), 1000.0, mats[1])
    layout = int(rng.integers(0, 3))
    if layout == 0:
        sb.world_bvh(0, None, 0.0, 1.0)
    elif layout == 1:  # two roots + a bare primitive
        sb.world_bvh(0, first_sphere, 0.0, 1.0)
        sb.world_bvh(first_sphere, sb.num_prims - first_sphere - 1, 0.0, 1.0)
        sb.world_prim(sb.num_prims - 1)
    else:              # plain list
        for i in range(sb.num_prims):
            sb.world_prim(i)
    return sb


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_trace_and_render(ctx, oracle, abi, camera, seed, node_path):
    sb = random_scene(abi, 1000 + seed)
    ctx.upload_scene(sb)
    ctx.set_camera(camera)
    osc = oracle.OracleScene(sb)
    rng = np.random.default_rng(seed)
    rays = np.zeros(3000, abi.RAY_DTYPE)
    rays["o"] = (rng.uniform(-4, 4, (len(rays), 3)) + np.array([0, 3, 2])).astype(np.float32)
    rays["d"] = rng.normal(size=(len(rays), 3)).astype(np.float32)
    rays["d"][::50, int(rng.integers(0, 3))] = 0.0  # axis-parallel rays: division by zero in the slab test
    rays["time"] = rng.random(len(rays)).astype(np.float32)
    rays["tMin"], rays["tMax"] = 0.001, np.inf
    for trav in (abi.SRT_TRAVERSE_FAITHFUL, abi.SRT_TRAVERSE_CLOSEST):
        want, got = osc.trace(rays, trav), ctx.trace(rays, trav)
        assert np.array_equal(got["prim"] >= 0, want["prim"] >= 0)
        m = want["prim"] >= 0
        assert np.array_equal(got["t"][m].view(np.uint32), want["t"][m].view(np.uint32))
        same = got["prim"] == want["prim"]
        if trav == abi.SRT_TRAVERSE_FAITHFUL:
            assert same.all()
            for f in ("p", "normal", "frontFace", "material", "nodeVisits", "boxPasses", "triTests", "sphereTests"):
                a, b = got[f][m], want[f][m]
                assert np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, b.view(np.uint32) if b.dtype.kind == "f" else b), f
        else:
            assert (~same).mean() < 5e-3  # exact ties between coincident primitives only
    p = abi.default_render_params(48, 27, 3, 5, seed=seed, count_stats=1)
    acc, rgba = render_counted(ctx, p, node_path)
    want, want_rgba, want_st = osc.render(camera, p, oracle.RNG_COUNTER, threads=4)
    assert np.array_equal(np.isnan(acc), np.isnan(want))
    bit = (acc.view(np.uint32) == want.view(np.uint32)).all(axis=-1)
    assert bit.mean() >= 0.98, bit.mean()
    ok = np.isnan(want) | (np.abs(acc - want) <= 0.05 * np.maximum(np.abs(want), 1e-3) + 0.05)
    assert ok.mean() >= 0.999
    st = ctx.stats()
    for k in ("samples", "rays"):
        assert abs(st[k] - want_st[k]) <= max(2, 1e-3 * want_st[k]), k
