"""The BASELINE.json configs as concrete scenes, built in the
reference's construction order so BVH sort ties and RNG draws line up with
main.cpp:54-154.  Returns SceneBuilder objects (abi.py)."""
import os

import numpy as np

from . import abi
from .abi import SceneBuilder
from .gltf import load_gltf

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")

_png_cache = {}


def _load_png(path, bpp):
    """stbi_load(path, ..., req_comp=bpp) (texture.h:115): 8-bit RGB or L bytes."""
    from PIL import Image
    key = (path, bpp)
    if key not in _png_cache:
        im = Image.open(path).convert("RGB" if bpp == 3 else "L")
        a = np.asarray(im, dtype=np.uint8)
        _png_cache[key] = a.reshape(a.shape[0], a.shape[1], bpp)
    return _png_cache[key]


def _ground(sb):
    # main.cpp:89-90; pbr ctor material.h:29-32 leaves metalness/roughness
    # uninitialised: defined as 0, 0.
    chk = sb.checker((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    return sb.add_sphere((0.0, -1000.0, 0.0), 1000.0, sb.pbr(albedo_tex=chk, metalness=0.0, roughness=0.0))


def _smooth_noise(rng, h, w, octaves):
    """Deterministic tileable value noise in [0,1]."""
    out = np.zeros((h, w), np.float64)
    amp, total = 1.0, 0.0
    for o in octaves:
        g = rng.random((max(2, h // o), max(2, w // o)))
        ys = (np.arange(h) / o) % g.shape[0]
        xs = (np.arange(w) / o) % g.shape[1]
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        y1, x1 = (y0 + 1) % g.shape[0], (x0 + 1) % g.shape[1]
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
        v = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x1] * (1 - fy) * fx
             + g[y1][:, x0] * fy * (1 - fx) + g[y1][:, x1] * fy * fx)
        out += amp * v
        total += amp
        amp *= 0.5
    return out / total


def iron_textures(seed=2, w=1024, h=512):
    """Seeded procedural stand-ins for rustediron2_{basecolor,normal,metallic,roughness}-2x1.png
    (main.cpp:133-136), which are missing blobs in the reference (.MISSING_LARGE_BLOBS:1-6).
    Returns (albedo RGB, normal RGB, metallic L, roughness L) uint8 arrays."""
    rng = np.random.default_rng(seed)
    rust = _smooth_noise(rng, h, w, (64, 32, 16, 8, 4))
    fine = _smooth_noise(rng, h, w, (8, 4, 2))
    mask = np.clip((rust - 0.48) * 6.0, 0.0, 1.0)  # 1 = rust, 0 = bare metal
    metal_col = np.array([0.56, 0.57, 0.58])
    rust_col = np.array([0.45, 0.22, 0.12])
    alb = (metal_col[None, None, :] * (1 - mask[..., None]) + rust_col[None, None, :] * mask[..., None])
    alb = alb * (0.75 + 0.5 * fine[..., None])
    albedo = np.clip(alb * 255.0, 0, 255).astype(np.uint8)
    hgt = rust * 0.7 + fine * 0.3
    dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
    dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
    n = np.stack([-dx * 24.0, -dy * 24.0, np.ones_like(dx)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    normal = np.clip(n * 127.5 + 127.5, 0, 255).astype(np.uint8)
    metallic = np.clip((1 - mask) * 255.0, 0, 255).astype(np.uint8)[..., None]
    roughness = np.clip((0.25 + 0.6 * mask + 0.15 * fine) * 255.0, 0, 255).astype(np.uint8)[..., None]
    return albedo, normal, metallic, roughness


def _iron_sphere(sb, seed=2):
    # main.cpp:133-141: pbr ctor material.h:47-52 (albedo factor 1,1,1,1; m=r=0 unused: maps present)
    a, n, m, r = iron_textures(seed)
    mat = sb.pbr(albedo_tex=sb.image(a, 3), normal_tex=sb.image(n, 3), metallic_tex=sb.image(m, 1),
                 roughness_tex=sb.image(r, 1), albedo=(1.0, 1.0, 1.0, 1.0), metalness=0.0, roughness=0.0)
    return sb.add_sphere((-3.0, 1.0, 0.0), 1.0, mat)


def scene_spheres():
    """Configs C1/C2: ground + three unit spheres (diffuse pbr / dielectric 1.5 / metal fuzz 0),
    positions and materials from main.cpp:90,124-125,140-144; wrapped in one bvhNode (main.cpp:146)."""
    sb = SceneBuilder()
    _ground(sb)
    sb.add_sphere((-3.0, 1.0, 0.0), 1.0, sb.pbr(albedo_tex=sb.solid(0.4 * 255, 0.2 * 255, 0.1 * 255),
                                                  metalness=0.0, roughness=0.0))
    sb.add_sphere((0.0, 1.0, 0.0), 1.0, sb.dielectric(1.5))
    sb.add_sphere((3.0, 1.0, 0.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0))
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def scene_iron():
    """Config C3: ground + PBR textured sphere (normal map + metallic/roughness maps)."""
    sb = SceneBuilder()
    _ground(sb)
    _iron_sphere(sb)
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def _try_png(path, bpp):
    """imagePNG(path, bpp): the pixels, or None for a failed load (texture.h:117-120) -- e.g. the `data:` URI of an
    embedded image, which gltfLoad hands to stbi_load as a file name (model.h:395-412)."""
    try:
        return _load_png(path, bpp)
    except (OSError, ValueError):
        return None


def add_model(sb, gltf_path):
    """model::create(path)->init() then objects.add(every triangle of every mesh) (main.cpp:60-86).  A triangle's
    material is its mesh's (model.h:108,178); all three maps are imagePNG(..., 3) (model.h:420-437) and the
    metallicRoughness map is stored but never sampled (material.h:190-200), so metallic / roughness come from
    the scalar factors.  Returns the number of triangles added."""
    added = 0
    for mesh in load_gltf(gltf_path):
        if not len(mesh["indices"]):
            continue
        m = mesh["material"]
        if m is None:
            raise ValueError("%s: triangles on a mesh without a material (the reference dereferences a null matPtr)" % gltf_path)
        alb = sb.image(_try_png(m["albedo"], 3), 3) if m["albedo"] else -1
        nrm = sb.image(_try_png(m["normal"], 3), 3) if m["normal"] else -1
        mat = sb.pbr(albedo_tex=alb, normal_tex=nrm, albedo=m["baseColorFactor"],
                     metalness=m["metallicFactor"], roughness=m["roughnessFactor"])
        sb.add_triangles(mesh["positions"], mesh["texcoords"], mesh["indices"], mat)
        added += len(mesh["indices"])
    return added


def add_masterchief(sb, gltf_path=None):
    """main.cpp:74-86 with the file main.cpp:74 names."""
    return add_model(sb, gltf_path or os.path.join(ASSETS, "masterchief2-separate-xf.gltf"))


def scene_masterchief_army(copies=5):
    """A mesh scene between the reference's own (4 043 nodes: a CU's LDS holds its tree) and the big soups: `copies`
    instances of the main.cpp mesh side by side (15 210 triangles at 5: what 16-bit primitive references address), with the ground, the
    light and the metal sphere of the HEAD scene, in one bvhNode.  A tree that is cache-resident, does not fit a CU's LDS
    and still has 16-bit thread links: the 256-thread kernel's regime (bench.py army_720p_1024spp; profiles/r03/sweep_form.txt)."""
    sb = SceneBuilder()
    first = len(sb.triangles)
    add_masterchief(sb)
    base = [t.copy() for t in sb.triangles[first:]]
    for c in range(1, copies):
        dx, dz = 2.6 * ((c + 1) // 2) * (1 if c % 2 else -1), -2.5 * (c % 3)
        for t in base:
            t2 = t.copy()
            t2["p"][..., 0] += np.float32(dx)
            t2["p"][..., 2] += np.float32(dz)
            sb.triangles.append(t2)
            sb._add_prims(abi.SRT_PRIM_TRIANGLE, sb._tri_count, len(t2))
            sb._tri_count += len(t2)
    _ground(sb)
    sb.add_sphere((-7.0, 4.0, 6.0), 1.0, sb.light((250.2, 220.9, 110.2)))
    sb.add_sphere((3.0, 1.0, 3.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0))
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def scene_gltf(gltf_path):
    """The `if (0)` / `else` branches of main.cpp:60-79 (square.gltf, scene.gltf, ...): the model's triangles and
    the checker ground in one bvhNode."""
    sb = SceneBuilder()
    add_model(sb, gltf_path)
    _ground(sb)
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def scene_masterchief(with_spheres=True, gltf_path=None):
    """Config C4/C5: the main.cpp HEAD scene (main.cpp:54-154): masterchief mesh (3042 triangles),
    ground, light sphere, iron sphere, metal sphere -> 3046 prims in one bvhNode.  gltf_path: another model file in
    the mesh's place (the square.gltf / scene.gltf branches of main.cpp:60-79)."""
    sb = SceneBuilder()
    add_masterchief(sb, gltf_path)
    _ground(sb)
    if with_spheres:
        sb.add_sphere((-7.0, 4.0, 6.0), 1.0, sb.light((250.2, 220.9, 110.2)))  # main.cpp:126-127
        _iron_sphere(sb)
        sb.add_sphere((3.0, 1.0, 0.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0))    # main.cpp:143-144
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def scene_soup(num_triangles, seed=7, extent=6.0, size=0.08, with_ground=True, builder=0):
    """Synthetic seeded triangle soup (SURVEY 8d 'Synthetic'): working sets beyond the caches."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    c = (rng.random((num_triangles, 1, 3), dtype=np.float32) - 0.5) * np.float32(2 * extent)
    c[..., 1] = c[..., 1] * 0.5 + np.float32(extent * 0.5)
    v = c + (rng.random((num_triangles, 3, 3), dtype=np.float32) - 0.5) * np.float32(2 * size)
    pos = v.reshape(-1, 3)
    uv = rng.random((num_triangles * 3, 2), dtype=np.float32)
    idx = np.arange(num_triangles * 3, dtype=np.int32).reshape(-1, 3)
    mat = sb.pbr(albedo_tex=sb.solid(0.7 * 255, 0.6 * 255, 0.5 * 255), metalness=0.0, roughness=0.5)
    sb.add_triangles(pos, uv, idx, mat)
    if with_ground:
        _ground(sb)
    sb.world_bvh(0, None, 0.0, 1.0, builder=builder)
    return sb


def scene_sphere_field():
    """The 22x22 field of small spheres that main.cpp:92-122 keeps commented out: moving
    diffuse spheres (sphere.h:47-52), fuzzy metals, glass, placed with the process-global generator
    (globals.h:30-35), plus the ground and the three big spheres of main.cpp:90,124-144.  Draws are taken
    in the order g++ evaluates that code (call arguments right to left); the reference never runs it, so
    the order is unpinned -- the scene's use is as a parity workload for the features no config exercises.
    In the reference's process the bvhNode constructor (main.cpp:146) goes on drawing from the same stream
    after these placements; `global_rng_draws` records how many were taken so that Context.upload_scene and
    the oracle build the tree from that point of the stream, not from a fresh generator."""
    from . import hipdev  # the global generator lives in the C-ABI library (no GPU needed)
    f32 = np.float32
    hipdev.host_random_reset()
    sb = SceneBuilder()
    sb.global_rng_draws = 0  # draws taken from the generator: the bvhNode build must continue after them

    def rf():
        sb.global_rng_draws += 1
        return hipdev.host_random_float()

    _ground(sb)
    glass = sb.dielectric(1.5)
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rf()
            z = f32(b) + f32(0.9) * f32(rf())        # 0.9f * randomFloat()
            x = f32(a + 0.9 * rf())                  # 0.9 * randomFloat() is a double product
            center = (float(x), 0.2, float(z))
            if np.sqrt(f32(x - f32(4.0)) ** 2 + f32(0.0) + z * z) > 0.9:
                if choose < 0.8:
                    cz, cy, cx = (f32(rf()) * f32(rf()) for _ in range(3))
                    mat = sb.pbr(albedo_tex=sb.solid(float(cx) * 255, float(cy) * 255, float(cz) * 255), metalness=0.0, roughness=0.0)
                    up = f32(0.5) * f32(rf())        # randomFloat(0, 0.5f)
                    sb.add_sphere(center, 0.2, mat, center1=(center[0], float(f32(0.2) + up), center[2]), time0=0.0, time1=1.0)
                elif choose < 0.95:
                    az, ay, ax = (f32(0.5) + f32(0.5) * f32(rf()) for _ in range(3))
                    fuzz = f32(0.5) * f32(rf())
                    sb.add_sphere(center, 0.2, sb.metal((float(ax), float(ay), float(az)), float(fuzz)))
                else:
                    sb.add_sphere(center, 0.2, glass)
    sb.add_sphere((-7.0, 4.0, 6.0), 1.0, sb.light((250.2, 220.9, 110.2)))
    sb.add_sphere((-3.0, 1.0, 0.0), 1.0, sb.pbr(albedo_tex=sb.solid(0.4 * 255, 0.2 * 255, 0.1 * 255), metalness=0.0, roughness=0.0))
    sb.add_sphere((0.0, 1.0, 0.0), 1.0, glass)
    sb.add_sphere((3.0, 1.0, 0.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0))
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


SCENES = {"army": scene_masterchief_army, "spheres": scene_spheres, "iron": scene_iron, "masterchief": scene_masterchief, "sphere_field": scene_sphere_field}
