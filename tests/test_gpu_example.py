"""End-to-end drop-in check on the GPU: the main.cpp-shaped C++ example (examples/main.cpp, built on
the host layer srt/*.h: gltfLoad + PNG decode + bvhNode + hipDevice + PNG write) renders the
main.cpp scene and must produce the same RGBA bytes as the Python scene path through the same
C-ABI library."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_example_matches_python_path(tmp_path, ctx, abi, srt, camera):
    from PIL import Image
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    data = tmp_path / "data"
    data.mkdir()
    for f in ("masterchief2-separate-xf.gltf", "masterchief2-separate-xf.bin", "Image_0.png", "Image_1.png"):
        shutil.copy(os.path.join(ROOT, "assets", f), data / f)
    a, n, m, r = srt.scenes.iron_textures()
    Image.fromarray(a).save(data / "rustediron2_basecolor-2x1.png")
    Image.fromarray(n).save(data / "rustediron2_normal-2x1.png")
    Image.fromarray(m[..., 0]).save(data / "rustediron2_metallic-2x1.png")
    Image.fromarray(r[..., 0]).save(data / "rustediron2_roughness-2x1.png")
    out = tmp_path / "test.png"
    env = dict(os.environ, SRT_DATA_DIR=str(data))
    subprocess.check_call([os.path.join(ROOT, "examples", "srt_main"), "--gltf", str(data / "masterchief2-separate-xf.gltf"),
                           "--height", "240", "--spp", "8", "--bounces", "4", "--chunks", "1", "--out", str(out)], env=env)
    got = np.asarray(Image.open(out).convert("RGBA"))
    ctx.upload_scene(srt.scenes.scene_masterchief())
    ctx.set_camera(camera)
    _, want = ctx.render_image(abi.default_render_params(426, 240, 8, 4, seed=1, spp_chunks=1))
    assert got.shape == want.shape == (240, 426, 4)
    assert np.array_equal(got, want)


def test_cpp_example_renders_a_data_uri_model(tmp_path, ctx, abi, srt, camera):
    """main.cpp:60-71, the square.gltf branch: a glTF whose buffer is a base64 `data:` URI (what cgltf_load_buffers
    decodes) goes through gltfLoad of the C++ host layer into the same scene and renders the same bytes as the
    Python path.  The file is written here with square.gltf's content shape (four vertices, two triangles, one
    untextured pbr material); /root/reference is not on the GPU box."""
    import base64
    import json
    from PIL import Image
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    data = tmp_path / "data"
    data.mkdir()
    pos = np.array([[-1, -1, 1], [1, -1, 1], [-1, 1, 1], [1, 1, 1]], "<f4")
    nrm = np.array([[0, 0, 1]] * 4, "<f4")
    uv = np.array([[0, 1], [1, 1], [0, 0], [1, 0]], "<f4")
    idx = np.array([0, 1, 3, 0, 3, 2], "<u2")
    raw = pos.tobytes() + nrm.tobytes() + uv.tobytes() + idx.tobytes()
    g = {"asset": {"version": "2.0"},
         "buffers": [{"byteLength": len(raw), "uri": "data:application/octet-stream;base64," + base64.b64encode(raw).decode()}],
         "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                         {"buffer": 0, "byteOffset": 96, "byteLength": 32}, {"buffer": 0, "byteOffset": 128, "byteLength": 12}],
         "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                       {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                       {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"},
                       {"bufferView": 3, "componentType": 5123, "count": 6, "type": "SCALAR"}],
         "materials": [{"doubleSided": True, "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1], "metallicFactor": 0,
                                                                       "roughnessFactor": 0.4}}],
         "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}]}
    path = data / "square.gltf"
    json.dump(g, open(path, "w"))
    a, n, m, r = srt.scenes.iron_textures()
    Image.fromarray(a).save(data / "rustediron2_basecolor-2x1.png")
    Image.fromarray(n).save(data / "rustediron2_normal-2x1.png")
    Image.fromarray(m[..., 0]).save(data / "rustediron2_metallic-2x1.png")
    Image.fromarray(r[..., 0]).save(data / "rustediron2_roughness-2x1.png")
    out = tmp_path / "square.png"
    subprocess.check_call([os.path.join(ROOT, "examples", "srt_main"), "--gltf", str(path), "--height", "144", "--spp", "8",
                           "--bounces", "4", "--chunks", "1", "--out", str(out)], env=dict(os.environ, SRT_DATA_DIR=str(data)))
    got = np.asarray(Image.open(out).convert("RGBA"))
    ctx.upload_scene(srt.scenes.scene_masterchief(gltf_path=str(path)))
    ctx.set_camera(camera)
    _, want = ctx.render_image(abi.default_render_params(256, 144, 8, 4, seed=1, spp_chunks=1))
    assert got.shape == want.shape == (144, 256, 4)
    assert np.array_equal(got, want)
    # the square is in the picture: rays through the image centre hit it in front of the sky
    assert len(ctx.bvh(0)) >= 5


def test_cpp_world_hit_on_the_host_classes(ctx, abi, srt):
    """Scene code that calls world.hit(r, tMin, tMax, rec) (hittable.h:26) compiles against the host mirror
    and gets the device's answer: examples/hit_probe.cpp builds the three-sphere scene with the reference's
    class vocabulary and calls hittableList::hit ray by ray; t, p, normal, frontFace and the material pointer
    must be what srtTraceRays returns for the same scene built through the Python path."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    out = subprocess.check_output([os.path.join(ROOT, "examples", "srt_hit_probe")]).decode().split("\n")
    rays = np.zeros(35, abi.RAY_DTYPE)
    k = 0
    for j in range(5):
        for i in range(7):
            rays[k]["o"] = (0.0, 3.0, 5.0)
            rays[k]["d"] = (-6.0 + 2.0 * i, -3.5 + 1.0 * j, -5.0)
            k += 1
    rays["time"], rays["tMin"], rays["tMax"] = 0.25, 0.001, np.inf
    ctx.upload_scene(srt.scenes.scene_spheres())
    want = ctx.trace(rays)
    mat_of_prim = {0: 0, 1: 1, 2: 2, 3: 3}  # scene_spheres adds ground, diffuse, glass, mirror in this order
    assert (want["prim"] >= 0).sum() >= 20 and (want["prim"] < 0).sum() >= 3
    for k in range(35):
        f = out[k].split()
        if want["prim"][k] < 0:
            assert f[0] == "miss", (k, out[k])
            continue
        assert f[0] == "hit", (k, out[k])
        vals = np.array([float.fromhex(x) for x in f[1:8]], np.float32)
        ref = np.concatenate([[want["t"][k]], want["p"][k], want["normal"][k]]).astype(np.float32)
        assert np.array_equal(vals.view(np.uint32), ref.view(np.uint32)), (k, vals, ref)
        assert int(f[8]) == int(want["frontFace"][k]) and int(f[9]) == mat_of_prim[int(want["prim"][k])]
    assert out[35] == "ball hit" and float.fromhex(out[36].split()[1]) == 2.0

    # ---- material::scatter / material::emitted / texture::value on the host classes (material.h:15-21, texture.h:13-16):
    # the same program went on to call rec.matPtr->scatter(r, rec, attenuation, scattered) for every hit of the grid.
    # Each material answers from a device context of its own and keys the counter RNG by its own call count; the same
    # (ray, hit record, key) through srtScatterRays on the Python-built scene must give the same bits.
    lines = out[37:]
    scat = [l.split() for l in lines if l.startswith("scatter ")]
    hit_idx = [k for k in range(35) if want["prim"][k] >= 0]
    assert len(scat) == len(hit_idx) >= 20
    calls = {}
    for f, k in zip(scat, hit_idx):
        m = int(want["material"][k])
        key = calls.get(m, 0)
        calls[m] = key + 1
        ref = ctx.scatter_test(rays[k:k + 1], want[k:k + 1], key)[0]
        got = np.array([float.fromhex(x) for x in f[2:12]], np.float32)
        assert int(f[1]) == int(ref[9]), (k, f, ref)
        assert np.array_equal(got[0:3].view(np.uint32), ref[0:3].view(np.uint32)), (k, "attenuation", got, ref)
        assert np.array_equal(got[3:6].view(np.uint32), ref[3:6].view(np.uint32)), (k, "direction")
        assert np.array_equal(got[6:9].view(np.uint32), ref[6:9].view(np.uint32)) and got[9] == np.float32(0.25)
    assert len(calls) >= 3  # ground, diffuse or glass, mirror were all hit
    em = [float.fromhex(x) for x in next(l for l in lines if l.startswith("emitted ")).split()[1:]]
    assert em == [4.0, 3.0, 2.0, 0.0, 0.0, 0.0]  # diffuseLight::emitted (material.h:144-150) / material::emitted (:18-20)
    # checker::value (texture.h:42-48): sign of sin(10x) sin(10y) sin(10z) picks odd / even, colours times 255
    chk = [[float.fromhex(x) for x in l.split()[1:]] for l in lines if l.startswith("checker ")]
    assert len(chk) == 6
    for k, c in enumerate(chk):
        p = np.float32([0.11 + 0.37 * k, 0.05 - 0.21 * k, 0.4 + 0.13 * k])
        sines = np.sin(np.float32(10.0) * p).prod()
        col = np.float32([0.9, 0.9, 0.9] if sines < 0 else [0.2, 0.3, 0.1]) * np.float32(255.0)
        assert np.array_equal(np.float32(c), col), (k, c, col, sines)
    # imagePNG::value (texture.h:129-148): nearest texel, v flipped, bytes as floats; the 3 x 2 image hit_probe wrote
    img = (10 + 13 * np.arange(18)).astype(np.uint8).reshape(2, 3, 3)
    tex = [[float.fromhex(x) for x in l.split()[1:]] for l in lines if l.startswith("texel ")]
    assert len(tex) == 4
    for (u, v), c in zip(((0.0, 0.0), (0.4, 0.9), (0.99, 0.2), (1.0, 1.0)), tex):
        i, j = min(int(u * 3), 2), min(int((1.0 - v) * 2), 1)
        assert c == [float(x) for x in img[j, i]], (u, v, c, img[j, i])
