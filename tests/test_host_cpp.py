"""The C++ host layer (sexy-raytracer_amd/host/srt/*.h: the reference's class names over the C-ABI)
against the Python scene path and the oracle, on CPU: glTF loader (gltfLoad semantics), PNG decoder
and writer, flattening, bvhNode built at construction from the process-global generator."""
import hashlib
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

TRI = np.dtype([("p", "<f4", (3, 3)), ("uv", "<f4", (3, 2)), ("material", "<i4")])
SPH = np.dtype([("c0", "<f4", 3), ("c1", "<f4", 3), ("t0", "<f4"), ("t1", "<f4"), ("r", "<f4"), ("material", "<i4")])
TEX = np.dtype([("kind", "<i4"), ("w", "<i4"), ("h", "<i4"), ("bpp", "<i4"), ("off", "<i8"), ("even", "<i4"),
                ("odd", "<i4"), ("color", "<f4", 3), ("pad", "<i4")])


def _read(f, dtype):
    n = struct.unpack("<q", f.read(8))[0]
    return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype)


@pytest.fixture(scope="module")
def dump(tmp_path_factory, dev):
    exe = os.path.join(ROOT, "examples", "srt_flatten_dump")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    d = tmp_path_factory.mktemp("flat")
    out, png = str(d / "flat.bin"), str(d / "roundtrip.png")
    env = dict(os.environ, SRT_DATA_DIR=os.path.join(ROOT, "assets"))
    subprocess.check_call([exe, os.path.join(ROOT, "assets", "masterchief2-separate-xf.gltf"), out, png], env=env)
    with open(out, "rb") as f:
        r = {"tri": _read(f, TRI), "sph": _read(f, SPH), "prims": _read(f, np.dtype(("<i4", 2))).reshape(-1, 2),
             "mat": _read(f, np.dtype(("<i4", 16))), "tex": _read(f, TEX), "texels": _read(f, "u1")}
        from importlib import import_module
        r["nodes"] = _read(f, import_module("sexy-raytracer_amd.abi").NODE_DTYPE)
        r["next"] = _read(f, "<f4")
    r["png"] = png
    return r


def _python_scene(srt):
    sb = srt.abi.SceneBuilder()
    srt.scenes.add_masterchief(sb)
    chk = sb.checker((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    sb.add_sphere((0.0, -1000.0, 0.0), 1000.0, sb.pbr(albedo_tex=chk))
    sb.add_sphere((3.0, 1.0, 0.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0))
    sb.world_bvh(0, None, 0.0, 1.0)
    return sb


def test_cpp_gltf_and_flatten_match_python(dump, srt):
    sb = _python_scene(srt)
    tri = np.concatenate(sb.triangles)
    assert len(dump["tri"]) == 3042 and dump["tri"]["p"].tobytes() == tri["p"].tobytes()
    assert dump["tri"]["uv"].tobytes() == tri["uv"].tobytes()
    assert np.array_equal(dump["prims"], np.concatenate(sb._prim_chunks))
    assert np.allclose(dump["sph"]["c0"], [[0, -1000, 0], [3, 1, 0]]) and dump["sph"]["r"].tolist() == [1000.0, 1.0]


def test_cpp_png_decoder_matches_pil(dump):
    from PIL import Image
    want = {}
    for name in ("Image_0.png", "Image_1.png"):
        a = np.asarray(Image.open(os.path.join(ROOT, "assets", name)).convert("RGB"), dtype=np.uint8)
        want[hashlib.sha1(a.tobytes()).hexdigest()] = a.shape
    seen = set()
    for t in dump["tex"]:
        if t["kind"] == 2 and t["w"] > 0:
            b = dump["texels"][t["off"]:t["off"] + int(t["w"]) * t["h"] * t["bpp"]]
            h = hashlib.sha1(b.tobytes()).hexdigest()
            assert h in want and want[h] == (t["h"], t["w"], t["bpp"])
            seen.add(h)
    assert seen == set(want)
    # the writer's output is a valid PNG that PIL decodes to the same bytes
    back = np.asarray(Image.open(dump["png"]).convert("RGB"), dtype=np.uint8)
    assert hashlib.sha1(back.tobytes()).hexdigest() in want


def test_cpp_bvhnode_is_the_reference_build(dump, srt, oracle):
    """bvhNode built at construction through srtBuildBvh == the oracle's bvh.h:55-95 restatement, and
    the process-global generator has advanced by exactly one draw per node (globals.h:30-35)."""
    sb = _python_scene(srt)
    onodes, _ = oracle.OracleScene(sb).bvh(0)
    assert dump["nodes"].tobytes() == onodes.tobytes()
    n = len(onodes)
    assert dump["next"][0] == oracle.rng_kat(n + 1)[n]


def test_cpp_material_flattening(dump):
    mats = dump["mat"].view(np.dtype([("type", "<i4"), ("tex", "<i4", 4), ("albedo", "<f4", 4), ("metalness", "<f4"),
                                       ("roughness", "<f4"), ("fuzz", "<f4"), ("ir", "<f4"), ("pad", "<i4", 3)]))
    mats = mats.reshape(-1)
    # two mesh materials (pbr: albedo+normal maps, metallic 0, roughness 1 = glTF default), ground pbr, metal
    assert mats["type"].tolist() == [0, 0, 0, 1]
    assert (mats["tex"][:2, :2] >= 0).all() and (mats["tex"][:2, 2:] == -1).all()
    assert mats["metalness"][:2].tolist() == [0.0, 0.0] and mats["roughness"][:2].tolist() == [1.0, 1.0]
    assert mats["metalness"][2] == 0.0 and mats["roughness"][2] == 0.0  # ctor material.h:29-32, defined as 0
    assert np.allclose(mats["albedo"][3][:3], [0.7, 0.6, 0.5]) and mats["fuzz"][3] == 0.0
    ground_tex = dump["tex"][mats["tex"][2][0]]
    assert ground_tex["kind"] == 1  # checker
    assert np.allclose(dump["tex"][ground_tex["even"]]["color"], [0.2, 0.3, 0.1])
    assert np.allclose(dump["tex"][ground_tex["odd"]]["color"], [0.9, 0.9, 0.9])


@pytest.mark.parametrize("mode,req", [("RGB", 3), ("RGB", 1), ("RGB", 4), ("L", 1), ("L", 3), ("RGBA", 3), ("RGBA", 4),
                                      ("LA", 1), ("LA", 4), ("P", 3), ("I;16", 1)])
def test_cpp_png_decoder_formats(tmp_path, mode, req):
    """srt/png.h (the stbi_load replacement) on every colour type PNG allows, converted to the requested
    component count with stb_image's rules; the reference only ever asks for 3 or 1 (texture.h:62,115)."""
    from PIL import Image
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    tool = os.path.join(ROOT, "examples", "srt_png_tool")
    rng = np.random.default_rng(abs(hash((mode, req))) % 1000)
    w, h = 37, 23
    if mode == "I;16":
        src = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        im = Image.fromarray(src)
        rgba = np.stack([src >> 8] * 3 + [np.full_like(src, 255)], -1).astype(np.uint8)  # high byte, as stb does
    elif mode == "P":
        idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
        pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
        im = Image.fromarray(idx, "P")
        im.putpalette(pal.tobytes())
        rgba = np.concatenate([pal[idx], np.full((h, w, 1), 255, np.uint8)], -1)
    else:
        nch = {"RGB": 3, "L": 1, "RGBA": 4, "LA": 2}[mode]
        src = rng.integers(0, 256, (h, w, nch), dtype=np.uint8)
        im = Image.fromarray(src[..., 0] if nch == 1 else src, mode)
        if nch <= 2:
            rgba = np.concatenate([np.repeat(src[..., :1], 3, -1), src[..., 1:2] if nch == 2 else np.full((h, w, 1), 255, np.uint8)], -1)
        else:
            rgba = np.concatenate([src[..., :3], src[..., 3:4] if nch == 4 else np.full((h, w, 1), 255, np.uint8)], -1)
    path = str(tmp_path / "in.png")
    im.save(path)
    raw = str(tmp_path / "out.raw")
    out = subprocess.check_output([tool, "decode", path, str(req), raw]).decode().split()
    assert (int(out[0]), int(out[1])) == (w, h)
    got = np.fromfile(raw, np.uint8).reshape(h, w, req)
    r, g, b, a = (rgba[..., k].astype(np.int32) for k in range(4))
    grey_src = mode in ("L", "LA", "I;16")
    y = r if grey_src else ((r * 77 + g * 150 + 29 * b) >> 8)
    want = {1: np.stack([y], -1), 2: np.stack([y, a], -1), 3: np.stack([r, g, b], -1), 4: np.stack([r, g, b, a], -1)}[req]
    assert np.array_equal(got, want.astype(np.uint8))
    # and the writer round-trips what it is given
    back = str(tmp_path / "back.png")
    subprocess.check_call([tool, "encode", raw, str(w), str(h), str(req), back])
    pil = np.asarray(Image.open(back))
    assert np.array_equal(pil.reshape(h, w, req), got)


def test_cpp_png_decoder_rejects_hostile_headers(tmp_path):
    """Header fields of a texture file named by a glTF are untrusted: huge or zero dimensions,
    an IHDR chunk of the wrong length and truncated files must fail cleanly, not overflow the size arithmetic."""
    import struct
    import zlib
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")])
    tool = os.path.join(ROOT, "examples", "srt_png_tool")

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)

    def png(width, height, ihdr_extra=b"", idat=None):
        ihdr = struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0) + ihdr_extra
        idat = zlib.compress(b"\0" * 64) if idat is None else idat
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

    cases = {"huge": png(0x7fffffff, 0x7fffffff), "wrap": png(0x10000001, 0x10), "zero": png(0, 5),
             "long_ihdr": png(4, 4, ihdr_extra=b"\0\0\0"), "short_idat": png(64, 64),
             "truncated": png(4, 4)[:40]}
    for name, blob in cases.items():
        path = str(tmp_path / (name + ".png"))
        open(path, "wb").write(blob)
        r = subprocess.run([tool, "decode", path, "3", str(tmp_path / "o.raw")], capture_output=True)
        assert r.returncode == 1 and b"decode failed" in r.stderr, (name, r.returncode, r.stderr[-200:])
    # a good file of the same shape still decodes
    good = png(4, 4, idat=zlib.compress(b"".join(b"\0" + bytes(range(12)) for _ in range(4))))
    path = str(tmp_path / "good.png")
    open(path, "wb").write(good)
    assert subprocess.run([tool, "decode", path, "3", str(tmp_path / "o.raw")], capture_output=True).returncode == 0


# ---- N1: every model file the reference ships loads the way gltfLoad loads it (model.h:301-460) ----
REF_DATA = "/root/reference/data"


def _fnv(b, h=1469598103934665603):
    for x in np.frombuffer(b, np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _fnv_fast(arr):
    """FNV-1a over the bytes of arr (numpy, vectorised per byte position would change the order: plain loop on small data)."""
    return _fnv(np.ascontiguousarray(arr).tobytes())


def _cpp_model_dump(path, data_dir):
    exe = os.path.join(ROOT, "examples", "srt_flatten_dump")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sexy-raytracer_amd", "host")], stdout=subprocess.DEVNULL)
    r = subprocess.run([exe, "--model", path], env=dict(os.environ, SRT_DATA_DIR=data_dir), stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, text=True)
    meshes, flat = [], None
    for line in r.stdout.splitlines():
        w = line.split()
        if w[:1] == ["mesh"]:
            meshes.append({"positions": int(w[3]), "texcoords": int(w[5]), "triangles": int(w[7]), "material": int(w[9]),
                           "hp": int(w[11]), "ht": int(w[13])})
        elif w[:2] == ["flattened", "triangles"]:
            flat = {"triangles": int(w[2]), "materials": int(w[4]), "textures": int(w[6]), "hash": int(w[8])}
    return r.returncode, meshes, flat


def _check_model_file(srt, path, data_dir, expect_meshes):
    import importlib
    gltf = importlib.import_module("sexy-raytracer_amd.gltf")
    rc, meshes, flat = _cpp_model_dump(path, data_dir)
    assert rc == 0, "C++ gltfLoad failed on %s" % path
    py = gltf.load_gltf(path)
    assert len(py) == len(meshes) == expect_meshes
    for a, b in zip(py, meshes):
        assert (len(a["positions"]), len(a["texcoords"]), len(a["indices"])) == (b["positions"], b["texcoords"], b["triangles"])
        assert (a["material"] is not None) == bool(b["material"])
        assert _fnv_fast(a["positions"]) == b["hp"] and _fnv_fast(a["texcoords"]) == b["ht"]
    # the triangles as objects.add() sees them (main.cpp:81-85): same vertices through both loaders
    sb = srt.abi.SceneBuilder()
    n = srt.scenes.add_model(sb, path)
    assert n == flat["triangles"] == sum(m["triangles"] for m in meshes)
    tri = np.concatenate(sb.triangles)
    h = 1469598103934665603
    for t in tri:
        h = _fnv(t["uv"].tobytes(), _fnv(t["p"].tobytes(), h))
    assert h == flat["hash"]
    return py, meshes, flat, sb


def test_data_uri_and_glb_loading_synthetic(tmp_path, srt):
    """cgltf_load_buffers' three buffer sources on the same mesh: external .bin (assets/), the same bytes as a base64
    `data:` URI, and as the BIN chunk of a .glb -- identical vertices through the C++ and the Python loader; and the
    failures gltfLoad turns into `false` (truncated base64, foreign URI scheme, bad .glb chunk magic)."""
    import base64
    import json
    import shutil
    import struct as st
    src = os.path.join(ROOT, "assets", "masterchief2-separate-xf.gltf")
    g = json.load(open(src))
    raw = open(os.path.join(ROOT, "assets", g["buffers"][0]["uri"]), "rb").read()
    for name in ("Image_0.png", "Image_1.png"):
        shutil.copy(os.path.join(ROOT, "assets", name), tmp_path / name)
    want = _check_model_file(srt, src, os.path.join(ROOT, "assets"), 2)[2]

    uri = dict(g)
    uri["buffers"] = [{"byteLength": len(raw), "uri": "data:application/octet-stream;base64," + base64.b64encode(raw).decode()}]
    p_uri = str(tmp_path / "uri.gltf")
    json.dump(uri, open(p_uri, "w"))
    assert _check_model_file(srt, p_uri, str(tmp_path), 2)[2] == want

    glb = dict(g)
    glb["buffers"] = [{"byteLength": len(raw)}]
    js = json.dumps(glb).encode()
    js += b" " * (-len(js) % 4)
    binc = raw + b"\0" * (-len(raw) % 4)
    body = st.pack("<II", len(js), 0x4E4F534A) + js + st.pack("<II", len(binc), 0x004E4942) + binc
    p_glb = str(tmp_path / "m.glb")
    open(p_glb, "wb").write(b"glTF" + st.pack("<II", 2, 12 + len(body)) + body)
    assert _check_model_file(srt, p_glb, str(tmp_path), 2)[2] == want

    import importlib
    gltf = importlib.import_module("sexy-raytracer_amd.gltf")
    bad = []
    trunc = dict(uri)
    trunc["buffers"] = [{"byteLength": len(raw), "uri": uri["buffers"][0]["uri"][:2000]}]
    bad.append(("trunc.gltf", json.dumps(trunc).encode()))
    http = dict(uri)
    http["buffers"] = [{"byteLength": len(raw), "uri": "http://example.invalid/m.bin"}]
    bad.append(("http.gltf", json.dumps(http).encode()))
    bad.append(("magic.glb", b"glTF" + st.pack("<II", 2, 12 + len(body)) + body.replace(st.pack("<I", 0x004E4942), b"XXXX", 1)))
    bad.append(("v1.glb", b"glTF" + st.pack("<II", 1, 12 + len(body)) + body))
    for name, data in bad:
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        assert _cpp_model_dump(p, str(tmp_path))[0] != 0, name
        with pytest.raises(gltf.GltfError):
            gltf.load_gltf(p)


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's data/ directory is not on this machine")
@pytest.mark.parametrize("name,meshes,triangles", [
    ("square.gltf", 1, 2),            # main.cpp:62, base64 data: URI buffer
    ("cube.gltf", 1, 12),             # main.cpp:61, data: URI
    ("masterchief.gltf", 5, 4074),    # data: URI, five primitives of one mesh
    ("masterchief2.gltf", 2, 3042),   # data: URI buffer AND embedded data: URI images (failed loads: magenta)
    ("masterchief.glb", 5, 4074),     # binary glTF, BIN chunk
    ("halo.glb", 2, 3042),            # binary glTF with images in bufferViews (no uri)
    ("scene.gltf", 15, 5614),         # main.cpp:78: fifteen meshes of one primitive -> everything lands in meshes[0]
    ("masterchief-sep.gltf", 4, 2016),
    ("masterchief2-separate.gltf", 2, 3042),
    ("masterchief2-separate-xf.gltf", 2, 3042),  # main.cpp:74
])
def test_reference_model_files_load_like_gltfload(srt, name, meshes, triangles):
    py, cpp, flat, sb = _check_model_file(srt, os.path.join(REF_DATA, name), REF_DATA, meshes)
    assert flat["triangles"] == triangles
    if name == "scene.gltf":
        # the meshes[primIndex] quirk (model.h:345,361,450): every primitive is primitive 0 of its own glTF mesh, so all
        # vertex data and triangles pile up in meshes[0] and every triangle wears the first primitive's material
        assert [m["triangles"] for m in cpp] == [5614] + [0] * 14 and cpp[0]["positions"] == 5740
        assert flat["materials"] == 1
        assert len({int(t["material"]) for t in np.concatenate(sb.triangles)}) == 1
    if name in ("masterchief2.gltf", "halo.glb"):
        # embedded images cannot be opened as files: imagePNG load failures, as in the reference (texture.h:117-120)
        assert all(t.width == 0 for t in sb.textures) and len(sb.textures) == flat["textures"] == 4
