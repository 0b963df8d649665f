"""Minimal glTF 2.0 reader with the semantics of the reference's gltfLoad
(model.h:301-460), for the Python-side scene builders used by tests/bench.  The C++ drop-in of the same
function is host/srt/model.h; tests/test_host_cpp.py holds the two against each other.

What gltfLoad does and this mirrors:
  * the file is JSON text or a binary .glb, buffers are external files, base64 `data:` URIs or the .glb's
    BIN chunk: what cgltf_parse_file + cgltf_load_buffers (file type 0 = detect, model.h:301-315) load;
    anything else raises GltfError (gltfLoad returns false)
  * walks meshes -> primitives only; node transforms are ignored (model.h:317-320)
  * POSITION (vec3 f32) and every TEXCOORD_n (vec2 f32) are read from
    bufferView.byteOffset, ignoring accessor byteOffset and stride (model.h:343,359)
  * indices are read as u16 whatever the componentType (model.h:448)
  * one `mesh` is pushed per primitive, but attribute data and triangles go to meshes[primIndex] -- the
    primitive's index inside ITS glTF mesh (model.h:345,361,450).  Identical for a file with one glTF mesh;
    with several (data/scene.gltf) everything piles up in the first entries and the triangles use the
    first primitive's material
  * material: baseColorTexture / normalTexture / metallicRoughnessTexture image
    URIs, base_color_factor, metallic_factor, roughness_factor with glTF defaults
    (1,1,1,1 / 1 / 1) when absent (cgltf fills these).  An embedded image (data: URI or bufferView) is a
    file name that cannot be opened: a failed imagePNG load (texture.h:117-120,130-131)
"""
import json
import os
import re
import struct

import numpy as np

_B64_RE = re.compile(r"^[A-Za-z0-9+/]*$")


class GltfError(ValueError):
    """gltfLoad would have returned false (or crashed) on this file."""


def _base64(text, size):
    """cgltf_load_buffer_base64: exactly `size` bytes are decoded; the padding is never reached, a character
    outside the alphabet before that fails the load."""
    import base64
    need = (size * 8 + 5) // 6
    head = text[:need]
    if len(head) < need or not _B64_RE.match(head):
        raise GltfError("bad base64 buffer")
    return base64.b64decode(head + "A" * (-need % 4))[:size]


def _uri_decode(uri):
    out, i = [], 0
    while i < len(uri):
        if uri[i] == "%" and i + 2 < len(uri) and all(c in "0123456789abcdefABCDEF" for c in uri[i + 1:i + 3]):
            out.append(chr(int(uri[i + 1:i + 3], 16)))
            i += 3
        else:
            out.append(uri[i])
            i += 1
    return "".join(out)


def _split(raw):
    """(json text, BIN chunk or None) of a .gltf / .glb file."""
    if len(raw) < 4 or raw[:4] != b"glTF":
        return raw.decode("utf-8"), None
    if len(raw) < 20:
        raise GltfError("short .glb")
    version, total, json_len, json_magic = struct.unpack_from("<IIII", raw, 4)
    if version != 2 or total > len(raw) or json_magic != 0x4E4F534A or 20 + json_len > total:
        raise GltfError("bad .glb header")
    text = raw[20:20 + json_len].decode("utf-8")
    at = 20 + json_len
    if at + 8 <= total:
        bin_len, bin_magic = struct.unpack_from("<II", raw, at)
        if bin_magic != 0x004E4942 or at + 8 + bin_len > total:
            raise GltfError("bad .glb BIN chunk")
        return text, raw[at + 8:at + 8 + bin_len]
    return text, None


def load_gltf(path):
    """Returns model->meshes as a list of dicts {positions (n,3) f32, texcoords (m,2) f32, indices (k,3) i32,
    material dict or None}; raises GltfError where gltfLoad fails."""
    try:
        with open(path, "rb") as f:
            raw = f.read()
    except OSError as e:
        raise GltfError(str(e))
    text, glb_bin = _split(raw)
    try:
        g = json.loads(text)
    except ValueError as e:
        raise GltfError("json: %s" % e)
    base = os.path.dirname(path)
    buffers = []
    for i, b in enumerate(g.get("buffers", [])):
        size = int(b.get("byteLength", 0))
        uri = b.get("uri")
        if uri is None:
            if i != 0 or glb_bin is None or len(glb_bin) < size:
                raise GltfError("buffer %d has no data" % i)
            buffers.append(glb_bin)
        elif uri.startswith("data:"):
            comma = uri.find(",")
            if comma < 7 or uri[comma - 7:comma] != ";base64":
                raise GltfError("data: URI that is not base64")
            buffers.append(_base64(uri[comma + 1:], size))
        elif "://" not in uri:
            try:
                with open(os.path.join(base, _uri_decode(uri)), "rb") as f:
                    buffers.append(f.read())
            except OSError as e:
                raise GltfError(str(e))
            if len(buffers[-1]) < size:
                raise GltfError("buffer file shorter than byteLength")
        else:
            raise GltfError("unsupported buffer URI scheme")

    def view(acc, dtype, count):
        try:
            bv = g["bufferViews"][acc["bufferView"]]
            buf, off = buffers[bv["buffer"]], bv.get("byteOffset", 0)
            return np.frombuffer(buf, dtype, count, off)
        except (KeyError, IndexError, ValueError) as e:
            raise GltfError("accessor outside its buffer: %s" % e)

    def image_file(texinfo):
        if texinfo is None:
            return None
        tex = g["textures"][texinfo["index"]]
        return os.path.join(base, g["images"][tex["source"]].get("uri", ""))

    meshes = []
    for gm in g.get("meshes", []):
        for prim_index, prim in enumerate(gm["primitives"]):
            new = {"positions": [], "texcoords": [], "indices": [], "material": None}
            meshes.append(new)
            target = meshes[prim_index]
            for name, idx in prim["attributes"].items():
                acc = g["accessors"][idx]
                if name == "POSITION" and acc["type"] == "VEC3":
                    target["positions"].append(view(acc, "<f4", acc["count"] * 3).reshape(-1, 3))
                if name.split("_")[0] == "TEXCOORD" and acc["type"] == "VEC2":
                    target["texcoords"].append(view(acc, "<f4", acc["count"] * 2).reshape(-1, 2))
            if "material" in prim:
                m = g["materials"][prim["material"]]
                pbr = m.get("pbrMetallicRoughness")
                if pbr is not None:
                    new["material"] = {
                        "albedo": image_file(pbr.get("baseColorTexture")),
                        "normal": image_file(m.get("normalTexture")),
                        "metallicRoughness": image_file(pbr.get("metallicRoughnessTexture")),
                        "baseColorFactor": tuple(float(x) for x in pbr.get("baseColorFactor", (1.0, 1.0, 1.0, 1.0))),
                        "metallicFactor": float(pbr.get("metallicFactor", 1.0)),
                        "roughnessFactor": float(pbr.get("roughnessFactor", 1.0)),
                    }
            if prim.get("mode", 4) == 4 and "indices" in prim:
                acc = g["accessors"][prim["indices"]]
                n = acc["count"] - acc["count"] % 3
                target["indices"].append(view(acc, "<u2", n).reshape(-1, 3).astype(np.int32))
    for m in meshes:
        m["positions"] = np.concatenate(m["positions"]).astype(np.float32) if m["positions"] else np.zeros((0, 3), np.float32)
        m["texcoords"] = np.concatenate(m["texcoords"]).astype(np.float32) if m["texcoords"] else np.zeros((0, 2), np.float32)
        m["indices"] = np.concatenate(m["indices"]) if m["indices"] else np.zeros((0, 3), np.int32)
        if len(m["indices"]) and (m["indices"].max() >= len(m["positions"]) or m["indices"].max() >= len(m["texcoords"])):
            raise GltfError("index beyond the vertex data (the reference would read out of bounds)")
    return meshes
