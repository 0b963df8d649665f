"""Minimal glTF 2.0 reader with the semantics of the reference's gltfLoad
(model.h:301-460), for the Python-side scene builders used by tests/bench.

What gltfLoad does and this mirrors:
  * walks meshes -> primitives only; node transforms are ignored (model.h:317-320)
  * POSITION (vec3 f32) and TEXCOORD (vec2 f32) are read from
    bufferView.offset, ignoring accessor byteOffset and stride (model.h:343,359)
  * indices are read as u16 whatever the componentType (model.h:448)
  * material: baseColorTexture / normalTexture / metallicRoughnessTexture image
    URIs, base_color_factor, metallic_factor, roughness_factor with glTF defaults
    (1,1,1,1 / 1 / 1) when absent (cgltf fills these)
"""
import json
import os

import numpy as np


def load_gltf(path):
    with open(path, "r") as f:
        g = json.load(f)
    base = os.path.dirname(path)
    buffers = []
    for b in g["buffers"]:
        with open(os.path.join(base, b["uri"]), "rb") as f:
            buffers.append(f.read())

    def view_bytes(acc):
        bv = g["bufferViews"][acc["bufferView"]]
        return buffers[bv["buffer"]], bv.get("byteOffset", 0)

    prims = []
    for mesh in g["meshes"]:
        for prim in mesh["primitives"]:
            out = {"positions": None, "texcoords": None, "indices": None, "material": None}
            for name, idx in prim["attributes"].items():
                acc = g["accessors"][idx]
                buf, off = view_bytes(acc)
                if name == "POSITION" and acc["type"] == "VEC3":
                    out["positions"] = np.frombuffer(buf, "<f4", acc["count"] * 3, off).reshape(-1, 3).copy()
                if name.startswith("TEXCOORD") and acc["type"] == "VEC2" and out["texcoords"] is None:
                    out["texcoords"] = np.frombuffer(buf, "<f4", acc["count"] * 2, off).reshape(-1, 2).copy()
            if prim.get("mode", 4) == 4 and "indices" in prim:
                acc = g["accessors"][prim["indices"]]
                buf, off = view_bytes(acc)
                out["indices"] = np.frombuffer(buf, "<u2", acc["count"], off).reshape(-1, 3).astype(np.int32)
            if "material" in prim:
                m = g["materials"][prim["material"]]
                pbr = m.get("pbrMetallicRoughness")
                if pbr is not None:
                    def uri(texinfo):
                        if texinfo is None:
                            return None
                        tex = g["textures"][texinfo["index"]]
                        return os.path.join(base, g["images"][tex["source"]]["uri"])
                    out["material"] = {
                        "albedo": uri(pbr.get("baseColorTexture")),
                        "normal": uri(m.get("normalTexture")),
                        "metallicRoughness": uri(pbr.get("metallicRoughnessTexture")),
                        "baseColorFactor": tuple(pbr.get("baseColorFactor", (1.0, 1.0, 1.0, 1.0))),
                        "metallicFactor": float(pbr.get("metallicFactor", 1.0)),
                        "roughnessFactor": float(pbr.get("roughnessFactor", 1.0)),
                    }
            prims.append(out)
    return prims
