// model.h -- host mirror of triangle / mesh / model and gltfLoad (model.h:18-460).
// triangle::hit, getNormal and calcTangentBasis are device / upload-time code
// (srt_kernels.hip triHit, srt_api.cpp srtUploadScene); this file keeps the containers, the bounding
// box the BVH build needs, and a glTF loader with gltfLoad's semantics.
#ifndef SRT_HOST_MODEL_H
#define SRT_HOST_MODEL_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "hittable.h"
#include "json.h"
#include "material.h"

using std::uint16_t;
using std::vector;

bool gltfLoad(std::string filename, shared_ptr<class model> model);

// primitive in glTF
class mesh : public hittable, public std::enable_shared_from_this<mesh> {
 public:
  shared_ptr<mesh> getPtr() { return shared_from_this(); }
  [[nodiscard]] static shared_ptr<mesh> create() { return shared_ptr<mesh>(new mesh()); }
  bool boundingBox(float, float, aabb&) const override { return true; }  // model.h:289-291
  int populate(sceneFlattener& f) const override;
  bool isPrimitive() const override { return false; }

 private:
  mesh() {}

 public:
  std::vector<shared_ptr<class triangle>> triangles;
  std::vector<vec3f> positions;
  std::vector<vec2f> texcoords;
  shared_ptr<material> matPtr;
  shared_ptr<class model> parentModel;
};

class triangle : public hittable, public std::enable_shared_from_this<triangle> {
 public:
  shared_ptr<triangle> getPtr() { return shared_from_this(); }
  [[nodiscard]] static shared_ptr<triangle> create(uint16_t index0, uint16_t index1, uint16_t index2,
                                                   shared_ptr<class mesh> srcMesh) {
    return shared_ptr<triangle>(new triangle(index0, index1, index2, srcMesh));
  }
  bool boundingBox(float, float, aabb& outputBox) const override {  // model.h:183-212
    vec3f mn(infinity, infinity, infinity), mx(-infinity, -infinity, -infinity);
    for (const auto& vertex : vertices)
      for (int axis = 0; axis < 3; axis++) {
        mn(axis) = std::min(mn(axis), parentMesh->positions[vertex](axis));
        mx(axis) = std::max(mx(axis), parentMesh->positions[vertex](axis));
      }
    for (int axis = 0; axis < 3; axis++)
      if (mn(axis) == mx(axis)) {
        mn(axis) -= 0.0001f;
        mx(axis) += 0.0001f;
      }
    outputBox = surroundingBox(aabb(mn, mx), aabb(mn, mx));
    return true;
  }
  // gathers the vertex data through the u16 indices, as triangle::hit does per call (model.h:108-111)
  int populate(sceneFlattener& f) const override {
    SrtTriangleIn t{};
    for (int i = 0; i < 3; ++i) {
      const vec3f& p = parentMesh->positions[vertices[i]];
      const vec2f& uv = parentMesh->texcoords[vertices[i]];
      for (int k = 0; k < 3; ++k) t.p[i][k] = p(k);
      t.uv[i][0] = uv(0);
      t.uv[i][1] = uv(1);
    }
    t.material = f.materialId(parentMesh->matPtr);
    return f.addTriangle(t);
  }

 private:
  triangle(uint16_t index0, uint16_t index1, uint16_t index2, shared_ptr<class mesh> srcMesh) : parentMesh(srcMesh) {
    vertices[0] = index0;
    vertices[1] = index1;
    vertices[2] = index2;
  }
  uint16_t vertices[3];
  shared_ptr<class mesh> parentMesh;
};

inline int mesh::populate(sceneFlattener& f) const {
  int first = (int)f.prims.size();
  for (const auto& t : triangles) t->populate(f);
  return first;
}

class model : public hittable, public std::enable_shared_from_this<model> {
 public:
  shared_ptr<model> getPtr() { return shared_from_this(); }
  [[nodiscard]] static shared_ptr<model> create(std::string fn) { return shared_ptr<model>(new model(fn)); }
  bool boundingBox(float, float, aabb&) const override { return true; }  // model.h:297-299
  int populate(sceneFlattener& f) const override {
    int first = (int)f.prims.size();
    for (const auto& m : meshes) m->populate(f);
    return first;
  }
  bool isPrimitive() const override { return false; }
  bool init() { return gltfLoad(filename, getPtr()); }  // model.h:92-94

 private:
  model(std::string fn) : filename(fn) {}

 public:
  std::string filename;
  std::vector<shared_ptr<mesh>> meshes;
};

// Directory the loader prepends to image URIs.  The reference hard-codes "../data/" (paths relative
// to build/, model.h:395,403,411); SRT_DATA_DIR overrides it.
inline std::string srtDataDir() {
  const char* e = getenv("SRT_DATA_DIR");
  std::string d = (e && *e) ? e : "../data/";
  if (!d.empty() && d.back() != '/') d += '/';
  return d;
}

// ---- what cgltf_parse_file + cgltf_load_buffers accept (model.h:301-315, cgltf file type 0 = detect) ----
// base64 as cgltf_load_buffer_base64 reads it: exactly `size` bytes are decoded (the '=' padding is never
// reached); a character outside the alphabet before that is an error.
inline bool srtBase64Decode(const char* src, size_t srcLen, size_t size, std::vector<uint8_t>& out) {
  out.clear();
  out.reserve(size);
  unsigned buffer = 0, bits = 0;
  size_t at = 0;
  for (size_t i = 0; i < size; ++i) {
    while (bits < 8) {
      if (at >= srcLen) return false;
      const char ch = src[at++];
      int index = (unsigned)(ch - 'A') < 26   ? (ch - 'A')
                  : (unsigned)(ch - 'a') < 26 ? (ch - 'a') + 26
                  : (unsigned)(ch - '0') < 10 ? (ch - '0') + 52
                  : ch == '+'                 ? 62
                  : ch == '/'                 ? 63
                                              : -1;
      if (index < 0) return false;
      buffer = (buffer << 6) | (unsigned)index;
      bits += 6;
    }
    out.push_back((uint8_t)(buffer >> (bits - 8)));
    bits -= 8;
  }
  return true;
}
// %XX escapes of a relative file URI (cgltf_decode_uri)
inline std::string srtUriDecode(const std::string& uri) {
  auto hex = [](char c) { return (unsigned)(c - '0') < 10 ? c - '0' : (unsigned)((c | 32) - 'a') < 6 ? (c | 32) - 'a' + 10 : -1; };
  std::string out;
  for (size_t i = 0; i < uri.size(); ++i) {
    if (uri[i] == '%' && i + 2 < uri.size() && hex(uri[i + 1]) >= 0 && hex(uri[i + 2]) >= 0) {
      out.push_back((char)(hex(uri[i + 1]) * 16 + hex(uri[i + 2])));
      i += 2;
    } else {
      out.push_back(uri[i]);
    }
  }
  return out;
}
// The JSON text and, for a binary .glb (magic "glTF", version 2: 12-byte header, JSON chunk, optional BIN chunk),
// the BIN chunk.  Returns false for what cgltf_parse rejects (short file, wrong version, wrong chunk magic).
inline bool srtGltfSplit(const std::string& file, std::string& json, std::vector<uint8_t>& bin, bool& hasBin) {
  hasBin = false;
  auto u32 = [&](size_t at) {
    uint32_t v;
    memcpy(&v, file.data() + at, 4);
    return v;
  };
  if (file.size() < 4 || u32(0) != 0x46546C67u) {  // not "glTF": JSON text
    json = file;
    return !file.empty();
  }
  if (file.size() < 20 || u32(4) != 2u || u32(8) > file.size()) return false;
  const size_t total = u32(8), jsonLen = u32(12);
  if (u32(16) != 0x4E4F534Au || 20 + jsonLen > total) return false;  // "JSON"
  json.assign(file, 20, jsonLen);
  const size_t binHdr = 20 + jsonLen;
  if (binHdr + 8 <= total) {
    const size_t binLen = u32(binHdr);
    if (u32(binHdr + 4) != 0x004E4942u || binHdr + 8 + binLen > total) return false;  // "BIN\0"
    bin.assign(file.begin() + binHdr + 8, file.begin() + binHdr + 8 + binLen);
    hasBin = true;
  }
  return true;
}

// gltfLoad, model.h:301-460, semantics kept:
//  * the file may be JSON text or a binary .glb; buffers may be external files (relative, %XX-decoded), base64
//    `data:` URIs, or the .glb's BIN chunk -- what cgltf_parse_file + cgltf_load_buffers load (:301-315); any
//    other URI scheme, a short buffer or a missing file fails the load
//  * meshes -> primitives only; node transforms ignored (:317-320)
//  * POSITION (vec3 f32) / TEXCOORD_n (vec2 f32, every set, appended) read from bufferView.byteOffset, accessor
//    byteOffset and stride ignored (:343,359); indices read as u16 whatever their component type (:448)
//  * attribute data and triangles go to model->meshes[primIndex] -- the reference indexes by the
//    primitive's index inside ITS gltf mesh, not by the mesh just pushed (:345,361,450): identical for
//    a single gltf mesh; with several meshes (data/scene.gltf) everything lands in the first primIndex entries
//  * images: "<data dir>/<uri>" loaded with 3 components (:420-431), so an embedded image (a `data:` URI, or a
//    .glb image that lives in a bufferView -- the reference appends a null uri there, undefined behaviour) is a
//    FAILED load: magenta, texture.h:130-131; material via the ctor :60-66 with baseColorFactor / metallicFactor /
//    roughnessFactor (glTF defaults 1,1,1,1 / 1 / 1)
inline bool gltfLoad(std::string filename, shared_ptr<model> model) {
  std::ifstream in(filename, std::ios::binary);
  if (!in) return false;
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string file = ss.str();
  std::string text;
  std::vector<uint8_t> glbBin;
  bool hasBin = false;
  if (!srtGltfSplit(file, text, glbBin, hasBin)) return false;
  srtJson g;
  if (!srtJsonParser(text).parse(g)) return false;
  const std::string dir = filename.find_last_of('/') == std::string::npos ? "" : filename.substr(0, filename.find_last_of('/') + 1);
  std::vector<std::vector<uint8_t>> buffers;
  for (size_t i = 0; i < g["buffers"].size(); ++i) {
    const srtJson& b = g["buffers"][i];
    const size_t size = (size_t)b["byteLength"].number(0);
    if (!b.has("uri")) {
      // cgltf_load_buffers: buffer 0 without a uri is the BIN chunk; any other stays without data (the reference
      // would then read through a null pointer: fail instead)
      if (i != 0 || !hasBin || glbBin.size() < size) return false;
      buffers.push_back(glbBin);
      continue;
    }
    const std::string& uri = b["uri"].str;
    if (uri.compare(0, 5, "data:") == 0) {
      const size_t comma = uri.find(',');
      if (comma == std::string::npos || comma < 7 || uri.compare(comma - 7, 7, ";base64") != 0) return false;
      std::vector<uint8_t> bytes;
      if (!srtBase64Decode(uri.data() + comma + 1, uri.size() - comma - 1, size, bytes)) return false;
      buffers.push_back(std::move(bytes));
    } else if (uri.find("://") == std::string::npos) {
      std::ifstream bf(dir + srtUriDecode(uri), std::ios::binary);
      if (!bf) return false;  // cgltf_load_buffers failure (:312-315)
      buffers.emplace_back((std::istreambuf_iterator<char>(bf)), std::istreambuf_iterator<char>());
      if (buffers.back().size() < size) return false;
    } else {
      return false;  // cgltf_result_unknown_format
    }
  }
  auto viewPtr = [&](const srtJson& accessor, size_t bytesNeeded) -> const uint8_t* {
    const srtJson& bv = g["bufferViews"][(size_t)accessor["bufferView"].number(-1)];
    size_t buf = (size_t)bv["buffer"].number(0), off = (size_t)bv["byteOffset"].number(0);
    if (bv.isNull() || buf >= buffers.size() || off + bytesNeeded > buffers[buf].size()) return nullptr;
    return buffers[buf].data() + off;
  };
  auto imageFile = [&](const srtJson& texInfo) -> std::string {
    if (texInfo.isNull()) return "";
    const srtJson& tex = g["textures"][(size_t)texInfo["index"].number(-1)];
    const srtJson& img = g["images"][(size_t)tex["source"].number(-1)];
    // no image -> no texture; an image without a uri (embedded in a bufferView) -> "<data dir>/" which cannot be loaded
    return img.isNull() ? "" : srtDataDir() + img["uri"].str;
  };

  for (size_t meshIndex = 0; meshIndex < g["meshes"].size(); ++meshIndex) {
    const srtJson& prims = g["meshes"][meshIndex]["primitives"];
    for (size_t primIndex = 0; primIndex < prims.size(); ++primIndex) {
      const srtJson& prim = prims[primIndex];
      shared_ptr<mesh> newMesh = mesh::create();
      model->meshes.push_back(newMesh);
      shared_ptr<mesh>& target = model->meshes[primIndex];

      for (const auto& attr : prim["attributes"].obj) {
        const srtJson& a = g["accessors"][(size_t)attr.second.number(-1)];
        size_t count = (size_t)a["count"].number(0);
        if (attr.first == "POSITION" && a["type"].str == "VEC3") {
          const uint8_t* p = viewPtr(a, count * 12);
          if (!p) return false;
          for (size_t i = 0; i < count; ++i) {
            float v[3];
            memcpy(v, p + i * 12, 12);
            target->positions.push_back(vec3f(v[0], v[1], v[2]));
          }
        }
        if (attr.first.compare(0, 8, "TEXCOORD") == 0 && a["type"].str == "VEC2") {
          const uint8_t* p = viewPtr(a, count * 8);
          if (!p) return false;
          for (size_t i = 0; i < count; ++i) {
            float v[2];
            memcpy(v, p + i * 8, 8);
            target->texcoords.push_back(vec2f(v[0], v[1]));
          }
        }
      }

      if (prim.has("material")) {
        const srtJson& m = g["materials"][(size_t)prim["material"].number(-1)];
        if (m.has("pbrMetallicRoughness")) {
          const srtJson& pbr = m["pbrMetallicRoughness"];
          std::string textureFile = imageFile(pbr["baseColorTexture"]);
          std::string normalMapFile = imageFile(m["normalTexture"]);
          std::string mrFile = imageFile(pbr["metallicRoughnessTexture"]);
          vec4f baseColor(1.0f, 1.0f, 1.0f, 1.0f);
          if (pbr["baseColorFactor"].size() == 4)
            for (int i = 0; i < 4; ++i) baseColor(i) = (float)pbr["baseColorFactor"][i].number(1.0);
          float metallicness = (float)pbr["metallicFactor"].number(1.0);
          float roughness = (float)pbr["roughnessFactor"].number(1.0);
          shared_ptr<imagePNG> albedoPNG, normalPNG, mrPNG;
          if (!textureFile.empty()) albedoPNG = make_shared<imagePNG>(textureFile.c_str(), 3);
          if (!normalMapFile.empty()) normalPNG = make_shared<imagePNG>(normalMapFile.c_str(), 3);
          if (!mrFile.empty()) mrPNG = make_shared<imagePNG>(mrFile.c_str(), 3);
          newMesh->matPtr = make_shared<pbrMetallicRoughness>(albedoPNG, normalPNG, mrPNG, baseColor, metallicness, roughness);
        }
      }

      if (prim["mode"].number(4) == 4 && prim.has("indices")) {  // cgltf_primitive_type_triangles
        const srtJson& a = g["accessors"][(size_t)prim["indices"].number(-1)];
        size_t count = (size_t)a["count"].number(0);
        const uint8_t* p = viewPtr(a, count * 2);
        if (!p) return false;
        for (size_t idx = 0; idx + 2 < count; idx += 3) {
          uint16_t i3[3];
          memcpy(i3, p + idx * 2, 6);
          if (i3[0] >= target->positions.size() || i3[1] >= target->positions.size() || i3[2] >= target->positions.size() ||
              std::max({i3[0], i3[1], i3[2]}) >= target->texcoords.size())
            return false;  // the reference would read out of bounds
          target->triangles.push_back(triangle::create(i3[0], i3[1], i3[2], target));
        }
      }
    }
  }
  return true;
}

#endif
