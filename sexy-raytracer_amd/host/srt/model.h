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

// gltfLoad, model.h:301-460, semantics kept:
//  * meshes -> primitives only; node transforms ignored (:317-320)
//  * POSITION (vec3 f32) / TEXCOORD (vec2 f32) read from bufferView.byteOffset, accessor byteOffset and
//    stride ignored (:343,359); indices read as u16 (:448)
//  * attribute data and triangles go to model->meshes[primIndex] -- the reference indexes by the
//    primitive's index inside ITS gltf mesh, not by the mesh just pushed (:345,361,450): identical for
//    a single gltf mesh, kept as is for several
//  * images: "<data dir>/<uri>" loaded with 3 components (:420-431); material via the ctor :60-66
//    with baseColorFactor / metallicFactor / roughnessFactor (glTF defaults 1,1,1,1 / 1 / 1)
inline bool gltfLoad(std::string filename, shared_ptr<model> model) {
  std::ifstream in(filename, std::ios::binary);
  if (!in) return false;
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  srtJson g;
  if (!srtJsonParser(text).parse(g)) return false;
  const std::string dir = filename.find_last_of('/') == std::string::npos ? "" : filename.substr(0, filename.find_last_of('/') + 1);
  std::vector<std::vector<uint8_t>> buffers;
  for (size_t i = 0; i < g["buffers"].size(); ++i) {
    std::ifstream b(dir + g["buffers"][i]["uri"].str, std::ios::binary);
    if (!b) return false;  // cgltf_load_buffers failure (:312-315)
    buffers.emplace_back((std::istreambuf_iterator<char>(b)), std::istreambuf_iterator<char>());
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
