// material.h -- host mirror of the reference's material classes (material.h:15-154): same class names and constructor
// signatures.  scatter() / emitted() keep the reference's signatures (material.h:17-20) and are answered by the device
// (csrc/srt_path.h shade(), through srtScatterRays: shade_device.h); there is no host copy of the BRDF arithmetic.
#ifndef SRT_HOST_MATERIAL_H
#define SRT_HOST_MATERIAL_H

#include "texture.h"

struct hitRecord;
class ray;
struct srtShadeSession;

class material {
 public:
  virtual ~material() {}
  virtual int populate(sceneFlattener& f) const = 0;  // appends an SrtMaterialIn, returns its id
  // material.h:17 and :18-20.  One device round trip per call; the random draws of a scatter come from the counter RNG,
  // keyed by the number of scatter() calls this material has answered (the reference draws from its global generator).
  virtual bool scatter(const ray& rIn, const hitRecord& record, color3f& attenuation, ray& scatterRay) const;
  virtual color3f emitted(float u, float v, const vec3f& p) const;

 private:
  mutable shared_ptr<srtShadeSession> shadeSession_;
};

class pbrMetallicRoughness : public material {  // material.h:23-85
 public:
  // material.h:25-40 leave metalness/roughness/anisotropy uninitialised (undefined behaviour in the
  // reference); they are 0 here, which is what reproduces the published renders.
  pbrMetallicRoughness(const color3f& a) : albedoMap(make_shared<solidColor>(a)), albedo(1.0f, 1.0f, 1.0f, 1.0f) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap) : albedoMap(aMap), albedo(1.0f, 1.0f, 1.0f, 1.0f) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, vec4f a) : albedoMap(aMap), albedo(a) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, shared_ptr<texture> nMap)
      : albedoMap(aMap), normalMap(nMap), albedo(1.0f, 1.0f, 1.0f, 1.0f) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, shared_ptr<texture> nMap, shared_ptr<texture> mMap,
                       shared_ptr<texture> rMap)
      : albedoMap(aMap), normalMap(nMap), metallicMap(mMap), roughnessMap(rMap), albedo(1.0f, 1.0f, 1.0f, 1.0f) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, shared_ptr<texture> nMap, shared_ptr<texture> mMap,
                       shared_ptr<texture> rMap, vec4f a)
      : albedoMap(aMap), normalMap(nMap), metallicMap(mMap), roughnessMap(rMap), albedo(a) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, shared_ptr<texture> nMap, shared_ptr<texture> mrMap, vec4f a)
      : albedoMap(aMap), normalMap(nMap), metallicRoughnessMap(mrMap), albedo(a) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, shared_ptr<texture> nMap, shared_ptr<texture> mrMap, vec4f a,
                       float m, float r)
      : albedoMap(aMap), normalMap(nMap), metallicRoughnessMap(mrMap), albedo(a), metalness(m), roughness(r) {}
  pbrMetallicRoughness(shared_ptr<texture> aMap, vec4f a, float m, float r)
      : albedoMap(aMap), albedo(a), metalness(m), roughness(r) {}

  int populate(sceneFlattener& f) const override {
    SrtMaterialIn m{};
    m.type = SRT_MAT_PBR;
    m.albedoTex = f.textureId(albedoMap);
    m.normalTex = f.textureId(normalMap);
    // metallicRoughnessMap is stored but never sampled by the reference (material.h:190-200)
    m.metallicTex = f.textureId(metallicMap);
    m.roughnessTex = f.textureId(roughnessMap);
    for (int i = 0; i < 4; ++i) m.albedo[i] = albedo(i);
    m.metalness = metalness;
    m.roughness = roughness;
    f.materials.push_back(m);
    return (int)f.materials.size() - 1;
  }

 public:
  shared_ptr<texture> albedoMap, normalMap, metallicRoughnessMap, metallicMap, roughnessMap;
  vec4f albedo;
  float metalness = 0, roughness = 0, anisotropy = 0;
};

class metal : public material {  // material.h:87-102
 public:
  metal(const color3f& a, float f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
  int populate(sceneFlattener& f) const override {
    SrtMaterialIn m{};
    m.type = SRT_MAT_METAL;
    m.albedoTex = m.normalTex = m.metallicTex = m.roughnessTex = -1;
    for (int i = 0; i < 3; ++i) m.albedo[i] = albedo(i);
    m.albedo[3] = 1.0f;
    m.fuzz = fuzz;
    f.materials.push_back(m);
    return (int)f.materials.size() - 1;
  }
  color3f albedo;
  float fuzz;
};

class dielectric : public material {  // material.h:104-137
 public:
  dielectric(float indexRefraction) : ir(indexRefraction) {}
  int populate(sceneFlattener& f) const override {
    SrtMaterialIn m{};
    m.type = SRT_MAT_DIELECTRIC;
    m.albedoTex = m.normalTex = m.metallicTex = m.roughnessTex = -1;
    m.ir = ir;
    f.materials.push_back(m);
    return (int)f.materials.size() - 1;
  }
  float ir;
};

class diffuseLight : public material {  // material.h:139-154
 public:
  diffuseLight(shared_ptr<texture> a) : emit(a) {}
  diffuseLight(color3f c) : emit(make_shared<solidColor>(c)) {}
  int populate(sceneFlattener& f) const override {
    SrtMaterialIn m{};
    m.type = SRT_MAT_LIGHT;
    m.albedoTex = f.textureId(emit);
    m.normalTex = m.metallicTex = m.roughnessTex = -1;
    f.materials.push_back(m);
    return (int)f.materials.size() - 1;
  }

 private:
  shared_ptr<texture> emit;
};

inline int sceneFlattener::materialId(const shared_ptr<material>& m) {
  if (!m) return -1;
  auto it = matIds_.find(m.get());
  if (it != matIds_.end()) return it->second;
  int id = m->populate(*this);
  matIds_[m.get()] = id;
  if ((int)materialPtrs.size() <= id) materialPtrs.resize(id + 1);
  materialPtrs[id] = m;
  return id;
}

#include "shade_device.h"

#endif
