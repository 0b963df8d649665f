// shade_device.h -- material::scatter / material::emitted / texture::value on the host classes, answered by the device.
// The reference evaluates these on the CPU per hit (material.h:91-245, texture.h:26-148); here the arithmetic exists once,
// in the kernels (csrc/srt_path.h shade / texValue).  A call flattens the material (or the texture, wrapped in a
// diffuseLight whose emitted() IS the texture's value, material.h:144-150) into a one-sphere scene on a device context
// of its own, kept for later calls, and asks srtScatterRays for the (ray, hit record) pair.  No host fallback: without a
// HIP device the call reports on std::cerr and returns black / false.  Included at the end of material.h.
#ifndef SRT_HOST_SHADE_DEVICE_H
#define SRT_HOST_SHADE_DEVICE_H

#include "hittable.h"

struct srtShadeSession {
  SrtContext* ctx = nullptr;
  int material = 0;
  uint64_t calls = 0;
  ~srtShadeSession() {
    if (ctx) srtDestroy(ctx);
  }
};

// the scene a shading call runs against: the flattened material and one unit sphere that wears it
inline shared_ptr<srtShadeSession> srtOpenShadeSession(sceneFlattener& f, int materialId) {
  auto s = make_shared<srtShadeSession>();
  if (srtCreate(0, &s->ctx) != 0) {
    std::cerr << "ERROR: material::scatter / emitted / texture::value need a HIP device (no host implementation)\n";
    s->ctx = nullptr;
    return nullptr;
  }
  SrtSphereIn ball{};
  ball.radius = 1.0f;
  ball.time1 = 1.0f;
  ball.material = materialId;
  const int first = f.addSphere(ball);
  f.world.push_back(SrtWorldItem{SRT_WORLD_PRIM, first, 1, 0.0f, 0.0f, 0, nullptr, 0, 0});
  SrtSceneDesc d = f.desc();
  if (srtUploadScene(s->ctx, &d) != 0) {
    std::cerr << "ERROR: " << srtLastError(s->ctx) << "\n";
    return nullptr;
  }
  s->material = materialId;
  return s;
}

inline bool srtShadeOnDevice(srtShadeSession& s, const ray& rIn, const hitRecord& rec, float out13[13]) {
  SrtRay in{};
  SrtHit h{};
  for (int i = 0; i < 3; ++i) {
    in.o[i] = rIn.o(i);
    in.d[i] = rIn.dir(i);
    h.p[i] = rec.p(i);
    h.normal[i] = rec.normal(i);
    h.tangent[i] = rec.tangent(i);
    h.bitangent[i] = rec.bitangent(i);
  }
  in.time = rIn.time;
  h.uv[0] = rec.uv(0);
  h.uv[1] = rec.uv(1);
  h.t = rec.t;
  h.frontFace = rec.frontFace ? 1 : 0;
  h.material = s.material;
  if (srtScatterRays(s.ctx, &in, &h, 1, s.calls++, out13) != 0) {
    std::cerr << "ERROR: " << srtLastError(s.ctx) << "\n";
    return false;
  }
  return true;
}

inline bool material::scatter(const ray& rIn, const hitRecord& record, color3f& attenuation, ray& scatterRay) const {
  if (!shadeSession_) {
    sceneFlattener f;
    const int id = populate(f);
    shadeSession_ = srtOpenShadeSession(f, id);
    if (!shadeSession_) return false;
  }
  float o[13];
  if (!srtShadeOnDevice(*shadeSession_, rIn, record, o)) return false;
  attenuation = color3f(o[0], o[1], o[2]);
  scatterRay = ray(vec3f(o[6], o[7], o[8]), vec3f(o[3], o[4], o[5]), rIn.time);
  return o[9] != 0.0f;
}

inline color3f material::emitted(float u, float v, const vec3f& p) const {
  if (!shadeSession_) {
    sceneFlattener f;
    const int id = populate(f);
    shadeSession_ = srtOpenShadeSession(f, id);
    if (!shadeSession_) return color3f(0, 0, 0);
  }
  hitRecord rec;
  rec.p = p;
  rec.normal = vec3f(0, 1, 0);
  rec.tangent = vec3f(1, 0, 0);
  rec.bitangent = vec3f(0, 0, 1);
  rec.uv = vec2f(u, v);
  rec.t = 1.0f;
  rec.frontFace = true;
  float o[13];
  const uint64_t calls = shadeSession_->calls;  // emitted() draws nothing: leave scatter()'s key sequence alone
  const bool ok = srtShadeOnDevice(*shadeSession_, ray(p + vec3f(0, 1, 0), vec3f(0, -1, 0), 0), rec, o);
  shadeSession_->calls = calls;
  return ok ? color3f(o[10], o[11], o[12]) : color3f(0, 0, 0);
}

inline color3f texture::value(float u, float v, const vec3f& p) const {
  if (!shadeSession_) {
    // a diffuseLight around this texture: its emitted(u, v, p) is emit->value(u, v, p) (material.h:144-150)
    sceneFlattener f;
    SrtMaterialIn m{};
    m.type = SRT_MAT_LIGHT;
    m.albedoTex = populate(f);
    m.normalTex = m.metallicTex = m.roughnessTex = -1;
    f.materials.push_back(m);
    shadeSession_ = srtOpenShadeSession(f, (int)f.materials.size() - 1);
    if (!shadeSession_) return color3f(0, 0, 0);
  }
  hitRecord rec;
  rec.p = p;
  rec.normal = vec3f(0, 1, 0);
  rec.tangent = vec3f(1, 0, 0);
  rec.bitangent = vec3f(0, 0, 1);
  rec.uv = vec2f(u, v);
  rec.t = 1.0f;
  rec.frontFace = true;
  float o[13];
  if (!srtShadeOnDevice(*shadeSession_, ray(p + vec3f(0, 1, 0), vec3f(0, -1, 0), 0), rec, o)) return color3f(0, 0, 0);
  return color3f(o[10], o[11], o[12]);
}

#endif
