// hittable.h -- host mirror of hittable.h:9-33.  The virtual surface kept on the host is what
// scene construction needs: boundingBox (BVH build), populate (the complete populateVector) and
// selectBvhAxis.  The intersection arithmetic lives on the device only; hit() keeps the reference's
// signature (hittable.h:26) so that scene code calling world.hit(r, tMin, tMax, rec) compiles and runs: it
// flattens the object once, uploads it to a private device context and traces the ray through
// srtTraceRays (FAITHFUL traversal: bvh.h:97-105 order, model.h:128).  One device round trip per call --
// for many rays use hipDevice::trace.  There is no host fallback: without a HIP device hit() reports on
// std::cerr and returns false.
#ifndef SRT_HOST_HITTABLE_H
#define SRT_HOST_HITTABLE_H

#include <iostream>
#include <vector>

#include "aabb.h"
#include "flatten.h"
#include "globals.h"
#include "ray.h"

class material;

struct hitRecord {  // hittable.h:9-22
  vec3f p;
  vec3f normal, tangent, bitangent;
  vec2f uv;
  float t = 0;
  bool frontFace = false;
  shared_ptr<material> matPtr;
  inline void setFaceNormal(const ray& r, const vec3f& outwardNormal) {
    frontFace = r.dir.dot(outwardNormal) < 0;
    normal = frontFace ? outwardNormal : -outwardNormal;
  }
};

struct srtHitSession {  // the object, flattened and resident on a device context of its own
  SrtContext* ctx = nullptr;
  std::vector<shared_ptr<material>> materials;
  ~srtHitSession() {
    if (ctx) srtDestroy(ctx);
  }
};

class hittable {
 public:
  virtual ~hittable() {}
  virtual bool boundingBox(float time0, float time1, aabb& outputBox) const = 0;
  virtual int selectBvhAxis() const { return randomInt(0, 2); }  // hittable.h:29
  // the reference's populateVector(hittableVector) (hittable.h:32), made complete: appends this
  // object to the flattener as primitive(s) / world item and returns the first primitive index,
  // or -1 for containers that added world items instead.
  virtual int populate(sceneFlattener& f) const = 0;
  virtual bool isPrimitive() const { return true; }

  // hittable.h:26
  bool hit(const ray& r, float tMin, float tMax, hitRecord& record) const {
    if (!session_) {
      auto s = make_shared<srtHitSession>();
      if (srtCreate(0, &s->ctx) != 0) {
        std::cerr << "ERROR: hittable::hit needs a HIP device (the hot path has no host implementation)\n";
        s->ctx = nullptr;
        return false;
      }
      sceneFlattener f;
      const int first = populate(f);
      if (isPrimitive()) f.world.push_back(SrtWorldItem{SRT_WORLD_PRIM, first, 1, 0.0f, 0.0f, 0, nullptr, 0, 0});
      SrtSceneDesc d = f.desc();
      if (srtUploadScene(s->ctx, &d) != 0) {
        std::cerr << "ERROR: " << srtLastError(s->ctx) << "\n";
        return false;
      }
      s->materials = f.materialPtrs;
      session_ = s;
    }
    SrtRay in{};
    for (int i = 0; i < 3; ++i) {
      in.o[i] = r.o(i);
      in.d[i] = r.dir(i);
    }
    in.time = r.time;
    in.tMin = tMin;
    in.tMax = tMax;
    SrtHit h{};
    if (srtTraceRays(session_->ctx, &in, 1, &h, SRT_TRAVERSE_FAITHFUL) != 0) {
      std::cerr << "ERROR: " << srtLastError(session_->ctx) << "\n";
      return false;
    }
    if (h.prim == SRT_NO_HIT) return false;
    for (int i = 0; i < 3; ++i) {
      record.p(i) = h.p[i];
      record.normal(i) = h.normal[i];
      record.tangent(i) = h.tangent[i];
      record.bitangent(i) = h.bitangent[i];
    }
    record.uv(0) = h.uv[0];
    record.uv(1) = h.uv[1];
    record.t = h.t;
    record.frontFace = h.frontFace != 0;
    record.matPtr = (h.material >= 0 && h.material < (int)session_->materials.size()) ? session_->materials[h.material] : nullptr;
    return true;
  }

 protected:
  // containers call this when their contents change (hittableList::add / clear)
  void dropHitSession() { session_.reset(); }

 private:
  mutable shared_ptr<srtHitSession> session_;
};

#endif
