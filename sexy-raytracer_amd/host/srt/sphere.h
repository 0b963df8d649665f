// sphere.h -- host mirror of sphere.h:8-106 (hit, getSphereUV, calcTangentBasis are device code).
#ifndef SRT_HOST_SPHERE_H
#define SRT_HOST_SPHERE_H

#include "hittable.h"
#include "material.h"

class sphere : public hittable {
 public:
  sphere() : t0(0), t1(0), radius(0) {}
  sphere(vec3f c0, vec3f c1, float time0, float time1, float r, shared_ptr<material> m)
      : center0(c0), center1(c1), t0(time0), t1(time1), radius(r), matPtr(m) {}

  vec3f center(float time) const {  // sphere.h:47-52
    if (center0 != center1) return center0 + ((time - t0) / (t1 - t0)) * (center1 - center0);
    return center0;
  }
  bool boundingBox(float time0, float time1, aabb& outputBox) const override {  // sphere.h:85-94
    vec3f rr(radius, radius, radius);
    outputBox = surroundingBox(aabb(center(time0) - rr, center(time0) + rr), aabb(center(time1) - rr, center(time1) + rr));
    return true;
  }
  int populate(sceneFlattener& f) const override {
    SrtSphereIn s{};
    for (int i = 0; i < 3; ++i) {
      s.center0[i] = center0(i);
      s.center1[i] = center1(i);
    }
    s.time0 = t0; s.time1 = t1; s.radius = radius;
    s.material = f.materialId(matPtr);
    return f.addSphere(s);
  }

 public:
  vec3f center0, center1;
  float t0, t1;
  float radius;
  shared_ptr<material> matPtr;
};

#endif
