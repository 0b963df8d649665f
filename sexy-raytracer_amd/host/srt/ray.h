#ifndef SRT_HOST_RAY_H
#define SRT_HOST_RAY_H
#include "vec3.h"
// ray.h:6-23.  Kept for API completeness (fixed-ray-set tracing through hipDevice::trace).
class ray {
 public:
  ray() : time(0) {}
  ray(const vec3f& origin, const vec3f& direction, float t = 0) : o(origin), dir(direction), time(t) {}
  vec3f at(float t) const { return o + t * dir; }
  vec3f o, dir;
  float time;
};
#endif
