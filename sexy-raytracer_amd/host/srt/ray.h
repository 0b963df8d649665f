// ray.h -- host mirror of ray.h:6-25 (same member names: o, dir, time).
#ifndef SRT_HOST_RAY_H
#define SRT_HOST_RAY_H

#include "vec3.h"

class ray {
 public:
  ray() : time(0) {}
  ray(const vec3f& origin, const vec3f& direction, float t = 0) : o(origin), dir(direction), time(t) {}
  vec3f at(float t) const { return o + t * dir; }  // ray.h:15-17

 public:
  vec3f o;
  vec3f dir;
  float time;
};

#endif
