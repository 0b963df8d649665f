#ifndef SRT_HOST_AABB_H
#define SRT_HOST_AABB_H
#include <cmath>
#include "vec3.h"
// aabb.h:6-43 (the slab test aabb::hit is device code: srt_kernels.hip boxHit)
class aabb {
 public:
  aabb() {}
  aabb(const vec3f& a, const vec3f& b) : minimum(a), maximum(b) {}
  vec3f minimum, maximum;
};
inline aabb surroundingBox(aabb box0, aabb box1) {
  vec3f small(fminf(box0.minimum(0), box1.minimum(0)), fminf(box0.minimum(1), box1.minimum(1)), fminf(box0.minimum(2), box1.minimum(2)));
  vec3f large(fmaxf(box0.maximum(0), box1.maximum(0)), fmaxf(box0.maximum(1), box1.maximum(1)), fmaxf(box0.maximum(2), box1.maximum(2)));
  return aabb(small, large);
}
#endif
