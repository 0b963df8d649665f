// Axis-aligned box of the host scene description (role of the reference's aabb.h:6-43).
// Only the data and the union are needed on the host; the slab test is device code
// (srt_kernels.hip boxHit / boxHitApprox).
#ifndef SRT_HOST_AABB_H
#define SRT_HOST_AABB_H

#include <cmath>

#include "vec3.h"

struct aabb {
  vec3f minimum, maximum;
  aabb() = default;
  aabb(const vec3f& lo, const vec3f& hi) : minimum(lo), maximum(hi) {}
};

// union of two boxes, component by component with fminf / fmaxf (aabb.h:33-43)
inline aabb surroundingBox(const aabb& p, const aabb& q) {
  aabb u;
  for (int k = 0; k < 3; ++k) {
    u.minimum(k) = fminf(p.minimum(k), q.minimum(k));
    u.maximum(k) = fmaxf(p.maximum(k), q.maximum(k));
  }
  return u;
}

#endif
