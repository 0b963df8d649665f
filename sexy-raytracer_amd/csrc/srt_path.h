// srt_path.h -- device functions of the per-pixel path-tracing loop, shared by the render kernels
// (srt_kernels.hip: the step-scheduler kernels; srt_wavefront.hip: the path-pool kernel): vectors in the reference's
// operation order (vec3.h), the counter RNG, the geometry tests (aabb.h:11-27, sphere.h:54-83, model.h:104-181), hit
// records (hittable.h:9-22), textures (texture.h), the BRDF (pbr.h:58-81), the four material::scatter bodies
// (material.h:91-245), camera::getRay (camera.h:40-46) and the exact chunk sums.  Everything is in an anonymous
// namespace: each translation unit gets its own inlined copy.
#ifndef SRT_PATH_H
#define SRT_PATH_H

#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#include "srt_device.h"

#define SRT_BLOCK 256
#define SRT_BLOCK_TREE 1024      // LDS-resident tree: one workgroup per CU
#define SRT_TREE_WAVES_PER_SIMD 4
#ifndef SRT_NODE_UNROLL
#define SRT_NODE_UNROLL 4  // node visits per evaluation of the burst loop's exit test
#endif
#ifndef SRT_NODE_UNROLL_CLOSEST
#define SRT_NODE_UNROLL_CLOSEST 4  // same for the near-child-first variant (tunable separately)
#endif
#ifndef SRT_PRIM_ROUNDS
#define SRT_PRIM_ROUNDS 2  // primitive tests per scheduling trip (2: the second object of a two-object leaf in the same trip;
                           // headline frame 3392 / 3425 / 3414 Msamples/s with 1 / 2 / 3, profiles/r02/prim_rounds.txt)
#endif
#ifndef SRT_RENDER_WAVES_PER_SIMD
#define SRT_RENDER_WAVES_PER_SIMD 5
#endif

namespace {

// ------------------------------------------------------------------ vectors
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// Eigen 3-vector dot: x*x' + (y*y' + z*z')
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec3.h:29-31: (x*x + y*y) + z*z
__device__ __forceinline__ float lenSq(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
__device__ __forceinline__ float dist3(V3 a, V3 b) { return sqrtf(lenSq(a - b)); }  // vec3.h:37-39
__device__ __forceinline__ V3 unitv(V3 v) {                                           // vec3.h:54-60
  float len = sqrtf(lenSq(v));
  if (len != 0) return mk(v.x / len, v.y / len, v.z / len);
  return v;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) {  // globals.h:17-24
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
__device__ __forceinline__ V3 reflect3(V3 v, V3 n) { return v - (2.0f * dot3(v, n)) * n; }  // vec3.h:76-78

#define SRT_EPS 1.1920928955078125e-07f /* FLT_EPSILON, globals.h:14 */
#define SRT_PI 3.14159274101257324f     /* float(3.1415926535897932385), globals.h:15 */
#define SRT_INF __builtin_inff()

// ------------------------------------------------------------------ RNG
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Pcg {
  uint64_t state;
  __device__ __forceinline__ void key(uint64_t seedMixed, uint32_t pixel, uint32_t sample) {
    state = mix64(seedMixed ^ (((uint64_t)pixel << 32) | (uint64_t)sample));
  }
  __device__ __forceinline__ uint32_t bits() {
    uint64_t old = state;
    state = old * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
  }
  // globals.h:30-35 as libstdc++ evaluates it: float(u32) * 2^-32, clamped below 1
  __device__ __forceinline__ float uniform() {
    float r = (float)bits() * 2.3283064365386963e-10f;
    return r >= 1.0f ? 0x1.fffffep-1f : r;
  }
  __device__ __forceinline__ float uniform(float lo, float hi) { return lo + (hi - lo) * uniform(); }  // globals.h:37-39
  // vec3.h:62-70 with vec3.h:45-47's g++ argument order (z, y, x)
  __device__ __forceinline__ V3 inUnitSphere() {
    while (true) {
      float z = uniform(-1.0f, 1.0f);
      float y = uniform(-1.0f, 1.0f);
      float x = uniform(-1.0f, 1.0f);
      V3 p = mk(x, y, z);
      if (lenSq(p) >= 1.0f) continue;
      return p;
    }
  }
  __device__ __forceinline__ void inUnitDisk(float& x, float& y) {  // vec3.h:88-95 (y first)
    while (true) {
      y = uniform(-1.0f, 1.0f);
      x = uniform(-1.0f, 1.0f);
      if ((x * x + y * y) + 0.0f * 0.0f >= 1.0f) continue;
      return;
    }
  }
};


// ------------------------------------------------------------------ scene fetches
// Raw buffer loads (SRSRC + 32-bit byte offset): one buffer_load_dwordx4 per 16-byte slot, no 64-bit
// address arithmetic, never split or sunk by the compiler, out-of-range reads return 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t makeRsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 bufLoad4(__amdgpu_buffer_rsrc_t r, int byteOffset) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byteOffset, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ------------------------------------------------------------------ per-ray reciprocals and their operand ranges
// The slab test (six divisions by the three ray-direction components per BVH node, aabb.h:14-17) is decided from
// one FMA per plane with a refined reciprocal per axis per ray (boxHitApprox below).  The error bound of that
// decision assumes normal operands: fastDivOperandOk certifies the ranges per ray (and the scene's box
// coordinates at upload); a ray outside them takes the IEEE divisions on every visit.
__device__ __forceinline__ float refinedRcp(float d) {  // within 1.5 ulp of 1/d: v_rcp_f32 + one Newton step
  float r0 = __builtin_amdgcn_rcpf(d);
  float e0 = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
// direction components in [2^-20, 2^20]; origin components 0 or in [2^-77, 2^30]
__device__ __forceinline__ bool fastDivOperandOk(float o, float d) {
  float ad = fabsf(d), ao = fabsf(o);
  // bitwise: no short-circuit branches (each would cost the wave scalar exec-mask bookkeeping)
  return (ad >= 0x1p-20f) & (ad <= 0x1p20f) & ((o == 0.0f) | ((ao >= 0x1p-77f) & (ao <= 0x1p30f)));
}

// ------------------------------------------------------------------ rays, hits
struct Ray {
  V3 o, d;
  float time;
};

struct Counters {
  uint32_t nodeVisits, boxPasses, triTests, sphereTests;
};

// ------------------------------------------------------------------ geometry tests
// aabb.h:11-27.  tMin only grows and tMax only shrinks through the three axes, so the
// per-axis early outs collapse to one final comparison.
__device__ __forceinline__ bool boxHit(float4 n0, float4 n1, const Ray& r, float tMin, float tMax) {
  float a, b;
  a = (n0.x - r.o.x) / r.d.x;
  b = (n1.x - r.o.x) / r.d.x;
  tMin = fmaxf(fminf(a, b), tMin);
  tMax = fminf(fmaxf(a, b), tMax);
  a = (n0.y - r.o.y) / r.d.y;
  b = (n1.y - r.o.y) / r.d.y;
  tMin = fmaxf(fminf(a, b), tMin);
  tMax = fminf(fmaxf(a, b), tMax);
  a = (n0.z - r.o.z) / r.d.z;
  b = (n1.z - r.o.z) / r.d.z;
  tMin = fmaxf(fminf(a, b), tMin);
  tMax = fminf(fmaxf(a, b), tMax);
  return !(tMax <= tMin);
}

// Slab test decided from one-FMA quotients with an error certificate.
// The reference computes a = fl(fl(n - o) / d) per plane (aabb.h:14-17).  With r = refinedRcp(d) (within
// 3u of 1/d, u = 2^-24) and m = fl(-o * r), both per ray, A = fma(n, r, m) satisfies
//   |A - a| <= 6.2u |A| + 1.03u |o/d|            (one rounding each in r, m, the fma and the reference's
//                                                  subtraction and division; operands certified normal)
// i.e. a relative part and an absolute part K <= 2^-23 M, M = max_k |m_k|.  x -> x +- (eps|x| + K) is
// monotone, so min/max carry the bound through: the approximate interval ends tMinA, tMaxA are within
// eps|t| + K of the reference's (tMin and the running closest t are exact), and the reference's decision
// (tMax <= tMin -> miss) is certain whenever |tMaxA - tMinA| > eps (|tMaxA| + |tMinA|) + 2K with
// eps = 2^-21 (1.3x the bound).  tolAbs = 2K = 2^-22 M per ray, or +inf for a ray outside fastDiv's
// operand ranges (every visit of such a ray is "undecided" and takes the IEEE divisions).
// Returns the approximate verdict and whether it is uncertain (the caller then runs the exact test).
// The min/max chain is written with the hardware instructions directly: fminf/fmaxf make the compiler
// quiet possible signalling NaNs first (a `v_max_f32 x, x` per live-in operand per visit), which buys
// nothing here -- v_min/v_max already return the other operand when one is a NaN, and the result only
// feeds the two comparisons of the certificate.
__device__ __forceinline__ float hwMin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hwMax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hwMin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float hwMax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// b: wave-uniform (a kernel argument), read from its scalar register
__device__ __forceinline__ float hwMaxUniform(float a, float b) {
  float r;
  asm("v_max_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
// UNIFORM_TMIN: tMin is wave-uniform (a kernel argument) and is read from its scalar register
template <bool UNIFORM_TMIN>
__device__ __forceinline__ bool boxHitApprox(float4 n0, float4 n1, V3 r1, V3 m, float tolAbs, float tMin, float tMax,
                                             bool& undecided) {
  const float ax = __builtin_fmaf(n0.x, r1.x, m.x), bx = __builtin_fmaf(n1.x, r1.x, m.x);
  const float ay = __builtin_fmaf(n0.y, r1.y, m.y), by = __builtin_fmaf(n1.y, r1.y, m.y);
  const float az = __builtin_fmaf(n0.z, r1.z, m.z), bz = __builtin_fmaf(n1.z, r1.z, m.z);
  const float lo = hwMax3(hwMin(ax, bx), hwMin(ay, by), hwMin(az, bz));
  tMin = UNIFORM_TMIN ? hwMaxUniform(lo, tMin) : hwMax(lo, tMin);
  tMax = hwMin(hwMin3(hwMax(ax, bx), hwMax(ay, by), hwMax(az, bz)), tMax);
  const float diff = tMax - tMin;
  const float tol = __builtin_fmaf(0x1p-21f, fabsf(tMax) + fabsf(tMin), tolAbs);
  undecided = !(fabsf(diff) > tol);  // also when a NaN got in, and always when tolAbs = +inf
  return diff > tol;
}
// Closest-hit traversal: the same one-FMA intervals used CONSERVATIVELY -- a box is entered unless it is certainly
// missed (tMaxA - tMinA < -tol), so no exact fallback is needed: a box entered needlessly costs time, never a hit.
// Returns the approximate entry distance for near-first ordering.
__device__ __forceinline__ bool boxMaybeHit(float4 lo, float4 hi, V3 r1, V3 m, float tolAbs, float tMin, float tMax, float& tEnter) {
  const float ax = __builtin_fmaf(lo.x, r1.x, m.x), bx = __builtin_fmaf(hi.x, r1.x, m.x);
  const float ay = __builtin_fmaf(lo.y, r1.y, m.y), by = __builtin_fmaf(hi.y, r1.y, m.y);
  const float az = __builtin_fmaf(lo.z, r1.z, m.z), bz = __builtin_fmaf(hi.z, r1.z, m.z);
  const float t0 = hwMax(hwMax3(hwMin(ax, bx), hwMin(ay, by), hwMin(az, bz)), tMin);
  const float t1 = hwMin(hwMin3(hwMax(ax, bx), hwMax(ay, by), hwMax(az, bz)), tMax);
  const float tol = __builtin_fmaf(0x1p-21f, fabsf(t1) + fabsf(t0), tolAbs);
  tEnter = t0;
  return !(t1 - t0 < -tol);  // NaN or tolAbs = +inf (uncertified ray): enter
}
// per ray: m = fl(-o * r) and the absolute part of the certificate's tolerance
__device__ __forceinline__ void slabSetup(const V3& o, const V3& r1, bool certified, V3& m, float& tolAbs) {
  m = mk(-(o.x * r1.x), -(o.y * r1.y), -(o.z * r1.z));
  const float M = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fabsf(m.z));
  tolAbs = certified ? 0x1p-22f * M : SRT_INF;
}

// sphere.h:47-52
__device__ __forceinline__ V3 sphereCenter(const float4* sp, float4 s0, float4 s1, float time) {
  V3 c0 = mk(s0.x, s0.y, s0.z);
  if (__float_as_int(s1.w) & (1 << 30)) {
    float4 s2 = sp[2];
    V3 c1 = mk(s1.x, s1.y, s1.z);
    return c0 + ((time - s2.x) / (s2.y - s2.x)) * (c1 - c0);
  }
  return c0;
}

// sphere.h:54-73 on loaded values: c = centre at the ray's time, radius.  Straight-line form of the
// reference's early returns (same comparisons, same NaN behaviour): the lanes of a wave are at
// different spheres/triangles anyway, so early exits only cost scalar exec-mask bookkeeping.
__device__ __forceinline__ bool sphereHitV(V3 center, float radius, const Ray& r, float a, float tMin, float tMax,
                                           float& tOut) {
  V3 oc = r.o - center;
  float halfB = dot3(oc, r.d);
  float c = lenSq(oc) - radius * radius;
  float disc = halfB * halfB - a * c;
  float sqrtd = sqrtf(disc);
  float root1 = (-halfB - sqrtd) / a;
  float root2 = (-halfB + sqrtd) / a;
  const bool out1 = root1 < tMin || root1 > tMax;
  const bool out2 = root2 < tMin || root2 > tMax;
  tOut = out1 ? root2 : root1;
  return !(disc < 0.0f) && !(out1 && out2);
}
__device__ __forceinline__ bool sphereHit(const float4* sp, const Ray& r, float a, float tMin, float tMax, float& tOut) {
  float4 s0 = sp[0], s1 = sp[1];
  return sphereHitV(sphereCenter(sp, s0, s1, r.time), s0.w, r, a, tMin, tMax, tOut);
}

// model.h:104-154.  n is precomputed on the host with the same operation order as
// getNormal (model.h:276-283).  CLOSEST adds the t > tMax rejection the reference lacks.
template <bool CLOSEST>
__device__ __forceinline__ bool triHitV(float4 q0, float4 q1, float4 q2, const Ray& r, float tMin, float tMax,
                                        float& tOut) {
  V3 v0 = mk(q0.x, q0.y, q0.z), v1 = mk(q1.x, q1.y, q1.z), v2 = mk(q2.x, q2.y, q2.z);
  V3 n = mk(q0.w, q1.w, q2.w);
  float NdotDir = dot3(n, r.d);
  // model.h:119-123: parallel, then back-face (dot(dir, n) has the same bits as dot(n, dir))
  bool ok = !(fabsf(NdotDir) < SRT_EPS) && !(NdotDir > 0);
  float d = -dot3(n, v0);
  float t = -(dot3(n, r.o) + d) / NdotDir;
  ok = ok && !(t < tMin);
  if (CLOSEST) ok = ok && !(t > tMax);
  V3 p = r.o + t * r.d;
  // the three inside-edge tests (model.h:135-154); a NaN passes, as in the reference
  ok = ok && !(dot3(n, cross3(v1 - v0, p - v0)) < 0);
  ok = ok && !(dot3(n, cross3(v2 - v1, p - v1)) < 0);
  ok = ok && !(dot3(n, cross3(v0 - v2, p - v2)) < 0);
  tOut = t;
  return ok;
}
template <bool CLOSEST>
__device__ __forceinline__ bool triHit(const float4* tr, const Ray& r, float tMin, float tMax, float& tOut) {
  return triHitV<CLOSEST>(tr[0], tr[1], tr[2], r, tMin, tMax, tOut);
}

// hittableList::hit over the world list (hittablelist.h:33-47) with bvhNode::hit
// (bvh.h:97-105) as an explicit DFS.  In the reference the tMax handed to any node or
// primitive is "the t of the most recent successful primitive hit, else the caller's
// tMax" (by induction over bvh.h:102-103), i.e. one running value `closest`.
// A single-object leaf (left == right, bvh.h:67-69) tests its object twice with
// identical outcome; it is tested once here.
template <bool CLOSEST, bool COUNT>
__device__ __forceinline__ int traverse(const DevScene& sc, const Ray& r, float tMin, float tMax, int32_t* stack,
                                        float& tHit, Counters& cnt) {
  const float a = lenSq(r.d);  // sphere.h:56, per ray
  // the render kernel's slab test: certified one-FMA decision, IEEE divisions when it cannot decide
  const bool certified = sc.fastDivScene != 0 && fastDivOperandOk(r.o.x, r.d.x) && fastDivOperandOk(r.o.y, r.d.y) &&
                         fastDivOperandOk(r.o.z, r.d.z);
  const V3 rcpD = mk(refinedRcp(r.d.x), refinedRcp(r.d.y), refinedRcp(r.d.z));
  V3 negOR;
  float slabTol;
  slabSetup(r.o, rcpD, certified, negOR, slabTol);
  float closest = tMax;
  int hitRef = SRT_REF_DONE;
  for (int w = 0; w < sc.numWorld; ++w) {
    int cur = sc.world[w];
    int sp = 0;
    while (true) {
      while (cur >= 0) {
        const float4* node = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.nodes) + cur);  // byte offset
        float4 n0 = node[0], n1 = node[1];
        if (COUNT) cnt.nodeVisits++;
        bool undecided;
        bool hitBox = boxHitApprox<false>(n0, n1, rcpD, negOR, slabTol, tMin, closest, undecided);
        if (undecided) hitBox = boxHit(n0, n1, r, tMin, closest);
        if (hitBox) {
          if (COUNT) cnt.boxPasses++;
          int left = __float_as_int(n0.w), right = __float_as_int(n1.w);
          if (CLOSEST) {  // near child first (see the render kernel's node step)
            const int axis = sc.nodeAxis[cur >> 5];
            const float dAxis = axis == 0 ? r.d.x : (axis == 1 ? r.d.y : r.d.z);
            if (axis < 3 && dAxis < 0.0f) {
              const int tmp = left;
              left = right;
              right = tmp;
            }
          }
          if (right != left && sp < sc.stackDepth) {  // capacity is guaranteed at upload (see srt_api.cpp Builder)
            stack[sp * SRT_BLOCK] = right;
            sp++;
          }
          cur = left;
        } else {
          if (sp == 0) {
            cur = SRT_REF_DONE;
            break;
          }
          sp--;
          cur = stack[sp * SRT_BLOCK];
        }
      }
      if (cur == SRT_REF_DONE) break;
      int pr = ~cur;
      float t;
      bool ok;
      if (pr & 1) {
        if (COUNT) cnt.sphereTests++;
        ok = sphereHit(sc.spheres + 3 * (pr >> 1), r, a, tMin, closest, t);
      } else {
        if (COUNT) cnt.triTests++;
        ok = triHit<CLOSEST>(sc.triTest + 3 * (pr >> 1), r, tMin, closest, t);
      }
      if (ok) {
        closest = t;
        hitRef = cur;
      }
      if (sp == 0) break;
      sp--;
      cur = stack[sp * SRT_BLOCK];
    }
  }
  tHit = closest;
  return hitRef;
}

// ------------------------------------------------------------------ hit record (hittable.h:9-22)
struct Record {
  V3 p, normal, tangent, bitangent;
  float u, v, t;
  bool frontFace;
  int material;  // index
  int matType;   // SRT_MAT_* | SRT_MAT_TEXTURED: travels in the primitive's material word, so the hit step branches before it loads
  bool isTri;
};

__device__ __forceinline__ void setFaceNormal(Record& rec, const Ray& r, V3 outward) {  // hittable.h:18-21
  rec.frontFace = dot3(r.d, outward) < 0;
  rec.normal = rec.frontFace ? outward : -outward;
}

// The reference fills every field of hitRecord on every hit (sphere.h:74-80, model.h:156-178).
// uv is only ever read by image-texture lookups and the tangent frame only by normal mapping, so
// the render kernel computes them when the hit material can use them (DevMaterial::flags, set at
// upload); `all` forces everything (fixed-ray-set output).
#define SRT_MAT_NEEDS_UV 1
#define SRT_MAT_NEEDS_TANGENT 2
// The two flags ride in the primitive's material word (bits 28-29; srt_api.cpp), so that a hit knows what to compute
// as soon as its shading record has arrived instead of after one more dependent load.

__device__ __forceinline__ void sphereRecord(const DevScene& sc, int idx, const Ray& r, float t, Record& rec, bool all) {
  const float4* sp = sc.spheres + 3 * idx;
  float4 s0 = sp[0], s1 = sp[1];
  rec.t = t;
  rec.p = r.o + t * r.d;                                                  // ray.h:15-17
  V3 outward = unitv(rec.p - sphereCenter(sp, s0, s1, r.time));           // sphere.h:76
  setFaceNormal(rec, r, outward);
  rec.material = __float_as_int(s1.w) & SRT_MAT_INDEX_MASK;
  rec.matType = (__float_as_int(s1.w) >> SRT_MAT_TYPE_SHIFT) & 15;
  rec.isTri = false;
  const int flags = all ? 3 : (__float_as_int(s1.w) >> SRT_MAT_FLAGS_SHIFT) & 3;
  rec.u = rec.v = 0.0f;
  if (flags & SRT_MAT_NEEDS_UV) {
    float theta = acosf(-outward.y);                                      // sphere.h:32-38
    float phi = atan2f(-outward.z, outward.x) + SRT_PI;
    rec.u = phi / (2.0f * SRT_PI);
    rec.v = theta / SRT_PI;
  }
  rec.tangent = rec.bitangent = mk(0.0f, 0.0f, 0.0f);
  if (flags & SRT_MAT_NEEDS_TANGENT) {
    // sphere.h:96-106; dot(n, UnitY) = n.x*0 + (n.y*1 + n.z*0)
    float ny = outward.x * 0.0f + (outward.y * 1.0f + outward.z * 0.0f);
    V3 b = (1.0f - fabsf(ny) < SRT_EPS) ? mk(-0.0f, -0.0f, -1.0f) : mk(0.0f, 1.0f, 0.0f);
    rec.tangent = unitv(cross3(b, outward));
    rec.bitangent = unitv(cross3(outward, rec.tangent));
  }
}

__device__ __forceinline__ void triRecord(const DevScene& sc, int idx, const Ray& r, float t, Record& rec, bool all) {
  const float4* tr = sc.triTest + 3 * idx;
  const float4* sh = sc.triShade + 4 * idx;
  float4 h0 = sh[0], h1 = sh[1], h2 = sh[2], h3 = sh[3];
  V3 p = r.o + t * r.d;
  rec.material = __float_as_int(h3.w) & SRT_MAT_INDEX_MASK;
  rec.matType = (__float_as_int(h3.w) >> SRT_MAT_TYPE_SHIFT) & 15;
  rec.isTri = true;
  rec.u = rec.v = 0.0f;
  if (all || ((__float_as_int(h3.w) >> SRT_MAT_FLAGS_SHIFT) & SRT_MAT_NEEDS_UV)) {
    float4 q0 = tr[0], q1 = tr[1], q2 = tr[2];
    V3 v0 = mk(q0.x, q0.y, q0.z), v1 = mk(q1.x, q1.y, q1.z), v2 = mk(q2.x, q2.y, q2.z);
    // inverse-distance weights (model.h:158-169)
    float d0 = dist3(p, v0), d1 = dist3(p, v1), d2 = dist3(p, v2);
    float denom = (1.0f / d0) + (1.0f / d1) + (1.0f / d2);
    float r0 = (1.0f / d0) / denom, r1 = (1.0f / d1) / denom, r2 = (1.0f / d2) / denom;
    rec.u = r0 * h0.w + r1 * h2.w + r2 * h3.y;
    rec.v = 1.0f - (r0 * h1.w + r1 * h3.x + r2 * h3.z);
  }
  rec.t = t;
  rec.p = p;
  setFaceNormal(rec, r, mk(h0.x, h0.y, h0.z));  // unitVector(normal) precomputed (model.h:172)
  rec.tangent = mk(h1.x, h1.y, h1.z);          // calcTangentBasis precomputed (model.h:214-235)
  rec.bitangent = mk(h2.x, h2.y, h2.z);
}

// ------------------------------------------------------------------ textures (texture.h)
// Texel storage (srt_api.cpp): images of 3 or 4 bytes per pixel are kept as one aligned dword per texel
// (RGB padded to RGBA8), fetched with ONE buffer_load_dword; 1- and 2-byte images keep the reference's
// byte rows, because texture.h:147 reads pixel[1] and pixel[2] of a 1-bpp image from the NEXT texels (and
// past the end of the buffer at the last texel: reads as 0 here and in the oracle -- raw buffer loads
// return 0 out of range).  DevTexture::offset is a byte offset into the texel buffer (< 2^31).
typedef __amdgpu_buffer_rsrc_t Rsrc;
template <bool COUNT>
__device__ __forceinline__ V3 texLeaf(const DevScene& sc, Rsrc rsTexels, int id, float u, float v, uint32_t& fetches) {
  const DevTexture& t = sc.textures[id];
  if (t.kind == SRT_TEX_SOLID) return mk(t.color[0], t.color[1], t.color[2]);  // texture.h:26-28
  // imagePNG::value, texture.h:129-148
  if (t.width == 0) return mk(1.0f, 0.0f, 1.0f);
  if (COUNT) fetches++;
  u = clampf(u, 0.0f, 1.0f);
  v = 1.0f - clampf(v, 0.0f, 1.0f);
  int i = (int)(u * (float)t.width);
  int j = (int)(v * (float)t.height);
  if (!(u == u)) i = 0;  // NaN uv: UB in the reference, defined as texel 0 here and in the oracle
  if (!(v == v)) j = 0;
  if (i >= t.width) i = t.width - 1;
  if (j >= t.height) j = t.height - 1;
  const int texel = j * t.width + i;
  if (t.bpp >= 3) {
    const uint32_t px = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, (int)t.offset + 4 * texel, 0, 0);
    return mk((float)(px & 0xffu), (float)((px >> 8) & 0xffu), (float)((px >> 16) & 0xffu));
  }
  const int at = (int)t.offset + t.bpp * texel;  // bpp == 1: the next two texels (texture.h:147)
  return mk((float)__builtin_amdgcn_raw_buffer_load_b8(rsTexels, at, 0, 0), (float)__builtin_amdgcn_raw_buffer_load_b8(rsTexels, at + 1, 0, 0),
            (float)__builtin_amdgcn_raw_buffer_load_b8(rsTexels, at + 2, 0, 0));
}

// floor(x / pi) and the distance of x / pi to the integers, in float-float arithmetic: q = x * (ih + il)
// with ih + il = 1/pi to 2^-51, carried as an unevaluated sum (two FMAs recover the product's rounding
// error), so that for |q| < 2^22 the fraction is known to 3e-8.  Returns false when x / pi is within 1e-6
// of an integer, zero, huge or not finite.
__device__ __forceinline__ bool piPeriods(float x, int& periods) {
  const float ih = 0x1.45f306p-2f, il = 0x1.b9391p-27f;
  const float qh = x * ih;
  const float ql = __builtin_fmaf(x, il, __builtin_fmaf(x, ih, -qh));
  float f = floorf(qh);
  float r = (qh - f) + ql;  // qh - f is exact
  const bool below = r < 0.0f, above = r >= 1.0f;
  f = below ? f - 1.0f : (above ? f + 1.0f : f);
  r = below ? r + 1.0f : (above ? r - 1.0f : r);
  periods = (int)f;
  return r > 1e-6f && r < 1.0f - 1e-6f && fabsf(qh) < 4194304.0f;
}

// checker::value's choice (texture.h:42-48): is sinf(10x) * sinf(10y) * sinf(10z) negative (the odd texture) at p?
// Only the sign of the product is used.  sin(a) is negative exactly when floor(a/pi) is odd, and with a/pi at least
// 1e-6 away from the integers every sine is at least 3e-6 in magnitude: sinf cannot lose its sign and the product
// cannot underflow, so three range reductions replace three sinf calls.  Arguments that close to a multiple of pi,
// zero, huge and non-finite ones take the reference's expression.
__device__ __forceinline__ bool checkerOdd(V3 p) {
  const float ax = 10.0f * p.x, ay = 10.0f * p.y, az = 10.0f * p.z;
  int kx, ky, kz;
  const bool cx = piPeriods(ax, kx), cy = piPeriods(ay, ky), cz = piPeriods(az, kz);
  if (cx && cy && cz) return ((kx ^ ky ^ kz) & 1) != 0;
  const float sines = sinf(ax) * sinf(ay) * sinf(az);
  return sines < 0;
}

template <bool COUNT>
__device__ __forceinline__ V3 texValue(const DevScene& sc, Rsrc rsTexels, int id, float u, float v, V3 p, uint32_t& fetches) {
  const DevTexture& t = sc.textures[id];
  if (t.kind == SRT_TEX_CHECKER) {  // texture.h:42-48
    const int child = checkerOdd(p) ? t.odd : t.even;
    return texLeaf<COUNT>(sc, rsTexels, child, u, v, fetches) * 255.0f;
  }
  return texLeaf<COUNT>(sc, rsTexels, id, u, v, fetches);
}

// ------------------------------------------------------------------ pbr.h
__device__ __forceinline__ float trowbridgeReitzNDF(float NdotH, float roughness) {  // pbr.h:58-65
  float alpha = roughness * roughness;
  float alpha2 = alpha * alpha;
  float NdotH2 = NdotH * NdotH;
  float b = NdotH2 * (alpha2 - 1.0f) + 1.0f;
  float denom = SRT_PI * (b * b);  // std::pow(b, 2.0f) == b*b correctly rounded
  return alpha2 / denom;
}
__device__ __forceinline__ float schlickGAF(float NdotV, float roughness) {  // pbr.h:69-73
  float k = ((roughness + 1.0f) * (roughness + 1.0f)) / 8.0f;
  return NdotV / (NdotV * (1.0f - k) + k);
}

// ------------------------------------------------------------------ materials
// One 128-byte record per material (DevScene::shadeRecs, srt_api.cpp): the material's scalars and its texture slots
// resolved to what a lookup needs, so that a hit issues ALL its descriptor loads at once and then ALL its texel
// loads at once -- two round trips where material -> texture descriptor -> texel, slot after slot, made up to nine.
//   +0   type, flags, metalness (fuzz / ir), roughness          +16  albedo
//   +32  the emit texture of a light, in full: (mode, width | r, height | g, texel byte offset | b)
//   +48  pbr: albedo and normal slots, two dwords each: (mode | width << 2 | height << 17, texel byte offset | id)
//   +64  pbr: metallic and roughness slots, likewise
//   +80  +96  an albedo slot that is a checker of two solid colours (the ground of every main.cpp scene): the even and
//        the odd colour, and bit 2 of the slot's first word set -- the hit reads them with the rest of the record instead of
//        walking texture -> checker -> solidColor descriptors one dependent load after the other
// Slot modes: 0 no texture, 1 solid colour (also the magenta of a failed load; lights only), 2 image of >= 3 bytes per
// pixel with both sides below 2^15 (one dword per texel), 3 anything else (checker, solid colour in a pbr slot, 1- and
// 2-byte images): texValue on the texture id.
#define SRT_SLOT_NONE 0u
#define SRT_SLOT_SOLID 1u
#define SRT_SLOT_IMAGE 2u
#define SRT_SLOT_GENERIC 3u
#define SRT_SLOT_CHECKER2 7u  // SRT_SLOT_GENERIC | 4: albedo slot only, colours at +80 / +96 of the record
// byte offset of the texel an image lookup reads (imagePNG::value, texture.h:129-146)
__device__ __forceinline__ int texelOffset(int width, int height, int base, float u, float v) {
  u = clampf(u, 0.0f, 1.0f);
  v = 1.0f - clampf(v, 0.0f, 1.0f);
  int i = (int)(u * (float)width);
  int j = (int)(v * (float)height);
  if (!(u == u)) i = 0;  // NaN uv: UB in the reference, defined as texel 0 here and in the oracle
  if (!(v == v)) j = 0;
  if (i >= width) i = width - 1;
  if (j >= height) j = height - 1;
  return base + 4 * (j * width + i);
}
// a pbr slot (two dwords): where its texel is, or 0 (the load is made and ignored) when it is not an image
__device__ __forceinline__ int slotTexelOffset(uint32_t mw, uint32_t aux, float u, float v) {
  const int off = texelOffset((int)((mw >> 2) & 0x7fffu), (int)(mw >> 17), (int)aux, u, v);
  return (mw & 3u) == SRT_SLOT_IMAGE ? off : 0;
}
template <bool COUNT>
__device__ __forceinline__ V3 slotValue(const DevScene& sc, Rsrc rsTexels, uint32_t mw, uint32_t aux, uint32_t px, float u, float v, V3 p,
                                        uint32_t& fetches) {
  if ((mw & 3u) != SRT_SLOT_IMAGE) return texValue<COUNT>(sc, rsTexels, (int)aux, u, v, p, fetches);  // SRT_SLOT_GENERIC
  if (COUNT) fetches++;
  return mk((float)(px & 0xffu), (float)((px >> 8) & 0xffu), (float)((px >> 16) & 0xffu));
}

// returns false when the path ends here (scatter == false); emitted is always set.
// WIDE: all four texel loads of a pbr hit in flight at once (the 128-register kernel); otherwise one lookup after the
// other, which keeps four registers live instead of sixteen (the 96-register kernels spill as it is).
template <bool COUNT, bool WIDE = false>
__device__ __forceinline__ bool shade(const DevScene& sc, Rsrc rsTexels, const Ray& rIn, const Record& rec, Pcg& rng, V3& att,
                                      Ray& out, V3& emitted, uint32_t& fetches, unsigned long long* stamp = nullptr) {
  const Rsrc rsMat = makeRsrc(sc.shadeRecs, sc.numMaterials * 128);
  const int at = rec.material * 128;
  emitted = mk(0.0f, 0.0f, 0.0f);  // material.h:18-20
  out.o = rec.p;
  out.time = rIn.time;
  if ((rec.matType & 3) == SRT_MAT_LIGHT) {  // material.h:144-150
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsMat, at + 32, 0, 0);
    if (t.x == SRT_SLOT_IMAGE) {
      const uint32_t px = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, texelOffset((int)t.y, (int)t.z, (int)t.w, rec.u, rec.v), 0, 0);
      if (COUNT) fetches++;
      emitted = mk((float)(px & 0xffu), (float)((px >> 8) & 0xffu), (float)((px >> 16) & 0xffu));
    } else if (t.x == SRT_SLOT_SOLID) {
      emitted = mk(__uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));  // texture.h:26-28
    } else {
      emitted = texValue<COUNT>(sc, rsTexels, (int)t.y, rec.u, rec.v, rec.p, fetches);
    }
    return false;
  }
  const u32x4 head = __builtin_amdgcn_raw_buffer_load_b128(rsMat, at, 0, 0);  // type, flags, metalness (fuzz / ir), roughness
  const float4 albedo = bufLoad4(rsMat, at + 16);
  const float metalness = __uint_as_float(head.z), roughness = __uint_as_float(head.w);
  switch (rec.matType & 3) {
    case SRT_MAT_METAL: {  // material.h:91-97
      V3 reflected = reflect3(unitv(rIn.d), rec.normal);
      V3 fz = rng.inUnitSphere();  // drawn even when fuzz == 0
      out.d = reflected + metalness * fz;
      att = mk(albedo.x, albedo.y, albedo.z);
      return dot3(out.d, rec.normal) > 0;
    }
    case SRT_MAT_DIELECTRIC: {  // material.h:108-136
      att = mk(1.0f, 1.0f, 1.0f);
      float ir = metalness;
      float ratio = rec.frontFace ? (1.0f / ir) : ir;
      V3 unitDir = unitv(rIn.d);
      float cosTheta = fminf(dot3(rec.normal, -unitDir), 1.0f);  // double fmin of floats is exact
      float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
      bool reflectIt = ratio * sinTheta > 1.0f;
      if (!reflectIt) {
        // reflectance (material.h:132-136): double pow(x, 5), expression in double, narrowed
        float r0 = (1.0f - ratio) / (1.0f + ratio);
        r0 = r0 * r0;
        double x = (double)(1.0f - cosTheta);
        double x2 = x * x;
        double x5 = x2 * x2 * x;
        float refl = (float)((double)r0 + (double)(1.0f - r0) * x5);
        reflectIt = refl > rng.uniform();
      }
      if (reflectIt) {
        out.d = reflect3(unitDir, rec.normal);
      } else {  // refract, vec3.h:80-86
        float ct = fminf(dot3(rec.normal, -unitDir), 1.0f);
        V3 perp = ratio * (unitDir + ct * rec.normal);
        V3 par = (-sqrtf(fabsf(1.0f - lenSq(perp)))) * rec.normal;
        out.d = perp + par;
      }
      return true;
    }
    default: {  // pbrMetallicRoughness::scatter, material.h:156-245
      V3 a0 = mk(albedo.x, albedo.y, albedo.z), normal = rec.normal;
      float mt = metalness, rg = roughness;
      if (rec.matType & SRT_MAT_TEXTURED) {  // some texture slot is in use: skipped by waves whose hits have none
        // WIDE: the whole record in one round trip (slots, and the two colours of a checker-of-solids albedo), then the
        // four texel loads together when any slot is an image (an absent or non-image slot reads texel 0 of the buffer and
        // ignores it); otherwise each load is made where its value is used
        const u32x4 tAN = __builtin_amdgcn_raw_buffer_load_b128(rsMat, at + 48, 0, 0), tMR = __builtin_amdgcn_raw_buffer_load_b128(rsMat, at + 64, 0, 0);
        float4 even = make_float4(0.0f, 0.0f, 0.0f, 0.0f), odd = even;
        if (WIDE) {
          even = bufLoad4(rsMat, at + 80);
          odd = bufLoad4(rsMat, at + 96);
        }
        uint32_t pA = 0, pN = 0, pM = 0, pR = 0;
        const bool anyImage = (tAN.x & 3u) == SRT_SLOT_IMAGE || (tAN.z & 3u) == SRT_SLOT_IMAGE || (tMR.x & 3u) == SRT_SLOT_IMAGE || (tMR.z & 3u) == SRT_SLOT_IMAGE;
        if (WIDE && anyImage) {
          pA = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, slotTexelOffset(tAN.x, tAN.y, rec.u, rec.v), 0, 0);
          pN = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, slotTexelOffset(tAN.z, tAN.w, rec.u, rec.v), 0, 0);
          pM = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, slotTexelOffset(tMR.x, tMR.y, rec.u, rec.v), 0, 0);
          pR = __builtin_amdgcn_raw_buffer_load_b32(rsTexels, slotTexelOffset(tMR.z, tMR.w, rec.u, rec.v), 0, 0);
        }
        auto fetch = [&](uint32_t mw, uint32_t aux, uint32_t early) {
          return WIDE || (mw & 3u) != SRT_SLOT_IMAGE ? early : __builtin_amdgcn_raw_buffer_load_b32(rsTexels, slotTexelOffset(mw, aux, rec.u, rec.v), 0, 0);
        };
        if ((tAN.x & 7u) == SRT_SLOT_CHECKER2) {
          // checker(solidColor, solidColor)::value = the chosen colour times 255 (texture.h:45-47), then material.h:165-166
          if (!WIDE) {
            even = bufLoad4(rsMat, at + 80);
            odd = bufLoad4(rsMat, at + 96);
          }
          const float4 c = checkerOdd(rec.p) ? odd : even;
          a0 = mk(c.x * 255.0f, c.y * 255.0f, c.z * 255.0f) / 255.0f;
        } else if ((tAN.x & 3u) != SRT_SLOT_NONE)
          a0 = slotValue<COUNT>(sc, rsTexels, tAN.x, tAN.y, fetch(tAN.x, tAN.y, pA), rec.u, rec.v, rec.p, fetches) / 255.0f;
        if ((tAN.z & 3u) != SRT_SLOT_NONE) {
          V3 nt = slotValue<COUNT>(sc, rsTexels, tAN.z, tAN.w, fetch(tAN.z, tAN.w, pN), rec.u, rec.v, rec.p, fetches);
          nt = mk(nt.x - 128.0f, nt.y - 128.0f, nt.z - 128.0f) / 128.0f;  // vec3.h:103-110
          // Matrix3f(T|B|N) * nt, each row reduced x + (y + z)
          V3 w = mk(rec.tangent.x * nt.x + (rec.bitangent.x * nt.y + rec.normal.x * nt.z),
                    rec.tangent.y * nt.x + (rec.bitangent.y * nt.y + rec.normal.y * nt.z),
                    rec.tangent.z * nt.x + (rec.bitangent.z * nt.y + rec.normal.z * nt.z));
          normal = unitv(w);
        }
        if ((tMR.x & 3u) != SRT_SLOT_NONE)
          mt = clampf(slotValue<COUNT>(sc, rsTexels, tMR.x, tMR.y, fetch(tMR.x, tMR.y, pM), rec.u, rec.v, rec.p, fetches).x / 255.0f, 0.0f, 1.0f);
        if ((tMR.z & 3u) != SRT_SLOT_NONE)
          rg = clampf(slotValue<COUNT>(sc, rsTexels, tMR.z, tMR.w, fetch(tMR.z, tMR.w, pR), rec.u, rec.v, rec.p, fetches).y / 255.0f, 0.0f, 1.0f);
      }

      if (COUNT && stamp) stamp[0] = clock64();  // textures done
      V3 sd = normal + unitv(rng.inUnitSphere());  // randomUnitVector, vec3.h:72-74
      if (COUNT && stamp) stamp[1] = clock64();  // direction drawn
      // nearZero (vec3.h:49-52): float |x| compared against the DOUBLE 1e-8
      if ((double)fabsf(sd.x) < 1e-8 && (double)fabsf(sd.y) < 1e-8 && (double)fabsf(sd.z) < 1e-8) sd = normal;
      sd = unitv(sd);
      out.d = sd;
      V3 viewVec = -unitv(rIn.d);
      V3 halfVec = unitv(sd + viewVec);
      float NdotL = fmaxf(dot3(normal, sd), 0.0f);
      float NdotH = fmaxf(dot3(normal, halfVec), 0.0f);
      float HdotV = fmaxf(dot3(halfVec, viewVec), 0.0f);
      float NdotV = fmaxf(dot3(normal, viewVec), 0.0f);
      V3 fr = mk(albedo.x, albedo.y, albedo.z);
      // lerp(0.4, fr, mt), vec3.h:97-101.  F0 for dielectrics is 0.4 (material.h:228)
      V3 F0 = mk((1.0f - mt) * 0.4f + mt * fr.x, (1.0f - mt) * 0.4f + mt * fr.y, (1.0f - mt) * 0.4f + mt * fr.z);
      float D = trowbridgeReitzNDF(NdotH, rg);
      // fresnelEpic (pbr.h:75-81): pow(2.0f, x) resolves to the double pow, narrowed to float
      float power = (float)exp2((double)((-5.55473f * HdotV - 6.98316f) * HdotV));
      V3 F = mk(F0.x + (1.0f - F0.x) * power, F0.y + (1.0f - F0.y) * power, F0.z + (1.0f - F0.z) * power);
      float G = schlickGAF(NdotL, rg) * schlickGAF(NdotV, rg);
      V3 fd = a0 / SRT_PI;
      fd = mk(fd.x * (1.0f - F.x), fd.y * (1.0f - F.y), fd.z * (1.0f - F.z));
      fd = fd * (1.0f - mt);
      fd = mk(fd.x * albedo.x, fd.y * albedo.y, fd.z * albedo.z);
      V3 fs = ((D * F) * G) / (4.0f * NdotV * NdotL + SRT_EPS);
      att = (fd + fs) * NdotL;
      return true;
    }
  }
}

// The camera as the kernel argument segment holds it, read where it is used (scalar loads of 27 words) instead of at the
// kernel's entry: the render kernels are one big loop, values loaded up front stay live through all of it, and the scalar
// registers they take are spilled into vector-register lanes and read back lane by lane -- on the vector ALU -- at every
// restart step.  `RenderArgs` must be the kernel's only argument (offset 0 of the segment).
__device__ __forceinline__ DevCamera cameraFromKernarg() {
  DevCamera c;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const unsigned char ConstByte;
  typedef __attribute__((address_space(4))) const float ConstFloat;
  ConstByte* p = (ConstByte*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));  // not hoisted out of the loop it is called in
  ConstFloat* f = (ConstFloat*)(p + offsetof(RenderArgs, cam));
  static_assert(sizeof(DevCamera) % sizeof(float) == 0, "DevCamera is all floats");
  float* out = reinterpret_cast<float*>(&c);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(DevCamera) / sizeof(float)); ++k) out[k] = f[k];
#else
  c = DevCamera();
#endif
  return c;
}

// camera::getRay, camera.h:40-46
__device__ __forceinline__ void cameraRay(const DevCamera& c, float s, float t, Pcg& rng, Ray& r) {
  float dx, dy;
  rng.inUnitDisk(dx, dy);
  // rd = lensRadius * p;  offset = rd.x * hor + rd.y * vert
  float rx = c.lensRadius * dx, ry = c.lensRadius * dy;
  V3 offset = rx * ld3(c.hor) + ry * ld3(c.vert);
  V3 origin = ld3(c.origin);
  r.o = origin + offset;
  r.d = ld3(c.lleft) + s * ld3(c.horizontal) + t * ld3(c.vertical) - origin - offset;
  r.time = rng.uniform(c.time0, c.time1);
}

// Chunk sums of a pixel (sppChunks > 1) are added EXACTLY: each item's float partial sum is converted to
// 64-bit fixed point (units of 2^-36), the integers are added, and the sum is rounded to float once.  Integer
// addition is associative, so the pixel sum does not depend on the order in which chunks finish (items of a
// pixel run on different waves), on the tile split or on the GPU count.  Two implementations of the same sum:
//   * scratch path: an item stores its float4 partial in slot [chunk][tile][pixel]; srt_sum_chunks_kernel
//     converts and adds the slots (one plain store per item in the render kernel, chunks x 16 B per pixel);
//   * atomic path (commitFixed): items add their fixed-point partials with three 64-bit integer atomics into
//     32 B per pixel; srt_finalize_kernel rounds.  Memory is O(pixels) for any chunk count, at the price of the
//     memory-side atomics (scattered 8-byte atomics retire at ~20 G/s chip-wide: 0.4-1 % on the 720p headline
//     with 64-256 chunks, a third of the time of a 240p frame with one-sample items).
// srtRenderTiles picks the scratch path while the rank's slots fit a memory budget, the atomic path beyond; the
// image is the same bit for bit.
// A float of 2^-12 or more converts exactly (its ulp is >= 2^-36; smaller ones are truncated to 2^-36
// absolute), so two chunk sums a, b >= 2^-12 give fl(a + b) bit for bit.  NaN / infinite partial sums (the
// r = 0 ground BRDF produces NaN samples) poison the channel as they would a float sum.
// Range: a partial sum of `limit` or more counts as infinite, limit = 2^26 / (sppChunks rounded up to a power of two)
// (RenderArgs::fixLimit: 2^16 at the 640-chunk cap), so that the 64-bit sum of a pixel's chunks stays below 2^62 units and
// cannot wrap.  Such a pixel resolves to white either way: a chunk of ~8 samples summing to 65 536 is a mean radiance of
// thousands.  Returns false for a value that is not representable (NaN / infinite / at or beyond the limit).
__device__ __forceinline__ bool toFixed36(float v, float limit, long long& q) {
  const float av = fabsf(v);
  if (!(av < limit)) return false;
  // |v| = hi + fr with hi = trunc(|v|) < 2^26 and fr in [0, 1), both exact; fixed = hi * 2^36 + trunc(fr * 2^36)
  const uint32_t hi = (uint32_t)av;
  const float fr = av - (float)hi;
  const uint32_t frHi = (uint32_t)(fr * 0x1p32f);  // top 32 bits of the fraction
  const float rest = fr * 0x1p32f - (float)frHi;   // exact: what is left below 2^-32, in [0, 1)
  const uint32_t frLo = (uint32_t)(rest * 16.0f);  // 4 more bits
  const unsigned long long m = ((unsigned long long)hi << 36) + ((unsigned long long)frHi << 4) + frLo;
  q = (long long)(v < 0.0f ? 0ull - m : m);
  return true;
}
// flags: bit k: NaN in channel k; bit 3+k: +inf; bit 6+k: -inf
__device__ __forceinline__ uint32_t nonFiniteFlag(float v, int k) { return (v != v) ? (1u << k) : (v > 0.0f ? (8u << k) : (64u << k)); }
__device__ __forceinline__ float fromFixed36(long long q, uint32_t flags, int k) {
  const bool nan = (flags >> k) & 1u, pinf = (flags >> (3 + k)) & 1u, ninf = (flags >> (6 + k)) & 1u;
  if (nan || (pinf && ninf)) return __builtin_nanf("");
  if (pinf) return SRT_INF;
  if (ninf) return -SRT_INF;
  return (float)((double)q * 0x1p-36);
}
__device__ __forceinline__ void commitFixed(SrtFixedAccum* f, V3 acc, float limit) {
  const float c[3] = {acc.x, acc.y, acc.z};
  long long* const ch = &f->r;
  uint32_t flags = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    long long q;
    if (!toFixed36(c[k], limit, q))
      flags |= nonFiniteFlag(c[k], k);
    else if (q != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(ch + k), (unsigned long long)q);
  }
  if (flags) atomicOr(&f->flags, flags);
}

}  // namespace

#endif
