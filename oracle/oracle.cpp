// oracle.cpp -- CPU restatement of the reference's per-pixel path-tracing loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (sexy-raytracer_amd/,
// include/) links, loads or calls this file; only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg do, as the checker / reported CPU baseline.
//
// PARITY PIN STATUS: the reference (swishersnaaake/sexy-raytracer) cannot be
// built in this image: vec3.h:7, camera.h:6 need Eigen, texture.h:6-7 need stb,
// model.h:4 needs cgltf, gl.h:9-10 need glad+glfw, all of which are empty,
// un-pinned git submodules (.gitmodules:1-15) and absent from the container.
// The reference has no tests, golden vectors or fixtures (CMakeLists.txt:13-14
// is empty CTest boilerplate).  This restatement is therefore pinned only by
//   (1) libstdc++'s mt19937/uniform_real_distribution<float> known answers
//       (the RNG the reference uses, globals.h:30-35),
//   (2) the BVH statistics of the main.cpp scene recorded in SURVEY.md A.5
//       (3046 prims -> 4043 nodes, 998 single / 1024 double leaves, depth 11),
//   (3) region means of the two published renders images/test-*.png
//       (statistical, tests/test_oracle_published.py).
// Eigen's reduction order in dot()/Matrix3f*vec (a0b0 + (a1b1 + a2b2)) is taken
// from Eigen's unrolled redux and is "parity unpinned" (no Eigen checkout).
//
// Every function cites the reference file:line it restates.  Arithmetic is
// IEEE binary32 with no FMA contraction (build with -ffp-contract=off, no
// -march) except the double-precision detours the reference takes.
//
// RNG: two modes behind one draw() call.
//   MT      one default-seeded std::mt19937 shared by BVH build and render,
//           consumed serially in y,x,s order with GCC's right-to-left argument
//           evaluation made explicit (vec3.h:42,46,90) -- the reference as built
//           with g++.
//   COUNTER PCG32 keyed by (seed, pixel, sample); identical to the HIP kernel's.

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <vector>

#include "../include/srt_hip.h"

namespace orc {

static const float infinity = std::numeric_limits<float>::infinity();  // globals.h:13
static const float epsilon = std::numeric_limits<float>::epsilon();    // globals.h:14
static const float pi = 3.1415926535897932385f;                        // globals.h:15

// ---------------------------------------------------------------- vec3 (vec3.h)
struct vec3 {
  float e[3];
  vec3() : e{0, 0, 0} {}
  vec3(float a, float b, float c) : e{a, b, c} {}
  float operator()(int i) const { return e[i]; }
  float& operator()(int i) { return e[i]; }
};
static inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a(0) + b(0), a(1) + b(1), a(2) + b(2)); }
static inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a(0) - b(0), a(1) - b(1), a(2) - b(2)); }
static inline vec3 operator-(const vec3& a) { return vec3(-a(0), -a(1), -a(2)); }
static inline vec3 operator*(float s, const vec3& a) { return vec3(s * a(0), s * a(1), s * a(2)); }
static inline vec3 operator*(const vec3& a, float s) { return vec3(a(0) * s, a(1) * s, a(2) * s); }
static inline vec3 operator/(const vec3& a, float s) { return vec3(a(0) / s, a(1) / s, a(2) / s); }
static inline bool operator!=(const vec3& a, const vec3& b) { return a(0) != b(0) || a(1) != b(1) || a(2) != b(2); }
// Eigen dot on a fixed 3-vector: unrolled redux splits [0,1) and [1,3).
static inline float dot(const vec3& a, const vec3& b) { return a(0) * b(0) + (a(1) * b(1) + a(2) * b(2)); }
static inline vec3 cross(const vec3& a, const vec3& b) {
  return vec3(a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0));
}
// vec3.h:29-31 (left to right)
static inline float lengthSquared(const vec3& v) { return v(0) * v(0) + v(1) * v(1) + v(2) * v(2); }
static inline float length(const vec3& v) { return sqrtf(lengthSquared(v)); }            // vec3.h:33-35
static inline float distance(const vec3& u, const vec3& v) { return sqrtf(lengthSquared(u - v)); }  // vec3.h:37-39
static inline vec3 unitVector(const vec3& v) {  // vec3.h:54-60
  float len = length(v);
  if (len != 0) return vec3(v(0) / len, v(1) / len, v(2) / len);
  return v;
}
static inline bool nearZero(const vec3& v) {  // vec3.h:49-52: float fabs compared with DOUBLE 1e-8
  const double s = 1e-8;
  return ((double)fabsf(v(0)) < s) && ((double)fabsf(v(1)) < s) && ((double)fabsf(v(2)) < s);
}
static inline vec3 reflect(const vec3& v, const vec3& n) { return v - (2.0f * dot(v, n)) * n; }  // vec3.h:76-78
static inline vec3 refract(const vec3& uv, const vec3& n, float eta) {                            // vec3.h:80-86
  double cosTheta = fmin((double)dot(n, -uv), (double)1.0f);  // double fmin; value is an exact float
  vec3 rOutPerp = eta * (uv + (float)cosTheta * n);           // Eigen casts the double scalar to float
  vec3 rOutParallel = (-sqrtf(fabsf(1.0f - lengthSquared(rOutPerp)))) * n;
  return rOutPerp + rOutParallel;
}
static inline vec3 lerp(const vec3& a, const vec3& b, float t) {  // vec3.h:97-101
  return vec3((1.0f - t) * a(0) + t * b(0), (1.0f - t) * a(1) + t * b(1), (1.0f - t) * a(2) + t * b(2));
}
static inline vec3 normalIntToFloat(const vec3& n) {  // vec3.h:103-110
  vec3 r(n(0) - 128.0f, n(1) - 128.0f, n(2) - 128.0f);
  return r / 128.0f;
}
static inline float clampf(float x, float lo, float hi) {  // globals.h:17-24 (NaN falls through)
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}

// ---------------------------------------------------------------- RNG (globals.h:30-43)
enum { RNG_MT = 0, RNG_COUNTER = 1 };

static inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Rng {
  int mode = RNG_MT;
  std::mt19937* mt = nullptr;
  uint64_t state = 0;
  uint64_t draws = 0;
  void key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    state = mix64(mix64(seed) ^ (((uint64_t)pixel << 32) | (uint64_t)sample));
  }
  uint32_t bits() {
    if (mode == RNG_MT) return (uint32_t)(*mt)();
    uint64_t old = state;  // PCG32 XSH-RR, fixed increment
    state = old * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
  }
  // libstdc++ generate_canonical<float,24>(mt19937) + uniform_real_distribution(0,1):
  // float(u32) / 2^32, clamped below 1 (random.tcc:3348-3380).
  float randomFloat() {
    draws++;
    float r = (float)bits() / 4294967296.0f;
    if (r >= 1.0f) r = 0x1.fffffep-1f;
    return r;
  }
  float randomFloat(float lo, float hi) { return lo + (hi - lo) * randomFloat(); }  // globals.h:37-39
  int randomInt(int lo, int hi) { return (int)randomFloat((float)lo, (float)(hi + 1)); }  // globals.h:41-43
  // vec3.h:45-47 built with g++: arguments evaluated right to left (z, y, x).
  vec3 randomVec3f(float lo, float hi) {
    float z = randomFloat(lo, hi);
    float y = randomFloat(lo, hi);
    float x = randomFloat(lo, hi);
    return vec3(x, y, z);
  }
  vec3 randomInUnitSphere() {  // vec3.h:62-70
    while (true) {
      vec3 p = randomVec3f(-1.0f, 1.0f);
      if (lengthSquared(p) >= 1.0f) continue;
      return p;
    }
  }
  vec3 randomUnitVector() { return unitVector(randomInUnitSphere()); }  // vec3.h:72-74
  vec3 randomInUnitDisk() {                                            // vec3.h:88-95 (y drawn first)
    while (true) {
      float y = randomFloat(-1.0f, 1.0f);
      float x = randomFloat(-1.0f, 1.0f);
      vec3 p(x, y, 0);
      if (lengthSquared(p) >= 1.0f) continue;
      return p;
    }
  }
};

// ---------------------------------------------------------------- ray.h
struct ray {
  vec3 o, dir;
  float time = 0;
  ray() {}
  ray(const vec3& o_, const vec3& d_, float t_ = 0) : o(o_), dir(d_), time(t_) {}
  vec3 at(float t) const { return o + t * dir; }  // ray.h:15-17
};

// ---------------------------------------------------------------- counters
struct Counters {
  uint64_t samples = 0, rays = 0, nodeVisits = 0, boxPasses = 0;
  uint64_t triCalls = 0, sphereCalls = 0, dupTri = 0, dupSphere = 0;
  uint64_t shadedTriHits = 0, texelFetches = 0;
  void add(const Counters& o) {
    samples += o.samples; rays += o.rays; nodeVisits += o.nodeVisits; boxPasses += o.boxPasses;
    triCalls += o.triCalls; sphereCalls += o.sphereCalls; dupTri += o.dupTri; dupSphere += o.dupSphere;
    shadedTriHits += o.shadedTriHits; texelFetches += o.texelFetches;
  }
};
static thread_local Counters* tlsCounters = nullptr;
#define COUNT(field) do { if (tlsCounters) tlsCounters->field++; } while (0)

// ---------------------------------------------------------------- aabb.h
struct aabb {
  vec3 minimum, maximum;
  aabb() {}
  aabb(const vec3& a, const vec3& b) : minimum(a), maximum(b) {}
  bool hit(const ray& r, float tMin, float tMax) const {  // aabb.h:11-27 (the unused unitVector at :12 has no effect)
    for (int axis = 0; axis < 3; axis++) {
      float t0 = fminf((minimum(axis) - r.o(axis)) / r.dir(axis), (maximum(axis) - r.o(axis)) / r.dir(axis));
      float t1 = fmaxf((minimum(axis) - r.o(axis)) / r.dir(axis), (maximum(axis) - r.o(axis)) / r.dir(axis));
      tMin = fmaxf(t0, tMin);
      tMax = fminf(t1, tMax);
      if (tMax <= tMin) return false;
    }
    return true;
  }
};
static aabb surroundingBox(const aabb& b0, const aabb& b1) {  // aabb.h:33-43
  vec3 small(fminf(b0.minimum(0), b1.minimum(0)), fminf(b0.minimum(1), b1.minimum(1)), fminf(b0.minimum(2), b1.minimum(2)));
  vec3 large(fmaxf(b0.maximum(0), b1.maximum(0)), fmaxf(b0.maximum(1), b1.maximum(1)), fmaxf(b0.maximum(2), b1.maximum(2)));
  return aabb(small, large);
}

// ---------------------------------------------------------------- texture.h
struct texture {
  virtual ~texture() {}
  virtual vec3 value(float u, float v, const vec3& p) const = 0;  // texture.h:13-16
};
struct solidColor : texture {  // texture.h:18-32
  vec3 colorValue;
  explicit solidColor(const vec3& c) : colorValue(c) {}
  vec3 value(float, float, const vec3&) const override { return colorValue; }
};
struct checker : texture {  // texture.h:34-52
  const texture *odd, *even;
  checker(const texture* even_, const texture* odd_) : odd(odd_), even(even_) {}
  vec3 value(float u, float v, const vec3& p) const override {
    float sines = sinf(10.0f * p(0)) * sinf(10.0f * p(1)) * sinf(10.0f * p(2));
    if (sines < 0) return odd->value(u, v, p) * 255.0f;
    return even->value(u, v, p) * 255.0f;
  }
};
struct imagePNG : texture {  // texture.h:109-153 (image3bpp :54-107 is the bpp=3 case)
  const uint8_t* data;
  const uint8_t* dataEnd;  // end of the owning texel buffer: defines the 1-bpp overrun
  int width, height, bpp, bytesPerScanline;
  imagePNG(const uint8_t* d, const uint8_t* end, int w, int h, int b)
      : data(d), dataEnd(end), width(w), height(h), bpp(b), bytesPerScanline(b * w) {}
  vec3 value(float u, float v, const vec3&) const override {
    if (data == nullptr) return vec3(1.0f, 0, 1.0f);
    COUNT(texelFetches);
    u = clampf(u, 0, 1.0f);
    v = 1.0f - clampf(v, 0, 1.0f);
    int i = (int)(u * width);  // NaN -> UB in the reference; defined here (and on the device) as 0
    int j = (int)(v * height);
    if (!(u == u)) i = 0;
    if (!(v == v)) j = 0;
    if (i >= width) i = width - 1;
    if (j >= height) j = height - 1;
    const uint8_t* pixel = data + (int64_t)j * bytesPerScanline + (int64_t)i * bpp;
    // bpp==1 reads pixel[1], pixel[2] from the next texels (texture.h:147); past the
    // end of the buffer the reference reads heap garbage -- defined as 0 here.
    float c[3];
    for (int k = 0; k < 3; k++) c[k] = (pixel + k < dataEnd) ? (float)pixel[k] : 0.0f;
    return vec3(c[0], c[1], c[2]);
  }
};

// ---------------------------------------------------------------- hittable.h
struct material;
struct hitRecord {  // hittable.h:9-22
  vec3 p, normal, tangent, bitangent;
  float uv[2] = {0, 0};
  float t = 0;
  bool frontFace = false;
  const material* matPtr = nullptr;
  int prim = SRT_NO_HIT;  // not in the reference: list index of the primitive
  bool isTri = false;     // not in the reference: for the shaded-triangle counter
  void setFaceNormal(const ray& r, const vec3& outwardNormal) {
    frontFace = dot(r.dir, outwardNormal) < 0;
    normal = frontFace ? outwardNormal : -outwardNormal;
  }
};
struct hittable {  // hittable.h:24-33
  virtual ~hittable() {}
  virtual bool hit(const ray& r, float tMin, float tMax, hitRecord& rec) const = 0;
  virtual bool boundingBox(float time0, float time1, aabb& out) const = 0;
};

// ---------------------------------------------------------------- material.h / pbr.h
struct material {
  virtual ~material() {}
  virtual bool scatter(const ray& rIn, const hitRecord& rec, vec3& attenuation, ray& scattered, Rng& rng) const = 0;
  virtual vec3 emitted(float, float, const vec3&) const { return vec3(0, 0, 0); }  // material.h:18-20
  int id = -1;
};

static float trowbridgeReitzNDF(float NdotH, float roughness) {  // pbr.h:58-65
  float alpha = roughness * roughness;
  float alpha2 = alpha * alpha;
  float NdotH2 = NdotH * NdotH;
  float denom = pi * std::pow(NdotH2 * (alpha2 - 1.0f) + 1.0f, 2.0f);  // float std::pow
  return alpha2 / denom;
}
static float schlickGAF(float NdotV, float roughness) {  // pbr.h:69-73
  float k = ((roughness + 1.0f) * (roughness + 1.0f)) / 8.0f;
  return NdotV / (NdotV * (1.0f - k) + k);
}
static vec3 fresnelEpic(const vec3& F0, float HdotV) {  // pbr.h:75-81: unqualified pow -> double pow
  float power = (float)pow((double)2.0f, (double)((-5.55473f * HdotV - 6.98316f) * HdotV));
  return vec3(F0(0) + (1.0f - F0(0)) * power, F0(1) + (1.0f - F0(1)) * power, F0(2) + (1.0f - F0(2)) * power);
}

struct pbrMetallicRoughness : material {  // material.h:23-85, scatter :156-245
  const texture *albedoMap = nullptr, *normalMap = nullptr, *metallicMap = nullptr, *roughnessMap = nullptr;
  float albedo[4] = {1, 1, 1, 1};
  // material.h:25-40 leave these uninitialised (UB); the scene description
  // always carries defined values (0,0 reproduces the published image).
  float metalness = 0, roughness = 0;
  bool scatter(const ray& rIn, const hitRecord& rec, vec3& attenuation, ray& scatterRay, Rng& rng) const override {
    vec3 normal;
    float m, r;
    if (albedoMap) {
      attenuation = albedoMap->value(rec.uv[0], rec.uv[1], rec.p);
      attenuation = attenuation / 255.0f;
    } else
      attenuation = vec3(albedo[0], albedo[1], albedo[2]);
    if (normalMap) {
      normal = normalMap->value(rec.uv[0], rec.uv[1], rec.p);
      normal = normalIntToFloat(normal);
      // Matrix3f(tangent | bitangent | normal) * n, rows reduced like dot()
      vec3 w;
      for (int i = 0; i < 3; i++)
        w(i) = rec.tangent(i) * normal(0) + (rec.bitangent(i) * normal(1) + rec.normal(i) * normal(2));
      normal = unitVector(w);
    } else
      normal = rec.normal;
    if (metallicMap)
      m = clampf(metallicMap->value(rec.uv[0], rec.uv[1], rec.p)(0) / 255.0f, 0, 1.0f);
    else
      m = metalness;
    if (roughnessMap)
      r = clampf(roughnessMap->value(rec.uv[0], rec.uv[1], rec.p)(1) / 255.0f, 0, 1.0f);
    else
      r = roughness;

    vec3 scatterDir = normal + rng.randomUnitVector();
    if (nearZero(scatterDir)) scatterDir = normal;
    scatterDir = unitVector(scatterDir);
    scatterRay = ray(rec.p, scatterDir, rIn.time);

    vec3 viewVec = -unitVector(rIn.dir);
    vec3 halfVec = unitVector(scatterRay.dir + viewVec);
    float NdotL = fmaxf(dot(normal, scatterRay.dir), 0);
    float NdotH = fmaxf(dot(normal, halfVec), 0);
    float HdotV = fmaxf(dot(halfVec, viewVec), 0);
    float NdotV = fmaxf(dot(normal, viewVec), 0);

    vec3 fresnelReflectance(albedo[0], albedo[1], albedo[2]);
    vec3 F0 = lerp(vec3(0.4f, 0.4f, 0.4f), fresnelReflectance, m);
    float D = trowbridgeReitzNDF(NdotH, r);
    vec3 F = fresnelEpic(F0, HdotV);
    float G = schlickGAF(NdotL, r) * schlickGAF(NdotV, r);

    vec3 finalDiffuse = attenuation / pi;
    for (int i = 0; i < 3; i++) finalDiffuse(i) *= (1.0f - F(i));
    finalDiffuse = finalDiffuse * (1.0f - m);
    for (int i = 0; i < 3; i++) finalDiffuse(i) *= albedo[i];
    vec3 finalSpecular = ((D * F) * G) / (4.0f * NdotV * NdotL + epsilon);
    attenuation = (finalDiffuse + finalSpecular) * NdotL;
    return true;
  }
};

struct metal : material {  // material.h:87-102
  vec3 albedo;
  float fuzz;
  metal(const vec3& a, float f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
  bool scatter(const ray& rIn, const hitRecord& rec, vec3& attenuation, ray& scatterRay, Rng& rng) const override {
    vec3 reflected = reflect(unitVector(rIn.dir), rec.normal);
    scatterRay = ray(rec.p, reflected + fuzz * rng.randomInUnitSphere(), rIn.time);  // drawn even when fuzz == 0
    attenuation = albedo;
    return dot(scatterRay.dir, rec.normal) > 0;
  }
};

struct dielectric : material {  // material.h:104-137
  float ir;
  explicit dielectric(float i) : ir(i) {}
  static float reflectance(float cosine, float refIDX) {  // material.h:132-136: double pow, double expression
    float r0 = (1.0f - refIDX) / (1.0f + refIDX);
    r0 = r0 * r0;
    return (float)((double)r0 + (double)(1.0f - r0) * pow((double)(1.0f - cosine), (double)5.0f));
  }
  bool scatter(const ray& rIn, const hitRecord& rec, vec3& attenuation, ray& scatterRay, Rng& rng) const override {
    attenuation = vec3(1.0f, 1.0f, 1.0f);
    float refractionRatio = rec.frontFace ? (1.0f / ir) : ir;
    vec3 unitDir = unitVector(rIn.dir);
    float cosTheta = (float)fmin((double)dot(rec.normal, -unitDir), (double)1.0f);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    bool cannotRefract = refractionRatio * sinTheta > 1.0f;
    vec3 dir;
    if (cannotRefract || reflectance(cosTheta, refractionRatio) > rng.randomFloat())  // draw only if not TIR
      dir = reflect(unitDir, rec.normal);
    else
      dir = refract(unitDir, rec.normal, refractionRatio);
    scatterRay = ray(rec.p, dir, rIn.time);
    return true;
  }
};

struct diffuseLight : material {  // material.h:139-154
  const texture* emit;
  explicit diffuseLight(const texture* e) : emit(e) {}
  bool scatter(const ray&, const hitRecord&, vec3&, ray&, Rng&) const override { return false; }
  vec3 emitted(float u, float v, const vec3& p) const override { return emit->value(u, v, p); }
};

// ---------------------------------------------------------------- sphere.h
struct sphere : hittable {
  vec3 center0, center1;
  float t0, t1, radius;
  const material* matPtr;
  int prim;
  vec3 center(float time) const {  // sphere.h:47-52
    if (center0 != center1) return center0 + ((time - t0) / (t1 - t0)) * (center1 - center0);
    return center0;
  }
  static void getSphereUV(const vec3& p, float* uv) {  // sphere.h:32-38
    float theta = acosf(-p(1));
    float phi = atan2f(-p(2), p(0)) + pi;
    uv[0] = phi / (2.0f * pi);
    uv[1] = theta / pi;
  }
  void calcTangentBasis(const vec3& normal, vec3& tangent, vec3& bitangent) const {  // sphere.h:96-106
    vec3 b;
    if (1.0f - fabsf(dot(normal, vec3(0, 1.0f, 0))) < epsilon)
      b = -vec3(0, 0, 1.0f);
    else
      b = vec3(0, 1.0f, 0);
    tangent = unitVector(cross(b, normal));
    bitangent = unitVector(cross(normal, tangent));
  }
  bool hit(const ray& r, float tMin, float tMax, hitRecord& rec) const override {  // sphere.h:54-83
    COUNT(sphereCalls);
    vec3 oc = r.o - center(r.time);
    float a = lengthSquared(r.dir);
    float halfB = dot(oc, r.dir);
    float c = lengthSquared(oc) - radius * radius;
    float discriminant = halfB * halfB - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = sqrtf(discriminant);
    float root = (-halfB - sqrtd) / a;
    if (root < tMin || root > tMax) {
      root = (-halfB + sqrtd) / a;
      if (root < tMin || root > tMax) return false;
    }
    rec.t = root;
    rec.p = r.at(rec.t);
    vec3 outwardNormal = unitVector(rec.p - center(r.time));
    rec.setFaceNormal(r, outwardNormal);
    getSphereUV(outwardNormal, rec.uv);
    rec.matPtr = matPtr;
    rec.prim = prim;
    rec.isTri = false;
    calcTangentBasis(outwardNormal, rec.tangent, rec.bitangent);
    return true;
  }
  bool boundingBox(float time0, float time1, aabb& out) const override {  // sphere.h:85-94
    vec3 rr(radius, radius, radius);
    aabb box0(center(time0) - rr, center(time0) + rr);
    aabb box1(center(time1) - rr, center(time1) + rr);
    out = surroundingBox(box0, box1);
    return true;
  }
};

// ---------------------------------------------------------------- model.h triangle
struct triangle : hittable {
  vec3 P[3];  // parentMesh->positions[vertices[i]]
  float UV[3][2];
  const material* matPtr;
  int prim;
  vec3 getNormal() const { return cross(P[1] - P[0], P[2] - P[0]); }  // model.h:276-283
  void calcTangentBasis(vec3& tangent, vec3& bitangent) const {        // model.h:214-235
    vec3 edge0 = P[1] - P[0];
    vec3 edge1 = P[2] - P[0];
    float dUV0[2] = {UV[1][0] - UV[0][0], UV[1][1] - UV[0][1]};
    float dUV1[2] = {UV[2][0] - UV[0][0], UV[2][1] - UV[0][1]};
    float f = (dUV0[0] * dUV1[1] - dUV1[0] * dUV0[1]);
    if (f == 0) f += epsilon;
    f = 1.0f / f;
    for (int i = 0; i < 3; i++) tangent(i) = f * (dUV1[1] * edge0(i) - dUV0[1] * edge1(i));
    tangent = unitVector(tangent);
    for (int i = 0; i < 3; i++) bitangent(i) = f * (-dUV1[0] * edge0(i) + dUV0[0] * edge1(i));
    bitangent = unitVector(bitangent);
  }
  bool hit(const ray& r, float tMin, float /*tMax: never tested, model.h:128*/, hitRecord& rec) const override {  // model.h:104-181
    COUNT(triCalls);
    vec3 normal = getNormal();
    float NdotDir = dot(normal, r.dir);
    if (fabsf(NdotDir) < epsilon) return false;
    if (dot(r.dir, normal) > 0) return false;  // back-face cull
    float d = -dot(normal, P[0]);
    float t = -(dot(normal, r.o) + d) / NdotDir;
    if (t < tMin) return false;
    vec3 p = r.o + t * r.dir;
    vec3 c;
    c = cross(P[1] - P[0], p - P[0]);
    if (dot(normal, c) < 0) return false;
    c = cross(P[2] - P[1], p - P[1]);
    if (dot(normal, c) < 0) return false;
    c = cross(P[0] - P[2], p - P[2]);
    if (dot(normal, c) < 0) return false;
    // inverse-distance weights, not barycentrics (model.h:158-169)
    float d0 = distance(p, P[0]), d1 = distance(p, P[1]), d2 = distance(p, P[2]);
    float denom = (1.0f / d0) + (1.0f / d1) + (1.0f / d2);
    float r0 = (1.0f / d0) / denom, r1 = (1.0f / d1) / denom, r2 = (1.0f / d2) / denom;
    float u = r0 * UV[0][0] + r1 * UV[1][0] + r2 * UV[2][0];
    float v = 1.0f - (r0 * UV[0][1] + r1 * UV[1][1] + r2 * UV[2][1]);
    vec3 outwardNormal = unitVector(normal);
    rec.t = t;
    rec.p = r.at(rec.t);
    rec.setFaceNormal(r, outwardNormal);
    rec.uv[0] = u;
    rec.uv[1] = v;
    rec.matPtr = matPtr;
    rec.prim = prim;
    rec.isTri = true;
    calcTangentBasis(rec.tangent, rec.bitangent);
    return true;
  }
  bool boundingBox(float, float, aabb& out) const override {  // model.h:183-212
    vec3 mn(infinity, infinity, infinity), mx(-infinity, -infinity, -infinity);
    for (int k = 0; k < 3; k++)
      for (int axis = 0; axis < 3; axis++) {
        mn(axis) = std::min(mn(axis), P[k](axis));
        mx(axis) = std::max(mx(axis), P[k](axis));
      }
    for (int axis = 0; axis < 3; axis++)
      if (mn(axis) == mx(axis)) {
        mn(axis) -= 0.0001f;
        mx(axis) += 0.0001f;
      }
    out = surroundingBox(aabb(mn, mx), aabb(mn, mx));
    return true;
  }
};

// ---------------------------------------------------------------- hittablelist.h
struct hittableList : hittable {
  std::vector<const hittable*> objects;
  bool hit(const ray& r, float tMin, float tMax, hitRecord& rec) const override {  // hittablelist.h:33-47
    hitRecord temp;
    bool any = false;
    float closest = tMax;
    for (const hittable* o : objects)
      if (o->hit(r, tMin, closest, temp)) {
        any = true;
        closest = temp.t;
        rec = temp;
      }
    return any;
  }
  bool boundingBox(float t0, float t1, aabb& out) const override {  // hittablelist.h:49-64
    if (objects.empty()) return false;
    aabb temp;
    bool first = true;
    for (const hittable* o : objects) {
      if (!o->boundingBox(t0, t1, temp)) return false;
      out = first ? temp : surroundingBox(out, temp);
      first = false;
    }
    return true;
  }
};

// ---------------------------------------------------------------- bvh.h
struct bvhNode : hittable {
  const hittable *left = nullptr, *right = nullptr;
  aabb box;
  static bool boxCompare(const hittable* a, const hittable* b, int axis) {  // bvh.h:34-41 (boxes at time 0,0)
    aabb boxA, boxB;
    a->boundingBox(0, 0, boxA);
    b->boundingBox(0, 0, boxB);
    return boxA.minimum(axis) < boxB.minimum(axis);
  }
  // bvh.h:55-95.  The reference copies the whole vector at every node (:57) and sorts
  // [start,end) of its copy; siblings only ever touch disjoint sub-ranges of what their
  // parent sorted, so sorting one shared vector in place yields the same tree.
  bvhNode(std::vector<const hittable*>& objects, size_t start, size_t end, float time0, float time1, Rng& rng,
          std::vector<std::unique_ptr<bvhNode>>& pool) {
    int axis = rng.randomInt(0, 2);
    auto comparator = [axis](const hittable* a, const hittable* b) { return boxCompare(a, b, axis); };
    size_t span = end - start;
    if (span == 1) {
      left = right = objects[start];
    } else if (span == 2) {
      if (comparator(objects[start], objects[start + 1])) {
        left = objects[start];
        right = objects[start + 1];
      } else {
        left = objects[start + 1];
        right = objects[start];
      }
    } else {
      std::sort(objects.begin() + start, objects.begin() + end, comparator);
      size_t mid = start + span / 2;
      bvhNode* l = new bvhNode(objects, start, mid, time0, time1, rng, pool);
      pool.emplace_back(l);
      bvhNode* r = new bvhNode(objects, mid, end, time0, time1, rng, pool);
      pool.emplace_back(r);
      left = l;
      right = r;
    }
    aabb boxLeft, boxRight;
    left->boundingBox(time0, time1, boxLeft);
    right->boundingBox(time0, time1, boxRight);
    box = surroundingBox(boxLeft, boxRight);
  }
  bool hit(const ray& r, float tMin, float tMax, hitRecord& rec) const override {  // bvh.h:97-105
    COUNT(nodeVisits);
    if (!box.hit(r, tMin, tMax)) return false;
    COUNT(boxPasses);
    bool hitLeft = left->hit(r, tMin, tMax, rec);
    if (left == right && tlsCounters) {  // single-object leaf tests its object twice (bvh.h:67-69)
      if (dynamic_cast<const triangle*>(left)) tlsCounters->dupTri++;
      else if (dynamic_cast<const sphere*>(left)) tlsCounters->dupSphere++;
    }
    bool hitRight = right->hit(r, tMin, hitLeft ? rec.t : tMax, rec);
    return hitLeft || hitRight;
  }
  bool boundingBox(float, float, aabb& out) const override {  // bvh.h:107-110
    out = box;
    return true;
  }
};

// ---------------------------------------------------------------- camera.h
static void makeCamera(const SrtCameraParams& in, SrtCamera& out) {  // camera.h:10-38
  vec3 eye(in.eye[0], in.eye[1], in.eye[2]), lookAt(in.lookAt[0], in.lookAt[1], in.lookAt[2]), up(in.up[0], in.up[1], in.up[2]);
  float theta = in.vfovDegrees * pi / 180.0f;  // deg2rad, globals.h:26-28
  double h = tan((double)(theta / 2.0f));      // unqualified tan(float) -> double tan
  double vpHeight = 2.0f * h;
  double vpWidth = in.aspect * vpHeight;
  vec3 w = unitVector(eye - lookAt);
  vec3 hor = unitVector(cross(up, w));
  vec3 vert = unitVector(cross(w, hor));
  vec3 origin = eye;
  vec3 horizontal = (float)(in.focusDist * vpWidth) * hor;  // double scalar cast to float by Eigen
  vec3 vertical = (float)(in.focusDist * vpHeight) * vert;
  vec3 lleft = origin - horizontal / 2.0f - vertical / 2.0f - in.focusDist * w;
  for (int i = 0; i < 3; i++) {
    out.origin[i] = origin(i); out.lleft[i] = lleft(i); out.horizontal[i] = horizontal(i); out.vertical[i] = vertical(i);
    out.w[i] = w(i); out.hor[i] = hor(i); out.vert[i] = vert(i);
  }
  out.lensRadius = in.aperture / 2.0f;
  out.time0 = in.time0;
  out.time1 = in.time1;
}
static inline vec3 v3(const float* f) { return vec3(f[0], f[1], f[2]); }
static ray getRay(const SrtCamera& c, float s, float t, Rng& rng) {  // camera.h:40-46
  vec3 rd = c.lensRadius * rng.randomInUnitDisk();
  vec3 offset = rd(0) * v3(c.hor) + rd(1) * v3(c.vert);
  vec3 origin = v3(c.origin);
  return ray(origin + offset, v3(c.lleft) + s * v3(c.horizontal) + t * v3(c.vertical) - origin - offset,
             rng.randomFloat(c.time0, c.time1));
}

// ---------------------------------------------------------------- main.cpp:33-52
static vec3 rayColor(const ray& r, const vec3& background, const hittable& world, int maxBounce, float tMin, Rng& rng) {
  hitRecord rec;
  if (maxBounce <= 0) return vec3(0, 0, 0);
  COUNT(rays);
  if (!world.hit(r, tMin, infinity, rec)) return background;
  ray scattered;
  vec3 attenuation;
  vec3 emitted = rec.matPtr->emitted(rec.uv[0], rec.uv[1], rec.p);
  if (rec.isTri) COUNT(shadedTriHits);
  if (!rec.matPtr->scatter(r, rec, attenuation, scattered, rng)) return emitted;
  vec3 newColor = rayColor(scattered, background, world, maxBounce - 1, tMin, rng);
  newColor = vec3(newColor(0) * attenuation(0), newColor(1) * attenuation(1), newColor(2) * attenuation(2));
  return emitted + newColor;
}

// ---------------------------------------------------------------- scene
struct Scene {
  std::mt19937 mt;  // the process-global generator of globals.h:32, one per scene handle
  std::vector<uint8_t> texels;
  std::vector<std::unique_ptr<texture>> textures;
  std::vector<std::unique_ptr<material>> materials;
  std::vector<std::unique_ptr<hittable>> prims;  // list order
  std::vector<int> primIsTri;
  std::vector<std::unique_ptr<bvhNode>> pool;
  std::vector<const bvhNode*> itemRoot;  // per world item (nullptr for bare prims)
  hittableList world;
  uint64_t buildDraws = 0;
};

static Scene* buildScene(const SrtSceneDesc* d, uint64_t preDraws) {
  auto s = std::make_unique<Scene>();
  s->texels.assign(d->texels, d->texels + d->numTexelBytes);
  const uint8_t* tbase = s->texels.data();
  const uint8_t* tend = tbase + s->texels.size();
  s->textures.resize(d->numTextures);
  // two passes: checkers reference other textures
  for (int i = 0; i < d->numTextures; i++) {
    const SrtTextureIn& t = d->textures[i];
    if (t.kind == SRT_TEX_SOLID)
      s->textures[i].reset(new solidColor(vec3(t.color[0], t.color[1], t.color[2])));
    else if (t.kind == SRT_TEX_IMAGE)
      s->textures[i].reset(new imagePNG(t.width > 0 ? tbase + t.texelOffset : nullptr, tend, t.width, t.height, t.bpp));
  }
  for (int i = 0; i < d->numTextures; i++) {
    const SrtTextureIn& t = d->textures[i];
    if (t.kind == SRT_TEX_CHECKER) s->textures[i].reset(new checker(s->textures[t.even].get(), s->textures[t.odd].get()));
  }
  auto tex = [&](int id) -> const texture* { return id >= 0 ? s->textures[id].get() : nullptr; };
  for (int i = 0; i < d->numMaterials; i++) {
    const SrtMaterialIn& m = d->materials[i];
    material* out = nullptr;
    if (m.type == SRT_MAT_PBR) {
      auto* p = new pbrMetallicRoughness();
      p->albedoMap = tex(m.albedoTex); p->normalMap = tex(m.normalTex);
      p->metallicMap = tex(m.metallicTex); p->roughnessMap = tex(m.roughnessTex);
      for (int k = 0; k < 4; k++) p->albedo[k] = m.albedo[k];
      p->metalness = m.metalness; p->roughness = m.roughness;
      out = p;
    } else if (m.type == SRT_MAT_METAL)
      out = new metal(vec3(m.albedo[0], m.albedo[1], m.albedo[2]), m.fuzz);
    else if (m.type == SRT_MAT_DIELECTRIC)
      out = new dielectric(m.ir);
    else
      out = new diffuseLight(tex(m.albedoTex));
    out->id = i;
    s->materials.emplace_back(out);
  }
  for (int i = 0; i < d->numPrims; i++) {
    const SrtPrimRef& pr = d->prims[i];
    if (pr.type == SRT_PRIM_TRIANGLE) {
      const SrtTriangleIn& ti = d->triangles[pr.index];
      auto* t = new triangle();
      for (int k = 0; k < 3; k++) {
        t->P[k] = vec3(ti.p[k][0], ti.p[k][1], ti.p[k][2]);
        t->UV[k][0] = ti.uv[k][0]; t->UV[k][1] = ti.uv[k][1];
      }
      t->matPtr = s->materials[ti.material].get();
      t->prim = i;
      s->prims.emplace_back(t);
      s->primIsTri.push_back(1);
    } else {
      const SrtSphereIn& si = d->spheres[pr.index];
      auto* sp = new sphere();
      sp->center0 = v3(si.center0); sp->center1 = v3(si.center1);
      sp->t0 = si.time0; sp->t1 = si.time1; sp->radius = si.radius;
      sp->matPtr = s->materials[si.material].get();
      sp->prim = i;
      s->prims.emplace_back(sp);
      s->primIsTri.push_back(0);
    }
  }
  Rng rng;
  rng.mode = RNG_MT;
  rng.mt = &s->mt;
  s->mt.discard(preDraws);  // one 32-bit draw per randomFloat() (generate_canonical<float, 24> over mt19937)
  for (int w = 0; w < d->numWorld; w++) {
    const SrtWorldItem& it = d->world[w];
    if (it.kind == SRT_WORLD_PRIM) {
      s->world.objects.push_back(s->prims[it.first].get());
      s->itemRoot.push_back(nullptr);
    } else {
      // main.cpp:146: make_shared<bvhNode>(objects, 0, 1) -> bvh.h:15-16
      std::vector<const hittable*> objs;
      for (int i = it.first; i < it.first + it.count; i++) objs.push_back(s->prims[i].get());
      bvhNode* root = new bvhNode(objs, 0, objs.size(), it.time0, it.time1, rng, s->pool);
      s->pool.emplace_back(root);
      s->world.objects.push_back(root);
      s->itemRoot.push_back(root);
    }
  }
  s->buildDraws = rng.draws;
  return s.release();
}

// pre-order flatten in the product's SrtBvhNode format, for topology comparison.
static int flatten(const Scene* s, const hittable* h, std::vector<SrtBvhNode>& out, int depth, int* maxDepth) {
  if (const bvhNode* n = dynamic_cast<const bvhNode*>(h)) {
    if (depth > *maxDepth) *maxDepth = depth;
    int idx = (int)out.size();
    out.emplace_back();
    int l = flatten(s, n->left, out, depth + 1, maxDepth);
    int r = (n->right == n->left) ? l : flatten(s, n->right, out, depth + 1, maxDepth);
    SrtBvhNode& o = out[idx];
    for (int k = 0; k < 3; k++) { o.bmin[k] = n->box.minimum(k); o.bmax[k] = n->box.maximum(k); }
    o.left = l;
    o.right = r;
    return idx;
  }
  if (const triangle* t = dynamic_cast<const triangle*>(h)) return ~t->prim;
  return ~static_cast<const sphere*>(h)->prim;
}

static void fillHit(const Scene* s, bool ok, const hitRecord& rec, SrtHit& h) {
  memset(&h, 0, sizeof(h));
  h.prim = SRT_NO_HIT;
  h.material = -1;
  if (!ok) return;
  h.prim = rec.prim;
  h.t = rec.t;
  for (int k = 0; k < 3; k++) {
    h.p[k] = rec.p(k); h.normal[k] = rec.normal(k); h.tangent[k] = rec.tangent(k); h.bitangent[k] = rec.bitangent(k);
  }
  h.uv[0] = rec.uv[0]; h.uv[1] = rec.uv[1];
  h.frontFace = rec.frontFace ? 1 : 0;
  h.material = rec.matPtr->id;
  (void)s;
}

static void writeColorTarget(uint8_t* data, int x, int y, int w, const float* sum, int numSamples) {  // color.h:25-41
  float scale = 1.0f / numSamples;
  uint8_t* pixel = &data[(y * w + x) * 4];
  for (int k = 0; k < 3; k++) {
    float c = sqrtf(sum[k] * scale);
    float v = 256 * clampf(c, 0.0f, 0.999f);
    pixel[k] = (v == v) ? (uint8_t)v : 0;  // NaN cast is UB in the reference; x86 yields 0
  }
  pixel[3] = 255;
}

}  // namespace orc

// =================================================================== C API
using namespace orc;

struct OrcStats {
  uint64_t samples, rays, nodeVisits, boxPasses, triCalls, sphereCalls, dupTri, dupSphere, shadedTriHits, texelFetches;
  uint64_t rngDraws;
};

extern "C" {

void* orc_scene_create(const SrtSceneDesc* d) { return buildScene(d, 0); }
// preDraws: randomFloat() calls the scene-construction code made on the process-global generator before the
// world's bvhNode is built (main.cpp:92-122 draws sphere placements from it; the reference then continues the
// SAME stream into bvh.h:60) -- the scene handle's generator skips that many draws first.
void* orc_scene_create2(const SrtSceneDesc* d, uint64_t preDraws) { return buildScene(d, preDraws); }
void orc_scene_destroy(void* h) { delete static_cast<Scene*>(h); }
uint64_t orc_build_draws(void* h) { return static_cast<Scene*>(h)->buildDraws; }

void orc_make_camera(const SrtCameraParams* in, SrtCamera* out) { makeCamera(*in, *out); }

// first n draws of a fresh default-seeded generator (globals.h:30-35)
void orc_rng_kat(int n, float* out) {
  std::mt19937 mt;
  Rng r;
  r.mode = RNG_MT;
  r.mt = &mt;
  for (int i = 0; i < n; i++) out[i] = r.randomFloat();
}
// same through libstdc++'s own distribution object, to pin Rng::randomFloat's closed form
void orc_rng_kat_libstdcxx(int n, float* out) {
  std::uniform_real_distribution<float> distribution(0.0f, 1.0f);
  std::mt19937 generator;
  for (int i = 0; i < n; i++) out[i] = distribution(generator);
}
// counter stream of (seed,pixel,sample)
void orc_rng_counter(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out) {
  Rng r;
  r.mode = RNG_COUNTER;
  r.key(seed, pixel, sample);
  for (int i = 0; i < n; i++) out[i] = r.randomFloat();
}
void orc_random_vec3_kat(float* out3) {  // (0.811584, -0.729046, 0.629447) with g++
  std::mt19937 mt;
  Rng r;
  r.mode = RNG_MT;
  r.mt = &mt;
  vec3 v = r.randomVec3f(-1.0f, 1.0f);
  out3[0] = v(0); out3[1] = v(1); out3[2] = v(2);
}

int orc_bvh_flatten(void* h, int item, SrtBvhNode* nodes, int capacity, int* count, int* depth) {
  Scene* s = static_cast<Scene*>(h);
  if (item < 0 || item >= (int)s->itemRoot.size() || !s->itemRoot[item]) return 1;
  std::vector<SrtBvhNode> out;
  int md = 0;
  flatten(s, s->itemRoot[item], out, 1, &md);
  *count = (int)out.size();
  if (depth) *depth = md;
  if (nodes) {
    if (capacity < (int)out.size()) return 2;
    memcpy(nodes, out.data(), out.size() * sizeof(SrtBvhNode));
  }
  return 0;
}

// world.hit(r, tMin, tMax, rec) for each ray; traversal==CLOSEST is the brute-force
// closest hit over all primitives (what the survey compared F4 against).
void orc_trace(void* h, const SrtRay* rays, int64_t n, SrtHit* hits, int traversal) {
  Scene* s = static_cast<Scene*>(h);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n; i++) {
    Counters c;
    tlsCounters = &c;
    ray r(v3(rays[i].o), v3(rays[i].d), rays[i].time);
    hitRecord rec;
    bool ok;
    if (traversal == SRT_TRAVERSE_FAITHFUL) {
      ok = s->world.hit(r, rays[i].tMin, rays[i].tMax, rec);
    } else {
      ok = false;
      float closest = rays[i].tMax;
      hitRecord temp;
      for (auto& p : s->prims)
        if (p->hit(r, rays[i].tMin, closest, temp) && temp.t <= closest) {
          ok = true;
          closest = temp.t;
          rec = temp;
        }
    }
    fillHit(s, ok, rec, hits[i]);
    hits[i].nodeVisits = (int)c.nodeVisits;
    hits[i].boxPasses = (int)c.boxPasses;
    hits[i].triTests = (int)(c.triCalls - c.dupTri);
    hits[i].sphereTests = (int)(c.sphereCalls - c.dupSphere);
    tlsCounters = nullptr;
  }
}

// material scatter known-answer entry: one scatter() call on a given hit record with the
// counter RNG keyed (seed, pixel, sample).  out: attenuation[3], dir[3], origin[3], ok, emitted[3]
void orc_scatter(void* h, const SrtRay* rIn, const SrtHit* hit, uint64_t seed, uint32_t pixel, uint32_t sample,
                 float* out13) {
  Scene* s = static_cast<Scene*>(h);
  Rng rng;
  rng.mode = RNG_COUNTER;
  rng.key(seed, pixel, sample);
  hitRecord rec;
  rec.p = v3(hit->p); rec.normal = v3(hit->normal); rec.tangent = v3(hit->tangent); rec.bitangent = v3(hit->bitangent);
  rec.uv[0] = hit->uv[0]; rec.uv[1] = hit->uv[1]; rec.t = hit->t; rec.frontFace = hit->frontFace != 0;
  rec.matPtr = s->materials[hit->material].get();
  ray r(v3(rIn->o), v3(rIn->d), rIn->time), sc;
  vec3 att;
  vec3 em = rec.matPtr->emitted(rec.uv[0], rec.uv[1], rec.p);
  bool ok = rec.matPtr->scatter(r, rec, att, sc, rng);
  for (int k = 0; k < 3; k++) { out13[k] = att(k); out13[3 + k] = sc.dir(k); out13[6 + k] = sc.o(k); out13[10 + k] = em(k); }
  out13[9] = ok ? 1.0f : 0.0f;
}

// main.cpp:200-227.  accum: float[W*H*4] image order (rgb sum, a = spp); rgba: uint8[W*H*4].
// rngMode MT: one serial stream continuing the scene's generator (threads forced to 1).
// rngMode COUNTER: rows in parallel over `threads` OpenMP threads.
// rowBegin/rowEnd bound the rows rendered (whole image: 0,H) so baselines can time a sample.
int orc_render(void* h, const SrtCamera* cam, const SrtRenderParams* p, int rngMode, int threads, int rowBegin,
               int rowEnd, float* accum, uint8_t* rgba, OrcStats* stats) {
  Scene* s = static_cast<Scene*>(h);
  const int W = p->imageWidth, H = p->imageHeight;
  const vec3 background(p->background[0], p->background[1], p->background[2]);
  if (rngMode == RNG_MT) threads = 1;
  if (threads < 1) threads = 1;
  Counters total;
  uint64_t draws = 0;
#pragma omp parallel num_threads(threads)
  {
    Counters local;
    Rng rng;
    rng.mode = rngMode;
    rng.mt = &s->mt;
    tlsCounters = stats ? &local : nullptr;
#pragma omp for schedule(dynamic, 1)
    for (int y = rowBegin; y < rowEnd; ++y) {
      for (int x = 0; x < W; ++x) {
        vec3 pixelColor(0, 0, 0);
        for (int sIdx = p->sampleFirst; sIdx < p->sampleFirst + p->spp; ++sIdx) {
          if (rngMode == RNG_COUNTER) rng.key(p->seed, (uint32_t)(y * W + x), (uint32_t)sIdx);
          COUNT(samples);
          float u = (float)(x + rng.randomFloat()) / (W - 1);           // main.cpp:210
          float v = (float)((H - y) + rng.randomFloat()) / (H - 1);     // main.cpp:211 (H - y)
          ray r = getRay(*cam, u, v, rng);
          vec3 c = rayColor(r, background, s->world, p->maxBounce, p->tMin, rng);
          pixelColor = pixelColor + c;  // main.cpp:217
        }
        if (accum) {
          float* a = &accum[((size_t)y * W + x) * 4];
          a[0] = pixelColor(0); a[1] = pixelColor(1); a[2] = pixelColor(2); a[3] = (float)p->spp;
        }
        if (rgba) writeColorTarget(rgba, x, y, W, pixelColor.e, p->spp);
      }
    }
#pragma omp critical
    {
      total.add(local);
      draws += rng.draws;
    }
    tlsCounters = nullptr;
  }
  if (stats) {
    stats->samples = total.samples; stats->rays = total.rays; stats->nodeVisits = total.nodeVisits;
    stats->boxPasses = total.boxPasses; stats->triCalls = total.triCalls; stats->sphereCalls = total.sphereCalls;
    stats->dupTri = total.dupTri; stats->dupSphere = total.dupSphere; stats->shadedTriHits = total.shadedTriHits;
    stats->texelFetches = total.texelFetches; stats->rngDraws = draws;
  }
  return 0;
}

// color.h:25-41 on an accumulator image
void orc_resolve(const float* accum, int W, int H, int spp, uint8_t* rgba) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) writeColorTarget(rgba, x, y, W, &accum[((size_t)y * W + x) * 4], spp);
}

}  // extern "C"
