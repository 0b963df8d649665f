// srt_api.cpp -- host side of the C ABI (include/srt_hip.h): context, scene flattening,
// the reference-order BVH build, uploads, launches and timing.
//
// Host arithmetic that feeds the kernels (BVH boxes, per-triangle normals and tangent
// frames, the camera frame) keeps the reference's operation order; this file is built
// with -ffp-contract=off and without -march so it rounds like the reference's x86-64 build.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <queue>
#include <random>
#include <string>
#include <vector>

#include "srt_device.h"
#include "srt_thread.h"

extern "C" {
int srt_launch_render(const RenderArgs* a, int traversal, int count, int ldsTree, int grid, size_t ldsBytes, hipStream_t stream);
int srt_render_occupancy(int traversal, int count, int ldsTree, size_t ldsBytes, int* blocksPerCU);
int srt_launch_render_wf(const RenderArgs* a, int profile, int grid, size_t ldsBytes, hipStream_t stream);
int srt_launch_finalize(const SrtFixedAccum* fix, float4* out, int n, int samples, hipStream_t stream);
int srt_launch_sum_chunks(const float4* buf, float4* out, int n, int chunks, float limit, hipStream_t stream);
int srt_launch_resolve(const ResolveArgs* a, hipStream_t stream);
int srt_launch_trace(const TraceArgs* a, int traversal, int grid, size_t ldsBytes, hipStream_t stream);
int srt_lbvh_build(const DevScene* sc, const int32_t* dRefs, int n, float time0, float time1, float4* outNodes,
                   uint8_t* outAxis, int base, int* depthOut);
int srt_ploc_build(const DevScene* sc, const int32_t* dRefs, int n, float time0, float time1, float4* outNodes,
                   uint8_t* outAxis, int base, int radius, int* depthOut);
int srt_pair_nodes(const DevScene* sc, float time0, float time1, float4* out);
int srt_wide_nodes(const float4* nodes2, int numNodes, float4* out);
int srt_launch_scatter(const DevScene* sc, const SrtRay* rays, const SrtHit* hits, float* out, uint64_t seed, int n,
                       hipStream_t stream);
}

namespace {

// ------------------------------------------------------------------ host vector math
struct H3 {
  float x, y, z;
};
inline H3 h3(const float* p) { return H3{p[0], p[1], p[2]}; }
inline H3 operator+(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 operator-(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 operator*(float s, H3 a) { return H3{s * a.x, s * a.y, s * a.z}; }
inline H3 operator/(H3 a, float s) { return H3{a.x / s, a.y / s, a.z / s}; }
inline H3 crossH(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float lenSqH(H3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }  // vec3.h:29-31
inline H3 unitH(H3 v) {                                                    // vec3.h:54-60
  float len = sqrtf(lenSqH(v));
  if (len != 0) return H3{v.x / len, v.y / len, v.z / len};
  return v;
}

struct Box {
  float mn[3], mx[3];
};
inline Box surrounding(const Box& a, const Box& b) {  // aabb.h:33-43
  Box r;
  for (int k = 0; k < 3; ++k) {
    r.mn[k] = fminf(a.mn[k], b.mn[k]);
    r.mx[k] = fmaxf(a.mx[k], b.mx[k]);
  }
  return r;
}

// ------------------------------------------------------------------ the global generator
// globals.h:30-35: function-local static default-seeded mt19937 + uniform_real_distribution<float>(0,1)
std::mt19937& hostGenerator() {
  static std::mt19937 generator;
  return generator;
}
float hostRandomFloat() {
  static std::uniform_real_distribution<float> distribution(0.0f, 1.0f);
  return distribution(hostGenerator());
}
int hostRandomInt(int lo, int hi) {  // globals.h:37-43
  float a = (float)lo, b = (float)(hi + 1);
  return static_cast<int>(a + (b - a) * hostRandomFloat());
}

// ------------------------------------------------------------------ primitives on the host
Box sphereBoxAt(const SrtSphereIn& s, float time) {  // sphere.h:47-52, 86-89
  H3 c0 = h3(s.center0), c1 = h3(s.center1);
  H3 c = c0;
  if (c0.x != c1.x || c0.y != c1.y || c0.z != c1.z) c = c0 + ((time - s.time0) / (s.time1 - s.time0)) * (c1 - c0);
  Box b;
  b.mn[0] = c.x - s.radius; b.mn[1] = c.y - s.radius; b.mn[2] = c.z - s.radius;
  b.mx[0] = c.x + s.radius; b.mx[1] = c.y + s.radius; b.mx[2] = c.z + s.radius;
  return b;
}
Box sphereBox(const SrtSphereIn& s, float t0, float t1) {  // sphere.h:85-94
  return surrounding(sphereBoxAt(s, t0), sphereBoxAt(s, t1));
}
Box triangleBox(const SrtTriangleIn& t) {  // model.h:183-212
  const float inf = std::numeric_limits<float>::infinity();
  Box b;
  for (int a = 0; a < 3; ++a) {
    b.mn[a] = inf;
    b.mx[a] = -inf;
  }
  for (int k = 0; k < 3; ++k)
    for (int a = 0; a < 3; ++a) {
      b.mn[a] = std::min(b.mn[a], t.p[k][a]);
      b.mx[a] = std::max(b.mx[a], t.p[k][a]);
    }
  for (int a = 0; a < 3; ++a)
    if (b.mn[a] == b.mx[a]) {
      b.mn[a] -= 0.0001f;
      b.mx[a] += 0.0001f;
    }
  return surrounding(b, b);
}

// ------------------------------------------------------------------ bvh.h:55-95
struct BuildNode {
  Box box;
  int32_t left, right;  // >= 0 node, < 0 ~primListIndex
  uint8_t axis = 3;     // split axis (left child = lower box minimum on it), 3 = unknown
};

struct Builder {
  const SrtSceneDesc* d;
  float time0, time1;
  std::vector<float> sortKey;    // boundingBox(0,0).minimum per prim (boxCompare, bvh.h:34-41), 3 per prim
  std::vector<int32_t> objects;  // prim list indices; the reference's `objects` vector
  std::vector<BuildNode> nodes;
  int maxPending = 0;

  Box primBox(int32_t prim, float t0, float t1) const {
    const SrtPrimRef& pr = d->prims[prim];
    return pr.type == SRT_PRIM_SPHERE ? sphereBox(d->spheres[pr.index], t0, t1) : triangleBox(d->triangles[pr.index]);
  }
  Box childBox(int32_t ref) const { return ref >= 0 ? nodes[ref].box : primBox(~ref, time0, time1); }

  // The reference copies the object vector at each node (bvh.h:57) and sorts [start,end)
  // of the copy; sibling subtrees only touch disjoint sub-ranges, so one shared vector
  // sorted in place gives the same tree.  Nodes are emitted in pre-order (the order
  // populateVector walks them, bvh.h:112-148).
  int32_t build(size_t start, size_t end, int pending) {
    int axis = hostRandomInt(0, 2);  // bvh.h:60: one draw per node, pre-order
    auto comparator = [this, axis](int32_t a, int32_t b) { return sortKey[3 * a + axis] < sortKey[3 * b + axis]; };
    int32_t me = (int32_t)nodes.size();
    nodes.emplace_back();
    size_t span = end - start;
    int32_t left, right;
    if (span == 1) {
      left = right = ~objects[start];
    } else if (span == 2) {
      if (comparator(objects[start], objects[start + 1])) {
        left = ~objects[start];
        right = ~objects[start + 1];
      } else {
        left = ~objects[start + 1];
        right = ~objects[start];
      }
      maxPending = std::max(maxPending, pending + 1);
    } else {
      std::sort(objects.begin() + start, objects.begin() + end, comparator);
      size_t mid = start + span / 2;
      maxPending = std::max(maxPending, pending + 1);
      // capacity must hold for either visiting order (CLOSEST descends into the near child first and
      // leaves the other one pending), so both children are entered with one more pending entry
      left = build(start, mid, pending + 1);
      right = build(mid, end, pending + 1);
    }
    BuildNode& n = nodes[me];
    n.left = left;
    n.right = right;
    n.axis = (uint8_t)axis;
    n.box = surrounding(childBox(left), childBox(right));  // bvh.h:88-94
    return me;
  }
};

// make_shared<bvhNode>(objects, time0, time1) over one world item (main.cpp:146, bvh.h:15-16),
// or adoption of a caller-built tree (validated by validateScene).
void buildItem(const SrtSceneDesc* d, const SrtWorldItem& it, Builder& b) {
  b.d = d;
  b.time0 = it.time0;
  b.time1 = it.time1;
  if (it.nodes) {
    b.nodes.resize(it.numNodes);
    std::vector<int> pending(it.numNodes, 0);  // stack entries held when the node is entered
    for (int i = 0; i < it.numNodes; ++i) {
      const SrtBvhNode& n = it.nodes[i];
      memcpy(b.nodes[i].box.mn, n.bmin, 12);
      memcpy(b.nodes[i].box.mx, n.bmax, 12);
      b.nodes[i].left = n.left;
      b.nodes[i].right = n.right;
      const bool two = n.right != n.left;
      if (two) b.maxPending = std::max(b.maxPending, pending[i] + 1);
      if (n.left >= 0) pending[n.left] = pending[i] + (two ? 1 : 0);
      if (two && n.right >= 0) pending[n.right] = pending[i] + 1;  // either child may be the one left pending
    }
    return;
  }
  b.sortKey.resize((size_t)d->numPrims * 3);
  b.objects.resize(it.count);
  for (int i = 0; i < it.count; ++i) {
    int prim = it.first + i;
    b.objects[i] = prim;
    Box bx = b.primBox(prim, 0, 0);  // boxCompare uses boundingBox(0, 0, ...) (bvh.h:37)
    for (int k = 0; k < 3; ++k) b.sortKey[3 * prim + k] = bx.mn[k];
  }
  b.nodes.reserve((size_t)it.count * 2);
  b.build(0, it.count, 0);
}

// ------------------------------------------------------------------ context
struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

// Diagnostic tunables of the work distribution and the wave scheduler.  Environment variables give the
// defaults ONCE, at srtCreate; srtSetTunable (include/srt_hip_test.h) changes them per context.  -1 = the
// library's own rule.
struct Tunables {
  int tileBlock, unitTiles, queues;
  int shadeMin, primMin, hitMin, fuseMin, nodeBurst;
  int plocRadius, fastDiv;
  int chunkScratchMb;
  int primAgainMin;
  int keepEighths;
  int ldsTree;
  int wavefront, wfPool, wfSwapMin, wfSwapBig, wfProfile;
  int wideNodes, attGlobal;
  int wfHybrid, wfResidentMax, wfFarRounds;
};

struct SrtContext {
  int device = 0;
  std::string error;
  Tunables tun{};
  hipDeviceProp_t prop;
  // device scene
  std::vector<DeviceBuffer> sceneBuffers;
  DevScene scene{};
  DevCamera cam{};
  bool haveScene = false, haveCamera = false;
  // host copies for srtGetBvh
  std::vector<std::vector<SrtBvhNode>> itemNodes;
  struct DeviceTree { int32_t base = -1, count = 0; };
  std::vector<DeviceTree> itemDeviceTree;  // LBVH items: where their nodes live in scene.nodes
  std::vector<int32_t> hostTriPrimId, hostSphPrimId;
  int bvhDepth = 0;
  // work areas
  int32_t* dQueue = nullptr;
  unsigned long long* dStats = nullptr;
  void* comm = nullptr;          // ncclComm_t (srt_comm.cpp)
  int commRanks[2] = {1, 0};     // number of ranks, this rank
  SrtAovRecord* dAov = nullptr;  // set only for the duration of srtRenderAov
  int32_t aovDepth = 0;
  DeviceBuffer chunkScratch;
  DeviceBuffer attScratch;  // LDS-resident-tree kernel: the lanes' attenuation stacks (srt_render_kernel LDSTREE)
  DeviceBuffer wfPool, wfAttHi;  // path-pool kernel: contexts and upper attenuation levels (srt_wavefront.hip)
  int32_t* dWfError = nullptr;
  DeviceBuffer tileTable;   // RenderArgs::tileXY for the image size and tile order below
  int32_t tileTableKey[3] = {0, 0, 0};
  int32_t lastLaunch[4] = {0, 0, 0, 0};  // srtGetLaunchInfo
  hipEvent_t evStart = nullptr, evStop = nullptr;
  bool timed = false;
  SrtStats lastStats{};
};

namespace {

int fail(SrtContext* ctx, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf;
  fprintf(stderr, "srt_hip: %s\n", buf);  // the reference reports on std::cerr (texture.h:64-67, bvh.h:37-38)
  return 1;
}

// A path-pool launch that gave up (a ring wait exceeded its bound, srt_wavefront.hip) has added to the context's error
// word: the frame is incomplete.  Checked wherever the host has waited for the device anyway.
int wfCheck(SrtContext* ctx) {
  if (ctx->dWfError && *(volatile int32_t*)ctx->dWfError != 0) {
    const int n = *(volatile int32_t*)ctx->dWfError;
    *(volatile int32_t*)ctx->dWfError = 0;
    return fail(ctx, "render: %d workgroup(s) of the path-pool kernel gave up waiting on a ring; the frame is incomplete", n);
  }
  return 0;
}

#define HIP_OK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) return fail(ctx, "%s -> %s", #call, hipGetErrorString(e_));       \
  } while (0)

template <typename T>
int uploadVec(SrtContext* ctx, const std::vector<T>& v, const T** out, size_t padBytes = 0) {
  DeviceBuffer b;
  b.bytes = std::max<size_t>(v.size() * sizeof(T) + padBytes, 16);
  HIP_OK(ctx, hipMalloc(&b.p, b.bytes));
  HIP_OK(ctx, hipMemset(b.p, 0, b.bytes));
  if (!v.empty()) HIP_OK(ctx, hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ctx->sceneBuffers.push_back(b);
  *out = static_cast<const T*>(b.p);
  return 0;
}

void freeScene(SrtContext* ctx) {
  for (auto& b : ctx->sceneBuffers) (void)hipFree(b.p);
  ctx->sceneBuffers.clear();
  ctx->haveScene = false;
}

int envInt(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

struct TunableName {
  const char* name;
  const char* env;
  int Tunables::*field;
  int dflt;
};
const TunableName kTunables[] = {
    {"tile_block", nullptr, &Tunables::tileBlock, SRT_TILE_BLOCK},  // no environment override: ranks must agree (tiles.py)
    {"unit_tiles", "SRT_UNIT_TILES", &Tunables::unitTiles, -1},
    {"queues", "SRT_QUEUES", &Tunables::queues, -1},
    {"shade_min", "SRT_SHADE_MIN", &Tunables::shadeMin, -1},    // -1: by traversal mode, see srtRenderTiles
    {"prim_min", "SRT_PRIM_MIN", &Tunables::primMin, 12},
    {"hit_min", "SRT_HIT_MIN", &Tunables::hitMin, 24},
    {"fuse_min", "SRT_FUSE_MIN", &Tunables::fuseMin, 32},
    {"node_burst", "SRT_NODE_BURST", &Tunables::nodeBurst, -1},
    {"ploc_radius", "SRT_PLOC_RADIUS", &Tunables::plocRadius, 64},
    {"fast_div", "SRT_FAST_DIV", &Tunables::fastDiv, 1},
    {"prim_again_min", "SRT_PRIM_AGAIN_MIN", &Tunables::primAgainMin, 4},
    {"keep_eighths", "SRT_KEEP_EIGHTHS", &Tunables::keepEighths, -1},
    {"chunk_scratch_mb", "SRT_CHUNK_SCRATCH_MB", &Tunables::chunkScratchMb, 12288},  // budget of the chunk-slot path
    {"lds_tree", "SRT_LDS_TREE", &Tunables::ldsTree, 1},  // FAITHFUL: node records in LDS when the whole array fits and has this many nodes; 0 = never
    // the path-pool kernel (srt_wavefront.hip) for the launches the LDS-resident tree serves whose tree has at least this
    // many nodes (traversal-heavy frames gain, shading-heavy ones with tiny trees lose: profiles/r03/wavefront.txt); 0 = never
    {"wavefront", "SRT_WAVEFRONT", &Tunables::wavefront, 256},
    {"wf_pool", "SRT_WF_POOL", &Tunables::wfPool, 2048},       // path contexts per workgroup (1024 lanes traverse)
    // lanes without a walk before a wave swaps finished walks for READY contexts; 0 = 32, and 16 in the hybrid form, whose
    // lanes are worth more refilled than waiting (profiles/r03/hybrid.txt)
    {"wf_swap_min", "SRT_WF_SWAP_MIN", &Tunables::wfSwapMin, 0},
    {"wf_swap_big", "SRT_WF_SWAP_BIG", &Tunables::wfSwapBig, 32},
    {"wf_profile", "SRT_WF_PROFILE", &Tunables::wfProfile, 0},  // 1: the profiling variant (tools/wf_profile.py, srtGetWfProfile)
    // closest-hit traversal: 128-byte records with four boxes (1, read at srtUploadScene) instead of the 64-byte two-box
    // records (0).  Measured, not faster: the same bytes in half as many, twice as large random requests, at 2 instead
    // of 3 workgroups per CU (three pending references per level) -- profiles/r03/wide_nodes.txt.  Off.
    {"wide_nodes", "SRT_WIDE_NODES", &Tunables::wideNodes, 0},
    // closest-hit traversal of trees with at least this many nodes: attenuation stacks in global memory (4 instead of 3
    // workgroups per CU); 0 = never.  +3 % at 4 M triangles, -5 % at 10 M (same file).  Off.
    {"att_global", "SRT_ATT_GLOBAL", &Tunables::attGlobal, 0},
    // path-pool kernel over a tree that does not fit into LDS: its top (the boxes with the largest surface, closed upward)
    // stays in LDS, the rest is read from global memory (srt_wavefront.hip HYBRID).  wf_hybrid 0 = never (such scenes run
    // the 256-thread kernel); wf_resident_max > 0 caps the resident nodes, which also sends trees that WOULD fit down
    // this path (the parity tests set it to a few dozen).  Both are read at srtUploadScene.
    {"wf_hybrid", "SRT_WF_HYBRID", &Tunables::wfHybrid, 1},
    {"wf_resident_max", "SRT_WF_RESIDENT_MAX", &Tunables::wfResidentMax, 0},
    // hybrid form: node visits per round, 1..4 (srt_wavefront.hip): the first of two overlaps the other lanes' loads.
    // 0 = 2 for trees of up to 2^20 nodes (cache-resident: +10 % and more), 1 beyond (HBM-bound soups of 4 M and 10 M
    // triangles lose 5-10 % with 2) -- profiles/r03/hybrid.txt
    {"wf_far_rounds", "SRT_WF_FAR_ROUNDS", &Tunables::wfFarRounds, 0},
};

size_t ldsBytesFor(const SrtContext* ctx, int maxBounce, int stackDepth, bool attGlobal = false) {
  // per-thread stacks plus one word of queue state per wave (srt_render_kernel)
  return (size_t)(stackDepth + 2 + (attGlobal ? 0 : 3 * maxBounce + 3)) * 256 * sizeof(int32_t) + 4 * sizeof(int32_t);
}

}  // namespace


namespace {
// every index the kernels (and the host builder) will follow
int validateScene(SrtContext* ctx, const SrtSceneDesc* d) {
  // counts and pointers first: everything below indexes these arrays and sizes std::vectors with the counts
  if (d->numTriangles < 0 || d->numSpheres < 0 || d->numPrims < 0 || d->numWorld < 0 || d->numMaterials < 0 ||
      d->numTextures < 0 || d->numTexelBytes < 0)
    return fail(ctx, "scene: negative element count");
  if ((d->numTriangles > 0 && !d->triangles) || (d->numSpheres > 0 && !d->spheres) || (d->numPrims > 0 && !d->prims) ||
      (d->numWorld > 0 && !d->world) || (d->numMaterials > 0 && !d->materials) || (d->numTextures > 0 && !d->textures) ||
      (d->numTexelBytes > 0 && !d->texels))
    return fail(ctx, "scene: null array with a non-zero count");
  if ((int64_t)d->numTriangles > 0x3fffffff || (int64_t)d->numSpheres > 0x3fffffff)
    return fail(ctx, "scene: too many primitives for 31-bit device references");
  for (int i = 0; i < d->numTextures; ++i) {
    const SrtTextureIn& t = d->textures[i];
    if (t.kind == SRT_TEX_CHECKER) {
      for (int c : {t.even, t.odd})
        if (c < 0 || c >= d->numTextures || d->textures[c].kind == SRT_TEX_CHECKER)
          return fail(ctx, "texture %d: checker children must be solid or image textures", i);
    } else if (t.kind == SRT_TEX_IMAGE) {
      if (t.width < 0 || t.height < 0 || (t.width > 0 && (t.bpp < 1 || t.bpp > 4)))
        return fail(ctx, "texture %d: bad image dimensions", i);
      if (t.width > 0 && (t.texelOffset < 0 || t.texelOffset + (int64_t)t.width * t.height * t.bpp > d->numTexelBytes))
        return fail(ctx, "texture %d: texels out of range", i);
    } else if (t.kind != SRT_TEX_SOLID)
      return fail(ctx, "texture %d: unknown kind %d", i, t.kind);
  }
  auto texOk = [&](int id) { return id >= -1 && id < d->numTextures; };
  for (int i = 0; i < d->numMaterials; ++i) {
    const SrtMaterialIn& m = d->materials[i];
    if (m.type < SRT_MAT_PBR || m.type > SRT_MAT_LIGHT) return fail(ctx, "material %d: unknown type %d", i, m.type);
    if (!texOk(m.albedoTex) || !texOk(m.normalTex) || !texOk(m.metallicTex) || !texOk(m.roughnessTex))
      return fail(ctx, "material %d: texture id out of range", i);
    if (m.type == SRT_MAT_LIGHT && m.albedoTex < 0) return fail(ctx, "material %d: light without emit texture", i);
  }
  for (int i = 0; i < d->numTriangles; ++i)
    if (d->triangles[i].material < 0 || d->triangles[i].material >= d->numMaterials)
      return fail(ctx, "triangle %d: material out of range", i);
  for (int i = 0; i < d->numSpheres; ++i)
    if (d->spheres[i].material < 0 || d->spheres[i].material >= d->numMaterials)
      return fail(ctx, "sphere %d: material out of range", i);
  for (int i = 0; i < d->numPrims; ++i) {
    const SrtPrimRef& p = d->prims[i];
    int lim = p.type == SRT_PRIM_SPHERE ? d->numSpheres : p.type == SRT_PRIM_TRIANGLE ? d->numTriangles : -1;
    if (p.index < 0 || p.index >= lim) return fail(ctx, "prim %d: bad type/index", i);
  }
  if (d->numWorld < 1) return fail(ctx, "scene has an empty world list");
  for (int w = 0; w < d->numWorld; ++w) {
    const SrtWorldItem& it = d->world[w];
    if (it.first < 0 || it.count < 1 || it.first + it.count > d->numPrims || (it.kind != SRT_WORLD_PRIM && it.kind != SRT_WORLD_BVH))
      return fail(ctx, "world item %d: bad range", w);
    if (it.kind == SRT_WORLD_BVH && it.nodes) {
      // a caller-built tree must be finite and acyclic: children point forward (pre-order)
      // ... and a tree: every node but the root has exactly one parent (the pending-stack capacity is
      // derived per node from its single parent, buildItem)
      if (it.numNodes < 1) return fail(ctx, "world item %d: prebuilt tree without nodes", w);
      std::vector<uint8_t> parents(it.numNodes, 0);
      for (int i = 0; i < it.numNodes; ++i) {
        const int32_t l = it.nodes[i].left, r = it.nodes[i].right;
        for (int32_t c : {l, r}) {
          if (c >= 0 ? (c <= i || c >= it.numNodes) : (~c < it.first || ~c >= it.first + it.count))
            return fail(ctx, "world item %d: prebuilt node %d has a bad child reference %d", w, i, c);
        }
        if (l >= 0 && ++parents[l] > 1) return fail(ctx, "world item %d: prebuilt node %d has more than one parent", w, l);
        if (r >= 0 && r != l && ++parents[r] > 1) return fail(ctx, "world item %d: prebuilt node %d has more than one parent", w, r);
        if (r >= 0 && r == l) return fail(ctx, "world item %d: prebuilt node %d lists one subtree twice", w, i);
      }
      for (int i = 1; i < it.numNodes; ++i)
        if (!parents[i]) return fail(ctx, "world item %d: prebuilt node %d is unreachable", w, i);
    }
  }

  return 0;
}
}  // namespace

extern "C" {

int srtCreate(int deviceOrdinal, SrtContext** out) {
  if (!out) return 1;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, "no HIP device available (%s)", hipGetErrorString(e));
  if (deviceOrdinal < 0 || deviceOrdinal >= n) return fail(nullptr, "device ordinal %d out of range [0,%d)", deviceOrdinal, n);
  SrtContext* ctx = new SrtContext();
  ctx->device = deviceOrdinal;
  for (const TunableName& t : kTunables) ctx->tun.*(t.field) = t.env ? envInt(t.env, t.dflt) : t.dflt;
  HIP_OK(ctx, hipSetDevice(deviceOrdinal));
  HIP_OK(ctx, hipGetDeviceProperties(&ctx->prop, deviceOrdinal));
  HIP_OK(ctx, hipMalloc((void**)&ctx->dQueue, SRT_MAX_QUEUES * 16 * sizeof(int32_t)));
  HIP_OK(ctx, hipMalloc((void**)&ctx->dStats, 96 * sizeof(unsigned long long)));
  HIP_OK(ctx, hipEventCreate(&ctx->evStart));
  HIP_OK(ctx, hipEventCreate(&ctx->evStop));
  // the word a path-pool workgroup that gave up adds to: host memory the device writes, so that the host can look at it
  // without a synchronisation of its own (wfCheck)
  HIP_OK(ctx, hipHostMalloc((void**)&ctx->dWfError, sizeof(int32_t), hipHostMallocMapped));
  *ctx->dWfError = 0;
  *out = ctx;
  return 0;
}

// srt_comm.cpp's view of the context
int srtCtxFail(SrtContext* ctx, const char* text) { return fail(ctx, "%s", text); }
int srtCtxDevice(const SrtContext* ctx) { return ctx->device; }
void** srtCtxCommSlot(SrtContext* ctx) { return &ctx->comm; }
int* srtCtxCommRanks(SrtContext* ctx) { return ctx->commRanks; }
int srtCommDestroy(SrtContext* ctx);

int srtDestroy(SrtContext* ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)srtCommDestroy(ctx);
  freeScene(ctx);
  if (ctx->chunkScratch.p) (void)hipFree(ctx->chunkScratch.p);
  if (ctx->attScratch.p) (void)hipFree(ctx->attScratch.p);
  if (ctx->wfPool.p) (void)hipFree(ctx->wfPool.p);
  if (ctx->wfAttHi.p) (void)hipFree(ctx->wfAttHi.p);
  if (ctx->dWfError) (void)hipHostFree(ctx->dWfError);
  if (ctx->tileTable.p) (void)hipFree(ctx->tileTable.p);
  if (ctx->dQueue) (void)hipFree(ctx->dQueue);
  if (ctx->dStats) (void)hipFree(ctx->dStats);
  if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
  if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
  delete ctx;
  return 0;
}

const char* srtLastError(const SrtContext* ctx) { return ctx ? ctx->error.c_str() : "no context"; }

float srtHostRandomFloat(void) { return hostRandomFloat(); }
void srtHostRandomReset(void) { hostGenerator().seed(std::mt19937::default_seed); }

// camera.h:10-38
int srtMakeCamera(const SrtCameraParams* in, SrtCamera* out) {
  if (!in || !out) return 1;
  const float pi = 3.1415926535897932385f;
  H3 eye = h3(in->eye), lookAt = h3(in->lookAt), up = h3(in->up);
  float theta = in->vfovDegrees * pi / 180.0f;  // deg2rad, globals.h:26-28
  double h = tan((double)(theta / 2.0f));       // camera.h:20: tan(float) is the double overload
  double vpHeight = 2.0f * h;
  double vpWidth = in->aspect * vpHeight;
  H3 w = unitH(eye - lookAt);
  H3 hor = unitH(crossH(up, w));
  H3 vert = unitH(crossH(w, hor));
  H3 horizontal = (float)(in->focusDist * vpWidth) * hor;  // Eigen casts the double scalar to float
  H3 vertical = (float)(in->focusDist * vpHeight) * vert;
  H3 lleft = eye - horizontal / 2.0f - vertical / 2.0f - in->focusDist * w;
  const H3* src[] = {&eye, &lleft, &horizontal, &vertical, &w, &hor, &vert};
  float* dst[] = {out->origin, out->lleft, out->horizontal, out->vertical, out->w, out->hor, out->vert};
  for (int i = 0; i < 7; ++i) {
    dst[i][0] = src[i]->x;
    dst[i][1] = src[i]->y;
    dst[i][2] = src[i]->z;
  }
  out->lensRadius = in->aperture / 2.0f;
  out->time0 = in->time0;
  out->time1 = in->time1;
  return 0;
}

int srtSetCamera(SrtContext* ctx, const SrtCamera* c) {
  if (!ctx || !c) return 1;
  DevCamera& d = ctx->cam;
  memcpy(d.origin, c->origin, 12); memcpy(d.lleft, c->lleft, 12); memcpy(d.horizontal, c->horizontal, 12);
  memcpy(d.vertical, c->vertical, 12); memcpy(d.hor, c->hor, 12); memcpy(d.vert, c->vert, 12);
  d.lensRadius = c->lensRadius;
  d.time0 = c->time0;
  d.time1 = c->time1;
  ctx->haveCamera = true;
  return 0;
}

static int srtUploadSceneImpl(SrtContext* ctx, const SrtSceneDesc* d) {
  if (!ctx || !d) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  freeScene(ctx);
  ctx->itemNodes.clear();
  ctx->itemDeviceTree.clear();
  ctx->bvhDepth = 0;

  if (validateScene(ctx, d)) return 1;

  // ---- primitive records.  Device arrays are indexed by the scene's own triangle /
  // sphere indices; the owning list index is kept for srtTraceRays' prim output.
  std::vector<int32_t> triPrimId(d->numTriangles, -1), sphPrimId(d->numSpheres, -1);
  for (int i = 0; i < d->numPrims; ++i) {
    const SrtPrimRef& p = d->prims[i];
    (p.type == SRT_PRIM_SPHERE ? sphPrimId : triPrimId)[p.index] = i;
  }
  std::vector<float4> triTest((size_t)d->numTriangles * 3), triShade((size_t)d->numTriangles * 4);
  const float eps = std::numeric_limits<float>::epsilon();
  for (int i = 0; i < d->numTriangles; ++i) {
    const SrtTriangleIn& t = d->triangles[i];
    H3 v0 = h3(t.p[0]), v1 = h3(t.p[1]), v2 = h3(t.p[2]);
    H3 n = crossH(v1 - v0, v2 - v0);  // getNormal, model.h:276-283
    triTest[3 * i + 0] = make_float4(v0.x, v0.y, v0.z, n.x);
    triTest[3 * i + 1] = make_float4(v1.x, v1.y, v1.z, n.y);
    triTest[3 * i + 2] = make_float4(v2.x, v2.y, v2.z, n.z);
    H3 nu = unitH(n);  // model.h:172
    // calcTangentBasis, model.h:214-235
    H3 e0 = v1 - v0, e1 = v2 - v0;
    float du0 = t.uv[1][0] - t.uv[0][0], dv0 = t.uv[1][1] - t.uv[0][1];
    float du1 = t.uv[2][0] - t.uv[0][0], dv1 = t.uv[2][1] - t.uv[0][1];
    float f = (du0 * dv1 - du1 * dv0);
    if (f == 0) f += eps;
    f = 1.0f / f;
    H3 tg = unitH(H3{f * (dv1 * e0.x - dv0 * e1.x), f * (dv1 * e0.y - dv0 * e1.y), f * (dv1 * e0.z - dv0 * e1.z)});
    H3 bt = unitH(H3{f * (-du1 * e0.x + du0 * e1.x), f * (-du1 * e0.y + du0 * e1.y), f * (-du1 * e0.z + du0 * e1.z)});
    float matBits;
    int32_t mat = t.material;
    memcpy(&matBits, &mat, 4);
    triShade[4 * i + 0] = make_float4(nu.x, nu.y, nu.z, t.uv[0][0]);
    triShade[4 * i + 1] = make_float4(tg.x, tg.y, tg.z, t.uv[0][1]);
    triShade[4 * i + 2] = make_float4(bt.x, bt.y, bt.z, t.uv[1][0]);
    triShade[4 * i + 3] = make_float4(t.uv[1][1], t.uv[2][0], t.uv[2][1], matBits);
  }
  std::vector<float4> spheres((size_t)d->numSpheres * 3);
  for (int i = 0; i < d->numSpheres; ++i) {
    const SrtSphereIn& s = d->spheres[i];
    bool moving = s.center0[0] != s.center1[0] || s.center0[1] != s.center1[1] || s.center0[2] != s.center1[2];
    int32_t bits = s.material | (moving ? (1 << 30) : 0);
    float fb;
    memcpy(&fb, &bits, 4);
    spheres[3 * i + 0] = make_float4(s.center0[0], s.center0[1], s.center0[2], s.radius);
    spheres[3 * i + 1] = make_float4(s.center1[0], s.center1[1], s.center1[2], fb);
    spheres[3 * i + 2] = make_float4(s.time0, s.time1, 0.0f, 0.0f);
  }
  // device index of a triangle: identity until the trees are built, then the order in which the host-built
  // trees' leaves reference the triangles (below), so that a leaf's records and its neighbours' sit together
  std::vector<int32_t> triDevIndex;
  auto devRef = [&](int32_t listRef) -> int32_t {  // ~primListIndex -> device prim ref
    const SrtPrimRef& p = d->prims[~listRef];
    if (p.type == SRT_PRIM_SPHERE) return ~((p.index << 1) | 1);
    return ~((triDevIndex.empty() ? p.index : triDevIndex[p.index]) << 1);
  };

  // ---- world: build each bvhNode (consumes the global generator in scene order)
  std::vector<float4> nodes;
  std::vector<uint8_t> nodeAxis;
  std::vector<int32_t> world;
  int stackDepth = 0;
  for (int w = 0; w < d->numWorld; ++w) {
    const SrtWorldItem& it = d->world[w];
    ctx->itemNodes.emplace_back();
    ctx->itemDeviceTree.emplace_back();
    if (it.kind == SRT_WORLD_PRIM) {
      world.push_back(devRef(~it.first));
      continue;
    }
    if (!it.nodes && (it.builder == SRT_BUILDER_LBVH || it.builder == SRT_BUILDER_PLOC)) {
      // device build (srt_lbvh.hip) after the primitive arrays are uploaded: reserve the node slots
      int32_t base = (int32_t)(nodes.size() / 2), cnt = std::max(it.count - 1, 1);
      nodes.resize(nodes.size() + 2 * (size_t)cnt, make_float4(0, 0, 0, 0));
      nodeAxis.resize(nodeAxis.size() + cnt, 3);
      ctx->itemDeviceTree.back().base = base;
      ctx->itemDeviceTree.back().count = cnt;
      world.push_back(SRT_NODE_REF(base));
      continue;
    }
    Builder b;
    buildItem(d, it, b);
    // (A device order with the two children of a node in one 64-byte line -- root, then sibling pairs depth-first --
    // was measured against this pre-order, where the LEFT child follows its parent: headline +0.3 %, 1 M-triangle
    // soup -5 %, 10 M +1.7 %, profiles/r02/node_pairs.txt.  Not kept.)
    int32_t base = (int32_t)(nodes.size() / 2);
    std::vector<SrtBvhNode>& hostNodes = ctx->itemNodes.back();
    hostNodes.resize(b.nodes.size());
    for (size_t i = 0; i < b.nodes.size(); ++i) {
      const BuildNode& n = b.nodes[i];
      SrtBvhNode& o = hostNodes[i];
      memcpy(o.bmin, n.box.mn, 12);
      memcpy(o.bmax, n.box.mx, 12);
      o.left = n.left;
      o.right = n.right;
      int32_t l = n.left >= 0 ? SRT_NODE_REF(n.left + base) : devRef(n.left);
      int32_t r = n.right >= 0 ? SRT_NODE_REF(n.right + base) : devRef(n.right);
      float lf, rf;
      memcpy(&lf, &l, 4);
      memcpy(&rf, &r, 4);
      nodes.push_back(make_float4(n.box.mn[0], n.box.mn[1], n.box.mn[2], lf));
      nodes.push_back(make_float4(n.box.mx[0], n.box.mx[1], n.box.mx[2], rf));
      nodeAxis.push_back(n.axis);
    }
    world.push_back(SRT_NODE_REF(base));
    stackDepth = std::max(stackDepth, b.maxPending);
    // tree depth for reporting: longest root->node chain
    std::vector<int> depth(b.nodes.size(), 1);
    for (size_t i = 0; i < b.nodes.size(); ++i) {  // pre-order: parents precede children
      ctx->bvhDepth = std::max(ctx->bvhDepth, depth[i]);
      if (b.nodes[i].left >= 0) depth[b.nodes[i].left] = depth[i] + 1;
      if (b.nodes[i].right >= 0) depth[b.nodes[i].right] = depth[i] + 1;
    }
  }

  // ---- materials / textures
  std::vector<DevMaterial> mats(d->numMaterials);
  for (int i = 0; i < d->numMaterials; ++i) {
    const SrtMaterialIn& m = d->materials[i];
    DevMaterial& o = mats[i];
    memset(&o, 0, sizeof o);
    o.type = m.type;
    o.albedoTex = m.albedoTex; o.normalTex = m.normalTex; o.metallicTex = m.metallicTex; o.roughnessTex = m.roughnessTex;
    memcpy(o.albedo, m.albedo, 16);
    o.metalness = m.type == SRT_MAT_METAL ? (m.fuzz < 1.0f ? m.fuzz : 1.0f)  // material.h:89
                  : m.type == SRT_MAT_DIELECTRIC ? m.ir : m.metalness;
    o.roughness = m.roughness;
    // which hitRecord fields this material can observe (srt_kernels.hip sphereRecord/triRecord)
    auto readsUv = [&](int tex) {
      if (tex < 0) return false;
      const SrtTextureIn& t = d->textures[tex];
      if (t.kind == SRT_TEX_IMAGE) return true;
      if (t.kind == SRT_TEX_CHECKER) return d->textures[t.even].kind == SRT_TEX_IMAGE || d->textures[t.odd].kind == SRT_TEX_IMAGE;
      return false;
    };
    bool uv = readsUv(m.albedoTex);
    if (m.type == SRT_MAT_PBR) uv = uv || readsUv(m.normalTex) || readsUv(m.metallicTex) || readsUv(m.roughnessTex);
    o.flags = (uv ? 1 : 0) | ((m.type == SRT_MAT_PBR && m.normalTex >= 0) ? 2 : 0);
  }
  // textures: 3-byte images are padded to one aligned dword per texel, so a lookup is one
  // buffer_load_dword; 1- and 2-byte images keep their byte rows (the 1-bpp quirk of texture.h:147 reads the
  // neighbouring texels)
  std::vector<DevTexture> texs(d->numTextures);
  std::vector<uint8_t> texels;
  for (int i = 0; i < d->numTextures; ++i) {
    const SrtTextureIn& t = d->textures[i];
    DevTexture& o = texs[i];
    memset(&o, 0, sizeof o);
    o.kind = t.kind; o.width = t.width; o.height = t.height; o.bpp = t.bpp;
    o.even = t.even; o.odd = t.odd;
    memcpy(o.color, t.color, 12);
    if (t.kind != SRT_TEX_IMAGE || t.width == 0) continue;
    const size_t n = (size_t)t.width * t.height;
    const uint8_t* src = d->texels + t.texelOffset;
    texels.resize((texels.size() + 3) & ~(size_t)3);  // dword aligned
    o.offset = (int64_t)texels.size();
    if (t.bpp == 3) {
      const size_t at = texels.size();
      texels.resize(at + 4 * n);
      for (size_t k = 0; k < n; ++k) {
        texels[at + 4 * k + 0] = src[3 * k + 0];
        texels[at + 4 * k + 1] = src[3 * k + 1];
        texels[at + 4 * k + 2] = src[3 * k + 2];
        texels[at + 4 * k + 3] = 255;
      }
    } else {
      texels.insert(texels.end(), src, src + n * t.bpp);
    }
    if (texels.size() > (size_t)0x7fffff00) return fail(ctx, "scene: more than 2 GiB of texels");
  }
  // ---- the material's flags ride in every primitive's material word (srt_device.h SRT_MAT_FLAGS_SHIFT), and the hit
  // step's 128-byte shading records (srt_kernels.hip shade)
  if (d->numMaterials > SRT_MAT_INDEX_MASK) return fail(ctx, "scene: more than %d materials", SRT_MAT_INDEX_MASK);
  auto withFlags = [&](float& word) {
    int32_t bits;
    memcpy(&bits, &word, 4);
    const int32_t m = bits & SRT_MAT_INDEX_MASK;
    if (m < d->numMaterials) {
      const DevMaterial& dm = mats[m];
      const bool textured = dm.type == SRT_MAT_PBR && (dm.albedoTex >= 0 || dm.normalTex >= 0 || dm.metallicTex >= 0 || dm.roughnessTex >= 0);
      bits |= (dm.flags & 3) << SRT_MAT_FLAGS_SHIFT | ((dm.type & 3) | (textured ? SRT_MAT_TEXTURED : 0)) << SRT_MAT_TYPE_SHIFT;
    }
    memcpy(&word, &bits, 4);
  };
  for (int i = 0; i < d->numTriangles; ++i) withFlags(triShade[4 * (size_t)i + 3].w);
  for (int i = 0; i < d->numSpheres; ++i) withFlags(spheres[3 * (size_t)i + 1].w);
  // material class per primitive reference (DevScene::primClass; filled after the triangles have their device order)
  auto classOf = [&](float word, bool sphere) -> uint8_t {
    int32_t bits;
    memcpy(&bits, &word, 4);
    const int type = (bits >> SRT_MAT_TYPE_SHIFT) & 3, flags = (bits >> SRT_MAT_FLAGS_SHIFT) & 3;
    if (type != SRT_MAT_PBR) return 2;
    if (!sphere) return 0;
    // a sphere whose pbr material reads uv or a normal map (the textured iron sphere: acosf / atan2f, a tangent frame,
    // four lookups) would make every hit step of the plain spheres (the ground) run that code too: it goes with "the rest"
    return flags ? 2 : 1;
  };
  std::vector<uint4> shadeRecs((size_t)8 * d->numMaterials, make_uint4(0, 0, 0, 0));
  for (int i = 0; i < d->numMaterials; ++i) {
    const DevMaterial& m = mats[i];
    auto f2u = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    uint4* r = &shadeRecs[(size_t)8 * i];
    const bool textured = m.type == SRT_MAT_PBR && (m.albedoTex >= 0 || m.normalTex >= 0 || m.metallicTex >= 0 || m.roughnessTex >= 0);
    r[0] = make_uint4((uint32_t)((m.type & 3) | (textured ? SRT_MAT_TEXTURED : 0)), (uint32_t)m.flags, f2u(m.metalness), f2u(m.roughness));
    r[1] = make_uint4(f2u(m.albedo[0]), f2u(m.albedo[1]), f2u(m.albedo[2]), f2u(m.albedo[3]));
    const int32_t ids[4] = {m.albedoTex, m.normalTex, m.metallicTex, m.roughnessTex};
    // +32: the emit texture of a light, in full
    if (m.type == SRT_MAT_LIGHT && ids[0] >= 0) {
      const DevTexture& t = texs[ids[0]];
      if (t.kind == SRT_TEX_SOLID)
        r[2] = make_uint4(1, f2u(t.color[0]), f2u(t.color[1]), f2u(t.color[2]));
      else if (t.kind == SRT_TEX_IMAGE && t.width == 0)
        r[2] = make_uint4(1, f2u(1.0f), f2u(0.0f), f2u(1.0f));  // failed load: magenta (texture.h:130-131)
      else if (t.kind == SRT_TEX_IMAGE && t.bpp >= 3)
        r[2] = make_uint4(2, (uint32_t)t.width, (uint32_t)t.height, (uint32_t)t.offset);
      else
        r[2] = make_uint4(3, (uint32_t)ids[0], 0, 0);  // checker, 1- and 2-byte images: texValue
    }
    // +48, +64: the four pbr slots, two dwords each
    uint32_t packed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) {
      if (ids[k] < 0) continue;  // mode 0: no texture
      const DevTexture& t = texs[ids[k]];
      if (t.kind == SRT_TEX_IMAGE && t.bpp >= 3 && t.width > 0 && t.width < 32768 && t.height < 32768) {
        packed[2 * k] = 2u | (uint32_t)t.width << 2 | (uint32_t)t.height << 17;
        packed[2 * k + 1] = (uint32_t)t.offset;
      } else {
        packed[2 * k] = 3u;  // everything else goes through texValue on the id
        packed[2 * k + 1] = (uint32_t)ids[k];
      }
    }
    // an albedo slot that is checker(solidColor, solidColor): both colours into the record (+80 even, +96 odd)
    if (m.type == SRT_MAT_PBR && ids[0] >= 0 && texs[ids[0]].kind == SRT_TEX_CHECKER) {
      const DevTexture& c = texs[ids[0]];
      if (c.even >= 0 && c.odd >= 0 && c.even < (int)texs.size() && c.odd < (int)texs.size() && texs[c.even].kind == SRT_TEX_SOLID &&
          texs[c.odd].kind == SRT_TEX_SOLID) {
        packed[0] = 7u;  // SRT_SLOT_CHECKER2
        r[5] = make_uint4(f2u(texs[c.even].color[0]), f2u(texs[c.even].color[1]), f2u(texs[c.even].color[2]), 0);
        r[6] = make_uint4(f2u(texs[c.odd].color[0]), f2u(texs[c.odd].color[1]), f2u(texs[c.odd].color[2]), 0);
      }
    }
    r[3] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    r[4] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
  }
  // ---- triangle records in tree order.  Device arrays were filled in the scene's own triangle order; a tree's
  // leaves reference them at random (a mesh's index order has nothing to do with the median splits), and with
  // millions of triangles every test is then a 48-byte gather from a cold place.  Renumber the triangles by
  // their first appearance in the (pre-order) node arrays of the host-built trees: the two triangles of a leaf
  // and the leaves of a subtree become neighbours in memory.  Triangles no host-built tree references keep
  // their relative order behind them.  References are rewritten, nothing else changes (primitive ids reported
  // by srtTraceRays go through triPrimId, which is permuted along).
  if (d->numTriangles > 1 && !nodes.empty()) {
    std::vector<int32_t> order(d->numTriangles, -1);
    int32_t next = 0;
    auto visit = [&](float bits) {
      int32_t r;
      memcpy(&r, &bits, 4);
      if (r >= 0 || r == SRT_REF_DONE) return;
      const int32_t pr = ~r;
      if ((pr & 1) == 0 && (pr >> 1) < d->numTriangles && order[pr >> 1] < 0) order[pr >> 1] = next++;
    };
    for (size_t i = 0; i + 1 < nodes.size(); i += 2) {
      visit(nodes[i].w);
      visit(nodes[i + 1].w);
    }
    if (next > 0) {
      for (int32_t i = 0; i < d->numTriangles; ++i)
        if (order[i] < 0) order[i] = next++;
      auto remap = [&](float& bits) {
        int32_t r;
        memcpy(&r, &bits, 4);
        if (r >= 0 || r == SRT_REF_DONE) return;
        const int32_t pr = ~r;
        if (pr & 1) return;
        r = ~(order[pr >> 1] << 1);
        memcpy(&bits, &r, 4);
      };
      for (size_t i = 0; i + 1 < nodes.size(); i += 2) {
        remap(nodes[i].w);
        remap(nodes[i + 1].w);
      }
      for (int32_t& wr : world) {
        float bits;
        memcpy(&bits, &wr, 4);
        remap(bits);
        memcpy(&wr, &bits, 4);
      }
      std::vector<float4> tt(triTest.size()), ts(triShade.size());
      std::vector<int32_t> tp(triPrimId.size());
      for (int32_t i = 0; i < d->numTriangles; ++i) {
        const int32_t j = order[i];
        for (int k = 0; k < 3; ++k) tt[3 * (size_t)j + k] = triTest[3 * (size_t)i + k];
        for (int k = 0; k < 4; ++k) ts[4 * (size_t)j + k] = triShade[4 * (size_t)i + k];
        tp[j] = triPrimId[i];
      }
      triTest.swap(tt);
      triShade.swap(ts);
      triPrimId.swap(tp);
      triDevIndex.swap(order);  // devRef of the device-built trees below
    }
  }
  // ---- thread links (srt_thread.h): the 16-bit form the LDS-resident-tree kernels walk (DevScene::nodeThread), and for a
  // tree that does not fit into LDS the path-pool kernel's hybrid records (32-bit references, resident nodes first)
  bool hostTrees = true;
  for (const auto& dt : ctx->itemDeviceTree) hostTrees = hostTrees && dt.base < 0;
  std::vector<int32_t> nodeThread;
  std::vector<float4> nodesWf;
  std::vector<int32_t> worldWf, primSecond;
  int32_t wfResident = 0;
  if (hostTrees && !nodes.empty()) {
    srtThreadLinks16(nodes, world, d->numTriangles, d->numSpheres, nodeThread);
    if (ctx->tun.wfHybrid > 0) {
      const size_t n = nodes.size() / 2;
      const size_t fits = (160 * 1024 - 64 * sizeof(int32_t) - 20 * 2048) / 32;       // beside a pool of 2048 contexts
      const size_t fitsWhole = (160 * 1024 - 64 * sizeof(int32_t) - 18 * 1024) / 32;  // the whole-tree form's smallest pool
      const size_t cap = ctx->tun.wfResidentMax > 0 ? std::min<size_t>(fits, (size_t)ctx->tun.wfResidentMax) : fits;
      if (n > (ctx->tun.wfResidentMax > 0 ? cap : fitsWhole))
        wfResident = srtHybridRecords(nodes, world, d->numTriangles, d->numSpheres, cap, nodesWf, worldWf, primSecond);
    }
  }
  std::vector<uint8_t> primClass((size_t)2 * std::max(d->numTriangles, d->numSpheres) + 2, 2);
  for (int i = 0; i < d->numTriangles; ++i) primClass[(size_t)i << 1] = classOf(triShade[4 * (size_t)i + 3].w, false);
  for (int i = 0; i < d->numSpheres; ++i) primClass[((size_t)i << 1) | 1] = classOf(spheres[3 * (size_t)i + 1].w, true);
  DevScene& s = ctx->scene;
  memset(&s, 0, sizeof s);
  if (!nodeThread.empty() && uploadVec(ctx, nodeThread, &s.nodeThread)) return 1;
  if (!nodesWf.empty() && (uploadVec(ctx, nodesWf, &s.nodesWf) || uploadVec(ctx, worldWf, &s.worldWf) || uploadVec(ctx, primSecond, &s.primSecond))) return 1;
  s.wfResident = wfResident;
  if (uploadVec(ctx, primClass, &s.primClass, 16)) return 1;
  s.numPrimClass = (int32_t)primClass.size();
  if (uploadVec(ctx, nodeAxis, &s.nodeAxis, 64) || uploadVec(ctx, nodes, &s.nodes) || uploadVec(ctx, triTest, &s.triTest) || uploadVec(ctx, triShade, &s.triShade) ||
      uploadVec(ctx, spheres, &s.spheres) || uploadVec(ctx, triPrimId, &s.triPrimId) ||
      uploadVec(ctx, sphPrimId, &s.sphPrimId) || uploadVec(ctx, world, &s.world) || uploadVec(ctx, mats, &s.materials) || uploadVec(ctx, shadeRecs, &s.shadeRecs) ||
      uploadVec(ctx, texs, &s.textures) || uploadVec(ctx, texels, &s.texels, 64))
    return 1;
  ctx->hostTriPrimId = triPrimId;
  ctx->hostSphPrimId = sphPrimId;
  // device-built trees
  bool lbvhCertificate = true;
  for (int w = 0; w < d->numWorld; ++w) {
    const SrtContext::DeviceTree& dt = ctx->itemDeviceTree[w];
    if (dt.base < 0) continue;
    const SrtWorldItem& it = d->world[w];
    std::vector<int32_t> refs(it.count);
    Builder pb;
    pb.d = d;
    for (int i = 0; i < it.count; ++i) {
      refs[i] = devRef(~(it.first + i));
      // node boxes of this tree are unions of primitive boxes: extend fastDiv's coordinate certificate to them
      Box bx = pb.primBox(it.first + i, it.time0, it.time1);
      for (int k = 0; k < 3; ++k)
        for (float c : {bx.mn[k], bx.mx[k]}) {
          float ac = fabsf(c);
          if (!(c == 0.0f || (ac >= 0x1p-77f && ac <= 0x1p30f))) lbvhCertificate = false;
        }
    }
    int32_t* dRefs = nullptr;
    HIP_OK(ctx, hipMalloc((void**)&dRefs, refs.size() * sizeof(int32_t)));
    hipError_t ce = hipMemcpy(dRefs, refs.data(), refs.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    int depth = 0;
    int rc = (int)ce;
    if (ce == hipSuccess)
      rc = it.builder == SRT_BUILDER_PLOC
               ? srt_ploc_build(&s, dRefs, it.count, it.time0, it.time1, const_cast<float4*>(s.nodes), const_cast<uint8_t*>(s.nodeAxis),
                                dt.base, ctx->tun.plocRadius, &depth)
               : srt_lbvh_build(&s, dRefs, it.count, it.time0, it.time1, const_cast<float4*>(s.nodes), const_cast<uint8_t*>(s.nodeAxis),
                                dt.base, &depth);
    (void)hipFree(dRefs);
    if (rc) return fail(ctx, "device BVH build of world item %d failed: %s", w, hipGetErrorString((hipError_t)rc));
    stackDepth = std::max(stackDepth, depth);
    ctx->bvhDepth = std::max(ctx->bvhDepth, depth);
  }
  s.numWorld = (int32_t)world.size();
  s.stackDepth = stackDepth;
  s.numNodes = (int32_t)(nodes.size() / 2);
  {
    // the closest-hit traversal's records: both children's boxes per node (srt_lbvh.hip pairNodes), built on the
    // device from the finished node array (host-built and device-built trees alike)
    float t0 = 0.0f, t1 = 0.0f;
    for (int w = 0; w < d->numWorld; ++w) {
      t0 = w ? std::min(t0, d->world[w].time0) : d->world[w].time0;
      t1 = w ? std::max(t1, d->world[w].time1) : d->world[w].time1;
    }
    DeviceBuffer b2;
    b2.bytes = std::max<size_t>((size_t)s.numNodes * 64, 64);
    HIP_OK(ctx, hipMalloc(&b2.p, b2.bytes));
    ctx->sceneBuffers.push_back(b2);
    s.nodes2 = static_cast<const float4*>(b2.p);
    int rc2 = srt_pair_nodes(&s, t0, t1, static_cast<float4*>(b2.p));
    if (rc2) return fail(ctx, "pairing the node records failed: %s", hipGetErrorString((hipError_t)rc2));
    // ... and the 128-byte records with four boxes, for trees big enough to be fetched from HBM (and small enough for
    // index * 128 to stay a 31-bit byte offset)
    if (ctx->tun.wideNodes > 0 && s.numNodes >= 2 && s.numNodes < (1 << 24)) {
      DeviceBuffer b4;
      b4.bytes = (size_t)s.numNodes * 128;
      HIP_OK(ctx, hipMalloc(&b4.p, b4.bytes));
      ctx->sceneBuffers.push_back(b4);
      rc2 = srt_wide_nodes(s.nodes2, s.numNodes, static_cast<float4*>(b4.p));
      if (rc2) return fail(ctx, "building the four-box records failed: %s", hipGetErrorString((hipError_t)rc2));
      s.nodes4 = static_cast<const float4*>(b4.p);
      s.wideStackDepth = 3 * ((ctx->bvhDepth + 1) / 2 + 1) + 1;
    }
  }
  const bool lbvhOk = lbvhCertificate;
  if (nodes.size() / 2 > (size_t)SRT_MAX_NODES) return fail(ctx, "scene: %zu BVH nodes exceed the %d the device references can address", nodes.size() / 2, SRT_MAX_NODES);
  s.texelBytes = (int32_t)texels.size();
  s.numNodes = (int32_t)(nodes.size() / 2);
  s.numTris = d->numTriangles;
  s.numSpheres = d->numSpheres;
  s.numMaterials = d->numMaterials;
  // fastDiv's operand certificate for the box coordinates (srt_kernels.hip): 0 or 2^-77 <= |c| <= 2^30
  s.fastDivScene = ctx->tun.fastDiv;
  for (const float4& v : nodes)
    for (float c : {v.x, v.y, v.z}) {
      float ac = fabsf(c);
      if (!(c == 0.0f || (ac >= 0x1p-77f && ac <= 0x1p30f))) s.fastDivScene = 0;
    }
  if (!lbvhOk) s.fastDivScene = 0;
  ctx->haveScene = true;
  return 0;
}

// Host-only: build world item `item` exactly as srtUploadScene does, without a device.
static int srtBuildBvhImpl(const SrtSceneDesc* d, int32_t item, SrtBvhNode* out, int32_t capacity, int32_t* count, int32_t* stackDepth) {
  if (!d || !count) return 1;
  if (validateScene(nullptr, d)) return 1;
  if (item < 0 || item >= d->numWorld || d->world[item].kind != SRT_WORLD_BVH) return fail(nullptr, "srtBuildBvh: item %d is not a bvh", item);
  Builder b;
  buildItem(d, d->world[item], b);
  *count = (int32_t)b.nodes.size();
  if (stackDepth) *stackDepth = b.maxPending;
  if (out) {
    if (capacity < *count) return fail(nullptr, "srtBuildBvh: capacity too small");
    for (size_t i = 0; i < b.nodes.size(); ++i) {
      memcpy(out[i].bmin, b.nodes[i].box.mn, 12);
      memcpy(out[i].bmax, b.nodes[i].box.mx, 12);
      out[i].left = b.nodes[i].left;
      out[i].right = b.nodes[i].right;
    }
  }
  return 0;
}

static int srtGetBvhImpl(SrtContext* ctx, int32_t item, SrtBvhNode* nodes, int32_t capacity, int32_t* count) {
  if (!ctx || !count) return 1;
  if (item < 0 || item >= (int32_t)ctx->itemNodes.size()) return fail(ctx, "srtGetBvh: item %d out of range", item);
  if (ctx->itemDeviceTree[item].base >= 0 && ctx->itemNodes[item].empty()) {
    // device-built tree: read it back once, converting child refs to the host convention
    const auto& dt = ctx->itemDeviceTree[item];
    std::vector<float4> raw((size_t)dt.count * 2);
    HIP_OK(ctx, hipMemcpy(raw.data(), ctx->scene.nodes + 2 * (size_t)dt.base, raw.size() * sizeof(float4), hipMemcpyDeviceToHost));
    auto& out = ctx->itemNodes[item];
    out.resize(dt.count);
    auto conv = [&](float bits) -> int32_t {
      int32_t r;
      memcpy(&r, &bits, 4);
      if (r >= 0) return SRT_NODE_INDEX(r) - dt.base;
      int32_t pr = ~r;
      return ~((pr & 1) ? ctx->hostSphPrimId[pr >> 1] : ctx->hostTriPrimId[pr >> 1]);
    };
    for (int i = 0; i < dt.count; ++i) {
      out[i].bmin[0] = raw[2 * i].x; out[i].bmin[1] = raw[2 * i].y; out[i].bmin[2] = raw[2 * i].z;
      out[i].bmax[0] = raw[2 * i + 1].x; out[i].bmax[1] = raw[2 * i + 1].y; out[i].bmax[2] = raw[2 * i + 1].z;
      out[i].left = conv(raw[2 * i].w);
      out[i].right = conv(raw[2 * i + 1].w);
    }
  }
  const auto& v = ctx->itemNodes[item];
  *count = (int32_t)v.size();
  if (nodes) {
    if (capacity < (int32_t)v.size()) return fail(ctx, "srtGetBvh: capacity %d < %zu", capacity, v.size());
    memcpy(nodes, v.data(), v.size() * sizeof(SrtBvhNode));
  }
  return 0;
}

int srtGetBvhDepth(SrtContext* ctx, int32_t* depth) {
  if (!ctx || !depth) return 1;
  *depth = ctx->bvhDepth;
  return 0;
}

int32_t srtNumTiles(int32_t w, int32_t h) {
  return ((w + SRT_TILE_W - 1) / SRT_TILE_W) * ((h + SRT_TILE_H - 1) / SRT_TILE_H);
}
int32_t srtNumLocalTiles(int32_t w, int32_t h, int32_t stride) {
  if (stride < 1) stride = 1;
  return (srtNumTiles(w, h) + stride - 1) / stride;
}

// Work items per pixel when the caller leaves the choice to the library (sppChunks == 0): about 8 samples per
// item, at least 128 items per pixel when there are that many samples (down to one sample per item), at most 640.
// It depends on the sample count alone, so that the chunk boundaries -- and with them the image, bit for bit --
// are the same for every tile split and GPU count.  Many items per pixel keep the tiles in flight few (a queue's
// waves pull consecutive items, i.e. the chunks of one tile, then of its neighbour), and SMALL items keep the end
// of a launch short: the last items to finish are single pixels of the mesh, ten times the average pixel's cost,
// and a rank's share of a frame feels that tail most.  720p headline at 5000 spp on the LDS-resident-tree kernel
// (profiles/r02/chunk_policy.txt): whole frame 1069.9 / 1067.9 / 1066.7 / 1068.1 ms with 157 / 314 / 628 / 1250
// chunks (as long as the chunk slots fit the scratch budget; the atomic path costs 1.2 %); one of 8 ranks' share
// 148.5 / 142.5 / 139.5 / 138.3 ms (133.7 would be an eighth of the frame).  Low sample counts: 64 spp on the 240p
// spheres frame run at 4.3 / 5.2 / 6.2 / 7.1 / 7.8 Gsamples/s with 4 / 8 / 16 / 32 / 64 chunks.
int32_t srtDefaultSppChunks(int32_t spp) {
  const int32_t bySize = (spp + 7) / 8, byCount = std::min(128, spp);
  return std::max(1, std::min(640, std::max(bySize, byCount)));
}

// The chunk count a render of this size will use: `sppChunks` when the caller gives one, else srtDefaultSppChunks(spp),
// and -1 when an explicit count does not fit.  Work items and chunk slots are indexed with 32-bit integers in the
// kernels: the slots of one chunk over the WHOLE image (not a rank's share: the plan, and with it the image bit for
// bit, must not depend on the tile split), plus the padding a work queue's last unit can add (a unit is at most 1024
// tiles; every queue counts its own items).  1280 x 720 allows 2166 chunks, 1920 x 1080 1003: the default plan
// (at most 640) always fits images below about 3 Mpixels; beyond that the default is clamped.
int32_t srtPlanSppChunks(int32_t imageWidth, int32_t imageHeight, int32_t spp, int32_t sppChunks) {
  if (imageWidth < 1 || imageHeight < 1 || spp < 1 || sppChunks < 0 || sppChunks > spp) return -1;
  const int64_t perChunk = ((int64_t)srtNumTiles(imageWidth, imageHeight) + 1024 + 64) * SRT_TILE_PIXELS;
  const int64_t maxChunks = (int64_t)0x7fffffff / perChunk;
  if (maxChunks < 1) return -1;
  if (sppChunks > 0) return sppChunks <= maxChunks ? sppChunks : -1;
  return (int32_t)std::min<int64_t>(srtDefaultSppChunks(spp), maxChunks);
}

static int checkParams(SrtContext* ctx, const SrtRenderParams* p) {
  if (!ctx->haveScene) return fail(ctx, "render: no scene uploaded");
  if (!ctx->haveCamera) return fail(ctx, "render: no camera set");
  if (p->imageWidth < 2 || p->imageHeight < 2) return fail(ctx, "render: image must be at least 2x2 (u,v divide by W-1,H-1)");
  if (p->imageWidth > 65535 * SRT_TILE_W || p->imageHeight > 65535 * SRT_TILE_H) return fail(ctx, "render: image larger than 65535 tiles a side");
  if (p->spp < 1) return fail(ctx, "render: spp must be >= 1");
  if (p->sampleFirst < 0 || (int64_t)p->sampleFirst + p->spp > 0x7fffffff) return fail(ctx, "render: bad sample range");
  if (p->maxBounce < 0 || p->maxBounce > SRT_MAX_BOUNCE) return fail(ctx, "render: maxBounce must be in [0,%d]", SRT_MAX_BOUNCE);
  if (p->tileStride < 1 || p->tileFirst < 0 || p->tileFirst >= p->tileStride) return fail(ctx, "render: bad tile split %d/%d", p->tileFirst, p->tileStride);
  if (p->sppChunks < 0 || p->sppChunks > p->spp) return fail(ctx, "render: sppChunks must be in [0, spp] (0 = library default)");
  return 0;
}

static int srtRenderTilesImpl(SrtContext* ctx, const SrtRenderParams* p, void* dAccumTiles, void* streamPtr) {
  if (!ctx || !p || !dAccumTiles) return 1;
  if (checkParams(ctx, p)) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = static_cast<hipStream_t>(streamPtr);
  RenderArgs a;
  memset(&a, 0, sizeof a);
  a.scene = ctx->scene;
  a.cam = ctx->cam;
  a.imageWidth = p->imageWidth;
  a.imageHeight = p->imageHeight;
  a.tilesX = (p->imageWidth + SRT_TILE_W - 1) / SRT_TILE_W;
  a.tilesY = (p->imageHeight + SRT_TILE_H - 1) / SRT_TILE_H;
  a.tileBlock = std::max(1, ctx->tun.tileBlock);
  a.numTiles = srtNumTiles(p->imageWidth, p->imageHeight);
  a.spp = p->spp;
  a.maxBounce = p->maxBounce;
  a.sampleFirst = p->sampleFirst;
  a.seed = p->seed;
  memcpy(a.background, p->background, 12);
  a.tMin = p->tMin;
  a.tileFirst = p->tileFirst;
  a.tileStride = p->tileStride;
  a.numLocalTiles = srtNumLocalTiles(p->imageWidth, p->imageHeight, p->tileStride);
  a.sppChunks = p->sppChunks > 0 ? p->sppChunks : srtDefaultSppChunks(p->spp);
  {
    // work queues (srt_render_kernel): units of >= 8 consecutive local tiles, about a dozen units per queue,
    // at most 64 queues.  Measured on the 720p headline frame (ms per launch, 1 rank / one of 8 ranks):
    // 1 queue 1916 / 253, 16 queues x 8 tiles 1818 / 238, 64 x 8: 1767 / 255, 64 x 16: 1744 / -.
    const auto pow2Floor = [](int v) { int r = 1; while (2 * r <= v) r *= 2; return r; };
    int unit = 8;
    const int unitsAt8 = (a.numLocalTiles + 7) / 8;
    if (unitsAt8 >= 2 * 12 * SRT_MAX_QUEUES) unit = 8 * pow2Floor(unitsAt8 / (12 * SRT_MAX_QUEUES));
    a.unitTiles = std::min(1024, std::max(1, ctx->tun.unitTiles > 0 ? ctx->tun.unitTiles : unit));
    const int units = (a.numLocalTiles + a.unitTiles - 1) / a.unitTiles;
    // ... and only while every wave still gets a few dozen groups: with few groups per wave (16 spp on a
    // 10 M-triangle soup: 14 400 groups for 5 120 waves) one counter balances better than stealing does.
    const int64_t groups = (int64_t)a.numLocalTiles * a.sppChunks, waves = (int64_t)ctx->prop.multiProcessorCount * 20;
    const int byUnits = pow2Floor(std::max(1, units / 12));
    const int byGroups = pow2Floor((int)std::max<int64_t>(1, std::min<int64_t>(SRT_MAX_QUEUES, groups / (2 * waves))));
    a.numQueues = std::min(SRT_MAX_QUEUES, std::max(1, ctx->tun.queues > 0 ? ctx->tun.queues : std::min(byUnits, byGroups)));
  }
  {
    const int32_t planned = srtPlanSppChunks(p->imageWidth, p->imageHeight, p->spp, p->sppChunks);
    if (planned < 1) return fail(ctx, "render: sppChunks %d x %d tiles exceeds 2^31 work items", p->sppChunks, a.numTiles);
    a.sppChunks = planned;
  }
  {
    // exact chunk sums cannot wrap: partial sums of 2^26 / (chunk count rounded up to a power of two) or more count as infinite
    int pow2 = 1;
    while (pow2 < a.sppChunks) pow2 *= 2;
    a.fixLimit = 0x1p26f / (float)pow2;
  }
  a.numWork = a.numLocalTiles * a.sppChunks * SRT_TILE_PIXELS;
  a.sppBase = a.spp / a.sppChunks;
  a.sppRem = a.spp % a.sppChunks;
  a.numUnits = (a.numLocalTiles + a.unitTiles - 1) / a.unitTiles;
  a.unitGroups = a.unitTiles * a.sppChunks;
  a.rcpUnitGroups = 1.0f / (float)a.unitGroups;
  a.rcpChunks = 1.0f / (float)a.sppChunks;
  if (ctx->tileTableKey[0] != p->imageWidth || ctx->tileTableKey[1] != p->imageHeight || ctx->tileTableKey[2] != a.tileBlock || !ctx->tileTable.p) {
    // the tile order as a table (once per image size): the kernel's restart step looks a tile up instead of dividing
    std::vector<uint32_t> table((size_t)a.numTiles);
    for (int32_t i = 0; i < a.numTiles; ++i) {
      int tx, ty;
      srtTileFromOrder(i, a.tilesX, a.tilesY, a.tileBlock, tx, ty);
      table[i] = (uint32_t)tx | (uint32_t)ty << 16;
    }
    if (ctx->tileTable.p) HIP_OK(ctx, hipFree(ctx->tileTable.p));
    ctx->tileTable = DeviceBuffer();
    HIP_OK(ctx, hipMalloc(&ctx->tileTable.p, std::max<size_t>(table.size() * 4, 16)));
    ctx->tileTable.bytes = table.size() * 4;
    HIP_OK(ctx, hipMemcpy(ctx->tileTable.p, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    ctx->tileTableKey[0] = p->imageWidth;
    ctx->tileTableKey[1] = p->imageHeight;
    ctx->tileTableKey[2] = a.tileBlock;
  }
  a.tileXY = static_cast<const uint32_t*>(ctx->tileTable.p);
  // Scheduler defaults by traversal mode (profiles/r02/scheduler_sweep.txt).  FAITHFUL on cache-resident scenes:
  // node bursts go on while half of their lanes are still at nodes, up to 64 visits, restarts at 24 waiting lanes
  // (+6 % on the headline frame against 6/8, 32, 16).  The closest-hit traversal over the 64-byte records is bound by
  // memory latency on large scenes and wants shorter bursts that give up sooner (10 M triangles: 87.6 against 77.8
  // Msamples/s), and does not care on small ones.
  const bool closestMode = p->traversal == SRT_TRAVERSE_CLOSEST;
  a.shadeMin = ctx->tun.shadeMin >= 0 ? ctx->tun.shadeMin : (closestMode ? 16 : 24);
  a.primMin = ctx->tun.primMin;
  a.hitMin = ctx->tun.hitMin;
  a.fuseMin = ctx->tun.fuseMin;
  a.nodeBurst = std::max(1, ctx->tun.nodeBurst > 0 ? ctx->tun.nodeBurst : (closestMode ? 32 : 64));
  a.primAgainMin = std::max(1, ctx->tun.primAgainMin);
  a.keepEighths = std::min(8, ctx->tun.keepEighths >= 0 ? ctx->tun.keepEighths : (closestMode ? 6 : 4));
  a.queue = ctx->dQueue;
  a.stats = (p->countStats || ctx->tun.wfProfile > 0) ? ctx->dStats : nullptr;
  a.aov = p->countStats ? ctx->dAov : nullptr;
  a.aovDepth = ctx->aovDepth;
  const size_t tilePixels = (size_t)a.numLocalTiles * SRT_TILE_PIXELS;
  a.out = static_cast<float4*>(dAccumTiles);
  a.fix = nullptr;
  a.chunkStride = 0;
  bool scratchPath = false;
  if (a.sppChunks > 1) {
    // Chunk sums are added exactly (srt_kernels.hip "Chunk sums").  Scratch path (a float4 slot per item, summed by
    // srt_sum_chunks_kernel) while this rank's slots fit the budget, else the atomic path (32 B per pixel, 0.4-1 %
    // slower); the two give the same bits, so the choice may differ from rank to rank.
    const size_t localSlots = tilePixels * a.sppChunks * sizeof(float4);
    // budget: the tunable, and never more than a quarter of what the device has free right now (a smaller, shared or
    // partitioned GPU takes the atomic path -- same bits -- instead of failing)
    size_t budget = (size_t)std::max(0, ctx->tun.chunkScratchMb) * 1024 * 1024, freeB = 0, totalB = 0;
    if (ctx->chunkScratch.bytes < localSlots && hipMemGetInfo(&freeB, &totalB) == hipSuccess) budget = std::min(budget, (freeB + ctx->chunkScratch.bytes) / 4);
    scratchPath = localSlots <= budget;
    size_t need = scratchPath ? localSlots : tilePixels * sizeof(SrtFixedAccum);
    if (ctx->chunkScratch.bytes < need) {
      if (ctx->chunkScratch.p) HIP_OK(ctx, hipFree(ctx->chunkScratch.p));
      ctx->chunkScratch = DeviceBuffer();
      if (hipMalloc(&ctx->chunkScratch.p, need) != hipSuccess) {
        (void)hipGetLastError();
        ctx->chunkScratch.p = nullptr;
        if (!scratchPath) return fail(ctx, "render: cannot allocate %zu B for the pixel sums", need);
        scratchPath = false;  // the slots do not fit after all: 32 B per pixel on the atomic path
        need = tilePixels * sizeof(SrtFixedAccum);
        HIP_OK(ctx, hipMalloc(&ctx->chunkScratch.p, need));
      }
      ctx->chunkScratch.bytes = need;
    }
    if (scratchPath) {
      a.out = static_cast<float4*>(ctx->chunkScratch.p);
      a.chunkStride = (int32_t)tilePixels;
    } else {
      a.fix = static_cast<SrtFixedAccum*>(ctx->chunkScratch.p);
      HIP_OK(ctx, hipMemsetAsync(a.fix, 0, tilePixels * sizeof(SrtFixedAccum), stream));
    }
  }
  // FAITHFUL on a scene whose whole node array fits into a CU's LDS: the LDS-resident-tree kernel (srt_render_kernel
  // LDSTREE), one workgroup of 1024 threads per CU, walking the threaded copy of the tree (no per-lane stack).
  const size_t ldsTreeBytes = (size_t)ctx->scene.numNodes * 32 + 16 * sizeof(int32_t);  // threaded tree: no stacks
  // (Even trees of a few dozen nodes gain: their frames are shading-bound, and the 128-register kernel keeps a hit's
  // texel loads in flight together where the 96-register one spills, profiles/r02/lds_tree.txt.)
  const bool ldsTree = p->traversal == SRT_TRAVERSE_FAITHFUL && ctx->tun.ldsTree > 0 && ctx->scene.numNodes >= ctx->tun.ldsTree && ldsTreeBytes <= 160 * 1024 &&
                       ctx->scene.nodeThread != nullptr;  // thread links exist: host-built trees, 15-bit references (srtUploadScene)
  // ... and when the attenuation stacks fit behind them as well they stay in LDS (ldsTreeMode 2): +2 to +5 % on the
  // small BASELINE scenes; the headline scene's tree leaves no room (mode 1: they live in global memory)
  const size_t attBytes = (size_t)(3 * p->maxBounce + 3) * 1024 * sizeof(float);
  const int ldsTreeMode = !ldsTree ? 0 : (ldsTreeBytes + attBytes <= 160 * 1024 ? 2 : 1);
  // closest-hit traversal over the four-box records: its stack holds up to three pending references per wide level; a
  // tree too deep for that beside the attenuation stacks (64 KB per workgroup at most: two workgroups per CU) walks the
  // two-box records instead
  const bool attGlobal256 = !ldsTree && p->traversal == SRT_TRAVERSE_CLOSEST && ctx->tun.attGlobal > 0 && ctx->scene.numNodes >= ctx->tun.attGlobal;
  if (p->traversal == SRT_TRAVERSE_CLOSEST && a.scene.nodes4) {
    const int need = std::max(a.scene.stackDepth, a.scene.wideStackDepth);
    if (ldsBytesFor(ctx, p->maxBounce, need, attGlobal256) <= 80 * 1024)
      a.scene.stackDepth = need;
    else
      a.scene.nodes4 = nullptr;
  } else {
    a.scene.nodes4 = nullptr;
  }
  const size_t lds = ldsTreeMode == 2 ? ldsTreeBytes + attBytes : ldsTree ? ldsTreeBytes : ldsBytesFor(ctx, p->maxBounce, a.scene.stackDepth, attGlobal256);
  if (lds > 160 * 1024) return fail(ctx, "render: BVH depth %d needs %zu B of LDS per workgroup", ctx->scene.stackDepth, lds);
  // The path-pool kernel (srt_wavefront.hip) serves what the LDS-resident tree serves, when its rings fit behind the
  // tree: one 1024-thread workgroup per CU, wfPool contexts each.  The counting variant stays with srt_render_kernel.
  // LDS behind the tree: 64 control words, six rings of 16-bit slots, and per context the (t, primitive) its walk ended at:
  // 18 bytes per context.  Ring capacity = pool size = the largest of 1024, 1536, 2048, 3072, 4096 that fits and does not
  // exceed the tunable (the headline scene's 129 KB tree leaves room for 1536).
  int wfPoolSize = 0, wfRingCap = 0, wfRingShift = 0, wfRingMul3 = 0;
  // Hybrid form: the tree's top in LDS, the rest read from global memory (scene.nodesWf, built at upload when the tree does
  // not fit or the tunable wf_resident_max asks for it).
  const bool hybrid = p->traversal == SRT_TRAVERSE_FAITHFUL && ctx->scene.nodesWf != nullptr && ctx->tun.wavefront > 0 && !p->countStats &&
                      ctx->scene.primClass != nullptr;
  const size_t wfFixed = (size_t)(hybrid ? ctx->scene.wfResident : ctx->scene.numNodes) * 32 + 64 * sizeof(int32_t);
  const size_t wfPerContext = hybrid ? 20 : 18;  // six ring slots of 16 bits, t, the primitive (16 bits; 32 in the hybrid form)
  {
    // ring counters are 32-bit and a 3 * 2^j ring cannot take their wrap-around: such rings only while a workgroup's
    // enqueues stay far below 2^32 (about three per sample)
    const double enqueuesPerGroup = 4.0 * (double)a.numLocalTiles * SRT_TILE_PIXELS * (double)p->spp / std::max(1, ctx->prop.multiProcessorCount);
    static const struct { int cap, shift, mul3; } kRings[] = {{4096, 12, 0}, {3072, 10, 1}, {2048, 11, 0}, {1536, 9, 1}, {1024, 10, 0}};
    for (const auto& r : kRings) {
      if (r.cap > std::max(1024, ctx->tun.wfPool) || wfFixed + wfPerContext * r.cap > 160 * 1024) continue;
      if (r.mul3 && enqueuesPerGroup > 2.0e9) continue;
      wfRingCap = r.cap;
      wfRingShift = r.shift;
      wfRingMul3 = r.mul3;
      break;
    }
    wfPoolSize = wfRingCap;
  }
  const size_t wfLds = wfFixed + wfPerContext * wfRingCap;
  const bool wavefront = wfRingCap > 0 && (hybrid || (ldsTree && ctx->tun.wavefront > 0 && ctx->scene.numNodes >= ctx->tun.wavefront && !p->countStats &&
                                                      ctx->scene.primClass != nullptr));
  if (!(wavefront && hybrid)) a.scene.nodesWf = nullptr;  // srt_launch_render_wf picks the form by this pointer
  int perCU = 0;
  if (wavefront || srt_render_occupancy(p->traversal, p->countStats, ldsTreeMode, lds, &perCU) != 0 || perCU < 1) perCU = 1;
  // persistent waves: enough workgroups to fill every CU, never more than there is work (4 or 16 waves each)
  const int wgWaves = ldsTree || wavefront ? 16 : 4;
  int grid = std::min(ctx->prop.multiProcessorCount * perCU, (a.numWork + SRT_TILE_PIXELS * wgWaves - 1) / (SRT_TILE_PIXELS * wgWaves));
  if (grid < 1) grid = 1;
  auto ensure = [&](DeviceBuffer& b, size_t need) -> int {
    if (b.bytes >= need) return 0;
    if (b.p) HIP_OK(ctx, hipFree(b.p));
    b = DeviceBuffer();
    HIP_OK(ctx, hipMalloc(&b.p, need));
    b.bytes = need;
    return 0;
  };
  if (wavefront) {
    // a workgroup never needs more contexts than it has work items
    const int64_t itemsPerGroup = ((int64_t)a.numWork + grid - 1) / grid;
    wfPoolSize = (int)std::max<int64_t>(64, std::min<int64_t>(wfPoolSize, itemsPerGroup + 63));
    if (ensure(ctx->wfPool, (size_t)grid * wfPoolSize * 128)) return 1;
    const int hiLevels = std::max(0, p->maxBounce - 4);
    if (ensure(ctx->wfAttHi, std::max<size_t>(16, (size_t)grid * 3 * hiLevels * wfPoolSize * sizeof(float)))) return 1;
    a.wfPool = static_cast<char*>(ctx->wfPool.p);
    a.wfAttHi = static_cast<float*>(ctx->wfAttHi.p);
    a.wfPoolSize = wfPoolSize;
    a.wfRingCap = wfRingCap;
    a.wfRingShift = wfRingShift;
    a.wfRingMul3 = wfRingMul3;
    a.wfSwapMin = ctx->tun.wfSwapMin > 0 ? std::min(64, ctx->tun.wfSwapMin) : (hybrid ? 16 : 32);
    a.wfFarRounds = ctx->tun.wfFarRounds > 0 ? std::min(4, ctx->tun.wfFarRounds) : (ctx->scene.numNodes <= (1 << 20) ? 2 : 1);
    a.wfSwapBig = std::max(a.wfSwapMin, std::min(64, ctx->tun.wfSwapBig));
    HIP_OK(ctx, hipHostGetDevicePointer((void**)&a.wfError, ctx->dWfError, 0));
  } else if (ldsTreeMode == 1 || attGlobal256) {
    if (ensure(ctx->attScratch, (size_t)(3 * p->maxBounce + 3) * grid * (ldsTree ? 1024 : 256) * sizeof(float))) return 1;
    a.attScratch = static_cast<float*>(ctx->attScratch.p);
  }
  HIP_OK(ctx, hipMemsetAsync(ctx->dQueue, 0, sizeof(int32_t) * 16 * a.numQueues, stream));
  if (a.stats) HIP_OK(ctx, hipMemsetAsync(ctx->dStats, 0, 96 * sizeof(unsigned long long), stream));
  HIP_OK(ctx, hipEventRecord(ctx->evStart, stream));
  ctx->lastLaunch[0] = wavefront ? (hybrid ? 4 : 3) : ldsTreeMode;
  ctx->lastLaunch[1] = grid;
  ctx->lastLaunch[2] = ldsTree || wavefront ? 1024 : 256;
  ctx->lastLaunch[3] = (int32_t)(wavefront ? wfLds : lds);
  int rc = wavefront ? srt_launch_render_wf(&a, ctx->tun.wfProfile > 0, grid, wfLds, stream) : srt_launch_render(&a, p->traversal, p->countStats, ldsTreeMode, grid, lds, stream);
  if (rc) return fail(ctx, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
  HIP_OK(ctx, hipEventRecord(ctx->evStop, stream));
  ctx->timed = true;
  if (a.fix) {
    rc = srt_launch_finalize(a.fix, static_cast<float4*>(dAccumTiles), (int)tilePixels, a.spp, stream);
    if (rc) return fail(ctx, "finalize launch failed: %s", hipGetErrorString((hipError_t)rc));
  } else if (scratchPath) {
    rc = srt_launch_sum_chunks(a.out, static_cast<float4*>(dAccumTiles), (int)tilePixels, a.sppChunks, a.fixLimit, stream);
    if (rc) return fail(ctx, "chunk sum launch failed: %s", hipGetErrorString((hipError_t)rc));
  }
  return 0;
}

int srtResolveTiles(SrtContext* ctx, const SrtRenderParams* p, const void* dGathered, void* dRgba, void* dAccumImage,
                    void* streamPtr) {
  if (!ctx || !p || !dGathered) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  ResolveArgs a;
  a.gathered = static_cast<const float4*>(dGathered);
  a.imageWidth = p->imageWidth;
  a.imageHeight = p->imageHeight;
  a.tilesX = (p->imageWidth + SRT_TILE_W - 1) / SRT_TILE_W;
  a.tileBlock = std::max(1, ctx->tun.tileBlock);
  a.tileStride = p->tileStride < 1 ? 1 : p->tileStride;
  a.numLocalTiles = srtNumLocalTiles(p->imageWidth, p->imageHeight, a.tileStride);
  a.spp = p->spp;
  a.rgba = static_cast<uint8_t*>(dRgba);
  a.accumImage = static_cast<float4*>(dAccumImage);
  int rc = srt_launch_resolve(&a, static_cast<hipStream_t>(streamPtr));
  if (rc) return fail(ctx, "resolve launch failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

int srtRenderImage(SrtContext* ctx, const SrtRenderParams* pIn, float* hAccum, uint8_t* hRgba) {
  if (!ctx || !pIn) return 1;
  SrtRenderParams p = *pIn;
  p.tileFirst = 0;
  p.tileStride = 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  const size_t nPix = (size_t)p.imageWidth * p.imageHeight;
  const size_t tileBytes = (size_t)srtNumTiles(p.imageWidth, p.imageHeight) * SRT_TILE_PIXELS * sizeof(float4);
  void *dTiles = nullptr, *dRgba = nullptr, *dAcc = nullptr;
  int rc = 1;
  do {
    if (hipMalloc(&dTiles, tileBytes) != hipSuccess) { fail(ctx, "hipMalloc tiles"); break; }
    if (hRgba && hipMalloc(&dRgba, nPix * 4) != hipSuccess) { fail(ctx, "hipMalloc rgba"); break; }
    if (hAccum && hipMalloc(&dAcc, nPix * sizeof(float4)) != hipSuccess) { fail(ctx, "hipMalloc accum"); break; }
    if (srtRenderTilesImpl(ctx, &p, dTiles, nullptr)) break;
    if (srtResolveTiles(ctx, &p, dTiles, dRgba, dAcc, nullptr)) break;
    if (hipDeviceSynchronize() != hipSuccess) { fail(ctx, "render kernel failed: %s", hipGetErrorString(hipGetLastError())); break; }
    if (wfCheck(ctx)) break;
    if (hRgba && hipMemcpy(hRgba, dRgba, nPix * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail(ctx, "copy rgba"); break; }
    if (hAccum && hipMemcpy(hAccum, dAcc, nPix * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) { fail(ctx, "copy accum"); break; }
    rc = 0;
  } while (0);
  if (dTiles) (void)hipFree(dTiles);
  if (dRgba) (void)hipFree(dRgba);
  if (dAcc) (void)hipFree(dAcc);
  return rc;
}

int srtTraceRays(SrtContext* ctx, const SrtRay* rays, int64_t n, SrtHit* hits, int32_t traversal) {
  if (!ctx || !rays || !hits || n < 0) return 1;
  if (!ctx->haveScene) return fail(ctx, "trace: no scene uploaded");
  if (n == 0) return 0;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  void *dRays = nullptr, *dHits = nullptr;
  int rc = 1;
  do {
    if (hipMalloc(&dRays, n * sizeof(SrtRay)) != hipSuccess || hipMalloc(&dHits, n * sizeof(SrtHit)) != hipSuccess) { fail(ctx, "trace: hipMalloc"); break; }
    if (hipMemcpy(dRays, rays, n * sizeof(SrtRay), hipMemcpyHostToDevice) != hipSuccess) { fail(ctx, "trace: copy in"); break; }
    TraceArgs a;
    a.scene = ctx->scene;
    a.rays = static_cast<const SrtRay*>(dRays);
    a.hits = static_cast<SrtHit*>(dHits);
    a.n = n;
    size_t lds = (size_t)std::max(ctx->scene.stackDepth, 1) * 256 * sizeof(int32_t);
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->prop.multiProcessorCount * 8);
    int e = srt_launch_trace(&a, traversal, grid, lds, nullptr);
    if (e) { fail(ctx, "trace launch failed: %s", hipGetErrorString((hipError_t)e)); break; }
    if (hipDeviceSynchronize() != hipSuccess) { fail(ctx, "trace kernel failed"); break; }
    if (hipMemcpy(hits, dHits, n * sizeof(SrtHit), hipMemcpyDeviceToHost) != hipSuccess) { fail(ctx, "trace: copy out"); break; }
    rc = 0;
  } while (0);
  if (dRays) (void)hipFree(dRays);
  if (dHits) (void)hipFree(dHits);
  return rc;
}

// test entry: material::scatter known answers through the kernel's own shade()
int srtScatterRays(SrtContext* ctx, const SrtRay* rays, const SrtHit* hits, int32_t n, uint64_t seed, float* out13);
int srtScatterTest(SrtContext* ctx, const SrtRay* rays, const SrtHit* hits, int32_t n, uint64_t seed, float* out13) {
  return srtScatterRays(ctx, rays, hits, n, seed, out13);  // the test hook's old name
}

int srtScatterRays(SrtContext* ctx, const SrtRay* rays, const SrtHit* hits, int32_t n, uint64_t seed, float* out13) {
  if (!ctx || !rays || !hits || !out13 || n < 1) return 1;
  if (!ctx->haveScene) return fail(ctx, "scatter: no scene uploaded");
  HIP_OK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < n; ++i)
    if (hits[i].material < 0) return fail(ctx, "scatter: hit %d has no material", i);
  void *dRays = nullptr, *dHits = nullptr, *dOut = nullptr;
  int rc = 1;
  do {
    if (hipMalloc(&dRays, n * sizeof(SrtRay)) != hipSuccess || hipMalloc(&dHits, n * sizeof(SrtHit)) != hipSuccess ||
        hipMalloc(&dOut, (size_t)n * 13 * 4) != hipSuccess) { fail(ctx, "scatter: hipMalloc"); break; }
    (void)hipMemcpy(dRays, rays, n * sizeof(SrtRay), hipMemcpyHostToDevice);
    (void)hipMemcpy(dHits, hits, n * sizeof(SrtHit), hipMemcpyHostToDevice);
    int e = srt_launch_scatter(&ctx->scene, (const SrtRay*)dRays, (const SrtHit*)dHits, (float*)dOut, seed, n, nullptr);
    if (e) { fail(ctx, "scatter launch failed"); break; }
    if (hipDeviceSynchronize() != hipSuccess) { fail(ctx, "scatter kernel failed"); break; }
    (void)hipMemcpy(out13, dOut, (size_t)n * 13 * 4, hipMemcpyDeviceToHost);
    rc = 0;
  } while (0);
  if (dRays) (void)hipFree(dRays);
  if (dHits) (void)hipFree(dHits);
  if (dOut) (void)hipFree(dOut);
  return rc;
}

int srtLastKernelMs(SrtContext* ctx, float* ms) {
  if (!ctx || !ms) return 1;
  if (!ctx->timed) return fail(ctx, "no render has been launched");
  HIP_OK(ctx, hipEventSynchronize(ctx->evStop));
  HIP_OK(ctx, hipEventElapsedTime(ms, ctx->evStart, ctx->evStop));
  return wfCheck(ctx);
}

int srtGetStats(SrtContext* ctx, SrtStats* out) {
  if (!ctx || !out) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  HIP_OK(ctx, hipDeviceSynchronize());
  unsigned long long v[18];
  HIP_OK(ctx, hipMemcpy(v, ctx->dStats, sizeof v, hipMemcpyDeviceToHost));
  out->samples = v[0]; out->rays = v[1]; out->nodeVisits = v[2]; out->boxPasses = v[3];
  out->triTests = v[4]; out->sphereTests = v[5]; out->shadedTriHits = v[6]; out->texelFetches = v[7];
  out->cyclesNode = v[8]; out->cyclesPrim = v[9]; out->cyclesShade = v[10]; out->cyclesTotal = v[11];
  out->stepsNode = v[12]; out->stepsPrim = v[13]; out->stepsShade = v[14];
  out->lanesNode = v[15]; out->lanesPrim = v[16]; out->lanesShade = v[17];
  return 0;
}

int srtDeviceInfo(SrtContext* ctx, char* name, int32_t nameCap, int32_t* numCUs, int32_t* clockMHz) {
  if (!ctx) return 1;
  if (name && nameCap > 0) {
    snprintf(name, nameCap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (numCUs) *numCUs = ctx->prop.multiProcessorCount;
  if (clockMHz) *clockMHz = ctx->prop.clockRate / 1000;
  return 0;
}


// No exception crosses the C boundary (std::vector / std::string allocations above may throw).
#define SRT_GUARDED(ctx, call)                                                  \
  try {                                                                         \
    return (call);                                                              \
  } catch (const std::exception& e) {                                           \
    return fail(ctx, "%s: %s", __func__, e.what());                             \
  } catch (...) {                                                               \
    return fail(ctx, "%s: unknown exception", __func__);                        \
  }
int srtUploadScene(SrtContext* ctx, const SrtSceneDesc* d) { SRT_GUARDED(ctx, srtUploadSceneImpl(ctx, d)); }
int srtBuildBvh(const SrtSceneDesc* d, int32_t item, SrtBvhNode* out, int32_t capacity, int32_t* count, int32_t* stackDepth) { SRT_GUARDED(nullptr, srtBuildBvhImpl(d, item, out, capacity, count, stackDepth)); }
int srtGetBvh(SrtContext* ctx, int32_t item, SrtBvhNode* nodes, int32_t capacity, int32_t* count) { SRT_GUARDED(ctx, srtGetBvhImpl(ctx, item, nodes, capacity, count)); }
int srtRenderTiles(SrtContext* ctx, const SrtRenderParams* p, void* dAccumTiles, void* streamPtr) { SRT_GUARDED(ctx, srtRenderTilesImpl(ctx, p, dAccumTiles, streamPtr)); }

/* include/srt_hip_test.h: the render kernel's own traversal, ray by ray */
static int srtRenderAovImpl(SrtContext* ctx, const SrtRenderParams* pIn, int32_t depth, SrtAovRecord* hOut) {
  if (!ctx || !pIn || !hOut || depth < 0) return 1;
  SrtRenderParams p = *pIn;
  p.spp = 1;
  p.sppChunks = 1;
  p.countStats = 1;
  p.tileFirst = 0;
  p.tileStride = 1;
  if (checkParams(ctx, &p)) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  const size_t nPix = (size_t)p.imageWidth * p.imageHeight;
  const size_t tileBytes = (size_t)srtNumTiles(p.imageWidth, p.imageHeight) * SRT_TILE_PIXELS * sizeof(float4);
  void* dTiles = nullptr;
  int rc = 1;
  do {
    if (hipMalloc(&dTiles, tileBytes) != hipSuccess || hipMalloc((void**)&ctx->dAov, nPix * sizeof(SrtAovRecord)) != hipSuccess) { fail(ctx, "aov: hipMalloc"); break; }
    if (hipMemset(ctx->dAov, 0, nPix * sizeof(SrtAovRecord)) != hipSuccess) { fail(ctx, "aov: memset"); break; }
    ctx->aovDepth = depth;
    if (srtRenderTilesImpl(ctx, &p, dTiles, nullptr)) break;
    if (hipDeviceSynchronize() != hipSuccess) { fail(ctx, "aov: render kernel failed: %s", hipGetErrorString(hipGetLastError())); break; }
    if (hipMemcpy(hOut, ctx->dAov, nPix * sizeof(SrtAovRecord), hipMemcpyDeviceToHost) != hipSuccess) { fail(ctx, "aov: copy out"); break; }
    rc = 0;
  } while (0);
  if (dTiles) (void)hipFree(dTiles);
  if (ctx->dAov) (void)hipFree(ctx->dAov);
  ctx->dAov = nullptr;
  return rc;
}
int srtRenderAov(SrtContext* ctx, const SrtRenderParams* p, int32_t depth, SrtAovRecord* hOut) { SRT_GUARDED(ctx, srtRenderAovImpl(ctx, p, depth, hOut)); }

/* include/srt_hip_test.h: sub-step profile of the counting variant's last launch */
int srtGetLaunchInfo(SrtContext* ctx, int32_t* out4) {
  if (!ctx || !out4) return 1;
  memcpy(out4, ctx->lastLaunch, sizeof ctx->lastLaunch);
  return 0;
}
int srtGetWfProfile(SrtContext* ctx, uint64_t* out46) {
  if (!ctx || !out46) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  HIP_OK(ctx, hipDeviceSynchronize());
  HIP_OK(ctx, hipMemcpy(out46, ctx->dStats + 32, 46 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}

// Host-only test hooks for srt_thread.h (no device needed): nodes8 = numNodes x 8 floats, the flattened records.
int srtTestThreadLinks16(const float* nodes8, int32_t numNodes, const int32_t* world, int32_t numWorld, int32_t numTriangles, int32_t numSpheres,
                         int32_t* outLinks) {
  if (!nodes8 || !world || !outLinks || numNodes < 0 || numWorld < 0) return -1;
  std::vector<float4> nodes(2 * (size_t)numNodes);
  if (numNodes) memcpy(nodes.data(), nodes8, nodes.size() * sizeof(float4));
  std::vector<int32_t> links;
  srtThreadLinks16(nodes, std::vector<int32_t>(world, world + numWorld), numTriangles, numSpheres, links);
  if (links.empty()) return 0;
  memcpy(outLinks, links.data(), links.size() * sizeof(int32_t));
  return 1;
}

int srtTestHybridRecords(const float* nodes8, int32_t numNodes, const int32_t* world, int32_t numWorld, int32_t numTriangles, int32_t numSpheres,
                         int32_t cap, float* outNodes8, int32_t* outWorld, int32_t* outSecond) {
  if (!nodes8 || !world || !outNodes8 || !outWorld || !outSecond || numNodes < 0 || numWorld < 0 || cap < 0) return -1;
  std::vector<float4> nodes(2 * (size_t)numNodes), wf;
  if (numNodes) memcpy(nodes.data(), nodes8, nodes.size() * sizeof(float4));
  std::vector<int32_t> w, second;
  const int32_t resident = srtHybridRecords(nodes, std::vector<int32_t>(world, world + numWorld), numTriangles, numSpheres, (size_t)cap, wf, w, second);
  if (resident <= 0) return 0;
  memcpy(outNodes8, wf.data(), wf.size() * sizeof(float4));
  memcpy(outWorld, w.data(), w.size() * sizeof(int32_t));
  memcpy(outSecond, second.data(), second.size() * sizeof(int32_t));
  return resident;
}

int srtGetShadeProfile(SrtContext* ctx, uint64_t* out10) {
  if (!ctx || !out10) return 1;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  HIP_OK(ctx, hipDeviceSynchronize());
  HIP_OK(ctx, hipMemcpy(out10, ctx->dStats + 18, 10 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}

/* include/srt_hip_test.h: per-context diagnostic tunables */
int srtSetTunable(SrtContext* ctx, const char* name, int32_t value) {
  if (!ctx || !name) return 1;
  for (const TunableName& t : kTunables)
    if (!strcmp(t.name, name)) {
      ctx->tun.*(t.field) = value;
      return 0;
    }
  return fail(ctx, "srtSetTunable: unknown tunable '%s'", name);
}
int srtGetTunable(SrtContext* ctx, const char* name, int32_t* value) {
  if (!ctx || !name || !value) return 1;
  for (const TunableName& t : kTunables)
    if (!strcmp(t.name, name)) {
      *value = ctx->tun.*(t.field);
      return 0;
    }
  return fail(ctx, "srtGetTunable: unknown tunable '%s'", name);
}

}  // extern "C"
