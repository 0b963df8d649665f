// srt_wavefront.hip -- the per-pixel path-tracing loop (main.cpp:200-227, rayColor main.cpp:33-52) with the paths
// kept in a POOL per workgroup instead of one per lane: lanes traverse, full waves shade.
//
// srt_kernels.hip's LDS-resident-tree kernel binds a path to a lane for its whole life, so every wave holds lanes at
// nodes, at primitives, waiting for a hit to be shaded and waiting for a restart, and each step kind runs at 29-39 of
// 64 lanes (profiles/r02/step_profile_lds_tree.txt); worse, a hit step serves whatever materials its ~38 lanes
// happen to have hit, so it executes the triangle record AND the sphere record, image lookups AND the checker, pbr AND
// metal AND light code on every execution.  Here, with one 1024-thread workgroup owning the CU and its LDS:
//
//   * a PATH CONTEXT (one work item in flight: pixel x sample chunk, its current ray, RNG, attenuations, partial
//     sum: one 128-byte line in global memory) belongs to no lane.  Each workgroup owns `poolSize` of them
//     (about 1.5 per lane).
//   * lanes are TRAVERSAL ENGINES: a lane holds only a ray and its walk of the threaded tree (DevScene::nodeThread,
//     bvh.h:97-105 in bvh.h's order).  When the walk ends it leaves (t, primitive) in the context's LDS slot, hands the
//     context to a queue and takes the next READY context (swap step: LDS traffic only, plus the new ray's one line).
//   * queues are rings of context ids in LDS: READY (a fresh ray to traverse), RESTART (path ended or ray missed:
//     main.cpp:39-40,49-51,217 and the next camera ray, main.cpp:204-216), NEWITEM (the item's last sample is in:
//     write its sum, pull the next work item with ONE atomic for the 64, main.cpp:200-203) and one HIT ring per
//     MATERIAL CLASS (triangle + pbr / sphere + pbr / everything else).  Any wave that finds 64 entries in a ring takes
//     them and runs that step for them: hit shading (main.cpp:42-51 + material.h), restarts and item pulls run at 64 of
//     64 lanes, and a hit step executes one class's code.  A wave with nothing to traverse serves partial batches,
//     which also drains the end of the frame.
//
// Same arithmetic as srt_kernels.hip (srt_path.h holds it), same samples in the same order per work item, the same
// exact chunk sums: accumulators are bit-identical to the step-scheduler kernels' (tests: node_path "wavefront").
// Termination: every spin on a ring slot is bounded; a wave that exceeds the bound raises the workgroup's abort word,
// every wave leaves at its next scheduling decision and the host reports the error (RenderArgs::wfError).  The
// grid drains when every context has found the work queues empty (live == 0).
#include "srt_path.h"

#define WF_BLOCK 1024
#define WF_CLASSES 3
#define WF_RING_READY 0
#define WF_RING_RESTART 1
#define WF_RING_HIT 2                   // + material class
#define WF_RING_NEWITEM (2 + WF_CLASSES)
#define WF_RINGS (3 + WF_CLASSES)
#define WF_SPIN_LIMIT (1 << 24)
#define WF_CTX_BYTES 128
// Context line (global memory, one per context id):  +0 A = ray origin, time   +16 B = ray direction (terminal radiance
// once the path has ended), -   +32 C = RNG state (2 words), -, depth | pend << 8   +48 D = partial pixel sum, sample
// count of the item   +64 E = output index, next sample, end sample, px | py << 16   +80 attenuations of bounces 0..3
// Control words (LDS, behind the tree): 0..15 the waves' current work queue; 16 + 2r ring r's tail (places reserved),
// 17 + 2r its head (places claimed) -- one 8-byte read gets both --, 28 live contexts, 29 abort.  Behind them the rings
// (16-bit slots), then per context the t (float) and the primitive (16 bits) its last walk ended at.
#define WF_CTL_TAIL(r) (16 + 2 * (r))
#define WF_CTL_HEAD(r) (17 + 2 * (r))
#define WF_CTL_LIVE 28
#define WF_CTL_ABORT 29
#define WF_CTL_WORDS 64
#define WF_PEND_MISS 0      // context meta word: depth | pend << 8
#define WF_PEND_TERMINAL 2
#define WF_WG __HIP_MEMORY_SCOPE_WORKGROUP

namespace {
enum { W_NODE = 0, W_PRIM = 1, W_SWAP = 2, W_SERVE = 3 };

__device__ __forceinline__ void bufStore4(Rsrc r, int off, float4 v) {
  u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, 0);
}
__device__ __forceinline__ void bufStore1(Rsrc r, int off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, 0); }
__device__ __forceinline__ uint32_t bufLoad1(Rsrc r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
}  // namespace

// SINGLE: the world list is one tree (as in srt_render_kernel).  PROFILE (tunable wf_profile, tools/wf_profile.py): per
// step kind, the clocks the waves spent in it, its executions and the lanes they served -> RenderArgs::stats[32 + ...]
// (kinds: 0 node visit, 1 primitive test, 2 swap, 3-5 hit step per class, 6 restart, 7 idle, 8 lost claim, 9 item pull,
// 10 visit of a node outside LDS -- hybrid form: wait for the records + visit; its clocks are part of kind 0's as well).
#define WF_PROF_KINDS 12
// HYBRID: the tree does not fit into LDS.  Its top -- the first `resident` records of DevScene::nodesWf, the boxes a ray is
// most likely to meet -- is kept there, the other records are read from global memory where the walk reaches them; a
// node round asks for the global lanes' records first and visits LDS nodes with the other lanes while they are on the way.
template <bool SINGLE, bool PROFILE, bool HYBRID>
__global__ __launch_bounds__(WF_BLOCK, 4) void srt_render_wf_kernel(const RenderArgs a) {

  // whole tree in LDS: 16-bit references, DONE = the 16-bit "no reference" sign-extended, a link = two of them.
  // HYBRID: 32-bit references, DONE = -2^29, a link = successor << 2 | what follows a leaf's first object (srt_device.h)
  constexpr int32_t DONE = HYBRID ? -(1 << 29) : (int32_t)0xFFFF8000;
  constexpr int32_t DONE_PAIR = HYBRID ? (int32_t)0x80000000 : (int32_t)0x80008000;  // "nothing follows, then done"
  constexpr int LINK_SHIFT = HYBRID ? 2 : 16;
  typedef typename std::conditional<HYBRID, int32_t, uint16_t>::type PrimSlot;
  extern __shared__ int32_t lds[];
  const DevScene& sc = a.scene;
  char* const ldsTree = reinterpret_cast<char*>(lds);
  const int resident = HYBRID ? sc.wfResident : sc.numNodes;  // nodes [0, resident) are in LDS
  const int treeBytes = resident * 32;
  int32_t* const ctl = reinterpret_cast<int32_t*>(ldsTree + treeBytes);
  const int RCAP = a.wfRingCap, POOL = a.wfPoolSize;
  uint16_t* const ringSlots = reinterpret_cast<uint16_t*>(ctl + WF_CTL_WORDS);
  float* const hitT = reinterpret_cast<float*>(ringSlots + WF_RINGS * RCAP);
  PrimSlot* const hitPrim = reinterpret_cast<PrimSlot*>(hitT + POOL);
  const int lane = threadIdx.x & 63;
  const unsigned long long laneBelow = (1ull << lane) - 1ull;
  const uint64_t seedMixed = mix64(a.seed);
  const V3 background = ld3(a.background);
  const Rsrc rsNodes = makeRsrc(HYBRID ? sc.nodesWf : sc.nodes, sc.numNodes * 32);
  const Rsrc rsTris = makeRsrc(sc.triTest, sc.numTris * 48);
  const Rsrc rsSpheres = makeRsrc(sc.spheres, sc.numSpheres * 48);
  const Rsrc rsTexels = makeRsrc(sc.texels, sc.texelBytes);
  const Rsrc rsClass = makeRsrc(sc.primClass, sc.numPrimClass);
  const Rsrc rsSecond = makeRsrc(sc.primSecond, HYBRID ? sc.numPrimClass * 4 : 0);
  // this workgroup's contexts: id << 7 is a context's byte offset; attenuation levels of bounces >= 4 live in
  // [level - 4][channel][id] in a second array
  const Rsrc rsPool = makeRsrc(a.wfPool + (size_t)blockIdx.x * POOL * WF_CTX_BYTES, POOL * WF_CTX_BYTES);
  const int hiLevels = a.maxBounce > 4 ? a.maxBounce - 4 : 0;
  const Rsrc rsAttHi = makeRsrc(a.wfAttHi + (size_t)blockIdx.x * 3 * hiLevels * POOL, 3 * hiLevels * POOL * 4);
  const bool singleRoot = SINGLE || sc.numWorld == 1;
  auto worldRef = [&](int k) {
    if (HYBRID) return sc.worldWf[k];
    const int r = sc.world[k];
    return r >= 0 ? r >> 5 : r;
  };
  // place of counter value c in a ring: c mod RCAP, RCAP = 2^j or 3 * 2^j (the pool size rounded up to such a number)
  const int ringShift = a.wfRingShift;
  const bool ringMul3 = a.wfRingMul3 != 0;
  auto ringPos = [&](uint32_t c) -> int {
    if (!ringMul3) return (int)(c & (uint32_t)(RCAP - 1));
    const uint32_t x = c >> ringShift;
    const uint32_t x3 = x - 3u * (uint32_t)(((unsigned long long)x * 0xAAAAAAABull) >> 33);
    return (int)((c & ((1u << ringShift) - 1u)) | (x3 << ringShift));
  };

  // ---- set-up: the threaded tree into LDS (as srt_render_kernel LDSTREE), empty rings, every context waits for an item
  {
    float4* dst = reinterpret_cast<float4*>(ldsTree);
    for (int i = threadIdx.x; i < resident * 2; i += WF_BLOCK) {
      float4 v = bufLoad4(rsNodes, 16 * i);
      if (!HYBRID) {  // nodesWf holds the threaded records already
        const int r = __float_as_int(v.w);
        if (i & 1)
          v.w = __int_as_float(sc.nodeThread[i >> 1]);
        else if (r >= 0)
          v.w = __int_as_float(r >> 5);
      }
      dst[i] = v;
    }
    for (int i = threadIdx.x; i < WF_CTL_WORDS; i += WF_BLOCK) ctl[i] = 0;
    for (int i = threadIdx.x; i < WF_RINGS * RCAP; i += WF_BLOCK) ringSlots[i] = 0;
    __syncthreads();
    for (int id = threadIdx.x; id < POOL; id += WF_BLOCK) ringSlots[WF_RING_NEWITEM * RCAP + id] = (uint16_t)(id + 1);
    if (threadIdx.x == 0) {
      ctl[WF_CTL_TAIL(WF_RING_NEWITEM)] = POOL;
      ctl[WF_CTL_LIVE] = POOL;
    }
    if (threadIdx.x < 16) ctl[threadIdx.x] = (int)(blockIdx.x % (unsigned)a.numQueues);  // the waves' home work queue
    __syncthreads();
  }
  int32_t* const waveQueue = ctl + (threadIdx.x >> 6);
  const int qHome = (int)(blockIdx.x % (unsigned)a.numQueues);
  const int unitItems = a.unitTiles * a.sppChunks * SRT_TILE_PIXELS;
  auto queueEnd = [&](int q) { return (q < a.numUnits ? (a.numUnits - q + a.numQueues - 1) / a.numQueues : 0) * unitItems; };

  auto raiseAbort = [&]() {
    if (__hip_atomic_exchange(&ctl[WF_CTL_ABORT], 1, __ATOMIC_RELAXED, WF_WG) == 0 && a.wfError) atomicAdd(a.wfError, 1);
  };
  // ---- rings.  A slot holds id + 1, 0 = empty.  Producers reserve places with one atomic per wave and fill them;
  // consumers claim places (compare-and-swap on the head, never beyond the tail) and take the ids, waiting the few
  // cycles a reserved place may still be unwritten.  Slots are re-used only after their consumer emptied them.
  // What a producer wrote for the context BEFORE (global stores: the caller fences; LDS: one wave's LDS operations
  // execute in order) is visible to the consumer that finds the id.
  // `ring` < 0: this lane enqueues nothing.  All rings of one call are served together: one atomic per ring, issued side
  // by side by the first lanes of the wave, then one look at the slots: two LDS round trips whatever the number of rings.
  auto enqueue = [&](int ring, int id) {
    int n = 0;  // lane r < WF_RINGS: how many lanes of this wave enqueue to ring r
#pragma unroll
    for (int r = 0; r < WF_RINGS; ++r) {
      const int c = __popcll(__ballot(ring == r));
      n = lane == r ? c : n;
    }
    int base = 0;
    if (lane < WF_RINGS && n > 0) base = __hip_atomic_fetch_add(&ctl[WF_CTL_TAIL(lane)], n, __ATOMIC_RELAXED, WF_WG);
    int myBase = 0;
    unsigned long long mine = 0;
#pragma unroll
    for (int r = 0; r < WF_RINGS; ++r) {
      const int b = __builtin_amdgcn_readlane(base, r);
      const unsigned long long m = __ballot(ring == r);
      if (ring == r) {
        myBase = b;
        mine = m;
      }
    }
    if (ring >= 0) {
      uint16_t* const s = ringSlots + ring * RCAP + ringPos((uint32_t)(myBase + __popcll(mine & laneBelow)));
      // (first look outside the loop: a loop header makes the compiler wait for every load the caller has in flight)
      if (__hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG) != 0) {
        int spins = 0;
        while (__hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG) != 0) {
          if (++spins > WF_SPIN_LIMIT) {
            raiseAbort();
            break;
          }
        }
      }
      __hip_atomic_store(s, (uint16_t)(id + 1), __ATOMIC_RELAXED, WF_WG);
    }
  };
  // claims up to `want` entries (none unless at least `atLeast` are there) for the lanes of `takers` in lane order;
  // returns how many, and the id for the lanes that got one (-1 otherwise).  Wave-uniform call.
  auto claim = [&](int r, unsigned long long takers, int want, int atLeast, int& id) -> int {
    int h = 0, k = 0;
    if (lane == 0) {
      const unsigned long long th = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&ctl[WF_CTL_TAIL(r)]), __ATOMIC_RELAXED, WF_WG);
      int t = (int)th;
      h = (int)(th >> 32);
      // first attempt outside the loop (as in enqueue: no loop header on the common path)
      bool again = false;
      {
        const int avail = (int)((uint32_t)t - (uint32_t)h);
        k = avail < want ? avail : want;
        if (k < atLeast || k <= 0) {
          k = 0;
        } else {
          int expected = h;
          if (!__hip_atomic_compare_exchange_strong(&ctl[WF_CTL_HEAD(r)], &expected, h + k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, WF_WG)) {
            h = expected;  // another wave claimed meanwhile: the tail can only have grown
            again = true;
          }
        }
      }
      while (again) {
        t = __hip_atomic_load(&ctl[WF_CTL_TAIL(r)], __ATOMIC_RELAXED, WF_WG);
        const int avail = (int)((uint32_t)t - (uint32_t)h);
        k = avail < want ? avail : want;
        if (k < atLeast || k <= 0) {
          k = 0;
          break;
        }
        int expected = h;
        if (__hip_atomic_compare_exchange_strong(&ctl[WF_CTL_HEAD(r)], &expected, h + k, __ATOMIC_RELAXED, __ATOMIC_RELAXED, WF_WG)) break;
        h = expected;
      }
    }
    h = __builtin_amdgcn_readfirstlane(h);
    k = __builtin_amdgcn_readfirstlane(k);
    id = -1;
    const int rank = __popcll(takers & laneBelow);
    if (((takers >> lane) & 1ull) && rank < k) {
      uint16_t* const s = ringSlots + r * RCAP + ringPos((uint32_t)(h + rank));
      int v = __hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG);
      if (v == 0) {
        int spins = 0;
        while ((v = __hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG)) == 0) {
          if (++spins > WF_SPIN_LIMIT) {
            raiseAbort();
            break;
          }
        }
      }
      __hip_atomic_store(s, (uint16_t)0, __ATOMIC_RELAXED, WF_WG);
      id = v - 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return k;
  };

  // ---- lane state: a traversal engine
  int path = -1;  // context id being traversed, -1 = idle
  int cur = DONE, hitRef = DONE, w = 0;
  int32_t link = DONE_PAIR;
  Ray ray;
  ray.o = ray.d = mk(0.0f, 0.0f, 0.0f);
  ray.time = 0.0f;
  float closest = SRT_INF, rayA = 0.0f, slabTol = SRT_INF;
  V3 rcpD = mk(0.0f, 0.0f, 0.0f), negOR = mk(0.0f, 0.0f, 0.0f);
  auto atNode = [&]() { return cur >= 0; };
  auto atPrim = [&]() { return (uint32_t)cur > (uint32_t)DONE; };
  auto popNext = [&]() {
    int next;
    if (HYBRID) {
      const int follows = link & 3;
      next = link >> 2;
      if (follows == 1) next = cur - 2;                                                                   // the next primitive of the same array
      if (follows == 2) next = (int)__builtin_amdgcn_raw_buffer_load_b32(rsSecond, (~cur) << 2, 0, 0);  // any other pair (rare)
      link &= ~3;
    } else {
      next = (int32_t)(int16_t)link;
      link >>= 16;
    }
    if (!SINGLE && !singleRoot && next == DONE && ++w < sc.numWorld) {
      next = worldRef(w);
      link = DONE_PAIR;
    }
    cur = next;
  };
  auto startTraversal = [&]() {  // world.hit(r, 0.001, infinity, rec)
    rayA = lenSq(ray.d);
    const bool certified = (sc.fastDivScene != 0) & fastDivOperandOk(ray.o.x, ray.d.x) & fastDivOperandOk(ray.o.y, ray.d.y) &
                           fastDivOperandOk(ray.o.z, ray.d.z);
    rcpD = mk(refinedRcp(ray.d.x), refinedRcp(ray.d.y), refinedRcp(ray.d.z));
    slabSetup(ray.o, rcpD, certified, negOR, slabTol);
    closest = SRT_INF;
    hitRef = DONE;
    link = DONE_PAIR;
    w = 0;
    cur = worldRef(0);
  };
  // the next camera ray of a context (main.cpp:204-216): A, B, C of its line
  auto cameraRayInto = [&](int at, uint32_t pxy, int s) {
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    Pcg rng;
    rng.key(seedMixed, (uint32_t)(py * a.imageWidth + px), (uint32_t)s);
    const float u = ((float)px + rng.uniform()) / (float)(a.imageWidth - 1);                      // main.cpp:210
    const float v = ((float)(a.imageHeight - py) + rng.uniform()) / (float)(a.imageHeight - 1);  // main.cpp:211
    Ray r;
    const DevCamera cam = cameraFromKernarg();
    cameraRay(cam, u, v, rng, r);
    bufStore4(rsPool, at, make_float4(r.o.x, r.o.y, r.o.z, r.time));
    bufStore4(rsPool, at + 16, make_float4(r.d.x, r.d.y, r.d.z, 0.0f));
    u32x4 Cn;
    Cn.x = (uint32_t)rng.state;
    Cn.y = (uint32_t)(rng.state >> 32);
    Cn.z = 0;
    Cn.w = 0;  // depth 0, in flight (a miss unless a hit step says otherwise)
    __builtin_amdgcn_raw_buffer_store_b128(Cn, rsPool, at + 32, 0, 0);
  };

  int idleTrips = 0;  // consecutive decisions that found nothing to do (bounded: see "Termination" above)
  uint32_t tick = 0;
  unsigned long long pCyc[WF_PROF_KINDS], pRuns[WF_PROF_KINDS], pLanes[WF_PROF_KINDS], pSched = 0;
  // what the scheduling decisions that looked at the rings saw, summed: decisions, lanes at nodes / at primitives /
  // finished / idle, READY fill, fill of the fullest served ring, RESTART fill
  unsigned long long pSaw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < WF_PROF_KINDS; ++k) pCyc[k] = pRuns[k] = pLanes[k] = 0;
  const unsigned long long pStart = PROFILE ? clock64() : 0;
  unsigned long long pT = pStart;  // the clock at the end of the last step: what follows until the next step begins is scheduling
  auto prof = [&](int kind, int lanes) {  // closes the step that began at pT
    if (PROFILE) {
      const unsigned long long now = clock64();
      pCyc[kind] += now - pT;
      pRuns[kind]++;
      pLanes[kind] += lanes;
      pT = now;
    }
  };
  // ---- the two traversal steps (as srt_render_kernel's, over the threaded tree in LDS)
  auto primStep = [&]() -> int {  // returns the lanes at nodes afterwards
    int nNodes;
      // ------------------------------------------------ sphere::hit / triangle::hit (as srt_render_kernel)
      for (int round = 0; round < SRT_PRIM_ROUNDS; ++round) {
        if (round > 0 && __popcll(__ballot(atPrim())) < a.primAgainMin) break;
        if (PROFILE) {
          pRuns[1]++;
          pLanes[1] += __popcll(__ballot(atPrim()));
        }
        if (atPrim()) {
          const int pr = ~cur;
          float t;
          bool ok;
          if (pr & 1) {
            const int off = (pr >> 1) * 48;
            const float4 s0 = bufLoad4(rsSpheres, off), s1 = bufLoad4(rsSpheres, off + 16);
            V3 center = mk(s0.x, s0.y, s0.z);
            if (__float_as_int(s1.w) & (1 << 30)) {  // sphere.h:47-52
              const float4 s2 = bufLoad4(rsSpheres, off + 32);
              center = center + ((ray.time - s2.x) / (s2.y - s2.x)) * (mk(s1.x, s1.y, s1.z) - center);
            }
            ok = sphereHitV(center, s0.w, ray, rayA, a.tMin, closest, t);
          } else {
            const int off = (pr >> 1) * 48;
            ok = triHitV<false>(bufLoad4(rsTris, off), bufLoad4(rsTris, off + 16), bufLoad4(rsTris, off + 32), ray, a.tMin, closest, t);
          }
          if (ok) {
            closest = t;
            hitRef = cur;
          }
          popNext();
        }
      }
      nNodes = __popcll(__ballot(atNode()));
      if (PROFILE) {
        const unsigned long long now = clock64();
        pCyc[1] += now - pT;
        pT = now;
      }
      return nNodes;
  };
  auto nodeBurst = [&](int nNodes) -> int {  // returns the lanes at nodes afterwards
      // ------------------------------------------------ bvhNode::hit, bvh.h:97-105, over the threaded tree in LDS
      const int keep = (nNodes * a.keepEighths) >> 3;
      int budget = a.nodeBurst, left;
      auto visit = [&](const float4 n0, const float4 n1) {  // aabb::hit (certified slab test, IEEE when undecided), then on
        bool undecided;
        bool hitBox = boxHitApprox<true>(n0, n1, rcpD, negOR, slabTol, a.tMin, closest, undecided);
        if (PROFILE) {  // how often the certificate cannot decide: lanes (-> lanes[8]) and wave visits that run the IEEE test (-> lanes[7])
          const unsigned long long mu = __ballot(undecided);
          pLanes[8] += __popcll(mu);
          pLanes[7] += mu != 0 ? 1 : 0;
        }
        if (undecided) hitBox = boxHit(n0, n1, ray, a.tMin, closest);
        link = __float_as_int(n1.w);
        cur = hitBox ? __float_as_int(n0.w) : (link >> LINK_SHIFT);
        if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {
          cur = worldRef(w);
          link = DONE_PAIR;
        }
      };
      auto nodeVisit = [&](bool mine) {
        const bool here = HYBRID ? mine && (uint32_t)cur < (uint32_t)resident : cur >= 0;
        if (PROFILE) {
          pRuns[0]++;
          pLanes[0] += __popcll(__ballot(here));
        }
        if (here) {
          const char* rec = ldsTree + (cur << 5);
          visit(*reinterpret_cast<const float4*>(rec), *reinterpret_cast<const float4*>(rec + 16));
        }
      };
      do {
        if (HYBRID) {
          // the lanes at a node outside LDS ask for its record; the others visit LDS nodes meanwhile (wfFarRounds - 1 times,
          // or until none of them is at one) and then fetch one more record from LDS, so that the last visit of the round
          // serves both groups in one execution
          const bool far = cur >= resident;
          float4 g0, g1;  // (read only by the lanes that load them below)
          if (far) {
            g0 = bufLoad4(rsNodes, cur << 5);
            g1 = bufLoad4(rsNodes, (cur << 5) + 16);
          }
          // no loop here: a loop header makes the compiler wait for the loads above before the LDS visits begin
          if (a.wfFarRounds > 1) nodeVisit(!far);
          if (a.wfFarRounds > 2) nodeVisit(!far);
          if (a.wfFarRounds > 3) nodeVisit(!far);
          const bool near = !far && (uint32_t)cur < (uint32_t)resident;
          if (near) {
            const char* rec = ldsTree + (cur << 5);
            g0 = *reinterpret_cast<const float4*>(rec);
            g1 = *reinterpret_cast<const float4*>(rec + 16);
          }
          const unsigned long long f0 = PROFILE ? clock64() : 0;
          if (far || near) visit(g0, g1);
          if (PROFILE) {
            pRuns[10]++;
            pLanes[10] += __popcll(__ballot(far));
            pLanes[11] += __popcll(__ballot(near));
            pCyc[10] += clock64() - f0;
          }
        } else {
#pragma unroll
          for (int u = 0; u < SRT_NODE_UNROLL; ++u) nodeVisit(true);
        }
        budget -= HYBRID ? a.wfFarRounds : SRT_NODE_UNROLL;
        left = __popcll(__ballot(atNode()));
      } while (budget > 0 && left >= keep);
      if (PROFILE) {
        const unsigned long long now = clock64();
        pCyc[0] += now - pT;
        pT = now;
      }
      return left;
  };
  // ---- swap step: finished walks out, READY contexts in
  auto swapStep = [&]() {
      // ------------------------------------------------ finished walks out, READY contexts in.  The new rays' lines are
      // asked for first and arrive while the finished walks are handed over (LDS traffic only).
      const bool fin = cur == DONE && path >= 0;
      const bool hit = fin && hitRef != DONE;
      const unsigned long long mNeed = __ballot(path < 0 || fin);
      int id = -1;
      // the class of the hit primitive's material picks the ring: asked for first, it is there when the claim is done
      int cls = 0;
      if (hit) cls = (int)__builtin_amdgcn_raw_buffer_load_b8(rsClass, ~hitRef, 0, 0);
      const unsigned long long q0 = PROFILE ? clock64() : 0;
      if (mNeed != 0) claim(WF_RING_READY, mNeed, __popcll(mNeed), 1, id);
      const unsigned long long q1 = PROFILE ? clock64() : 0;
      float4 A = make_float4(0.0f, 0.0f, 0.0f, 0.0f), B = A;
      if (id >= 0) {
        A = bufLoad4(rsPool, id << 7);
        B = bufLoad4(rsPool, (id << 7) + 16);
      }
      if (hit) {
        // what the hit step needs beyond the ray: t and the primitive (LDS).  A miss leaves nothing: the context already
        // says "in flight, nothing hit".
        __hip_atomic_store(&hitT[path], closest, __ATOMIC_RELAXED, WF_WG);
        __hip_atomic_store(&hitPrim[path], (PrimSlot)hitRef, __ATOMIC_RELAXED, WF_WG);
      }
      asm volatile("" ::: "memory");  // the ring slot is written after them (one wave's LDS operations execute in order)
      enqueue(!fin ? -1 : (hit ? WF_RING_HIT + cls : WF_RING_RESTART), path);
      if (fin) {
        path = -1;
        hitRef = DONE;
      }
      const unsigned long long q2 = PROFILE ? clock64() : 0;
      if (id >= 0) {
        ray.o = mk(A.x, A.y, A.z);
        ray.d = mk(B.x, B.y, B.z);
        ray.time = A.w;
        path = id;
        startTraversal();
      }
      if (PROFILE) {
        const unsigned long long q3 = clock64();
        pSaw[5] += q1 - q0;  // swap: claim
        pSaw[6] += q2 - q1;  // swap: hand-over of the finished walks
        pSaw[7] += q3 - q2;  // swap: new rays arrive, set-up
      }
      prof(2, __popcll(__ballot(fin)) + __popcll(__ballot(id >= 0)));
  };
  // ---- serve step: up to 64 contexts of ring `bestRing` (none unless `serveAtLeast` are there); returns how many
  auto serveStep = [&](int bestRing, int serveAtLeast) -> int {
      // ------------------------------------------------ serve a ring: 64 contexts in the same state
      int id;
      const unsigned long long h0 = PROFILE ? clock64() : 0;
      const int k = claim(bestRing, ~0ull, 64, serveAtLeast, id);
      if (k == 0) {  // another wave took them
        prof(8, 0);
        return 0;
      }
      const int at = id << 7;
      if (bestRing >= WF_RING_HIT && bestRing < WF_RING_HIT + WF_CLASSES) {
        // ---------------------------- rayColor's hit branch (main.cpp:42-51): one path vertex per lane
        bool toReady = false, toRestart = false;
        if (id >= 0) {
          const float4 A = bufLoad4(rsPool, at), B = bufLoad4(rsPool, at + 16);
          const u32x4 C = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 32, 0, 0);
          const float tHit = __hip_atomic_load(&hitT[id], __ATOMIC_RELAXED, WF_WG);
          const int pr = HYBRID ? ~(int)__hip_atomic_load(&hitPrim[id], __ATOMIC_RELAXED, WF_WG)
                                : ~(int)(int16_t)__hip_atomic_load(&hitPrim[id], __ATOMIC_RELAXED, WF_WG);
          Ray rIn;
          rIn.o = mk(A.x, A.y, A.z);
          rIn.d = mk(B.x, B.y, B.z);
          rIn.time = A.w;
          Pcg rng;
          rng.state = (uint64_t)C.x | ((uint64_t)C.y << 32);
          int depth = (int)(C.w & 0xffu);
          Record rec;
          const unsigned long long h1 = PROFILE ? clock64() : 0;
          if (pr & 1)
            sphereRecord(sc, pr >> 1, rIn, tHit, rec, false);
          else
            triRecord(sc, pr >> 1, rIn, tHit, rec, false);
          V3 att, emitted;
          Ray next;
          uint32_t fetches = 0;
          const unsigned long long h2 = PROFILE ? clock64() : 0;
          const bool scattered = shade<false, true>(sc, rsTexels, rIn, rec, rng, att, next, emitted, fetches, nullptr);
          if (PROFILE && bestRing == WF_RING_HIT + 1) {  // the ground's class: claim + context / record / shade (lane 0's clocks)
            const unsigned long long h3 = clock64();
            pSaw[1] += h1 - h0;
            pSaw[2] += h2 - h1;
            pSaw[3] += h3 - h2;
            pSaw[4] += 1;
          }
          V3 terminal = emitted;  // main.cpp:46-47
          bool done = true;
          if (scattered) {
            // emitted is (0,0,0) for every scattering material (material.h:18-20): keep the attenuation of this level
            if (depth < 4) {
              bufStore1(rsPool, at + 80 + 12 * depth, __float_as_uint(att.x));
              bufStore1(rsPool, at + 84 + 12 * depth, __float_as_uint(att.y));
              bufStore1(rsPool, at + 88 + 12 * depth, __float_as_uint(att.z));
            } else {
              const int slot = (3 * (depth - 4) * POOL + id) * 4;
              bufStore1(rsAttHi, slot, __float_as_uint(att.x));
              bufStore1(rsAttHi, slot + POOL * 4, __float_as_uint(att.y));
              bufStore1(rsAttHi, slot + POOL * 8, __float_as_uint(att.z));
            }
            depth++;
            done = depth >= a.maxBounce;  // main.cpp:36-37: out of bounces -> black
            terminal = mk(0.0f, 0.0f, 0.0f);
          }
          u32x4 Cn;
          Cn.x = (uint32_t)rng.state;
          Cn.y = (uint32_t)(rng.state >> 32);
          Cn.z = 0;
          if (done) {
            // the path ends here: its terminal radiance takes the (dead) direction's place
            bufStore4(rsPool, at + 16, make_float4(terminal.x, terminal.y, terminal.z, 0.0f));
            Cn.w = (uint32_t)depth | (WF_PEND_TERMINAL << 8);
            toRestart = true;
          } else {
            bufStore4(rsPool, at, make_float4(next.o.x, next.o.y, next.o.z, next.time));
            bufStore4(rsPool, at + 16, make_float4(next.d.x, next.d.y, next.d.z, 0.0f));
            Cn.w = (uint32_t)depth;
            toReady = true;
          }
          __builtin_amdgcn_raw_buffer_store_b128(Cn, rsPool, at + 32, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        enqueue(toReady ? WF_RING_READY : (toRestart ? WF_RING_RESTART : -1), id);
        prof(3 + bestRing - WF_RING_HIT, k);
      } else if (bestRing == WF_RING_RESTART) {
        // ---------------------------- a sample is in: miss / path end (main.cpp:39-40,49-51), pixel sum (main.cpp:217),
        // next camera ray of the item (main.cpp:204-216) -- or, after the item's last sample, its sum goes out and the
        // context goes for the next item
        bool toReady = false, toNewItem = false;
        if (id >= 0) {
          // the whole line at once: one round trip
          const float4 B = bufLoad4(rsPool, at + 16);
          const u32x4 C = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 32, 0, 0);
          float4 D = bufLoad4(rsPool, at + 48);
          const u32x4 E = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 64, 0, 0);
          const float4 F = bufLoad4(rsPool, at + 80), G = bufLoad4(rsPool, at + 96), H = bufLoad4(rsPool, at + 112);
          const int outIndex = (int)E.x, sEnd = (int)E.z;
          int s = (int)E.y;
          const int depth = (int)(C.w & 0xffu), pend = (int)((C.w >> 8) & 0xffu);
          V3 L = background;  // main.cpp:39-40 (the ray went out of the scene)
          if (pend == WF_PEND_TERMINAL) L = mk(B.x, B.y, B.z);
          // unwind the recursion: emitted + newColor * attenuation, innermost first (main.cpp:49-51)
          const float lv[12] = {F.x, F.y, F.z, F.w, G.x, G.y, G.z, G.w, H.x, H.y, H.z, H.w};
          for (int j = depth - 1; j >= 4; --j) {
            const int slot = (3 * (j - 4) * POOL + id) * 4;
            const float ax = __uint_as_float(bufLoad1(rsAttHi, slot)), ay = __uint_as_float(bufLoad1(rsAttHi, slot + POOL * 4)),
                        az = __uint_as_float(bufLoad1(rsAttHi, slot + POOL * 8));
            L = mk(0.0f + L.x * ax, 0.0f + L.y * ay, 0.0f + L.z * az);
          }
#pragma unroll
          for (int j = 3; j >= 0; --j)
            if (j < depth) L = mk(0.0f + L.x * lv[3 * j], 0.0f + L.y * lv[3 * j + 1], 0.0f + L.z * lv[3 * j + 2]);
          D.x += L.x;  // main.cpp:217
          D.y += L.y;
          D.z += L.z;
          s++;
          if (s < sEnd) {
            bufStore4(rsPool, at + 48, D);
            bufStore1(rsPool, at + 68, (uint32_t)s);
            cameraRayInto(at, E.w, s);
            toReady = true;
          } else {
            // the item is complete: its partial sum goes out (exact chunk sums: srt_path.h)
            if (outIndex >= 0) {
              if (a.fix)
                commitFixed(a.fix + outIndex, mk(D.x, D.y, D.z), a.fixLimit);
              else
                a.out[outIndex] = D;
            }
            toNewItem = true;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        enqueue(toReady ? WF_RING_READY : (toNewItem ? WF_RING_NEWITEM : -1), id);
        prof(6, k);
      } else {
        // ---------------------------- the next work item for every context here, with one atomic for the wave
        // (main.cpp:200-203: the pixel loop; work queues as in srt_render_kernel), and its first camera ray
        const bool need = id >= 0;
        const unsigned long long mF2 = __ballot(need);
        const int q = __builtin_amdgcn_readfirstlane(*waveQueue);
        bool gotItem = false, alive = true;
        int idx = 0;
        if (q >= 0) {
          int base = 0;
          if (lane == 0) base = atomicAdd(a.queue + 16 * q, __popcll(mF2));
          base = __builtin_amdgcn_readfirstlane(base);
          idx = base + __popcll(mF2 & laneBelow);
          const int qEnd = queueEnd(q);
          gotItem = need && idx < qEnd;
          if (base + __popcll(mF2) > qEnd) {
            // drained: lane 0 looks at every queue's counter, the wave moves to the fullest one (own XCD first)
            int nq = -1;
            if (lane == 0) {
              int bestLeft = 0;
              bool bestOwn = false;
              for (int kq = 0; kq < a.numQueues; ++kq) {
                const int left = queueEnd(kq) - __hip_atomic_load(a.queue + 16 * kq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool own = ((kq ^ qHome) & 7) == 0;
                if (left > 0 && ((own && !bestOwn) || (own == bestOwn && left > bestLeft))) {
                  bestLeft = left;
                  bestOwn = own;
                  nq = kq;
                }
              }
              *waveQueue = nq;
            }
            nq = __builtin_amdgcn_readfirstlane(nq);
            if (nq < 0) alive = false;
          }
        } else {
          alive = false;
        }
        bool toReady = false, again = false, retired = false;
        if (need) {
          if (!gotItem) {
            retired = !alive;  // nothing left anywhere: this context is done
            again = alive;     // the wave's queue ran dry under it: pull again from the new one
          } else {
            // idx -> (local tile, chunk, pixel of the tile), without integer divisions (as srt_render_kernel)
            const int group = idx >> 6, ln = idx & 63;
            int u = (int)((float)group * a.rcpUnitGroups);
            int inUnit = group - u * a.unitGroups;
            if (inUnit < 0) {
              u--;
              inUnit += a.unitGroups;
            } else if (inUnit >= a.unitGroups) {
              u++;
              inUnit -= a.unitGroups;
            }
            int tileInUnit = (int)((float)inUnit * a.rcpChunks);
            int chunk = inUnit - tileInUnit * a.sppChunks;
            if (chunk < 0) {
              tileInUnit--;
              chunk += a.sppChunks;
            } else if (chunk >= a.sppChunks) {
              tileInUnit++;
              chunk -= a.sppChunks;
            }
            const int localTile = (q + u * a.numQueues) * a.unitTiles + tileInUnit;  // may pad past numLocalTiles
            const int tile = a.tileFirst + localTile * a.tileStride;
            const uint32_t txy = a.tileXY[tile < a.numTiles ? tile : 0];
            const int px = (int)(txy & 0xffffu) * SRT_TILE_W + (ln & (SRT_TILE_W - 1));
            const int py = (int)(txy >> 16) * SRT_TILE_H + (ln >> 3);
            const uint32_t pxy = (uint32_t)px | ((uint32_t)py << 16);
            const int s0 = a.sampleFirst + chunk * a.sppBase + min(chunk, a.sppRem);
            const int s1 = s0 + a.sppBase + (chunk < a.sppRem ? 1 : 0);
            // maxBounce <= 0: rayColor returns black before tracing anything (main.cpp:36-37)
            const bool valid = localTile < a.numLocalTiles && tile < a.numTiles && px < a.imageWidth && py < a.imageHeight && a.maxBounce > 0;
            const int outIndex = localTile < a.numLocalTiles ? chunk * a.chunkStride + localTile * SRT_TILE_PIXELS + ln : -1;
            const float4 D = make_float4(0.0f, 0.0f, 0.0f, (float)(s1 - s0));
            if (valid) {
              bufStore4(rsPool, at + 48, D);
              u32x4 En;
              En.x = (uint32_t)outIndex;
              En.y = (uint32_t)s0;
              En.z = (uint32_t)s1;
              En.w = pxy;
              __builtin_amdgcn_raw_buffer_store_b128(En, rsPool, at + 64, 0, 0);
              cameraRayInto(at, pxy, s0);
              toReady = true;
            } else {
              // an item with nothing to trace (a pixel outside the image, or no bounces): its zero sum is written at once
              if (outIndex >= 0 && !a.fix) a.out[outIndex] = D;
              again = true;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        enqueue(toReady ? WF_RING_READY : (again ? WF_RING_NEWITEM : -1), id);
        const unsigned long long mR = __ballot(retired);
        if (mR != 0 && lane == __ffsll((long long)mR) - 1) __hip_atomic_fetch_sub(&ctl[WF_CTL_LIVE], __popcll(mR), __ATOMIC_RELAXED, WF_WG);
        prof(9, k);
      }
      return k;
  };
  auto flushProfile = [&]() {
  if (PROFILE && a.stats && lane == 0) {
      for (int k = 0; k < WF_PROF_KINDS; ++k) {
        atomicAdd(&a.stats[32 + k], pCyc[k]);
        atomicAdd(&a.stats[32 + WF_PROF_KINDS + k], pRuns[k]);
        atomicAdd(&a.stats[32 + 2 * WF_PROF_KINDS + k], pLanes[k]);
      }
      atomicAdd(&a.stats[32 + 3 * WF_PROF_KINDS], pSched);
      atomicAdd(&a.stats[32 + 3 * WF_PROF_KINDS + 1], (unsigned long long)(clock64() - pStart));
      for (int k = 0; k < 8; ++k) atomicAdd(&a.stats[32 + 3 * WF_PROF_KINDS + 2 + k], pSaw[k]);
    }
  };
  bool traverseFirst = false;  // the last decision was "go on traversing": one step before the rings are looked at again
  for (;;) {
    // ---- traverse for as long as the rings need no look: a tight loop of node bursts and primitive steps.  The rings are
    // looked at (below) when this wave has lanes to refill, nothing to traverse, or every fourth pass: with sixteen waves
    // deciding, a full batch is still seen within a fraction of the time it took to fill.  EVERY traversal step runs in
    // this loop (a decision that picks one comes back here with traverseFirst set): a second place to run them made the
    // compiler copy the lanes' whole state, forty registers, on its way back to the loop header.
    int nN, nP, free, nNknown = -1;  // (a step that ends with a count of the lanes at nodes hands it to the next pass)
    for (;;) {
      nN = nNknown >= 0 ? nNknown : __popcll(__ballot(atNode()));
      nP = __popcll(__ballot(atPrim()));
      free = 64 - nN - nP;  // lanes whose walk is over or that hold no context (cur == DONE either way)
      if (nN + nP == 0) break;
      if (!traverseFirst && (free >= a.wfSwapMin || (++tick & 3u) == 0)) break;
      traverseFirst = false;
      if (PROFILE) {
        const unsigned long long now = clock64();
        pSched += now - pT;
        pT = now;
      }
      if (nP >= a.primMin || nN == 0) {
        const int nNodes = primStep();
        nNknown = nNodes >= a.fuseMin ? nodeBurst(nNodes) : nNodes;
      } else {
        nNknown = nodeBurst(nN);
      }
    }
    // ---- scheduling decision with the rings in view
    const int nF = __popcll(__ballot(cur == DONE && path >= 0));
    int pick, serveAtLeast = 64, bestRing = WF_RING_RESTART;
    {
      unsigned long long* cw = reinterpret_cast<unsigned long long*>(ctl + 16);
      // relaxed 64-bit atomic loads: fresh values every time, two words per LDS read, broadcast to the wave
      const unsigned long long w0 = __hip_atomic_load(cw + 0, __ATOMIC_RELAXED, WF_WG), w1 = __hip_atomic_load(cw + 1, __ATOMIC_RELAXED, WF_WG),
                               w2 = __hip_atomic_load(cw + 2, __ATOMIC_RELAXED, WF_WG), w3 = __hip_atomic_load(cw + 3, __ATOMIC_RELAXED, WF_WG),
                               w4 = __hip_atomic_load(cw + 4, __ATOMIC_RELAXED, WF_WG), w5 = __hip_atomic_load(cw + 5, __ATOMIC_RELAXED, WF_WG),
                               w6 = __hip_atomic_load(cw + 6, __ATOMIC_RELAXED, WF_WG);
      // words 16 + 2r / 17 + 2r tail / head of ring r, 28 live, 29 abort
      if ((int)(w6 >> 32) != 0) break;  // abort
      auto fill = [](unsigned long long th) { return (int)((uint32_t)th - (uint32_t)(th >> 32)); };
      const int readyAvail = __builtin_amdgcn_readfirstlane(fill(w0));
      const int av[5] = {fill(w1), fill(w2), fill(w3), fill(w4), fill(w5)};  // RESTART, HIT 0..2, NEWITEM
      int bestAvail = av[0];
#pragma unroll
      for (int r = 1; r < 5; ++r)
        if (av[r] > bestAvail) {
          bestAvail = av[r];
          bestRing = WF_RING_RESTART + r;
        }
      bestRing = __builtin_amdgcn_readfirstlane(bestRing);
      bestAvail = __builtin_amdgcn_readfirstlane(bestAvail);
      // what a swap would move: finished walks out, READY contexts into the lanes without one
      const int swapGain = nF + (readyAvail < free ? readyAvail : free);
      if (bestAvail >= 64)
        pick = W_SERVE;  // a full batch is always worth a step (and feeds READY)
      else if (swapGain >= a.wfSwapMin)
        pick = W_SWAP;
      else if (nP >= a.primMin || (nN == 0 && nP > 0))
        pick = W_PRIM;
      else if (nN > 0)
        pick = W_NODE;
      else if (swapGain > 0)
        pick = W_SWAP;
      else if (bestAvail > 0) {
        pick = W_SERVE;  // nothing to traverse here and nothing READY: serve what there is
        serveAtLeast = 1;
      } else {
        if (__builtin_amdgcn_readfirstlane((int)w6) <= 0) break;  // every context has retired
        if (++idleTrips > WF_SPIN_LIMIT) {
          raiseAbort();
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        prof(7, 0);
        continue;
      }
      idleTrips = 0;
      if (PROFILE) {
        pSaw[0]++;
        (void)readyAvail;
      }
    }
    if (PROFILE) {
      const unsigned long long now = clock64();
      pSched += now - pT;
      pT = now;
    }

    if (pick == W_PRIM || pick == W_NODE) {
      traverseFirst = true;  // (the loop above picks the step by the same rule)
    } else if (pick == W_SWAP) {
      swapStep();
    } else if (pick == W_SERVE) {
      serveStep(bestRing, serveAtLeast);
    }
  }
  flushProfile();
}

extern "C" {
int srt_launch_render_wf(const RenderArgs* a, int profile, int grid, size_t ldsBytes, hipStream_t stream) {
  typedef void (*Kernel)(const RenderArgs);
  // (hybrid form: the single-root instance is worth +12 to +15 % on cache-resident trees and costs 5 % on the HBM-bound
  // soups of 4 M triangles and more, where the shorter visit only crowds the memory system: profiles/r03/hybrid.txt)
  const Kernel k = a->scene.nodesWf ? (profile ? srt_render_wf_kernel<false, true, true>
                                       : a->scene.numWorld == 1 && a->scene.numNodes <= (1 << 20) ? srt_render_wf_kernel<true, false, true>
                                                                                                   : srt_render_wf_kernel<false, false, true>)
                   : profile       ? srt_render_wf_kernel<false, true, false>
                                   : (a->scene.numWorld == 1 ? srt_render_wf_kernel<true, false, false> : srt_render_wf_kernel<false, false, false>);
  if (ldsBytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(WF_BLOCK), ldsBytes, stream, *a);
  return (int)hipGetLastError();
}
}
