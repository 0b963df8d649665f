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
//     sum: 128 bytes in global memory, L2-resident) belongs to no lane.  Each workgroup owns `poolSize` of them
//     (about 1.5 per lane).
//   * lanes are TRAVERSAL ENGINES: a lane holds only a ray and its walk of the threaded tree (DevScene::nodeThread,
//     bvh.h:97-105 in bvh.h's order).  When the walk ends it writes (t, primitive) into the context, hands the
//     context to a queue and takes the next READY context (swap step).
//   * queues are rings of context ids in LDS: READY (a fresh ray to traverse), RESTART (path ended or ray missed:
//     main.cpp:39-40,49-51,217 and the next camera ray, main.cpp:204-216) and one HIT ring per MATERIAL CLASS
//     (triangle + pbr / sphere + pbr / everything else).  Any wave that finds 64 entries in a ring takes them and
//     runs that step for them: hit shading (main.cpp:42-51 + material.h) and restarts run at 64 of 64 lanes, and
//     a hit step executes one class's code.  A wave with nothing to traverse serves partial batches, which also
//     drains the end of the frame.
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
#define WF_RINGS (2 + WF_CLASSES)
#define WF_SPIN_LIMIT (1 << 24)
#define WF_CTX_BYTES 128
// control words (LDS, behind the tree): 0..15 the waves' current work queue; 16 + 2r / 17 + 2r ring r's tail (reserved)
// and head (claimed); 32 live contexts; 33 abort
#define WF_CTL_TAIL(r) (16 + 2 * (r))
#define WF_CTL_HEAD(r) (17 + 2 * (r))
#define WF_CTL_LIVE 32
#define WF_CTL_ABORT 33
#define WF_CTL_WORDS 64
// context meta word: depth | pend << 8
#define WF_PEND_NONE 3  // nothing to add (fresh context, empty item)
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

// SINGLE: the world list is one tree (as in srt_render_kernel)
template <bool SINGLE>
__global__ __launch_bounds__(WF_BLOCK, 4) void srt_render_wf_kernel(const RenderArgs a) {
  constexpr int32_t DONE = (int32_t)0xFFFF8000;       // the 16-bit "no reference", sign-extended
  constexpr int32_t DONE_PAIR = (int32_t)0x80008000;  // both halves of a thread link
  extern __shared__ int32_t lds[];
  const DevScene& sc = a.scene;
  char* const ldsTree = reinterpret_cast<char*>(lds);
  const int treeBytes = sc.numNodes * 32;
  int32_t* const ctl = reinterpret_cast<int32_t*>(ldsTree + treeBytes);
  uint16_t* const ringSlots = reinterpret_cast<uint16_t*>(ctl + WF_CTL_WORDS);
  const int RCAP = a.wfRingCap, RMASK = RCAP - 1, POOL = a.wfPoolSize;
  const int lane = threadIdx.x & 63;
  const unsigned long long laneBelow = (1ull << lane) - 1ull;
  const uint64_t seedMixed = mix64(a.seed);
  const V3 background = ld3(a.background);
  const Rsrc rsNodes = makeRsrc(sc.nodes, sc.numNodes * 32);
  const Rsrc rsTris = makeRsrc(sc.triTest, sc.numTris * 48);
  const Rsrc rsSpheres = makeRsrc(sc.spheres, sc.numSpheres * 48);
  const Rsrc rsTexels = makeRsrc(sc.texels, sc.texelBytes);
  const Rsrc rsClass = makeRsrc(sc.primClass, sc.numPrimClass);
  // this workgroup's contexts: id << 7 is a context's byte offset; its upper attenuation levels (bounce >= 4) live in
  // [level - 4][channel][id] behind the pools
  const Rsrc rsPool = makeRsrc(a.wfPool + (size_t)blockIdx.x * POOL * WF_CTX_BYTES, POOL * WF_CTX_BYTES);
  const int hiLevels = a.maxBounce > 4 ? a.maxBounce - 4 : 0;
  const Rsrc rsAttHi = makeRsrc(a.wfAttHi + (size_t)blockIdx.x * 3 * hiLevels * POOL, 3 * hiLevels * POOL * 4);
  const bool singleRoot = SINGLE || sc.numWorld == 1;
  auto localRef = [&](int r) { return r >= 0 ? r >> 5 : r; };

  // ---- set-up: the threaded tree into LDS (as srt_render_kernel LDSTREE), empty rings, every context in RESTART
  {
    float4* dst = reinterpret_cast<float4*>(ldsTree);
    for (int i = threadIdx.x; i < sc.numNodes * 2; i += WF_BLOCK) {
      float4 v = bufLoad4(rsNodes, 16 * i);
      const int r = __float_as_int(v.w);
      if (i & 1)
        v.w = __int_as_float(sc.nodeThread[i >> 1]);
      else if (r >= 0)
        v.w = __int_as_float(r >> 5);
      dst[i] = v;
    }
    for (int i = threadIdx.x; i < WF_CTL_WORDS; i += WF_BLOCK) ctl[i] = 0;
    for (int i = threadIdx.x; i < WF_RINGS * RCAP; i += WF_BLOCK) ringSlots[i] = 0;
    __syncthreads();
    for (int id = threadIdx.x; id < POOL; id += WF_BLOCK) {
      ringSlots[WF_RING_RESTART * RCAP + id] = (uint16_t)(id + 1);
      const int at = id << 7;
      bufStore4(rsPool, at + 32, make_float4(0.0f, 0.0f, __int_as_float(DONE), __int_as_float(WF_PEND_NONE << 8)));
      bufStore4(rsPool, at + 48, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
      bufStore4(rsPool, at + 64, make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f));  // outIndex, s, sEnd, px | py << 16
    }
    if (threadIdx.x == 0) {
      ctl[WF_CTL_TAIL(WF_RING_RESTART)] = POOL;
      ctl[WF_CTL_LIVE] = POOL;
    }
    if (threadIdx.x < 16) ctl[threadIdx.x] = (int)(blockIdx.x % (unsigned)a.numQueues);  // the waves' home work queue
    __syncthreads();  // includes the wait for the context stores
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
  auto enqueue = [&](int r, bool want, int id) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add(&ctl[WF_CTL_TAIL(r)], __popcll(m), __ATOMIC_RELAXED, WF_WG);
    base = __shfl(base, leader);
    if (want) {
      uint16_t* const s = ringSlots + r * RCAP + ((base + __popcll(m & laneBelow)) & RMASK);
      int spins = 0;
      while (__hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG) != 0) {
        if (++spins > WF_SPIN_LIMIT) {
          raiseAbort();
          break;
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
      h = __hip_atomic_load(&ctl[WF_CTL_HEAD(r)], __ATOMIC_RELAXED, WF_WG);
      for (;;) {
        const int t = __hip_atomic_load(&ctl[WF_CTL_TAIL(r)], __ATOMIC_RELAXED, WF_WG);
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
      uint16_t* const s = ringSlots + r * RCAP + ((h + rank) & RMASK);
      int v, spins = 0;
      while ((v = __hip_atomic_load(s, __ATOMIC_RELAXED, WF_WG)) == 0) {
        if (++spins > WF_SPIN_LIMIT) {
          raiseAbort();
          break;
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
    int next = (int32_t)(int16_t)link;
    link >>= 16;
    if (!SINGLE && !singleRoot && next == DONE && ++w < sc.numWorld) {
      next = localRef(sc.world[w]);
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
    cur = localRef(sc.world[0]);
  };

  int idleTrips = 0;  // consecutive decisions that found nothing to do (bounded: see "Termination" above)
  for (;;) {
    // ---- scheduling decision
    const unsigned long long mN = __ballot(atNode()), mP = __ballot(atPrim()), mF = __ballot(cur == DONE && path >= 0),
                             mI = __ballot(path < 0);
    const int nN = __popcll(mN), nP = __popcll(mP), nF = __popcll(mF), nI = __popcll(mI);
    // ring fill (every lane reads the same words: broadcast reads)
    int avail[WF_RINGS];
#pragma unroll
    for (int r = 0; r < WF_RINGS; ++r)
      avail[r] = (int)((uint32_t)__hip_atomic_load(&ctl[WF_CTL_TAIL(r)], __ATOMIC_RELAXED, WF_WG) -
                       (uint32_t)__hip_atomic_load(&ctl[WF_CTL_HEAD(r)], __ATOMIC_RELAXED, WF_WG));
    if (__hip_atomic_load(&ctl[WF_CTL_ABORT], __ATOMIC_RELAXED, WF_WG) != 0) break;
    int bestRing = WF_RING_RESTART, bestAvail = avail[WF_RING_RESTART];
#pragma unroll
    for (int r = WF_RING_HIT; r < WF_RINGS; ++r)
      if (avail[r] > bestAvail) {
        bestAvail = avail[r];
        bestRing = r;
      }
    bestRing = __builtin_amdgcn_readfirstlane(bestRing);
    bestAvail = __builtin_amdgcn_readfirstlane(bestAvail);
    const int readyAvail = __builtin_amdgcn_readfirstlane(avail[WF_RING_READY]);
    const bool canSwap = nF > 0 || (nI > 0 && readyAvail > 0);
    int pick, serveAtLeast = 64;
    if (nF + nI >= a.wfSwapBig && canSwap)
      pick = W_SWAP;  // half the wave has nothing to traverse
    else if (bestAvail >= 64)
      pick = W_SERVE;  // a full batch is always worth a step
    else if (nF + nI >= a.wfSwapMin && canSwap)
      pick = W_SWAP;
    else if (nP >= a.primMin || (nN == 0 && nP > 0))
      pick = W_PRIM;
    else if (nN > 0)
      pick = W_NODE;
    else if (canSwap)
      pick = W_SWAP;
    else if (bestAvail > 0) {
      pick = W_SERVE;  // nothing to traverse here: serve what there is
      serveAtLeast = 1;
    } else {
      if (__hip_atomic_load(&ctl[WF_CTL_LIVE], __ATOMIC_RELAXED, WF_WG) <= 0) break;  // every context has retired
      if (++idleTrips > WF_SPIN_LIMIT) {
        raiseAbort();
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    idleTrips = 0;

    int nNodes = nN;
    if (pick == W_PRIM) {
      // ------------------------------------------------ sphere::hit / triangle::hit (as srt_render_kernel)
      for (int round = 0; round < SRT_PRIM_ROUNDS; ++round) {
        if (round > 0 && __popcll(__ballot(atPrim())) < a.primAgainMin) break;
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
      if (nNodes >= a.fuseMin) pick = W_NODE;
    }

    if (pick == W_NODE) {
      // ------------------------------------------------ bvhNode::hit, bvh.h:97-105, over the threaded tree in LDS
      const int keep = (nNodes * a.keepEighths) >> 3;
      int budget = a.nodeBurst;
      auto nodeVisit = [&]() {
        if (atNode()) {
          const float4 n0 = *reinterpret_cast<const float4*>(ldsTree + (cur << 5));
          const float4 n1 = *reinterpret_cast<const float4*>(ldsTree + (cur << 5) + 16);
          bool undecided;
          bool hitBox = boxHitApprox<true>(n0, n1, rcpD, negOR, slabTol, a.tMin, closest, undecided);
          if (undecided) hitBox = boxHit(n0, n1, ray, a.tMin, closest);
          link = __float_as_int(n1.w);
          cur = hitBox ? __float_as_int(n0.w) : (link >> 16);
          if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {
            cur = localRef(sc.world[w]);
            link = DONE_PAIR;
          }
        }
      };
      do {
#pragma unroll
        for (int u = 0; u < SRT_NODE_UNROLL; ++u) nodeVisit();
        budget -= SRT_NODE_UNROLL;
      } while (budget > 0 && __popcll(__ballot(atNode())) >= keep);
    } else if (pick == W_SWAP) {
      // ------------------------------------------------ finished walks out, READY contexts in
      const bool fin = cur == DONE && path >= 0;
      const bool hit = fin && hitRef != DONE;
      int cls = -1;
      if (hit) {
        // what the hit step needs beyond the ray: t and the primitive; the class of its material picks the ring.
        // (A miss stores nothing: the context already says "in flight, nothing hit".)
        bufStore1(rsPool, (path << 7) + 28, __float_as_uint(closest));
        bufStore1(rsPool, (path << 7) + 40, (uint32_t)hitRef);
        cls = (int)__builtin_amdgcn_raw_buffer_load_b8(rsClass, ~hitRef, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      enqueue(WF_RING_RESTART, fin && !hit, path);
#pragma unroll
      for (int c = 0; c < WF_CLASSES; ++c) enqueue(WF_RING_HIT + c, hit && cls == c, path);
      if (fin) path = -1;
      const unsigned long long mNeed = __ballot(path < 0);
      int id = -1;
      if (mNeed != 0) claim(WF_RING_READY, mNeed, __popcll(mNeed), 1, id);
      if (id >= 0) {
        const float4 A = bufLoad4(rsPool, id << 7), B = bufLoad4(rsPool, (id << 7) + 16);
        ray.o = mk(A.x, A.y, A.z);
        ray.d = mk(B.x, B.y, B.z);
        ray.time = A.w;
        path = id;
        startTraversal();
      }
    } else if (pick == W_SERVE) {
      // ------------------------------------------------ serve a ring: 64 contexts in the same state
      int id;
      const int k = claim(bestRing, ~0ull, 64, serveAtLeast, id);
      if (k == 0) continue;  // another wave took them
      const int at = id << 7;
      if (bestRing >= WF_RING_HIT) {
        // ---------------------------- rayColor's hit branch (main.cpp:42-51): one path vertex per lane
        bool toReady = false, toRestart = false;
        if (id >= 0) {
          const float4 A = bufLoad4(rsPool, at), B = bufLoad4(rsPool, at + 16);
          const u32x4 C = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 32, 0, 0);
          Ray rIn;
          rIn.o = mk(A.x, A.y, A.z);
          rIn.d = mk(B.x, B.y, B.z);
          rIn.time = A.w;
          Pcg rng;
          rng.state = (uint64_t)C.x | ((uint64_t)C.y << 32);
          int depth = (int)(C.w & 0xffu);
          Record rec;
          const int pr = ~(int)C.z;
          if (pr & 1)
            sphereRecord(sc, pr >> 1, rIn, B.w, rec, false);
          else
            triRecord(sc, pr >> 1, rIn, B.w, rec, false);
          V3 att, emitted;
          Ray next;
          uint32_t fetches = 0;
          const bool scattered = shade<false, true>(sc, rsTexels, rIn, rec, rng, att, next, emitted, fetches, nullptr);
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
          Cn.z = (uint32_t)DONE;
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
        enqueue(WF_RING_READY, toReady, id);
        enqueue(WF_RING_RESTART, toRestart, id);
      } else {
        // ---------------------------- path restart: miss / path end (main.cpp:39-40,49-51), pixel sum (main.cpp:217),
        // next work item, next camera ray (main.cpp:204-216)
        bool toReady = false, again = false, retired = false;
        float4 D = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int outIndex = -1, s = 0, sEnd = 0;
        uint32_t pxy = 0;
        if (id >= 0) {
          const u32x4 C = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 32, 0, 0);
          const u32x4 E = __builtin_amdgcn_raw_buffer_load_b128(rsPool, at + 64, 0, 0);
          D = bufLoad4(rsPool, at + 48);
          outIndex = (int)E.x;
          s = (int)E.y;
          sEnd = (int)E.z;
          pxy = E.w;
          const int depth = (int)(C.w & 0xffu), pend = (int)((C.w >> 8) & 0xffu);
          if (pend != WF_PEND_NONE) {
            V3 L = background;  // main.cpp:39-40 (pend 0: the ray went out of the scene)
            if (pend == WF_PEND_TERMINAL) {
              const float4 T = bufLoad4(rsPool, at + 16);
              L = mk(T.x, T.y, T.z);
            }
            // unwind the recursion: emitted + newColor * attenuation, innermost first (main.cpp:49-51)
            for (int j = depth - 1; j >= 0; --j) {
              float ax, ay, az;
              if (j < 4) {
                ax = __uint_as_float(bufLoad1(rsPool, at + 80 + 12 * j));
                ay = __uint_as_float(bufLoad1(rsPool, at + 84 + 12 * j));
                az = __uint_as_float(bufLoad1(rsPool, at + 88 + 12 * j));
              } else {
                const int slot = (3 * (j - 4) * POOL + id) * 4;
                ax = __uint_as_float(bufLoad1(rsAttHi, slot));
                ay = __uint_as_float(bufLoad1(rsAttHi, slot + POOL * 4));
                az = __uint_as_float(bufLoad1(rsAttHi, slot + POOL * 8));
              }
              L = mk(0.0f + L.x * ax, 0.0f + L.y * ay, 0.0f + L.z * az);
            }
            D.x += L.x;  // main.cpp:217
            D.y += L.y;
            D.z += L.z;
            s++;
          }
        }
        // work item finished (or none yet): write it, pull the next one with one atomic per wave
        const bool needItem = id >= 0 && s >= sEnd;
        const unsigned long long mF2 = __ballot(needItem);
        if (mF2 != 0) {
          if (needItem && outIndex >= 0) {
            if (a.fix)
              commitFixed(a.fix + outIndex, mk(D.x, D.y, D.z));
            else
              a.out[outIndex] = D;
          }
          const int leader = __ffsll((long long)mF2) - 1;
          const int q = __builtin_amdgcn_readfirstlane(*waveQueue);
          bool gotItem = false, alive = true;
          int idx = 0;
          if (q >= 0) {
            int base = 0;
            if (lane == leader) base = atomicAdd(a.queue + 16 * q, __popcll(mF2));
            base = __shfl(base, leader);
            idx = base + __popcll(mF2 & laneBelow);
            const int qEnd = queueEnd(q);
            gotItem = needItem && idx < qEnd;
            if (base + __popcll(mF2) > qEnd) {
              // drained: the leader looks at every queue's counter, the wave moves to the fullest one (own XCD first)
              int nq = -1;
              if (lane == leader) {
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
              nq = __shfl(nq, leader);
              if (!gotItem && nq < 0) alive = false;
            }
          } else {
            alive = false;
          }
          if (needItem) {
            outIndex = -1;
            D = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (!gotItem) {
              s = sEnd = 0;
              retired = !alive;  // nothing left anywhere: this context is done
              again = alive;     // an empty item: pull again from the wave's new queue
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
              pxy = (uint32_t)px | ((uint32_t)py << 16);
              const int s0 = a.sampleFirst + chunk * a.sppBase + min(chunk, a.sppRem);
              const int s1 = s0 + a.sppBase + (chunk < a.sppRem ? 1 : 0);
              D.w = (float)(s1 - s0);
              const bool valid = localTile < a.numLocalTiles && tile < a.numTiles && px < a.imageWidth && py < a.imageHeight && a.maxBounce > 0;
              s = s0;
              sEnd = valid ? s1 : s0;
              outIndex = localTile < a.numLocalTiles ? chunk * a.chunkStride + localTile * SRT_TILE_PIXELS + ln : -1;
              again = !valid;  // an item with nothing to trace (outside the image, or no bounces): its zero sum is written next time round
            }
          }
        }
        if (id >= 0) {
          u32x4 Cn;
          Cn.z = (uint32_t)DONE;
          Cn.w = (uint32_t)(WF_PEND_NONE << 8);
          Cn.x = Cn.y = 0;
          if (!retired && s < sEnd) {
            const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
            Pcg rng;
            rng.key(seedMixed, (uint32_t)(py * a.imageWidth + px), (uint32_t)s);
            const float u = ((float)px + rng.uniform()) / (float)(a.imageWidth - 1);                      // main.cpp:210
            const float v = ((float)(a.imageHeight - py) + rng.uniform()) / (float)(a.imageHeight - 1);  // main.cpp:211
            Ray r;
            cameraRay(a.cam, u, v, rng, r);
            bufStore4(rsPool, at, make_float4(r.o.x, r.o.y, r.o.z, r.time));
            bufStore4(rsPool, at + 16, make_float4(r.d.x, r.d.y, r.d.z, 0.0f));
            Cn.x = (uint32_t)rng.state;
            Cn.y = (uint32_t)(rng.state >> 32);
            Cn.w = 0;  // depth 0, in flight
            toReady = true;
            again = false;
          }
          __builtin_amdgcn_raw_buffer_store_b128(Cn, rsPool, at + 32, 0, 0);
          bufStore4(rsPool, at + 48, D);
          u32x4 En;
          En.x = (uint32_t)outIndex;
          En.y = (uint32_t)s;
          En.z = (uint32_t)sEnd;
          En.w = pxy;
          __builtin_amdgcn_raw_buffer_store_b128(En, rsPool, at + 64, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        enqueue(WF_RING_READY, toReady, id);
        enqueue(WF_RING_RESTART, again && !toReady, id);
        const unsigned long long mR = __ballot(retired);
        if (mR != 0 && lane == __ffsll((long long)mR) - 1) __hip_atomic_fetch_sub(&ctl[WF_CTL_LIVE], __popcll(mR), __ATOMIC_RELAXED, WF_WG);
      }
    }
  }
}

extern "C" {
int srt_launch_render_wf(const RenderArgs* a, int grid, size_t ldsBytes, hipStream_t stream) {
  typedef void (*Kernel)(const RenderArgs);
  const Kernel k = a->scene.numWorld == 1 ? srt_render_wf_kernel<true> : srt_render_wf_kernel<false>;
  if (ldsBytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(WF_BLOCK), ldsBytes, stream, *a);
  return (int)hipGetLastError();
}
}
