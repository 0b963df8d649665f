// srt_kernels.hip -- the per-pixel path-tracing loop as hand-written HIP for gfx950 (MI355X).
//
// Replaces main.cpp:200-227 (pixel/sample loop), camera::getRay (camera.h:40-46),
// rayColor (main.cpp:33-52), hittableList/bvhNode/sphere/triangle::hit
// (hittablelist.h:33-47, bvh.h:97-105, sphere.h:54-83, model.h:104-181), the four
// material::scatter bodies (material.h:91-97,108-126,144-150,156-245), pbr.h:58-81,
// texture::value (texture.h:26-28,42-48,129-148) and writeColorTarget (color.h:25-41).
//
// Execution model (CDNA4), details at srt_render_kernel:
//   * every lane is a persistent worker: it pulls a work item (one pixel x one chunk of its
//     samples; 64 consecutive items = one 8x8 tile) from an atomic counter, runs the chunk's samples in
//     index order and adds them in that order (main.cpp:204-218), hands in one float partial sum and pulls
//     the next item; the partial sums of a pixel are added exactly (64-bit fixed point, "Chunk sums" below).
//     Accumulators are reproducible bit for bit and independent of scheduling, tiling and GPU count.
//   * a wave-level scheduler runs ONE kind of step per trip -- BVH node visits, primitive tests, hit
//     shading or path restarts -- for the lanes that are in that state, picked from ballot counts, so the
//     64 lanes stay busy although their paths are at different depths of different subtrees.
//   * bounce recursion is a loop; the per-bounce attenuations sit in a small per-lane LDS stack and are
//     folded innermost-first at path end, which reproduces the recursion's rounding
//     (E + A*(E' + A'*(...))) exactly.
//   * BVH traversal is a per-lane DFS in the reference's fixed left-then-right order with the pending
//     right children on a per-lane LDS stack laid out [slot][thread] (bank = thread: conflict-free).
//   * a scene whose whole node array fits a CU's 160 KB of LDS is traversed OUT OF LDS: one workgroup of 1024
//     threads per CU copies the node records in and keeps 16-bit stacks behind them (LDSTREE, see the kernel);
//     larger scenes read the records through the vector L1 with 256-thread workgroups.
//   * a hit reads ONE flattened record per material (scalars + texture slots resolved at upload), its type and
//     flags travel in the primitive's material word: two round trips to memory per shaded hit (shade).
//   * RNG: PCG32 keyed by (seed, pixel, sample) per lane.
//   * all arithmetic keeps the reference's operation order; built with -ffp-contract=off; divisions and
//     sqrt are IEEE (hipcc default); the slab test is decided from one FMA per plane under an error
//     certificate, with the IEEE divisions for the visits it cannot decide (boxHitApprox).
//   * SRT_TRAVERSE_CLOSEST (not the parity path) walks 64-byte records that hold both children's boxes,
//     nearer child first.
// No MFMA: there is no dense contraction on this path.

#include "srt_path.h"


// =================================================================== render
// main.cpp:200-227 for the tiles of this rank.
//
// Every lane is an independent persistent worker: it pulls a work item (one pixel x one
// sample chunk) from its wave's current work queue (see "work queues" below), runs that pixel's samples
// in index order, writes the partial sum and pulls the next item.  A lane is always in exactly one of four states --
// at a BVH node, at a primitive, at a hit to be shaded (path vertex), or at a path restart (path
// ended or ray missed: add the sample, new camera ray, possibly a new work item) -- and each trip
// round the wave's loop executes ONE kind of step for the lanes
// that are in that state, chosen by ballot counts.  Box tests, primitive tests and shading
// are therefore each executed by a well-filled wave although the 64 paths are at different
// depths of different trees (the reference's 1.74 rays/sample x 55 node visits/ray vary by
// two orders of magnitude from ray to ray).
enum { M_NODE = 0, M_PRIM = 1, M_SHADE = 2, M_EXIT = 3, M_HIT = 4 };

// SINGLE: the world list is one tree (the usual case, main.cpp:146) -- known at compile time, the per-visit
// "next root of the world list?" test disappears from the node and primitive steps.
//
// LDSTREE (FAITHFUL, a scene whose whole node array fits): ONE workgroup of 1024 threads per CU keeps the node
// records in LDS -- 160 KB per CU is the one memory on this chip that takes a wave's 64 scattered 32-byte reads
// without going through the vector L1, whose address unit is the busiest unit of the 256-thread kernel (78 % of
// the cycles on the headline frame; a divergent dwordx4 wave-load costs it 16 + 0.45 x lines cycles,
// tools/ubench_tcp.hip, against ~11 cycles for the same two reads from LDS, tools/ubench_lds.hip).
// The LDS copy is THREADED (round 3): bvh.h:102-103 visits left, then right, always -- a fixed order needs no stack.
// A record's first link is its left child (node INDEX, so a visit's address is cur << 5, or primitive reference),
// its second link packs two 16-bit references (DevScene::nodeThread, computed at upload): high half = where the
// traversal goes when this node's subtree is done (its sibling's subtree, or an ancestor's, or "done"), low half =
// what follows a leaf's first object (its second object, or the high half again).  A miss takes the high half, a hit
// the left child; a lane at a primitive carries the pair and shifts it.  No per-lane stack in LDS at all: the 28 KB
// the headline scene's 16-bit stacks took are free, and a visit is two ds_read_b128 and nothing else.
// The attenuation stack lives in global memory (three coalesced stores per bounce) unless the tree leaves room
// (ATTLDS).  Same records, same arithmetic, same decisions as the stack walk.
template <bool CLOSEST, bool COUNT, bool SINGLE, bool LDSTREE, bool ATTLDS>
__global__ __launch_bounds__(LDSTREE ? SRT_BLOCK_TREE : SRT_BLOCK, LDSTREE ? SRT_TREE_WAVES_PER_SIMD : SRT_RENDER_WAVES_PER_SIMD) void srt_render_kernel(
    const RenderArgs a) {
  static_assert(LDSTREE || !ATTLDS, "ATTLDS qualifies the LDS-resident-tree kernel");
  static_assert(!(CLOSEST && LDSTREE), "the LDS-resident tree serves the FAITHFUL traversal");
  constexpr int BLOCK = LDSTREE ? SRT_BLOCK_TREE : SRT_BLOCK;  // threads per workgroup = stride of the [slot][thread] arrays
  // "no reference": what popping the empty stack yields.  LDSTREE: the 16-bit sentinel, sign-extended.
  constexpr int32_t DONE = LDSTREE ? (int32_t)0xFFFF8000 : SRT_REF_DONE;
  typedef typename std::conditional<LDSTREE, int16_t, int32_t>::type StackT;
  extern __shared__ int32_t lds[];
  // per-thread LDS slots, [slot][thread]: stackDepth+2 traversal slots (slot 0 holds a sentinel, the last
  // one is a spare for the node step's unconditional store), then 3*maxBounce attenuation floats and 3
  // floats of terminal radiance.  LDSTREE: node records first, then the (16-bit) traversal slots.
  char* const ldsTree = reinterpret_cast<char*>(lds);
  const int treeBytes = LDSTREE ? a.scene.numNodes * 32 : 0;
  // LDSTREE: no stack (threaded tree); the slot arithmetic below then counts zero stack slots
  const int stackSlots = LDSTREE ? 0 : a.scene.stackDepth + 2;
  StackT* const stackBase = reinterpret_cast<StackT*>(ldsTree + treeBytes) + threadIdx.x;
  if (!LDSTREE) *stackBase = (StackT)DONE;  // popping the empty stack yields "done"; nothing ever stores to slot 0 again
  // attenuation slots: behind the traversal slots (256-thread kernel), behind the 16 queue words (LDSTREE with room,
  // ATTLDS), or in global memory (LDSTREE with a tree that leaves no room)
  // (256-thread kernel with RenderArgs::attScratch set: in global memory too -- the closest-hit traversal of a scene far
  // larger than the caches is bound by requests in flight, and 15 KB less LDS per workgroup is one more workgroup per CU)
  const bool attGlobal = LDSTREE ? !ATTLDS : a.attScratch != nullptr;
  float* attStack = attGlobal ? a.attScratch + (size_t)blockIdx.x * BLOCK + threadIdx.x
                    : !LDSTREE ? reinterpret_cast<float*>(lds + stackSlots * BLOCK + threadIdx.x)
                               : reinterpret_cast<float*>(ldsTree + treeBytes + 64) + threadIdx.x;
  const int attStride = attGlobal ? (int)gridDim.x * BLOCK : BLOCK;
  const int attSlots = attGlobal ? 0 : 3 * a.maxBounce + 3;  // LDS slots per thread behind the traversal stack
  const int lane = threadIdx.x & 63;
  const uint64_t seedMixed = mix64(a.seed);
  const V3 background = ld3(a.background);
  const DevScene& sc = a.scene;
  const __amdgpu_buffer_rsrc_t rsNodes = makeRsrc(sc.nodes, sc.numNodes * 32);
  const __amdgpu_buffer_rsrc_t rsNodes2 = makeRsrc(sc.nodes2, CLOSEST ? sc.numNodes * 64 : 0);
  const bool wide = CLOSEST && sc.nodes4 != nullptr;  // closest-hit traversal over the 128-byte four-box records
  const __amdgpu_buffer_rsrc_t rsNodes4 = makeRsrc(sc.nodes4, wide ? (int)((unsigned)sc.numNodes * 128u) : 0);
  const __amdgpu_buffer_rsrc_t rsTris = makeRsrc(sc.triTest, sc.numTris * 48);
  const __amdgpu_buffer_rsrc_t rsSpheres = makeRsrc(sc.spheres, sc.numSpheres * 48);
  const __amdgpu_buffer_rsrc_t rsTexels = makeRsrc(sc.texels, sc.texelBytes);
  const bool singleRoot = SINGLE || sc.numWorld == 1;
  if (LDSTREE) {
    // node records into LDS, child references of nodes as indices (a 16-bit stack slot holds them; byte offset = index << 5)
    float4* dst = reinterpret_cast<float4*>(ldsTree);
    for (int i = threadIdx.x; i < sc.numNodes * 2; i += BLOCK) {
      float4 v = bufLoad4(rsNodes, 16 * i);
      const int r = __float_as_int(v.w);
      if (i & 1)
        v.w = __int_as_float(sc.nodeThread[i >> 1]);  // second link: (after a leaf's first object) | (after this subtree) << 16
      else if (r >= 0)
        v.w = __int_as_float(r >> 5);
      dst[i] = v;
    }
    __syncthreads();
  }
  // a reference as the world list holds it -> as this kernel's lanes hold it
  auto localRef = [&](int r) { return LDSTREE && r >= 0 ? r >> 5 : r; };

  unsigned long long cSamples = 0, cRays = 0, cNodes = 0, cBox = 0, cTri = 0, cSph = 0, cShTri = 0, cTex = 0;
  // scheduler profile (COUNT variant only; wave-uniform)
  unsigned long long pCyc[3] = {0, 0, 0}, pSteps[3] = {0, 0, 0}, pLanes[3] = {0, 0, 0};
  // sub-step profile of the two shading kinds (COUNT variant; wave-uniform clocks): hit step = record /
  // textures / direction draw / BRDF + bookkeeping; restart step; executions and lanes of each kind
  unsigned long long pSub[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pStart = COUNT ? clock64() : 0;

  // ---- work queues.  The local tiles (in tile order) are cut into units of unitTiles tiles, dealt
  // round-robin to numQueues queues with one counter each, so that the waves of a workgroup (home queue
  // blockIdx % numQueues) stay on a few neighbouring tiles: their rays touch the same part of the scene,
  // and since workgroups are dealt to the XCDs round-robin (XCD = blockIdx % 8) each L2 sees a few patches
  // of the image instead of all tiles in flight.
  // A wave that finds its queue drained moves to the queue with the most items left (steals), preferring
  // the queues of its own XCD; when none has any left its idle lanes leave.  The wave's current queue
  // lives in one LDS word (-1: everything drained) so that every lane sees it whichever lanes pulled last.
  int32_t* waveQueue = LDSTREE ? reinterpret_cast<int32_t*>(ldsTree + treeBytes) + (threadIdx.x >> 6)
                               : lds + (stackSlots + attSlots) * BLOCK + (threadIdx.x >> 6);
  const int qHome = (int)(blockIdx.x % (unsigned)a.numQueues);
  if (lane == 0) *waveQueue = qHome;
  const int unitItems = a.unitTiles * a.sppChunks * SRT_TILE_PIXELS;
  auto queueEnd = [&](int q) { return (q < a.numUnits ? (a.numUnits - q + a.numQueues - 1) / a.numQueues : 0) * unitItems; };

  // ---- lane state
  // A lane's state is read off its traversal registers: at a node (cur >= 0), at a primitive (cur < 0 but
  // not DONE), traversal finished with a hit to shade (cur == DONE, hitRef set), or waiting for a restart
  // (both DONE).  Nothing is recomputed per node visit to keep track of it.
  bool alive = true;
  int pend = 0;          // path end waiting to be added at the restart step: 1 = miss (background), 2 = terminal in LDS
  int s = 0, sEnd = 0;   // samples [s, sEnd) of the current work item remain
  int sCount = 0, outIndex = -1;
  int px = 0, py = 0;
  uint32_t pixel = 0;
  V3 acc = mk(0.0f, 0.0f, 0.0f);
  Ray ray;
  ray.o = ray.d = mk(0.0f, 0.0f, 0.0f);
  ray.time = 0.0f;
  Pcg rng;
  rng.state = 0;
  int depth = 0;
  // traversal state (hittableList::hit over the world list + bvhNode::hit as a DFS; see traverse())
  int cur = DONE, w = 0, hitRef = DONE;
  StackT* sptr = stackBase;  // top of this lane's stack of pending references (slot 0 = sentinel); unused by LDSTREE
  // LDSTREE (threaded tree): the two references that follow the current primitive, 16 bits each (low half first)
  constexpr int32_t DONE_PAIR = (int32_t)0x80008000;
  int32_t link = DONE_PAIR;
  auto atNode = [&]() { return cur >= 0; };
  auto atPrim = [&]() { return (uint32_t)cur > (uint32_t)DONE; };
  auto atHit = [&]() { return cur == DONE && hitRef != DONE; };
  auto atRestart = [&]() { return cur == DONE && hitRef == DONE && alive; };
  float closest = SRT_INF, rayA = 0.0f;
  // slab test per ray (boxHitApprox): refined reciprocals of ray.d, m = -o * rcpD, absolute tolerance
  V3 rcpD = mk(0.0f, 0.0f, 0.0f), negOR = mk(0.0f, 0.0f, 0.0f);
  float slabTol = SRT_INF;
  int dirNeg = 0;  // CLOSEST: sign bits of ray.d, for the near-child-first order

  // next pending reference after the current subtree is done; ends the traversal when none is left.
  // Written with selects rather than nested branches: every divergent `if` costs the wave half a
  // dozen scalar exec-mask instructions, and the scalar unit is shared by the CU's four SIMDs.
  uint32_t aovN0 = 0, aovB0 = 0, aovT0 = 0, aovS0 = 0;  // COUNT: counters at the start of the current ray
  // srtRenderAov: what this kernel's own traversal made of the ray at bounce aovDepth of the first sample
  auto writeAov = [&](int ref) {
    SrtAovRecord r;
    r.o[0] = ray.o.x; r.o[1] = ray.o.y; r.o[2] = ray.o.z;
    r.d[0] = ray.d.x; r.d[1] = ray.d.y; r.d[2] = ray.d.z;
    r.time = ray.time;
    r.valid = 1;
    r.prim = SRT_NO_HIT;
    r.t = 0.0f;
    if (ref != DONE) {
      const int pr = ~ref;
      r.prim = (pr & 1) ? sc.sphPrimId[pr >> 1] : sc.triPrimId[pr >> 1];
      r.t = closest;
    }
    r.nodeVisits = (int32_t)((uint32_t)cNodes - aovN0);
    r.boxPasses = (int32_t)((uint32_t)cBox - aovB0);
    r.triTests = (int32_t)((uint32_t)cTri - aovT0);
    r.sphereTests = (int32_t)((uint32_t)cSph - aovS0);
    r.pad[0] = r.pad[1] = 0;
    a.aov[pixel] = r;
  };
  auto popNext = [&](int next) {  // next = *sptr, read early by the caller: the sentinel when nothing is pending
    if (LDSTREE) {  // threaded tree: the low half of the pair is next, the high half follows it
      next = (int32_t)(int16_t)link;
      link >>= 16;
    } else {
      sptr -= BLOCK;
    }
    if (!SINGLE && !singleRoot && next == DONE && ++w < sc.numWorld) {  // once per ray: next root of the world list
      next = localRef(sc.world[w]);
      if (CLOSEST && next >= 0) next <<= (wide ? 2 : 1);
      sptr = stackBase;
      link = DONE_PAIR;
    }
    cur = next;
  };
  // world.hit(r, 0.001, infinity, rec): start the traversal of the world list
  auto startTraversal = [&]() {
    if (COUNT) {
      cRays++;
      aovN0 = (uint32_t)cNodes;
      aovB0 = (uint32_t)cBox;
      aovT0 = (uint32_t)cTri;
      aovS0 = (uint32_t)cSph;
    }
    rayA = lenSq(ray.d);  // sphere.h:56
    const bool certified = (sc.fastDivScene != 0) & fastDivOperandOk(ray.o.x, ray.d.x) & fastDivOperandOk(ray.o.y, ray.d.y) &
                           fastDivOperandOk(ray.o.z, ray.d.z);
    rcpD = mk(refinedRcp(ray.d.x), refinedRcp(ray.d.y), refinedRcp(ray.d.z));
    slabSetup(ray.o, rcpD, certified, negOR, slabTol);
    if (CLOSEST) dirNeg = (ray.d.x < 0.0f ? 1 : 0) | (ray.d.y < 0.0f ? 2 : 0) | (ray.d.z < 0.0f ? 4 : 0);
    closest = SRT_INF;
    hitRef = DONE;
    sptr = stackBase;
    link = DONE_PAIR;
    w = 0;
    cur = localRef(sc.world[0]);
    if (CLOSEST && cur >= 0) cur <<= (wide ? 2 : 1);  // node references of the closest-hit traversal address the 64- / 128-byte records
    pend = 1;  // a miss unless a hit-shading step says otherwise
  };

  // Two loop levels: the traversal steps run in the inner one, where the lanes' ray (origin, direction, reciprocals, ...) is
  // loop-invariant; the shading steps, which replace it, in the outer one.  With all four step kinds in ONE loop the
  // compiler copied that state -- some forty registers, twice -- on every trip back to the loop header.
  for (;;) {
    int pick, pk, nS = 0, nH = 0;
    unsigned long long pT0 = 0;
    for (;;) {
    const unsigned long long mN = __ballot(atNode()), mP = __ballot(atPrim()), mS = __ballot(atRestart()),
                             mH = __ballot(atHit());
    const int nN = __popcll(mN), nP = __popcll(mP);
    nS = __popcll(mS);
    nH = __popcll(mH);
    if ((nN | nP | nS | nH) == 0) {
      pick = -1;
      break;
    }
    if (nH >= a.hitMin || (nN | nP | nS) == 0)
      pick = M_HIT;
    else if (nS >= a.shadeMin || (nN | nP) == 0)
      pick = M_SHADE;
    else if (nP >= a.primMin || nN == 0)
      pick = M_PRIM;
    else
      pick = M_NODE;
    pick = __builtin_amdgcn_readfirstlane(pick);
    pT0 = COUNT ? clock64() : 0;
    pk = pick == M_HIT ? 2 : pick;  // profile slot: hit shading and restarts share the "shade" row
    if (COUNT && pick != M_NODE) {
      pSteps[pk]++;
      pLanes[pk] += pick == M_PRIM ? nP : (pick == M_HIT ? nH : nS);
    }

    if (pick == M_HIT || pick == M_SHADE) break;  // -> the outer level
    int nNodes = nN;
    if (pick == M_PRIM) {
      // ------------------------------------------------ sphere::hit / triangle::hit
      // A two-object leaf leaves its second primitive pending right behind the first (bvh.h:70-78): when enough
      // lanes are at a primitive again after the first test, test those in the same trip.
      for (int round = 0; round < SRT_PRIM_ROUNDS; ++round) {
        if (round > 0) {
          const int again = __popcll(__ballot(atPrim()));
          if (again < a.primAgainMin) break;
          if (COUNT) {
            pSteps[M_PRIM]++;
            pLanes[M_PRIM] += again;
          }
        }
        if (atPrim()) {
          const int pending = LDSTREE ? 0 : (int)*sptr;  // for popNext, read while the primitive's record is on its way
          int pr = ~cur;
          float t;
          bool ok;
          if (pr & 1) {
            if (COUNT) cSph++;
            const int off = (pr >> 1) * 48;
            float4 s0 = bufLoad4(rsSpheres, off), s1 = bufLoad4(rsSpheres, off + 16);
            V3 center = mk(s0.x, s0.y, s0.z);
            if (__float_as_int(s1.w) & (1 << 30)) {  // sphere.h:47-52
              float4 s2 = bufLoad4(rsSpheres, off + 32);
              center = center + ((ray.time - s2.x) / (s2.y - s2.x)) * (mk(s1.x, s1.y, s1.z) - center);
            }
            ok = sphereHitV(center, s0.w, ray, rayA, a.tMin, closest, t);
          } else {
            if (COUNT) cTri++;
            const int off = (pr >> 1) * 48;
            ok = triHitV<CLOSEST>(bufLoad4(rsTris, off), bufLoad4(rsTris, off + 16), bufLoad4(rsTris, off + 32), ray,
                                  a.tMin, closest, t);
          }
          if (ok) {
            closest = t;
            hitRef = cur;
          }
          popNext(pending);
        }
      }
      // most of these lanes are back at nodes now: go on with a node burst in the same trip instead of
      // paying for another scheduling decision (scalar work) first
      nNodes = __popcll(__ballot(atNode()));
      if (nNodes >= a.fuseMin) {
        if (COUNT) {
          const unsigned long long now = clock64();
          pCyc[pk] += now - pT0;
          pT0 = now;
          pk = M_NODE;
        }
        pick = M_NODE;
      }
    }

    if (pick == M_NODE) {
      // ------------------------------------------------ bvhNode::hit, bvh.h:97-105
      // several visits per scheduling decision while most of the node lanes are still at nodes
      const int keep = (nNodes * a.keepEighths) >> 3;  // the burst goes on while this many lanes are still at nodes
      int budget = a.nodeBurst;
      // SRT_NODE_UNROLL visits per loop trip: the "enough lanes left at nodes?" test is scalar work, and the scalar unit is
      // shared by the CU's four SIMDs
      constexpr int UNROLL = CLOSEST ? SRT_NODE_UNROLL_CLOSEST : SRT_NODE_UNROLL;
      auto nodeVisit = [&]() {
          if (COUNT) {
            pSteps[M_NODE]++;
            pLanes[M_NODE] += __popcll(__ballot(atNode()));
          }
          if (CLOSEST && wide) {
            // closest-hit traversal over the 128-byte records (DevScene::nodes4: the boxes of a node's four grandchildren):
            // one full line per visit, four boxes tested, the nearest entered, the other hits left pending; conservative
            // one-FMA intervals as below (a box is entered unless it is certainly missed)
            if (atNode()) {
              float4 lo[4], hi[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                lo[k] = bufLoad4(rsNodes4, cur + 16 * k);
                hi[k] = bufLoad4(rsNodes4, cur + 64 + 16 * k);
              }
              const int top = *sptr;
              float t[4];
              bool h[4];
              int ref[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                ref[k] = __float_as_int(lo[k].w);
                h[k] = boxMaybeHit(lo[k], hi[k], rcpD, negOR, slabTol, a.tMin, closest, t[k]);
              }
              if (slabTol == SRT_INF) {  // a ray outside the certified operand ranges: the reference's own test (see below)
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = boxHit(lo[k], hi[k], ray, a.tMin, closest);
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                h[k] = h[k] && ref[k] != SRT_REF_DONE;
                t[k] = h[k] ? t[k] : SRT_INF;
                if (COUNT) {
                  cNodes += ref[k] != SRT_REF_DONE ? 1 : 0;  // boxes tested, 32 bytes each
                  cBox += h[k] ? 1 : 0;
                }
              }
              // the nearest hit is entered; the others go on the stack (slot order)
              const int n01 = t[1] < t[0] ? 1 : 0, n23 = t[3] < t[2] ? 3 : 2;
              const float t01 = hwMin(t[0], t[1]), t23 = hwMin(t[2], t[3]);
              const int nearK = t23 < t01 ? n23 : n01;
              const bool any = h[0] || h[1] || h[2] || h[3];
              int nearRef = ref[0];
              nearRef = nearK == 1 ? ref[1] : nearRef;
              nearRef = nearK == 2 ? ref[2] : nearRef;
              nearRef = nearK == 3 ? ref[3] : nearRef;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                sptr[BLOCK] = (StackT)ref[k];  // the slot above the top is free; live only if sptr is bumped
                int move = (h[k] && nearK != k) ? 1 : 0;
                asm("" : "+v"(move));
                sptr += move * BLOCK;
              }
              if (!any) sptr -= BLOCK;
              cur = any ? nearRef : top;
              if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {
                cur = sc.world[w] >= 0 ? sc.world[w] << 2 : sc.world[w];
                sptr = stackBase;
              }
            }
            return;
          }
          if (CLOSEST) {
            // closest-hit traversal over the 64-byte records (both children's boxes per node, DevScene::nodes2):
            // one 64-byte request per visit, two boxes tested, the nearer child entered first, no exact fallback
            if (atNode()) {
              const float4 l0 = bufLoad4(rsNodes2, cur), l1 = bufLoad4(rsNodes2, cur + 16), r0 = bufLoad4(rsNodes2, cur + 32),
                           r1 = bufLoad4(rsNodes2, cur + 48);
              const int top = *sptr;
              if (COUNT) cNodes += 2;  // two boxes, 2 x 32 bytes, per record
              const int left = __float_as_int(l0.w), right = __float_as_int(l1.w);
              float tl, tr;
              bool hl = boxMaybeHit(l0, l1, rcpD, negOR, slabTol, a.tMin, closest, tl);
              bool hr = boxMaybeHit(r0, r1, rcpD, negOR, slabTol, a.tMin, closest, tr);
              if (slabTol == SRT_INF) {
                // a ray outside the certified operand ranges (a zero or tiny direction component ...): "cannot rule
                // the box out" would let it walk the whole tree -- on a 10 M-triangle scene a handful of such rays
                // then take longer than all the others together -- so these take the reference's own test
                hl = boxHit(l0, l1, ray, a.tMin, closest);
                hr = boxHit(r0, r1, ray, a.tMin, closest);
              }
              hr = hr && right != left;
              if (COUNT) cBox += (hl ? 1 : 0) + (hr ? 1 : 0);
              const bool both = hl && hr, leftFirst = !hr || (hl && !(tr < tl));
              const int nearRef = leftFirst ? left : right, farRef = leftFirst ? right : left;
              sptr[BLOCK] = (StackT)farRef;  // the slot above the top is free; live only if sptr is bumped
              int move = (hl || hr) ? (both ? 1 : 0) : -1;
              asm("" : "+v"(move));
              sptr += move * BLOCK;
              cur = (hl || hr) ? nearRef : top;
              if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {
                cur = sc.world[w] >= 0 ? sc.world[w] << 1 : sc.world[w];
                sptr = stackBase;
              }
            }
            return;
          }
          if (LDSTREE) {
            // threaded tree out of LDS: two reads, the box test, two selects
            if (atNode()) {
              const float4 n0 = *reinterpret_cast<const float4*>(ldsTree + (cur << 5));
              const float4 n1 = *reinterpret_cast<const float4*>(ldsTree + (cur << 5) + 16);
              if (COUNT) cNodes++;
              bool undecided;
              bool hitBox = boxHitApprox<true>(n0, n1, rcpD, negOR, slabTol, a.tMin, closest, undecided);
              if (undecided) hitBox = boxHit(n0, n1, ray, a.tMin, closest);
              if (COUNT && hitBox) cBox++;
              // hit: the left child (a leaf's first object then carries the pair: second object | what follows the
              // subtree); miss: what follows the subtree.  A lane at a node never needs its pair, so it is overwritten
              // unconditionally.
              link = __float_as_int(n1.w);
              cur = hitBox ? __float_as_int(n0.w) : (link >> 16);
              if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {  // once per ray: next root of the world list
                cur = localRef(sc.world[w]);
                link = DONE_PAIR;
              }
            }
            return;
          }
          if (atNode()) {
            const float4 n0 = bufLoad4(rsNodes, cur), n1 = bufLoad4(rsNodes, cur + 16);  // node references are byte offsets
            const int axis = CLOSEST ? sc.nodeAxis[cur >> 5] : 3;  // issued with the node record, used after the box test
            const int top = *sptr;  // pending reference, or the sentinel: read while the node record is on its way
            if (COUNT) cNodes++;
            // certified one-FMA test for every lane; the few visits it cannot decide, and every visit of a ray
            // outside fastDiv's operand ranges (slabTol = inf), take the IEEE divisions
            bool undecided;
            bool hitBox = boxHitApprox<true>(n0, n1, rcpD, negOR, slabTol, a.tMin, closest, undecided);
            if (undecided) hitBox = boxHit(n0, n1, ray, a.tMin, closest);
            if (COUNT && hitBox) cBox++;
            // descend left and leave right pending, or take the next pending reference (selects, see popNext)
            int left = __float_as_int(n0.w), right = __float_as_int(n1.w);
            if (CLOSEST) {
              // the closest hit does not depend on the visiting order: take the child on the ray's near side
              // first (left = lower side of the split axis) so that far subtrees get culled by `closest`
              // (dirNeg: bit k set when ray.d[k] < 0, bit 3 clear for "no usable axis")
              if ((dirNeg >> axis) & 1) {
                const int tmp = left;
                left = right;
                right = tmp;
              }
            }
            sptr[BLOCK] = (StackT)right;   // the slot above the top is free; it becomes live only if sptr is bumped
            // hit: descend left, right stays pending (a single-object leaf has left == right: nothing pending);
            // miss: take the pending reference.  The stack cannot overflow: its capacity (stackDepth slots + a
            // spare) is the tree's maximum number of pending references, computed or verified at upload.
            int move = hitBox ? (right != left ? 1 : 0) : -1;  // slots; selects of inline constants
            asm("" : "+v"(move));  // keep it in slots: folded into bytes it needs two literal moves per visit
            sptr += move * BLOCK;  // one shift-add
            cur = hitBox ? left : top;
            if (!SINGLE && !singleRoot && cur == DONE && ++w < sc.numWorld) {  // once per ray: next root of the world list
              cur = localRef(sc.world[w]);
              sptr = stackBase;
            }
          }
      };
      do {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) nodeVisit();
        budget -= UNROLL;
      } while (budget > 0 && __popcll(__ballot(atNode())) >= keep);
    }
    if (COUNT) pCyc[pk] += clock64() - pT0;
    }  // traversal steps
    if (pick < 0) break;
    bool go = false;  // this lane has a new ray to send into the world
    if (pick == M_HIT) {
      // ------------------------------------------------ rayColor's hit branch (main.cpp:42-51): one path vertex
      const unsigned long long h0 = COUNT ? clock64() : 0;
      unsigned long long hStamp[2] = {h0, h0}, h1 = h0;
      if (COUNT) {
        pSub[5]++;
        pSub[6] += nH;
      }
      if (atHit()) {
        if (COUNT && a.aov && depth == a.aovDepth && s == a.sampleFirst) writeAov(hitRef);
        Record rec;
        int pr = ~hitRef;
        if (pr & 1)
          sphereRecord(sc, pr >> 1, ray, closest, rec, false);
        else
          triRecord(sc, pr >> 1, ray, closest, rec, false);
        V3 att, emitted;
        Ray next;
        uint32_t fetches = 0;
        if (COUNT && rec.isTri) cShTri++;
        if (COUNT) h1 = clock64();
        bool scattered = shade<COUNT, LDSTREE>(sc, rsTexels, ray, rec, rng, att, next, emitted, fetches, COUNT ? hStamp : nullptr);
        if (COUNT) cTex += fetches;
        V3 terminal = emitted;  // main.cpp:46-47
        bool done = true;
        if (scattered) {
          // emitted is (0,0,0) for every scattering material (material.h:18-20)
          attStack[(3 * depth + 0) * attStride] = att.x;
          attStack[(3 * depth + 1) * attStride] = att.y;
          attStack[(3 * depth + 2) * attStride] = att.z;
          ray = next;
          depth++;
          done = depth >= a.maxBounce;  // main.cpp:36-37: out of bounces -> black
          terminal = mk(0.0f, 0.0f, 0.0f);
        }
        if (done) {
          // the path ends here: leave its terminal radiance for the restart step
          attStack[(3 * a.maxBounce + 0) * attStride] = terminal.x;
          attStack[(3 * a.maxBounce + 1) * attStride] = terminal.y;
          attStack[(3 * a.maxBounce + 2) * attStride] = terminal.z;
          pend = 2;
          hitRef = DONE;  // shaded: the lane now waits for a restart step
        } else {
          go = true;  // -> startTraversal() below, one copy of it for both shading steps
        }
      }
      if (COUNT) {
        // stamps are per lane; lanes that skipped a section keep the previous stamp: take the wave maximum
        unsigned long long t1 = h1, t2 = hStamp[0], t3 = hStamp[1];
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long o1 = __shfl_xor(t1, off), o2 = __shfl_xor(t2, off), o3 = __shfl_xor(t3, off);
          t1 = o1 > t1 ? o1 : t1;
          t2 = o2 > t2 ? o2 : t2;
          t3 = o3 > t3 ? o3 : t3;
        }
        const unsigned long long h4 = clock64();
        t2 = t2 < t1 ? t1 : t2;
        t3 = t3 < t2 ? t2 : t3;
        pSub[0] += t1 - h0;
        pSub[1] += t2 - t1;
        pSub[2] += t3 - t2;
        pSub[3] += h4 - t3;
      }
    } else {
      // ------------------------------------------------ path restart: miss / path end (main.cpp:39-40,49-51),
      // pixel sum (main.cpp:217), next work item, next camera ray (main.cpp:204-216)
      const unsigned long long r0 = COUNT ? clock64() : 0;
      if (COUNT) {
        pSub[7]++;
        pSub[8] += nS;
      }
      if (atRestart()) {
        if (pend != 0) {
          if (COUNT && a.aov && pend == 1 && depth == a.aovDepth && s == a.sampleFirst) writeAov(DONE);
          V3 L = background;  // main.cpp:39-40
          if (pend == 2)
            L = mk(attStack[(3 * a.maxBounce + 0) * attStride], attStack[(3 * a.maxBounce + 1) * attStride],
                   attStack[(3 * a.maxBounce + 2) * attStride]);
          pend = 0;
          // unwind the recursion: emitted + newColor * attenuation, innermost first (main.cpp:49-51)
          for (int j = depth - 1; j >= 0; --j) {
            float ax = attStack[(3 * j + 0) * attStride], ay = attStack[(3 * j + 1) * attStride],
                  az = attStack[(3 * j + 2) * attStride];
            L = mk(0.0f + L.x * ax, 0.0f + L.y * ay, 0.0f + L.z * az);
          }
          acc = acc + L;  // main.cpp:217
          s++;
        }
        if (s >= sEnd) {
          // work item finished (or none yet): write it, pull the next one with one atomic per wave
          if (outIndex >= 0) {
            if (a.fix)
              commitFixed(a.fix + outIndex, acc, a.fixLimit);
            else
              a.out[outIndex] = make_float4(acc.x, acc.y, acc.z, (float)sCount);
          }
          const unsigned long long mF = __ballot(1);
          const int leader = __ffsll((long long)mF) - 1;
          int base = 0;
          const int q = *waveQueue;  // wave-uniform
          bool gotItem = false;
          int idx = 0;
          if (q >= 0) {
            // queue q holds the units q, q + Q, q + 2Q, ... of the unit order (a unit = unitTiles consecutive
            // local tiles, all their chunks): every queue walks the whole image
            if (lane == leader) base = atomicAdd(a.queue + 16 * q, __popcll(mF));
            base = __shfl(base, leader);
            idx = base + __popcll(mF & ((1ull << lane) - 1ull));
            const int qEnd = queueEnd(q);
            gotItem = idx < qEnd;
            if (base + __popcll(mF) > qEnd) {
              // drained (wave-uniform): the leader looks at every queue's counter and the wave moves to the
              // fullest one, own XCD first.  Counters only grow, so "nothing left anywhere" is final.
              int nq = -1;
              if (lane == leader) {
                int bestLeft = 0;
                bool bestOwn = false;
                for (int k = 0; k < a.numQueues; ++k) {
                  const int left = queueEnd(k) - __hip_atomic_load(a.queue + 16 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  const bool own = ((k ^ qHome) & 7) == 0;
                  if (left > 0 && ((own && !bestOwn) || (own == bestOwn && left > bestLeft))) {
                    bestLeft = left;
                    bestOwn = own;
                    nq = k;
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
          outIndex = -1;
          if (!gotItem) {
            // nothing from this queue: an empty item; pulls again from the new queue on the next restart step
            s = sEnd = 0;
          } else {
            // idx -> (local tile, chunk, pixel of the tile); 64 consecutive items = one tile, one chunk.
            // No integer divisions here: a wave runs this branch whenever ANY of its restarting lanes has finished an
            // item (7 restart steps out of 10 on the headline frame), and the seven divisions of the plain
            // decomposition were ~270 of its ~350 VALU instructions.  The two quotients come from float reciprocals
            // computed on the host (off by at most one: group < 2^25, quotients < 2^20; corrected with the exact
            // integer remainder), the tile's place in the blocked order from a table (RenderArgs::tileXY).
            const int group = idx >> 6, ln = idx & 63;  // unitItems is a multiple of 64
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
            const uint32_t txy = a.tileXY[tile < a.numTiles ? tile : 0];  // srtTileFromOrder(tile): tx | ty << 16
            const int tx = (int)(txy & 0xffffu), ty = (int)(txy >> 16);
            px = tx * SRT_TILE_W + (ln & (SRT_TILE_W - 1));
            py = ty * SRT_TILE_H + (ln >> 3);
            pixel = (uint32_t)(py * a.imageWidth + px);
            // spp split: the first sppRem chunks get sppBase + 1 samples, the others sppBase
            const int s0 = a.sampleFirst + chunk * a.sppBase + min(chunk, a.sppRem);
            const int s1 = s0 + a.sppBase + (chunk < a.sppRem ? 1 : 0);
            sCount = s1 - s0;
            // maxBounce <= 0: rayColor returns black before tracing anything (main.cpp:36-37)
            const bool valid = localTile < a.numLocalTiles && tile < a.numTiles && px < a.imageWidth && py < a.imageHeight &&
                               a.maxBounce > 0;
            s = s0;
            sEnd = valid ? s1 : s0;
            acc = mk(0.0f, 0.0f, 0.0f);
            // slot of this item: [chunk][localTile][pixel] on the scratch path (chunkStride = numLocalTiles * 64),
            // [localTile][pixel] otherwise (chunkStride = 0)
            outIndex = localTile < a.numLocalTiles ? chunk * a.chunkStride + localTile * SRT_TILE_PIXELS + ln : -1;
          }
        }
        if (alive && s < sEnd) {
          rng.key(seedMixed, pixel, (uint32_t)s);
          float u = ((float)px + rng.uniform()) / (float)(a.imageWidth - 1);                      // main.cpp:210
          float v = ((float)(a.imageHeight - py) + rng.uniform()) / (float)(a.imageHeight - 1);  // main.cpp:211
          const DevCamera cam = cameraFromKernarg();
          cameraRay(cam, u, v, rng, ray);
          depth = 0;
          if (COUNT) cSamples++;
          go = true;
        }
        // else: exit, or an empty item (pixel outside the image): stays in M_SHADE and pulls again
      }
      if (COUNT) pSub[4] += clock64() - r0;
    }
    if (go) startTraversal();
    if (COUNT) pCyc[pk] += clock64() - pT0;
  }

  if (COUNT && a.stats) {
    unsigned long long v[8] = {cSamples, cRays, cNodes, cBox, cTri, cSph, cShTri, cTex};
    for (int k = 0; k < 8; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if (lane == 0) atomicAdd(&a.stats[k], x);
    }
    if (lane == 0) {
      for (int k = 0; k < 3; ++k) {
        atomicAdd(&a.stats[8 + k], pCyc[k]);
        atomicAdd(&a.stats[12 + k], pSteps[k]);
        atomicAdd(&a.stats[15 + k], pLanes[k]);
      }
      atomicAdd(&a.stats[11], (unsigned long long)(clock64() - pStart));
      for (int k = 0; k < 10; ++k) atomicAdd(&a.stats[18 + k], pSub[k]);
    }
  }
}

// atomic path: round the exact sums.  samples = what every pixel of the launch received (the items of a pixel
// partition its sample range, so the count needs no atomics).
__global__ void srt_finalize_kernel(const SrtFixedAccum* fix, float4* out, int n, int samples) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SrtFixedAccum f = fix[i];
  out[i] = make_float4(fromFixed36(f.r, f.flags, 0), fromFixed36(f.g, f.flags, 1), fromFixed36(f.b, f.flags, 2), (float)samples);
}

// scratch path: the same exact sum over the chunk slots buf[c][i]
__global__ void srt_sum_chunks_kernel(const float4* buf, float4* out, int n, int chunks, float limit) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long sum[3] = {0, 0, 0};
  uint32_t flags = 0;
  float count = 0.0f;
  for (int c = 0; c < chunks; ++c) {
    const float4 v = buf[(size_t)c * n + i];
    const float ch[3] = {v.x, v.y, v.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      long long q;
      if (toFixed36(ch[k], limit, q))
        sum[k] += q;
      else
        flags |= nonFiniteFlag(ch[k], k);
    }
    count += v.w;
  }
  out[i] = make_float4(fromFixed36(sum[0], flags, 0), fromFixed36(sum[1], flags, 1), fromFixed36(sum[2], flags, 2), count);
}

// =================================================================== resolve (color.h:25-41)
__global__ void srt_resolve_kernel(const ResolveArgs a) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.imageWidth * a.imageHeight) return;
  int x = idx % a.imageWidth, y = idx / a.imageWidth;
  int tile = srtOrderFromTile(x / SRT_TILE_W, y / SRT_TILE_H, a.tilesX, (a.imageHeight + SRT_TILE_H - 1) / SRT_TILE_H, a.tileBlock);
  int lane = (y % SRT_TILE_H) * SRT_TILE_W + (x % SRT_TILE_W);
  int rank = tile % a.tileStride, local = tile / a.tileStride;
  float4 v = a.gathered[((size_t)rank * a.numLocalTiles + local) * SRT_TILE_PIXELS + lane];
  if (a.accumImage) a.accumImage[idx] = v;
  if (a.rgba) {
    float scale = 1.0f / (float)a.spp;
    float c[3] = {v.x, v.y, v.z};
    uint8_t o[4];
    for (int k = 0; k < 3; ++k) {
      float g = sqrtf(c[k] * scale);
      float q = 256.0f * clampf(g, 0.0f, 0.999f);
      o[k] = (q == q) ? (uint8_t)q : (uint8_t)0;  // NaN -> 0 (what the reference's UB cast yields on x86)
    }
    o[3] = 255;
    reinterpret_cast<uchar4*>(a.rgba)[idx] = make_uchar4(o[0], o[1], o[2], o[3]);
  }
}

// =================================================================== fixed ray set
template <bool CLOSEST>
__global__ __launch_bounds__(SRT_BLOCK) void srt_trace_kernel(const TraceArgs a) {
  extern __shared__ int32_t lds[];
  int32_t* stack = lds + threadIdx.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    SrtRay in = a.rays[i];
    Ray r;
    r.o = ld3(in.o);
    r.d = ld3(in.d);
    r.time = in.time;
    Counters cnt = {0, 0, 0, 0};
    float tHit;
    int ref = traverse<CLOSEST, true>(a.scene, r, in.tMin, in.tMax, stack, tHit, cnt);
    SrtHit h;
    h.prim = SRT_NO_HIT;
    h.t = 0;
    for (int k = 0; k < 3; ++k) h.p[k] = h.normal[k] = h.tangent[k] = h.bitangent[k] = 0;
    h.uv[0] = h.uv[1] = 0;
    h.frontFace = 0;
    h.material = -1;
    if (ref != SRT_REF_DONE) {
      Record rec;
      int pr = ~ref;
      if (pr & 1) {
        sphereRecord(a.scene, pr >> 1, r, tHit, rec, true);
        h.prim = a.scene.sphPrimId[pr >> 1];
      } else {
        triRecord(a.scene, pr >> 1, r, tHit, rec, true);
        h.prim = a.scene.triPrimId[pr >> 1];
      }
      h.t = rec.t;
      h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
      h.normal[0] = rec.normal.x; h.normal[1] = rec.normal.y; h.normal[2] = rec.normal.z;
      h.tangent[0] = rec.tangent.x; h.tangent[1] = rec.tangent.y; h.tangent[2] = rec.tangent.z;
      h.bitangent[0] = rec.bitangent.x; h.bitangent[1] = rec.bitangent.y; h.bitangent[2] = rec.bitangent.z;
      h.uv[0] = rec.u; h.uv[1] = rec.v;
      h.frontFace = rec.frontFace ? 1 : 0;
      h.material = rec.material;
    }
    h.nodeVisits = (int)cnt.nodeVisits;
    h.boxPasses = (int)cnt.boxPasses;
    h.triTests = (int)cnt.triTests;
    h.sphereTests = (int)cnt.sphereTests;
    a.hits[i] = h;
  }
}

// material scatter known-answer kernel: one shade() per entry (tests only drive it
// through srtScatterTest; same device function the render kernel uses)
struct ScatterArgs {
  DevScene scene;
  const SrtRay* rays;
  const SrtHit* hits;
  float* out;  // 13 floats per entry
  uint64_t seed;
  int n;
};
__global__ void srt_scatter_kernel(const ScatterArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  Ray r;
  r.o = ld3(a.rays[i].o);
  r.d = ld3(a.rays[i].d);
  r.time = a.rays[i].time;
  const SrtHit& h = a.hits[i];
  Record rec;
  rec.p = ld3(h.p); rec.normal = ld3(h.normal); rec.tangent = ld3(h.tangent); rec.bitangent = ld3(h.bitangent);
  rec.u = h.uv[0]; rec.v = h.uv[1]; rec.t = h.t; rec.frontFace = h.frontFace != 0; rec.material = h.material;
  rec.matType = (int)a.scene.shadeRecs[8 * h.material].x;  // a caller-made hit record has no material word: type | SRT_MAT_TEXTURED from the record
  rec.isTri = false;
  Pcg rng;
  rng.key(mix64(a.seed), (uint32_t)i, 0u);
  V3 att = mk(0, 0, 0), em;
  Ray out;
  out.d = mk(0, 0, 0);
  uint32_t fetches = 0;
  bool ok = shade<false>(a.scene, makeRsrc(a.scene.texels, a.scene.texelBytes), r, rec, rng, att, out, em, fetches);
  float* o = a.out + 13 * i;
  o[0] = att.x; o[1] = att.y; o[2] = att.z;
  o[3] = out.d.x; o[4] = out.d.y; o[5] = out.d.z;
  o[6] = out.o.x; o[7] = out.o.y; o[8] = out.o.z;
  o[9] = ok ? 1.0f : 0.0f;
  o[10] = em.x; o[11] = em.y; o[12] = em.z;
}

// =================================================================== launch wrappers (host)
extern "C" {

namespace {
typedef void (*RenderKernel)(const RenderArgs);
// ldsTree: 0 = node records through the L1 (256 threads), 1 = LDS-resident tree (FAITHFUL, 1024 threads, attenuation
// stacks in global memory), 2 = the same with the attenuation stacks in LDS too
RenderKernel renderVariant(const RenderArgs* a, int traversal, int count, int ldsTree) {
  const bool closest = traversal == SRT_TRAVERSE_CLOSEST, single = a == nullptr || a->scene.numWorld == 1;
  if (ldsTree == 2 && !closest) {
    if (count) return srt_render_kernel<false, true, false, true, true>;
    return single ? srt_render_kernel<false, false, true, true, true> : srt_render_kernel<false, false, false, true, true>;
  }
  if (ldsTree && !closest) {
    if (count) return srt_render_kernel<false, true, false, true, false>;
    return single ? srt_render_kernel<false, false, true, true, false> : srt_render_kernel<false, false, false, true, false>;
  }
  if (count) return closest ? srt_render_kernel<true, true, false, false, false> : srt_render_kernel<false, true, false, false, false>;
  if (single) return closest ? srt_render_kernel<true, false, true, false, false> : srt_render_kernel<false, false, true, false, false>;
  return closest ? srt_render_kernel<true, false, false, false, false> : srt_render_kernel<false, false, false, false, false>;
}
}  // namespace

// ldsTree != 0: an LDS-resident-tree variant (FAITHFUL): workgroups of SRT_BLOCK_TREE threads, up to 160 KB of LDS each
int srt_launch_render(const RenderArgs* a, int traversal, int count, int ldsTree, int grid, size_t ldsBytes, hipStream_t stream) {
  const RenderKernel k = renderVariant(a, traversal, count, ldsTree);
  if (ldsBytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(ldsTree ? SRT_BLOCK_TREE : SRT_BLOCK), ldsBytes, stream, *a);
  return (int)hipGetLastError();
}

int srt_render_occupancy(int traversal, int count, int ldsTree, size_t ldsBytes, int* blocksPerCU) {
  const RenderKernel k = renderVariant(nullptr, traversal, count, ldsTree);
  if (ldsBytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k, ldsTree ? SRT_BLOCK_TREE : SRT_BLOCK, ldsBytes);
}

int srt_launch_finalize(const SrtFixedAccum* fix, float4* out, int n, int samples, hipStream_t stream) {
  hipLaunchKernelGGL(srt_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fix, out, n, samples);
  return (int)hipGetLastError();
}

int srt_launch_sum_chunks(const float4* buf, float4* out, int n, int chunks, float limit, hipStream_t stream) {
  hipLaunchKernelGGL(srt_sum_chunks_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, buf, out, n, chunks, limit);
  return (int)hipGetLastError();
}

int srt_launch_resolve(const ResolveArgs* a, hipStream_t stream) {
  int n = a->imageWidth * a->imageHeight;
  hipLaunchKernelGGL(srt_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *a);
  return (int)hipGetLastError();
}

int srt_launch_trace(const TraceArgs* a, int traversal, int grid, size_t ldsBytes, hipStream_t stream) {
  if (traversal == SRT_TRAVERSE_CLOSEST)
    hipLaunchKernelGGL(srt_trace_kernel<true>, dim3(grid), dim3(SRT_BLOCK), ldsBytes, stream, *a);
  else
    hipLaunchKernelGGL(srt_trace_kernel<false>, dim3(grid), dim3(SRT_BLOCK), ldsBytes, stream, *a);
  return (int)hipGetLastError();
}

int srt_launch_scatter(const DevScene* sc, const SrtRay* rays, const SrtHit* hits, float* out, uint64_t seed, int n,
                       hipStream_t stream) {
  ScatterArgs a{*sc, rays, hits, out, seed, n};
  hipLaunchKernelGGL(srt_scatter_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
