/*
 * srt_hip_test.h -- test hooks and diagnostics of libsrt_hip.so.
 *
 * NOT part of the drop-in boundary (include/srt_hip.h): nothing here is needed to render.  These entry
 * points let the parity tests reach device functions of the hot path in isolation (the kernel's own
 * shading function, the render kernel's own traversal per ray) and let the
 * measurement tools vary the work distribution.  HOST pointers throughout.
 */
#ifndef SRT_HIP_TEST_H
#define SRT_HIP_TEST_H

#include "srt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: material::scatter + emitted (material.h:15-21) through the kernel's own
 * shading function for n (ray, hit record) pairs; entry i draws from the counter RNG
 * keyed (seed, pixel=i, sample=0).  out13 per entry: attenuation[3], scattered dir[3],
 * scattered origin[3], scatter's bool, emitted[3].  HOST pointers. */
int srtScatterTest(SrtContext* ctx, const SrtRay* rays, const SrtHit* hits, int32_t n, uint64_t seed, float* out13);

/* Per-ray view into the RENDER kernel's own traversal (not srtTraceRays' kernel): renders sample
 * p->sampleFirst of every pixel with the counting variant of srt_render_kernel and records, per pixel, the
 * ray that path traced at bounce `depth` (0 = the camera ray) and what the kernel's node / primitive steps
 * made of it.  Feeding the recorded rays to the oracle's world.hit() checks the render kernel's slab-test
 * certificate, stack handling and step scheduler ray by ray.  hOut: W*H records in image order. */
typedef struct SrtAovRecord {
  float o[3], d[3], time; /* the ray (tMin = p->tMin, tMax = +inf, main.cpp:39) */
  int32_t valid;          /* 0: the pixel's path ended before this bounce */
  int32_t prim;           /* index into prims[], SRT_NO_HIT on a miss */
  float t;
  int32_t nodeVisits, boxPasses, triTests, sphereTests;
  int32_t pad[2];
} SrtAovRecord;
int srtRenderAov(SrtContext* ctx, const SrtRenderParams* p, int32_t depth, SrtAovRecord* hOut);

/* Sub-step profile of the most recent countStats launch (shader clocks summed over waves, diagnostics only):
 * out10 = { hit step: hit record, textures, direction draw, BRDF + bookkeeping; restart step;
 *           hit-step executions, hit-step lanes, restart-step executions, restart-step lanes, reserved }. */
int srtGetShadeProfile(SrtContext* ctx, uint64_t* out10);

/* Step profile of the most recent launch of the path-pool kernel's profiling variant (tunable "wf_profile" = 1;
 * csrc/srt_wavefront.hip): out46 = { clocks[12], executions[12], lanes[12] } for the step kinds node visit, primitive
 * test, swap, hit step of class 0 / 1 / 2, restart, idle, lost claim, item pull, visit of a node outside LDS (hybrid form:
 * the clocks are the wait for the records plus the visit), unused; scheduling clocks, total wave clocks; then what the
 * scheduling decisions that looked at the rings saw, summed: decisions, lanes at nodes / at primitives / finished / idle,
 * READY fill, fill of the fullest served ring, RESTART fill. */
int srtGetWfProfile(SrtContext* ctx, uint64_t* out46);

/* Host-only (no device): the thread links srtUploadScene builds for the stackless walks (csrc/srt_thread.h) from a
 * flattened node array -- nodes8 = numNodes x 8 floats, (bmin.xyz, left) (bmax.xyz, right), node references = index * 32,
 * primitives = ~(index << 1 | sphere) -- and the world list's references.
 * srtTestThreadLinks16: one word per node (DevScene::nodeThread) -> outLinks[numNodes]; returns 1, or 0 when the forest has
 * no 16-bit threaded form.
 * srtTestHybridRecords: the path-pool kernel's hybrid records (DevScene::nodesWf / worldWf / primSecond) with at most `cap`
 * resident nodes -> outNodes8[numNodes x 8], outWorld[numWorld], outSecond[2 * max(numTriangles, numSpheres) + 2]; returns
 * the number of resident nodes, or 0 when the forest has no threaded form.  Both return -1 on a null argument. */
int srtTestThreadLinks16(const float* nodes8, int32_t numNodes, const int32_t* world, int32_t numWorld, int32_t numTriangles, int32_t numSpheres,
                         int32_t* outLinks);
int srtTestHybridRecords(const float* nodes8, int32_t numNodes, const int32_t* world, int32_t numWorld, int32_t numTriangles, int32_t numSpheres,
                         int32_t cap, float* outNodes8, int32_t* outWorld, int32_t* outSecond);

/* The most recent render-kernel launch: out4 = { 0 node records through the L1, 1 the step-scheduler kernel over the
 * LDS-resident threaded tree (FAITHFUL, node array small enough for a CU's LDS; tunable "lds_tree" = 0 switches it
 * off), 2 the same with the attenuation stacks in LDS as well, 3 the path-pool kernel over the same tree (tunable
 * "wavefront" = 0 switches it off), 4 the path-pool kernel's hybrid form for a tree that does not fit (its top in LDS, the
 * rest read from global memory; tunable "wf_hybrid" = 0 switches it off, "wf_resident_max" caps the resident nodes -- both
 * are read by srtUploadScene); workgroups; threads per workgroup; LDS bytes per workgroup }. */
int srtGetLaunchInfo(SrtContext* ctx, int32_t* out4);

/* Per-context diagnostic tunables of the work distribution and the wave scheduler ("queues", "unit_tiles",
 * "tile_block", "shade_min", "prim_min", "hit_min", "fuse_min", "node_burst", "ploc_radius", "fast_div",
 * "prim_again_min", "keep_eighths", "lds_tree", "chunk_scratch_mb": memory budget of the chunk-slot path of the exact chunk sum, 0 forces the atomic path).  Defaults come from the library (and, for the scheduler thresholds, from SRT_*
 * environment variables read ONCE at srtCreate); none of them changes the image
 * (tests/test_gpu_properties.py::test_image_independent_of_work_distribution).  "tile_block" has no
 * environment override: every rank and the host untile (tiles.py) must agree on it. */
int srtSetTunable(SrtContext* ctx, const char* name, int32_t value);
int srtGetTunable(SrtContext* ctx, const char* name, int32_t* value);

#ifdef __cplusplus
}
#endif
#endif /* SRT_HIP_TEST_H */
