/*
 * srt_hip.h -- C ABI of the MI355X-native path-tracing hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  It sits where the reference keeps its (disabled) device seam
 *
 *     glDevice::init(w, h, std::vector<hittableIndexed>&)       gl.h:28, gl.h:194-267
 *     glDevice::rtFrame(void* frame, w, h, objects)             gl.h:29, gl.h:269-320
 *     glDevice::terminate()                                     gl.h:30
 *     hittableVector::build(list) / hittable::populateVector    hittablevector.h:27-31, hittable.h:32
 *
 * and replaces the body of the integrator loop main.cpp:200-227 (pixel/sample
 * loop -> camera::getRay -> rayColor -> writeColorTarget).
 *
 * Conventions (the reference's: bool returns + text on stderr, caller-owned
 * pixel buffers, main.cpp:182,239): every entry point returns int, 0 = ok,
 * non-zero = error with text available from srtLastError().  No exceptions
 * cross this boundary.  One context per GPU; calls on one context are not
 * thread-safe; different contexts may be driven from different host threads.
 */
#ifndef SRT_HIP_H
#define SRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ enums */
enum { SRT_PRIM_TRIANGLE = 0, SRT_PRIM_SPHERE = 1 };
/* material.h: pbrMetallicRoughness :23, metal :87, dielectric :104, diffuseLight :139 */
enum { SRT_MAT_PBR = 0, SRT_MAT_METAL = 1, SRT_MAT_DIELECTRIC = 2, SRT_MAT_LIGHT = 3 };
/* texture.h: solidColor :18, checker :34, imagePNG/image3bpp :109/:54 */
enum { SRT_TEX_SOLID = 0, SRT_TEX_CHECKER = 1, SRT_TEX_IMAGE = 2 };
enum { SRT_WORLD_PRIM = 0, SRT_WORLD_BVH = 1 };
/* who builds an SRT_WORLD_BVH item's tree when the caller supplies none:
 * REFERENCE = bvh.h:55-95 on the host (the tree FAITHFUL traversal semantics are defined on);
 * LBVH = a linear BVH built on the device (Morton order + Karras' hierarchy), PLOC = parallel
 * locally-ordered clustering on the device (surface-area driven, tighter trees, a few times the
 * LBVH's build time): both for SRT_TRAVERSE_CLOSEST rendering of large scenes. */
enum { SRT_BUILDER_REFERENCE = 0, SRT_BUILDER_LBVH = 1, SRT_BUILDER_PLOC = 2 };
/* traversal semantics: FAITHFUL = bvh.h:97-105 order with triangle::hit's
 * missing tMax test (model.h:128); CLOSEST adds the t < closest test. */
enum { SRT_TRAVERSE_FAITHFUL = 0, SRT_TRAVERSE_CLOSEST = 1 };

#define SRT_TILE_W 8
#define SRT_TILE_H 8
#define SRT_TILE_PIXELS 64
#define SRT_TILE_BLOCK 8 /* tiles are ordered in 8x8 blocks of tiles, see "Tiles" below */
#define SRT_MAX_BOUNCE 16
#define SRT_NO_HIT (-1)

/* ------------------------------------------------- scene description (in) */

/* One triangle with its vertex data already gathered through the mesh's u16
 * indices (model.h:108-111).  64 bytes. */
typedef struct SrtTriangleIn {
  float p[3][3];   /* positions  parentMesh->positions[vertices[i]]  */
  float uv[3][2];  /* texcoords  parentMesh->texcoords[vertices[i]]  */
  int32_t material;
} SrtTriangleIn;

/* sphere.h:11-15 ctor arguments. */
typedef struct SrtSphereIn {
  float center0[3];
  float center1[3];
  float time0, time1;
  float radius;
  int32_t material;
} SrtSphereIn;

/* One entry of a hittableList::objects vector (hittablelist.h:30), in list order. */
typedef struct SrtPrimRef {
  int32_t type;  /* SRT_PRIM_* */
  int32_t index; /* into triangles[] or spheres[] */
} SrtPrimRef;

/* One entry of the world list handed to rayColor (main.cpp:146,187):
 * a bare primitive, or a bvhNode over prims[first, first+count) with the
 * (time0,time1) given to its ctor (bvh.h:15-16).  nodes == NULL: the library
 * builds the tree (bvh.h:55-95, consuming the global generator); otherwise the
 * caller supplies an already built tree (e.g. from srtBuildBvh at bvhNode
 * construction time, or its own builder): numNodes records in pre-order, node 0
 * the root, child >= 0 a node index GREATER than its parent's, child < 0 a
 * primitive ~child inside [first, first+count). */
typedef struct SrtWorldItem {
  int32_t kind; /* SRT_WORLD_* */
  int32_t first;
  int32_t count;
  float time0, time1;
  int32_t numNodes;
  const struct SrtBvhNode* nodes;
  int32_t builder; /* SRT_BUILDER_* (ignored when nodes != NULL) */
  int32_t pad;
} SrtWorldItem;

/* material.h.  Field use per type:
 *   PBR        albedoTex/normalTex/metallicTex/roughnessTex (-1 = null ptr),
 *              albedo[4] = albedo factor, metalness, roughness
 *   METAL      albedo[0..2] = albedo, fuzz (already clamped to <=1 by the ctor, :89)
 *   DIELECTRIC ir
 *   LIGHT      albedoTex = emit texture                                        */
typedef struct SrtMaterialIn {
  int32_t type;
  int32_t albedoTex, normalTex, metallicTex, roughnessTex;
  float albedo[4];
  float metalness, roughness;
  float fuzz, ir;
  int32_t pad[3];
} SrtMaterialIn;

/* texture.h.  IMAGE: width*height*bpp bytes at texels + texelOffset, row
 * stride bpp*width (texture.h:122); width==0 means "failed to load" and
 * samples as (1,0,1) (texture.h:130-131).  CHECKER: even/odd are texture
 * ids of SOLID or IMAGE textures (texture.h:37-40). */
typedef struct SrtTextureIn {
  int32_t kind;
  int32_t width, height, bpp;
  int64_t texelOffset;
  int32_t even, odd;
  float color[3];
  int32_t pad;
} SrtTextureIn;

typedef struct SrtSceneDesc {
  int32_t numTriangles;
  const SrtTriangleIn* triangles;
  int32_t numSpheres;
  const SrtSphereIn* spheres;
  int32_t numPrims;
  const SrtPrimRef* prims; /* list order: decides BVH sort ties */
  int32_t numWorld;
  const SrtWorldItem* world;
  int32_t numMaterials;
  const SrtMaterialIn* materials;
  int32_t numTextures;
  const SrtTextureIn* textures;
  int64_t numTexelBytes;
  const uint8_t* texels;
} SrtSceneDesc;

/* camera.h:10-38 ctor arguments, and the members the ctor derives. */
typedef struct SrtCameraParams {
  float eye[3], lookAt[3], up[3];
  float vfovDegrees, aspect, aperture, focusDist, time0, time1;
} SrtCameraParams;

typedef struct SrtCamera {
  float origin[3], lleft[3], horizontal[3], vertical[3];
  float w[3], hor[3], vert[3];
  float lensRadius, time0, time1;
} SrtCamera;

/* ------------------------------------------------------ flattened BVH (out) */
/* 32 bytes: one node visit = two 16-byte loads.  child >= 0: node index in
 * the same array; child < 0: primitive, ~child = index into prims[]. */
typedef struct SrtBvhNode {
  float bmin[3];
  int32_t left;
  float bmax[3];
  int32_t right;
} SrtBvhNode;

/* ----------------------------------------------------------- fixed ray set */
typedef struct SrtRay {
  float o[3];
  float d[3];
  float time;
  float tMin, tMax;
} SrtRay;

/* hitRecord (hittable.h:9-22) plus what the reference never records: the
 * primitive id (index into prims[]) and traversal counters. */
typedef struct SrtHit {
  int32_t prim; /* SRT_NO_HIT on miss */
  float t;
  float p[3];
  float normal[3], tangent[3], bitangent[3];
  float uv[2];
  int32_t frontFace;
  int32_t material;
  int32_t nodeVisits, boxPasses, triTests, sphereTests;
} SrtHit;

/* ----------------------------------------------------------------- render */
typedef struct SrtRenderParams {
  int32_t imageWidth, imageHeight;
  int32_t spp;       /* numSamples, main.cpp:178 */
  int32_t maxBounce; /* main.cpp:180 */
  uint64_t seed;     /* counter RNG key; see DESIGN.md "RNG" */
  float background[3]; /* main.cpp:170 */
  float tMin;          /* 0.001f, main.cpp:39 */
  int32_t traversal;   /* SRT_TRAVERSE_* */
  /* multi-GPU tile split: this call renders tiles tileFirst, tileFirst+tileStride, ... */
  int32_t tileFirst, tileStride;
  /* One work item = one pixel x one chunk of its samples.  Samples are summed in index order inside a
   * chunk (a float running sum, main.cpp:217).  1 = a single running sum per pixel, the reference's order
   * (main.cpp:204-218), bit-reproducible against the oracle.  > 1: the chunks' float sums are added EXACTLY
   * (64-bit fixed point with 2^-36 resolution) and rounded to float once, so the pixel sum does not depend on
   * the order in which chunks finish, on the tile split or on the GPU count; it differs from the single
   * running sum only by the re-association of the float sum (<= 2e-5 relative), and a chunk sum of
   * 2^26 / (chunk count rounded up to a power of two) or more counts as infinite (the pixel is white either
   * way).  Device memory held by the context until srtDestroy: 16 bytes per pixel PER CHUNK of this rank's
   * tiles (one slot per work item, summed by a second kernel) while that fits a budget -- the smaller of
   * 12 GiB (tunable chunk_scratch_mb) and a quarter of the free device memory --, otherwise, or when that
   * allocation fails, 32 bytes per pixel whatever the chunk count (64-bit integer atomics, 0.4-1 % slower);
   * both give the same bits.  0 = library default, srtPlanSppChunks(width, height, spp, 0) (fastest). */
  int32_t sppChunks;
  int32_t countStats; /* 1: run the counting variant and fill srtGetStats() */
  /* progressive rendering: this call renders samples [sampleFirst, sampleFirst + spp) of every
   * pixel (RNG keys use the absolute sample index), so passes can be added up, checkpointed and
   * resumed; the reference only writes once at the very end (main.cpp:235-237). */
  int32_t sampleFirst;
} SrtRenderParams;

/* counters behind the algorithmic-bytes figure */
typedef struct SrtStats {
  uint64_t samples, rays;
  uint64_t nodeVisits, boxPasses, triTests, sphereTests;
  uint64_t shadedTriHits, texelFetches;
  /* wave-scheduler profile of the counting variant (diagnostics, not part of any parity claim):
   * shader clocks spent in, executions of, and lanes active in each step kind, summed over waves */
  uint64_t cyclesNode, cyclesPrim, cyclesShade, cyclesTotal;
  uint64_t stepsNode, stepsPrim, stepsShade;
  uint64_t lanesNode, lanesPrim, lanesShade;
} SrtStats;

typedef struct SrtContext SrtContext;

/* ------------------------------------------------------------ entry points */

/* replaces glDevice::init's context part (gl.h:194-215). */
int srtCreate(int deviceOrdinal, SrtContext** out);
/* replaces glDevice::terminate (gl.h:322-324). */
int srtDestroy(SrtContext* ctx);
const char* srtLastError(const SrtContext* ctx);

/* Host helper: camera ctor arithmetic, camera.h:10-38 (double tan, etc). */
int srtMakeCamera(const SrtCameraParams* in, SrtCamera* out);

/* The process-global default-seeded mt19937 the reference draws everything
 * from (globals.h:30-35).  The BVH builder consumes it exactly as
 * bvh.h:60 does; scene code may use it as randomFloat(). */
float srtHostRandomFloat(void);
void srtHostRandomReset(void);

/* replaces hittableVector::build + the SSBO upload (hittablevector.h:27-31,
 * gl.h:240-262): builds every SRT_WORLD_BVH with bvh.h:55-95 semantics,
 * precomputes per-triangle constants, copies everything to HBM.
 * Device memory: 32 B per node, 112 B per triangle, 48 B per sphere, the texels; trees too large for a compute
 * unit's LDS (more than about 4 500 nodes) get a second, threaded copy of the node array for the render kernel
 * that keeps their top in LDS: +32 B per node, +8 B per primitive. */
int srtUploadScene(SrtContext* ctx, const SrtSceneDesc* scene);
int srtSetCamera(SrtContext* ctx, const SrtCamera* cam);

/* Host-only (no GPU needed): the bvhNode build of world item `item`, exactly as
 * srtUploadScene performs it.  out may be NULL to query the node count. */
int srtBuildBvh(const SrtSceneDesc* scene, int32_t item, SrtBvhNode* out, int32_t capacity, int32_t* count,
                int32_t* stackDepth);

/* Flattened tree of world item `item` (host copy), for topology parity tests.
 * Call with nodes == NULL to get the count. */
int srtGetBvh(SrtContext* ctx, int32_t item, SrtBvhNode* nodes, int32_t capacity, int32_t* count);
int srtGetBvhDepth(SrtContext* ctx, int32_t* depth);

/* Tiles: the image is cut into 8x8-pixel tiles.  Tiles are numbered along a blocked curve: the tile
 * grid is cut into SRT_TILE_BLOCK x SRT_TILE_BLOCK blocks (edge blocks smaller), blocks row-major,
 * tiles row-major inside a block with row iy rotated by iy, so that tiles issued together form a
 * compact 2-D patch (coherent rays) and a rank's tiles do not line up in columns.  Positions of
 * this order are the unit of the multi-GPU split (rank r of N renders positions r, r+N, ...)
 * and of the output layout ([local position][64 pixels] float4); inside the kernel lanes pull
 * (pixel, sample chunk) work items.  srtResolveTiles undoes the order; callers never need it
 * (host mirror for tests: sexy-raytracer_amd/tiles.py).
 * srtNumLocalTiles gives the (rank-padded) tile count of one rank: ceil(numTiles / tileStride). */
int32_t srtNumTiles(int32_t imageWidth, int32_t imageHeight);
int32_t srtNumLocalTiles(int32_t imageWidth, int32_t imageHeight, int32_t tileStride);

int32_t srtDefaultSppChunks(int32_t spp);
/* The chunk count a render will use: sppChunks itself when > 0 (-1 if that many chunk slots over the whole image do
 * not fit the kernels' 32-bit work-item index: srtRenderTiles then fails), else srtDefaultSppChunks(spp), lowered only
 * for images beyond ~3 Mpixels.  A function of the image size and the sample count alone -- never of the tile split --
 * so that every rank of every split adds the same chunks.  Host only. */
int32_t srtPlanSppChunks(int32_t imageWidth, int32_t imageHeight, int32_t spp, int32_t sppChunks);

/* The hot path.  Asynchronous on `stream` (a hipStream_t, NULL = default).
 * dAccumTiles: DEVICE pointer, float4[numLocalTiles * 64], tile-major,
 * rgb = sum over samples of rayColor, a = sample count. */
int srtRenderTiles(SrtContext* ctx, const SrtRenderParams* p, void* dAccumTiles, void* stream);

/* color.h:25-41 (writeColorTarget) over a gathered, rank-major tile buffer
 * float4[tileStride][numLocalTiles*64]: un-permutes tiles into image order.
 * dRgba: DEVICE uint8[W*H*4] or NULL; dAccumImage: DEVICE float4[W*H] or NULL. */
int srtResolveTiles(SrtContext* ctx, const SrtRenderParams* p, const void* dGatheredTiles,
                    void* dRgba, void* dAccumImage, void* stream);

/* Multi-GPU: one process per GPU, the scene replicated, rank r of N renders tile positions
 * r, r+N, ... (SrtRenderParams.tileFirst / tileStride), and the path's only collective is ONE gather of the
 * ranks' equal-sized tile buffers to rank 0 over RCCL (ncclGather), after which rank 0 calls
 * srtResolveTiles.  The reference is single-device (gl.h:28-31); these entries extend its device seam.
 *   srtCommGetUniqueId  rank 0: 128 opaque bytes (an ncclUniqueId) to hand to the other ranks by the host
 *                       program's own means (file, MPI, a launcher's key-value store)
 *   srtCommInit         every rank: joins the communicator (collective: returns when all ranks have)
 *   srtGatherTiles      every rank: dLocalTiles = float4[numLocalTiles*64] (DEVICE); rank 0 also passes
 *                       dGathered = float4[numRanks][numLocalTiles*64] (DEVICE), others NULL.
 *                       Asynchronous on `stream`.  With one rank it degenerates to a device copy.
 *   srtRenderImageRanks the blocking main.cpp:182-227 form across the ranks (collective): render own tiles,
 *                       the one gather, rank 0 resolves into its caller-owned HOST buffers.  Before the gather the
 *                       ranks agree (a 4-byte all-reduce) that every one of them rendered: if one failed, ALL return
 *                       non-zero together instead of some waiting in the gather.  A rank that is lost altogether
 *                       leaves its peers in a collective: they then need srtCommDestroy (ncclCommAbort semantics).
 *   srtCommDestroy      also done by srtDestroy */
#define SRT_COMM_ID_BYTES 128
int srtCommGetUniqueId(void* id128);
int srtCommInit(SrtContext* ctx, const void* id128, int32_t numRanks, int32_t rank);
int srtGatherTiles(SrtContext* ctx, const SrtRenderParams* p, const void* dLocalTiles, void* dGathered, void* stream);
int srtRenderImageRanks(SrtContext* ctx, const SrtRenderParams* p, float* hAccum, uint8_t* hRgba);
int srtCommDestroy(SrtContext* ctx);

/* Blocking convenience with caller-owned HOST buffers: the main.cpp:182,224
 * `target` contract.  hAccum (float[W*H*4]) and hRgba (uint8[W*H*4]) may be NULL. */
int srtRenderImage(SrtContext* ctx, const SrtRenderParams* p, float* hAccum, uint8_t* hRgba);

/* Fixed-ray-set parity entry: world.hit(r, tMin, tMax, rec) for n rays. HOST pointers. */
int srtTraceRays(SrtContext* ctx, const SrtRay* rays, int64_t n, SrtHit* hits, int32_t traversal);

/* material::scatter + material::emitted (material.h:15-21; texture::value through a light's emitted, texture.h:13-16) for
 * n (incoming ray, hit record) pairs, evaluated by the render kernels' own shading function.  hits[i].material indexes
 * the uploaded scene's materials; p, normal, tangent, bitangent, uv, t, frontFace are read as the reference's hitRecord
 * fields.  Random draws come from the counter RNG keyed (seed, i, 0) -- the reference draws from its process-global
 * generator.  out13 per entry: attenuation[3], scattered direction[3], scattered origin[3], scatter's bool (0 / 1),
 * emitted[3].  HOST pointers; blocking. */
int srtScatterRays(SrtContext* ctx, const SrtRay* rays, const SrtHit* hits, int32_t n, uint64_t seed, float* out13);

/* Duration of the most recent srtRenderTiles kernel, from HIP events recorded
 * on its stream (synchronises on the stop event). */
int srtLastKernelMs(SrtContext* ctx, float* ms);
int srtGetStats(SrtContext* ctx, SrtStats* out);
/* device properties the bench prints next to its numbers */
int srtDeviceInfo(SrtContext* ctx, char* name, int32_t nameCap, int32_t* numCUs, int32_t* clockMHz);

#ifdef __cplusplus
}
#endif
#endif /* SRT_HIP_H */
