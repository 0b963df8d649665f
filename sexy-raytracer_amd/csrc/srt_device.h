// srt_device.h -- HBM-resident scene layout shared by the host uploader and the kernels.
//
// Every record is a whole number of 16-byte slots so one lane fetches it with
// dwordx4 loads; arrays are separate buffers (nodes / triangle test records /
// triangle shading records / spheres / materials / textures / texels) instead of
// the reference's 160-byte hittableIndexed mega-struct (hittableindexed.h:24-38).
#ifndef SRT_DEVICE_H
#define SRT_DEVICE_H

#include <stdint.h>

#include "../../include/srt_hip.h"
#include "../../include/srt_hip_test.h"

// Child / world reference encoding on the device:
//   ref >= 0          BVH node: the BYTE OFFSET of its record in `nodes` (index * 32), so that a visit needs no
//                     address arithmetic; SRT_NODE_REF / SRT_NODE_INDEX convert
//   ref <  0          primitive: r = ~ref, (r & 1) = 1 sphere / 0 triangle, r >> 1 = index
//   SRT_REF_DONE      traversal sentinel (never a valid primitive)
#define SRT_REF_DONE ((int32_t)0x80000000)
#define SRT_NODE_REF(index) ((int32_t)(index) << 5)
#define SRT_NODE_INDEX(ref) ((int32_t)(ref) >> 5)
#define SRT_MAX_NODES (1 << 25) /* byte offsets (index * 64 in the closest-hit records) stay below 2^31 */
#define SRT_MAX_QUEUES 64
// a primitive's material word: material index, the material's type (SRT_MAT_*) and SRT_MAT_NEEDS_* flags,
// (spheres) bit 30 = moving
#define SRT_MAT_INDEX_MASK 0x00ffffff
#define SRT_MAT_TYPE_SHIFT 24   /* four bits: SRT_MAT_* (two bits) | SRT_MAT_TEXTURED */
#define SRT_MAT_TEXTURED 4      /* a pbr material with at least one texture slot in use */
#define SRT_MAT_FLAGS_SHIFT 28

struct DevMaterial {  // 48 B
  int32_t type;
  int32_t albedoTex, normalTex, metallicTex, roughnessTex;
  float albedo[4];
  float metalness, roughness;  // METAL: metalness = fuzz; DIELECTRIC: metalness = ir
  int32_t flags;               // bit0: some texture of this material reads uv; bit1: it has a normal map
};

struct DevTexture {  // 48 B
  int32_t kind;
  int32_t width, height, bpp;
  int64_t offset;  // byte offset into texels (< 2^31); 3-byte images are stored 4 bytes per texel
  int32_t even, odd;
  float color[3];
  int32_t pad;
};

struct DevScene {
  // 2 x float4 per node: (bmin.xyz, left) (bmax.xyz, right), children as references (above)  -- 32 B / node visit
  const float4* nodes;
  // 3 x float4 per triangle: (v0.xyz, n.x) (v1.xyz, n.y) (v2.xyz, n.z), n = (v1-v0)x(v2-v0)  -- 48 B / test
  const float4* triTest;
  // 4 x float4 per triangle, read once per shaded hit:
  // (N^.xyz, uv0.u) (T.xyz, uv0.v) (B.xyz, uv1.u) (uv1.v, uv2.u, uv2.v, material)
  const float4* triShade;
  // 3 x float4 per sphere: (c0.xyz, radius) (c1.xyz, material | moving<<30) (t0, t1, -, -)   -- 16..48 B / test
  const float4* spheres;
  // SRT_TRAVERSE_CLOSEST only: one 64-byte record per node holding BOTH children's boxes,
  // (lmin.xyz, left) (lmax.xyz, right) (rmin.xyz, -) (rmax.xyz, -), node children as byte offsets into this array
  // (index * 64); built from `nodes` on the device after upload (srt_lbvh.hip, srt_pair_nodes).  One 64-byte request
  // per visit tests two boxes; the 32-byte records use half of every request they cause.
  const float4* nodes2;
  // SRT_TRAVERSE_CLOSEST, wide form: one 128-byte record per node with the boxes of its (up to) four grandchildren,
  // slot k: (min_k.xyz, reference_k) at +16k, (max_k.xyz, -) at +64 + 16k, node references = byte offsets into this array
  // (index * 128), SRT_REF_DONE = unused slot; built from nodes2 (srt_lbvh.hip wideNodes).  Null: walk nodes2.
  const float4* nodes4;
  int32_t wideStackDepth;  // pending references the wide walk can pile up: 3 per wide level
  // 1 byte per node: the axis its children are split on (left = lower side), 3 = unknown.  Read only by
  // SRT_TRAVERSE_CLOSEST, which visits the nearer child first; FAITHFUL keeps bvh.h's left-then-right.
  const uint8_t* nodeAxis;
  // The LDS-resident-tree kernel walks the tree without a stack (bvh.h:102-103 is a fixed left-then-right order): one
  // word per node, two 16-bit references -- low half: what follows a leaf's first object (its second object, or the high
  // half again for a single-object leaf); high half: where the traversal goes when this node's subtree is done (node
  // INDEX, primitive reference, or 0x8000 = done).  Null unless every tree was built on the host, every node's children
  // are two nodes or two primitives, and every reference fits 15 bits (srt_api.cpp threadTree).
  const int32_t* nodeThread;
  // path-pool kernel (srt_wavefront.hip): the material CLASS of every primitive, indexed by ~reference
  // (index << 1 | sphere): 0 triangle with a pbr material, 1 sphere with a pbr material that reads neither uv nor a normal
  // map, 2 anything else.  A hit goes to its class's ring, so a hit step runs one class's code.
  const uint8_t* primClass;
  int32_t numPrimClass;
  // path-pool kernel, hybrid form (a tree that does not fit into LDS): threaded records with 32-bit references --
  // (bmin.xyz, reference taken on a box hit) (bmax.xyz, successor << 2 | what follows a leaf's first object: 0 nothing,
  // 1 the next primitive of the same array, 2 primSecond[~first]) -- renumbered so that the wfResident nodes kept in LDS
  // come first (srt_api.cpp: the boxes of largest surface, closed upward), the world list's roots in that numbering, and
  // the second object of the leaves whose objects are not neighbours in one array (indexed like primClass).  "Done" is
  // -2^29 here.  Null when the tree fits (or is not a host-built tree of two-node / two-primitive nodes).
  const float4* nodesWf;
  const int32_t* worldWf;
  const int32_t* primSecond;
  int32_t wfResident;
  const int32_t* triPrimId;  // device index -> index into the scene's prims[] list
  const int32_t* sphPrimId;
  const int32_t* world;  // refs, world-list order
  int32_t numWorld;
  int32_t stackDepth;  // max pending right children over all trees
  int32_t numNodes, numTris, numSpheres;  // record counts (buffer-resource extents)
  int32_t fastDivScene;  // 1: every box coordinate is 0 or in [2^-77, 2^30] (see fastDiv in srt_kernels.hip)
  const DevMaterial* materials;
  // what the hit step reads: 128 bytes per material -- scalars, a light's emit texture, the four pbr texture slots
  // resolved to (mode, width, height, texel offset); layout at srt_kernels.hip "materials"
  const uint4* shadeRecs;
  int32_t numMaterials;
  const DevTexture* textures;
  const uint8_t* texels;  // images of >= 3 bytes per pixel as one dword per texel (RGBA8), others as byte rows
  int32_t texelBytes;     // extent of `texels` (buffer-resource bound: reads past it return 0)
};

struct DevCamera {
  float origin[3], lleft[3], horizontal[3], vertical[3], hor[3], vert[3];
  float lensRadius, time0, time1;
};

struct SrtFixedAccum {  // 32 B per pixel of a rank's tile buffer: exact sum of the items' partial sums (commitFixed)
  long long r, g, b;    // units of 2^-36
  uint32_t flags;       // bit k: a NaN partial sum in channel k; bit 3+k: +inf; bit 6+k: -inf
  uint32_t pad;
};

struct RenderArgs {
  DevScene scene;
  DevCamera cam;
  int32_t imageWidth, imageHeight, tilesX, tilesY, numTiles;
  int32_t spp, maxBounce, sampleFirst;
  uint64_t seed;
  float background[3];
  float tMin;
  int32_t tileFirst, tileStride, numLocalTiles;
  int32_t tileBlock;   // tile order: row-major inside tileBlock x tileBlock blocks of tiles (srtTileFromOrder)
  int32_t numQueues;   // work counters; a wave starts on queue blockIdx % numQueues
  int32_t unitTiles, numUnits;  // queue q owns units q, q+numQueues, ... of unitTiles consecutive local tiles
  int32_t sppChunks;
  int32_t sppBase, sppRem;  // chunk c renders samples [c * sppBase + min(c, sppRem), ...): spp = sppChunks * sppBase + sppRem
  int32_t numWork;  // numLocalTiles * sppChunks * 64 (one item = one pixel x one sample chunk)
  int32_t shadeMin, primMin, hitMin;  // wave scheduler thresholds (lanes waiting before that step kind runs)
  int32_t fuseMin;            // lanes at nodes after a primitive step for a node burst to follow in the same trip
  int32_t nodeBurst;          // max node visits per scheduling decision
  int32_t keepEighths;        // a node burst goes on while nNodes * keepEighths / 8 of its lanes are still at nodes
  int32_t primAgainMin;       // lanes at a primitive again after a primitive step for a second round in the same trip
  int32_t* queue;   // persistent-wave work counters, 16 ints apart (zeroed before launch)
  float4* out;      // sppChunks == 1: the caller's [localTile][64] buffer (the reference's float running sum, written
                    // directly); scratch path: chunk slots [chunk][localTile][64]
  int32_t chunkStride;  // scratch path: numLocalTiles * 64, else 0
  SrtFixedAccum* fix;   // atomic path: [localTile][64], items add their fixed-point partial sums here (null otherwise)
  float fixLimit;       // exact chunk sums: a partial sum of this much or more counts as infinite (toFixed36)
  unsigned long long* stats;  // 8 counters (SrtStats order) or nullptr
  SrtAovRecord* aov;          // counting variant only: per-pixel record of the ray at bounce aovDepth (srtRenderAov)
  int32_t aovDepth;
  float* attScratch;          // LDS-resident-tree variant: [3 * maxBounce + 3][grid * 1024] attenuation slots in global memory
  // work-item decomposition without divisions (restart step): groups of 64 items per unit, float reciprocals of
  // unitGroups and sppChunks, and srtTileFromOrder as a table over the image's tiles (tx | ty << 16)
  int32_t unitGroups;
  float rcpUnitGroups, rcpChunks;
  const uint32_t* tileXY;
  // path-pool kernel (srt_wavefront.hip): wfPoolSize contexts of 128 B per workgroup, the attenuation levels beyond
  // the four a context holds ([workgroup][level - 4][channel][context]), ring capacity (a power of two >= wfPoolSize),
  // lanes with nothing to traverse before a wave swaps finished walks for READY contexts (wfSwapMin when no full batch
  // waits to be served, wfSwapBig regardless), and the word a workgroup that gave up (bounded spin exceeded) adds to
  char* wfPool;
  float* wfAttHi;
  int32_t wfPoolSize, wfRingCap, wfSwapMin, wfSwapBig;
  int32_t wfRingShift, wfRingMul3;  // wfRingCap = (wfRingMul3 ? 3 : 1) << wfRingShift
  int32_t wfFarRounds;  // hybrid form: node visits per round (the last one serves the lanes outside LDS as well)
  int32_t* wfError;
};

struct TraceArgs {
  DevScene scene;
  const SrtRay* rays;
  SrtHit* hits;
  int64_t n;
};

struct ResolveArgs {
  const float4* gathered;  // [rank][localTile][64]
  int32_t imageWidth, imageHeight, tilesX;
  int32_t tileStride, numLocalTiles;
  int32_t tileBlock;
  int32_t spp;
  uint8_t* rgba;      // may be null
  float4* accumImage;  // may be null
};


// ---- tile order.  Tiles are numbered along a blocked curve, not row-major: the image is cut into
// B x B blocks of tiles (edge blocks smaller), blocks row-major, tiles row-major inside a block with row
// iy rotated by iy.  Work is issued in this order (so the tiles in flight at any time form a compact 2-D
// patch and the rays in flight stay coherent) and ranks take every tileStride-th position of it (the
// rotation keeps a rank's tiles from lining up in columns).  B = 1 is plain row-major.
#if defined(__HIPCC__)
#define SRT_HD __host__ __device__
#else
#define SRT_HD
#endif
SRT_HD inline void srtTileFromOrder(int i, int tilesX, int tilesY, int B, int& tx, int& ty) {
  const int by = i / (B * tilesX), r = i - by * B * tilesX;
  const int bh = (tilesY - by * B) < B ? (tilesY - by * B) : B;
  const int bx = r / (B * bh), r2 = r - bx * B * bh;
  const int bw = (tilesX - bx * B) < B ? (tilesX - bx * B) : B;
  const int iy = r2 / bw, ixr = r2 - iy * bw;
  int ix = ixr - iy % bw;
  ix = ix < 0 ? ix + bw : ix;
  tx = bx * B + ix;
  ty = by * B + iy;
}
SRT_HD inline int srtOrderFromTile(int tx, int ty, int tilesX, int tilesY, int B) {
  const int by = ty / B, iy = ty - by * B, bx = tx / B, ix = tx - bx * B;
  const int bh = (tilesY - by * B) < B ? (tilesY - by * B) : B;
  const int bw = (tilesX - bx * B) < B ? (tilesX - bx * B) : B;
  return by * B * tilesX + bx * B * bh + iy * bw + (ix + iy) % bw;
}

#endif
