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

// Child / world reference encoding on the device:
//   ref >= 0          BVH node index
//   ref <  0          primitive: r = ~ref, (r & 1) = 1 sphere / 0 triangle, r >> 1 = index
//   SRT_REF_DONE      traversal sentinel (never a valid primitive)
#define SRT_REF_DONE ((int32_t)0x80000000)

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
  int64_t offset;  // byte offset into texels
  int32_t even, odd;
  float color[3];
  int32_t pad;
};

struct DevScene {
  // 2 x float4 per node: (bmin.xyz, left) (bmax.xyz, right)      -- 32 B / node visit
  const float4* nodes;
  // 3 x float4 per triangle: (v0.xyz, n.x) (v1.xyz, n.y) (v2.xyz, n.z), n = (v1-v0)x(v2-v0)  -- 48 B / test
  const float4* triTest;
  // 4 x float4 per triangle, read once per shaded hit:
  // (N^.xyz, uv0.u) (T.xyz, uv0.v) (B.xyz, uv1.u) (uv1.v, uv2.u, uv2.v, material)
  const float4* triShade;
  // 3 x float4 per sphere: (c0.xyz, radius) (c1.xyz, material | moving<<30) (t0, t1, -, -)   -- 16..48 B / test
  const float4* spheres;
  // 1 byte per node: the axis its children are split on (left = lower side), 3 = unknown.  Read only by
  // SRT_TRAVERSE_CLOSEST, which visits the nearer child first; FAITHFUL keeps bvh.h's left-then-right.
  const uint8_t* nodeAxis;
  const int32_t* triPrimId;  // device index -> index into the scene's prims[] list
  const int32_t* sphPrimId;
  const int32_t* world;  // refs, world-list order
  int32_t numWorld;
  int32_t stackDepth;  // max pending right children over all trees
  int32_t numNodes, numTris, numSpheres;  // record counts (buffer-resource extents)
  int32_t fastDivScene;  // 1: every box coordinate is 0 or in [2^-77, 2^30] (see fastDiv in srt_kernels.hip)
  const DevMaterial* materials;
  const DevTexture* textures;
  const uint8_t* texels;  // zero padded by >= 16 bytes
};

struct DevCamera {
  float origin[3], lleft[3], horizontal[3], vertical[3], hor[3], vert[3];
  float lensRadius, time0, time1;
};

struct RenderArgs {
  DevScene scene;
  DevCamera cam;
  int32_t imageWidth, imageHeight, tilesX, numTiles;
  int32_t spp, maxBounce, sampleFirst;
  uint64_t seed;
  float background[3];
  float tMin;
  int32_t tileFirst, tileStride, numLocalTiles;
  int32_t sppChunks;
  int32_t numWork;  // numLocalTiles * sppChunks * 64 (one item = one pixel x one sample chunk)
  int32_t shadeMin, primMin, hitMin;  // wave scheduler thresholds (lanes waiting before that step kind runs)
  int32_t nodeBurst;          // max node visits per scheduling decision
  int32_t* queue;   // persistent-wave work counter (zeroed before launch)
  float4* out;      // [chunk][localTile][64]
  unsigned long long* stats;  // 8 counters (SrtStats order) or nullptr
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
  int32_t spp;
  uint8_t* rgba;      // may be null
  float4* accumImage;  // may be null
};

#endif
