// srt_comm.cpp -- multi-GPU side of the C ABI (include/srt_hip.h, "Multi-GPU"): one process per GPU,
// one RCCL communicator per context, and the path's ONLY data-path collective: one ncclGather of the
// ranks' equal-sized tile buffers to rank 0 (the reference has no multi-device code, its
// device seam gl.h:28-31 is single-GPU).  Pixels are independent (main.cpp:200-227 carries no state
// between pixels once the RNG is counter-based), the scene is replicated: nothing else is exchanged.
//
// The unique id is created by rank 0 (srtCommGetUniqueId) and handed to the other ranks by whatever
// the host program uses to start its processes (a file, MPI, torch.distributed's store); this library
// does not open sockets of its own.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "srt_device.h"

extern "C" {
// implemented in srt_api.cpp
int srtCtxFail(SrtContext* ctx, const char* text);
int srtCtxDevice(const SrtContext* ctx);
void** srtCtxCommSlot(SrtContext* ctx);  // where the context keeps its ncclComm_t (opaque there)
int* srtCtxCommRanks(SrtContext* ctx);   // [0] = number of ranks, [1] = this rank

int srtCommGetUniqueId(void* id128) {
  if (!id128) return 1;
  static_assert(sizeof(ncclUniqueId) == SRT_COMM_ID_BYTES, "SRT_COMM_ID_BYTES must match ncclUniqueId");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) {
    fprintf(stderr, "srt_hip: ncclGetUniqueId -> %s\n", ncclGetErrorString(r));
    return 1;
  }
  memcpy(id128, &id, sizeof id);
  return 0;
}

int srtCommInit(SrtContext* ctx, const void* id128, int32_t numRanks, int32_t rank) {
  if (!ctx || !id128) return 1;
  if (numRanks < 1 || rank < 0 || rank >= numRanks) return srtCtxFail(ctx, "srtCommInit: rank out of range");
  if (*srtCtxCommSlot(ctx)) return srtCtxFail(ctx, "srtCommInit: this context already has a communicator");
  if (hipSetDevice(srtCtxDevice(ctx)) != hipSuccess) return srtCtxFail(ctx, "srtCommInit: hipSetDevice failed");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, numRanks, id, rank);
  if (r != ncclSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "srtCommInit: ncclCommInitRank(%d of %d) -> %s", rank, numRanks, ncclGetErrorString(r));
    return srtCtxFail(ctx, buf);
  }
  *srtCtxCommSlot(ctx) = comm;
  srtCtxCommRanks(ctx)[0] = numRanks;
  srtCtxCommRanks(ctx)[1] = rank;
  return 0;
}

int srtCommDestroy(SrtContext* ctx) {
  if (!ctx) return 0;
  void** slot = srtCtxCommSlot(ctx);
  if (*slot) {
    (void)hipSetDevice(srtCtxDevice(ctx));
    (void)ncclCommDestroy(static_cast<ncclComm_t>(*slot));
    *slot = nullptr;
  }
  srtCtxCommRanks(ctx)[0] = 1;
  srtCtxCommRanks(ctx)[1] = 0;
  return 0;
}

// One gather: every rank sends its float4[numLocalTiles * 64] tile buffer (what srtRenderTiles wrote for
// tileFirst = rank, tileStride = numRanks); rank 0 receives float4[numRanks][numLocalTiles * 64], the layout
// srtResolveTiles un-permutes.  Asynchronous on `stream`.
int srtGatherTiles(SrtContext* ctx, const SrtRenderParams* p, const void* dLocalTiles, void* dGathered, void* streamPtr) {
  if (!ctx || !p || !dLocalTiles) return 1;
  const int numRanks = srtCtxCommRanks(ctx)[0], rank = srtCtxCommRanks(ctx)[1];
  if (p->tileStride != numRanks || p->tileFirst != rank)
    return srtCtxFail(ctx, "srtGatherTiles: the tile split of the render parameters is not this communicator's (tileStride = ranks, tileFirst = rank)");
  const size_t count = (size_t)srtNumLocalTiles(p->imageWidth, p->imageHeight, p->tileStride) * SRT_TILE_PIXELS * 4;  // floats
  hipStream_t stream = static_cast<hipStream_t>(streamPtr);
  if (hipSetDevice(srtCtxDevice(ctx)) != hipSuccess) return srtCtxFail(ctx, "srtGatherTiles: hipSetDevice failed");
  if (numRanks == 1) {  // nothing to exchange: the gathered buffer is the local one
    if (dGathered && dGathered != dLocalTiles &&
        hipMemcpyAsync(dGathered, dLocalTiles, count * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return srtCtxFail(ctx, "srtGatherTiles: copy failed");
    return 0;
  }
  ncclComm_t comm = static_cast<ncclComm_t>(*srtCtxCommSlot(ctx));
  if (!comm) return srtCtxFail(ctx, "srtGatherTiles: no communicator (srtCommInit)");
  if (rank == 0 && !dGathered) return srtCtxFail(ctx, "srtGatherTiles: rank 0 needs the gathered buffer");
  ncclResult_t r = ncclGather(dLocalTiles, dGathered, count, ncclFloat32, 0, comm, stream);
  if (r != ncclSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "srtGatherTiles: ncclGather -> %s", ncclGetErrorString(r));
    return srtCtxFail(ctx, buf);
  }
  return 0;
}

// main.cpp:182-227 across the ranks of the communicator (collective: every rank calls it with the same
// parameters): each rank renders its tile positions, ONE gather, rank 0 resolves into the caller-owned
// HOST buffers (hAccum float[W*H*4], hRgba uint8[W*H*4]; either may be NULL, both are ignored on other ranks).
int srtRenderImageRanks(SrtContext* ctx, const SrtRenderParams* pIn, float* hAccum, uint8_t* hRgba) {
  if (!ctx || !pIn) return 1;
  const int numRanks = srtCtxCommRanks(ctx)[0], rank = srtCtxCommRanks(ctx)[1];
  SrtRenderParams p = *pIn;
  p.tileFirst = rank;
  p.tileStride = numRanks;
  if (hipSetDevice(srtCtxDevice(ctx)) != hipSuccess) return srtCtxFail(ctx, "srtRenderImageRanks: hipSetDevice failed");
  const size_t nPix = (size_t)p.imageWidth * p.imageHeight;
  const size_t localBytes = (size_t)srtNumLocalTiles(p.imageWidth, p.imageHeight, numRanks) * SRT_TILE_PIXELS * sizeof(float4);
  void *dLocal = nullptr, *dGathered = nullptr, *dRgba = nullptr, *dAcc = nullptr;
  int32_t* dAgree = nullptr;
  int rc = 1;
  do {
    // Everything that can fail on ONE rank happens before the gather, and the ranks agree on it first: a rank that
    // broke out here while the others entered ncclGather would leave them waiting for ever.  The agreement is a
    // 4-byte all-reduce (minimum of the ranks' status); only a rank that cannot even allocate those 4 bytes leaves
    // without taking part -- its peers then need ncclCommAbort (srtCommDestroy), as after any lost rank.
    if (numRanks > 1 && hipMalloc((void**)&dAgree, sizeof(int32_t)) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: hipMalloc (4 bytes)"); break; }
    int32_t ok = 1;
    if (hipMalloc(&dLocal, localBytes) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: hipMalloc"); ok = 0; }
    if (ok && rank == 0) {
      if (hipMalloc(&dGathered, localBytes * numRanks) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: hipMalloc"); ok = 0; }
      if (ok && hRgba && hipMalloc(&dRgba, nPix * 4) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: hipMalloc"); ok = 0; }
      if (ok && hAccum && hipMalloc(&dAcc, nPix * sizeof(float4)) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: hipMalloc"); ok = 0; }
    }
    if (ok && srtRenderTiles(ctx, &p, dLocal, nullptr)) ok = 0;
    if (ok && hipDeviceSynchronize() != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: render kernel failed"); ok = 0; }
    if (numRanks > 1) {
      ncclComm_t comm = static_cast<ncclComm_t>(*srtCtxCommSlot(ctx));
      int32_t agreed = 0;
      if (!comm) { srtCtxFail(ctx, "srtRenderImageRanks: no communicator (srtCommInit)"); break; }
      if (hipMemcpy(dAgree, &ok, sizeof ok, hipMemcpyHostToDevice) != hipSuccess ||
          ncclAllReduce(dAgree, dAgree, 1, ncclInt32, ncclMin, comm, nullptr) != ncclSuccess ||
          hipMemcpy(&agreed, dAgree, sizeof agreed, hipMemcpyDeviceToHost) != hipSuccess) {
        srtCtxFail(ctx, "srtRenderImageRanks: the ranks could not agree on the render's status");
        break;
      }
      if (!agreed) {
        if (ok) srtCtxFail(ctx, "srtRenderImageRanks: another rank failed before the gather");
        break;
      }
    } else if (!ok) {
      break;
    }
    if (srtGatherTiles(ctx, &p, dLocal, dGathered, nullptr)) break;
    if (rank == 0 && srtResolveTiles(ctx, &p, dGathered, dRgba, dAcc, nullptr)) break;
    if (hipDeviceSynchronize() != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: gather or resolve failed"); break; }
    if (rank == 0) {
      if (hRgba && hipMemcpy(hRgba, dRgba, nPix * 4, hipMemcpyDeviceToHost) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: copy rgba"); break; }
      if (hAccum && hipMemcpy(hAccum, dAcc, nPix * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) { srtCtxFail(ctx, "srtRenderImageRanks: copy accum"); break; }
    }
    rc = 0;
  } while (0);
  if (dAgree) (void)hipFree(dAgree);
  for (void* q : {dLocal, dGathered, dRgba, dAcc})
    if (q) (void)hipFree(q);
  return rc;
}

}  // extern "C"
