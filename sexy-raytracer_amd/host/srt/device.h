// device.h -- hipDevice: the working counterpart of the reference's disabled glDevice seam
// (gl.h:16-40: init / rtFrame / terminate; main.cpp:190-199,231).
//   init(w, h, world)                       flatten (populate) + upload to HBM
//   rtFrame(target, w, h, cam, bg, spp, b)  render the frame into the caller-owned RGBA8 target
//   terminate()
//   uniqueId / initRanks                    multi-GPU: one process per GPU; rtFrame then renders this rank's
//                                           tiles, the library gathers them with ONE ncclGather and rank 0's
//                                           target receives the frame (include/srt_hip.h "Multi-GPU")
// Returns bool and reports on std::cerr like the reference.  There is no CPU fallback.
#ifndef SRT_HOST_DEVICE_H
#define SRT_HOST_DEVICE_H

#include <iostream>
#include <vector>

#include "camera.h"
#include "hittablelist.h"

class hipDevice {
 public:
  hipDevice() {}
  ~hipDevice() { terminate(); }
  hipDevice(const hipDevice&) = delete;

  bool init(int w, int h, const hittableList& world, int deviceOrdinal = 0) {
    width = w;
    height = h;
    if (srtCreate(deviceOrdinal, &ctx) != 0) {
      std::cerr << "ERROR: no usable HIP device\n";
      ctx = nullptr;
      return false;
    }
    sceneFlattener f;
    world.populate(f);
    SrtSceneDesc d = f.desc();
    if (srtUploadScene(ctx, &d) != 0) return error();
    numPrims = (int)f.prims.size();
    return true;
  }

  // Multi-GPU.  Rank 0 obtains an id (128 bytes) and hands it to the other processes by its own means;
  // every process then calls initRanks after init.  Single-process programs never call these.
  static bool uniqueId(void* id128) { return srtCommGetUniqueId(id128) == 0; }
  bool initRanks(const void* id128, int numRanks, int rank) {
    if (!ctx) return false;
    if (srtCommInit(ctx, id128, numRanks, rank) != 0) return error();
    ranks = numRanks;
    return true;
  }

  // the pixel loop main.cpp:200-227 for the whole frame; frameData = uint8[w*h*4] (main.cpp:182)
  bool rtFrame(void* frameData, int w, int h, const camera& cam, const color3f& background, int numSamples,
               int maxBounce, uint64_t seed = 1, float* accum = nullptr) {
    if (!ctx) return false;
    if (srtSetCamera(ctx, &cam.data()) != 0) return error();
    SrtRenderParams p{};
    p.imageWidth = w; p.imageHeight = h; p.spp = numSamples; p.maxBounce = maxBounce; p.seed = seed;
    for (int i = 0; i < 3; ++i) p.background[i] = background(i);
    p.tMin = 0.001f;  // main.cpp:39
    p.traversal = SRT_TRAVERSE_FAITHFUL;
    p.tileFirst = 0; p.tileStride = 1;
    p.sppChunks = sppChunks;
    if (ranks > 1) {  // collective: every rank renders its tiles, one gather, rank 0 fills its target
      if (srtRenderImageRanks(ctx, &p, accum, static_cast<uint8_t*>(frameData)) != 0) return error();
    } else if (srtRenderImage(ctx, &p, accum, static_cast<uint8_t*>(frameData)) != 0) {
      return error();
    }
    (void)srtLastKernelMs(ctx, &lastKernelMs);
    return true;
  }

  bool trace(const std::vector<SrtRay>& rays, std::vector<SrtHit>& hits) {
    hits.resize(rays.size());
    if (!ctx || srtTraceRays(ctx, rays.data(), (int64_t)rays.size(), hits.data(), SRT_TRAVERSE_FAITHFUL) != 0) return error();
    return true;
  }

  void terminate() {
    if (ctx) srtDestroy(ctx);
    ctx = nullptr;
  }

 public:
  int sppChunks = 0;  // 0 = library default; 1 = the reference's single running sum per pixel
  float lastKernelMs = 0;
  int numPrims = 0;

 private:
  bool error() {
    std::cerr << "ERROR: " << (ctx ? srtLastError(ctx) : "no context") << "\n";
    return false;
  }
  SrtContext* ctx = nullptr;
  int width = 0, height = 0;
  int ranks = 1;
};

#endif
