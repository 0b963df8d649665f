// main.cpp-shaped driver: the reference's program flow (camera -> scene -> render -> PNG,
// main.cpp:156-242) on the MI355X path.  Scene setup uses the same vocabulary as the reference
// (model::create(...)->init(), sphere, pbrMetallicRoughness, checker, diffuseLight, metal, bvhNode);
// the pixel loop is hipDevice::rtFrame instead of the CPU loops.
//
//   usage: srt_main [--gltf file] [--height H] [--spp N] [--bounces B] [--out file.png] [--chunks K]
//   SRT_DATA_DIR selects the directory of the glTF's images (default "../data/", as the reference).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "srt/bvh.h"
#include "srt/camera.h"
#include "srt/color.h"
#include "srt/device.h"
#include "srt/globals.h"
#include "srt/hittablelist.h"
#include "srt/material.h"
#include "srt/model.h"
#include "srt/sphere.h"

static std::string gltfPath = "../data/masterchief2-separate-xf.gltf";

static shared_ptr<hittable> unitSphere(float x, float y, float z, shared_ptr<material> m, float r = 1.0f) {
  return make_shared<sphere>(vec3f(x, y, z), vec3f(x, y, z), 0, 1.0f, r, m);
}

// the scene of main.cpp:54-154: mesh triangles, checker ground, light, textured sphere, mirror
static hittableList buildScene(bool& ok) {
  hittableList objects, scene;
  auto chief = model::create(gltfPath);
  ok = chief->init();
  if (!ok) std::cerr << "ERROR: could not load " << gltfPath << "\n";
  for (const auto& m : chief->meshes)
    for (const auto& tri : m->triangles) objects.add(tri);

  auto ground = make_shared<pbrMetallicRoughness>(make_shared<checker>(color3f(0.2f, 0.3f, 0.1f), color3f(0.9f, 0.9f, 0.9f)));
  objects.add(unitSphere(0, -1000, 0, ground, 1000));
  objects.add(unitSphere(-7.0f, 4.0f, 6.0f, make_shared<diffuseLight>(color3f(250.2f, 220.9f, 110.2f))));

  const std::string d = srtDataDir();
  auto iron = make_shared<pbrMetallicRoughness>(make_shared<imagePNG>((d + "rustediron2_basecolor-2x1.png").c_str(), 3),
                                                make_shared<imagePNG>((d + "rustediron2_normal-2x1.png").c_str(), 3),
                                                make_shared<imagePNG>((d + "rustediron2_metallic-2x1.png").c_str(), 1),
                                                make_shared<imagePNG>((d + "rustediron2_roughness-2x1.png").c_str(), 1),
                                                vec4f(1.0f, 1.0f, 1.0f, 1.0f));
  objects.add(unitSphere(-3.0f, 1.0f, 0.0f, iron));
  objects.add(unitSphere(3.0f, 1.0f, 0.0f, make_shared<metal>(color3f(0.7f, 0.6f, 0.5f), 0.0f)));

  scene.add(make_shared<bvhNode>(objects, 0, 1));
  return scene;
}

int main(int argc, char** argv) {
  int imageHeight = 720, numSamples = 5000, maxBounce = 4, chunks = 0;
  std::string out = "test.png";
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--gltf")) gltfPath = argv[i + 1];
    else if (!strcmp(argv[i], "--height")) imageHeight = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--spp")) numSamples = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--bounces")) maxBounce = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--chunks")) chunks = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--out")) out = argv[i + 1];
  }
  const float aspect = 16.0f / 9.0f;
  const int imageWidth = static_cast<int>(imageHeight * aspect);
  camera mainCamera(vec3f(0.0f, 3.0f, 5.0f), vec3f(0, 2.5f, 0), vec3f(0, 1.0f, 0), 70.0f, aspect, 0.1f, 10.0f, 0, 1.0f);
  color3f background(0.53f, 0.81f, 0.92f);
  uint8_t* target = static_cast<uint8_t*>(malloc(sizeof(uint8_t) * 4 * imageWidth * imageHeight));

  bool ok = false;
  hittableList world = buildScene(ok);
  if (!ok) return 1;

  hipDevice device;
  if (!device.init(imageWidth, imageHeight, world)) return 1;
  device.sppChunks = chunks;
  auto t0 = std::chrono::steady_clock::now();
  if (!device.rtFrame(target, imageWidth, imageHeight, mainCamera, background, numSamples, maxBounce)) return 1;
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  device.terminate();

  stbi_write_png(out.c_str(), imageWidth, imageHeight, 4, target, 4 * imageWidth);
  free(target);
  std::cerr << imageWidth << "x" << imageHeight << " @" << numSamples << " spp: " << device.numPrims << " primitives, kernel "
            << device.lastKernelMs << " ms (" << (double)imageWidth * imageHeight * numSamples / device.lastKernelMs / 1e3
            << " Msamples/s), wall " << sec << " s -> " << out << "\nDone.\n";
  return 0;
}
