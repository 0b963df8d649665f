// Builds a scene through the C++ host layer and dumps its flattened form (no GPU needed):
// used by tests/test_host_cpp.py to compare the C++ scene-build path with the Python one.
//   srt_flatten_dump <gltf> <out.bin> [png-roundtrip.png]
//   srt_flatten_dump --model <gltf|glb>      what gltfLoad made of the file: one line per mesh (vertex, texcoord,
//                                            triangle counts, material factors, FNV-1a of the raw arrays) and one for
//                                            the triangles as objects.add() would see them (main.cpp:81-85)
#include <cstdio>
#include <cstring>
#include <iostream>

#include "srt/bvh.h"
#include "srt/hittablelist.h"
#include "srt/material.h"
#include "srt/model.h"
#include "srt/sphere.h"

template <typename T>
static void put(FILE* f, const std::vector<T>& v) {
  int64_t n = (int64_t)v.size();
  fwrite(&n, 8, 1, f);
  if (n) fwrite(v.data(), sizeof(T), v.size(), f);
}

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

static int dumpModel(const char* path) {
  auto m = model::create(path);
  if (!m->init()) {
    std::cout << "gltfLoad failed\n";
    return 1;
  }
  sceneFlattener f;
  for (size_t i = 0; i < m->meshes.size(); ++i) {
    const auto& ms = m->meshes[i];
    uint64_t hp = 1469598103934665603ull, ht = hp;
    for (const auto& p : ms->positions) {
      float v[3] = {p(0), p(1), p(2)};
      hp = fnv(v, 12, hp);
    }
    for (const auto& t : ms->texcoords) {
      float v[2] = {t(0), t(1)};
      ht = fnv(v, 8, ht);
    }
    std::cout << "mesh " << i << " positions " << ms->positions.size() << " texcoords " << ms->texcoords.size() << " triangles "
              << ms->triangles.size() << " material " << (ms->matPtr ? 1 : 0) << " hp " << hp << " ht " << ht << "\n";
    if (ms->matPtr || ms->triangles.empty())
      for (const auto& t : ms->triangles) t->populate(f);
  }
  uint64_t h = 1469598103934665603ull;
  for (const auto& t : f.triangles) h = fnv(t.uv, sizeof(t.uv), fnv(t.p, sizeof(t.p), h));
  std::cout << "flattened triangles " << f.triangles.size() << " materials " << f.materials.size() << " textures " << f.textures.size()
            << " hash " << h << "\n";
  for (const auto& mt : f.materials)
    std::cout << "material type " << mt.type << " albedo " << mt.albedo[0] << " " << mt.albedo[1] << " " << mt.albedo[2] << " " << mt.albedo[3]
              << " metalness " << mt.metalness << " roughness " << mt.roughness << " albedoTex " << mt.albedoTex << " normalTex " << mt.normalTex << "\n";
  for (const auto& t : f.textures) std::cout << "texture kind " << t.kind << " " << t.width << "x" << t.height << " bpp " << t.bpp << "\n";
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  if (!strcmp(argv[1], "--model")) return dumpModel(argv[2]);
  srtHostRandomReset();
  hittableList objects, scene;
  auto m = model::create(argv[1]);
  if (!m->init()) {
    std::cerr << "gltfLoad failed\n";
    return 1;
  }
  for (const auto& ms : m->meshes)
    for (const auto& t : ms->triangles) objects.add(t);
  auto ground = make_shared<pbrMetallicRoughness>(make_shared<checker>(color3f(0.2f, 0.3f, 0.1f), color3f(0.9f, 0.9f, 0.9f)));
  objects.add(make_shared<sphere>(vec3f(0, -1000, 0), vec3f(0, -1000, 0), 0, 1.0f, 1000, ground));
  objects.add(make_shared<sphere>(vec3f(3.0f, 1.0f, 0), vec3f(3.0f, 1.0f, 0), 0, 1.0f, 1.0f, make_shared<metal>(color3f(0.7f, 0.6f, 0.5f), 0.0f)));
  scene.add(make_shared<bvhNode>(objects, 0, 1));
  float next = randomFloat();  // the generator position after the build (4043-ish draws)

  sceneFlattener f;
  scene.populate(f);
  FILE* fp = fopen(argv[2], "wb");
  if (!fp) return 1;
  put(fp, f.triangles);
  put(fp, f.spheres);
  put(fp, f.prims);
  put(fp, f.materials);
  put(fp, f.textures);
  put(fp, f.texels);
  put(fp, f.trees.at(0));
  std::vector<float> tail = {next};
  put(fp, tail);
  fclose(fp);
  if (argc > 3) {  // PNG writer/reader round trip on the first image texture
    for (const auto& t : f.textures)
      if (t.kind == SRT_TEX_IMAGE && t.width > 0) {
        if (!stbi_write_png(argv[3], t.width, t.height, t.bpp, f.texels.data() + t.texelOffset, t.width * t.bpp)) return 1;
        int w, h, c;
        uint8_t* back = stbi_load(argv[3], &w, &h, &c, t.bpp);
        bool same = back && w == t.width && h == t.height && !memcmp(back, f.texels.data() + t.texelOffset, (size_t)w * h * t.bpp);
        free(back);
        if (!same) {
          std::cerr << "png round trip mismatch\n";
          return 1;
        }
        break;
      }
  }
  std::cout << f.triangles.size() << " triangles, " << f.spheres.size() << " spheres, " << f.trees.at(0).size() << " nodes\n";
  return 0;
}
