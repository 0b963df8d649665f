// hit_probe: scene code in the reference's style that CALLS world.hit() on the host (hittable.h:26,
// hittablelist.h:33-47).  The mirror classes forward the call to the device (host/srt/hittable.h); this
// program prints one line per ray for tests/test_gpu_example.py to compare with the batch entry point.
//   srt_hit_probe            (the three-sphere scene of configs C1/C2, rays through a 7x5 grid of the camera)
// After the 35 hit lines and the bare-primitive lines it calls, in the reference's own words, rec.matPtr->scatter(r, rec,
// attenuation, scattered) on every hit (material.h:17, main.cpp:45), a light's emitted() (material.h:18-20, main.cpp:43)
// and texture::value() on a checker and on a PNG it writes first (texture.h:15): all answered by the device.
#include <cstdio>

#include "srt/bvh.h"
#include "srt/camera.h"
#include "srt/hittablelist.h"
#include "srt/material.h"
#include "srt/sphere.h"

static shared_ptr<hittable> unitSphere(float x, float y, float z, shared_ptr<material> m, float r = 1.0f) {
  return make_shared<sphere>(vec3f(x, y, z), vec3f(x, y, z), 0, 1.0f, r, m);
}

int main() {
  srtHostRandomReset();
  hittableList objects, world;
  auto ground = make_shared<pbrMetallicRoughness>(make_shared<checker>(color3f(0.2f, 0.3f, 0.1f), color3f(0.9f, 0.9f, 0.9f)));
  auto diffuse = make_shared<pbrMetallicRoughness>(color3f(0.4f * 255, 0.2f * 255, 0.1f * 255));
  auto glass = make_shared<dielectric>(1.5f);
  auto mirror = make_shared<metal>(color3f(0.7f, 0.6f, 0.5f), 0.0f);
  shared_ptr<material> mats[4] = {ground, diffuse, glass, mirror};
  objects.add(unitSphere(0, -1000, 0, ground, 1000));
  objects.add(unitSphere(-3.0f, 1.0f, 0.0f, diffuse));
  objects.add(unitSphere(0.0f, 1.0f, 0.0f, glass));
  objects.add(unitSphere(3.0f, 1.0f, 0.0f, mirror));
  world.add(make_shared<bvhNode>(objects, 0, 1));  // main.cpp:146

  const vec3f eye(0.0f, 3.0f, 5.0f);
  for (int j = 0; j < 5; ++j)
    for (int i = 0; i < 7; ++i) {
      ray r(eye, vec3f(-6.0f + 2.0f * i, -3.5f + 1.0f * j, -5.0f), 0.25f);
      hitRecord rec;
      if (world.hit(r, 0.001f, infinity, rec)) {  // main.cpp:39
        int m = -1;
        for (int k = 0; k < 4; ++k)
          if (rec.matPtr == mats[k]) m = k;
        printf("hit %a %a %a %a %a %a %a %d %d\n", rec.t, rec.p(0), rec.p(1), rec.p(2), rec.normal(0), rec.normal(1), rec.normal(2),
               rec.frontFace ? 1 : 0, m);
      } else {
        printf("miss\n");
      }
    }
  // a bare primitive answers too (sphere::hit, sphere.h:54-83)
  hitRecord rec;
  auto ball = unitSphere(0, 0, -3, mirror);
  printf("%s\n", ball->hit(ray(vec3f(0, 0, 0), vec3f(0, 0, -1), 0), 0.001f, infinity, rec) ? "ball hit" : "ball miss");
  printf("t %a\n", rec.t);

  // material::scatter on the hits of the grid, in grid order (each material counts its own calls: the RNG key)
  for (int j = 0; j < 5; ++j)
    for (int i = 0; i < 7; ++i) {
      ray r(eye, vec3f(-6.0f + 2.0f * i, -3.5f + 1.0f * j, -5.0f), 0.25f);
      hitRecord h;
      if (!world.hit(r, 0.001f, infinity, h)) continue;
      color3f attenuation(0, 0, 0);
      ray scattered;
      const bool ok = h.matPtr->scatter(r, h, attenuation, scattered);  // main.cpp:45
      printf("scatter %d %a %a %a %a %a %a %a %a %a %a\n", ok ? 1 : 0, attenuation(0), attenuation(1), attenuation(2), scattered.dir(0),
             scattered.dir(1), scattered.dir(2), scattered.o(0), scattered.o(1), scattered.o(2), scattered.time);
    }
  // material::emitted (main.cpp:43): a light answers with its texture, everything else with black
  auto lamp = make_shared<diffuseLight>(color3f(4.0f, 3.0f, 2.0f));
  const color3f e0 = lamp->emitted(0.3f, 0.6f, vec3f(1, 2, 3)), e1 = mirror->emitted(0.3f, 0.6f, vec3f(1, 2, 3));
  printf("emitted %a %a %a %a %a %a\n", e0(0), e0(1), e0(2), e1(0), e1(1), e1(2));
  // texture::value: the checker's sign pattern on p (texture.h:42-48: sin(10x) sin(10y) sin(10z), colours times 255) ...
  checker squares(color3f(0.2f, 0.3f, 0.1f), color3f(0.9f, 0.9f, 0.9f));
  for (int k = 0; k < 6; ++k) {
    const vec3f p(0.11f + 0.37f * k, 0.05f - 0.21f * k, 0.4f + 0.13f * k);
    const color3f c = squares.value(0.5f, 0.5f, p);
    printf("checker %a %a %a\n", c(0), c(1), c(2));
  }
  // ... and an image's nearest texel, v flipped (texture.h:129-148), from a PNG written here
  uint8_t px[2 * 3 * 3];
  for (int k = 0; k < 18; ++k) px[k] = (uint8_t)(10 + 13 * k);
  const char* file = "/tmp/srt_hit_probe_texture.png";
  if (stbi_write_png(file, 3, 2, 3, px, 9)) {
    imagePNG image(file, 3);
    const float uv[4][2] = {{0.0f, 0.0f}, {0.4f, 0.9f}, {0.99f, 0.2f}, {1.0f, 1.0f}};
    for (int k = 0; k < 4; ++k) {
      const color3f c = image.value(uv[k][0], uv[k][1], vec3f(0, 0, 0));
      printf("texel %a %a %a\n", c(0), c(1), c(2));
    }
  }
  return 0;
}
