// hit_probe: scene code in the reference's style that CALLS world.hit() on the host (hittable.h:26,
// hittablelist.h:33-47).  The mirror classes forward the call to the device (host/srt/hittable.h); this
// program prints one line per ray for tests/test_gpu_example.py to compare with the batch entry point.
//   srt_hit_probe            (the three-sphere scene of configs C1/C2, rays through a 7x5 grid of the camera)
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
  return 0;
}
