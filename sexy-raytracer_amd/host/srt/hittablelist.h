// hittablelist.h -- host mirror of hittablelist.h:12-64.
#ifndef SRT_HOST_HITTABLELIST_H
#define SRT_HOST_HITTABLELIST_H

#include <vector>

#include "hittable.h"

class hittableList : public hittable {
 public:
  hittableList() {}
  hittableList(shared_ptr<hittable> object) { add(object); }
  void clear() {
    objects.clear();
    dropHitSession();
  }
  void add(shared_ptr<hittable> object) {
    objects.push_back(object);
    dropHitSession();
  }

  bool boundingBox(float time0, float time1, aabb& outputBox) const override {  // hittablelist.h:49-64
    if (objects.empty()) return false;
    aabb tempBox;
    bool firstBox = true;
    for (const auto& object : objects) {
      if (!object->boundingBox(time0, time1, tempBox)) return false;
      outputBox = firstBox ? tempBox : surroundingBox(outputBox, tempBox);
      firstBox = false;
    }
    return true;
  }
  // as the world handed to rayColor (main.cpp:187,217): one world item per element, in order
  int populate(sceneFlattener& f) const override {
    for (const auto& object : objects) {
      int first = object->populate(f);
      if (object->isPrimitive()) f.world.push_back(SrtWorldItem{SRT_WORLD_PRIM, first, 1, 0.0f, 0.0f, 0, nullptr});
    }
    return -1;
  }
  bool isPrimitive() const override { return false; }

 public:
  std::vector<shared_ptr<hittable>> objects;
};

#endif
