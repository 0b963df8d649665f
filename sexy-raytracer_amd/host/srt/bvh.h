// bvh.h -- host mirror of bvhNode (bvh.h:9-110).  The tree is built at construction, like the
// reference's ctor (bvh.h:55-95: one generator draw per node in pre-order, std::sort by the boxes'
// minimum on the drawn axis, median split), by libsrt_hip.so's builder (srtBuildBvh), so the
// process-global generator is consumed at the same point of scene setup as in the reference.
#ifndef SRT_HOST_BVH_H
#define SRT_HOST_BVH_H

#include <iostream>
#include <vector>

#include "hittablelist.h"

class bvhNode : public hittable {
 public:
  bvhNode() {}
  bvhNode(const hittableList& list, float time0, float time1) : bvhNode(list.objects, 0, list.objects.size(), time0, time1) {}
  bvhNode(const std::vector<shared_ptr<hittable>>& srcObjects, size_t start, size_t end, float time0, float time1)
      : objects(srcObjects.begin() + start, srcObjects.begin() + end), t0(time0), t1(time1) {
    sceneFlattener f;
    SrtMaterialIn dummy{};
    dummy.type = SRT_MAT_DIELECTRIC;
    dummy.albedoTex = dummy.normalTex = dummy.metallicTex = dummy.roughnessTex = -1;
    dummy.ir = 1.0f;
    for (const auto& o : objects) {
      if (!o->isPrimitive()) {
        std::cerr << "No bounding box in bvhNode constructor.\n";  // bvh.h:37-38: only primitives are supported
        continue;
      }
      o->populate(f);
    }
    // geometry only: materials are irrelevant to the build
    f.materials.assign(1, dummy);
    for (auto& t : f.triangles) t.material = 0;
    for (auto& s : f.spheres) s.material = 0;
    f.textures.clear();
    f.texels.clear();
    f.world.assign(1, SrtWorldItem{SRT_WORLD_BVH, 0, (int32_t)f.prims.size(), time0, time1, 0, nullptr});
    SrtSceneDesc d = f.desc();
    int32_t count = 0, depth = 0;
    nodes.resize(f.prims.empty() ? 0 : 2 * f.prims.size());
    if (f.prims.empty() || srtBuildBvh(&d, 0, nodes.data(), (int32_t)nodes.size(), &count, &depth) != 0) {
      std::cerr << "No bounding box in bvh node constructor.\n";  // bvh.h:90-92
      count = 0;
    }
    nodes.resize(count);
    if (count) box = aabb(vec3f(nodes[0].bmin[0], nodes[0].bmin[1], nodes[0].bmin[2]),
                          vec3f(nodes[0].bmax[0], nodes[0].bmax[1], nodes[0].bmax[2]));
  }

  bool boundingBox(float, float, aabb& outputBox) const override {  // bvh.h:107-110
    outputBox = box;
    return true;
  }
  // as an element of the world list: one SRT_WORLD_BVH item carrying the tree built at construction
  int populate(sceneFlattener& f) const override {
    const int32_t first = (int32_t)f.prims.size();
    for (const auto& o : objects)
      if (o->isPrimitive()) o->populate(f);
    f.trees.emplace_back(nodes);
    for (auto& n : f.trees.back()) {  // primitive refs are relative to this node's own list
      if (n.left < 0) n.left = ~(~n.left + first);
      if (n.right < 0) n.right = ~(~n.right + first);
    }
    f.world.push_back(SrtWorldItem{SRT_WORLD_BVH, first, (int32_t)f.prims.size() - first, t0, t1,
                                   (int32_t)f.trees.back().size(), f.trees.back().data()});
    return -1;
  }
  bool isPrimitive() const override { return false; }

 public:
  std::vector<shared_ptr<hittable>> objects;
  std::vector<SrtBvhNode> nodes;  // pre-order, the layout populateVector walks (bvh.h:112-148)
  aabb box;
  float t0 = 0, t1 = 0;
};

#endif
