// flatten.h -- the complete version of the reference's flatten seam
// (hittable::populateVector -> hittableVector::build, hittable.h:32, hittablevector.h:27-31):
// walks the kept virtual hittable / material / texture classes into an SrtSceneDesc.
#ifndef SRT_HOST_FLATTEN_H
#define SRT_HOST_FLATTEN_H

#include <cstdint>
#include <map>
#include <vector>

#include "globals.h"

class material;
class texture;

class sceneFlattener {
 public:
  std::vector<SrtTriangleIn> triangles;
  std::vector<SrtSphereIn> spheres;
  std::vector<SrtPrimRef> prims;
  std::vector<SrtWorldItem> world;
  std::vector<SrtMaterialIn> materials;
  std::vector<shared_ptr<material>> materialPtrs;  // materials[i] was produced by materialPtrs[i] (hitRecord::matPtr)
  std::vector<SrtTextureIn> textures;
  std::vector<uint8_t> texels;
  std::vector<std::vector<SrtBvhNode>> trees;  // remapped prebuilt trees, kept alive for desc()

  int materialId(const shared_ptr<material>& m);  // material.h
  int textureId(const shared_ptr<texture>& t);    // texture.h

  int addSphere(const SrtSphereIn& s) {
    spheres.push_back(s);
    prims.push_back(SrtPrimRef{SRT_PRIM_SPHERE, (int32_t)spheres.size() - 1});
    return (int)prims.size() - 1;
  }
  int addTriangle(const SrtTriangleIn& t) {
    triangles.push_back(t);
    prims.push_back(SrtPrimRef{SRT_PRIM_TRIANGLE, (int32_t)triangles.size() - 1});
    return (int)prims.size() - 1;
  }

  SrtSceneDesc desc() const {
    SrtSceneDesc d;
    d.numTriangles = (int32_t)triangles.size(); d.triangles = triangles.data();
    d.numSpheres = (int32_t)spheres.size(); d.spheres = spheres.data();
    d.numPrims = (int32_t)prims.size(); d.prims = prims.data();
    d.numWorld = (int32_t)world.size(); d.world = world.data();
    d.numMaterials = (int32_t)materials.size(); d.materials = materials.data();
    d.numTextures = (int32_t)textures.size(); d.textures = textures.data();
    d.numTexelBytes = (int64_t)texels.size(); d.texels = texels.data();
    return d;
  }

 private:
  std::map<const void*, int> matIds_, texIds_;
  friend class material;
  friend class texture;
};

#endif
