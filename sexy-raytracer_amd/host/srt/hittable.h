// hittable.h -- host mirror of hittable.h:24-33.  The virtual surface kept on the host is what
// scene construction needs: boundingBox (BVH build), populate (the complete populateVector) and
// selectBvhAxis.  hit() runs on the device; hitRecord results of fixed rays come back as SrtHit.
#ifndef SRT_HOST_HITTABLE_H
#define SRT_HOST_HITTABLE_H

#include "aabb.h"
#include "flatten.h"
#include "globals.h"

class material;

class hittable {
 public:
  virtual ~hittable() {}
  virtual bool boundingBox(float time0, float time1, aabb& outputBox) const = 0;
  virtual int selectBvhAxis() const { return randomInt(0, 2); }  // hittable.h:29
  // the reference's populateVector(hittableVector) (hittable.h:32), made complete: appends this
  // object to the flattener as primitive(s) / world item and returns the first primitive index,
  // or -1 for containers that added world items instead.
  virtual int populate(sceneFlattener& f) const = 0;
  virtual bool isPrimitive() const { return true; }
};

#endif
