// camera.h -- host mirror of camera.h:8-62: same ctor; the frame arithmetic (double tan etc.) is
// srtMakeCamera; getRay runs on the device (srt_kernels.hip cameraRay).
#ifndef SRT_HOST_CAMERA_H
#define SRT_HOST_CAMERA_H

#include "globals.h"

class camera {
 public:
  camera(vec3f eye, vec3f lookAt, vec3f up, float vFOV, float aspect, float aperture, float focusDist, float time0,
         float time1) {
    SrtCameraParams p{};
    for (int i = 0; i < 3; ++i) {
      p.eye[i] = eye(i);
      p.lookAt[i] = lookAt(i);
      p.up[i] = up(i);
    }
    p.vfovDegrees = vFOV; p.aspect = aspect; p.aperture = aperture; p.focusDist = focusDist;
    p.time0 = time0; p.time1 = time1;
    srtMakeCamera(&p, &pod);
  }
  const SrtCamera& data() const { return pod; }

 private:
  SrtCamera pod;
};

#endif
