// common.h -- what scene-building code takes from the reference's globals.h (constants, clamp, deg2rad, the
// process-global generator).  Scene-description side only: the hot path runs on the device.
#ifndef SRT_HOST_COMMON_H
#define SRT_HOST_COMMON_H

#include <cmath>
#include <limits>
#include <memory>

#include "../../../include/srt_hip.h"

using std::make_shared;
using std::shared_ptr;

// the three constants scene code uses (globals.h:13-15)
constexpr float infinity = std::numeric_limits<float>::infinity();
constexpr float epsilon = std::numeric_limits<float>::epsilon();
constexpr float pi = 3.1415926535897932385f;

// globals.h:17-28
inline float clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float deg2rad(float degrees) { return degrees * pi / 180.0f; }

// The reference draws everything from one default-seeded mt19937 (globals.h:30-43).  That generator
// lives inside libsrt_hip.so, so scene code calling these and the bvhNode build share one stream.
inline float randomFloat() { return srtHostRandomFloat(); }
inline float randomFloat(float lo, float hi) { return lo + (hi - lo) * srtHostRandomFloat(); }
inline int randomInt(int lo, int hi) {
  const float a = (float)lo, b = (float)(hi + 1);
  return static_cast<int>(a + (b - a) * srtHostRandomFloat());
}

#include "vec3.h"

#endif
