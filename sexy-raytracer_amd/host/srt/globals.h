// globals.h -- host mirror of the reference's globals.h (constants, clamp, deg2rad, the
// process-global generator).  Scene-description side only: the hot path runs on the device.
#ifndef SRT_HOST_GLOBALS_H
#define SRT_HOST_GLOBALS_H

#include <cmath>
#include <limits>
#include <memory>

#include "../../../include/srt_hip.h"

using std::make_shared;
using std::shared_ptr;

const float infinity = std::numeric_limits<float>::infinity();  // globals.h:13
const float epsilon = std::numeric_limits<float>::epsilon();    // globals.h:14
const float pi = 3.1415926535897932385f;                        // globals.h:15

inline float clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }  // globals.h:17-24
inline float deg2rad(float degrees) { return degrees * pi / 180.0f; }                          // globals.h:26-28

// globals.h:30-43: one default-seeded mt19937 for the whole process.  It lives inside
// libsrt_hip.so so that scene code and the bvhNode build draw from the same stream.
inline float randomFloat() { return srtHostRandomFloat(); }
inline float randomFloat(float lo, float hi) { return lo + (hi - lo) * randomFloat(); }
inline int randomInt(int lo, int hi) { return static_cast<int>(randomFloat((float)lo, (float)(hi + 1))); }

#include "vec3.h"

#endif
