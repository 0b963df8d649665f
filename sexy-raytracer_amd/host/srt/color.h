// color.h -- host mirror of writeColorTarget (color.h:25-41) for callers that post-process float
// accumulators themselves; the renderer's own tone map is srtResolveTiles on the device.
#ifndef SRT_HOST_COLOR_H
#define SRT_HOST_COLOR_H

#include <cmath>
#include <cstdint>

#include "globals.h"

inline void writeColorTarget(uint8_t* data, int x, int y, int w, int /*h*/, int bpp, color3f pixelColor, int numSamples) {
  float scale = 1.0f / numSamples;
  uint8_t* pixel = &(data[(y * w + x) * bpp]);
  for (int k = 0; k < 3; ++k) {
    float c = 256 * clamp(sqrtf(pixelColor(k) * scale), 0.0f, 0.999f);
    pixel[k] = (c == c) ? static_cast<uint8_t>(c) : 0;
  }
  pixel[3] = 255;
}

#endif
