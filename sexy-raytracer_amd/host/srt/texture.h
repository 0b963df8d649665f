// texture.h -- host mirror of the reference's texture classes (texture.h:13-153).  A texture knows how to describe
// itself to the flattener; value(u, v, p) keeps the reference's signature (texture.h:15) and is answered by the device
// (csrc/srt_path.h texValue, through srtScatterRays: shade_device.h) -- there is no host copy of the lookup arithmetic.
#ifndef SRT_HOST_TEXTURE_H
#define SRT_HOST_TEXTURE_H

#include <cstdlib>
#include <iostream>

#include "flatten.h"
#include "png.h"

struct srtShadeSession;

class texture {
 public:
  virtual ~texture() {}
  virtual int populate(sceneFlattener& f) const = 0;  // appends an SrtTextureIn, returns its id
  // texture.h:15.  One device round trip per call (shade_device.h).
  virtual color3f value(float u, float v, const vec3f& p) const;

 private:
  mutable shared_ptr<srtShadeSession> shadeSession_;
};

class solidColor : public texture {  // texture.h:18-32
 public:
  solidColor() {}
  solidColor(color3f c) : colorValue(c) {}
  solidColor(float r, float g, float b) : solidColor(color3f(r, g, b)) {}
  int populate(sceneFlattener& f) const override {
    SrtTextureIn t{};
    t.kind = SRT_TEX_SOLID;
    t.even = t.odd = -1;
    for (int i = 0; i < 3; ++i) t.color[i] = colorValue(i);
    f.textures.push_back(t);
    return (int)f.textures.size() - 1;
  }

 private:
  color3f colorValue;
};

class checker : public texture {  // texture.h:34-52
 public:
  checker() {}
  checker(shared_ptr<texture> _even, shared_ptr<texture> _odd) : odd(_odd), even(_even) {}
  checker(color3f c1, color3f c2) : odd(make_shared<solidColor>(c2)), even(make_shared<solidColor>(c1)) {}
  int populate(sceneFlattener& f) const override {
    SrtTextureIn t{};
    t.kind = SRT_TEX_CHECKER;
    t.even = f.textureId(even);
    t.odd = f.textureId(odd);
    f.textures.push_back(t);
    return (int)f.textures.size() - 1;
  }

 private:
  shared_ptr<texture> odd, even;
};

class imagePNG : public texture {  // texture.h:109-153; stbi_load -> srtPngLoad (png.h)
 public:
  imagePNG(int bytesPP) : data(nullptr), width(0), height(0), bpp(bytesPP), bytesPerScanline(0) {}
  imagePNG(const char* filename, int bytesPP) : bpp(bytesPP) {
    int componentsPP = bpp;
    data = srtPngLoad(filename, &width, &height, &componentsPP, componentsPP);
    if (!data) {
      std::cerr << "ERROR: Could not load image file '" << filename << "'\n";  // texture.h:117-120
      width = height = 0;
    }
    bytesPerScanline = bpp * width;
  }
  imagePNG(const imagePNG&) = delete;
  ~imagePNG() { free(data); }
  int populate(sceneFlattener& f) const override {
    SrtTextureIn t{};
    t.kind = SRT_TEX_IMAGE;
    t.even = t.odd = -1;
    t.bpp = bpp;
    if (data) {
      t.width = width;
      t.height = height;
      t.texelOffset = (int64_t)f.texels.size();
      f.texels.insert(f.texels.end(), data, data + (size_t)width * height * bpp);
    }
    f.textures.push_back(t);
    return (int)f.textures.size() - 1;
  }

 protected:
  uint8_t* data;
  int width, height, bpp;
  int bytesPerScanline;
};

class image3bpp : public imagePNG {  // texture.h:54-107: the bpp = 3 case
 public:
  const static int bpp = 3;
  image3bpp() : imagePNG(3) {}
  image3bpp(const char* filename) : imagePNG(filename, 3) {}
};

inline int sceneFlattener::textureId(const shared_ptr<texture>& t) {
  if (!t) return -1;
  auto it = texIds_.find(t.get());
  if (it != texIds_.end()) return it->second;
  int id = t->populate(*this);
  texIds_[t.get()] = id;
  return id;
}

#endif
