// png.h -- PNG decode / encode for the scene-build path, replacing the stb calls the reference
// makes (stbi_load at texture.h:62,115; stbi_write_png at main.cpp:237).  zlib does the inflate /
// deflate.  Decodes 1- to 16-bit grey, grey+alpha, RGB, RGBA and palette PNGs (non-interlaced) and
// converts to the requested component count with stb_image's rules (luma = (77r+150g+29b)>>8,
// 16-bit -> high byte); for the 8-bit RGB / L files this path loads, the bytes are the file's own.
#ifndef SRT_HOST_PNG_H
#define SRT_HOST_PNG_H

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

inline uint32_t srtBe32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// returns malloc'ed w*h*reqComp bytes (caller frees), or nullptr.  *comp = components in the file.
inline uint8_t* srtPngLoad(const char* filename, int* w, int* h, int* comp, int reqComp) {
  FILE* fp = fopen(filename, "rb");
  if (!fp) return nullptr;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + n);
  fclose(fp);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return nullptr;
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    uint32_t len = srtBe32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    const uint8_t* data = &file[pos + 8];
    if (pos + 12 + (size_t)len > file.size()) return nullptr;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return nullptr;  // the fixed IHDR layout is read below
      width = srtBe32(data); height = srtBe32(data + 4);
      depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  const bool lowDepth = depth == 1 || depth == 2 || depth == 4;  // allowed for grey and palette only
  // header fields are untrusted: bound the image before any size arithmetic (32768^2 x 8 bytes < 2^43 fits
  // size_t; the int outputs *w, *h stay positive)
  if (width > 32768u || height > 32768u) return nullptr;
  if (!width || !height || interlace || !(depth == 8 || depth == 16 || lowDepth) || (lowDepth && ctype != 0 && ctype != 3) ||
      (ctype == 3 && depth == 16))
    return nullptr;
  int fileComp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!fileComp) return nullptr;
  const size_t bps = depth >= 8 ? depth / 8 : 1;
  const size_t stride = lowDepth ? ((size_t)width * depth + 7) / 8 : (size_t)width * fileComp * bps;
  const size_t bpp = lowDepth ? 1 : fileComp * bps;
  std::vector<uint8_t> raw((stride + 1) * height);
  uLongf rawLen = raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), idat.size()) != Z_OK || rawLen != raw.size()) return nullptr;
  // unfilter (PNG spec section 9)
  std::vector<uint8_t> img(stride * height);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t ft = raw[y * (stride + 1)];
    const uint8_t* in = &raw[y * (stride + 1) + 1];
    uint8_t* out = &img[y * stride];
    const uint8_t* up = y ? &img[(y - 1) * stride] : nullptr;
    for (size_t x = 0; x < stride; ++x) {
      int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
      int v = in[x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return nullptr;
      }
      out[x] = (uint8_t)v;
    }
  }
  if (lowDepth) {  // unpack 1/2/4-bit samples (MSB first); grey is scaled to 8 bits, palette indices stay
    std::vector<uint8_t> wide((size_t)width * height);
    const int mask = (1 << depth) - 1, scale = ctype == 0 ? 255 / mask : 1;
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) {
        const size_t bit = (size_t)x * depth;
        const int v = (img[y * stride + (bit >> 3)] >> (8 - depth - (bit & 7))) & mask;
        wide[(size_t)y * width + x] = (uint8_t)(v * scale);
      }
    img.swap(wide);
  }
  // to 8-bit RGBA-ish working pixels
  int srcComp = fileComp;
  std::vector<uint8_t> px;
  if (ctype == 3) {
    srcComp = trns.empty() ? 3 : 4;
    px.resize((size_t)width * height * srcComp);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
      unsigned idx = img[i];
      for (int k = 0; k < 3; ++k) px[i * srcComp + k] = idx * 3 + k < plte.size() ? plte[idx * 3 + k] : 0;
      if (srcComp == 4) px[i * 4 + 3] = idx < trns.size() ? trns[idx] : 255;
    }
  } else {
    px.resize((size_t)width * height * srcComp);
    for (size_t i = 0; i < px.size(); ++i) px[i] = img[i * bps];  // 16-bit: high byte
  }
  if (comp) *comp = srcComp;
  if (reqComp < 1 || reqComp > 4) reqComp = srcComp;
  uint8_t* out = (uint8_t*)malloc((size_t)width * height * reqComp);
  if (!out) return nullptr;
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    const uint8_t* s = &px[i * srcComp];
    uint8_t r, g, b, a = 255;
    if (srcComp <= 2) { r = g = b = s[0]; if (srcComp == 2) a = s[1]; }
    else { r = s[0]; g = s[1]; b = s[2]; if (srcComp == 4) a = s[3]; }
    uint8_t y = srcComp <= 2 ? s[0] : (uint8_t)(((r * 77) + (g * 150) + (29 * b)) >> 8);
    uint8_t* d = &out[i * reqComp];
    switch (reqComp) {
      case 1: d[0] = y; break;
      case 2: d[0] = y; d[1] = a; break;
      case 3: d[0] = r; d[1] = g; d[2] = b; break;
      default: d[0] = r; d[1] = g; d[2] = b; d[3] = a; break;
    }
  }
  *w = (int)width;
  *h = (int)height;
  return out;
}

// stbi_write_png(filename, w, h, comp, data, strideBytes) -> 1 on success (main.cpp:237)
inline int srtPngWrite(const char* filename, int w, int h, int comp, const void* data, int strideBytes) {
  if (w < 1 || h < 1 || comp < 1 || comp > 4 || !data) return 0;
  static const int ctypeOf[5] = {0, 0, 4, 2, 6};
  const size_t row = (size_t)w * comp;
  std::vector<uint8_t> raw((row + 1) * h);
  for (int y = 0; y < h; ++y) {
    raw[y * (row + 1)] = 0;
    memcpy(&raw[y * (row + 1) + 1], (const uint8_t*)data + (size_t)y * strideBytes, row);
  }
  uLongf zlen = compressBound(raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), raw.size(), 6) != Z_OK) return 0;
  FILE* fp = fopen(filename, "wb");
  if (!fp) return 0;
  auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
    uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                      (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    fwrite(hdr, 1, 8, fp);
    if (len) fwrite(d, 1, len, fp);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, d, len);
    uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    fwrite(c, 1, 4, fp);
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  fwrite(sig, 1, 8, fp);
  uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                      (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                      8, (uint8_t)ctypeOf[comp], 0, 0, 0};
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), (uint32_t)zlen);
  chunk("IEND", nullptr, 0);
  return fclose(fp) == 0;
}

// the names scene code written for the reference uses
inline uint8_t* stbi_load(const char* fn, int* w, int* h, int* comp, int reqComp) { return srtPngLoad(fn, w, h, comp, reqComp); }
inline int stbi_write_png(const char* fn, int w, int h, int comp, const void* data, int stride) { return srtPngWrite(fn, w, h, comp, data, stride); }

#endif
