// png_tool: decodes a PNG with the host layer's loader (srt/png.h, the stbi_load replacement) and writes
// the raw bytes, or re-encodes them with the writer; used by tests/test_host_cpp.py.
//   srt_png_tool decode <in.png> <req_comp> <out.raw>     (prints "w h comp")
//   srt_png_tool encode <in.raw> <w> <h> <comp> <out.png>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "srt/png.h"

int main(int argc, char** argv) {
  if (argc >= 5 && !strcmp(argv[1], "decode")) {
    int w = 0, h = 0, comp = 0, req = atoi(argv[3]);
    uint8_t* px = stbi_load(argv[2], &w, &h, &comp, req);
    if (!px) {
      fprintf(stderr, "decode failed\n");
      return 1;
    }
    FILE* f = fopen(argv[4], "wb");
    if (!f) return 1;
    fwrite(px, 1, (size_t)w * h * (req ? req : comp), f);
    fclose(f);
    free(px);
    printf("%d %d %d\n", w, h, comp);
    return 0;
  }
  if (argc >= 7 && !strcmp(argv[1], "encode")) {
    int w = atoi(argv[3]), h = atoi(argv[4]), comp = atoi(argv[5]);
    std::vector<uint8_t> buf((size_t)w * h * comp);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) return 1;
    fclose(f);
    return stbi_write_png(argv[6], w, h, comp, buf.data(), w * comp) ? 0 : 1;
  }
  return 2;
}
