// TEST INFRASTRUCTURE ONLY.  Compiles the per-block device math of
// image-compression_amd/csrc/*_block.h for the HOST (g++ -DICAMD_HOST_EMULATION: the gfx950
// instruction wrappers in ic_device.h are replaced by plain-C equivalents) so that the exact
// kernel arithmetic can be checked against the oracle in the CPU-only test tier.
// Never linked into libic_amd.so; the product has no CPU path.
#ifndef ICAMD_HOST_EMULATION
#error "build with -DICAMD_HOST_EMULATION"
#endif
#include <algorithm>
#include <cstring>

#include "dxt_block.h"
#include "etc1_block.h"
#include "pvrtc_block.h"
#include "decode_block.h"

using namespace icamd;

extern "C" int emul_encode(int codec, int strategy, int comps, int swap, uint32_t h, uint32_t w, uint32_t gh,
                           uint32_t gw, uint32_t stride, const uint8_t *src, uint8_t *out) {
  if (codec == 3) return emul_pvrtc2(src, w, out);
  const uint32_t rows = (std::max(h, gh) + 3) / 4, cols = (std::max(w, gw) + 3) / 4;
  for (uint32_t br = 0; br < rows; ++br)
    for (uint32_t bc = 0; bc < cols; ++bc) {
      uint32_t px[16];
      if (comps == 4) load_block<4>(src, h, w, stride, br * 4, bc * 4, px);
      else load_block<3>(src, h, w, stride, br * 4, bc * 4, px);
      uint8_t *o = out + ((size_t)br * cols + bc) * (codec == 1 ? 16 : 8);
      BlockStash stash;
      if (codec == 0) {
        Out8 c = encode_dxt_color_block(px, swap != 0, false, stash);
        memcpy(o, &c, 8);
      } else if (codec == 1) {
        const bool one_pixel = bc * 4 >= w && br * 4 >= h;
        Out8 a = encode_dxt5_alpha_block(px, one_pixel);
        Out8 c = encode_dxt_color_block(px, swap != 0, true, stash);
        memcpy(o, &a, 8);
        memcpy(o + 8, &c, 8);
      } else {
        Out8 c = encode_etc1_block(px, (uint32_t)strategy);
        memcpy(o, &c, 8);
      }
    }
  return 1;
}

extern "C" int emul_decode(int codec, int swap, uint32_t h, uint32_t w, uint32_t pad, const uint8_t *blocks, uint8_t *out) {
  const int comps = codec == 1 ? 4 : 3;
  const uint32_t rows = (h + 3) / 4, cols = (w + 3) / 4;
  const size_t stride = (size_t)w * comps + pad;
  for (uint32_t br = 0; br < rows; ++br)
    for (uint32_t bc = 0; bc < cols; ++bc) {
      const uint32_t *b = reinterpret_cast<const uint32_t *>(blocks + ((size_t)br * cols + bc) * (codec == 1 ? 16 : 8));
      uint32_t px[16];
      if (codec == 1) { decode_dxt_colors(b[2], b[3], swap != 0, true, px); decode_dxt5_alpha(b[0], b[1], px); }
      else if (codec == 0) decode_dxt_colors(b[0], b[1], swap != 0, false, px);
      else decode_etc1(b[0], b[1], px);
      for (uint32_t y = 0; y < 4 && br * 4 + y < h; ++y)
        for (uint32_t x = 0; x < 4 && bc * 4 + x < w; ++x)
          memcpy(out + (br * 4 + y) * stride + (size_t)(bc * 4 + x) * comps, &px[4 * y + x], comps);
    }
  return 1;
}
