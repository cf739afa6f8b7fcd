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
