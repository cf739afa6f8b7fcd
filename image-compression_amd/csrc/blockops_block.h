// blockops_block.h -- compressed-domain block operations, one output block per lane:
// Pad's replicated blocks (internal/dxtc_compressor.cc:594-696, etc_compressor.cc:645-698), Downsample's
// decode -> 2x2 average -> re-encode (internal/compressor4x4_helper.h:264-391,594-636, pixel4x4.h:152-162,
// color_util.h:335-380) and the DXT1 -> ETC1 transcoder (internal/dxtc_to_etc_transcoder.cc:29-40).
#ifndef ICAMD_BLOCKOPS_BLOCK_H_
#define ICAMD_BLOCKOPS_BLOCK_H_

#include "decode_block.h"
#include "dxt_block.h"
#include "etc1_block.h"

namespace icamd {

enum { kPadColumn = 0, kPadRow = 1, kPadCorner = 2 };  // replicate pixel column 3 / pixel row 3 / pixel (3,3)

// DXT colour index bits (4 bytes, one per pixel row, 2 bits per pixel).
ICAMD_DEV uint32_t dxt_pad_color_bits(uint32_t bits, int kind) {
  if (kind == kPadColumn) return ((bits >> 6) & 0x03030303u) * 0x55u;  // CopyColumn3ColorBits per row
  if (kind == kPadRow) return (bits >> 24) * 0x01010101u;              // row 3 everywhere
  return (bits >> 30) * 0x55555555u;                                    // pixel (3,3) everywhere
}

// DXT5 alpha codes: lo24 = pixels 0..7, hi24 = pixels 8..15, 3 bits each.
ICAMD_DEV void dxt5_pad_alpha_codes(uint32_t &lo24, uint32_t &hi24, int kind) {
  if (kind == kPadColumn) {
    const uint32_t r0 = (lo24 >> 9) & 7u, r1 = (lo24 >> 21) & 7u, r2 = (hi24 >> 9) & 7u, r3 = (hi24 >> 21) & 7u;
    lo24 = r0 * 0x249u | (r1 * 0x249u) << 12;
    hi24 = r2 * 0x249u | (r3 * 0x249u) << 12;
  } else if (kind == kPadRow) {
    const uint32_t row3 = hi24 >> 12;
    lo24 = hi24 = row3 | row3 << 12;
  } else {
    lo24 = hi24 = (hi24 >> 21) * 0x249249u;
  }
}

// CreateSolidBlock (etc_compressor.cc:595-617): differential mode, base = colour >> 3, everything else zero.
ICAMD_DEV Out8 etc1_solid_block(uint32_t rgb) {
  const uint32_t hi = 2u | (bfe(rgb, 0, 8) >> 3) << 27 | (bfe(rgb, 8, 8) >> 3) << 19 | (bfe(rgb, 16, 8) >> 3) << 11;
  Out8 o = { perm(0u, hi, 0x00010203u), 0u };
  return o;
}

// ETC pad block from the decoded source block.
ICAMD_DEV Out8 etc1_pad_block(uint32_t w0, uint32_t w1, int kind, uint32_t strategy) {
  uint32_t src[16], px[16];
  decode_etc1(w0, w1, src);
  if (kind == kPadCorner) return etc1_solid_block(src[15]);
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int x = 0; x < 4; ++x) px[4 * y + x] = kind == kPadColumn ? src[4 * y + 3] : src[12 + x];
  }
  return encode_etc1_block(px, strategy);
}

// Average4ColorsFast (color_util.h:335-380) on packed pixels: per channel (a+b+c+d)/4.
ICAMD_DEV uint32_t average4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const uint32_t m = 0x00ff00ffu;
  const uint32_t rb = (((a & m) + (b & m) + (c & m) + (d & m)) >> 2) & m;
  const uint32_t ga = ((((a >> 8) & m) + ((b >> 8) & m) + ((c >> 8) & m) + ((d >> 8) & m)) >> 2) & m;
  return rb | ga << 8;
}

// StoreDownsampledPixels4x4 (pixel4x4.h:152-162): the 2x2 averages of src go to dst's 2x2 quadrant (tr, tc).
ICAMD_DEV void store_downsampled(const uint32_t src[16], int tr, int tc, uint32_t dst[16]) {
  ICAMD_UNROLL
  for (int r = 0; r < 2; ++r) {
    ICAMD_UNROLL
    for (int c = 0; c < 2; ++c)
      dst[4 * (tr + r) + tc + c] = average4(src[8 * r + 2 * c], src[8 * r + 2 * c + 1], src[8 * r + 4 + 2 * c],
                                            src[8 * r + 4 + 2 * c + 1]);
  }
}

// Decode any block (DXT1 / DXT5 / ETC1, codec ids of ic_amd.h) to packed pixels, no red/blue swap
// (Downsample and the transcoder always pass swap = false).
template <int CODEC>
ICAMD_DEV void decode_any(const uint32_t *w, uint32_t px[16]) {
  if (CODEC == 1) {
    decode_dxt_colors(w[2], w[3], false, true, px);
    decode_dxt5_alpha(w[0], w[1], px);
  } else if (CODEC == 0) {
    decode_dxt_colors(w[0], w[1], false, false, px);
  } else {
    decode_etc1(w[0], w[1], px);
  }
}

// Encode packed pixels with any codec, no swap (what Downsample's encode functor does).  out: 2 or 4 dwords.
template <int CODEC>
ICAMD_DEV void encode_any(const uint32_t px[16], uint32_t strategy, BlockStash &stash, uint32_t *out) {
  if (CODEC == 1) {
    const Out8 a = encode_dxt5_alpha_block(px, false);
    const Out8 c = encode_dxt_color_block(px, false, true, stash);
    out[0] = a.lo; out[1] = a.hi; out[2] = c.lo; out[3] = c.hi;
  } else if (CODEC == 0) {
    const Out8 c = encode_dxt_color_block(px, false, false, stash);
    out[0] = c.lo; out[1] = c.hi;
  } else {
    const Out8 c = encode_etc1_block(px, strategy);
    out[0] = c.lo; out[1] = c.hi;
  }
}

}  // namespace icamd
#endif  // ICAMD_BLOCKOPS_BLOCK_H_
