// decode_block.h -- DXT1 / DXT5 / ETC1 block decoders, one block per lane ("next" row 8f.1).
// Bit-exact with DecodeDxt1Block / DecodeDxt5Block (internal/dxtc_compressor.cc:167-267) and
// Etc1BlockDecoder (internal/etc_compressor.cc:198-289).  Decoded pixels are returned as dwords in the
// OUTPUT memory byte order (byte0 = first channel written, byte3 = alpha for DXT5).
#ifndef ICAMD_DECODE_BLOCK_H_
#define ICAMD_DECODE_BLOCK_H_

#include "ic_device.h"

namespace icamd {

// ExtendToRgb888 (color_util.h:232-236) of a packed 565 colour -> 0x00BBGGRR
ICAMD_DEV uint32_t expand565_packed(uint32_t c) {
  const uint32_t r = c >> 11, g = (c >> 5) & 63u, b = c & 31u;
  return ((r << 3) | (r >> 2)) | ((g << 2) | (g >> 4)) << 8 | ((b << 3) | (b >> 2)) << 16;
}

// per-channel (wa*a + wb*b) / (wa + wb) on 0x00BBGGRR colours (CombineUint8Fast, color_util.h:288-291)
ICAMD_DEV uint32_t blend_packed(uint32_t a, uint32_t b, uint32_t wa, uint32_t wb) {
  uint32_t o = 0;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t s = wa * bfe(a, 8 * ch, 8) + wb * bfe(b, 8 * ch, 8);
    o |= (wa + wb == 3u ? div3(s) : s >> 1) << (8 * ch);
  }
  return o;
}

// blk: the 8 colour bytes as two little-endian dwords.  always4 = DXT5's colour block.
ICAMD_DEV void decode_dxt_colors(uint32_t w0, uint32_t bits, bool swap, bool always4, uint32_t px[16]) {
  const uint32_t c0 = w0 & 0xffffu, c1 = w0 >> 16;
  uint32_t col[4];
  col[0] = expand565_packed(c0);
  col[1] = expand565_packed(c1);
  if (swap) {  // SwapRedAndBlue (dxtc.cc:179-182): stored R goes to the third byte
    col[0] = perm(col[0], col[0], 0x03000102u);
    col[1] = perm(col[1], col[1], 0x03000102u);
  }
  if (c0 == c1) {
    col[2] = col[3] = col[1];
  } else if (always4 || c0 > c1) {
    col[2] = blend_packed(col[0], col[1], 2, 1);
    col[3] = blend_packed(col[0], col[1], 1, 2);
  } else {
    col[2] = blend_packed(col[0], col[1], 1, 1);
    col[3] = 0;
  }
  ICAMD_UNROLL
  for (int p = 0; p < 16; ++p) {
    const uint32_t code = (bits >> (2 * p)) & 3u;
    px[p] = code == 0u ? col[0] : code == 1u ? col[1] : code == 2u ? col[2] : col[3];
  }
}

// DXT5 alpha block (two dwords) -> alpha into byte 3 of px[] (DecodeAlphaValues, dxtc.cc:195-217).
ICAMD_DEV void decode_dxt5_alpha(uint32_t w0, uint32_t w1, uint32_t px[16]) {
  const uint32_t a0 = w0 & 0xffu, a1 = (w0 >> 8) & 0xffu;
  uint32_t t[8];
  t[0] = a0; t[1] = a1;
  if (a0 > a1) {
    ICAMD_UNROLL
    for (int k = 1; k <= 6; ++k) t[1 + k] = div7((uint32_t)(7 - k) * a0 + (uint32_t)k * a1);
  } else {
    ICAMD_UNROLL
    for (int k = 1; k <= 4; ++k) t[1 + k] = div5((uint32_t)(5 - k) * a0 + (uint32_t)k * a1);
    t[6] = 0u; t[7] = 255u;
  }
  const uint32_t lo24 = w0 >> 16 | (w1 & 0xffu) << 16, hi24 = w1 >> 8;  // codes of pixels 0-7 / 8-15
  ICAMD_UNROLL
  for (int p = 0; p < 16; ++p) {
    const uint32_t code = ((p < 8 ? lo24 : hi24) >> (3 * (p & 7))) & 7u;
    uint32_t a = t[0];
    ICAMD_UNROLL
    for (int k = 1; k < 8; ++k) a = code == (uint32_t)k ? t[k] : a;
    px[p] = (px[p] & 0x00ffffffu) | a << 24;
  }
}

ICAMD_DEV uint32_t clamp255(int32_t v) { return (uint32_t)imin(imax(v, 0), 255); }

// w0, w1: the 8 block bytes as little-endian dwords (memory holds hi word then lo word, big-endian).
ICAMD_DEV void decode_etc1(uint32_t w0, uint32_t w1, uint32_t px[16]) {
  const uint32_t hi = perm(0u, w0, 0x00010203u), lo = perm(0u, w1, 0x00010203u);
  const bool flip = hi & 1u, diff = hi & 2u;
  const uint32_t cw0 = (hi >> 5) & 7u, cw1 = (hi >> 2) & 7u;
  int32_t base[2][3];
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    if (diff) {
      const int32_t b5 = (int32_t)((hi >> (27 - 8 * ch)) & 31u);
      const int32_t d3 = (int32_t)((hi >> (24 - 8 * ch)) & 7u);
      const int32_t s5 = b5 + (d3 >= 4 ? d3 - 8 : d3);  // ExtendSignBit, bit_util.h:61-69
      base[0][ch] = (b5 << 3) | ((b5 >> 2) & 7);        // Extend5Bit, color_util.h:200-202
      base[1][ch] = (s5 << 3) | ((s5 >> 2) & 7);
    } else {
      const int32_t q0 = (int32_t)((hi >> (28 - 8 * ch)) & 15u), q1 = (int32_t)((hi >> (24 - 8 * ch)) & 15u);
      base[0][ch] = q0 * 17;
      base[1][ch] = q1 * 17;
    }
  }
  // modifier magnitudes {a, b} of each sub-block's codeword (etc.cc:101-110); index k: +a, +b, -a, -b
  const uint32_t tab_a[2] = { 2u | 5u << 8 | 9u << 16 | 13u << 24, 18u | 24u << 8 | 33u << 16 | 47u << 24 };
  const uint32_t tab_b[2] = { 8u | 17u << 8 | 29u << 16 | 42u << 24, 60u | 80u << 8 | 106u << 16 | 183u << 24 };
  int32_t ma[2], mb[2];
  ma[0] = (int32_t)bfe(cw0 < 4u ? tab_a[0] : tab_a[1], 8 * (cw0 & 3u), 8);
  mb[0] = (int32_t)bfe(cw0 < 4u ? tab_b[0] : tab_b[1], 8 * (cw0 & 3u), 8);
  ma[1] = (int32_t)bfe(cw1 < 4u ? tab_a[0] : tab_a[1], 8 * (cw1 & 3u), 8);
  mb[1] = (int32_t)bfe(cw1 < 4u ? tab_b[0] : tab_b[1], 8 * (cw1 & 3u), 8);
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int x = 0; x < 4; ++x) {
      const int p = 4 * x + y;  // etc.cc:131-137
      const uint32_t k = ((lo >> p) & 1u) | ((lo >> (p + 16)) & 1u) << 1;
      const bool second = flip ? y >= 2 : x >= 2;
      const int32_t mag = (k & 1u) ? (second ? mb[1] : mb[0]) : (second ? ma[1] : ma[0]);
      const int32_t m = (k & 2u) ? -mag : mag;
      uint32_t c = 0;
      ICAMD_UNROLL
      for (int ch = 0; ch < 3; ++ch) c |= clamp255((second ? base[1][ch] : base[0][ch]) + m) << (8 * ch);
      px[4 * y + x] = c;
    }
  }
}

}  // namespace icamd
#endif  // ICAMD_DECODE_BLOCK_H_
