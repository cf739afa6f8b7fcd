// decode_block.h -- DXT1 / DXT5 / ETC1 (and, as an extension, PVRTC1 2bpp) block decoders, one block per lane
// ("next" row 8f.1).
// Bit-exact with DecodeDxt1Block / DecodeDxt5Block (internal/dxtc_compressor.cc:167-267) and
// Etc1BlockDecoder (internal/etc_compressor.cc:198-289).  Decoded pixels are returned as dwords in the
// OUTPUT memory byte order (byte0 = first channel written, byte3 = alpha for DXT5).
#ifndef ICAMD_DECODE_BLOCK_H_
#define ICAMD_DECODE_BLOCK_H_

#include "ic_device.h"
#include "pvrtc_block.h"  // channel-pair helpers, bilerp_pair

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

// ---- PVRTC1 2bpp decoder: EXTENSION, PARITY UNPINNED (the reference has none, pvrtc.cc:669-672).  Written from the
// encoder's own rules; the plain-C statement of the same rules, with the citations, is oracle/ic_oracle.c
// (pvrtc_decode_image), which the GPU tests compare this against.

// EncodeColors (pvrtc.cc:356-388) undone: the two stored colours as RGBA dwords, fields widened by bit replication
// exactly like ApplyBitDepthReduction (pvrtc.cc:93-106), so decoder colours == the encoder's reduced colours.
ICAMD_DEV uint32_t pvrtc_rep5(uint32_t v) { return v << 3 | v >> 2; }
ICAMD_DEV uint32_t pvrtc_rep4(uint32_t v) { return v << 4 | v; }
ICAMD_DEV uint32_t pvrtc_rep3(uint32_t v) { return v << 5 | v << 2 | v >> 1; }
ICAMD_DEV void pvrtc_unpack_colors(uint32_t c, uint32_t &col_a, uint32_t &col_b) {
  col_a = (c & (1u << 15)) ? (pvrtc_rep5(bfe(c, 10, 5)) | pvrtc_rep5(bfe(c, 5, 5)) << 8 | pvrtc_rep4(bfe(c, 1, 4)) << 16 | 0xff000000u)
                           : (pvrtc_rep4(bfe(c, 8, 4)) | pvrtc_rep4(bfe(c, 4, 4)) << 8 | pvrtc_rep3(bfe(c, 1, 3)) << 16 |
                              pvrtc_rep3(bfe(c, 12, 3)) << 24);
  col_b = (c & (1u << 31)) ? (pvrtc_rep5(bfe(c, 26, 5)) | pvrtc_rep5(bfe(c, 21, 5)) << 8 | pvrtc_rep5(bfe(c, 16, 5)) << 16 | 0xff000000u)
                           : (pvrtc_rep4(bfe(c, 24, 4)) | pvrtc_rep4(bfe(c, 20, 4)) << 8 | pvrtc_rep4(bfe(c, 16, 4)) << 16 |
                              pvrtc_rep3(bfe(c, 28, 3)) << 24);
}

// Weight of colour B (in eighths: modulation 0..3 = 0, 3, 5, 8; ApplyModulation, pvrtc.cc:120-144) that a block
// STORES for its pixel (x, y); 2BPP blocks store nothing for the odd checkerboard pixels (caller interpolates).
ICAMD_DEV uint32_t pvrtc_stored_weight(uint32_t data, bool two_bpp, uint32_t x, uint32_t y) {
  if (!two_bpp) return ((data >> (8u * y + x)) & 1u) * 8u;          // 1BPP: bit 8y+x, 0 -> A, 1 -> B
  const uint32_t pos = 2u * (4u * y + (x >> 1)), s = (data >> pos) & 3u;
  if (pos == 0u || pos == 20u) return (s >> 1) * 8u;                 // the low bit is a sub-mode flag, pvrtc.cc:474-487
  return (0x08050300u >> (8u * s)) & 0xffu;
}

// One 8x4 block.  words[d] = {modulation word, colour word} of the blocks at (dx, dy) = nb index 3*(dy+1)+(dx+1),
// toroidal wrap applied by the caller.  px[8*y + x] = decoded R,G,B,A dword.
ICAMD_DEV void decode_pvrtc2_block(const uint32_t mod[9], const uint32_t col[9], uint32_t px[32]) {
  PvrtcAB nb[3][3];
  ICAMD_UNROLL
  for (int i = 0; i < 9; ++i) {
    uint32_t a, b;
    pvrtc_unpack_colors(col[i], a, b);
    PvrtcAB e = { pair_rb(a), pair_ga(a), pair_rb(b), pair_ga(b) };
    nb[i / 3][i % 3] = e;
  }
  const uint32_t data = mod[4];
  const bool two = (col[4] & 1u) != 0u;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int x = 0; x < 8; ++x) {
      uint32_t w;
      if (!two || ((x ^ y) & 1) == 0) {
        w = pvrtc_stored_weight(data, two, (uint32_t)x, (uint32_t)y);
      } else {
        // the four orthogonal neighbours all store a weight (even checkerboard parity, or a 1BPP block)
        const int lb = x == 0 ? 3 : 4, rb = x == 7 ? 5 : 4, ub = y == 0 ? 1 : 4, db = y == 3 ? 7 : 4;
        const uint32_t l = pvrtc_stored_weight(mod[lb], (col[lb] & 1u) != 0u, (uint32_t)((x + 7) & 7), (uint32_t)y);
        const uint32_t r = pvrtc_stored_weight(mod[rb], (col[rb] & 1u) != 0u, (uint32_t)((x + 1) & 7), (uint32_t)y);
        const uint32_t u = pvrtc_stored_weight(mod[ub], (col[ub] & 1u) != 0u, (uint32_t)x, (uint32_t)((y + 3) & 3));
        const uint32_t d = pvrtc_stored_weight(mod[db], (col[db] & 1u) != 0u, (uint32_t)x, (uint32_t)((y + 1) & 3));
        w = !(data & 1u) ? (l + r + u + d + 2u) >> 2 : (data & (1u << 20)) ? (u + d + 1u) >> 1 : (l + r + 1u) >> 1;
      }
      // GetInterpolatedColor2BPP (pvrtc.cc:208-237): 2x2 sources and weights of pixel (x, y)
      const int x0 = x < 4 ? 0 : 1, y0 = y < 2 ? 0 : 1;
      const uint32_t xw = (uint32_t)((x + 4) & 7), yw = (uint32_t)((y + 2) & 3);
      const PvrtcAB &c00 = nb[y0][x0], &c01 = nb[y0][x0 + 1], &c10 = nb[y0 + 1][x0], &c11 = nb[y0 + 1][x0 + 1];
      const uint32_t a_rb = bilerp_pair(c00.a_rb, c01.a_rb, c10.a_rb, c11.a_rb, xw, yw);
      const uint32_t a_ga = bilerp_pair(c00.a_ga, c01.a_ga, c10.a_ga, c11.a_ga, xw, yw);
      const uint32_t b_rb = bilerp_pair(c00.b_rb, c01.b_rb, c10.b_rb, c11.b_rb, xw, yw);
      const uint32_t b_ga = bilerp_pair(c00.b_ga, c01.b_ga, c10.b_ga, c11.b_ga, xw, yw);
      const uint32_t rbv = (((8u - w) * a_rb + w * b_rb) >> 3) & 0x00ff00ffu;
      const uint32_t gav = (((8u - w) * a_ga + w * b_ga) >> 3) & 0x00ff00ffu;
      px[8 * y + x] = unpair(rbv, gav);
    }
  }
}

}  // namespace icamd
#endif  // ICAMD_DECODE_BLOCK_H_
