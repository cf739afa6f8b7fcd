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

// One 8x4 block.  mod[d] / col[d] = modulation word / colour word of the blocks at (dx, dy) = index 3*(dy+1)+(dx+1),
// toroidal wrap applied by the caller.  px[8*y + x] = decoded R,G,B,A dword.
//
// Weights (of colour B, in eighths) are handled as BYTES, four pixels per dword (r03):
//   * the block's own stored weights of a pixel row are expanded from the row's 8 modulation bits by shifts and masks
//     (1BPP: bit -> 0 / 8; 2BPP: the four 2-bit samples s -> 3 s - (s >> 1) = {0, 3, 5, 8}, placed at the even-parity
//     pixels; the two samples whose low bit is a sub-mode flag -- bit positions 0 and 20, pvrtc.cc:474-487 -- keep only
//     their high bit);
//   * the weights a 2BPP block does NOT store (odd checkerboard parity) are the mode's average of the orthogonal
//     neighbours' stored weights, computed on the packed bytes: (l + r + u + d + 2) >> 2 needs no carries between bytes
//     (4 * 8 + 2 < 256); only the 12 border weights come from the neighbour blocks' words, one by one.
// Colours follow the encoder's separable walk (pvrtc_row_mods_v): the three block columns are blended vertically once
// per pixel row, then P(xw + 1) = P(xw) + (VR - VL) steps through each half row with P = 256 * colour on 16-bit lanes,
// whose high bytes are the reference's truncated 8-bit channels (pvrtc.cc:228-236); the final blend
// ((8 - w) A + w B) >> 3 (ApplyModulation, pvrtc.cc:120-144) is three packed 16-bit instructions per channel pair.
// A block's two colours as the four channel pairs the interpolation works on: a_rb, a_ga, b_rb, b_ga.
ICAMD_DEV void pvrtc_expand_colors(uint32_t colour_word, uint32_t out[4]) {
  uint32_t a, b;
  pvrtc_unpack_colors(colour_word, a, b);
  out[0] = pair_rb(a); out[1] = pair_ga(a); out[2] = pair_rb(b); out[3] = pair_ga(b);
}

// v_pk_mad_u16 with ONE 16-bit lane of w feeding both products (the compiler folds the broadcast into op_sel):
// lane k of the result = a.lane[k] * w.lane[L] + c.lane[k]  (mod 2^16)
template <int L>
ICAMD_DEV uint32_t pk_mad_u16_lane(uint32_t a, uint32_t w, uint32_t c) {
#if defined(ICAMD_HOST_EMULATION)
  const uint32_t ww = (w >> (16 * L)) & 0xffffu;
  return pk_mad_u16(a, ww | ww << 16, c);
#else
  const icamd_dxt_us2 v = as_us2(w);
  return from_us2(as_us2(a) * __builtin_shufflevector(v, v, L, L) + as_us2(c));
#endif
}

// The eight weights one block STORES for its pixel row y (bits = the row's byte of the modulation word): out[h] byte
// x & 3 of half h = x >> 2; 2BPP rows hold 0 at the pixels the block stores nothing for (odd checkerboard parity).
//   1BPP: bit x -> 8 * bit.  nibble * 0x204081 puts bit i at 8 i (the sixteen partial products land on distinct bits:
//         i + 7 k for i, k in 0..3).
//   2BPP: sample j (2 bits) belongs to pixel 2 j + (y & 1); {0, 3, 5, 8}[s] = 3 s - (s >> 1); the two samples whose low
//         bit is a sub-mode flag (bit positions 0 and 20, pvrtc.cc:474-487) keep only their high bit: (s >> 1) * 8.
template <int Y>
ICAMD_DEV void pvrtc_stored_row(uint32_t data, bool two, uint32_t out[2]) {
  // (the nibble enters pre-multiplied by 8 so that both factors stay below 2^24: v_mul_u32_u24, not the quarter-rate full multiply)
  const uint32_t n0 = (Y == 0 ? data << 3 : data >> (8 * Y - 3)) & 0x78u, n1 = (data >> (8 * Y + 1)) & 0x78u;
  const uint32_t one0 = umad24(n0, 0x204081u, 0u) & 0x08080808u;
  const uint32_t one1 = umad24(n1, 0x204081u, 0u) & 0x08080808u;
  const uint32_t bits = bfe(data, 8 * Y, 8);
  const uint32_t t = bits | bits << 12;
  const uint32_t S = (t | t << 6) & 0x03030303u;                                     // byte j = s_j
  uint32_t w4 = 3u * S - ((S >> 1) & 0x01010101u);                                   // byte j = {0, 3, 5, 8}[s_j]
  if (Y == 0) w4 = (w4 & 0xffffff00u) | ((S & 0x00000002u) << 2);
  if (Y == 2) w4 = (w4 & 0xff00ffffu) | ((S & 0x00020000u) << 2);
  const uint32_t two0 = perm(0u, w4, (Y & 1) ? 0x010c000cu : 0x0c010c00u);
  const uint32_t two1 = perm(0u, w4, (Y & 1) ? 0x030c020cu : 0x0c030c02u);
  out[0] = two ? two0 : one0;
  out[1] = two ? two1 : one1;
}

// C[r][c][v]: expanded colours of the 3 x 3 block neighbourhood.  mod / col are indexed like the neighbourhood
// (3 * (dy + 1) + dx + 1) but only the block's own words (4) and its four orthogonal neighbours' (1, 3, 5, 7) are read.
// emit(y, row): the eight decoded pixels of pixel row y (R,G,B,A dwords), called as soon as the row is complete -- the
// kernels store from there, so a block's eight stores leave spread over its arithmetic instead of in one burst at its end.
template <typename Emit>
ICAMD_DEV void decode_pvrtc2_block_rows(const uint32_t C[3][3][4], const uint32_t mod[9], const uint32_t col[9], Emit emit) {
  const uint32_t data = mod[4];
  const bool two = (col[4] & 1u) != 0u;

  // ---- stored weights of the block's own pixels and of the rows above / below it (r05: whole rows, one routine)
  uint32_t W[4][2], up[2], dn[2];
  pvrtc_stored_row<0>(data, two, W[0]);
  pvrtc_stored_row<1>(data, two, W[1]);
  pvrtc_stored_row<2>(data, two, W[2]);
  pvrtc_stored_row<3>(data, two, W[3]);
  pvrtc_stored_row<3>(mod[1], (col[1] & 1u) != 0u, up);
  pvrtc_stored_row<0>(mod[7], (col[7] & 1u) != 0u, dn);
  // ---- the 4 weights of the left / right blocks that the block's unstored pixels look at (all of even parity there)
  const uint32_t l1 = pvrtc_stored_weight(mod[3], (col[3] & 1u) != 0u, 7u, 1u), l3 = pvrtc_stored_weight(mod[3], (col[3] & 1u) != 0u, 7u, 3u);
  const uint32_t r0 = pvrtc_stored_weight(mod[5], (col[5] & 1u) != 0u, 0u, 0u), r2 = pvrtc_stored_weight(mod[5], (col[5] & 1u) != 0u, 0u, 2u);
  // ---- weights of all 32 pixels.  The three interpolation modes are one formula on packed bytes: with hs = l + r and
  // vs = u + d, (hs + vs + 2) >> 2 is the four-neighbour average, and the two-neighbour ones are the same expression
  // with hs or vs taken twice ((2 vs + 2) >> 2 == (vs + 1) >> 1); no carries between bytes (4 * 8 + 2 < 256).
  const uint32_t only_v = ((data & 1u) != 0u && (data & (1u << 20)) != 0u) ? 0xffffffffu : 0u;
  const uint32_t only_h = ((data & 1u) != 0u && (data & (1u << 20)) == 0u) ? 0xffffffffu : 0u;
  uint32_t Wf[4][2];
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const uint32_t left[2] = { y == 1 ? (W[y][0] << 8 | l1) : y == 3 ? (W[y][0] << 8 | l3) : W[y][0] << 8, alignbit(W[y][1], W[y][0], 24) };
    const uint32_t right[2] = { alignbit(W[y][1], W[y][0], 8), y == 0 ? (W[y][1] >> 8 | r0 << 24) : y == 2 ? (W[y][1] >> 8 | r2 << 24) : W[y][1] >> 8 };
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t u = y > 0 ? W[y - 1][h] : up[h], d = y < 3 ? W[y + 1][h] : dn[h];
      const uint32_t hs = left[h] + right[h], vs = u + d;
      const uint32_t first = (vs & only_v) | (hs & ~only_v), second = (hs & only_h) | (vs & ~only_h);
      const uint32_t interp = ((first + second + 0x02020202u) >> 2) & 0x0f0f0f0fu;
      const uint32_t stored_at = (y & 1) ? 0xff00ff00u : 0x00ff00ffu;  // even checkerboard parity
      Wf[y][h] = two ? ((W[y][h] & stored_at) | (interp & ~stored_at)) : W[y][h];
    }
  }
  // ---- colours.  r05: the weights enter as 32 w and 256 - 32 w on 16-bit lanes, two pixels per register, the lane picked by
  // the multiply itself (op_sel): (256 - 32 w) A + 32 w B = 32 ((8 - w) A + w B) <= 65 280 still fits a lane, and its HIGH
  // byte is ((8 - w) A + w B) >> 3 -- one byte permute assembles the pixel, no shift.
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const int r0b = y < 2 ? 0 : 1;                     // block rows (r0b, r0b + 1) bracket this pixel row
    const uint32_t yw = (uint32_t)((y + 2) & 3);
    uint32_t V[3][4], row[8];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c)
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) V[c][v] = vblend_pair(yw, C[r0b][c][v], C[r0b + 1][c][v]);
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      uint32_t P[4], D[4];
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        const uint32_t vl = V[h][v], vr = V[h + 1][v];
        D[v] = vr - vl;
        P[v] = h == 0 ? (vl + vr) << 2 : vl << 3;     // xw = 4 / xw = 0
      }
      uint32_t w32[2], iw32[2];
      w32[0] = perm(0u, Wf[y][h], 0x0c010c00u) << 5;   // lanes: 32 w of pixels 0, 1
      w32[1] = perm(0u, Wf[y][h], 0x0c030c02u) << 5;   //                       2, 3
      iw32[0] = 0x01000100u - w32[0];
      iw32[1] = 0x01000100u - w32[1];
      ICAMD_UNROLL
      for (int j = 0; j < 4; ++j) {
        const uint32_t a_rb = pk_lshr16(P[0], 8), a_ga = pk_lshr16(P[1], 8), b_rb = pk_lshr16(P[2], 8), b_ga = pk_lshr16(P[3], 8);
        uint32_t rb, ga;
        if (j & 1) {
          rb = pk_mad_u16_lane<1>(b_rb, w32[j >> 1], pk_mad_u16_lane<1>(a_rb, iw32[j >> 1], 0u));
          ga = pk_mad_u16_lane<1>(b_ga, w32[j >> 1], pk_mad_u16_lane<1>(a_ga, iw32[j >> 1], 0u));
        } else {
          rb = pk_mad_u16_lane<0>(b_rb, w32[j >> 1], pk_mad_u16_lane<0>(a_rb, iw32[j >> 1], 0u));
          ga = pk_mad_u16_lane<0>(b_ga, w32[j >> 1], pk_mad_u16_lane<0>(a_ga, iw32[j >> 1], 0u));
        }
        row[4 * h + j] = perm(ga, rb, 0x07030501u);  // R, G, B, A = the high bytes of rb.lo, ga.lo, rb.hi, ga.hi
        if (j < 3) {
          ICAMD_UNROLL
          for (int v = 0; v < 4; ++v) P[v] += D[v];
        }
      }
    }
    emit(y, row);
  }
}
ICAMD_DEV void decode_pvrtc2_block_expanded(const uint32_t C[3][3][4], const uint32_t mod[9], const uint32_t col[9],
                                            uint32_t px[32]) {
  decode_pvrtc2_block_rows(C, mod, col, [&](int y, const uint32_t row[8]) {
    ICAMD_UNROLL
    for (int x = 0; x < 8; ++x) px[8 * y + x] = row[x];
  });
}

// ---- PVRTC1 4 bpp decoder (r05): EXTENSION of an extension, PARITY UNPINNED (the reference has neither a 4 bpp format nor a
// PVRTC decoder; BASELINE names the format).  The inverse of the 4 bpp encoder of pvrtc_block.h, the plain-C statement of the
// same rules is oracle/ic_oracle.c pvrtc4_decode_image.  4 x 4 blocks, block centres at (2, 2): pixel (x, y) blends the low
// resolution colours with weights (x + 2) & 3, (y + 2) & 3 out of 4 in both directions, toroidal; every pixel stores its own
// 2-bit value: weights 0, 3, 5, 8 eighths of B (ApplyModulation, pvrtc.cc:120-144), or -- colour word bit 0 set, which the
// encoder never writes -- PVRTC1's punch-through 0, 4, 4, 8 with alpha 0 for value 2.
// C[r][c][v]: expanded colours of the 3 x 3 block neighbourhood; emit(y, row): the four pixels of pixel row y.
// Same walk as the 2 bpp decoder at scale 256: V = 64 x vertical blend, P(px) = (4 - px) VL + px VR stepped by VR - VL.
ICAMD_DEV uint32_t vblend4_pair(uint32_t py, uint32_t top, uint32_t bot) {
  if (py == 0u) return top << 6;
  if (py == 2u) return (top + bot) << 5;
  return (py == 1u ? 3u * top + bot : top + 3u * bot) << 4;
}
template <typename Emit>
ICAMD_DEV void decode_pvrtc4_block_rows(const uint32_t C[3][3][4], uint32_t data, bool punch, Emit emit) {
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const int r0b = y < 2 ? 0 : 1;                     // block rows (r0b, r0b + 1) bracket this pixel row
    const uint32_t py = (uint32_t)((y + 2) & 3);
    uint32_t V[3][4], row[4];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c)
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) V[c][v] = vblend4_pair(py, C[r0b][c][v], C[r0b + 1][c][v]);
    // the row's four 2-bit values on 16-bit lanes: pixels (0, 1) and (2, 3)
    const uint32_t bits = bfe(data, 8 * y, 8), hb = bits >> 4;
    const uint32_t S[2] = { (bits | bits << 14) & 0x00030003u, (hb | hb << 14) & 0x00030003u };
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t std_w = 3u * S[h] - ((S[h] >> 1) & 0x00010001u);                   // {0, 3, 5, 8}
      const uint32_t pt_w = (((S[h] + 0x00010001u) >> 1) & 0x00030003u) << 2;            // {0, 4, 4, 8}
      const uint32_t w32 = (punch ? pt_w : std_w) << 5, iw32 = 0x01000100u - w32;
      const uint32_t hole = punch ? ((S[h] >> 1) & ~S[h] & 0x00010001u) : 0u;            // value 2 of a punch-through block
      uint32_t P[4], D[4];
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        const uint32_t vl = V[h][v], vr = V[h + 1][v];
        D[v] = vr - vl;
        P[v] = h == 0 ? (vl + vr) << 1 : vl << 2;      // px = 2 / px = 0
      }
      ICAMD_UNROLL
      for (int j = 0; j < 2; ++j) {
        const uint32_t a_rb = pk_lshr16(P[0], 8), a_ga = pk_lshr16(P[1], 8), b_rb = pk_lshr16(P[2], 8), b_ga = pk_lshr16(P[3], 8);
        uint32_t rb, ga;
        if (j) {
          rb = pk_mad_u16_lane<1>(b_rb, w32, pk_mad_u16_lane<1>(a_rb, iw32, 0u));
          ga = pk_mad_u16_lane<1>(b_ga, w32, pk_mad_u16_lane<1>(a_ga, iw32, 0u));
        } else {
          rb = pk_mad_u16_lane<0>(b_rb, w32, pk_mad_u16_lane<0>(a_rb, iw32, 0u));
          ga = pk_mad_u16_lane<0>(b_ga, w32, pk_mad_u16_lane<0>(a_ga, iw32, 0u));
        }
        const uint32_t keep = ((hole >> (16 * j)) & 1u) ? 0x00ffffffu : 0xffffffffu;
        row[2 * h + j] = perm(ga, rb, 0x07030501u) & keep;
        if (j == 0) {
          ICAMD_UNROLL
          for (int v = 0; v < 4; ++v) P[v] += D[v];
        }
      }
    }
    emit(y, row);
  }
}
// col[d] = colour word of the block at (dx, dy) = index 3 * (dy + 1) + dx + 1, data = the block's own modulation word
ICAMD_DEV void decode_pvrtc4_block(uint32_t data, const uint32_t col[9], uint32_t px[16]) {
  uint32_t C[3][3][4];
  ICAMD_UNROLL
  for (int i = 0; i < 9; ++i) pvrtc_expand_colors(col[i], C[i / 3][i % 3]);
  decode_pvrtc4_block_rows(C, data, (col[4] & 1u) != 0u, [&](int y, const uint32_t row[4]) {
    ICAMD_UNROLL
    for (int x = 0; x < 4; ++x) px[4 * y + x] = row[x];
  });
}

ICAMD_DEV void decode_pvrtc2_block(const uint32_t mod[9], const uint32_t col[9], uint32_t px[32]) {
  uint32_t C[3][3][4];  // [block row][block column][a_rb, a_ga, b_rb, b_ga]
  ICAMD_UNROLL
  for (int i = 0; i < 9; ++i) pvrtc_expand_colors(col[i], C[i / 3][i % 3]);
  decode_pvrtc2_block_expanded(C, mod, col, px);
}

}  // namespace icamd
#endif  // ICAMD_DECODE_BLOCK_H_
