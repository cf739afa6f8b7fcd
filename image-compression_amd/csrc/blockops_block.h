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

// The same with four lanes per pad block (kSmallerError; etc1_block.h encode_etc1_block_quad): every lane of the quad decodes
// the source block and builds the pad block's pixels, the search is split.  t = lane & 3; *writes: this lane stores.
#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV Out8 etc1_pad_block_quad(uint32_t w0, uint32_t w1, int kind) {
#else
ICAMD_DEV Out8 etc1_pad_block_quad(uint32_t w0, uint32_t w1, int kind, uint32_t t, bool *writes) {
#endif
  uint32_t src[16], px[16];
  decode_etc1(w0, w1, src);
#if !defined(ICAMD_HOST_EMULATION)
  if (kind == kPadCorner) { *writes = t == 0u; return etc1_solid_block(src[15]); }
#else
  if (kind == kPadCorner) return etc1_solid_block(src[15]);
#endif
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int x = 0; x < 4; ++x) px[4 * y + x] = kind == kPadColumn ? src[4 * y + 3] : src[12 + x];
  }
#if defined(ICAMD_HOST_EMULATION)
  return encode_etc1_block_quad(px);
#else
  return encode_etc1_block_quad(px, t, writes);
#endif
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

// ---- DownsampleBlocks2x2 of DXT blocks without ever materialising the 64 source pixels (r04) ------------------------
// A DXT colour block holds at most four colours.  Keep the palette as three "planes" -- byte k of P[ch] = channel ch of
// palette entry k -- and the four 2-bit indices of a 2x2 pixel quad as the four bytes of a selector: ONE v_perm_b32 then
// yields the quad's four values of a channel, and their sum is one v_sad_u8 / v_dot4.  Per output pixel that is
// 4 (selector) + 3 x 2 (perm + sum) + 4 (pack) instructions instead of four 4-way palette look-ups and a 14-instruction
// average -- bit-exact with DecodeDxt1Block / DecodeDxt5Block (dxtc.cc:167-267) followed by Average4ColorsFast
// (color_util.h:335-380): (a + b + c + d) / 4 per channel, truncating.
//
// Palette planes of the colour block whose first dword is w0 (c0 | c1 << 16).  always4 = DXT5's colour block.
// floor(n / 3) for n <= 765 = 3 * 255 is byte 2 of (85 n + 1) * 257: 85 * 257 = 65535 / 3, so the product is n / 3 in units of
// 2^16 minus n / 3 plus 257 -- the + 1 keeps exact multiples of three from falling one short.  Checked for every n.
constexpr bool third_is_byte2_of_product() {
  for (unsigned n = 0; n <= 765; ++n)
    if (((((85u * n + 1u) * 257u) >> 16) & 0xffu) != n / 3u || (85u * n + 1u) * 257u >= (1u << 24)) return false;
  return true;
}
static_assert(third_is_byte2_of_product(), "(85 n + 1) * 257 >> 16 != n / 3");

ICAMD_DEV void dxt_palette_planes(uint32_t w0, bool always4, uint32_t P[3]) {
  // ExtendToRgb888 (color_util.h:232-236) on both endpoints at once: X0 | X1 << 16 per channel
  const uint32_t r = (w0 >> 11) & 0x001f001fu, g = (w0 >> 5) & 0x003f003fu, b = w0 & 0x001f001fu;
  const uint32_t X[3] = { (r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2) };
  const uint32_t c0 = w0 & 0xffffu, c1 = w0 >> 16;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    // X[ch]: endpoint 0 in byte 0, endpoint 1 in byte 2 (the right shifts leak endpoint 1's low bits into byte 1: every use
    // below gives that byte the weight zero).  (2 a + b) / 3 and (a + 2 b) / 3 (CombineUint8Fast, color_util.h:288-291) *(r06)*:
    // 85 (2 a + b) + 1 is ONE v_dot4 with the weights (170, 85) and the accumulator 1, times 257 one v_lshl_add, and the third
    // is byte 2 of that (third_is_byte2_of_product) -- which a v_perm picks up while it assembles the plane: 6 instructions per
    // channel where the multiply-shift form took 11.  c0 == c1 gives x2 = x3 = x1, which is what the decoder's special case for
    // equal endpoints produces (dxtc.cc:184-186)
    const uint32_t t2 = udot4(X[ch], 0x005500aau, 1u), t3 = udot4(X[ch], 0x00aa0055u, 1u);
    const uint32_t p2 = (t2 << 8) + t2, p3 = (t3 << 8) + t3;
    const uint32_t q = perm(p3, p2, 0x0c0c0602u);      // [x2, x3, 0, 0]
    P[ch] = perm(q, X[ch], 0x05040200u);               // [x0, x1, x2, x3]
  }
  // DXT1's three-colour mode (c0 < c1): entry 2 = (a + b) / 2, entry 3 = black (dxtc.cc:187-193).  Our own encoder emits
  // it only from the constant-colour path; a wave without such a block skips this.
  if (!always4 && !wave_all(c0 >= c1)) {
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      const uint32_t x0 = X[ch] & 0xffu, x1 = bfe(X[ch], 16, 8);
      const uint32_t alt = x0 | x1 << 8 | ((x0 + x1) >> 1) << 16;
      P[ch] = c0 < c1 ? alt : P[ch];
    }
  }
}

// t: the index bytes of two pixel rows at bits 0-7 and 16-23 (2 bits per pixel).  Returns the selector of the 2x2 quad
// at columns 2 QX, 2 QX + 1: byte i = index of quad pixel i (order: row 0 left, right, row 1 left, right).
template <int QX>
ICAMD_DEV uint32_t dxt_quad_selector(uint32_t t) {
  const uint32_t v = (QX ? t >> 4 : t) & 0x000f000fu;  // the two rows' nibbles
  return (v | v << 6) & 0x03030303u;                    // nibble = (hi2 << 2 | lo2): lo2 stays, hi2 moves up one byte
}

// Sum / 4 of the quad's four palette entries, packed R | G << 8 | B << 16 (byte 3 = 0).
ICAMD_DEV uint32_t dxt_quad_average(const uint32_t P[3], uint32_t sel) {
  // 64 * (sum of the four values) has sum / 4 in byte 1 (sum <= 1020: nothing reaches byte 2); two v_perm collect the three
  // bytes (r06: 2 instructions where masks, shifts and ors took 5)
  const uint32_t tr = udot4(perm(P[0], P[0], sel), 0x40404040u, 0u);
  const uint32_t tg = udot4(perm(P[1], P[1], sel), 0x40404040u, 0u);
  const uint32_t tb = udot4(perm(P[2], P[2], sel), 0x40404040u, 0u);
  return perm(tb, perm(tg, tr, 0x0c0c0501u), 0x0c050100u);
}

// The eight alpha values of a DXT5 alpha block (first dword w0 = a0 | a1 << 8 | codes...) as two dwords of bytes
// (DecodeAlphaValues, dxtc.cc:195-217).
// floor(n / 7) for n <= 7 * 255 and floor(n / 5) for n <= 5 * 255 as BYTE 2 of one 24-bit product (9363 = ceil(2^16 / 7),
// 13108 = ceil(2^16 / 5)): no shift -- a v_perm picks the byte up while it assembles the table.  Checked for every n.
constexpr bool quotient_is_byte2(unsigned d, unsigned mul, unsigned max) {
  for (unsigned n = 0; n <= max; ++n)
    if ((((n * mul) >> 16) & 0xffu) != n / d || n * mul >= (1u << 24)) return false;
  return true;
}
static_assert(quotient_is_byte2(7, 9363, 1785) && quotient_is_byte2(5, 13108, 1275), "sevenths / fifths as byte 2 of a 24-bit product");

ICAMD_DEV void dxt5_alpha_planes(uint32_t w0, uint32_t &tlo, uint32_t &thi) {
  const uint32_t a0 = w0 & 0xffu, a1 = (w0 >> 8) & 0xffu;
  const bool eight = a0 > a1;
  uint32_t lo8 = 0, hi8 = 0, lo6 = 0, hi6 = 0;
  // *(r06)* (7 - k) a0 + k a1 is ONE v_dot4 on the block's first dword (a0, a1 in bytes 0, 1; the code bytes get weight zero),
  // the division one 24-bit multiply whose byte 2 is the quotient: 2 instructions per value where two mads, a multiply and a
  // shift took 4, and the table is assembled by v_perm (5 / 4 instead of 6 / 5 shifts and ors)
  if (!wave_all(!eight)) {  // some lane interpolates six values in sevenths
    uint32_t p[7];
    ICAMD_UNROLL
    for (int k = 1; k <= 6; ++k) p[k] = umad24(udot4(w0, (uint32_t)(k << 8 | (7 - k)), 0u), 9363u, 0u);  // byte 2 = value 1 + k
    const uint32_t v23 = perm(p[2], p[1], 0x0c0c0602u);
    lo8 = perm(v23, w0, 0x05040100u);                                  // [a0, a1, t2, t3]
    hi8 = perm(perm(p[6], p[5], 0x06020c0cu), perm(p[4], p[3], 0x0c0c0602u), 0x07060100u);  // [t4, t5, t6, t7]
  }
  if (!wave_all(eight)) {   // some lane interpolates four values in fifths, then 0 and 255
    uint32_t p[5];
    ICAMD_UNROLL
    for (int k = 1; k <= 4; ++k) p[k] = umad24(udot4(w0, (uint32_t)(k << 8 | (5 - k)), 0u), 13108u, 0u);
    const uint32_t v23 = perm(p[2], p[1], 0x0c0c0602u);
    lo6 = perm(v23, w0, 0x05040100u);
    hi6 = perm(p[4], p[3], 0x0c0c0602u) | 0xff000000u;  // [t4, t5, 0, 255]
  }
  tlo = eight ? lo8 : lo6;
  thi = eight ? hi8 : hi6;
}

// h24: the 3-bit alpha codes of two pixel rows (pixels 0..7 of the pair, 24 bits).  Selector of the quad at columns
// 2 QX, 2 QX + 1: byte i = code (0..7) of quad pixel i -- v_perm_b32 over {thi, tlo} looks all four alphas up at once.
template <int QX>
ICAMD_DEV uint32_t dxt5_quad_alpha_selector(uint32_t h24) {
  const uint32_t x = QX ? h24 >> 6 : h24;
  const uint32_t y = (x & 0x3fu) | bfe(x, 12, 6) << 16;  // codes (0, 1) at bits 0-5, (2, 3) at bits 16-21
  return (y | y << 5) & 0x07070707u;
}

// px[16] <- the 2x2 averages of the 8x8 pixels that the four blocks s[i][j] (i = block row, j = block column; 2 or 4
// dwords each) decode to.  CODEC: ICAMD_DXT1 / ICAMD_DXT5 numbering of ic_amd.h (0 / 1).
template <int CODEC>
ICAMD_DEV void dxt_downsample_2x2(const uint32_t *const s[2][2], uint32_t px[16]) {
  ICAMD_UNROLL
  for (int i = 0; i < 2; ++i) {
    ICAMD_UNROLL
    for (int j = 0; j < 2; ++j) {
      const uint32_t *w = s[i][j];
      const uint32_t cw0 = CODEC == 1 ? w[2] : w[0], bits = CODEC == 1 ? w[3] : w[1];
      uint32_t P[3];
      dxt_palette_planes(cw0, CODEC == 1, P);
      uint32_t tlo = 0, thi = 0, lo24 = 0, hi24 = 0;
      if (CODEC == 1) {
        dxt5_alpha_planes(w[0], tlo, thi);
        lo24 = w[0] >> 16 | (w[1] & 0xffu) << 16;  // codes of pixels 0-7 / 8-15
        hi24 = w[1] >> 8;
      }
      ICAMD_UNROLL
      for (int qy = 0; qy < 2; ++qy) {
        const uint32_t t = perm(0u, bits, qy ? 0x0c030c02u : 0x0c010c00u);  // index bytes of rows (2 qy, 2 qy + 1)
        uint32_t q0 = dxt_quad_average(P, dxt_quad_selector<0>(t));
        uint32_t q1 = dxt_quad_average(P, dxt_quad_selector<1>(t));
        if (CODEC == 1) {
          const uint32_t h = qy ? hi24 : lo24;
          const uint32_t a0 = udot4(perm(thi, tlo, dxt5_quad_alpha_selector<0>(h)), 0x40404040u, 0u);
          const uint32_t a1 = udot4(perm(thi, tlo, dxt5_quad_alpha_selector<1>(h)), 0x40404040u, 0u);
          q0 = perm(a0, q0, 0x05020100u);  // byte 3 <- byte 1 of 64 * sum = sum / 4
          q1 = perm(a1, q1, 0x05020100u);
        }
        // StoreDownsampledPixels4x4 (pixel4x4.h:152-162): block (i, j) fills the 2x2 quadrant at (2 i, 2 j)
        px[4 * (2 * i + qy) + 2 * j] = q0;
        px[4 * (2 * i + qy) + 2 * j + 1] = q1;
      }
    }
  }
}

// ---- the same for ETC1 (r04).  A sub-block has four colours -- base + {a, b, -a, -b}, each channel clamped to 0..255
// (etc.cc:101-125) -- so an ETC1 block is two four-entry palettes; a 2x2 pixel quad never straddles the two sub-blocks
// (they split the block at column 2 or at row 2), so one palette serves a whole quad.
//
// P[s][ch]: byte k of = channel ch of sub-block s's candidate k (k = modifier index: +a, +b, -a, -b).  Returns flip.
ICAMD_DEV bool etc1_palette_planes(uint32_t w0, uint32_t P[2][3]) {
  const uint32_t hi = perm(0u, w0, 0x00010203u);  // big-endian word (etc.cc:172-180)
  const bool diff = (hi & 2u) != 0u;
  const uint32_t cws[2] = { (hi >> 5) & 7u, (hi >> 2) & 7u };
  ICAMD_UNROLL
  for (int sb = 0; sb < 2; ++sb) {
    const uint32_t cw = cws[sb], sh = (cw & 3u) * 8u;
    const uint32_t a = bfe(cw < 4u ? kEtcModA_lo : kEtcModA_hi, sh, 8), b = bfe(cw < 4u ? kEtcModB_lo : kEtcModB_hi, sh, 8);
    const uint32_t ab = a << 8 | b << 24;  // the two magnitudes in the high bytes of the 16-bit lanes
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      // base colour of this sub-block exactly as decode_etc1 / the reference's decoder reconstruct it (etc.cc:198-273):
      // plain int arithmetic, so a differential base whose 5-bit sum leaves 0..31 is negative or above 255 here and only
      // the final base + modifier is clamped
      int32_t sbase;
      if (diff) {
        const int32_t b5 = (int32_t)((hi >> (27 - 8 * ch)) & 31u), d3 = (int32_t)((hi >> (24 - 8 * ch)) & 7u);
        const int32_t v5 = sb == 0 ? b5 : b5 + (d3 >= 4 ? d3 - 8 : d3);  // ExtendSignBit, bit_util.h:61-69
        sbase = (v5 << 3) | ((v5 >> 2) & 7);                              // Extend5Bit, color_util.h:200-202
      } else {
        sbase = (int32_t)(((hi >> ((sb == 0 ? 28 : 24) - 8 * ch)) & 15u) * 17u);
      }
      // candidates clamp(sbase +/- m): with the value in the high byte of a 16-bit lane, saturating adds / subtracts clamp
      // at 255 / 0 by themselves for sbase in 0..255; the (rare) out-of-range bases take the plain-integer form
      const uint32_t b2 = ((uint32_t)sbase & 0xffu) * 0x01000100u;
      uint32_t up = pk_addsat_u16(b2, ab), dn = pk_subsat_u16(b2, ab);
      uint32_t plane = perm(dn, up, 0x07050301u);
      if (!wave_all(sbase >= 0 && sbase <= 255)) {
        const uint32_t k0 = clamp255(sbase + (int32_t)a), k1 = clamp255(sbase + (int32_t)b);
        const uint32_t k2 = clamp255(sbase - (int32_t)a), k3 = clamp255(sbase - (int32_t)b);
        const uint32_t slow = k0 | k1 << 8 | k2 << 16 | k3 << 24;
        plane = (sbase >= 0 && sbase <= 255) ? plane : slow;
      }
      P[sb][ch] = plane;
    }
  }
  return (hi & 1u) != 0u;
}

// Selector of the quad at columns (2 QX, 2 QX + 1), rows (2 QY, 2 QY + 1): byte i = modifier index of one of the quad's
// pixels (the ORDER inside the quad does not matter for a sum).  lo = the block's big-endian index word: bit 4x + y holds
// the LSB of pixel (x, y)'s index, bit 16 + 4x + y the MSB (etc.cc:131-156).
template <int QX, int QY>
ICAMD_DEV uint32_t etc1_quad_selector(uint32_t lo) {
  const uint32_t z = (8 * QX + 2 * QY) ? lo >> (8 * QX + 2 * QY) : lo;  // the quad's bits now sit at 0, 1, 4, 5 (+ 16)
  // bits 0, 1, 4, 5 -> bit 0 of bytes 0..3: one multiply (the partial products that collide carry no further than bit 14)
  const uint32_t l = umad24(z & 0x33u, 0x81081u, 0u) & 0x01010101u;
  const uint32_t m = umad24(bfe(z, 16, 6) & 0x33u, 0x81081u, 0u) & 0x01010101u;
  return l | m << 1;
}

// px[16] <- 2x2 averages of the 8x8 pixels of four ETC1 blocks s[i][j] (2 dwords each); see dxt_downsample_2x2.
ICAMD_DEV void etc1_downsample_2x2(const uint32_t *const s[2][2], uint32_t px[16]) {
  ICAMD_UNROLL
  for (int i = 0; i < 2; ++i) {
    ICAMD_UNROLL
    for (int j = 0; j < 2; ++j) {
      uint32_t P[2][3];
      const bool flip = etc1_palette_planes(s[i][j][0], P);
      const uint32_t lo = perm(0u, s[i][j][1], 0x00010203u);
      // sub-block of quad (qx, qy): flip ? qy : qx
      uint32_t Pm[2][3];  // palettes of the quads (1, 0) and (0, 1)
      ICAMD_UNROLL
      for (int ch = 0; ch < 3; ++ch) {
        Pm[0][ch] = flip ? P[0][ch] : P[1][ch];  // (qx, qy) = (1, 0)
        Pm[1][ch] = flip ? P[1][ch] : P[0][ch];  // (0, 1)
      }
      px[4 * (2 * i) + 2 * j] = dxt_quad_average(P[0], etc1_quad_selector<0, 0>(lo));
      px[4 * (2 * i) + 2 * j + 1] = dxt_quad_average(Pm[0], etc1_quad_selector<1, 0>(lo));
      px[4 * (2 * i + 1) + 2 * j] = dxt_quad_average(Pm[1], etc1_quad_selector<0, 1>(lo));
      px[4 * (2 * i + 1) + 2 * j + 1] = dxt_quad_average(P[1], etc1_quad_selector<1, 1>(lo));
    }
  }
}

// ---- the block DECODERS on the same planes (r04): a pixel ROW's four indices become one selector, one v_perm per channel
// looks the row up, and two v_perm per output dword interleave the channels in memory order -- the 16 pixels of a block are
// never selected one by one (decode_dxt_colors / decode_dxt5_alpha / decode_etc1 spend 7-16 instructions per pixel on 4- and
// 8-way selects).  Same bytes as those functions (checked block by block in tests/host_emul).
//
// Selector of pixel row y of a DXT colour block: byte x = 2-bit index of pixel (x, y).
ICAMD_DEV uint32_t dxt_row_selector(uint32_t bits, int y) {
  const uint32_t v = bfe(bits, 8u * (uint32_t)y, 8u);
  const uint32_t t = v | v << 12;
  return (t | t << 6) & 0x03030303u;
}
// Selector of pixel row y of a DXT5 alpha block: byte x = 3-bit code of pixel (x, y).  lo24 / hi24 = codes of pixels 0-7 / 8-15.
ICAMD_DEV uint32_t dxt5_row_alpha_selector(uint32_t lo24, uint32_t hi24, int y) {
  const uint32_t h = y < 2 ? lo24 : hi24, x = (y & 1) ? h >> 12 : h;
  const uint32_t t = (x & 0x3fu) | bfe(x, 6, 6) << 16;
  return (t | t << 5) & 0x07070707u;
}
// Selector of pixel row y of an ETC1 block for v_perm over {palette of sub-block 1, palette of sub-block 0}: byte x = modifier
// index of pixel (x, y) + 4 if the pixel belongs to sub-block 1.  lo = big-endian index word (bit 4x + y: LSB, + 16: MSB).
ICAMD_DEV uint32_t etc1_row_selector(uint32_t lo, int y, bool flip) {
  const uint32_t w = (y ? lo >> y : lo) & 0x11111111u;                       // bits 0, 4, 8, 12 (LSBs) and 16, 20, 24, 28 (MSBs)
  const uint32_t l = perm(0u, w, 0x0c010c00u), m = perm(0u, w, 0x0c030c02u);  // bits 0, 4, 16, 20 of each
  const uint32_t lb = (l | l << 4) & 0x01010101u, mb = (m | m << 4) & 0x01010101u;
  const uint32_t sub = flip ? (y >= 2 ? 0x04040404u : 0u) : 0x04040000u;     // sub-block 1: y >= 2 (flip) / x >= 2
  return lb | mb << 1 | sub;
}
// Interleave three planar rows (bytes x = 0..3 of each channel) into the 12 bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3.
ICAMD_DEV void interleave_rgb_row(uint32_t r, uint32_t g, uint32_t b, uint32_t out[3]) {
  out[0] = perm(b, perm(g, r, 0x010c0400u), 0x03040100u);   // [R0 G0 .. R1] then B0 into byte 2
  out[1] = perm(r, perm(b, g, 0x020c0501u), 0x03060100u);   // [G1 B1 .. G2] then R2 into byte 2
  out[2] = perm(g, perm(r, b, 0x030c0702u), 0x03070100u);   // [B2 R3 .. B3] then G3 into byte 2
}
// ... and four planar rows into the four R G B A pixels of the row.
ICAMD_DEV void interleave_rgba_row(uint32_t r, uint32_t g, uint32_t b, uint32_t a, uint32_t out[4]) {
  const uint32_t rg01 = perm(g, r, 0x05010400u), rg23 = perm(g, r, 0x07030602u);
  const uint32_t ba01 = perm(a, b, 0x05010400u), ba23 = perm(a, b, 0x07030602u);
  out[0] = perm(ba01, rg01, 0x05040100u);
  out[1] = perm(ba01, rg01, 0x07060302u);
  out[2] = perm(ba23, rg23, 0x05040100u);
  out[3] = perm(ba23, rg23, 0x07060302u);
}
// A whole block to its four pixel rows in output memory order: rows[y][0..2] = the 12 bytes of an RGB888 row (CODEC 0: DXT1,
// 2: ETC1), rows[y][0..3] = the four RGBA pixels (CODEC 1: DXT5).  swap: stored R goes to the third byte (kBGR / kBGRA).
template <int CODEC>
ICAMD_DEV void decode_block_rows(const uint32_t *w, bool swap, uint32_t rows[4][4]) {
  if (CODEC == 2) {
    uint32_t P[2][3];
    const bool flip = etc1_palette_planes(w[0], P);
    const uint32_t lo = perm(0u, w[1], 0x00010203u);
    ICAMD_UNROLL
    for (int y = 0; y < 4; ++y) {
      const uint32_t sel = etc1_row_selector(lo, y, flip);
      interleave_rgb_row(perm(P[1][0], P[0][0], sel), perm(P[1][1], P[0][1], sel), perm(P[1][2], P[0][2], sel), rows[y]);
    }
    return;
  }
  uint32_t P[3];
  dxt_palette_planes(CODEC == 1 ? w[2] : w[0], CODEC == 1, P);
  const uint32_t bits = CODEC == 1 ? w[3] : w[1];
  const uint32_t pr = swap ? P[2] : P[0], pb = swap ? P[0] : P[2];
  uint32_t tlo = 0, thi = 0, lo24 = 0, hi24 = 0;
  if (CODEC == 1) {
    dxt5_alpha_planes(w[0], tlo, thi);
    lo24 = w[0] >> 16 | (w[1] & 0xffu) << 16;
    hi24 = w[1] >> 8;
  }
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const uint32_t sel = dxt_row_selector(bits, y);
    const uint32_t r = perm(pr, pr, sel), g = perm(P[1], P[1], sel), b = perm(pb, pb, sel);
    if (CODEC == 1) interleave_rgba_row(r, g, b, perm(thi, tlo, dxt5_row_alpha_selector(lo24, hi24, y)), rows[y]);
    else interleave_rgb_row(r, g, b, rows[y]);
  }
}

// ---- TranscodeDxt1ToEtc1 in the palette domain (r04).  dxtc_to_etc_transcoder.cc:29-40 decodes the DXT1 block and runs
// EncodeEtc1Block(kHeuristic) on the 16 pixels.  The pixels are four colours at most, so everything kHeuristic computes
// follows from the palette and the 2-bit indices without materialising a pixel:
//   * quadrant sums / deviation sums (etc.cc:299-312, 415-455, 553-574) from the channel planes through the quad selectors
//     (as in dxt_downsample_2x2); the reference's "(2,2) twice, never (3,3)" quadrant is one weighted v_dot4;
//   * ComputeCodewordError (etc.cc:350-385) for the FOUR palette colours instead of the eight pixels of a sub-block: 16 keys
//     per sub-block instead of 32, same keys and tie rule as eval_codeword;
//   * the pixels' new 2-bit indices are a 4-entry table look-up of their old ones, done on bit planes: the DXT index bits
//     are gathered straight into ETC bit order (bit 4x + y, etc.cc:131-137) by v_dot4 with power-of-two weights, and the
//     table look-up is three v_bfi per plane and sub-block.
// Bit-exact with decode_dxt_colors + encode_etc1_block(px, 3) (checked block by block in tests/host_emul).
ICAMD_DEV Out8 transcode_dxt1_block_to_etc1(uint32_t w0, uint32_t bits) {
  uint32_t P[3];
  dxt_palette_planes(w0, false, P);
  // the four palette colours as pixels (R | G << 8 | B << 16)
  const uint32_t t01 = perm(P[1], P[0], 0x05010400u), t23 = perm(P[1], P[0], 0x07030602u);  // R G R G of entries (0, 1) / (2, 3)
  const uint32_t col[4] = { perm(P[2], t01, 0x0c040100u), perm(P[2], t01, 0x0c050302u),
                            perm(P[2], t23, 0x0c060100u), perm(P[2], t23, 0x0c070302u) };
  // quadrant q = 2 (y >= 2) + (x >= 2): its four bytes of each channel, and their sums
  const uint32_t rows01 = perm(0u, bits, 0x0c010c00u), rows23 = perm(0u, bits, 0x0c030c02u);
  const uint32_t sel[4] = { dxt_quad_selector<0>(rows01), dxt_quad_selector<1>(rows01),
                            dxt_quad_selector<0>(rows23), dxt_quad_selector<1>(rows23) };
  uint32_t qpk[4][3], qs[4][3];
  ICAMD_UNROLL
  for (int q = 0; q < 4; ++q) {
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      qpk[q][ch] = perm(P[ch], P[ch], sel[q]);
      qs[q][ch] = sad_u8(qpk[q][ch], 0u, 0u);
    }
  }
  // the partition (etc.cc:553-574): the fourth quadrant sum uses pixel (2,2) twice and never (3,3) -- quad byte 0 twice,
  // byte 3 not at all
  uint32_t e_lr = 0, e_tb = 0;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t q3 = udot4(qpk[3][ch], 0x00010102u, 0u);
    const uint32_t l = (qs[0][ch] + qs[2][ch]) >> 3, r = (qs[1][ch] + q3) >> 3;
    const uint32_t t = (qs[0][ch] + qs[1][ch]) >> 3, b = (qs[2][ch] + q3) >> 3;
    const uint32_t dlr = sad_u32(l, r, 0u), dtb = sad_u32(t, b, 0u);
    e_lr = umad24(dlr, dlr, e_lr);
    e_tb = umad24(dtb, dtb, e_tb);
  }
  const bool flip = !(e_lr > e_tb);
  // sub-block sums and channel bytes: left | right = quadrants (0, 2) | (1, 3), top | bottom = (0, 1) | (2, 3)
  uint32_t s0[3], s1[3], pk[2][3][2];
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t mid0 = flip ? qs[1][ch] : qs[2][ch], mid1 = flip ? qs[2][ch] : qs[1][ch];
    s0[ch] = qs[0][ch] + mid0;
    s1[ch] = mid1 + qs[3][ch];
    pk[0][ch][0] = qpk[0][ch];
    pk[0][ch][1] = flip ? qpk[1][ch] : qpk[2][ch];
    pk[1][ch][0] = flip ? qpk[2][ch] : qpk[1][ch];
    pk[1][ch][1] = qpk[3][ch];
  }
  // FindBestSubblockEncoding's base colours (etc.cc:460-542), as encode_flip computes them
  uint32_t q5a[3], q5b[3];
  bool diff_mode = true;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    q5a[ch] = s0[ch] >> 6;
    q5b[ch] = s1[ch] >> 6;
    const int32_t d = (int32_t)q5b[ch] - (int32_t)q5a[ch];
    diff_mode = diff_mode && d >= -4 && d <= 3;
  }
  uint32_t hi = flip ? 1u : 0u, b0[3], b1[3];
  if (diff_mode) hi |= 2u;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t d3 = (q5b[ch] - q5a[ch]) & 7u, qa = s0[ch] >> 7, qb = s1[ch] >> 7;
    hi |= diff_mode ? (q5a[ch] << (27 - 8 * ch) | d3 << (24 - 8 * ch)) : (qa << (28 - 8 * ch) | qb << (24 - 8 * ch));
    b0[ch] = diff_mode ? ((q5a[ch] << 3) | (q5a[ch] >> 2)) : qa * 17u;
    b1[ch] = diff_mode ? ((q5b[ch] << 3) | (q5b[ch] >> 2)) : qb * 17u;
  }
  // per sub-block: FindCodewordHeuristic (etc.cc:415-455), then the best modifier of each PALETTE colour
  uint32_t fld[2][4];  // 32 E + (3 - k) of the winner; bits 0-1 = 3 - k
  uint32_t cws[2];
  ICAMD_UNROLL
  for (int sb = 0; sb < 2; ++sb) {
    const uint32_t *bc = sb ? b1 : b0;
    const EtcBase base = { bc[0] << 24 | bc[2] << 8, bc[1] << 8 };
    const uint32_t br4 = perm(0u, bc[0], 0u), bg4 = perm(0u, bc[1], 0u), bb4 = perm(0u, bc[2], 0u);  // byte 0 replicated (a plain multiply is a quarter-rate v_mul_lo_u32)
    const uint32_t sr = sad_u8(pk[sb][0][0], br4, sad_u8(pk[sb][0][1], br4, 0u));
    const uint32_t sg = sad_u8(pk[sb][1][0], bg4, sad_u8(pk[sb][1][1], bg4, 0u));
    const uint32_t sbl = sad_u8(pk[sb][2][0], bb4, sad_u8(pk[sb][2][1], bb4, 0u));
    const uint32_t dev = umax3(sr >> 3, sg >> 3, sbl >> 3);
    const uint32_t cw = (dev > 144u) + (dev > 93u) + (dev > 70u) + (dev > 51u) + (dev > 35u) + (dev > 23u) + (dev > 12u);
    const uint32_t sh = (cw & 3u) * 8u;
    const uint32_t a = bfe(cw < 4u ? kEtcModA_lo : kEtcModA_hi, sh, 8), b = bfe(cw < 4u ? kEtcModB_lo : kEtcModB_hi, sh, 8);
    uint32_t v[4];
    int32_t c[4];
    build_candidates(base, a, b, v, c);
    ICAMD_UNROLL
    for (int i = 0; i < 4; ++i) {
      const int32_t k0 = (int32_t)(udot4(col[i], v[0], 0u) << 6) + c[0];
      const int32_t k1 = (int32_t)(udot4(col[i], v[1], 0u) << 6) + c[1];
      const int32_t k2 = (int32_t)(udot4(col[i], v[2], 0u) << 6) + c[2];
      const int32_t k3 = (int32_t)(udot4(col[i], v[3], 0u) << 6) + c[3];
      fld[sb][i] = (uint32_t)imax(imax3(k0, k1, k2), k3);
    }
    cws[sb] = cw;
  }
  hi |= cws[0] << 5 | cws[1] << 2;
  // old index bits in ETC order: plane bit 4x + y = bit of pixel (x, y).  Column x of the low (high) index bit sits at bit
  // 2x (2x + 1) of each row's byte: mask it, and a v_dot4 with weights 2^(y + 2x') drops the column's four bits at its
  // nibble (two columns per pass so that the weights stay below 256; the upper two columns come from `bits >> 4`).
  const uint32_t bl = bits, bh = bits >> 1, bl4 = bits >> 4, bh4 = bits >> 5;
  const uint32_t L = udot4(bl & 0x04040404u, 0x20100804u, udot4(bl & 0x01010101u, 0x08040201u, 0u)) |
                     udot4(bl4 & 0x04040404u, 0x20100804u, udot4(bl4 & 0x01010101u, 0x08040201u, 0u)) << 8;
  const uint32_t H = udot4(bh & 0x04040404u, 0x20100804u, udot4(bh & 0x01010101u, 0x08040201u, 0u)) |
                     udot4(bh4 & 0x04040404u, 0x20100804u, udot4(bh4 & 0x01010101u, 0x08040201u, 0u)) << 8;
  // table look-up on the planes: field (3 - k) bit B of the pixel = fld[S][old index] bit B.  mask(x) = all ones iff the bit
  // is set (v_bfe_i32 of a 1-bit field); v_bfi(m, a, b) = (m & a) | (~m & b) picks by the old index's bits.
  uint32_t plane[2][2];  // [sub-block][bit of 3 - k]
  ICAMD_UNROLL
  for (int sb = 0; sb < 2; ++sb) {
    ICAMD_UNROLL
    for (int bit = 0; bit < 2; ++bit) {
      uint32_t m[4];
      ICAMD_UNROLL
      for (int i = 0; i < 4; ++i) m[i] = bit_mask(fld[sb][i], (uint32_t)bit);
      const uint32_t lo2 = (L & m[1]) | (~L & m[0]), hi2 = (L & m[3]) | (~L & m[2]);
      plane[sb][bit] = (H & hi2) | (~H & lo2);
    }
  }
  // sub-block 1 owns x >= 2 (bits 8-15) for flip = 0, y >= 2 (bits 2, 3 of every nibble) for flip = 1
  const uint32_t r1 = flip ? 0xccccu : 0xff00u;
  const uint32_t f_lsb = (r1 & plane[1][0]) | (~r1 & plane[0][0]), f_msb = (r1 & plane[1][1]) | (~r1 & plane[0][1]);
  // index k = 3 - field: both bits inverted; LSB plane in bits 0-15, MSB plane in bits 16-31 (etc.cc:150-156)
  const uint32_t lo = ~((f_lsb & 0xffffu) | f_msb << 16);
  Out8 o = { perm(0u, hi, 0x00010203u), perm(0u, lo, 0x00010203u) };
  return o;
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
