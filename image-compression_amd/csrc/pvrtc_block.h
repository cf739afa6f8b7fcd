// pvrtc_block.h -- PVRTC1 2bpp (8x4-pixel blocks) per-block and per-pixel math.
//
// Bit-exact with internal/pvrtc_compressor.cc (Morph :506-521, Modulate :527-540, Encode :551-580),
// restructured so that one lane owns one 8x4 block with its 32 pixels in VGPRs:
//  * GetExtremesFast (:255-329): the 5 fitness axes' "first minimum / first maximum" become unsigned
//    min / max reductions over keys value*32 + idx  /  value*32 + (31-idx); the R,G,B,A keys are one
//    v_dot4_u32_u8 each (weight 32 on one byte, idx as accumulator);
//  * ColorDiff (:74-77), an L1 distance over 4 bytes, is one v_sad_u8;
//  * the bilinear up-sampling (:173-237) works on 16-bit channel pairs 0x00RR00BB / 0x00GG00AA so
//    two channels share each 24-bit multiply-add (max 32*255 = 8160 per lane: no carry between lanes).
#ifndef ICAMD_PVRTC_BLOCK_H_
#define ICAMD_PVRTC_BLOCK_H_

#include "ic_device.h"

namespace icamd {

// A block's two colours after ApplyColorChannelReduction, expanded to channel pairs.
struct PvrtcAB {
  uint32_t a_rb, a_ga, b_rb, b_ga;
};

ICAMD_DEV uint32_t pair_rb(uint32_t c) { return c & 0x00ff00ffu; }
ICAMD_DEV uint32_t pair_ga(uint32_t c) { return (c >> 8) & 0x00ff00ffu; }
ICAMD_DEV uint32_t unpair(uint32_t rb, uint32_t ga) { return rb | ga << 8; }

// ApplyBitDepthReduction (pvrtc.cc:93-106) on one 8-bit channel.
ICAMD_DEV uint32_t bit_depth_reduce(uint32_t v, uint32_t depth) {
  const uint32_t e = v & (0xffu << (8 - depth)) & 0xffu;
  uint32_t r = e | e >> depth;
  if (depth <= 3) r |= e >> (2 * depth);
  return r;
}

// ApplyColorChannelReduction (pvrtc.cc:337-349).  Note the alpha 224..254 promotion: a translucent
// colour whose alpha reduces to 255 keeps its 4/4/3(4)-bit RGB but is later stored as opaque.
ICAMD_DEV uint32_t channel_reduce(uint32_t c, bool is_b) {
  const uint32_t r = bfe(c, 0, 8), g = bfe(c, 8, 8), b = bfe(c, 16, 8), a = c >> 24;
  const bool opaque = a == 255u;
  const uint32_t r5 = bit_depth_reduce(r, 5), g5 = bit_depth_reduce(g, 5);
  const uint32_t r4 = bit_depth_reduce(r, 4), g4 = bit_depth_reduce(g, 4);
  const uint32_t bo = is_b ? bit_depth_reduce(b, 5) : bit_depth_reduce(b, 4);
  const uint32_t bt = is_b ? bit_depth_reduce(b, 4) : bit_depth_reduce(b, 3);
  const uint32_t a3 = bit_depth_reduce(a, 3);
  return opaque ? (r5 | g5 << 8 | bo << 16 | 255u << 24) : (r4 | g4 << 8 | bt << 16 | a3 << 24);
}

// Per-lane 32-dword stash (same idea as BlockStash in dxt_block.h): pixel at a data-dependent index.
#if defined(ICAMD_HOST_EMULATION)
struct Stash32 {
  uint32_t v[32];
  void put(const uint32_t px[32]) { for (int i = 0; i < 32; ++i) v[i] = px[i]; }
  uint32_t get(uint32_t idx) const { return v[idx]; }
};
#else
struct Stash32 {
  uint32_t *base;       // &lds[0][thread][0]
  uint32_t row_dwords;  // threads * 4
  __device__ __forceinline__ void put(const uint32_t px[32]) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<uint4 *>(base + q * row_dwords) = make_uint4(px[4 * q], px[4 * q + 1], px[4 * q + 2], px[4 * q + 3]);
  }
  __device__ __forceinline__ uint32_t get(uint32_t idx) const { return base[(idx >> 2) * row_dwords + (idx & 3u)]; }
};
#endif

// GetExtremesFast (pvrtc.cc:255-329) on a block's 32 pixels px[4*... raster: idx = 8*y + x].
// image0 = pixel 0 of the whole image: the reference initialises every "max" candidate index to 0
// (an IMAGE index, pvrtc.cc:268-269) and only replaces it when a fitness value > 0 is seen.
// Returns the two extreme colours, ordered so that colour A is not brighter than colour B.
ICAMD_DEV void pvrtc_extremes(const uint32_t px[32], uint32_t image0, Stash32 &stash, uint32_t &col_a, uint32_t &col_b) {
  uint32_t kmin[5], kmax[5];
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) { kmin[i] = 0xffffffffu; kmax[i] = 0u; }
  ICAMD_UNROLL
  for (int p = 0; p < 32; ++p) {
    const uint32_t c = px[p];
    // lightness = (77r + 150g + 28b) / 256, then key = lightness*32 + idx
    const uint32_t l32 = (udot4(c, 0x001c964du, 0u) >> 3) & ~31u;
    kmin[0] = umin(kmin[0], l32 | (uint32_t)p);
    kmax[0] = umax(kmax[0], l32 | (uint32_t)(31 - p));
    ICAMD_UNROLL
    for (int ch = 0; ch < 4; ++ch) {
      const uint32_t w = 32u << (8 * ch);
      kmin[ch + 1] = umin(kmin[ch + 1], udot4(c, w, (uint32_t)p));
      kmax[ch + 1] = umax(kmax[ch + 1], udot4(c, w, (uint32_t)(31 - p)));
    }
  }
  stash.put(px);
  uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    const uint32_t lo = stash.get(kmin[i] & 31u);
    const uint32_t hi_block = stash.get(31u - (kmax[i] & 31u));
    const uint32_t hi = (kmax[i] >> 5) == 0u ? image0 : hi_block;  // never-updated max -> image pixel 0
    const uint32_t d = sad_u8(lo, hi, 0u);
    const bool better = (i == 0) || d > best_diff;  // strict '>' scan from best_pair = 0 (pvrtc.cc:309-316)
    best_lo = better ? lo : best_lo;
    best_hi = better ? hi : best_hi;
    best_diff = better ? d : best_diff;
  }
  // ColorBrightnessOrder (pvrtc.cc:240-243, 323-328): swap only if strictly darker
  const uint32_t s_lo = udot4(best_lo, 0x01010101u, 0u), s_hi = udot4(best_hi, 0x01010101u, 0u);
  const bool swap = s_hi < s_lo;
  col_a = swap ? best_hi : best_lo;
  col_b = swap ? best_lo : best_hi;
}

// One channel pair of GetInterpolatedColor2BPP / Interpolate4_2BPP (pvrtc.cc:173-237):
// ((4-yw)(8-xw) c00 + (4-yw) xw c01 + yw (8-xw) c10 + yw xw c11) / 32 on both 16-bit lanes.
ICAMD_DEV uint32_t bilerp_pair(uint32_t c00, uint32_t c01, uint32_t c10, uint32_t c11, uint32_t xw, uint32_t yw) {
  const uint32_t a = (4u - yw) * (8u - xw), b = (4u - yw) * xw, c = yw * (8u - xw), d = yw * xw;
  return ((a * c00 + b * c01 + c * c10 + d * c11) >> 5) & 0x00ff00ffu;
}

// BestModulation (pvrtc.cc:148-166) for one pixel given the up-sampled A and B colours as pairs.
// Scans mod 0..3 and stops at the first step that does not improve (NOT a full argmin).
ICAMD_DEV uint32_t best_modulation(uint32_t pixel, uint32_t a_rb, uint32_t a_ga, uint32_t b_rb, uint32_t b_ga) {
  const uint32_t c0 = unpair(a_rb, a_ga), c3 = unpair(b_rb, b_ga);
  // ApplyModulation (pvrtc.cc:120-144): (5A+3B)/8 and (3A+5B)/8 per channel; <= 2040 per 16-bit lane
  const uint32_t c1 = unpair(((5u * a_rb + 3u * b_rb) >> 3) & 0x00ff00ffu, ((5u * a_ga + 3u * b_ga) >> 3) & 0x00ff00ffu);
  const uint32_t c2 = unpair(((3u * a_rb + 5u * b_rb) >> 3) & 0x00ff00ffu, ((3u * a_ga + 5u * b_ga) >> 3) & 0x00ff00ffu);
  const uint32_t d0 = sad_u8(pixel, c0, 0u), d1 = sad_u8(pixel, c1, 0u);
  const uint32_t d2 = sad_u8(pixel, c2, 0u), d3 = sad_u8(pixel, c3, 0u);
  const bool s1 = d1 < d0, s2 = s1 && d2 < d1, s3 = s2 && d3 < d2;
  return (uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3;
}

// Modulation value of the pixel at in-block position (XI, YI) of a block whose 3x3 block neighbourhood
// of reduced colours is nb[dy+1][dx+1] (toroidal wrap already applied by the caller).
template <int XI, int YI>
ICAMD_DEV uint32_t pvrtc_pixel_mod(uint32_t pixel, const PvrtcAB nb[3][3]) {
  constexpr int x0 = XI < 4 ? 0 : 1, y0 = YI < 2 ? 0 : 1;      // top-left of the 2x2 sources, pvrtc.cc:216-223
  constexpr uint32_t xw = (XI + 4) & 7, yw = (YI + 2) & 3;      // pvrtc.cc:226-227
  const PvrtcAB &c00 = nb[y0][x0], &c01 = nb[y0][x0 + 1], &c10 = nb[y0 + 1][x0], &c11 = nb[y0 + 1][x0 + 1];
  return best_modulation(pixel,
                         bilerp_pair(c00.a_rb, c01.a_rb, c10.a_rb, c11.a_rb, xw, yw),
                         bilerp_pair(c00.a_ga, c01.a_ga, c10.a_ga, c11.a_ga, xw, yw),
                         bilerp_pair(c00.b_rb, c01.b_rb, c10.b_rb, c11.b_rb, xw, yw),
                         bilerp_pair(c00.b_ga, c01.b_ga, c10.b_ga, c11.b_ga, xw, yw));
}

// EncodeColors (pvrtc.cc:356-388); colours are the channel-reduced RGBA dwords.
ICAMD_DEV uint32_t pvrtc_pack_colors(uint32_t ca, uint32_t cb, bool mode_1bpp) {
  const uint32_t ar = bfe(ca, 0, 8), ag = bfe(ca, 8, 8), ab = bfe(ca, 16, 8), aa = ca >> 24;
  const uint32_t br = bfe(cb, 0, 8), bg = bfe(cb, 8, 8), bb = bfe(cb, 16, 8), ba = cb >> 24;
  const uint32_t va = aa == 255u ? (1u << 15 | (ab >> 4) << 1 | (ag >> 3) << 5 | (ar >> 3) << 10)
                                 : ((ab >> 5) << 1 | (ag >> 4) << 4 | (ar >> 4) << 8 | (aa >> 5) << 12);
  const uint32_t vb = ba == 255u ? (1u << 31 | (bb >> 3) << 16 | (bg >> 3) << 21 | (br >> 3) << 26)
                                 : ((bb >> 4) << 16 | (bg >> 4) << 20 | (br >> 4) << 24 | (ba >> 5) << 28);
  return va | vb | (mode_1bpp ? 0u : 1u);
}

// CalculateBlockModulationMode + CalculateBlockModulationData (pvrtc.cc:395-496) for one block.
// rows[y][0..1]: the block's modulation values as bytes (pixel x of row y = byte x&3 of rows[y][x>>2]);
// right_col: byte y = modulation of the pixel right of (7, y); below[0..1]: row below (bytes, x order).
// Returns the 32-bit modulation word; *mode_1bpp tells EncodeColors which flag to store.
ICAMD_DEV uint32_t pvrtc_block_modulation(const uint32_t rows[4][2], uint32_t right_col, const uint32_t below[2],
                                          bool *mode_1bpp) {
  // pixels best served by an intermediate value (1 or 2): low bit xor high bit of each byte
  uint32_t inter = 0, hc = 0, vc = 0;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t r = rows[y][h];
#if defined(ICAMD_HOST_EMULATION)
      inter += (uint32_t)__builtin_popcount((r ^ (r >> 1)) & 0x01010101u);
#else
      inter += (uint32_t)__popc((r ^ (r >> 1)) & 0x01010101u);
#endif
      // "horizontal_count" in the source sums |m - m(x, y+1)|, "vertical_count" |m - m(x+1, y)|
      // (the names are swapped there, pvrtc.cc:426-429; kept as the reference computes them).
      const uint32_t down = y < 3 ? rows[y + 1][h] : below[h];
      hc = sad_u8(r, down, hc);
      // neighbour to the right: bytes shifted by one pixel; the last byte comes from the next
      // dword of the row or from the right-hand block's first column
      const uint32_t next = h == 0 ? rows[y][1] : (bfe(right_col, 8 * y, 8));
      const uint32_t right = alignbit(next, r, 8);
      vc = sad_u8(r, right, vc);
    }
  }
  // modes: 0 = 1BPP, 1 = average-4, 2 = vertical, 3 = horizontal (pvrtc.cc:433-446)
  uint32_t mode = 1u;
  if (inter <= 4u) mode = 0u;
  else if (vc > 10u && vc > hc * 2u) mode = 2u;
  else if (hc > 10u && hc > vc * 2u) mode = 3u;

  uint32_t d1 = 0, d2 = 0;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int x = 0; x < 8; ++x) {
      const uint32_t m = bfe(rows[y][x >> 2], 8 * (x & 3), 8);
      d1 |= (m >> 1) << (8 * y + x);  // 1BPP: one bit per pixel, raster order
      if (((x ^ y) & 1) == 0) {        // 2BPP modes: checkerboard samples, 2 bits each
        const int bitpos = 2 * (4 * y + (x >> 1));
        uint32_t bit = m;
        if (bitpos == 0) bit = mode == 1u ? (m & 2u) : (m | 1u);        // average-4 vs "other"
        else if (bitpos == 20) bit = mode == 2u ? (m | 1u) : (m & 2u);  // vertical vs horizontal
        d2 |= bit << bitpos;
      }
    }
  }
  *mode_1bpp = mode == 0u;
  return mode == 0u ? d1 : d2;
}

// FromZOrder inverse (pvrtc.cc:80-86): x occupies the odd bits, y the even bits of the block index.
ICAMD_DEV uint32_t spread_bits16(uint32_t v) {
  v = (v | v << 8) & 0x00ff00ffu;
  v = (v | v << 4) & 0x0f0f0f0fu;
  v = (v | v << 2) & 0x33333333u;
  v = (v | v << 1) & 0x55555555u;
  return v;
}
ICAMD_DEV uint32_t pvrtc_z_index(uint32_t bx, uint32_t by) { return spread_bits16(bx) << 1 | spread_bits16(by); }

#if defined(ICAMD_HOST_EMULATION)
// Three-pass host driver over the device math above (tests/host_emul only).
template <int XI, int YI>
static inline void emul_mods_xy(const uint32_t px[32], const PvrtcAB nb[3][3], uint8_t mods[32]) {
  mods[8 * YI + XI] = (uint8_t)pvrtc_pixel_mod<XI, YI>(px[8 * YI + XI], nb);
  if constexpr (XI + 1 < 8) emul_mods_xy<XI + 1, YI>(px, nb, mods);
  else if constexpr (YI + 1 < 4) emul_mods_xy<0, YI + 1>(px, nb, mods);
}

static inline int emul_pvrtc2(const uint8_t *src, uint32_t n, uint8_t *out) {
  const uint32_t bw = n / 8, bh = n / 4;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(src);
  PvrtcAB *ab = new PvrtcAB[(size_t)bw * bh];
  uint32_t *ca = new uint32_t[(size_t)bw * bh], *cb = new uint32_t[(size_t)bw * bh];
  uint8_t *mods = new uint8_t[(size_t)n * n];
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      uint32_t px[32];
      for (int i = 0; i < 32; ++i) px[i] = img[(size_t)(by * 4 + i / 8) * n + bx * 8 + i % 8];
      Stash32 st;
      uint32_t a, b;
      pvrtc_extremes(px, img[0], st, a, b);
      a = channel_reduce(a, false);
      b = channel_reduce(b, true);
      ca[by * bw + bx] = a; cb[by * bw + bx] = b;
      PvrtcAB e = { pair_rb(a), pair_ga(a), pair_rb(b), pair_ga(b) };
      ab[by * bw + bx] = e;
    }
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      uint32_t px[32];
      for (int i = 0; i < 32; ++i) px[i] = img[(size_t)(by * 4 + i / 8) * n + bx * 8 + i % 8];
      PvrtcAB nb[3][3];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          nb[dy][dx] = ab[((by + bh + dy - 1) % bh) * bw + (bx + bw + dx - 1) % bw];
      uint8_t m[32];
      emul_mods_xy<0, 0>(px, nb, m);
      for (int i = 0; i < 32; ++i) mods[(size_t)(by * 4 + i / 8) * n + bx * 8 + i % 8] = m[i];
    }
  for (uint32_t by = 0; by < bh; ++by)
    for (uint32_t bx = 0; bx < bw; ++bx) {
      uint32_t rows[4][2], right = 0, below[2] = { 0, 0 };
      for (int y = 0; y < 4; ++y)
        for (int h = 0; h < 2; ++h) {
          rows[y][h] = 0;
          for (int x = 0; x < 4; ++x) rows[y][h] |= (uint32_t)mods[(size_t)(by * 4 + y) * n + bx * 8 + 4 * h + x] << (8 * x);
        }
      for (int y = 0; y < 4; ++y) right |= (uint32_t)mods[(size_t)(by * 4 + y) * n + ((bx * 8 + 8) & (n - 1))] << (8 * y);
      for (int x = 0; x < 8; ++x) below[x >> 2] |= (uint32_t)mods[(size_t)((by * 4 + 4) & (n - 1)) * n + bx * 8 + x] << (8 * (x & 3));
      bool one_bpp;
      const uint32_t data = pvrtc_block_modulation(rows, right, below, &one_bpp);
      const uint32_t colors = pvrtc_pack_colors(ca[by * bw + bx], cb[by * bw + bx], one_bpp);
      uint32_t *o = reinterpret_cast<uint32_t *>(out) + 2 * (size_t)pvrtc_z_index(bx, by);
      o[0] = data;
      o[1] = colors;
    }
  delete[] ab; delete[] ca; delete[] cb; delete[] mods;
  return 1;
}
#endif

}  // namespace icamd
#endif  // ICAMD_PVRTC_BLOCK_H_
