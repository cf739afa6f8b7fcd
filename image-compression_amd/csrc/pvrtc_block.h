// pvrtc_block.h -- PVRTC1 2bpp (8x4-pixel blocks) per-block and per-pixel math.
//
// Bit-exact with internal/pvrtc_compressor.cc (Morph :506-521, Modulate :527-540, Encode :551-580),
// restructured so that a lane owns whole 8x4 blocks with their pixels in VGPRs:
//  * GetExtremesFast (:255-329): the 5 fitness axes' "first minimum / first maximum" become unsigned
//    min / max reductions over keys value*256 + idx  /  value*256 + (31-idx), each key (pair) built by one v_perm_b32:
//    lightness is a v_dot4_u32_u8 with 32-bit keys, the R,B and G,A channels are two 16-bit keys per dword reduced
//    with v_pk_min/max_u16;
//  * ColorDiff (:74-77), an L1 distance over 4 bytes, is one v_sad_u8;
//  * ApplyColorChannelReduction (:337-349) is SWAR on the RGBA dword;
//  * the bilinear up-sampling (:173-237) is separable and incremental on 16-bit channel pairs 0x00RR00BB /
//    0x00GG00AA carried at scale 256 (max 65 280 per lane: no carry between lanes), so the truncated 8-bit
//    channels are the lanes' high bytes; the 5:3 / 3:5 blends (:111-135) are nested v_lerp_u8 byte averages;
//  * pvrtc_encode_strip: one lane walks a vertical strip of blocks, so the row below a block is computed once.
#ifndef ICAMD_PVRTC_BLOCK_H_
#define ICAMD_PVRTC_BLOCK_H_

#include "dxt_block.h"  // pk_lshr16
#include "ic_device.h"
#if defined(ICAMD_HOST_EMULATION)
#include <string.h>
#endif

namespace icamd {

// A block's two colours after ApplyColorChannelReduction, expanded to channel pairs.
struct PvrtcAB {
  uint32_t a_rb, a_ga, b_rb, b_ga;
};
// ... and as the two RGBA dwords the morph kernel stores (8 bytes per block).
struct PvrtcColors {
  uint32_t a, b;
};

ICAMD_DEV uint32_t pair_rb(uint32_t c) { return c & 0x00ff00ffu; }
ICAMD_DEV uint32_t pair_ga(uint32_t c) { return (c >> 8) & 0x00ff00ffu; }
ICAMD_DEV uint32_t unpair(uint32_t rb, uint32_t ga) { return rb | ga << 8; }

// ApplyBitDepthReduction (pvrtc.cc:93-106) on one 8-bit channel: keep the top `depth` bits, replicate them downwards.
constexpr uint32_t bit_depth_reduce(uint32_t v, uint32_t depth) {
  const uint32_t e = v & (0xffu << (8 - depth)) & 0xffu;
  return e | e >> depth | (depth <= 3 ? e >> (2 * depth) : 0u);
}

// ApplyColorChannelReduction (pvrtc.cc:337-349), channel by channel as the reference does it:
//   colour A: opaque R5 G5 B4, translucent R4 G4 B3 A3;   colour B: opaque R5 G5 B5, translucent R4 G4 B4 A3.
// Note the alpha 224..254 promotion: a translucent colour whose alpha reduces to 255 keeps its 4/4/3(4)-bit RGB
// but is later stored as opaque (pvrtc_pack_colors tests the REDUCED alpha).
constexpr uint32_t channel_reduce_by_channel(uint32_t c, bool is_b) {
  const uint32_t r = c & 0xffu, g = (c >> 8) & 0xffu, b = (c >> 16) & 0xffu, a = c >> 24;
  return a == 255u ? (bit_depth_reduce(r, 5) | bit_depth_reduce(g, 5) << 8 | bit_depth_reduce(b, is_b ? 5 : 4) << 16 | 255u << 24)
                   : (bit_depth_reduce(r, 4) | bit_depth_reduce(g, 4) << 8 | bit_depth_reduce(b, is_b ? 4 : 3) << 16 |
                      bit_depth_reduce(a, 3) << 24);
}

// The same on all four channels at once (SWAR on the RGBA dword): the shifted copies are masked so that nothing
// crosses a byte boundary.  ~10 integer ops per colour instead of ~45.
constexpr uint32_t channel_reduce(uint32_t c, bool is_b) {
  uint32_t ro = 0, rt = 0;
  if (is_b) {
    const uint32_t eo = c & 0x00f8f8f8u;
    ro = eo | ((eo >> 5) & 0x00070707u);
    const uint32_t et = c & 0xe0f0f0f0u;
    rt = et | ((et >> 4) & 0x000f0f0fu) | ((et >> 3) & 0x1c000000u) | ((et >> 6) & 0x03000000u);
  } else {
    const uint32_t eo = c & 0x00f0f8f8u;
    ro = eo | ((eo >> 5) & 0x00000707u) | ((eo >> 4) & 0x000f0000u);
    const uint32_t et = c & 0xe0e0f0f0u;
    rt = et | ((et >> 4) & 0x00000f0fu) | ((et >> 3) & 0x1c1c0000u) | ((et >> 6) & 0x03030000u);
  }
  return (c >> 24) == 255u ? (ro | 0xff000000u) : rt;
}
// every value of every channel, next to all-zero and all-one neighbours, for both colours and both alpha classes
constexpr bool channel_reduce_matches_reference() {
  for (uint32_t v = 0; v < 256; ++v)
    for (uint32_t sh = 0; sh < 32; sh += 8)
      for (uint32_t bg = 0; bg < 2; ++bg)
        for (uint32_t alpha_ff = 0; alpha_ff < 2; ++alpha_ff) {
          uint32_t c = ((bg ? 0xffffffffu : 0u) & ~(0xffu << sh)) | v << sh;
          if (alpha_ff) c |= 0xff000000u;
          if (channel_reduce(c, false) != channel_reduce_by_channel(c, false)) return false;
          if (channel_reduce(c, true) != channel_reduce_by_channel(c, true)) return false;
        }
  return true;
}
static_assert(channel_reduce_matches_reference(), "SWAR channel reduction differs from the per-channel form");

// Scheduling fence: keeps hipcc from interleaving independent pixels / rows, which would multiply the live
// registers (the encode kernel wants <= 64 VGPRs; thread-level parallelism covers the latency instead).
#if defined(ICAMD_HOST_EMULATION)
#define ICAMD_SCHED_FENCE() ((void)0)
ICAMD_DEV uint32_t popcount_u32(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
#else
#define ICAMD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
ICAMD_DEV uint32_t popcount_u32(uint32_t v) { return (uint32_t)__popc(v); }
#endif

// Per-lane 32-dword stash (same idea as BlockStash in dxt_block.h): pixel at a data-dependent index.
#if defined(ICAMD_HOST_EMULATION)
struct Stash32 {
  uint32_t v[32];
  void put(const uint32_t px[32]) { for (int i = 0; i < 32; ++i) v[i] = px[i]; }
  uint32_t get(uint32_t idx) const { return v[idx]; }
};
#else
struct Stash32 {
  uint32_t *base;       // the lane's 4 dwords in plane 0
  uint32_t row_dwords;  // distance between the 8 planes
  bool filled = false;  // the kernel already placed the pixels (compile-time constant after inlining)
  __device__ __forceinline__ void put(const uint32_t px[32]) {
    if (filled) return;
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
  // keys: value*256 + p (min side) and value*256 + (31-p) (max side): an unsigned min / max over them is the
  // reference's "first pixel with the strictly smallest / largest value".  A key pair is ONE v_perm_b32: the channel
  // byte of the pixel next to an index byte taken from a register that holds four consecutive indices.  The max-side
  // key is the min-side key plus (31 - 2p): one full-rate add.  The lightness axis uses 32-bit keys (byte 1 of the
  // 16-bit dot product is the reference's (77r + 150g + 28b) / 256); the R,B and G,A axes are two 16-bit keys per
  // dword, reduced with v_pk_min/max_u16.
  uint32_t kmin_l = 0xffffffffu, kmax_l = 0u, kmin_rb = 0xffffffffu, kmax_rb = 0u, kmin_ga = 0xffffffffu, kmax_ga = 0u;
  ICAMD_UNROLL
  for (int p = 0; p < 32; p += 2) {
    uint32_t kl[2];
    ICAMD_UNROLL
    for (int q = 0; q < 2; ++q) {
      const uint32_t c = px[p + q], i = (uint32_t)((p + q) & 3);
      const uint32_t idx4 = (uint32_t)((p + q) & ~3) * 0x01010101u + 0x03020100u;  // bytes: 4 consecutive indices
      const uint32_t up = (uint32_t)(31 - 2 * (p + q)) * 0x00010001u;
      // {hi, lo} = {c or dot, idx4}: selector bytes 0..3 pick an index byte, 4..7 a byte of the pixel
      kl[q] = perm(udot4(c, 0x001c964du, 0u), idx4, 0x0c0c0500u | i);              // [idx, lightness, 0, 0]
      const uint32_t k_rb = perm(c, idx4, 0x06000400u | i | i << 16);              // [idx, R, idx, B]
      const uint32_t k_ga = perm(c, idx4, 0x07000500u | i | i << 16);              // [idx, G, idx, A]
      kmin_rb = pk_min_u16(kmin_rb, k_rb);
      kmin_ga = pk_min_u16(kmin_ga, k_ga);
      kmax_rb = pk_max_u16(kmax_rb, k_rb + up);
      kmax_ga = pk_max_u16(kmax_ga, k_ga + up);
    }
    kmin_l = umin3(kmin_l, kl[0], kl[1]);
    kmax_l = umax3(kmax_l, kl[0] + (uint32_t)(31 - 2 * p), kl[1] + (uint32_t)(31 - 2 * (p + 1)));
    if ((p & 6) == 6) {  // one pixel row at a time: stops the optimiser from regrouping the reductions by axis
      kmin_l = opaque(kmin_l); kmax_l = opaque(kmax_l);  // (which keeps ~64 masked pixel values alive)
      kmin_rb = opaque(kmin_rb); kmax_rb = opaque(kmax_rb);
      kmin_ga = opaque(kmin_ga); kmax_ga = opaque(kmax_ga);
      ICAMD_SCHED_FENCE();
    }
  }
  // axis order of the reference: lightness, R, G, B, A (pvrtc.cc:259-266)
  const uint32_t kmin[5] = { kmin_l, kmin_rb & 0xffffu, kmin_ga & 0xffffu, kmin_rb >> 16, kmin_ga >> 16 };
  const uint32_t kmax[5] = { kmax_l, kmax_rb & 0xffffu, kmax_ga & 0xffffu, kmax_rb >> 16, kmax_ga >> 16 };
  stash.put(px);
  uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    const uint32_t lo = stash.get(kmin[i] & 31u);
    const uint32_t hi_block = stash.get(31u - (kmax[i] & 31u));
    const uint32_t hi = (kmax[i] >> 8) == 0u ? image0 : hi_block;  // never-updated max -> image pixel 0
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

// 8 * ((4-yw)*top + yw*bot) on a channel pair (both 16-bit lanes; <= 8*4*255 per lane).
ICAMD_DEV uint32_t vblend_pair(uint32_t yw, uint32_t top, uint32_t bot) {
  if (yw == 0u) return top << 5;
  if (yw == 2u) return (top + bot) << 4;
  return (yw == 1u ? 3u * top + bot : top + 3u * bot) << 3;
}

// floor((5a + 3b) / 8) per byte as three nested floor-averages: with m = (a+b)>>1,
//   (b + m) >> 1 = floor((a + 3b) / 4)   and   (a + floor((a + 3b) / 4)) >> 1 = floor((5a + 3b) / 8)
// (an integer can be moved inside a floor, and floor(floor(x/2)/2) = floor(x/4)); checked for all 65 536 pairs.
constexpr bool blend53_is_nested_average() {
  for (unsigned a = 0; a < 256; ++a)
    for (unsigned b = 0; b < 256; ++b) {
      const unsigned m = (a + b) >> 1;
      if (((a + ((b + m) >> 1)) >> 1) != (5 * a + 3 * b) / 8) return false;
    }
  return true;
}
static_assert(blend53_is_nested_average(), "(5a+3b)/8 != avg(a, avg(b, avg(a,b)))");

// Modulation value of one pixel from the horizontally accumulated sums P[] = 256 * (up-sampled A_rb, A_ga,
// B_rb, B_ga) -- the reference's truncated 8-bit channels (pvrtc.cc:228-236, sum / 32) are therefore exactly the
// HIGH BYTES of the four 16-bit lanes, and one v_perm_b32 per colour packs them as R,G,B,A.  The two intermediate
// colours (5A+3B)/8 and (3A+5B)/8 (pvrtc.cc:111-135) are nested byte averages (v_lerp_u8, all four channels per
// instruction), the four L1 distances are v_sad_u8.  The value (0..3) is ADDED into `acc` at the byte whose unit
// is `unit` (1, 1<<8, ...).  Same decisions as best_modulation().
#if !defined(ICAMD_HOST_EMULATION) && !defined(ICAMD_PVRTC_NO_SCAN_SDWA)
// The early-exit scan  s1 + (s1 && s2) + (s1 && s2 && s3)  as nested selects  e1 ? (e2 ? (e3 ? 3 : 2) : 1) : 0  on VCC, the
// last select writing byte J of `acc` in place (SDWA dst_sel, the other bytes preserved): 3 v_cmp + 3 v_cndmask and no scalar
// instruction, where the plain expression compiles to 3 v_cmp + 2 s_and_b64 + 2 v_cndmask + v_addc + v_lshl_add (r05: -2 %
// on the one-pass kernel, profiles/r05_ab_pvrtc_onepass.log; -DICAMD_PVRTC_NO_SCAN_SDWA builds the plain form).  The byte of
// `acc` that `unit` addresses must be zero on entry.
ICAMD_DEV uint32_t scan_into_byte(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t unit, uint32_t acc) {
  uint32_t x;
  const uint32_t three = 3u, zero = 0u;
#define ICAMD_SCAN_HEAD                                                                                                   \
  "v_cmp_lt_u32_e32 vcc, %[d3], %[d2]\n\tv_cndmask_b32_e32 %[x], 2, %[three], vcc\n\t"                                   \
  "v_cmp_lt_u32_e32 vcc, %[d2], %[d1]\n\tv_cndmask_b32_e32 %[x], 1, %[x], vcc\n\tv_cmp_lt_u32_e32 vcc, %[d1], %[d0]\n\t"
#define ICAMD_SCAN_TAIL(B)                                                                                                \
  "v_cndmask_b32_sdwa %[acc], %[zero], %[x], vcc dst_sel:" B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
#define ICAMD_SCAN_OPS : [acc] "+v"(acc), [x] "=&v"(x) : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [three] "v"(three), [zero] "v"(zero) : "vcc"
  if (unit == 1u) asm(ICAMD_SCAN_HEAD "v_cndmask_b32_e32 %[acc], 0, %[x], vcc" ICAMD_SCAN_OPS);  // acc == 0: the whole dword
  else if (unit == 1u << 8) asm(ICAMD_SCAN_HEAD ICAMD_SCAN_TAIL("BYTE_1") ICAMD_SCAN_OPS);
  else if (unit == 1u << 16) asm(ICAMD_SCAN_HEAD ICAMD_SCAN_TAIL("BYTE_2") ICAMD_SCAN_OPS);
  else asm(ICAMD_SCAN_HEAD ICAMD_SCAN_TAIL("BYTE_3") ICAMD_SCAN_OPS);
#undef ICAMD_SCAN_HEAD
#undef ICAMD_SCAN_TAIL
#undef ICAMD_SCAN_OPS
  return acc;
}
#else
ICAMD_DEV uint32_t scan_into_byte(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t unit, uint32_t acc) {
  const bool s1 = d1 < d0, s2 = s1 && d2 < d1, s3 = s2 && d3 < d2;  // stop at the first non-improving step
  return acc + ((uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3) * unit;
}
#endif
// ---- two pixels per scan (r06) ----------------------------------------------------------------------------------------------
// The four L1 distances of a pixel are at most 1 020, so TWO pixels' distances share a dword: v_sad_u8 writes the first pixel's
// into the low half, v_sad_hi_u8 ((sad << 16) + accumulator) adds the second pixel's on top -- still one instruction per distance.
// The early-exit scan then runs on both 16-bit lanes at once: the sign bit of d(k+1) - d(k) (v_pk_sub_u16, |difference| < 2^15)
// is "step k + 1 improves", the chain s1, s1 && s2, s1 && s2 && s3 is two ANDs on the raw differences, and the value 0 .. 3 is
// the sum of the three sign bits -- 9 instructions for two pixels where the compare / select chain takes 12, no VCC, no inline
// asm (hipcc follows every asm statement with an s_nop).  Same decisions as best_modulation().  -DICAMD_PVRTC_SCAN_SDWA keeps
// the one-pixel form.
ICAMD_DEV void modulation_colours(const uint32_t P[4], uint32_t c[4]) {
  const uint32_t kSel = 0x07030501u;  // bytes: lo.b1, hi.b1, lo.b3, hi.b3  = R, G, B, A
  c[0] = perm(P[1], P[0], kSel);
  c[3] = perm(P[3], P[2], kSel);
  const uint32_t m = avg_u8(c[0], c[3]);
  c[1] = avg_u8(c[0], avg_u8(c[3], m));
  c[2] = avg_u8(c[3], avg_u8(c[0], m));
}
// d[k]: distances of two pixels to their own colour k, one per 16-bit lane -> the two modulation values, one per lane
ICAMD_DEV uint32_t scan_pair(const uint32_t d[4]) {
  const uint32_t s1 = pk_sub_u16(d[1], d[0]), s2 = pk_sub_u16(d[2], d[1]), s3 = pk_sub_u16(d[3], d[2]);
  const uint32_t s12 = s1 & s2, s123 = s12 & s3;
  return pk_lshr16(s1, 15) + pk_lshr16(s12, 15) + pk_lshr16(s123, 15);
}
ICAMD_DEV uint32_t accumulate_mod(uint32_t pixel, const uint32_t P[4], uint32_t unit, uint32_t acc) {
  const uint32_t kSel = 0x07030501u;  // bytes: lo.b1, hi.b1, lo.b3, hi.b3  = R, G, B, A
  const uint32_t c0 = perm(P[1], P[0], kSel), c3 = perm(P[3], P[2], kSel);
  const uint32_t m = avg_u8(c0, c3);
  const uint32_t c1 = avg_u8(c0, avg_u8(c3, m)), c2 = avg_u8(c3, avg_u8(c0, m));
  const uint32_t d0 = sad_u8(pixel, c0, 0u), d1 = sad_u8(pixel, c1, 0u);
  const uint32_t d2 = sad_u8(pixel, c2, 0u), d3 = sad_u8(pixel, c3, 0u);
  return scan_into_byte(d0, d1, d2, d3, unit, acc);
}

// The 8 modulation values of one pixel row of a block (bytes of row[0..1], x order), and optionally the value of
// the pixel just right of the row (first pixel of the right-hand block).  top[c] / bot[c], c = 0..2: reduced
// colours of the block columns (left, centre, right) in the two block rows that bracket this pixel row;
// yw = vertical weight of `bot` (0..3).  Separable form of pvrtc.cc:173-237: blend the three block columns
// vertically once ((4-yw)*top + yw*bot), then walk each half row with P(xw+1) = P(xw) + (VR - VL):
//   x_in 0..3: sources (left, centre), xw = 4..7, P(4) = 4 (VL + VR)
//   x_in 4..7: sources (centre, right), xw = 0..3, P(0) = 8 VL
// with everything pre-scaled by 8 (vblend_pair) so that P = 256 * colour: 16-bit lanes, max 65 280, no carries;
//   pixel right of the row = x_in 0 of the next block: sources (centre, right), xw = 4
// (a*c00 + b*c01 + c*c10 + d*c11 with a..d = (4-yw)(8-xw), (4-yw)xw, yw(8-xw), yw*xw is exactly
//  (8-xw)*VL + xw*VR; the division by 32 is accumulate_mod's "take the high byte".)
// V[c][v]: 8 * vertical blend of block column c (left, centre, right), v = a_rb, a_ga, b_rb, b_ga
template <bool WITH_RIGHT>
ICAMD_DEV void pvrtc_row_mods_v(const uint32_t V[3][4], const uint32_t *pixels, uint32_t right_pixel, uint32_t row[2],
                                uint32_t *right_mod) {
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    uint32_t P[4], D[4];
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) {
      const uint32_t vl = V[h][v], vr = V[h + 1][v];
      D[v] = vr - vl;
      P[v] = h == 0 ? (vl + vr) << 2 : vl << 3;
    }
    uint32_t acc = 0;
    ICAMD_UNROLL
    for (int j = 0; j < 4; ++j) {
      // opaque(): finish this pixel (compares included) before the next one starts, otherwise the optimiser
      // sinks all eight pixels' decisions to the end of the row and keeps their distances alive until then
      acc = opaque(accumulate_mod(pixels[4 * h + j], P, 1u << (8 * j), acc));
      ICAMD_SCHED_FENCE();
      if (j < 3) {
        ICAMD_UNROLL
        for (int v = 0; v < 4; ++v) P[v] += D[v];
      }
    }
    row[h] = acc;
  }
  if (WITH_RIGHT) {
    uint32_t P[4];
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) P[v] = (V[1][v] + V[2][v]) << 2;
    *right_mod = accumulate_mod(right_pixel, P, 1u, 0u);
  }
}

// The same from the walk's own bases (one-pass kernel, r05): P0 / D0 = first value and step of the left half row (x_in 0..3,
// sources left | centre), P1 / D1 of the right half row (centre | right).  Both are linear in the vertical weight, so the
// strip walk steps THEM from pixel row to pixel row (16 adds) instead of stepping the three column blends and re-deriving
// P and D in every row (12 + 16).  The bases are left untouched: pixel j uses base + j * step built by three-operand adds.
ICAMD_DEV void pvrtc_row_mods_pd(const uint32_t P0[4], const uint32_t D0[4], const uint32_t P1[4], const uint32_t D1[4],
                                 const uint32_t *pixels, uint32_t row[2]) {
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    const uint32_t *Pb = h ? P1 : P0, *D = h ? D1 : D0;
    uint32_t P[4] = { Pb[0], Pb[1], Pb[2], Pb[3] };
    uint32_t acc = 0;
    ICAMD_UNROLL
    for (int j = 0; j < 4; ++j) {
      acc = opaque(accumulate_mod(pixels[4 * h + j], P, 1u << (8 * j), acc));
      ICAMD_SCHED_FENCE();
      if (j < 3) {
        ICAMD_UNROLL
        for (int v = 0; v < 4; ++v) P[v] += D[v];
      }
    }
    row[h] = acc;
  }
}

// ---- the walk on 64-bit register pairs (r06) ----------------------------------------------------------------------------
// The four sums of a walk are two (rb, ga) word pairs; as ONE 64-bit integer each -- word v at bits 32 (v & 1) -- a pair steps
// with one v_lshl_add_u64 (4.4 clocks at two waves per SIMD against 2 x 3.5 for two v_add_u32 next to half-rate instructions:
// scripts/ubench_u64.hip).  Exact: every quantity of the walk is a LINEAR function of the colours, 16-bit lanes of a word may be
// negative on the way (steps, differences), so the whole chain is computed modulo 2^64 -- borrows cross the word boundary exactly
// as they cross the lane boundary inside a word in the 32-bit form -- and the values that are READ (the sums at the pixels) have
// all four lanes in 0 .. 65 280, so their words are the 32-bit form's words.
// -DICAMD_PVRTC_WALK32 builds the 32-bit form (A/B; profiles/r06_ab_pvrtc_walk64.log: 16 x 4096^2 0.3887 -> 0.3769 ms).
#if !defined(ICAMD_PVRTC_WALK32) && !defined(ICAMD_PVRTC_WALK64)
#define ICAMD_PVRTC_WALK64 1
#endif
typedef unsigned long long icamd_u64;
#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV icamd_u64 pack64(uint32_t lo, uint32_t hi) { return (icamd_u64)hi << 32 | lo; }
ICAMD_DEV uint32_t lo32(icamd_u64 v) { return (uint32_t)v; }
ICAMD_DEV uint32_t hi32(icamd_u64 v) { return (uint32_t)(v >> 32); }
#else
// (as a two-element vector: hipcc then keeps the pair in one aligned register pair whose halves are written in place; the
// shift-and-or form is canonicalised to zext(lo) + (hi << 32) and a pair add becomes v_lshl_add_u64 + v_add_u32)
typedef uint32_t icamd_u32x2 __attribute__((ext_vector_type(2)));
ICAMD_DEV icamd_u64 pack64(uint32_t lo, uint32_t hi) {
  const icamd_u32x2 v = { lo, hi };
  return __builtin_bit_cast(icamd_u64, v);
}
ICAMD_DEV uint32_t lo32(icamd_u64 v) { return __builtin_bit_cast(icamd_u32x2, v).x; }
ICAMD_DEV uint32_t hi32(icamd_u64 v) { return __builtin_bit_cast(icamd_u32x2, v).y; }
#endif
// the pair of two SIGNED words (each below 2^31 in magnitude, given modulo 2^32) as hi * 2^32 + lo modulo 2^64
ICAMD_DEV icamd_u64 pack64_signed(uint32_t lo, uint32_t hi) { return pack64(lo, hi + (uint32_t)((int32_t)lo >> 31)); }
template <int S>
ICAMD_DEV icamd_u64 shl_add64(icamd_u64 a, icamd_u64 b) {  // (a << S) + b, S = 0 .. 4
  static_assert(S >= 0 && S <= 4, "v_lshl_add_u64 shifts by at most 4");
#if defined(ICAMD_HOST_EMULATION) || !defined(ICAMD_PVRTC_WALK64_ASM)
  return (a << S) + b;  // (hipcc selects v_lshl_add_u64 for it on gfx950 and, unlike after an asm, knows which hazards it has)
#else
  icamd_u64 r;
  asm("v_lshl_add_u64 %0, %1, %3, %2" : "=v"(r) : "v"(a), "v"(b), "n"(S));
  return r;
#endif
}
ICAMD_DEV icamd_u64 add64(icamd_u64 a, icamd_u64 b) { return shl_add64<0>(a, b); }
ICAMD_DEV icamd_u64 opaque64(icamd_u64 v) {
#if !defined(ICAMD_HOST_EMULATION)
  asm volatile("" : "+v"(v));
#endif
  return v;
}
// pvrtc_row_mods_pd with the bases as pairs: P*[p] = words (2 p, 2 p + 1) of the 32-bit form
ICAMD_DEV void pvrtc_row_mods_pd64(const icamd_u64 P0[2], const icamd_u64 D0[2], const icamd_u64 P1[2], const icamd_u64 D1[2],
                                   const uint32_t *pixels, uint32_t row[2]) {
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    const icamd_u64 *Pb = h ? P1 : P0, *D = h ? D1 : D0;
    icamd_u64 Q[2] = { Pb[0], Pb[1] };
#if defined(ICAMD_PVRTC_SCAN_SDWA)
    uint32_t acc = 0;
    ICAMD_UNROLL
    for (int j = 0; j < 4; ++j) {
      const uint32_t P[4] = { lo32(Q[0]), hi32(Q[0]), lo32(Q[1]), hi32(Q[1]) };
      acc = opaque(accumulate_mod(pixels[4 * h + j], P, 1u << (8 * j), acc));
      ICAMD_SCHED_FENCE();
      if (j < 3) {
        Q[0] = add64(Q[0], D[0]);
        Q[1] = add64(Q[1], D[1]);
      }
    }
    row[h] = acc;
#else
    // pixels (0, 2) and (1, 3) of the half row share their scans: the values land in bytes 0, 2 of one word and, shifted, 1, 3
    uint32_t d[2][4], val[2] = { 0u, 0u };
    ICAMD_UNROLL
    for (int j = 0; j < 4; ++j) {
      const uint32_t P[4] = { lo32(Q[0]), hi32(Q[0]), lo32(Q[1]), hi32(Q[1]) };
      uint32_t c[4];
      modulation_colours(P, c);
      const uint32_t px = pixels[4 * h + j];
      ICAMD_UNROLL
      for (int k = 0; k < 4; ++k) d[j & 1][k] = j < 2 ? sad_u8(px, c[k], 0u) : sad_hi_u8(px, c[k], d[j & 1][k]);
      if (j >= 2) val[j & 1] = opaque(scan_pair(d[j & 1]));
      else {
        ICAMD_UNROLL
        for (int k = 0; k < 4; ++k) d[j][k] = opaque(d[j][k]);
      }
      ICAMD_SCHED_FENCE();
      if (j < 3) {
        Q[0] = add64(Q[0], D[0]);
        Q[1] = add64(Q[1], D[1]);
      }
    }
    row[h] = val[0] | val[1] << 8;
#endif
  }
}

template <bool WITH_RIGHT>
ICAMD_DEV void pvrtc_row_mods(uint32_t yw, const PvrtcAB top[3], const PvrtcAB bot[3], const uint32_t *pixels,
                              uint32_t right_pixel, uint32_t row[2], uint32_t *right_mod) {
  uint32_t V[3][4];
  ICAMD_UNROLL
  for (int c = 0; c < 3; ++c) {
    V[c][0] = vblend_pair(yw, top[c].a_rb, bot[c].a_rb);
    V[c][1] = vblend_pair(yw, top[c].a_ga, bot[c].a_ga);
    V[c][2] = vblend_pair(yw, top[c].b_rb, bot[c].b_rb);
    V[c][3] = vblend_pair(yw, top[c].b_ga, bot[c].b_ga);
  }
  pvrtc_row_mods_v<WITH_RIGHT>(V, pixels, right_pixel, row, right_mod);
}

ICAMD_DEV PvrtcAB pvrtc_expand(const PvrtcColors &c) {
  PvrtcAB e = { pair_rb(c.a), pair_ga(c.a), pair_rb(c.b), pair_ga(c.b) };
  return e;
}
// the same from packed RGBA colours
template <bool WITH_RIGHT>
ICAMD_DEV void pvrtc_row_mods(uint32_t yw, const PvrtcColors top[3], const PvrtcColors bot[3], const uint32_t *pixels,
                              uint32_t right_pixel, uint32_t row[2], uint32_t *right_mod) {
  PvrtcAB t[3], b[3];
  ICAMD_UNROLL
  for (int c = 0; c < 3; ++c) {
    t[c] = pvrtc_expand(top[c]);
    b[c] = pvrtc_expand(bot[c]);
  }
  pvrtc_row_mods<WITH_RIGHT>(yw, t, b, pixels, right_pixel, row, right_mod);
}

// All modulation values a block's encoding depends on, from its 3x3 block neighbourhood nb (toroidal wrap
// applied by the caller): its own 32 (rows[y][h]: byte x&3 of rows[y][x>>2] = pixel (x, y)), the pixel column
// right of it (right_col: byte y; right_px[y] = first pixel of row y of the right-hand block) and the pixel row
// below it (below[0..1]; below_px[0..7] = first pixel row of the block below).  CalculateBlockModulationMode
// looks one pixel right and one pixel down (pvrtc.cc:416-429), so these 12 extra values make the block
// self-contained: no exchange with other lanes is needed.
ICAMD_DEV void pvrtc_block_mods(const uint32_t px[32], const uint32_t right_px[4], const uint32_t below_px[8],
                                const PvrtcColors nb[3][3], uint32_t rows[4][2], uint32_t *right_col, uint32_t below[2]) {
  uint32_t rc = 0;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const int y0 = y < 2 ? 0 : 1;
    uint32_t m;
    pvrtc_row_mods<true>((uint32_t)((y + 2) & 3), nb[y0], nb[y0 + 1], &px[8 * y], right_px[y], rows[y], &m);
    rc |= m << (8 * y);
  }
  *right_col = rc;
  // first pixel row of the block below: y_in = 0 there -> block rows (centre, below), weight 2
  pvrtc_row_mods<false>(2u, nb[1], nb[2], below_px, 0u, below, nullptr);
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
  uint32_t inter = 0, hc = 0, vc = 0, d1 = 0, d2 = 0;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      const uint32_t r = rows[y][h];
      // pixels best served by an intermediate value (1 or 2): low bit xor high bit of each byte
#if defined(ICAMD_HOST_EMULATION)
      inter += (uint32_t)__builtin_popcount((r ^ (r >> 1)) & 0x01010101u);
#else
      inter += (uint32_t)__popc((r ^ (r >> 1)) & 0x01010101u);
#endif
      // "horizontal_count" in the source sums |m - m(x, y+1)|, "vertical_count" |m - m(x+1, y)|
      // (the names are swapped there, pvrtc.cc:426-429; kept as the reference computes them).
      const uint32_t down = y < 3 ? rows[y + 1][h] : below[h];
      hc = sad_u8(r, down, hc);
      // neighbour to the right: bytes shifted by one pixel; the last byte comes from the next dword of the
      // row or from the right-hand block's first column
      const uint32_t next = h == 0 ? rows[y][1] : (bfe(right_col, 8 * y, 8));
      vc = sad_u8(r, alignbit(next, r, 8), vc);
      // 1BPP word: bit 8y+x = m >> 1.  The four high bits of a dword's bytes are gathered into a nibble by
      // one multiply (bit 8j+1 -> bit 24+j; no two partial products collide below bit 28).
      const int pos = 8 * y + 4 * h;
      d1 |= ((((r >> 1) & 0x01010101u) * 0x01020408u) >> 24) << pos;
      // 2BPP word: checkerboard samples ((x^y)&1 == 0), 2 bits each, in raster order: bytes 0,2 of the dword on
      // even rows, bytes 1,3 on odd rows -> one nibble at the same position as the 1BPP nibble.
      const uint32_t v = ((y & 1) ? r >> 8 : r) & 0x00030003u;
      d2 |= ((v | v >> 14) & 0xfu) << pos;
    }
  }
  // modes: 0 = 1BPP, 1 = average-4, 2 = vertical, 3 = horizontal (pvrtc.cc:433-446)
  uint32_t mode = 1u;
  if (inter <= 4u) mode = 0u;
  else if (vc > 10u && vc > hc * 2u) mode = 2u;
  else if (hc > 10u && hc > vc * 2u) mode = 3u;
  // The samples at bit 0 (0,0) and bit 20 (4,2) keep only their high bit; the low bit selects the sub-mode
  // (pvrtc.cc:474-487): bit 0 = "not average-4", bit 20 = "vertical".
  d2 = mode == 1u ? (d2 & ~1u) : (d2 | 1u);
  d2 = mode == 2u ? (d2 | 1u << 20) : (d2 & ~(1u << 20));
  *mode_1bpp = mode == 0u;
  return mode == 0u ? d1 : d2;
}

// Row-streaming form of pvrtc_block_mods + pvrtc_block_modulation: the block is consumed one pixel row at a
// time (rows 0..3 of the block, then the first row of the block below), so only one row of pixels, its 9
// modulation values and a handful of counters are live at any moment -- this is what keeps the encode kernel
// at <= 64 VGPRs (8 waves per SIMD), where the full-rate add/and/shift instructions actually pay off.
// load(r, pixels[8], &right): r = 0..3 -> pixel row r of the block and the pixel right of it;
//                             r = 4    -> first pixel row of the block below (right unused).

template <typename RowLoader>
ICAMD_DEV void pvrtc_encode_block_rows(RowLoader &load, const PvrtcColors nb[3][3], uint32_t *data_out,
                                       bool *mode_1bpp) {
  uint32_t inter = 0, hc = 0, vc = 0, d1 = 0, d2 = 0, prev[2] = { 0, 0 };
  uint32_t cur[8], cur_right = 0, nxt[8], nxt_right = 0;
  load(0, cur, &cur_right);
  ICAMD_UNROLL
  for (int r = 0; r < 5; ++r) {
    if (r < 4) load(r + 1, nxt, &nxt_right);  // prefetch the next row while this one is processed
    ICAMD_SCHED_FENCE();
    uint32_t row[2], right_mod = 0;
    if (r < 4) {
      const int y0 = r < 2 ? 0 : 1;
      pvrtc_row_mods<true>((uint32_t)((r + 2) & 3), nb[y0], nb[y0 + 1], cur, cur_right, row, &right_mod);
    } else {
      pvrtc_row_mods<false>(2u, nb[1], nb[2], cur, 0u, row, nullptr);  // y_in = 0 of the block below
    }
    if (r > 0) {  // "horizontal_count" = sum |m - m(x, y+1)| (names swapped in the source, pvrtc.cc:426-429)
      hc = sad_u8(prev[0], row[0], hc);
      hc = sad_u8(prev[1], row[1], hc);
    }
    if (r < 4) {
      // "vertical_count" = sum |m - m(x+1, y)|: bytes shifted by one pixel, last one from the right-hand block
      vc = sad_u8(row[0], alignbit(row[1], row[0], 8), vc);
      vc = sad_u8(row[1], alignbit(right_mod, row[1], 8), vc);
      ICAMD_UNROLL
      for (int h = 0; h < 2; ++h) {
        const uint32_t m = row[h];
        inter += popcount_u32((m ^ (m >> 1)) & 0x01010101u);  // values 1 or 2: low bit xor high bit
        const int pos = 8 * r + 4 * h;
        d1 |= ((((m >> 1) & 0x01010101u) * 0x01020408u) >> 24) << pos;  // 1BPP: bit 8y+x = m >> 1
        const uint32_t v = ((r & 1) ? m >> 8 : m) & 0x00030003u;         // 2BPP: checkerboard samples
        d2 |= ((v | v >> 14) & 0xfu) << pos;
      }
      prev[0] = row[0];
      prev[1] = row[1];
      ICAMD_UNROLL
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
      cur_right = nxt_right;
    }
    ICAMD_SCHED_FENCE();
  }
  uint32_t mode = 1u;  // 0 = 1BPP, 1 = average-4, 2 = vertical, 3 = horizontal (pvrtc.cc:433-446)
  if (inter <= 4u) mode = 0u;
  else if (vc > 10u && vc > hc * 2u) mode = 2u;
  else if (hc > 10u && hc > vc * 2u) mode = 3u;
  d2 = mode == 1u ? (d2 & ~1u) : (d2 | 1u);                 // pvrtc.cc:474-487
  d2 = mode == 2u ? (d2 | 1u << 20) : (d2 & ~(1u << 20));
  *mode_1bpp = mode == 0u;
  *data_out = mode == 0u ? d1 : d2;
}

// ---- strip form: one lane encodes K vertically adjacent blocks of one block column ---------------------------------
// Walking down a column, the pixel row below a block IS row 0 of the next block, so the "below" halo row of
// pvrtc_encode_block_rows (8 of its 12 redundant modulation values plus one row set-up) is computed once instead of
// twice; only the last block of a strip still pays for it.  A block is finished (its mode decided, its words stored)
// right after row 0 of the block under it.  44 -> 36 + 8/K modulation values per block.
struct PvrtcBlockAcc {
  uint32_t hc, vc, d1, d2;
  uint32_t u01, u23;    // rows (0, 1) / (2, 3): byte x = m(x, y) | m(x, y + 1) << 2 | m(x + 4, y) << 4 | m(x + 4, y + 1) << 6
  uint32_t col0, col7;  // EXCHANGE only: byte y = modulation of pixel (0, y) / (7, y) of the block
};
// one pixel row (y = 0..3, compile-time after unrolling) of a block: everything except the vertical differences.
// The bit gathers of CalculateBlockModulationData (pvrtc.cc:456-496) are v_dot4_u32_u8 with power-of-two weights: the
// modulation values sit one per byte, so  sum_x (m_x & 2) * 2^x  is twice the row's eight 1BPP bits, and
// sum_j m_(2j + odd row) * 4^j  its four checkerboard samples (2 bits each) -- one dot per half row instead of a
// shift-mask-multiply-shift chain (r03).
// EXCHANGE: the value right of the row is not computed here -- the term |m(7, y) - m(8, y)| is added when the block is
// finished, from the right-hand neighbour's own column-0 values (pvrtc_encode_strip); the row only records its two
// outer values.
template <bool EXCHANGE>
ICAMD_DEV void pvrtc_acc_row(PvrtcBlockAcc &A, int y, const uint32_t row[2], uint32_t right_mod) {
  A.vc = sad_u8(row[0], alignbit(row[1], row[0], 8), A.vc);   // "vertical_count" = sum |m - m(x+1, y)| (pvrtc.cc:426-429)
  if (EXCHANGE) {
    A.vc = sad_u8(row[1], perm(row[1], row[1], 0x03030201u), A.vc);  // bytes (5, 6, 7, 7): the last term is 0 here
    // byte y of col0 / col7 <- byte 0 of row[0] / byte 3 of row[1]; selector 4 + i keeps byte i of the old value
    const uint32_t keep = 0x07060504u & ~(0xffu << (8 * y));
    A.col0 = perm(A.col0, row[0], keep);
    A.col7 = perm(A.col7, row[1], keep | 0x03u << (8 * y));
  } else {
    A.vc = sad_u8(row[1], alignbit(right_mod, row[1], 8), A.vc);
  }
  // both half rows in one word (r06): byte x = m(x) | m(x + 4) << 4 (at most 51), so that ONE dot product per gather sees
  // all eight values -- a weight w on byte x is w on m(x) and 16 w on m(x + 4), which is what both gathers want
  const uint32_t u = row[0] | row[1] << 4;
  // 1BPP word: bit 8y + x = m >> 1;  sum_x (2 hi(x) + 32 hi(x + 4)) 2^x = twice the row's eight bits
  const uint32_t twice = udot4(u & 0x22222222u, 0x08040201u, 0u);
  A.d1 |= y == 0 ? twice >> 1 : twice << (8 * y - 1);
  // 2BPP word: the samples with (x ^ y) & 1 == 0, 2 bits each, raster order -> byte y
  A.d2 |= udot4(u, (y & 1) ? 0x04000100u : 0x00040001u, 0u) << (8 * y);
  // values 1 or 2 are counted at the end, two rows per dword (fields at bits (0, 1), (4, 5) of a byte | the next row's << 2)
  if (y == 0) A.u01 = u;
  else if (y == 1) A.u01 |= u << 2;
  else if (y == 2) A.u23 = u;
  else A.u23 |= u << 2;
}
ICAMD_DEV uint32_t pvrtc_acc_finish(const PvrtcBlockAcc &A, bool *mode_1bpp) {
  // pixels best served by an intermediate value (1 or 2): low bit xor high bit of each 2-bit field
  const uint32_t inter = popcount_u32((A.u01 ^ (A.u01 >> 1)) & 0x55555555u) + popcount_u32((A.u23 ^ (A.u23 >> 1)) & 0x55555555u);
  uint32_t mode = 1u;  // 0 = 1BPP, 1 = average-4, 2 = vertical, 3 = horizontal (pvrtc.cc:433-446)
  if (inter <= 4u) mode = 0u;
  else if (A.vc > 10u && A.vc > A.hc * 2u) mode = 2u;
  else if (A.hc > 10u && A.hc > A.vc * 2u) mode = 3u;
  uint32_t d2 = mode == 1u ? (A.d2 & ~1u) : (A.d2 | 1u);  // pvrtc.cc:474-487
  d2 = mode == 2u ? (d2 | 1u << 20) : (d2 & ~(1u << 20));
  *mode_1bpp = mode == 0u;
  return mode == 0u ? A.d1 : d2;
}

// Modulation value of the pixel at x_in = 0, row y_in (0..3, a RUN-TIME value) of a block, from the reduced colours of
// the four blocks its interpolation uses: columns (left neighbour, own) x block rows (upper, lower), where (upper,
// lower) = (by - 1, by) for y_in < 2 and (by, by + 1) otherwise (pvrtc.cc:216-227).  With xw = 4 the four bilinear
// weights are 4 (4 - yw), 4 (4 - yw), 4 yw, 4 yw, so the /32 of Interpolate4_2BPP is an exact >> 3 of
// (4 - yw)(c00 + c01) + yw (c10 + c11) (<= 2040 per 16-bit lane).  Used once per strip by the encode kernel for the
// column right of each wave (the lanes in between get these values from their right-hand neighbour lane).
ICAMD_DEV uint32_t pvrtc_left_edge_mod(uint32_t pixel, uint32_t y_in, const PvrtcColors &ul, const PvrtcColors &uc,
                                       const PvrtcColors &ll, const PvrtcColors &lc) {
  const uint32_t yw = (y_in + 2u) & 3u, uw = 4u - yw;
  const uint32_t a_rb = ((uw * (pair_rb(ul.a) + pair_rb(uc.a)) + yw * (pair_rb(ll.a) + pair_rb(lc.a))) >> 3) & 0x00ff00ffu;
  const uint32_t a_ga = ((uw * (pair_ga(ul.a) + pair_ga(uc.a)) + yw * (pair_ga(ll.a) + pair_ga(lc.a))) >> 3) & 0x00ff00ffu;
  const uint32_t b_rb = ((uw * (pair_rb(ul.b) + pair_rb(uc.b)) + yw * (pair_rb(ll.b) + pair_rb(lc.b))) >> 3) & 0x00ff00ffu;
  const uint32_t b_ga = ((uw * (pair_ga(ul.b) + pair_ga(uc.b)) + yw * (pair_ga(ll.b) + pair_ga(lc.b))) >> 3) & 0x00ff00ffu;
  return best_modulation(pixel, a_rb, a_ga, b_rb, b_ga);
}

// load_px(r, pixels[8], &right): pixel row r of the strip, r = 0 .. 4 K (row 4 K = first row of the block below the
//                                strip), and the pixel right of it; toroidal wrap is the loader's business.
// load_colours(j, c[3]):         reduced colours of block row j of the strip (j = -1 .. K), columns left/centre/right.
// store(j, data, mode_1bpp, own): block j of the strip is finished; own = its reduced colours.
// EXCHANGE: the modulation values right of a block (pvrtc.cc:426-429 looks one pixel right) are not computed by the
//   lane -- 4 of the 37 values a block costs -- but fetched when block j is finished:
// right_of(j, col0):             given this lane's column-0 values of block j (byte y = row y), returns those of the
//                                block to its right.  On the device consecutive lanes are consecutive block columns
//                                walking the same rows in lock-step, so this is a one-lane shuffle (the last lane of a
//                                wave reads values its workgroup computed up front with pvrtc_left_edge_mod).
//
// The walk is organised by COLOUR-ROW PAIRS, not by blocks: rows 2, 3 of block s-1 and rows 0, 1 of block s all
// interpolate between the colours of block rows s-1 (A) and s (B), with vertical weights 0, 1, 2, 3
// (pvrtc.cc:216-227).  So the twelve vertical blends 8 ((4 - yw) A + yw B) of a pixel row are set up once per four
// rows (32 A, and the step 8 (B - A)) and then just stepped -- twelve full-rate adds per row instead of re-expanding
// six colours and re-blending them; two pixel-row buffers alternate, so no row is ever copied.
template <bool EXCHANGE, typename PixelRowLoader, typename ColourRowLoader, typename BlockStore, typename RightOf>
ICAMD_DEV void pvrtc_encode_strip(uint32_t k_blocks, PixelRowLoader &load_px, ColourRowLoader &load_colours,
                                  BlockStore &store, RightOf &right_of) {
  PvrtcColors cc[3];
  uint32_t A[3][4];  // colour row s-1 as channel pairs
  load_colours(-1, cc);
  ICAMD_UNROLL
  for (int c = 0; c < 3; ++c) {
    A[c][0] = pair_rb(cc[c].a); A[c][1] = pair_ga(cc[c].a); A[c][2] = pair_rb(cc[c].b); A[c][3] = pair_ga(cc[c].b);
  }
  load_colours(0, cc);
  uint32_t buf0[8], buf1[8], right0 = 0, right1 = 0, prev[2] = { 0, 0 };
  ICAMD_UNROLL
  for (int i = 0; i < 8; ++i) buf1[i] = 0;
  load_px(0u, buf0, &right0);
  PvrtcBlockAcc acc = { 0, 0, 0, 0, 0, 0, 0, 0 };
  PvrtcColors own = cc[1];
  ICAMD_NOUNROLL
  for (uint32_t s = 0;; ++s) {
    // colour rows (s-1, s): V = 32 A, dV = 8 (B - A).  Plain 32-bit arithmetic on the 16-bit channel pairs: every
    // intermediate V is a true blend with both lanes in [0, 8160], so borrows between the lanes cancel exactly.
    uint32_t V[3][4], dV[3][4];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c) {
      const uint32_t b[4] = { pair_rb(cc[c].a), pair_ga(cc[c].a), pair_rb(cc[c].b), pair_ga(cc[c].b) };
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        V[c][v] = A[c][v] << 5;
        dV[c][v] = (b[v] - A[c][v]) << 3;
        A[c][v] = b[v];
      }
    }
    const PvrtcColors own_next = cc[1];
    if (s < k_blocks) load_colours((int)s + 1, cc);  // next segment's colours, in flight during these rows
    uint32_t row[2], right_mod = 0;
    if (s > 0) {
      // rows 2 and 3 of block s-1: weights 0 and 1
      load_px(4u * s - 1u, buf1, &right1);
      ICAMD_SCHED_FENCE();
      pvrtc_row_mods_v<!EXCHANGE>(V, buf0, right0, row, &right_mod);
      acc.hc = sad_u8(prev[0], row[0], acc.hc);  // "horizontal_count" = sum |m - m(x, y+1)| (pvrtc.cc:426-429)
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      pvrtc_acc_row<EXCHANGE>(acc, 2, row, right_mod);
      prev[0] = row[0]; prev[1] = row[1];
      ICAMD_UNROLL
      for (int c = 0; c < 3; ++c)
        ICAMD_UNROLL
        for (int v = 0; v < 4; ++v) V[c][v] += dV[c][v];
      ICAMD_SCHED_FENCE();
      load_px(4u * s, buf0, &right0);
      ICAMD_SCHED_FENCE();
      pvrtc_row_mods_v<!EXCHANGE>(V, buf1, right1, row, &right_mod);
      acc.hc = sad_u8(prev[0], row[0], acc.hc);
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      pvrtc_acc_row<EXCHANGE>(acc, 3, row, right_mod);
      prev[0] = row[0]; prev[1] = row[1];
      ICAMD_UNROLL
      for (int c = 0; c < 3; ++c)
        ICAMD_UNROLL
        for (int v = 0; v < 4; ++v) V[c][v] += dV[c][v];
      ICAMD_SCHED_FENCE();
    } else {
      ICAMD_UNROLL
      for (int c = 0; c < 3; ++c)
        ICAMD_UNROLL
        for (int v = 0; v < 4; ++v) V[c][v] += 2u * dV[c][v];  // the strip starts at weight 2
    }
    // row 0 of block s, weight 2 -- for s == k_blocks the row below the strip, which only completes block K-1
    if (s < k_blocks) load_px(4u * s + 1u, buf1, &right1);
    ICAMD_SCHED_FENCE();
    pvrtc_row_mods_v<!EXCHANGE>(V, buf0, right0, row, &right_mod);
    if (s > 0) {  // the vertical differences across the block boundary, then block s-1 is complete
      acc.hc = sad_u8(prev[0], row[0], acc.hc);
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      if (EXCHANGE) acc.vc = sad_u8(acc.col7, right_of(s - 1u, acc.col0), acc.vc);  // sum_y |m(7, y) - m(8, y)|
      bool one_bpp;
      const uint32_t data = pvrtc_acc_finish(acc, &one_bpp);
      store(s - 1u, data, one_bpp, own);
    }
    if (s == k_blocks) break;
    own = own_next;
    acc.hc = acc.vc = acc.d1 = acc.d2 = 0;  // (u01 / u23 / col0 / col7 are overwritten piece by piece)
    pvrtc_acc_row<EXCHANGE>(acc, 0, row, right_mod);
    prev[0] = row[0]; prev[1] = row[1];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c)
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) V[c][v] += dV[c][v];
    ICAMD_SCHED_FENCE();
    // row 1 of block s, weight 3
    load_px(4u * s + 2u, buf0, &right0);
    ICAMD_SCHED_FENCE();
    pvrtc_row_mods_v<!EXCHANGE>(V, buf1, right1, row, &right_mod);
    acc.hc = sad_u8(prev[0], row[0], acc.hc);
    acc.hc = sad_u8(prev[1], row[1], acc.hc);
    pvrtc_acc_row<EXCHANGE>(acc, 1, row, right_mod);
    prev[0] = row[0]; prev[1] = row[1];
    ICAMD_SCHED_FENCE();
  }
}

// ---- one-pass form (r05): the lane that encodes a block column also MORPHS it ----------------------------------------
// GetExtremesFast (pvrtc.cc:255-329) consumed one pixel row at a time: the same keys as pvrtc_extremes, the block's four
// rows arriving in four calls (Q = row inside the block, compile-time), the ten data-dependent pixel look-ups of the final
// scan batched into one call of `lookup10` (on the device: ten ds_read_b32 from the pixel-row ring under one wait).
// The (rb, ga) "first maximum" keys as ONE running pair (r06).  The plain form is max over pixels p of k_p + up_p with
// up_p = (N - 1 - 2 p) per 16-bit lane (N = 32 or 16 pixels): two word adds per pixel.  With M_p = (that maximum up to p) - up_p,
//   M_0 = k_0,   M_p = max(M_(p-1) + 2, k_p)   per lane,
// the SAME constant is added every time, to the running value instead of the new key -- so (M_rb, M_ga) steps as a 64-bit pair
// with one v_lshl_add_u64 (see "the walk on 64-bit register pairs": at two waves per SIMD 4.4 clocks against 2 x 3.5).  Lanes
// stay in 0 .. 65 280 + 2 N: M_p >= k_p >= 0 and the maximum is at most 65 280 + N - 1.  The keys the scan reads are
// M_(N-1) + up_(N-1) = M - (N - 1) per lane (pvrtc_keys_max_words).  -DICAMD_PVRTC_KEYS_UP builds the plain form.
struct PvrtcMorphKeys {
  uint32_t min_l, max_l, min_rb, min_ga;
#if defined(ICAMD_PVRTC_KEYS_UP)
  uint32_t max_rb, max_ga;
#else
  icamd_u64 max_pair;  // M_rb | M_ga << 32; first assigned by pixel 0 of a block
#endif
};
ICAMD_DEV void pvrtc_keys_reset(PvrtcMorphKeys &k) {
  k.min_l = k.min_rb = k.min_ga = 0xffffffffu;
  k.max_l = 0u;
#if defined(ICAMD_PVRTC_KEYS_UP)
  k.max_rb = k.max_ga = 0u;
#else
  k.max_pair = 0u;
#endif
}
// one pixel's (rb, ga) keys into the running maxima; P = pixel index in the block (compile-time after unrolling), N = pixels
template <int N>
ICAMD_DEV void pvrtc_keys_max_step(PvrtcMorphKeys &k, int P, uint32_t k_rb, uint32_t k_ga) {
#if defined(ICAMD_PVRTC_KEYS_UP)
  const uint32_t up = (uint32_t)(N - 1 - 2 * P) * 0x00010001u;
  k.max_rb = pk_max_u16(k.max_rb, k_rb + up);
  k.max_ga = pk_max_u16(k.max_ga, k_ga + up);
#else
  if (P == 0) {
    k.max_pair = pack64(k_rb, k_ga);
  } else {
    const icamd_u64 m = add64(k.max_pair, pack64(0x00020002u, 0x00020002u));
    k.max_pair = pack64(pk_max_u16(lo32(m), k_rb), pk_max_u16(hi32(m), k_ga));
  }
#endif
}
template <int N>
ICAMD_DEV void pvrtc_keys_max_words(const PvrtcMorphKeys &k, uint32_t &max_rb, uint32_t &max_ga) {
#if defined(ICAMD_PVRTC_KEYS_UP)
  max_rb = k.max_rb;
  max_ga = k.max_ga;
#else
  max_rb = lo32(k.max_pair) - (uint32_t)(N - 1) * 0x00010001u;
  max_ga = hi32(k.max_pair) - (uint32_t)(N - 1) * 0x00010001u;
#endif
}
ICAMD_DEV void pvrtc_keys_opaque(PvrtcMorphKeys &k) {
  k.min_l = opaque(k.min_l); k.max_l = opaque(k.max_l);
  k.min_rb = opaque(k.min_rb); k.min_ga = opaque(k.min_ga);
#if defined(ICAMD_PVRTC_KEYS_UP)
  k.max_rb = opaque(k.max_rb); k.max_ga = opaque(k.max_ga);
#else
  k.max_pair = opaque64(k.max_pair);
#endif
}
template <int Q>
ICAMD_DEV void pvrtc_keys_row(PvrtcMorphKeys &k, const uint32_t px[8]) {
  ICAMD_UNROLL
  for (int x = 0; x < 8; x += 2) {
    uint32_t kl[2];
    ICAMD_UNROLL
    for (int q = 0; q < 2; ++q) {
      const int p = 8 * Q + x + q;
      const uint32_t c = px[x + q], i = (uint32_t)(p & 3);
      const uint32_t idx4 = (uint32_t)(p & ~3) * 0x01010101u + 0x03020100u;
      kl[q] = perm(udot4(c, 0x001c964du, 0u), idx4, 0x0c0c0500u | i);
      const uint32_t k_rb = perm(c, idx4, 0x06000400u | i | i << 16);
      const uint32_t k_ga = perm(c, idx4, 0x07000500u | i | i << 16);
      // (pixel 0 of a block ASSIGNS the running keys: the reset values -- all ones / zero -- never win against a key)
      k.min_rb = p == 0 ? k_rb : pk_min_u16(k.min_rb, k_rb);
      k.min_ga = p == 0 ? k_ga : pk_min_u16(k.min_ga, k_ga);
      pvrtc_keys_max_step<32>(k, p, k_rb, k_ga);
    }
    const int p = 8 * Q + x;
    k.min_l = p == 0 ? umin(kl[0], kl[1]) : umin3(k.min_l, kl[0], kl[1]);
    k.max_l = p == 0 ? umax(kl[0] + 31u, kl[1] + 29u)
                     : umax3(k.max_l, kl[0] + (uint32_t)(31 - 2 * p), kl[1] + (uint32_t)(31 - 2 * (p + 1)));
  }
  pvrtc_keys_opaque(k);
  ICAMD_SCHED_FENCE();
}
// lookup10(idx[10], out[10]): out[i] = pixel idx[i] (0..31, raster inside the block) of the block whose rows were just consumed
template <typename Lookup10>
ICAMD_DEV void pvrtc_keys_finish(const PvrtcMorphKeys &k, uint32_t image0, Lookup10 &lookup10, uint32_t &col_a, uint32_t &col_b) {
  uint32_t max_rb, max_ga;
  pvrtc_keys_max_words<32>(k, max_rb, max_ga);
  const uint32_t kmin[5] = { k.min_l, k.min_rb & 0xffffu, k.min_ga & 0xffffu, k.min_rb >> 16, k.min_ga >> 16 };
  const uint32_t kmax[5] = { k.max_l, max_rb & 0xffffu, max_ga & 0xffffu, max_rb >> 16, max_ga >> 16 };
  uint32_t idx[10], v[10];
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    idx[2 * i] = kmin[i] & 31u;
    idx[2 * i + 1] = 31u - (kmax[i] & 31u);
  }
  lookup10(idx, v);
  uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    const uint32_t lo = v[2 * i];
    const uint32_t hi = (kmax[i] >> 8) == 0u ? image0 : v[2 * i + 1];  // never-updated max -> image pixel 0 (pvrtc.cc:268-269)
    const uint32_t d = sad_u8(lo, hi, 0u);
    const bool better = (i == 0) || d > best_diff;
    best_lo = better ? lo : best_lo;
    best_hi = better ? hi : best_hi;
    best_diff = better ? d : best_diff;
  }
  const uint32_t s_lo = udot4(best_lo, 0x01010101u, 0u), s_hi = udot4(best_hi, 0x01010101u, 0u);
  const bool swap = s_hi < s_lo;
  col_a = swap ? best_hi : best_lo;
  col_b = swap ? best_lo : best_hi;
}

// One lane = one block column of a strip of K blocks (block rows 0 .. K-1 of the strip), walking pixel rows -4 .. 4 K + 3:
// every row is consumed twice from the same row ring -- by the morph when it arrives (row m) and by the modulation five
// rows later (row e = m - 5): rows 2, 3 of block s-1 and rows 0, 1 of block s interpolate between colour rows s-1 and s
// (pvrtc.cc:216-227), so block s must be morphed (its last row is 4 s + 3) before row 4 s - 2 is modulated.  A "tick"
// hands over both rows.  Per SEGMENT s = -1 .. K+1 (four ticks):
//   tick(4 s + 3):  last row of block s -> its two colours;  exchange(): the colours of the block columns left and right
//                   (neighbour lanes; on the device wave-edge lanes go through LDS, which is where the workgroup's one
//                   barrier per segment sits) and, riding on the same barrier, the column-0 modulation values of the block
//                   right of block s-2 -- which is why a block is finished one segment late (block j in segment j+2):
//                   its last term  sum_y |m(7, y) - m(8, y)|  (pvrtc.cc:426-429) needs the right-hand lane's values;
//   rows 4 s - 2, 4 s - 1 (block s-1 rows 2, 3), 4 s (block s row 0, closes block s-1's vertical differences), 4 s + 1.
// tick(m, mp, ep):       pixel rows m (morph) and m - 5 (modulation) of the strip; wrap and clamping are the caller's.
// lookup10(idx, out):    see pvrtc_keys_finish; refers to the block whose last row the latest tick delivered.
// exchange(s, own, col0, left, right, right_col0): own = colours of block row s of this column, col0 = this lane's column-0
//                        values of block s-2; returns the colours left / right of `own` and the column-0 values of the block
//                        right of block s-2.
// store(j, data, one_bpp, own): block j of the strip is finished.
// K must be >= 1; segments -1 and K+1 only morph / only finish.
template <typename Tick, typename Lookup10, typename Exchange, typename BlockStore>
ICAMD_DEV void pvrtc_onepass_strip(uint32_t k_blocks, uint32_t image0, Tick &tick, Lookup10 &lookup10, Exchange &exchange,
                                   BlockStore &store) {
  const int K = (int)k_blocks;
  PvrtcMorphKeys keys;
  pvrtc_keys_reset(keys);
  uint32_t mp[8], ep[8];
  uint32_t A[3][4];  // colour row s-1 as channel pairs
  ICAMD_UNROLL
  for (int c = 0; c < 3; ++c)
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) A[c][v] = 0u;
  PvrtcBlockAcc acc = { 0, 0, 0, 0, 0, 0, 0, 0 }, def = { 0, 0, 0, 0, 0, 0, 0, 0 };
  PvrtcColors own_acc = { 0u, 0u }, own_def = { 0u, 0u };
  uint32_t prev[2] = { 0u, 0u };
  tick(-4, mp, ep); pvrtc_keys_row<0>(keys, mp);
  tick(-3, mp, ep); pvrtc_keys_row<1>(keys, mp);
  tick(-2, mp, ep); pvrtc_keys_row<2>(keys, mp);
  PvrtcColors cc[3] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
#if defined(ICAMD_PVRTC_WALK64)
  icamd_u64 P0[2] = { 0u, 0u }, D0[2] = { 0u, 0u }, P1[2] = { 0u, 0u }, D1[2] = { 0u, 0u };  // the walks' bases, carried
#endif
  ICAMD_NOUNROLL
  for (int s = -1;; ++s) {
    if (s <= K) {
      tick(4 * s + 3, mp, ep);
      pvrtc_keys_row<3>(keys, mp);
      uint32_t a, c;
      pvrtc_keys_finish(keys, image0, lookup10, a, c);
      cc[1].a = channel_reduce(a, false);
      cc[1].b = channel_reduce(c, true);
      pvrtc_keys_reset(keys);
    }
    uint32_t right_col0 = 0u;
    exchange(s, cc[1], def.col0, cc[0], cc[2], right_col0);
    if (s >= 2) {
      def.vc = sad_u8(def.col7, right_col0, def.vc);  // sum_y |m(7, y) - m(8, y)|
      bool one_bpp;
      const uint32_t data = pvrtc_acc_finish(def, &one_bpp);
      store((uint32_t)(s - 2), data, one_bpp, own_def);
    }
    if (s > K) break;
    // colour rows (s-1, s): V = 32 A, dV = 8 (B - A) -- see pvrtc_encode_strip
#if defined(ICAMD_PVRTC_WALK64)
    // The bases of the horizontal walks and their steps per pixel row, straight from the colour rows A (s-1) and B (s), E = B - A:
    //   left half row:  D = V[1] - V[0] = 32 (A1 - A0),  P = 4 (V[0] + V[1]) = 128 (A0 + A1);   per row + 8 (E1 - E0), + 32 (E0 + E1)
    //   right half row: D = V[2] - V[1] = 32 (A2 - A1),  P = 8 V[1] = 256 A1;                   per row + 8 (E2 - E1), + 64 E1
    // (sums / shifts commute modulo 2^32: the same words as deriving them from V and dV, 15 instead of 19 instructions per
    // channel pair), then as 64-bit pairs (pvrtc_row_mods_pd64) made exact modulo 2^64: the steps are SIGNED quantities below
    // 2^31 in magnitude per word (lanes of at most 16 320), so the pair's high word owes the low word's sign --
    // hi + (lo >> 31, arithmetic); the P bases have non-negative lanes and need nothing.
    // The bases themselves are CARRIED from segment to segment: four row steps lead from colour row s-1 to colour row s, so
    // a fourth ICAMD_ROW_STEP at the end of the segment leaves exactly the next segment's bases (32 (B1 - B0) = 32 (A1 - A0)
    // + 4 * 8 (E1 - E0), ...; zero before the first segment, like A) -- 8 pair adds instead of deriving them from A again.
    icamd_u64 dP0[2], dD0[2], dP1[2], dD1[2];
    {
      uint32_t ep0[4], ed0[4], ep1[4], ed1[4];
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        uint32_t e[3];
        ICAMD_UNROLL
        for (int c = 0; c < 3; ++c) {
          const uint32_t b = v == 0 ? pair_rb(cc[c].a) : v == 1 ? pair_ga(cc[c].a) : v == 2 ? pair_rb(cc[c].b) : pair_ga(cc[c].b);
          e[c] = b - A[c][v];
          A[c][v] = b;
        }
        ed0[v] = (e[1] - e[0]) << 3;
        ep0[v] = (e[0] + e[1]) << 5;
        ed1[v] = (e[2] - e[1]) << 3;
        ep1[v] = e[1] << 6;
      }
      ICAMD_UNROLL
      for (int p = 0; p < 2; ++p) {
        dD0[p] = pack64_signed(ed0[2 * p], ed0[2 * p + 1]); dD1[p] = pack64_signed(ed1[2 * p], ed1[2 * p + 1]);
        dP0[p] = pack64_signed(ep0[2 * p], ep0[2 * p + 1]); dP1[p] = pack64_signed(ep1[2 * p], ep1[2 * p + 1]);
      }
    }
#define ICAMD_ROW_MODS(px_, row_) pvrtc_row_mods_pd64(P0, D0, P1, D1, px_, row_)
#define ICAMD_ROW_STEP()                                                                                     \
  ICAMD_UNROLL                                                                                               \
  for (int p = 0; p < 2; ++p) {                                                                              \
    P0[p] = add64(P0[p], dP0[p]); D0[p] = add64(D0[p], dD0[p]);                                              \
    P1[p] = add64(P1[p], dP1[p]); D1[p] = add64(D1[p], dD1[p]);                                              \
  }
#else
    uint32_t V[3][4], dV[3][4];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c) {
      const uint32_t b[4] = { pair_rb(cc[c].a), pair_ga(cc[c].a), pair_rb(cc[c].b), pair_ga(cc[c].b) };
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        V[c][v] = A[c][v] << 5;
        dV[c][v] = (b[v] - A[c][v]) << 3;
        A[c][v] = b[v];
      }
    }
#if !defined(ICAMD_PVRTC_ONEPASS_STEP_V)
    // ... and from them the bases of the horizontal walks and THEIR steps per pixel row (pvrtc_row_mods_pd; everything is
    // linear in the vertical weight, and sums / shifts commute modulo 2^32, so stepping these equals re-deriving them):
    //   left half row:  D = V[1] - V[0],  P = 4 (V[0] + V[1]);     right half row:  D = V[2] - V[1],  P = 8 V[1]
    uint32_t P0[4], D0[4], P1[4], D1[4], dP0[4], dD0[4], dP1[4], dD1[4];
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) {
      D0[v] = V[1][v] - V[0][v];      dD0[v] = dV[1][v] - dV[0][v];
      P0[v] = (V[0][v] + V[1][v]) << 2; dP0[v] = (dV[0][v] + dV[1][v]) << 2;
      D1[v] = V[2][v] - V[1][v];      dD1[v] = dV[2][v] - dV[1][v];
      P1[v] = V[1][v] << 3;           dP1[v] = dV[1][v] << 3;
    }
#define ICAMD_ROW_MODS(px_, row_) pvrtc_row_mods_pd(P0, D0, P1, D1, px_, row_)
#define ICAMD_ROW_STEP()                                                                             \
  ICAMD_UNROLL                                                                                       \
  for (int v = 0; v < 4; ++v) { P0[v] += dP0[v]; D0[v] += dD0[v]; P1[v] += dP1[v]; D1[v] += dD1[v]; }
#else
#define ICAMD_ROW_MODS(px_, row_) pvrtc_row_mods_v<false>(V, px_, 0u, row_, nullptr)
#define ICAMD_ROW_STEP()                                      \
  ICAMD_UNROLL                                                \
  for (int c = 0; c < 3; ++c)                                 \
    ICAMD_UNROLL                                              \
    for (int v = 0; v < 4; ++v) V[c][v] += dV[c][v];
#endif
#endif
    uint32_t row[2];
    if (s >= 1) {  // row 2 of block s-1, weight 0
      ICAMD_ROW_MODS(ep, row);
      acc.hc = sad_u8(prev[0], row[0], acc.hc);  // "horizontal_count" = sum |m - m(x, y+1)| (pvrtc.cc:426-429)
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      pvrtc_acc_row<true>(acc, 2, row, 0u);
      prev[0] = row[0]; prev[1] = row[1];
    }
    ICAMD_ROW_STEP()
    ICAMD_SCHED_FENCE();
    tick(4 * s + 4, mp, ep);
    pvrtc_keys_row<0>(keys, mp);
    if (s >= 1) {  // row 3 of block s-1, weight 1
      ICAMD_ROW_MODS(ep, row);
      acc.hc = sad_u8(prev[0], row[0], acc.hc);
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      pvrtc_acc_row<true>(acc, 3, row, 0u);
      prev[0] = row[0]; prev[1] = row[1];
    }
    ICAMD_ROW_STEP()
    ICAMD_SCHED_FENCE();
    tick(4 * s + 5, mp, ep);
    pvrtc_keys_row<1>(keys, mp);
    if (s >= 0) {  // row 0 of block s, weight 2 -- for s == K the row below the strip, which only completes block K-1
      ICAMD_ROW_MODS(ep, row);
      if (s >= 1) {
        acc.hc = sad_u8(prev[0], row[0], acc.hc);
        acc.hc = sad_u8(prev[1], row[1], acc.hc);
        def = acc;  // complete but for the right-hand column: finished in the next segment
        own_def = own_acc;
      }
      own_acc = cc[1];
      acc.hc = acc.vc = acc.d1 = acc.d2 = 0u;
      pvrtc_acc_row<true>(acc, 0, row, 0u);
      prev[0] = row[0]; prev[1] = row[1];
    }
    ICAMD_ROW_STEP()
    ICAMD_SCHED_FENCE();
    tick(4 * s + 6, mp, ep);
    pvrtc_keys_row<2>(keys, mp);
    if (s >= 0 && s < K) {  // row 1 of block s, weight 3
      ICAMD_ROW_MODS(ep, row);
      acc.hc = sad_u8(prev[0], row[0], acc.hc);
      acc.hc = sad_u8(prev[1], row[1], acc.hc);
      pvrtc_acc_row<true>(acc, 1, row, 0u);
      prev[0] = row[0]; prev[1] = row[1];
    }
#if defined(ICAMD_PVRTC_WALK64)
    ICAMD_ROW_STEP()  // weight 4 = colour row s itself = the next segment's weight 0
#endif
    ICAMD_SCHED_FENCE();
  }
}
#undef ICAMD_ROW_MODS
#undef ICAMD_ROW_STEP

// ---- PVRTC1 4 bpp (r05): EXTENSION, PARITY UNPINNED -- BASELINE.json's config 5 names "PVRTC 4bpp", the reference only
// has 2 bpp (public/pvrtc_compressor.h:15-18).  The 2 bpp rules above with 4 x 4-pixel blocks, exactly as
// oracle/ic_oracle.c (pvrtc4_encode_image) restates them: GetExtremesFast over 16 pixels, the same channel reduction,
// BestModulation against A / B up-sampled with weights (x + 2) & 3, (y + 2) & 3 out of 4 in both directions, every pixel's
// 2-bit value stored at bits 2 (4 y + x), colour word with bit 0 clear.
// GetExtremesFast (pvrtc.cc:255-329) on px[4 y + x]: the keys of pvrtc_extremes with 4-bit indices.
ICAMD_DEV void pvrtc4_extremes(const uint32_t px[16], uint32_t image0, BlockStash &stash, uint32_t &col_a, uint32_t &col_b) {
  uint32_t kmin_l = 0xffffffffu, kmax_l = 0u, kmin_rb = 0xffffffffu, kmax_rb = 0u, kmin_ga = 0xffffffffu, kmax_ga = 0u;
  ICAMD_UNROLL
  for (int p = 0; p < 16; p += 2) {
    uint32_t kl[2];
    ICAMD_UNROLL
    for (int q = 0; q < 2; ++q) {
      const uint32_t c = px[p + q], i = (uint32_t)((p + q) & 3);
      const uint32_t idx4 = (uint32_t)((p + q) & ~3) * 0x01010101u + 0x03020100u;
      const uint32_t up = (uint32_t)(15 - 2 * (p + q)) * 0x00010001u;  // max-side key = value * 256 + (15 - idx)
      kl[q] = perm(udot4(c, 0x001c964du, 0u), idx4, 0x0c0c0500u | i);
      const uint32_t k_rb = perm(c, idx4, 0x06000400u | i | i << 16), k_ga = perm(c, idx4, 0x07000500u | i | i << 16);
      kmin_rb = pk_min_u16(kmin_rb, k_rb);
      kmin_ga = pk_min_u16(kmin_ga, k_ga);
      kmax_rb = pk_max_u16(kmax_rb, k_rb + up);
      kmax_ga = pk_max_u16(kmax_ga, k_ga + up);
    }
    kmin_l = umin3(kmin_l, kl[0], kl[1]);
    kmax_l = umax3(kmax_l, kl[0] + (uint32_t)(15 - 2 * p), kl[1] + (uint32_t)(15 - 2 * (p + 1)));
  }
  const uint32_t kmin[5] = { kmin_l, kmin_rb & 0xffffu, kmin_ga & 0xffffu, kmin_rb >> 16, kmin_ga >> 16 };
  const uint32_t kmax[5] = { kmax_l, kmax_rb & 0xffffu, kmax_ga & 0xffffu, kmax_rb >> 16, kmax_ga >> 16 };
  stash.put(px);
  uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    const uint32_t lo = stash.get(kmin[i] & 15u);
    const uint32_t hi_block = stash.get(15u - (kmax[i] & 15u));
    const uint32_t hi = (kmax[i] >> 8) == 0u ? image0 : hi_block;  // never-updated max -> image pixel 0 (pvrtc.cc:268-269)
    const uint32_t d = sad_u8(lo, hi, 0u);
    const bool better = (i == 0) || d > best_diff;
    best_lo = better ? lo : best_lo;
    best_hi = better ? hi : best_hi;
    best_diff = better ? d : best_diff;
  }
  const bool swap = udot4(best_hi, 0x01010101u, 0u) < udot4(best_lo, 0x01010101u, 0u);
  col_a = swap ? best_hi : best_lo;
  col_b = swap ? best_lo : best_hi;
}

// The block's 32-bit modulation word from its pixels and the reduced colours of its 3 x 3 block neighbourhood (toroidal wrap
// applied by the caller).  Separable like the 2 bpp walk: per pixel row the three block columns are blended vertically,
// V = 4 ((4 - yw) top + yw bottom), then each half row walks P(xw + 1) = P(xw) + 4 (VR - VL) from P = 8 (VL + VR) (x = 0, 1:
// left | centre, xw = 2, 3) or P = 16 VL (x = 2, 3: centre | right, xw = 0, 1) -- P = 256 x colour on 16-bit lanes
// (<= 65 280), so accumulate_mod's "take the high bytes" is the oracle's sum / 16.
ICAMD_DEV uint32_t pvrtc4_block_data(const uint32_t px[16], const PvrtcColors nb[3][3]) {
  uint32_t C[3][3][4];
  ICAMD_UNROLL
  for (int r = 0; r < 3; ++r)
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c) {
      C[r][c][0] = pair_rb(nb[r][c].a); C[r][c][1] = pair_ga(nb[r][c].a);
      C[r][c][2] = pair_rb(nb[r][c].b); C[r][c][3] = pair_ga(nb[r][c].b);
    }
  uint32_t data = 0;
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    const int r0 = y < 2 ? 0 : 1;
    const uint32_t yw = (uint32_t)((y + 2) & 3);
    uint32_t V[3][4];
    ICAMD_UNROLL
    for (int c = 0; c < 3; ++c)
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) V[c][v] = vblend_pair(yw, C[r0][c][v], C[r0 + 1][c][v]) >> 1;  // 8 x blend -> 4 x blend
    uint32_t acc = 0;
    ICAMD_UNROLL
    for (int h = 0; h < 2; ++h) {
      uint32_t P[4], D[4];
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        const uint32_t vl = V[h][v], vr = V[h + 1][v];
        D[v] = (vr - vl) << 2;
        P[v] = h == 0 ? (vl + vr) << 3 : vl << 4;
      }
      acc = opaque(accumulate_mod(px[4 * y + 2 * h], P, 1u << (16 * h), acc));
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) P[v] += D[v];
      acc = opaque(accumulate_mod(px[4 * y + 2 * h + 1], P, 1u << (16 * h + 8), acc));
    }
    data |= udot4(acc, 0x40100401u, 0u) << (8 * y);  // bytes (values 0..3) -> four 2-bit fields
  }
  return data;
}

// One-pass form of the 4 bpp encoder (r05), the 2 bpp walk of pvrtc_onepass_strip with 4-pixel rows: one lane = one 4-pixel
// block column of a strip of K blocks.  The vertical structure is the 2 bpp one (blocks are 4 rows tall in both formats: rows
// 2, 3 of block s-1 and rows 0, 1 of block s interpolate between colour rows s-1 and s with weights 0, 1, 2, 3), so a tick
// again hands over pixel row m for the morph and row m - 5 for the modulation.  What falls away: the mode decision and its
// neighbour terms (every pixel's value is stored), hence no deferred finish and no column exchange -- block s-1 is complete
// after its row 3 in segment s.
//   tick(m, mp[4], ep[4]); lookup10 as in pvrtc_keys_finish (indices 0..15); exchange(s, own, left, right): colours only;
//   store(j, data, own).
template <int Q>
ICAMD_DEV void pvrtc4_keys_row(PvrtcMorphKeys &k, const uint32_t px[4]) {
  ICAMD_UNROLL
  for (int x = 0; x < 4; x += 2) {
    uint32_t kl[2];
    ICAMD_UNROLL
    for (int q = 0; q < 2; ++q) {
      const int p = 4 * Q + x + q;
      const uint32_t c = px[x + q], i = (uint32_t)(p & 3);
      const uint32_t idx4 = (uint32_t)(p & ~3) * 0x01010101u + 0x03020100u;
      kl[q] = perm(udot4(c, 0x001c964du, 0u), idx4, 0x0c0c0500u | i);
      const uint32_t k_rb = perm(c, idx4, 0x06000400u | i | i << 16), k_ga = perm(c, idx4, 0x07000500u | i | i << 16);
      k.min_rb = p == 0 ? k_rb : pk_min_u16(k.min_rb, k_rb);
      k.min_ga = p == 0 ? k_ga : pk_min_u16(k.min_ga, k_ga);
      pvrtc_keys_max_step<16>(k, p, k_rb, k_ga);
    }
    const int p = 4 * Q + x;
    k.min_l = p == 0 ? umin(kl[0], kl[1]) : umin3(k.min_l, kl[0], kl[1]);
    k.max_l = p == 0 ? umax(kl[0] + 15u, kl[1] + 13u)
                     : umax3(k.max_l, kl[0] + (uint32_t)(15 - 2 * p), kl[1] + (uint32_t)(15 - 2 * (p + 1)));
  }
  pvrtc_keys_opaque(k);
  ICAMD_SCHED_FENCE();
}
template <typename Lookup10>
ICAMD_DEV void pvrtc4_keys_finish(const PvrtcMorphKeys &k, uint32_t image0, Lookup10 &lookup10, uint32_t &col_a, uint32_t &col_b) {
  uint32_t max_rb, max_ga;
  pvrtc_keys_max_words<16>(k, max_rb, max_ga);
  const uint32_t kmin[5] = { k.min_l, k.min_rb & 0xffffu, k.min_ga & 0xffffu, k.min_rb >> 16, k.min_ga >> 16 };
  const uint32_t kmax[5] = { k.max_l, max_rb & 0xffffu, max_ga & 0xffffu, max_rb >> 16, max_ga >> 16 };
  uint32_t idx[10], v[10];
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    idx[2 * i] = kmin[i] & 15u;
    idx[2 * i + 1] = 15u - (kmax[i] & 15u);
  }
  lookup10(idx, v);
  uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
  ICAMD_UNROLL
  for (int i = 0; i < 5; ++i) {
    const uint32_t lo = v[2 * i];
    const uint32_t hi = (kmax[i] >> 8) == 0u ? image0 : v[2 * i + 1];
    const uint32_t d = sad_u8(lo, hi, 0u);
    const bool better = (i == 0) || d > best_diff;
    best_lo = better ? lo : best_lo;
    best_hi = better ? hi : best_hi;
    best_diff = better ? d : best_diff;
  }
  const bool swap = udot4(best_hi, 0x01010101u, 0u) < udot4(best_lo, 0x01010101u, 0u);
  col_a = swap ? best_hi : best_lo;
  col_b = swap ? best_lo : best_hi;
}
// the four values of one pixel row as the row's 8 data bits (pixel x at bits 2 x)
ICAMD_DEV uint32_t pvrtc4_row_bits(const uint32_t P0[4], const uint32_t D0[4], const uint32_t P1[4], const uint32_t D1[4],
                                   const uint32_t px[4]) {
  uint32_t acc = 0;
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    const uint32_t *Pb = h ? P1 : P0, *D = h ? D1 : D0;
    uint32_t P[4] = { Pb[0], Pb[1], Pb[2], Pb[3] };
    acc = opaque(accumulate_mod(px[2 * h], P, 1u << (16 * h), acc));
    ICAMD_SCHED_FENCE();
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) P[v] += D[v];
    acc = opaque(accumulate_mod(px[2 * h + 1], P, 1u << (16 * h + 8), acc));
    ICAMD_SCHED_FENCE();
  }
  return udot4(acc, 0x40100401u, 0u);
}
// ... with the bases as 64-bit pairs (see pvrtc_row_mods_pd64)
ICAMD_DEV uint32_t pvrtc4_row_bits64(const icamd_u64 P0[2], const icamd_u64 D0[2], const icamd_u64 P1[2], const icamd_u64 D1[2],
                                     const uint32_t px[4]) {
  // (one-pixel scans here: at this kernel's four waves per SIMD the compare / select chain is the cheaper one -- the two-pixel
  // form of pvrtc_row_mods_pd64, -DICAMD_PVRTC4_SCAN_PAIR, measured 0.4167 -> 0.4244 ms on 16 x 4096^2)
#if !defined(ICAMD_PVRTC4_SCAN_PAIR)
  uint32_t acc = 0;
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    const icamd_u64 *Pb = h ? P1 : P0, *D = h ? D1 : D0;
    icamd_u64 Q[2] = { Pb[0], Pb[1] };
    ICAMD_UNROLL
    for (int j = 0; j < 2; ++j) {
      const uint32_t P[4] = { lo32(Q[0]), hi32(Q[0]), lo32(Q[1]), hi32(Q[1]) };
      acc = opaque(accumulate_mod(px[2 * h + j], P, 1u << (16 * h + 8 * j), acc));
      ICAMD_SCHED_FENCE();
      if (j == 0) {
        Q[0] = add64(Q[0], D[0]);
        Q[1] = add64(Q[1], D[1]);
      }
    }
  }
  return udot4(acc, 0x40100401u, 0u);
#else
  // pixels (0, 2) and (1, 3) of the row share their scans (scan_pair): values in bytes 0, 2 and, shifted, 1, 3
  uint32_t d[2][4], val[2] = { 0u, 0u };
  ICAMD_UNROLL
  for (int h = 0; h < 2; ++h) {
    const icamd_u64 *Pb = h ? P1 : P0, *D = h ? D1 : D0;
    icamd_u64 Q[2] = { Pb[0], Pb[1] };
    ICAMD_UNROLL
    for (int j = 0; j < 2; ++j) {
      const uint32_t P[4] = { lo32(Q[0]), hi32(Q[0]), lo32(Q[1]), hi32(Q[1]) };
      uint32_t c[4];
      modulation_colours(P, c);
      const uint32_t pixel = px[2 * h + j];
      ICAMD_UNROLL
      for (int k = 0; k < 4; ++k) d[j][k] = h == 0 ? sad_u8(pixel, c[k], 0u) : sad_hi_u8(pixel, c[k], d[j][k]);
      if (h == 1) val[j] = opaque(scan_pair(d[j]));
      else {
        ICAMD_UNROLL
        for (int k = 0; k < 4; ++k) d[j][k] = opaque(d[j][k]);
      }
      ICAMD_SCHED_FENCE();
      if (j == 0) {
        Q[0] = add64(Q[0], D[0]);
        Q[1] = add64(Q[1], D[1]);
      }
    }
  }
  return udot4(val[0] | val[1] << 8, 0x40100401u, 0u);
#endif
}
template <typename Tick, typename Lookup10, typename Exchange, typename BlockStore>
ICAMD_DEV void pvrtc4_onepass_strip(uint32_t k_blocks, uint32_t image0, Tick &tick, Lookup10 &lookup10, Exchange &exchange,
                                    BlockStore &store) {
  const int K = (int)k_blocks;
  PvrtcMorphKeys keys;
  pvrtc_keys_reset(keys);
  uint32_t mp[4], ep[4];
  uint32_t A[3][4];
  ICAMD_UNROLL
  for (int c = 0; c < 3; ++c)
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) A[c][v] = 0u;
  uint32_t data = 0u;
  PvrtcColors own_acc = { 0u, 0u };
  tick(-4, mp, ep); pvrtc4_keys_row<0>(keys, mp);
  tick(-3, mp, ep); pvrtc4_keys_row<1>(keys, mp);
  tick(-2, mp, ep); pvrtc4_keys_row<2>(keys, mp);
  PvrtcColors cc[3] = { { 0u, 0u }, { 0u, 0u }, { 0u, 0u } };
#if defined(ICAMD_PVRTC_WALK64)
  icamd_u64 P0[2] = { 0u, 0u }, D0[2] = { 0u, 0u }, P1[2] = { 0u, 0u }, D1[2] = { 0u, 0u };  // the walks' bases, carried
#endif
  ICAMD_NOUNROLL
  for (int s = -1;; ++s) {
    {
      tick(4 * s + 3, mp, ep);
      pvrtc4_keys_row<3>(keys, mp);
      uint32_t a, c;
      pvrtc4_keys_finish(keys, image0, lookup10, a, c);
      cc[1].a = channel_reduce(a, false);
      cc[1].b = channel_reduce(c, true);
      pvrtc_keys_reset(keys);
    }
    exchange(s, cc[1], cc[0], cc[2]);
    // colour rows (s-1, s): V = 16 A + w * 4 (B - A) for weight w = 0..3; from it the walks' bases and their steps per pixel row:
    //   x = 0, 1: D = 4 (V[1] - V[0]), P = 8 (V[0] + V[1]);   x = 2, 3: D = 4 (V[2] - V[1]), P = 16 V[1]
#if defined(ICAMD_PVRTC_WALK64)
    // (the bases are carried and stepped a fourth time at the end of the segment, the steps come straight from E = B - A: see
    // pvrtc_onepass_strip; here D = 64 (A1 - A0), P = 128 (A0 + A1) | D = 64 (A2 - A1), P = 256 A1)
    icamd_u64 dP0[2], dD0[2], dP1[2], dD1[2];
    {
      uint32_t ep0[4], ed0[4], ep1[4], ed1[4];
      ICAMD_UNROLL
      for (int v = 0; v < 4; ++v) {
        uint32_t e[3];
        ICAMD_UNROLL
        for (int c = 0; c < 3; ++c) {
          const uint32_t b = v == 0 ? pair_rb(cc[c].a) : v == 1 ? pair_ga(cc[c].a) : v == 2 ? pair_rb(cc[c].b) : pair_ga(cc[c].b);
          e[c] = b - A[c][v];
          A[c][v] = b;
        }
        ed0[v] = (e[1] - e[0]) << 4;
        ep0[v] = (e[0] + e[1]) << 5;
        ed1[v] = (e[2] - e[1]) << 4;
        ep1[v] = e[1] << 6;
      }
      ICAMD_UNROLL
      for (int p = 0; p < 2; ++p) {  // (signed steps: lanes of at most 16 320)
        dD0[p] = pack64_signed(ed0[2 * p], ed0[2 * p + 1]); dD1[p] = pack64_signed(ed1[2 * p], ed1[2 * p + 1]);
        dP0[p] = pack64_signed(ep0[2 * p], ep0[2 * p + 1]); dP1[p] = pack64_signed(ep1[2 * p], ep1[2 * p + 1]);
      }
    }
#define ICAMD_ROW4_BITS(px_) pvrtc4_row_bits64(P0, D0, P1, D1, px_)
#define ICAMD_ROW4_STEP()                                                                                    \
  ICAMD_UNROLL                                                                                               \
  for (int p = 0; p < 2; ++p) {                                                                              \
    P0[p] = add64(P0[p], dP0[p]); D0[p] = add64(D0[p], dD0[p]);                                              \
    P1[p] = add64(P1[p], dP1[p]); D1[p] = add64(D1[p], dD1[p]);                                              \
  }
#else
    uint32_t P0[4], D0[4], P1[4], D1[4], dP0[4], dD0[4], dP1[4], dD1[4];
    ICAMD_UNROLL
    for (int v = 0; v < 4; ++v) {
      uint32_t Vc[3], dVc[3];
      ICAMD_UNROLL
      for (int c = 0; c < 3; ++c) {
        const uint32_t b = v == 0 ? pair_rb(cc[c].a) : v == 1 ? pair_ga(cc[c].a) : v == 2 ? pair_rb(cc[c].b) : pair_ga(cc[c].b);
        Vc[c] = A[c][v] << 4;
        dVc[c] = (b - A[c][v]) << 2;
        A[c][v] = b;
      }
      D0[v] = (Vc[1] - Vc[0]) << 2;   dD0[v] = (dVc[1] - dVc[0]) << 2;
      P0[v] = (Vc[0] + Vc[1]) << 3;   dP0[v] = (dVc[0] + dVc[1]) << 3;
      D1[v] = (Vc[2] - Vc[1]) << 2;   dD1[v] = (dVc[2] - dVc[1]) << 2;
      P1[v] = Vc[1] << 4;             dP1[v] = dVc[1] << 4;
    }
#define ICAMD_ROW4_BITS(px_) pvrtc4_row_bits(P0, D0, P1, D1, px_)
#define ICAMD_ROW4_STEP()                                                                            \
  ICAMD_UNROLL                                                                                       \
  for (int v = 0; v < 4; ++v) { P0[v] += dP0[v]; D0[v] += dD0[v]; P1[v] += dP1[v]; D1[v] += dD1[v]; }
#endif
    if (s >= 1) data |= ICAMD_ROW4_BITS(ep) << 16;  // row 2 of block s-1, weight 0
    ICAMD_ROW4_STEP()
    tick(4 * s + 4, mp, ep);
    pvrtc4_keys_row<0>(keys, mp);
    if (s >= 1) {  // row 3 of block s-1, weight 1: the block is complete
      data |= ICAMD_ROW4_BITS(ep) << 24;
      store((uint32_t)(s - 1), data, own_acc);
    }
    if (s == K) break;
    ICAMD_ROW4_STEP()
    tick(4 * s + 5, mp, ep);
    pvrtc4_keys_row<1>(keys, mp);
    if (s >= 0) {  // row 0 of block s, weight 2
      own_acc = cc[1];
      data = ICAMD_ROW4_BITS(ep);
    }
    ICAMD_ROW4_STEP()
    tick(4 * s + 6, mp, ep);
    pvrtc4_keys_row<2>(keys, mp);
    if (s >= 0) data |= ICAMD_ROW4_BITS(ep) << 8;  // row 1 of block s, weight 3
#if defined(ICAMD_PVRTC_WALK64)
    ICAMD_ROW4_STEP()  // weight 4 = colour row s itself = the next segment's weight 0
#endif
#undef ICAMD_ROW4_STEP
#undef ICAMD_ROW4_BITS
  }
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
static inline uint32_t spread_bits16_host(uint32_t v) {
  v = (v | v << 8) & 0x00ff00ffu; v = (v | v << 4) & 0x0f0f0f0fu; v = (v | v << 2) & 0x33333333u; v = (v | v << 1) & 0x55555555u;
  return v;
}
// PVRTC 4 bpp (extension): the device math above over a whole image (tests/host_emul only).
static inline int emul_pvrtc4(const uint8_t *src, uint32_t n, uint8_t *out) {
  const uint32_t lw = n / 4;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(src);
  PvrtcColors *col = new PvrtcColors[(size_t)lw * lw];
  for (uint32_t by = 0; by < lw; ++by)
    for (uint32_t bx = 0; bx < lw; ++bx) {
      uint32_t px[16], a, b;
      for (int i = 0; i < 16; ++i) px[i] = img[(size_t)(by * 4 + i / 4) * n + bx * 4 + i % 4];
      BlockStash st;
      pvrtc4_extremes(px, img[0], st, a, b);
      col[by * lw + bx].a = channel_reduce(a, false);
      col[by * lw + bx].b = channel_reduce(b, true);
    }
  for (uint32_t by = 0; by < lw; ++by)
    for (uint32_t bx = 0; bx < lw; ++bx) {
      uint32_t px[16];
      for (int i = 0; i < 16; ++i) px[i] = img[(size_t)(by * 4 + i / 4) * n + bx * 4 + i % 4];
      PvrtcColors nb[3][3];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) nb[dy][dx] = col[((by + lw + dy - 1) % lw) * lw + (bx + lw + dx - 1) % lw];
      uint32_t *o = reinterpret_cast<uint32_t *>(out) + 2 * (size_t)(spread_bits16_host(bx) << 1 | spread_bits16_host(by));
      o[0] = pvrtc4_block_data(px, nb);
      o[1] = pvrtc_pack_colors(nb[1][1].a, nb[1][1].b, true);  // bit 0 clear: standard modulation
    }
  // the one-pass walker (what icamd_pvrtc4_onepass_kernel runs) must reproduce every block for several strip heights
  int ok = 1;
  for (uint32_t k_blocks = 1; k_blocks <= 8 && k_blocks <= lw; k_blocks *= 2)
    for (uint32_t by0 = 0; by0 < lw; by0 += k_blocks)
      for (uint32_t bx = 0; bx < lw; ++bx) {
        int last_m = -100;
        auto tick = [&](int m, uint32_t *mp, uint32_t *ep) {
          last_m = m;
          const uint32_t ym = (by0 * 4 + (uint32_t)m) & (n - 1), ye = (by0 * 4 + (uint32_t)(m - 5)) & (n - 1);
          for (int x = 0; x < 4; ++x) {
            mp[x] = img[(size_t)ym * n + bx * 4 + x];
            ep[x] = img[(size_t)ye * n + bx * 4 + x];
          }
        };
        auto lookup10 = [&](const uint32_t idx[10], uint32_t v[10]) {
          const uint32_t y0 = (by0 * 4 + (uint32_t)(last_m - 3)) & (n - 1);
          for (int i = 0; i < 10; ++i) v[i] = img[(size_t)(y0 + idx[i] / 4) * n + bx * 4 + idx[i] % 4];
        };
        auto exchange = [&](int s, const PvrtcColors &own, PvrtcColors &left, PvrtcColors &right) {
          const size_t row = (size_t)((by0 + lw + (uint32_t)s) % lw) * lw;
          if (own.a != col[row + bx].a || own.b != col[row + bx].b) ok = 0;
          left = col[row + (bx + lw - 1) % lw];
          right = col[row + (bx + 1) % lw];
        };
        uint32_t stored = 0;
        auto store = [&](uint32_t j, uint32_t data, const PvrtcColors &own) {
          const uint32_t *o = reinterpret_cast<const uint32_t *>(out) + 2 * (size_t)(spread_bits16_host(bx) << 1 | spread_bits16_host(by0 + j));
          if (o[0] != data || o[1] != pvrtc_pack_colors(own.a, own.b, true)) ok = 0;
          stored |= 1u << j;
        };
        pvrtc4_onepass_strip(k_blocks, img[0], tick, lookup10, exchange, store);
        if (!ok || stored != (1u << k_blocks) - 1u) { delete[] col; return 0; }
      }
  delete[] col;
  return 1;
}
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
  uint32_t *self_right = new uint32_t[(size_t)bw * bh], *self_below = new uint32_t[(size_t)2 * bw * bh];
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
      uint8_t m[32], m2[32];
      emul_mods_xy<0, 0>(px, nb, m);  // generic per-pixel path
      uint32_t rows[4][2], right_px[4], below_px[8], right_col, below[2];
      for (int y = 0; y < 4; ++y) right_px[y] = img[(size_t)(by * 4 + y) * n + ((bx * 8 + 8) & (n - 1))];
      for (int x = 0; x < 8; ++x) below_px[x] = img[(size_t)((by * 4 + 4) & (n - 1)) * n + bx * 8 + x];
      PvrtcColors nbc[3][3];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx) {
          const size_t o = ((by + bh + dy - 1) % bh) * bw + (bx + bw + dx - 1) % bw;
          nbc[dy][dx].a = ca[o];
          nbc[dy][dx].b = cb[o];
        }
      pvrtc_block_mods(px, right_px, below_px, nbc, rows, &right_col, below);  // separable path (what the kernel runs)
      for (int i = 0; i < 32; ++i) m2[i] = (uint8_t)(rows[i / 8][(i % 8) >> 2] >> (8 * (i & 3)));
      if (memcmp(m, m2, 32) != 0) return 0;
      for (int i = 0; i < 32; ++i) mods[(size_t)(by * 4 + i / 8) * n + bx * 8 + i % 8] = m[i];
      self_right[by * bw + bx] = right_col;
      self_below[2 * (by * bw + bx)] = below[0];
      self_below[2 * (by * bw + bx) + 1] = below[1];
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
      // pvrtc_left_edge_mod (what the encode kernel precomputes for the column right of each wave) == the per-pixel path
      for (uint32_t y = 0; y < 4; ++y) {
        const uint32_t up = (y < 2 ? by + bh - 1 : by) % bh, dn = (up + 1) % bh, lx = (bx + bw - 1) % bw;
        const PvrtcColors ul = { ca[up * bw + lx], cb[up * bw + lx] }, uc = { ca[up * bw + bx], cb[up * bw + bx] };
        const PvrtcColors ll = { ca[dn * bw + lx], cb[dn * bw + lx] }, lc = { ca[dn * bw + bx], cb[dn * bw + bx] };
        if (pvrtc_left_edge_mod(img[(size_t)(by * 4 + y) * n + bx * 8], y, ul, uc, ll, lc) !=
            mods[(size_t)(by * 4 + y) * n + bx * 8]) return 0;
      }
      // the halo values each lane recomputes for itself must equal the neighbours' own values
      if (right != self_right[by * bw + bx] || below[0] != self_below[2 * (by * bw + bx)] ||
          below[1] != self_below[2 * (by * bw + bx) + 1]) return 0;
      bool one_bpp;
      const uint32_t data = pvrtc_block_modulation(rows, right, below, &one_bpp);
      const uint32_t colors = pvrtc_pack_colors(ca[by * bw + bx], cb[by * bw + bx], one_bpp);
      {  // the row-streaming encoder (what the kernel runs) must agree
        PvrtcColors nbc[3][3];
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {
            const size_t o2 = ((by + bh + dy - 1) % bh) * bw + (bx + bw + dx - 1) % bw;
            nbc[dy][dx].a = ca[o2];
            nbc[dy][dx].b = cb[o2];
          }
        auto loader = [&](int r, uint32_t *pixels, uint32_t *right_px) {
          const uint32_t yy = (by * 4 + (uint32_t)r) & (n - 1);
          for (int x = 0; x < 8; ++x) pixels[x] = img[(size_t)yy * n + bx * 8 + x];
          *right_px = img[(size_t)yy * n + ((bx * 8 + 8) & (n - 1))];
        };
        uint32_t data2;
        bool one2;
        pvrtc_encode_block_rows(loader, nbc, &data2, &one2);
        if (data2 != data || one2 != one_bpp) return 0;
      }
      uint32_t *o = reinterpret_cast<uint32_t *>(out) + 2 * (size_t)pvrtc_z_index(bx, by);
      o[0] = data;
      o[1] = colors;
    }
  // the strip encoder (what the kernel runs) must reproduce every block for several strip heights
  for (uint32_t k_blocks = 1; k_blocks <= 8 && k_blocks <= bh; k_blocks *= 2)
    for (uint32_t by0 = 0; by0 < bh; by0 += k_blocks)
      for (uint32_t bx = 0; bx < bw; ++bx) {
        auto load_px = [&](uint32_t r, uint32_t *pixels, uint32_t *right_px) {
          const uint32_t yy = (by0 * 4 + r) & (n - 1);
          for (int x = 0; x < 8; ++x) pixels[x] = img[(size_t)yy * n + bx * 8 + x];
          *right_px = img[(size_t)yy * n + ((bx * 8 + 8) & (n - 1))];
        };
        auto load_colours = [&](int j, PvrtcColors c[3]) {
          const uint32_t yy = (by0 + bh + (uint32_t)j) % bh;
          for (int dx = 0; dx < 3; ++dx) {
            const size_t o2 = (size_t)yy * bw + (bx + bw + dx - 1) % bw;
            c[dx].a = ca[o2];
            c[dx].b = cb[o2];
          }
        };
        int ok = 1;
        auto store = [&](uint32_t j, uint32_t data, bool one_bpp, const PvrtcColors &own) {
          const uint32_t *o = reinterpret_cast<const uint32_t *>(out) + 2 * (size_t)pvrtc_z_index(bx, by0 + j);
          if (o[0] != data || o[1] != pvrtc_pack_colors(own.a, own.b, one_bpp)) ok = 0;
        };
        auto no_right = [&](uint32_t, uint32_t) -> uint32_t { return 0u; };
        pvrtc_encode_strip<false>(k_blocks, load_px, load_colours, store, no_right);
        if (!ok) return 0;
        // EXCHANGE form: the right-hand values come from the neighbouring column (here: the per-pixel reference path)
        auto right_of = [&](uint32_t j, uint32_t col0) -> uint32_t {
          uint32_t own0 = 0, r = 0;
          for (int y = 0; y < 4; ++y) {
            own0 |= (uint32_t)mods[(size_t)((by0 + j) * 4 + y) * n + bx * 8] << (8 * y);
            r |= (uint32_t)mods[(size_t)((by0 + j) * 4 + y) * n + ((bx * 8 + 8) & (n - 1))] << (8 * y);
          }
          if (own0 != col0) ok = 0;  // what the lane hands to its left-hand neighbour
          return r;
        };
        pvrtc_encode_strip<true>(k_blocks, load_px, load_colours, store, right_of);
        if (!ok) return 0;
        // the one-pass walker (r05: what icamd_pvrtc2_onepass_kernel runs): morphs its own column, is handed the neighbour
        // columns' colours and column-0 values (here: the reference values; what it hands out is checked against them)
        int last_m = -100;
        auto tick = [&](int m, uint32_t *mp, uint32_t *ep) {
          last_m = m;
          const uint32_t ym = (by0 * 4 + (uint32_t)m) & (n - 1), ye = (by0 * 4 + (uint32_t)(m - 5)) & (n - 1);
          for (int x = 0; x < 8; ++x) {
            mp[x] = img[(size_t)ym * n + bx * 8 + x];
            ep[x] = img[(size_t)ye * n + bx * 8 + x];
          }
        };
        auto lookup10 = [&](const uint32_t idx[10], uint32_t v[10]) {
          const uint32_t y0 = (by0 * 4 + (uint32_t)(last_m - 3)) & (n - 1);  // first row of the block just consumed
          for (int i = 0; i < 10; ++i) v[i] = img[(size_t)(y0 + idx[i] / 8) * n + bx * 8 + idx[i] % 8];
        };
        auto exchange = [&](int s, const PvrtcColors &own, uint32_t col0, PvrtcColors &left, PvrtcColors &right, uint32_t &right_col0) {
          if (s <= (int)k_blocks) {
            const uint32_t yy = (by0 + bh + (uint32_t)s) % bh;
            const size_t o2 = (size_t)yy * bw;
            if (own.a != ca[o2 + bx] || own.b != cb[o2 + bx]) ok = 0;
            left.a = ca[o2 + (bx + bw - 1) % bw]; left.b = cb[o2 + (bx + bw - 1) % bw];
            right.a = ca[o2 + (bx + 1) % bw]; right.b = cb[o2 + (bx + 1) % bw];
          }
          if (s >= 2) {
            uint32_t own0 = 0, r = 0;
            for (int y = 0; y < 4; ++y) {
              own0 |= (uint32_t)mods[(size_t)((by0 + (uint32_t)s - 2) * 4 + y) * n + bx * 8] << (8 * y);
              r |= (uint32_t)mods[(size_t)((by0 + (uint32_t)s - 2) * 4 + y) * n + ((bx * 8 + 8) & (n - 1))] << (8 * y);
            }
            if (own0 != col0) ok = 0;
            right_col0 = r;
          }
        };
        uint32_t stored = 0;
        auto store1 = [&](uint32_t j, uint32_t data, bool one_bpp, const PvrtcColors &own) {
          store(j, data, one_bpp, own);
          stored |= 1u << j;
        };
        pvrtc_onepass_strip(k_blocks, img[0], tick, lookup10, exchange, store1);
        if (!ok || stored != (k_blocks >= 32 ? 0xffffffffu : (1u << k_blocks) - 1u)) return 0;
      }
  delete[] ab; delete[] ca; delete[] cb; delete[] mods; delete[] self_right; delete[] self_below;
  return 1;
}
#endif

}  // namespace icamd
#endif  // ICAMD_PVRTC_BLOCK_H_
