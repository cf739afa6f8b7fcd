// etc1_block.h -- ETC1 block encoder (all four CompressionStrategy values), one block per lane.
//
// Bit-exact with EncodeEtc1Block (internal/etc_compressor.cc:545-586) and everything under it,
// restructured for CDNA4 integer VALU.  The reference's hot loop (etc.cc:350-409) evaluates, per
// sub-block pixel p and per candidate colour v = clamp(base + modifier), the squared distance
//     |p - v|^2 = |p|^2 - (2 p.v - |v|^2)
// and keeps the smallest (ties: lowest modifier index, then lowest codeword).  |p|^2 does not depend
// on the candidate, so we maximise E = 2 p.v - |v|^2 instead:
//   * p.v is ONE v_dot4_u32_u8 on the packed pixel / candidate dwords (clamping is already baked
//     into v, so this is exact for every base colour, unlike a Sum(d)-based shortcut);
//   * key = 32*E + (3-k) folds the tie rule into a signed max (v_lshl_add_u32 + v_max3_i32); the key sum >> 5 is Sum(E);
//   * the 2-bit winner is shifted into an index word with v_alignbit_b32;
//   * the 32 candidates of a sub-block are built with packed saturating 16-bit adds
//     (v_pk_add_u16 / v_pk_sub_u16 clamp on 0xRR00BB00) + one v_perm_b32 each;
//   * Sum|p|^2 over the 16 pixels is the same for both flips, so "error_lr <= error_tb"
//     (etc.cc:583) is decided on Sum(E) alone.
// 3.5-4.3 k integer instructions per block at kSmallerError: this codec is VALU-bound, not HBM-bound.
#ifndef ICAMD_ETC1_BLOCK_H_
#define ICAMD_ETC1_BLOCK_H_

#include "dxt_block.h"  // Out8
#include "ic_device.h"

namespace icamd {

// Diagnostics build only (-DICAMD_ETC1_STATS, never in libic_amd.so as shipped): per-wave counts of which evaluation each
// codeword of each search took, read back with icamd_debug_etc1_stats (scripts/etc1_path_stats.py).
#if defined(ICAMD_ETC1_STATS) && !defined(ICAMD_HOST_EMULATION)
__device__ unsigned int g_etc1_stats[16];
#define ICAMD_ETC1_COUNT(i) do { if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_etc1_stats[i], 1u); } while (0)
#else
#define ICAMD_ETC1_COUNT(i) ((void)0)
#endif

#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV uint32_t pk_addsat_u16(uint32_t a, uint32_t b) {
  uint32_t lo = (a & 0xffffu) + (b & 0xffffu), hi = (a >> 16) + (b >> 16);
  if (lo > 0xffffu) lo = 0xffffu;
  if (hi > 0xffffu) hi = 0xffffu;
  return hi << 16 | lo;
}
ICAMD_DEV uint32_t pk_subsat_u16(uint32_t a, uint32_t b) {
  uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return (ah > bh ? ah - bh : 0u) << 16 | (al > bl ? al - bl : 0u);
}
ICAMD_DEV uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {
  return (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16) + c;
}
#else
typedef unsigned short icamd_us2 __attribute__((ext_vector_type(2)));
// v_pk_add_u16 ... clamp / v_pk_sub_u16 ... clamp
ICAMD_DEV uint32_t pk_addsat_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(icamd_us2, a),
                                                                     __builtin_bit_cast(icamd_us2, b)));
}
ICAMD_DEV uint32_t pk_subsat_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(icamd_us2, a),
                                                                     __builtin_bit_cast(icamd_us2, b)));
}
// v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi + c
ICAMD_DEV uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(icamd_us2, a), __builtin_bit_cast(icamd_us2, b), c, false);
}
#endif

// ETC1 modifier table (OES_compressed_ETC1_RGB8_texture; etc.cc:101-110): row cw = {a, b, -a, -b}.
// Packed as bytes so a per-lane codeword can look its row up with two v_bfe.
constexpr uint32_t kEtcModA_lo = 2u | 5u << 8 | 9u << 16 | 13u << 24;
constexpr uint32_t kEtcModA_hi = 18u | 24u << 8 | 33u << 16 | 47u << 24;
constexpr uint32_t kEtcModB_lo = 8u | 17u << 8 | 29u << 16 | 42u << 24;
constexpr uint32_t kEtcModB_hi = 60u | 80u << 8 | 106u << 16 | 183u << 24;
constexpr int kEtcA[8] = { 2, 5, 9, 13, 18, 24, 33, 47 };
constexpr int kEtcB[8] = { 8, 17, 29, 42, 60, 80, 106, 183 };

// A sub-block's base colour in the two forms the candidate builder needs.
struct EtcBase {
  uint32_t rb_hi;  // 0xRR00BB00
  uint32_t g_hi;   // 0x0000GG00
};

// Pixel j (0..7) of sub-block S under FLIP, enumerated in ascending ETC bit position 4x+y
// (etc.cc:131-137) so that the index field order equals the output bit order.
//   FLIP=0 (left|right halves, etc.cc:466-467): x in {2S, 2S+1}, y = 0..3  -> bit 8S + j
//   FLIP=1 (top|bottom halves, etc.cc:464-465): y in {2S, 2S+1}, x = 0..3 -> bit 4(j>>1) + 2S + (j&1)
//   FLIP=2: the caller already gathered the 16 pixels in (sub-block, j) order for a per-lane flip
template <int FLIP, int S>
constexpr int sub_pixel(int j) {
  return FLIP == 2 ? 8 * S + j
       : FLIP == 0 ? 4 * (j & 3) + (2 * S + (j >> 2))       // raster index 4y + x
                   : 4 * (2 * S + (j & 1)) + (j >> 1);
}

// The 4 candidate colours of one codeword (modifiers +a, +b, -a, -b; etc.cc:121-125 clamps each
// channel to 0..255) as packed R,G,B,0 dwords, and the per-candidate constant (3-k) - 32|v|^2.
ICAMD_DEV void build_candidates(const EtcBase &base, uint32_t a, uint32_t b, uint32_t v[4], int32_t c[4]) {
  const uint32_t a1 = a << 8, b1 = b << 8;
  const uint32_t a2 = a1 | a1 << 16, b2 = b1 | b1 << 16;  // modifier in the high byte of both halves (no multiply: a
                                                          // run-time modifier -- kHeuristic -- would make it a v_mul_lo_u32)
  // byte0 = R (rb byte 3), byte1 = G (g byte 1), byte2 = B (rb byte 1), byte3 = 0
  const uint32_t sel = 0x0c050107u;
  v[0] = perm(pk_addsat_u16(base.rb_hi, a2), pk_addsat_u16(base.g_hi, a1), sel);
  v[1] = perm(pk_addsat_u16(base.rb_hi, b2), pk_addsat_u16(base.g_hi, b1), sel);
  v[2] = perm(pk_subsat_u16(base.rb_hi, a2), pk_subsat_u16(base.g_hi, a1), sel);
  v[3] = perm(pk_subsat_u16(base.rb_hi, b2), pk_subsat_u16(base.g_hi, b1), sel);
  ICAMD_UNROLL
  // opaque(): the constant must exist as one (negative) register so that every key is a single v_lshl_add_u32;
  // left to itself the optimiser keeps +4|v|^2 for k = 3 and spends a shift and a subtract per pixel on it
  for (int k = 0; k < 4; ++k) c[k] = (int32_t)opaque((uint32_t)((3 - k) - 32 * (int32_t)udot4(v[k], v[k], 0u)));
}

// ComputeCodewordError (etc.cc:350-385) for one codeword over the 8 pixels of a sub-block.
// Per pixel and candidate the key is 32 E + (3 - k) with E = 2 p.v - |v|^2: the signed max picks the best E and, on
// ties, the lowest k (etc.cc:366-379).  The eight tie-break fields add up to at most 24 < 32, so an arithmetic shift
// of the key sum by 5 returns exactly Sum_p max_k E.  *fields receives the eight 2-bit values (3 - best_k) in bits
// 16..31, field j = pixel sub_pixel<FLIP,S>(j).
template <int FLIP, int S>
ICAMD_DEV int32_t eval_codeword(const uint32_t px[16], const uint32_t v[4], const int32_t c[4], uint32_t *fields) {
  int32_t sum = 0;
  uint32_t acc = 0;
  ICAMD_UNROLL
  for (int j = 0; j < 8; ++j) {
    const uint32_t p = px[sub_pixel<FLIP, S>(j)];
    const int32_t k0 = (int32_t)(udot4(p, v[0], 0u) << 6) + c[0];
    const int32_t k1 = (int32_t)(udot4(p, v[1], 0u) << 6) + c[1];
    const int32_t k2 = (int32_t)(udot4(p, v[2], 0u) << 6) + c[2];
    const int32_t k3 = (int32_t)(udot4(p, v[3], 0u) << 6) + c[3];
    const int32_t m = imax(imax3(k0, k1, k2), k3);
    sum += m;
    acc = alignbit((uint32_t)m, acc, 2);
  }
  *fields = acc;
  return sum >> 5;
}

struct EtcSubResult {
  int32_t score;    // Sum_p max_k E for the chosen codeword (larger = smaller error)
  uint32_t cw;      // chosen codeword 0..7
  uint32_t fields;  // see eval_codeword
};

// Unclamped shortcut.  When base_c +/- b stays inside [0,255] for all three channels, every candidate is
// v = base + m*(1,1,1) and  E(m) = 2 p.v - |v|^2 = E0 + 2 m s - 3 m^2  with  s = Sum(p) - Sum(base),
// E0 = 2 p.base - |base|^2.  Then (all comparisons exact integers, same tie rules as etc.cc:366-379):
//   * the sign of s picks the side: s >= 0 -> {+a, +b} (indices 0,1), s < 0 -> {-a, -b} (indices 2,3);
//     f(+m) - f(-m) = 4 m s, and s = 0 ties go to the lower index, i.e. the positive side;
//   * within the side, +/-a wins unless 2b|s| - 3b^2 > 2a|s| - 3a^2, i.e. a wins iff 2|s| <= 3 (a + b)  (tie -> a).
// So a pixel contributes  f = max(2a|s| - 3a^2, 2b|s| - 3b^2) = (2a|s| - 3a^2) + (b - a) relu(2|s| - 3(a + b)),  and with
// relu(y) = (y + |y|) / 2 the sub-block's score collapses to sums (r03):
//     Sum_j f_j = ( (a + b) S2 + (b - a) Sum_j |2|s_j| - 3(a + b)| ) / 2  -  12 (a^2 + b^2),     S2 = Sum_j 2|s_j|
// (the numerator is even: S2 is, and either b - a is or all eight terms of the second sum are odd).  Per pixel and
// codeword that is ONE v_sad_u16 with accumulate; which modifier each pixel took is worked out afterwards, for the
// winning codeword only (search_codewords).  abs2[j] = 2|s_j| of pixel j of the sub-block.  Returns Sum_j f_j (WITHOUT
// the E0 part).
ICAMD_DEV int32_t eval_codeword_unclamped(const uint32_t abs2[8], uint32_t s2, int32_t a, int32_t b) {
  uint32_t dev = 0;
  ICAMD_UNROLL
  for (int j = 0; j < 8; ++j) dev = sad_u32(abs2[j], (uint32_t)(3 * (a + b)), dev);
  return (imad24(a + b, (int32_t)s2, imad24(b - a, (int32_t)dev, 0)) >> 1) - 12 * (a * a + b * b);
}

// Mixed tier: the codeword's +/-a candidates are unclamped in every lane of the wave (base_c +/- a inside [0,255]) while
// +/-b may clamp somewhere.  +/-a then follow the formula -- the better of the two is on the side of sign(s), with
// E = E0 + 2a|s| - 3a^2 -- and only +/-b are built and evaluated exactly: per pixel one v_mad + 2 (v_dot4 + v_lshl_add)
// + v_max3 instead of 4 (v_dot4 + v_lshl_add) + v_max3 + v_max.  k0[j] = 32 E0_j + (3 - k) of the a candidate on the
// pixel's side (k = 0 for s >= 0, 2 for s < 0); all keys of this evaluation carry a common offset of +96 a^2, taken off
// the sum at the end.  Same key format and tie rules as eval_codeword.
template <int FLIP, int S>
ICAMD_DEV int32_t eval_codeword_mixed(const uint32_t px[16], const uint32_t abs2[8], const int32_t k0[8], const EtcBase &base,
                                      int32_t a, int32_t b, uint32_t *fields) {
  const uint32_t b2 = (uint32_t)b * 0x01000100u, b1 = (uint32_t)b << 8, sel = 0x0c050107u;
  const uint32_t vp = perm(pk_addsat_u16(base.rb_hi, b2), pk_addsat_u16(base.g_hi, b1), sel);
  const uint32_t vn = perm(pk_subsat_u16(base.rb_hi, b2), pk_subsat_u16(base.g_hi, b1), sel);
  const int32_t cp = (int32_t)opaque((uint32_t)(2 + 96 * a * a - 32 * (int32_t)udot4(vp, vp, 0u)));  // k = 1
  const int32_t cn = (int32_t)opaque((uint32_t)(0 + 96 * a * a - 32 * (int32_t)udot4(vn, vn, 0u)));  // k = 3
  int32_t sum = 0;
  uint32_t acc = 0;
  ICAMD_UNROLL
  for (int j = 0; j < 8; ++j) {
    const uint32_t p = px[sub_pixel<FLIP, S>(j)];
    const int32_t ka = imad24((int32_t)abs2[j], 32 * a, k0[j]);
    const int32_t kp = (int32_t)(udot4(p, vp, 0u) << 6) + cp;
    const int32_t kn = (int32_t)(udot4(p, vn, 0u) << 6) + cn;
    const int32_t m = imax3(ka, kp, kn);
    sum += m;
    acc = alignbit((uint32_t)m, acc, 2);
  }
  *fields = acc;
  return (sum >> 5) - 24 * a * a;
}

// FindBestCodeword (etc.cc:391-409): first codeword with the strictly smallest error.
// bmin / bmax: smallest / largest channel of the decoded base colour (decides, per codeword and for the whole
// wave at once, whether the unclamped shortcut applies); sub_sum[]: channel sums of the sub-block's 8 pixels;
// psum[]: 2 (r + g + b) of each of the 16 pixels.
// skip: this lane's result will not be used (a one-colour block inside a mixed wave, encoded by
// encode_etc1_constant_block instead) -- it takes no part in the wave-uniform decisions below, so a lane that would
// veto the shortcut, the tier or a pruning step for everyone (a saturated flat colour, typically) no longer does.
// (SKIP is a template parameter so that waves without such lanes run exactly the code they ran before: r03 A/B, a run-time
// flag alone cost noise content 5 %.)
// ICAMD_ETC1_NO_PSUM (A/B only): recompute 2 (r + g + b) of a pixel where it is used instead of keeping 16 of them live
#if defined(ICAMD_ETC1_NO_PSUM)
#define ICAMD_PSUM(q) udot4(px[q], 0x00020202u, 0u)
#else
#define ICAMD_PSUM(q) psum[q]
#endif
template <int FLIP, int S, bool TIER, bool PRUNE, bool SKIP = false>
ICAMD_DEV EtcSubResult search_codewords(const uint32_t px[16], const uint32_t psum[16], const EtcBase &base,
                                        const uint32_t bch[3], const uint32_t sub_sum[3], bool skip_lane = false) {
  const bool skip = SKIP && skip_lane;
  const uint32_t bsum = bch[0] + bch[1] + bch[2];
  const uint32_t base_px = bch[0] | bch[1] << 8 | bch[2] << 16;  // the base colour as a pixel dword
  const uint32_t bmin = umin3(bch[0], bch[1], bch[2]), bmax = umax3(bch[0], bch[1], bch[2]);
  const uint32_t room = umin(bmin, 255u - bmax);  // a modifier m leaves every channel of base +/- m unclamped iff m <= room
  // The modifiers grow with the codeword, so once a codeword clamps somewhere in the wave every later one does too:
  // `fast` is a wave-uniform flag that only ever goes from true to false, and when even codeword 0 clamps (bright /
  // dark / saturated regions) none of the shortcut's per-pixel preparation is executed.
  bool fast = wave_all(skip || room >= (uint32_t)kEtcB[0]);
  // per pixel 2|s| (psum[] holds 2 (r + g + b)) and their sum, shared by all unclamped codewords
  const uint32_t bsum2 = 2u * bsum;
  uint32_t abs2[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, s2 = 0;
  int32_t k0[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  int32_t e0_sum = 0;
  bool a_fits = false;       // mixed tier usable (set below); then wave-uniform and monotone like `fast`
  if (fast) {
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) {
      abs2[j] = sad_u32(ICAMD_PSUM((sub_pixel<FLIP, S>(j))), bsum2, 0u);
      s2 += abs2[j];
    }
    // Sum_j E0 = 2 * (base . sub_sum) - 8 |base|^2 : puts the shortcut's scores on the scale of eval_codeword
    e0_sum = 2 * (int32_t)umad24(bch[0], sub_sum[0], umad24(bch[1], sub_sum[1], umad24(bch[2], sub_sum[2], 0u))) -
           8 * (int32_t)udot4(base_px, base_px, 0u);
  }
  // Wave-uniform exact pruning of codewords that cannot win.  In channel c every candidate on the positive side of
  // codeword cw (+a, +b) differs from the base colour by at least min(a_cw, 255 - base_c), every one on the negative
  // side by at least min(a_cw, base_c) (clamping, etc.cc:121-125, can only shorten a step down to the distance to the
  // boundary), while every pixel of the sub-block lies within dev_c of the base colour.  A pixel's error under cw is
  // therefore at least min over the two sides of Sum_c (step_c - dev_c)^2 (terms clipped at 0), the sub-block's 8 times
  // that.  A codeword whose bound exceeds the error of the best one so far in every lane is skipped; the reference
  // keeps the FIRST codeword with the strictly smallest error (etc.cc:401), so nothing it would pick is lost.  Two
  // stages keep the cost off busy content: the L1 deviation (one v_sad_u8 per pixel) decides for the whole wave
  // whether the per-channel deviations are worth computing at all.
  // The instantiation with the mixed tier is only entered by waves of busy blocks (etc1_busy_wave), where this test
  // practically never passes: it leaves pruning out altogether (a performance choice; pruning never changes a result).
  bool prunable = false;
  if (PRUNE) {
    uint32_t d1 = 0;
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) d1 = umax(d1, sad_u8(px[sub_pixel<FLIP, S>(j)] & 0x00ffffffu, base_px, 0u));
    // a per-channel deviation is at least a third of the L1 one, and no step exceeds a_7 = 47
    prunable = wave_all(skip || d1 < 3u * (uint32_t)kEtcA[7]);
  }
  // Mixed tier (eval_codeword_mixed) for the codewords the shortcut does not reach: prepared only on busy content --
  // where pruning is on, most of those codewords are skipped anyway.
  if (TIER && fast && !wave_all(skip || room >= (uint32_t)kEtcB[7])) {
    a_fits = true;
    // 32 E0_j + tie field of the a candidate on the pixel's side (3 for s >= 0, 1 for s < 0)
    const int32_t c3 = 3 - 32 * (int32_t)udot4(base_px, base_px, 0u);
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) {
      const uint32_t q = sub_pixel<FLIP, S>(j);
      const int32_t neg = (int32_t)(ICAMD_PSUM(q) - bsum2) >> 31;  // -1 iff s < 0
      k0[j] = (int32_t)(udot4(px[q], base_px, 0u) << 6) + c3 + 2 * neg;
    }
  }
  uint32_t dev_rb = 0, dev_g = 0;      // per-channel maximum deviation: R | B << 16, G
  uint32_t room_rb_up = 0, room_rb_dn = 0, room_g_up = 0, room_g_dn = 0;
  int32_t sum_sq = 0;                  // Sum_j |p_j|^2: error of a codeword = sum_sq - its score
  if (prunable) {
    uint32_t rb_max = 0, rb_min = 0xffffffffu, g_max = 0, g_min = 0xffu;
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) {
      const uint32_t q = px[sub_pixel<FLIP, S>(j)] & 0x00ffffffu;
      sum_sq = (int32_t)udot4(q, q, (uint32_t)sum_sq);
      const uint32_t rb = q & 0x00ff00ffu, g = q >> 8 & 0xffu;
      rb_max = pk_max_u16(rb_max, rb);
      rb_min = pk_min_u16(rb_min, rb);
      g_max = umax(g_max, g);
      g_min = umin(g_min, g);
    }
    const uint32_t base_rb = bch[0] | bch[2] << 16;
    dev_rb = pk_max_u16(pk_subsat_u16(rb_max, base_rb), pk_subsat_u16(base_rb, rb_min));
    dev_g = umax(g_max - umin(g_max, bch[1]), bch[1] - umin(bch[1], g_min));
    room_rb_dn = base_rb;
    room_rb_up = 0x00ff00ffu - base_rb;
    room_g_dn = bch[1];
    room_g_up = 255u - bch[1];
  }
  // mid-tones everywhere in the wave: no step of any codeword is shortened, the bound is 24 (a_cw - max_c dev_c)^2
  const bool roomy = prunable && wave_all(skip || room >= (uint32_t)kEtcA[7]);
  const uint32_t dev_max = umax3(dev_rb & 0xffffu, dev_rb >> 16, dev_g);
  ICAMD_ETC1_COUNT(4);
  if (fast) ICAMD_ETC1_COUNT(5);
  if (prunable) ICAMD_ETC1_COUNT(6);
  if (TIER) ICAMD_ETC1_COUNT(7);
  EtcSubResult r;
  r.score = 0; r.cw = 0; r.fields = 0;
  uint32_t fast_mask = 0;  // wave-uniform: bit cw set iff that codeword took the shortcut
  ICAMD_UNROLL
  for (int cw = 0; cw < 8; ++cw) {
    int32_t s;
    uint32_t f;
    // (r03) the shortcut costs a dozen instructions per codeword -- less than a pruning test -- so only codewords that
    // would take an exact evaluation are tested
    if (cw > 0 && fast) fast = wave_all(skip || room >= (uint32_t)kEtcB[cw]);
    if (fast) {
      // nothing to prune
    } else if (cw > 0 && roomy) {
      const int32_t t = kEtcA[cw] - (int32_t)dev_max;
      // (opaque: left alone the optimiser regroups 24 t^2 into (24 t) * t with a quarter-rate v_mul_lo_u32)
      const int32_t tt = (int32_t)opaque((uint32_t)imad24(t, t, 0));
      if (wave_all(skip || (t > 0 && imad24(tt, 24, 0) > sum_sq - r.score))) { ICAMD_ETC1_COUNT(3); continue; }
    } else if (cw > 0 && prunable) {
      const uint32_t a2 = (uint32_t)kEtcA[cw] * 0x00010001u;
      const uint32_t up_rb = pk_subsat_u16(pk_min_u16(a2, room_rb_up), dev_rb);
      const uint32_t dn_rb = pk_subsat_u16(pk_min_u16(a2, room_rb_dn), dev_rb);
      const uint32_t ug = umin((uint32_t)kEtcA[cw], room_g_up), dg = umin((uint32_t)kEtcA[cw], room_g_dn);
      const uint32_t up_g = ug - umin(ug, dev_g), dn_g = dg - umin(dg, dev_g);
      const uint32_t lb_up = udot2_u16(up_rb, up_rb, umad24(up_g, up_g, 0u)), lb_dn = udot2_u16(dn_rb, dn_rb, umad24(dn_g, dn_g, 0u));
      if (wave_all(skip || (int32_t)(8u * umin(lb_up, lb_dn)) > sum_sq - r.score)) { ICAMD_ETC1_COUNT(3); continue; }
    }
    if (fast) {
      ICAMD_ETC1_COUNT(0);
      s = eval_codeword_unclamped(abs2, s2, kEtcA[cw], kEtcB[cw]) + e0_sum;
      f = 0u;  // worked out below if this codeword wins
      fast_mask |= 1u << cw;
    } else {
      if (TIER && a_fits) a_fits = wave_all(skip || room >= (uint32_t)kEtcA[cw]);
      if (TIER && a_fits) {
        ICAMD_ETC1_COUNT(1);
        s = eval_codeword_mixed<FLIP, S>(px, abs2, k0, base, kEtcA[cw], kEtcB[cw], &f);
      } else {
        uint32_t v[4];
        int32_t c[4];
        ICAMD_ETC1_COUNT(2);
        ICAMD_ETC1_COUNT(8 + cw);  // which codewords end up in the exact evaluation
        build_candidates(base, (uint32_t)kEtcA[cw], (uint32_t)kEtcB[cw], v, c);
        s = eval_codeword<FLIP, S>(px, v, c, &f);
      }
    }
    const bool better = cw == 0 || s > r.score;
    r.score = better ? s : r.score;
    r.cw = better ? (uint32_t)cw : r.cw;
    // (shortcut codewords come before all others -- `fast` is monotone -- and their fields are worked out below)
    if (!fast) r.fields = better ? f : r.fields;
  }
  // The winner's index fields in the common format (3 - k per pixel, 2 bits, bits 16..31) when it took the shortcut:
  // high bit = (s >= 0), low bit = (magnitude a chosen) = (2|s| <= 3 (a + b)).  Both are sign bits of one subtraction,
  // shifted in pixel 7 first; skipped when no lane of the wave was won by a shortcut codeword.
  const bool won_fast = ((fast_mask >> r.cw) & 1u) != 0u;
  if (!wave_all(skip || !won_fast)) {
    const uint32_t sh = (r.cw & 3u) * 8u;
    const uint32_t thr = 3u * (bfe(r.cw < 4u ? kEtcModA_lo : kEtcModA_hi, sh, 8) + bfe(r.cw < 4u ? kEtcModB_lo : kEtcModB_hi, sh, 8));
    uint32_t acc = 0;
    ICAMD_UNROLL
    for (int j = 7; j >= 0; --j) {
      acc = alignbit(acc, ICAMD_PSUM((sub_pixel<FLIP, S>(j))) - bsum2, 31);  // s < 0
      acc = alignbit(acc, thr - abs2[j], 31);                          // 2|s| > 3 (a + b): magnitude b
    }
    r.fields = won_fast ? ~acc << 16 : r.fields;
  }
  return r;
}

// FindCodewordHeuristic (etc.cc:415-455): codeword from the largest mean absolute deviation.
// packed (optional): the sub-block's eight R bytes, G bytes and B bytes as two dwords each ([channel][half], any pixel
// order) -- the deviation sums are then two v_sad_u8 per channel instead of a v_bfe + v_sad per pixel and channel.
template <int FLIP, int S>
ICAMD_DEV EtcSubResult heuristic_codeword(const uint32_t px[16], const EtcBase &base, uint32_t br, uint32_t bg,
                                          uint32_t bb, const uint32_t (*packed)[2] = nullptr) {
  uint32_t sr = 0, sg = 0, sb = 0;
  if (packed) {
    const uint32_t br4 = perm(0u, br, 0u), bg4 = perm(0u, bg, 0u), bb4 = perm(0u, bb, 0u);  // byte 0 replicated (a plain multiply is a quarter-rate v_mul_lo_u32)
    sr = sad_u8(packed[0][0], br4, sad_u8(packed[0][1], br4, 0u));
    sg = sad_u8(packed[1][0], bg4, sad_u8(packed[1][1], bg4, 0u));
    sb = sad_u8(packed[2][0], bb4, sad_u8(packed[2][1], bb4, 0u));
  } else {
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) {
      const uint32_t p = px[sub_pixel<FLIP, S>(j)];
      sr = sad_u32(br, bfe(p, 0, 8), sr);
      sg = sad_u32(bg, bfe(p, 8, 8), sg);
      sb = sad_u32(bb, bfe(p, 16, 8), sb);
    }
  }
  const uint32_t dev = umax3(sr >> 3, sg >> 3, sb >> 3);
  const uint32_t cw = (dev > 144u) + (dev > 93u) + (dev > 70u) + (dev > 51u) + (dev > 35u) + (dev > 23u) + (dev > 12u);
  const uint32_t sh = (cw & 3u) * 8u;
  const uint32_t a = bfe(cw < 4u ? kEtcModA_lo : kEtcModA_hi, sh, 8);
  const uint32_t b = bfe(cw < 4u ? kEtcModB_lo : kEtcModB_hi, sh, 8);
  EtcSubResult r;
  uint32_t v[4];
  int32_t c[4];
  build_candidates(base, a, b, v, c);
  r.score = eval_codeword<FLIP, S>(px, v, c, &r.fields);
  r.cw = cw;
  return r;
}

// The base colours of a partition's two sub-blocks from their channel sums s0 / s1 (R, G, B), and the high word's
// flip / diff / colour bits (etc.cc:460-521): differential 5-5-5 + 3-bit deltas when every delta fits, else individual 4-4-4.
ICAMD_DEV uint32_t etc1_partition_bases(const uint32_t s0[3], const uint32_t s1[3], uint32_t flip_bit, uint32_t b0[3], uint32_t b1[3]) {
  // ComputeAverageColor (etc.cc:299-312): sum/8; QuantizeRgbFast<5>: >>3; <4>: >>4 (color_util.h:142-148)
  uint32_t q5a[3], q5b[3];
  bool diff_mode = true;
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    q5a[ch] = s0[ch] >> 6;
    q5b[ch] = s1[ch] >> 6;
    const int32_t d = (int32_t)q5b[ch] - (int32_t)q5a[ch];
    diff_mode = diff_mode && d >= -4 && d <= 3;
  }
  uint32_t hi = flip_bit;
  if (diff_mode) {
    hi |= 2u;
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      const uint32_t d3 = (q5b[ch] - q5a[ch]) & 7u;  // 3-bit two's complement (bit_util.h:46-57)
      hi |= q5a[ch] << (27 - 8 * ch) | d3 << (24 - 8 * ch);
      b0[ch] = (q5a[ch] << 3) | (q5a[ch] >> 2);  // Extend5Bit, color_util.h:200-202
      b1[ch] = (q5b[ch] << 3) | (q5b[ch] >> 2);
    }
  } else {
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      const uint32_t qa = s0[ch] >> 7, qb = s1[ch] >> 7;
      hi |= qa << (28 - 8 * ch) | qb << (24 - 8 * ch);
      b0[ch] = qa * 17u;  // Extend4Bit, color_util.h:193-195
      b1[ch] = qb * 17u;
    }
  }
  return hi;
}

struct EtcFlipResult {
  int32_t score;          // sum of both sub-block scores (Sum over the 16 pixels of max_k E)
  uint32_t hi;            // high word (flip, diff, codewords, colours), etc.cc:43-61
  uint32_t f0, f1;        // index fields of sub-block 0 / 1
};

// FindBestSubblockEncoding (etc.cc:460-542).  s0[], s1[] = channel sums (R,G,B) of the two sub-blocks.
template <int FLIP, bool TIER = false, bool PRUNE = true, bool SKIP = false>
ICAMD_DEV EtcFlipResult encode_flip(const uint32_t px[16], const uint32_t psum[16], const uint32_t s0[3],
                                    const uint32_t s1[3], bool heuristic, uint32_t flip_bit = (uint32_t)FLIP,
                                    bool skip = false, const uint32_t (*packed)[3][2] = nullptr) {
  uint32_t b0[3], b1[3];  // decoded base colours (what the decoder will reconstruct)
  const uint32_t hi = etc1_partition_bases(s0, s1, flip_bit, b0, b1);
  EtcBase e0 = { b0[0] << 24 | b0[2] << 8, b0[1] << 8 };
  EtcBase e1 = { b1[0] << 24 | b1[2] << 8, b1[1] << 8 };
  EtcSubResult r0, r1;
  if (heuristic) {
    r0 = heuristic_codeword<FLIP, 0>(px, e0, b0[0], b0[1], b0[2], packed ? packed[0] : nullptr);
    r1 = heuristic_codeword<FLIP, 1>(px, e1, b1[0], b1[1], b1[2], packed ? packed[1] : nullptr);
  } else {
    r0 = search_codewords<FLIP, 0, TIER, PRUNE, SKIP>(px, psum, e0, b0, s0, skip);
    r1 = search_codewords<FLIP, 1, TIER, PRUNE, SKIP>(px, psum, e1, b1, s1, skip);
  }
  EtcFlipResult out;
  out.hi = hi | r0.cw << 5 | r1.cw << 2;
  out.score = r0.score + r1.score;
  out.f0 = r0.fields;
  out.f1 = r1.fields;
  return out;
}

// Turn the two sub-blocks' index fields into the low word (etc.cc:150-156, :539): bit 4x+y gets the
// LSB of pixel (x,y)'s index, bit 16+4x+y the MSB.
ICAMD_DEV uint32_t assemble_indices(uint32_t f0, uint32_t f1, bool flip) {
  // fields hold (3 - k): invert; then z = [16 fields of sub0 | sub1] as pairs (lsb,msb)
  const uint32_t z = ~((f0 >> 16) | (f1 & 0xffff0000u));
  // split even (lsb) and odd (msb) bits: lsb plane -> low half, msb plane -> high half
  uint32_t lsb = z & 0x55555555u, msb = (z >> 1) & 0x55555555u;
  lsb = (lsb | lsb >> 1) & 0x33333333u; msb = (msb | msb >> 1) & 0x33333333u;
  lsb = (lsb | lsb >> 2) & 0x0f0f0f0fu; msb = (msb | msb >> 2) & 0x0f0f0f0fu;
  lsb = (lsb | lsb >> 4) & 0x00ff00ffu; msb = (msb | msb >> 4) & 0x00ff00ffu;
  lsb = (lsb | lsb >> 8) & 0x0000ffffu; msb = (msb | msb >> 8) & 0x0000ffffu;
  // now bit j of each plane = field j of sub-block 0 (j < 8) / sub-block 1 (j >= 8)
  uint32_t planes = lsb | msb << 16;
  if (flip) {
    // FLIP=1: field j of sub-block S belongs at bit 4(j>>1) + 2S + (j&1): spread bit pairs to
    // nibble starts, sub-block 1 shifted up by 2.
    uint32_t s0 = planes & 0x00ff00ffu, s1 = (planes >> 8) & 0x00ff00ffu;
    s0 = (s0 | s0 << 4) & 0x0f0f0f0fu; s1 = (s1 | s1 << 4) & 0x0f0f0f0fu;
    s0 = (s0 | s0 << 2) & 0x33333333u; s1 = (s1 | s1 << 2) & 0x33333333u;
    planes = s0 | s1 << 2;
  }
  return planes;
}

// EncodeEtc1Block (etc.cc:545-586).  Source channel order is always R,G,B (EtcCompressor accepts kRGB
// only, etc.cc:751-754).  Returns the 8 output bytes: hi word then lo word, each big-endian (etc.cc:172-180).
// Wave-uniform content probe for the kSmallerError kernels: true when at least three quarters of the wave's blocks have
// pixels whose sums r + g + b spread by 2 * 141 or more -- such a block has a pixel at least 141 (L1) from any colour it
// could average to, which switches the pruning of its searches off (search_codewords: d1 >= 3 * 47) and makes the mixed
// tier what its exact evaluations want.  The instantiation with the tier runs ~5 % slower wherever the tier is not used
// (register pressure), so a wave of mostly calm blocks takes the one without.  Only a performance choice: both produce
// the same bytes.
// (threshold sweep 564 / 200 / 100 / 50: profiles/r03_ab_etc1_mixed_tier.log, P)
#ifndef ICAMD_ETC1_BUSY_SPREAD
#define ICAMD_ETC1_BUSY_SPREAD (4u * 141u)
#endif
ICAMD_DEV uint32_t etc1_block_spread(const uint32_t px[16]) {
  uint32_t lo = 0xffffffffu, hi = 0u;
  ICAMD_UNROLL
  for (int p = 0; p < 16; ++p) {
    const uint32_t t = udot4(px[p], 0x00020202u, 0u);  // the same 2 (r + g + b) the searches use
    lo = umin(lo, t);
    hi = umax(hi, t);
  }
  return hi - lo;
}
ICAMD_DEV bool etc1_busy_wave(uint32_t spread) { return wave_count(spread >= ICAMD_ETC1_BUSY_SPREAD) >= 48u; }
ICAMD_DEV bool etc1_busy_wave(const uint32_t px[16]) { return etc1_busy_wave(etc1_block_spread(px)); }

// Per lane: the block is ONE colour (the empty regions of texture atlases, UI, vector art, padding).  Two stages so
// that other content pays one compare and a vote: the 16 pixels are only compared when the probe's spread is 0 in some
// lane of the wave.  Byte 3 of a pixel (the ignored alpha of RGBA8 sources) does not count.
ICAMD_DEV bool etc1_constant_block(const uint32_t px[16], uint32_t spread) {
  if (wave_all(spread != 0u)) return false;
  uint32_t diff = 0;
  ICAMD_UNROLL
  for (int p = 1; p < 16; ++p) diff |= px[p] ^ px[0];
  return (diff & 0x00ffffffu) == 0u;
}
// etc1_busy_wave among the lanes that take part in the searches (the others are one-colour blocks): three quarters of them
ICAMD_DEV bool etc1_busy_wave(uint32_t spread, bool skip) {
  return 4u * wave_count(!skip && spread >= ICAMD_ETC1_BUSY_SPREAD) >= 3u * wave_count(!skip);
}

// EncodeEtc1Block (etc.cc:545-586) for a block whose 16 pixels are the colour p (R | G << 8 | B << 16), strategies
// kSplitHorizontally / kSplitVertically / kSmallerError.  Every sub-block of either partition then holds eight copies of
// p: ComputeAverageColor (etc.cc:299-312) returns p itself, the 5-bit colours of the two halves are equal, so the block
// is differential with zero differences (etc.cc:486-505), both halves get the same codeword and every pixel the same
// modifier index -- ONE pixel against the 32 candidates (same keys and tie rules as eval_codeword: first codeword with
// the strictly smallest error, lowest index inside it) instead of four searches over eight pixels.  The two partitions
// of kSmallerError tie, which keeps flip = 0 (etc.cc:583).
ICAMD_DEV Out8 encode_etc1_constant_block(uint32_t p, uint32_t strategy) {
  const uint32_t pix = p & 0x00ffffffu;
  uint32_t hi = (strategy == 0u ? 1u : 0u) | 2u;  // kSplitHorizontally is the flip = 1 partition (etc.cc:549-551)
  uint32_t bch[3];
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t q5 = bfe(pix, 8 * ch, 8) >> 3;   // (8 p) / 8 >> 3
    hi |= q5 << (27 - 8 * ch);
    bch[ch] = (q5 << 3) | (q5 >> 2);                // Extend5Bit, color_util.h:200-202
  }
  const EtcBase base = { bch[0] << 24 | bch[2] << 8, bch[1] << 8 };
  int32_t best_e = 0;
  uint32_t best_cw = 0, best_field = 0;
  ICAMD_UNROLL
  for (int cw = 0; cw < 8; ++cw) {
    uint32_t v[4];
    int32_t c[4];
    build_candidates(base, (uint32_t)kEtcA[cw], (uint32_t)kEtcB[cw], v, c);
    const int32_t k0 = (int32_t)(udot4(pix, v[0], 0u) << 6) + c[0];
    const int32_t k1 = (int32_t)(udot4(pix, v[1], 0u) << 6) + c[1];
    const int32_t k2 = (int32_t)(udot4(pix, v[2], 0u) << 6) + c[2];
    const int32_t k3 = (int32_t)(udot4(pix, v[3], 0u) << 6) + c[3];
    const int32_t m = imax(imax3(k0, k1, k2), k3);  // 32 E + (3 - k) of the nearest candidate
    const int32_t e = m >> 5;                        // the sub-block's score is 8 E: comparing E is comparing errors
    const bool better = cw == 0 || e > best_e;
    best_e = better ? e : best_e;
    best_cw = better ? (uint32_t)cw : best_cw;
    best_field = better ? ((uint32_t)m & 3u) : best_field;
  }
  hi |= best_cw << 5 | best_cw << 2;
  const uint32_t k = 3u - best_field;  // all sixteen pixels take modifier index k: LSB plane in bits 0-15, MSB plane in 16-31
  const uint32_t lo = ((k & 1u) ? 0x0000ffffu : 0u) | ((k & 2u) ? 0xffff0000u : 0u);
  Out8 o = { perm(0u, hi, 0x00010203u), perm(0u, lo, 0x00010203u) };
  return o;
}

// TIER: compile the mixed tier (eval_codeword_mixed) into the codeword searches.  It pays on busy content only and its
// mere presence costs smooth / flat content 3-5 % (r03 A/B), so the kSmallerError kernels carry both instantiations and
// pick one per wave (etc1_busy_wave).
template <bool TIER = false, bool PRUNE = true, bool SKIP = false>
ICAMD_DEV Out8 encode_etc1_block(const uint32_t px[16], uint32_t strategy, bool skip = false) {
  // per-quadrant channel sums; quadrant q = 2*(y>=2) + (x>=2)
  uint32_t qs[4][3];
  uint32_t qpk[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };  // kHeuristic: the quadrant's four R / G / B bytes
  if (strategy == 3u) {
    // kHeuristic is short enough for the sums to matter: a 4 x 3 byte transpose per quadrant (seven v_perm) gives each
    // channel's four bytes in one dword; a quadrant sum is then ONE v_sad_u8 against 0, and the deviation sums of
    // FindCodewordHeuristic two v_sad_u8 per channel and sub-block (heuristic_codeword) -- 58 instructions where the
    // per-pixel forms take 144.
    ICAMD_UNROLL
    for (int q = 0; q < 4; ++q) {
      const int i0 = 8 * (q >> 1) + 2 * (q & 1);
      const uint32_t p0 = px[i0], p1 = px[i0 + 1], p2 = px[i0 + 4], p3 = px[i0 + 5];
      const uint32_t t0 = perm(p1, p0, 0x05010400u), t1 = perm(p3, p2, 0x05010400u);  // r r g g
      const uint32_t t2 = perm(p1, p0, 0x0c0c0602u), t3 = perm(p3, p2, 0x0c0c0602u);  // b b 0 0
      qpk[q][0] = perm(t1, t0, 0x05040100u);
      qpk[q][1] = perm(t1, t0, 0x07060302u);
      qpk[q][2] = perm(t3, t2, 0x05040100u);
      ICAMD_UNROLL
      for (int ch = 0; ch < 3; ++ch) qs[q][ch] = sad_u8(qpk[q][ch], 0u, 0u);
    }
  } else {
    ICAMD_UNROLL
    for (int q = 0; q < 4; ++q) {
      qs[q][0] = qs[q][1] = qs[q][2] = 0;
      ICAMD_UNROLL
      for (int i = 0; i < 4; ++i) {
        const int y = 2 * (q >> 1) + (i >> 1), x = 2 * (q & 1) + (i & 1);
        const uint32_t p = px[4 * y + x];
        qs[q][0] = udot4(p, 0x00000001u, qs[q][0]);
        qs[q][1] = udot4(p, 0x00000100u, qs[q][1]);
        qs[q][2] = udot4(p, 0x00010000u, qs[q][2]);
      }
    }
  }
  uint32_t psum[16];  // 2 (r + g + b) per pixel (for the unclamped shortcut)
  ICAMD_UNROLL
  for (int p = 0; p < 16; ++p) psum[p] = udot4(px[p], 0x00020202u, 0u);
  uint32_t left[3], right[3], top[3], bottom[3];
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    left[ch] = qs[0][ch] + qs[2][ch];
    right[ch] = qs[1][ch] + qs[3][ch];
    top[ch] = qs[0][ch] + qs[1][ch];
    bottom[ch] = qs[2][ch] + qs[3][ch];
  }
  EtcFlipResult res;
  bool flip;
  if (strategy == 0u) {  // kSplitHorizontally: top|bottom only
    res = encode_flip<1, TIER, PRUNE, SKIP>(px, psum, top, bottom, false, 1u, skip);
    flip = true;
  } else if (strategy == 1u) {  // kSplitVertically: left|right only
    res = encode_flip<0, TIER, PRUNE, SKIP>(px, psum, left, right, false, 0u, skip);
    flip = false;
  } else if (strategy == 3u) {  // kHeuristic, etc.cc:553-574: one evaluation, partition chosen per lane
    // the reference's fourth quadrant sum uses pixel (2,2) twice and never (3,3) (etc.cc:563-564)
    uint32_t e_lr = 0, e_tb = 0;
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      const uint32_t q3 = qs[3][ch] - bfe(px[15], 8 * ch, 8) + bfe(px[10], 8 * ch, 8);
      const uint32_t l = (qs[0][ch] + qs[2][ch]) >> 3, r = (qs[1][ch] + q3) >> 3;
      const uint32_t t = (qs[0][ch] + qs[1][ch]) >> 3, b = (qs[2][ch] + q3) >> 3;
      const uint32_t dlr = sad_u32(l, r, 0u), dtb = sad_u32(t, b, 0u);
      e_lr = umad24(dlr, dlr, e_lr);
      e_tb = umad24(dtb, dtb, e_tb);
    }
    flip = !(e_lr > e_tb);
    // The partition differs per lane; instead of evaluating both (or diverging), every lane gathers its 16 pixels
    // in the order of ITS partition -- (sub-block, j) as sub_pixel<flip> enumerates them -- with one v_cndmask per
    // position that differs, and the single evaluation below runs on that list.
    uint32_t pl[16];
    ICAMD_UNROLL
    for (int j = 0; j < 8; ++j) {
      pl[j] = sub_pixel<0, 0>(j) == sub_pixel<1, 0>(j) ? px[sub_pixel<0, 0>(j)]
                                                       : (flip ? px[sub_pixel<1, 0>(j)] : px[sub_pixel<0, 0>(j)]);
      pl[8 + j] = sub_pixel<0, 1>(j) == sub_pixel<1, 1>(j) ? px[sub_pixel<0, 1>(j)]
                                                           : (flip ? px[sub_pixel<1, 1>(j)] : px[sub_pixel<0, 1>(j)]);
    }
    uint32_t sa[3], sb[3];
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      sa[ch] = flip ? top[ch] : left[ch];
      sb[ch] = flip ? bottom[ch] : right[ch];
    }
    // the two sub-blocks' channel bytes: left | right = quadrants (0, 2) | (1, 3), top | bottom = (0, 1) | (2, 3)
    uint32_t packed[2][3][2];
    ICAMD_UNROLL
    for (int ch = 0; ch < 3; ++ch) {
      packed[0][ch][0] = qpk[0][ch];
      packed[0][ch][1] = flip ? qpk[1][ch] : qpk[2][ch];
      packed[1][ch][0] = flip ? qpk[2][ch] : qpk[1][ch];
      packed[1][ch][1] = qpk[3][ch];
    }
    res = encode_flip<2>(pl, psum, sa, sb, true, flip ? 1u : 0u, false, packed);
    res.score = 0;
  } else {  // kSmallerError (and the reference's default: label)
    const EtcFlipResult r0 = encode_flip<0, TIER, PRUNE, SKIP>(px, psum, left, right, false, 0u, skip);
    const EtcFlipResult r1 = encode_flip<1, TIER, PRUNE, SKIP>(px, psum, top, bottom, false, 1u, skip);
    // error_lr <= error_tb  <=>  score_lr >= score_tb  (same Sum|p|^2 on both sides)
    flip = !(r0.score >= r1.score);
    res.hi = flip ? r1.hi : r0.hi;
    res.f0 = flip ? r1.f0 : r0.f0;
    res.f1 = flip ? r1.f1 : r0.f1;
    res.score = 0;
  }
  const uint32_t lo = assemble_indices(res.f0, res.f1, flip);
  // big-endian words in memory = byte-reversed little-endian dwords
  Out8 o = { perm(0u, res.hi, 0x00010203u), perm(0u, lo, 0x00010203u) };
  return o;
}

// ---- kSmallerError with FOUR lanes per block (r05) -------------------------------------------------------------------
// For launches of FEW blocks that each need the whole search -- the pad blocks of an ETC1 Pad: 4 352 of them around a 4096^2
// texture, one per lane that is 65 waves of ~3 800 dependent instructions on 65 of the chip's 1 024 SIMDs.  The four searches
// of EncodeEtc1Block (etc.cc:575-583: two partitions x two sub-blocks) are independent once the partition's base colours are
// known, so lane t of a quad takes partition f = t >> 1, sub-block s = t & 1: it computes ITS partition's bases (both lanes of
// a partition do: a few dozen instructions), gathers its eight pixels into the (sub-block 0, j) slots and runs the one search
// every lane runs (search_codewords<2, 0>: the per-lane-partition form kHeuristic already uses).  The results meet by quad
// permutes (etc1_quad_finish): same keys, same tie rules, same bytes as encode_etc1_block.
struct EtcQuadPart {
  int32_t score;         // Sum over the lane's eight pixels of max_k E
  uint32_t cw, fields;   // its sub-block's codeword and index fields
  uint32_t hi;           // its partition's high word without the codewords (equal in both lanes of a partition)
};
template <bool TIER = false, bool PRUNE = true>
ICAMD_DEV EtcQuadPart etc1_quad_part(const uint32_t px[16], uint32_t f, uint32_t s) {
  uint32_t qs[4][3];  // per-quadrant channel sums; quadrant q = 2*(y>=2) + (x>=2)
  ICAMD_UNROLL
  for (int q = 0; q < 4; ++q) {
    qs[q][0] = qs[q][1] = qs[q][2] = 0;
    ICAMD_UNROLL
    for (int i = 0; i < 4; ++i) {
      const uint32_t p = px[4 * (2 * (q >> 1) + (i >> 1)) + 2 * (q & 1) + (i & 1)];
      qs[q][0] = udot4(p, 0x00000001u, qs[q][0]);
      qs[q][1] = udot4(p, 0x00000100u, qs[q][1]);
      qs[q][2] = udot4(p, 0x00010000u, qs[q][2]);
    }
  }
  uint32_t sa[3], sb[3];  // sub-blocks 0 / 1 of the lane's partition: left | right (f = 0) or top | bottom (f = 1)
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    sa[ch] = qs[0][ch] + (f ? qs[1][ch] : qs[2][ch]);
    sb[ch] = qs[3][ch] + (f ? qs[2][ch] : qs[1][ch]);
  }
  uint32_t b0[3], b1[3], bm[3], sm[3];
  EtcQuadPart out;
  out.hi = etc1_partition_bases(sa, sb, f, b0, b1);
  ICAMD_UNROLL
  for (int ch = 0; ch < 3; ++ch) {
    bm[ch] = s ? b1[ch] : b0[ch];
    sm[ch] = s ? sb[ch] : sa[ch];
  }
  uint32_t pl[16], ps[16];
  ICAMD_UNROLL
  for (int j = 0; j < 8; ++j) {
    const uint32_t v0 = s ? px[sub_pixel<0, 1>(j)] : px[sub_pixel<0, 0>(j)], v1 = s ? px[sub_pixel<1, 1>(j)] : px[sub_pixel<1, 0>(j)];
    pl[j] = f ? v1 : v0;
    ps[j] = udot4(pl[j], 0x00020202u, 0u);
    pl[8 + j] = 0u; ps[8 + j] = 0u;  // (never read: search_codewords<2, 0> looks at slots 0..7 only)
  }
  const EtcBase e = { bm[0] << 24 | bm[2] << 8, bm[1] << 8 };
  const EtcSubResult r = search_codewords<2, 0, TIER, PRUNE, false>(pl, ps, e, bm, sm);
  out.score = r.score; out.cw = r.cw; out.fields = r.fields;
  return out;
}
// mine: this lane's part; mate: the other sub-block of the same partition; other_score: the other partition's total.
// *writes: this lane holds the block (sub-block 0 of the winning partition).  error_lr <= error_tb keeps flip = 0 (etc.cc:583).
ICAMD_DEV Out8 etc1_quad_finish(const EtcQuadPart &mine, const EtcQuadPart &mate, int32_t other_score, uint32_t f, uint32_t s,
                                bool *writes) {
  const int32_t total = mine.score + mate.score;
  const int32_t score_lr = f ? other_score : total, score_tb = f ? total : other_score;
  const bool flip = !(score_lr >= score_tb);
  *writes = s == 0u && (f != 0u) == flip;
  const uint32_t hi = mine.hi | mine.cw << 5 | mate.cw << 2;
  const uint32_t lo = assemble_indices(mine.fields, mate.fields, flip);
  Out8 o = { perm(0u, hi, 0x00010203u), perm(0u, lo, 0x00010203u) };
  return o;
}
#if defined(ICAMD_HOST_EMULATION)
// the four lanes of a quad, one after the other (the host build has no lanes): what the quad kernels compute
ICAMD_DEV Out8 encode_etc1_block_quad(const uint32_t px[16]) {
  EtcQuadPart part[4];
  for (uint32_t t = 0; t < 4; ++t) part[t] = etc1_quad_part(px, t >> 1, t & 1u);
  Out8 out = { 0u, 0u };
  int writers = 0;
  for (uint32_t t = 0; t < 4; ++t) {
    bool writes = false;
    const Out8 o = etc1_quad_finish(part[t], part[t ^ 1u], part[t ^ 2u].score + part[t ^ 3u].score, t >> 1, t & 1u, &writes);
    if (writes) { out = o; ++writers; }
  }
  if (writers != 1) { out.lo = 0xdeadbeefu; out.hi = (uint32_t)writers; }  // (a test would see it)
  return out;
}
#else
// quad permutes (DPP quad_perm [1,0,3,2] / [2,3,0,1]): lane t reads lane t ^ 1 / t ^ 2 of its quad
ICAMD_DEV uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xb1, 0xf, 0xf, true); }
ICAMD_DEV uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4e, 0xf, 0xf, true); }
// Called by ALL FOUR lanes of a quad (t = lane & 3) with the same pixels.  *writes: this lane stores the result.
ICAMD_DEV Out8 encode_etc1_block_quad(const uint32_t px[16], uint32_t t, bool *writes) {
  const uint32_t f = t >> 1, s = t & 1u;
  const EtcQuadPart mine = etc1_quad_part(px, f, s);
  EtcQuadPart mate;
  mate.score = (int32_t)quad_xor1((uint32_t)mine.score);
  mate.cw = quad_xor1(mine.cw);
  mate.fields = quad_xor1(mine.fields);
  mate.hi = mine.hi;
  const int32_t total = mine.score + mate.score;
  const int32_t other = (int32_t)quad_xor2((uint32_t)total);
  return etc1_quad_finish(mine, mate, other, f, s, writes);
}
#endif

}  // namespace icamd
#endif  // ICAMD_ETC1_BLOCK_H_
