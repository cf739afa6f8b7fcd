// etc1_kernels.hip -- ETC1 encode kernels for gfx950 (MI355X); see etc1_block.h for the math.
// Same one-block-per-lane tile mapping as dxt_kernels.hip, with 16 x 16-block tiles.  VALU-bound (3.5-4.3 k integer instructions per block at
// kSmallerError, 98 % issue utilisation); the 3.5 / 4.5 B/px of HBM traffic are a small fraction of the roofline.
#include <cstdlib>
#include "etc1_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

// Calm waves take the instantiation WITHOUT the mixed tier: with it (and pruning) smooth content measured 1.466 -> 1.505 ms
// and flat 1.35 -> 1.42 ms (r03, profiles/r03_ab_etc1_mixed_tier.log) -- there the tier's register pressure costs more than
// its arithmetic saves.
#ifndef ICAMD_ETC1_CALM_TIER
#define ICAMD_ETC1_CALM_TIER false
#endif

namespace icamd {

// STRATEGY is a compile-time constant: one kernel per EtcCompressor::CompressionStrategy, so that the straight-line
// code of kSmallerError (two partitions x two sub-blocks x eight codewords, unrolled) does not carry the single-partition
// and heuristic paths along -- the kernel is bound by instruction issue AND sensitive to its code footprint
// (profiles/r03_ab_etc1_*.log).  The default: label of etc_compressor.cc:575-584 makes every other value kSmallerError.
// The part of etc1_encode_one after the block is in registers: the per-wave choice of instantiation.  constant / busy are
// per-lane properties of the block (one colour; sums r + g + b spread by ICAMD_ETC1_BUSY_SPREAD or more).
template <int STRATEGY>
__device__ __forceinline__ Out8 etc1_encode_classified(const uint32_t px[16], bool constant, bool busy) {
  Out8 c;
  // One-colour blocks are encoded by a form of their own (one pixel against the 32 candidates).  A wave of nothing else
  // skips the searches altogether; inside a mixed wave those lanes neither vote in the searches' wave-uniform
  // decisions nor count for the content probe, and their results are replaced afterwards.
  if (wave_all(!constant)) {  // (the common case: exactly the code of a build without the one-colour forms)
    if (wave_count(busy) >= 48u) c = encode_etc1_block<true, false>(px, (uint32_t)STRATEGY);  // busy wave: mixed tier, no pruning
    else c = encode_etc1_block<ICAMD_ETC1_CALM_TIER, true>(px, (uint32_t)STRATEGY);           // calm wave: pruning (+ tier?)
  } else if (wave_all(constant)) {
    c = encode_etc1_constant_block(px[0], (uint32_t)STRATEGY);
  } else {
    if (4u * wave_count(!constant && busy) >= 3u * wave_count(!constant))
      c = encode_etc1_block<true, false, true>(px, (uint32_t)STRATEGY, constant);
    else c = encode_etc1_block<ICAMD_ETC1_CALM_TIER, true, true>(px, (uint32_t)STRATEGY, constant);
    const Out8 cc = encode_etc1_constant_block(px[0], (uint32_t)STRATEGY);
    c.lo = constant ? cc.lo : c.lo;
    c.hi = constant ? cc.hi : c.hi;
  }
  return c;
}

// STRATEGY is a compile-time constant: one kernel per EtcCompressor::CompressionStrategy, so that the straight-line
// code of kSmallerError (two partitions x two sub-blocks x eight codewords, unrolled) does not carry the single-partition
// and heuristic paths along -- the kernel is bound by instruction issue AND sensitive to its code footprint
// (profiles/r03_ab_etc1_*.log).  The default: label of etc_compressor.cc:575-584 makes every other value kSmallerError.
//
// REGROUP (r04, the searching strategies): a one-colour block is encoded by a form that costs a tenth of the searches, but
// only a wave of 64 such blocks skips the searches -- one noisy tile among flat ones (UI, atlases, map tiles) makes its
// 63 neighbours wait.  A workgroup (256 blocks) that holds BOTH kinds therefore partitions itself first: per-wave ballots
// give the counts and the rank inside the kind, the 16 pixel dwords travel through LDS to the lane of their rank -- one-colour
// blocks first, the rank rotated by the workgroup's index so that the searching waves of neighbouring workgroups land on
// different SIMDs -- and each lane stores its result straight to the block it now holds.  Workgroups of one kind (all of
// noise / photographic content) pay two ballots, one barrier and a few adds, nothing else.
// (r04 also measured the full version of VERDICT r03 item 2 -- a counting sort of all 256 blocks by (one colour | calm |
//  busy) x room-to-0/255 class, results returned through LDS: noise 248.7 -> 233.6, smooth 182.4 -> 174.6, flat 338.7 ->
//  328.6 Gpix/s, i.e. slower everywhere.  scripts/etc1_search_sim.py shows why: a block's four searches have four different
//  rooms (two partitions x two sub-blocks), so one per-block key homogenises none of them on noise, smooth gradients are
//  homogeneous per wave already, and a returning barrier parks the finished one-colour waves behind the searching one.)
// NOT SHIPPED (default 0): three versions measured, none faster on any content -- profiles/r04_ab_etc1_regroup.log (v3, this
// code: noise 249.3 -> 241.6, smooth 180.9 -> 178.3, flat 332.9 -> 336.9 Gpix/s).  Kept behind the macro for A/B runs.
#ifndef ICAMD_ETC1_REGROUP
#define ICAMD_ETC1_REGROUP 0
#endif

// Lane -> block inside a 16 x 16-block tile.  ICAMD_ETC1_WAVE_8X8: a wave covers 8 x 8 blocks (32 x 32 pixels) instead of
// 16 x 4 (64 x 16 pixels): more compact content per wave-uniform decision, 96- instead of 192-byte row segments per wave.
#ifndef ICAMD_ETC1_WAVE_8X8
#define ICAMD_ETC1_WAVE_8X8 0
#endif
// ICAMD_ETC1_WAVE_WORKGROUPS (r05, default): every WAVE of a tile is a workgroup of its own (64 lanes; blockIdx.x = 4 x tile
// column + wave).  Same tiles, same waves, same content per wave -- but a wave no longer shares its workgroup's fate.  The
// search's cost depends on the content (shortcuts, pruning, one-colour forms are decided per wave), and a four-wave workgroup
// is only replaced as a whole: on smooth 1024^2 textures the PMC counters showed 2.5 resident waves per SIMD of 4
// (SQ_WAVE_CYCLES; noise: 3.7) while every resident wave issued as on noise.  A/B (profiles/r05_ab_etc1_wave_workgroups.log):
// c4 smooth 143 -> 182, flat 345 -> 433 Gpix/s, 16 x 4096^2 smooth 182 -> 200, flat 337 -> 418, noise 253 -> 254 (= 0):
// what three rounds of instruction-level work on the smooth case (r03 / r04) could not move was an occupancy effect.
#ifndef ICAMD_ETC1_WAVE_WORKGROUPS
#define ICAMD_ETC1_WAVE_WORKGROUPS 1
#endif
#if ICAMD_ETC1_WAVE_WORKGROUPS && ICAMD_ETC1_REGROUP
#error "ICAMD_ETC1_REGROUP exchanges blocks between the four waves of a tile: it needs ICAMD_ETC1_WAVE_WORKGROUPS=0"
#endif
// (kHeuristic has no content-dependent path: it keeps the four-wave workgroups, whose dispatch costs a quarter as much)
constexpr bool etc1_wave_workgroups(int strategy) { return ICAMD_ETC1_WAVE_WORKGROUPS != 0 && strategy != 3; }
// ICAMD_ETC1_XCD_COLUMNS (r06, 3-byte sources): which tile column a workgroup takes.  A wave of a 16 x 16-block RGB888 tile reads
// 192-byte row segments: tiles 2 k and 2 k + 1 together cover three 128-byte lines and SHARE the middle one.  Workgroups are dealt
// to the eight XCDs round-robin in launch order (x fastest), each XCD with an L2 of its own -- with the plain mapping (tile column
// = blockIdx.x >> 2, wave = blockIdx.x & 3) the two tiles of such a pair ALWAYS sit on different XCDs (their workgroup indices
// differ by 4), so every shared line crosses the fabric twice: FETCH_SIZE 1.29 x the algorithmic bytes in every round's profile.
// Mode 2 (default): XCD r = blockIdx.x & 7 takes wave r & 3 of the tile PAIRS with parity r >> 2, q = blockIdx.x >> 3 counting its
// tiles along the row -- the two tiles of a pair run the same wave on the SAME XCD one dispatch apart, and the second reader finds
// the line in L2: counter traffic 1.286 -> 1.009 x (kHeuristic, four-wave workgroups: 1.286 -> 1.14, and + 1.7 % Mpixels/s), same
// tiles, same waves, same lanes, same bytes out, 4 scalar instructions, time unchanged on every content.
// Modes 1 (the left / right HALF of a tile row per XCD group) and 4 (QUADS of tiles in the order A B B A) reach the same traffic
// and LOSE 8 % on smooth content (c4: 182.5 -> 167.5 Gpixel/s): the partition is static, the search's cost follows the image, and a
// gradient along the row loads the XCD groups unequally -- the finest grouping that still keeps a pair together is the one to
// take (profiles/r06_ab_etc1_xcd.log).  0 = the plain mapping.
#ifndef ICAMD_ETC1_XCD_COLUMNS
#define ICAMD_ETC1_XCD_COLUMNS 2
#endif
// blockIdx.x -> (tile column, wave of the tile) for one-wave workgroups; -> tile column for four-wave workgroups (wave unused)
// (bx, gx: blockIdx.x and gridDim.x)
// ICAMD_ETC1_ROTATE_ROWS (r06): the order in which an XCD's workgroups of a tile row take its tiles is ROTATED by
// the row (+ image) index.  An XCD's workgroups are dealt to its shader engines in a fixed rotation too (XCD-local index mod 4), and
// with 8 workgroups per XCD and tile row (config c4: 1024^2 textures) that made engine e the owner of tiles q = e, e + 4 of EVERY
// row: a static partition of the image by column, which a gradient along the row loads unequally -- the same effect the coarse XCD
// groupings showed (profiles/r06_ab_etc1_xcd.log), one level down.  SQ / SPI counters: 3.09 resident waves per SIMD on smooth
// content with room in the SIMDs (profiles/r06_valu_counters.txt).  With the rotation every engine sees every column within four
// rows: c4 smooth 182 -> 194 Gpixel/s, noise / flat / 16 x 4096^2 (32 workgroups per XCD and row: already spread) unchanged
// (profiles/r06_ab_etc1_rotate.log; kHeuristic's four-wave workgroups, 4096-px rows = 8 per XCD and row: smooth + 2 %, 2048^2 + 3-6 %).
// Four scalar instructions, any extent of at least four such tiles; bx mod 8 -- the XCD -- is kept.
#ifndef ICAMD_ETC1_ROTATE_ROWS
#define ICAMD_ETC1_ROTATE_ROWS 1
#endif
template <int COMPS, bool WAVE_WORKGROUPS>
__device__ __forceinline__ void etc1_tile_of_workgroup(uint32_t bx, uint32_t gx, uint32_t &tile_col, uint32_t &wave, uint32_t by = 0u) {
  constexpr uint32_t kMode = ICAMD_ETC1_XCD_COLUMNS;
  if (ICAMD_ETC1_ROTATE_ROWS && (gx & 7u) == 0u && gx >= 32u) {
    const uint32_t nq = gx >> 3;
    uint32_t q = (bx >> 3) + (by & 3u);  // (four engines: a rotation by row mod 4 is all it takes, and it needs no division)
    q -= q >= nq ? nq : 0u;
    bx = (q << 3) | (bx & 7u);
  }
  if (WAVE_WORKGROUPS) {
    tile_col = bx >> 2;
    wave = bx & 3u;
    if (kMode == 0u || COMPS != 3) return;
    // XCD r = bx & 7 (workgroups are dealt round-robin in launch order, x fastest, and gx is a multiple of 8 below): g = r >> 2
    // picks the group of tile columns, q = bx >> 3 counts the group's tiles along the row; tiles per row = gx / 4
    const uint32_t r = bx & 7u, g = r >> 2, q = bx >> 3;
    if (kMode == 1u && (gx & 7u) == 0u) {
      wave = r & 3u;
      tile_col = g * (gx >> 3) + q;
    } else if (kMode == 2u && (gx & 15u) == 0u) {
      wave = r & 3u;
      tile_col = 4u * (q >> 1) + 2u * g + (q & 1u);
    } else if (kMode == 4u && (gx & 31u) == 0u) {
      const uint32_t m = q >> 2;
      wave = r & 3u;
      tile_col = 8u * m + 4u * (g ^ (m & 1u)) + (q & 3u);
    }
  } else {
    // four-wave workgroups (kHeuristic): XCD r = bx & 7 takes every 8th RUN of tiles; runs of gx / 8 (mode 1), 2 or 4 tiles
    tile_col = bx;
    wave = 0u;
    if (kMode == 0u || COMPS != 3) return;
    const uint32_t r = bx & 7u, q = bx >> 3;
    if (kMode == 1u && (gx & 7u) == 0u) tile_col = r * (gx >> 3) + q;
    else if (kMode == 2u && (gx & 15u) == 0u) tile_col = 16u * (q >> 1) + 2u * r + (q & 1u);
    else if (kMode == 4u && (gx & 31u) == 0u) tile_col = 32u * (q >> 2) + 4u * r + (q & 3u);
  }
}
template <int STRATEGY, int COMPS>
__device__ __forceinline__ TileCoord etc1_locate_tile(const GridParams &P, uint32_t bx, uint32_t by, uint32_t bz, uint32_t gx) {
  if (etc1_wave_workgroups(STRATEGY)) {
  TileCoord t;
  const uint32_t cols = 1u << P.log2_tile_cols, rows = 256u >> P.log2_tile_cols;
  uint32_t tile_col, wave;
  etc1_tile_of_workgroup<COMPS, true>(bx, gx, tile_col, wave, by + bz);
  const uint32_t vt = threadIdx.x + 64u * wave;
  t.lx = vt & (cols - 1u);
  t.ly = vt >> P.log2_tile_cols;
  t.bcol0 = tile_col * cols;
  t.brow0 = (by + P.tile_row0) * rows;
  t.bcol = t.bcol0 + t.lx;
  t.brow = t.brow0 + t.ly;
  t.img = bz;
  t.full = t.bcol0 + cols <= P.block_cols && t.brow0 + rows <= P.block_rows;
  t.interior = (t.bcol0 + cols) * 4u <= P.width && (t.brow0 + rows) * 4u <= P.height;
  t.valid = t.full || (t.bcol < P.block_cols && t.brow < P.block_rows);
  return t;
  }
  uint32_t tile_col, wave;
  etc1_tile_of_workgroup<COMPS, false>(bx, gx, tile_col, wave, by + bz);
  TileCoord t = locate_tile<false>(P, tile_col);
  if (ICAMD_ETC1_WAVE_8X8 && P.log2_tile_cols == 4u) {
    const uint32_t tid = threadIdx.x;
    t.lx = (tid & 7u) + ((tid >> 6) & 1u) * 8u;
    t.ly = ((tid >> 3) & 7u) + (tid >> 7) * 8u;
    t.bcol = t.bcol0 + t.lx;
    t.brow = t.brow0 + t.ly;
    t.valid = t.full || (t.bcol < P.block_cols && t.brow < P.block_rows);
  }
  return t;
}

template <int COMPS, int STRATEGY>
__device__ __forceinline__ void etc1_encode_one(const GridParams &P, uint32_t bx, uint32_t by, uint32_t bz, uint32_t gx) {
  const TileCoord t = etc1_locate_tile<STRATEGY, COMPS>(P, bx, by, bz, gx);
  if (STRATEGY == 3 || !ICAMD_ETC1_REGROUP) {
    if (!t.valid) return;
    uint32_t px[16];
    load_tile_block<COMPS>(P, t, px);
    Out8 c;
    if (STRATEGY == 3) {
      c = encode_etc1_block<false>(px, 3u);
    } else {
      const uint32_t spread = etc1_block_spread(px);
      c = etc1_encode_classified<STRATEGY>(px, etc1_constant_block(px, spread), spread >= ICAMD_ETC1_BUSY_SPREAD);
    }
    store_stream8(tile_dst<8>(P, t), c.lo, c.hi);
    return;
  }
  __shared__ uint32_t lds_px[17][kThreadsPerWorkgroup];  // [pixel 0..15, (busy | destination offset)][rank]
  __shared__ uint32_t lds_cnt[2][4];                      // per wave: one-colour blocks, searching blocks
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  uint32_t px[16];
  uint32_t kind = 2u;  // lanes without a block (partial tiles at the image's edges)
  bool busy = false;
  if (t.valid) {
    load_tile_block<COMPS>(P, t, px);
    const uint32_t spread = etc1_block_spread(px);
    busy = spread >= ICAMD_ETC1_BUSY_SPREAD;
    kind = etc1_constant_block(px, spread) ? 0u : 1u;
  }
  // counts per wave by ballot (no atomics: 64 lanes on one LDS counter serialise), one LDS word per wave and kind
  const uint64_t m_const = __ballot(kind == 0u), m_search = __ballot(kind == 1u);
  if ((tid & 63u) == 0u) {
    lds_cnt[0][wave] = (uint32_t)__popcll(m_const);
    lds_cnt[1][wave] = (uint32_t)__popcll(m_search);
  }
  __syncthreads();
  uint32_t n_const = 0, n_search = 0, before_const = 0, before_search = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4u; ++w) {
    const uint32_t c = lds_cnt[0][w], q = lds_cnt[1][w];
    before_const += w < wave ? c : 0u;
    before_search += w < wave ? q : 0u;
    n_const += c;
    n_search += q;
  }
  uint8_t *const dst_base = P.dst + (uint64_t)t.img * P.dst_image_stride + ((uint64_t)t.brow0 * P.block_cols + t.bcol0) * 8u;
  const uint32_t dst_off = (t.ly * P.block_cols + t.lx) * 8u;
  if (n_const == 0u || n_search == 0u) {  // one kind only (workgroup-uniform): nothing to regroup
    if (kind == 2u) return;
    const Out8 c = etc1_encode_classified<STRATEGY>(px, kind == 0u, busy);
    store_stream8(dst_base + dst_off, c.lo, c.hi);
    return;
  }
  // rank: one-colour blocks, then searching blocks, then the lanes without a block; rotated by a hash of the workgroup's
  // index -- a CU receives workgroups whose indices differ by a multiple of the CU count, so anything periodic in
  // blockIdx would put the searching wave of every resident workgroup on the same SIMD
  const uint32_t lane_lt = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_const >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_const, 0u));
  const uint32_t lane_lt_s = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_search >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_search, 0u));
  const uint32_t wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const uint32_t rot = ((wg * 0x9e3779b1u) >> 30) * 64u;
  const uint32_t tail = n_const + n_search;  // (lanes without a block: any distinct slots past the blocks)
  const uint32_t plain = kind == 0u ? before_const + lane_lt : kind == 1u ? n_const + before_search + lane_lt_s
                                    : tail + (tid - (before_const + lane_lt + before_search + lane_lt_s));
  const uint32_t rank = (plain + rot) & 255u;
#pragma unroll
  for (int i = 0; i < 16; ++i) lds_px[i][rank] = kind == 2u ? 0u : px[i];
  // destination offsets are multiples of 8: bit 0 carries `busy`, bit 1 `one colour`, bit 2 `no block`
  lds_px[16][rank] = dst_off | (busy ? 1u : 0u) | (kind == 0u ? 2u : 0u) | (kind == 2u ? 4u : 0u);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) px[i] = lds_px[i][tid];
  const uint32_t tag = lds_px[16][tid];
  if (tag & 4u) return;
  const Out8 c = etc1_encode_classified<STRATEGY>(px, (tag & 2u) != 0u, (tag & 1u) != 0u);
  store_stream8(dst_base + (tag & ~7u), c.lo, c.hi);
}

// ---- kSmallerError for SMALL launches (r05): four lanes per block.  A 256^2 texture is 4 096 blocks = 64 waves of ~2 600
// dependent instructions on 64 of the chip's 1 024 SIMDs: latency, not throughput.  With the four searches of a block on four
// lanes (etc1_block.h encode_etc1_block_quad, the form the Pad border uses) it is 256 waves of ~900.  The tile stays 16 x 16
// blocks; its 1 024 lanes are 16 one-wave workgroups of 16 blocks (one row of the tile) x 4 lanes.  One-colour blocks keep
// their own form (lane 0 of the quad writes it).  launch_etc1 takes this kernel while the launch has at most
// kEtc1QuadMaxBlocks (36 864) blocks -- above that every SIMD has a wave of the one-lane form anyway, which needs fewer instructions
// in total.  Same bytes: tests/test_gpu_parity.py runs both forms on the same inputs.
template <int COMPS>
__device__ __forceinline__ void etc1_encode_quad(const GridParams &P) {
  TileCoord t;
  const uint32_t cols = 1u << P.log2_tile_cols, rows = 256u >> P.log2_tile_cols;
  const uint32_t vb = (threadIdx.x >> 2) + 16u * (blockIdx.x & 15u), q = threadIdx.x & 3u;  // block of the tile, lane of the quad
  t.lx = vb & (cols - 1u);
  t.ly = vb >> P.log2_tile_cols;
  t.bcol0 = (blockIdx.x >> 4) * cols;
  t.brow0 = (blockIdx.y + P.tile_row0) * rows;
  t.bcol = t.bcol0 + t.lx;
  t.brow = t.brow0 + t.ly;
  t.img = blockIdx.z;
  t.full = t.bcol0 + cols <= P.block_cols && t.brow0 + rows <= P.block_rows;
  t.interior = (t.bcol0 + cols) * 4u <= P.width && (t.brow0 + rows) * 4u <= P.height;
  t.valid = t.full || (t.bcol < P.block_cols && t.brow < P.block_rows);
  if (!t.valid) return;  // (whole quads)
  uint32_t px[16];
  load_tile_block<COMPS>(P, t, px);
  uint32_t diff = 0;
#pragma unroll
  for (int p = 1; p < 16; ++p) diff |= px[p] ^ px[0];
  Out8 c;
  bool writes;
  if ((diff & 0x00ffffffu) == 0u) {  // one colour (per quad): the form of its own, no search
    c = encode_etc1_constant_block(px[0], 2u);
    writes = q == 0u;
  } else {
    c = encode_etc1_block_quad(px, q, &writes);
  }
  if (writes) store_stream8(tile_dst<8>(P, t), c.lo, c.hi);
}

extern "C" {
__global__ void __launch_bounds__(64) icamd_etc1_rgb888_quad_kernel(GridParams P) { etc1_encode_quad<3>(P); }
__global__ void __launch_bounds__(64) icamd_etc1_rgba8_quad_kernel(GridParams P) { etc1_encode_quad<4>(P); }
}

extern "C" {

// (amdgpu_waves_per_eu(4): the search must fit 128 VGPRs -- left alone the allocator takes 139 for kSmallerError with the
// mixed tier, i.e. 3 waves per SIMD, which costs smooth / flat content 10-18 %)
#define ICAMD_ETC1_KERNEL(name, comps, strategy)                                                                      \
  __global__ void __launch_bounds__(kThreadsPerWorkgroup) __attribute__((amdgpu_waves_per_eu(4))) name(GridParams P) { \
    etc1_encode_one<comps, strategy>(P, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);                               \
  }
ICAMD_ETC1_KERNEL(icamd_etc1_rgb888_kernel, 3, 2)        // kSmallerError (the reference's default)
ICAMD_ETC1_KERNEL(icamd_etc1_rgba8_kernel, 4, 2)
ICAMD_ETC1_KERNEL(icamd_etc1_rgb888_split_h_kernel, 3, 0)  // kSplitHorizontally
ICAMD_ETC1_KERNEL(icamd_etc1_rgba8_split_h_kernel, 4, 0)
ICAMD_ETC1_KERNEL(icamd_etc1_rgb888_split_v_kernel, 3, 1)  // kSplitVertically
ICAMD_ETC1_KERNEL(icamd_etc1_rgba8_split_v_kernel, 4, 1)
ICAMD_ETC1_KERNEL(icamd_etc1_rgb888_heuristic_kernel, 3, 3)  // kHeuristic
ICAMD_ETC1_KERNEL(icamd_etc1_rgba8_heuristic_kernel, 4, 3)
#undef ICAMD_ETC1_KERNEL

}  // extern "C"

#if defined(ICAMD_ETC1_STATS)
// diagnostics build: read (and optionally clear) the path counters
extern "C" __attribute__((visibility("default"))) int icamd_debug_etc1_stats(unsigned int *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_etc1_stats), 64) != hipSuccess) return -1;
  if (reset) {
    unsigned int z[16] = { 0 };
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_etc1_stats), z, 64) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// measured (profiles/r05_ab_etc1_small.log): one call alone 17.1 -> 12.2 us at 256^2, 17.4 -> 12.8 at 512^2, 17.9 -> 16.5 at 768^2,
// 18.3 -> 19.0 at 1024^2 (there every SIMD has a wave of the one-lane form, which needs fewer instructions in total)
constexpr uint32_t kEtc1QuadMaxBlocks = 36864;  // one 768^2 texture, two 512^2, nine 256^2

const char *etc1_kernel_name(int comps) { return comps == 4 ? "icamd_etc1_rgba8_kernel" : "icamd_etc1_rgb888_kernel"; }

hipError_t launch_etc1(int comps, const GridParams &P, hipStream_t stream) {
  // 16 x 16-block tiles (a wave = 16 x 4 blocks = 64 x 16 pixels) instead of 256 x 1: the encoder's wave-uniform
  // decisions (unclamped shortcut, codeword pruning) fire far more often on compact waves, and at 7 % of the HBM
  // roofline the narrower loads cost nothing: noise 1.47 = 1.47 ms, smooth 1.85 -> 1.61 ms, flat 1.90 -> 1.72 ms (r01)
  // (RGB888: a 16-block tile row is 192 bytes -- it straddles 128-byte lines that the neighbouring tile reads too, and the
  //  counters show 1.29 x the algorithmic bytes; 32 x 8-block tiles make it 384 bytes = three whole lines:
  //  ICAMD_ETC1_RGB888_TILE_LOG2, A/B in profiles/r04_ab_etc1_tiles_waves.log: traffic 1.33 -> 1.00 x but smooth -21 %, flat -17 %: a wave becomes 32 x 2 blocks)
#ifndef ICAMD_ETC1_RGB888_TILE_LOG2
#define ICAMD_ETC1_RGB888_TILE_LOG2 4u
#endif
  const uint32_t cap = comps == 3 ? ICAMD_ETC1_RGB888_TILE_LOG2 : 4u;
  typedef void (*Kernel)(GridParams);
  static const Kernel kernels[2][4] = {
    { icamd_etc1_rgb888_split_h_kernel, icamd_etc1_rgb888_split_v_kernel, icamd_etc1_rgb888_kernel, icamd_etc1_rgb888_heuristic_kernel },
    { icamd_etc1_rgba8_split_h_kernel, icamd_etc1_rgba8_split_v_kernel, icamd_etc1_rgba8_kernel, icamd_etc1_rgba8_heuristic_kernel } };
  const uint32_t strategy = P.etc_strategy < 4u ? P.etc_strategy : 2u;
  // r05: small kSmallerError launches take four lanes per block (etc1_encode_quad); ICAMD_ETC1_QUAD_MAX_BLOCKS overrides the
  // threshold (0 = never) for the A/B
  static const uint64_t quad_max = [] {
    const char *e = getenv("ICAMD_ETC1_QUAD_MAX_BLOCKS");
    return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)kEtc1QuadMaxBlocks;
  }();
  if (strategy == 2u && (uint64_t)P.block_rows * P.block_cols * P.n_images <= quad_max && P.block_cols < (1u << 20)) {
    const Kernel q = comps == 4 ? icamd_etc1_rgba8_quad_kernel : icamd_etc1_rgb888_quad_kernel;
    return launch_tiled(q, q, P, stream, 4u, 1, true, 4u);
  }
  // (r06, measured and NOT shipped -- profiles/r06_ab_etc1_persistent.log: PERSISTENT waves, four one-wave workgroups per SIMD each
  //  walking 1 / 4 096 of the launch's tile waves, to keep every slot filled on content whose per-wave cost varies (3.09 resident
  //  waves per SIMD on smooth content against 3.86 on noise).  16-38 % SLOWER on every content, scrambled walk or not: the hardware
  //  does not spread a grid that exactly fills the chip evenly -- 2.74 resident waves per SIMD, the late workgroups run alone at
  //  the end -- and a work queue that would make late starters harmless needs a per-launch counter.  Removed.
  //  Second form: k = 2 / 4 / 8 tile waves per one-wave workgroup, taken from k distant places of the launch, the grid still many
  //  times the chip: c4 smooth + 5 % at k = 4 (182 -> 192 Gpixel/s: averaging the content does refill the slots), but the loop costs
  //  the search 2.3 % on noise and 3.5 % on flat content (24 bytes of scratch, 106 SGPRs) and 16 x 4096^2 smooth loses 2.5 %.  Removed.)
  const Kernel k = kernels[comps == 4 ? 1 : 0][strategy];
  return launch_tiled(k, k, P, stream, cap, 1, etc1_wave_workgroups(P.etc_strategy < 4u ? (int)P.etc_strategy : 2));
}

}  // namespace icamd
