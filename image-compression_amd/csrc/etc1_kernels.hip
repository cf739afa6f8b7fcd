// etc1_kernels.hip -- ETC1 encode kernels for gfx950 (MI355X); see etc1_block.h for the math.
// Same one-block-per-lane tile mapping as dxt_kernels.hip, with 16 x 16-block tiles.  VALU-bound (3.5-4.3 k integer instructions per block at
// kSmallerError, 98 % issue utilisation); the 3.5 / 4.5 B/px of HBM traffic are a small fraction of the roofline.
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
template <int COMPS, int STRATEGY>
__device__ __forceinline__ void etc1_encode_one(const GridParams &P) {
  const TileCoord t = locate_tile<false>(P);
  if (!t.valid) return;
  uint32_t px[16];
  load_tile_block<COMPS>(P, t, px);
  Out8 c;
  if (STRATEGY == 3) {
    c = encode_etc1_block<false>(px, 3u);
  } else {
    // One-colour blocks are encoded by a form of their own (one pixel against the 32 candidates).  A wave of nothing else
    // skips the searches altogether; inside a mixed wave those lanes neither vote in the searches' wave-uniform
    // decisions nor count for the content probe, and their results are replaced afterwards.
    const uint32_t spread = etc1_block_spread(px);
    const bool constant = etc1_constant_block(px, spread);
    if (wave_all(!constant)) {  // (the common case: exactly the code of a build without the one-colour forms)
      if (etc1_busy_wave(spread)) c = encode_etc1_block<true, false>(px, (uint32_t)STRATEGY);  // busy wave: mixed tier, no pruning
      else c = encode_etc1_block<ICAMD_ETC1_CALM_TIER, true>(px, (uint32_t)STRATEGY);          // calm wave: pruning (+ tier?)
    } else if (wave_all(constant)) {
      c = encode_etc1_constant_block(px[0], (uint32_t)STRATEGY);
    } else {
      if (etc1_busy_wave(spread, constant)) c = encode_etc1_block<true, false, true>(px, (uint32_t)STRATEGY, constant);
      else c = encode_etc1_block<ICAMD_ETC1_CALM_TIER, true, true>(px, (uint32_t)STRATEGY, constant);
      const Out8 cc = encode_etc1_constant_block(px[0], (uint32_t)STRATEGY);
      c.lo = constant ? cc.lo : c.lo;
      c.hi = constant ? cc.hi : c.hi;
    }
  }
  store_stream8(tile_dst<8>(P, t), c.lo, c.hi);
}

extern "C" {

// (amdgpu_waves_per_eu(4): the search must fit 128 VGPRs -- left alone the allocator takes 139 for kSmallerError with the
// mixed tier, i.e. 3 waves per SIMD, which costs smooth / flat content 10-18 %)
#define ICAMD_ETC1_KERNEL(name, comps, strategy)                                                                      \
  __global__ void __launch_bounds__(kThreadsPerWorkgroup) __attribute__((amdgpu_waves_per_eu(4))) name(GridParams P) { \
    etc1_encode_one<comps, strategy>(P);                                                                              \
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

const char *etc1_kernel_name(int comps) { return comps == 4 ? "icamd_etc1_rgba8_kernel" : "icamd_etc1_rgb888_kernel"; }

hipError_t launch_etc1(int comps, const GridParams &P, hipStream_t stream) {
  // 16 x 16-block tiles (a wave = 16 x 4 blocks = 64 x 16 pixels) instead of 256 x 1: the encoder's wave-uniform
  // decisions (unclamped shortcut, codeword pruning) fire far more often on compact waves, and at 7 % of the HBM
  // roofline the narrower loads cost nothing: noise 1.47 = 1.47 ms, smooth 1.85 -> 1.61 ms, flat 1.90 -> 1.72 ms (r01)
  const uint32_t cap = 4u;
  typedef void (*Kernel)(GridParams);
  static const Kernel kernels[2][4] = {
    { icamd_etc1_rgb888_split_h_kernel, icamd_etc1_rgb888_split_v_kernel, icamd_etc1_rgb888_kernel, icamd_etc1_rgb888_heuristic_kernel },
    { icamd_etc1_rgba8_split_h_kernel, icamd_etc1_rgba8_split_v_kernel, icamd_etc1_rgba8_kernel, icamd_etc1_rgba8_heuristic_kernel } };
  const Kernel k = kernels[comps == 4 ? 1 : 0][P.etc_strategy < 4u ? P.etc_strategy : 2u];
  return launch_tiled(k, k, P, stream, cap);
}

}  // namespace icamd
