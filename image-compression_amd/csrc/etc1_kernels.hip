// etc1_kernels.hip -- ETC1 encode kernels for gfx950 (MI355X); see etc1_block.h for the math.
// Same one-block-per-lane tile mapping as dxt_kernels.hip, with 16 x 16-block tiles.  VALU-bound (3.5-4.3 k integer instructions per block at
// kSmallerError, 98 % issue utilisation); the 3.5 / 4.5 B/px of HBM traffic are a small fraction of the roofline.
#include "etc1_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

template <int COMPS, bool WIDE>
__device__ __forceinline__ void etc1_encode_one(const GridParams &P) {
  const TileCoord t = locate_tile<WIDE>(P);
  if (!t.valid) return;
  uint32_t px[16];
  load_tile_block<COMPS>(P, t, px);
  const Out8 c = encode_etc1_block(px, P.etc_strategy);
  store_stream8(tile_dst<8>(P, t), c.lo, c.hi);
}

extern "C" {

__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_etc1_rgb888_kernel(GridParams P) { etc1_encode_one<3, false>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_etc1_rgba8_kernel(GridParams P) { etc1_encode_one<4, false>(P); }

}  // extern "C"

const char *etc1_kernel_name(int comps) { return comps == 4 ? "icamd_etc1_rgba8_kernel" : "icamd_etc1_rgb888_kernel"; }

hipError_t launch_etc1(int comps, const GridParams &P, hipStream_t stream) {
  // 16 x 16-block tiles (a wave = 16 x 4 blocks = 64 x 16 pixels) instead of 256 x 1: the encoder's wave-uniform
  // decisions (unclamped shortcut, codeword pruning) fire far more often on compact waves, and at 7 % of the HBM
  // roofline the narrower loads cost nothing: noise 1.47 = 1.47 ms, smooth 1.85 -> 1.61 ms, flat 1.90 -> 1.72 ms (r01)
  const uint32_t cap = 4u;
  return comps == 4 ? launch_tiled(icamd_etc1_rgba8_kernel, icamd_etc1_rgba8_kernel, P, stream, cap)
                    : launch_tiled(icamd_etc1_rgb888_kernel, icamd_etc1_rgb888_kernel, P, stream, cap);
}

}  // namespace icamd
