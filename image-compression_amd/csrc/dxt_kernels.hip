// dxt_kernels.hip -- DXT1 / DXT5 encode kernels for gfx950 (MI355X).
//
// One 4x4 block per lane, 256 lanes per workgroup, one workgroup per tile of consecutive blocks of a
// block row (locate_tile, ic_device.h): consecutive lanes read consecutive 16-byte (RGBA8) or 12-byte
// (RGB888) row segments -- a wave's four row loads are 1 KiB / 768 B contiguous each -- and write
// consecutive 8/16-byte blocks (reference raster order, internal/compressor4x4_helper.h:202-214).
// 4.5 / 3.5 / 5 algorithmic bytes per pixel (DXT1 from RGBA8 / RGB888, DXT5) and 255 / 276 / 614 integer VALU
// instructions per block (r01): DXT1 from RGBA8 streams at the practical HBM rate, the other two are bound by
// instruction issue (see dxt_block.h).
#include "dxt_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

template <int COMPS, bool DXT5, bool WIDE>
__device__ __forceinline__ void dxt_encode_one(const GridParams &P) {
  const TileCoord t = locate_tile<WIDE>(P);
  if (!t.valid) return;
  uint32_t px[16];
  load_tile_block<COMPS>(P, t, px);
  const bool swap = P.swap_rb != 0;
  __shared__ uint32_t lds_px[4][kThreadsPerWorkgroup][4];
  BlockStash stash;
  stash.base = &lds_px[0][threadIdx.x][0];
  if (DXT5) {
    // has_one_pixel: block entirely right of AND below the image (pixel4x4.cc:58)
    const bool one_pixel = (t.bcol * 4 >= P.width) && (t.brow * 4 >= P.height);
    const Out8 a = encode_dxt5_alpha_block(px, one_pixel);
    const Out8 c = encode_dxt_color_block(px, swap, true, stash);
    // alpha block then colour block, dxtc.cc:94-96
    store_stream16(tile_dst<16>(P, t), a.lo, a.hi, c.lo, c.hi);
  } else {
    const Out8 c = encode_dxt_color_block(px, swap, false, stash);
    store_stream8(tile_dst<8>(P, t), c.lo, c.hi);
  }
}

// Two vertically adjacent blocks per lane (256 x 2-block tiles): all eight row loads are issued before the first
// block is encoded (twice the bytes in flight per wave: the 12-byte-per-lane loads of a 3-byte source keep a quarter
// less in flight than RGBA8's), and the tile prologue (coordinates, 64-bit bases) is paid once for two blocks.
template <int COMPS, bool DXT5>
__device__ __forceinline__ void dxt_encode_two(const GridParams &P) {
  const TileCoord t = locate_tile<true, 2>(P);
  __shared__ uint32_t lds_px[4][kThreadsPerWorkgroup][4];
  BlockStash stash;
  stash.base = &lds_px[0][threadIdx.x][0];
  const bool swap = P.swap_rb != 0;
  TileCoord tr[2];
  uint32_t px[2][16];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    tr[r] = t;
    tr[r].brow0 = t.brow0 + (uint32_t)r;
    tr[r].brow = tr[r].brow0;
    tr[r].valid = t.full || (t.bcol < P.block_cols && tr[r].brow < P.block_rows);
    if (tr[r].valid) load_tile_block<COMPS>(P, tr[r], px[r]);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (!tr[r].valid) continue;
    if (DXT5) {
      const bool one_pixel = (tr[r].bcol * 4 >= P.width) && (tr[r].brow * 4 >= P.height);
      const Out8 a = encode_dxt5_alpha_block(px[r], one_pixel);
      const Out8 c = encode_dxt_color_block(px[r], swap, true, stash);
      store_stream16(tile_dst<16>(P, tr[r]), a.lo, a.hi, c.lo, c.hi);
    } else {
      const Out8 c = encode_dxt_color_block(px[r], swap, false, stash);
      store_stream8(tile_dst<8>(P, tr[r]), c.lo, c.hi);
    }
  }
}

extern "C" {

// (r03: two HORIZONTALLY adjacent blocks per lane -- 24 contiguous bytes per row as a 16-byte + an 8-byte load, one
// 16-byte store, 512 x 1-block tiles -- measured 10 % slower than this kernel on every content,
// profiles/r03_ab_rgb888_hpair.log; the variant is in the history at commit "DXT colour index search in O(1)".)
// 3-byte sources only (r02 A/B, 16 x 4096^2, ms per launch, one block vs two blocks per lane): RGB888 0.183 -> 0.175
// (noise), 0.190 -> 0.183 (flat), 0.174 -> 0.172 (smooth); RGBA8 0.190 -> 0.201 and DXT5 0.253 -> 0.259 got slower
// (67-71 VGPRs instead of 51-54) and keep one block per lane.
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgb888_x2_kernel(GridParams P) { dxt_encode_two<3, false>(P); }
#if defined(ICAMD_DXT1_RGBA8_X2_SMALL)
// A/B only (r04, VERDICT r03 item 6: "fewer, fatter workgroups for <= 64 MiB launches so that a launch is one residency round"):
// one 4096^2 image per call 16.0 -> 17.1 us with an event pair, 13.9 -> 15.2 us back to back -- slower, not shipped
// (profiles/r04_single_image_timeline.txt, section c)
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgba8_x2_kernel(GridParams P) { dxt_encode_two<4, false>(P); }
#endif

// *_kernel: 256 x 1-block tiles (block grids more than 128 columns wide); *_narrow_kernel: any tile shape
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgba8_kernel(GridParams P) { dxt_encode_one<4, false, true>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt5_rgba8_kernel(GridParams P) { dxt_encode_one<4, true, true>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgba8_narrow_kernel(GridParams P) { dxt_encode_one<4, false, false>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgb888_narrow_kernel(GridParams P) { dxt_encode_one<3, false, false>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt5_rgba8_narrow_kernel(GridParams P) { dxt_encode_one<4, true, false>(P); }

}  // extern "C"

const char *dxt_kernel_name(int codec, int comps) {
  if (codec == ICAMD_DXT5) return "icamd_dxt5_rgba8_kernel";
  return comps == 4 ? "icamd_dxt1_rgba8_kernel" : "icamd_dxt1_rgb888_x2_kernel";
}

hipError_t launch_dxt(int codec, int comps, const GridParams &P, hipStream_t stream) {
  if (codec == ICAMD_DXT5) {
    if (comps != 4) return hipErrorInvalidValue;
    return launch_tiled(icamd_dxt5_rgba8_kernel, icamd_dxt5_rgba8_narrow_kernel, P, stream);
  }
#if defined(ICAMD_DXT1_RGBA8_X2_SMALL)
  if (comps == 4 && (uint64_t)P.n_images * P.block_rows * P.block_cols <= (1ull << 20))  // one 4096^2 image or less
    return launch_tiled(icamd_dxt1_rgba8_x2_kernel, icamd_dxt1_rgba8_narrow_kernel, P, stream, 8, 2);
#endif
  if (comps == 4) return launch_tiled(icamd_dxt1_rgba8_kernel, icamd_dxt1_rgba8_narrow_kernel, P, stream);
  return launch_tiled(icamd_dxt1_rgb888_x2_kernel, icamd_dxt1_rgb888_narrow_kernel, P, stream, 8, 2);
}

}  // namespace icamd
