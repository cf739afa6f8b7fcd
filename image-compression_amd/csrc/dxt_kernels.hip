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

// A/B variant (r03): two HORIZONTALLY adjacent blocks per lane, 512 x 1-block tiles -- a lane reads
// 24 contiguous bytes per pixel row (one 16-byte + one 8-byte load; a wave 1 536 contiguous bytes) and writes its two
// blocks with one 16-byte store.  Built only with -DICAMD_RGB888_HPAIR (profiles/r03_ab_rgb888_hpair.log has the result).
#if defined(ICAMD_RGB888_HPAIR)
__device__ __forceinline__ void dxt1_rgb888_hpair(const GridParams &P) {
  __shared__ uint32_t lds_px[4][kThreadsPerWorkgroup][4];
  BlockStash stash;
  stash.base = &lds_px[0][threadIdx.x][0];
  const bool swap = P.swap_rb != 0;
  const uint32_t brow = blockIdx.y + P.tile_row0, bcol0 = blockIdx.x * 512u, bcol = bcol0 + 2u * threadIdx.x;
  const uint32_t img = blockIdx.z;
  if (bcol >= P.block_cols) return;
  const bool interior = (bcol0 + 512u) * 4u <= P.width && (brow + 1u) * 4u <= P.height && !P.force_gather;
  uint32_t px[2][16];
  const uint8_t *image = P.src + (uint64_t)img * P.src_image_stride;
  if (interior) {
    const uint8_t *base = image + (uint64_t)(brow * 4u) * P.row_stride + (uint64_t)bcol0 * 12u;
    const uint32_t off = threadIdx.x * 24u;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const uint8_t *row = base + (uint32_t)(off + (uint32_t)y * P.row_stride);
      const U4 a = load_stream(reinterpret_cast<const U4 *>(row));
      const U2 b = *reinterpret_cast<const U2 *>(row + 16);
      px[0][4 * y + 0] = a.x;
      px[0][4 * y + 1] = alignbit(a.y, a.x, 24);
      px[0][4 * y + 2] = alignbit(a.z, a.y, 16);
      px[0][4 * y + 3] = a.z >> 8;
      px[1][4 * y + 0] = a.w;
      px[1][4 * y + 1] = alignbit(b.x, a.w, 24);
      px[1][4 * y + 2] = alignbit(b.y, b.x, 16);
      px[1][4 * y + 3] = b.y >> 8;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (bcol + h < P.block_cols)
        load_block<3>(image, P.height, P.width, P.row_stride, brow * 4u, (bcol + h) * 4u, px[h], !P.force_gather);
  }
  uint8_t *dst = P.dst + (uint64_t)img * P.dst_image_stride + ((uint64_t)brow * P.block_cols + bcol) * 8u;
  const Out8 c0 = encode_dxt_color_block(px[0], swap, false, stash);
  if (bcol + 1u < P.block_cols) {
    const Out8 c1 = encode_dxt_color_block(px[1], swap, false, stash);
    store_stream16(dst, c0.lo, c0.hi, c1.lo, c1.hi);
  } else {
    store_stream8(dst, c0.lo, c0.hi);
  }
}
extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgb888_hpair_kernel(GridParams P) { dxt1_rgb888_hpair(P); }
#endif

extern "C" {

// 3-byte sources only (r02 A/B, 16 x 4096^2, ms per launch, one block vs two blocks per lane): RGB888 0.183 -> 0.175
// (noise), 0.190 -> 0.183 (flat), 0.174 -> 0.172 (smooth); RGBA8 0.190 -> 0.201 and DXT5 0.253 -> 0.259 got slower
// (67-71 VGPRs instead of 51-54) and keep one block per lane.
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgb888_x2_kernel(GridParams P) { dxt_encode_two<3, false>(P); }

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
  if (comps == 4) return launch_tiled(icamd_dxt1_rgba8_kernel, icamd_dxt1_rgba8_narrow_kernel, P, stream);
#if defined(ICAMD_RGB888_HPAIR)
  if (P.block_cols > 256 && (uint64_t)P.row_stride * 4u < (1ull << 31) && P.block_rows <= 65535u && P.n_images <= 65535u) {
    GridParams Q = P;
    Q.log2_tile_cols = 8; Q.tile_row0 = 0;
    Q.force_gather = (uint64_t)P.row_stride * 3u + 16384u >= (1ull << 32) ? 1u : 0u;
    hipLaunchKernelGGL(icamd_dxt1_rgb888_hpair_kernel, dim3((P.block_cols + 511u) / 512u, P.block_rows, P.n_images),
                       dim3(kThreadsPerWorkgroup), 0, stream, Q);
    return hipGetLastError();
  }
#endif
  return launch_tiled(icamd_dxt1_rgb888_x2_kernel, icamd_dxt1_rgb888_narrow_kernel, P, stream, 8, 2);
}

}  // namespace icamd
