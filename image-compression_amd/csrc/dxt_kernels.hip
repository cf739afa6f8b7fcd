// dxt_kernels.hip -- DXT1 / DXT5 encode kernels for gfx950 (MI355X).
//
// One 4x4 block per lane, 256 lanes per workgroup.  Block k of the launch is block
// (k / cols, k % cols) of image k / blocks_per_image, so consecutive lanes read
// consecutive 16-byte (RGBA8) or 12-byte (RGB888) row segments -- a wave's four row
// loads are 1 KiB / 768 B contiguous each -- and write consecutive 8/16-byte blocks
// (reference raster order, internal/compressor4x4_helper.h:202-214).
// HBM-bound by design: 4.5 / 3.5 / 5 algorithmic bytes per pixel (DXT1 from RGBA8 /
// RGB888, DXT5), ~300-650 integer VALU ops per block (see dxt_block.h).
#include "dxt_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

template <int COMPS, bool DXT5>
__device__ __forceinline__ void dxt_encode_one(const GridParams &P, uint32_t k) {
  uint32_t img, brow, bcol;
  locate_block(P, k, img, brow, bcol);
  const uint8_t *src = P.src + (size_t)img * P.src_image_stride;
  uint32_t px[16];
  load_block<COMPS>(src, P.height, P.width, P.row_stride, brow * 4, bcol * 4, px);
  const bool swap = P.swap_rb != 0;
  __shared__ uint32_t lds_px[4][kThreadsPerWorkgroup][4];
  BlockStash stash;
  stash.base = &lds_px[0][threadIdx.x][0];
  if (DXT5) {
    // has_one_pixel: block entirely right of AND below the image (pixel4x4.cc:58)
    const bool one_pixel = (bcol * 4 >= P.width) && (brow * 4 >= P.height);
    const Out8 a = encode_dxt5_alpha_block(px, one_pixel);
    const Out8 c = encode_dxt_color_block(px, swap, true, stash);
    // alpha block then colour block, dxtc.cc:94-96
    store_stream16(P.dst + (size_t)img * P.dst_image_stride + (size_t)(k - img * P.blocks_per_image) * 16, a.lo, a.hi,
                   c.lo, c.hi);
  } else {
    const Out8 c = encode_dxt_color_block(px, swap, false, stash);
    store_stream8(P.dst + (size_t)img * P.dst_image_stride + (size_t)(k - img * P.blocks_per_image) * 8, c.lo, c.hi);
  }
}

extern "C" {

__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgba8_kernel(GridParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) dxt_encode_one<4, false>(P, k);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_rgb888_kernel(GridParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) dxt_encode_one<3, false>(P, k);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt5_rgba8_kernel(GridParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) dxt_encode_one<4, true>(P, k);
}

}  // extern "C"

const char *dxt_kernel_name(int codec, int comps) {
  if (codec == ICAMD_DXT5) return "icamd_dxt5_rgba8_kernel";
  return comps == 4 ? "icamd_dxt1_rgba8_kernel" : "icamd_dxt1_rgb888_kernel";
}

hipError_t launch_dxt(int codec, int comps, const GridParams &P, hipStream_t stream) {
  if (P.total_blocks == 0) return hipSuccess;
  const dim3 grid((P.total_blocks + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  if (codec == ICAMD_DXT5) {
    if (comps != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(icamd_dxt5_rgba8_kernel, grid, block, 0, stream, P);
  } else if (comps == 4) {
    hipLaunchKernelGGL(icamd_dxt1_rgba8_kernel, grid, block, 0, stream, P);
  } else {
    hipLaunchKernelGGL(icamd_dxt1_rgb888_kernel, grid, block, 0, stream, P);
  }
  return hipGetLastError();
}

}  // namespace icamd
