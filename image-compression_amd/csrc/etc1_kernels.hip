// etc1_kernels.hip -- ETC1 encode kernels for gfx950 (MI355X); see etc1_block.h for the math.
// Same one-block-per-lane raster mapping as dxt_kernels.hip.  VALU-bound (~3.8 k integer ops per
// block at kSmallerError); the 3.5 / 4.5 B/px of HBM traffic are a small fraction of the roofline.
#include "etc1_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

template <int COMPS>
__device__ __forceinline__ void etc1_encode_one(const GridParams &P, uint32_t k) {
  uint32_t img, brow, bcol;
  locate_block(P, k, img, brow, bcol);
  const uint8_t *src = P.src + (size_t)img * P.src_image_stride;
  uint32_t px[16];
  load_block<COMPS>(src, P.height, P.width, P.row_stride, brow * 4, bcol * 4, px);
  const Out8 c = encode_etc1_block(px, P.etc_strategy);
  store_stream8(P.dst + (size_t)img * P.dst_image_stride + (size_t)(k - img * P.blocks_per_image) * 8, c.lo, c.hi);
}

extern "C" {

__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_etc1_rgb888_kernel(GridParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) etc1_encode_one<3>(P, k);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_etc1_rgba8_kernel(GridParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) etc1_encode_one<4>(P, k);
}

}  // extern "C"

const char *etc1_kernel_name(int comps) { return comps == 4 ? "icamd_etc1_rgba8_kernel" : "icamd_etc1_rgb888_kernel"; }

hipError_t launch_etc1(int comps, const GridParams &P, hipStream_t stream) {
  if (P.total_blocks == 0) return hipSuccess;
  const dim3 grid((P.total_blocks + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  if (comps == 4)
    hipLaunchKernelGGL(icamd_etc1_rgba8_kernel, grid, block, 0, stream, P);
  else
    hipLaunchKernelGGL(icamd_etc1_rgb888_kernel, grid, block, 0, stream, P);
  return hipGetLastError();
}

}  // namespace icamd
