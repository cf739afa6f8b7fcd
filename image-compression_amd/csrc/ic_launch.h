// ic_launch.h -- host-side launch entry points of the kernel translation units
// (dxt_kernels.hip, etc1_kernels.hip, pvrtc_kernels.hip, decode_kernels.hip); called by ic_capi.hip.
#ifndef ICAMD_IC_LAUNCH_H_
#define ICAMD_IC_LAUNCH_H_

#include <hip/hip_runtime.h>

#include "ic_device.h"

namespace icamd {

constexpr int kThreadsPerWorkgroup = 256;  // 4 waves; one 4x4 block per lane

// codec: ICAMD_DXT1 / ICAMD_DXT5; comps: source bytes per pixel (3 or 4; DXT5 requires 4).
hipError_t launch_dxt(int codec, int comps, const GridParams &P, hipStream_t stream);
hipError_t launch_etc1(int comps, const GridParams &P, hipStream_t stream);

// PVRTC1 2bpp: square power-of-two RGBA8 images, n_images of them.
struct PvrtcParams {
  const uint8_t *src;
  uint8_t *dst;
  uint64_t src_image_stride, dst_image_stride;
  uint32_t size;      // width == height
  uint32_t log2_size;
  uint32_t n_images;
};
hipError_t launch_pvrtc2(const PvrtcParams &P, hipStream_t stream);

struct DecodeParams {
  const uint8_t *blocks;
  uint8_t *pixels;
  uint64_t src_image_stride, dst_image_stride;
  uint32_t height, width, block_rows, block_cols, row_stride;
  uint32_t blocks_per_image, total_blocks, swap_rb;
  FastDiv div_bpi, div_cols;
};
hipError_t launch_decode(int codec, const DecodeParams &P, hipStream_t stream);

// Compressed-domain operations on one image's block grid.
struct BlockOpParams {
  const uint8_t *src;
  uint8_t *dst;
  uint32_t in_rows, in_cols;    // source block grid
  uint32_t out_rows, out_cols;  // result block grid
  uint32_t total_out;
  uint32_t etc_strategy;
  uint32_t src_height, src_width;  // uncompressed pixels of the source (Downsample's single-block case)
  FastDiv div_out_cols;
};
hipError_t launch_pad(int codec, const BlockOpParams &P, hipStream_t stream);
hipError_t launch_downsample(int codec, const BlockOpParams &P, hipStream_t stream);
hipError_t launch_transcode_dxt1_to_etc1(void *blocks, uint32_t n_blocks, hipStream_t stream);

const char *dxt_kernel_name(int codec, int comps);
const char *etc1_kernel_name(int comps);
const char *pvrtc2_kernel_name();

}  // namespace icamd
#endif  // ICAMD_IC_LAUNCH_H_
