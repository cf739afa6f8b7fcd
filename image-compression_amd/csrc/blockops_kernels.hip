// blockops_kernels.hip -- Pad / Downsample / DXT1->ETC1 transcode kernels: one output
// block per lane, coalesced 8/16-byte block loads and stores.  See blockops_block.h for the per-block math.
#include <cstdlib>
#include "blockops_block.h"
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

namespace {
constexpr int kWords(int codec) { return codec == ICAMD_DXT5 ? 4 : 2; }
}

// (r, c) of output block k: the source block it copies / replicates, and whether it is a pad block.
template <int CODEC, int STRATEGY>
__device__ __forceinline__ void pad_block(const BlockOpParams &P, uint32_t img, uint32_t r, uint32_t c, bool copy_interior,
                                          bool make_border) {
  constexpr int W = kWords(CODEC);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(P.src + (size_t)img * P.src_image_stride);
  uint32_t *dst = reinterpret_cast<uint32_t *>(P.dst + (size_t)img * P.dst_image_stride) + ((size_t)r * P.out_cols + c) * W;
  const bool in_rows = r < P.in_rows, in_cols = c < P.in_cols;
  if (in_rows && in_cols ? !copy_interior : !make_border) return;
  const uint32_t sr = in_rows ? r : P.in_rows - 1, sc = in_cols ? c : P.in_cols - 1;
  const uint32_t *s = src + ((size_t)sr * P.in_cols + sc) * W;
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = s[i];
  if (in_rows && in_cols) {
#pragma unroll
    for (int i = 0; i < W; ++i) dst[i] = w[i];
    return;
  }
  const int kind = in_rows ? kPadColumn : (in_cols ? kPadRow : kPadCorner);  // helper.h:427-470
  if (CODEC == ICAMD_ETC1) {
    const Out8 o = etc1_pad_block(w[0], w[1], kind, (uint32_t)STRATEGY);  // compile-time strategy: see downsample_one
    dst[0] = o.lo; dst[1] = o.hi;
  } else if (CODEC == ICAMD_DXT1) {
    dst[0] = w[0];
    dst[1] = dxt_pad_color_bits(w[1], kind);
  } else {
    uint32_t lo24 = w[0] >> 16 | (w[1] & 0xffu) << 16, hi24 = w[1] >> 8;
    dxt5_pad_alpha_codes(lo24, hi24, kind);
    dst[0] = (w[0] & 0xffffu) | lo24 << 16;
    dst[1] = lo24 >> 16 | hi24 << 8;
    dst[2] = w[2];
    dst[3] = dxt_pad_color_bits(w[3], kind);
  }
}

// DXT: one pass over the output grid (a pad block is a few bit edits).  ETC1 (r04): a pad block is a decode + re-encode -- up to
// the whole kSmallerError search -- so the copy of the image's own blocks runs as one light kernel over the grid (PART 1: 7 VGPRs,
// 8 waves per SIMD) and the pad blocks as a second, small launch over the BORDER only (PART 2: the in_rows x extra columns to
// the right, then the extra rows over the full width); in one kernel the searches' 121 VGPRs capped the copy at 4 waves per SIMD
// and every wave that touched the border ran the search (28.6 -> ~7 us per 4096^2 image padded by 8 pixels).
// Batched launches (icamd_pad_batch_device, r05): n_images equally shaped grids, out_per_image work items each (output blocks;
// PART 2: border blocks).
template <int CODEC, int STRATEGY, int PART>
__device__ __forceinline__ void pad_one(const BlockOpParams &P, uint32_t k) {
  uint32_t img = 0;
  if (P.n_images > 1) {
    if (PART == 3) {  // work items are quad lanes of pad blocks
      img = fastdiv(k, P.div_border_lanes_per_image);
      k -= img * P.border_lanes_per_image;
    } else {
      img = fastdiv(k, P.div_out_per_image);
      k -= img * P.out_per_image;
    }
  }
  if (PART == 3) {  // PART 2 with four lanes per pad block (kSmallerError): work item k = 4 * border block + quad lane
    const uint32_t t = k & 3u;
    k >>= 2;
    const uint32_t dc = P.out_cols - P.in_cols, right = P.in_rows * dc;
    uint32_t r, c;
    if (k < right) { r = k / dc; c = P.in_cols + (k - r * dc); }
    else { const uint32_t j = k - right; r = P.in_rows + j / P.out_cols; c = j - (r - P.in_rows) * P.out_cols; }
    const bool in_rows = r < P.in_rows, in_cols = c < P.in_cols;
    const uint32_t sr = in_rows ? r : P.in_rows - 1, sc = in_cols ? c : P.in_cols - 1;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(P.src + (size_t)img * P.src_image_stride) + ((size_t)sr * P.in_cols + sc) * 2;
    uint32_t *dst = reinterpret_cast<uint32_t *>(P.dst + (size_t)img * P.dst_image_stride) + ((size_t)r * P.out_cols + c) * 2;
    bool writes = false;
    const Out8 o = etc1_pad_block_quad(s[0], s[1], in_rows ? kPadColumn : (in_cols ? kPadRow : kPadCorner), t, &writes);
    if (writes) { dst[0] = o.lo; dst[1] = o.hi; }
    return;
  }
  if (PART == 2) {
    const uint32_t dc = P.out_cols - P.in_cols, right = P.in_rows * dc;
    uint32_t r, c;
    if (k < right) { r = k / dc; c = P.in_cols + (k - r * dc); }
    else { const uint32_t j = k - right; r = P.in_rows + j / P.out_cols; c = j - (r - P.in_rows) * P.out_cols; }
    pad_block<CODEC, STRATEGY>(P, img, r, c, false, true);
    return;
  }
  const uint32_t r = fastdiv(k, P.div_out_cols), c = k - r * P.out_cols;
  pad_block<CODEC, STRATEGY>(P, img, r, c, true, PART == 0);
}

// STRATEGY: the ETC1 re-encode strategy as a compile-time constant (one kernel per strategy, like the encoders: the
// kSmallerError search is not carried along by a kHeuristic downsample and vice versa); ignored for DXT.
// QUAD (ETC1 kSmallerError, small grids -- the low levels of a mip chain): work item k = 4 * output block + quad lane; the four
// lanes build the same sixteen pixels and split the re-encode's searches (etc1_block.h encode_etc1_block_quad).
// (img, r, c): image of a batched launch and the output block's position in its grid; kk = r * out_cols + c
template <int CODEC, int STRATEGY, bool QUAD = false>
__device__ __forceinline__ void downsample_at(const BlockOpParams &P, uint32_t img, uint32_t r, uint32_t c, uint32_t kk,
                                              uint32_t quad_lane, BlockStash &stash) {
  constexpr int W = kWords(CODEC);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(P.src + (size_t)img * P.src_image_stride);
  uint32_t px[16], tmp[16];
  if (P.in_rows > 1 && P.in_cols > 1) {  // DownsampleBlocks2x2
    if (CODEC != ICAMD_ETC1) {
      // the two blocks of a source row are adjacent in memory: one 16-byte (DXT1) / two 16-byte (DXT5) loads per row
      uint32_t w[2][2][W];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t *q = src + ((size_t)(2 * r + i) * P.in_cols + 2 * c) * W;
#pragma unroll
        for (int h = 0; h < 2 * W / 4; ++h) {
          const U4 v = *reinterpret_cast<const U4 *>(q + 4 * h);
          uint32_t *d = &w[0][0][0] + (i * 2 * W + 4 * h);
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
      const uint32_t *const s4[2][2] = { { w[0][0], w[0][1] }, { w[1][0], w[1][1] } };
      dxt_downsample_2x2<CODEC>(s4, px);
    } else {
      uint32_t w[2][2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const U4 v = *reinterpret_cast<const U4 *>(src + ((size_t)(2 * r + i) * P.in_cols + 2 * c) * 2);
        w[i][0][0] = v.x; w[i][0][1] = v.y; w[i][1][0] = v.z; w[i][1][1] = v.w;
      }
      const uint32_t *const s4[2][2] = { { w[0][0], w[0][1] }, { w[1][0], w[1][1] } };
      etc1_downsample_2x2(s4, px);
    }
  } else if (P.in_rows > 1) {  // DownsampleBlocks2x1: one block column
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      decode_any<CODEC>(src + (size_t)(2 * r + i) * W, tmp);
      store_downsampled(tmp, 2 * i, 0, px);
      store_downsampled(tmp, 2 * i, 2, px);
    }
  } else if (P.in_cols > 1) {  // DownsampleBlocks1x2: one block row
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      decode_any<CODEC>(src + (size_t)(2 * c + j) * W, tmp);
      store_downsampled(tmp, 0, 2 * j, px);
      store_downsampled(tmp, 2, 2 * j, px);
    }
  } else {  // a single block of 4, 2 or 1 pixels per side: replicate to 4x4 first (helper.h:338-387)
    decode_any<CODEC>(src, tmp);
    if (P.src_width == 1) {
#pragma unroll
      for (int y = 0; y < 4; ++y) tmp[4 * y + 1] = tmp[4 * y + 2] = tmp[4 * y + 3] = tmp[4 * y];
    } else if (P.src_width == 2) {
#pragma unroll
      for (int y = 0; y < 4; ++y) { tmp[4 * y + 2] = tmp[4 * y]; tmp[4 * y + 3] = tmp[4 * y + 1]; }
    }
    if (P.src_height == 1) {
#pragma unroll
      for (int x = 0; x < 4; ++x) tmp[4 + x] = tmp[8 + x] = tmp[12 + x] = tmp[x];
    } else if (P.src_height == 2) {
#pragma unroll
      for (int x = 0; x < 4; ++x) { tmp[8 + x] = tmp[x]; tmp[12 + x] = tmp[4 + x]; }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) store_downsampled(tmp, 2 * i, 2 * j, px);
  }
  uint8_t *dst = P.dst + (size_t)img * P.dst_image_stride + (size_t)kk * (W * 4);
  if (QUAD) {
    bool writes = false;
    const Out8 o = encode_etc1_block_quad(px, quad_lane, &writes);
    if (writes) store_stream8(dst, o.lo, o.hi);
    return;
  }
  uint32_t out[4];
  encode_any<CODEC>(px, CODEC == ICAMD_ETC1 ? (uint32_t)STRATEGY : 0u, stash, out);
  if (W == 4) store_stream16(dst, out[0], out[1], out[2], out[3]);
  else store_stream8(dst, out[0], out[1]);
}
// work item k of a linear launch -> (image, row, column) by two multiply-high divisions
template <int CODEC, int STRATEGY, bool QUAD = false>
__device__ __forceinline__ void downsample_one(const BlockOpParams &P, uint32_t k, BlockStash &stash) {
  const uint32_t quad_lane = QUAD ? (k & 3u) : 0u;
  if (QUAD) k >>= 2;
  uint32_t img = 0, kk = k;
  if (P.n_images > 1) {
    img = fastdiv(k, P.div_out_per_image);
    kk = k - img * P.out_per_image;
  }
  const uint32_t r = fastdiv(kk, P.div_out_cols), c = kk - r * P.out_cols;
  downsample_at<CODEC, STRATEGY, QUAD>(P, img, r, c, kk, quad_lane, stash);
}

// Lanes per workgroup: the kernels that run an ETC1 codeword SEARCH per output block (Downsample and the Pad border with
// kSplitHorizontally / kSplitVertically / kSmallerError) are launched as one-wave workgroups like the encoders (r05,
// etc1_kernels.hip ICAMD_ETC1_WAVE_WORKGROUPS: the search's cost depends on the content, and a four-wave workgroup holds its
// slots until its slowest wave is done); everything else keeps 256.
constexpr int blockop_lanes(int codec, int strategy, int part) {
  return (codec == ICAMD_ETC1 && strategy != 3 && part != 1) ? 64 : kThreadsPerWorkgroup;
}
#define ICAMD_PAD_KERNEL(NAME, CODEC, STRATEGY, PART)                                                          \
  extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pad_##NAME##_kernel(BlockOpParams P) { \
    const uint32_t k = blockIdx.x * (uint32_t)blockop_lanes(CODEC, STRATEGY, PART) + threadIdx.x;              \
    if (k < P.total_out) pad_one<CODEC, STRATEGY, PART>(P, k);                                                 \
  }
#define ICAMD_DOWNSAMPLE_KERNEL(NAME, CODEC, STRATEGY)                                                         \
  extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup)                                           \
  icamd_downsample_##NAME##_kernel(BlockOpParams P) {                                                          \
    /* the per-lane pixel stash is only used by the DXT colour encoder */                                      \
    __shared__ uint32_t lds_px[CODEC == ICAMD_ETC1 ? 1 : 4][CODEC == ICAMD_ETC1 ? 1 : kThreadsPerWorkgroup][4]; \
    BlockStash stash;                                                                                          \
    stash.base = CODEC == ICAMD_ETC1 ? &lds_px[0][0][0] : &lds_px[0][threadIdx.x][0];                          \
    const uint32_t k = blockIdx.x * (uint32_t)blockop_lanes(CODEC, STRATEGY, 0) + threadIdx.x;                 \
    if (k < P.total_out) downsample_one<CODEC, STRATEGY>(P, k, stash);                                         \
  }

// ROW TILES (r06; grids of at least 256 output columns, DXT and ETC1 kHeuristic): blockIdx = (column tile, output row, image),
// so image and row are workgroup-uniform -- the two multiply-high divisions, their quarter-rate multiplies back and the 64-bit
// row addressing move to the scalar unit, a lane adds a 32-bit column offset (the encoders' tile form).
#define ICAMD_DOWNSAMPLE_ROWS_KERNEL(NAME, CODEC, STRATEGY)                                                    \
  extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup)                                           \
  icamd_downsample_##NAME##_rows_kernel(BlockOpParams P) {                                                     \
    __shared__ uint32_t lds_px[CODEC == ICAMD_ETC1 ? 1 : 4][CODEC == ICAMD_ETC1 ? 1 : kThreadsPerWorkgroup][4]; \
    BlockStash stash;                                                                                          \
    stash.base = CODEC == ICAMD_ETC1 ? &lds_px[0][0][0] : &lds_px[0][threadIdx.x][0];                          \
    const uint32_t c = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x, r = blockIdx.y, img = blockIdx.z;      \
    if (c < P.out_cols) downsample_at<CODEC, STRATEGY>(P, img, r, c, r * P.out_cols + c, 0u, stash);           \
  }
ICAMD_PAD_KERNEL(dxt1, ICAMD_DXT1, 0, 0)
ICAMD_PAD_KERNEL(dxt5, ICAMD_DXT5, 0, 0)
ICAMD_PAD_KERNEL(etc1_copy, ICAMD_ETC1, 0, 1)
ICAMD_PAD_KERNEL(etc1_border_split_h, ICAMD_ETC1, 0, 2)
ICAMD_PAD_KERNEL(etc1_border_split_v, ICAMD_ETC1, 1, 2)
ICAMD_PAD_KERNEL(etc1_border, ICAMD_ETC1, 2, 2)
ICAMD_PAD_KERNEL(etc1_border_heuristic, ICAMD_ETC1, 3, 2)
// kSmallerError (r05): four lanes per pad block, and -- the split search needs 61 VGPRs where the whole one needed 121, so the
// copy no longer loses occupancy to it -- in the SAME launch as the copy of the image's own blocks: the first border_wgs
// workgroups are the pad blocks (they start first and are the long ones), the others copy.  One launch per Pad call.
extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pad_etc1_quad_kernel(BlockOpParams P) {
  if (blockIdx.x < P.border_wgs) {
    const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
    if (k < P.border_lanes) pad_one<ICAMD_ETC1, 2, 3>(P, k);
  } else {
    const uint32_t k = (blockIdx.x - P.border_wgs) * kThreadsPerWorkgroup + threadIdx.x;
    if (k < P.total_out) pad_one<ICAMD_ETC1, 2, 1>(P, k);
  }
}
ICAMD_DOWNSAMPLE_KERNEL(dxt1, ICAMD_DXT1, 0)
ICAMD_DOWNSAMPLE_KERNEL(dxt5, ICAMD_DXT5, 0)
ICAMD_DOWNSAMPLE_KERNEL(etc1_split_h, ICAMD_ETC1, 0)
ICAMD_DOWNSAMPLE_KERNEL(etc1_split_v, ICAMD_ETC1, 1)
ICAMD_DOWNSAMPLE_KERNEL(etc1, ICAMD_ETC1, 2)  // kSmallerError (and every value the reference's default: label maps to it)
ICAMD_DOWNSAMPLE_KERNEL(etc1_heuristic, ICAMD_ETC1, 3)
ICAMD_DOWNSAMPLE_ROWS_KERNEL(dxt1, ICAMD_DXT1, 0)
ICAMD_DOWNSAMPLE_ROWS_KERNEL(dxt5, ICAMD_DXT5, 0)
ICAMD_DOWNSAMPLE_ROWS_KERNEL(etc1_heuristic, ICAMD_ETC1, 3)
// kSmallerError on grids of at most kDownsampleQuadMaxBlocks output blocks (r05): four lanes per output block.  A 512^2 level is
// 4 096 output blocks = 64 waves of ~3 000 dependent instructions on 64 of 1 024 SIMDs; the quad form makes it 256 waves of ~1 500.
constexpr uint32_t kDownsampleQuadMaxBlocks = 36864;
extern "C" __global__ void __launch_bounds__(64) icamd_downsample_etc1_quad_kernel(BlockOpParams P) {
  BlockStash stash;
  stash.base = nullptr;  // (only the DXT colour encoder parks pixels)
  const uint32_t k = blockIdx.x * 64u + threadIdx.x;
  if ((k >> 2) < P.total_out) downsample_one<ICAMD_ETC1, 2, true>(P, k, stash);
}

extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_to_etc1_kernel(uint2 *blocks, uint32_t n) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k >= n) return;
  const uint2 b = blocks[k];
  // DecodeDxt1Block(block, swap = false) + EncodeEtc1Block(..., kHeuristic), without materialising the pixels
  const Out8 o = transcode_dxt1_block_to_etc1(b.x, b.y);
  blocks[k] = make_uint2(o.lo, o.hi);
}

// A/B switch (ICAMD_PAD_BORDER_QUAD=0: one lane per pad block, the r04 form); read once
static bool pad_border_quad() {
  static const bool on = [] { const char *e = getenv("ICAMD_PAD_BORDER_QUAD"); return !(e && e[0] == '0'); }();
  return on;
}

hipError_t launch_pad(int codec, const BlockOpParams &P, hipStream_t stream) {
  if (P.total_out == 0) return hipSuccess;
  const dim3 grid((P.total_out + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  if (codec == ICAMD_DXT1) hipLaunchKernelGGL(icamd_pad_dxt1_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_DXT5) hipLaunchKernelGGL(icamd_pad_dxt5_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_ETC1) {
    const uint64_t border = (uint64_t)P.in_rows * (P.out_cols - P.in_cols) + (uint64_t)(P.out_rows - P.in_rows) * P.out_cols;
    if (border && P.etc_strategy != 0u && P.etc_strategy != 1u && P.etc_strategy != 3u && pad_border_quad() &&
        border * 4u * P.n_images < (1ull << 31)) {
      BlockOpParams Q = P;
      Q.border_lanes_per_image = (uint32_t)border * 4u;
      Q.div_border_lanes_per_image = make_fastdiv(Q.border_lanes_per_image);
      Q.border_lanes = Q.border_lanes_per_image * P.n_images;
      Q.border_wgs = (Q.border_lanes + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup;
      hipLaunchKernelGGL(icamd_pad_etc1_quad_kernel, dim3(grid.x + Q.border_wgs), block, 0, stream, Q);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(icamd_pad_etc1_copy_kernel, grid, block, 0, stream, P);
    BlockOpParams B = P;  // the pad blocks only: right of the image, then below it
    B.out_per_image = (uint32_t)border;
    B.div_out_per_image = make_fastdiv(border ? (uint32_t)border : 1u);
    B.total_out = (uint32_t)(border * P.n_images);
    if (border) {
      const uint32_t lanes = (uint32_t)blockop_lanes(ICAMD_ETC1, P.etc_strategy == 3u ? 3 : 2, 2);
      const dim3 bgrid((B.total_out + lanes - 1) / lanes), bblock(lanes);
      if (P.etc_strategy == 0u) hipLaunchKernelGGL(icamd_pad_etc1_border_split_h_kernel, bgrid, bblock, 0, stream, B);
      else if (P.etc_strategy == 1u) hipLaunchKernelGGL(icamd_pad_etc1_border_split_v_kernel, bgrid, bblock, 0, stream, B);
      else if (P.etc_strategy == 3u) hipLaunchKernelGGL(icamd_pad_etc1_border_heuristic_kernel, bgrid, bblock, 0, stream, B);
      else hipLaunchKernelGGL(icamd_pad_etc1_border_kernel, bgrid, bblock, 0, stream, B);
    }
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_downsample(int codec, const BlockOpParams &P, hipStream_t stream) {
  if (P.total_out == 0) return hipSuccess;
  const dim3 grid((P.total_out + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  // row tiles where a row fills a workgroup (ICAMD_DOWNSAMPLE_ROW_TILES=0 keeps the linear launch for the A/B)
  static const bool row_tiles_on = [] { const char *e = getenv("ICAMD_DOWNSAMPLE_ROW_TILES"); return !(e && e[0] == '0'); }();
  const bool rows = row_tiles_on && P.in_rows > 1 && P.in_cols > 1 && P.out_cols >= (uint32_t)kThreadsPerWorkgroup &&
                    P.out_rows <= 65535u && P.n_images <= 65535u;
  const dim3 rgrid((P.out_cols + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup, P.out_rows, P.n_images);
  if (codec == ICAMD_DXT1) {
    if (rows) hipLaunchKernelGGL(icamd_downsample_dxt1_rows_kernel, rgrid, block, 0, stream, P);
    else hipLaunchKernelGGL(icamd_downsample_dxt1_kernel, grid, block, 0, stream, P);
  } else if (codec == ICAMD_DXT5) {
    if (rows) hipLaunchKernelGGL(icamd_downsample_dxt5_rows_kernel, rgrid, block, 0, stream, P);
    else hipLaunchKernelGGL(icamd_downsample_dxt5_kernel, grid, block, 0, stream, P);
  } else if (codec == ICAMD_ETC1 && P.etc_strategy == 3u && rows) {
    hipLaunchKernelGGL(icamd_downsample_etc1_heuristic_rows_kernel, rgrid, block, 0, stream, P);
  } else if (codec == ICAMD_ETC1) {
    const uint32_t lanes = (uint32_t)blockop_lanes(ICAMD_ETC1, P.etc_strategy == 3u ? 3 : 2, 0);
    const dim3 egrid((P.total_out + lanes - 1) / lanes), eblock(lanes);
    if (P.etc_strategy == 0u) hipLaunchKernelGGL(icamd_downsample_etc1_split_h_kernel, egrid, eblock, 0, stream, P);
    else if (P.etc_strategy == 1u) hipLaunchKernelGGL(icamd_downsample_etc1_split_v_kernel, egrid, eblock, 0, stream, P);
    else if (P.etc_strategy == 3u) hipLaunchKernelGGL(icamd_downsample_etc1_heuristic_kernel, egrid, eblock, 0, stream, P);
    else if (P.total_out <= kDownsampleQuadMaxBlocks && pad_border_quad())  // (the same A/B switch as the Pad border: ICAMD_PAD_BORDER_QUAD=0)
      hipLaunchKernelGGL(icamd_downsample_etc1_quad_kernel, dim3((P.total_out * 4u + 63u) / 64u), dim3(64), 0, stream, P);
    else hipLaunchKernelGGL(icamd_downsample_etc1_kernel, egrid, eblock, 0, stream, P);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_transcode_dxt1_to_etc1(void *blocks, uint32_t n_blocks, hipStream_t stream) {
  if (n_blocks == 0) return hipSuccess;
  const dim3 grid((n_blocks + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  hipLaunchKernelGGL(icamd_dxt1_to_etc1_kernel, grid, block, 0, stream, static_cast<uint2 *>(blocks), n_blocks);
  return hipGetLastError();
}

// ---- CreateSolidImage / CopySubimage on device-resident block grids (helper.h:522-592) ----

// One block replicated n times: lane i of the grid-stride loop writes block i (8 or 16 bytes, streaming stores; the
// pointer only needs the 4-byte alignment of the other block-domain operations).
struct FillParams {
  uint8_t *dst;
  uint64_t n_blocks;
  uint32_t w[4];
  uint32_t words;  // 2 or 4
};
// n <= kFillBatch images of blocks_per_image blocks each, image i filled with its own block w[i] (icamd_create_solid_batch_device)
constexpr uint32_t kFillBatch = 64;
struct FillBatchParams {
  uint8_t *dst;
  uint64_t dst_image_stride;
  uint32_t blocks_per_image, words, n;
  uint32_t w[kFillBatch][4];
};
extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_fill_blocks_batch_kernel(FillBatchParams P) {
  const uint32_t img = blockIdx.y;
  uint8_t *dst = P.dst + (size_t)img * P.dst_image_stride;
  const uint32_t w0 = P.w[img][0], w1 = P.w[img][1], w2 = P.w[img][2], w3 = P.w[img][3];
  const uint32_t stride = gridDim.x * kThreadsPerWorkgroup;
  for (uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x; k < P.blocks_per_image; k += stride) {
    if (P.words == 4) store_stream16(dst + (size_t)k * 16u, w0, w1, w2, w3);
    else store_stream8(dst + (size_t)k * 8u, w0, w1);
  }
}
extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_fill_blocks_kernel(FillParams P) {
  const uint64_t stride = (uint64_t)gridDim.x * kThreadsPerWorkgroup;
  for (uint64_t k = (uint64_t)blockIdx.x * kThreadsPerWorkgroup + threadIdx.x; k < P.n_blocks; k += stride) {
    if (P.words == 4) store_stream16(P.dst + k * 16u, P.w[0], P.w[1], P.w[2], P.w[3]);
    else store_stream8(P.dst + k * 8u, P.w[0], P.w[1]);
  }
}

// Sub-rectangle of a block grid: output block (r, c) = source block (r0 + r, c0 + c); grid = (column chunks, rows),
// consecutive lanes = consecutive blocks of one row (coalesced on both sides).
struct SubimageParams {
  const uint8_t *src;
  uint8_t *dst;
  uint32_t src_cols, r0, c0, rows, cols, words, row_first;
  uint64_t src_image_stride, dst_image_stride;  // blockIdx.z = image (icamd_copy_subimage_batch_device)
};
extern "C" __global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_copy_subimage_kernel(SubimageParams P) {
  const uint32_t c = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x, r = P.row_first + blockIdx.y;
  if (c >= P.cols) return;
  const size_t so = (size_t)blockIdx.z * P.src_image_stride + ((size_t)(P.r0 + r) * P.src_cols + P.c0 + c) * (P.words * 4u);
  const size_t dof = (size_t)blockIdx.z * P.dst_image_stride + ((size_t)r * P.cols + c) * (P.words * 4u);
  if (P.words == 4) {
    const U4 v = *reinterpret_cast<const U4 *>(P.src + so);
    store_stream16(P.dst + dof, v.x, v.y, v.z, v.w);
  } else {
    const U2 v = *reinterpret_cast<const U2 *>(P.src + so);
    store_stream8(P.dst + dof, v.x, v.y);
  }
}

hipError_t launch_fill_blocks(void *dst, uint64_t n_blocks, int block_bytes, const uint32_t words[4], hipStream_t stream) {
  if (n_blocks == 0) return hipSuccess;
  FillParams P;
  P.dst = static_cast<uint8_t *>(dst);
  P.n_blocks = n_blocks;
  P.words = (uint32_t)block_bytes / 4u;
  for (int i = 0; i < 4; ++i) P.w[i] = words[i];
  uint64_t wgs = (n_blocks + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup;
  if (wgs > 256u * 64u) wgs = 256u * 64u;  // 8 waves on every SIMD several times over; the loop covers the rest
  (void)hipGetLastError();
  hipLaunchKernelGGL(icamd_fill_blocks_kernel, dim3((uint32_t)wgs), dim3(kThreadsPerWorkgroup), 0, stream, P);
  return hipGetLastError();
}

hipError_t launch_fill_blocks_batch(void *dst, uint64_t dst_image_stride, uint32_t blocks_per_image, int block_bytes,
                                    const uint32_t (*words)[4], uint32_t n_images, hipStream_t stream) {
  if (blocks_per_image == 0 || n_images == 0) return hipSuccess;
  (void)hipGetLastError();
  uint32_t wgs = (blocks_per_image + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup;
  for (uint32_t first = 0; first < n_images; first += kFillBatch) {
    FillBatchParams P;
    P.n = n_images - first < kFillBatch ? n_images - first : kFillBatch;
    P.dst = static_cast<uint8_t *>(dst) + (uint64_t)first * dst_image_stride;
    P.dst_image_stride = dst_image_stride;
    P.blocks_per_image = blocks_per_image;
    P.words = (uint32_t)block_bytes / 4u;
    for (uint32_t i = 0; i < P.n; ++i)
      for (int j = 0; j < 4; ++j) P.w[i][j] = words[first + i][j];
    // ~8 waves on every SIMD over the whole launch; the loop covers the rest of an image
    const uint32_t per_image = (256u * 32u + P.n - 1) / P.n;
    hipLaunchKernelGGL(icamd_fill_blocks_batch_kernel, dim3(wgs < per_image ? wgs : per_image, P.n), dim3(kThreadsPerWorkgroup), 0, stream, P);
  }
  return hipGetLastError();
}

hipError_t launch_copy_subimage(int block_bytes, const void *src, uint32_t src_cols, uint32_t r0, uint32_t c0,
                                uint32_t rows, uint32_t cols, void *dst, hipStream_t stream, uint32_t n_images,
                                uint64_t src_image_stride, uint64_t dst_image_stride) {
  if (rows == 0 || cols == 0 || n_images == 0) return hipSuccess;
  SubimageParams P;
  P.src = static_cast<const uint8_t *>(src);
  P.dst = static_cast<uint8_t *>(dst);
  P.src_cols = src_cols; P.r0 = r0; P.c0 = c0; P.rows = rows; P.cols = cols;
  P.words = (uint32_t)block_bytes / 4u;
  P.src_image_stride = src_image_stride;
  P.dst_image_stride = dst_image_stride;
  (void)hipGetLastError();
  const uint32_t gx = (cols + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup;
  for (uint32_t img0 = 0; img0 < n_images; img0 += 65535u) {
    const uint32_t nz = n_images - img0 < 65535u ? n_images - img0 : 65535u;
    P.src = static_cast<const uint8_t *>(src) + (uint64_t)img0 * src_image_stride;
    P.dst = static_cast<uint8_t *>(dst) + (uint64_t)img0 * dst_image_stride;
    for (uint32_t first = 0; first < rows; first += 65535u) {
      P.row_first = first;
      hipLaunchKernelGGL(icamd_copy_subimage_kernel, dim3(gx, rows - first < 65535u ? rows - first : 65535u, nz),
                         dim3(kThreadsPerWorkgroup), 0, stream, P);
    }
  }
  return hipGetLastError();
}

}  // namespace icamd
