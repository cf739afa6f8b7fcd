// ic_launch.h -- host-side launch entry points of the kernel translation units
// (dxt_kernels.hip, etc1_kernels.hip, pvrtc_kernels.hip, decode_kernels.hip); called by ic_capi.hip.
#ifndef ICAMD_IC_LAUNCH_H_
#define ICAMD_IC_LAUNCH_H_

#include <hip/hip_runtime.h>

#include <atomic>

#include "ic_device.h"

namespace icamd {

constexpr int kThreadsPerWorkgroup = 256;  // 4 waves; one 4x4 block per lane

// Compute units of a device (cached per ordinal): launch shapes and time models count the workgroup slots of THIS device -- a
// partitioned MI355X (CPX: 32 CUs per partition) or another part must not be taken for 256 CUs.  A wrong figure
// costs time, never bytes.
inline uint32_t device_compute_units(int dev) {
  static std::atomic<uint32_t> cached[64];
  if (dev >= 0 && dev < 64) {
    const uint32_t c = cached[dev].load();
    if (c) return c;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    cus = 256;
  }
  if (dev >= 0 && dev < 64) cached[dev].store((uint32_t)cus);
  return (uint32_t)cus;
}

// Tile shape for a block grid `block_cols` wide (GridParams::log2_tile_cols): the smallest power of two that covers
// a block row, at most 256.
inline uint32_t tile_log2_cols(uint32_t block_cols) {
  uint32_t l = 0;
  while (l < 8 && (1u << l) < block_cols) ++l;
  return l;
}
// Launches `kernel` over every tile of every image; images go into grid.z in chunks of at most 65 535, tile rows into
// grid.y in chunks of at most 65 535 (GridParams::tile_row0), so any uint32 geometry the reference accepts runs.
// max_log2_cols < 8 asks for squarer tiles (e.g. 4: 16 x 16 blocks, a wave = 16 x 4 blocks = 64 x 16 pixels):
// worse coalescing, but the lanes of a wave see more homogeneous content, which is what wave-uniform shortcuts need.
template <typename Kernel>
// wave_workgroups: every wave of a tile is launched as its own 64-lane workgroup (grid.x = 4 x column tiles; the kernel
// undoes it: etc1_locate_tile)
hipError_t launch_tiled(Kernel wide_kernel, Kernel narrow_kernel, GridParams P, hipStream_t stream,
                        uint32_t max_log2_cols = 8, uint32_t wide_rows = 1, bool wave_workgroups = false, uint32_t lanes_per_block = 1) {
  if (P.n_images == 0 || P.block_rows == 0 || P.block_cols == 0) return hipSuccess;
  P.log2_tile_cols = tile_log2_cols(P.block_cols);
  if (P.log2_tile_cols > max_log2_cols) P.log2_tile_cols = max_log2_cols;
  // A lane addresses its block with a 32-bit offset from the tile's (64-bit, uniform) origin: up to 4 * 256 - 1 rows
  // of stride in a 1 x 256 tile.  Rows of 4 MiB and more (or grids of 2^20 block columns and more, for the output
  // offset) get 256 x 1 tiles, where the offset is at most three rows; rows of more than a third of 4 GiB take the
  // 64-bit clamp-to-edge gather for every block.
  if ((uint64_t)P.row_stride * 1024u >= (1ull << 32) || (uint64_t)P.block_cols * 4096u >= (1ull << 32)) P.log2_tile_cols = 8;
  P.force_gather = (uint64_t)P.row_stride * 3u + 8192u >= (1ull << 32) ? 1u : 0u;
  const uint32_t cols = 1u << P.log2_tile_cols, rows = P.log2_tile_cols == 8 ? wide_rows : 256u >> P.log2_tile_cols;
  const uint32_t gx = (uint32_t)(((uint64_t)P.block_cols + cols - 1) / cols);
  const uint32_t gy = (uint32_t)(((uint64_t)P.block_rows + rows - 1) / rows);
  (void)hipGetLastError();  // do not attribute a stale error of another library on this thread to these launches
  for (uint32_t first = 0; first < P.n_images; first += 65535u) {
    const uint32_t count = P.n_images - first < 65535u ? P.n_images - first : 65535u;
    for (uint32_t row0 = 0; row0 < gy; row0 += 65535u) {
      GridParams Q = P;
      Q.src = P.src + (uint64_t)first * P.src_image_stride;
      Q.dst = P.dst + (uint64_t)first * P.dst_image_stride;
      Q.tile_row0 = row0;
      const uint32_t gyc = gy - row0 < 65535u ? gy - row0 : 65535u;
      // (lanes_per_block = 4: the tile's 256 blocks are 16 one-wave workgroups of 16 blocks x 4 lanes, etc1_locate_tile_quad)
      hipLaunchKernelGGL(P.log2_tile_cols == 8 ? wide_kernel : narrow_kernel,
                         dim3(wave_workgroups ? gx * 4u * lanes_per_block : gx, gyc, count),
                         dim3(wave_workgroups ? 64 : kThreadsPerWorkgroup), 0, stream, Q);
    }
  }
  return hipGetLastError();
}

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
  // region_blocks != 0: encode only the blocks [region_first, region_first + region_blocks) of ONE image's Z-order
  // output (a power-of-two, aligned range = a rectangle of blocks); dst receives just those 8 * region_blocks bytes
  uint32_t region_first = 0, region_blocks = 0;
  // the library's own host-buffer path: never borrow the caller-owned workspace (icamd_pvrtc2_set_workspace)
  bool internal_workspace = false;
};
hipError_t launch_pvrtc2(const PvrtcParams &P, hipStream_t stream);
// PVRTC1 4 bpp (extension, parity unpinned): same parameters (no regions), 4 x 4-pixel blocks, size * size / 2 bytes per image
hipError_t launch_pvrtc4(const PvrtcParams &P, hipStream_t stream);
size_t pvrtc4_workspace_bytes(uint32_t size, uint32_t n_images);
const char *pvrtc4_kernel_name();
// Scratch the PVRTC encoder needs between its two kernels for n_images size x size textures (8 bytes per block of one
// launch group), and the thread-local caller-owned override of the library's internal scratch buffer.
size_t pvrtc2_workspace_bytes(uint32_t size, uint32_t n_images);
void pvrtc2_set_workspace(void *d_workspace, size_t bytes);
// this thread's library-owned workspace slot (0 / 1) for the PVRTC launches that follow: one per alternating stream
void pvrtc2_select_workspace(int slot);
// which kernels whole-texture PVRTC launches take: mode 0 = automatic, 1 = always morph + encode, 2 = the one-pass kernel
// wherever it is eligible (512^2 ... 4096^2); log2_strip < 0 = automatic strip height of the one-pass kernel
void pvrtc2_tune(int mode, int log2_strip);

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
  // Downsample / Pad: n_images equally shaped block grids per launch (total_out = out_rows * out_cols * n_images)
  uint32_t n_images = 1, out_per_image = 0;
  uint64_t src_image_stride = 0, dst_image_stride = 0;  // bytes
  FastDiv div_out_per_image = { 0, 0, 1 };
  // ETC1 Pad, kSmallerError, ONE launch (r05): the first border_wgs workgroups are the pad blocks' quad lanes
  // (border_lanes = 4 * pad blocks over all images, border_lanes_per_image of them per image), the rest copy the grid
  uint32_t border_wgs = 0, border_lanes = 0, border_lanes_per_image = 0;
  FastDiv div_border_lanes_per_image = { 0, 0, 1 };
};
hipError_t launch_pad(int codec, const BlockOpParams &P, hipStream_t stream);
hipError_t launch_downsample(int codec, const BlockOpParams &P, hipStream_t stream);
hipError_t launch_transcode_dxt1_to_etc1(void *blocks, uint32_t n_blocks, hipStream_t stream);
// CreateSolidImage: `words` (block_bytes / 4 of them) replicated n_blocks times.  CopySubimage: rows x cols blocks
// starting at block (r0, c0) of a grid src_cols blocks wide.
hipError_t launch_fill_blocks(void *dst, uint64_t n_blocks, int block_bytes, const uint32_t words[4], hipStream_t stream);
hipError_t launch_copy_subimage(int block_bytes, const void *src, uint32_t src_cols, uint32_t r0, uint32_t c0,
                                uint32_t rows, uint32_t cols, void *dst, hipStream_t stream, uint32_t n_images = 1,
                                uint64_t src_image_stride = 0, uint64_t dst_image_stride = 0);
// n_images grids of blocks_per_image blocks, dst_image_stride bytes apart, image i filled with words[i]
hipError_t launch_fill_blocks_batch(void *dst, uint64_t dst_image_stride, uint32_t blocks_per_image, int block_bytes,
                                    const uint32_t (*words)[4], uint32_t n_images, hipStream_t stream);
// Diagnostics: one wave spins for `ticks` periods of the constant-rate clock (s_memrealtime) and records how many shader
// cycles (s_memtime) passed meanwhile: d_out[0] = shader cycles, d_out[1] = constant-rate ticks.
hipError_t launch_clock_probe(uint64_t *d_out, uint64_t ticks, hipStream_t stream);

const char *dxt_kernel_name(int codec, int comps);
const char *etc1_kernel_name(int comps);
const char *pvrtc2_kernel_name();

}  // namespace icamd
#endif  // ICAMD_IC_LAUNCH_H_
