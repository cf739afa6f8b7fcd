// decode_kernels.hip -- DXT1 / DXT5 / ETC1 decode kernels for gfx950 ("next" row 8f.1): one block per
// lane, 8/16-byte coalesced block loads, 12/16-byte row-segment stores (Compressor4x4Helper::Decompress,
// internal/compressor4x4_helper.h:218-262: blocks past the image edge are clipped).
#include "blockops_block.h"  // decode_block.h + the palette-plane row decoders
#include "ic_launch.h"
#include "ic_amd.h"

namespace icamd {

template <int CODEC>
__device__ __forceinline__ void decode_one(const DecodeParams &P, uint32_t k) {
  constexpr int COMPS = CODEC == ICAMD_DXT5 ? 4 : 3;
  const uint32_t img = fastdiv(k, P.div_bpi);
  const uint32_t rem = k - img * P.blocks_per_image;
  const uint32_t brow = fastdiv(rem, P.div_cols), bcol = rem - brow * P.block_cols;
  const uint8_t *src = P.blocks + (size_t)img * P.src_image_stride + (size_t)rem * (CODEC == ICAMD_DXT5 ? 16 : 8);
  const bool swap = P.swap_rb != 0;
  uint32_t w[4] = { 0, 0, 0, 0 };
  if (CODEC == ICAMD_DXT5) {
    const U4 v = load_stream(reinterpret_cast<const U4 *>(src));  // no alignment assumed: the caller owns the block pointer
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
    const U2 v = load_stream(reinterpret_cast<const U2 *>(src));
    w[0] = v.x; w[1] = v.y;
  }
  uint8_t *dst = P.pixels + (size_t)img * P.dst_image_stride;
  const uint32_t row = brow * 4, col = bcol * 4;
  if (row + 4 <= P.height && col + 4 <= P.width) {
    // whole block inside the image: rows straight in memory order from the palette planes (blockops_block.h)
    uint32_t rows[4][4];
    decode_block_rows<CODEC>(w, swap, rows);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      uint8_t *q = dst + (size_t)(row + y) * P.row_stride + (size_t)col * COMPS;
      // r05: non-temporal -- a wave's store instruction writes whole lines (64 x 16 or 12 contiguous bytes), each byte once
      if (COMPS == 4) store_stream16(q, rows[y][0], rows[y][1], rows[y][2], rows[y][3]);
      else store_stream12(q, rows[y][0], rows[y][1], rows[y][2]);
    }
  } else {  // clipped at the image's edge (helper.h:218-262): pixel by pixel
    uint32_t px[16];
    if (CODEC == ICAMD_DXT5) {
      decode_dxt_colors(w[2], w[3], swap, true, px);
      decode_dxt5_alpha(w[0], w[1], px);
    } else if (CODEC == ICAMD_DXT1) {
      decode_dxt_colors(w[0], w[1], swap, false, px);
    } else {
      decode_etc1(w[0], w[1], px);
    }
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (row + y < P.height && col + x < P.width) {
          uint8_t *q = dst + (size_t)(row + y) * P.row_stride + (size_t)(col + x) * COMPS;
          const uint32_t v = px[4 * y + x];
          q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16);
          if (COMPS == 4) q[3] = (uint8_t)(v >> 24);
        }
  }
}

extern "C" {
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt1_decode_kernel(DecodeParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) decode_one<ICAMD_DXT1>(P, k);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_dxt5_decode_kernel(DecodeParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) decode_one<ICAMD_DXT5>(P, k);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_etc1_decode_kernel(DecodeParams P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k < P.total_blocks) decode_one<ICAMD_ETC1>(P, k);
}
}  // extern "C"

// PVRTC1 2 bpp / 4 bpp (extensions, see decode_block.h): one block (8 x 4 / 4 x 4 pixels) per lane, lanes in raster order of the
// block grid; the nine block words come from their Z-order slots.  Small textures (block grids below 32 x 8).
template <int BPP>
__device__ __forceinline__ void pvrtc_decode_generic(const DecodeParams &P) {
  const uint32_t k = blockIdx.x * kThreadsPerWorkgroup + threadIdx.x;
  if (k >= P.total_blocks) return;
  const uint32_t img = fastdiv(k, P.div_bpi), rem = k - img * P.blocks_per_image;
  const uint32_t by = fastdiv(rem, P.div_cols), bx = rem - by * P.block_cols;  // block_cols = width / 8 (2 bpp), / 4 (4 bpp)
  const U2 *blocks = reinterpret_cast<const U2 *>(P.blocks + (size_t)img * P.src_image_stride);
  uint32_t mod[9], col[9];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const uint32_t nx = (bx + (uint32_t)dx) & (P.block_cols - 1u), ny = (by + (uint32_t)dy) & (P.block_rows - 1u);
      const U2 w = blocks[spread_bits16(nx) << 1 | spread_bits16(ny)];
      mod[3 * (dy + 1) + dx + 1] = w.x;
      col[3 * (dy + 1) + dx + 1] = w.y;
    }
  uint8_t *dst = P.pixels + (size_t)img * P.dst_image_stride + (size_t)(by * 4u) * P.row_stride + (size_t)bx * (BPP == 2 ? 32u : 16u);
  if (BPP == 2) {
    uint32_t px[32];
    decode_pvrtc2_block(mod, col, px);
#pragma unroll
    for (int y = 0; y < 4; ++y) {  // (one 64-bit product above; the rows advance by additions)
      U4 v0 = { px[8 * y], px[8 * y + 1], px[8 * y + 2], px[8 * y + 3] };
      U4 v1 = { px[8 * y + 4], px[8 * y + 5], px[8 * y + 6], px[8 * y + 7] };
      *reinterpret_cast<U4 *>(dst) = v0;
      *reinterpret_cast<U4 *>(dst + 16) = v1;
      dst += P.row_stride;
    }
  } else {
    uint32_t px[16];
    decode_pvrtc4_block(mod[4], col, px);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      U4 v0 = { px[4 * y], px[4 * y + 1], px[4 * y + 2], px[4 * y + 3] };
      *reinterpret_cast<U4 *>(dst) = v0;
      dst += P.row_stride;
    }
  }
}
// The same for block grids of at least 32 x 8: one workgroup per TILE of 32 x 8 blocks.  Every lane expands its own block's
// colour word once (the packed-field -> channel-pair expansion is a quarter of the per-block work when each lane does it for all
// nine neighbours), the 84 blocks of the one-block ring around the tile are expanded by the first 84 lanes, the pairs (16 B per
// block) meet in LDS, one barrier.  The modulation / mode words of the four orthogonal neighbours (2 bpp only: its unstored
// pixels look at them) still come from memory (L1 hits).
constexpr uint32_t kPvrtcTileW = 32, kPvrtcTileH = 8;
template <int BPP>
__device__ __forceinline__ void pvrtc_decode_tile(const DecodeParams &P, U4 *pairs, U4 (*turn)[128]) {
  const uint32_t log2_cols = 31u - (uint32_t)__builtin_clz(P.block_cols), log2_rows = 31u - (uint32_t)__builtin_clz(P.block_rows);
  const uint32_t tiles_x = P.block_cols >> 5, log2_tx = log2_cols - 5u, log2_tiles = log2_tx + log2_rows - 3u;
  const uint32_t img = blockIdx.x >> log2_tiles, tile = blockIdx.x & ((1u << log2_tiles) - 1u);
  const uint32_t bx0 = (tile & (tiles_x - 1u)) << 5, by0 = (tile >> log2_tx) << 3;
  const uint32_t cmask = P.block_cols - 1u, rmask = P.block_rows - 1u;
  const U2 *blocks = reinterpret_cast<const U2 *>(P.blocks + (size_t)img * P.src_image_stride);
  const uint32_t lx = threadIdx.x & 31u, ly = threadIdx.x >> 5;
  const uint32_t bx = bx0 + lx, by = by0 + ly;
  auto word_at = [&](uint32_t x, uint32_t y) { return blocks[spread_bits16(x & cmask) << 1 | spread_bits16(y & rmask)]; };
  // the four orthogonal neighbours step in the Z-order domain itself: x lives on the odd bits, y on the even ones; filling
  // the other coordinate's bits with ones lets a carry run across them, zeros let a borrow, and the spread grid mask wraps
  const uint32_t mx = spread_bits16(cmask) << 1, my = spread_bits16(rmask);
  auto publish = [&](uint32_t cell, uint32_t colour_word) {
    uint32_t e[4];
    pvrtc_expand_colors(colour_word, e);
    const U4 v = { e[0], e[1], e[2], e[3] };
    pairs[cell] = v;
  };
  const uint32_t sx = spread_bits16(bx) << 1, sy = spread_bits16(by);
  const U2 own = blocks[sx | sy];
  publish((ly + 1u) * (kPvrtcTileW + 2u) + lx + 1u, own.y);
  if (threadIdx.x < 2u * (kPvrtcTileW + 2u) + 2u * kPvrtcTileH) {  // the ring: top row, bottom row, left column, right column
    const uint32_t t = threadIdx.x;
    uint32_t cx, cy;  // cell coordinates in the (W + 2) x (H + 2) array
    if (t < kPvrtcTileW + 2u) { cx = t; cy = 0u; }
    else if (t < 2u * (kPvrtcTileW + 2u)) { cx = t - (kPvrtcTileW + 2u); cy = kPvrtcTileH + 1u; }
    else if (t < 2u * (kPvrtcTileW + 2u) + kPvrtcTileH) { cx = 0u; cy = t - 2u * (kPvrtcTileW + 2u) + 1u; }
    else { cx = kPvrtcTileW + 1u; cy = t - 2u * (kPvrtcTileW + 2u) - kPvrtcTileH + 1u; }
    publish(cy * (kPvrtcTileW + 2u) + cx, word_at(bx0 + cx - 1u, by0 + cy - 1u).y);
  }
  uint32_t mod[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, col[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  mod[4] = own.x; col[4] = own.y;
  if (BPP == 2) {
    { const U2 w = blocks[sx | ((sy - 1u) & my)]; mod[1] = w.x; col[1] = w.y; }
    { const U2 w = blocks[((sx - 2u) & mx) | sy]; mod[3] = w.x; col[3] = w.y; }
    { const U2 w = blocks[(((sx | 0x55555555u) + 2u) & mx) | sy]; mod[5] = w.x; col[5] = w.y; }
    { const U2 w = blocks[sx | (((sy | 0xaaaaaaaau) + 1u) & my)]; mod[7] = w.x; col[7] = w.y; }
  }
  __syncthreads();
  uint32_t C[3][3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const U4 v = pairs[(ly + (uint32_t)r) * (kPvrtcTileW + 2u) + lx + (uint32_t)c];
      C[r][c][0] = v.x; C[r][c][1] = v.y; C[r][c][2] = v.z; C[r][c][3] = v.w;
    }
  const size_t row_stride = P.row_stride;
  if (BPP == 4) {
    // a lane's 16 bytes of a pixel row are one store: lanes 0-31 / 32-63 of a wave write 512 contiguous bytes (whole lines) of
    // the two block rows they hold; non-temporal, row by row as the rows are finished
    uint8_t *dst = P.pixels + (size_t)img * P.dst_image_stride + (size_t)(by * 4u) * row_stride + (size_t)bx * 16u;
    decode_pvrtc4_block_rows(C, own.x, (own.y & 1u) != 0u, [&](int y, const uint32_t row[4]) {
      store_stream16(dst, row[0], row[1], row[2], row[3]);
      dst += row_stride;
    });
    return;
  }
  // r05: a lane's 32 bytes of a pixel row leave as two 16-byte stores, and with the lanes' own data those would each fill every
  // other 16 bytes of the lines they touch (measured: this store pattern ALONE took 0.30 ms per 16 x 4096^2, the arithmetic 0.21).
  // Each wave therefore turns its row segment round in 2 KiB of LDS -- lane i writes pieces 2 i and 2 i + 1, reads pieces i and
  // 64 + i -- so that one store instruction writes one whole kilobyte of ONE pixel row (lanes 0-31 hold block row 2 w of the
  // tile, lanes 32-63 block row 2 w + 1), non-temporal: 0.166 ms for the same bytes.  No barrier: LDS is in order within a wave.
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  U4 *const mine = turn[wave];
  uint8_t *dst = P.pixels + (size_t)img * P.dst_image_stride + (size_t)((by0 + 2u * wave) * 4u) * row_stride + (size_t)bx0 * 32u + lane * 16u;
  const size_t next_block_row = 4u * row_stride;
  decode_pvrtc2_block_rows(C, mod, col, [&](int y, const uint32_t row[8]) {  // (one 64-bit product above; the rows advance by additions)
    const U4 v0 = { row[0], row[1], row[2], row[3] }, v1 = { row[4], row[5], row[6], row[7] };
    mine[2u * lane] = v0;
    mine[2u * lane + 1u] = v1;
    __builtin_amdgcn_wave_barrier();
    const U4 a = mine[lane], b = mine[64u + lane];
    __builtin_amdgcn_wave_barrier();
    store_stream16(dst, a.x, a.y, a.z, a.w);
    store_stream16(dst + next_block_row, b.x, b.y, b.z, b.w);
    dst += row_stride;
  });
}

extern "C" {
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pvrtc2_decode_kernel(DecodeParams P) { pvrtc_decode_generic<2>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pvrtc4_decode_kernel(DecodeParams P) { pvrtc_decode_generic<4>(P); }
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pvrtc2_decode_tile_kernel(DecodeParams P) {
  __shared__ U4 pairs[(kPvrtcTileH + 2) * (kPvrtcTileW + 2)];
  __shared__ U4 turn[kThreadsPerWorkgroup / 64][128];
  pvrtc_decode_tile<2>(P, pairs, turn);
}
__global__ void __launch_bounds__(kThreadsPerWorkgroup) icamd_pvrtc4_decode_tile_kernel(DecodeParams P) {
  __shared__ U4 pairs[(kPvrtcTileH + 2) * (kPvrtcTileW + 2)];
  pvrtc_decode_tile<4>(P, pairs, nullptr);
}
}  // extern "C"

hipError_t launch_decode(int codec, const DecodeParams &P, hipStream_t stream) {
  if (P.total_blocks == 0) return hipSuccess;
  (void)hipGetLastError();  // a stale error of another library on this thread is not this launch's
  const dim3 grid((P.total_blocks + kThreadsPerWorkgroup - 1) / kThreadsPerWorkgroup), block(kThreadsPerWorkgroup);
  if (codec == ICAMD_DXT1) hipLaunchKernelGGL(icamd_dxt1_decode_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_DXT5) hipLaunchKernelGGL(icamd_dxt5_decode_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_ETC1) hipLaunchKernelGGL(icamd_etc1_decode_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_PVRTC2 && P.block_cols >= kPvrtcTileW && P.block_rows >= kPvrtcTileH &&
           (P.block_cols & (P.block_cols - 1u)) == 0u && (P.block_rows & (P.block_rows - 1u)) == 0u)
    hipLaunchKernelGGL(icamd_pvrtc2_decode_tile_kernel, grid, block, 0, stream, P);  // whole tiles: total_blocks / 256 workgroups
  else if (codec == ICAMD_PVRTC2) hipLaunchKernelGGL(icamd_pvrtc2_decode_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_PVRTC4 && P.block_cols >= kPvrtcTileW && P.block_rows >= kPvrtcTileH &&
           (P.block_cols & (P.block_cols - 1u)) == 0u && (P.block_rows & (P.block_rows - 1u)) == 0u)
    hipLaunchKernelGGL(icamd_pvrtc4_decode_tile_kernel, grid, block, 0, stream, P);
  else if (codec == ICAMD_PVRTC4) hipLaunchKernelGGL(icamd_pvrtc4_decode_kernel, grid, block, 0, stream, P);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace icamd
