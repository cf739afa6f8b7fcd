// pvrtc_kernels.hip -- fused PVRTC1 2bpp encoder for gfx950 (MI355X).
//
// The reference runs three whole-image passes (Morph -> Modulate -> Encode, pvrtc_compressor.cc:586-597)
// through three heap images.  Here ONE kernel does all three per 16x16-block tile (128x64 pixels) with
// the intermediate A/B colours and modulation values living in LDS only:
//
//   workgroup = 320 lanes = 256 "tile" lanes (one 8x4 block each, its 32 pixels held in VGPRs for the
//               whole kernel) + one "ring" wave that owns the 68 halo blocks around the tile;
//   phase 1   every lane: extremes -> channel-reduced A/B of its block -> LDS ab[18][18]         (Morph)
//   phase 2   tile lanes: 32 modulation values from the 3x3 block neighbourhood in LDS;
//             ring lanes right of / below the tile: the 4 / 8 values the tile's edge needs      (Modulate)
//             first row / first column of every block -> LDS for the neighbours
//   phase 3   tile lanes: mode decision (needs the pixel column right of and the row below the
//             block), modulation word, colour word, one 8-byte store at the block's Z-order index (Encode)
//
// Tile lanes are numbered in Z-order inside the tile, so a wave's 64 stores are 512 contiguous bytes
// (pvrtc.cc:551-580 emits blocks in Z-order) and its loads are 8 runs of 256 contiguous bytes per row.
// Block coordinates wrap toroidally (pvrtc.cc:208-237, :416-423), which also makes images smaller than
// one tile work unchanged.  Source pixels are read ~1.27x (halo), everything else stays on chip;
// algorithmic traffic is 4 B/px in + 0.25 B/px out.
#include "ic_launch.h"
#include "ic_amd.h"
#include "pvrtc_block.h"

namespace icamd {

namespace {

constexpr int kTile = 16;               // blocks per tile edge
constexpr int kHalo = kTile + 2;        // 18
constexpr int kTileLanes = kTile * kTile;
constexpr int kLanes = kTileLanes + 64; // + one ring wave
constexpr int kRingBlocks = 4 * kTile + 4;

struct ModEdge {   // what neighbouring blocks need from a block's modulation values
  uint32_t row0[2];  // first pixel row, bytes in x order
  uint32_t col0;     // first pixel column, byte y
};

// compact the odd / even bits of an 8-bit Z-order lane id into 4-bit tile coordinates
__device__ __forceinline__ uint32_t compact_even_bits8(uint32_t v) {
  v &= 0x55u;
  v = (v | v >> 1) & 0x33u;
  v = (v | v >> 2) & 0x0fu;
  return v;
}

// Ring slot r (0..67) -> halo coordinates.  Slots 0..63 are the four edges (right column first, then
// bottom row: the two that must also produce modulation values), 64..67 the corners.
__device__ __forceinline__ void ring_coords(uint32_t r, uint32_t &lx, uint32_t &ly) {
  const uint32_t side = r >> 4, i = (r & 15u) + 1u;
  if (r >= 64u) { lx = (r & 1u) ? kHalo - 1 : 0; ly = (r & 2u) ? kHalo - 1 : 0; }
  else if (side == 0u) { lx = kHalo - 1; ly = i; }   // right column
  else if (side == 1u) { lx = i; ly = kHalo - 1; }   // bottom row
  else if (side == 2u) { lx = i; ly = 0; }           // top row
  else { lx = 0; ly = i; }                           // left column
}

template <int XI, int YI>
__device__ __forceinline__ void all_mods(const uint32_t px[32], const PvrtcAB nb[3][3], uint32_t rows[4][2]) {
  const uint32_t m = pvrtc_pixel_mod<XI, YI>(px[8 * YI + XI], nb);
  rows[YI][XI >> 2] |= m << (8 * (XI & 3));
  if constexpr (XI + 1 < 8) all_mods<XI + 1, YI>(px, nb, rows);
  else if constexpr (YI + 1 < 4) all_mods<0, YI + 1>(px, nb, rows);
}

}  // namespace

extern "C" __global__ void __launch_bounds__(kLanes) icamd_pvrtc2_kernel(PvrtcParams P) {
  __shared__ uint32_t lds_stash[8][kLanes][4];  // 40 KiB: per-lane pixel stash for index lookups
  __shared__ PvrtcAB lds_ab[kHalo][kHalo];      // 5 KiB
  __shared__ ModEdge lds_edge[kHalo][kHalo];    // 4 KiB

  const uint32_t tid = threadIdx.x;
  const uint32_t n = P.size, bw_mask = (n >> 3) - 1u, bh_mask = (n >> 2) - 1u;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(P.src + (size_t)blockIdx.z * P.src_image_stride);
  const uint32_t image0 = img[0];
  const bool is_tile = tid < kTileLanes;

  Stash32 stash;
  stash.base = &lds_stash[0][tid][0];
  stash.row_dwords = kLanes * 4;

  uint32_t px[32];
  uint32_t lx = 0, ly = 0, col_a = 0, col_b = 0;

  // ---- phase 1: load the block, find its two colours, publish them
  auto phase1 = [&](uint32_t hx, uint32_t hy) {
    const uint32_t bx = (blockIdx.x * kTile + hx - 1u) & bw_mask;
    const uint32_t by = (blockIdx.y * kTile + hy - 1u) & bh_mask;
    const uint32_t *p = img + (size_t)(by * 4u) * n + bx * 8u;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const uint4 v0 = *reinterpret_cast<const uint4 *>(p + (size_t)y * n);
      const uint4 v1 = *reinterpret_cast<const uint4 *>(p + (size_t)y * n + 4);
      px[8 * y + 0] = v0.x; px[8 * y + 1] = v0.y; px[8 * y + 2] = v0.z; px[8 * y + 3] = v0.w;
      px[8 * y + 4] = v1.x; px[8 * y + 5] = v1.y; px[8 * y + 6] = v1.z; px[8 * y + 7] = v1.w;
    }
    uint32_t a, b;
    pvrtc_extremes(px, image0, stash, a, b);
    col_a = channel_reduce(a, false);
    col_b = channel_reduce(b, true);
    PvrtcAB e = { pair_rb(col_a), pair_ga(col_a), pair_rb(col_b), pair_ga(col_b) };
    lds_ab[hy][hx] = e;
  };

  if (is_tile) {
    lx = compact_even_bits8(tid >> 1) + 1u;  // x from the odd bits, y from the even bits (pvrtc.cc:83-84)
    ly = compact_even_bits8(tid) + 1u;
    phase1(lx, ly);
  } else {
    const uint32_t r = tid - kTileLanes;
    if (r < kRingBlocks - 64u) {  // the four corners first, so the edge block's pixels are the ones kept
      ring_coords(r + 64u, lx, ly);
      phase1(lx, ly);
    }
    ring_coords(r, lx, ly);
    phase1(lx, ly);
  }
  __syncthreads();

  // ---- phase 2: modulation values
  uint32_t rows[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
  PvrtcAB nb[3][3];
  if (is_tile) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) nb[dy][dx] = lds_ab[ly + dy - 1][lx + dx - 1];
    all_mods<0, 0>(px, nb, rows);
    ModEdge e;
    e.row0[0] = rows[0][0];
    e.row0[1] = rows[0][1];
    e.col0 = (rows[0][0] & 0xffu) | (rows[1][0] & 0xffu) << 8 | (rows[2][0] & 0xffu) << 16 | (rows[3][0] & 0xffu) << 24;
    lds_edge[ly][lx] = e;
  } else if (lx == kHalo - 1 && ly >= 1 && ly <= kTile) {
    // halo block right of the tile: only its first pixel column is needed (sources: columns lx-1, lx)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      nb[dy][0] = lds_ab[ly + dy - 1][lx - 1];
      nb[dy][1] = lds_ab[ly + dy - 1][lx];
      nb[dy][2] = nb[dy][1];  // never read for x_in < 4
    }
    ModEdge e;
    e.row0[0] = e.row0[1] = 0;
    e.col0 = pvrtc_pixel_mod<0, 0>(px[0], nb) | pvrtc_pixel_mod<0, 1>(px[8], nb) << 8 |
             pvrtc_pixel_mod<0, 2>(px[16], nb) << 16 | pvrtc_pixel_mod<0, 3>(px[24], nb) << 24;
    lds_edge[ly][lx] = e;
  } else if (ly == kHalo - 1 && lx >= 1 && lx <= kTile) {
    // halo block below the tile: only its first pixel row is needed (sources: rows ly-1, ly)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      nb[0][dx] = lds_ab[ly - 1][lx + dx - 1];
      nb[1][dx] = lds_ab[ly][lx + dx - 1];
      nb[2][dx] = nb[1][dx];  // never read for y_in < 2
    }
    ModEdge e;
    e.row0[0] = pvrtc_pixel_mod<0, 0>(px[0], nb) | pvrtc_pixel_mod<1, 0>(px[1], nb) << 8 |
                pvrtc_pixel_mod<2, 0>(px[2], nb) << 16 | pvrtc_pixel_mod<3, 0>(px[3], nb) << 24;
    e.row0[1] = pvrtc_pixel_mod<4, 0>(px[4], nb) | pvrtc_pixel_mod<5, 0>(px[5], nb) << 8 |
                pvrtc_pixel_mod<6, 0>(px[6], nb) << 16 | pvrtc_pixel_mod<7, 0>(px[7], nb) << 24;
    e.col0 = 0;
    lds_edge[ly][lx] = e;
  }
  __syncthreads();

  // ---- phase 3: mode, modulation word, colour word, store
  if (is_tile) {
    const uint32_t bx = blockIdx.x * kTile + lx - 1u, by = blockIdx.y * kTile + ly - 1u;
    if (bx <= bw_mask && by <= bh_mask) {
      const uint32_t right_col = lds_edge[ly][lx + 1].col0;
      const uint32_t below[2] = { lds_edge[ly + 1][lx].row0[0], lds_edge[ly + 1][lx].row0[1] };
      bool one_bpp;
      const uint32_t data = pvrtc_block_modulation(rows, right_col, below, &one_bpp);
      const uint32_t colors = pvrtc_pack_colors(col_a, col_b, one_bpp);
      uint8_t *dst = P.dst + (size_t)blockIdx.z * P.dst_image_stride + (size_t)pvrtc_z_index(bx, by) * 8u;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(data, colors);
    }
  }
}

const char *pvrtc2_kernel_name() { return "icamd_pvrtc2_kernel"; }

hipError_t launch_pvrtc2(const PvrtcParams &P, hipStream_t stream) {
  if (P.n_images == 0) return hipSuccess;
  const uint32_t bw = P.size / 8, bh = P.size / 4;
  const dim3 grid((bw + kTile - 1) / kTile, (bh + kTile - 1) / kTile, P.n_images), block(kLanes);
  if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(icamd_pvrtc2_kernel, grid, block, 0, stream, P);
  return hipGetLastError();
}

}  // namespace icamd
