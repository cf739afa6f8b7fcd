// pvrtc_kernels.hip -- PVRTC1 2bpp encoder for gfx950 (MI355X): two barrier-free kernels.
//
// The reference runs three whole-image passes (Morph -> Modulate -> Encode, pvrtc_compressor.cc:586-597)
// through three heap images (A, B: 8 B per block; modulation: 1 B per pixel).  Here:
//
//   icamd_pvrtc2_morph_kernel   one 8x4 block per lane, raster order (a wave reads 64 x 32 B = 2 KiB contiguous
//                               per pixel row): GetExtremesFast + ApplyColorChannelReduction -> the block's two
//                               reduced colours, 8 B per block, into a workspace (0.25 B/px).           (Morph)
//   icamd_pvrtc2_encode_kernel  one lane per vertical strip of 8 blocks of one block column; consecutive lanes =
//                               consecutive columns (a wave reads 2 KiB contiguous per pixel row, its 8-byte block
//                               stores land at the blocks' Z-order slots, pvrtc.cc:551-580).  The lane walks down
//                               its 32 pixel rows plus the one below the strip, re-reading the pixels and the
//                               pixel right of each row, keeps a rolling 3x3 neighbourhood of reduced colours,
//                               and computes every modulation value its mode decisions depend on itself
//                               (pvrtc.cc:416-429 looks one pixel right / down): 36 + 1 per block instead of the
//                               32 a block owns.  No barrier, no inter-lane exchange of modulation values: the
//                               reference's 1 B/px modulation image never exists.  The finished 8-byte blocks are
//                               parked in a per-wave LDS tile laid out in Z order and written out as 512-byte
//                               contiguous runs (16 B per lane) at the end of the strip.   (Modulate + Encode)
//
// A first fused version (one 320-lane workgroup per 16x16-block tile, A/B and modulation edges exchanged
// through an LDS halo, two barriers) measured 1.0-1.25 ms per 16 x 4096^2 launch: with 50 KiB of LDS and a
// fifth "ring" wave per workgroup it was latency/occupancy-bound (SQ_WAIT_ANY 50 %).  Recomputing 12 halo values
// per lane (+37 % modulation work) removes every dependency between lanes.
// Launch order is morph(all images of a group) then encode(same group), groups as large as the workspace allows
// (see launch_pvrtc2); toroidal wrap (pvrtc.cc:216-227,416-423) is applied to block / pixel coordinates.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ic_launch.h"
#include "ic_amd.h"
#include "pvrtc_block.h"

namespace icamd {

namespace {

constexpr int kMorphLanes = 256;
constexpr int kEncodeLanes = 256;
#ifndef ICAMD_PVRTC_FULL_CHIP_WAVES
#define ICAMD_PVRTC_FULL_CHIP_WAVES 2u
#endif
constexpr uint32_t kFullChipLanes = 256u * 4u * 64u * ICAMD_PVRTC_FULL_CHIP_WAVES;  // two waves on each of the 1 024 SIMDs

__device__ __forceinline__ void load_block32(const uint32_t *p, uint32_t n, uint32_t px[32]) {
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const U4 v0 = load_stream(reinterpret_cast<const U4 *>(p + (size_t)y * n));      // non-temporal: streamed once
    const U4 v1 = load_stream(reinterpret_cast<const U4 *>(p + (size_t)y * n + 4));
    px[8 * y + 0] = v0.x; px[8 * y + 1] = v0.y; px[8 * y + 2] = v0.z; px[8 * y + 3] = v0.w;
    px[8 * y + 4] = v1.x; px[8 * y + 5] = v1.y; px[8 * y + 6] = v1.z; px[8 * y + 7] = v1.w;
  }
}

// Per-thread grow-only device buffer.  Reuse from a different stream waits (on the device) for the previous user.
// NOT usable under stream capture: a captured graph would bake in a pointer that a later, larger call frees, and
// replays would bypass the event bookkeeping -- callers that capture provide their own workspace
// (icamd_pvrtc2_set_workspace), one per graph.
struct Workspace {
  int device = -1;
  void *ptr = nullptr;
  size_t capacity = 0;
  hipEvent_t done = nullptr;
  hipStream_t last_stream = nullptr;
  bool in_flight = false;
  void *user_ptr = nullptr;  // caller-owned workspace (thread-local override); no allocation, no events
  size_t user_bytes = 0;
  bool using_user = false;   // between acquire and release: the caller-owned workspace is the one in use
  // No destructor: a Workspace is never destroyed (TlsWorkspace below parks it in a leaked process-wide list when its
  // thread ends) -- no HIP call may run from a thread_local destructor, where the runtime can already be gone.
  // internal_only: the library's own staging-stream calls (icamd_compress of a PVRTC image) never borrow the
  // caller-owned workspace -- that one belongs to the caller's streams and graphs.
  hipError_t acquire(size_t bytes, hipStream_t stream, void **out, bool internal_only = false) {
    using_user = user_ptr && !internal_only;
    if (using_user) {
      if (bytes > user_bytes) return hipErrorInvalidValue;
      *out = user_ptr;
      return hipSuccess;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return hipErrorStreamCaptureUnsupported;
    (void)hipGetLastError();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != device || bytes > capacity) {
      if (ptr) (void)hipFree(ptr);  // hipFree waits for outstanding work on the buffer
      ptr = nullptr; capacity = 0; in_flight = false; device = dev;
      if (done) { (void)hipEventDestroy(done); done = nullptr; }
      e = hipMalloc(&ptr, bytes);
      if (e != hipSuccess) return e;
      capacity = bytes;
    }
    if (!done) {
      e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    if (in_flight && last_stream != stream) {
      e = hipStreamWaitEvent(stream, done, 0);
      if (e != hipSuccess) return e;
    }
    *out = ptr;
    return hipSuccess;
  }
  hipError_t release(hipStream_t stream) {
    if (using_user) return hipSuccess;
    last_stream = stream;
    in_flight = true;
    return hipEventRecord(done, stream);
  }
};
// Parked workspaces of threads that ended (leaked on purpose at process exit, re-used by later threads).
std::mutex &g_ws_mutex = *new std::mutex();
std::vector<Workspace *> &g_ws_parked = *new std::vector<Workspace *>();
// Two slots per thread: a caller that alternates between two streams (icamd_encode_batch_sharded_device's workers) selects
// the slot that goes with the stream (pvrtc2_select_workspace), so that a run on one stream does not wait for the previous
// run on the other through the shared buffer's event.  Slot 0 is the default and the one a caller-owned
// workspace (icamd_pvrtc2_set_workspace) overrides.
struct TlsWorkspace {
  Workspace *p[2] = { nullptr, nullptr };
  int slot = 0;
  Workspace &get() {
    Workspace *&w = p[slot];
    if (!w) {
      // (created on first use only: a thread whose launches all take the one-pass kernel never owns a workspace, and
      // slot 1 exists only in threads that really alternate between two streams)
      int dev = -1;
      (void)hipGetDevice(&dev);
      std::lock_guard<std::mutex> lock(g_ws_mutex);
      if (!g_ws_parked.empty()) {
        // prefer a parked workspace of THIS device: taking another device's costs a hipFree + hipMalloc in acquire()
        size_t pick = g_ws_parked.size() - 1;
        for (size_t i = 0; i < g_ws_parked.size(); ++i)
          if (g_ws_parked[i]->device == dev) { pick = i; break; }
        w = g_ws_parked[pick];
        g_ws_parked.erase(g_ws_parked.begin() + (long)pick);
      } else {
        w = new Workspace();
      }
    }
    return *w;
  }
  ~TlsWorkspace() {
    for (Workspace *w : p) {
      if (!w) continue;
      w->user_ptr = nullptr;  // the override was this thread's
      w->user_bytes = 0;
      try {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        g_ws_parked.push_back(w);  // no HIP call here
      } catch (...) {
        // (the list could not grow: this workspace stays allocated until the process ends; a destructor must not throw)
      }
    }
  }
};
thread_local TlsWorkspace g_tls_workspace;

// images of one launch pair: as many as 4 GiB of pixels (and the 32-bit block index) allow
uint64_t pvrtc_group(uint32_t size, uint32_t n_images) {
  const uint64_t image_bytes = (uint64_t)size * size * 4u;
  uint64_t group = image_bytes ? (4096ull << 20) / image_bytes : 1;
  if (group < 1) group = 1;
  if (group > n_images) group = n_images;
  return group;
}

}  // namespace

struct PvrtcLaunch {
  const uint8_t *src;
  uint8_t *dst;
  uint2 *ab;                // workspace: reduced (A, B) colours, [image][by][bx]
  uint64_t src_image_stride, dst_image_stride;
  uint32_t size, log2_bw;   // width == height; log2(width / 8)
  uint32_t log2_bpi;        // log2(blocks per image)
  uint32_t log2_strip;      // encode kernel: log2(blocks per lane), a vertical strip of one block column
  uint32_t total_blocks;    // blocks per image * images in this launch
  uint32_t total_strips;    // encoded blocks of this launch >> log2_strip
  // Encoded region of each image: the blocks whose Z-order index lies in [z_first, z_first + 2^log2_rblocks), a
  // rectangle of 2^log2_rw x 2^(log2_rblocks - log2_rw) blocks at (rx0, ry0).  Whole image: 0, 0, log2_bw, log2_bpi, 0.
  uint32_t rx0, ry0, log2_rw, log2_rblocks, z_first;
  uint32_t stage_stores;    // encode kernel: park the strip's blocks in LDS and write them out in Z-order runs
  // One-pass kernel, halo form (r06; textures wider than one workgroup, regions): a workgroup covers 2^log2_wgc block columns
  // of the 2^log2_rw the rectangle is wide
  uint32_t log2_wgc = 0;
};

// Morph: kMorphBlocksPerLane blocks per lane, software-pipelined with two pixel buffers -- the loads of the next
// block are in flight while the current one is reduced, so a wave always has 8 KiB outstanding and the HBM stream
// does not drain during its ~700-instruction compute phases (5 waves/SIMD: the 32 KiB stash caps the occupancy).
constexpr int kMorphBlocksPerLane = 4;

// LDS of a morph workgroup: the per-lane pixel stash for index lookups, 8 planes x 256 lanes x 16 B = 32 KiB
constexpr uint32_t kLdsDwords = 8 * kMorphLanes * 4;

template <int kBlocksPerLane, bool DENSE = false>
__device__ __forceinline__ void pvrtc2_morph(const PvrtcLaunch &L, uint32_t wg, uint32_t *lds) {
  const uint32_t k0 = wg * (kMorphLanes * kBlocksPerLane) + threadIdx.x;
  const uint32_t n = L.size, log2_n = L.log2_bw + 3u, bpi_mask = (1u << L.log2_bpi) - 1u, bw_mask = (1u << L.log2_bw) - 1u;
  const uint32_t lane = threadIdx.x & 63u;
  Stash32 stash;
  stash.base = lds + threadIdx.x * 4u;  // [plane][lane][4 dwords]
  stash.row_dwords = kMorphLanes * 4;
  stash.filled = DENSE;
  // DENSE (block grids at least 64 columns wide: a wave's 64 blocks are one 2 KiB run per pixel row): lane l fetches
  // the l-th and the (64 + l)-th 16-byte piece of that run -- two fully dense 1 KiB accesses per row instead of two
  // half-dense 2 KiB ones -- and the pieces are handed to the lanes that own them through the LDS stash, which the
  // index lookups need filled anyway (the owner reads its 32 pixels back with 8 ds_read_b128).  The kernel is bound
  // by its memory accesses, not by issue (removing a quarter of its VALU work moved it by 1 %): r02 A/B 0.533 -> 0.510 ms
  // for the whole codec on noise, 0.529 -> 0.506 on flat, 0.494 -> 0.485 on smooth (profiles/r02_ab_morph_dense.log).
  auto fetch = [&](uint32_t k, uint32_t px[32]) {
    const uint32_t kb = DENSE ? k - lane : k;  // DENSE: first block of the wave
    const uint32_t image = kb >> L.log2_bpi, b = kb & bpi_mask;
    const uint32_t by = b >> L.log2_bw, bx = b & bw_mask;
    const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
    if (DENSE) {
      // n is a power of two <= 32 768 (launch_pvrtc2 refuses more): a pixel's index in its image fits 32 bits, no multiply
      const uint32_t *row0 = img + (((by * 4u) << log2_n) + bx * 8u + lane * 4u);
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const U4 a = load_stream(reinterpret_cast<const U4 *>(row0 + (size_t)y * n));
        const U4 c = load_stream(reinterpret_cast<const U4 *>(row0 + (size_t)y * n + 256));
        px[8 * y + 0] = a.x; px[8 * y + 1] = a.y; px[8 * y + 2] = a.z; px[8 * y + 3] = a.w;
        px[8 * y + 4] = c.x; px[8 * y + 5] = c.y; px[8 * y + 6] = c.z; px[8 * y + 7] = c.w;
      }
    } else {
      load_block32(img + (((by * 4u) << log2_n) + bx * 8u), n, px);
    }
  };
  auto reduce = [&](uint32_t k, uint32_t px[32]) {
    const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)(k >> L.log2_bpi) * L.src_image_stride);
    if (DENSE) {
      // piece A of row y = columns 4 (lane & 1) .. of block (lane >> 1): plane 2 y + (lane & 1) of that owner's slot;
      // piece B the same for block 32 + (lane >> 1)
      uint32_t *wave_lds = lds + (threadIdx.x & ~63u) * 4u;
      uint32_t *slot = wave_lds + (lane & 1u) * stash.row_dwords + (lane >> 1) * 4u;
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        *reinterpret_cast<uint4 *>(slot + (2 * y) * stash.row_dwords) =
            make_uint4(px[8 * y], px[8 * y + 1], px[8 * y + 2], px[8 * y + 3]);
        *reinterpret_cast<uint4 *>(slot + (2 * y) * stash.row_dwords + 32 * 4) =
            make_uint4(px[8 * y + 4], px[8 * y + 5], px[8 * y + 6], px[8 * y + 7]);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(stash.base + q * stash.row_dwords);
        px[4 * q] = v.x; px[4 * q + 1] = v.y; px[4 * q + 2] = v.z; px[4 * q + 3] = v.w;
      }
    }
    uint32_t a, c;
    pvrtc_extremes(px, img[0], stash, a, c);
    L.ab[k] = make_uint2(channel_reduce(a, false), channel_reduce(c, true));
  };
  uint32_t buf_a[32], buf_b[32];
  if (k0 < L.total_blocks) fetch(k0, buf_a);
  if (kBlocksPerLane == 1) {
    // pin the eight loads together in front of the reduction (left alone, the optimiser sinks each row's loads
    // down to its first use: four exposed memory round trips and 188 VGPRs)
#pragma unroll
    for (int i = 0; i < 32; ++i) buf_a[i] = opaque(buf_a[i]);
    if (k0 < L.total_blocks) reduce(k0, buf_a);
    return;
  }
#pragma unroll
  for (int i = 0; i < kBlocksPerLane; i += 2) {
    const uint32_t ka = k0 + (uint32_t)i * kMorphLanes, kb = ka + kMorphLanes, kc = kb + kMorphLanes;
    if (kb < L.total_blocks) fetch(kb, buf_b);
    if (ka < L.total_blocks) reduce(ka, buf_a);
    if (i + 2 < kBlocksPerLane && kc < L.total_blocks) fetch(kc, buf_a);
    if (kb < L.total_blocks) reduce(kb, buf_b);
  }
}

extern "C" __global__ void __launch_bounds__(kMorphLanes) icamd_pvrtc2_morph_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds[kLdsDwords];
  pvrtc2_morph<kMorphBlocksPerLane>(L, blockIdx.x, lds);
}
// block grids at least 64 columns wide (textures of 512^2 and more): dense row accesses, transposed through the stash
extern "C" __global__ void __launch_bounds__(kMorphLanes) icamd_pvrtc2_morph_dense_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds[kLdsDwords];
  pvrtc2_morph<kMorphBlocksPerLane, true>(L, blockIdx.x, lds);
}
// small launches (a few textures of <= 1024^2): one block per lane, four times as many workgroups
extern "C" __global__ void __launch_bounds__(kMorphLanes) icamd_pvrtc2_morph_small_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds[kLdsDwords];
  pvrtc2_morph<1>(L, blockIdx.x, lds);
}

// Morph of a region plus its one-block ring (toroidal wrap), for encoding part of an image (one rank's share of a
// PVRTC texture sharded by Z-order range): grid = (column chunks, rows) of the (rw + 2) x (rh + 2) rectangle.  Where
// the ring wraps onto the region itself (a region as wide / tall as the image) a block is reduced twice to the same
// value.  Single image only.
extern "C" __global__ void __launch_bounds__(kMorphLanes) icamd_pvrtc2_morph_rect_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds_stash[8][kMorphLanes][4];
  const uint32_t cx = blockIdx.x * kMorphLanes + threadIdx.x, rw = 1u << L.log2_rw;
  if (cx >= rw + 2u) return;
  const uint32_t n = L.size, bw_mask = (1u << L.log2_bw) - 1u, bh_mask = (2u << L.log2_bw) - 1u;
  const uint32_t bx = (L.rx0 + cx - 1u) & bw_mask, by = (L.ry0 + blockIdx.y - 1u) & bh_mask;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src);
  uint32_t px[32];
  load_block32(img + (size_t)(by * 4u) * n + bx * 8u, n, px);
#pragma unroll
  for (int i = 0; i < 32; ++i) px[i] = opaque(px[i]);  // all eight loads in flight before the reduction starts
  Stash32 stash;
  stash.base = &lds_stash[0][threadIdx.x][0];
  stash.row_dwords = kMorphLanes * 4;
  uint32_t a, c;
  pvrtc_extremes(px, img[0], stash, a, c);
  L.ab[(by << L.log2_bw) + bx] = make_uint2(channel_reduce(a, false), channel_reduce(c, true));
}

// LDS tile of the staged stores: a wave's lanes are grouped in chunks of 2^sb consecutive block columns (sb =
// log2_strip); a chunk's 2^sb x 2^sb blocks are 2^(2 sb) consecutive Z-order slots (x in the odd bits, y in the even
// bits, pvrtc.cc:80-86) = one contiguous run of the output.  Chunk stride padded by two slots against bank conflicts.
constexpr uint32_t kStageSlots = (kEncodeLanes / 8) * 66;  // sb = 3: 32 chunks x (64 + 2) slots of 8 bytes

// EXCHANGE (regions at least 64 block columns wide, i.e. every whole texture of 512^2 and more): a lane does not
// compute the modulation values right of its blocks (4 of the 37 a block costs) -- its right-hand neighbour LANE owns
// that block column and walks the same pixel rows in lock-step, so the four values of a block arrive with one
// one-lane shuffle when the block is finished.  Only the last lane of a wave has no neighbour in the wave: the values
// right of ITS strip (4 << log2_strip pixel rows) are computed up front by the workgroup, one pixel per lane
// (pvrtc_left_edge_mod), parked in 128 bytes of LDS, and the single barrier of the kernel follows -- before the strips
// start, where it costs nothing.  -9 % executed instructions per block (r03).
// DMA (with EXCHANGE): the strip's pixel rows are fetched by LDS-DMA (global_load_lds_dwordx4, no destination VGPRs) into a
// per-wave ring of kRowRing row slots, three rows ahead of their use -- three times the bytes in flight of the register
// path, whose depth the 168-VGPR budget caps at one row (profiles/r03_ab_pvrtc_prefetch.log: 15 % of the kernel is HBM
// latency).  A row slot is [half][lane][4 pixels] (2 KiB), read back by its lane with two conflict-free ds_read_b128.
// The waits are counted by hand (s_waitcnt vmcnt(4): everything but the two youngest rows has landed) and the reads are
// inline asm, because hipcc drains ALL outstanding loads (vmcnt(0)) before any LDS read it can see while a DMA is in flight.
constexpr uint32_t kRowRing = 4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <bool EXCHANGE, bool DMA = false>
__device__ __forceinline__ void pvrtc2_encode(const PvrtcLaunch &L, uint32_t wg, uint32_t *lds, uint32_t *lds_edge,
                                              uint32_t *lds_rows = nullptr) {
  uint2 *lds_out = reinterpret_cast<uint2 *>(lds);
  const uint32_t k = wg * kEncodeLanes + threadIdx.x;
  const uint32_t n = L.size, log2_n = L.log2_bw + 3u, bw_mask = (1u << L.log2_bw) - 1u, bh_mask = (2u << L.log2_bw) - 1u;
  const uint32_t sb = L.log2_strip;
  const uint32_t log2_spi = L.log2_rblocks - sb;  // log2(strips per image (region))
  // consecutive lanes = consecutive block columns of one strip row: a wave reads 2 KiB contiguous per pixel row
  auto locate = [&](uint32_t kk, uint32_t &image, uint32_t &bx, uint32_t &by0) {
    image = kk >> log2_spi;
    const uint32_t s = kk & ((1u << log2_spi) - 1u);
    bx = L.rx0 + (s & ((1u << L.log2_rw) - 1u));
    by0 = L.ry0 + ((s >> L.log2_rw) << sb);
  };
  const uint32_t chunk_slots = (1u << (2u * sb)) + 2u;
  if (EXCHANGE) {
    // lane t < 4 * rows: pixel row r = t % rows of the strip of wave w = t / rows, in the block column right of that
    // wave's last lane; byte r of the wave's 32-byte LDS row = its modulation value
    const uint32_t log2_rows = 2u + sb, t = threadIdx.x;
    if (t < (4u << log2_rows)) {
      const uint32_t w = t >> log2_rows, r = t & ((1u << log2_rows) - 1u);
      const uint32_t kk = wg * kEncodeLanes + 64u * w + 63u;
      if (kk < L.total_strips) {
        uint32_t image, bx, by0;
        locate(kk, image, bx, by0);
        const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
        const uint2 *ab = L.ab + ((size_t)image << L.log2_bpi);
        const uint32_t cx = (bx + 1u) & bw_mask, y_in = r & 3u;
        const uint32_t by = (by0 + (r >> 2)) & bh_mask;
        const uint32_t up = (y_in < 2u ? by - 1u : by) & bh_mask, dn = (up + 1u) & bh_mask;
        const uint2 ul = ab[(up << L.log2_bw) + bx], uc = ab[(up << L.log2_bw) + cx];
        const uint2 ll = ab[(dn << L.log2_bw) + bx], lc = ab[(dn << L.log2_bw) + cx];
        const uint32_t pixel = img[(((by0 * 4u + r) & (n - 1u)) << log2_n) + cx * 8u];
        const PvrtcColors c_ul = { ul.x, ul.y }, c_uc = { uc.x, uc.y }, c_ll = { ll.x, ll.y }, c_lc = { lc.x, lc.y };
        reinterpret_cast<uint8_t *>(lds_edge)[w * 32u + r] = (uint8_t)pvrtc_left_edge_mod(pixel, y_in, c_ul, c_uc, c_ll, c_lc);
      }
    }
    __syncthreads();
  }
  if (k < L.total_strips) {
    uint32_t image, bx, by0;
    locate(k, image, bx, by0);
    const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
    const uint2 *ab = L.ab + ((size_t)image << L.log2_bpi);
    uint2 *dst = reinterpret_cast<uint2 *>(L.dst + (size_t)image * L.dst_image_stride);
    const uint32_t xl = (bx - 1u) & bw_mask, xr = (bx + 1u) & bw_mask;

    // DMA: this wave's ring (all 64 lanes of a wave share the strip row and the image when the region is >= 64 columns wide)
    // (r04) the wave's index is made a scalar explicitly: the ring slot (M0) is then computed on the scalar unit instead of
    // two v_readfirstlane + vector adds per pixel row (encode kernel 268.0 -> 264.4 us, profiles/r04_ab_pvrtc_scalar_ring.log).
    // The same for the strip's row (readfirstlane of by0 * 4, -DICAMD_PVRTC_SCALAR_ROW) produced wrong first blocks of strips
    // in r04.  r05 found why: not the SGPR operands, but the colour loads' wait (load_colours below) -- that build merely
    // scheduled differently.  With the wait fixed the scalar-row build is bit-exact; it stays an A/B option (the kernel now
    // only serves launches the one-pass kernel does not take).
    const uint32_t wave_s = DMA ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : threadIdx.x >> 6;
    lds_u32 *ring = DMA ? (lds_u32 *)(lds_rows + wave_s * (kRowRing * 512u)) : nullptr;
    const uint32_t ring_lane_byte = DMA ? (uint32_t)(uintptr_t)ring + (threadIdx.x & 63u) * 16u : 0u;
    // pixel row r of the strip -> slot r % kRowRing.  The walk ends at row 4 K (the first row below the strip); the three
    // requests past it keep the wait counts uniform but re-fetch row 4 K (an L2 hit) instead of 3 / 32 more HBM bytes.
    const uint32_t last_row = 4u << sb;
#if defined(ICAMD_PVRTC_SCALAR_ROW)  // r04's failing experiment, kept reproducible (profiles/r04_ab_pvrtc_scalar_ring.log)
    const uint32_t by0_dma = (uint32_t)__builtin_amdgcn_readfirstlane((int)by0);
#else
    const uint32_t by0_dma = by0;
#endif
    auto dma_row = [&](uint32_t r) {
      const uint32_t *q = img + ((((by0_dma * 4u + (r < last_row ? r : last_row)) & (n - 1u)) << log2_n) + bx * 8u);
      lds_u32 *slot = ring + (r & (kRowRing - 1u)) * 512u;
      __builtin_amdgcn_global_load_lds(q, slot, 16, 0, 0);             // lane l: its first four pixels -> slot + 16 l
      __builtin_amdgcn_global_load_lds(q + 4, slot + 256, 16, 0, 0);   // ... its last four -> slot + 1024 + 16 l
    };
    if (DMA) {
#pragma unroll
      for (uint32_t r = 0; r + 1 < kRowRing; ++r) dma_row(r);
    }
#if defined(ICAMD_PVRTC_FUSION_PROBE)
    // r04 measurement aid, NEVER in the shipped library: what would one pass over the pixels cost in situ?  Every pixel row the
    // strip consumes also feeds the morph reduction of its block (the five axes' first-min / first-max keys); every fourth row
    // the block is finished like pvrtc_extremes does it (ten data-dependent pixel look-ups -- here from the row ring --, five
    // v_sad_u8, the best-axis scan, the brightness order, two channel reductions).  The results are only kept alive, not used:
    // the encode kernel's output is unchanged, its time shows the VALU / register price of fusing the morph in.
    uint32_t pr_lmin = 0xffffffffu, pr_lmax = 0u, pr_rbmin = 0xffffffffu, pr_rbmax = 0u, pr_gamin = 0xffffffffu, pr_gamax = 0u, pr_acc = 0u;
    auto probe_row = [&](uint32_t r, const uint32_t *pixels) {
      const uint32_t q = r & 3u;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const uint32_t c = pixels[x], i = (uint32_t)(x & 3);
        const uint32_t idx4 = (8u * q + (uint32_t)(x & 4)) * 0x01010101u + 0x03020100u;
        const uint32_t up1 = 31u - 2u * (8u * q + (uint32_t)x), up = up1 * 0x00010001u;
        const uint32_t kl = perm(udot4(c, 0x001c964du, 0u), idx4, 0x0c0c0500u | i);
        const uint32_t k_rb = perm(c, idx4, 0x06000400u | i | i << 16), k_ga = perm(c, idx4, 0x07000500u | i | i << 16);
        pr_rbmin = pk_min_u16(pr_rbmin, k_rb); pr_gamin = pk_min_u16(pr_gamin, k_ga);
        pr_rbmax = pk_max_u16(pr_rbmax, k_rb + up); pr_gamax = pk_max_u16(pr_gamax, k_ga + up);
        pr_lmin = umin(pr_lmin, kl); pr_lmax = umax(pr_lmax, kl + up1);
      }
      if (q == 3u) {
        const uint32_t kmin[5] = { pr_lmin, pr_rbmin & 0xffffu, pr_gamin & 0xffffu, pr_rbmin >> 16, pr_gamin >> 16 };
        const uint32_t kmax[5] = { pr_lmax, pr_rbmax & 0xffffu, pr_gamax & 0xffffu, pr_rbmax >> 16, pr_gamax >> 16 };
        uint32_t best_diff = 0, best_lo = 0, best_hi = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
          uint32_t lo, hi;
          const uint32_t i0 = kmin[a] & 31u, i1 = 31u - (kmax[a] & 31u);
          const uint32_t a0 = ring_lane_byte + (i0 >> 3) * 2048u + ((i0 >> 2) & 1u) * 1024u + (i0 & 3u) * 4u;
          const uint32_t a1 = ring_lane_byte + (i1 >> 3) * 2048u + ((i1 >> 2) & 1u) * 1024u + (i1 & 3u) * 4u;
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
          hi = (kmax[a] >> 8) == 0u ? img[0] * 0u + lo : hi;
          const uint32_t d = sad_u8(lo, hi, 0u);
          const bool better = (a == 0) || d > best_diff;
          best_lo = better ? lo : best_lo; best_hi = better ? hi : best_hi; best_diff = better ? d : best_diff;
        }
        const bool swap = udot4(best_hi, 0x01010101u, 0u) < udot4(best_lo, 0x01010101u, 0u);
        pr_acc ^= channel_reduce(swap ? best_hi : best_lo, false) + channel_reduce(swap ? best_lo : best_hi, true);
        pr_lmin = pr_rbmin = pr_gamin = 0xffffffffu; pr_lmax = pr_rbmax = pr_gamax = 0u;
      }
    };
#endif
    auto load_px = [&](uint32_t r, uint32_t *pixels, uint32_t *right_px) {
      if (DMA) {
        // rows r + 1 and r + 2 (four DMA instructions) may still be in flight; row r and everything older has landed
        uint4 v0, v1;
        const uint32_t addr = ring_lane_byte + (r & (kRowRing - 1u)) * 2048u;
        asm volatile("s_waitcnt vmcnt(4)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1) : "v"(addr) : "memory");
        pixels[0] = v0.x; pixels[1] = v0.y; pixels[2] = v0.z; pixels[3] = v0.w;
        pixels[4] = v1.x; pixels[5] = v1.y; pixels[6] = v1.z; pixels[7] = v1.w;
        // the slot of row r - 1 (read one call ago, its ds_reads long complete) takes row r + 3
        dma_row(r + kRowRing - 1u);
#if defined(ICAMD_PVRTC_FUSION_PROBE)
        probe_row(r, pixels);
#endif
        return;
      }
      // n = 2^log2_n <= 32 768: the pixel index fits 32 bits -- a shift, not a 64-bit multiply
      const uint32_t *q = img + (((by0 * 4u + r) & (n - 1u)) << log2_n);
      // (plain loads: the neighbouring lanes' / rows' re-use of these lines wants the cache -- non-temporal was 4 % slower)
      const uint4 v0 = *reinterpret_cast<const uint4 *>(q + bx * 8u), v1 = *reinterpret_cast<const uint4 *>(q + bx * 8u + 4);
      pixels[0] = v0.x; pixels[1] = v0.y; pixels[2] = v0.z; pixels[3] = v0.w;
      pixels[4] = v1.x; pixels[5] = v1.y; pixels[6] = v1.z; pixels[7] = v1.w;
      if (!EXCHANGE) *right_px = q[xr * 8u];
    };
    PvrtcColors *requested = nullptr;  // DMA: the colour row whose loads hipcc does not know about
    auto load_colours = [&](int j, PvrtcColors c[3]) {
      const uint2 *row = ab + (((by0 + (uint32_t)j) & bh_mask) << L.log2_bw);
      if (DMA) {
        // Loads hipcc does not know about, so that it does not drain the row ring (vmcnt(0)) where their results are used:
        // colour row j >= 2 is requested a whole block (four load_px calls, eight younger DMA instructions) before its
        // first use, so the vmcnt(4) of those calls has retired it; rows -1, 0 and 1 are used sooner and are waited for here.
        uint2 l, m, r;
        asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %4, off\n\tglobal_load_dwordx2 %2, %5, off"
                     : "=&v"(l), "=&v"(m), "=&v"(r) : "v"(row + xl), "v"(row + bx), "v"(row + xr) : "memory");
        // The wait must CARRY the loaded registers ("+v"): an asm volatile without operands only orders against other volatile
        // asm and memory operations, and the scheduler is free to place a pure-VALU use of l / m / r between the load asm and a
        // wait that does not name them.  That is what broke r04's scalar-row build (uses of the first colour row scheduled in
        // front of the wait: wrong first blocks of strips, deterministic per build -- profiles/r04_ab_pvrtc_scalar_ring.log;
        // root-caused in r05) and what the shipped build escaped only by scheduling luck.
        if (j <= 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(l), "+v"(m), "+v"(r) :: "memory");
        requested = c;
        c[0].a = l.x; c[0].b = l.y;
        c[1].a = m.x; c[1].b = m.y;
        c[2].a = r.x; c[2].b = r.y;
        return;
      }
      const uint2 l = row[xl], m = row[bx], r = row[xr];
      c[0].a = l.x; c[0].b = l.y;
      c[1].a = m.x; c[1].b = m.y;
      c[2].a = r.x; c[2].b = r.y;
    };
    const uint32_t zx = spread_bits16(bx) << 1;  // pvrtc.cc:80-86: x in the odd bits, y in the even bits
    // this lane's slots in the wave's LDS tile: chunk = lane >> sb, x inside the chunk = lane & (2^sb - 1)
    const uint32_t sub = (1u << sb) - 1u;
    uint2 *slot0 = lds_out + (threadIdx.x >> sb) * chunk_slots + (spread_bits16(threadIdx.x & sub) << 1);
    auto store = [&](uint32_t j, uint32_t data, bool one_bpp, const PvrtcColors &own) {
      const uint2 v = make_uint2(data, pvrtc_pack_colors(own.a, own.b, one_bpp));
      if (DMA) {  // (stores hipcc can see are merged into one flat store, which drains the row ring)
        // A block is stored after the third load_px of its walk step, whose vmcnt(4) has retired the colour loads issued
        // at the top of the step: from here on their results may be used, and no use can be scheduled before this point.
        asm volatile("" : "+v"(requested[0].a), "+v"(requested[0].b), "+v"(requested[1].a), "+v"(requested[1].b),
                          "+v"(requested[2].a), "+v"(requested[2].b));
        if (L.stage_stores)
          asm volatile("ds_write_b64 %0, %1" :: "v"((uint32_t)(uintptr_t)(lds_u32 *)(uint32_t *)(slot0 + spread_bits16(j))), "v"(v) : "memory");
        else
          asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst + ((zx | spread_bits16(by0 + j)) - L.z_first)), "v"(v) : "memory");
        return;
      }
      if (L.stage_stores) slot0[spread_bits16(j)] = v;
      else dst[(zx | spread_bits16(by0 + j)) - L.z_first] = v;
    };
    // column-0 values of the block to the right: lane + 1's own; the wave's last lane reads the precomputed ones
    const uint32_t *edge_row = lds_edge + (threadIdx.x >> 6) * 8u;
    const bool wave_end = (threadIdx.x & 63u) == 63u;
    auto right_of = [&](uint32_t j, uint32_t col0) -> uint32_t {
      const uint32_t from_lane = (uint32_t)__shfl_down((int)col0, 1);
      if (DMA) {  // an LDS read hipcc can see would drain the row ring (vmcnt(0)); the edge values were written before the barrier
        uint32_t e;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"((uint32_t)(uintptr_t)(lds_u32 *)(edge_row + j)) : "memory");
        return wave_end ? e : from_lane;
      }
      return wave_end ? edge_row[j] : from_lane;
    };
    pvrtc_encode_strip<EXCHANGE>(1u << sb, load_px, load_colours, store, right_of);
#if defined(ICAMD_PVRTC_FUSION_PROBE)
    asm volatile("" :: "v"(pr_acc));
#endif
  }
  if (!L.stage_stores) return;
  // Write-out: every lane stores 16 bytes (two Z-adjacent blocks) per round; 32 lanes cover one 512-byte run (sb = 3).
  // Only lanes of this wave wrote the slots it reads (chunks never straddle a wave), and a wave's LDS operations
  // complete in program order, so no barrier is needed.
  const uint32_t wave_lane0 = threadIdx.x & ~63u, lane = threadIdx.x & 63u;
  const uint32_t pairs_per_chunk = 1u << (2u * sb - 1u), log2_ppc = 2u * sb - 1u;
  for (uint32_t t = 0; t < (1u << (sb - 1u)); ++t) {
    const uint32_t pair = t * 64u + lane;                 // index among the wave's 2^(5 + sb) slot pairs
    const uint32_t chunk = pair >> log2_ppc, within = (pair & (pairs_per_chunk - 1u)) << 1;
    const uint32_t src_lane = wave_lane0 + (chunk << sb); // first lane of the chunk that produced these slots
    const uint32_t kk = wg * kEncodeLanes + src_lane;
    if (kk >= L.total_strips) continue;
    uint32_t image, bx, by0;
    locate(kk, image, bx, by0);
    const uint2 *p = lds_out + (src_lane >> sb) * chunk_slots + within;
    const uint2 a = p[0], b = p[1];
    uint2 *dst = reinterpret_cast<uint2 *>(L.dst + (size_t)image * L.dst_image_stride);
    const uint32_t z = ((spread_bits16(bx) << 1) | spread_bits16(by0)) - L.z_first + within;
    store_stream16(dst + z, a.x, a.y, b.x, b.y);
  }
}

extern "C" __global__ void __launch_bounds__(kEncodeLanes) icamd_pvrtc2_encode_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds[kStageSlots * 2];
  __shared__ uint32_t lds_edge[4 * 8];  // per wave: 32 bytes = the values right of its last lane's strip
#if !defined(ICAMD_PVRTC_NO_ROW_DMA)  // (the register-path build, for A/B runs: -DICAMD_PVRTC_NO_ROW_DMA)
  __shared__ __attribute__((aligned(16))) uint32_t lds_rows[4 * kRowRing * 512];  // per wave: kRowRing row slots of 2 KiB
#if defined(ICAMD_PVRTC_XCD_REMAP)
  const uint32_t nwg = gridDim.x, wg = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
  pvrtc2_encode<true, true>(L, wg, lds, lds_edge, lds_rows);
#else
  pvrtc2_encode<true, true>(L, blockIdx.x, lds, lds_edge, lds_rows);
#endif
#else
  pvrtc2_encode<true>(L, blockIdx.x, lds, lds_edge);
#endif
}
// regions fewer than 64 block columns wide (textures below 512^2): a wave holds several strip rows, every lane computes
// the values right of its blocks itself
extern "C" __global__ void __launch_bounds__(kEncodeLanes) icamd_pvrtc2_encode_narrow_kernel(PvrtcLaunch L) {
  __shared__ uint32_t lds[kStageSlots * 2];
  pvrtc2_encode<false>(L, blockIdx.x, lds, nullptr);
}

// ---- one-pass kernel (r05) ---------------------------------------------------------------------------------------------
// ONE read of the pixels, no workspace: the lane that encodes a block column also morphs it (pvrtc_onepass_strip).
//   * a workgroup is one WHOLE block row of the image wide (bw = 64 ... 512 lanes, 1 ... 8 waves: textures of 512^2 ... 4096^2)
//     and K = 4 ... 64 blocks tall: the toroidal wrap (pvrtc.cc:216-227) stays inside the workgroup, so NO block column is
//     morphed twice; vertically a strip morphs K + 2 block rows for K encoded (K = 32: + 6 % of the morph, which is a third
//     of the work);
//   * pixel rows arrive by LDS-DMA in a per-wave ring of 8 rows (16 KiB): row m is morphed when it lands and modulated five
//     rows later from the same slot, which row m + 3 then takes -- three rows (6 KiB per wave) in flight, waits counted by
//     hand (every tick issues exactly one row and waits with vmcnt(4); stores only make that wait stricter);
//   * a block's colours go to the neighbour lanes by DPP wave shifts; the two edge lanes of every wave go through 32 bytes
//     of LDS per wave and parity and ONE s_barrier per block row, which also carries the column-0 modulation values the
//     last lane of a wave needs from the first lane of the next (the reason blocks are finished one segment late);
//   * finished blocks are parked in a per-wave LDS tile (4 block rows x 64 columns, Z order) and leave as 128-byte runs.
// LDS: waves x (16 KiB ring + 2.25 KiB tile) + 64 B x waves = 150 KiB for 8 waves: one workgroup per CU, 2 waves per SIMD,
// which is what the ~200 VGPRs of the fused walk allow anyway.  Everything that touches LDS or memory inside the walk is
// inline asm or a DMA builtin: hipcc drains the ring (vmcnt(0)) in front of any LDS access it can see.
typedef uint32_t icamd_u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOnePassRing = 8;
constexpr uint32_t kOnePassChunkSlots = 18;                                  // 4 x 4 blocks in Z order + 2 (16-byte aligned stride)
constexpr uint32_t kOnePassWaveDwords = kOnePassRing * 512u + 16u * kOnePassChunkSlots * 2u;  // ring + tile: 18 688 bytes
constexpr uint32_t kOnePassXchDwords = 8;                                    // per wave and parity: [lo.a lo.b col0 - | hi.a hi.b - -]

// lane i <- v of lane i - 1 (wave_shr:1); lane 0, which has no source lane, keeps `edge` (the DPP "old" operand: no select after it)
__device__ __forceinline__ uint32_t dpp_from_lower_lane(uint32_t v, uint32_t edge) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);
}
// lane i <- v of lane i + 1 (wave_shl:1); lane 63 keeps `edge`
__device__ __forceinline__ uint32_t dpp_from_upper_lane(uint32_t v, uint32_t edge) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false);
}

// HALO (r06): the workgroup covers 2^log2_wgc block columns of a WIDER rectangle -- a whole texture of 8192^2 and more, whose block
// row does not fit 512 lanes, or a region of one texture (icamd_pvrtc2_encode_region_device) -- so the wrap no longer closes inside
// it.  What the first and the last lane need from outside: the colours of the block column beyond the edge (every segment) and,
// for the last lane, the column-0 modulation values of the blocks right of its own (the term sum_y |m(7, y) - m(8, y)|,
// pvrtc.cc:426-429).  A PROLOGUE reduces those two columns (and the workgroup's own last one) itself -- 3 (K + 3) blocks, one per
// lane, with the pair path's routine and the idle row ring as its stash -- and computes the 4 K column-0 values from them, one
// pixel per lane (pvrtc_left_edge_mod, the pair path's routine for the same job): no pre-pass, no scratch memory, so these
// launches too can be captured without a workspace -- and from then on the edge waves read
// their neighbours' records at the same point, with the same instructions, as every other wave reads the next wave's: the walk, its
// ring protocol and its instruction count are unchanged.  Output and strip coordinates are LOCAL to the rectangle (an aligned
// Z-order range is the Z order of its local coordinates), pixels are addressed in the texture.
template <bool HALO>
__device__ __forceinline__ void pvrtc2_onepass_body(const PvrtcLaunch &L, uint32_t *lds_dyn) {
  const uint32_t n = L.size, log2_n = L.log2_bw + 3u;
  const uint32_t sb = L.log2_strip, K = 1u << sb;
  const uint32_t log2_rh = L.log2_rblocks - L.log2_rw;
  const uint32_t log2_spi = (HALO ? log2_rh : L.log2_bw + 1u) - sb;  // strips per image (rectangle) = its block rows >> sb
  const uint32_t log2_g = HALO ? L.log2_rw - L.log2_wgc : 0u;       // workgroups per block row of the rectangle
  const uint32_t group = HALO ? blockIdx.x & ((1u << log2_g) - 1u) : 0u, strip_id = blockIdx.x >> log2_g;
  const uint32_t image = strip_id >> log2_spi, by0_local = (strip_id & ((1u << log2_spi) - 1u)) << sb;
  const uint32_t cx0_local = group << L.log2_wgc;
  const uint32_t by0 = (HALO ? L.ry0 : 0u) + by0_local;
  const uint32_t bx_local = cx0_local + threadIdx.x, bx = (HALO ? L.rx0 : 0u) + bx_local, lane = threadIdx.x & 63u;
  const uint32_t W = blockDim.x >> 6;
  const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
  uint2 *dst = reinterpret_cast<uint2 *>(L.dst + (size_t)image * L.dst_image_stride);
  const uint32_t image0 = opaque(img[0]);  // pvrtc.cc:268-269: never-updated maxima refer to IMAGE pixel 0

  lds_u32 *ring = (lds_u32 *)(lds_dyn + wave_s * kOnePassWaveDwords);
  const uint32_t ring_lane_byte = (uint32_t)(uintptr_t)ring + lane * 16u;
  const uint32_t tile_byte = (uint32_t)(uintptr_t)ring + kOnePassRing * 2048u;
  const uint32_t xch_byte = (uint32_t)(uintptr_t)(lds_u32 *)(lds_dyn + W * kOnePassWaveDwords);
  // HALO: after the exchange slots, per segment s = -1 .. K + 1: the colours left of lane 0 (8 bytes), a 16-byte record
  // { colours right of the last lane, column-0 values of the block right of block s - 2, - } -- the layout of the exchange
  // slots -- and the colours of the workgroup's own last column (the prologue's edge values need them before the walk has them)
  uint32_t halo_left_byte = 0, halo_right_byte = 0;
  if (HALO) {
    uint32_t *tab = lds_dyn + W * (kOnePassWaveDwords + 2u * kOnePassXchDwords);
    uint2 *tl = reinterpret_cast<uint2 *>(tab);
    uint4 *tr = reinterpret_cast<uint4 *>(tab + 2u * (K + 4u));  // ((K + 4) * 8 bytes: a multiple of 16)
    uint2 *tlast = reinterpret_cast<uint2 *>(tab + 2u * (K + 4u) + 4u * (K + 3u));
    halo_left_byte = (uint32_t)(uintptr_t)(lds_u32 *)tab;
    halo_right_byte = (uint32_t)(uintptr_t)(lds_u32 *)(tab + 2u * (K + 4u));
    const uint32_t bw_mask = (1u << L.log2_bw) - 1u, bh_mask = (2u << L.log2_bw) - 1u;
    const uint32_t first_bx = bx - threadIdx.x, wgc = 1u << L.log2_wgc;
    // (1) GetExtremesFast + channel reduction of the three block columns the workgroup does not walk itself or needs early --
    // left of it, its own last one, right of it -- for the strip's K + 3 colour rows: 3 (K + 3) blocks, one per lane and round
    // (the pair path's routine; its per-lane pixel stash aliases the row ring, which is idle until the walk starts).  The same
    // columns are reduced again by the neighbouring workgroup: 3 / 512 of the morph.
    Stash32 stash;
    stash.base = lds_dyn + threadIdx.x * 4u;
    stash.row_dwords = blockDim.x * 4u;
    for (uint32_t t = threadIdx.x; t < 3u * (K + 3u); t += blockDim.x) {
      const uint32_t col = t < K + 3u ? 0u : (t < 2u * (K + 3u) ? 1u : 2u), e = t - col * (K + 3u);
      const uint32_t hx = (col == 0u ? first_bx - 1u : (col == 1u ? first_bx + wgc - 1u : first_bx + wgc)) & bw_mask;
      const uint32_t hy = (by0 + e - 1u) & bh_mask;  // segment s = e - 1 refers to block row by0 + s
      uint32_t px[32];
      load_block32(img + (size_t)(hy * 4u) * n + hx * 8u, n, px);
      uint32_t a, c;
      pvrtc_extremes(px, image0, stash, a, c);
      const uint2 v = make_uint2(channel_reduce(a, false), channel_reduce(c, true));
      if (col == 0u) tl[e] = v;
      else if (col == 1u) tlast[e] = v;
      else { tr[e].x = v.x; tr[e].y = v.y; }
    }
    __syncthreads();
    // (2) the column-0 modulation values of the blocks right of the last lane's, one pixel per lane and round
    for (uint32_t t = threadIdx.x; t < 4u * K; t += blockDim.x) {  // (a 64-lane workgroup with 64-block strips: four rounds)
      // pixel row y_in of the block right of block j (pvrtc.cc:216-227 with xw = 4)
      const uint32_t j = t >> 2, y_in = t & 3u;
      const uint32_t right_bx = (first_bx + wgc) & bw_mask;
      const uint32_t y = ((by0 + j) * 4u + y_in) & (n - 1u);
      const uint32_t pixel = img[(y << log2_n) + right_bx * 8u];
      const uint32_t up = j + (y_in < 2u ? 0u : 1u);  // entry of the upper of the two colour rows (entry e = block row e - 1)
      const uint2 ul = tlast[up], ll = tlast[up + 1u];
      const PvrtcColors cul = { ul.x, ul.y }, cuc = { tr[up].x, tr[up].y }, cll = { ll.x, ll.y }, clc = { tr[up + 1u].x, tr[up + 1u].y };
      reinterpret_cast<uint8_t *>(&tr[j + 3u].z)[y_in] = (uint8_t)pvrtc_left_edge_mod(pixel, y_in, cul, cuc, cll, clc);
    }
    __syncthreads();
  }

  // pixel row m of the strip (-4 ... 4 K + 3; requests past the end re-fetch the last row so that the wait counts stay
  // uniform) -> ring slot m mod 8, [half][lane][4 pixels]
  const int last_row = (int)(4u * K + 3u);
  auto dma_row = [&](int m) {
    const uint32_t y = (by0 * 4u + (uint32_t)(m < last_row ? m : last_row)) & (n - 1u);
    const uint32_t *q = img + ((y << log2_n) + bx * 8u);
    lds_u32 *slot = ring + ((uint32_t)m & (kOnePassRing - 1u)) * 512u;
    __builtin_amdgcn_global_load_lds(q, slot, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(q + 4, slot + 256, 16, 0, 0);
  };
  dma_row(-4); dma_row(-3); dma_row(-2);
  int cur_m = -4;
  auto tick = [&](int m, uint32_t *mp, uint32_t *ep) {
    cur_m = m;
    // rows m + 1 and m + 2 (four DMA instructions) may still be in flight; row m and everything older has landed
    const uint32_t addr_m = ring_lane_byte + ((uint32_t)m & 7u) * 2048u, addr_e = ring_lane_byte + ((uint32_t)(m + 3) & 7u) * 2048u;
    uint4 m0, m1, e0, e1;
    asm volatile("s_waitcnt vmcnt(4)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                 "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(m0), "=&v"(m1), "=&v"(e0), "=&v"(e1) : "v"(addr_m), "v"(addr_e) : "memory");
    mp[0] = m0.x; mp[1] = m0.y; mp[2] = m0.z; mp[3] = m0.w; mp[4] = m1.x; mp[5] = m1.y; mp[6] = m1.z; mp[7] = m1.w;
    ep[0] = e0.x; ep[1] = e0.y; ep[2] = e0.z; ep[3] = e0.w; ep[4] = e1.x; ep[5] = e1.y; ep[6] = e1.z; ep[7] = e1.w;
    dma_row(m + 3);  // into the slot of row m - 5, whose reads have just completed
  };
  auto lookup10 = [&](const uint32_t idx[10], uint32_t v[10]) {
    // the block whose last row is cur_m: rows cur_m - 3 ... cur_m sit in four consecutive slots (cur_m - 3 is a multiple of 4)
    const uint32_t base = ring_lane_byte + ((uint32_t)(cur_m - 3) & 7u) * 2048u;
    uint32_t a[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {  // pixel idx = 8 y + x sits at (idx >> 2) * 1024 + (idx & 3) * 4: v_bfe, v_lshl_add, v_and, v_lshl_add
      const uint32_t half_row = opaque(__builtin_amdgcn_ubfe(idx[i], 2u, 3u));  // (opaque: hipcc turns it back into shift + and + add + and_or)
      a[i] = ((idx[i] & 3u) << 2) + ((half_row << 10) + base);
    }
    asm volatile("ds_read_b32 %0, %10\n\tds_read_b32 %1, %11\n\tds_read_b32 %2, %12\n\tds_read_b32 %3, %13\n\t"
                 "ds_read_b32 %4, %14\n\tds_read_b32 %5, %15\n\tds_read_b32 %6, %16\n\tds_read_b32 %7, %17\n\t"
                 "ds_read_b32 %8, %18\n\tds_read_b32 %9, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9])
                 : "memory");
  };
  const uint32_t left_wave = (wave_s == 0u ? W : wave_s) - 1u, right_wave = wave_s + 1u == W ? 0u : wave_s + 1u;
  auto exchange = [&](int s, const PvrtcColors &own, uint32_t col0, PvrtcColors &left, PvrtcColors &right, uint32_t &right_col0) {
    const uint32_t x = xch_byte + ((uint32_t)s & 1u) * (W * kOnePassXchDwords * 4u);
    const uint32_t mine = x + wave_s * (kOnePassXchDwords * 4u);
    const uint2 c = make_uint2(own.a, own.b);
    if (lane == 0u) asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %0, %2 offset:8" :: "v"(mine), "v"(c), "v"(col0) : "memory");
    if (lane == 63u) asm volatile("ds_write_b64 %0, %1 offset:16" :: "v"(mine), "v"(c) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint2 l;
    uint4 r;
    // (HALO: the first wave's left neighbour and the last wave's right neighbour are the prologue's records of segment s)
    const uint32_t al = HALO && wave_s == 0u ? halo_left_byte + (uint32_t)(s + 1) * 8u - 16u : x + left_wave * (kOnePassXchDwords * 4u);
    const uint32_t ar = HALO && wave_s + 1u == W ? halo_right_byte + (uint32_t)(s + 1) * 16u : x + right_wave * (kOnePassXchDwords * 4u);
    asm volatile("ds_read_b64 %0, %2 offset:16\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l), "=&v"(r) : "v"(al), "v"(ar) : "memory");
    // (every lane has read the same two records; only the edge lanes keep them)
    left.a = dpp_from_lower_lane(own.a, l.x);   left.b = dpp_from_lower_lane(own.b, l.y);
    right.a = dpp_from_upper_lane(own.a, r.x);  right.b = dpp_from_upper_lane(own.b, r.y);
    right_col0 = dpp_from_upper_lane(col0, r.z);
  };
  // this lane's slot of block row jj (0..3) in the wave's tile: chunk = lane / 4, Z order inside (x odd bits, y even bits)
  const uint32_t tile_lane_byte = tile_byte + ((lane >> 2) * kOnePassChunkSlots + (((lane & 1u) | (lane & 2u) << 1) << 1)) * 8u;
  const uint32_t zx = spread_bits16(bx_local) << 1;
  // write-out of four finished block rows: 2 rounds, a lane stores 16 bytes (two Z-adjacent blocks), 8 lanes one 128-byte run
  uint32_t flush_lds[2], flush_zx[2];
#pragma unroll
  for (uint32_t t = 0; t < 2; ++t) {
    const uint32_t p = t * 64u + lane, chunk = p >> 3, within = (p & 7u) << 1;
    flush_lds[t] = tile_byte + (chunk * kOnePassChunkSlots + within) * 8u;
    flush_zx[t] = (spread_bits16(cx0_local + wave_s * 64u + 4u * chunk) << 1) + within;
  }
  auto store = [&](uint32_t j, uint32_t data, bool one_bpp, const PvrtcColors &own) {
    const uint2 v = make_uint2(data, pvrtc_pack_colors(own.a, own.b, one_bpp));
    if (!L.stage_stores) {
      asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst + (zx | spread_bits16(by0_local + j))), "v"(v) : "memory");
      return;
    }
    const uint32_t jj = j & 3u;
    asm volatile("ds_write_b64 %0, %1" :: "v"(tile_lane_byte + ((jj & 1u) | (jj & 2u) << 1) * 8u), "v"(v) : "memory");
    if (jj != 3u) return;
    const uint32_t zy = spread_bits16(by0_local + j - 3u);  // a multiple of 4: its bits do not meet `within`
#pragma unroll
    for (uint32_t t = 0; t < 2; ++t) {
      icamd_u32x4 q;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q) : "v"(flush_lds[t]) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst + (flush_zx[t] | zy)), "v"(q) : "memory");
    }
  };
  pvrtc_onepass_strip(K, image0, tick, lookup10, exchange, store);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row requests past the end of the strip
#if !defined(ICAMD_PVRTC_ONEPASS_NO_VGPR_CLAIM)
  // Claim 176 VGPRs (the walk uses ~120): LDS already limits a CU to 8 waves, and with at most two waves per SIMD the
  // dispatcher cannot stack the one- and two-wave workgroups of 512^2 / 1024^2 textures three or four deep on one SIMD while
  // another idles once workgroups retire out of step (r05 A/B, 256 x 1024^2 in 16-block strips: 0.534 -> 0.425 ms).
  asm volatile("" ::: "v175");
#endif
}

extern "C" __global__ void __launch_bounds__(512) icamd_pvrtc2_onepass_kernel(PvrtcLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  pvrtc2_onepass_body<false>(L, lds_dyn);
}
extern "C" __global__ void __launch_bounds__(512) icamd_pvrtc2_onepass_halo_kernel(PvrtcLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  pvrtc2_onepass_body<true>(L, lds_dyn);
}

// the kernel whole-texture launches of 512^2 ... 4096^2 take (launch_pvrtc2); smaller / larger / partial ones: morph + encode
const char *pvrtc2_kernel_name() { return "icamd_pvrtc2_onepass_kernel"; }

namespace {
// inverse of pvrtc_z_index on the host: x from the odd bits, y from the even bits
uint32_t compact_even_bits_host(uint32_t v) {
  v &= 0x55555555u;
  v = (v | v >> 1) & 0x33333333u;
  v = (v | v >> 2) & 0x0f0f0f0fu;
  v = (v | v >> 4) & 0x00ff00ffu;
  v = (v | v >> 8) & 0x0000ffffu;
  return v;
}
}  // namespace

// ---- path selection: one pass (whole textures of 512^2 ... 4096^2, enough of them to fill the chip) or morph + encode --------
namespace {
std::atomic<int> g_path_mode{-1};   // -1: not read yet; 0 auto, 1 always two kernels, 2 one pass wherever it is eligible
std::atomic<int> g_path_strip{-2};  // -2: not read yet; -1 auto, else log2(blocks per strip) of the one-pass kernel
// The environment is read ONCE, by whichever of the first launch and the first icamd_pvrtc2_tune comes first; a tune() then
// stores over it, so a launch on another thread can no longer read the environment on top of a tune() call.
std::once_flag g_path_env_once;
void read_path_env() {
  std::call_once(g_path_env_once, [] {
    const char *m = getenv("ICAMD_PVRTC2_PATH"), *k = getenv("ICAMD_PVRTC2_STRIP");
    int mode = 0;
    if (m && !strcmp(m, "two")) mode = 1;
    else if (m && !strcmp(m, "one")) mode = 2;
    g_path_strip.store(k && *k ? atoi(k) : -1);
    g_path_mode.store(mode);
  });
}
// Strip height of the one-pass kernel for n_images size^2 textures, or -1 where the morph + encode pair is the better choice.
// Time model, fitted on an MI355X (profiles/r05_ab_pvrtc_onepass.log, within 5 % of every measured shape from 1 x 512^2 to
// 16 x 4096^2): a workgroup of K-block strips takes 11 + 5.8 K us whatever its width (its waves walk K + 2 block rows at two
// waves per SIMD), a CU holds 8 / waves-per-workgroup of them, a launch takes ceil(workgroups / slots) such rounds; the pair
// takes 10 us + 52.6 us per million blocks.  forced >= 0: that strip height; always: the best strip height, no comparison.
int onepass_log2_strip(uint32_t log2_size, uint64_t n_images, int forced, bool always, uint32_t compute_units) {
  const uint32_t log2_bw = log2_size - 3u, log2_bh = log2_size - 2u;
  if (log2_bw < 6u || log2_bw > 9u) return -1;  // a workgroup is one block row wide: 64 ... 512 lanes
  // (a forced strip may be as tall as the texture, log2(size / 4): one workgroup per texture -- tests and A/B runs use it)
  if (forced >= 0) return forced < 2 ? 2 : (forced > (int)log2_bh ? (int)log2_bh : forced);
  const uint64_t slots = (uint64_t)compute_units * (8u >> (log2_bw - 6u));
  int best = -1;
  double best_us = 0.0;
  for (int sb = 2; sb <= 6 && sb <= (int)log2_bh; ++sb) {
    const uint64_t wgs = n_images << (log2_bh - (uint32_t)sb);
    const double us = (double)((wgs + slots - 1) / slots) * (11.0 + 5.8 * (double)(1u << sb));
    if (best < 0 || us <= best_us) { best_us = us; best = sb; }
  }
  const double pair_us = 10.0 + 52.6e-6 * (double)(n_images << (log2_bw + log2_bh));
  return always || best_us < 0.97 * pair_us ? best : -1;
}
hipError_t launch_pvrtc2_onepass(const PvrtcParams &P, int sb, hipStream_t stream) {
  const uint32_t log2_bw = P.log2_size - 3u, bw = 1u << log2_bw, waves = bw >> 6;
  const size_t lds_bytes = (size_t)waves * (kOnePassWaveDwords + 2u * kOnePassXchDwords) * 4u;
  // more than 64 KiB of dynamic LDS has to be allowed per function and device, once
  static std::atomic<uint64_t> allowed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !((allowed.load() >> dev) & 1u)) {  // (ordinals past 63: every time)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(icamd_pvrtc2_onepass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(8u * (kOnePassWaveDwords + 2u * kOnePassXchDwords) * 4u));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) allowed.fetch_or(1ull << dev);
  }
  PvrtcLaunch L;
  L.src = P.src;
  L.dst = P.dst;
  L.ab = nullptr;
  L.src_image_stride = P.src_image_stride;
  L.dst_image_stride = P.dst_image_stride;
  L.size = P.size;
  L.log2_bw = log2_bw;
  L.log2_bpi = 2 * P.log2_size - 5;
  L.log2_strip = (uint32_t)sb;
  L.rx0 = L.ry0 = L.z_first = 0;
  L.log2_rw = log2_bw;
  L.log2_rblocks = L.log2_bpi;
  L.total_blocks = L.total_strips = 0;
  // the tile's write-out issues 16-byte stores: only when every image's output is 16-byte aligned (the contract asks for 8)
  L.stage_stores = (reinterpret_cast<uintptr_t>(P.dst) % 16u == 0 && (P.n_images == 1 || P.dst_image_stride % 16u == 0)) ? 1u : 0u;
  const uint32_t strips_per_image = (P.size / 4u) >> sb;
  // (launch_pvrtc2 has checked blocks per image x images < 2^31, so the grid fits)
  hipLaunchKernelGGL(icamd_pvrtc2_onepass_kernel, dim3((uint32_t)(P.n_images * (uint64_t)strips_per_image)), dim3(bw), lds_bytes, stream, L);
  return hipGetLastError();
}

// Halo form (r06): the rectangle (rx0, ry0, 2^log2_rw x 2^log2_rh blocks; a whole texture: 0, 0, bw, bh) of each image, workgroups
// of min(2^log2_rw, 512) lanes, strips of 2^sb block rows.  One launch, no scratch memory.
struct HaloRect {
  uint32_t rx0, ry0, log2_rw, log2_rh, z_first;
};
constexpr uint32_t kOnePassHaloTableBytes = (64u + 4u) * 8u + (64u + 3u) * 16u + (64u + 3u) * 8u;  // strips of at most 64 blocks
hipError_t launch_pvrtc2_onepass_halo(const PvrtcParams &P, const HaloRect &R, int sb, hipStream_t stream) {
  const uint32_t log2_bw = P.log2_size - 3u;
  const uint32_t log2_wgc = R.log2_rw < 9u ? R.log2_rw : 9u, lanes = 1u << log2_wgc, waves = lanes >> 6;
  const uint32_t groups = 1u << (R.log2_rw - log2_wgc);
  const size_t lds_bytes = (size_t)waves * (kOnePassWaveDwords + 2u * kOnePassXchDwords) * 4u + kOnePassHaloTableBytes;
  static std::atomic<uint64_t> allowed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !((allowed.load() >> dev) & 1u)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(icamd_pvrtc2_onepass_halo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(8u * (kOnePassWaveDwords + 2u * kOnePassXchDwords) * 4u + kOnePassHaloTableBytes));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) allowed.fetch_or(1ull << dev);
  }
  PvrtcLaunch L;
  L.src = P.src;
  L.dst = P.dst;
  L.ab = nullptr;
  L.src_image_stride = P.src_image_stride;
  L.dst_image_stride = P.dst_image_stride;
  L.size = P.size;
  L.log2_bw = log2_bw;
  L.log2_bpi = 2 * P.log2_size - 5;
  L.log2_strip = (uint32_t)sb;
  L.rx0 = R.rx0;
  L.ry0 = R.ry0;
  L.z_first = R.z_first;
  L.log2_rw = R.log2_rw;
  L.log2_rblocks = R.log2_rw + R.log2_rh;
  L.total_blocks = L.total_strips = 0;
  L.log2_wgc = log2_wgc;
  L.stage_stores = (reinterpret_cast<uintptr_t>(P.dst) % 16u == 0 && (P.n_images == 1 || P.dst_image_stride % 16u == 0)) ? 1u : 0u;
  (void)hipGetLastError();
  const uint64_t wgs = (uint64_t)P.n_images * ((1ull << R.log2_rh) >> sb) * groups;
  hipLaunchKernelGGL(icamd_pvrtc2_onepass_halo_kernel, dim3((uint32_t)wgs), dim3(lanes), lds_bytes, stream, L);
  return hipGetLastError();
}
// Strip height of the halo form for a rectangle of 2^log2_rw x 2^log2_rh blocks per image, or -1 where the pair is the better
// choice: the time model of onepass_log2_strip (a K-block strip workgroup: 11 + 5.8 K us whatever its width, 8 / waves of them per
// CU) + 2 us for the prologue; the pair: 10 us + 52.6 us per million blocks.
int onepass_halo_log2_strip(const HaloRect &R, uint64_t n_images, int forced, bool always, uint32_t compute_units) {
  if (R.log2_rw < 6u || R.log2_rh < 2u) return -1;  // at least one wave wide and one 4-block strip tall
  const uint32_t log2_wgc = R.log2_rw < 9u ? R.log2_rw : 9u;
  const uint64_t groups = 1ull << (R.log2_rw - log2_wgc);
  const int max_sb = R.log2_rh < 6u ? (int)R.log2_rh : 6;
  if (forced >= 0) return forced < 2 ? 2 : (forced > max_sb ? max_sb : forced);
  const uint64_t slots = (uint64_t)compute_units * (8u >> (log2_wgc - 6u));
  int best = -1;
  double best_us = 0.0;
  for (int sb = 2; sb <= max_sb; ++sb) {
    const uint64_t wgs = (n_images << (R.log2_rh - (uint32_t)sb)) * groups;
    const double us = 2.0 + (double)((wgs + slots - 1) / slots) * (11.0 + 5.8 * (double)(1u << sb));
    if (best < 0 || us <= best_us) { best_us = us; best = sb; }
  }
  const double pair_us = 10.0 + 52.6e-6 * (double)(n_images << (R.log2_rw + R.log2_rh));
  return always || best_us < 0.97 * pair_us ? best : -1;
}
}  // namespace

// One image, blocks [z_first, z_first + n_blocks) of its Z-order output (n_blocks a power of two, z_first a multiple
// of it: a rectangle of the block grid).  Reads the region's pixels and a one-block ring around it only.
static hipError_t launch_pvrtc2_region(const PvrtcParams &P, hipStream_t stream) {
  const uint32_t log2_bw = P.log2_size - 3, log2_bpi = 2 * P.log2_size - 5;
  uint32_t m = 0;
  while ((1u << m) < P.region_blocks) ++m;
  if ((1u << m) != P.region_blocks || m > log2_bpi || (P.region_first & (P.region_blocks - 1u)) != 0 ||
      (uint64_t)P.region_first + P.region_blocks > (1ull << log2_bpi))
    return hipErrorInvalidValue;
  // r06: regions at least one wave wide and one 4-block strip tall take the one-pass kernel's halo form where the time model
  // prefers it (one read of the pixels, one launch, no scratch) -- the multi-GPU split of ONE large texture (sharding.pvrtc_region)
  // no longer pays the pair's second pass over the pixels; icamd_pvrtc2_tune(1, ...) keeps the pair for A/B runs and tests
  read_path_env();
  if (g_path_mode.load() != 1) {
    const uint32_t lrw = m / 2;
    const HaloRect R = { compact_even_bits_host(P.region_first >> 1), compact_even_bits_host(P.region_first), lrw, m - lrw, P.region_first };
    const bool force = g_path_mode.load() == 2;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int hsb = onepass_halo_log2_strip(R, 1, force ? g_path_strip.load() : -1, force, device_compute_units(dev));
    if (hsb >= 0) return launch_pvrtc2_onepass_halo(P, R, hsb, stream);
  }
  uint2 *ab = nullptr;
  Workspace &ws = g_tls_workspace.get();
  hipError_t e = ws.acquire(((size_t)sizeof(uint2)) << log2_bpi, stream, reinterpret_cast<void **>(&ab));
  if (e != hipSuccess) return e;
  (void)hipGetLastError();
  PvrtcLaunch L;
  L.src = P.src;
  L.dst = P.dst;
  L.ab = ab;
  L.src_image_stride = L.dst_image_stride = 0;
  L.size = P.size;
  L.log2_bw = log2_bw;
  L.log2_bpi = log2_bpi;
  L.log2_rblocks = m;
  L.log2_rw = m / 2;  // x owns the odd bits of the Z index: floor(m/2) of the low m bits
  const uint32_t log2_rh = m - L.log2_rw;
  L.rx0 = compact_even_bits_host(P.region_first >> 1);
  L.ry0 = compact_even_bits_host(P.region_first);
  L.z_first = P.region_first;
  L.log2_strip = log2_rh < 3 ? log2_rh : 3;
  while (L.log2_strip > 0 && (P.region_blocks >> L.log2_strip) < kFullChipLanes) --L.log2_strip;
  // the staged write-out issues 16-byte stores: only when the region's output is 16-byte aligned (8 is the contract)
  const bool dst16 = reinterpret_cast<uintptr_t>(P.dst) % 16u == 0;
  L.stage_stores = (L.log2_strip >= 1 && L.log2_rw >= L.log2_strip && dst16) ? 1u : 0u;
  L.total_blocks = 1u << log2_bpi;
  L.total_strips = P.region_blocks >> L.log2_strip;
  const uint32_t rw = 1u << L.log2_rw, rh = 1u << log2_rh;
  const dim3 gm((rw + 2 + kMorphLanes - 1) / kMorphLanes, rh + 2), ge((L.total_strips + kEncodeLanes - 1) / kEncodeLanes);
  hipLaunchKernelGGL(icamd_pvrtc2_morph_rect_kernel, gm, dim3(kMorphLanes), 0, stream, L);
  hipLaunchKernelGGL(L.log2_rw >= 6 ? icamd_pvrtc2_encode_kernel : icamd_pvrtc2_encode_narrow_kernel, ge, dim3(kEncodeLanes),
                     0, stream, L);
  e = hipGetLastError();
  const hipError_t e2 = ws.release(stream);
  return e != hipSuccess ? e : e2;
}

void pvrtc2_tune(int mode, int log2_strip) {
  read_path_env();
  g_path_mode.store(mode < 0 || mode > 2 ? 0 : mode);
  g_path_strip.store(log2_strip < 0 ? -1 : log2_strip);
}

hipError_t launch_pvrtc2(const PvrtcParams &P, hipStream_t stream) {
  if (P.n_images == 0) return hipSuccess;
  if (P.region_blocks != 0) return P.n_images == 1 ? launch_pvrtc2_region(P, stream) : hipErrorInvalidValue;
  read_path_env();
  if (g_path_mode.load() != 1 && (uint64_t)(P.size / 8) * (P.size / 4) * P.n_images < (1ull << 31)) {
    const bool force = g_path_mode.load() == 2;  // (a strip height only counts together with mode 2)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int sb = onepass_log2_strip(P.log2_size, P.n_images, force ? g_path_strip.load() : -1, force, device_compute_units(dev));
    if (sb >= 0) return launch_pvrtc2_onepass(P, sb, stream);
    if (P.log2_size - 3u > 9u) {  // textures of 8192^2 and more (r06): a block row is two or more workgroups wide -> halo form
      const HaloRect R = { 0u, 0u, P.log2_size - 3u, P.log2_size - 2u, 0u };
      const int hsb = onepass_halo_log2_strip(R, P.n_images, force ? g_path_strip.load() : -1, force, device_compute_units(dev));
      if (hsb >= 0) return launch_pvrtc2_onepass_halo(P, R, hsb, stream);
    }
  }
  const uint32_t bw = P.size / 8, bh = P.size / 4;
  const uint64_t bpi = (uint64_t)bw * bh;
  // Images per launch pair.  Every launch boundary costs a drain/fill of ~4 waves per SIMD: measured 0.67 / 0.60 /
  // 0.58 / 0.57 ms per 16 x 4096^2 for groups of 64 MiB / 128 MiB / 512 MiB / 1 GiB of pixels (r01).  So: as many
  // images per launch as the 32-bit block index and a 256 MiB workspace allow.  (r02: hosting the encode workgroups
  // of one image group and the morph workgroups of the next in ONE grid, 1 : 2 interleaved -- the encode role is
  // VALU-bound, the morph role memory-bound -- was measured at 0.55 / 0.59 / 0.68 ms for 2 / 4 / 8 stages against
  // 0.53 ms for the two plain kernels: the shared 167-VGPR allocation and the extra fill/drain phases cost more than
  // the overlap gains.  Removed.)
  const uint64_t group = pvrtc_group(P.size, P.n_images);
  if (bpi * group >= (1ull << 31)) return hipErrorInvalidValue;

  // Workspace for the reduced colours (8 B per block of one image group).  A grow-only hipMalloc buffer per host
  // thread: stream-ordered pool memory (hipMallocAsync) proved unusable for producer->consumer kernels on ROCm 7.2
  // (a reused pool block was observed zero-filled underneath the first kernel; scripts/coh_test.hip).
  uint2 *ab = nullptr;
  Workspace &ws = g_tls_workspace.get();
  hipError_t e = ws.acquire((size_t)(bpi * group * sizeof(uint2)), stream, reinterpret_cast<void **>(&ab), P.internal_workspace);
  if (e != hipSuccess) return e;
  (void)hipGetLastError();
  PvrtcLaunch L;
  L.ab = ab;
  L.src_image_stride = P.src_image_stride;
  L.dst_image_stride = P.dst_image_stride;
  L.size = P.size;
  L.log2_bw = P.log2_size - 3;
  L.log2_bpi = 2 * P.log2_size - 5;
  // strip height: 8 blocks (32 pixel rows) amortise the one halo row per strip to 1/32 of the modulation work
  // while a 4096^2 image still yields 1 024 waves; never more than the image's block rows (size / 4)
  L.log2_strip = P.log2_size - 2 < 3 ? P.log2_size - 2 : 3;
  // ... and never so tall that a small launch leaves the chip empty: a strip is one long dependent instruction
  // stream (~1 200 per block), so below ~2 waves per SIMD of strips, shorter strips finish sooner
  // (one 1024^2 texture: 40 us with 8-block strips, a quarter of that with 1-block strips)
  while (L.log2_strip > 0 && ((bpi * group) >> L.log2_strip) < kFullChipLanes) --L.log2_strip;
  L.rx0 = L.ry0 = L.z_first = 0;
  L.log2_rw = L.log2_bw;
  L.log2_rblocks = L.log2_bpi;
  // the staged write-out issues 16-byte stores: taken only when every image's output is 16-byte aligned (the
  // contract asks for 8); otherwise each block is stored on its own, 8 bytes at its Z-order slot
  const bool dst16 = reinterpret_cast<uintptr_t>(P.dst) % 16u == 0 && (P.n_images == 1 || P.dst_image_stride % 16u == 0);
  for (uint64_t first = 0; first < P.n_images; first += group) {
    const uint64_t count = (P.n_images - first < group) ? P.n_images - first : group;
    // images [i0, i0 + cnt) of this chunk as one launch descriptor
    auto part = [&](uint64_t i0, uint64_t cnt, uint32_t log2_strip) {
      PvrtcLaunch Q = L;
      Q.src = P.src + (first + i0) * P.src_image_stride;
      Q.dst = P.dst + (first + i0) * P.dst_image_stride;
      Q.ab = ab + i0 * bpi;
      Q.log2_strip = log2_strip;
      Q.stage_stores = (log2_strip >= 1 && Q.log2_rw >= log2_strip && dst16) ? 1u : 0u;
      Q.total_blocks = (uint32_t)(bpi * cnt);
      Q.total_strips = Q.total_blocks >> log2_strip;
      return Q;
    };
    const PvrtcLaunch Q = part(0, count, L.log2_strip);
    // one block per lane up to AND INCLUDING two waves per SIMD of four-block lanes -- one 4096^2 texture is exactly that:
    // morph 17.0 -> 14.7 us per call (r03, rocprofv3; the memory-bound kernel wants the extra waves in flight)
    const bool small = Q.total_blocks <= (uint32_t)kMorphBlocksPerLane * kFullChipLanes;
    const uint32_t per_wg = kMorphLanes * (small ? 1 : kMorphBlocksPerLane);
    const dim3 gm((Q.total_blocks + per_wg - 1) / per_wg), ge((Q.total_strips + kEncodeLanes - 1) / kEncodeLanes);
    if (small) hipLaunchKernelGGL(icamd_pvrtc2_morph_small_kernel, gm, dim3(kMorphLanes), 0, stream, Q);
    else if (Q.log2_bw >= 6) hipLaunchKernelGGL(icamd_pvrtc2_morph_dense_kernel, gm, dim3(kMorphLanes), 0, stream, Q);
    else hipLaunchKernelGGL(icamd_pvrtc2_morph_kernel, gm, dim3(kMorphLanes), 0, stream, Q);
    hipLaunchKernelGGL(Q.log2_rw >= 6 ? icamd_pvrtc2_encode_kernel : icamd_pvrtc2_encode_narrow_kernel, ge,
                       dim3(kEncodeLanes), 0, stream, Q);
  }
  e = hipGetLastError();
  const hipError_t e2 = ws.release(stream);
  return e != hipSuccess ? e : e2;
}

// ---- PVRTC1 4 bpp (r05): EXTENSION, PARITY UNPINNED (pvrtc_block.h; the reference has no 4 bpp mode) -----------------------
// Two kernels like the 2 bpp pair, one 4 x 4 block per lane, lanes in Z ORDER (lane k of an image = block with Z index k:
// pvrtc.cc:80-86): the 8-byte stores of both kernels are then fully coalesced, a wave covers an 8 x 8-block patch whose
// pixel rows are 128-byte runs, and the eight neighbour blocks' colours (8 bytes each in the workspace, indexed by Z as
// well) mostly sit in the same patch.  x +- 1 / y +- 1 are carried out on the interleaved index itself.
struct Pvrtc4Launch {
  const uint8_t *src;
  uint8_t *dst;
  uint2 *ab;  // workspace: reduced colours, [image][z]
  uint64_t src_image_stride, dst_image_stride;
  uint32_t log2_n, log2_bpi, total_blocks;
};
__device__ __forceinline__ uint32_t compact_even_bits_dev(uint32_t v) {
  v &= 0x55555555u;
  v = (v | v >> 1) & 0x33333333u;
  v = (v | v >> 2) & 0x0f0f0f0fu;
  v = (v | v >> 4) & 0x00ff00ffu;
  v = (v | v >> 8) & 0x0000ffffu;
  return v;
}
__device__ __forceinline__ void pvrtc4_load_block(const Pvrtc4Launch &L, const uint32_t *img, uint32_t z, uint32_t px[16]) {
  const uint32_t bx = compact_even_bits_dev(z >> 1), by = compact_even_bits_dev(z);
  const uint32_t *p = img + (((by * 4u) << L.log2_n) + bx * 4u);
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const U4 v = load_stream(reinterpret_cast<const U4 *>(p + ((size_t)y << L.log2_n)));
    px[4 * y] = v.x; px[4 * y + 1] = v.y; px[4 * y + 2] = v.z; px[4 * y + 3] = v.w;
  }
}
extern "C" __global__ void __launch_bounds__(kMorphLanes) icamd_pvrtc4_morph_kernel(Pvrtc4Launch L) {
  __shared__ uint32_t lds_px[4][kMorphLanes][4];
  const uint32_t k = blockIdx.x * kMorphLanes + threadIdx.x;
  if (k >= L.total_blocks) return;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)(k >> L.log2_bpi) * L.src_image_stride);
  uint32_t px[16];
  pvrtc4_load_block(L, img, k & ((1u << L.log2_bpi) - 1u), px);
  BlockStash stash;
  stash.base = &lds_px[0][threadIdx.x][0];
  uint32_t a, c;
  pvrtc4_extremes(px, img[0], stash, a, c);
  L.ab[k] = make_uint2(channel_reduce(a, false), channel_reduce(c, true));
}
extern "C" __global__ void __launch_bounds__(kEncodeLanes) icamd_pvrtc4_encode_kernel(Pvrtc4Launch L) {
  const uint32_t k = blockIdx.x * kEncodeLanes + threadIdx.x;
  if (k >= L.total_blocks) return;
  const uint32_t image = k >> L.log2_bpi, mask = (1u << L.log2_bpi) - 1u, z = k & mask;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
  const uint2 *ab = L.ab + ((size_t)image << L.log2_bpi);
  // toroidal neighbours on the interleaved index: x lives in the odd bits, y in the even bits
  const uint32_t X = 0xaaaaaaaau & mask, Y = 0x55555555u & mask, zx = z & X, zy = z & Y;
  const uint32_t xs[3] = { (zx - 1u) & X, zx, ((zx | ~X) + 1u) & X }, ys[3] = { (zy - 1u) & Y, zy, ((zy | ~Y) + 1u) & Y };
  PvrtcColors nb[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint2 v = ab[xs[c] | ys[r]];
      nb[r][c].a = v.x;
      nb[r][c].b = v.y;
    }
  uint32_t px[16];
  pvrtc4_load_block(L, img, z, px);
  const uint32_t data = pvrtc4_block_data(px, nb);
  uint8_t *dst = L.dst + (size_t)image * L.dst_image_stride + (size_t)z * 8u;
  store_stream8(dst, data, pvrtc_pack_colors(nb[1][1].a, nb[1][1].b, true));  // bit 0 clear: standard modulation
}

// One-pass form (r05), the 4 bpp twin of icamd_pvrtc2_onepass_kernel: a workgroup is one whole block row of the texture wide
// (size / 4 lanes: 64 ... 1 024 = textures of 256^2 ... 4096^2), one lane = one 4-pixel block column of a strip
// (pvrtc4_onepass_strip).  A row slot of the per-wave ring is 64 lanes x 16 bytes = 1 KiB (one DMA instruction per row), eight
// of them plus a 1 KiB tile of finished blocks (2 block rows, 32-byte runs) = 9 KiB per wave: sixteen waves per CU, four per
// SIMD, which a 1 024-lane workgroup needs anyway (and which caps the walk at 128 VGPRs).
constexpr uint32_t kOnePass4WaveDwords = kOnePassRing * 256u + 256u;  // ring + tile: 9 216 bytes
constexpr uint32_t kOnePass4XchDwords = 4;                            // per wave and parity: lo.a lo.b hi.a hi.b
struct Pvrtc4OnePass {
  const uint8_t *src;
  uint8_t *dst;
  uint64_t src_image_stride, dst_image_stride;
  uint32_t log2_n, log2_strip, stage_stores;
};
extern "C" __global__ void __launch_bounds__(1024) icamd_pvrtc4_onepass_kernel(Pvrtc4OnePass L) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t n = 1u << L.log2_n, sb = L.log2_strip, K = 1u << sb;
  const uint32_t log2_spi = L.log2_n - 2u - sb;  // strips per image = (size / 4) >> sb
  const uint32_t image = blockIdx.x >> log2_spi, by0 = (blockIdx.x & ((1u << log2_spi) - 1u)) << sb;
  const uint32_t bx = threadIdx.x, lane = threadIdx.x & 63u;
  const uint32_t W = blockDim.x >> 6;
  const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t *img = reinterpret_cast<const uint32_t *>(L.src + (size_t)image * L.src_image_stride);
  uint2 *dst = reinterpret_cast<uint2 *>(L.dst + (size_t)image * L.dst_image_stride);
  const uint32_t image0 = opaque(img[0]);
  lds_u32 *ring = (lds_u32 *)(lds_dyn + wave_s * kOnePass4WaveDwords);
  const uint32_t ring_lane_byte = (uint32_t)(uintptr_t)ring + lane * 16u;
  const uint32_t tile_byte = (uint32_t)(uintptr_t)ring + kOnePassRing * 1024u;
  const uint32_t xch_byte = (uint32_t)(uintptr_t)(lds_u32 *)(lds_dyn + W * kOnePass4WaveDwords);
  const int last_row = (int)(4u * K + 3u);
  auto dma_row = [&](int m) {
    const uint32_t y = (by0 * 4u + (uint32_t)(m < last_row ? m : last_row)) & (n - 1u);
    __builtin_amdgcn_global_load_lds(img + ((y << L.log2_n) + bx * 4u), ring + ((uint32_t)m & (kOnePassRing - 1u)) * 256u, 16, 0, 0);
  };
  dma_row(-4); dma_row(-3); dma_row(-2);
  int cur_m = -4;
  auto tick = [&](int m, uint32_t *mp, uint32_t *ep) {
    cur_m = m;
    // rows m + 1 and m + 2 (one DMA instruction each) may still be in flight
    const uint32_t addr_m = ring_lane_byte + ((uint32_t)m & 7u) * 1024u, addr_e = ring_lane_byte + ((uint32_t)(m + 3) & 7u) * 1024u;
    uint4 m0, e0;
    asm volatile("s_waitcnt vmcnt(2)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(m0), "=&v"(e0) : "v"(addr_m), "v"(addr_e) : "memory");
    mp[0] = m0.x; mp[1] = m0.y; mp[2] = m0.z; mp[3] = m0.w;
    ep[0] = e0.x; ep[1] = e0.y; ep[2] = e0.z; ep[3] = e0.w;
    dma_row(m + 3);
  };
  auto lookup10 = [&](const uint32_t idx[10], uint32_t v[10]) {
    const uint32_t base = ring_lane_byte + ((uint32_t)(cur_m - 3) & 7u) * 1024u;
    uint32_t a[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {  // pixel idx = 4 y + x sits at y * 1024 + x * 4 (see the 2 bpp kernel's lookup)
      const uint32_t y = opaque(__builtin_amdgcn_ubfe(idx[i], 2u, 2u));
      a[i] = ((idx[i] & 3u) << 2) + ((y << 10) + base);
    }
    asm volatile("ds_read_b32 %0, %10\n\tds_read_b32 %1, %11\n\tds_read_b32 %2, %12\n\tds_read_b32 %3, %13\n\t"
                 "ds_read_b32 %4, %14\n\tds_read_b32 %5, %15\n\tds_read_b32 %6, %16\n\tds_read_b32 %7, %17\n\t"
                 "ds_read_b32 %8, %18\n\tds_read_b32 %9, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9])
                 : "memory");
  };
  const uint32_t left_wave = (wave_s == 0u ? W : wave_s) - 1u, right_wave = wave_s + 1u == W ? 0u : wave_s + 1u;
  auto exchange = [&](int s, const PvrtcColors &own, PvrtcColors &left, PvrtcColors &right) {
    const uint32_t x = xch_byte + ((uint32_t)s & 1u) * (W * kOnePass4XchDwords * 4u);
    const uint32_t mine = x + wave_s * (kOnePass4XchDwords * 4u);
    const uint2 c = make_uint2(own.a, own.b);
    if (lane == 0u) asm volatile("ds_write_b64 %0, %1" :: "v"(mine), "v"(c) : "memory");
    if (lane == 63u) asm volatile("ds_write_b64 %0, %1 offset:8" :: "v"(mine), "v"(c) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint2 l, r;
    const uint32_t al = x + left_wave * (kOnePass4XchDwords * 4u), ar = x + right_wave * (kOnePass4XchDwords * 4u);
    asm volatile("ds_read_b64 %0, %2 offset:8\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l), "=&v"(r) : "v"(al), "v"(ar) : "memory");
    left.a = dpp_from_lower_lane(own.a, l.x);   left.b = dpp_from_lower_lane(own.b, l.y);
    right.a = dpp_from_upper_lane(own.a, r.x);  right.b = dpp_from_upper_lane(own.b, r.y);
  };
  // tile: chunk = lane / 2 (two block columns x two block rows = four consecutive Z slots = 32 bytes)
  const uint32_t tile_lane_byte = tile_byte + ((lane >> 1) * 4u + ((lane & 1u) << 1)) * 8u;
  const uint32_t zx = spread_bits16(bx) << 1;
  const uint32_t flush_lds = tile_byte + lane * 16u;                                        // pair `lane`: chunk lane / 2, slots 2 (lane & 1) ..
  const uint32_t flush_zx = (spread_bits16(wave_s * 64u + (lane & ~1u)) << 1) + ((lane & 1u) << 1);
  auto store = [&](uint32_t j, uint32_t data, const PvrtcColors &own) {
    const uint2 v = make_uint2(data, pvrtc_pack_colors(own.a, own.b, true));  // bit 0 clear: standard modulation
    if (!L.stage_stores) {
      asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst + (zx | spread_bits16(by0 + j))), "v"(v) : "memory");
      return;
    }
    asm volatile("ds_write_b64 %0, %1" :: "v"(tile_lane_byte + (j & 1u) * 8u), "v"(v) : "memory");
    if (!(j & 1u)) return;
    icamd_u32x4 q;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q) : "v"(flush_lds) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst + (flush_zx | spread_bits16(by0 + j - 1u))), "v"(q) : "memory");
  };
  pvrtc4_onepass_strip(K, image0, tick, lookup10, exchange, store);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

namespace {
// Strip height of the 4 bpp one-pass kernel, or -1 where the pair is the better choice.  Measured (r05, profiles/
// r05_ab_pvrtc4_onepass.log): a strip workgroup takes 11 + 5.8 K us on a full CU whatever its width -- the 2 bpp kernel's
// figure: the same pixels per CU and block row --, a CU holds 16 / waves-per-workgroup of them; the pair takes 10 us + 1.87 us per
// million pixels.  Below ~8 Mpixel per launch neither fills the chip and the pair's two short kernels are as fast or faster
// (1 x 2048^2: 15 vs 21 us), so it keeps those.
int onepass4_log2_strip(uint32_t log2_size, uint64_t n_images, int forced, bool always, uint32_t compute_units) {
  const uint32_t log2_bw = log2_size - 2u;  // blocks per row = lanes per workgroup
  if (log2_bw < 6u || log2_bw > 10u) return -1;
  if (forced >= 0) return forced < 1 ? 1 : (forced > (int)log2_bw ? (int)log2_bw : forced);
  const uint64_t pixels = n_images << (2u * log2_size);
  if (!always && pixels < (8ull << 20)) return -1;
  const uint64_t slots = (uint64_t)compute_units * (16u >> (log2_bw - 6u));
  int best = -1;
  double best_us = 0.0;
  for (int sb = 2; sb <= 6 && sb <= (int)log2_bw; ++sb) {
    const uint64_t wgs = n_images << (log2_bw - (uint32_t)sb);
    const double us = (double)((wgs + slots - 1) / slots) * (11.0 + 5.8 * (double)(1u << sb));
    if (best < 0 || us <= best_us) { best_us = us; best = sb; }
  }
  const double pair_us = 10.0 + 1.87e-6 * (double)pixels;
  return always || best_us < 0.97 * pair_us ? best : -1;
}
hipError_t launch_pvrtc4_onepass(const PvrtcParams &P, int sb, hipStream_t stream) {
  const uint32_t lanes = P.size / 4u, waves = lanes >> 6;
  const size_t lds_bytes = (size_t)waves * (kOnePass4WaveDwords + 2u * kOnePass4XchDwords) * 4u;
  static std::atomic<uint64_t> allowed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !((allowed.load() >> dev) & 1u)) {  // (ordinals past 63: every time)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(icamd_pvrtc4_onepass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(16u * (kOnePass4WaveDwords + 2u * kOnePass4XchDwords) * 4u));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) allowed.fetch_or(1ull << dev);
  }
  Pvrtc4OnePass L;
  L.src = P.src;
  L.dst = P.dst;
  L.src_image_stride = P.src_image_stride;
  L.dst_image_stride = P.dst_image_stride;
  L.log2_n = P.log2_size;
  L.log2_strip = (uint32_t)sb;
  L.stage_stores = (reinterpret_cast<uintptr_t>(P.dst) % 16u == 0 && (P.n_images == 1 || P.dst_image_stride % 16u == 0)) ? 1u : 0u;
  const uint32_t strips_per_image = lanes >> sb;
  hipLaunchKernelGGL(icamd_pvrtc4_onepass_kernel, dim3((uint32_t)(P.n_images * (uint64_t)strips_per_image)), dim3(lanes), lds_bytes, stream, L);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_pvrtc4(const PvrtcParams &P, hipStream_t stream) {
  if (P.n_images == 0) return hipSuccess;
  if (P.region_blocks != 0) return hipErrorInvalidValue;
  read_path_env();
  if (g_path_mode.load() != 1 && (uint64_t)(P.size / 4) * (P.size / 4) * P.n_images < (1ull << 31)) {
    const bool force = g_path_mode.load() == 2;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int sb = onepass4_log2_strip(P.log2_size, P.n_images, force ? g_path_strip.load() : -1, force, device_compute_units(dev));
    if (sb >= 0) return launch_pvrtc4_onepass(P, sb, stream);
  }
  const uint64_t bpi = (uint64_t)(P.size / 4) * (P.size / 4);
  const uint64_t group = pvrtc_group(P.size, P.n_images);
  if (bpi * group >= (1ull << 31)) return hipErrorInvalidValue;
  uint2 *ab = nullptr;
  Workspace &ws = g_tls_workspace.get();
  hipError_t e = ws.acquire((size_t)(bpi * group * sizeof(uint2)), stream, reinterpret_cast<void **>(&ab), P.internal_workspace);
  if (e != hipSuccess) return e;
  (void)hipGetLastError();
  for (uint64_t first = 0; first < P.n_images; first += group) {
    const uint64_t count = (P.n_images - first < group) ? P.n_images - first : group;
    Pvrtc4Launch L;
    L.src = P.src + first * P.src_image_stride;
    L.dst = P.dst + first * P.dst_image_stride;
    L.ab = ab;
    L.src_image_stride = P.src_image_stride;
    L.dst_image_stride = P.dst_image_stride;
    L.log2_n = P.log2_size;
    L.log2_bpi = 2 * P.log2_size - 4;
    L.total_blocks = (uint32_t)(bpi * count);
    const dim3 grid((L.total_blocks + kMorphLanes - 1) / kMorphLanes);
    hipLaunchKernelGGL(icamd_pvrtc4_morph_kernel, grid, dim3(kMorphLanes), 0, stream, L);
    hipLaunchKernelGGL(icamd_pvrtc4_encode_kernel, grid, dim3(kEncodeLanes), 0, stream, L);
  }
  e = hipGetLastError();
  const hipError_t e2 = ws.release(stream);
  return e != hipSuccess ? e : e2;
}
size_t pvrtc4_workspace_bytes(uint32_t size, uint32_t n_images) {
  if (n_images == 0) return 0;
  return (size_t)((uint64_t)(size / 4) * (size / 4) * pvrtc_group(size, n_images) * sizeof(uint2));
}
const char *pvrtc4_kernel_name() { return "icamd_pvrtc4_onepass_kernel"; }

size_t pvrtc2_workspace_bytes(uint32_t size, uint32_t n_images) {
  if (n_images == 0) return 0;
  return (size_t)((uint64_t)(size / 8) * (size / 4) * pvrtc_group(size, n_images) * sizeof(uint2));
}
void pvrtc2_select_workspace(int slot) { g_tls_workspace.slot = slot & 1; }
void pvrtc2_set_workspace(void *d_workspace, size_t bytes) {
  const int keep = g_tls_workspace.slot;
  g_tls_workspace.slot = 0;
  Workspace &ws = g_tls_workspace.get();
  g_tls_workspace.slot = keep;
  ws.user_ptr = d_workspace;
  ws.user_bytes = d_workspace ? bytes : 0;
}

}  // namespace icamd
