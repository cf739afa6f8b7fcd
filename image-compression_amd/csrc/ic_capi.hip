// ic_capi.hip -- implementation of the C ABI declared in include/ic_amd.h.
//
// Argument validation mirrors the reference's public wrappers (the bool they return
// becomes ICAMD_OK / ICAMD_FALSE); everything else is geometry set-up and kernel
// launches.  There is no CPU encode path in this library.
#include "ic_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "containers.h"
#include "ic_abi.h"
#include "ic_launch.h"

// Error text and the exception barrier of the C ABI: declared in ic_abi.h (shared with rccl_gather.hip), defined here.
namespace icamd {
thread_local char g_last_error[kErrorChars] = "";

void set_last_error(const char *text) noexcept {
  std::snprintf(g_last_error, kErrorChars, "%s", text ? text : "");
}

int fail(int code, const char *what, hipError_t e) noexcept {
  if (e != hipSuccess)
    std::snprintf(g_last_error, kErrorChars, "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
  else
    std::snprintf(g_last_error, kErrorChars, "%s", what);
  return code;
}

int abi_exception() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    return fail(ICAMD_ERR_ALLOC, "out of host memory (std::bad_alloc)");
  } catch (const std::system_error &e) {
    char buf[kErrorChars];
    std::snprintf(buf, sizeof buf, "system resource unavailable (std::system_error: %s)", e.what());
    return fail(ICAMD_ERR_ALLOC, buf);
  } catch (const std::exception &e) {
    char buf[kErrorChars];
    std::snprintf(buf, sizeof buf, "unexpected C++ exception: %s", e.what());
    return fail(ICAMD_ERR_HIP, buf);
  } catch (...) {
    return fail(ICAMD_ERR_HIP, "unexpected C++ exception");
  }
}
}  // namespace icamd

namespace {
using icamd::abi_exception;
using icamd::fail;
using icamd::g_last_error;
using icamd::kErrorChars;
using icamd::set_last_error;

// Worker threads of the two batch entry points.  A worker's body never lets an exception out (that would be std::terminate);
// the guard joins whatever was started before the vectors the workers write to go out of scope, on every path out.
struct JoinAll {
  std::vector<std::thread> &threads;
  ~JoinAll() {
    for (std::thread &t : threads)
      if (t.joinable()) t.join();
  }
};
// A device list is a handful of GPUs, possibly each named a few times (two workers per GPU overlap their copies): a longer
// list is a caller's mistake, and one worker thread per entry must not be something a caller can ask 10 000 of.
constexpr int kMaxDeviceListEntries = 256;
// Text of a worker's first failure (fixed size: written from inside the worker without allocating).
struct WorkerError {
  char text[kErrorChars] = "";
  bool empty() const { return text[0] == 0; }
  void set(const char *t) noexcept { std::snprintf(text, sizeof text, "%s", t ? t : ""); }
};

#define ICAMD_HIP(call, what)                                   \
  do {                                                          \
    hipError_t e_ = (call);                                     \
    if (e_ != hipSuccess) return fail(ICAMD_ERR_HIP, what, e_); \
  } while (0)

uint32_t num_blocks4(uint32_t n) { return (n + 3) / 4; }  // helper.h:86-88
bool is_pow2(uint32_t x) { return x != 0 && !(x & (x - 1)); }
int format_components(int format) {  // compressed_image.h:188-199
  return (format == ICAMD_RGB || format == ICAMD_BGR) ? 3 : (format == ICAMD_RGBA || format == ICAMD_BGRA) ? 4 : 0;
}
bool format_swaps(int format) { return format == ICAMD_BGR || format == ICAMD_BGRA; }  // compressed_image.h:202-204

uint32_t ilog2(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x) ++l;
  return l;
}

// Per-thread device staging for the host-buffer entry points (grow-only).
struct Staging {
  int device = -1;
  hipStream_t stream = nullptr, stream2 = nullptr;  // two streams: bands of one image alternate between them
  void *d_in = nullptr, *d_out = nullptr;
  size_t cap_in = 0, cap_out = 0;
  void *h_in = nullptr, *h_out = nullptr;  // pinned host mirrors (batch workers only)
  size_t cap_hin = 0, cap_hout = 0;
  // No destructor: a Staging is never destroyed (see TlsStaging below) -- no HIP call may run from a thread_local or
  // static destructor, where the HIP runtime can already be gone.
  void drop() {
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
    if (stream) (void)hipStreamDestroy(stream);
    if (stream2) (void)hipStreamDestroy(stream2);
    d_in = d_out = h_in = h_out = nullptr;
    cap_in = cap_out = cap_hin = cap_hout = 0;
    stream = stream2 = nullptr;
  }
  int ensure(size_t in_bytes, size_t out_bytes, bool pinned = false) {
    int dev = 0;
    ICAMD_HIP(hipGetDevice(&dev), "hipGetDevice");
    if (dev != device) {  // thread moved to another device: drop the old buffers
      drop();
      device = dev;
    }
    if (!stream) ICAMD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    if (!stream2) ICAMD_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking), "hipStreamCreate");
    if (in_bytes > cap_in) {
      if (d_in) (void)hipFree(d_in);
      d_in = nullptr; cap_in = 0;
      if (hipMalloc(&d_in, in_bytes) != hipSuccess) return fail(ICAMD_ERR_ALLOC, "hipMalloc(input staging)");
      cap_in = in_bytes;
    }
    if (out_bytes > cap_out) {
      if (d_out) (void)hipFree(d_out);
      d_out = nullptr; cap_out = 0;
      if (hipMalloc(&d_out, out_bytes) != hipSuccess) return fail(ICAMD_ERR_ALLOC, "hipMalloc(output staging)");
      cap_out = out_bytes;
    }
    if (pinned && in_bytes > cap_hin) {
      if (h_in) (void)hipHostFree(h_in);
      h_in = nullptr; cap_hin = 0;
      if (hipHostMalloc(&h_in, in_bytes, hipHostMallocDefault) != hipSuccess) return fail(ICAMD_ERR_ALLOC, "hipHostMalloc(input)");
      cap_hin = in_bytes;
    }
    if (pinned && out_bytes > cap_hout) {
      if (h_out) (void)hipHostFree(h_out);
      h_out = nullptr; cap_hout = 0;
      if (hipHostMalloc(&h_out, out_bytes, hipHostMallocDefault) != hipSuccess) return fail(ICAMD_ERR_ALLOC, "hipHostMalloc(output)");
      cap_hout = out_bytes;
    }
    return ICAMD_OK;
  }
};

// Every Staging lives in (or is on loan from) this process-wide pool: icamd_compress_batch's short-lived worker
// threads take one and give it back, so repeated batches allocate nothing, and a thread's own staging (tls_staging())
// returns to the pool when the thread ends instead of being destroyed.  The pool itself is heap-allocated and never
// destroyed: at process exit the HIP runtime may already be gone, so nothing here calls into it from a destructor --
// the driver reclaims device memory and streams with the process.
std::mutex &g_pool_mutex = *new std::mutex();
std::vector<std::unique_ptr<Staging>> &g_pool = *new std::vector<std::unique_ptr<Staging>>();

std::unique_ptr<Staging> pool_take(int device) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  for (size_t i = 0; i < g_pool.size(); ++i)
    if (g_pool[i]->device == device) {
      std::unique_ptr<Staging> s = std::move(g_pool[i]);
      g_pool.erase(g_pool.begin() + (long)i);
      return s;
    }
  return std::unique_ptr<Staging>(new Staging());
}
void pool_give(std::unique_ptr<Staging> s) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  g_pool.push_back(std::move(s));
}

// The calling thread's staging for the host-buffer entry points: borrowed from the pool on first use (any device --
// Staging::ensure re-targets it), handed back -- not freed -- by the thread_local destructor.
struct TlsStaging {
  Staging *p = nullptr;
  Staging &get() {
    if (!p) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      p = pool_take(dev).release();
    }
    return *p;
  }
  ~TlsStaging() {
    if (!p) return;
    try {
      pool_give(std::unique_ptr<Staging>(p));  // no HIP call here
    } catch (...) {
      // (the pool could not grow: the staging's device buffers stay allocated until the process ends)
    }
  }
};
thread_local TlsStaging g_tls_staging;
Staging &tls_staging() { return g_tls_staging.get(); }

// ---- one large image per call as two concurrent bands (experiment knob) ----
// ICAMD_SPLIT_SINGLE=1 (read once): icamd_encode_device / icamd_compress_device calls with ONE image of at least
// ICAMD_SPLIT_SINGLE_MIN_BLOCKS (default 2^20 = one 4096^2 image) 4x4 blocks run as two kernels, the lower band on a per-thread
// side stream between a fork and a join event.  What it measured is in profiles/r06_ab_single_image_split.log and DESIGN.md 5.
int split_single_mode() {
  static const int mode = [] { const char *e = getenv("ICAMD_SPLIT_SINGLE"); return e && e[0] == '1' ? 1 : 0; }();
  return mode;
}
uint64_t split_single_min_blocks() {
  static const uint64_t n = [] {
    const char *e = getenv("ICAMD_SPLIT_SINGLE_MIN_BLOCKS");
    const long long v = e && *e ? atoll(e) : 0;
    return v > 0 ? (uint64_t)v : (1ull << 20);
  }();
  return n;
}
struct SplitLane {
  int device = -1;
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
// The calling thread's side stream and events on the current device (created on first use, re-created when the thread moves to
// another device; never destroyed: no HIP call may run from a thread_local destructor).  nullptr: could not be created.
SplitLane *tls_split_lane() {
  static thread_local SplitLane lane;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (lane.device != dev || !lane.side) {
    lane = SplitLane();
    if (hipStreamCreateWithFlags(&lane.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&lane.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&lane.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      lane = SplitLane();
      return nullptr;
    }
    lane.device = dev;
  }
  return &lane;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(ICAMD_ERR_NO_DEVICE, "no HIP device available; this library has no CPU path", e);
  return ICAMD_OK;
}

// What (compressor, format) encodes to; returns false when the reference's Compress would.
bool resolve_codec(int compressor, int format, int *codec, int *comps, bool *swap) {
  const int c = format_components(format);
  if (c == 0) return false;
  *comps = c;
  *swap = format_swaps(format);
  if (compressor == ICAMD_COMPRESSOR_DXTC) {  // dxtc.cc:741-749: 3 components -> DXT1, else DXT5
    *codec = c == 3 ? ICAMD_DXT1 : ICAMD_DXT5;
    return true;
  }
  if (compressor == ICAMD_COMPRESSOR_ETC) {  // etc.cc:751-754: kRGB only
    if (format != ICAMD_RGB) return false;
    *codec = ICAMD_ETC1;
    return true;
  }
  return false;
}

bool blockop_codec(int compressor, int format, int *codec) {
  int comps;
  bool swap;
  return resolve_codec(compressor, format, codec, &comps, &swap);
}

// host-buffer wrappers: stage in, run, stage out
template <typename F>
int staged_blockop(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_size, bool in_place, F &&run) {
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  Staging &st = tls_staging();
  rc = st.ensure(std::max<size_t>(in_size, 1), std::max<size_t>(out_size, 1));
  if (rc != ICAMD_OK) return rc;
  hipStream_t s = st.stream;
  ICAMD_HIP(hipMemcpyAsync(st.d_in, in, in_size, hipMemcpyHostToDevice, s), "H2D copy");
  rc = run(st.d_in, st.d_out, s);
  if (rc != ICAMD_OK) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  ICAMD_HIP(hipMemcpyAsync(out, in_place ? st.d_in : st.d_out, out_size, hipMemcpyDeviceToHost, s), "D2H copy");
  ICAMD_HIP(hipStreamSynchronize(s), "stream synchronize");
  return ICAMD_OK;
}

// PVRTC branch of icamd_encode_device.  internal_workspace: the call comes from the library's own host-buffer path
// (its staging stream), which must not borrow the workspace a caller registered for its own streams / graphs.
int pvrtc_encode_device_impl(int src_components, uint32_t height, uint32_t width, uint32_t row_stride_bytes,
                             uint32_t n_images, size_t src_image_stride_bytes, size_t dst_image_stride_bytes,
                             const void *d_src, void *d_dst, hipStream_t stream, bool internal_workspace) {
  // preconditions of PvrtcCompressor::Compress, pvrtc.cc:636-650 (source always read as RGBA8888)
  if (!is_pow2(width) || !is_pow2(height) || width != height || width % 8 || height % 4) return ICAMD_FALSE;
  // the reference sizes its output with the uint32 product width * height / 4 (pvrtc.cc:631-634), which wraps to 0 at
  // 65536^2 -- it would then write 2^30 bytes into a zero-byte buffer; refused here
  if (width >= 65536u) return ICAMD_FALSE;
  if (src_components != 4 || row_stride_bytes != width * 4u) return ICAMD_FALSE;
  if (reinterpret_cast<uintptr_t>(d_src) % 16u || reinterpret_cast<uintptr_t>(d_dst) % 8u ||
      (n_images > 1 && (src_image_stride_bytes % 16u || dst_image_stride_bytes % 8u)))
    return fail(ICAMD_ERR_ARG, "PVRTC: source must be 16-byte aligned, output 8-byte aligned");
  icamd::PvrtcParams P;
  P.src = static_cast<const uint8_t *>(d_src);
  P.dst = static_cast<uint8_t *>(d_dst);
  P.src_image_stride = src_image_stride_bytes;
  P.dst_image_stride = dst_image_stride_bytes;
  P.size = width;
  P.log2_size = ilog2(width);
  P.n_images = n_images;
  P.internal_workspace = internal_workspace;
  ICAMD_HIP(icamd::launch_pvrtc2(P, stream), "launch pvrtc2");
  return ICAMD_OK;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

const char *icamd_version(void) { return "image-compression_amd 0.6 (gfx950)"; }
const char *icamd_last_error(void) { return g_last_error; }

int icamd_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} ICAMD_ABI_CATCH

const char *icamd_kernel_name(int codec, int src_components) {
  switch (codec) {
    case ICAMD_DXT1: case ICAMD_DXT5: return icamd::dxt_kernel_name(codec, src_components);
    case ICAMD_ETC1: return icamd::etc1_kernel_name(src_components);
    case ICAMD_PVRTC2: return icamd::pvrtc2_kernel_name();
    case ICAMD_PVRTC4: return icamd::pvrtc4_kernel_name();
  }
  return "";
}

int icamd_supports_format(int compressor, int format) try {
  if (format_components(format) == 0) return 0;
  if (compressor == ICAMD_COMPRESSOR_DXTC) return 1;                      // dxtc.cc:707-710
  if (compressor == ICAMD_COMPRESSOR_ETC) return format == ICAMD_RGB;     // etc.cc:713-717
  if (compressor == ICAMD_COMPRESSOR_PVRTC) return format == ICAMD_RGBA;  // pvrtc.cc:607-609
  return 0;
} ICAMD_ABI_CATCH

size_t icamd_compute_compressed_data_size(int compressor, int format, uint32_t height, uint32_t width) {
  if (compressor == ICAMD_COMPRESSOR_PVRTC) return (size_t)(width * height / 4);  // pvrtc.cc:631-634 (uint32 product)
  if (height == 0 || width == 0) return 0;
  const size_t blocks = (size_t)std::max(1u, num_blocks4(height)) * std::max(1u, num_blocks4(width));
  if (compressor == ICAMD_COMPRESSOR_DXTC) return blocks * (format_components(format) == 3 ? 8u : 16u);  // dxtc.cc:276-280,725-733
  if (compressor == ICAMD_COMPRESSOR_ETC) return format == ICAMD_RGB ? blocks * 8u : 0;                  // etc.cc:734-745
  return 0;
}

size_t icamd_encoded_size(int codec, uint32_t grid_height, uint32_t grid_width) {
  if (codec == ICAMD_PVRTC2) return (size_t)grid_width * grid_height / 4;
  if (codec == ICAMD_PVRTC4) return (size_t)grid_width * grid_height / 2;
  return (size_t)num_blocks4(grid_height) * num_blocks4(grid_width) * (codec == ICAMD_DXT5 ? 16u : 8u);
}

int icamd_encode_device(int codec, int etc_strategy, int src_components, int swap_rb,
                        uint32_t height, uint32_t width, uint32_t grid_height, uint32_t grid_width,
                        uint32_t row_stride_bytes, uint32_t n_images,
                        size_t src_image_stride_bytes, size_t dst_image_stride_bytes,
                        const void *d_src, void *d_dst, void *hip_stream) try {
  if (!d_src || !d_dst || height == 0 || width == 0) return ICAMD_FALSE;
  if (src_components != 3 && src_components != 4) return fail(ICAMD_ERR_ARG, "src_components must be 3 or 4");
  if (n_images == 0) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);

  if (codec == ICAMD_PVRTC2)
    return pvrtc_encode_device_impl(src_components, height, width, row_stride_bytes, n_images, src_image_stride_bytes,
                                    dst_image_stride_bytes, d_src, d_dst, stream, false);
  if (codec == ICAMD_PVRTC4) {
    // EXTENSION (parity unpinned, include/ic_amd.h): PVRTC1 4 bpp under the preconditions of PvrtcCompressor::Compress
    // (pvrtc.cc:636-650: square power of two, RGBA8888, no row padding), at least 8 x 8
    if (!is_pow2(width) || width != height || width < 8u || width >= 65536u) return ICAMD_FALSE;
    if (src_components != 4 || row_stride_bytes != width * 4u) return ICAMD_FALSE;
    if (reinterpret_cast<uintptr_t>(d_src) % 16u || reinterpret_cast<uintptr_t>(d_dst) % 8u ||
        (n_images > 1 && (src_image_stride_bytes % 16u || dst_image_stride_bytes % 8u)))
      return fail(ICAMD_ERR_ARG, "PVRTC: source must be 16-byte aligned, output 8-byte aligned");
    icamd::PvrtcParams P;
    P.src = static_cast<const uint8_t *>(d_src);
    P.dst = static_cast<uint8_t *>(d_dst);
    P.src_image_stride = src_image_stride_bytes;
    P.dst_image_stride = dst_image_stride_bytes;
    P.size = width;
    P.log2_size = ilog2(width);
    P.n_images = n_images;
    ICAMD_HIP(icamd::launch_pvrtc4(P, stream), "launch pvrtc4");
    return ICAMD_OK;
  }
  if (codec != ICAMD_DXT1 && codec != ICAMD_DXT5 && codec != ICAMD_ETC1) return fail(ICAMD_ERR_ARG, "unknown codec");
  if (codec == ICAMD_DXT5 && src_components != 4) return fail(ICAMD_ERR_ARG, "DXT5 needs a 4-component source");
  if (row_stride_bytes < width * (uint32_t)src_components) return fail(ICAMD_ERR_ARG, "row stride smaller than a row");

  icamd::GridParams P;
  P.src = static_cast<const uint8_t *>(d_src);
  P.dst = static_cast<uint8_t *>(d_dst);
  P.src_image_stride = src_image_stride_bytes;
  P.dst_image_stride = dst_image_stride_bytes;
  P.height = height;
  P.width = width;
  P.block_rows = num_blocks4(std::max(height, grid_height));  // helper.h:487-488,501-502
  P.block_cols = num_blocks4(std::max(width, grid_width));
  P.row_stride = row_stride_bytes;
  P.n_images = n_images;
  P.swap_rb = swap_rb ? 1u : 0u;
  P.etc_strategy = (uint32_t)etc_strategy;
  P.log2_tile_cols = P.tile_row0 = P.force_gather = 0;
  auto launch = [&](const icamd::GridParams &G, hipStream_t s) {
    return codec == ICAMD_ETC1 ? icamd::launch_etc1(src_components, G, s) : icamd::launch_dxt(codec, src_components, G, s);
  };
  // ONE large image per call: optionally as two bands of block rows, the second on a side stream that is
  // forked from and joined back into the caller's stream by events (legal under stream capture too).  Off by default: see
  // split_single_mode() for what it measured.
  if (n_images == 1 && split_single_mode() != 0 && (uint64_t)P.block_rows * P.block_cols >= split_single_min_blocks() && P.block_rows >= 2) {
    SplitLane *lane = tls_split_lane();
    if (lane) {
      const uint32_t top = P.block_rows / 2u;
      icamd::GridParams A = P, B = P;
      A.block_rows = top;
      A.height = std::min(height, top * 4u);
      B.block_rows = P.block_rows - top;
      B.height = height > top * 4u ? height - top * 4u : 0u;
      if (B.height != 0) {  // (a CompressAndPad grid whose lower band lies wholly below the image is not split)
        B.src = P.src + (size_t)top * 4u * row_stride_bytes;
        B.dst = P.dst + (size_t)top * P.block_cols * (codec == ICAMD_DXT5 ? 16u : 8u);
        ICAMD_HIP(hipEventRecord(lane->fork, stream), "split: fork event");
        ICAMD_HIP(hipStreamWaitEvent(lane->side, lane->fork, 0), "split: fork wait");
        ICAMD_HIP(launch(B, lane->side), "launch (lower band)");
        ICAMD_HIP(launch(A, stream), "launch (upper band)");
        ICAMD_HIP(hipEventRecord(lane->join, lane->side), "split: join event");
        ICAMD_HIP(hipStreamWaitEvent(stream, lane->join, 0), "split: join wait");
        return ICAMD_OK;
      }
    }
  }
  ICAMD_HIP(launch(P, stream), codec == ICAMD_ETC1 ? "launch etc1" : "launch dxt");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_pvrtc2_encode_region_device(uint32_t size, uint32_t first_block, uint32_t n_blocks, const void *d_src,
                                      void *d_dst_region, void *hip_stream) try {
  if (!d_src || !d_dst_region || n_blocks == 0) return ICAMD_FALSE;
  if (!is_pow2(size) || size < 8) return ICAMD_FALSE;  // pvrtc.cc:640-646
  if (size >= 65536u) return ICAMD_FALSE;              // as icamd_encode_device: the kernels index pixels with 32 bits
  const uint64_t blocks = (uint64_t)(size / 8) * (size / 4);
  if (!is_pow2(n_blocks) || (first_block & (n_blocks - 1u)) != 0 || (uint64_t)first_block + n_blocks > blocks)
    return fail(ICAMD_ERR_ARG, "PVRTC region must be a power-of-two, aligned range of the image's blocks");
  if (reinterpret_cast<uintptr_t>(d_src) % 16u || reinterpret_cast<uintptr_t>(d_dst_region) % 8u)
    return fail(ICAMD_ERR_ARG, "PVRTC: source must be 16-byte aligned, output 8-byte aligned");
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  icamd::PvrtcParams P;
  P.src = static_cast<const uint8_t *>(d_src);
  P.dst = static_cast<uint8_t *>(d_dst_region);
  P.src_image_stride = P.dst_image_stride = 0;
  P.size = size;
  P.log2_size = ilog2(size);
  P.n_images = 1;
  P.region_first = first_block;
  P.region_blocks = n_blocks;
  ICAMD_HIP(icamd::launch_pvrtc2(P, static_cast<hipStream_t>(hip_stream)), "launch pvrtc2 region");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_host_register(void *host_ptr, size_t bytes) try {
  if (!host_ptr || bytes == 0) return fail(ICAMD_ERR_ARG, "icamd_host_register: null buffer");
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  ICAMD_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault), "hipHostRegister");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_host_unregister(void *host_ptr) try {
  if (!host_ptr) return fail(ICAMD_ERR_ARG, "icamd_host_unregister: null buffer");
  ICAMD_HIP(hipHostUnregister(host_ptr), "hipHostUnregister");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

size_t icamd_pvrtc2_workspace_size(uint32_t size, uint32_t n_images) {
  if (!is_pow2(size) || size < 8) return 0;
  return icamd::pvrtc2_workspace_bytes(size, n_images);
}

size_t icamd_pvrtc4_workspace_size(uint32_t size, uint32_t n_images) {
  if (!is_pow2(size) || size < 8) return 0;
  return icamd::pvrtc4_workspace_bytes(size, n_images);
}

int icamd_pvrtc2_set_workspace(void *d_workspace, size_t bytes) try {
  if (d_workspace && reinterpret_cast<uintptr_t>(d_workspace) % 8u) return fail(ICAMD_ERR_ARG, "workspace must be 8-byte aligned");
  if (d_workspace) {
    // the kernels of this thread's later calls write through this pointer: it must be device memory of the current device
    hipPointerAttribute_t attr;
    int dev = -1;
    if (hipPointerGetAttributes(&attr, d_workspace) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        attr.type != hipMemoryTypeDevice || attr.device != dev) {
      (void)hipGetLastError();
      return fail(ICAMD_ERR_ARG, "workspace must be device memory of the current HIP device");
    }
  }
  icamd::pvrtc2_set_workspace(d_workspace, bytes);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_pvrtc2_tune(int mode, int log2_strip) try {
  if (mode < 0 || mode > 2) return fail(ICAMD_ERR_ARG, "icamd_pvrtc2_tune: mode must be 0 (auto), 1 (two kernels) or 2 (one pass)");
  icamd::pvrtc2_tune(mode, log2_strip);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_compress_and_pad_device(int compressor, int etc_strategy, int format,
                                  uint32_t height, uint32_t width,
                                  uint32_t padded_height, uint32_t padded_width,
                                  uint32_t padding_bytes_per_row,
                                  const void *d_buffer, void *d_out, size_t out_size, void *hip_stream) try {
  if (compressor == ICAMD_COMPRESSOR_PVRTC) return ICAMD_FALSE;  // pvrtc.cc:684-691
  // dxtc.cc:799-818 / etc.cc:787-800
  if (!d_buffer || !d_out || height == 0 || width == 0) return ICAMD_FALSE;
  int codec, comps;
  bool swap;
  if (!resolve_codec(compressor, format, &codec, &comps, &swap)) return ICAMD_FALSE;
  const uint32_t gh = std::max(height, padded_height), gw = std::max(width, padded_width);
  const size_t need = icamd_encoded_size(codec, gh, gw);
  if (out_size != need) return ICAMD_FALSE;  // compressor4x4_helper.cc:34-41
  return icamd_encode_device(codec, etc_strategy, comps, swap, height, width, gh, gw,
                             width * (uint32_t)comps + padding_bytes_per_row, 1, 0, 0, d_buffer, d_out, hip_stream);
} ICAMD_ABI_CATCH

int icamd_compress_device(int compressor, int etc_strategy, int format,
                          uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                          const void *d_buffer, void *d_out, size_t out_size, void *hip_stream) try {
  if (compressor == ICAMD_COMPRESSOR_PVRTC) {
    // pvrtc.cc:636-667.  `format` is deliberately not validated (neither does the reference).
    if (!d_buffer || !d_out || height == 0 || width == 0) return ICAMD_FALSE;
    if (!is_pow2(width) || !is_pow2(height) || width != height) return ICAMD_FALSE;
    if (padding_bytes_per_row != 0) return ICAMD_FALSE;
    if (width % 8 != 0 || height % 4 != 0) return ICAMD_FALSE;
    if (width >= 65536u) return ICAMD_FALSE;  // the reference's uint32 width * height / 4 wraps (see pvrtc_encode_device_impl)
    if (out_size != (size_t)(width * height / 4)) return ICAMD_FALSE;
    return icamd_encode_device(ICAMD_PVRTC2, 0, 4, 0, height, width, height, width, width * 4u, 1, 0, 0,
                               d_buffer, d_out, hip_stream);
  }
  return icamd_compress_and_pad_device(compressor, etc_strategy, format, height, width, height, width,
                                       padding_bytes_per_row, d_buffer, d_out, out_size, hip_stream);
} ICAMD_ABI_CATCH

// Measured on the MI355X box (r01, 32 x 2048^2 kRGB -> DXT1, workers on one device): pageable copies 21 GB/s with one
// worker; page-locked mirrors 10 / 14 / 20 / 22 GB/s with 1 / 2 / 4 / 8 workers -- the extra CPU memcpy costs more
// than the asynchronous DMA gains, so the batch path keeps the pageable copies.
constexpr bool kBatchPinnedStaging = false;

// Source bytes per band of the host-buffer pipeline (whole block rows): the kernel and the D2H copy of band i run
// underneath the H2D copy of band i+1.  Measured on the MI355X box (r02, one 4096^2 kRGB image = 48 MiB, pageable
// buffers, ms per call): bands of 2 / 4 / 8 / 16 / 24 / 32 MiB -> 2.33 / 1.91 / 1.38 / 1.32 / 1.21 / 1.09 (one band):
// a PCIe copy needs tens of MiB to reach its ~50 GB/s, and there is little to hide (D2H + kernel = 15 % of the H2D
// time), so bands are large and images below 64 MiB go in one piece.
constexpr size_t kHostBandBytes = (size_t)32 << 20;

// The host-buffer drop-in (Compressor::Compress / CompressAndPad, compressor.h:77-80,114-119): argument validation
// first (nothing is staged or copied for a call the reference would refuse), then DXT / ETC images are cut into bands
// of whole block rows -- blocks are independent and stored row-major (helper.h:202-214), so a band is an image of its
// own and its blocks are one contiguous byte range of the output -- and band i's H2D copy, kernel and D2H copy are
// enqueued on stream i % 2: the copy engines and the kernel of consecutive bands overlap.  Caller buffers that are
// page-locked (icamd_host_register / hipHostMalloc) are DMA-ed directly at the PCIe rate; pageable ones go through the
// runtime's own staging.  PVRTC (toroidal neighbourhood, Z-order output) is staged whole.
// `pinned`: copy through the Staging's page-locked mirrors (batch workers only, see above).
static int compress_host_common(Staging &st, bool pinned, bool and_pad, int compressor, int etc_strategy, int format,
                                uint32_t height, uint32_t width, uint32_t padded_height, uint32_t padded_width,
                                uint32_t padding_bytes_per_row, const uint8_t *buffer, uint8_t *out,
                                size_t out_size) {
  if (!buffer || !out || height == 0 || width == 0) return ICAMD_FALSE;
  const bool pvrtc = compressor == ICAMD_COMPRESSOR_PVRTC;
  int codec = ICAMD_PVRTC2, comps = 4;  // PVRTC: the buffer is reinterpreted as RGBA8888 whatever `format`, pvrtc.cc:664
  bool swap = false;
  uint32_t gh = height, gw = width;
  if (pvrtc) {
    if (and_pad) return ICAMD_FALSE;  // pvrtc.cc:684-691
    // pvrtc.cc:636-667
    if (!is_pow2(width) || !is_pow2(height) || width != height || padding_bytes_per_row != 0) return ICAMD_FALSE;
    if (width % 8 != 0 || height % 4 != 0) return ICAMD_FALSE;
    if (width >= 65536u) return ICAMD_FALSE;  // the reference's uint32 width * height / 4 wraps (see pvrtc_encode_device_impl)
    if (out_size != (size_t)(width * height / 4)) return ICAMD_FALSE;
  } else {
    if (!resolve_codec(compressor, format, &codec, &comps, &swap)) return ICAMD_FALSE;  // dxtc.cc:735-750, etc.cc:747-758
    gh = std::max(height, padded_height);
    gw = std::max(width, padded_width);
    if (out_size != icamd_encoded_size(codec, gh, gw)) return ICAMD_FALSE;  // compressor4x4_helper.cc:34-41
  }
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  const size_t stride = (size_t)width * comps + padding_bytes_per_row;
  if (stride > 0xffffffffull) return fail(ICAMD_ERR_ARG, "row stride does not fit 32 bits");
  // the reference never touches the padding after the LAST row (pixel4x4.h:47-48 addresses row * stride + col)
  const size_t in_bytes = (size_t)(height - 1) * stride + (size_t)width * comps;
  rc = st.ensure(in_bytes, std::max<size_t>(out_size, 1), pinned);
  if (rc != ICAMD_OK) return rc;
  uint8_t *d_in = static_cast<uint8_t *>(st.d_in), *d_out = static_cast<uint8_t *>(st.d_out);
  const uint8_t *h_src = buffer;
  uint8_t *h_dst = out;
  if (pinned) {
    std::memcpy(st.h_in, buffer, in_bytes);
    h_src = static_cast<const uint8_t *>(st.h_in);
    h_dst = static_cast<uint8_t *>(st.h_out);
  }
  if (pvrtc) {
    hipStream_t s = st.stream;
    ICAMD_HIP(hipMemcpyAsync(d_in, h_src, in_bytes, hipMemcpyHostToDevice, s), "H2D copy");
    rc = pvrtc_encode_device_impl(4, height, width, width * 4u, 1, 0, 0, d_in, d_out, s, true);
    if (rc != ICAMD_OK) {
      (void)hipStreamSynchronize(s);
      return rc;
    }
    ICAMD_HIP(hipMemcpyAsync(h_dst, d_out, out_size, hipMemcpyDeviceToHost, s), "D2H copy");
    ICAMD_HIP(hipStreamSynchronize(s), "stream synchronize");
  } else {
    const uint32_t block_bytes = codec == ICAMD_DXT5 ? 16u : 8u;
    const uint32_t img_block_rows = num_blocks4(height), grid_block_rows = num_blocks4(gh), block_cols = num_blocks4(gw);
    uint64_t band_rows = std::max<uint64_t>(1, kHostBandBytes / (4u * stride));  // block rows per band
    if (band_rows * 2 > img_block_rows) band_rows = img_block_rows;               // small images: one band
    int status = ICAMD_OK;
    uint32_t band = 0;
    for (uint64_t r0 = 0; r0 < img_block_rows && status == ICAMD_OK; r0 += band_rows, ++band) {
      const bool last = r0 + band_rows >= img_block_rows;
      const uint32_t rows_b = (uint32_t)(last ? img_block_rows - r0 : band_rows);        // block rows with pixels
      const uint32_t h_b = (uint32_t)std::min<uint64_t>((uint64_t)rows_b * 4u, height - r0 * 4u);
      const uint32_t grid_rows_b = last ? (uint32_t)(grid_block_rows - r0) : rows_b;     // + the pad rows below the image
      const size_t src_off = (size_t)r0 * 4u * stride;
      const size_t src_len = (size_t)(h_b - 1) * stride + (size_t)width * comps;
      const size_t dst_off = (size_t)r0 * block_cols * block_bytes, dst_len = (size_t)grid_rows_b * block_cols * block_bytes;
      hipStream_t s = (band & 1u) ? st.stream2 : st.stream;
      hipError_t e = hipMemcpyAsync(d_in + src_off, h_src + src_off, src_len, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) { status = fail(ICAMD_ERR_HIP, "H2D copy", e); break; }
      status = icamd_encode_device(codec, etc_strategy, comps, swap, h_b, width, grid_rows_b * 4u, gw, (uint32_t)stride, 1, 0,
                                   0, d_in + src_off, d_out + dst_off, s);
      if (status != ICAMD_OK) break;
      e = hipMemcpyAsync(h_dst + dst_off, d_out + dst_off, dst_len, hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) status = fail(ICAMD_ERR_HIP, "D2H copy", e);
    }
    const hipError_t e1 = hipStreamSynchronize(st.stream), e2 = hipStreamSynchronize(st.stream2);
    if (status != ICAMD_OK) return status;
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(ICAMD_ERR_HIP, "stream synchronize", e1 != hipSuccess ? e1 : e2);
  }
  if (pinned) std::memcpy(out, st.h_out, out_size);
  return ICAMD_OK;
}

int icamd_compress(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                   uint32_t padding_bytes_per_row, const uint8_t *buffer, uint8_t *out, size_t out_size) try {
  return compress_host_common(tls_staging(), false, false, compressor, etc_strategy, format, height, width, height, width,
                              padding_bytes_per_row, buffer, out, out_size);
} ICAMD_ABI_CATCH

int icamd_compress_and_pad(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                           uint32_t padded_height, uint32_t padded_width, uint32_t padding_bytes_per_row,
                           const uint8_t *buffer, uint8_t *out, size_t out_size) try {
  return compress_host_common(tls_staging(), false, true, compressor, etc_strategy, format, height, width, padded_height,
                              padded_width, padding_bytes_per_row, buffer, out, out_size);
} ICAMD_ABI_CATCH

// One launch: fewer than 2^31 blocks (the decode kernels index blocks with 32 bits).
static int decode_launch(int codec, int swap_rb, uint32_t height, uint32_t width, uint32_t row_stride,
                         uint32_t n_images, size_t src_image_stride, size_t dst_image_stride, const uint8_t *blocks,
                         uint8_t *pixels, hipStream_t stream) {
  icamd::DecodeParams P;
  P.blocks = blocks;
  P.pixels = pixels;
  P.src_image_stride = src_image_stride;
  P.dst_image_stride = dst_image_stride;
  P.height = height;
  P.width = width;
  P.block_rows = num_blocks4(height);
  P.block_cols = codec == ICAMD_PVRTC2 ? width / 8u : num_blocks4(width);  // PVRTC 2 bpp: 8x4-pixel blocks
  P.row_stride = row_stride;
  P.blocks_per_image = P.block_rows * P.block_cols;
  P.total_blocks = P.blocks_per_image * n_images;
  P.swap_rb = swap_rb ? 1u : 0u;
  P.div_bpi = icamd::make_fastdiv(P.blocks_per_image);
  P.div_cols = icamd::make_fastdiv(P.block_cols);
  ICAMD_HIP(icamd::launch_decode(codec, P, stream), "launch decode");
  return ICAMD_OK;
}

int icamd_decode_device(int codec, int swap_rb, uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                        uint32_t n_images, size_t src_image_stride_bytes, size_t dst_image_stride_bytes,
                        const void *d_blocks, void *d_pixels, void *hip_stream) try {
  if (!d_blocks || !d_pixels || height == 0 || width == 0) return ICAMD_FALSE;
  if (codec != ICAMD_DXT1 && codec != ICAMD_DXT5 && codec != ICAMD_ETC1 && codec != ICAMD_PVRTC2 && codec != ICAMD_PVRTC4)
    return ICAMD_FALSE;
  if (n_images == 0) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const uint8_t *blocks = static_cast<const uint8_t *>(d_blocks);
  uint8_t *pixels = static_cast<uint8_t *>(d_pixels);
  if (codec == ICAMD_PVRTC2 || codec == ICAMD_PVRTC4) {
    // EXTENSION (the reference's PvrtcCompressor::Decompress returns false, pvrtc.cc:669-672; 4 bpp: no such format there at
    // all): the sizes PvrtcCompressor::Compress accepts (pvrtc.cc:636-650), RGBA8 out, no row padding, one launch per <= 2^30 blocks
    if (!is_pow2(width) || width != height || width < 8 || padding_bytes_per_row != 0) return ICAMD_FALSE;
    const uint64_t pv_bpi = (uint64_t)(width / (codec == ICAMD_PVRTC2 ? 8 : 4)) * (height / 4);
    if (pv_bpi >= (1ull << 31)) return fail(ICAMD_ERR_ARG, "PVRTC texture too large to decode");
    const uint64_t per_launch = std::max<uint64_t>(1, ((1ull << 31) - 1) / pv_bpi);
    for (uint64_t first = 0; first < n_images; first += per_launch) {
      const uint32_t count = (uint32_t)std::min<uint64_t>(per_launch, n_images - first);
      rc = decode_launch(codec, 0, height, width, width * 4u, count, src_image_stride_bytes, dst_image_stride_bytes,
                         blocks + first * src_image_stride_bytes, pixels + first * dst_image_stride_bytes, stream);
      if (rc != ICAMD_OK) return rc;
    }
    return ICAMD_OK;
  }
  const uint32_t block_bytes = codec == ICAMD_DXT5 ? 16u : 8u;
  const uint32_t row_stride = width * (codec == ICAMD_DXT5 ? 4u : 3u) + padding_bytes_per_row;
  const uint64_t block_cols = num_blocks4(width), bpi = (uint64_t)num_blocks4(height) * block_cols;
  const uint64_t kMaxBlocks = (1ull << 31) - 1;
  if (bpi <= kMaxBlocks) {  // whole images, as many per launch as the 32-bit block index allows
    const uint64_t per_launch = std::max<uint64_t>(1, kMaxBlocks / bpi);
    for (uint64_t first = 0; first < n_images; first += per_launch) {
      const uint32_t count = (uint32_t)std::min<uint64_t>(per_launch, n_images - first);
      rc = decode_launch(codec, swap_rb, height, width, row_stride, count, src_image_stride_bytes,
                         dst_image_stride_bytes, blocks + first * src_image_stride_bytes,
                         pixels + first * dst_image_stride_bytes, stream);
      if (rc != ICAMD_OK) return rc;
    }
    return ICAMD_OK;
  }
  // a single image of 2^31 blocks or more: bands of block rows (blocks are row-major, helper.h:218-262)
  const uint32_t band = (uint32_t)(kMaxBlocks / block_cols);
  for (uint32_t i = 0; i < n_images; ++i)
    for (uint64_t r0 = 0; r0 < num_blocks4(height); r0 += band) {
      const uint32_t rows = (uint32_t)std::min<uint64_t>((uint64_t)band * 4u, (uint64_t)height - r0 * 4u);
      rc = decode_launch(codec, swap_rb, rows, width, row_stride, 1, 0, 0,
                         blocks + i * src_image_stride_bytes + r0 * block_cols * block_bytes,
                         pixels + i * dst_image_stride_bytes + r0 * 4u * row_stride, stream);
      if (rc != ICAMD_OK) return rc;
    }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_decompress(int compressor, int format, uint32_t height, uint32_t width, uint32_t padding_bytes_per_row,
                     const uint8_t *blocks, size_t blocks_size, uint8_t *out, size_t out_size) try {
  if (!blocks || !out || height == 0 || width == 0) return ICAMD_FALSE;
  int codec, comps;
  bool swap;
  if (!resolve_codec(compressor, format, &codec, &comps, &swap)) return ICAMD_FALSE;
  if (blocks_size != icamd_encoded_size(codec, height, width)) return ICAMD_FALSE;
  const size_t need = (size_t)height * ((size_t)width * comps + padding_bytes_per_row);
  if (out_size != need) return ICAMD_FALSE;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  Staging &st = tls_staging();
  rc = st.ensure(blocks_size, need);
  if (rc != ICAMD_OK) return rc;
  hipStream_t s = st.stream;
  ICAMD_HIP(hipMemcpyAsync(st.d_in, blocks, blocks_size, hipMemcpyHostToDevice, s), "H2D copy");
  if (padding_bytes_per_row) ICAMD_HIP(hipMemsetAsync(st.d_out, 0, need, s), "memset");
  rc = icamd_decode_device(codec, swap, height, width, padding_bytes_per_row, 1, 0, 0, st.d_in, st.d_out, s);
  if (rc != ICAMD_OK) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  ICAMD_HIP(hipMemcpyAsync(out, st.d_out, need, hipMemcpyDeviceToHost, s), "D2H copy");
  ICAMD_HIP(hipStreamSynchronize(s), "stream synchronize");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

// EXTENSION (parity unpinned, see icamd_decode_device): host-buffer PVRTC 2bpp decode.  Kept apart from
// icamd_decompress, which answers ICAMD_FALSE for PVRTC like the reference (pvrtc_compressor.cc:669-672).
int icamd_pvrtc2_decompress(uint32_t size, const uint8_t *blocks, size_t blocks_size, uint8_t *out, size_t out_size) try {
  if (!blocks || !out || !is_pow2(size) || size < 8) return ICAMD_FALSE;
  if (blocks_size != (size_t)size * size / 4 || out_size != (size_t)size * size * 4) return ICAMD_FALSE;
  return staged_blockop(blocks, blocks_size, out, out_size, false, [&](void *din, void *dout, hipStream_t s) {
    return icamd_decode_device(ICAMD_PVRTC2, 0, size, size, 0, 1, 0, 0, din, dout, s);
  });
} ICAMD_ABI_CATCH

// ---- compressed-domain operations

int icamd_pad_batch_device(int compressor, int etc_strategy, int format, uint32_t ch, uint32_t cw, uint32_t n_images,
                           const void *d_blocks, size_t src_image_stride_bytes, uint32_t ph, uint32_t pw, void *d_out,
                           size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream) try {
  int codec;
  if (!d_blocks || !d_out || !blockop_codec(compressor, format, &codec)) return ICAMD_FALSE;
  if ((reinterpret_cast<uintptr_t>(d_blocks) | reinterpret_cast<uintptr_t>(d_out) | src_image_stride_bytes | dst_image_stride_bytes) % 4u)
    return fail(ICAMD_ERR_ARG, "block pointers and image strides must be 4-byte aligned");
  icamd::BlockOpParams P;
  P.in_rows = num_blocks4(ch); P.in_cols = num_blocks4(cw);
  P.out_rows = num_blocks4(ph); P.out_cols = num_blocks4(pw);
  if (P.in_rows == 0 || P.in_cols == 0 || P.out_rows < P.in_rows || P.out_cols < P.in_cols) return ICAMD_FALSE;
  if (out_size_per_image != icamd_encoded_size(codec, ph, pw)) return ICAMD_FALSE;
  // (a non-zero stride is checked for one image too: a caller that passes one states how far its buffer reaches)
  if (((n_images > 1 || src_image_stride_bytes != 0) && src_image_stride_bytes < icamd_encoded_size(codec, ch, cw)) ||
      ((n_images > 1 || dst_image_stride_bytes != 0) && dst_image_stride_bytes < out_size_per_image))
    return fail(ICAMD_ERR_ARG, "image stride smaller than an image");
  if (n_images == 0) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  const uint64_t per = (uint64_t)P.out_rows * P.out_cols;
  if (per >= (1ull << 31)) return fail(ICAMD_ERR_ARG, "more than 2^31 blocks in one image");
  P.etc_strategy = (uint32_t)etc_strategy;
  P.src_height = ch; P.src_width = cw;
  P.div_out_cols = icamd::make_fastdiv(P.out_cols);
  P.out_per_image = (uint32_t)per;
  P.div_out_per_image = icamd::make_fastdiv(P.out_per_image);
  P.src_image_stride = src_image_stride_bytes;
  P.dst_image_stride = dst_image_stride_bytes;
  // as many images per launch as the 32-bit block index allows
  const uint64_t group = std::max<uint64_t>(1, ((1ull << 31) - 1) / per);
  for (uint64_t first = 0; first < n_images; first += group) {
    const uint64_t count = std::min<uint64_t>(group, n_images - first);
    P.src = static_cast<const uint8_t *>(d_blocks) + first * src_image_stride_bytes;
    P.dst = static_cast<uint8_t *>(d_out) + first * dst_image_stride_bytes;
    P.n_images = (uint32_t)count;
    P.total_out = (uint32_t)(per * count);
    ICAMD_HIP(icamd::launch_pad(codec, P, static_cast<hipStream_t>(hip_stream)), "launch pad");
  }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_pad_device(int compressor, int etc_strategy, int format, uint32_t ch, uint32_t cw, const void *d_blocks,
                     uint32_t ph, uint32_t pw, void *d_out, size_t out_size, void *hip_stream) try {
  return icamd_pad_batch_device(compressor, etc_strategy, format, ch, cw, 1, d_blocks, 0, ph, pw, d_out, 0, out_size, hip_stream);
} ICAMD_ABI_CATCH

int icamd_downsample_batch_device(int compressor, int etc_strategy, int format, uint32_t uh, uint32_t uw, uint32_t n_images,
                                  const void *d_blocks, size_t src_image_stride_bytes, void *d_out,
                                  size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream) try {
  int codec;
  if (!d_blocks || !d_out || uh == 0 || uw == 0 || !blockop_codec(compressor, format, &codec)) return ICAMD_FALSE;
  if ((reinterpret_cast<uintptr_t>(d_blocks) | reinterpret_cast<uintptr_t>(d_out) | src_image_stride_bytes | dst_image_stride_bytes) % 4u)
    return fail(ICAMD_ERR_ARG, "block pointers and image strides must be 4-byte aligned");
  icamd::BlockOpParams P;
  P.in_rows = num_blocks4(uh); P.in_cols = num_blocks4(uw);
  // helper.h:281-284, :340-341
  if ((P.in_rows > 1 && P.in_rows % 2) || (P.in_cols > 1 && P.in_cols % 2)) return ICAMD_FALSE;
  if (P.in_rows == 1 && P.in_cols == 1 && (uh == 3 || uw == 3)) return ICAMD_FALSE;
  const uint32_t dh = (uh + 1) / 2, dw = (uw + 1) / 2;
  P.out_rows = num_blocks4(dh); P.out_cols = num_blocks4(dw);
  if (out_size_per_image != icamd_encoded_size(codec, dh, dw)) return ICAMD_FALSE;
  // (a non-zero stride is checked for one image too: a caller that passes one states how far its buffer reaches)
  if ((n_images > 1 || src_image_stride_bytes != 0) && src_image_stride_bytes < icamd_encoded_size(codec, uh, uw))
    return fail(ICAMD_ERR_ARG, "source image stride smaller than an image");
  if ((n_images > 1 || dst_image_stride_bytes != 0) && dst_image_stride_bytes < out_size_per_image)
    return fail(ICAMD_ERR_ARG, "destination image stride smaller than an image");
  if (n_images == 0) return ICAMD_OK;  // (after the checks: a call the reference would refuse is refused for any count)
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  const uint64_t per = (uint64_t)P.out_rows * P.out_cols;
  if (per >= (1ull << 31)) return fail(ICAMD_ERR_ARG, "more than 2^31 blocks in one image");
  P.etc_strategy = (uint32_t)etc_strategy;
  P.src_height = uh; P.src_width = uw;
  P.div_out_cols = icamd::make_fastdiv(P.out_cols);
  P.out_per_image = (uint32_t)per;
  P.div_out_per_image = icamd::make_fastdiv(P.out_per_image);
  P.src_image_stride = src_image_stride_bytes;
  P.dst_image_stride = dst_image_stride_bytes;
  // as many images per launch as the 32-bit block index allows
  const uint64_t group = std::max<uint64_t>(1, ((1ull << 31) - 1) / per);
  for (uint64_t first = 0; first < n_images; first += group) {
    const uint64_t count = std::min<uint64_t>(group, n_images - first);
    P.src = static_cast<const uint8_t *>(d_blocks) + first * src_image_stride_bytes;
    P.dst = static_cast<uint8_t *>(d_out) + first * dst_image_stride_bytes;
    P.n_images = (uint32_t)count;
    P.total_out = (uint32_t)(per * count);
    ICAMD_HIP(icamd::launch_downsample(codec, P, static_cast<hipStream_t>(hip_stream)), "launch downsample");
  }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_downsample_device(int compressor, int etc_strategy, int format, uint32_t uh, uint32_t uw,
                            const void *d_blocks, void *d_out, size_t out_size, void *hip_stream) try {
  return icamd_downsample_batch_device(compressor, etc_strategy, format, uh, uw, 1, d_blocks, 0, d_out, 0, out_size, hip_stream);
} ICAMD_ABI_CATCH

int icamd_transcode_dxt1_to_etc1_device(void *d_blocks, size_t n_bytes, void *hip_stream) try {
  if (!d_blocks) return ICAMD_FALSE;
  if (reinterpret_cast<uintptr_t>(d_blocks) % 8u) return fail(ICAMD_ERR_ARG, "block pointer must be 8-byte aligned");
  if (n_bytes < 8) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  const size_t kChunk = (size_t)1 << 30;  // blocks per launch (32-bit block index in the kernel)
  for (size_t first = 0; first < n_bytes / 8; first += kChunk) {
    const size_t count = std::min(kChunk, n_bytes / 8 - first);
    ICAMD_HIP(icamd::launch_transcode_dxt1_to_etc1(static_cast<uint8_t *>(d_blocks) + first * 8, (uint32_t)count,
                                                   static_cast<hipStream_t>(hip_stream)), "launch transcode");
  }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_pad(int compressor, int etc_strategy, int format, uint32_t ch, uint32_t cw, const uint8_t *blocks,
              uint32_t ph, uint32_t pw, uint8_t *out, size_t out_size) try {
  int codec;
  if (!blocks || !out || !blockop_codec(compressor, format, &codec)) return ICAMD_FALSE;
  if (num_blocks4(ph) < num_blocks4(ch) || num_blocks4(pw) < num_blocks4(cw)) return ICAMD_FALSE;
  if (out_size != icamd_encoded_size(codec, ph, pw)) return ICAMD_FALSE;
  return staged_blockop(blocks, icamd_encoded_size(codec, ch, cw), out, out_size, false,
                        [&](void *din, void *dout, hipStream_t s) {
                          return icamd_pad_device(compressor, etc_strategy, format, ch, cw, din, ph, pw, dout, out_size, s);
                        });
} ICAMD_ABI_CATCH

int icamd_downsample(int compressor, int etc_strategy, int format, uint32_t uh, uint32_t uw, const uint8_t *blocks,
                     uint8_t *out, size_t out_size) try {
  int codec;
  if (!blocks || !out || uh == 0 || uw == 0 || !blockop_codec(compressor, format, &codec)) return ICAMD_FALSE;
  const uint32_t r = num_blocks4(uh), c = num_blocks4(uw);
  if ((r > 1 && r % 2) || (c > 1 && c % 2) || (r == 1 && c == 1 && (uh == 3 || uw == 3))) return ICAMD_FALSE;
  if (out_size != icamd_encoded_size(codec, (uh + 1) / 2, (uw + 1) / 2)) return ICAMD_FALSE;
  return staged_blockop(blocks, icamd_encoded_size(codec, uh, uw), out, out_size, false,
                        [&](void *din, void *dout, hipStream_t s) {
                          return icamd_downsample_device(compressor, etc_strategy, format, uh, uw, din, dout, out_size, s);
                        });
} ICAMD_ABI_CATCH

int icamd_transcode_dxt1_to_etc1(uint8_t *blocks, size_t n_bytes) try {
  if (!blocks) return ICAMD_FALSE;
  if (n_bytes < 8) return ICAMD_OK;
  return staged_blockop(blocks, n_bytes, blocks, n_bytes - n_bytes % 8, true, [&](void *din, void *, hipStream_t s) {
    return icamd_transcode_dxt1_to_etc1_device(din, n_bytes, s);
  });
} ICAMD_ABI_CATCH

int icamd_compress_batch(int compressor, int etc_strategy, int format, uint32_t height, uint32_t width,
                         uint32_t padding_bytes_per_row, uint32_t n_images, const uint8_t *const *buffers,
                         uint8_t *const *outs, size_t out_size, const int *devices, int n_devices, int *statuses) try {
  if (n_images == 0) return ICAMD_OK;
  if (!buffers || !outs || !devices || n_devices <= 0) return fail(ICAMD_ERR_ARG, "icamd_compress_batch: null list");
  if (n_devices > kMaxDeviceListEntries) return fail(ICAMD_ERR_ARG, "icamd_compress_batch: device list longer than 256 entries");
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  int visible = 0;
  ICAMD_HIP(hipGetDeviceCount(&visible), "hipGetDeviceCount");
  for (int d = 0; d < n_devices; ++d)
    if (devices[d] < 0 || devices[d] >= visible) return fail(ICAMD_ERR_ARG, "icamd_compress_batch: bad device ordinal");
  std::vector<int> local(n_images, ICAMD_OK);
  std::vector<WorkerError> errors((size_t)n_devices);
  std::vector<std::thread> workers;
  workers.reserve((size_t)n_devices);
  int not_started_from = n_devices;  // list entries from here on have no worker (thread creation failed)
  {
    JoinAll join_on_every_path_out{workers};
    for (int d = 0; d < n_devices && (uint32_t)d < n_images; ++d) {
      try {
        workers.emplace_back([&, d]() noexcept {
          auto fail_all = [&](int code, const char *what) {
            for (uint64_t i = (uint64_t)d; i < n_images; i += (uint64_t)n_devices) local[i] = code;
            if (errors[(size_t)d].empty()) errors[(size_t)d].set(what);
          };
          uint64_t reached = (uint64_t)d;  // the image being worked on: what an exception leaves undone starts here
          try {
            // each worker owns its device context and a pooled Staging (stream, device buffers, pinned host mirrors)
            if (hipSetDevice(devices[d]) != hipSuccess) return fail_all(ICAMD_ERR_HIP, "hipSetDevice failed");
            std::unique_ptr<Staging> st = pool_take(devices[d]);
            for (; reached < n_images; reached += (uint64_t)n_devices) {
              const uint32_t i = (uint32_t)reached;
              local[i] = compress_host_common(*st, kBatchPinnedStaging, false, compressor, etc_strategy, format, height, width, height, width,
                                              padding_bytes_per_row, buffers[i], outs[i], out_size);
              if (local[i] < 0 && errors[(size_t)d].empty()) errors[(size_t)d].set(g_last_error);
            }
            pool_give(std::move(st));
          } catch (...) {
            const int code = abi_exception();
            for (uint64_t i = reached; i < n_images; i += (uint64_t)n_devices) local[i] = code;
            if (errors[(size_t)d].empty()) errors[(size_t)d].set(g_last_error);
          }
        });
      } catch (...) {  // std::system_error from the thread's creation (or bad_alloc): the workers started so far finish
        (void)abi_exception();
        not_started_from = d;
        break;
      }
    }
  }  // all started workers joined
  for (int d = not_started_from; d < n_devices; ++d) {
    for (uint32_t i = (uint32_t)d; i < n_images; i += (uint32_t)n_devices) local[i] = ICAMD_ERR_ALLOC;
    errors[(size_t)d].set("icamd_compress_batch: could not start a worker thread");
  }
  int first = ICAMD_OK;
  for (uint32_t i = 0; i < n_images; ++i) {
    if (statuses) statuses[i] = local[i];
    if (first == ICAMD_OK && local[i] != ICAMD_OK) first = local[i];
  }
  if (first < 0)
    for (const WorkerError &e : errors)
      if (!e.empty()) { set_last_error(e.text); break; }
  return first;
} ICAMD_ABI_CATCH

// ---- CreateSolidImage / CopySubimage

namespace {
// Quantize8<n> (color_util.h:156-164)
uint32_t quantize8(uint32_t v, uint32_t bits) {
  const uint32_t i = v * ((1u << bits) - 1u) + 128u;
  return (i + (i >> 8)) >> 8;
}
// The one block CreateSolidImage replicates, as little-endian dwords; returns its size in bytes, 0 where the reference
// returns false.  DXT (dxtc_compressor.cc:42-49,77-82,820-839): c0 = c1 = RGB565(color) -- no red/blue swap --, index
// bits zero; DXT5 puts alpha0 = alpha1 = color[3] and zero codes in front.  ETC (etc_compressor.cc:595-617,802-812):
// differential mode, 5-bit base = color >> 3 (the "adjusted" colour computed there is never used), zero difference,
// codeword 0 twice, all indices 0; big-endian high word, then the zero low word.
uint32_t solid_block(int compressor, int format, const uint8_t *color, uint32_t w[4]) {
  w[0] = w[1] = w[2] = w[3] = 0;
  const int comps = format_components(format);
  if (comps == 0 || !color) return 0;
  if (compressor == ICAMD_COMPRESSOR_ETC) {
    if (format != ICAMD_RGB) return 0;
    const uint32_t hi = 2u | (uint32_t)(color[0] >> 3) << 27 | (uint32_t)(color[1] >> 3) << 19 | (uint32_t)(color[2] >> 3) << 11;
    w[0] = __builtin_bswap32(hi);
    return 8;
  }
  if (compressor != ICAMD_COMPRESSOR_DXTC) return 0;  // pvrtc_compressor.cc:693-698
  const uint32_t c565 = quantize8(color[0], 5) << 11 | quantize8(color[1], 6) << 5 | quantize8(color[2], 5);
  if (comps == 3) {
    w[0] = c565 | c565 << 16;
    return 8;
  }
  w[0] = (uint32_t)color[3] | (uint32_t)color[3] << 8;
  w[2] = c565 | c565 << 16;
  return 16;
}
// CopySubimage's argument check (helper.h:555-563) and geometry
bool subimage_geometry(int compressor, int format, uint32_t ch, uint32_t cw, uint32_t row, uint32_t col, uint32_t h,
                       uint32_t w, int *block_bytes) {
  int codec;
  if (!blockop_codec(compressor, format, &codec)) return false;
  *block_bytes = codec == ICAMD_DXT5 ? 16 : 8;
  if (row % 4 || col % 4 || h % 4 || w % 4) return false;
  // 64-bit sums: the reference's uint32 start + extent can wrap and then accept a window outside the image
  return !(row > ch || col > cw || (uint64_t)row + h > ch || (uint64_t)col + w > cw);
}
}  // namespace

int icamd_create_solid_batch_device(int compressor, int format, uint32_t height, uint32_t width, uint32_t n_images,
                                    const uint8_t *colors, void *d_out, size_t dst_image_stride_bytes, size_t out_size_per_image,
                                    void *hip_stream) try {
  const int comps = format_components(format);
  uint32_t probe[4];
  if (!colors || !d_out || comps == 0) return ICAMD_FALSE;
  const uint32_t bb = solid_block(compressor, format, colors, probe);
  if (bb == 0) return ICAMD_FALSE;
  const uint64_t n = (uint64_t)num_blocks4(height) * num_blocks4(width);
  if (out_size_per_image != n * bb) return ICAMD_FALSE;  // compressor4x4_helper.cc:34-41
  if ((reinterpret_cast<uintptr_t>(d_out) | dst_image_stride_bytes) % 4u) return fail(ICAMD_ERR_ARG, "block pointers and image strides must be 4-byte aligned");
  if ((n_images > 1 || dst_image_stride_bytes != 0) && dst_image_stride_bytes < out_size_per_image)
    return fail(ICAMD_ERR_ARG, "image stride smaller than an image");
  if (n >= (1ull << 32)) return fail(ICAMD_ERR_ARG, "more than 2^32 blocks in one image");
  if (n_images == 0) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  std::vector<uint32_t> words((size_t)n_images * 4u);
  for (uint32_t i = 0; i < n_images; ++i) (void)solid_block(compressor, format, colors + (size_t)i * (size_t)comps, &words[(size_t)i * 4u]);
  ICAMD_HIP(icamd::launch_fill_blocks_batch(d_out, dst_image_stride_bytes, (uint32_t)n, (int)bb,
                                            reinterpret_cast<const uint32_t (*)[4]>(words.data()), n_images,
                                            static_cast<hipStream_t>(hip_stream)), "launch fill");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_create_solid_device(int compressor, int format, uint32_t height, uint32_t width, const uint8_t *color,
                              void *d_out, size_t out_size, void *hip_stream) try {
  uint32_t w[4];
  const uint32_t bb = solid_block(compressor, format, color, w);
  if (bb == 0 || !d_out) return ICAMD_FALSE;
  const uint64_t n = (uint64_t)num_blocks4(height) * num_blocks4(width);
  if (out_size != n * bb) return ICAMD_FALSE;  // compressor4x4_helper.cc:34-41
  if (reinterpret_cast<uintptr_t>(d_out) % 4u) return fail(ICAMD_ERR_ARG, "block pointers must be 4-byte aligned");
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  ICAMD_HIP(icamd::launch_fill_blocks(d_out, n, (int)bb, w, static_cast<hipStream_t>(hip_stream)), "launch fill");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

// Host buffers in, host buffers out: replicating one block is byte shuffling with nothing to offload (a PCIe round
// trip would be pure overhead), exactly what the reference does (helper.h:536-540).
int icamd_create_solid(int compressor, int format, uint32_t height, uint32_t width, const uint8_t *color, uint8_t *out,
                       size_t out_size) try {
  uint32_t w[4];
  const uint32_t bb = solid_block(compressor, format, color, w);
  if (bb == 0 || !out) return ICAMD_FALSE;
  const uint64_t n = (uint64_t)num_blocks4(height) * num_blocks4(width);
  if (out_size != n * bb) return ICAMD_FALSE;
  for (uint64_t i = 0; i < n; ++i) std::memcpy(out + i * bb, w, bb);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_copy_subimage_batch_device(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                                     uint32_t n_images, const void *d_blocks, size_t src_image_stride_bytes, uint32_t start_row,
                                     uint32_t start_column, uint32_t height, uint32_t width, void *d_out,
                                     size_t dst_image_stride_bytes, size_t out_size_per_image, void *hip_stream) try {
  int bb;
  if (!d_blocks || !d_out ||
      !subimage_geometry(compressor, format, compressed_height, compressed_width, start_row, start_column, height, width, &bb))
    return ICAMD_FALSE;
  const uint32_t rows = num_blocks4(height), cols = num_blocks4(width);
  if (out_size_per_image != (size_t)rows * cols * (size_t)bb) return ICAMD_FALSE;
  if ((reinterpret_cast<uintptr_t>(d_blocks) | reinterpret_cast<uintptr_t>(d_out) | src_image_stride_bytes | dst_image_stride_bytes) % 4u)
    return fail(ICAMD_ERR_ARG, "block pointers and image strides must be 4-byte aligned");
  if (((n_images > 1 || src_image_stride_bytes != 0) &&
       src_image_stride_bytes < (size_t)num_blocks4(compressed_height) * num_blocks4(compressed_width) * (size_t)bb) ||
      ((n_images > 1 || dst_image_stride_bytes != 0) && dst_image_stride_bytes < out_size_per_image))
    return fail(ICAMD_ERR_ARG, "image stride smaller than an image");
  if (n_images == 0) return ICAMD_OK;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  ICAMD_HIP(icamd::launch_copy_subimage(bb, d_blocks, num_blocks4(compressed_width), start_row / 4u, start_column / 4u,
                                        rows, cols, d_out, static_cast<hipStream_t>(hip_stream), n_images,
                                        src_image_stride_bytes, dst_image_stride_bytes), "launch copy_subimage");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_copy_subimage_device(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                               const void *d_blocks, uint32_t start_row, uint32_t start_column, uint32_t height,
                               uint32_t width, void *d_out, size_t out_size, void *hip_stream) try {
  return icamd_copy_subimage_batch_device(compressor, format, compressed_height, compressed_width, 1, d_blocks, 0, start_row,
                                          start_column, height, width, d_out, 0, out_size, hip_stream);
} ICAMD_ABI_CATCH

// Host form: block-row memcpys like the reference (helper.h:583-589); nothing to offload.
int icamd_copy_subimage(int compressor, int format, uint32_t compressed_height, uint32_t compressed_width,
                        const uint8_t *blocks, uint32_t start_row, uint32_t start_column, uint32_t height,
                        uint32_t width, uint8_t *out, size_t out_size) try {
  int bb;
  if (!blocks || !out ||
      !subimage_geometry(compressor, format, compressed_height, compressed_width, start_row, start_column, height, width, &bb))
    return ICAMD_FALSE;
  const uint32_t rows = num_blocks4(height), cols = num_blocks4(width), src_cols = num_blocks4(compressed_width);
  if (out_size != (size_t)rows * cols * (size_t)bb) return ICAMD_FALSE;
  const uint8_t *src = blocks + ((size_t)(start_row / 4u) * src_cols + start_column / 4u) * (size_t)bb;
  for (uint32_t r = 0; r < rows; ++r)
    std::memcpy(out + (size_t)r * cols * bb, src + (size_t)r * src_cols * bb, (size_t)cols * bb);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

// ---- diagnostics

uint32_t icamd_wall_clock_rate_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return 0;
  return (uint32_t)khz;
}

int icamd_clock_probe_device(void *d_out16, uint32_t duration_us, void *hip_stream) try {
  if (!d_out16 || reinterpret_cast<uintptr_t>(d_out16) % 8u) return fail(ICAMD_ERR_ARG, "clock probe: 8-byte aligned 16-byte buffer needed");
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  const uint32_t khz = icamd_wall_clock_rate_khz();
  if (khz == 0) return fail(ICAMD_ERR_HIP, "hipDeviceAttributeWallClockRate unavailable");
  // an exported entry point must not be able to park a wave on the device for an hour (uint32 microseconds = 71 minutes)
  if (duration_us > ICAMD_CLOCK_PROBE_MAX_US) return fail(ICAMD_ERR_ARG, "clock probe: duration above ICAMD_CLOCK_PROBE_MAX_US (10 s)");
  const uint64_t ticks = (uint64_t)duration_us * khz / 1000u;
  ICAMD_HIP(icamd::launch_clock_probe(static_cast<uint64_t *>(d_out16), ticks, static_cast<hipStream_t>(hip_stream)), "launch clock probe");
  return ICAMD_OK;
} ICAMD_ABI_CATCH

// ---- multi-GPU from one process, device-resident

int icamd_encode_batch_sharded_device(int codec, int etc_strategy, int src_components, int swap_rb, uint32_t height,
                                      uint32_t width, uint32_t row_stride_bytes, uint32_t n_images,
                                      const void *const *d_srcs, void *const *d_dsts, const int *devices, int n_devices,
                                      int gather_device, void *d_gathered, size_t gathered_image_stride_bytes,
                                      int *statuses) try {
  if (n_images == 0) return ICAMD_OK;
  if (!d_srcs || !devices || n_devices <= 0) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: null list");
  if (n_devices > kMaxDeviceListEntries) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: device list longer than 256 entries");
  const bool gather = gather_device >= 0;
  if (!gather && !d_dsts) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: no output (d_dsts and no gather)");
  if (gather && !d_gathered) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: gather without a buffer");
  if (height == 0 || width == 0) return ICAMD_FALSE;
  int rc = require_device();
  if (rc != ICAMD_OK) return rc;
  int visible = 0;
  ICAMD_HIP(hipGetDeviceCount(&visible), "hipGetDeviceCount");
  for (int d = 0; d < n_devices; ++d)
    if (devices[d] < 0 || devices[d] >= visible) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: bad device ordinal");
  if (gather && gather_device >= visible) return fail(ICAMD_ERR_ARG, "icamd_encode_batch_sharded_device: bad gather device");
  const size_t out_size = icamd_encoded_size(codec, height, width);
  if (gather && gathered_image_stride_bytes < out_size) return fail(ICAMD_ERR_ARG, "gathered image stride smaller than an image");
  int prev_device = 0;
  (void)hipGetDevice(&prev_device);
  std::vector<int> local(n_images, ICAMD_OK);
  std::vector<WorkerError> errors((size_t)n_devices);
  std::vector<std::thread> workers;
  workers.reserve((size_t)n_devices);
  int not_started_from = n_devices;  // list entries from here on have no worker (thread creation failed)
  {
  JoinAll join_on_every_path_out{workers};
  for (int d = 0; d < n_devices && (uint32_t)d < n_images; ++d) {
    try {
    workers.emplace_back([&, d]() noexcept {
      const int dev = devices[d];
      auto fail_all = [&](int code, const char *what) {
        for (uint64_t i = (uint64_t)d; i < n_images; i += (uint64_t)n_devices) local[i] = code;
        errors[(size_t)d].set(what);
      };
      try {
      if (hipSetDevice(dev) != hipSuccess) return fail_all(ICAMD_ERR_HIP, "hipSetDevice failed");
      if (gather && dev != gather_device) {
        // direct xGMI copies into the gather buffer: without peer access hipMemcpyPeerAsync bounces through host memory.
        // Best effort -- "already enabled" and "not supported" both leave a working (if slower) copy path.
        // (ICAMD_DISABLE_PEER_ACCESS=1: test knob -- take the host-bounced copy path on a box that has peer access)
        int can = 0;
        const char *no_peer = getenv("ICAMD_DISABLE_PEER_ACCESS");
        if (!(no_peer && no_peer[0] == '1') && hipDeviceCanAccessPeer(&can, dev, gather_device) == hipSuccess && can)
          (void)hipDeviceEnablePeerAccess(gather_device, 0);
        (void)hipGetLastError();
      }
      std::unique_ptr<Staging> st = pool_take(dev);
      // this worker's images, in order
      std::vector<uint32_t> mine;
      for (uint64_t i = (uint64_t)d; i < n_images; i += (uint64_t)n_devices) mine.push_back((uint32_t)i);
      auto slot_of = [&](uint32_t i) { return static_cast<uint8_t *>(d_gathered) + (size_t)i * gathered_image_stride_bytes; };
      // where image i is encoded to: its own buffer, its gather slot (images of the gather device), or scratch (nullptr here)
      auto target_of = [&](uint32_t i) -> uint8_t * {
        if (d_dsts && d_dsts[i]) return static_cast<uint8_t *>(d_dsts[i]);
        return (gather && dev == gather_device) ? slot_of(i) : nullptr;
      };
      // Consecutive images of a worker whose sources AND targets are evenly spaced (a batch laid out as one array, the
      // usual case) go out as ONE launch of up to kRun images: a 1024^2 texture alone fills a quarter of the chip's wave
      // slots, so one launch per texture leaves an MI355X mostly idle (bench.py single_image: 97 vs 252 Gpix/s for ETC1).
      const size_t kRun = 64;
      auto run_length = [&](size_t j, size_t *src_stride, size_t *dst_stride) -> size_t {
        const uint32_t i0 = mine[j];
        if (!d_srcs[i0] || j + 1 >= mine.size() || !d_srcs[mine[j + 1]]) return 1;
        const uint8_t *s0 = static_cast<const uint8_t *>(d_srcs[i0]), *s1 = static_cast<const uint8_t *>(d_srcs[mine[j + 1]]);
        uint8_t *t0 = target_of(i0), *t1 = target_of(mine[j + 1]);
        if (s1 <= s0 || (t0 == nullptr) != (t1 == nullptr) || (t0 && t1 <= t0)) return 1;
        const size_t ss = (size_t)(s1 - s0), ds = t0 ? (size_t)(t1 - t0) : out_size;
        if (ds < out_size || ss % 16u || ds % 16u) return 1;  // (PVRTC wants 16-byte strides; harmless for the others)
        size_t len = 2;
        while (len < kRun && j + len < mine.size()) {
          const uint32_t in = mine[j + len];
          const uint8_t *sn = static_cast<const uint8_t *>(d_srcs[in]);
          uint8_t *tn = target_of(in);
          if (!sn || sn != s0 + len * ss || (tn == nullptr) != (t0 == nullptr) || (t0 && tn != t0 + len * ds)) break;
          ++len;
        }
        *src_stride = ss;
        *dst_stride = ds;
        return len;
      };
      // scratch for the images that have no buffer on this device (gather only): two halves, so that the peer copies of
      // one run overlap the encode of the next (two streams, alternating)
      size_t scratch_images = 0;
      for (size_t j = 0; j < mine.size(); ++j)
        if (gather && target_of(mine[j]) == nullptr) scratch_images = std::min(kRun, std::max<size_t>(scratch_images + 1, 1));
      const size_t scratch_bytes = scratch_images ? scratch_images * out_size : 1;
      if (st->ensure(scratch_bytes, scratch_bytes) != ICAMD_OK) {
        pool_give(std::move(st));
        return fail_all(ICAMD_ERR_ALLOC, "staging allocation failed");
      }
      size_t run_no = 0;
      for (size_t j = 0; j < mine.size(); ++run_no) {
        size_t src_stride = 0, dst_stride = 0;
        size_t len = run_length(j, &src_stride, &dst_stride);
        const uint32_t i0 = mine[j];
        hipStream_t s = (run_no & 1u) ? st->stream2 : st->stream;
        uint8_t *scratch = static_cast<uint8_t *>((run_no & 1u) ? st->d_in : st->d_out);
        uint8_t *own = target_of(i0);
        if (!own && gather) len = std::min(len, scratch_images);
        auto set_all = [&](int code) { for (size_t k = 0; k < len; ++k) local[mine[j + k]] = code; };
        if (!d_srcs[i0]) { local[i0] = ICAMD_FALSE; j += 1; continue; }
        if (!own && !gather) {  // nowhere to put this image's blocks
          local[i0] = ICAMD_ERR_ARG;
          if (errors[(size_t)d].empty()) errors[(size_t)d].set("icamd_encode_batch_sharded_device: image without an output buffer");
          j += 1;
          continue;
        }
        uint8_t *target = own ? own : scratch;
        if (codec == ICAMD_PVRTC2) icamd::pvrtc2_select_workspace((int)(run_no & 1u));  // one scratch buffer per stream
        const int rc_run = icamd_encode_device(codec, etc_strategy, src_components, swap_rb, height, width, height, width,
                                               row_stride_bytes, (uint32_t)len, len > 1 ? src_stride : 0,
                                               len > 1 ? (own ? dst_stride : out_size) : 0, d_srcs[i0], target, s);
        set_all(rc_run);
        if (rc_run == ICAMD_OK && gather) {
          for (size_t k = 0; k < len; ++k) {
            const uint32_t i = mine[j + k];
            uint8_t *from = own ? own + k * dst_stride : scratch + k * out_size;
            if (from == slot_of(i)) continue;  // encoded straight into its slot
            // the encoded image travels device -> device (xGMI between GPUs); no host staging
            const hipError_t e = dev == gather_device ? hipMemcpyAsync(slot_of(i), from, out_size, hipMemcpyDeviceToDevice, s)
                                                      : hipMemcpyPeerAsync(slot_of(i), gather_device, from, dev, out_size, s);
            if (e != hipSuccess) local[i] = fail(ICAMD_ERR_HIP, "gather copy", e);
          }
        }
        for (size_t k = 0; k < len; ++k)
          if (local[mine[j + k]] < 0 && errors[(size_t)d].empty()) errors[(size_t)d].set(g_last_error);
        j += len;
      }
      icamd::pvrtc2_select_workspace(0);
      const hipError_t e1 = hipStreamSynchronize(st->stream), e2 = hipStreamSynchronize(st->stream2);
      if (e1 != hipSuccess || e2 != hipSuccess) fail_all(ICAMD_ERR_HIP, "stream synchronize failed");
      pool_give(std::move(st));
      } catch (...) {
        // what was enqueued may still be running on the worker's streams: this device's images all count as failed
        const int code = abi_exception();
        fail_all(code, g_last_error);
        (void)hipDeviceSynchronize();
      }
    });
    } catch (...) {  // std::system_error from the thread's creation (or bad_alloc): the workers started so far finish
      (void)abi_exception();
      not_started_from = d;
      break;
    }
  }
  }  // all started workers joined
  for (int d = not_started_from; d < n_devices; ++d) {
    for (uint64_t i = (uint64_t)d; i < n_images; i += (uint64_t)n_devices) local[i] = ICAMD_ERR_ALLOC;
    errors[(size_t)d].set("icamd_encode_batch_sharded_device: could not start a worker thread");
  }
  (void)hipSetDevice(prev_device);
  int first = ICAMD_OK;
  for (uint32_t i = 0; i < n_images; ++i) {
    if (statuses) statuses[i] = local[i];
    if (first == ICAMD_OK && local[i] != ICAMD_OK) first = local[i];
  }
  if (first < 0)
    for (const WorkerError &e : errors)
      if (!e.empty()) { set_last_error(e.text); break; }
  return first;
} ICAMD_ABI_CATCH

// ---- container framing (extension; csrc/containers.h) ----
size_t icamd_container_size(int container, int codec, uint32_t height, uint32_t width, uint32_t levels) {
  using namespace icamd;
  if (!container_supports(container, codec) || height == 0 || width == 0 || levels == 0 || levels > 32) return 0;
  if (container == ICAMD_CONTAINER_PKM && (levels != 1 || height > 65532u || width > 65532u)) return 0;
  if (codec == ICAMD_PVRTC2 && (height != width || !is_pow2(width))) return 0;
  size_t total = container_header_size(container);
  for (uint32_t l = 0; l < levels; ++l) {
    if (l > 0 && (height >> l) == 0 && (width >> l) == 0) return 0;  // past the 1 x 1 level
    const size_t b = container_level_bytes(codec, height, width, l);
    if (b == 0 || b > 0xffffffffull) return 0;
    total += container_level_prefix(container) + b;
  }
  return total;
}

int icamd_container_write(int container, int codec, uint32_t height, uint32_t width, uint32_t levels,
                          const uint8_t *const *level_data, const size_t *level_sizes, uint8_t *out, size_t out_size) try {
  using namespace icamd;
  if (container < ICAMD_CONTAINER_DDS || container > ICAMD_CONTAINER_PVR) return fail(ICAMD_ERR_ARG, "unknown container");
  if (codec < ICAMD_DXT1 || codec > ICAMD_PVRTC2) return fail(ICAMD_ERR_ARG, "unknown codec");
  if (!level_data || !level_sizes || !out) return ICAMD_FALSE;
  const size_t need = icamd_container_size(container, codec, height, width, levels);
  if (need == 0 || out_size != need) return ICAMD_FALSE;
  for (uint32_t l = 0; l < levels; ++l)
    if (!level_data[l] || level_sizes[l] != container_level_bytes(codec, height, width, l)) return ICAMD_FALSE;
  container_write_header(container, codec, height, width, levels, level_sizes[0], out);
  uint8_t *p = out + container_header_size(container);
  for (uint32_t l = 0; l < levels; ++l) {
    if (container_level_prefix(container)) {
      put_le32(p, (uint32_t)level_sizes[l]);  // KTX imageSize; block streams are multiples of 8 bytes: no mip padding
      p += 4;
    }
    std::memcpy(p, level_data[l], level_sizes[l]);
    p += level_sizes[l];
  }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

#pragma GCC visibility pop
}  // extern "C"
