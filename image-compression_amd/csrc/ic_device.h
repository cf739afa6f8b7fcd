// ic_device.h -- shared device-side helpers for the gfx950 block-encode kernels.
//
// Conventions used by every kernel in this directory:
//  * one 4x4 block per lane (DXT/ETC); consecutive lanes own consecutive block
//    columns, so a wave reads 64 x 16 B (RGBA8) or 64 x 12 B (RGB888) contiguous
//    bytes per pixel row and writes 64 x 8/16 B contiguous output;
//  * a pixel lives in one VGPR as a dword whose bytes are the source bytes in
//    MEMORY order: byte0 = first channel (R, or B for kBGR*), byte1 = G,
//    byte2 = third channel, byte3 = alpha (undefined for 3-byte sources --
//    every consumer either masks it or multiplies it by a zero weight);
//  * all arithmetic is 32-bit integer, like the reference (no floats).
//
// ICAMD_HOST_EMULATION (set ONLY by tests/host_emul, compiled with g++) swaps the
// gfx950 instruction wrappers below for plain-C equivalents so the per-block
// math in *_block.h can be unit-tested against the oracle without a GPU.  It is
// never defined when libic_amd.so is built; the product has no CPU path.
#ifndef ICAMD_IC_DEVICE_H_
#define ICAMD_IC_DEVICE_H_

#include <stdint.h>
#include <stddef.h>

#if defined(ICAMD_HOST_EMULATION)
#define ICAMD_DEV static inline
#define ICAMD_UNROLL
#define ICAMD_NOUNROLL
#else
#include <hip/hip_runtime.h>
#define ICAMD_DEV __device__ __forceinline__
#define ICAMD_UNROLL _Pragma("unroll")
#define ICAMD_NOUNROLL _Pragma("nounroll")
#endif

namespace icamd {

// Host-precomputed unsigned division n / d for n < 2^31 (block ids), d >= 1:
// q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
  uint32_t mul, shift, d;
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}

// Geometry of one launch over a batch of equally shaped images.
struct GridParams {
  const uint8_t *src;
  uint8_t *dst;
  uint64_t src_image_stride;  // bytes between images
  uint64_t dst_image_stride;
  uint32_t height, width;          // source image (pixels)
  uint32_t block_rows, block_cols; // emitted block grid (>= image for CompressAndPad)
  uint32_t row_stride;             // bytes between source rows
  uint32_t n_images;
  uint32_t swap_rb;                // source is B,G,R(,A)
  uint32_t etc_strategy;
  uint32_t log2_tile_cols;         // a workgroup covers 2^log2_tile_cols x (256 >> log2_tile_cols) blocks
  uint32_t tile_row0;              // first tile row of this launch (grids of more than 65 535 tile rows are chunked)
  uint32_t force_gather;           // rows too long for 32-bit lane offsets: every block takes the 64-bit gather path
};

// ---- thin wrappers over the gfx950 instructions the kernels rely on ----
#if defined(ICAMD_HOST_EMULATION)

ICAMD_DEV uint32_t umulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
ICAMD_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
  return c;
}
ICAMD_DEV uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) { return (a > b ? a - b : b - a) + c; }
ICAMD_DEV uint32_t sad_u16x2(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return (al > bl ? al - bl : bl - al) + (ah > bh ? ah - bh : bh - ah) + c;
}
ICAMD_DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; ++i) {
    int x = (a >> (8 * i)) & 0xff, y = (b >> (8 * i)) & 0xff;
    c += (uint32_t)(x > y ? x - y : y - x);
  }
  return c;
}
ICAMD_DEV uint32_t sad_hi_u8(uint32_t a, uint32_t b, uint32_t c) { return (sad_u8(a, b, 0u) << 16) + c; }
ICAMD_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31));
}
ICAMD_DEV uint32_t avg_u8(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) r |= ((((a >> (8 * i)) & 0xff) + ((b >> (8 * i)) & 0xff)) >> 1) << (8 * i);
  return r;
}
ICAMD_DEV uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    uint32_t s = (sel >> (8 * i)) & 0xff;
    uint32_t b = s <= 7 ? (uint32_t)((v >> (8 * s)) & 0xff) : (s == 0x0c ? 0u : 0xffu);
    r |= b << (8 * i);
  }
  return r;
}
ICAMD_DEV uint32_t bfe(uint32_t v, uint32_t off, uint32_t w) { return (v >> off) & ((1u << w) - 1u); }
ICAMD_DEV uint32_t bit_mask(uint32_t v, uint32_t bit) { return (v >> bit) & 1u ? 0xffffffffu : 0u; }
ICAMD_DEV int32_t imad24(int32_t a, int32_t b, int32_t c) { return a * b + c; }
ICAMD_DEV uint32_t umad24(uint32_t a, uint32_t b, uint32_t c) { return a * b + c; }
ICAMD_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
ICAMD_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
ICAMD_DEV int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
ICAMD_DEV int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

#else  // gfx950

ICAMD_DEV uint32_t umulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
// v_dot4_u32_u8: a.b0*b.b0 + a.b1*b.b1 + a.b2*b.b2 + a.b3*b.b3 + c
ICAMD_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// |a - b| + c for 16-bit operands
// PRECONDITION a, b < 65536.  Emitted as v_sad_u16 (|a.lo16 - b.lo16| + |a.hi16 - b.hi16| + c; the high halves
// are both zero), which has a compiler builtin; v_sad_u32 only exists as inline asm, and every asm statement costs
// hazard s_nops and blocks scheduling.
ICAMD_DEV uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u16(a, b, c); }
// v_sad_u16 proper: |a.lo16 - b.lo16| + |a.hi16 - b.hi16| + c
ICAMD_DEV uint32_t sad_u16x2(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u16(a, b, c); }
// v_sad_u8: sum over the 4 bytes of |a.b - b.b|, plus c
ICAMD_DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u8(a, b, c); }
ICAMD_DEV uint32_t sad_hi_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_hi_u8(a, b, c); }  // (sad << 16) + c
// v_lerp_u8 with a zero rounding operand: per byte (a + b) >> 1 -- four floor-averages in one instruction
ICAMD_DEV uint32_t avg_u8(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0u); }
// v_alignbit_b32: low 32 bits of ({hi,lo} >> sh)
ICAMD_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// v_perm_b32: byte i of the result = byte sel.b[i] of the 8-byte value {hi,lo}
// (selector 0..3 -> lo bytes, 4..7 -> hi bytes, 0x0c -> 0x00).
ICAMD_DEV uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
ICAMD_DEV uint32_t bfe(uint32_t v, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(v, off, w); }
// v_bfe_i32 of a one-bit field: all ones iff bit `bit` of v is set
ICAMD_DEV uint32_t bit_mask(uint32_t v, uint32_t bit) { return (uint32_t)__builtin_amdgcn_sbfe((int32_t)v, bit, 1u); }
// v_mad_i32_i24: a * b + c.  PRECONDITION |a|, |b| < 2^23 (the compiler cannot prove it and would emit
// v_mul_lo_u32 + v_add_u32 for the plain expression).
ICAMD_DEV int32_t imad24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }
// v_mad_u32_u24: the same for unsigned operands below 2^24 (v_mul_lo_u32, what the plain product compiles to when the
// range is not provable, issues at a quarter of the rate)
ICAMD_DEV uint32_t umad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
ICAMD_DEV uint32_t umin(uint32_t a, uint32_t b) { return min(a, b); }
ICAMD_DEV uint32_t umax(uint32_t a, uint32_t b) { return max(a, b); }
ICAMD_DEV int32_t imin(int32_t a, int32_t b) { return min(a, b); }
ICAMD_DEV int32_t imax(int32_t a, int32_t b) { return max(a, b); }

#endif

// True iff the predicate holds in every active lane of the wave (the emulation has one "lane").
#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV bool wave_all(bool p) { return p; }
ICAMD_DEV uint32_t wave_count(bool p) { return p ? 64u : 0u; }
#else
ICAMD_DEV bool wave_all(bool p) { return __all(p ? 1 : 0) != 0; }
// number of active lanes of the wave in which the predicate holds
ICAMD_DEV uint32_t wave_count(bool p) { return (uint32_t)__popcll(__ballot(p ? 1 : 0)); }
#endif

// Value the optimiser must treat as freshly produced (blocks common-subexpression elimination across uses).
#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV uint32_t opaque(uint32_t v) { return v; }
#else
ICAMD_DEV uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}
#endif

// The compiler folds these into v_min3/v_max3.
ICAMD_DEV uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return umin(umin(a, b), c); }
ICAMD_DEV uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return umax(umax(a, b), c); }
ICAMD_DEV int32_t imax3(int32_t a, int32_t b, int32_t c) { return imax(imax(a, b), c); }

ICAMD_DEV uint32_t fastdiv(uint32_t n, const FastDiv &f) { return (umulhi32(n, f.mul) + n) >> f.shift; }

// Unaligned-tolerant vector loads (gfx950 global memory runs in unaligned-access
// mode; the backend emits one global_load_dwordx3/x4 for these).
struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U3 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(1))) U2 { uint32_t x, y; };

// Streaming access: every source byte is read once and every output byte written once, so the encoders mark them
// non-temporal (global_load/store ... nt).  Measured on the headline kernel: 5.85 -> 6.17 TB/s (r01 A/B).
#if defined(ICAMD_HOST_EMULATION)
ICAMD_DEV U4 load_stream(const U4 *p) { return *p; }
ICAMD_DEV U3 load_stream(const U3 *p) { return *p; }
ICAMD_DEV U2 load_stream(const U2 *p) { return *p; }
#else
typedef uint32_t icamd_u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t icamd_u32x3_u __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t icamd_u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
ICAMD_DEV U4 load_stream(const U4 *p) {
  const icamd_u32x4_u v = __builtin_nontemporal_load(reinterpret_cast<const icamd_u32x4_u *>(p));
  U4 r = { v.x, v.y, v.z, v.w };
  return r;
}
ICAMD_DEV U3 load_stream(const U3 *p) {
  const icamd_u32x3_u v = __builtin_nontemporal_load(reinterpret_cast<const icamd_u32x3_u *>(p));
  U3 r = { v.x, v.y, v.z };
  return r;
}
ICAMD_DEV U2 load_stream(const U2 *p) {
  const icamd_u32x2_u v = __builtin_nontemporal_load(reinterpret_cast<const icamd_u32x2_u *>(p));
  U2 r = { v.x, v.y };
  return r;
}
ICAMD_DEV void store_stream12(void *p, uint32_t a, uint32_t b, uint32_t c) {
  const icamd_u32x3_u v = { a, b, c };
  __builtin_nontemporal_store(v, reinterpret_cast<icamd_u32x3_u *>(p));
}
// 8- and 16-byte block stores (no alignment assumed: the caller owns the output pointer)
ICAMD_DEV void store_stream8(void *p, uint32_t a, uint32_t b) {
  const icamd_u32x2_u v = { a, b };
  __builtin_nontemporal_store(v, reinterpret_cast<icamd_u32x2_u *>(p));
}
ICAMD_DEV void store_stream16(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const icamd_u32x4_u v = { a, b, c, d };
  __builtin_nontemporal_store(v, reinterpret_cast<icamd_u32x4_u *>(p));
}
#endif

#if !defined(ICAMD_HOST_EMULATION)
// Launch geometry of the encoders: grid = (column tiles, row tiles, images), one workgroup per tile of
// 2^log2_tile_cols x (256 >> log2_tile_cols) blocks (256 x 1 for images at least 1024 pixels wide).  Image, tile row
// and tile column are workgroup-uniform (blockIdx), so all 64-bit address arithmetic runs on the scalar unit and a
// lane only adds a 32-bit offset (global_load ... v_off, s[base:base+1]); no division anywhere.  A wave still covers
// 64 consecutive blocks of one block row (or whole rows of a narrower image).
struct TileCoord {
  uint32_t img, brow, bcol;  // this lane's block
  uint32_t brow0, bcol0;     // first block of the workgroup's tile (uniform)
  uint32_t ly, lx;           // the lane's position inside the tile
  bool full;                 // uniform: the whole tile lies inside the block grid
  bool interior;             // uniform: every block of the tile lies inside the source image (4 wide loads each)
  bool valid;                // this lane's block exists
};
// WIDE = the tile is 256 x 1 blocks (block grids more than 128 columns wide): the lane's row IS the tile's row, so every
// row-dependent quantity is workgroup-uniform as well.
// WIDE_ROWS > 1: the workgroup covers WIDE_ROWS block rows and every lane encodes WIDE_ROWS vertically adjacent blocks
// (the returned coordinate is the first one; the caller steps brow0 / brow).
// tile_col: the tile column this workgroup takes (blockIdx.x, unless the kernel deals the columns out differently)
template <bool WIDE, uint32_t WIDE_ROWS = 1>
__device__ __forceinline__ TileCoord locate_tile(const GridParams &P, uint32_t tile_col) {
  TileCoord t;
  const uint32_t cols = WIDE ? 256u : 1u << P.log2_tile_cols, rows = WIDE ? WIDE_ROWS : 256u >> P.log2_tile_cols;
  t.lx = WIDE ? threadIdx.x : threadIdx.x & (cols - 1u);
  t.ly = WIDE ? 0u : threadIdx.x >> P.log2_tile_cols;
  t.bcol0 = tile_col * cols;
  t.brow0 = (blockIdx.y + P.tile_row0) * rows;
  t.bcol = t.bcol0 + t.lx;
  t.brow = t.brow0 + t.ly;
  t.img = blockIdx.z;
  t.full = t.bcol0 + cols <= P.block_cols && t.brow0 + rows <= P.block_rows;
  t.interior = (t.bcol0 + cols) * 4u <= P.width && (t.brow0 + rows) * 4u <= P.height;
  t.valid = t.full || (t.bcol < P.block_cols && t.brow < P.block_rows);
  return t;
}
template <bool WIDE, uint32_t WIDE_ROWS = 1>
__device__ __forceinline__ TileCoord locate_tile(const GridParams &P) {
  return locate_tile<WIDE, WIDE_ROWS>(P, blockIdx.x);
}
// The block's first source byte = uniform 64-bit base + 32-bit lane offset (<= 4 * 256 rows of stride).
struct TileSrc {
  const uint8_t *base;  // workgroup-uniform
  uint32_t off;         // per lane
};
template <int COMPS>
__device__ __forceinline__ TileSrc tile_src(const GridParams &P, const TileCoord &t) {
  TileSrc r;
  r.base = P.src + (uint64_t)t.img * P.src_image_stride + (uint64_t)(t.brow0 * 4u) * P.row_stride +
           (uint64_t)t.bcol0 * (4u * COMPS);
  r.off = t.ly * 4u * P.row_stride + t.lx * (4u * COMPS);
  return r;
}
// Loads the lane's block: four wide loads when the block (for an interior tile: without looking) lies inside the
// image, the clamp-to-edge gather otherwise.
template <int COMPS>
__device__ __forceinline__ void load_tile_block(const GridParams &P, const TileCoord &t, uint32_t px[16]);
// Address of the block's output bytes (BYTES per block, raster order inside the image).
template <int BYTES>
__device__ __forceinline__ uint8_t *tile_dst(const GridParams &P, const TileCoord &t) {
  uint8_t *base = P.dst + (uint64_t)t.img * P.dst_image_stride +
                  ((uint64_t)t.brow0 * P.block_cols + t.bcol0) * (uint64_t)BYTES;
  return base + (uint32_t)((t.ly * P.block_cols + t.lx) * (uint32_t)BYTES);
}
#endif

// Gather one 4x4 block (reference: internal/pixel4x4.h:44-67, pixel4x4.cc:23-59).
// Interior blocks take 4 wide loads; edge blocks replicate the last row/column
// (clamp-to-edge), byte by byte.
// Interior block: four wide streaming loads.  `base` + `off` = address of the block's first pixel; the kernels pass
// a workgroup-uniform base and a 32-bit lane offset so the loads use the scalar-base addressing mode.
template <int COMPS>
ICAMD_DEV void load_block_interior(const uint8_t *__restrict__ base, uint32_t off, uint32_t stride, uint32_t px[16]) {
  ICAMD_UNROLL
  for (int y = 0; y < 4; ++y) {
    // the row advance stays in the 32-bit lane offset (launch_tiled: tiles are 256 x 1 when 1024 * stride >= 2^32, and
    // rows too long even for three strides take the 64-bit gather, GridParams::force_gather)
    const uint8_t *row = base + (uint32_t)(off + (uint32_t)y * stride);
    if (COMPS == 4) {
      U4 v = load_stream(reinterpret_cast<const U4 *>(row));
      px[4 * y + 0] = v.x; px[4 * y + 1] = v.y; px[4 * y + 2] = v.z; px[4 * y + 3] = v.w;
    } else {
      U3 v = load_stream(reinterpret_cast<const U3 *>(row));
      px[4 * y + 0] = v.x;
      px[4 * y + 1] = alignbit(v.y, v.x, 24);
      px[4 * y + 2] = alignbit(v.z, v.y, 16);
      px[4 * y + 3] = v.z >> 8;
    }
  }
}

// wide_ok = false: rows too long for the interior path's 32-bit row offsets -- gather every pixel with 64-bit addresses.
template <int COMPS>
ICAMD_DEV void load_block(const uint8_t *__restrict__ img, uint32_t h, uint32_t w, uint32_t stride,
                          uint32_t row, uint32_t col, uint32_t px[16], bool wide_ok = true) {
  if (wide_ok && row + 4 <= h && col + 4 <= w) {
    load_block_interior<COMPS>(img + (size_t)row * stride + (size_t)col * COMPS, 0u, stride, px);
  } else {
    ICAMD_UNROLL
    for (int y = 0; y < 4; ++y) {
      uint32_t sy = umin(row + y, h - 1);
      ICAMD_UNROLL
      for (int x = 0; x < 4; ++x) {
        uint32_t sx = umin(col + x, w - 1);
        const uint8_t *q = img + (size_t)sy * stride + (size_t)sx * COMPS;
        uint32_t v = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16;
        if (COMPS == 4) v |= (uint32_t)q[3] << 24;
        px[4 * y + x] = v;
      }
    }
  }
}

#if !defined(ICAMD_HOST_EMULATION)
template <int COMPS>
__device__ __forceinline__ void load_tile_block(const GridParams &P, const TileCoord &t, uint32_t px[16]) {
  if (!P.force_gather && (t.interior || (t.brow * 4u + 4u <= P.height && t.bcol * 4u + 4u <= P.width))) {
    const TileSrc ts = tile_src<COMPS>(P, t);
    load_block_interior<COMPS>(ts.base, ts.off, P.row_stride, px);
  } else {
    load_block<COMPS>(P.src + (size_t)t.img * P.src_image_stride, P.height, P.width, P.row_stride, t.brow * 4u,
                      t.bcol * 4u, px, !P.force_gather);
  }
}
#endif

// Exact small-constant divisions on the operand ranges the encoders produce;
// the ranges are checked exhaustively at compile time below.
ICAMD_DEV uint32_t div3(uint32_t x) { return (x * 683u) >> 11; }   // x <= 765  = 3*255
ICAMD_DEV uint32_t div5(uint32_t x) { return (x * 3277u) >> 14; }  // x <= 1275 = 5*255
ICAMD_DEV uint32_t div7(uint32_t x) { return (x * 9363u) >> 16; }  // x <= 1785 = 7*255

namespace detail {
constexpr bool check_div(uint32_t d, uint32_t mul, uint32_t sh, uint32_t max) {
  for (uint32_t x = 0; x <= max; ++x)
    if (((x * mul) >> sh) != x / d) return false;
  return true;
}
static_assert(check_div(3, 683, 11, 765), "div3 magic");
static_assert(check_div(5, 3277, 14, 1275), "div5 magic");
static_assert(check_div(7, 9363, 16, 1785), "div7 magic");
}  // namespace detail

}  // namespace icamd
#endif  // ICAMD_IC_DEVICE_H_
