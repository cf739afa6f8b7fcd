// r05 micro-benchmark (gfx950): what the HBM delivers to the plainest streaming kernels over 1 GiB -- read only (non-temporal and
// plain dwordx4 loads, 16 B per lane and instruction, 1 / 4 loads in flight per lane), write only (non-temporal / plain), copy
// (float4, the figure MI355X_MICROARCH.md quotes), and a read : write mix of 8 : 1 like DXT1 from RGBA8 -- the ceiling the
// roofline fractions of the memory-bound encoders should be read against.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_hbm.hip -o scripts/scratch/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ inline u4 ld(const u4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline void st(u4 *p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
// each workgroup streams UNROLL x 256 x 16 bytes
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) k_read(const u4 *src, u4 *sink, uint32_t n16) {
  const uint32_t base = blockIdx.x * (256u * UNROLL) + threadIdx.x;
  u4 acc = { 0, 0, 0, 0 };
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { const uint32_t k = base + i * 256u; if (k < n16) acc ^= ld<NT>(src + k); }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc;  // (never: keeps the loads)
}
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) k_write(u4 *dst, uint32_t n16, uint32_t seed) {
  const uint32_t base = blockIdx.x * (256u * UNROLL) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { const uint32_t k = base + i * 256u; if (k < n16) st<NT>(dst + k, (u4){ k, seed, k ^ seed, 7u }); }
}
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) k_copy(const u4 *src, u4 *dst, uint32_t n16) {
  const uint32_t base = blockIdx.x * (256u * UNROLL) + threadIdx.x;
  u4 v[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { const uint32_t k = base + i * 256u; v[i] = k < n16 ? ld<NT>(src + k) : (u4){ 0, 0, 0, 0 }; }
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { const uint32_t k = base + i * 256u; if (k < n16) st<NT>(dst + k, v[i]); }
}
// 8 : 1 like DXT1 <- RGBA8: a lane reads 64 bytes (4 x 16, rows of a 4 x 4-pixel block 16 KiB apart) and writes 8
// stride16: distance between pixel rows in 16-byte units (= row16 for a dense image; larger = padded rows)
__global__ void __launch_bounds__(256) k_mix(const u4 *src, u2 *dst, uint32_t n_blocks, uint32_t row16, uint32_t stride16) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n_blocks) return;
  const uint32_t brow = k / row16, bcol = k - brow * row16;  // row16 = blocks per row
  const u4 *p = src + (size_t)brow * 4u * stride16 + bcol;
  u4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + stride16), c = __builtin_nontemporal_load(p + 2 * stride16),
     d = __builtin_nontemporal_load(p + 3 * stride16);
  a ^= b; c ^= d; a ^= c;
  __builtin_nontemporal_store((u2){ a.x ^ a.y, a.z ^ a.w }, dst + k);
}
// the same bytes with ONE load per lane: lane t of a quad reads pixel row t of its block, lane 0 of the quad writes the 8 bytes
// (quad-xor'ed so that the loads are kept).  MODE 0: the quad = 4 adjacent lanes; MODE 1: row t of a wave's 16 blocks in lanes 16 t .. 16 t + 15
template <int MODE> __global__ void __launch_bounds__(256) k_mix_quad(const u4 *src, u2 *dst, uint32_t n_blocks, uint32_t row16) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  uint32_t k, t;
  if (MODE == 0) { k = g >> 2; t = g & 3u; }
  else { const uint32_t wave = g >> 6, lane = g & 63u; k = wave * 16u + (lane & 15u); t = lane >> 4; }
  if (k >= n_blocks) return;
  const uint32_t brow = k / row16, bcol = k - brow * row16;
  u4 a = __builtin_nontemporal_load(src + ((size_t)brow * 4u + t) * row16 + bcol);
  uint32_t x = a.x ^ a.y, y = a.z ^ a.w;
  if (MODE == 0) {
    x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xb1, 0xf, 0xf, true); y ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)y, 0xb1, 0xf, 0xf, true);
    x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4e, 0xf, 0xf, true); y ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)y, 0x4e, 0xf, 0xf, true);
  } else {
    x ^= (uint32_t)__shfl_xor((int)x, 16); y ^= (uint32_t)__shfl_xor((int)y, 16);
    x ^= (uint32_t)__shfl_xor((int)x, 32); y ^= (uint32_t)__shfl_xor((int)y, 32);
  }
  if (t == 0u) __builtin_nontemporal_store((u2){ x, y }, dst + k);
}
template <typename F> double timeit(F f, int reps = 30) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main() {
  const size_t bytes = 1ull << 30; const uint32_t n16 = (uint32_t)(bytes / 16);
  u4 *a, *b, *sink; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4096);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
#define RUN(label, bytes_moved, ...) { double ms = timeit([&] { __VA_ARGS__; }); printf("%-34s %.4f ms  %7.1f GB/s\n", label, ms, (bytes_moved) / ms / 1e6); }
  RUN("read  nt, 1 load / lane", (double)bytes, hipLaunchKernelGGL((k_read<true, 1>), dim3(n16 / 256), dim3(256), 0, 0, a, sink, n16))
  RUN("read  nt, 4 loads / lane", (double)bytes, hipLaunchKernelGGL((k_read<true, 4>), dim3(n16 / 1024), dim3(256), 0, 0, a, sink, n16))
  RUN("read  plain, 4 loads / lane", (double)bytes, hipLaunchKernelGGL((k_read<false, 4>), dim3(n16 / 1024), dim3(256), 0, 0, a, sink, n16))
  RUN("write nt, 1 store / lane", (double)bytes, hipLaunchKernelGGL((k_write<true, 1>), dim3(n16 / 256), dim3(256), 0, 0, b, n16, 3u))
  RUN("write nt, 4 stores / lane", (double)bytes, hipLaunchKernelGGL((k_write<true, 4>), dim3(n16 / 1024), dim3(256), 0, 0, b, n16, 3u))
  RUN("write plain, 4 stores / lane", (double)bytes, hipLaunchKernelGGL((k_write<false, 4>), dim3(n16 / 1024), dim3(256), 0, 0, b, n16, 3u))
  RUN("copy  nt, 1 / lane (read + write)", 2.0 * bytes, hipLaunchKernelGGL((k_copy<true, 1>), dim3(n16 / 256), dim3(256), 0, 0, a, b, n16))
  RUN("copy  nt, 4 / lane (read + write)", 2.0 * bytes, hipLaunchKernelGGL((k_copy<true, 4>), dim3(n16 / 1024), dim3(256), 0, 0, a, b, n16))
  RUN("copy  plain, 4 / lane", 2.0 * bytes, hipLaunchKernelGGL((k_copy<false, 4>), dim3(n16 / 1024), dim3(256), 0, 0, a, b, n16))
  { const uint32_t row16 = 1024, nb = n16 / 4;  // 4096-pixel rows of RGBA8 = 1024 x 16 B
    RUN("mix 8 : 1 (DXT1 <- RGBA8, 4096 px rows)", 1.125 * bytes, hipLaunchKernelGGL(k_mix, dim3(nb / 256), dim3(256), 0, 0, a, (u2 *)b, nb, row16, row16))
    RUN("mix 8 : 1, one load per lane (quads)", 1.125 * bytes, hipLaunchKernelGGL(k_mix_quad<0>, dim3(nb / 64), dim3(256), 0, 0, a, (u2 *)b, nb, row16))
    RUN("mix 8 : 1, one load per lane (16 x 4)", 1.125 * bytes, hipLaunchKernelGGL(k_mix_quad<1>, dim3(nb / 64), dim3(256), 0, 0, a, (u2 *)b, nb, row16))
    const uint32_t pad = 1024 + 16, nbp = (uint32_t)((bytes / 16 / pad / 4) * 1024);  // rows padded by 256 bytes
    RUN("mix 8 : 1, rows padded by 256 B", 72.0 * nbp, hipLaunchKernelGGL(k_mix, dim3(nbp / 256), dim3(256), 0, 0, a, (u2 *)b, nbp, row16, pad))
    const uint32_t pad2 = 1024 + 2, nbq = (uint32_t)((bytes / 16 / pad2 / 4) * 1024);  // rows padded by 32 bytes
    RUN("mix 8 : 1, rows padded by 32 B", 72.0 * nbq, hipLaunchKernelGGL(k_mix, dim3(nbq / 256), dim3(256), 0, 0, a, (u2 *)b, nbq, row16, pad2))
    RUN("mix 8 : 1, 8192 px rows", 1.125 * bytes, hipLaunchKernelGGL(k_mix, dim3(nb / 256), dim3(256), 0, 0, a, (u2 *)b, nb, 2048u, 2048u))
    RUN("mix 8 : 1, 1024 px rows", 1.125 * bytes, hipLaunchKernelGGL(k_mix, dim3(nb / 256), dim3(256), 0, 0, a, (u2 *)b, nb, 256u, 256u)) }
  return 0;
}
