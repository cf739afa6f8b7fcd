// r05 micro-benchmark (gfx950): issue rate of v_lshl_add_u64 against v_add_u32, alone and mixed with a half-rate op, at 2 and 8
// waves per SIMD (the PVRTC one-pass kernel runs 2: is a 64-bit add one issue slot, and what does it cost next to v_perm?).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_u64.hip -o gpurun_out/ubench_u64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define BODY(ASM)                                                                                                         \
  uint64_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;                                           \
  uint64_t b = s + threadIdx.x;                                                                                           \
  uint32_t c0 = threadIdx.x, c1 = c0 * 3, c2 = c0 * 5, c3 = c0 * 7, d = (uint32_t)s;                                      \
  for (int i = 0; i < ITER; ++i)                                                                                          \
    asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b), "v"(d)); \
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3;
#define K(NAME, ASM)                                                                            \
  __global__ void __launch_bounds__(256) k_##NAME(uint64_t *out, uint64_t s) {                 \
    extern __shared__ uint32_t pad[];                                                           \
    BODY(ASM)                                                                                   \
  }
// 8 instructions per iteration each
K(add32, "v_add_u32 %4, %4, %9\nv_add_u32 %5, %5, %9\nv_add_u32 %6, %6, %9\nv_add_u32 %7, %7, %9\nv_add_u32 %4, %4, %9\nv_add_u32 %5, %5, %9\nv_add_u32 %6, %6, %9\nv_add_u32 %7, %7, %9\n")
K(add64, "v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\nv_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\n")
K(perm, "v_perm_b32 %4, %4, %9, %5\nv_perm_b32 %5, %5, %9, %6\nv_perm_b32 %6, %6, %9, %7\nv_perm_b32 %7, %7, %9, %4\nv_perm_b32 %4, %4, %9, %5\nv_perm_b32 %5, %5, %9, %6\nv_perm_b32 %6, %6, %9, %7\nv_perm_b32 %7, %7, %9, %4\n")
K(perm_add32, "v_perm_b32 %4, %4, %9, %5\nv_add_u32 %5, %5, %9\nv_perm_b32 %6, %6, %9, %7\nv_add_u32 %7, %7, %9\nv_perm_b32 %4, %4, %9, %5\nv_add_u32 %5, %5, %9\nv_perm_b32 %6, %6, %9, %7\nv_add_u32 %7, %7, %9\n")
K(perm_add64, "v_perm_b32 %4, %4, %9, %5\nv_lshl_add_u64 %0, %0, 0, %8\nv_perm_b32 %6, %6, %9, %7\nv_lshl_add_u64 %1, %1, 0, %8\nv_perm_b32 %4, %4, %9, %5\nv_lshl_add_u64 %2, %2, 0, %8\nv_perm_b32 %6, %6, %9, %7\nv_lshl_add_u64 %3, %3, 0, %8\n")
K(perm_salu, "v_perm_b32 %4, %4, %9, %5\ns_and_b64 s[20:21], s[20:21], s[22:23]\nv_perm_b32 %6, %6, %9, %7\ns_and_b64 s[20:21], s[20:21], s[22:23]\nv_perm_b32 %4, %4, %9, %5\ns_and_b64 s[20:21], s[20:21], s[22:23]\nv_perm_b32 %6, %6, %9, %7\ns_and_b64 s[20:21], s[20:21], s[22:23]\n")
K(cmp_cnd, "v_cmp_lt_u32 vcc, %4, %9\nv_cndmask_b32 %5, %5, %6, vcc\nv_cmp_lt_u32 vcc, %6, %9\nv_cndmask_b32 %7, %7, %4, vcc\nv_cmp_lt_u32 vcc, %4, %9\nv_cndmask_b32 %5, %5, %6, vcc\nv_cmp_lt_u32 vcc, %6, %9\nv_cndmask_b32 %7, %7, %4, vcc\n")
template <typename F> void run(const char *name, F f, int lds_bytes, const char *occ) {
  uint64_t *out; hipMalloc(&out, 1024 * 16 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 16;
  hipLaunchKernelGGL(f, dim3(grid), dim3(256), lds_bytes, 0, out, 1ull);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(f, dim3(grid), dim3(256), lds_bytes, 0, out, 1ull);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double insts = (double)grid * 4 /*waves*/ * ITER * 8;
  printf("%-12s %-14s %.3f ms  %.2f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, occ, ms, insts / ms / 1e6,
         1024 * 2.4e9 / (insts / (ms * 1e-3)));
  hipFree(out);
}
int main() {
  const int lds2 = 70 * 1024, lds8 = 16 * 1024;  // 2 workgroups of 4 waves per CU = 2 waves per SIMD; 8 per CU = 8 per SIMD
#define R(N) hipFuncSetAttribute((const void *)k_##N, hipFuncAttributeMaxDynamicSharedMemorySize, lds2); run(#N, k_##N, lds2, "2 waves/SIMD"); run(#N, k_##N, lds8, "8 waves/SIMD");
  R(add32) R(add64) R(perm) R(perm_add32) R(perm_add64) R(perm_salu) R(cmp_cnd)
  return 0;
}
