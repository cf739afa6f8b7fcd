// r05 micro-benchmark (gfx950): issue rate of the packed 16-bit integer ops the PVRTC decoder's blend is made of (v_pk_mad_u16,
// v_pk_mul_lo_u16, v_pk_lshrrev_b16, v_pk_add_u16) against v_add_u32 / v_perm_b32, and of their 32-bit stand-ins
// (v_mad_u32_u24, v_mul_u32_u24, v_lshrrev_b32 + v_and), at 2 and 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_pk16.hip -o gpurun_out/ubench_pk16
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
K(perm, "v_perm_b32 %4, %4, %9, %5\nv_perm_b32 %5, %5, %9, %6\nv_perm_b32 %6, %6, %9, %7\nv_perm_b32 %7, %7, %9, %4\nv_perm_b32 %4, %4, %9, %5\nv_perm_b32 %5, %5, %9, %6\nv_perm_b32 %6, %6, %9, %7\nv_perm_b32 %7, %7, %9, %4\n")
K(pk_mad, "v_pk_mad_u16 %4, %4, %9, %5\nv_pk_mad_u16 %5, %5, %9, %6\nv_pk_mad_u16 %6, %6, %9, %7\nv_pk_mad_u16 %7, %7, %9, %4\nv_pk_mad_u16 %4, %4, %9, %5\nv_pk_mad_u16 %5, %5, %9, %6\nv_pk_mad_u16 %6, %6, %9, %7\nv_pk_mad_u16 %7, %7, %9, %4\n")
K(pk_mul, "v_pk_mul_lo_u16 %4, %4, %9\nv_pk_mul_lo_u16 %5, %5, %9\nv_pk_mul_lo_u16 %6, %6, %9\nv_pk_mul_lo_u16 %7, %7, %9\nv_pk_mul_lo_u16 %4, %4, %9\nv_pk_mul_lo_u16 %5, %5, %9\nv_pk_mul_lo_u16 %6, %6, %9\nv_pk_mul_lo_u16 %7, %7, %9\n")
K(pk_lshr, "v_pk_lshrrev_b16 %4, 8, %4\nv_pk_lshrrev_b16 %5, 8, %5\nv_pk_lshrrev_b16 %6, 8, %6\nv_pk_lshrrev_b16 %7, 8, %7\nv_pk_lshrrev_b16 %4, 8, %4\nv_pk_lshrrev_b16 %5, 8, %5\nv_pk_lshrrev_b16 %6, 8, %6\nv_pk_lshrrev_b16 %7, 8, %7\n")
K(pk_add, "v_pk_add_u16 %4, %4, %9\nv_pk_add_u16 %5, %5, %9\nv_pk_add_u16 %6, %6, %9\nv_pk_add_u16 %7, %7, %9\nv_pk_add_u16 %4, %4, %9\nv_pk_add_u16 %5, %5, %9\nv_pk_add_u16 %6, %6, %9\nv_pk_add_u16 %7, %7, %9\n")
K(mad24, "v_mad_u32_u24 %4, %4, %9, %5\nv_mad_u32_u24 %5, %5, %9, %6\nv_mad_u32_u24 %6, %6, %9, %7\nv_mad_u32_u24 %7, %7, %9, %4\nv_mad_u32_u24 %4, %4, %9, %5\nv_mad_u32_u24 %5, %5, %9, %6\nv_mad_u32_u24 %6, %6, %9, %7\nv_mad_u32_u24 %7, %7, %9, %4\n")
K(mul24, "v_mul_u32_u24 %4, %4, %9\nv_mul_u32_u24 %5, %5, %9\nv_mul_u32_u24 %6, %6, %9\nv_mul_u32_u24 %7, %7, %9\nv_mul_u32_u24 %4, %4, %9\nv_mul_u32_u24 %5, %5, %9\nv_mul_u32_u24 %6, %6, %9\nv_mul_u32_u24 %7, %7, %9\n")
K(lshr32, "v_lshrrev_b32 %4, 8, %4\nv_lshrrev_b32 %5, 8, %5\nv_lshrrev_b32 %6, 8, %6\nv_lshrrev_b32 %7, 8, %7\nv_lshrrev_b32 %4, 8, %4\nv_lshrrev_b32 %5, 8, %5\nv_lshrrev_b32 %6, 8, %6\nv_lshrrev_b32 %7, 8, %7\n")
K(and_or, "v_and_or_b32 %4, %4, %9, %5\nv_and_or_b32 %5, %5, %9, %6\nv_and_or_b32 %6, %6, %9, %7\nv_and_or_b32 %7, %7, %9, %4\nv_and_or_b32 %4, %4, %9, %5\nv_and_or_b32 %5, %5, %9, %6\nv_and_or_b32 %6, %6, %9, %7\nv_and_or_b32 %7, %7, %9, %4\n")
K(bfe, "v_bfe_u32 %4, %4, 8, 8\nv_bfe_u32 %5, %5, 8, 8\nv_bfe_u32 %6, %6, 8, 8\nv_bfe_u32 %7, %7, 8, 8\nv_bfe_u32 %4, %4, 8, 8\nv_bfe_u32 %5, %5, 8, 8\nv_bfe_u32 %6, %6, 8, 8\nv_bfe_u32 %7, %7, 8, 8\n")
K(lshl_add, "v_lshl_add_u32 %4, %4, 3, %5\nv_lshl_add_u32 %5, %5, 3, %6\nv_lshl_add_u32 %6, %6, 3, %7\nv_lshl_add_u32 %7, %7, 3, %4\nv_lshl_add_u32 %4, %4, 3, %5\nv_lshl_add_u32 %5, %5, 3, %6\nv_lshl_add_u32 %6, %6, 3, %7\nv_lshl_add_u32 %7, %7, 3, %4\n")
K(pk_mad_sel, "v_pk_mad_u16 %4, %4, %9, %5 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %5, %5, %9, %6 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %6, %6, %9, %7 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %7, %7, %9, %4 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %4, %4, %9, %5 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %5, %5, %9, %6 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %6, %6, %9, %7 op_sel_hi:[1,0,1]\nv_pk_mad_u16 %7, %7, %9, %4 op_sel_hi:[1,0,1]\n")
K(mul24_sdwa, "v_mul_u32_u24_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %4, %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %5, %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %6, %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_mul_u32_u24_sdwa %7, %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
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
  R(add32) R(perm) R(pk_mad) R(pk_mul) R(pk_lshr) R(pk_add) R(mad24) R(mul24) R(lshr32) R(and_or) R(bfe) R(lshl_add) R(pk_mad_sel) R(mul24_sdwa)
  return 0;
}
