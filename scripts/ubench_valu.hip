// Instruction-rate micro-benchmark for the integer VALU ops the encoders lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o gpurun_out/ubench_valu ; run on the GPU box.
// Each kernel runs ITER x 8 independent chains of ONE instruction per lane; rate = lane-ops / s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 4096

#define DEF_KERNEL(NAME, ASM)                                                            \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t s) {           \
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;        \
    uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;     \
    uint32_t b = s + threadIdx.x, c = s * 3 + 1;                                         \
    for (int i = 0; i < ITER; ++i) {                                                     \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)       \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c));                                                    \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
  }

// operand numbering: %0..%7 accumulators, %8 = b, %9 = c
#define OP3(x, INS) INS " %" #x ", %" #x ", %8, %9\n"
#define A_ADD(x) "v_add_u32 %" #x ", %" #x ", %8\n"
#define A_DOT4(x) "v_dot4_u32_u8 %" #x ", %" #x ", %8, %9\n"
#define A_SADU32(x) "v_sad_u32 %" #x ", %" #x ", %8, %9\n"
#define A_SADU8(x) "v_sad_u8 %" #x ", %" #x ", %8, %9\n"
#define A_MIN3(x) "v_min3_u32 %" #x ", %" #x ", %8, %9\n"
#define A_MAX3I(x) "v_max3_i32 %" #x ", %" #x ", %8, %9\n"
#define A_MIN(x) "v_min_u32 %" #x ", %" #x ", %8\n"
#define A_ALIGNBIT(x) "v_alignbit_b32 %" #x ", %" #x ", %8, 2\n"
#define A_PERM(x) "v_perm_b32 %" #x ", %" #x ", %8, %9\n"
#define A_PKADD(x) "v_pk_add_u16 %" #x ", %" #x ", %8 clamp\n"
#define A_PKMAD(x) "v_pk_mad_u16 %" #x ", %" #x ", %8, %9\n"
#define A_MAD24(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %9\n"
#define A_MUL24(x) "v_mul_u32_u24 %" #x ", %" #x ", %8\n"
#define A_MULLO(x) "v_mul_lo_u32 %" #x ", %" #x ", %8\n"
#define A_MULHI(x) "v_mul_hi_u32 %" #x ", %" #x ", %8\n"
#define A_LSHLADD(x) "v_lshl_add_u32 %" #x ", %" #x ", 3, %8\n"
#define A_LSHLOR(x) "v_lshl_or_b32 %" #x ", %" #x ", 3, %8\n"
#define A_ANDOR(x) "v_and_or_b32 %" #x ", %" #x ", %8, %9\n"
#define A_BFE(x) "v_bfe_u32 %" #x ", %" #x ", 3, 8\n"
#define A_LSHR(x) "v_lshrrev_b32 %" #x ", 3, %" #x "\n"
#define A_ADD3(x) "v_add3_u32 %" #x ", %" #x ", %8, %9\n"
#define A_OR3(x) "v_or3_b32 %" #x ", %" #x ", %8, %9\n"
#define A_BCNT(x) "v_bcnt_u32_b32 %" #x ", %" #x ", %8\n"
#define A_CMPCND(x) "v_cmp_lt_u32 vcc, %" #x ", %8\n v_cndmask_b32 %" #x ", %" #x ", %9, vcc\n"
#define A_SDWA(x) "v_add_u32_sdwa %" #x ", %" #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define A_MAD64(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %9\n"

DEF_KERNEL(add_u32, A_ADD)
DEF_KERNEL(dot4_u32_u8, A_DOT4)
DEF_KERNEL(sad_u32, A_SADU32)
DEF_KERNEL(sad_u8, A_SADU8)
DEF_KERNEL(min3_u32, A_MIN3)
DEF_KERNEL(max3_i32, A_MAX3I)
DEF_KERNEL(min_u32, A_MIN)
DEF_KERNEL(alignbit, A_ALIGNBIT)
DEF_KERNEL(perm_b32, A_PERM)
DEF_KERNEL(pk_add_u16_clamp, A_PKADD)
DEF_KERNEL(pk_mad_u16, A_PKMAD)
DEF_KERNEL(mad_u32_u24, A_MAD24)
DEF_KERNEL(mul_u32_u24, A_MUL24)
DEF_KERNEL(mul_lo_u32, A_MULLO)
DEF_KERNEL(mul_hi_u32, A_MULHI)
DEF_KERNEL(lshl_add_u32, A_LSHLADD)
DEF_KERNEL(lshl_or_b32, A_LSHLOR)
DEF_KERNEL(and_or_b32, A_ANDOR)
DEF_KERNEL(bfe_u32, A_BFE)
DEF_KERNEL(lshrrev_b32, A_LSHR)
DEF_KERNEL(add3_u32, A_ADD3)
DEF_KERNEL(or3_b32, A_OR3)
DEF_KERNEL(bcnt_u32, A_BCNT)
DEF_KERNEL(cmp_plus_cndmask, A_CMPCND)
DEF_KERNEL(add_u32_sdwa, A_SDWA)


#define A_MOV(x) "v_mov_b32 %" #x ", %8\n"
#define A_AND(x) "v_and_b32 %" #x ", %" #x ", %8\n"
#define A_OR(x) "v_or_b32 %" #x ", %" #x ", %8\n"
#define A_XOR(x) "v_xor_b32 %" #x ", %" #x ", %8\n"
#define A_SUB(x) "v_sub_u32 %" #x ", %" #x ", %8\n"
#define A_LSHL(x) "v_lshlrev_b32 %" #x ", 3, %" #x "\n"
#define A_LSHLV(x) "v_lshlrev_b32 %" #x ", %8, %" #x "\n"
#define A_ASHR(x) "v_ashrrev_i32 %" #x ", 3, %" #x "\n"
#define A_MAXU(x) "v_max_u32 %" #x ", %" #x ", %8\n"
#define A_MAXI(x) "v_max_i32 %" #x ", %" #x ", %8\n"
#define A_CNDMASK(x) "v_cndmask_b32 %" #x ", %" #x ", %8, vcc\n"
#define A_CMPONLY(x) "v_cmp_lt_u32 vcc, %" #x ", %8\n"
#define A_MULI24(x) "v_mul_i32_i24 %" #x ", %" #x ", %8\n"
#define A_FMA(x) "v_fma_f32 %" #x ", %" #x ", %8, %9\n"
#define A_FADD(x) "v_add_f32 %" #x ", %" #x ", %8\n"
#define A_FMUL(x) "v_mul_f32 %" #x ", %" #x ", %8\n"
#define A_FMAC(x) "v_fmac_f32 %" #x ", %8, %9\n"
#define A_ADDLSHL(x) "v_add_lshl_u32 %" #x ", %" #x ", %8, 2\n"
#define A_BFI(x) "v_bfi_b32 %" #x ", %" #x ", %8, %9\n"
#define A_MED3(x) "v_med3_u32 %" #x ", %" #x ", %8, %9\n"
#define A_PKADDNC(x) "v_pk_add_u16 %" #x ", %" #x ", %8\n"
#define A_PKMIN(x) "v_pk_min_u16 %" #x ", %" #x ", %8\n"
#define A_PKMAXI(x) "v_pk_max_i16 %" #x ", %" #x ", %8\n"
#define A_PKLSHR(x) "v_pk_lshrrev_b16 %" #x ", 3, %" #x "\n"
#define A_PKMUL(x) "v_pk_mul_lo_u16 %" #x ", %" #x ", %8\n"
#define A_ADDU16(x) "v_add_u16 %" #x ", %" #x ", %8\n"
#define A_MADU16(x) "v_mad_u16 %" #x ", %" #x ", %8, %9\n"
#define A_DOT2(x) "v_dot2_u32_u16 %" #x ", %" #x ", %8, %9\n"
#define A_DOT8(x) "v_dot8_u32_u4 %" #x ", %" #x ", %8, %9\n"
#define A_DOT4I(x) "v_dot4_i32_i8 %" #x ", %" #x ", %8, %9\n"
#define A_MSAD(x) "v_msad_u8 %" #x ", %" #x ", %8, %9\n"
#define A_LERP(x) "v_lerp_u8 %" #x ", %" #x ", %8, %9\n"
#define A_SADHI(x) "v_sad_hi_u8 %" #x ", %" #x ", %8, %9\n"
#define A_SADU16(x) "v_sad_u16 %" #x ", %" #x ", %8, %9\n"
#define A_MADI24(x) "v_mad_i32_i24 %" #x ", %" #x ", %8, %9\n"
#define A_XAD(x) "v_xad_u32 %" #x ", %" #x ", %8, %9\n"
#define A_MIN3I(x) "v_min3_i32 %" #x ", %" #x ", %8, %9\n"
#define A_PKFMA(x) "v_pk_fma_f16 %" #x ", %" #x ", %8, %9\n"
#define A_CVTPKU8(x) "v_cvt_pk_u8_f32 %" #x ", %" #x ", %8, %9\n"
#define A_MBCNT(x) "v_mbcnt_lo_u32_b32 %" #x ", %" #x ", %8\n"
#define A_ADDCO(x) "v_add_co_u32 %" #x ", vcc, %" #x ", %8\n"
#define A_ADDC(x) "v_addc_co_u32 %" #x ", vcc, %" #x ", %8, vcc\n"
DEF_KERNEL(mov_b32, A_MOV) DEF_KERNEL(and_b32, A_AND) DEF_KERNEL(or_b32, A_OR) DEF_KERNEL(xor_b32, A_XOR)
DEF_KERNEL(sub_u32, A_SUB) DEF_KERNEL(lshlrev_imm, A_LSHL) DEF_KERNEL(lshlrev_vgpr, A_LSHLV) DEF_KERNEL(ashrrev_imm, A_ASHR)
DEF_KERNEL(max_u32, A_MAXU) DEF_KERNEL(max_i32, A_MAXI) DEF_KERNEL(cndmask_vcc, A_CNDMASK) DEF_KERNEL(cmp_only, A_CMPONLY)
DEF_KERNEL(mul_i32_i24, A_MULI24) DEF_KERNEL(fma_f32, A_FMA) DEF_KERNEL(add_f32, A_FADD) DEF_KERNEL(mul_f32, A_FMUL)
DEF_KERNEL(fmac_f32, A_FMAC) DEF_KERNEL(add_lshl_u32, A_ADDLSHL) DEF_KERNEL(bfi_b32, A_BFI) DEF_KERNEL(med3_u32, A_MED3)
DEF_KERNEL(pk_add_u16, A_PKADDNC) DEF_KERNEL(pk_min_u16, A_PKMIN) DEF_KERNEL(pk_max_i16, A_PKMAXI) DEF_KERNEL(pk_lshrrev_b16, A_PKLSHR)
DEF_KERNEL(pk_mul_lo_u16, A_PKMUL) DEF_KERNEL(add_u16, A_ADDU16) DEF_KERNEL(mad_u16, A_MADU16) DEF_KERNEL(dot2_u32_u16, A_DOT2)
DEF_KERNEL(dot8_u32_u4, A_DOT8) DEF_KERNEL(dot4_i32_i8, A_DOT4I) DEF_KERNEL(msad_u8, A_MSAD) DEF_KERNEL(lerp_u8, A_LERP)
DEF_KERNEL(sad_hi_u8, A_SADHI) DEF_KERNEL(sad_u16, A_SADU16) DEF_KERNEL(mad_i32_i24, A_MADI24) DEF_KERNEL(xad_u32, A_XAD)
DEF_KERNEL(min3_i32, A_MIN3I) DEF_KERNEL(pk_fma_f16, A_PKFMA) DEF_KERNEL(cvt_pk_u8_f32, A_CVTPKU8) DEF_KERNEL(mbcnt_lo, A_MBCNT)
DEF_KERNEL(add_co_u32, A_ADDCO) DEF_KERNEL(addc_co_u32, A_ADDC)

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Entry { const char *name; kern_t k; int ops_per_slot; };

int main() {
  const int blocks = 256 * 8 * 4, threads = 256;  // 8 waves/SIMD resident, 4 rounds
  uint32_t *d;
  hipMalloc(&d, (size_t)blocks * threads * 4);
  Entry es[] = {
#define E(n) { #n, k_##n, 1 }
    E(add_u32), E(dot4_u32_u8), E(sad_u32), E(sad_u8), E(min3_u32), E(max3_i32), E(min_u32), E(alignbit), E(perm_b32),
    E(pk_add_u16_clamp), E(pk_mad_u16), E(mad_u32_u24), E(mul_u32_u24), E(mul_lo_u32), E(mul_hi_u32), E(lshl_add_u32),
    E(lshl_or_b32), E(and_or_b32), E(bfe_u32), E(lshrrev_b32), E(add3_u32), E(or3_b32), E(bcnt_u32),
    { "cmp_plus_cndmask", k_cmp_plus_cndmask, 2 }, E(add_u32_sdwa),
    E(mov_b32), E(and_b32), E(or_b32), E(xor_b32), E(sub_u32), E(lshlrev_imm), E(lshlrev_vgpr), E(ashrrev_imm), E(max_u32),
    E(max_i32), E(cndmask_vcc), E(cmp_only), E(mul_i32_i24), E(fma_f32), E(add_f32), E(mul_f32), E(fmac_f32), E(add_lshl_u32),
    E(bfi_b32), E(med3_u32), E(pk_add_u16), E(pk_min_u16), E(pk_max_i16), E(pk_lshrrev_b16), E(pk_mul_lo_u16), E(add_u16),
    E(mad_u16), E(dot2_u32_u16), E(dot8_u32_u4), E(dot4_i32_i8), E(msad_u8), E(lerp_u8), E(sad_hi_u8), E(sad_u16),
    E(mad_i32_i24), E(xad_u32), E(min3_i32), E(pk_fma_f16), E(cvt_pk_u8_f32), E(mbcnt_lo), E(add_co_u32), E(addc_co_u32),
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-20s %12s %12s\n", "instruction", "Tlane-op/s", "rel. to add");
  double base = 0;
  for (auto &e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    double ops = (double)blocks * threads * ITER * 8.0 * e.ops_per_slot;
    double rate = ops / (best * 1e-3) / 1e12;
    if (base == 0) base = rate;
    printf("%-20s %12.2f %12.2f   (%.3f ms)\n", e.name, rate, rate / base, best);
  }
  return 0;
}
