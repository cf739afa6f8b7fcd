// Probe: semantics and issue rate of the quad-SAD instructions on gfx950 (v_qsad_pk_u16_u8, v_mqsad_pk_u16_u8,
// v_mqsad_u32_u8).  Build: hipcc --offload-arch=gfx950 -O3 scripts/probe_qsad.hip -o gpurun_out/probe_qsad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void sem(uint64_t *out, uint64_t s0, uint32_t s1, uint64_t acc) {
  out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc);
  out[1] = __builtin_amdgcn_mqsad_pk_u16_u8(s0, s1, acc);
}

#define ITER 2048
template <int MODE>
__global__ void __launch_bounds__(256) rate(uint64_t *out, uint32_t s) {
  uint64_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (2 * i + 1) + s;
  const uint32_t ref = s * 7 + 1;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_amdgcn_qsad_pk_u16_u8(a[i], ref, a[i]);
      if (MODE == 1) a[i] = __builtin_amdgcn_mqsad_pk_u16_u8(a[i], ref, a[i]);
      if (MODE == 2) a[i] = (uint64_t)__builtin_amdgcn_sad_u8((uint32_t)a[i], ref, (uint32_t)a[i]);
    }
  }
  uint64_t r = 0;
  for (int i = 0; i < 8; ++i) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  uint64_t *d;
  hipMalloc(&d, 256 * 8 * 4 * 256 * 8);
  struct { uint64_t s0; uint32_t s1; uint64_t acc; } cases[] = {
    { 0x50463c32281e140aull, 0x00000019u, 0 },            // bytes 10,20,..,80 ; ref (25,0,0,0)
    { 0x50463c32281e140aull, 0x00000119u, 0 },            // ref (25,1,0,0)
    { 0x50463c32281e140aull, 0x19000000u, 0 },            // ref (0,0,0,25)
    { 0x50463c32281e140aull, 0x00000019u, 0x0004000300020001ull },
    { 0x5046003200001400ull, 0x05050505u, 0 },            // zeros in src0: is the mask on src0 or src1?
  };
  for (auto &c : cases) {
    sem<<<1, 1>>>(d, c.s0, c.s1, c.acc);
    uint64_t h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s0=%016llx s1=%08x acc=%016llx  qsad=%016llx  mqsad=%016llx\n", (unsigned long long)c.s0, c.s1,
           (unsigned long long)c.acc, (unsigned long long)h[0], (unsigned long long)h[1]);
  }
  const int blocks = 256 * 8 * 4;
  const char *names[3] = { "qsad_pk_u16_u8", "mqsad_pk_u16_u8", "sad_u8" };
  for (int m = 0; m < 3; ++m) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m == 0) rate<0><<<blocks, 256>>>(d, 3);
      if (m == 1) rate<1><<<blocks, 256>>>(d, 3);
      if (m == 2) rate<2><<<blocks, 256>>>(d, 3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %.2f T lane-instr/s\n", names[m], (double)blocks * 256 * ITER * 8 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
