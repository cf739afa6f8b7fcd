// diag_kernels.hip -- measurement aid, not part of the encode path: the effective shader clock under load.
//
// gfx950 has two free-running 64-bit counters a wave can read: s_memtime ticks once per SHADER cycle (it follows the
// DVFS clock; MI355X_MICROARCH.md "s_memtime tick = shader cycle"), s_memrealtime at a constant rate
// (hipDeviceAttributeWallClockRate, 100 MHz).  One sleeping wave that reads both at the start and at the end of an
// interval therefore measures the mean shader clock of that interval, while the encode kernels being benchmarked
// occupy the rest of the chip from another stream (bench.py's `clock` object).
#include "ic_launch.h"

namespace icamd {

extern "C" __global__ void __launch_bounds__(64) icamd_clock_probe_kernel(uint64_t *out, uint64_t ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  uint64_t r1;
  do {
    __builtin_amdgcn_s_sleep(64);  // stay out of the issue slots of the kernels being measured
    r1 = __builtin_amdgcn_s_memrealtime();
  } while (r1 - r0 < ticks);
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}

hipError_t launch_clock_probe(uint64_t *d_out, uint64_t ticks, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(icamd_clock_probe_kernel, dim3(1), dim3(64), 0, stream, d_out, ticks);
  return hipGetLastError();
}

}  // namespace icamd
