// Cross-XCD visibility probe: kernel A (2 workgroups) writes buf[i] = gen*1000+i; kernel B (2 workgroups) reads the
// element written by the OTHER workgroup.  Repeated for several generations on the same buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void writer(unsigned *buf, unsigned gen) { unsigned i = blockIdx.x * 256 + threadIdx.x; buf[i] = gen * 1000u + i; }
__global__ void reader(const unsigned *buf, unsigned *out) { unsigned i = blockIdx.x * 256 + threadIdx.x; out[i] = buf[i ^ 256u]; }
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0 hipMalloc, 1 hipMallocAsync per generation
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned *out; hipMalloc(&out, 512 * 4);
  unsigned *buf = nullptr; if (mode == 0) hipMalloc(&buf, 512 * 4);
  unsigned h[512]; int bad_total = 0;
  for (unsigned gen = 1; gen <= 6; ++gen) {
    if (mode == 1) hipMallocAsync((void **)&buf, 512 * 4, s);
    hipLaunchKernelGGL(writer, dim3(2), dim3(256), 0, s, buf, gen);
    hipLaunchKernelGGL(reader, dim3(2), dim3(256), 0, s, buf, out);
    if (mode == 1) hipFreeAsync(buf, s);
    hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    int bad = 0; for (unsigned i = 0; i < 512; ++i) bad += h[i] != gen * 1000u + (i ^ 256u);
    printf("mode %d gen %u buf %p bad %d (e.g. got %u want %u)\n", mode, gen, (void *)buf, bad, h[0], gen * 1000u + 256u);
    bad_total += bad;
  }
  return bad_total != 0;
}
