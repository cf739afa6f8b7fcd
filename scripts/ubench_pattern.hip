// Read-bandwidth probe for the PVRTC morph kernel's access pattern on gfx950: every lane owns one 8x4-pixel block
// (4 rows x 32 B).  A: the lane loads its own 32 B per row as two dwordx4 (stride-32 B across lanes per instruction).
// B: the wave loads the same bytes fully coalesced (instruction i, lane l -> 16 B chunk i*64+l of the wave's 2 KiB
// row segment); data would then need an LDS transpose.  Both XOR everything and store 8 B per lane.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_pattern.hip -o build_ab/ubench_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k(const uint32_t *img, uint2 *out, uint32_t n, uint32_t log2_bw, uint32_t log2_bpi) {
  const uint32_t kk = blockIdx.x * 256 + threadIdx.x;
  const uint32_t image = kk >> log2_bpi, b = kk & ((1u << log2_bpi) - 1u);
  const uint32_t by = b >> log2_bw, bx = b & ((1u << log2_bw) - 1u);
  const uint32_t *base = img + (size_t)image * n * n;
  u4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    const uint32_t *p = base + (size_t)(by * 4u) * n + bx * 8u;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u4 *q = reinterpret_cast<const u4 *>(p + (size_t)y * n + 4 * h);
        acc ^= NT ? __builtin_nontemporal_load(q) : *q;
      }
  } else {
    const uint32_t lane = threadIdx.x & 63u, bx0 = bx - lane;  // wave covers blocks bx0..bx0+63 of block row by
    const uint32_t *p = base + (size_t)(by * 4u) * n + bx0 * 8u;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u4 *q = reinterpret_cast<const u4 *>(p + (size_t)y * n + (h * 64 + lane) * 4);
        acc ^= NT ? __builtin_nontemporal_load(q) : *q;
      }
  }
  out[kk] = make_uint2(acc.x ^ acc.y, acc.z ^ acc.w);
}

int main() {
  const uint32_t n = 4096, images = 16, log2_bw = 9, log2_bpi = 19;
  const size_t bytes = (size_t)images * n * n * 4;
  uint32_t *img; uint2 *out;
  hipMalloc(&img, bytes); hipMemset(img, 1, bytes);
  const uint32_t total = images << log2_bpi;
  hipMalloc(&out, (size_t)total * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = { "own-32B", "own-32B nt", "coalesced", "coalesced nt" };
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k<0, false><<<total / 256, 256>>>(img, out, n, log2_bw, log2_bpi);
      if (v == 1) k<0, true><<<total / 256, 256>>>(img, out, n, log2_bw, log2_bpi);
      if (v == 2) k<1, false><<<total / 256, 256>>>(img, out, n, log2_bw, log2_bpi);
      if (v == 3) k<1, true><<<total / 256, 256>>>(img, out, n, log2_bw, log2_bpi);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%-14s %.1f us  %.2f TB/s (read+write)\n", names[v], best * 1e3, (bytes + total * 8.0) / (best * 1e-3) / 1e12);
  }
  return 0;
}
