// r05 micro-benchmark: what bounds icamd_compress_batch (32 x 12 MiB pageable host buffers -> device)?
// pageable hipMemcpyAsync by size and by thread count, hipHostRegister cost, registered-copy rate, memcpy-to-pinned rate.
// Build: hipcc -O2 scripts/ubench_h2d.hip -o scripts/scratch/ubench_h2d -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MiB = 1 << 20;
  const int N = 32;
  const size_t sz = 12 * MiB;
  std::vector<char *> host(N);
  for (auto &h : host) { h = (char *)malloc(sz); memset(h, 1, sz); }
  for (int threads : {1, 2, 4, 8}) {
    std::vector<char *> dev(threads);
    std::vector<hipStream_t> st(threads);
    for (int t = 0; t < threads; ++t) { hipMalloc((void **)&dev[t], sz); hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); }
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      std::vector<std::thread> w;
      for (int t = 0; t < threads; ++t)
        w.emplace_back([&, t]() { hipSetDevice(0); for (int i = t; i < N; i += threads) { hipMemcpyAsync(dev[t], host[i], sz, hipMemcpyHostToDevice, st[t]); hipStreamSynchronize(st[t]); } });
      for (auto &x : w) x.join();
      double dt = now() - t0;
      if (rep) printf("pageable H2D 32 x 12 MiB, %d thread(s): %.2f ms  %.1f GB/s\n", threads, dt * 1e3, N * sz / dt / 1e9);
    }
    for (int t = 0; t < threads; ++t) { hipFree(dev[t]); hipStreamDestroy(st[t]); }
  }
  // register / copy / unregister per buffer
  char *d; hipMalloc((void **)&d, 64 * MiB);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int rep = 0; rep < 2; ++rep) {
    double treg = 0, tcopy = 0, tunreg = 0;
    for (int i = 0; i < N; ++i) {
      double a = now(); hipHostRegister(host[i], sz, hipHostRegisterDefault);
      double b = now(); hipMemcpyAsync(d, host[i], sz, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
      double c = now(); hipHostUnregister(host[i]);
      double e = now(); treg += b - a; tcopy += c - b; tunreg += e - c;
    }
    if (rep) printf("per 12 MiB buffer: hipHostRegister %.3f ms, registered H2D %.3f ms (%.1f GB/s), hipHostUnregister %.3f ms\n", treg / N * 1e3, tcopy / N * 1e3, sz / (tcopy / N) / 1e9, tunreg / N * 1e3);
  }
  // memcpy into a pinned mirror with T threads, then one DMA
  char *pin; hipHostMalloc((void **)&pin, N * sz, hipHostMallocDefault);
  char *dbig; hipMalloc((void **)&dbig, N * sz);
  for (int threads : {1, 2, 4, 8, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      std::vector<std::thread> w;
      for (int t = 0; t < threads; ++t) w.emplace_back([&, t]() { for (int i = t; i < N; i += threads) memcpy(pin + i * sz, host[i], sz); });
      for (auto &x : w) x.join();
      double t1 = now();
      hipMemcpyAsync(dbig, pin, N * sz, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
      double t2 = now();
      if (rep) printf("memcpy 32 x 12 MiB into pinned, %2d thread(s): %.2f ms (%.1f GB/s); one 384 MiB DMA: %.2f ms (%.1f GB/s)\n", threads, (t1 - t0) * 1e3, N * sz / (t1 - t0) / 1e9, (t2 - t1) * 1e3, N * sz / (t2 - t1) / 1e9);
    }
  }
  for (size_t m : {4, 12, 24, 48, 96}) {
    char *h = (char *)malloc(m * MiB); memset(h, 2, m * MiB);
    hipMemcpyAsync(dbig, h, m * MiB, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < 8; ++i) { hipMemcpyAsync(dbig, h, m * MiB, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
    double dt = (now() - t0) / 8;
    printf("pageable H2D of %zu MiB: %.3f ms  %.1f GB/s\n", m, dt * 1e3, m * MiB / dt / 1e9);
    free(h);
  }
  return 0;
}
