#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer drop-in entry points (icamd_compress: H2D + kernel + D2H)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cases = [("DXT1 kRGB", T.DXTC, T.RGB, 2), ("DXT5 kRGBA", T.DXTC, T.RGBA, 2), ("ETC1 kRGB smaller-error", T.ETC, T.RGB, 2),
         ("ETC1 kRGB heuristic", T.ETC, T.RGB, 3), ("PVRTC kRGBA", T.PVRTC, T.RGBA, 2), ("ETC1 kRGB smaller-error", T.ETC, T.RGB, 2),
         ("DXT1 kRGB", T.DXTC, T.RGB, 2)]
for name, comp, fmt, strategy in cases:
    img = T.s_noise(n, n, T.comps_of(fmt), index=1).reshape(-1)
    pkg.compress_host(comp, fmt, img, n, n, etc_strategy=strategy)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = pkg.compress_host(comp, fmt, img, n, n, etc_strategy=strategy); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print("%-26s %4dx%d  best %.2f ms/call (all: %s)  %.0f Mpix/s  (%.1f GB/s of source)" % (
        name, n, n, dt * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts), n * n / dt / 1e6, img.size / dt / 1e9))

# icamd_compress_batch: several worker threads (one stream + staging buffers each) on the SAME device overlap one
# image's H2D copy with another's kernel and D2H copy.
m = 2048
batch = [T.s_noise(m, m, 3, index=i).reshape(-1) for i in range(32)]
for devices in ([0], [0, 0], [0, 0, 0, 0], [0] * 8):
    pkg.compress_batch_host(T.DXTC, T.RGB, batch, m, m, devices)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); outs = pkg.compress_batch_host(T.DXTC, T.RGB, batch, m, m, devices); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print("compress_batch DXT1 kRGB 32 x %dx%d, %d worker(s) on device 0: %.2f ms  %.0f Mpix/s  (%.1f GB/s of source)" % (
        m, m, len(devices), dt * 1e3, 32 * m * m / dt / 1e6, 32 * m * m * 3 / dt / 1e9))
