#!/usr/bin/env python3
"""Where do the microseconds of ONE config-sized image per call go?

 * size sweep (one image per call, sizes 1024^2 .. 8192^2, rotating through >= 256 MiB of distinct sources): a linear
   fit  t = t0 + pixels / rate  separates the per-call fixed cost (dispatch -> first wave, ramp, tail) from the streaming rate;
 * the same calls issued round-robin on TWO streams (what a caller with independent textures can do): the ramp of one
   call overlaps the tail of the previous one;
 * run under `rocprofv3 --kernel-trace` the kernel's own begin/end timestamps (no event packets) give the pure kernel
   duration and the gap between consecutive kernels (scripts/summarize in the calling shell script)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_amd_loader

pkg = ic_amd_loader.load_package()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
codec = {"dxt1": pkg.DXT1, "dxt5": pkg.DXT5}[os.environ.get("CODEC", "dxt1")]
comps = 4
bpp = 4.5 if codec == pkg.DXT1 else 5.0
calls = int(os.environ.get("CALLS", "512"))


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


g = torch.Generator(device=dev)
g.manual_seed(3)
pool = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device=dev, generator=g)  # 1 GiB of distinct source bytes
fit = []
for size in (1024, 2048, 4096, 8192):
    nbytes = size * size * comps
    n_img = max(1, pool.numel() // nbytes)
    per = pkg.encoded_size(codec, size, size)
    out = torch.empty((n_img, per), dtype=torch.uint8, device=dev)
    imgs = [pool[i * nbytes:(i + 1) * nbytes] for i in range(n_img)]
    s = torch.cuda.current_stream()

    def call(i, stream):
        k = i % n_img
        pkg.encode_device(codec, imgs[k], size, size, comps, n_images=1, out=out[k:k + 1], stream=stream)
    for i in range(300):
        call(i, s)
    torch.cuda.synchronize()
    # (a) event pair per call (what bench.py's single_image leg reports)
    st = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    en = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    for i in range(calls):
        st[i].record(s); call(i, s); en[i].record(s)
    torch.cuda.synchronize()
    pair = median([a.elapsed_time(b) for a, b in zip(st, en)])
    # (b) one event between consecutive calls (period = kernel + gap)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
    for i in range(calls):
        ev[i].record(s); call(i, s)
    ev[calls].record(s)
    torch.cuda.synchronize()
    period = median([ev[i].elapsed_time(ev[i + 1]) for i in range(calls)])
    # (c) no events inside: whole run / calls
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for i in range(calls):
        call(i, s)
    e1.record(s)
    torch.cuda.synchronize()
    bare = e0.elapsed_time(e1) / calls
    # (d) two streams round-robin
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    e0, e1, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    torch.cuda.synchronize()
    e0.record(sa); sb.wait_event(e0)
    for i in range(calls):
        call(i, sa if i % 2 == 0 else sb)
    eb.record(sb); sa.wait_event(eb); e1.record(sa)
    torch.cuda.synchronize()
    two = e0.elapsed_time(e1) / calls
    algo = size * size * bpp
    print("size %5d: event-pair %.2f us (frac %.3f) | period %.2f us (%.3f) | bare %.2f us (%.3f) | two streams %.2f us (%.3f)" % (
        size, pair * 1e3, algo / (pair * 1e-3) / 8e12, period * 1e3, algo / (period * 1e-3) / 8e12,
        bare * 1e3, algo / (bare * 1e-3) / 8e12, two * 1e3, algo / (two * 1e-3) / 8e12))
    fit.append((size * size, bare))
    del out
# least squares t = t0 + px / rate over the bare per-call times
n = len(fit)
sx = sum(p for p, _ in fit); sy = sum(t for _, t in fit)
sxx = sum(p * p for p, _ in fit); sxy = sum(p * t for p, t in fit)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
t0 = (sy - slope * sx) / n
print("fit: t = %.2f us + pixels / %.0f Gpix/s  (streaming rate = %.0f GB/s algorithmic)" % (t0 * 1e3, 1 / slope / 1e6, bpp / slope / 1e6))
