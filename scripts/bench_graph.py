#!/usr/bin/env python3
"""Launch-bound regime: N separate encode calls on small textures, eager vs captured once into a HIP graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
n, size = 64, 256
src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device="cuda")
for name, codec in (("DXT1", T.DXT1), ("ETC1 heuristic", T.ETC1), ("PVRTC", T.PVRTC2)):
    strategy = 3 if codec == T.ETC1 else 2
    out = torch.zeros((n, pkg.encoded_size(codec, size, size)), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def loop():
            for i in range(n):
                pkg.encode_device(codec, src[i], size, size, 4, etc_strategy=strategy, out=out[i:i + 1], stream=s)
        loop(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): loop()
        s.synchronize()
        eager = (time.perf_counter() - t0) / 20
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            loop()
        g.replay(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        s.synchronize()
        graph = (time.perf_counter() - t0) / 20
    batch_out = torch.zeros_like(out)
    pkg.encode_device(codec, src, size, size, 4, etc_strategy=strategy, n_images=n, out=batch_out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): pkg.encode_device(codec, src, size, size, 4, etc_strategy=strategy, n_images=n, out=batch_out)
    torch.cuda.synchronize()
    batched = (time.perf_counter() - t0) / 20
    assert torch.equal(out, batch_out)
    print("%-15s %d x %d^2: eager %.1f us/texture, graph replay %.1f us/texture, one batched call %.2f us/texture" % (
        name, n, size, eager / n * 1e6, graph / n * 1e6, batched / n * 1e6))
