"""r05: one-pass kernel, time against the number of workgroup rounds (narrow textures)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(5)
def timeit(src, out, size, n, reps=100):
    for _ in range(30): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for size in (512, 1024, 2048, 4096):
    bw = size // 8; W = bw // 64; slots = 256 * (8 // W); bh = size // 4
    nmax = (1 << 28) // (size * size)
    src = torch.randint(0, 256, (2 * nmax, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((2 * nmax, size * size // 4), dtype=torch.uint8, device=dev)
    for sb in (3, 4, 5, 6, 7):
        K = 1 << sb
        if K > bh: continue
        line = []
        for rounds in (0.5, 1, 1.5, 2, 3, 4):
            n = int(rounds * slots * K / bh)
            if n < 1 or n > 2 * nmax: continue
            pkg.pvrtc_tune(2, sb)
            ms = timeit(src, out, size, n)
            line.append("%.1fr n=%d %.3fms %.0fGpix/s" % (n * (bh // K) / slots, n, ms, n * size * size / ms / 1e6))
        print("size %d K=%d: " % (size, K) + " | ".join(line), flush=True)
