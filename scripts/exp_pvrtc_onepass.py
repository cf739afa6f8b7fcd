"""r05 experiment: the PVRTC one-pass kernel against the morph + encode pair -- parity (pair, oracle) and time."""
import os, sys, time, hashlib
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T

dev = torch.device("cuda:0")
def enc(src, size, n):
    return pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n)

bad = 0
g = torch.Generator(device="cuda"); g.manual_seed(5)
for size in (512, 1024, 2048, 4096):
    for n in (1, 3):
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
        src[:, : size // 4, : size // 2] = src[:, :1, :1]
        if n > 1:
            src[1, :, :, 3] = 255
            src[2] = (src[2] >> 3) + 100
        pkg.pvrtc_tune(1, -1)
        ref = enc(src, size, n).clone()
        torch.cuda.synchronize()
        for sb in (2, 3, 4, 5, 6):
            pkg.pvrtc_tune(2, sb)
            out = torch.zeros_like(ref)
            pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
            torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            if not same:
                bad += 1
                d = (out != ref).view(n, -1, 8).any(dim=2)
                idx = d.nonzero()[:5].tolist()
                print("MISMATCH size", size, "n", n, "sb", sb, "blocks differing", int(d.sum()), idx)
        if size <= 1024:
            want = T.oracle_encode(T.PVRTC2, src[0].cpu().numpy(), size, size, 4)
            if ref[0].cpu().numpy().tobytes() != want:
                bad += 1; print("pair != oracle", size)
    print("size", size, "done, bad so far", bad, flush=True)
# unaligned (8 mod 16) destination: direct stores
size, n = 1024, 2
src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
per = size * size // 4
pkg.pvrtc_tune(1, -1)
ref = enc(src, size, n).clone()
buf = torch.zeros(n * per + 64, dtype=torch.uint8, device=dev)
pkg.pvrtc_tune(2, 3)
o = buf[8:8 + n * per].view(n, per)
st = pkg.lib().icamd_encode_device(T.PVRTC2, 2, 4, 0, size, size, size, size, size * 4, n, size * size * 4, per,
                                   __import__("ctypes").c_void_p(src.data_ptr()), __import__("ctypes").c_void_p(o.data_ptr()), None)
torch.cuda.synchronize()
print("unaligned dst rc", st, "equal", bool(torch.equal(o, ref)))
if not torch.equal(o, ref): bad += 1
print("PARITY", "OK" if bad == 0 else "FAILED %d" % bad, flush=True)

# timing: 16 x 4096^2, events over launches
size, n = 4096, 16
src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty((n, size * size // 4), dtype=torch.uint8, device=dev)
def timeit(label, reps=150):
    for _ in range(60): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-14s %.4f ms  %.0f Gpix/s  frac %.3f" % (label, ms, n * size * size / ms / 1e6, n * size * size * 4.25 / (ms * 1e-3) / 8e12), flush=True)
for rnd in range(3):
    pkg.pvrtc_tune(1, -1); timeit("pair")
    for sb in (3, 4, 5, 6):
        pkg.pvrtc_tune(2, sb); timeit("onepass K=%d" % (1 << sb))
for (size, n) in ((2048, 64), (1024, 256), (512, 1024), (4096, 1), (4096, 4), (2048, 8)):
    src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((n, size * size // 4), dtype=torch.uint8, device=dev)
    print("-- %d x %d^2" % (n, size))
    pkg.pvrtc_tune(1, -1); timeit("pair")
    for sb in (2, 3, 4, 5):
        pkg.pvrtc_tune(2, sb); timeit("onepass K=%d" % (1 << sb))
    pkg.pvrtc_tune(0, -1); timeit("auto")
