import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(5)
bad = 0
for size in (256, 512, 1024, 2048, 4096):
    for n in (1, 3):
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
        src[:, : size // 4, : size // 2] = src[:, :1, :1]
        if n > 1:
            src[1, :, :, 3] = 255; src[2] = (src[2] >> 3) + 100
        pkg.pvrtc_tune(1, -1)
        ref = pkg.encode_device(T.PVRTC4, src, size, size, 4, n_images=n).clone()
        torch.cuda.synchronize()
        if size <= 1024:
            if ref[0].cpu().numpy().tobytes() != T.oracle_encode(T.PVRTC4, src[0].cpu().numpy(), size, size, 4):
                bad += 1; print("pair != oracle", size)
        for sb in (1, 2, 3, 4, 5, 6, 7):
            if (1 << sb) > size // 4: continue
            pkg.pvrtc_tune(2, sb)
            out = torch.zeros_like(ref)
            pkg.encode_device(T.PVRTC4, src, size, size, 4, n_images=n, out=out)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                d = (out != ref).view(n, -1, 8).any(dim=2)
                print("MISMATCH size", size, "n", n, "sb", sb, "blocks", int(d.sum()), d.nonzero()[:6].tolist())
    print("size", size, "bad so far", bad, flush=True)
print("PARITY", "OK" if not bad else "FAILED")
def timeit(src, out, size, n, reps=150):
    for _ in range(40): pkg.encode_device(T.PVRTC4, src, size, size, 4, n_images=n, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): pkg.encode_device(T.PVRTC4, src, size, size, 4, n_images=n, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for size, ns in ((4096, (16, 4, 1)), (2048, (64, 8, 1)), (1024, (256, 16, 1)), (512, (1024, 64, 4)), (256, (4096, 64))):
    for n in ns:
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
        out = torch.empty((n, size * size // 2), dtype=torch.uint8, device=dev)
        row = []
        pkg.pvrtc_tune(1, -1); row.append("pair %.1f" % timeit(src, out, size, n))
        pkg.pvrtc_tune(0, -1); row.append("auto %.1f" % timeit(src, out, size, n))
        for sb in (2, 3, 4, 5, 6, 7):
            if (1 << sb) > size // 4: continue
            pkg.pvrtc_tune(2, sb); row.append("K%d %.1f" % (1 << sb, timeit(src, out, size, n)))
        print("%4d x %4d^2 us: " % (n, size) + "  ".join(row), flush=True)
