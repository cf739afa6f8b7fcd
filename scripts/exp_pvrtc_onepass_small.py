import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(5)
def timeit(src, out, size, n, reps=300):
    for _ in range(50): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for size, ns in ((512, (1, 4, 16, 64, 128, 256)), (1024, (1, 2, 4, 16, 32, 64)), (2048, (1, 2, 4, 8, 16)), (4096, (1, 2, 3, 4, 5, 8))):
    for n in ns:
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
        out = torch.empty((n, size * size // 4), dtype=torch.uint8, device=dev)
        row = []
        pkg.pvrtc_tune(1, -1); row.append("pair %.1f" % timeit(src, out, size, n))
        pkg.pvrtc_tune(0, -1); row.append("auto %.1f" % timeit(src, out, size, n))
        for sb in (2, 3, 4, 5):
            pkg.pvrtc_tune(2, sb); row.append("K%d %.1f" % (1 << sb, timeit(src, out, size, n)))
        print("%4d x %4d^2 us: " % (n, size) + "  ".join(row), flush=True)
