"""Runs ON THE GPU BOX (r06): the one-pass kernel's HALO form (textures of 8192^2 and more, regions of one texture) against the
morph + encode pair -- ms per launch from HIP events, outputs compared (torch.equal) with the pair's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_amd_loader
pkg = ic_amd_loader.load_package()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from image_compression_amd import sharding
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)


def timed(fn, reps):
    for _ in range(max(3, reps // 4)): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for size, n in ((8192, 4), (8192, 1), (16384, 1), (4096, 16)):
    src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((n, size * size // 4), dtype=torch.uint8, device=dev)
    ref = None
    row = []
    for mode, sb in ((1, -1), (2, 3), (2, 4), (2, 5), (2, 6), (0, -1)):
        pkg.pvrtc_tune(mode, sb)
        fn = lambda: pkg.encode_device(pkg.PVRTC2, src, size, size, 4, n_images=n, out=out)
        t = timed(fn, 40 if size <= 8192 else 15)
        if ref is None:
            ref = out.clone()
        ok = torch.equal(out, ref)
        px = n * size * size
        row.append("%s %.4f ms %.0f Gpix/s frac %.3f %s" % ("pair" if mode == 1 else ("auto" if mode == 0 else "K=%d" % (1 << sb)), t,
                   px / t / 1e6, px * 4.25 / (t * 1e-3) / 8e12, "ok" if ok else "MISMATCH"))
    print("%d x %d^2: " % (n, size) + " | ".join(row), flush=True)
    del src, out, ref
    torch.cuda.empty_cache()
# regions: one 4096^2 / 8192^2 texture as 8 regions (the 8-GPU split of sharding.pvrtc_region), all eight launched back to back
for size in (4096, 8192):
    src = torch.randint(0, 256, (size, size, 4), dtype=torch.uint8, device=dev, generator=g)
    world = 8
    regs = [sharding.pvrtc_region(size, world, r) for r in range(world)]
    outs = [torch.empty(r["dst_bytes"], dtype=torch.uint8, device=dev) for r in regs]
    row, ref = [], None
    for mode, sb in ((1, -1), (2, 4), (2, 6), (0, -1)):
        pkg.pvrtc_tune(mode, sb)
        def fn():
            for r, o in zip(regs, outs):
                pkg.pvrtc_encode_region_device(src, size, r["first_block"], r["n_blocks"], out=o)
        t = timed(fn, 30)
        cat = torch.cat(outs)
        if ref is None:
            ref = cat.clone()
        row.append("%s %.4f ms per 8 regions (frac %.3f) %s" % ("pair" if mode == 1 else ("auto" if mode == 0 else "K=%d" % (1 << sb)), t,
                   size * size * 4.25 / (t * 1e-3) / 8e12, "ok" if torch.equal(cat, ref) else "MISMATCH"))
    print("%d^2 in 8 regions: " % size + " | ".join(row), flush=True)
pkg.pvrtc_tune(0, -1)
