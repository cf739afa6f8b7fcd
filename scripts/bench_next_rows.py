#!/usr/bin/env python3
"""Device-resident throughput of the SURVEY 8f "next" rows: block decoders, Downsample, Pad, DXT1->ETC1 transcode.
Prints one line per kernel: Mpixels/s (source or result pixels, whichever is larger) and algorithmic GB/s, and (r05) the
parity of image 0 of exactly the timed call against the oracle (tests/ic_testlib: test infrastructure, the checker only)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
L = pkg.lib()
BAD = []
# r06: `--json PATH` additionally writes every leg as an object (bench.py's `next_rows` field reads it); `--core` runs only the legs
# VERDICT r05 item 5 puts a bar on (decoders, Downsample, transcode) -- what bench.py's default line can afford
JSON_PATH = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
CORE = "--core" in sys.argv
LEGS = []
_print = print

def print(*a, **k):  # every result line is "<leg name> ... <GB/s> GB/s ... (<ms> ms ...) parity: ..."; recorded as it is printed
    _print(*a, **k)
    import re as _re
    line = " ".join(str(x) for x in a)
    m_g = _re.search(r"([0-9.]+) GB/s", line)
    m_t = _re.search(r"\(([0-9.]+) ms", line)
    if m_g and m_t and "parity:" in line:
        name = _re.split(r"\s+[0-9.]+ (?:Mpix/s|GB/s)", line)[0].strip()
        name = _re.sub(r"\s+", " ", name)
        LEGS.append({"leg": name, "algorithmic_GBps": float(m_g.group(1)), "frac": round(float(m_g.group(1)) / 8000.0, 4),
                     "ms_per_call": float(m_t.group(1)), "parity": "bit-exact" if "bit-exact" in line else "MISMATCH",
                     "one_launch": "one call per image" not in line})

def parity(what, got, want):
    """got: device tensor (image 0 of the timed call's output), want: the oracle's bytes."""
    torch.cuda.synchronize()
    ok = want is not None and got.cpu().numpy().tobytes() == want
    if not ok:
        BAD.append(what)
    return "parity: bit-exact vs oracle (image 0)" if ok else "parity: MISMATCH vs oracle"
dev = torch.device("cuda:0")
n, batch = 4096, 16
stream = torch.cuda.current_stream()
sh = ctypes.c_void_p(stream.cuda_stream)

def timeit(fn, reps=20):
    """Seconds per call in the steady state: like bench.py's preconditioning (the first launches after an idle period
    run 10-30 % slower than the steady state of the same kernel), 0.25 s of untimed calls first, then at least `reps` calls and
    at least 60 ms between two events on the launch stream."""
    if os.environ.get("ICAMD_NEXT_ROWS_QUICK") == "1":  # the profiled passes (gpu_profile_next_rows.sh): few launches, small CSVs
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n0 = 0
    while time.perf_counter() - t0 < 0.25:
        fn(); n0 += 1
        if n0 % 8 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / max(n0, 1)
    reps = max(reps, min(2000, int(0.06 / max(per, 1e-6))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

g = torch.Generator(device=dev); g.manual_seed(1)
for name, codec, comps, bb in [("dxt1", 0, 3, 8), ("dxt5", 1, 4, 16), ("etc1", 2, 3, 8)]:
    src = torch.randint(0, 256, (batch, n, n, comps if codec != 1 else 4), dtype=torch.uint8, device=dev, generator=g)
    blocks = pkg.encode_device(codec, src, n, n, src.shape[-1], n_images=batch)
    torch.cuda.synchronize()
    del src
    per_in = blocks.shape[1]
    out_comps = 4 if codec == 1 else 3
    out = torch.empty((batch, n * n * out_comps), dtype=torch.uint8, device=dev)
    def dec():
        rc = L.icamd_decode_device(codec, 0, n, n, 0, batch, per_in, out.shape[1], ctypes.c_void_p(blocks.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), sh)
        assert rc == 0
    t = timeit(dec)
    px = batch * n * n
    b0 = blocks[0].cpu().numpy().tobytes()
    print("decode %-5s %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms per %d x %d^2)  %s" % (
        name, px / t / 1e6, (blocks.numel() + out.numel()) / t / 1e9, t * 1e3, batch, n,
        parity("decode " + name, out[0], T.oracle_decode(codec, b0, n, n).tobytes())))
    # Downsample / Pad / transcode work on one image's grid per call: loop over the batch inside the timed region
    compressor, fmt = (1, 0) if codec == 2 else (0, 0 if codec == 0 else 2)
    dn = torch.empty((batch, per_in // 4), dtype=torch.uint8, device=dev)
    def down():
        for i in range(batch):
            rc = L.icamd_downsample_device(compressor, 2, fmt, n, n, ctypes.c_void_p(blocks[i].data_ptr()),
                                           ctypes.c_void_p(dn[i].data_ptr()), dn.shape[1], sh)
            assert rc == 0
    t = timeit(down, 5)
    print("downsample %-5s %8.0f Mpix/s source  %6.0f GB/s algorithmic (%.3f ms per %d images, one call per image)  %s" % (
        name, px / t / 1e6, (blocks.numel() + dn.numel()) / t / 1e9, t * 1e3, batch,
        parity("downsample " + name, dn[0], T.oracle_downsample(compressor, fmt, b0, n, n, 2))))
    # the same as ONE batched launch (icamd_downsample_batch_device, r04), per ETC1 re-encode strategy
    for strat in ((2, 3) if codec == 2 else (2,)):
        def down_b():
            rc = L.icamd_downsample_batch_device(compressor, strat, fmt, n, n, batch, ctypes.c_void_p(blocks.data_ptr()), per_in,
                                                 ctypes.c_void_p(dn.data_ptr()), dn.shape[1], dn.shape[1], sh)
            assert rc == 0
        t = timeit(down_b, 10)
        print("downsample %-5s batched%s %8.0f Mpix/s source  %6.0f GB/s algorithmic (%.3f ms per %d images, one launch)  %s" % (
            name, " strategy %d" % strat if codec == 2 else "", px / t / 1e6, (blocks.numel() + dn.numel()) / t / 1e9, t * 1e3, batch,
            parity("downsample batched %s s%d" % (name, strat), dn[0], T.oracle_downsample(compressor, fmt, b0, n, n, strat))))
    if CORE:
        if codec == 0:
            work = blocks.clone()
            def tr():
                rc = L.icamd_transcode_dxt1_to_etc1_device(ctypes.c_void_p(work.data_ptr()), work.numel(), sh)
                assert rc == 0
            t = timeit(tr, 10)
            work.copy_(blocks)
            tr()
            print("transcode dxt1->etc1 %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms)  %s" % (
                px / t / 1e6, 2 * work.numel() / t / 1e9, t * 1e3, parity("transcode", work[0], T.oracle_transcode(b0))))
            del work
        del out, dn, blocks
        torch.cuda.empty_cache()
        continue
    # Pad (helper.h:393-477) to a grid two block rows / columns larger: a copy plus one re-encoded / bit-edited border
    ph = pw = n + 8
    pout = torch.empty((batch, ((ph + 3) // 4) * ((pw + 3) // 4) * bb), dtype=torch.uint8, device=dev)
    def pad():
        for i in range(batch):
            rc = L.icamd_pad_device(compressor, 2, fmt, n, n, ctypes.c_void_p(blocks[i].data_ptr()), ph, pw,
                                    ctypes.c_void_p(pout[i].data_ptr()), pout.shape[1], sh)
            assert rc == 0
    t = timeit(pad, 5)
    print("pad %-5s to %d^2 %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms per %d images, one call per image)  %s" % (
        name, ph, px / t / 1e6, (blocks.numel() + pout.numel()) / t / 1e9, t * 1e3, batch,
        parity("pad " + name, pout[0], T.oracle_pad(compressor, fmt, b0, n, n, ph, pw, 2))))
    def pad_b():
        rc = L.icamd_pad_batch_device(compressor, 2, fmt, n, n, batch, ctypes.c_void_p(blocks.data_ptr()), per_in, ph, pw,
                                      ctypes.c_void_p(pout.data_ptr()), pout.shape[1], pout.shape[1], sh)
        assert rc == 0
    pout.zero_()
    t = timeit(pad_b, 10)
    print("pad %-5s to %d^2 batched %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms per %d images, one call)  %s" % (
        name, ph, px / t / 1e6, (blocks.numel() + pout.numel()) / t / 1e9, t * 1e3, batch,
        parity("pad batched " + name, pout[batch - 1], T.oracle_pad(compressor, fmt, blocks[batch - 1].cpu().numpy().tobytes(), n, n, ph, pw, 2))))
    sub = torch.empty((batch, per_in // 4), dtype=torch.uint8, device=dev)
    def subimage():
        for i in range(batch):
            rc = L.icamd_copy_subimage_device(compressor, fmt, n, n, ctypes.c_void_p(blocks[i].data_ptr()), n // 4, n // 4, n // 2, n // 2,
                                              ctypes.c_void_p(sub[i].data_ptr()), sub.shape[1], sh)
            assert rc == 0
    t = timeit(subimage, 5)
    print("copy_subimage %-5s %d^2 of %d^2  %6.0f GB/s (%.3f ms per %d images, one call per image)  %s" % (
        name, n // 2, n, 2 * sub.numel() / t / 1e9, t * 1e3, batch,
        parity("copy_subimage " + name, sub[0], T.oracle_copy_subimage(compressor, fmt, b0, n, n, n // 4, n // 4, n // 2, n // 2))))
    def subimage_b():
        rc = L.icamd_copy_subimage_batch_device(compressor, fmt, n, n, batch, ctypes.c_void_p(blocks.data_ptr()), per_in, n // 4, n // 4,
                                                n // 2, n // 2, ctypes.c_void_p(sub.data_ptr()), sub.shape[1], sub.shape[1], sh)
        assert rc == 0
    sub.zero_()
    t = timeit(subimage_b, 10)
    print("copy_subimage %-5s %d^2 of %d^2 batched  %6.0f GB/s (%.3f ms per %d images, one call)  %s" % (
        name, n // 2, n, 2 * sub.numel() / t / 1e9, t * 1e3, batch,
        parity("copy_subimage batched " + name, sub[batch - 1], T.oracle_copy_subimage(compressor, fmt, blocks[batch - 1].cpu().numpy().tobytes(), n, n, n // 4, n // 4, n // 2, n // 2))))
    colour = (ctypes.c_uint8 * 4)(200, 100, 50, 255)
    def solid():
        for i in range(batch):
            rc = L.icamd_create_solid_device(compressor, fmt, n, n, colour, ctypes.c_void_p(blocks[i].data_ptr()), per_in, sh)
            assert rc == 0
    if codec == 2:
        keep = blocks.clone()
    t = timeit(solid, 5) if codec == 2 else None  # (overwrites `blocks`: last use of the ETC1 grid; restored below)
    if t:
        print("create_solid %-5s %6.0f GB/s written (%.3f ms per %d images of %d^2, one call per image)  %s" % (
            name, blocks.numel() / t / 1e9, t * 1e3, batch, n,
            parity("create_solid " + name, blocks[0], T.oracle_create_solid(compressor, fmt, n, n, list(colour)))))
        colours = (ctypes.c_uint8 * (4 * batch))(*[(37 * i + j * 50) & 255 for i in range(batch) for j in range(3)])
        def solid_b():
            rc = L.icamd_create_solid_batch_device(compressor, fmt, n, n, batch, colours, ctypes.c_void_p(blocks.data_ptr()), per_in, per_in, sh)
            assert rc == 0
        t = timeit(solid_b, 10)
        print("create_solid %-5s batched %6.0f GB/s written (%.3f ms per %d images of %d^2, one call)  %s" % (
            name, blocks.numel() / t / 1e9, t * 1e3, batch, n,
            parity("create_solid batched " + name, blocks[batch - 1], T.oracle_create_solid(compressor, fmt, n, n, [colours[3 * (batch - 1) + j] for j in range(3)] + [0]))))
        blocks.copy_(keep); del keep
    del pout, sub
    if codec == 0:
        work = blocks.clone()
        def tr():
            rc = L.icamd_transcode_dxt1_to_etc1_device(ctypes.c_void_p(work.data_ptr()), work.numel(), sh)
            assert rc == 0
        t = timeit(tr, 10)
        work.copy_(blocks)  # (the timed calls transcoded their own output again and again: one clean pass for the check)
        tr()
        print("transcode dxt1->etc1 %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms)  %s" % (
            px / t / 1e6, 2 * work.numel() / t / 1e9, t * 1e3, parity("transcode", work[0], T.oracle_transcode(b0))))
    del out, dn, blocks
    torch.cuda.empty_cache()

# PVRTC 2 bpp decoder (extension: the reference has none, pvrtc_compressor.cc:669-672)
src = torch.randint(0, 256, (batch, n, n, 4), dtype=torch.uint8, device=dev, generator=g)
blocks = pkg.encode_device(3, src, n, n, 4, n_images=batch)
torch.cuda.synchronize()
del src
out = torch.empty((batch, n * n * 4), dtype=torch.uint8, device=dev)
def dec_pvrtc():
    rc = L.icamd_decode_device(3, 0, n, n, 0, batch, blocks.shape[1], out.shape[1], ctypes.c_void_p(blocks.data_ptr()),
                               ctypes.c_void_p(out.data_ptr()), sh)
    assert rc == 0
t = timeit(dec_pvrtc)
print("decode pvrtc %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms per %d x %d^2)  %s (extension: the reference has no PVRTC decoder)" % (
    batch * n * n / t / 1e6, (blocks.numel() + out.numel()) / t / 1e9, t * 1e3, batch, n,
    parity("decode pvrtc", out[0], T.oracle_decode(3, blocks[0].cpu().numpy().tobytes(), n, n).tobytes())))
def finish():
    print("next rows: %s" % ("ALL LEGS bit-exact vs oracle at the timed shape" if not BAD else "MISMATCH in: " + ", ".join(BAD)))
    if JSON_PATH:
        import json
        with open(JSON_PATH, "w") as f:
            json.dump({"legs": LEGS, "all_bit_exact": not BAD, "shape": "%d x %d^2, device-resident, steady state (0.25 s of untimed calls first)" % (batch, n),
                       "library": L.icamd_version().decode()}, f)
    sys.exit(1 if BAD else 0)
if CORE:
    finish()
# PVRTC 4 bpp decoder (r05: the decoder of the 4 bpp extension encoder; parity unpinned like it)
del blocks
src = torch.randint(0, 256, (batch, n, n, 4), dtype=torch.uint8, device=dev, generator=g)
blocks = pkg.encode_device(4, src, n, n, 4, n_images=batch)
torch.cuda.synchronize()
del src
def dec_pvrtc4():
    rc = L.icamd_decode_device(4, 0, n, n, 0, batch, blocks.shape[1], out.shape[1], ctypes.c_void_p(blocks.data_ptr()),
                               ctypes.c_void_p(out.data_ptr()), sh)
    assert rc == 0
t = timeit(dec_pvrtc4)
print("decode pvrtc4 %8.0f Mpix/s  %6.0f GB/s algorithmic (%.3f ms per %d x %d^2)  %s (extension of an extension: no reference for the format)" % (
    batch * n * n / t / 1e6, (blocks.numel() + out.numel()) / t / 1e9, t * 1e3, batch, n,
    parity("decode pvrtc4", out[0], T.oracle_decode(4, blocks[0].cpu().numpy().tobytes(), n, n).tobytes())))
finish()
