#!/usr/bin/env python3
"""Long seeded random parity soak on the GPU box: the GPU tier's test_random_soak_matches_oracle with more seeds.
usage: python scripts/parity_soak.py [n_seeds [seed_base]]   (each seed = 400 block-codec cases + 60 PVRTC cases; seed_base 0 =
the cases every round repeats, anything else = fresh ones)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed_base = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
t0 = time.time(); cases = 0; bad = 0
for seed in range(n_seeds):
    for (codec, comps, swap, strategy, h, w, pad, img) in T.soak_cases(0xA000 + seed_base + seed, 400, 60, max_h=260, max_w=400, max_log2_pvrtc=6):
        src = T.with_row_padding(img, pad)
        stride = w * comps + pad
        want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, stride=stride)
        out = pkg.encode_device(codec, torch.from_numpy(np.ascontiguousarray(src)).cuda(), h, w, comps, swap_rb=bool(swap),
                                etc_strategy=strategy, row_stride_bytes=stride)
        torch.cuda.synchronize()
        cases += 1
        if out.cpu().numpy().tobytes() != want:
            bad += 1
            print("MISMATCH", seed, codec, comps, swap, strategy, h, w, pad)
print("encode soak: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))

# compressed-domain operations and decoders ("next" rows): random geometry, random content, all strategies
rng = np.random.Generator(np.random.PCG64(0xB10C + seed_base))
ops = 0
for it in range(60 * n_seeds):
    compressor, fmt, strategy = [(T.DXTC, T.RGB, 2), (T.DXTC, T.BGR, 2), (T.DXTC, T.RGBA, 2), (T.DXTC, T.BGRA, 2),
                                 (T.ETC, T.RGB, 0), (T.ETC, T.RGB, 1), (T.ETC, T.RGB, 2), (T.ETC, T.RGB, 3)][it % 8]
    comps = T.comps_of(fmt)
    h, w = int(rng.integers(1, 120)), int(rng.integers(1, 160))
    img = T.soak_image(rng, h, w, comps)
    blocks = T.oracle_compress(compressor, fmt, img.reshape(-1), h, w, 0, strategy)
    ch, cw = 4 * ((h + 3) // 4), 4 * ((w + 3) // 4)
    ph, pw = ch + 4 * int(rng.integers(0, 4)), cw + 4 * int(rng.integers(0, 4))
    checks = [("downsample", pkg.downsample_host(compressor, fmt, blocks, h, w, strategy),
               T.oracle_downsample(compressor, fmt, blocks, h, w, strategy))]
    if (ph, pw) != (ch, cw):
        checks.append(("pad", pkg.pad_host(compressor, fmt, blocks, ch, cw, ph, pw, strategy),
                       T.oracle_pad(compressor, fmt, blocks, ch, cw, ph, pw, strategy)))
    codec = T.ETC1 if compressor == T.ETC else (T.DXT1 if comps == 3 else T.DXT5)
    swap = fmt in (T.BGR, T.BGRA)
    pad = int(rng.integers(0, 6))
    dec = pkg.decode_device(codec, torch.from_numpy(np.frombuffer(blocks, np.uint8).copy()).cuda(), h, w, swap_rb=swap,
                            padding_bytes_per_row=pad)
    torch.cuda.synchronize()
    checks.append(("decode", dec.cpu().numpy().reshape(-1).tobytes(),
                   np.asarray(T.oracle_decode(codec, blocks, h, w, swap=int(swap), pad=pad)).tobytes()))
    if codec == T.DXT1:
        checks.append(("transcode", pkg.transcode_dxt1_to_etc1_host(blocks), T.oracle_transcode(blocks)))
    for name, got, want in checks:
        ops += 1
        if got != want:
            bad += 1
            print("MISMATCH", name, compressor, fmt, strategy, h, w, ph, pw, pad)
print("block-op / decoder soak: %d checks, %d mismatches in total, %.1f s" % (ops, bad, time.time() - t0))

# r02: DXT5 blocks that drive the O(1) alpha index search through every (alpha0, alpha1, alpha) on the DEVICE
# (the CPU tier runs the same sweep through the host-emulated math), and PVRTC encode -> decode on random sizes
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernel_math_host as KM
alphas = KM._alpha_blocks_exhaustive()
g = np.random.Generator(np.random.PCG64(6))
m = alphas.shape[0]
chunk = 1 << 17
for s0 in range(0, m, chunk):
    a = alphas[s0:s0 + chunk]
    k = a.shape[0]
    img = np.zeros((4, 4 * k, 4), np.uint8)
    img[..., :3] = g.integers(0, 256, (4, 4 * k, 3), dtype=np.uint8)
    img[..., 3] = a.reshape(k, 4, 4).transpose(1, 0, 2).reshape(4, 4 * k)
    out = pkg.encode_device(T.DXT5, torch.from_numpy(img).cuda(), 4, 4 * k, 4)
    torch.cuda.synchronize()
    if out.cpu().numpy().tobytes() != T.oracle_encode(T.DXT5, img, 4, 4 * k, 4, threads=16):
        bad += 1
        print("MISMATCH dxt5 alpha sweep chunk", s0)
print("dxt5 alpha sweep on the device: %d blocks, %d mismatches in total, %.1f s" % (m, bad, time.time() - t0))
for it in range(20 * n_seeds):
    n = 8 << int(g.integers(0, 7))
    words = g.integers(0, 256, n * n // 4, dtype=np.uint8) if it % 2 else \
        np.frombuffer(T.oracle_encode(T.PVRTC2, T.soak_image(g, n, n, 4), n, n, 4), np.uint8)
    dec = pkg.decode_device(T.PVRTC2, torch.from_numpy(words.copy()).cuda(), n, n)
    torch.cuda.synchronize()
    if dec.cpu().numpy().tobytes() != T.oracle_decode(T.PVRTC2, words.tobytes(), n, n).tobytes():
        bad += 1
        print("MISMATCH pvrtc decode", n, it)
print("pvrtc decode soak done, %d mismatches in total, %.1f s" % (bad, time.time() - t0))

# r03: the constructed decision-boundary sets of tests/test_kernel_math_host.py (DXT colour index thresholds, ETC1
# shortcut decision points) through the device kernels, at a size where whole waves take each path
import test_kernel_math_host as K
g = np.random.Generator(np.random.PCG64(0xC0DE))
checked = 0
for rep in range(max(1, n_seeds // 50)):
    n = 1 << 17
    for codec, comps, swap in ((T.DXT1, 3, 0), (T.DXT1, 3, 1), (T.DXT1, 4, 0), (T.DXT5, 4, 0), (T.DXT5, 4, 1)):
        strip = K.dxt_boundary_blocks(g, n, comps)
        want = T.oracle_encode(codec, strip, 4, 4 * n, comps, swap, 2, threads=32)
        out = pkg.encode_device(codec, torch.from_numpy(strip).cuda(), 4, 4 * n, comps, swap_rb=bool(swap))
        torch.cuda.synchronize()
        checked += n
        if out.cpu().numpy().tobytes() != want:
            bad += 1
            print("MISMATCH dxt boundary set", codec, comps, swap)
    for comps, strategy in ((3, 2), (3, 0), (3, 1), (4, 2)):
        strip = K.etc_shortcut_blocks(g, n, comps)
        # a block-row strip gives waves of 64 consecutive blocks; also as a 16-row image so that the 16 x 4-block waves mix rows
        for h, w in ((4, 4 * n), (64, n // 4)):
            img = np.ascontiguousarray(strip.reshape(4, 16, n // 4, comps).transpose(1, 0, 2, 3).reshape(64, n // 4, comps)) if h == 64 else strip
            want = T.oracle_encode(T.ETC1, img, h, w, comps, 0, strategy, threads=32)
            out = pkg.encode_device(T.ETC1, torch.from_numpy(img).cuda(), h, w, comps, etc_strategy=strategy)
            torch.cuda.synchronize()
            checked += n
            if out.cpu().numpy().tobytes() != want:
                bad += 1
                print("MISMATCH etc shortcut set", comps, strategy, h, w)
print("constructed boundary sets on the device: %d blocks, %d mismatches in total, %.1f s" % (checked, bad, time.time() - t0))

# r04: the palette-plane kernels (decoders, Downsample incl. the batched entry, DXT1 -> ETC1 transcode) on ARBITRARY block words
# -- DXT1 three-colour mode, equal endpoints, DXT5 six-value alpha, ETC1 differential bases outside 0..31 -- random geometry
g = np.random.Generator(np.random.PCG64(0xD04))
words = 0
for it in range(12 * n_seeds):
    compressor, fmt, codec, strategy = [(T.DXTC, T.RGB, T.DXT1, 2), (T.DXTC, T.BGR, T.DXT1, 2), (T.DXTC, T.RGBA, T.DXT5, 2),
                                        (T.DXTC, T.BGRA, T.DXT5, 2), (T.ETC, T.RGB, T.ETC1, 3), (T.ETC, T.RGB, T.ETC1, 0)][it % 6]
    bb = 16 if codec == T.DXT5 else 8
    rows, cols, n = 2 * int(g.integers(1, 24)), 2 * int(g.integers(1, 40)), int(g.integers(1, 5))
    h, w = 4 * rows, 4 * cols
    raw = g.integers(0, 256, size=(n, rows * cols, bb), dtype=np.uint8)
    kind = int(g.integers(0, 4))
    if codec != T.ETC1 and kind == 1:
        c = raw[:, :, bb - 8:bb - 4]
        raw[:, :, bb - 8:bb - 4] = np.sort(c.copy().view(np.uint16), axis=2).view(np.uint8)
    elif codec != T.ETC1 and kind == 2:
        raw[:, :, bb - 6:bb - 4] = raw[:, :, bb - 8:bb - 6]
    elif codec == T.DXT5 and kind == 3:
        raw[:, :, :2] = np.sort(raw[:, :, :2], axis=2)
    d = torch.from_numpy(raw.reshape(n, -1).copy()).cuda()
    got = pkg.downsample_device(compressor, fmt, d, h, w, etc_strategy=strategy, n_images=n)
    swap = fmt in (T.BGR, T.BGRA)
    dec = pkg.decode_device(codec, d[0].contiguous(), h, w, swap_rb=swap)
    torch.cuda.synchronize()
    for i in range(n):
        words += 1
        if got[i].cpu().numpy().tobytes() != T.oracle_downsample(compressor, fmt, raw[i].tobytes(), h, w, strategy):
            bad += 1
            print("MISMATCH downsample of arbitrary words", codec, fmt, h, w, i, kind)
    words += 1
    if dec.cpu().numpy().reshape(-1).tobytes() != np.asarray(T.oracle_decode(codec, raw[0].tobytes(), h, w, swap=int(swap))).tobytes():
        bad += 1
        print("MISMATCH decode of arbitrary words", codec, fmt, h, w, kind)
    if codec == T.DXT1:
        words += 1
        if pkg.transcode_dxt1_to_etc1_host(raw[0].tobytes()) != T.oracle_transcode(raw[0].tobytes()):
            bad += 1
            print("MISMATCH transcode of arbitrary words", h, w, kind)
print("arbitrary block words through the palette-plane kernels: %d checks, %d mismatches in total, %.1f s" % (words, bad, time.time() - t0))

# r05: the PVRTC one-pass kernel (textures of 512^2 and more: the soak above never reaches it), forced with a random strip height,
# random batch and content, against the oracle; and the PVRTC 4 bpp extension against the oracle's restatement
g = np.random.Generator(np.random.PCG64(0x0E9A55))
one = 0
try:
    for it in range(max(4, n_seeds // 2)):
        n = int(g.choice([512, 512, 1024, 1024, 2048]))
        cnt = int(g.integers(1, 4))
        imgs = np.stack([T.soak_image(g, n, n, 4) for _ in range(cnt)])
        sb = int(g.integers(2, 7))
        pkg.pvrtc_tune(2, sb)
        out = pkg.encode_device(T.PVRTC2, torch.from_numpy(imgs).cuda(), n, n, 4, n_images=cnt)
        torch.cuda.synchronize()
        for i in range(cnt):
            one += 1
            if out[i].cpu().numpy().tobytes() != T.oracle_encode(T.PVRTC2, imgs[i], n, n, 4, threads=16):
                bad += 1
                print("MISMATCH pvrtc one-pass", n, cnt, sb, i)
finally:
    pkg.pvrtc_tune(0, -1)
# r06: the one-pass kernel's HALO form -- regions of one texture (the multi-GPU split, sharding.pvrtc_region) at least 64 block
# columns wide, forced with a random strip height; the region is encoded from the WHOLE texture here (the GPU tier checks that
# nothing outside region + ring is read), random world size, every rank's bytes against the oracle's range
from image_compression_amd import sharding as _sh
halo = 0
try:
    for it in range(max(4, n_seeds // 2)):
        n = int(g.choice([512, 1024, 1024, 2048, 2048, 4096]))
        img = T.soak_image(g, n, n, 4)
        want = T.oracle_encode(T.PVRTC2, img, n, n, 4, threads=16)
        d = torch.from_numpy(np.ascontiguousarray(img)).cuda()
        world = int(g.choice([1, 2, 4, 8, 16]))
        pkg.pvrtc_tune(2, int(g.integers(2, 7)))
        for rank in range(world):
            r = _sh.pvrtc_region(n, world, rank)
            if r["blocks_w"] < 64 or r["blocks_h"] < 4:
                continue
            out = pkg.pvrtc_encode_region_device(d, n, r["first_block"], r["n_blocks"])
            torch.cuda.synchronize()
            halo += 1
            if out.cpu().numpy().tobytes() != want[r["dst_offset_bytes"]:r["dst_offset_bytes"] + r["dst_bytes"]]:
                bad += 1
                print("MISMATCH pvrtc halo form", n, world, rank)
finally:
    pkg.pvrtc_tune(0, -1)
four = 0
for it in range(8 * n_seeds):
    n = 1 << int(g.integers(3, 10))
    img = T.soak_image(g, n, n, 4)
    out = pkg.encode_device(T.PVRTC4, torch.from_numpy(np.ascontiguousarray(img)).cuda(), n, n, 4)
    torch.cuda.synchronize()
    four += 1
    if out.cpu().numpy().tobytes() != T.oracle_encode(T.PVRTC4, img, n, n, 4):
        bad += 1
        print("MISMATCH pvrtc4 (extension)", n, it)
print("pvrtc one-pass soak: %d textures; halo-form regions: %d; pvrtc4 (extension) soak: %d textures; %d mismatches in total, %.1f s" % (one, halo, four, bad, time.time() - t0))
sys.exit(1 if bad else 0)
