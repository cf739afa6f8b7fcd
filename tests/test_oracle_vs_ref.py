"""Pins oracle/ic_oracle.c (our C restatement) against the COMPILED REFERENCE
(oracle/_ref/libic_ref.so, built from /root/reference by `make -C oracle _ref`).
Runs only where that build exists (the build container); on other boxes the
committed fixtures in tests/golden/ carry the pin (test_golden.py)."""
import numpy as np
import pytest

import ic_testlib as T

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")]

DXT_ETC_CASES = [(T.DXTC, T.RGB, s) for s in (2,)] + [(T.DXTC, T.BGR, 2), (T.DXTC, T.RGBA, 2), (T.DXTC, T.BGRA, 2)] + \
    [(T.ETC, T.RGB, s) for s in (0, 1, 2, 3)]


@pytest.mark.parametrize("compressor,fmt,strategy", DXT_ETC_CASES)
@pytest.mark.parametrize("gen", ["noise", "smooth", "flat", "mixed"])
def test_images_match(compressor, fmt, strategy, gen):
    for (h, w, pad) in [(64, 64, 0), (61, 59, 3), (128, 96, 0), (5, 3, 0), (1, 1, 0), (4, 4, 1), (9, 2, 7)]:
        img = T.GENERATORS[gen](h, w, T.comps_of(fmt), index=h * 131 + w)
        src = T.with_row_padding(img, pad)
        a = T.ref_compress(compressor, fmt, src, h, w, pad, strategy)
        b = T.oracle_compress(compressor, fmt, src, h, w, pad, strategy)
        assert a is not None and a == b, (gen, h, w, pad)


@pytest.mark.parametrize("compressor,fmt,strategy", DXT_ETC_CASES)
def test_compress_and_pad_matches(compressor, fmt, strategy):
    for (h, w, ph, pw, pad) in [(30, 30, 40, 48, 8), (4, 4, 8, 8, 0), (7, 9, 7, 20, 0), (16, 16, 8, 8, 0), (3, 3, 17, 3, 2)]:
        img = T.s_mixed(h, w, T.comps_of(fmt), index=7)
        src = T.with_row_padding(img, pad)
        a = T.ref_compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy)
        b = T.oracle_compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy)
        assert a is not None and a == b, (h, w, ph, pw)


def test_random_blocks_dxt_etc():
    # 2^17 independent random 4x4 blocks per codec as one 4 x (4*N) strip (plus structured ones)
    n = 1 << 15
    g = np.random.Generator(np.random.PCG64(99))
    for compressor, fmt, strategy in DXT_ETC_CASES:
        c = T.comps_of(fmt)
        strip = g.integers(0, 256, size=(4, 4 * n, c), dtype=np.uint8)
        # make a quarter of the blocks low-variance so the constant-colour path and ETC diff mode are hit
        base = g.integers(0, 256, size=(1, n // 4, 1, c), dtype=np.int64)
        jit = g.integers(-2, 3, size=(4, n // 4, 4, c), dtype=np.int64)
        strip[:, : n, :] = strip[:, : n, :]
        low = np.clip(base + jit, 0, 255).astype(np.uint8).reshape(4, n, c)
        strip[:, : n] = low
        a = T.ref_compress(compressor, fmt, strip, 4, 4 * n, 0, strategy)
        b = T.oracle_compress(compressor, fmt, strip, 4, 4 * n, 0, strategy)
        assert a == b


def test_all_solid_colours_on_lattice():
    # every 8-bit value through the constant-colour table, per channel, for all four formats
    vals = np.arange(256, dtype=np.uint8)
    for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
        c = T.comps_of(fmt)
        for ch in range(3):
            img = np.zeros((4, 4 * 256 * 4, c), np.uint8)
            others = [0, 37, 128, 255]
            for oi, o in enumerate(others):
                blockcols = np.repeat(vals, 4)
                sl = slice(oi * 1024, (oi + 1) * 1024)
                img[:, sl, :3] = o
                img[:, sl, ch] = blockcols
                if c == 4:
                    img[:, sl, 3] = blockcols[::-1]
            a = T.ref_compress(T.DXTC, fmt, img, 4, img.shape[1])
            b = T.oracle_compress(T.DXTC, fmt, img, 4, img.shape[1])
            assert a == b


def test_pvrtc_matches():
    for n in (8, 16, 32, 64, 128, 256):
        for gen in ("noise", "smooth", "flat", "mixed"):
            img = T.GENERATORS[gen](n, n, 4, index=n)
            a = T.ref_compress(T.PVRTC, T.RGBA, img, n, n)
            b = T.oracle_compress(T.PVRTC, T.RGBA, img, n, n)
            assert a is not None and a == b, (n, gen)
    # zero-axis quirk: blocks whose channel maxima are all zero reference image pixel 0
    img = np.zeros((32, 32, 4), np.uint8)
    img[0, 0] = (250, 3, 7, 255)
    img[8:, :, 1] = 200
    img[:, 16:, 3] = 255
    assert T.ref_compress(T.PVRTC, T.RGBA, img, 32, 32) == T.oracle_compress(T.PVRTC, T.RGBA, img, 32, 32)


def test_argument_validation_matches():
    img = T.s_noise(16, 16, 4)
    cases = []
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for (h, w, pad) in [(16, 16, 0), (8, 8, 0), (8, 16, 0), (12, 12, 0), (16, 16, 4), (0, 4, 0), (4, 0, 0), (4, 4, 0)]:
                cases.append((compressor, fmt, h, w, pad))
    for compressor, fmt, h, w, pad in cases:
        assert T.ref_size(compressor, fmt, h, w) == T.oracle_size(compressor, fmt, h, w)
        n = T.ref_size(compressor, fmt, h, w)
        if h * (w * T.comps_of(fmt) + pad) > img.size:
            continue
        for out_size in (n, n + 8):
            a = T.ref_compress(compressor, fmt, img, h, w, pad, out_size=out_size)
            b = T.oracle_compress(compressor, fmt, img, h, w, pad, out_size=out_size)
            assert a == b, (compressor, fmt, h, w, pad, out_size)


def test_decoders_match():
    for compressor, fmt, codec in [(T.DXTC, T.RGB, T.DXT1), (T.DXTC, T.BGR, T.DXT1), (T.DXTC, T.RGBA, T.DXT5),
                                   (T.DXTC, T.BGRA, T.DXT5), (T.ETC, T.RGB, T.ETC1)]:
        for (h, w) in [(64, 64), (13, 7)]:
            img = T.s_mixed(h, w, T.comps_of(fmt), index=5)
            blocks = T.ref_compress(compressor, fmt, img, h, w)
            a = T.ref_decompress(compressor, fmt, blocks, h, w)
            b = T.oracle_decode(codec, blocks, h, w, swap=int(fmt in (T.BGR, T.BGRA)))
            assert a is not None and np.array_equal(a, b), (compressor, fmt, h, w)
    # arbitrary (not encoder-produced) DXT blocks, including 3-colour mode
    g = np.random.Generator(np.random.PCG64(5))
    blocks = g.integers(0, 256, size=64 * 64 // 16 * 8, dtype=np.uint8).tobytes()
    assert np.array_equal(T.ref_decompress(T.DXTC, T.RGB, blocks, 64, 64), T.oracle_decode(T.DXT1, blocks, 64, 64))
    blocks = g.integers(0, 256, size=64 * 64 // 16 * 16, dtype=np.uint8).tobytes()
    assert np.array_equal(T.ref_decompress(T.DXTC, T.RGBA, blocks, 64, 64), T.oracle_decode(T.DXT5, blocks, 64, 64))
    # arbitrary ETC1 words: half of them differential, many with base + delta outside 0..31 (etc_compressor.cc:198-273)
    for seed in range(8):
        blocks = T.random_blocks(T.ETC1, 128, 128, 200 + seed)
        assert np.array_equal(T.ref_decompress(T.ETC, T.RGB, blocks, 128, 128), T.oracle_decode(T.ETC1, blocks, 128, 128))


# ---- compressed-domain operations

OPS_CASES = [(T.DXTC, T.RGB, 2), (T.DXTC, T.BGR, 2), (T.DXTC, T.RGBA, 2), (T.DXTC, T.BGRA, 2), (T.ETC, T.RGB, 0),
             (T.ETC, T.RGB, 2), (T.ETC, T.RGB, 3)]


@pytest.mark.parametrize("compressor,fmt,strategy", OPS_CASES)
def test_pad_downsample_subimage_match(compressor, fmt, strategy):
    for (h, w) in [(32, 48), (13, 7), (64, 8), (8, 64), (4, 4), (2, 2), (1, 4), (4, 1), (16, 4), (4, 16), (3, 8)]:
        img = T.s_mixed(h, w, T.comps_of(fmt), index=h + w)
        blocks = T.ref_compress(compressor, fmt, img, h, w, 0, strategy)
        ch, cw = 4 * ((h + 3) // 4), 4 * ((w + 3) // 4)
        for (ph, pw) in [(h + 9, w + 5), (ch, cw + 8), (ch + 4, cw), (h, w), (ch + 1, cw + 1)]:
            a = T.ref_pad(compressor, fmt, blocks, h, w, ph, pw, strategy)
            b = T.oracle_pad(compressor, fmt, blocks, ch, cw, ph, pw, strategy)
            assert a is not None and a[0] == b, (h, w, ph, pw)
        a = T.ref_downsample(compressor, fmt, blocks, h, w, strategy)
        b = T.oracle_downsample(compressor, fmt, blocks, h, w, strategy)
        assert (a[0] if a else None) == b, (h, w)
        for (r, c, sh, sw) in [(0, 0, ch, cw), (4, 4, 4, 4), (0, 4, 8, 4), (4, 0, 4, 8), (2, 0, 4, 4), (0, 0, ch + 4, 4)]:
            a = T.ref_copy_subimage(compressor, fmt, blocks, h, w, r, c, sh, sw)
            b = T.oracle_copy_subimage(compressor, fmt, blocks, ch, cw, r, c, sh, sw)
            assert a == b, (h, w, r, c, sh, sw)


def test_solid_and_transcode_match():
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for color in [(200, 100, 50, 77), (0, 0, 0, 0), (255, 255, 255, 255), (1, 2, 3, 4), (129, 7, 250, 128)]:
                for (h, w) in [(8, 12), (5, 3), (1, 1)]:
                    assert T.ref_create_solid(compressor, fmt, h, w, color) == T.oracle_create_solid(compressor, fmt, h, w, color)
    for (h, w) in [(64, 64), (13, 7)]:
        for gen in ("noise", "smooth", "flat", "mixed"):
            img = T.GENERATORS[gen](h, w, 3, index=9)
            blocks = T.ref_compress(T.DXTC, T.RGB, img, h, w)
            assert T.ref_transcode(blocks, h, w) == T.oracle_transcode(blocks)
    g = np.random.Generator(np.random.PCG64(3))
    raw = g.integers(0, 256, size=8 * 4096, dtype=np.uint8).tobytes()  # arbitrary DXT1 blocks (3-colour mode etc.)
    assert T.ref_transcode(raw, 256, 256) == T.oracle_transcode(raw)
