"""GPU tier (-m gpu): the HIP kernels, called through the C ABI (libic_amd.so), against
(1) the committed golden vectors generated from the compiled reference, and
(2) the oracle (oracle/ic_oracle.c) on the same seeded inputs, bit-for-bit (every codec is pure integer).
"""
import hashlib

import numpy as np
import pytest

import golden_cases as G
import ic_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    import ic_amd_loader
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    p = ic_amd_loader.load_package()
    assert p.lib().icamd_device_count() >= 1
    return p


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().tobytes()


# ---- golden vectors through the reference-shaped entry points (host buffers, like Compressor::Compress)

def test_golden_known_answers_host_api(pkg):
    def compress(compressor, fmt, src, h, w, pad, strategy):
        return pkg.compress_host(compressor, fmt, src, h, w, padding_bytes_per_row=pad, etc_strategy=strategy)

    def compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy):
        return pkg.compress_host(compressor, fmt, src, h, w, padding_bytes_per_row=pad, etc_strategy=strategy,
                                 padded=(ph, pw))
    assert G.check_kats(compress, compress_and_pad) >= 20
    assert G.check_mixed64(compress) == 9


def test_golden_hashes_device_api(pkg):
    def compress(compressor, fmt, src, h, w, pad, strategy):
        out = pkg.compress_device(compressor, fmt, _dev(src), h, w, padding_bytes_per_row=pad, etc_strategy=strategy)
        return None if out is None else _host(out)

    def compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy):
        out = pkg.compress_device(compressor, fmt, _dev(src), h, w, padding_bytes_per_row=pad, etc_strategy=strategy,
                                  padded=(ph, pw))
        return None if out is None else _host(out)
    assert G.check_hashes(compress, compress_and_pad) > 100


# ---- oracle comparisons on seeded inputs

CASES = [(T.DXT1, 3, 0, 2), (T.DXT1, 3, 1, 2), (T.DXT1, 4, 0, 2), (T.DXT1, 4, 1, 2), (T.DXT5, 4, 0, 2),
         (T.DXT5, 4, 1, 2), (T.ETC1, 3, 0, 0), (T.ETC1, 3, 0, 1), (T.ETC1, 3, 0, 2), (T.ETC1, 3, 0, 3),
         (T.ETC1, 4, 0, 2), (T.ETC1, 4, 0, 3)]
SHAPES = [(64, 64, 0), (61, 59, 3), (128, 260, 0), (5, 3, 0), (1, 1, 0), (4, 4, 1), (9, 2, 7), (257, 1023, 5)]


@pytest.mark.parametrize("codec,comps,swap,strategy", CASES)
def test_encode_matches_oracle(pkg, codec, comps, swap, strategy):
    for gen in ("noise", "smooth", "flat", "mixed"):
        for (h, w, pad) in SHAPES:
            img = T.GENERATORS[gen](h, w, comps, index=h * 7 + w)
            src = T.with_row_padding(img, pad)
            stride = w * comps + pad
            want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, stride=stride)
            out = pkg.encode_device(codec, _dev(src), h, w, comps, swap_rb=bool(swap), etc_strategy=strategy,
                                    row_stride_bytes=stride)
            assert _host(out) == want, (gen, h, w, pad)


@pytest.mark.parametrize("codec,comps,swap,strategy", CASES)
def test_compress_and_pad_grid_matches_oracle(pkg, codec, comps, swap, strategy):
    for (h, w, gh, gw, pad) in [(30, 30, 40, 48, 8), (4, 4, 8, 8, 0), (7, 9, 7, 20, 0), (3, 3, 17, 3, 2)]:
        img = T.s_mixed(h, w, comps, index=60)
        src = T.with_row_padding(img, pad)
        stride = w * comps + pad
        want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, gh=gh, gw=gw, stride=stride)
        out = pkg.encode_device(codec, _dev(src), h, w, comps, swap_rb=bool(swap), etc_strategy=strategy,
                                grid_height=gh, grid_width=gw, row_stride_bytes=stride)
        assert _host(out) == want, (h, w, gh, gw)


def test_argument_validation_matches_oracle(pkg):
    img = T.s_noise(16, 16, 4)
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for (h, w, pad) in [(16, 16, 0), (8, 8, 0), (8, 16, 0), (12, 12, 0), (16, 16, 4), (0, 4, 0), (4, 0, 0)]:
                n = T.oracle_size(compressor, fmt, h, w)
                assert pkg.compute_compressed_data_size(compressor, fmt, h, w) == n
                if h * (w * T.comps_of(fmt) + pad) > img.size:
                    continue
                for out_size in (n, n + 8):
                    want = T.oracle_compress(compressor, fmt, img, h, w, pad, out_size=out_size)
                    got = pkg.compress_host(compressor, fmt, img.reshape(-1), h, w, padding_bytes_per_row=pad,
                                            out_size=out_size)
                    assert got == want, (compressor, fmt, h, w, pad, out_size)
    assert pkg.compress_host(T.PVRTC, T.RGBA, img.reshape(-1), 16, 16, padded=(16, 16)) is None  # pvrtc.cc:684-691


def test_batch_launch_equals_per_image(pkg):
    import torch
    n, h, w = 5, 36, 52
    for codec, comps in ((T.DXT1, 4), (T.DXT5, 4), (T.ETC1, 3)):
        imgs = np.stack([T.s_mixed(h, w, comps, index=i) for i in range(n)])
        out = pkg.encode_device(codec, _dev(imgs), h, w, comps, n_images=n)
        torch.cuda.synchronize()
        for i in range(n):
            assert out[i].cpu().numpy().tobytes() == T.oracle_encode(codec, imgs[i], h, w, comps)


def test_batch_of_more_images_than_one_grid_dimension(pkg):
    """70 000 images of 4x4 pixels in one call (grid.z holds 65 535): byte-identical to ONE 4 x 280 000 image."""
    n = 70000
    rng = np.random.Generator(np.random.PCG64(7))
    for codec, comps in ((T.DXT1, 3), (T.ETC1, 3), (T.DXT5, 4)):
        imgs = rng.integers(0, 256, (n, 4, 4, comps), dtype=np.uint8)
        out = pkg.encode_device(codec, _dev(imgs), 4, 4, comps, n_images=n)
        want = T.oracle_encode(codec, imgs.reshape(4 * n, 4, comps), 4 * n, 4, comps)
        assert _host(out) == want, codec


# ---- BASELINE.json full sizes

def test_full_size_dxt1_4096_rgba8_and_rgb888(pkg):
    h = w = 4096
    for comps in (4, 3):
        img = T.s_smooth(h, w, comps, index=4)
        img[:1024, :1024] = T.s_flat(1024, 1024, comps, index=4)
        img[1024:2048, :1024] = T.s_noise(1024, 1024, comps, index=4)
        out = pkg.encode_device(T.DXT1, _dev(img), h, w, comps)
        got = _host(out)
        want = T.oracle_encode(T.DXT1, img, h, w, comps, threads=8)
        assert got == want
        # size-independent property: block-aligned crops encode to the corresponding block rows/cols
        crop = np.ascontiguousarray(img[512:1024, 256:1280])
        sub = _host(pkg.encode_device(T.DXT1, _dev(crop), 512, 1024, comps))
        full = np.frombuffer(got, np.uint8).reshape(1024, 1024, 8)
        assert sub == full[128:256, 64:320].tobytes()
        # kBGR / kBGRA order of the same bytes (R and B swapped at the pixel read, dxtc.cc:288,295,333; const path quirk :360)
        got_bgr = _host(pkg.encode_device(T.DXT1, _dev(img), h, w, comps, swap_rb=True))
        assert got_bgr == T.oracle_encode(T.DXT1, img, h, w, comps, swap=1, threads=8)
        assert got_bgr != got


def test_full_size_dxt5_8192(pkg):
    h = w = 8192
    img = T.s_smooth(h, w, 4, index=5)
    img[:2048, :2048] = T.s_noise(2048, 2048, 4, index=5)
    for swap in (0, 1):  # kRGBA, kBGRA
        out = pkg.encode_device(T.DXT5, _dev(img), h, w, 4, swap_rb=bool(swap))
        got = _host(out)
        want = T.oracle_encode(T.DXT5, img, h, w, 4, swap=swap, threads=8)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), swap


def test_large_single_image_16384_dxt1(pkg):
    """1 GiB source in one image (offsets beyond 2^30, 16 M blocks): hash against the oracle, which runs
    slab-parallel on the host cores; plus a wide, short image whose block rows span many column tiles."""
    import os
    import torch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    h = w = 16384
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
    src[: h // 2, : w // 2] = 200  # constant-colour quadrant
    out = pkg.encode_device(T.DXT1, src, h, w, 4)
    got = _host(out)
    want = T.oracle_encode(T.DXT1, src.cpu().numpy(), h, w, 4, threads=cores)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    del src, out
    h, w = 12, 70001  # ragged width, 17 501 block columns = 69 column tiles
    img = T.s_mixed(h, w, 3, index=9)
    assert _host(pkg.encode_device(T.ETC1, _dev(img), h, w, 3, etc_strategy=3)) == T.oracle_encode(T.ETC1, img, h, w, 3, 0, 3)


def test_etc1_every_colour_as_a_one_colour_block(pkg):
    """All 2^24 colours as constant 4x4 blocks (the wave-uniform one-colour form of the ETC1 kernels: one pixel against the
    32 candidates) against the oracle's full search, kSmallerError and kSplitHorizontally; plus images that mix one-colour
    waves, one-colour blocks inside busy waves and noise, so that all three wave forms run next to each other."""
    import os
    import torch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    side = 4096                                   # 4096 x 4096 blocks = every colour once; two halves of 2^23 blocks
    for half in range(2):
        idx = torch.arange(half << 23, (half + 1) << 23, dtype=torch.int64, device="cuda").reshape(side // 2, side)
        rgb = torch.stack([idx & 255, (idx >> 8) & 255, idx >> 16], dim=2).to(torch.uint8)   # (2048, 4096, 3) block colours
        src = rgb.repeat_interleave(4, dim=0).repeat_interleave(4, dim=1).contiguous()        # 8192 x 16384 px
        del idx, rgb
        h, w = src.shape[0], src.shape[1]
        host = src.cpu().numpy()
        for strategy in (2, 0):
            got = _host(pkg.encode_device(T.ETC1, src, h, w, 3, etc_strategy=strategy))
            want = T.oracle_encode(T.ETC1, host, h, w, 3, 0, strategy, threads=cores)
            assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), (half, strategy)
        del src, host
    g = np.random.Generator(np.random.PCG64(31))
    h = w = 1024
    img = g.integers(0, 256, size=(h // 16, w // 16, 1, 1, 3), dtype=np.uint8)               # 16 x 16 px tiles of one colour
    img = np.broadcast_to(img, (h // 16, w // 16, 16, 16, 3)).transpose(0, 2, 1, 3, 4).reshape(h, w, 3).copy()
    noisy = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    tile_is_noise = np.repeat(np.repeat(g.random((h // 16, w // 16)) < 0.2, 16, axis=0), 16, axis=1)
    img[tile_is_noise] = noisy[tile_is_noise]
    img[512:, :512] = noisy[512:, :512]
    for comps in (3, 4):
        im = img if comps == 3 else np.concatenate([img, noisy[..., :1]], axis=2)
        for strategy in (0, 1, 2):
            assert _host(pkg.encode_device(T.ETC1, _dev(im), h, w, comps, etc_strategy=strategy)) == \
                T.oracle_encode(T.ETC1, im, h, w, comps, 0, strategy, threads=8), (comps, strategy)


def test_etc1_batch_1024(pkg):
    import torch
    # config 4 shape (1024x1024 textures, batch sharded over GPUs): a per-GPU sub-batch here, oracle on 2 of them
    n, h, w = 8, 1024, 1024
    imgs = np.stack([T.s_smooth(h, w, 3, index=i) if i % 2 else T.s_noise(h, w, 3, index=i) for i in range(n)])
    out = pkg.encode_device(T.ETC1, _dev(imgs), h, w, 3, n_images=n)
    torch.cuda.synchronize()
    for i in (0, n - 1):
        assert out[i].cpu().numpy().tobytes() == T.oracle_encode(T.ETC1, imgs[i], h, w, 3, threads=8)
    # golden: 1024^2 noise/smooth hashes are in hashes.json and were checked by test_golden_hashes_device_api


# ---- PVRTC1 2bpp (fused tile kernel with LDS halo)

def test_pvrtc_matches_oracle(pkg):
    for n in (8, 16, 32, 64, 128, 256, 512):
        for gen in ("noise", "smooth", "flat", "mixed"):
            img = T.GENERATORS[gen](n, n, 4, index=n)
            out = pkg.encode_device(T.PVRTC2, _dev(img), n, n, 4)
            assert _host(out) == T.oracle_encode(T.PVRTC2, img, n, n, 4), (n, gen)
    img = np.zeros((32, 32, 4), np.uint8)  # never-updated maxima refer to image pixel 0 (pvrtc.cc:268-269)
    img[0, 0] = (250, 3, 7, 255)
    img[8:, :, 1] = 200
    img[:, 16:, 3] = 255
    assert _host(pkg.encode_device(T.PVRTC2, _dev(img), 32, 32, 4)) == T.oracle_encode(T.PVRTC2, img, 32, 32, 4)


def test_pvrtc_batch_and_full_size_4096(pkg):
    import torch
    imgs = np.stack([T.s_mixed(128, 128, 4, index=i) for i in range(3)])
    out = pkg.encode_device(T.PVRTC2, _dev(imgs), 128, 128, 4, n_images=3)
    torch.cuda.synchronize()
    for i in range(3):
        assert out[i].cpu().numpy().tobytes() == T.oracle_encode(T.PVRTC2, imgs[i], 128, 128, 4)
    n = 4096  # BASELINE.json config 5
    img = T.s_smooth(n, n, 4, index=6)
    img[:1024, :1024] = T.s_noise(1024, 1024, 4, index=6)
    img[1024:2048, :1024] = T.s_flat(1024, 1024, 4, index=6)
    got = _host(pkg.encode_device(T.PVRTC2, _dev(img), n, n, 4))
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(T.oracle_encode(T.PVRTC2, img, n, n, 4)).hexdigest()
    # toroidal property: rolling the image by whole tiles permutes blocks but keeps each block's bytes
    # (every block sees the same wrapped neighbourhood) as long as image pixel 0 stays the same colour
    small = T.s_noise(256, 256, 4, index=8)
    small[..., :3] |= 1  # no all-zero channel, so pixel 0 is never consulted
    a = np.frombuffer(_host(pkg.encode_device(T.PVRTC2, _dev(small), 256, 256, 4)), np.uint8).reshape(-1, 8)
    rolled = np.ascontiguousarray(np.roll(small, (64, 128), axis=(0, 1)))
    b = np.frombuffer(_host(pkg.encode_device(T.PVRTC2, _dev(rolled), 256, 256, 4)), np.uint8).reshape(-1, 8)

    def z(bx, by):
        r = 0
        for j in range(16):
            r |= ((bx >> j) & 1) << (2 * j + 1) | ((by >> j) & 1) << (2 * j)
        return r
    for (bx, by) in [(0, 0), (5, 7), (31, 63), (16, 16), (17, 3)]:
        assert a[z(bx, by)].tobytes() == b[z((bx + 16) % 32, (by + 16) % 64)].tobytes()


def test_pvrtc_large_batches(pkg):
    """Batches of large textures in one call (whole-batch morph launch, then whole-batch encode launch with the
    LDS-staged Z-order stores), every texture against the oracle."""
    import torch
    for n, size in ((7, 4096), (5, 2048), (9, 1024)):
        g = torch.Generator(device="cuda")
        g.manual_seed(size + n)
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device="cuda", generator=g)
        src[1::3, :, :, 3] = 255                       # opaque textures
        src[2::3] = (src[2::3] >> 3) + 100             # low-contrast textures (1BPP blocks)
        src[:, : size // 4, : size // 2] = src[:, :1, :1]  # flat area, colour of each texture's pixel 0
        out = pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n)
        torch.cuda.synchronize()
        host = src.cpu().numpy()
        for i in range(n):
            want = T.oracle_encode(T.PVRTC2, host[i], size, size, 4)
            assert hashlib.sha256(out[i].cpu().numpy().tobytes()).hexdigest() == hashlib.sha256(want).hexdigest(), (n, size, i)


@pytest.fixture
def pvrtc_auto(pkg):
    """icamd_pvrtc2_tune is process-wide: whatever a test forces, the automatic selection is back afterwards."""
    yield
    assert pkg.pvrtc_tune(0, -1)


def test_pvrtc_onepass_kernel_every_strip_height_matches_oracle(pkg, pvrtc_auto):
    """r05: icamd_pvrtc2_onepass_kernel (morph + modulate + encode in one read of the pixels; one workgroup = one whole
    block row of the texture wide, 1 / 2 / 4 / 8 waves) forced for every eligible size and every strip height, against the
    oracle: all contents, the image-pixel-0 rule, a batch with padded image strides, an 8-mod-16 destination (no staged
    stores) -- and the same inputs through the morph + encode pair."""
    import ctypes
    import torch
    for n in (512, 1024, 2048):
        imgs = [T.GENERATORS[gen](n, n, 4, index=n + 1) for gen in ("noise", "smooth", "flat", "mixed")]
        z = np.zeros((n, n, 4), np.uint8)  # never-updated maxima refer to image pixel 0 (pvrtc.cc:268-269)
        z[0, 0] = (250, 3, 7, 255)
        z[n // 4:, :, 1] = 200
        z[:, n // 2:, 3] = 255
        imgs.append(z)
        want = [T.oracle_encode(T.PVRTC2, im, n, n, 4, threads=8) for im in imgs]
        d = _dev(np.stack(imgs))
        for mode, strips in ((1, (-1,)), (2, (2, 3, 4, 5, 6))):
            for sb in strips:
                assert pkg.pvrtc_tune(mode, sb)
                out = pkg.encode_device(T.PVRTC2, d, n, n, 4, n_images=len(imgs))
                torch.cuda.synchronize()
                for i in range(len(imgs)):
                    assert out[i].cpu().numpy().tobytes() == want[i], (n, mode, sb, i)
                one = pkg.encode_device(T.PVRTC2, d[3], n, n, 4)  # one texture per call
                assert _host(one) == want[3], (n, mode, sb)
    # 4096^2 (eight waves per workgroup): one mixed texture, every strip height
    n = 4096
    img = T.s_smooth(n, n, 4, index=16)
    img[:1024, :1024] = T.s_noise(1024, 1024, 4, index=16)
    img[1024:2048, 2048:] = T.s_flat(1024, 2048, 4, index=16)
    want = hashlib.sha256(T.oracle_encode(T.PVRTC2, img, n, n, 4, threads=8)).hexdigest()
    d = _dev(img)
    for sb in (2, 3, 4, 5, 6):
        assert pkg.pvrtc_tune(2, sb)
        assert hashlib.sha256(_host(pkg.encode_device(T.PVRTC2, d, n, n, 4))).hexdigest() == want, sb
    # padded image strides on both sides, and a destination that is only 8-byte aligned
    n, cnt = 1024, 3
    per = n * n // 4
    imgs = np.stack([T.s_mixed(n, n, 4, index=40 + i) for i in range(cnt)])
    want = [T.oracle_encode(T.PVRTC2, imgs[i], n, n, 4, threads=8) for i in range(cnt)]
    src = torch.zeros((cnt, n * n * 4 + 4096), dtype=torch.uint8, device="cuda")
    src[:, : n * n * 4] = _dev(imgs).view(cnt, -1)
    for off, dst_stride in ((0, per + 256), (8, per + 8), (8, per + 16)):
        buf = torch.zeros(cnt * dst_stride + 64, dtype=torch.uint8, device="cuda")
        for mode, sb in ((2, 3), (2, 5), (1, -1)):
            assert pkg.pvrtc_tune(mode, sb)
            buf.zero_()
            st = pkg.lib().icamd_encode_device(T.PVRTC2, 2, 4, 0, n, n, n, n, n * 4, cnt, n * n * 4 + 4096, dst_stride,
                                               ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(buf.data_ptr() + off), None)
            assert st == 0
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            for i in range(cnt):
                assert host[off + i * dst_stride: off + i * dst_stride + per].tobytes() == want[i], (off, dst_stride, mode, sb, i)
            assert not host[off + (cnt - 1) * dst_stride + per:].any() and not host[:off].any()


def test_pvrtc4_extension_matches_the_oracles_restatement_and_decodes(pkg, pvrtc_auto):
    """PVRTC1 4 bpp (EXTENSION, parity unpinned: BASELINE config 5 names it, the reference has none): the device kernels
    against oracle/ic_oracle.c's restatement of the same rules (8^2 ... 4096^2, all contents, the image-pixel-0 rule, batches
    with padded strides), and the decoded result against the source (the only outside check there is)."""
    import torch
    for n in (8, 16, 32, 64, 256, 1024):
        for gen in ("noise", "smooth", "flat", "mixed"):
            img = T.GENERATORS[gen](n, n, 4, index=n + 9)
            out = pkg.encode_device(T.PVRTC4, _dev(img), n, n, 4)
            assert _host(out) == T.oracle_encode(T.PVRTC4, img, n, n, 4), (n, gen)
    img = np.zeros((32, 32, 4), np.uint8)
    img[0, 0] = (250, 3, 7, 255)
    img[8:, :, 1] = 200
    img[:, 16:, 3] = 255
    assert _host(pkg.encode_device(T.PVRTC4, _dev(img), 32, 32, 4)) == T.oracle_encode(T.PVRTC4, img, 32, 32, 4)
    # both forms on every eligible size: the morph + encode pair and the one-pass kernel (icamd_pvrtc4_onepass_kernel: a
    # workgroup = one block row of the texture, 64 ... 1 024 lanes) at every strip height
    for n in (256, 512, 2048):
        imgs = np.stack([T.GENERATORS[gen](n, n, 4, index=n + 2) for gen in ("noise", "smooth", "flat", "mixed")])
        want = [T.oracle_encode(T.PVRTC4, im, n, n, 4) for im in imgs]
        d = _dev(imgs)
        for mode, strips in ((1, (-1,)), (2, (1, 2, 3, 4, 5, 6, 7))):
            for sb in strips:
                assert pkg.pvrtc_tune(mode, sb)
                out = pkg.encode_device(T.PVRTC4, d, n, n, 4, n_images=4)
                torch.cuda.synchronize()
                for i in range(4):
                    assert out[i].cpu().numpy().tobytes() == want[i], (n, mode, sb, i)
    assert pkg.pvrtc_tune(2, 4)  # (the rest of this test: the one-pass form where eligible, then automatic)
    n, cnt = 512, 5
    imgs = np.stack([T.s_mixed(n, n, 4, index=70 + i) for i in range(cnt)])
    src = torch.zeros((cnt, n * n * 4 + 64), dtype=torch.uint8, device="cuda")
    src[:, : n * n * 4] = _dev(imgs).view(cnt, -1)
    per = n * n // 2
    out = torch.zeros((cnt, per + 24), dtype=torch.uint8, device="cuda")
    assert pkg.lib().icamd_encode_device(T.PVRTC4, 2, 4, 0, n, n, n, n, n * 4, cnt, n * n * 4 + 64, per + 24, src.data_ptr(),
                                         out.data_ptr(), None) == pkg.OK
    torch.cuda.synchronize()
    for i in range(cnt):
        assert out[i, :per].cpu().numpy().tobytes() == T.oracle_encode(T.PVRTC4, imgs[i], n, n, 4), i
        assert not out[i, per:].any()
    # an 8-mod-16 destination: no staged 16-byte stores
    buf = torch.zeros(cnt * per + 64, dtype=torch.uint8, device="cuda")
    assert pkg.lib().icamd_encode_device(T.PVRTC4, 2, 4, 0, n, n, n, n, n * 4, cnt, n * n * 4 + 64, per, src.data_ptr(),
                                         buf.data_ptr() + 8, None) == pkg.OK
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    for i in range(cnt):
        assert host[8 + i * per: 8 + (i + 1) * per].tobytes() == T.oracle_encode(T.PVRTC4, imgs[i], n, n, 4), i
    assert pkg.pvrtc_tune(0, -1)
    n = 4096
    img = T.s_smooth(n, n, 4, index=21)
    img[:1024, :1024] = T.s_noise(1024, 1024, 4, index=21)
    got = _host(pkg.encode_device(T.PVRTC4, _dev(img), n, n, 4))  # (16 Mpixel: the automatic choice is the one-pass kernel)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(T.oracle_encode(T.PVRTC4, img, n, n, 4)).hexdigest()
    assert pkg.pvrtc_tune(1, -1)
    assert _host(pkg.encode_device(T.PVRTC4, _dev(img), n, n, 4)) == got
    dec = T.oracle_decode(T.PVRTC4, got, n, n).reshape(n, n, 4).astype(np.float64)
    mse = float(((dec[1024:] - img[1024:].astype(np.float64)) ** 2).mean())
    assert 10 * np.log10(255.0 * 255.0 / mse) > 18.0  # the smooth part (ramps + 5-bit noise + random alpha)
    assert pkg.encoded_size(T.PVRTC4, 64, 64) == 2048 and pkg.pvrtc4_workspace_size(64, 3) == 3 * 256 * 8
    assert pkg.lib().icamd_encode_device(T.PVRTC4, 2, 4, 0, 16, 32, 16, 32, 128, 1, 0, 0, src.data_ptr(), out.data_ptr(), None) == pkg.FALSE
    assert pkg.lib().icamd_encode_device(T.PVRTC4, 2, 3, 0, 16, 16, 16, 16, 48, 1, 0, 0, src.data_ptr(), out.data_ptr(), None) == pkg.FALSE


def test_pvrtc_automatic_path_selection_and_both_paths_on_a_full_batch(pkg, pvrtc_auto):
    """The launch shapes BASELINE config 5 is quoted on (16 x 4096^2) and its neighbours: the automatic selection, the
    forced pair and the forced one-pass kernel produce the same bytes; texture 0 and the last one against the oracle."""
    import torch
    for n, size in ((16, 4096), (64, 2048), (256, 1024), (600, 512), (3, 4096), (1, 2048)):
        g = torch.Generator(device="cuda")
        g.manual_seed(size * 3 + n)
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device="cuda", generator=g)
        src[1::3, :, :, 3] = 255
        src[2::3] = (src[2::3] >> 3) + 100
        outs = []
        for mode in (0, 1, 2):
            assert pkg.pvrtc_tune(mode, -1)
            outs.append(pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n).clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (n, size)
        for i in (0, n - 1):
            want = T.oracle_encode(T.PVRTC2, src[i].cpu().numpy(), size, size, 4, threads=8)
            assert outs[0][i].cpu().numpy().tobytes() == want, (n, size, i)


def test_pvrtc_onepass_launches_can_be_captured_without_a_workspace(pkg, pvrtc_auto):
    """The one-pass kernel keeps nothing between kernels, so a launch that takes it can be captured into a HIP graph with
    no caller-owned workspace (the pair refuses that, test_pvrtc_graphs_keep_their_own_workspace)."""
    import torch
    n, size = 4, 1024
    imgs = np.stack([T.s_mixed(size, size, 4, index=900 + i) for i in range(n)])
    d = _dev(imgs)
    out = torch.zeros((n, size * size // 4), dtype=torch.uint8, device="cuda")
    assert pkg.pvrtc_tune(2, -1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            assert pkg.encode_device(T.PVRTC2, d, size, size, 4, n_images=n, out=out, stream=s) is not None
        for _ in range(2):
            out.zero_()
            g.replay()
            s.synchronize()
            for i in range(n):
                assert out[i].cpu().numpy().tobytes() == T.oracle_encode(T.PVRTC2, imgs[i], size, size, 4, threads=8), i
    assert pkg.pvrtc_tune(1, -1)
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(pkg.BackendError):
            with torch.cuda.graph(g2, stream=s):
                pkg.encode_device(T.PVRTC2, d, size, size, 4, n_images=n, out=out, stream=s)
    torch.cuda.synchronize()


def test_const_colour_table_as_compiled_into_the_library(pkg):
    """the oracle and the product compile in the SAME dxtc_const_table.inc, so a corrupted table would
    pass every product-vs-oracle test.  Pin it without the oracle: (1) the file's 2 048 values hash to the pinned SHA-256;
    (2) the library's encoding of one-colour blocks that read every table row, in both channel widths, equals the bytes
    the compiled reference produced for them (tests/golden/solid_ramps_dxt1_*.bin), through the device entry point, the
    host drop-in, and the RGBA8 extension."""
    assert hashlib.sha256(T.const_table_bytes()).hexdigest() == T.CONST_TABLE_SHA256
    img = T.solid_ramp_image()
    h, w = img.shape[:2]
    rgba = np.concatenate([img, np.full((h, w, 1), 77, np.uint8)], axis=2)
    for fmt, swap in ((T.RGB, False), (T.BGR, True)):
        want = G.load_bin("solid_ramps_dxt1_%s.bin" % G.FMT_NAMES[fmt])
        assert _host(pkg.encode_device(T.DXT1, _dev(img), h, w, 3, swap_rb=swap)) == want
        assert _host(pkg.encode_device(T.DXT1, _dev(rgba), h, w, 4, swap_rb=swap)) == want
        assert pkg.compress_host(T.DXTC, fmt, img.reshape(-1), h, w) == want


# ---- "next" row 8f.1: decoders

def test_decoders_match_oracle_and_golden(pkg):
    import torch
    for c in G.load("decode_hashes.json"):
        img = T.s_mixed(c["h"], c["w"], T.comps_of(c["format"]), index=c["index"])
        comp = c["compressor"]
        codec = T.ETC1 if comp == T.ETC else (T.DXT1 if T.comps_of(c["format"]) == 3 else T.DXT5)
        swap = c["format"] in (T.BGR, T.BGRA)
        blocks = pkg.compress_device(comp, c["format"], _dev(img), c["h"], c["w"])
        assert hashlib.sha256(_host(blocks)).hexdigest() == c["blocks_sha256"]
        px = pkg.decode_device(codec, blocks.contiguous(), c["h"], c["w"], swap_rb=swap)
        assert hashlib.sha256(_host(px)).hexdigest() == c["pixels_sha256"]
    # arbitrary block words decoded by the compiled reference -> committed hashes (incl. ETC1 differential blocks that
    # leave the 5-bit range)
    for c in G.load("random_decode_hashes.json"):
        blocks = np.frombuffer(T.random_blocks(c["codec"], c["h"], c["w"], c["seed"]), np.uint8)
        px = pkg.decode_device(c["codec"], _dev(blocks), c["h"], c["w"], swap_rb=c["format"] in (T.BGR, T.BGRA))
        assert hashlib.sha256(_host(px)).hexdigest() == c["pixels_sha256"], c
    g = np.random.Generator(np.random.PCG64(11))
    for codec, comps in ((T.DXT1, 3), (T.DXT5, 4), (T.ETC1, 3)):
        for (h, w, pad) in [(64, 64, 0), (13, 7, 0), (9, 9, 5), (1024, 1024, 0)]:
            n = pkg.encoded_size(codec, h, w)
            blocks = g.integers(0, 256, size=n, dtype=np.uint8)
            # (ETC1: random words include differential blocks whose base + delta leaves 0..31 -- the reference's decoder is
            # fully defined there (Extend5Bit masks, ClampTo8Bits: etc_compressor.cc:198-273) and so is ours)
            want = T.oracle_decode(codec, blocks.tobytes(), h, w, pad=pad)
            got = pkg.decode_device(codec, _dev(blocks), h, w, padding_bytes_per_row=pad)
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy().reshape(-1), want), (codec, h, w, pad)


def test_pvrtc_decoder_matches_oracle(pkg):
    """PVRTC 2bpp decoder on the device (extension, parity unpinned: the reference has no PVRTC decoder) against the
    oracle's statement of the same rules: encoder output, random block words, a batch, and the 4096^2 size."""
    import torch
    rng = np.random.Generator(np.random.PCG64(31))
    for n in (8, 16, 32, 128, 256, 512):  # (256^2 and more take the tiled kernel; 256^2 is one tile wide: its ring wraps onto itself)
        cases = [T.oracle_encode(T.PVRTC2, T.GENERATORS[gen](n, n, 4, index=n + 1), n, n, 4) for gen in ("noise", "mixed", "flat")]
        cases.append(rng.integers(0, 256, size=n * n // 4, dtype=np.uint8).tobytes())
        for blocks in cases:
            dec = pkg.decode_device(T.PVRTC2, _dev(np.frombuffer(blocks, np.uint8)), n, n)
            assert _host(dec) == T.oracle_decode(T.PVRTC2, blocks, n, n).tobytes(), n
    for n, k in ((64, 5), (256, 3), (1024, 2)):
        words = rng.integers(0, 256, size=(k, n * n // 4), dtype=np.uint8)
        dec = pkg.decode_device(T.PVRTC2, _dev(words), n, n, n_images=k)
        torch.cuda.synchronize()
        for i in range(k):
            assert dec[i].cpu().numpy().tobytes() == T.oracle_decode(T.PVRTC2, words[i].tobytes(), n, n).tobytes(), (n, i)
    n = 4096
    img = T.s_smooth(n, n, 4, index=12)
    enc = pkg.encode_device(T.PVRTC2, _dev(img), n, n, 4)
    dec = pkg.decode_device(T.PVRTC2, enc.reshape(-1), n, n)
    want = T.oracle_decode(T.PVRTC2, _host(enc), n, n)
    assert hashlib.sha256(_host(dec)).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()
    # refusals: not square / not a power of two / row padding; the host-buffer Decompress stays `false` like the reference
    assert pkg.decode_device(T.PVRTC2, enc.reshape(-1), 4096, 2048) is None
    assert pkg.decode_device(T.PVRTC2, enc.reshape(-1), 24, 24) is None
    assert pkg.decode_device(T.PVRTC2, enc.reshape(-1), 64, 64, padding_bytes_per_row=4) is None
    # the host-buffer form of the extension (icamd_decompress itself answers false for PVRTC, like the reference)
    small = T.oracle_encode(T.PVRTC2, T.s_mixed(64, 64, 4, index=9), 64, 64, 4)
    assert pkg.pvrtc_decompress_host(small, 64) == T.oracle_decode(T.PVRTC2, small, 64, 64).tobytes()
    assert pkg.pvrtc_decompress_host(small, 32) is None


def test_pvrtc4_decoder_matches_oracle(pkg):
    """r05: PVRTC 4 bpp decoder on the device (the decoder of the 4 bpp extension encoder; parity unpinned twice over) against
    the oracle's statement of the same rules: the device encoder's output, random block words (half of them punch-through
    blocks, which the encoder never writes), batches through both kernels (block grids below / from 32 x 8), 4096^2, and
    the round trip encode -> decode on the device against encode -> decode in the oracle."""
    import torch
    rng = np.random.Generator(np.random.PCG64(37))
    for n in (8, 16, 32, 64, 128, 256, 512):  # (128^2 is the first size of the tiled kernel: one tile wide, its ring wraps onto itself)
        cases = [T.oracle_encode(T.PVRTC4, T.GENERATORS[gen](n, n, 4, index=n + 2), n, n, 4) for gen in ("noise", "mixed", "flat")]
        cases.append(rng.integers(0, 256, size=n * n // 2, dtype=np.uint8).tobytes())
        for blocks in cases:
            dec = pkg.decode_device(T.PVRTC4, _dev(np.frombuffer(blocks, np.uint8)), n, n)
            assert _host(dec) == T.oracle_decode(T.PVRTC4, blocks, n, n).tobytes(), n
    for n, k in ((64, 5), (128, 3), (1024, 2)):
        words = rng.integers(0, 256, size=(k, n * n // 2), dtype=np.uint8)
        dec = pkg.decode_device(T.PVRTC4, _dev(words), n, n, n_images=k)
        torch.cuda.synchronize()
        for i in range(k):
            assert dec[i].cpu().numpy().tobytes() == T.oracle_decode(T.PVRTC4, words[i].tobytes(), n, n).tobytes(), (n, i)
    n = 4096
    img = T.s_smooth(n, n, 4, index=13)
    enc = pkg.encode_device(T.PVRTC4, _dev(img), n, n, 4)
    dec = pkg.decode_device(T.PVRTC4, enc.reshape(-1), n, n)
    want = T.oracle_decode(T.PVRTC4, _host(enc), n, n)
    assert hashlib.sha256(_host(dec)).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()
    mse = float(((want.reshape(n, n, 4).astype(np.float64) - img.astype(np.float64)) ** 2).mean())
    assert 10.0 * np.log10(255.0 * 255.0 / mse) > 18.0  # (s_smooth carries 5 bits of noise and a random alpha: 20.6 dB)
    assert pkg.decode_device(T.PVRTC4, enc.reshape(-1), 4096, 2048) is None
    assert pkg.decode_device(T.PVRTC4, enc.reshape(-1), 24, 24) is None
    assert pkg.decode_device(T.PVRTC4, enc.reshape(-1), 64, 64, padding_bytes_per_row=4) is None


def test_pvrtc_call_sequences_host_api(pkg):
    # Regression: results must not depend on what ran before in the process (workspace / staging reuse).
    # A 128^2 image spans two workgroups of each PVRTC kernel, 64^2 and 8^2 only one.
    big = T.s_mixed(128, 128, 4, index=77)
    assert pkg.compress_host(T.DXTC, T.BGRA, big.reshape(-1), 128, 128) == T.oracle_compress(T.DXTC, T.BGRA, big, 128, 128)
    for n in (64, 8, 128, 16, 256, 128, 8, 512, 128):
        img = T.s_mixed(n, n, 4, index=n + 1)
        assert pkg.compress_host(T.PVRTC, T.RGBA, img.reshape(-1), n, n) == T.oracle_encode(T.PVRTC2, img, n, n, 4), n


def test_slab_sharding_of_one_image_on_device(pkg):
    # one large DXT/ETC image split into block-row slabs (what each rank of an N-GPU job encodes);
    # here the "ranks" run one after the other on the single GPU and their byte ranges must tile the whole output.
    import torch
    from image_compression_amd import sharding
    h, w, pad = 1022, 515, 3
    for codec, comps, bb in ((T.DXT1, 3, 8), (T.DXT5, 4, 16), (T.ETC1, 3, 8)):
        img = T.s_mixed(h, w, comps, index=21)
        src = T.with_row_padding(img, pad)
        stride = w * comps + pad
        d = _dev(src)
        whole = _host(pkg.encode_device(codec, d, h, w, comps, row_stride_bytes=stride))
        assert whole == T.oracle_encode(codec, src, h, w, comps, stride=stride, threads=8)
        for world in (2, 3, 8):
            parts = []
            for rank in range(world):
                g = sharding.slab_geometry(h, w, comps, stride, bb, world, rank)
                slab = d[g["src_offset_bytes"]:].contiguous()
                out = pkg.encode_device(codec, slab, g["pixel_rows"], w, comps, row_stride_bytes=stride,
                                        grid_height=g["block_rows"] * 4, grid_width=w)
                torch.cuda.synchronize()
                parts.append(out.cpu().numpy().tobytes())
                assert len(parts[-1]) == g["dst_bytes"]
            assert b"".join(parts) == whole, (codec, world)


# ---- "next" rows 8f.2-4: Pad / Downsample / DXT1->ETC1 transcode kernels

def test_blockops_match_oracle(pkg):
    # every ETC1 strategy: each has its own border kernel (icamd_pad_etc1_border[_split_h|_split_v|_heuristic]_kernel); the three
    # padded sizes give a rows-and-columns, a columns-only and a rows-only border, i.e. both halves of its index mapping
    for compressor, fmt, strategy in [(T.DXTC, T.RGB, 2), (T.DXTC, T.BGR, 2), (T.DXTC, T.RGBA, 2), (T.ETC, T.RGB, 0),
                                      (T.ETC, T.RGB, 1), (T.ETC, T.RGB, 2), (T.ETC, T.RGB, 3)]:
        for (h, w) in [(256, 512), (32, 48), (13, 7), (64, 8), (8, 64), (4, 4), (2, 2), (1, 4), (3, 8)]:
            img = T.s_mixed(h, w, T.comps_of(fmt), index=h + w)
            blocks = T.oracle_compress(compressor, fmt, img, h, w, 0, strategy)
            ch, cw = 4 * ((h + 3) // 4), 4 * ((w + 3) // 4)
            for (ph, pw) in [(h + 9, w + 5), (ch, cw + 8), (ch + 4, cw)]:
                assert pkg.pad_host(compressor, fmt, blocks, ch, cw, ph, pw, strategy) == \
                    T.oracle_pad(compressor, fmt, blocks, ch, cw, ph, pw, strategy), (compressor, fmt, h, w, ph, pw)
            assert pkg.downsample_host(compressor, fmt, blocks, h, w, strategy) == \
                T.oracle_downsample(compressor, fmt, blocks, h, w, strategy), (compressor, fmt, h, w)
    # a mip chain entirely in the compressed domain: 1024 -> 512 -> ... -> 4
    img = T.s_smooth(1024, 1024, 3, index=2)
    cur = T.oracle_compress(T.DXTC, T.RGB, img, 1024, 1024)
    n = 1024
    while n > 4:
        nxt = pkg.downsample_host(T.DXTC, T.RGB, cur, n, n)
        assert nxt == T.oracle_downsample(T.DXTC, T.RGB, cur, n, n), n
        cur, n = nxt, n // 2
    g = np.random.Generator(np.random.PCG64(3))
    raw = g.integers(0, 256, size=8 * 65536, dtype=np.uint8).tobytes()
    assert pkg.transcode_dxt1_to_etc1_host(raw) == T.oracle_transcode(raw)
    enc = T.oracle_compress(T.DXTC, T.RGB, T.s_mixed(512, 512, 3, index=4), 512, 512)
    assert pkg.transcode_dxt1_to_etc1_host(enc) == T.oracle_transcode(enc)


def test_etc1_small_launches_four_lanes_per_block_match_oracle(pkg):
    """r05: kSmallerError launches of at most 36 864 blocks run four lanes per block (icamd_etc1_rgb888/rgba8_quad_kernel).  The
    same inputs through that form (threshold forced up), through the one-lane form (threshold 0) and with the shipped
    threshold -- ragged sizes, RGBA, a batch, one-colour blocks inside searching waves, row padding -- all against the oracle;
    the threshold is read once per process, hence the child processes."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
bad = 0
for (h, w, comps, gen) in [(256, 256, 3, "noise"), (256, 256, 4, "mixed"), (61, 59, 3, "mixed"), (5, 3, 3, "noise"), (1, 1, 3, "flat"),
                           (128, 512, 3, "flat"), (512, 512, 3, "smooth"), (64, 1024, 4, "smooth")]:
    img = T.GENERATORS[gen](h, w, comps, index=h + w)
    out = pkg.encode_device(T.ETC1, torch.from_numpy(img).cuda(), h, w, comps)
    torch.cuda.synchronize()
    bad += out.cpu().numpy().tobytes() != T.oracle_encode(T.ETC1, img, h, w, comps)
imgs = np.stack([T.s_mixed(128, 128, 3, index=50 + i) for i in range(5)])
imgs[2, 32:64, :, :] = imgs[2, 0, 0, :]          # a band of one-colour blocks inside searching waves
out = pkg.encode_device(T.ETC1, torch.from_numpy(imgs).cuda(), 128, 128, 3, n_images=5)
torch.cuda.synchronize()
for i in range(5):
    bad += out[i].cpu().numpy().tobytes() != T.oracle_encode(T.ETC1, imgs[i], 128, 128, 3)
pad = T.with_row_padding(T.s_noise(37, 41, 3, index=4), 7)
got = pkg.compress_host(T.ETC, T.RGB, pad, 37, 41, padding_bytes_per_row=7)
bad += got != T.oracle_compress(T.ETC, T.RGB, pad, 37, 41, 7)
print("KERNEL", pkg.lib().icamd_version().decode())
print("BAD", bad)
""" % (T.ROOT, T.ROOT)
    for setting in ("0", str(1 << 40), None):
        env = dict(os.environ)
        env.pop("ICAMD_ETC1_QUAD_MAX_BLOCKS", None)
        if setting is not None:
            env["ICAMD_ETC1_QUAD_MAX_BLOCKS"] = setting
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("BAD 0"), (setting, r.stdout[-300:], r.stderr[-800:])


def test_etc1_pad_quad_lanes_and_one_lane_forms_match_oracle(pkg):
    """r05: the kSmallerError Pad runs as ONE launch whose first workgroups are the pad blocks with FOUR lanes each
    (encode_etc1_block_quad).  Arbitrary ETC1 block words (saturated bases, clamping codewords, partition ties), batched
    with an image stride, rows-only / columns-only / both borders; and the r04 form (one lane per pad block, two launches:
    ICAMD_PAD_BORDER_QUAD=0, read once per process -> a child process) on the same inputs."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
g = np.random.Generator(np.random.PCG64(41))
bad = 0
for trial in range(6):
    h, w = (64, 128) if trial < 4 else (8, 4)
    raw = g.integers(0, 256, size=(h // 4) * (w // 4) * 8, dtype=np.uint8)
    if trial in (2, 3): raw = (raw & (0xc0 if trial == 2 else 0x81)).astype(np.uint8)
    for (ph, pw) in [(h + 8, w + 12), (h, w + 4), (h + 4, w)]:
        got = pkg.pad_host(T.ETC, T.RGB, raw.tobytes(), h, w, ph, pw, 2)
        bad += got != T.oracle_pad(T.ETC, T.RGB, raw.tobytes(), h, w, ph, pw, 2)
n, h, w, ph, pw = 3, 32, 64, 40, 72
grids = [g.integers(0, 256, size=(h // 4) * (w // 4) * 8, dtype=np.uint8) for _ in range(n)]
src = torch.zeros((n, grids[0].size + 40), dtype=torch.uint8, device="cuda")
for i, q in enumerate(grids): src[i, :q.size] = torch.from_numpy(q).cuda()
out = pkg.pad_batch_device(T.ETC, T.RGB, src, h, w, ph, pw, etc_strategy=2)  # (image stride = the tensor's row: 40 spare bytes)
torch.cuda.synchronize()
for i, q in enumerate(grids):
    bad += out[i].cpu().numpy().tobytes() != T.oracle_pad(T.ETC, T.RGB, q.tobytes(), h, w, ph, pw, 2)
# the same switch covers Downsample's small grids (icamd_downsample_etc1_quad_kernel): a mip chain 256 -> 4 of arbitrary words,
# every level against the oracle's
cur, s = g.integers(0, 256, size=(256 // 4) ** 2 * 8, dtype=np.uint8).tobytes(), 256
while s > 4:
    nxt = pkg.downsample_host(T.ETC, T.RGB, cur, s, s, 2)
    bad += nxt != T.oracle_downsample(T.ETC, T.RGB, cur, s, s, 2)
    cur, s = T.oracle_downsample(T.ETC, T.RGB, cur, s, s, 2), s // 2
print("BAD", bad)
""" % (T.ROOT, T.ROOT)
    for quad in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ICAMD_PAD_BORDER_QUAD=quad), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("BAD 0"), (quad, r.stdout[-300:], r.stderr[-600:])


def test_batched_block_operations_match_oracle(pkg):
    """r05: icamd_pad_batch_device / icamd_copy_subimage_batch_device / icamd_create_solid_batch_device -- n equally shaped
    grids per launch (padded image strides on the source side), every image against the oracle's per-image result; and the
    argument checks they share with icamd_downsample_batch_device (a refused geometry is refused for ANY image count)."""
    import ctypes
    import torch
    n = 5
    for compressor, fmt, strategy in [(T.DXTC, T.RGB, 2), (T.DXTC, T.RGBA, 2), (T.ETC, T.RGB, 0), (T.ETC, T.RGB, 1), (T.ETC, T.RGB, 2),
                                      (T.ETC, T.RGB, 3)]:
        for (h, w) in [(64, 96), (12, 8), (4, 4)]:
            imgs = [T.s_mixed(h, w, T.comps_of(fmt), index=31 * i + h) for i in range(n)]
            grids = [T.oracle_compress(compressor, fmt, im, h, w, 0, strategy) for im in imgs]
            per = len(grids[0])
            src = torch.zeros((n, per + 24), dtype=torch.uint8, device="cuda")  # 24 spare bytes per image: a real stride
            for i, gbytes in enumerate(grids):
                src[i, :per] = _dev(np.frombuffer(gbytes, np.uint8))
            for (ph, pw) in [(h + 9, w + 5), (h, w + 8), (h + 4, w)]:
                out = pkg.pad_batch_device(compressor, fmt, src, h, w, ph, pw, etc_strategy=strategy)
                torch.cuda.synchronize()
                for i in range(n):
                    assert out[i].cpu().numpy().tobytes() == T.oracle_pad(compressor, fmt, grids[i], h, w, ph, pw, strategy), \
                        (compressor, fmt, strategy, h, w, ph, pw, i)
            if h >= 8 and w >= 16:
                out = pkg.copy_subimage_batch_device(compressor, fmt, src, h, w, 4, 4, h - 4, w - 8)
                torch.cuda.synchronize()
                for i in range(n):
                    assert out[i].cpu().numpy().tobytes() == T.oracle_copy_subimage(compressor, fmt, grids[i], h, w, 4, 4, h - 4, w - 8)
        colors = [((10 * i + 3) & 255, (255 - 7 * i) & 255, 40 + i, (200 + i) & 255) for i in range(70)]  # more images than one fill launch holds
        out = pkg.create_solid_batch_device(compressor, fmt, 40, 52, colors)
        torch.cuda.synchronize()
        for i, c in enumerate(colors):
            assert out[i].cpu().numpy().tobytes() == T.oracle_create_solid(compressor, fmt, 40, 52, list(c)[:T.comps_of(fmt)] + [0] * (4 - T.comps_of(fmt))), i
    # argument checks
    L = pkg.lib()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p(buf.data_ptr())
    for count in (0, 1, 3):  # an odd block grid is refused for any count, also for none
        assert L.icamd_downsample_batch_device(T.DXTC, 2, T.RGB, 12, 16, count, p, 4096, p, 4096, 8, None) == pkg.FALSE
        assert L.icamd_pad_batch_device(T.DXTC, 2, T.RGB, 16, 16, count, p, 4096, 8, 24, p, 4096, 96, None) == pkg.FALSE  # shrinks
        assert L.icamd_copy_subimage_batch_device(T.DXTC, T.RGB, 16, 16, count, p, 4096, 2, 0, 8, 8, p, 4096, 32, None) == pkg.FALSE
    assert L.icamd_downsample_batch_device(T.DXTC, 2, T.RGB, 16, 16, 0, p, 4096, p, 4096, 32, None) == pkg.OK
    assert L.icamd_downsample_batch_device(T.DXTC, 2, T.RGB, 16, 16, 1, p, 64, p, 4096, 32, None) < 0   # stride < one image
    assert L.icamd_pad_batch_device(T.DXTC, 2, T.RGB, 16, 16, 2, p, 64, 24, 24, p, 4096, 288, None) < 0
    assert L.icamd_create_solid_batch_device(T.PVRTC, T.RGBA, 8, 8, 1, (ctypes.c_uint8 * 4)(1, 2, 3, 4), p, 0, 16, None) == pkg.FALSE
    with pytest.raises(ValueError):
        pkg.downsample_device(T.DXTC, T.RGB, buf[:64].view(1, -1), 16, 16)  # 64 bytes: a 16 x 16 DXT1 grid has 128
    torch.cuda.synchronize()


def test_mip_chain_in_containers(pkg):
    """8(f) row 4 tail: a device-encoded texture and its compressed-domain mip chain framed as KTX / DDS / PVR; the levels
    read back out of the file image are the oracle's levels (the framing itself is pinned in tests/test_containers.py)."""
    import struct
    for codec, compressor, fmt, container in [(0, T.DXTC, T.RGB, pkg.CONTAINER_DDS), (1, T.DXTC, T.RGBA, pkg.CONTAINER_KTX),
                                             (2, T.ETC, T.RGB, pkg.CONTAINER_PVR)]:
        n = 256
        img = T.s_smooth(n, n, T.comps_of(fmt), index=11)
        levels = [pkg.compress_host(compressor, fmt, img, n, n)]
        want = [T.oracle_compress(compressor, fmt, img, n, n)]
        while n > 4:  # Downsample needs at least a 2 x 2 block grid (compressor4x4_helper.h:594-636)
            levels.append(pkg.downsample_host(compressor, fmt, levels[-1], n, n))
            want.append(T.oracle_downsample(compressor, fmt, want[-1], n, n))
            n //= 2
        blob = pkg.container_write(container, codec, 256, 256, levels)
        assert blob is not None and len(blob) == pkg.container_size(container, codec, 256, 256, len(levels))
        off = {pkg.CONTAINER_DDS: 128, pkg.CONTAINER_KTX: 64, pkg.CONTAINER_PVR: 52}[container]
        for lvl in want:
            if container == pkg.CONTAINER_KTX:
                assert struct.unpack_from("<I", blob, off)[0] == len(lvl)
                off += 4
            assert blob[off:off + len(lvl)] == lvl
            off += len(lvl)
        assert off == len(blob)
    # PKM: one ETC1 level of a ragged size
    img = T.s_mixed(61, 59, 3, index=5)
    blob = pkg.container_write(pkg.CONTAINER_PKM, 2, 61, 59, [pkg.compress_host(T.ETC, T.RGB, img, 61, 59)])
    assert blob[:6] == b"PKM 10" and blob[16:] == T.oracle_compress(T.ETC, T.RGB, img, 61, 59)
    # PVR: a PVRTC texture (Z-order block stream as the encoder emits it)
    img = T.s_smooth(64, 64, 4, index=6)
    blob = pkg.container_write(pkg.CONTAINER_PVR, 3, 64, 64, [pkg.compress_host(T.PVRTC, T.RGBA, img, 64, 64)])
    assert blob[52:] == T.oracle_compress(T.PVRTC, T.RGBA, img, 64, 64)


def test_single_process_multi_device_batch(pkg):
    # icamd_compress_batch: one worker thread, stream and staging set per device-list entry.  The GPU box has one
    # GPU, so device 0 is listed several times, which exercises the same concurrency as several devices would.
    n, h, w = 13, 96, 160
    for compressor, fmt, codec in ((T.DXTC, T.RGB, T.DXT1), (T.DXTC, T.BGRA, T.DXT5), (T.ETC, T.RGB, T.ETC1)):
        comps = T.comps_of(fmt)
        imgs = [T.s_mixed(h, w, comps, index=i) for i in range(n)]
        want = [T.oracle_compress(compressor, fmt, im, h, w) for im in imgs]
        for devices in ([0], [0, 0, 0], [0] * 8):
            assert pkg.compress_batch_host(compressor, fmt, imgs, h, w, devices) == want
    imgs = [T.s_mixed(64, 64, 4, index=i) for i in range(6)]
    assert pkg.compress_batch_host(T.PVRTC, T.RGBA, imgs, 64, 64, [0, 0, 0]) == \
        [T.oracle_encode(T.PVRTC2, im, 64, 64, 4) for im in imgs]
    assert pkg.compress_batch_host(T.ETC, T.RGBA, imgs, 64, 64, [0, 0]) == [None] * 6  # reference: false


# ---- seeded random soak: many small images of random geometry / content / format against the oracle

def test_random_soak_matches_oracle(pkg):
    n = 0
    for (codec, comps, swap, strategy, h, w, pad, img) in T.soak_cases(0x50AC, 260, 40):
        src = T.with_row_padding(img, pad)
        stride = w * comps + pad
        want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, stride=stride)
        out = pkg.encode_device(codec, _dev(src), h, w, comps, swap_rb=bool(swap), etc_strategy=strategy,
                                row_stride_bytes=stride)
        assert _host(out) == want, (codec, comps, swap, strategy, h, w, pad)
        n += 1
    assert n == 300


# ---- one PVRTC texture sharded by Z-order range: regions on one device == the whole image

def test_pvrtc_region_sharding_of_one_image(pkg):
    from image_compression_amd import sharding
    rng = np.random.Generator(np.random.PCG64(0x9E))
    for size, worlds in ((8, (1, 2)), (32, (2, 4, 8, 32)), (256, (2, 4, 8)), (1024, (8, 64))):
        img = T.soak_image(rng, size, size, 4)
        want = T.oracle_encode(T.PVRTC2, img, size, size, 4)
        assert _host(pkg.encode_device(T.PVRTC2, _dev(img), size, size, 4)) == want
        bw, bh = size // 8, size // 4
        for world in worlds:
            got = bytearray(len(want))
            for rank in range(world):
                g = sharding.pvrtc_region(size, world, rank)
                # the rank's copy of the image: only its rectangle, a one-block ring (toroidal) and pixel (0, 0) are
                # real; everything else is garbage and must not influence the result
                keep = np.zeros((size, size), bool)
                ys = [(g["block_y0"] - 1 + j) % bh for j in range(g["blocks_h"] + 2)]
                xs = [(g["block_x0"] - 1 + i) % bw for i in range(g["blocks_w"] + 2)]
                for by in ys:
                    for bx in xs:
                        keep[by * 4:by * 4 + 4, bx * 8:bx * 8 + 8] = True
                keep[0, 0] = True
                local = np.where(keep[..., None], img, rng.integers(0, 256, img.shape, dtype=np.uint8))
                out = pkg.pvrtc_encode_region_device(_dev(local), size, g["first_block"], g["n_blocks"])
                got[g["dst_offset_bytes"]:g["dst_offset_bytes"] + g["dst_bytes"]] = _host(out)
            assert bytes(got) == want, (size, world)
    # argument checks: misaligned / non power-of-two ranges are errors, bad sizes return false
    src = _dev(np.zeros((32, 32, 4), np.uint8))
    with pytest.raises(pkg.BackendError):
        pkg.pvrtc_encode_region_device(src, 32, 3, 4)
    with pytest.raises(pkg.BackendError):
        pkg.pvrtc_encode_region_device(src, 32, 0, 6)
    assert pkg.pvrtc_encode_region_device(src, 24, 0, 2) is None


def test_concurrent_host_api_calls_from_several_threads(pkg):
    """The reference has no mutable global state (threading): concurrent Compress calls must stay safe.
    Four host threads hammer the host-buffer entry points (per-thread streams / staging / PVRTC workspace)."""
    import threading
    work = [(T.DXTC, T.RGB, 3, 2), (T.DXTC, T.RGBA, 4, 2), (T.ETC, T.RGB, 3, 2), (T.PVRTC, T.RGBA, 4, 2),
            (T.ETC, T.RGB, 3, 3), (T.DXTC, T.BGR, 3, 2)]
    cases = []
    for i in range(24):
        compressor, fmt, comps, strategy = work[i % len(work)]
        n = 64 if compressor == T.PVRTC else 52 + 4 * (i % 5)
        img = T.s_mixed(n, n, comps, index=100 + i)
        cases.append((compressor, fmt, strategy, n, img, T.oracle_compress(compressor, fmt, img.reshape(-1), n, n,
                                                                         strategy=strategy)))
    failures = []

    def run(tid):
        for rep in range(3):
            for j, (compressor, fmt, strategy, n, img, want) in enumerate(cases):
                if (j + tid) % 4 and rep:  # different interleavings per thread
                    continue
                got = pkg.compress_host(compressor, fmt, img.reshape(-1), n, n, etc_strategy=strategy)
                if got != want:
                    failures.append((tid, j))
    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures[:5]


def test_device_entry_points_are_graph_capturable(pkg):
    """The device entry points only enqueue kernels, so a loop over many small textures can be captured once into a
    HIP graph and replayed (launch-bound regime).  PVRTC needs scratch memory between its two kernels: under capture
    the caller provides it (icamd_pvrtc2_set_workspace), one buffer per graph."""
    import torch
    n, size = 12, 64
    imgs = np.stack([T.s_mixed(size, size, 4, index=300 + i) for i in range(n)])
    src = _dev(imgs)
    for codec in (T.DXT1, T.DXT5, T.ETC1, T.PVRTC2):
        per = pkg.encoded_size(codec, size, size)
        out = torch.zeros((n, per), dtype=torch.uint8, device="cuda")
        s = torch.cuda.Stream()
        ws = torch.empty(pkg.pvrtc_workspace_size(size, 1), dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(s):
            if codec == T.PVRTC2:
                pkg.pvrtc_set_workspace(ws)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for i in range(n):
                        pkg.encode_device(codec, src[i], size, size, 4, out=out[i:i + 1], stream=s)
            finally:
                pkg.pvrtc_set_workspace(None)
            g.replay()
            s.synchronize()
        for i in range(n):
            assert out[i].cpu().numpy().tobytes() == T.oracle_encode(codec, imgs[i], size, size, 4), (codec, i)


def test_pvrtc_graphs_keep_their_own_workspace(pkg):
    """Two PVRTC graphs of different sizes, each with its own caller-provided workspace, stay valid across later
    (larger) eager calls on the same thread and can be replayed concurrently on different streams; without a
    caller workspace a PVRTC call under capture is refused instead of baking the library's buffer into the graph."""
    import torch
    sizes = (64, 256)
    imgs = [T.s_mixed(sz, sz, 4, index=700 + sz) for sz in sizes]
    srcs = [_dev(im) for im in imgs]
    outs = [torch.zeros((1, pkg.encoded_size(T.PVRTC2, sz, sz)), dtype=torch.uint8, device="cuda") for sz in sizes]
    wss = [torch.empty(pkg.pvrtc_workspace_size(sz, 1), dtype=torch.uint8, device="cuda") for sz in sizes]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    graphs = []
    for i, sz in enumerate(sizes):
        with torch.cuda.stream(streams[i]):
            pkg.pvrtc_set_workspace(wss[i])
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i]):
                    pkg.encode_device(T.PVRTC2, srcs[i], sz, sz, 4, out=outs[i], stream=streams[i])
            finally:
                pkg.pvrtc_set_workspace(None)
            graphs.append(g)
    # an eager, larger call in between (grows / replaces the library's own scratch buffer)
    big = T.s_noise(1024, 1024, 4, index=5)
    assert _host(pkg.encode_device(T.PVRTC2, _dev(big), 1024, 1024, 4)) == T.oracle_encode(T.PVRTC2, big, 1024, 1024, 4)
    for rep in range(3):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
        torch.cuda.synchronize()
        for i, sz in enumerate(sizes):
            assert outs[i].cpu().numpy().tobytes() == T.oracle_encode(T.PVRTC2, imgs[i], sz, sz, 4), (rep, sz)
    # too small a workspace is an argument error, not an overrun
    pkg.pvrtc_set_workspace(wss[0])
    try:
        with pytest.raises(pkg.BackendError):
            pkg.encode_device(T.PVRTC2, srcs[1], sizes[1], sizes[1], 4)
    finally:
        pkg.pvrtc_set_workspace(None)
    torch.cuda.synchronize()


# ---- BASELINE config 4 at its stated per-GPU size, and the N > 1 paths

def test_config_c4_per_gpu_share_128_textures(pkg):
    """Config 4 = 1024 textures of 1024^2, ETC1 kSmallerError, sharded over 8 GPUs by texture_range: rank 3's share
    (textures 384..511) encoded by ONE launch, every texture hash-checked against the oracle."""
    import os
    import torch
    from image_compression_amd import sharding
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    begin, end = sharding.texture_range(1024, 8, 3)
    assert (begin, end) == (384, 512)
    n, size = end - begin, 1024
    g = torch.Generator(device="cuda")
    g.manual_seed(0x1234ABCD + 3)
    src = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, device="cuda", generator=g)
    # a quarter of the textures: mid-tones only (the unclamped shortcut fires), smooth ramps, flat saturated tiles
    src[1::4] = (src[1::4] >> 2) + 96
    ramp = (torch.arange(size, device="cuda").view(1, size, 1, 1) // 4).to(torch.uint8)
    src[2::4] = (src[2::4] >> 5) + ramp
    src[3::4] = (src[3::4, ::16, ::16] >> 7).mul(255).repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)
    out = pkg.encode_device(T.ETC1, src, size, size, 3, etc_strategy=T.SMALLER_ERROR, n_images=n)
    torch.cuda.synchronize()
    host_src, host_out = src.cpu().numpy(), out.cpu().numpy()
    for i in range(n):
        want = T.oracle_encode(T.ETC1, host_src[i], size, size, 3, 0, T.SMALLER_ERROR, threads=cores)
        assert hashlib.sha256(host_out[i].tobytes()).hexdigest() == hashlib.sha256(want).hexdigest(), begin + i


def _run_bench(args, timeout=600):
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(lines[-1])


def test_bench_watchdog_prints_the_line_when_the_extra_legs_overrun(pkg):
    """r05: after the contract's K timed steps nothing may cost the run its line.  With a 2 s watchdog the default command is
    still inside its extra legs when it fires: ONE line, exit code 0, the headline fields and the roofline present, marked;
    and over two gloo ranks every rank leaves (the launcher would otherwise report the survivors' time-outs)."""
    d = _run_bench(["--steps", "3", "--warmup", "1", "--watchdog-seconds", "2", "--precondition-seconds", "0"], timeout=300)
    assert "watchdog" in d and d["value"] > 0 and d["n_gpus"] == 1 and d["roofline"]["frac"] > 0 and d["config"]["preset"] == "c2"
    assert "slab" not in d  # (the last leg cannot have finished within 2 s)
    d = _run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--watchdog-seconds", "2",
                    "--precondition-seconds", "0"], timeout=300)
    assert "watchdog" in d and d["n_gpus"] == 2 and d["value"] > 0


def test_bench_bare_command_self_launches_ranks(pkg):
    """`python bench.py --gpus 2` outside torch.distributed.run starts its own two ranks and times both regions
    (encode only; encode -> gather on rank 0, overlapped).  On a 1-GPU box the ranks share the GPU over gloo with the
    gather staged through the host (debug backend); with >= 2 GPUs this is the real RCCL path."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    for extra in (["--workload", "dxt1_rgba8", "--size", "1024", "--batch", "3"], ["--config", "c4"]):
        if extra == ["--config", "c4"] and backend == "gloo":
            extra = ["--workload", "etc1_rgb888", "--size", "256", "--batch", "5"]
        d = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", backend, "--no-cpu-baseline"] + extra)
        assert d["n_gpus"] == 2 and d["steps"] == 4 and d["parity"].startswith("bit-exact")
        assert d["value"] > 0 and d["value_with_gather"] > 0 and d["gather_ms"] > 0 and d["rank0_copy_matches"] is True
        # (no upper bound of value_with_gather against value: 4 steps of two different timed regions are not
        # comparable to better than the clock / warm-up noise; correctness of the gathered bytes is what is asserted)


def test_bench_config_presets(pkg):
    for cfg, codec, size in (("c3", "dxt5_rgba8", 8192), ("c5", "pvrtc2_rgba8", 4096)):
        d = _run_bench(["--config", cfg, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-host-api",
                        "--sustained-seconds", "0.5"])
        assert d["config"]["codec"] == codec and d["config"]["texture"] == [size, size] and d["config"]["preset"] == cfg
        assert d["parity"].startswith("bit-exact") and d["roofline"]["bound"] == "valu"


def test_bench_default_line_carries_every_baseline_config_and_the_slab_legs(pkg):
    """The line the driver records (bench.py with no --config): c2 headline + `configs` {c3, c4, c5} + `slab` (one 4096^2 /
    8192^2 / 16384^2 image in block-row slabs, gathered into rank 0's final buffer), each with its own parity; and the
    slab mode as a headline line, over the RCCL code path with one forced rank."""
    d = _run_bench(["--steps", "3", "--warmup", "1", "--extra-steps", "2", "--no-sustained", "--no-single-image", "--no-host-api",
                    "--no-cpu-baseline", "--precondition-seconds", "0"], timeout=900)
    assert d["config"]["preset"] == "c2" and d["parity"].startswith("bit-exact")
    assert sorted(d["configs"]) == ["c3", "c4", "c5", "c5_4bpp", "c5_8192"]
    for name, leg in d["configs"].items():
        assert leg["parity"].startswith("bit-exact") and leg["value"] > 0 and 0 < leg["roofline"]["frac"] < 1, (name, leg)
    assert d["configs"]["c4"]["textures_per_gpu_per_step"] == 1024 and d["configs"]["c4"]["etc_strategy"] == 2
    assert d["configs"]["c5"]["roofline"]["traffic"] and d["roofline"]["traffic"]
    # r05: `traffic` is measured in the run itself (two rocprofv3 --pmc child passes per launch shape) and agrees with the
    # committed profile of the same launch shape; algorithmic bytes <= traffic < 1.1 x algorithmic for the one-pass kernels
    for name, roof, algo in [("c2", d["roofline"], 16 * 4096 * 4096 * 4.5), ("c5", d["configs"]["c5"]["roofline"], 16 * 4096 * 4096 * 4.25),
                             ("c5_4bpp", d["configs"]["c5_4bpp"]["roofline"], 16 * 4096 * 4096 * 4.5),
                             ("c5_8192", d["configs"]["c5_8192"]["roofline"], 4 * 8192 * 8192 * 4.25)]:  # r06: the halo form
        assert roof["traffic_source"].startswith("measured in this run"), (name, roof["traffic_source"])
        assert algo <= roof["traffic"] < 1.1 * algo, (name, roof["traffic"], algo)
        if roof["traffic_committed_profile"]:
            assert abs(roof["traffic"] / roof["traffic_committed_profile"] - 1) < 0.02, (name, roof)
    for name, leg in d["slab"].items():
        assert leg["parity"].startswith("bit-exact") and leg["value"] > 0 and leg["value_with_gather"] > 0, (name, leg)
        assert leg["distinct_images_rotated"] >= 5 and leg["distinct_source_MiB_per_rank"] >= 320, (name, leg)
    for name, leg in d["configs"].items():  # r05: every leg says at which clock and VALU issue fraction it ran
        assert leg["roofline"]["effective_clock_MHz"] > 500, (name, leg["roofline"])
        # (r05: from SQ_INSTS_VALU counted in the run; r06: clock and kernel time of the SAME launches -- a fraction above 1 is
        # not a measurement)
        assert 0 < leg["roofline"]["valu_frac"] <= 1.0, (name, leg["roofline"])
        assert leg["roofline"]["clock_window_kernel_ms"] > 0
        assert leg["roofline"]["valu_profile"].startswith("measured in this run"), (name, leg["roofline"])
    other = d["configs"]["c4"]["other_contents"]
    assert sorted(other) == ["flat", "smooth"] and all(v["parity"].startswith("bit-exact") and 0 < v["valu_frac"] <= 1.0 for v in other.values())
    assert d["link_probe"] is None and "value_with_gather_ceiling" in d["scaling_headline"]
    d = _run_bench(["--gpus", "1", "--force-distributed", "--backend", "nccl", "--shard", "slab", "--workload", "dxt5_rgba8",
                    "--size", "8192", "--steps", "3"])
    assert d["scaling"] == "strong" and d["parity"].startswith("bit-exact") and d["value_with_gather"] > 0
    assert d["config"]["world_size"] == 1 and d["slab_block_rows"] == [2048]
    # two ranks on this box's one GPU over gloo (the gather staged through the host): unequal slabs, batched isend / irecv
    import torch
    if torch.cuda.device_count() < 2:
        d = _run_bench(["--gpus", "2", "--backend", "gloo", "--shard", "slab", "--workload", "dxt1_rgb888", "--size", "2052", "--steps", "3"])
        assert d["n_gpus"] == 2 and d["slab_block_rows"] == [256, 257] and d["parity"].startswith("bit-exact") and d["value_with_gather"] > 0


def test_bench_rehearsal_of_the_drivers_eight_rank_command(pkg):
    """The driver's first 8-GPU SCALE run is one shot, so its exact command is rehearsed here: `bench.py --gpus 8` (default
    line: c2 + configs {c3, c4, c5} + slab legs, every leg with its gather).  With fewer than 8 GPUs the eight ranks share
    the box's GPU(s) over gloo (collectives staged through the host): every split, count, buffer shape, barrier and parity
    check of the 8-rank run executes; only the transport differs from RCCL."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    d = _run_bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--extra-steps", "2", "--precondition-seconds", "0",
                    "--backend", backend], timeout=1500)
    assert d["n_gpus"] == 8 and d["config"]["world_size"] == 8 and d["scaling"] == "weak"
    assert d["parity"].startswith("bit-exact") and d["config"]["textures_per_step_all_gpus"] == 8 * 16
    assert d["value"] > 0 and d["value_with_gather"] > 0 and d["rank0_copy_matches"] is True
    assert d["gather_ranks"] == 8 and d["gather_bound_GBps"] > 0 and d["value_with_gather_ceiling"] > 0
    assert d["gather_bytes_into_rank0_per_step"] == 7 * 16 * 4096 * 4096 // 2
    lp = d["link_probe"]
    assert lp["peers"] == 7 and len(lp["per_peer_alone_GBps"]) == 7 and lp["payload_intact"] and lp["xgmi_links_into_rank0"] == 7
    assert "value_with_gather_ceiling" in d["scaling_headline"]
    assert sorted(d["configs"]) == ["c3", "c4", "c5", "c5_4bpp", "c5_8192"]
    for name, leg in d["configs"].items():
        assert leg["parity"].startswith("bit-exact") and leg["value"] > 0 and leg["value_with_gather"] > 0, (name, leg)
        assert leg["rank0_copy_matches"] is True and leg["gather_ranks"] == 8 and leg["gather_bound_GBps"] > 0, (name, leg)
    c4 = d["configs"]["c4"]  # 1 024 textures over 8 ranks by texture_range: 128 each, strong scaling
    assert c4["textures_per_gpu_per_step"] == 128 and c4["scaling"] == "strong"
    assert sorted(c4["other_contents"]) == ["flat", "smooth"]
    for content, leg in c4["other_contents"].items():
        assert leg["parity"].startswith("bit-exact") and leg["value"] > 0, (content, leg)
    assert d["configs"]["c3"]["scaling"] == "weak" and d["configs"]["c5"]["textures_per_gpu_per_step"] == 16
    for name, size in (("c2_one_4096", 4096), ("c3_one_8192", 8192), ("one_16384", 16384)):
        leg = d["slab"][name]
        assert leg["parity"].startswith("bit-exact") and leg["value"] > 0 and leg["value_with_gather"] > 0, (name, leg)
        assert leg["slab_block_rows"] == [size // 4 // 8] * 8 and leg["scaling"] == "strong" and leg["gather_ranks"] == 8
        # every rank's slabs are read from HBM, not from the 256 MiB Infinity Cache
        assert leg["distinct_images_rotated"] >= 5 and leg["distinct_source_MiB_per_rank"] >= 320, (name, leg)


def test_rccl_code_path_executes_with_a_single_rank(pkg):
    """The pool's boxes have one GPU, so the >= 2-GPU tests above never run there.  RCCL itself works with a world of one
    rank: this runs the same worker (process-group init on the device, dist.gather / all_gather_into_tensor of device
    tensors through sharding.gather_to_root / gather_output with the world-size-1 shortcuts switched off) and bench.py's
    distributed code path (barriers, all-reduces, the double-buffered encode -> gather region on a side stream) over RCCL."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ICAMD_FORCE_COLLECTIVES="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29643",
                        os.path.join(T.ROOT, "tests", "nccl_worker.py")], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "NCCL_WORKER_OK" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])
    for impl in ("torch", "c"):  # r06: the gather legs default to the library's own collective (icamd_gather_blocks_rccl)
        d = _run_bench(["--gpus", "1", "--force-distributed", "--backend", "nccl", "--steps", "4", "--warmup", "1",
                        "--workload", "dxt1_rgba8", "--size", "1024", "--batch", "3", "--no-cpu-baseline", "--no-host-api",
                        "--no-sustained", "--no-single-image", "--precondition-seconds", "0"]
                       + (["--gather-impl", "torch"] if impl == "torch" else []))
        assert d["n_gpus"] == 1 and d["parity"].startswith("bit-exact")
        assert d["value_with_gather"] > 0 and d["gather_ms"] > 0 and d["rank0_copy_matches"] is True
        if impl == "torch":
            assert "backend nccl" in d["gather"] and "--gather-impl torch" in d["gather_impl"]
        else:
            assert "icamd_gather_blocks_rccl" in d["gather"] and "icamd_gather_blocks_rccl" in d["gather_impl"], d["gather_impl"]


def test_every_inter_gpu_path_on_a_multi_gpu_box(pkg):
    """The ONE test that needs >= 2 GPUs (the pool's boxes have one; every path below also has a 1-GPU / gloo twin that
    runs everywhere).  On the first multi-GPU box it exercises, in one go:
      1. RCCL, one rank per GPU (tests/nccl_worker.py): texture_range sharding, dist.gather with equal counts, the batched
         isend / irecv gather with UNEQUAL counts (n = 4 world + 1), all_gather_into_tensor;
      2. icamd_compress_batch on distinct devices (host buffers, one worker thread per device);
      3. icamd_encode_batch_sharded_device on distinct devices: hipMemcpyPeerAsync into the gather buffer with peer access
         enabled (default) and refused (ICAMD_DISABLE_PEER_ACCESS=1, a fresh process: peer access is sticky);
      4. bench.py --gpus 2 as the driver launches it: headline + c3 / c4 / c5 legs + the one-large-image slab legs with their
         gathers, parity asserted in every leg."""
    import os
    import subprocess
    import sys
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs >= 2 GPUs (the pool's boxes have one)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for world, port in ((min(n_dev, 4), 29641), (min(n_dev, 3), 29645)):
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(T.ROOT, "tests", "nccl_worker.py")], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0 and "NCCL_WORKER_OK" in p.stdout, (world, p.stdout[-3000:], p.stderr[-3000:])
    # 2
    n, h, w = 11, 200, 264
    imgs = [T.s_mixed(h, w, 3, index=i) for i in range(n)]
    want = [T.oracle_compress(T.ETC, T.RGB, im, h, w) for im in imgs]
    assert pkg.compress_batch_host(T.ETC, T.RGB, imgs, h, w, list(range(n_dev))) == want
    assert pkg.compress_batch_host(T.ETC, T.RGB, imgs, h, w, [n_dev - 1, 0]) == want
    # 3 (peer access refused first, in its own process; then enabled, here)
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n" \
           "import torch, ic_amd_loader, test_gpu_parity as G\npkg = ic_amd_loader.load_package()\n" \
           "G._sharded_case(pkg, pkg.DXT1, 4, 512, 2 * torch.cuda.device_count() + 1, list(range(torch.cuda.device_count())))\n" \
           "print('NO_PEER_OK')" % (T.ROOT, os.path.join(T.ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(env, ICAMD_DISABLE_PEER_ACCESS="1"))
    assert p.returncode == 0 and "NO_PEER_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    devices = list(range(n_dev))
    _sharded_case(pkg, pkg.DXT1, 4, 512, 2 * len(devices) + 1, devices)
    _sharded_case(pkg, pkg.ETC1, 3, 256, len(devices) + 1, devices)
    _sharded_case(pkg, pkg.PVRTC2, 4, 256, len(devices) + 2, devices)
    # 4
    d = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--extra-steps", "2", "--no-sustained", "--no-single-image",
                    "--no-host-api", "--no-cpu-baseline", "--precondition-seconds", "0"], timeout=1200)
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["rank0_copy_matches"] is True
    for name, leg in d["configs"].items():
        assert leg.get("parity", "").startswith("bit-exact") and leg.get("rank0_copy_matches") is True, (name, leg)
    for name, leg in d["slab"].items():
        assert leg.get("parity", "").startswith("bit-exact") and leg.get("value_with_gather"), (name, leg)
    d = _run_bench(["--gpus", "2", "--shard", "slab", "--workload", "dxt5_rgba8", "--size", "8192", "--steps", "3"], timeout=600)
    assert d["scaling"] == "strong" and d["parity"].startswith("bit-exact") and d["value_with_gather"] > 0


# ---- geometries beyond one launch's limits (chunked inside the library; the reference accepts any uint32 size)

def test_more_than_65535_tile_rows(pkg):
    """300 000 x 1024 px (wide grid: 256 x 1 tiles -> 75 000 tile rows, chunked over grid.y) and a 4-pixel-wide strip."""
    import os
    import torch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    h, w = 300000, 1024
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    src = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    src[100000:200000] = (src[100000:200000] >> 3) + 90
    got = _host(pkg.encode_device(T.DXT1, src, h, w, 3))
    want = T.oracle_encode(T.DXT1, src.cpu().numpy(), h, w, 3, threads=cores)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    dec = pkg.decode_device(T.DXT1, torch.frombuffer(bytearray(got), dtype=torch.uint8).cuda(), h, w)
    want_px = T.oracle_decode(T.DXT1, want[: 8 * 256 * 64], 256, 1024)
    assert _host(dec)[: 256 * 1024 * 3] == want_px.tobytes()
    del src, dec
    h, w = 300000, 8
    img = T.s_noise(h, w, 4, index=8)
    assert _host(pkg.encode_device(T.DXT5, _dev(img), h, w, 4)) == T.oracle_encode(T.DXT5, img, h, w, 4, threads=cores)


def test_rows_longer_than_32_bit_lane_offsets(pkg):
    """Row strides of 4 MiB and more (256 x 1 tiles) and of more than 4 GiB / 3 (the 64-bit gather path)."""
    import torch
    for stride, h, w in ((5 << 20, 64, 96), ((3 << 29) + 12, 9, 40)):
        buf = torch.zeros(stride * (h - 1) + w * 3, dtype=torch.uint8, device="cuda")
        img = T.s_mixed(h, w, 3, index=stride % 97)
        rows = _dev(img.reshape(h, w * 3))
        for y in range(h):
            buf[y * stride: y * stride + w * 3] = rows[y]
        for codec, strategy in ((T.DXT1, 2), (T.ETC1, 2), (T.ETC1, 3)):
            out = pkg.encode_device(codec, buf, h, w, 3, etc_strategy=strategy, row_stride_bytes=stride,
                                    src_image_stride_bytes=0)
            assert _host(out) == T.oracle_encode(codec, img, h, w, 3, 0, strategy), (stride, codec)
        del buf


def test_batch_of_more_than_2_31_blocks(pkg):
    """2 049 x 4096^2 DXT1 = 2^31 + 2^20 blocks in ONE call (16 GiB of output): every image reads the same source
    (image stride 0), so every image's output must equal the first, and the first must equal the oracle."""
    import os
    import torch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    n, size = 2049, 4096
    img = T.s_smooth(size, size, 4, index=21)
    img[:1024] = T.s_noise(1024, size, 4, index=21)
    src = _dev(img)
    per = pkg.encoded_size(T.DXT1, size, size)
    out = torch.zeros((n, per), dtype=torch.uint8, device="cuda")
    assert pkg.encode_device(T.DXT1, src, size, size, 4, n_images=n, src_image_stride_bytes=0, out=out) is not None
    torch.cuda.synchronize()
    assert out[0].cpu().numpy().tobytes() == T.oracle_encode(T.DXT1, img, size, size, 4, threads=cores)
    first = out[0].view(torch.int64)
    for i in range(1, n, 64):
        assert bool((out[i:i + 64].view(torch.int64) == first).all()), i
    # the decoder on the same scale: 2 049 images of blocks -> pixels of the last one
    dec = pkg.decode_device(T.DXT1, out[n - 1].contiguous(), size, size)
    assert _host(dec) == T.oracle_decode(T.DXT1, out[0].cpu().numpy().tobytes(), size, size).tobytes()


def test_host_api_band_pipeline(pkg):
    """icamd_compress / icamd_compress_and_pad on images large enough for several bands (bands of whole block rows,
    two streams): ragged heights, row padding, pad grids below / right of the image, page-locked caller buffers."""
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    for compressor, fmt, codec, h, w, pad, padded in (
            (T.DXTC, T.RGB, T.DXT1, 4096, 4096, 0, None), (T.DXTC, T.BGRA, T.DXT5, 2051, 3000, 7, None),
            (T.ETC, T.RGB, T.ETC1, 2999, 2048, 3, None), (T.DXTC, T.BGR, T.DXT1, 3001, 2500, 5, (3100, 2600)),
            (T.DXTC, T.RGBA, T.DXT5, 1500, 4000, 0, (1500, 4100)), (T.ETC, T.RGB, T.ETC1, 2100, 1900, 0, (2133, 1900)),
            # more than 64 MiB of source: three 32 MiB bands on two streams, the last one ragged / with pad rows below
            (T.DXTC, T.RGBA, T.DXT5, 5001, 4000, 0, None), (T.DXTC, T.RGB, T.DXT1, 6001, 4100, 4, (6100, 4100)),
            (T.ETC, T.RGB, T.ETC1, 5999, 4099, 1, None)):
        comps = T.comps_of(fmt)
        img = T.s_smooth(h, w, comps, index=h % 13)
        img[: h // 3] = T.s_noise(h // 3, w, comps, index=w % 11)
        src = T.with_row_padding(img, pad)[: (h - 1) * (w * comps + pad) + w * comps]  # no padding after the LAST row
        if padded is None:
            want = T.oracle_encode(codec, src, h, w, comps, int(fmt in (T.BGR, T.BGRA)), stride=w * comps + pad, threads=cores)
        else:
            want = T.oracle_encode(codec, src, h, w, comps, int(fmt in (T.BGR, T.BGRA)), gh=padded[0], gw=padded[1],
                                   stride=w * comps + pad, threads=cores)
        got = pkg.compress_host(compressor, fmt, src, h, w, padding_bytes_per_row=pad, padded=padded)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), (compressor, fmt, h, w, pad, padded)
    # page-locked caller buffers (icamd_host_register): same bytes
    h = w = 2048
    img = np.ascontiguousarray(T.s_mixed(h, w, 3, index=2))
    out = np.zeros(pkg.compute_compressed_data_size(T.DXTC, T.RGB, h, w), np.uint8)
    pkg.host_register(img)
    pkg.host_register(out)
    try:
        assert pkg.compress_host(T.DXTC, T.RGB, img, h, w, out=out) is out
        assert out.tobytes() == T.oracle_encode(T.DXT1, img, h, w, 3, threads=cores)
    finally:
        pkg.host_unregister(out)
        pkg.host_unregister(img)
    # a wrong out_size is refused before anything is staged (and a huge one allocates nothing)
    assert pkg.compress_host(T.DXTC, T.RGB, img, h, w, out_size=8) is None


# ---- SURVEY 8f row 2 on device: CreateSolidImage / CopySubimage on device-resident block grids

def test_create_solid_and_copy_subimage_on_device_match_oracle(pkg):
    import torch
    rng = np.random.default_rng(11)
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for (h, w) in ((4, 4), (1, 1), (9, 5), (257, 1023), (2048, 2048)):
                color = [int(v) for v in rng.integers(0, 256, 4)]
                want = T.oracle_create_solid(compressor, fmt, h, w, color)
                got = pkg.create_solid_device(compressor, fmt, h, w, color)
                assert (None if got is None else _host(got)) == want, (compressor, fmt, h, w)
    # a wrong output size is the reference's external-storage mismatch
    out = torch.empty(100, dtype=torch.uint8, device="cuda")
    assert pkg.lib().icamd_create_solid_device(T.DXTC, T.RGB, 8, 8, (pkg.ctypes.c_uint8 * 4)(1, 2, 3, 4),
                                               pkg.ctypes.c_void_p(out.data_ptr()), 100, None) == pkg.FALSE
    for compressor, fmt in ((T.DXTC, T.RGB), (T.DXTC, T.BGRA), (T.ETC, T.RGB), (T.ETC, T.RGBA), (T.PVRTC, T.RGBA)):
        ch, cw = 1024, 2052
        bb = 8 if (compressor == T.ETC or T.comps_of(fmt) == 3) else 16
        blocks = rng.integers(0, 256, (ch // 4) * (cw // 4) * bb, dtype=np.uint8)
        d_blocks = _dev(blocks)
        for (r, c, sh, sw) in ((0, 0, ch, cw), (4, 8, 8, 12), (ch - 4, cw - 4, 4, 4), (8, 0, 0, cw), (0, 0, ch + 4, cw),
                               (512, 1024, 512, 1028), (2, 0, 4, 4), (ch, cw, 0, 0), (0, 4, 1024, 2048), (4, 4, 21, 8)):
            want = T.oracle_copy_subimage(compressor, fmt, blocks.tobytes(), ch, cw, r, c, sh, sw)
            got = pkg.copy_subimage_device(compressor, fmt, d_blocks, ch, cw, r, c, sh, sw)
            assert (None if got is None else _host(got)) == want, (compressor, fmt, r, c, sh, sw)


# ---- device-resident images on a device list, optional gather into one device's buffer

def _sharded_case(pkg, codec, comps, size, n, devices, strategy=2):
    import torch
    imgs = [T.s_mixed(size, size, comps, index=40 + i) for i in range(n)]
    srcs = [torch.from_numpy(im).to("cuda:%d" % devices[i % len(devices)]) for i, im in enumerate(imgs)]
    want = [T.oracle_encode(codec, im, size, size, comps, 0, strategy) for im in imgs]
    per = pkg.encoded_size(codec, size, size)
    # (a) per-image outputs only
    outs = [torch.zeros(per, dtype=torch.uint8, device=s.device) for s in srcs]
    st, outs, _ = pkg.encode_batch_sharded_device(codec, srcs, size, size, comps, devices, etc_strategy=strategy, outs=outs)
    assert st == [0] * n
    for i in range(n):
        assert outs[i].cpu().numpy().tobytes() == want[i], ("outs", codec, i)
    # (b) gather only (no per-image buffers): scratch + device-to-device copies into the root buffer
    for root in sorted(set(devices)):
        st, _, gathered = pkg.encode_batch_sharded_device(codec, srcs, size, size, comps, devices, etc_strategy=strategy,
                                                          gather_device=root)
        assert st == [0] * n and gathered.device.index == root
        g = gathered.cpu().numpy()
        for i in range(n):
            assert g[i].tobytes() == want[i], ("gather", codec, root, i)
    # (c) both, with a wider stride in the gathered buffer and one image lacking its own buffer
    outs2 = [torch.zeros(per, dtype=torch.uint8, device=s.device) for s in srcs]
    outs2[1] = None
    gathered = torch.zeros((n, per + 64), dtype=torch.uint8, device="cuda:%d" % devices[0])
    st, _, gathered = pkg.encode_batch_sharded_device(codec, srcs, size, size, comps, devices, etc_strategy=strategy,
                                                      outs=outs2, gather_device=devices[0], gathered=gathered)
    assert st == [0] * n
    g = gathered.cpu().numpy()
    for i in range(n):
        assert g[i, :per].tobytes() == want[i] and not g[i, per:].any()
        if outs2[i] is not None:
            assert outs2[i].cpu().numpy().tobytes() == want[i]


def test_sharded_device_batch_one_gpu_listed_several_times(pkg):
    for codec, comps, size in ((pkg.DXT1, 3, 256), (pkg.DXT1, 4, 512), (pkg.DXT5, 4, 256), (pkg.ETC1, 3, 128),
                               (pkg.PVRTC2, 4, 256)):
        _sharded_case(pkg, codec, comps, size, 7, [0, 0, 0])
    # argument errors: a bad ordinal is an error status, not a crash; a refused geometry is the reference's `false`
    import torch
    src = torch.zeros(64 * 64 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.BackendError):
        pkg.encode_batch_sharded_device(pkg.DXT1, [src], 64, 64, 4, [0], gather_device=99,
                                        gathered=torch.zeros((1, 2048), dtype=torch.uint8, device="cuda"))
    st, _, _ = pkg.encode_batch_sharded_device(pkg.PVRTC2, [src], 48, 48, 4, [0], gather_device=0)
    assert st == [pkg.FALSE]
    # an image with neither its own output buffer nor a gather slot is an argument error of that image only
    good = torch.zeros(pkg.encoded_size(pkg.DXT1, 64, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.BackendError):
        pkg.encode_batch_sharded_device(pkg.DXT1, [src, src], 64, 64, 4, [0], outs=[good, None])


def test_sharded_device_batch_of_evenly_spaced_images(pkg):
    """Images laid out as one array (sources and outputs evenly spaced) go out as batched launches, up to 64 per launch;
    an image that breaks the spacing splits the runs.  Same bytes as one call per image."""
    import torch
    for codec, comps, size, n in ((pkg.DXT1, 4, 128, 23), (pkg.ETC1, 3, 64, 200), (pkg.PVRTC2, 4, 64, 23), (pkg.DXT5, 4, 64, 9)):
        imgs = np.stack([T.s_mixed(size, size, comps, index=900 + i) for i in range(n)])
        want = [T.oracle_encode(codec, imgs[i], size, size, comps, 0, 2) for i in range(n)]
        per = pkg.encoded_size(codec, size, size)
        src = torch.from_numpy(imgs).cuda()
        for devices in ([0], [0, 0, 0]):
            srcs = [src[i] for i in range(n)]
            out = torch.zeros((n, per), dtype=torch.uint8, device="cuda")
            st, _, _ = pkg.encode_batch_sharded_device(codec, srcs, size, size, comps, devices, outs=[out[i] for i in range(n)])
            assert st == [0] * n
            o = out.cpu().numpy()
            assert all(o[i].tobytes() == want[i] for i in range(n)), (codec, devices, "outs")
            # gather only, padded slots (slot spacing = len(devices) x the gather stride within a worker)
            gathered = torch.zeros((n, per + 32), dtype=torch.uint8, device="cuda")
            st, _, gathered = pkg.encode_batch_sharded_device(codec, srcs, size, size, comps, devices, gather_device=0,
                                                              gathered=gathered)
            g = gathered.cpu().numpy()
            assert st == [0] * n and all(g[i, :per].tobytes() == want[i] and not g[i, per:].any() for i in range(n))
            # an image elsewhere in memory in the middle of the array, and one missing source
            lone = src[n // 2].clone()
            srcs2 = list(srcs)
            srcs2[n // 2] = lone
            out.zero_()
            st, _, _ = pkg.encode_batch_sharded_device(codec, srcs2, size, size, comps, devices, outs=[out[i] for i in range(n)])
            o = out.cpu().numpy()
            assert st == [0] * n and all(o[i].tobytes() == want[i] for i in range(n)), (codec, devices, "split runs")


def test_pvrtc_workspace_must_be_device_memory(pkg):
    import numpy as np
    host = np.zeros(4096, np.uint8)
    assert pkg.lib().icamd_pvrtc2_set_workspace(pkg.ctypes.c_void_p(host.ctypes.data), host.size) == -4  # ICAMD_ERR_ARG
    assert pkg.pvrtc_set_workspace(None)


def test_clock_probe_reports_a_plausible_shader_clock(pkg):
    import torch
    s = torch.cuda.Stream()
    res = pkg.clock_probe(20000, s)
    torch.cuda.synchronize()
    r = res()
    assert r is not None and 15.0 < r["interval_ms"] < 200.0
    assert 100.0 < r["shader_MHz"] < 3000.0, r


def test_process_that_used_every_entry_point_exits_cleanly(pkg):
    """No HIP call may run from a thread_local / static destructor (the runtime can be gone by then): a process that
    exercised the host-buffer API (thread-local staging), PVRTC (thread-local workspace), worker threads and the batch
    entry points must exit 0 with no HIP error in its AMD_LOG_LEVEL=1 log."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, threading
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
def work():
    for compressor, fmt, comps in ((pkg.COMPRESSOR_DXTC, pkg.RGB, 3), (pkg.COMPRESSOR_DXTC, pkg.RGBA, 4),
                                   (pkg.COMPRESSOR_ETC, pkg.RGB, 3), (pkg.COMPRESSOR_PVRTC, pkg.RGBA, 4)):
        img = T.s_mixed(64, 64, comps, index=3)
        out = pkg.compress_host(compressor, fmt, img, 64, 64)
        assert out is not None
        if compressor != pkg.COMPRESSOR_PVRTC:
            assert pkg.pad_host(compressor, fmt, out, 64, 64, 72, 80) is not None
            assert pkg.downsample_host(compressor, fmt, out, 64, 64) is not None
    d = torch.from_numpy(T.s_mixed(64, 64, 4, index=5)).cuda()
    pkg.encode_device(pkg.PVRTC2, d, 64, 64, 4); pkg.encode_device(pkg.DXT5, d, 64, 64, 4)
    torch.cuda.synchronize()
work()
ts = [threading.Thread(target=work) for _ in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
imgs = [T.s_mixed(64, 64, 3, index=i) for i in range(5)]
assert all(o is not None for o in pkg.compress_batch_host(pkg.COMPRESSOR_DXTC, pkg.RGB, imgs, 64, 64, [0, 0]))
srcs = [torch.from_numpy(T.s_mixed(64, 64, 4, index=i)).cuda() for i in range(5)]
st, _, g = pkg.encode_batch_sharded_device(pkg.PVRTC2, srcs, 64, 64, 4, [0, 0], gather_device=0)
assert st == [0] * 5
print("WORK DONE")
''' % (T.ROOT, os.path.join(T.ROOT, "tests"))
    env = dict(os.environ, AMD_LOG_LEVEL="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert "WORK DONE" in p.stdout
    bad = [l for l in p.stderr.splitlines() if "hipError" in l or "HIP error" in l or "Assertion" in l or "core dumped" in l]
    assert not bad, bad[:10]


def test_downsample_batch_and_arbitrary_block_words(pkg):
    """icamd_downsample_batch_device (r04 extension): n equally shaped grids in one launch == n single calls == the oracle;
    and the DXT fast path (palette planes + quad selectors) on arbitrary block words -- three-colour DXT1 blocks, equal
    endpoints, DXT5's six-value alpha -- against the oracle's decode-average-encode."""
    import torch
    g = np.random.Generator(np.random.PCG64(23))
    for compressor, fmt, codec, strategy in ((T.DXTC, T.RGB, T.DXT1, 2), (T.DXTC, T.RGBA, T.DXT5, 2), (T.ETC, T.RGB, T.ETC1, 3),
                                            (T.ETC, T.RGB, T.ETC1, 2), (T.ETC, T.RGB, T.ETC1, 1)):
        bb = 16 if codec == T.DXT5 else 8
        h, w, n = 64, 96, 5
        raw = g.integers(0, 256, size=(n, (h // 4) * (w // 4), bb), dtype=np.uint8)
        if codec != T.ETC1:
            col = raw[1, :, bb - 8:]
            col[:, :4] = np.sort(col[:, :4].copy().view(np.uint16), axis=1).view(np.uint8)  # c0 <= c1
            raw[2, :, bb - 6:bb - 4] = raw[2, :, bb - 8:bb - 6]                               # c0 == c1
            if codec == T.DXT5:
                raw[3, :, :2] = np.sort(raw[3, :, :2], axis=1)                                # alpha0 <= alpha1
        else:
            raw[:] = np.stack([np.frombuffer(T.oracle_compress(T.ETC, T.RGB, T.s_mixed(h, w, 3, index=70 + i), h, w), np.uint8)
                               .reshape(-1, 8) for i in range(n)])
        want = [T.oracle_downsample(compressor, fmt, raw[i].tobytes(), h, w, strategy) for i in range(n)]
        got = pkg.downsample_device(compressor, fmt, _dev(raw.reshape(n, -1)), h, w, etc_strategy=strategy, n_images=n)
        torch.cuda.synchronize()
        for i in range(n):
            assert got[i].cpu().numpy().tobytes() == want[i], (codec, strategy, i)
            assert pkg.downsample_host(compressor, fmt, raw[i].tobytes(), h, w, strategy) == want[i]
    # a batch with padded strides, and a refused geometry
    raw = g.integers(0, 256, size=(3, 16 * 16 * 8 + 24), dtype=np.uint8)
    d = _dev(raw)
    out = torch.zeros((3, 8 * 8 * 8 + 16), dtype=torch.uint8, device="cuda")
    rc = pkg.lib().icamd_downsample_batch_device(T.DXTC, 2, T.RGB, 64, 64, 3, pkg.ctypes.c_void_p(d.data_ptr()), raw.shape[1],
                                                 pkg.ctypes.c_void_p(out.data_ptr()), out.shape[1], 8 * 8 * 8, None)
    torch.cuda.synchronize()
    assert rc == 0
    for i in range(3):
        assert out[i, :512].cpu().numpy().tobytes() == T.oracle_downsample(T.DXTC, T.RGB, raw[i, :2048].tobytes(), 64, 64, 2)
        assert not out[i, 512:].any()
    assert pkg.lib().icamd_downsample_batch_device(T.DXTC, 2, T.RGB, 12, 64, 3, pkg.ctypes.c_void_p(d.data_ptr()), raw.shape[1],
                                                   pkg.ctypes.c_void_p(out.data_ptr()), out.shape[1], 8 * 2 * 8, None) == 1  # 3 block rows
