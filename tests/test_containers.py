"""Container framing (DDS / KTX 1.1 / PKM / PVR v3) of the encoders' block streams -- SURVEY.md 8(f) row 4, tail.

The reference has no container code (compressed_image.h:52-66 stops at format + dimensions), so these tests pin the C-ABI
against headers assembled here, field by field, from the public format descriptions with struct.pack -- an independent
second construction, not a reference output.  Host-only: no GPU, no oracle.
"""
import importlib
import struct

import numpy as np
import pytest

pkg = importlib.import_module("image-compression_amd")
DXT1, DXT5, ETC1, PVRTC2 = 0, 1, 2, 3
DDS, KTX, PKM, PVR = pkg.CONTAINER_DDS, pkg.CONTAINER_KTX, pkg.CONTAINER_PKM, pkg.CONTAINER_PVR


def level_bytes(codec, h, w, l):
    lh, lw = max(1, h >> l), max(1, w >> l)
    if codec == PVRTC2:
        return lw * lh // 4
    return ((lh + 3) // 4) * ((lw + 3) // 4) * (16 if codec == DXT5 else 8)


def make_levels(codec, h, w, n, seed=7):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, level_bytes(codec, h, w, l), dtype=np.uint8).tobytes() for l in range(n)]


def expected_dds(codec, h, w, levels):
    n = len(levels)
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000 | (0x20000 if n > 1 else 0)
    caps = 0x1000 | ((0x8 | 0x400000) if n > 1 else 0)
    pf = struct.pack("<II4sIIIII", 32, 0x4, b"DXT1" if codec == DXT1 else b"DXT5", 0, 0, 0, 0, 0)
    hdr = b"DDS " + struct.pack("<IIIIIII", 124, flags, h, w, len(levels[0]), 0, n) + b"\0" * 44 + pf + \
        struct.pack("<IIIII", caps, 0, 0, 0, 0)
    assert len(hdr) == 128
    return hdr + b"".join(levels)


def expected_ktx(codec, h, w, levels):
    internal = {DXT1: 0x83F0, DXT5: 0x83F3, ETC1: 0x8D64, PVRTC2: 0x8C03}[codec]
    base = 0x1908 if codec in (DXT5, PVRTC2) else 0x1907
    hdr = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A]) + \
        struct.pack("<13I", 0x04030201, 0, 1, 0, internal, base, w, h, 0, 0, 1, len(levels), 0)
    assert len(hdr) == 64
    return hdr + b"".join(struct.pack("<I", len(b)) + b for b in levels)


def expected_pkm(h, w, levels):
    return b"PKM 10" + struct.pack(">HHHHH", 0, (w + 3) & ~3, (h + 3) & ~3, w, h) + levels[0]


def expected_pvr(codec, h, w, levels):
    fmt = {PVRTC2: 1, ETC1: 6, DXT1: 7, DXT5: 11}[codec]
    hdr = struct.pack("<IIQIIIIIIIII", 0x03525650, 0, fmt, 0, 0, h, w, 1, 1, 1, len(levels), 0)
    assert len(hdr) == 52
    return hdr + b"".join(levels)


CASES = [
    (DDS, DXT1, 64, 64, 1), (DDS, DXT1, 64, 32, 7), (DDS, DXT5, 61, 59, 3), (DDS, DXT5, 4096, 4096, 13),
    (KTX, DXT1, 30, 30, 1), (KTX, DXT5, 256, 128, 9), (KTX, ETC1, 61, 59, 6), (KTX, PVRTC2, 64, 64, 4),
    (PKM, ETC1, 61, 59, 1), (PKM, ETC1, 1024, 1024, 1),
    (PVR, PVRTC2, 256, 256, 6), (PVR, PVRTC2, 8, 8, 1), (PVR, ETC1, 128, 64, 8), (PVR, DXT1, 5, 3, 3), (PVR, DXT5, 16, 16, 5),
]


@pytest.mark.parametrize("container,codec,h,w,n", CASES)
def test_container_bytes_match_the_format_descriptions(container, codec, h, w, n):
    levels = make_levels(codec, h, w, n)
    want = {DDS: lambda: expected_dds(codec, h, w, levels), KTX: lambda: expected_ktx(codec, h, w, levels),
            PKM: lambda: expected_pkm(h, w, levels), PVR: lambda: expected_pvr(codec, h, w, levels)}[container]()
    assert pkg.container_size(container, codec, h, w, n) == len(want)
    got = pkg.container_write(container, codec, h, w, levels)
    assert got == want


def test_level_sizes_are_the_encoders_sizes():
    # a level's block stream is what Compress returns for that size (compressor4x4_helper.h:594-636 halves the same way)
    for codec, comp, fmt in ((DXT1, pkg.COMPRESSOR_DXTC, pkg.RGB), (DXT5, pkg.COMPRESSOR_DXTC, pkg.RGBA), (ETC1, pkg.COMPRESSOR_ETC, pkg.RGB)):
        for h, w in ((61, 59), (256, 64), (5, 3)):
            for l in range(4):
                assert level_bytes(codec, h, w, l) == pkg.compute_compressed_data_size(comp, fmt, max(1, h >> l), max(1, w >> l))
    for size in (8, 64, 512):
        assert level_bytes(PVRTC2, size, size, 0) == pkg.compute_compressed_data_size(pkg.COMPRESSOR_PVRTC, pkg.RGBA, size, size)


def test_container_refusals():
    # codecs a container has no code for
    assert pkg.container_size(DDS, ETC1, 64, 64, 1) == 0
    assert pkg.container_size(DDS, PVRTC2, 64, 64, 1) == 0
    assert pkg.container_size(PKM, DXT1, 64, 64, 1) == 0
    # PKM: one level, 16-bit dimensions
    assert pkg.container_size(PKM, ETC1, 64, 64, 2) == 0
    assert pkg.container_size(PKM, ETC1, 65536, 4, 1) == 0
    # level counts: none, or past the 1 x 1 level
    assert pkg.container_size(KTX, DXT1, 64, 64, 0) == 0
    assert pkg.container_size(KTX, DXT1, 64, 64, 7) > 0
    assert pkg.container_size(KTX, DXT1, 64, 64, 8) == 0
    assert pkg.container_size(KTX, DXT1, 64, 16, 7) > 0   # 1 x 1 is reached by the LARGER dimension
    # PVRTC: the encoder's domain only (square powers of two, levels of 8 x 8 and up)
    assert pkg.container_size(PVR, PVRTC2, 64, 32, 1) == 0
    assert pkg.container_size(PVR, PVRTC2, 48, 48, 1) == 0
    assert pkg.container_size(PVR, PVRTC2, 64, 64, 4) > 0
    assert pkg.container_size(PVR, PVRTC2, 64, 64, 5) == 0
    assert pkg.container_size(KTX, DXT1, 0, 64, 1) == 0
    # wrong level size / wrong output size: false, like a size mismatch in Compress (compressor4x4_helper.cc:34-41)
    levels = make_levels(DXT1, 64, 64, 2)
    assert pkg.container_write(KTX, DXT1, 64, 64, [levels[0], levels[1][:-8]]) is None
    assert pkg.container_write(DDS, ETC1, 64, 64, make_levels(ETC1, 64, 64, 1)) is None
    with pytest.raises(pkg.BackendError):
        pkg.container_write(9, DXT1, 64, 64, levels)
    with pytest.raises(pkg.BackendError):
        pkg.container_write(KTX, 9, 64, 64, levels)


def parse_ktx(blob):
    f = struct.unpack_from("<13I", blob, 12)
    assert f[0] == 0x04030201
    w, h, n = f[6], f[7], f[11]
    off, out = 64 + f[12], []
    for _ in range(n):
        (sz,) = struct.unpack_from("<I", blob, off)
        out.append(blob[off + 4:off + 4 + sz])
        off += 4 + sz + (-sz) % 4
    assert off == len(blob)
    return f[4], h, w, out


def test_ktx_round_trip_through_a_reader():
    levels = make_levels(ETC1, 100, 36, 7)
    internal, h, w, got = parse_ktx(pkg.container_write(KTX, ETC1, 100, 36, levels))
    assert (internal, h, w) == (0x8D64, 100, 36) and got == levels
