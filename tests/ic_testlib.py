"""Shared test plumbing: ctypes loaders for the oracle / compiled reference, and the
deterministic synthetic-image generators of SURVEY.md 8(d).

TEST INFRASTRUCTURE.  The product library (libic_amd.so) is loaded through the
package loader in image-compression_amd/, never from here.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libic_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libic_ref.so")

# block codecs (oracle/ic_oracle.h, include/ic_amd.h use the same numbering)
DXT1, DXT5, ETC1, PVRTC2 = 0, 1, 2, 3
PVRTC4 = 4  # extension, parity unpinned (the reference has no 4 bpp mode)
# reference compressor classes / formats / ETC strategies
DXTC, ETC, PVRTC = 0, 1, 2
RGB, BGR, RGBA, BGRA = 0, 1, 2, 3
SPLIT_H, SPLIT_V, SMALLER_ERROR, HEURISTIC = 0, 1, 2, 3

u32, sz, vp, ci = ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int


def comps_of(fmt):
    return 3 if fmt in (RGB, BGR) else 4


def _ptr(a):
    return a.ctypes.data if a is not None else None


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "ic_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libic_oracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        L = ctypes.CDLL(build_oracle())
        L.ico_compute_compressed_data_size.restype = sz
        L.ico_compute_compressed_data_size.argtypes = [ci, ci, u32, u32]
        L.ico_compress.restype = ci
        L.ico_compress.argtypes = [ci, ci, ci, u32, u32, u32, vp, vp, sz]
        L.ico_compress_and_pad.restype = ci
        L.ico_compress_and_pad.argtypes = [ci, ci, ci, u32, u32, u32, u32, u32, vp, vp, sz]
        L.ico_encode.restype = ci
        L.ico_encode.argtypes = [ci, ci, ci, ci, u32, u32, u32, u32, u32, vp, vp, ci]
        L.ico_encoded_size.restype = sz
        L.ico_encoded_size.argtypes = [ci, u32, u32]
        L.ico_decode.restype = ci
        L.ico_decode.argtypes = [ci, ci, u32, u32, u32, vp, vp]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        L = ctypes.CDLL(REF_SO)
        L.ref_compressed_size.restype = sz
        L.ref_compressed_size.argtypes = [ci, ci, u32, u32]
        L.ref_compress.restype = ci
        L.ref_compress.argtypes = [ci, ci, ci, u32, u32, u32, vp, vp, sz, vp]
        L.ref_compress_owned.restype = ctypes.c_long
        L.ref_compress_owned.argtypes = [ci, ci, ci, u32, u32, u32, vp, vp, sz]
        L.ref_compress_and_pad.restype = ci
        L.ref_compress_and_pad.argtypes = [ci, ci, ci, u32, u32, u32, u32, u32, vp, vp, sz, vp]
        L.ref_decompress.restype = ctypes.c_long
        L.ref_decompress.argtypes = [ci, ci, u32, u32, u32, u32, u32, vp, sz, vp, sz]
        _ref = L
    return _ref


# ------------------------------------------------------------------ wrappers


def oracle_size(compressor, fmt, h, w):
    return oracle().ico_compute_compressed_data_size(compressor, fmt, h, w)


def oracle_compress(compressor, fmt, img_bytes, h, w, pad=0, strategy=SMALLER_ERROR, out_size=None):
    """Returns bytes, or None when the oracle returns false."""
    n = oracle_size(compressor, fmt, h, w) if out_size is None else out_size
    out = np.zeros(max(n, 1), np.uint8)
    src = np.ascontiguousarray(img_bytes, dtype=np.uint8)
    ok = oracle().ico_compress(compressor, strategy, fmt, h, w, pad, _ptr(src), _ptr(out), n)
    return out[:n].tobytes() if ok else None


def oracle_compress_and_pad(compressor, fmt, img_bytes, h, w, ph, pw, pad=0, strategy=SMALLER_ERROR):
    n = oracle_size(compressor, fmt, max(h, ph), max(w, pw))
    out = np.zeros(max(n, 1), np.uint8)
    src = np.ascontiguousarray(img_bytes, dtype=np.uint8)
    ok = oracle().ico_compress_and_pad(compressor, strategy, fmt, h, w, ph, pw, pad, _ptr(src), _ptr(out), n)
    return out[:n].tobytes() if ok else None


def oracle_encode(codec, src, h, w, comps, swap=0, strategy=SMALLER_ERROR, gh=None, gw=None, stride=None,
                  threads=1):
    gh = h if gh is None else gh
    gw = w if gw is None else gw
    stride = w * comps if stride is None else stride
    n = oracle().ico_encoded_size(codec, max(gh, h), max(gw, w))
    out = np.zeros(max(n, 1), np.uint8)
    src = np.ascontiguousarray(src, dtype=np.uint8)
    ok = oracle().ico_encode(codec, strategy, comps, swap, h, w, gh, gw, stride, _ptr(src), _ptr(out), threads)
    return out[:n].tobytes() if ok else None


def oracle_decode(codec, blocks, h, w, swap=0, pad=0):
    comps = 4 if codec in (DXT5, PVRTC2, PVRTC4) else 3
    out = np.zeros(h * (w * comps + pad), np.uint8)
    b = np.frombuffer(blocks, np.uint8)
    ok = oracle().ico_decode(codec, swap, h, w, pad, _ptr(b), _ptr(out))
    return out if ok else None


def ref_size(compressor, fmt, h, w):
    return ref().ref_compressed_size(compressor, fmt, h, w)


def ref_compress(compressor, fmt, img_bytes, h, w, pad=0, strategy=SMALLER_ERROR, out_size=None):
    n = ref_size(compressor, fmt, h, w) if out_size is None else out_size
    out = np.zeros(max(n, 1), np.uint8)
    src = np.ascontiguousarray(img_bytes, dtype=np.uint8)
    ok = ref().ref_compress(compressor, strategy, fmt, h, w, pad, _ptr(src), _ptr(out), n, None)
    return out[:n].tobytes() if ok else None


def ref_compress_and_pad(compressor, fmt, img_bytes, h, w, ph, pw, pad=0, strategy=SMALLER_ERROR):
    n = ref_size(compressor, fmt, max(h, ph), max(w, pw))
    out = np.zeros(max(n, 1), np.uint8)
    src = np.ascontiguousarray(img_bytes, dtype=np.uint8)
    ok = ref().ref_compress_and_pad(compressor, strategy, fmt, h, w, ph, pw, pad, _ptr(src), _ptr(out), n, None)
    return out[:n].tobytes() if ok else None


def ref_decompress(compressor, fmt, blocks, h, w, pad=0):
    comps = comps_of(fmt)
    ch, cw = 4 * ((h + 3) // 4), 4 * ((w + 3) // 4)
    cap = h * (w * comps + pad) + 64
    out = np.zeros(cap, np.uint8)
    b = np.frombuffer(blocks, np.uint8).copy()
    n = ref().ref_decompress(compressor, fmt, h, w, ch, cw, pad, _ptr(b), b.size, _ptr(out), cap)
    return None if n < 0 else out[:n]


# --------------------------------------------------------- synthetic images
# integer-only generators so host, device and every box agree.
# numpy's PCG64 is deterministic across platforms for integers().

SEED0 = 0x1234ABCD


def _rng(seed_index):
    return np.random.Generator(np.random.PCG64(SEED0 + seed_index))


def s_noise(h, w, comps, index=0):
    return _rng(index).integers(0, 256, size=(h, w, comps), dtype=np.uint8)


def s_smooth(h, w, comps, index=0):
    g = _rng(1000 + index)
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    n = g.integers(0, 32, size=(h, w), dtype=np.int64)
    img = np.empty((h, w, comps), np.uint8)
    img[..., 0] = (255 * x // max(w, 1) + n) & 255
    img[..., 1] = (255 * y // max(h, 1) + n) & 255
    img[..., 2] = (255 * (x + y) // max(w + h, 1) + n) & 255
    if comps == 4:
        a = g.integers(0, 256, size=(h, w), dtype=np.int64)
        keep = g.integers(0, 8, size=(h, w)) != 0
        img[..., 3] = np.where(keep, 255, a)
    return img


def s_flat(h, w, comps, index=0):
    g = _rng(2000 + index)
    th, tw = (h + 15) // 16, (w + 15) // 16
    tiles = g.integers(0, 256, size=(th, tw, comps), dtype=np.uint8)
    img = np.repeat(np.repeat(tiles, 16, axis=0), 16, axis=1)[:h, :w].copy()
    noisy = g.integers(0, 8, size=(th, tw)) == 0
    mask = np.repeat(np.repeat(noisy, 16, axis=0), 16, axis=1)[:h, :w]
    noise = g.integers(0, 256, size=(h, w, comps), dtype=np.uint8)
    img[mask] = noise[mask]
    return img


def s_mixed(h, w, comps, index=0):
    """Quadrant mix + special cases (near-constant, black, white, alpha 224..255, channel-zero areas)."""
    g = _rng(3000 + index)
    img = s_noise(h, w, comps, 50 + index)
    h2, w2 = h // 2, w // 2
    img[:h2, w2:] = s_smooth(h, w, comps, index)[:h2, w2:]
    img[h2:, :w2] = s_flat(h, w, comps, index)[h2:, :w2]
    q = img[h2:, w2:]
    qh, qw = q.shape[:2]
    base = g.integers(0, 256, size=(1, 1, comps), dtype=np.int64)
    jitter = g.integers(-3, 4, size=(qh, qw, comps), dtype=np.int64)
    q[...] = np.clip(base + jitter, 0, 255).astype(np.uint8)
    if qh >= 8 and qw >= 24:
        q[:4, :8] = 0
        q[:4, 8:16] = 255
        q[4:8, :8, 0] = 0  # a zero red channel (PVRTC max-index quirk)
        if comps == 4:
            q[4:8, 8:16, 3] = g.integers(224, 256, size=(4, 8), dtype=np.uint8)
            q[:4, 16:24, 3] = 0
    return img


def random_blocks(codec, h, w, seed):
    """Seeded arbitrary block words for an h x w image of `codec` (decoder tests; golden fixtures refer to the seed)."""
    n = ((h + 3) // 4) * ((w + 3) // 4) * (16 if codec == DXT5 else 8)
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def solid_ramp_image():
    """16 x 1024 RGB: block (row r, column v) is one colour -- (v, v, v), (v, 0, 0), (0, v, 0), (0, 0, v) for r = 0..3.
    Its DXT1 encoding reads every row of the constant-colour endpoint table for both channel widths."""
    img = np.zeros((16, 1024, 3), np.uint8)
    v = np.repeat(np.arange(256, dtype=np.uint8), 4)
    img[0:4, :, :] = v[None, :, None]
    img[4:8, :, 0] = v[None, :]
    img[8:12, :, 1] = v[None, :]
    img[12:16, :, 2] = v[None, :]
    return img


CONST_TABLE_SHA256 = "50ee564dd108fbbbe79da46d22f7dfc95d18c140f8112eecfb68ba9820e300cf"


def const_table_bytes():
    """The 2 048 values of image-compression_amd/csrc/dxtc_const_table.inc as the product and the oracle compile them in."""
    import re
    with open(os.path.join(ROOT, "image-compression_amd", "csrc", "dxtc_const_table.inc")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return bytes(int(t) for t in re.findall(r"\d+", text))


GENERATORS = {"noise": s_noise, "smooth": s_smooth, "flat": s_flat, "mixed": s_mixed}


def with_row_padding(img, pad):
    """(h, w, c) -> flat bytes with `pad` junk bytes after each row."""
    h = img.shape[0]
    flat = img.reshape(h, -1)
    if pad == 0:
        return np.ascontiguousarray(flat).reshape(-1)
    junk = np.full((h, pad), 0xA5, np.uint8)
    return np.ascontiguousarray(np.concatenate([flat, junk], axis=1)).reshape(-1)


# ------------------------------------------- compressed-domain operations

def _bind_ops():
    L = oracle()
    if getattr(L, "_ops_bound", False):
        return L
    L.ico_create_solid.restype = ci
    L.ico_create_solid.argtypes = [ci, ci, u32, u32, vp, vp]
    L.ico_copy_subimage.restype = ci
    L.ico_copy_subimage.argtypes = [ci, ci, u32, u32, vp, u32, u32, u32, u32, vp]
    L.ico_pad.restype = ci
    L.ico_pad.argtypes = [ci, ci, ci, u32, u32, vp, u32, u32, vp]
    L.ico_downsample.restype = ci
    L.ico_downsample.argtypes = [ci, ci, ci, u32, u32, vp, vp]
    L.ico_transcode_dxt1_to_etc1.restype = None
    L.ico_transcode_dxt1_to_etc1.argtypes = [vp, sz]
    L._ops_bound = True
    return L


def _bind_ref_ops():
    L = ref()
    if getattr(L, "_ops_bound", False):
        return L
    L.ref_create_solid.restype = ci
    L.ref_create_solid.argtypes = [ci, ci, u32, u32, vp, vp, sz]
    L.ref_pad.restype = ci
    L.ref_pad.argtypes = [ci, ci, ci, u32, u32, u32, u32, vp, sz, u32, u32, vp, sz, vp, vp]
    L.ref_downsample.restype = ci
    L.ref_downsample.argtypes = [ci, ci, ci, u32, u32, u32, u32, vp, sz, vp, sz, vp, vp]
    L.ref_copy_subimage.restype = ci
    L.ref_copy_subimage.argtypes = [ci, ci, u32, u32, u32, u32, vp, sz, u32, u32, u32, u32, vp, sz, vp, vp]
    L.ref_transcode_dxt1_to_etc1.restype = None
    L.ref_transcode_dxt1_to_etc1.argtypes = [u32, u32, u32, u32, vp, sz]
    L._ops_bound = True
    return L


def _b4(n):
    return 4 * ((n + 3) // 4)


def oracle_create_solid(compressor, fmt, h, w, color):
    n = ((h + 3) // 4) * ((w + 3) // 4) * (8 if (compressor == ETC or comps_of(fmt) == 3) else 16)
    out = np.zeros(max(n, 1), np.uint8)
    c = np.asarray(color, np.uint8)
    ok = _bind_ops().ico_create_solid(compressor, fmt, h, w, _ptr(c), _ptr(out))
    return out[:n].tobytes() if ok else None


def ref_create_solid(compressor, fmt, h, w, color):
    n = ((h + 3) // 4) * ((w + 3) // 4) * (8 if (compressor == ETC or comps_of(fmt) == 3) else 16)
    out = np.zeros(max(n, 1), np.uint8)
    c = np.asarray(color, np.uint8)
    ok = _bind_ref_ops().ref_create_solid(compressor, fmt, h, w, _ptr(c), _ptr(out), n)
    return out[:n].tobytes() if ok else None


def oracle_copy_subimage(compressor, fmt, blocks, ch, cw, row, col, h, w):
    bb = 8 if (compressor == ETC or comps_of(fmt) == 3) else 16
    n = ((h + 3) // 4) * ((w + 3) // 4) * bb
    out = np.zeros(max(n, 1), np.uint8)
    b = np.frombuffer(blocks, np.uint8)
    ok = _bind_ops().ico_copy_subimage(compressor, fmt, ch, cw, _ptr(b), row, col, h, w, _ptr(out))
    return out[:n].tobytes() if ok else None


def ref_copy_subimage(compressor, fmt, blocks, uh, uw, row, col, h, w):
    b = np.frombuffer(blocks, np.uint8).copy()
    out = np.zeros(b.size + 64, np.uint8)
    n = sz(0)
    ok = _bind_ref_ops().ref_copy_subimage(compressor, fmt, uh, uw, _b4(uh), _b4(uw), _ptr(b), b.size, row, col, h, w,
                                           _ptr(out), out.size, ctypes.byref(n), None)
    return out[:n.value].tobytes() if ok else None


def oracle_pad(compressor, fmt, blocks, ch, cw, ph, pw, strategy=SMALLER_ERROR):
    bb = 8 if (compressor == ETC or comps_of(fmt) == 3) else 16
    cap = max(((max(ph, ch) + 3) // 4) * ((max(pw, cw) + 3) // 4) * bb, len(blocks)) + 64
    out = np.zeros(cap, np.uint8)
    b = np.frombuffer(blocks, np.uint8)
    rc = _bind_ops().ico_pad(compressor, strategy, fmt, ch, cw, _ptr(b), ph, pw, _ptr(out))
    if rc == 0:
        return None
    n = len(blocks) if rc == 2 else ((ph + 3) // 4) * ((pw + 3) // 4) * bb
    return out[:n].tobytes()


def ref_pad(compressor, fmt, blocks, uh, uw, ph, pw, strategy=SMALLER_ERROR):
    b = np.frombuffer(blocks, np.uint8).copy()
    bb = 8 if (compressor == ETC or comps_of(fmt) == 3) else 16
    cap = max(((max(ph, uh) + 3) // 4) * ((max(pw, uw) + 3) // 4) * bb, b.size) + 64
    out = np.zeros(cap, np.uint8)
    n = sz(0)
    meta = (u32 * 5)()
    ok = _bind_ref_ops().ref_pad(compressor, strategy, fmt, uh, uw, _b4(uh), _b4(uw), _ptr(b), b.size, ph, pw, _ptr(out),
                                 out.size, ctypes.byref(n), meta)
    return (out[:n.value].tobytes(), tuple(meta)) if ok else None


def oracle_downsample(compressor, fmt, blocks, uh, uw, strategy=SMALLER_ERROR):
    bb = 8 if (compressor == ETC or comps_of(fmt) == 3) else 16
    dh, dw = (uh + 1) // 2, (uw + 1) // 2
    n = ((dh + 3) // 4) * ((dw + 3) // 4) * bb
    out = np.zeros(max(n, 1), np.uint8)
    b = np.frombuffer(blocks, np.uint8)
    ok = _bind_ops().ico_downsample(compressor, strategy, fmt, uh, uw, _ptr(b), _ptr(out))
    return out[:n].tobytes() if ok else None


def ref_downsample(compressor, fmt, blocks, uh, uw, strategy=SMALLER_ERROR):
    b = np.frombuffer(blocks, np.uint8).copy()
    out = np.zeros(b.size + 64, np.uint8)
    n = sz(0)
    meta = (u32 * 5)()
    ok = _bind_ref_ops().ref_downsample(compressor, strategy, fmt, uh, uw, _b4(uh), _b4(uw), _ptr(b), b.size, _ptr(out),
                                        out.size, ctypes.byref(n), meta)
    return (out[:n.value].tobytes(), tuple(meta)) if ok else None


def oracle_transcode(blocks):
    b = np.frombuffer(blocks, np.uint8).copy()
    _bind_ops().ico_transcode_dxt1_to_etc1(_ptr(b), b.size)
    return b.tobytes()


def ref_transcode(blocks, uh, uw):
    b = np.frombuffer(blocks, np.uint8).copy()
    _bind_ref_ops().ref_transcode_dxt1_to_etc1(uh, uw, _b4(uh), _b4(uw), _ptr(b), b.size)
    return b.tobytes()


# ---- seeded random soak cases shared by the CPU tier (host-emulated kernel math) and the GPU tier

SOAK_FORMATS = [(DXT1, 3, 0, 2), (DXT1, 3, 1, 2), (DXT1, 4, 0, 2), (DXT1, 4, 1, 2), (DXT5, 4, 0, 2), (DXT5, 4, 1, 2),
                (ETC1, 3, 0, 0), (ETC1, 3, 0, 1), (ETC1, 3, 0, 2), (ETC1, 3, 0, 3), (ETC1, 4, 0, 2), (ETC1, 4, 0, 3)]


def soak_image(rng, h, w, comps):
    kind = int(rng.integers(0, 6))
    if kind == 0:    # full-range noise
        img = rng.integers(0, 256, (h, w, comps), dtype=np.uint8)
    elif kind == 1:  # mid-tones only: every ETC1 codeword up to b = 80 stays unclamped (wave-uniform shortcut fires)
        img = rng.integers(100, 156, (h, w, comps), dtype=np.uint8)
    elif kind == 2:  # mid-tone left half, saturated right half: waves with and without clamping lanes
        img = rng.integers(100, 156, (h, w, comps), dtype=np.uint8)
        img[:, w // 2:] = rng.choice(np.array([0, 3, 252, 255], dtype=np.uint8), (h, w - w // 2, comps))
    elif kind == 3:  # few distinct colours (constant-colour blocks, ties)
        pal = rng.integers(0, 256, (3, comps), dtype=np.uint8)
        img = pal[rng.integers(0, 3, (h // 4 + 1, w // 4 + 1))].repeat(4, axis=0).repeat(4, axis=1)[:h, :w]
    elif kind == 5:  # 16 x 16 tiles of one colour, a quarter of them noise: one-colour blocks next to busy ones in a wave
        th, tw = h // 16 + 1, w // 16 + 1
        img = rng.integers(0, 256, (th, tw, comps), dtype=np.uint8).repeat(16, axis=0).repeat(16, axis=1)[:h, :w].copy()
        noisy = (rng.random((th, tw)) < 0.25).repeat(16, axis=0).repeat(16, axis=1)[:h, :w]
        img[noisy] = rng.integers(0, 256, (int(noisy.sum()), comps), dtype=np.uint8)
    else:            # gradients plus a little noise
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 255 // max(h + w - 2, 1)),
                         255 - (x * 255 // max(w - 1, 1))][:comps], axis=-1)
        img = np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)
    if comps == 4 and rng.integers(0, 2):  # alpha with exact 0 / 255 runs (DXT5 6-alpha mode, PVRTC opaque flags)
        img = img.copy()
        img[..., 3] = rng.choice(np.array([0, 255, 255, 255, 17, 128, 240], dtype=np.uint8), (h, w))
    return np.ascontiguousarray(img)



def soak_cases(seed, n_block_codecs, n_pvrtc, max_h=200, max_w=300, max_log2_pvrtc=5):
    """Yields (codec, comps, swap, strategy, h, w, pad, img): random geometry, content and format."""
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(n_block_codecs):
        codec, comps, swap, strategy = SOAK_FORMATS[int(rng.integers(0, len(SOAK_FORMATS)))]
        h, w, pad = int(rng.integers(1, max_h)), int(rng.integers(1, max_w)), int(rng.integers(0, 9))
        yield codec, comps, swap, strategy, h, w, pad, soak_image(rng, h, w, comps)
    for _ in range(n_pvrtc):  # PVRTC: square power-of-two RGBA images
        size = 8 << int(rng.integers(0, max_log2_pvrtc + 1))
        swap = int(rng.integers(0, 2))
        yield PVRTC2, 4, swap, 0, size, size, 0, soak_image(rng, size, size, 4)
