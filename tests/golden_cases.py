"""Iterators over the committed golden fixtures (tests/golden/, generated from the
compiled reference by tests/golden/make_golden.py).  Used by test_golden.py with
the oracle (CPU, `-m "not gpu"`) and by test_gpu_parity.py with the HIP path."""
import hashlib
import json
import os

import numpy as np

import ic_testlib as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMT_NAMES = {T.RGB: "rgb", T.BGR: "bgr", T.RGBA: "rgba", T.BGRA: "bgra"}
COMP_NAMES = {T.DXTC: "dxtc", T.ETC: "etc", T.PVRTC: "pvrtc"}


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_bin(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def check_kats(compress, compress_and_pad):
    """compress(compressor, fmt, src_bytes(np.uint8 1-D), h, w, pad, strategy) -> bytes|None."""
    n = 0
    for k in load("kat.json"):
        src = np.frombuffer(bytes.fromhex(k["input_hex"]), np.uint8)
        if k["and_pad"]:
            got = compress_and_pad(k["compressor"], k["format"], src, k["h"], k["w"], k["and_pad"][0],
                                   k["and_pad"][1], k["pad"], k["strategy"])
        else:
            got = compress(k["compressor"], k["format"], src, k["h"], k["w"], k["pad"], k["strategy"])
        assert got is not None and got.hex() == k["output_hex"], k["name"]
        n += 1
    return n


def check_hashes(compress, compress_and_pad, max_pixels=None, only=None):
    n = 0
    for c in load("hashes.json"):
        if max_pixels is not None and c["h"] * c["w"] > max_pixels:
            continue
        if only is not None and not only(c):
            continue
        img = T.GENERATORS[c["gen"]](c["h"], c["w"], T.comps_of(c["format"]), index=c["index"])
        assert hashlib.sha256(img.tobytes()).hexdigest() == c["input_sha256"], "synthetic generator drifted"
        src = T.with_row_padding(img, c["pad"])
        if c.get("and_pad"):
            got = compress_and_pad(c["compressor"], c["format"], src, c["h"], c["w"], c["and_pad"][0],
                                   c["and_pad"][1], c["pad"], c["strategy"])
        else:
            got = compress(c["compressor"], c["format"], src, c["h"], c["w"], c["pad"], c["strategy"])
        assert got is not None and hashlib.sha256(got).hexdigest() == c["sha256"], c
        n += 1
    return n


def check_mixed64(compress):
    n = 0
    for fn in sorted(os.listdir(GOLDEN)):
        if not fn.startswith("mixed64_"):
            continue
        _, cn, fname, s = fn[:-4].split("_")
        compressor = {v: k for k, v in COMP_NAMES.items()}[cn]
        fmt = {v: k for k, v in FMT_NAMES.items()}[fname]
        strategy = int(s[1:])
        img = T.s_mixed(64, 64, T.comps_of(fmt), index=128)
        got = compress(compressor, fmt, img.reshape(-1), 64, 64, 0, strategy)
        with open(os.path.join(GOLDEN, fn), "rb") as f:
            want = f.read()
        if got != want:
            a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
            bad = np.nonzero(a != b)[0]
            raise AssertionError("%s: %d bytes differ, first at %d" % (fn, bad.size, bad[0]))
        n += 1
    return n
