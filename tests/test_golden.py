"""CPU (`-m "not gpu"`): the oracle (oracle/ic_oracle.c) against the committed golden
vectors that tests/golden/make_golden.py generated from the compiled reference.
This is the pin that travels: it runs on boxes where /root/reference is absent."""
import hashlib

import numpy as np

import golden_cases as G
import ic_testlib as T


def _compress(compressor, fmt, src, h, w, pad, strategy):
    return T.oracle_compress(compressor, fmt, src, h, w, pad, strategy)


def _compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy):
    return T.oracle_compress_and_pad(compressor, fmt, src, h, w, ph, pw, pad, strategy)


def test_oracle_known_answers():
    assert G.check_kats(_compress, _compress_and_pad) >= 20


def test_oracle_mixed64_full_bytes():
    assert G.check_mixed64(_compress) == 9


def test_oracle_hashes():
    assert G.check_hashes(_compress, _compress_and_pad) > 100


def test_oracle_decoders_golden():
    for c in G.load("decode_hashes.json"):
        img = T.s_mixed(c["h"], c["w"], T.comps_of(c["format"]), index=c["index"])
        blocks = T.oracle_compress(c["compressor"], c["format"], img.reshape(-1), c["h"], c["w"])
        assert hashlib.sha256(blocks).hexdigest() == c["blocks_sha256"]
        codec = T.ETC1 if c["compressor"] == T.ETC else (T.DXT1 if T.comps_of(c["format"]) == 3 else T.DXT5)
        px = T.oracle_decode(codec, blocks, c["h"], c["w"], swap=int(c["format"] in (T.BGR, T.BGRA)))
        assert hashlib.sha256(px.tobytes()).hexdigest() == c["pixels_sha256"]


def test_oracle_rgba_extension_equals_alpha_stripped_rgb():
    # the RGBA8 -> DXT1/ETC1 extension is defined as "reference output for the alpha-stripped RGB888 image"
    img = T.s_mixed(64, 48, 4, index=3)
    rgb = np.ascontiguousarray(img[..., :3])
    for codec, comp in ((T.DXT1, T.DXTC), (T.ETC1, T.ETC)):
        want = T.oracle_compress(comp, T.RGB, rgb.reshape(-1), 64, 48)
        got = T.oracle_encode(codec, img.reshape(-1), 64, 48, 4)
        assert got == want


def test_oracle_threads_agree():
    img = T.s_noise(128, 64, 4, index=9)
    for codec in (T.DXT1, T.DXT5, T.ETC1):
        assert T.oracle_encode(codec, img, 128, 64, 4, threads=1) == T.oracle_encode(codec, img, 128, 64, 4, threads=5)


def test_oracle_decoders_on_arbitrary_block_words_golden():
    """Seeded random block words (DXT1 3-colour mode, DXT5 6-value alpha, ETC1 differential blocks that leave 0..31)
    decoded by the reference -> committed hashes."""
    for c in G.load("random_decode_hashes.json"):
        blocks = T.random_blocks(c["codec"], c["h"], c["w"], c["seed"])
        assert hashlib.sha256(blocks).hexdigest() == c["blocks_sha256"], "seeded block generator drifted"
        px = T.oracle_decode(c["codec"], blocks, c["h"], c["w"], swap=int(c["format"] in (T.BGR, T.BGRA)))
        assert hashlib.sha256(px.tobytes()).hexdigest() == c["pixels_sha256"], c


def test_const_colour_table_pinned():
    """The one data file the oracle and the product both compile in (dxtc_const_table.inc): its 2 048 values hash to the
    pinned SHA-256, and -- independently of that file -- the oracle's encoding of every table row equals the bytes the
    compiled reference produced (tests/golden/solid_ramps_dxt1_*.bin)."""
    assert hashlib.sha256(T.const_table_bytes()).hexdigest() == T.CONST_TABLE_SHA256
    img = T.solid_ramp_image()
    for fmt in (T.RGB, T.BGR):
        got = T.oracle_compress(T.DXTC, fmt, img.reshape(-1), img.shape[0], img.shape[1])
        assert got == G.load_bin("solid_ramps_dxt1_%s.bin" % G.FMT_NAMES[fmt])
