"""CPU tier, property-based: random shapes / strides / formats / pad grids.
  * kernel math (host-emulated device code) == oracle, always;
  * oracle == compiled reference, where oracle/_ref exists (build container)."""
import ctypes

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import ic_testlib as T
from test_kernel_math_host import emul, emul_encode  # noqa: F401  (fixture + helper)

CODECS = [(T.DXT1, 3), (T.DXT1, 4), (T.DXT5, 4), (T.ETC1, 3), (T.ETC1, 4)]


def _image(seed, h, w, comps, pad, style):
    g = np.random.Generator(np.random.PCG64(seed))
    if style == 0:
        img = g.integers(0, 256, size=(h, w, comps), dtype=np.uint8)
    elif style == 1:  # few distinct values: flat blocks, ties, 0/255 alphas
        img = g.choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), size=(h, w, comps))
    else:  # low variance around a random base
        base = g.integers(0, 256, size=(1, 1, comps))
        img = np.clip(base + g.integers(-4, 5, size=(h, w, comps)), 0, 255).astype(np.uint8)
    return T.with_row_padding(img, pad)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31), h=st.integers(1, 21), w=st.integers(1, 21), pad=st.integers(0, 9),
       ci=st.integers(0, len(CODECS) - 1), swap=st.integers(0, 1), strategy=st.integers(0, 3),
       gh=st.integers(0, 12), gw=st.integers(0, 12), style=st.integers(0, 2))
def test_kernel_math_equals_oracle_on_random_geometry(emul, seed, h, w, pad, ci, swap, strategy, gh, gw, style):
    codec, comps = CODECS[ci]
    if codec == T.ETC1:
        swap = 0
    src = _image(seed, h, w, comps, pad, style)
    stride = w * comps + pad
    grid_h, grid_w = h + gh, w + gw
    want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, gh=grid_h, gw=grid_w, stride=stride)
    got = emul_encode(emul, codec, src, h, w, comps, swap, strategy, gh=grid_h, gw=grid_w, stride=stride)
    assert got == want


@pytest.mark.ref
@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")
@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**31), h=st.integers(1, 21), w=st.integers(1, 21), pad=st.integers(0, 9),
       case=st.sampled_from([(T.DXTC, T.RGB), (T.DXTC, T.BGR), (T.DXTC, T.RGBA), (T.DXTC, T.BGRA), (T.ETC, T.RGB)]),
       strategy=st.integers(0, 3), gh=st.integers(0, 12), gw=st.integers(0, 12), style=st.integers(0, 2))
def test_oracle_equals_reference_on_random_geometry(seed, h, w, pad, case, strategy, gh, gw, style):
    compressor, fmt = case
    src = _image(seed, h, w, T.comps_of(fmt), pad, style)
    a = T.ref_compress_and_pad(compressor, fmt, src, h, w, h + gh, w + gw, pad, strategy)
    b = T.oracle_compress_and_pad(compressor, fmt, src, h, w, h + gh, w + gw, pad, strategy)
    assert a is not None and a == b


@pytest.mark.ref
@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")
@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31), log2n=st.integers(3, 6), style=st.integers(0, 2))
def test_pvrtc_oracle_equals_reference_random(seed, log2n, style):
    n = 1 << log2n
    src = _image(seed, n, n, 4, 0, style)
    assert T.ref_compress(T.PVRTC, T.RGBA, src, n, n) == T.oracle_compress(T.PVRTC, T.RGBA, src, n, n)


@settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31), log2n=st.integers(3, 6), style=st.integers(0, 2))
def test_pvrtc_kernel_math_equals_oracle_random(emul, seed, log2n, style):
    n = 1 << log2n
    src = _image(seed, n, n, 4, 0, style)
    assert emul_encode(emul, T.PVRTC2, src, n, n, 4) == T.oracle_encode(T.PVRTC2, src, n, n, 4)
