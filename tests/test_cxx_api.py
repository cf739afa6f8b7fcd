"""The C++ drop-in classes (image-compression_amd/cxx: DxtcCompressor / EtcCompressor / PvrtcCompressor with the
reference's public API) driven by tests/cxx/api_driver.cc.  The same source is compiled against the reference
(api_driver_ref, build container only) and against our library (api_driver_amd); the transcripts must be identical.
tests/golden/api_driver_ref.txt is the committed transcript of api_driver_ref."""
import os
import subprocess

import pytest

import ic_testlib as T

BUILD = os.path.join(T.ROOT, "tests", "cxx", "build")
GOLDEN = os.path.join(T.ROOT, "tests", "golden", "api_driver_ref.txt")


def _run(exe):
    return subprocess.run([os.path.join(BUILD, exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                          check=True).stdout.decode()


@pytest.mark.ref
@pytest.mark.skipif(not (os.path.isdir("/root/reference") and T.have_ref()), reason="needs the reference build")
def test_reference_transcript_is_the_committed_golden():
    subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "tests", "cxx"), "ref"], stdout=subprocess.DEVNULL)
    assert _run("api_driver_ref") == open(GOLDEN).read()


def test_cxx_headers_compile_and_link_without_gpu():
    # the drop-in headers must at least build into a program here (no GPU needed for that)
    subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "image-compression_amd")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "tests", "cxx"), os.path.join(BUILD, "api_driver_amd")],
                          stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(BUILD, "api_driver_amd"))


@pytest.mark.gpu
def test_cxx_api_transcript_matches_reference():
    if not os.path.exists(os.path.join(BUILD, "api_driver_amd")):
        subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "image-compression_amd")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "tests", "cxx"), os.path.join(BUILD, "api_driver_amd")],
                              stdout=subprocess.DEVNULL)
    got = _run("api_driver_amd")
    want = open(GOLDEN).read()
    if got != want:
        g, w = got.splitlines(), want.splitlines()
        for i, (a, b) in enumerate(zip(g, w)):
            assert a == b, "line %d:\n  ours: %s\n  ref:  %s" % (i + 1, a, b)
        assert len(g) == len(w)


@pytest.mark.gpu
def test_cxx_pvrtc_decompress_opt_in_extension():
    """PvrtcCompressor::Decompress returns false like the reference (pvrtc_compressor.cc:669-672) unless
    ICAMD_PVRTC_DECOMPRESS_EXTENSION=1 opts into the (parity-unpinned) decoder: with it, the only transcript lines
    that change are PVRTC `decompress -> false` lines turning into `decompress -> true hash=...`."""
    exe = os.path.join(BUILD, "api_driver_amd")
    base = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600, check=True).stdout.decode().splitlines()
    env = dict(os.environ, ICAMD_PVRTC_DECOMPRESS_EXTENSION="1")
    ext = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600, check=True, env=env).stdout.decode().splitlines()
    assert len(base) == len(ext)
    changed = [(a, b) for a, b in zip(base, ext) if a != b]
    assert changed and all(a == "  decompress -> false" and b.startswith("  decompress -> true hash=") for a, b in changed)


@pytest.mark.gpu
def test_cxx_device_resident_extension_equals_the_host_drop_in():
    """r05: CompressDevice / CompressAndPadDevice / CompressBatchDevice on DxtcCompressor / EtcCompressor / PvrtcCompressor
    (an extension, compressor.h) leave in HBM exactly the bytes the host-buffer drop-in returns -- whose transcript is pinned
    against the reference above -- for every format / strategy / ragged shape / padded grid, and refuse what it refuses."""
    exe = os.path.join(BUILD, "device_driver_amd")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(T.ROOT, "tests", "cxx"), exe], stdout=subprocess.DEVNULL)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "all equal" in out.splitlines()[-1], out[-3000:]
    assert "DIFFERENT" not in out and int(out.splitlines()[-1].split()[2]) >= 90
