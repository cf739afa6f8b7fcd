"""CPU tier: the two CPU-side restatements -- oracle/ic_oracle.c (the plain-C port of the reference) and
tests/host_emul (the device per-block math compiled for the host) -- built with AddressSanitizer +
UndefinedBehaviorSanitizer and driven over ragged geometries with exactly-sized heap buffers.
Test infrastructure only; nothing here is linked into libic_amd.so."""
import os
import subprocess

import pytest

import ic_testlib as T

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(T.ROOT, "image-compression_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = tmp_path_factory.mktemp("san")
    exe = os.path.join(out, "san_driver")
    oracle_o = os.path.join(out, "ic_oracle.o")
    subprocess.check_call(["gcc"] + SAN + ["-std=c99", "-I" + os.path.join(T.ROOT, "oracle"), "-I" + CSRC, "-c",
                                           os.path.join(T.ROOT, "oracle", "ic_oracle.c"), "-o", oracle_o])
    subprocess.check_call(["g++"] + SAN + ["-std=c++17", "-DICAMD_HOST_EMULATION", "-I" + CSRC,
                                           "-I" + os.path.join(T.ROOT, "oracle"), "-o", exe,
                                           os.path.join(HERE, "sanitize", "san_driver.cc"),
                                           os.path.join(HERE, "host_emul", "emul.cc"), oracle_o, "-lpthread"])
    return exe


def test_oracle_and_host_emulation_are_clean_under_asan_ubsan(driver):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([driver], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-6000:])
    assert "all checks passed" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
