"""CPU tier: the C-ABI library builds, loads, exports every function include/ic_amd.h declares, answers the
host-only queries like the reference, and -- without a GPU -- fails LOUDLY instead of falling back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

import ic_testlib as T


@pytest.fixture(scope="module")
def pkg():
    import ic_amd_loader
    return ic_amd_loader.load_package()


def declared_functions():
    text = open(os.path.join(T.ROOT, "include", "ic_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icamd_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkg):
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(pkg.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(pkg.EXPORTS) == names  # the Python binding list is complete too


def test_host_only_queries_match_the_oracle(pkg):
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for (h, w) in [(0, 5), (1, 1), (5, 5), (8, 8), (61, 59), (4096, 4096), (16, 32)]:
                assert pkg.compute_compressed_data_size(compressor, fmt, h, w) == T.oracle_size(compressor, fmt, h, w)
    lib = pkg.lib()
    assert [lib.icamd_supports_format(T.ETC, f) for f in range(4)] == [1, 0, 0, 0]
    assert [lib.icamd_supports_format(T.PVRTC, f) for f in range(4)] == [0, 0, 1, 0]
    assert all(lib.icamd_supports_format(T.DXTC, f) for f in range(4))
    assert pkg.kernel_name(T.DXT1, 4) == "icamd_dxt1_rgba8_kernel"


def test_no_gpu_means_a_loud_error_not_a_cpu_result(pkg):
    if pkg.lib().icamd_device_count() > 0:
        pytest.skip("a GPU is present")
    img = T.s_noise(8, 8, 3)
    with pytest.raises(pkg.BackendError):
        pkg.compress_host(T.DXTC, T.RGB, img.reshape(-1), 8, 8)
    out = np.zeros(32, np.uint8)
    rc = pkg.lib().icamd_compress(T.DXTC, 2, T.RGB, 8, 8, 0, img.ctypes.data, out.ctypes.data, 32)
    assert rc < 0 and not out.any()
    assert b"no HIP device" in pkg.lib().icamd_last_error()
    # argument errors are still reported the reference's way (false), before any device is needed
    assert pkg.lib().icamd_compress(T.DXTC, 2, T.RGB, 0, 8, 0, img.ctypes.data, out.ctypes.data, 32) == 1


def test_header_is_plain_c_and_links_from_c(pkg, tmp_path):
    """include/ic_amd.h is a C header (no C++ or torch types): a C99 translation unit that calls the host-only queries
    compiles with gcc -std=c99 -Wall -Werror -pedantic, links against libic_amd.so and runs without a GPU."""
    import subprocess
    src = tmp_path / "abi_c99.c"
    src.write_text(r'''
#include <stdio.h>
#include "ic_amd.h"
int main(void) {
  size_t n = icamd_compute_compressed_data_size(ICAMD_COMPRESSOR_DXTC, ICAMD_RGB, 61, 59);
  int ok = icamd_supports_format(ICAMD_COMPRESSOR_ETC, ICAMD_RGB) && !icamd_supports_format(ICAMD_COMPRESSOR_ETC, ICAMD_RGBA);
  /* a refused call needs no device: the reference's `false` */
  int rc = icamd_compress(ICAMD_COMPRESSOR_DXTC, ICAMD_ETC_SMALLER_ERROR, ICAMD_RGB, 0, 8, 0, (const uint8_t *)"", (uint8_t *)&n, 8);
  printf("%lu %d %d %lu %s\n", (unsigned long)n, ok, rc, (unsigned long)icamd_encoded_size(ICAMD_PVRTC2, 64, 64), icamd_version());
  return 0;
}
''')
    exe = tmp_path / "abi_c99"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(T.ROOT, "include"),
                           "-o", str(exe), str(src), "-L" + libdir, "-lic_amd", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, timeout=120, check=True).stdout.decode().split()
    assert out[:4] == [str(T.oracle_size(T.DXTC, T.RGB, 61, 59)), "1", "1", str(64 * 64 // 4)]


def test_host_forms_of_create_solid_and_copy_subimage_match_oracle_and_reference(pkg):
    """icamd_create_solid / icamd_copy_subimage (host buffers; byte shuffling, no device needed -- the reference's
    helper.h:522-592) against the oracle and, where it was built, the compiled reference."""
    import numpy as np
    import ic_testlib as T
    rng = np.random.default_rng(7)
    for compressor in (T.DXTC, T.ETC, T.PVRTC):
        for fmt in (T.RGB, T.BGR, T.RGBA, T.BGRA):
            for (h, w) in ((4, 4), (1, 1), (9, 5), (64, 20), (0, 8)):
                color = [int(v) for v in rng.integers(0, 256, 4)]
                got = pkg.create_solid_host(compressor, fmt, h, w, color)
                assert got == T.oracle_create_solid(compressor, fmt, h, w, color), (compressor, fmt, h, w)
                if T.have_ref() and h and w:
                    assert got == T.ref_create_solid(compressor, fmt, h, w, color), (compressor, fmt, h, w)
    for compressor, fmt in ((T.DXTC, T.RGB), (T.DXTC, T.RGBA), (T.ETC, T.RGB), (T.ETC, T.RGBA), (T.PVRTC, T.RGBA)):
        h, w = 21, 30
        ch, cw = 24, 32
        bb = 8 if (compressor == T.ETC or T.comps_of(fmt) == 3) else 16
        blocks = rng.integers(0, 256, (ch // 4) * (cw // 4) * bb, dtype=np.uint8).tobytes()
        for (r, c, sh, sw) in ((0, 0, 24, 32), (4, 8, 8, 12), (20, 28, 4, 4), (8, 0, 0, 32), (0, 0, 28, 32), (4, 4, 21, 8),
                               (2, 0, 4, 4), (24, 32, 0, 0), (0, 28, 4, 8), (4, 4, 0xfffffffc, 4)):
            got = pkg.copy_subimage_host(compressor, fmt, blocks, ch, cw, r, c, sh, sw)
            if sh < 1 << 20:
                assert got == T.oracle_copy_subimage(compressor, fmt, blocks, ch, cw, r, c, sh, sw), (compressor, fmt, r, c, sh, sw)
            else:
                assert got is None  # start + extent wraps 32 bits: refused here (the reference would read out of bounds)


def test_batch_entry_points_refuse_absurd_device_lists_before_starting_a_thread(pkg):
    """one worker thread per device-list entry must not be something a caller can ask 10 000 of; the check
    needs no device (it comes before anything is allocated or started)."""
    lib = pkg.lib()
    n = 10000
    devs = (ctypes.c_int * n)(*([0] * n))
    one = (ctypes.c_void_p * 1)(0)
    st = (ctypes.c_int * 1)(5)
    rc = lib.icamd_compress_batch(T.DXTC, 2, T.RGB, 8, 8, 0, 1, one, one, 32, devs, n, st)
    assert rc == -4 and b"device list" in lib.icamd_last_error()
    rc = lib.icamd_encode_batch_sharded_device(pkg.DXT1, 2, 3, 0, 8, 8, 24, 1, one, one, devs, n, -1, None, 0, st)
    assert rc == -4 and b"device list" in lib.icamd_last_error()
    # a long error text is cut to the fixed buffer, never allocated for (icamd_last_error() is a plain thread-local array)
    assert len(lib.icamd_last_error()) < 256


def test_rccl_entry_points_validate_before_they_need_rccl_or_a_device(pkg):
    """The C gather's argument checks (include/ic_amd.h) answer without a GPU; binding librccl is reported by
    icamd_rccl_available, never by an exception or an abort."""
    lib = pkg.lib()
    assert lib.icamd_rccl_available() in (0, 1)
    counts = (ctypes.c_size_t * 2)(8, 8)
    assert lib.icamd_gather_blocks_rccl(None, 0, 2, 0, counts, None, None, None, None) == -4
    fake = ctypes.c_void_p(1)  # never dereferenced: the checks below come first
    assert lib.icamd_gather_blocks_rccl(fake, 2, 2, 0, counts, None, None, None, None) == -4
    assert lib.icamd_gather_blocks_rccl(fake, 0, 2, 5, counts, None, None, None, None) == -4
    assert lib.icamd_gather_blocks_rccl(fake, 1, 2, 0, counts, None, None, None, None) == -4   # bytes to send, no d_local
    assert lib.icamd_gather_blocks_rccl(fake, 0, 2, 0, None, None, None, None, None) == -4
    assert lib.icamd_rccl_get_unique_id(None) == -4
    assert lib.icamd_rccl_comm_init(None, 1, 0, None) == -4
    assert lib.icamd_rccl_comm_destroy(None) == 0
