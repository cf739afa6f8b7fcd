"""GPU tier (-m gpu), round 6: the C ABI under failure (no path from a C caller to std::terminate) and the library's own
RCCL gather (icamd_gather_blocks_rccl) in a world of one rank -- what a 1-GPU box can execute of it."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ic_testlib as T

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PRELOAD = os.path.join(HERE, "cxx", "build", "libfail_pthread.so")


@pytest.fixture(scope="module")
def pkg():
    import torch
    import ic_amd_loader
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return ic_amd_loader.load_package()


def _child(code, env=None, timeout=600):
    head = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n" % (T.ROOT, HERE)
    return subprocess.run([sys.executable, "-c", head + code], capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})))


THREAD_FAILURE = r'''
import ctypes, numpy as np, torch, ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
hook = ctypes.CDLL(%r)
hook.icamd_test_fail_pthread_after.argtypes = [ctypes.c_int]
L = pkg.lib()
n, h, w = 9, 64, 72
imgs = [T.s_mixed(h, w, 3, index=i) for i in range(n)]
want = [T.oracle_compress(T.DXTC, T.RGB, im, h, w) for im in imgs]
devices = [0, 0, 0]
assert pkg.compress_batch_host(T.DXTC, T.RGB, imgs, h, w, devices) == want        # warm-up: contexts, pooled stagings

def batch(allowed):
    size = len(want[0])
    outs = [np.zeros(size, np.uint8) for _ in range(n)]
    ins = (ctypes.c_void_p * n)(*[im.ctypes.data for im in imgs])
    ops = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    devs = (ctypes.c_int * 3)(*devices)
    st = (ctypes.c_int * n)(*([7] * n))
    hook.icamd_test_fail_pthread_after(allowed)
    rc = L.icamd_compress_batch(T.DXTC, 2, T.RGB, h, w, 0, n, ins, ops, size, devs, 3, st)
    hook.icamd_test_fail_pthread_after(-1)
    return rc, list(st), [o.tobytes() for o in outs], L.icamd_last_error().decode()

# the second worker cannot be created: worker 0's images are done and right, the others report ICAMD_ERR_ALLOC, the call returns
for allowed in (1, 0, 2):
    rc, st, outs, err = batch(allowed)
    assert rc == -3, (allowed, rc, st, err)
    for i in range(n):
        if i %% 3 < allowed:
            assert st[i] == 0 and outs[i] == want[i], (allowed, i, st)
        else:
            assert st[i] == -3, (allowed, i, st)
    assert "worker thread" in err, err
assert hook.icamd_test_pthread_refusals() == 3
# ... and the library is still usable afterwards
assert pkg.compress_batch_host(T.DXTC, T.RGB, imgs, h, w, devices) == want

# the device-resident twin
srcs = [torch.from_numpy(im).cuda() for im in imgs]
size = len(want[0])
dsts = [torch.zeros(size, dtype=torch.uint8, device="cuda") for _ in range(n)]
sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
devs = (ctypes.c_int * 3)(0, 0, 0)
st = (ctypes.c_int * n)()
assert L.icamd_encode_batch_sharded_device(pkg.DXT1, 2, 3, 0, h, w, w * 3, n, sp, dp, devs, 3, -1, None, 0, st) == 0   # warm-up
hook.icamd_test_fail_pthread_after(1)
rc = L.icamd_encode_batch_sharded_device(pkg.DXT1, 2, 3, 0, h, w, w * 3, n, sp, dp, devs, 3, -1, None, 0, st)
hook.icamd_test_fail_pthread_after(-1)
torch.cuda.synchronize()
assert rc == -3 and [s for s in st] == [0 if i %% 3 == 0 else -3 for i in range(n)], (rc, list(st))
assert all(dsts[i].cpu().numpy().tobytes() == want[i] for i in range(0, n, 3))
assert "worker thread" in L.icamd_last_error().decode()
print("THREAD_FAILURE_OK")
'''


def test_a_worker_thread_that_cannot_be_created_is_a_status_not_a_terminate(pkg):
    """std::thread's constructor throws std::system_error when pthread_create fails; with workers already
    running that used to destroy joinable threads (std::terminate inside a C caller).  pthread_create is made to fail through an
    LD_PRELOAD interposer (the tests run as root, RLIMIT_NPROC does not bind): the started workers finish and are joined, their
    images are right, the others report ICAMD_ERR_ALLOC, the process lives on."""
    assert os.path.exists(PRELOAD), "tests/cxx/build/libfail_pthread.so is not built (make -C tests/cxx)"
    p = _child(THREAD_FAILURE % PRELOAD, env={"LD_PRELOAD": PRELOAD})
    assert p.returncode == 0 and "THREAD_FAILURE_OK" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


RCCL_ONE_RANK = r'''
import ctypes, numpy as np, torch, ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
L = pkg.lib()
assert L.icamd_rccl_available() == 1, L.icamd_last_error()
torch.cuda.set_device(0)
g = pkg.RcclGather(0, 1, lambda raw: raw)     # world of one rank: the id needs no transport
size = 256
imgs = np.stack([T.s_mixed(size, size, 4, index=i) for i in range(3)])
local = pkg.encode_device(pkg.DXT1, torch.from_numpy(imgs).cuda(), size, size, 4, n_images=3)
want = b"".join(T.oracle_encode(pkg.DXT1, imgs[i], size, size, 4) for i in range(3))
side = torch.cuda.Stream()
done = torch.cuda.Event()
done.record(torch.cuda.current_stream())
# (a) into a separate buffer, on a side stream, default offsets (prefix sums)
root = torch.full((local.numel() + 64,), 0xA5, dtype=torch.uint8, device="cuda")
counts = (ctypes.c_size_t * 1)(local.numel())
with torch.cuda.stream(side):
    side.wait_event(done)
    st = L.icamd_gather_blocks_rccl(g.comm, 0, 1, 0, counts, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(root.data_ptr()), None,
                                    ctypes.c_void_p(side.cuda_stream))
assert st == 0, L.icamd_last_error()
side.synchronize()
got = root.cpu().numpy()
assert got[:local.numel()].tobytes() == want and (got[local.numel():] == 0xA5).all()
# (b) explicit offset; (c) in place (d_local already is the slot): nothing to do, still ICAMD_OK
root.fill_(0x5A)
offs = (ctypes.c_size_t * 1)(32)
assert L.icamd_gather_blocks_rccl(g.comm, 0, 1, 0, counts, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(root.data_ptr()), offs, None) == 0
torch.cuda.synchronize()
got = root.cpu().numpy()
assert got[32:32 + local.numel()].tobytes() == want and (got[:32] == 0x5A).all() and (got[32 + local.numel():] == 0x5A).all()
assert L.icamd_gather_blocks_rccl(g.comm, 0, 1, 0, counts, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(local.data_ptr()), None, None) == 0
# (d) the Python wrapper the bench uses, through sharding.gather_to_root
from image_compression_amd import sharding
import os
os.environ["ICAMD_FORCE_COLLECTIVES"] = "1"
bufs = sharding.alloc_gather_buffers(local, [3], 0)
bufs[0].fill_(0)
sharding.gather_to_root(local, bufs, [3], 0, rccl=g)
torch.cuda.synchronize()
assert bufs[0].cpu().numpy().tobytes() == want
# argument errors
assert L.icamd_gather_blocks_rccl(g.comm, 1, 1, 0, counts, None, None, None, None) == -4
assert L.icamd_gather_blocks_rccl(g.comm, 0, 1, 0, counts, ctypes.c_void_p(local.data_ptr()), None, None, None) == -4
assert L.icamd_gather_blocks_rccl(None, 0, 1, 0, counts, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(root.data_ptr()), None, None) == -4
zero = (ctypes.c_size_t * 1)(0)
assert L.icamd_gather_blocks_rccl(g.comm, 0, 1, 0, zero, None, None, None, None) == 0
g.destroy()
print("RCCL_ONE_RANK_OK", os.environ.get("ICAMD_RCCL_SELF_SENDRECV", "0"))
'''


@pytest.mark.parametrize("self_sendrecv", ["0", "1"])
def test_c_rccl_gather_in_a_world_of_one_rank(pkg, self_sendrecv):
    """icamd_rccl_get_unique_id / comm_init / icamd_gather_blocks_rccl / comm_destroy on the box's one GPU: librccl bound by dlopen
    (PyTorch's copy, already in the process), a communicator of one rank, root's own range copied into the gather buffer on a
    side stream; with ICAMD_RCCL_SELF_SENDRECV=1 the same bytes travel through ncclSend / ncclRecv to the rank itself -- RCCL's
    send / receive kernel path, not only its set-up.  Between two GPUs this has never run (the pool's boxes have one):
    tests/nccl_worker.py does the same at any world size."""
    p = _child(RCCL_ONE_RANK, env={"ICAMD_RCCL_SELF_SENDRECV": self_sendrecv, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert p.returncode == 0 and ("RCCL_ONE_RANK_OK %s" % self_sendrecv) in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


def test_rccl_binding_without_torch_in_the_process(pkg):
    """A C caller has no PyTorch: librccl comes from the loader's search path (ROCm's lib directory is on libic_amd.so's rpath)."""
    code = r'''
import ctypes
L = ctypes.CDLL(%r)
L.icamd_last_error.restype = ctypes.c_char_p
assert L.icamd_rccl_available() == 1, L.icamd_last_error()
uid = (ctypes.c_uint8 * 128)()
assert L.icamd_rccl_get_unique_id(uid) == 0, L.icamd_last_error()
comm = ctypes.c_void_p()
assert L.icamd_rccl_comm_init(ctypes.byref(comm), 1, 0, uid) == 0, L.icamd_last_error()
assert L.icamd_rccl_comm_destroy(comm) == 0
print("PLAIN_C_RCCL_OK")
''' % pkg.LIB_PATH
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0 and "PLAIN_C_RCCL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_single_image_strides_are_checked_like_batches(pkg):
    """a non-zero image stride is validated for one image too, in all four batch block operations, and the Python
    wrappers refuse a [1, bytes] tensor that is smaller than the source grid before any kernel reads it."""
    import torch
    blocks = torch.zeros((1, 8 * 4 * 4), dtype=torch.uint8, device="cuda")  # a 16 x 16 DXT1 grid
    small = blocks[:, :64].contiguous()
    with pytest.raises(ValueError):
        pkg.pad_batch_device(T.DXTC, T.RGB, small, 16, 16, 24, 24)
    with pytest.raises(ValueError):
        pkg.copy_subimage_batch_device(T.DXTC, T.RGB, small, 16, 16, 0, 0, 8, 8)
    L = pkg.lib()
    import ctypes
    out = torch.zeros((1, 8 * 6 * 6), dtype=torch.uint8, device="cuda")
    p, o = ctypes.c_void_p(small.data_ptr()), ctypes.c_void_p(out.data_ptr())
    assert L.icamd_pad_batch_device(T.DXTC, 2, T.RGB, 16, 16, 1, p, 64, 24, 24, o, out.shape[1], 8 * 6 * 6, None) == -4
    assert L.icamd_pad_batch_device(T.DXTC, 2, T.RGB, 16, 16, 1, p, 128, 24, 24, o, 8, 8 * 6 * 6, None) == -4
    assert L.icamd_copy_subimage_batch_device(T.DXTC, T.RGB, 16, 16, 1, p, 64, 0, 0, 8, 8, o, 32, 32, None) == -4
    assert L.icamd_copy_subimage_batch_device(T.DXTC, T.RGB, 16, 16, 1, p, 128, 0, 0, 8, 8, o, 8, 32, None) == -4
    color = (ctypes.c_uint8 * 4)(1, 2, 3, 4)
    assert L.icamd_create_solid_batch_device(T.DXTC, T.RGB, 16, 16, 1, color, o, 8, 128, None) == -4
    # stride 0 with one image (the single-image entry points pass it) and exact strides still work
    assert pkg.pad_batch_device(T.DXTC, T.RGB, blocks, 16, 16, 24, 24) is not None
    assert L.icamd_pad_batch_device(T.DXTC, 2, T.RGB, 16, 16, 1, ctypes.c_void_p(blocks.data_ptr()), 0, 24, 24, o, 0, 8 * 6 * 6, None) == 0
    torch.cuda.synchronize()


@pytest.fixture
def pvrtc_auto(pkg):
    yield
    pkg.pvrtc_tune(0, -1)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_pvrtc_onepass_halo_form_on_textures_wider_than_a_workgroup(pkg, pvrtc_auto):
    """r06: textures of 8192^2 and more -- a block row is 1 024 lanes and more -- take the one-pass kernel
    in its HALO form (icamd_pvrtc2_onepass_halo_kernel: two or more 512-lane workgroups per block row; the colours of the
    columns either side of every workgroup boundary from a tiny pre-pass, the column-0 values right of a workgroup's last lane
    from its prologue) instead of the morph + encode pair.  Forced at the shortest and the tallest strip, automatic, and the
    pair itself, against the oracle; a batch of two with padded strides; an 8-mod-16 destination (no staged stores)."""
    import hashlib
    import torch
    n = 8192
    img = T.s_smooth(n, n, 4, index=61)
    img[:2048, :3072] = T.s_noise(2048, 3072, 4, index=61)
    img[4096:6144, 4000:4200] = T.s_flat(2048, 200, 4, index=61)  # a flat band across the workgroup boundary at x = 4096
    img[:, 8184:] = T.s_noise(n, 8, 4, index=62)                    # ... and noise either side of the toroidal seam
    img[:, :8] = T.s_noise(n, 8, 4, index=63)
    want = T.oracle_encode(T.PVRTC2, img, n, n, 4)
    d = _dev(img)
    for mode, sb in ((2, 2), (2, 6), (0, -1), (1, -1)):
        assert pkg.pvrtc_tune(mode, sb)
        out = pkg.encode_device(T.PVRTC2, d, n, n, 4)
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == want, (mode, sb)
    # the halo form keeps nothing between kernels either (its prologue reduces the boundary columns itself): such a launch is
    # captured into a HIP graph without a caller-owned workspace, like the plain one-pass kernel's
    assert pkg.pvrtc_tune(2, -1)
    gs = torch.cuda.Stream()
    cap = torch.zeros(n * n // 4, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(gs):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs):
            assert pkg.encode_device(T.PVRTC2, d, n, n, 4, out=cap.view(1, -1), stream=gs) is not None
        g.replay()
        gs.synchronize()
    assert cap.cpu().numpy().tobytes() == want
    del g
    # two textures, padded image strides, destination 8 mod 16
    assert pkg.pvrtc_tune(2, 4)
    img2 = np.ascontiguousarray(img[::-1, ::-1])
    want2 = T.oracle_encode(T.PVRTC2, img2, n, n, 4)
    per_in, per_out = n * n * 4 + 4096, n * n // 4 + 64
    src = torch.zeros(2 * per_in, dtype=torch.uint8, device="cuda")
    src[:n * n * 4] = d.view(-1)
    src[per_in:per_in + n * n * 4] = _dev(img2).view(-1)
    buf = torch.zeros(2 * per_out + 8, dtype=torch.uint8, device="cuda")
    import ctypes
    rc = pkg.lib().icamd_encode_device(pkg.PVRTC2, 0, 4, 0, n, n, n, n, n * 4, 2, per_in, per_out, ctypes.c_void_p(src.data_ptr()),
                                       ctypes.c_void_p(buf.data_ptr() + 8), None)
    assert rc == 0, pkg.lib().icamd_last_error()
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert got[8:8 + n * n // 4].tobytes() == want and got[8 + per_out:8 + per_out + n * n // 4].tobytes() == want2
    assert not got[:8].any() and not got[8 + n * n // 4:8 + per_out].any()


def test_pvrtc_regions_through_the_halo_form_and_through_the_pair(pkg, pvrtc_auto):
    """The multi-GPU split of ONE texture (sharding.pvrtc_region -> icamd_pvrtc2_encode_region_device): regions at least 64
    block columns wide now take the one-pass kernel's halo form.  Every region of 1024^2 ... 4096^2 textures split 2 ... 16 ways,
    forced through the halo form (shortest / tallest strip), left to the selection, and through the pair -- each from a copy of
    the texture in which everything outside region + ring + pixel (0, 0) is garbage -- against the oracle."""
    from image_compression_amd import sharding
    import torch
    rng = np.random.Generator(np.random.PCG64(0x6A))
    for size, worlds in ((1024, (2, 4)), (2048, (2, 8)), (4096, (8, 16))):
        img = T.soak_image(rng, size, size, 4)
        want = T.oracle_encode(T.PVRTC2, img, size, size, 4, threads=8)
        bw, bh = size // 8, size // 4
        for world in worlds:
            locals_ = []
            for rank in range(world):
                g = sharding.pvrtc_region(size, world, rank)
                rows, cols = np.zeros(size, bool), np.zeros(size, bool)
                for j in range(g["blocks_h"] + 2):
                    by = (g["block_y0"] - 1 + j) % bh
                    rows[by * 4:by * 4 + 4] = True
                for i in range(g["blocks_w"] + 2):
                    bx = (g["block_x0"] - 1 + i) % bw
                    cols[bx * 8:bx * 8 + 8] = True
                keep = np.outer(rows, cols)
                keep[0, 0] = True
                locals_.append((g, _dev(np.where(keep[..., None], img, rng.integers(0, 256, img.shape, dtype=np.uint8)))))
            for mode, sb in ((2, 2), (2, 6), (0, -1), (1, -1)):
                assert pkg.pvrtc_tune(mode, sb)
                got = bytearray(len(want))
                for g, local in locals_:
                    out = pkg.pvrtc_encode_region_device(local, size, g["first_block"], g["n_blocks"])
                    torch.cuda.synchronize()
                    got[g["dst_offset_bytes"]:g["dst_offset_bytes"] + g["dst_bytes"]] = out.cpu().numpy().tobytes()
                assert bytes(got) == want, (size, world, mode, sb)
