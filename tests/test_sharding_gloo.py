"""N>1 path on CPU: world_size-2 `gloo` process group.  The sharding / gather logic of
image-compression_amd/sharding.py is exercised with the ORACLE standing in for the device encoder (tests only);
the assembled result must equal the oracle's output for the whole batch / whole image."""
import pytest
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ic_testlib as T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_sharding():
    import importlib.util
    path = os.path.join(T.ROOT, "image-compression_amd", "sharding.py")
    spec = importlib.util.spec_from_file_location("icamd_sharding", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding()
    ok = True
    # (1) batch of independent textures, ETC1 (config 4 shape in miniature) + PVRTC
    for codec, comps, size in ((T.ETC1, 3, 32), (T.PVRTC2, 4, 32), (T.DXT1, 4, 20)):
        n = 6 if 6 % world == 0 else 2 * world  # equal per-rank counts for the one-collective gather
        batch = np.stack([T.s_mixed(size, size, comps, index=i) for i in range(n)])

        def enc(t):
            arr = t.numpy()
            outs = [np.frombuffer(T.oracle_encode(codec, arr[i], size, size, comps), np.uint8) for i in range(arr.shape[0])]
            return torch.from_numpy(np.stack(outs).copy())
        local, gathered = sh.encode_batch_sharded(enc, torch.from_numpy(batch), world, rank)
        full = np.stack([np.frombuffer(T.oracle_encode(codec, batch[i], size, size, comps), np.uint8) for i in range(n)])
        ok &= bool(np.array_equal(gathered.reshape(n, -1).numpy(), full))
        # gather to rank 0 only
        _, g0 = sh.encode_batch_sharded(enc, torch.from_numpy(batch), world, rank, gather_dst=0)
        if rank == 0:
            ok &= bool(np.array_equal(g0.reshape(n, -1).numpy(), full))
        else:
            ok &= g0 is None
    # (2) one large DXT5 image sharded by block rows, ragged height, row padding
    h, w, comps, pad = 37, 29, 4, 5
    img = T.s_mixed(h, w, comps, index=3)
    src = T.with_row_padding(img, pad)
    stride = w * comps + pad
    geo = sh.slab_geometry(h, w, comps, stride, 16, world, rank)
    slab_src = src[geo["src_offset_bytes"]:]
    # the slab is encoded as its own image; only the LAST slab may be ragged / replicate the bottom edge
    slab = T.oracle_encode(T.DXT5, slab_src, geo["pixel_rows"], w, comps, stride=stride,
                           gh=geo["block_rows"] * 4, gw=w)
    whole = T.oracle_encode(T.DXT5, src, h, w, comps, stride=stride)
    ok &= slab == whole[geo["dst_offset_bytes"]: geo["dst_offset_bytes"] + geo["dst_bytes"]]
    sizes = [sh.slab_geometry(h, w, comps, stride, 16, world, r)["dst_bytes"] for r in range(world)]
    ok &= sum(sizes) == len(whole)
    # (2b) fewer block rows than ranks: the empty slab is an empty tensor, the others still tile the image
    h1, w1 = 3, 21
    img1 = T.s_mixed(h1, w1, 3, index=8)
    flat1 = torch.from_numpy(np.ascontiguousarray(img1).reshape(-1))

    def enc_slab(slab, rows, grid_rows):
        return torch.from_numpy(np.frombuffer(T.oracle_encode(T.ETC1, slab.numpy(), rows, w1, 3, gh=grid_rows, gw=w1), np.uint8).copy())
    blocks1, geo1 = sh.encode_slab(enc_slab, flat1, h1, w1, 3, w1 * 3, 8, world, rank)
    whole1 = T.oracle_encode(T.ETC1, img1, h1, w1, 3)
    ok &= blocks1.numpy().tobytes() == whole1[geo1["dst_offset_bytes"]: geo1["dst_offset_bytes"] + geo1["dst_bytes"]]
    ok &= (blocks1.numel() == 0) == (geo1["block_rows"] == 0)
    # (3) gather_to_root: config 4's texture_range split, equal and unequal per-rank counts, rank-0 receive buffers
    for n in (8, 7):
        counts = [e - b for b, e in (sh.texture_range(n, world, r) for r in range(world))]
        b, e = sh.texture_range(n, world, rank)
        full = torch.arange(n * 5, dtype=torch.uint8).reshape(n, 5)
        local = full[b:e].clone()
        bufs = sh.alloc_gather_buffers(local, counts, rank)
        sh.gather_to_root(local, bufs, counts, rank)
        if rank == 0:
            ok &= bool(torch.equal(torch.cat(bufs), full))
        else:
            ok &= bufs is None
        # the debugging path bench.py uses when several ranks share one GPU
        bufs = sh.alloc_gather_buffers(local, counts, rank)
        sh.gather_to_root(local, bufs, counts, rank, host_staged=True)
        if rank == 0:
            ok &= bool(torch.equal(torch.cat(bufs), full))
    # (4) gather_output inside a sub-group whose root is not global rank 0 (group ranks != global ranks)
    sub = dist.new_group(ranks=list(range(world))[::-1])  # group rank g <-> global rank world-1-g
    local = torch.full((3,), rank, dtype=torch.uint8)
    got = sh.gather_output(local, world, dst=0, group=sub)
    if dist.get_rank(sub) == 0:
        want = torch.stack([torch.full((3,), dist.get_global_rank(sub, g), dtype=torch.uint8) for g in range(world)])
        ok &= got is not None and bool(torch.equal(got, want))
    else:
        ok &= got is None
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    results[rank] = int(flag.item())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])  # 3: unequal texture counts and slab heights on the ranks; 8: the driver's N
def test_sharded_encode_and_gather(world):
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        assert dict(results) == {r: 1 for r in range(world)}


def test_ranges_partition_exactly():
    sh = _load_sharding()
    for n in (1, 7, 8, 128, 1024):
        for world in (1, 2, 3, 4, 8):
            spans = [sh.texture_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_pvrtc_regions_partition_the_block_grid():
    """pvrtc_region: the ranks' Z-order ranges are disjoint rectangles that tile the block grid, and each rectangle's
    blocks are exactly the blocks whose Z index (x in the odd bits, y in the even bits, pvrtc.cc:80-86) is in range."""
    sh = _load_sharding()

    def z_index(bx, by):
        z = 0
        for b in range(16):
            z |= ((by >> b) & 1) << (2 * b) | ((bx >> b) & 1) << (2 * b + 1)
        return z
    for size in (8, 16, 64, 512):
        bw, bh = size // 8, size // 4
        for world in (1, 2, 4, 8, 16):
            if world > bw * bh:
                continue
            owner = {}
            for rank in range(world):
                g = sh.pvrtc_region(size, world, rank)
                assert g["n_blocks"] * world == bw * bh and g["blocks_w"] * g["blocks_h"] == g["n_blocks"]
                assert g["dst_offset_bytes"] == 8 * g["first_block"] and g["dst_bytes"] == 8 * g["n_blocks"]
                for by in range(g["block_y0"], g["block_y0"] + g["blocks_h"]):
                    for bx in range(g["block_x0"], g["block_x0"] + g["blocks_w"]):
                        assert (bx, by) not in owner and bx < bw and by < bh
                        owner[(bx, by)] = rank
                        assert g["first_block"] <= z_index(bx, by) < g["first_block"] + g["n_blocks"]
            assert len(owner) == bw * bh
    import pytest
    with pytest.raises(ValueError):
        sh.pvrtc_region(64, 3, 0)


# ---- bench.py's one-large-image slab mode (strong scaling + gather into rank 0's final buffer) over gloo, with the
# oracle standing in for the device encoder (tests only): every rank's slab and the gathered image must check out.

class _OraclePkg:
    """The two entry points bench.slab_leg uses, backed by the oracle on CPU tensors."""

    @staticmethod
    def encoded_size(codec, h, w):
        return int(T.oracle().ico_encoded_size(codec, h, w))

    @staticmethod
    def kernel_name(codec, comps):
        return "oracle-stand-in"

    @staticmethod
    def encode_device(codec, src, height, width, comps, *, n_images=1, out=None, stream=None, etc_strategy=2, **kw):
        assert n_images == 1 and not src.is_cuda
        got = T.oracle_encode(codec, src.numpy().reshape(-1)[:height * width * comps], height, width, comps, 0, etc_strategy)
        out.view(-1)[:len(got)].copy_(torch.from_numpy(np.frombuffer(got, np.uint8).copy()))
        return out


def _slab_worker(rank, world, port, results):
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, T.ROOT)
    import bench
    sh = _load_sharding()
    ctx = bench.Ctx(torch, dist, rank, world, torch.device("cpu"), True, "gloo")
    ok = True
    # what the line says about the gather (r05): measured inbound rate of rank 0 -> gather_bound_GBps / ceiling in every leg
    probe = bench.link_probe(ctx, mib=1, reps=1)
    ok &= probe["peers"] == world - 1 and len(probe["per_peer_alone_GBps"]) == world - 1 and probe["payload_intact"] \
        and probe["all_peers_at_once_GBps_into_rank0"] > 0 and probe["xgmi_links_into_rank0"] == min(world - 1, 7)
    for workload, size, content in (("dxt1_rgba8", 64, "noise"), ("dxt5_rgba8", 40, "smooth"), ("dxt1_rgb888", 12, "flat")):
        res = bench.slab_leg(ctx, _OraclePkg, sh, workload, size, 2, content, probe=probe, rotate_max=6)
        good = res["parity"].startswith("bit-exact") and res.get("value_with_gather") is not None and res["scaling"] == "strong" \
            and sum(res["slab_block_rows"]) == (size + 3) // 4 and res["distinct_images_rotated"] == 6 \
            and res["gather_ranks"] == world and res["gather_bound_GBps"] == probe["all_peers_at_once_GBps_into_rank0"] \
            and res["value_with_gather_ceiling"] > 0
        if not good:
            print("slab_leg failed on rank %d: %r" % (rank, res), file=sys.stderr, flush=True)
        ok &= good
    # a corrupted encoder must be caught by the slab check
    class Bad(_OraclePkg):
        @staticmethod
        def encode_device(*a, **kw):
            out = _OraclePkg.encode_device(*a, **kw)
            if rank == world - 1:
                out.view(-1)[0] ^= 1
            return out
    res = bench.slab_leg(ctx, Bad, sh, "dxt1_rgba8", 64, 1, "noise", rotate_max=5)
    ok &= res["parity"].startswith("MISMATCH")
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    results[rank] = int(flag.item())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])  # 3: unequal slab heights -> the batched isend / irecv gather; 8: the driver's N
def test_bench_slab_mode_over_gloo(world):
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_slab_worker, args=(world, port, results), nprocs=world, join=True)
        assert dict(results) == {r: 1 for r in range(world)}


def _rccl_setup_worker(rank, world, port, results):
    """r06: sharding.make_rccl_gather on ranks WITHOUT a GPU -- every rank must come out of it with the same exception after the
    same sequence of collectives (id broadcast, agreement), none left waiting inside ncclCommInitRank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ic_amd_loader
    pkg = ic_amd_loader.load_package()
    sh = _load_sharding()
    try:
        sh.make_rccl_gather(pkg, rank, world, torch.device("cpu"))
        outcome = "created"
    except pkg.BackendError as e:
        outcome = "BackendError"
    # still in step: one more collective after the failed set-up must complete on every rank
    flag = torch.tensor([1 if outcome == "BackendError" else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    results[rank] = (outcome, int(flag.item()))
    dist.destroy_process_group()


def test_rccl_gather_setup_fails_in_step_on_every_rank_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the set-up would succeed (covered by the gpu tier)")
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_rccl_setup_worker, args=(2, port, results), nprocs=2, join=True)
        assert dict(results) == {0: ("BackendError", 1), 1: ("BackendError", 1)}
