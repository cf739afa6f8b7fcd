"""Worker of tests/test_gpu_parity.py::test_nccl_sharded_encode_and_gather_on_real_gpus (one rank per GPU, RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ic_amd_loader  # noqa: E402
import ic_testlib as T  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    pkg = ic_amd_loader.load_package()
    from image_compression_amd import sharding
    ok = True
    # r06: the library's own collective next to torch.distributed's -- icamd_gather_blocks_rccl on its own ncclComm_t (equal and
    # unequal counts take the same grouped send / receive), checked against the oracle like the other
    rccl = sharding.make_rccl_gather(pkg, rank, world, dev)
    for codec, comps, size, n in ((T.ETC1, 3, 256, 4 * world), (T.DXT1, 4, 128, 4 * world + 1), (T.PVRTC2, 4, 64, 2 * world)):
        batch = np.stack([T.s_mixed(size, size, comps, index=i) for i in range(n)])
        counts = [e - b for b, e in (sharding.texture_range(n, world, r) for r in range(world))]
        b, e = sharding.texture_range(n, world, rank)
        local_out = pkg.encode_device(codec, torch.from_numpy(batch[b:e]).to(dev), size, size, comps, n_images=e - b)
        bufs = sharding.alloc_gather_buffers(local_out, counts, rank)
        sharding.gather_to_root(local_out, bufs, counts, rank)
        torch.cuda.synchronize()
        if rank == 0:
            got = torch.cat(bufs).cpu().numpy()
            for i in range(n):
                ok &= got[i].tobytes() == T.oracle_encode(codec, batch[i], size, size, comps)
        bufs_c = sharding.alloc_gather_buffers(local_out, counts, rank)
        if rank == 0:
            for t in bufs_c:
                t.fill_(0xA5)
        sharding.gather_to_root(local_out, bufs_c, counts, rank, rccl=rccl)
        torch.cuda.synchronize()
        if rank == 0:
            ok &= torch.cat(bufs_c).equal(torch.cat(bufs))
        if n % world == 0:  # equal counts: the all-gather form as well
            allg = sharding.gather_output(local_out, world)
            torch.cuda.synchronize()
            ok &= allg.reshape(n, -1)[b:e].equal(local_out)
    rccl.destroy()
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0 and int(flag.item()) == 1:
        print("NCCL_WORKER_OK")
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
