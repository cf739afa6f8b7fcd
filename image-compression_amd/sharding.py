"""Multi-GPU sharding of the block-encode path: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on MI355X; "gloo" in the CPU test tier).

Every 4x4 (DXT/ETC) block and every PVRTC texture is independent, so there is NO collective on the data path:
  * one PVRTC texture can still be split by Z-order range (`pvrtc_region`): each rank recomputes the reduced colours
    of a one-block ring around its rectangle instead of exchanging them;
  * a batch of textures is split into contiguous per-rank ranges (`texture_range`);
  * one large DXT/ETC image is split into contiguous slabs of block rows (`block_row_range`); blocks are stored
    row-major (reference compressor4x4_helper.h:202-214), so each slab's output is one contiguous byte range.
The only exchange is the optional gather of the compressed output (1/6..1/8 of the input bytes), `gather_output`.
"""
import torch
import torch.distributed as dist


def texture_range(n_textures, world_size, rank):
    """Contiguous, balanced split of [0, n_textures) -> (begin, end) for `rank`."""
    return n_textures * rank // world_size, n_textures * (rank + 1) // world_size


def block_row_range(block_rows, world_size, rank):
    """Contiguous, balanced split of a single image's block rows -> (begin, end)."""
    return block_rows * rank // world_size, block_rows * (rank + 1) // world_size


def slab_geometry(height, width, components, row_stride_bytes, block_bytes, world_size, rank):
    """For one DXT/ETC image sharded by block rows: what this rank reads and where its blocks go.
    Returns dict(pixel_row0, pixel_rows, src_offset_bytes, dst_offset_bytes, dst_bytes).  The last slab keeps the
    image's ragged bottom edge, so edge replication (pixel4x4.cc:23-59) is unchanged."""
    rows = (height + 3) // 4
    cols = (width + 3) // 4
    b0, b1 = block_row_range(rows, world_size, rank)
    y0 = b0 * 4
    y1 = min(height, b1 * 4)
    return {"block_row0": b0, "block_rows": b1 - b0, "pixel_row0": y0, "pixel_rows": max(0, y1 - y0),
            "src_offset_bytes": y0 * row_stride_bytes, "dst_offset_bytes": b0 * cols * block_bytes,
            "dst_bytes": (b1 - b0) * cols * block_bytes}


def encode_slab(encode_fn, src, height, width, components, row_stride_bytes, block_bytes, world_size, rank):
    """This rank's slab of ONE DXT/ETC image: encode_fn(slab_bytes, pixel_rows, grid_rows) -> uint8 tensor of the slab's
    blocks, called with the rank's byte range of `src` (a flat uint8 tensor).  Ranks whose share is empty (fewer block
    rows than ranks) get an empty tensor instead of a refused zero-height encode.  Returns (blocks, geometry)."""
    geo = slab_geometry(height, width, components, row_stride_bytes, block_bytes, world_size, rank)
    if geo["pixel_rows"] == 0:
        return torch.empty((0,), dtype=torch.uint8, device=src.device), geo
    return encode_fn(src[geo["src_offset_bytes"]:], geo["pixel_rows"], geo["block_rows"] * 4), geo


def pvrtc_region(size, world_size, rank):
    """One PVRTC 2 bpp texture (size x size) sharded over `world_size` (a power of two) ranks by Z-order range: rank r
    owns blocks [r * n, (r + 1) * n) of the output, n = blocks / world_size -- a rectangle of the block grid
    (reference pvrtc_compressor.cc:80-86: x in the odd bits, y in the even bits of the block index) and one contiguous
    byte range of the final buffer.  Returns dict(first_block, n_blocks, dst_offset_bytes, dst_bytes, block_x0,
    block_y0, blocks_w, blocks_h): what to pass to icamd_pvrtc2_encode_region_device, where the bytes go, and which
    blocks (plus a one-block toroidal ring, plus pixel (0, 0)) the rank has to hold."""
    blocks = (size // 8) * (size // 4)
    if world_size < 1 or world_size & (world_size - 1) or world_size > blocks:
        raise ValueError("world_size must be a power of two not larger than the block count")
    n = blocks // world_size
    first = rank * n

    def compact(v):  # every other bit of v, starting with bit 0
        out, bit = 0, 0
        while v:
            out |= (v & 1) << bit
            v >>= 2
            bit += 1
        return out
    m = n.bit_length() - 1
    return {"first_block": first, "n_blocks": n, "dst_offset_bytes": first * 8, "dst_bytes": n * 8,
            "block_x0": compact(first >> 1), "block_y0": compact(first), "blocks_w": 1 << (m // 2),
            "blocks_h": 1 << (m - m // 2)}


def _force_collectives():
    """ICAMD_FORCE_COLLECTIVES=1: issue the collectives even in a world of one rank -- how the pool's 1-GPU boxes get to
    execute the RCCL code path at all (communicator set-up, device-tensor gather / all-gather, stream ordering)."""
    import os
    return os.environ.get("ICAMD_FORCE_COLLECTIVES") == "1"


def _global_rank(group, group_rank):
    """torch.distributed addresses peers (dst= / src=) by GLOBAL rank, even inside a sub-group."""
    if group is None or group is dist.group.WORLD:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def gather_output(local, world_size, dst=None, group=None):
    """Gathers equally sized per-rank compressed buffers.  dst=None: all-gather (every rank gets
    [world, ...]); dst=r (a rank of `group`): only that rank receives (others get None).  One collective, after the
    encode."""
    if world_size == 1 and not _force_collectives():
        return local.unsqueeze(0)
    if dst is None:
        local = local.contiguous()
        flat = torch.empty((world_size * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)  # concatenation along dim 0: the layout both RCCL and gloo accept
        dist.all_gather_into_tensor(flat, local, group=group)
        return flat.view((world_size,) + tuple(local.shape))
    rank = dist.get_rank(group)  # rank inside `group`, the space `dst` is given in
    bufs = [torch.empty_like(local) for _ in range(world_size)] if rank == dst else None
    dist.gather(local.contiguous(), bufs, dst=_global_rank(group, dst), group=group)
    return torch.stack(bufs) if rank == dst else None


def alloc_gather_buffers(local, counts, rank, dst=0):
    """Receive buffers on `dst` for gather_to_root: one [counts[r], ...] tensor per rank (None elsewhere)."""
    if rank != dst:
        return None
    return [torch.empty((c,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) for c in counts]


def make_rccl_gather(pkg, rank, world, device):
    """The library's own communicator for gather_to_root(..., rccl=...): rank 0's ncclUniqueId goes to the others through the
    default process group (a 128-byte broadcast: on the device with the nccl backend, on the host with gloo)."""
    on_device = dist.get_backend() == "nccl"

    def broadcast_bytes(raw):
        t = torch.zeros(pkg.RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8)
        if raw is not None:
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
        if on_device:
            t = t.to(device)
        if world > 1 or _force_collectives():
            dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def agree(ok):
        if not (world > 1 or _force_collectives()):
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if on_device:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))
    return pkg.RcclGather(rank, world, broadcast_bytes, agree)


def gather_to_root(local, bufs, counts, rank, dst=0, group=None, host_staged=False, rccl=None):
    """Gathers per-rank compressed slabs (rank r holds counts[r] textures' worth of blocks) into `bufs` on rank `dst`
    (ranks of `group`).  Enqueued on the CURRENT stream and not synchronised, so the caller can run it on a side
    stream underneath the next batch's encode.
    rccl: an `RcclGather` (make_rccl_gather) -- the LIBRARY's collective, icamd_gather_blocks_rccl: one grouped ncclSend /
    ncclRecv exchange, equal or unequal counts alike, every peer writing its own slab of rank dst's HBM over its own xGMI link.
    Without it, torch.distributed: equal counts one dist.gather, unequal counts (n_textures % world != 0) batched
    point-to-point.  host_staged: the gloo debugging path (device tensors staged through host memory)."""
    world = len(counts)
    if rccl is not None and not host_staged and (world > 1 or _force_collectives()):
        # byte counts: root reads every rank's from its receive buffers (the C entry point only looks at a sender's own count
        # on the sending side), a sender states its own
        if rank == dst:
            counts_bytes = [b.numel() * b.element_size() for b in bufs]
            if counts_bytes[rank] != local.numel() * local.element_size():
                raise ValueError("gather_to_root: root's own buffer does not have the size of its local output")
        else:
            counts_bytes = [0] * world
            counts_bytes[rank] = local.numel() * local.element_size()
        rccl.gather(local.contiguous().view(-1).view(torch.uint8), bufs if rank == dst else None, counts_bytes, root=dst)
        return
    if world == 1 and not _force_collectives():
        bufs[0].copy_(local)
        return
    gdst = _global_rank(group, dst)
    if host_staged:
        h = local.cpu()
        hb = [torch.empty((c,) + tuple(local.shape[1:]), dtype=local.dtype) for c in counts] if rank == dst else None
        if len(set(counts)) == 1:
            dist.gather(h, hb, dst=gdst, group=group)
        else:
            _p2p_gather(h, hb, counts, rank, dst, group)
        if rank == dst:
            for b, x in zip(bufs, hb):
                b.copy_(x)
        return
    if len(set(counts)) == 1:
        dist.gather(local.contiguous(), bufs if rank == dst else None, dst=gdst, group=group)
    else:
        _p2p_gather(local.contiguous(), bufs, counts, rank, dst, group)


def _p2p_gather(local, bufs, counts, rank, dst, group):
    if rank == dst:
        bufs[dst].copy_(local)
        ops = [dist.P2POp(dist.irecv, bufs[r], _global_rank(group, r), group) for r in range(len(counts))
               if r != dst and counts[r] > 0]
    else:
        ops = [dist.P2POp(dist.isend, local, _global_rank(group, dst), group)] if counts[rank] > 0 else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def encode_batch_sharded(encode_fn, textures, world_size, rank, gather_dst=None, gather=True):
    """textures: [n, h, w, c] uint8 tensor visible to every rank (or just this rank's slice semantics: only
    textures[begin:end] is touched).  encode_fn(batch[k,h,w,c]) -> [k, bytes] uint8 tensor.
    Returns (local_output, gathered or None).  n must divide evenly for the gather (equal counts)."""
    n = textures.shape[0]
    b, e = texture_range(n, world_size, rank)
    local = encode_fn(textures[b:e])
    if not gather or world_size == 1:
        return local, (local.unsqueeze(0) if gather else None)
    if n % world_size != 0:
        raise ValueError("gather needs n_textures divisible by world_size (equal per-rank counts)")
    return local, gather_output(local, world_size, dst=gather_dst)
