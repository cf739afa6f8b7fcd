#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X block-encode backend.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W            (N > 1, one rank per GPU)

Metric (BASELINE.json): Mpixels/s DXT1 encode of 4096x4096 RGBA8 textures, inputs already resident
in HBM, measured through the device-resident C-ABI entry point (icamd_encode_device).

A *step* is one pass of the hot path over one batch of synthetic input: `--batch` (default 16)
distinct 4096x4096 RGBA8 textures per rank, encoded by ONE kernel launch.  16 textures = 1 GiB of
source per step, well past the 256 MiB Infinity Cache, so the reads really come from HBM.
Multi-GPU (weak scaling): every rank encodes its own batch (independent textures -> no data-path
collective; the compressed slabs stay resident in each GPU's HBM).  `--gather` additionally times one
RCCL all-gather of the compressed output after the timed region and reports it as `gather_ms`.

One JSON line is printed by rank 0.  Besides the driver contract it carries
  roofline     -- algorithmic bytes per launch / mean launch duration (HIP events on the launch stream)
  cpu_baseline -- the oracle port (oracle/ic_oracle.c) timed on this host's cores on a bounded sample
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# codec name -> (codec id, source components, swap, algorithmic bytes per pixel (read + write), label)
WORKLOADS = {
    "dxt1_rgba8": (0, 4, 4.5, "DXT1"),
    "dxt1_rgb888": (0, 3, 3.5, "DXT1"),
    "dxt5_rgba8": (1, 4, 5.0, "DXT5"),
    "etc1_rgb888": (2, 3, 3.5, "ETC1"),
    "etc1_rgba8": (2, 4, 4.5, "ETC1"),
    "pvrtc2_rgba8": (3, 4, 4.25, "PVRTC1-2bpp"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dxt1_rgba8", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=4096, help="texture width = height")
    ap.add_argument("--batch", type=int, default=16, help="textures per rank per step (one launch)")
    ap.add_argument("--content", default="noise", choices=["noise", "smooth", "flat"])
    ap.add_argument("--etc-strategy", type=int, default=2)
    ap.add_argument("--gather", action="store_true", help="also time one RCCL all-gather of the compressed output")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args()


def make_batch(torch, content, batch, size, comps, device, seed):
    """Synthetic textures generated on the device (integer-only, seeded)."""
    g = torch.Generator(device=device)
    g.manual_seed(0x1234ABCD + seed)
    if content == "noise":
        return torch.randint(0, 256, (batch, size, size, comps), dtype=torch.uint8, device=device, generator=g)
    y = torch.arange(size, device=device, dtype=torch.int32).view(1, size, 1)
    x = torch.arange(size, device=device, dtype=torch.int32).view(1, 1, size)
    if content == "smooth":
        n = torch.randint(0, 32, (batch, size, size), dtype=torch.int32, device=device, generator=g)
        chans = [(255 * x // size + n) & 255, (255 * y // size + n) & 255, (255 * (x + y) // (2 * size) + n) & 255]
        if comps == 4:
            a = torch.randint(0, 256, (batch, size, size), dtype=torch.int32, device=device, generator=g)
            keep = torch.randint(0, 8, (batch, size, size), dtype=torch.int32, device=device, generator=g) != 0
            chans.append(torch.where(keep, torch.full_like(a, 255), a))
        return torch.stack(chans, dim=-1).to(torch.uint8).contiguous()
    tiles = torch.randint(0, 256, (batch, size // 16, size // 16, comps), dtype=torch.uint8, device=device, generator=g)
    img = tiles.repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)
    noisy = (torch.randint(0, 8, (batch, size // 16, size // 16, 1), device=device, generator=g) == 0)
    noisy = noisy.repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)
    noise = torch.randint(0, 256, img.shape, dtype=torch.uint8, device=device, generator=g)
    return torch.where(noisy, noise, img).contiguous()


def cpu_baseline(T, codec, comps, size, strategy, host_img):
    """Times the oracle port on this host's cores on ONE texture of the workload (bounded sample)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, cores)
    # bounded: ETC1 exhaustive search is ~10x slower per pixel, so sample a band of rows for it
    rows = size if codec != 2 else max(4, min(size, (1 << 22) // size // 4 * 4))
    sample = host_img[:rows] if codec != 3 else host_img
    if codec == 3:
        rows = size
    t0 = time.perf_counter()
    reps = 0
    while True:
        out = T.oracle_encode(codec, sample, rows, size, comps, 0, strategy, threads=threads if codec != 3 else 1)
        reps += 1
        if time.perf_counter() - t0 > 3.0 or reps >= 8:
            break
    dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter()
    T.oracle_encode(codec, sample, rows, size, comps, 0, strategy, threads=1)
    dt1 = time.perf_counter() - t1
    assert out is not None
    ref_info = None
    if T.have_ref():
        # the compiled reference itself (oracle/_ref, built from /root/reference in the build container and shipped
        # as a binary): single thread, its own entry point -- DXT1/ETC1 only exist for 3-byte pixels there
        import numpy as np
        compressor = {0: T.DXTC, 1: T.DXTC, 2: T.ETC, 3: T.PVRTC}[codec]
        fmt = {0: T.RGB, 1: T.RGBA, 2: T.RGB, 3: T.RGBA}[codec]
        rimg = sample if T.comps_of(fmt) == comps else np.ascontiguousarray(sample[..., :3])
        t2 = time.perf_counter()
        r = T.ref_compress(compressor, fmt, rimg.reshape(-1), rows, size, 0, strategy)
        dt2 = time.perf_counter() - t2
        if r is not None:
            ref_info = {"value": rows * size / dt2 / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                        "entry_point": "%sCompressor::Compress(%s)" % ({T.DXTC: "Dxtc", T.ETC: "Etc", T.PVRTC: "Pvrtc"}[compressor],
                                                                       {T.RGB: "kRGB", T.RGBA: "kRGBA"}[fmt])}
    return {
        "reference_single_thread": ref_info,
        "value": rows * size / dt / 1e6, "unit": "Mpixels/s", "cores": threads if codec != 3 else 1, "kind": "port",
        "single_thread_value": rows * size / dt1 / 1e6,
        "sample": "oracle/ic_oracle.c (plain-C port of the reference, -O2), %dx%d px of one workload texture, "
                  "%d rep(s), slab-parallel over block rows with %d pthreads; plus one single-thread pass"
                  % (size, rows, reps, threads if codec != 3 else 1),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import ic_amd_loader
    pkg = ic_amd_loader.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            sys.exit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the backend has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    codec, comps, bytes_per_px, label = WORKLOADS[args.workload]
    size, batch = args.size, args.batch
    src = make_batch(torch, args.content, batch, size, comps, device, seed=rank)
    per_image_out = pkg.encoded_size(codec, size, size)
    out = torch.empty((batch, per_image_out), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream()

    def step():
        r = pkg.encode_device(codec, src, size, size, comps, etc_strategy=args.etc_strategy, n_images=batch, out=out,
                              stream=stream)
        assert r is not None

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = None
    if args.gather and distributed:
        from image_compression_amd import sharding
        gathered = sharding.gather_output(out, world)  # warm-up (communicator set-up)
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        gathered = sharding.gather_output(out, world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert torch.equal(gathered[rank], out)

    pixels_per_step_rank = batch * size * size
    total_pixels = pixels_per_step_rank * world * args.steps
    value = total_pixels / elapsed / 1e6

    result = {
        "metric": "Mpixels/s encode (%s, %dx%d %s)" % (label, size, size, "RGBA8" if comps == 4 else "RGB888"),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic (%s, seeded, generated on device)" % args.content,
        "config": {"workload": "%s encode of %d x %dx%d %s textures per GPU per step (one launch), device-resident"
                               % (label, batch, size, size, "RGBA8" if comps == 4 else "RGB888"),
                   "codec": args.workload, "textures_per_gpu_per_step": batch, "texture": [size, size],
                   "src_bytes_per_pixel": comps, "etc_strategy": args.etc_strategy if codec == 2 else None,
                   "parallelism": "independent textures per GPU (no data-path collective)",
                   "kernel": pkg.kernel_name(codec, comps)},
    }
    if gather_ms is not None:
        result["gather_ms"] = round(gather_ms, 3)

    if rank == 0:
        algo_bytes = pixels_per_step_rank * bytes_per_px
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload)
        result["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "kernel": pkg.kernel_name(codec, comps), "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_pixel": bytes_per_px, "algorithmic_bytes_per_launch": int(algo_bytes),
            "read_roofline_frac": round((pixels_per_step_rank * comps / (kernel_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS, 4),
        }
        # Informational: how busy the integer VALU is.  Instructions per wave come from the committed PMC profile of
        # this workload (SQ_INSTS_VALU / SQ_WAVES); a wave instruction occupies its SIMD for 4 clocks at the 16
        # lanes/clk base rate, 1 024 SIMDs, 2.4 GHz.  Not a second roofline in the contract's sense.
        spath = os.path.join(ROOT, "profiles", "r01_%s_summary.json" % args.workload)
        if os.path.exists(spath) and (size, batch) == (4096, 16):
            with open(spath) as f:
                kernels = json.load(f).get("kernels", {})
            issue_s = 0.0
            for kinfo in kernels.values():
                if "valu_insts_per_wave" in kinfo and "SQ_WAVES" in kinfo:
                    issue_s += kinfo["SQ_WAVES"] * kinfo["valu_insts_per_wave"] * 4.0 / (1024 * 2.4e9)
            if issue_s > 0:
                result["roofline"]["valu_issue_frac"] = round(issue_s / (kernel_ms * 1e-3), 3)
        import ic_testlib as T
        host0 = src[0].cpu().numpy()
        if not args.no_verify:
            got = out[0].cpu().numpy().tobytes()
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
            want = T.oracle_encode(codec, host0, size, size, comps, 0, args.etc_strategy,
                                   threads=1 if codec == 3 else cores)
            result["parity"] = "bit-exact vs oracle (texture 0 of the batch)" if got == want else "MISMATCH vs oracle"
            if got != want:
                print(json.dumps(result))
                sys.exit("parity check failed")
            if codec != 3:
                # informational: quality of the encoded texture 0, decoded again on the device
                try:
                    dec = pkg.decode_device(codec, out[0].contiguous(), size, size)
                    dcomps = 4 if codec == 1 else 3
                    a = dec.view(size, size, dcomps).to(torch.float64)
                    b = src[0][..., :dcomps].to(torch.float64)
                    mse = float(((a - b) ** 2).mean())
                    result["psnr_db"] = None if mse == 0 else round(10.0 * math.log10(255.0 * 255.0 / mse), 2)
                except Exception as e:  # never let the informational figure take the benchmark line down
                    result["psnr_db"] = "unavailable: %s" % e
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(T, codec, comps, size, args.etc_strategy, host0)
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
