#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X block-encode backend.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 needs one process per GPU: when the script is not already running under
torch.distributed.run (no RANK in the environment) it re-launches itself as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
so the bare command works; launched by the driver under torch.distributed.run it just reads RANK / LOCAL_RANK /
WORLD_SIZE.

Metric (BASELINE.json): Mpixels/s DXT1 encode of 4096x4096 RGBA8 textures, inputs already resident in HBM, measured
through the device-resident C-ABI entry point (icamd_encode_device).  `--config c2|c3|c4|c5` selects the other
BASELINE.json configurations exactly as worded there (c2 is the default).

A *step* is one pass of the hot path over one batch of synthetic input: `batch` distinct textures per rank, encoded by
ONE kernel launch (16 x 4096^2 RGBA8 = 1 GiB of source per step, well past the 256 MiB Infinity Cache, so the reads
really come from HBM).  Multi-GPU: every rank encodes its own textures (independent textures -> no data-path
collective).  The timed region is exactly K such steps between barriers -> `value`.

After it, for N > 1, a second region of K steps times encode -> RCCL gather of the compressed output to rank 0
("wall time from first launch to completion of the RCCL gather on rank 0"), with the gather of batch k on
a second stream overlapping the encode of batch k+1 -> `value_with_gather`, `gather_ms` (one un-overlapped gather).

`configs` (same line): after the c2 legs the OTHER BASELINE configurations -- c3 (DXT5 8192^2), c4 (ETC1 kSmallerError, 1024 x
1024^2 over the ranks), c5 (PVRTC 2bpp 4096^2) -- run a handful of timed steps each, with their own roofline and parity.
`slab` (same line): ONE large image (4096^2 / 8192^2 DXT5 / 16384^2) split into block-row slabs over the ranks
(sharding.slab_geometry, reference compressor4x4_helper.h:202-214: row-major blocks -> contiguous output ranges), strong
scaling, plus the gather of the slabs into rank 0's final buffer.  `--shard slab` makes that the headline line instead.

One JSON line is printed by rank 0.  Besides the driver contract it carries
  roofline     -- algorithmic bytes per launch / mean launch duration (HIP events on the launch stream); `bound` says
                  which unit limits the kernel; `hbm_frac` and (where a matching PMC profile is committed) `valu_frac`
  cpu_baseline -- the oracle port (oracle/ic_oracle.c) timed on this host's cores on a bounded sample
  host_api     -- the host-buffer drop-in (icamd_compress, H2D + kernel + D2H), never the reported `value`
"""
import argparse
import json
import math
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# name -> (codec id, source components, algorithmic bytes per pixel (read + write), label, limiting unit)
WORKLOADS = {
    "dxt1_rgba8": (0, 4, 4.5, "DXT1", "hbm"),
    "dxt1_rgb888": (0, 3, 3.5, "DXT1", "hbm"),
    "dxt5_rgba8": (1, 4, 5.0, "DXT5", "valu"),
    "etc1_rgb888": (2, 3, 3.5, "ETC1", "valu"),
    "etc1_rgba8": (2, 4, 4.5, "ETC1", "valu"),
    "pvrtc2_rgba8": (3, 4, 4.25, "PVRTC1-2bpp", "valu"),
    # EXTENSION, parity unpinned (the reference has no 4 bpp mode): checked against the oracle's restatement and by decoding
    "pvrtc4_rgba8": (4, 4, 4.5, "PVRTC1-4bpp", "valu"),
}

# BASELINE.json configs[1..4] (configs[0] is the reference's own CPU case = the cpu_baseline leg).
#   total_textures: the configuration fixes the WHOLE job (strong scaling: split over the ranks by texture_range);
#   batch: textures per rank per step (weak scaling)
CONFIGS = {
    "c2": dict(workload="dxt1_rgba8", size=4096, batch=16,
               text="DXT1 encode 4096x4096 RGBA8 on 1x MI355X, bit-exact vs reference"),
    "c3": dict(workload="dxt5_rgba8", size=8192, batch=4,
               text="DXT5 (RGB+alpha) encode 8192x8192 on 1x MI355X, block-per-lane + alpha endpoint kernel"),
    "c4": dict(workload="etc1_rgb888", size=1024, total_textures=1024, etc_strategy=2,
               text="ETC1 high-quality search (kSmallerError), batch of 1024x 1024x1024 textures sharded across the "
                    "GPUs (texture_range per rank, RCCL gather)"),
    "c5": dict(workload="pvrtc2_rgba8", size=4096, batch=16,
               text="PVRTC 2bpp encode 4096x4096 on 1x MI355X (the reference has no 4bpp mode, SURVEY D3)"),
    # not a BASELINE configuration: config 5's codec on textures whose block row no longer fits one workgroup (r06: the one-pass
    # kernel's halo form; until r05 the morph + encode pair at 0.31 with 2.2 x the traffic) -- parity pinned like c5
    "c5_8192": dict(workload="pvrtc2_rgba8", size=8192, batch=4,
                    text="PVRTC 2bpp encode 8192x8192 (4 textures per launch): a block row is two workgroups wide"),
    # not a BASELINE-pinned leg: config 5's literal "PVRTC 4bpp" as an EXTENSION (no reference implementation to pin it to)
    "c5_4bpp": dict(workload="pvrtc4_rgba8", size=4096, batch=16,
                    text="PVRTC 4bpp encode 4096x4096 on 1x MI355X -- EXTENSION, parity unpinned: the 2bpp rules of the "
                         "reference with 4x4 blocks (SURVEY D3); checked against the oracle's restatement and by decoding"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="a BASELINE.json configuration preset")  # (+ c5_4bpp)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=None, help="texture width = height")
    ap.add_argument("--batch", type=int, default=None, help="textures per rank per step (one launch)")
    ap.add_argument("--content", default="noise", choices=["noise", "smooth", "flat"])
    ap.add_argument("--etc-strategy", type=int, default=None)
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the encode->gather region")
    ap.add_argument("--no-next-rows", action="store_true",
                    help="default line, N = 1: skip the `next_rows` legs (decoders, Downsample, DXT1->ETC1 transcode of 16 x 4096^2)")
    ap.add_argument("--gather-impl", choices=["c", "torch"], default="c",
                    help="N > 1, nccl backend: the gather legs use the library's own collective (icamd_gather_blocks_rccl on its own "
                         "ncclComm_t, default) or torch.distributed's dist.gather / isend-irecv")
    ap.add_argument("--shard", default="textures", choices=["textures", "slab"],
                    help="textures: every rank encodes its own textures (weak scaling; c4: strong, texture_range). "
                         "slab: ONE --size^2 image of the workload split into block-row slabs over the ranks (strong "
                         "scaling) + gather of the slabs to rank 0")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the c3 / c4 / c5 legs of the default (c2) line")
    ap.add_argument("--no-slab", action="store_true", help="skip the one-large-image slab legs of the default (c2) line")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of each extra-config / slab leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained-run leg (N = 1)")
    ap.add_argument("--no-single-image", action="store_true", help="skip the one-image-per-launch leg (N = 1)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0)
    ap.add_argument("--precondition-seconds", type=float, default=1.0,
                    help="untimed back-to-back launches of the step before the W warm-up steps, so that the timed K "
                         "steps see the chip's steady power state rather than its ramp out of idle (0 disables)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic in this run (two short child runs of the workload under "
                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE); use the committed profiles/traffic.json instead")
    ap.add_argument("--watchdog-seconds", type=float, default=480.0,
                    help="after the headline measurement: if the extra legs (gather, configs, slab -- collectives that have never run "
                         "between real GPUs) are still going after this long, rank 0 prints the line with what it has and every rank exits 0")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # internal: three launches, no output
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the torch.distributed / RCCL code path (init, barriers, all-reduce, encode->gather region) "
                         "even with a single rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: debugging only (several ranks sharing one GPU; the gather is staged through the host)")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config or "c2"] if (args.config or not (args.workload or args.size or args.batch)) else {}
    args.preset = (args.config or "c2") if preset else None
    args.workload = args.workload or preset.get("workload", "dxt1_rgba8")
    args.size = args.size or preset.get("size", 4096)
    args.total_textures = preset.get("total_textures") if args.batch is None else None
    args.batch = args.batch or preset.get("batch", 16)
    if args.etc_strategy is None:
        args.etc_strategy = preset.get("etc_strategy", 2)
    return args


def relaunch_distributed(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks of this same command."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.call(cmd, env=env))


def make_batch(torch, content, batch, size, comps, device, seed, height=None, row0=0):
    """Synthetic textures generated on the device (integer-only, seeded): `batch` images of `height` (default: size) rows
    x `size` columns.  row0: first row of a slab inside its size x size image (only the smooth ramp depends on it)."""
    h = size if height is None else height
    if str(device) == "cpu":
        g = torch.Generator()
    else:
        g = torch.Generator(device=device)
    g.manual_seed(0x1234ABCD + seed)
    if content == "noise":
        return torch.randint(0, 256, (batch, h, size, comps), dtype=torch.uint8, device=device, generator=g)
    y = (torch.arange(h, device=device, dtype=torch.int32) + row0).view(1, h, 1)
    x = torch.arange(size, device=device, dtype=torch.int32).view(1, 1, size)
    if content == "smooth":
        n = torch.randint(0, 32, (batch, h, size), dtype=torch.int32, device=device, generator=g)
        chans = [(255 * x // size + n) & 255, (255 * y // size + n) & 255, (255 * (x + y) // (2 * size) + n) & 255]
        if comps == 4:
            a = torch.randint(0, 256, (batch, h, size), dtype=torch.int32, device=device, generator=g)
            keep = torch.randint(0, 8, (batch, h, size), dtype=torch.int32, device=device, generator=g) != 0
            chans.append(torch.where(keep, torch.full_like(a, 255), a))
        return torch.stack(chans, dim=-1).to(torch.uint8).contiguous()
    th, tw = (h + 15) // 16, (size + 15) // 16
    tiles = torch.randint(0, 256, (batch, th, tw, comps), dtype=torch.uint8, device=device, generator=g)
    img = tiles.repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)
    noisy = (torch.randint(0, 8, (batch, th, tw, 1), device=device, generator=g) == 0)
    noisy = noisy.repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)
    noise = torch.randint(0, 256, img.shape, dtype=torch.uint8, device=device, generator=g)
    return torch.where(noisy, noise, img)[:, :h, :size].contiguous()


def cpu_baseline(T, codec, comps, size, strategy, host_img):
    """Times the oracle port on this host's cores on ONE texture of the workload (bounded sample)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, cores)
    # bounded: ETC1 exhaustive search is ~10x slower per pixel, so sample a band of rows for it
    rows = size if codec != 2 else max(4, min(size, (1 << 22) // size // 4 * 4))
    sample = host_img[:rows] if codec not in (3, 4) else host_img
    if codec in (3, 4):
        rows = size
    # one untimed pass (thread start-up, page faults), then about a second of wall time of back-to-back passes: on a
    # 256-thread host that is ~90 passes of a 4096^2 DXT1 texture = ~15 s of single-core work
    out = T.oracle_encode(codec, sample, rows, size, comps, 0, strategy, threads=threads if codec not in (3, 4) else 1)
    t0 = time.perf_counter()
    reps = 0
    while True:
        out = T.oracle_encode(codec, sample, rows, size, comps, 0, strategy, threads=threads if codec not in (3, 4) else 1)
        reps += 1
        if time.perf_counter() - t0 > 1.0 or reps >= 200:
            break
    dt = (time.perf_counter() - t0) / reps
    single_passes = 3
    t1 = time.perf_counter()
    for _ in range(single_passes):
        T.oracle_encode(codec, sample, rows, size, comps, 0, strategy, threads=1)
    dt1 = (time.perf_counter() - t1) / single_passes
    assert out is not None
    ref_info = None
    if T.have_ref():
        # the compiled reference itself (oracle/_ref, built from /root/reference in the build container and shipped
        # as a binary): single thread, its own entry point -- DXT1/ETC1 only exist for 3-byte pixels there
        import numpy as np
        if codec == 4:  # extension: the reference has no PVRTC 4 bpp
            return {"reference_single_thread": None, "value": rows * size / dt / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                    "single_thread_value": rows * size / dt1 / 1e6,
                    "sample": "oracle/ic_oracle.c pvrtc4_encode_image (extension, parity unpinned), one %dx%d texture, %d passes" % (size, size, reps)}
        compressor = {0: T.DXTC, 1: T.DXTC, 2: T.ETC, 3: T.PVRTC}[codec]
        fmt = {0: T.RGB, 1: T.RGBA, 2: T.RGB, 3: T.RGBA}[codec]
        rimg = sample if T.comps_of(fmt) == comps else np.ascontiguousarray(sample[..., :3])
        t2 = time.perf_counter()
        for _ in range(single_passes):
            r = T.ref_compress(compressor, fmt, rimg.reshape(-1), rows, size, 0, strategy)
        dt2 = (time.perf_counter() - t2) / single_passes
        if r is not None:
            ref_info = {"value": rows * size / dt2 / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                        "entry_point": "%sCompressor::Compress(%s)" % ({T.DXTC: "Dxtc", T.ETC: "Etc", T.PVRTC: "Pvrtc"}[compressor],
                                                                       {T.RGB: "kRGB", T.RGBA: "kRGBA"}[fmt])}
    # (r06) the compiled reference's own figure -- "the reference's CPU path" -- rides at the top level of this
    # object too, and in `sample`, not only in the nested `reference_single_thread`
    ref_text = "; the compiled reference was not shipped with this run (oracle/_ref absent)"
    if ref_info:
        ref_text = "; compiled reference itself (oracle/_ref = /root/reference built in place, %s, 1 thread): %.1f Mpixels/s" \
            % (ref_info["entry_point"], ref_info["value"])
    return {
        "reference_single_thread": ref_info,
        "reference_value": ref_info["value"] if ref_info else None, "reference_cores": 1 if ref_info else None,
        "reference_kind": "reference" if ref_info else None,
        "value": rows * size / dt / 1e6, "unit": "Mpixels/s", "cores": threads if codec not in (3, 4) else 1, "kind": "port",
        "single_thread_value": rows * size / dt1 / 1e6,
        "sample": "oracle/ic_oracle.c (plain-C port of the reference, -O2), %dx%d px of one workload texture, "
                  "%d timed pass(es) after one untimed, slab-parallel over block rows with %d pthreads (= %.1f s of "
                  "single-core work); plus %d single-thread passes of the port (%.1f Mpixels/s)%s"
                  % (size, rows, reps, threads if codec not in (3, 4) else 1, reps * dt1, single_passes, rows * size / dt1 / 1e6, ref_text),
    }


def host_api_leg(pkg, T, codec, size, strategy):
    """Steady state of the host-buffer drop-in (Compressor::Compress, compressor.h:77-80) on ONE texture of the
    workload's size in the reference's own input format (kRGB for DXT1 / ETC1, kRGBA for DXT5 / PVRTC)."""
    import numpy as np
    compressor = {0: pkg.COMPRESSOR_DXTC, 1: pkg.COMPRESSOR_DXTC, 2: pkg.COMPRESSOR_ETC, 3: pkg.COMPRESSOR_PVRTC}[codec]
    fmt = {0: pkg.RGB, 1: pkg.RGBA, 2: pkg.RGB, 3: pkg.RGBA}[codec]
    comps = 3 if fmt == pkg.RGB else 4
    img = T.s_noise(size, size, comps, index=99)
    first = pkg.compress_host(compressor, fmt, img, size, size, etc_strategy=strategy)
    if first is None:
        return None
    want = T.oracle_compress({0: T.DXTC, 1: T.DXTC, 2: T.ETC, 3: T.PVRTC}[codec], fmt, img, size, size, 0, strategy) \
        if size <= 4096 and codec != 2 else None

    def timed(out):
        for _ in range(2):
            pkg.compress_host(compressor, fmt, img, size, size, etc_strategy=strategy, out=out)
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            pkg.compress_host(compressor, fmt, img, size, size, etc_strategy=strategy, out=out)
        return (time.perf_counter() - t0) / reps
    out = np.zeros(len(first), np.uint8)
    dt = timed(out)
    res = {"entry_point": "icamd_compress (bands of block rows: H2D + kernel + D2H pipelined on two streams)",
           "texture": [size, size], "format": "kRGB" if comps == 3 else "kRGBA",
           "pageable": {"ms_per_call": round(dt * 1e3, 4), "value": round(size * size / dt / 1e6, 1), "unit": "Mpixels/s",
                        "source_GBps": round(size * size * comps / dt / 1e9, 2)},
           "parity": None if want is None else ("bit-exact vs oracle" if first == want else "MISMATCH vs oracle")}
    try:  # the same call on caller buffers page-locked once with icamd_host_register
        pkg.host_register(img)
        pkg.host_register(out)
        try:
            dtp = timed(out)
            res["page_locked"] = {"ms_per_call": round(dtp * 1e3, 4), "value": round(size * size / dtp / 1e6, 1),
                                  "unit": "Mpixels/s", "source_GBps": round(size * size * comps / dtp / 1e9, 2),
                                  "parity": None if want is None else ("bit-exact vs oracle" if out.tobytes() == want
                                                                       else "MISMATCH vs oracle")}
        finally:
            pkg.host_unregister(out)
            pkg.host_unregister(img)
    except Exception as e:
        res["page_locked"] = "unavailable: %s" % e
    return res


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return None if n == 0 else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


class SclkSampler:
    """Samples the GPU core clock the driver reports (sysfs hwmon freq1_input / pp_dpm_sclk, else `rocm-smi`) from a
    host thread while a timed leg runs.  Secondary evidence next to the in-kernel clock probe; absent sources are
    reported as such, never guessed."""

    def __init__(self, pci_address=None, period_s=0.02):
        import glob
        import threading
        self.period = period_s
        self.samples = []
        self.source = None
        self._stop = threading.Event()
        self._thread = None

        def mine(paths):
            # the node's sysfs shows every GPU of the host, the process sees one: keep the card whose PCI address is
            # the HIP device's (a wrong card's clock is worse than none)
            out = []
            for p in paths:
                card = p.split("/device/")[0] + "/device"
                if pci_address and os.path.basename(os.path.realpath(card)).lower() == pci_address.lower():
                    out.append(p)
            return out if pci_address else (paths if len(paths) == 1 else [])
        cands = mine(sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")))
        if cands:
            self.source, self._path, self._read = "sysfs hwmon freq1_input of %s" % pci_address, cands[0], self._read_hwmon
        else:
            cands = mine(sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")))
            if cands:
                self.source, self._path, self._read = "sysfs pp_dpm_sclk of %s" % pci_address, cands[0], self._read_dpm
            else:
                import shutil
                if shutil.which("rocm-smi"):
                    self.source, self._path, self._read = "rocm-smi --showclocks", None, self._read_smi
                    self.period = max(self.period, 0.25)

    def _read_hwmon(self):
        with open(self._path) as f:
            return float(f.read().strip()) / 1e6

    def _read_dpm(self):
        with open(self._path) as f:
            for line in f:
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        return None

    def _read_smi(self):
        import re
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
        return float(m.group(1)) if m else None

    def _run(self):
        while not self._stop.is_set():
            try:
                v = self._read()
                if v:
                    self.samples.append(v)
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.source:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=10)

    def summary(self):
        if not self.samples:
            return {"source": self.source, "samples": 0}
        return {"source": self.source, "samples": len(self.samples), "median_MHz": round(_median(self.samples), 1),
                "min_MHz": round(min(self.samples), 1), "max_MHz": round(max(self.samples), 1)}


def sustained_leg(torch, pkg, step_fn, stream, kernel_ms_hint, seconds, algo_bytes):
    """>= `seconds` (default 5: long enough for a once-per-second rocm-smi sampler to see the GPU busy) of back-to-back launches of the workload (one event between consecutive launches, so a period
    includes whatever gap the launches leave), with the shader clock measured underneath by a one-wave probe kernel
    on a second stream (s_memtime / s_memrealtime) and the driver-reported sclk sampled from the host.  Reports the
    median period of the first 20 launches, of every later 10 % slice and of the last 20 %."""
    n = int(min(40000, max(200, math.ceil(seconds * 1e3 / max(kernel_ms_hint, 1e-3)))))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    probe_stream = torch.cuda.Stream()
    est_ms = n * kernel_ms_hint
    pci = None
    try:
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    bufs = [pkg.clock_probe_buffer() for _ in range(3)]  # cleared before the launches the probes overlap
    torch.cuda.synchronize()
    time.sleep(0.5)  # start from an idle chip, like a fresh caller would: the first decile shows the ramp
    probes = []
    with SclkSampler(pci) as sclk:
        t0 = time.perf_counter()
        # three probes: the first 10 % of the run, the middle, the last 25 %
        try:
            probes.append(("first_10pct", pkg.clock_probe(int(est_ms * 1e3 * 0.10), probe_stream, bufs[0])))
        except Exception as e:
            probes.append(("error", str(e)))
        for i in range(n):
            if i == n // 2 and probes and probes[0][0] != "error":
                probes.append(("middle_10pct", pkg.clock_probe(int(est_ms * 1e3 * 0.10), probe_stream, bufs[1])))
            if i == (n * 7) // 10 and probes and probes[0][0] != "error":
                probes.append(("last_25pct", pkg.clock_probe(int(est_ms * 1e3 * 0.25), probe_stream, bufs[2])))
            ev[i].record(stream)
            step_fn(i)
        ev[n].record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    slices = []
    for d in range(10):
        lo, hi = (n * d) // 10, (n * (d + 1)) // 10
        slices.append(round(_median(per[lo:hi]), 5))
    last = per[(n * 8) // 10:]
    last_ms = _median(last)
    clock = {}
    for name, res in probes:
        if name == "error":
            clock["probe_error"] = res
            continue
        r = res()
        if r:
            clock[name] = {"shader_MHz": round(r["shader_MHz"], 1), "interval_ms": round(r["interval_ms"], 2)}
    clock["method"] = ("one sleeping wave on a second stream reads s_memtime (shader cycles) and s_memrealtime (constant "
                       "%s kHz) at both ends of its interval while the launches run" % pkg.lib().icamd_wall_clock_rate_khz())
    clock["driver_sclk"] = sclk.summary()
    return {
        "launches": n, "wall_s": round(wall, 3), "mean_ms_per_launch_wall": round(wall / n * 1e3, 5),
        "median_ms_first_20": round(_median(per[:20]), 5),
        "median_ms_ramp": {"%d-%d" % (a, b): round(_median(per[a:b]), 5) for a, b in
                           ((0, 20), (20, 50), (50, 100), (100, 200), (200, 400), (400, 800)) if b <= n},
        "median_ms_by_decile": slices,
        "median_ms_last_20pct": round(last_ms, 5), "min_ms": round(min(per), 5), "max_ms": round(max(per), 5),
        "achieved_GBps_last_20pct": round(algo_bytes / (last_ms * 1e-3) / 1e9, 1),
        "frac_last_20pct": round(algo_bytes / (last_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "timing": "HIP events on the launch stream, one between consecutive launches (period = kernel + launch gap)",
    }, clock


def single_image_leg(torch, pkg, codec, comps, src, size, batch, strategy, stream, bytes_per_px, calls=None):
    """The literal shape of the BASELINE configuration and of Compressor::Compress (compressor.h:77-80): ONE texture
    per call, rotating through the batch's distinct textures (>= 256 MiB of distinct sources, so every call reads HBM,
    not the Infinity Cache), a HIP event pair around every call."""
    per_image_out = pkg.encoded_size(codec, size, size)
    out = torch.empty((batch, per_image_out), dtype=torch.uint8, device=src.device)
    distinct_bytes = batch * size * size * comps
    calls = calls or max(256, min(2000, 8 * batch * max(1, (256 << 20) // max(distinct_bytes, 1))))

    def call(i):
        k = i % batch
        r = pkg.encode_device(codec, src[k], size, size, comps, etc_strategy=strategy, n_images=1, out=out[k:k + 1],
                              stream=stream)
        assert r is not None
    for i in range(min(calls, 2 * batch)):
        call(i)
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    t0 = time.perf_counter()
    for i in range(calls):
        starts[i].record(stream)
        call(i)
        ends[i].record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    med = _median(per)
    algo = size * size * bytes_per_px
    # The same calls WITHOUT a timestamp packet pair around each (one event pair around the whole run): r04's kernel trace
    # (profiles/r04_single_image_timeline.txt) shows the pair itself costs ~2 us per call -- the kernel runs 13.5 us from
    # dispatch to completion for a 4096^2 DXT1 image either way.  And round-robin over two streams, which is what a caller
    # with independent textures can do to overlap one call's ramp with the previous call's tail.
    def run_bare(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        joins = [torch.cuda.Event() for _ in streams[1:]]
        torch.cuda.synchronize()
        e0.record(streams[0])
        for s2 in streams[1:]:
            s2.wait_event(e0)
        for i in range(calls):
            k = i % batch
            r = pkg.encode_device(codec, src[k], size, size, comps, etc_strategy=strategy, n_images=1, out=out[k:k + 1],
                                  stream=streams[i % len(streams)])
            assert r is not None
        for j, s2 in zip(joins, streams[1:]):
            j.record(s2)
            streams[0].wait_event(j)
        e1.record(streams[0])
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / calls
        return {"ms_per_call": round(t, 5), "value": round(size * size / (t * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
                "frac": round(algo / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    bare = two = None
    try:
        bare = run_bare([stream])
        two = run_bare([torch.cuda.Stream(device=src.device), torch.cuda.Stream(device=src.device)])
    except Exception as e:  # diagnostic legs
        bare = bare or "unavailable: %s" % e
    # The same calls captured once into a HIP graph (one call per distinct texture) and replayed: what a caller with many
    # single textures gets when the host's per-call launch cost is taken out (PVRTC needs its scratch
    # memory handed over for the capture, icamd_pvrtc2_set_workspace).
    graph = None
    try:
        gs = torch.cuda.Stream(device=src.device)
        ws = torch.empty(max(1, pkg.pvrtc_workspace_size(size, 1) if codec == 3 else pkg.pvrtc4_workspace_size(size, 1)),
                         dtype=torch.uint8, device=src.device) if codec in (3, 4) else None
        with torch.cuda.stream(gs):
            if ws is not None:
                pkg.pvrtc_set_workspace(ws)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=gs):
                    for k in range(batch):
                        pkg.encode_device(codec, src[k], size, size, comps, etc_strategy=strategy, n_images=1,
                                          out=out[k:k + 1], stream=gs)
            finally:
                if ws is not None:
                    pkg.pvrtc_set_workspace(None)
            for _ in range(3):
                g.replay()
            gs.synchronize()
            reps = max(8, min(200, calls // batch))
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            for i in range(reps):
                marks[i].record(gs)
                g.replay()
            marks[reps].record(gs)
            gs.synchronize()
        gper = _median([marks[i].elapsed_time(marks[i + 1]) for i in range(reps)]) / batch
        graph = {"calls_per_graph": batch, "replays": reps, "median_ms_per_call": round(gper, 5),
                 "value": round(size * size / (gper * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
                 "frac": round(algo / (gper * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        del g
    except Exception as e:  # a diagnostic leg: never fatal
        graph = "unavailable: %s" % e
    return {
        "graph_replay": graph, "back_to_back_no_events": bare, "two_streams_round_robin": two,
        "texture": [size, size], "distinct_textures_rotated": batch, "distinct_source_MiB": distinct_bytes >> 20,
        "calls": calls, "median_ms_per_call": round(med, 5), "min_ms": round(min(per), 5), "max_ms": round(max(per), 5),
        "back_to_back_ms_per_call_wall": round(wall / calls * 1e3, 5),
        "value": round(size * size / (med * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
        "achieved_GBps": round(algo / (med * 1e-3) / 1e9, 1), "frac": round(algo / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "note": "one icamd_encode_device call = one config-sized image; HIP event pair per call on the launch stream",
    }


def valu_fraction(workload, content, etc_strategy, codec, pixels_per_launch, kernel_ms, clock_mhz, preset=None, live_insts=None,
                  live_cycles=None):
    """Integer-VALU issue fraction of the sustained run, from measured quantities only:
      executed VALU wave-instructions per kernel  -- SQ_INSTS_VALU of the committed PMC profile of EXACTLY this workload /
                                                    content / ETC strategy (profiles/valu_insts.json);
      issue clocks per instruction                -- 4 at the base rate (16 lanes / clk / SIMD: the measured ceiling of
                                                    scripts/ubench_valu.hip, 38.6 T lane-ops/s at 2.36 GHz), 2 for the
                                                    add / sub / and / or / xor / right-shift / mov ops that issue at twice
                                                    that rate, weighted with each kernel's static mix (profiles/*_isa.json);
      clock                                       -- the shader clock the probe measured during this run (or 2 400 MHz
                                                    nominal when the probe is unavailable, which is then said so)."""
    mpix = pixels_per_launch / 1e6
    vi = None
    if live_insts:  # r05: counted in this run by live_traffic()'s third pass, over exactly this launch shape and content
        total = sum(live_insts.values())
        vi = {"valu_wave_insts_per_Mpixel": total / mpix, "per_kernel_valu_wave_insts_per_Mpixel": {k: v / mpix for k, v in live_insts.items()},
              "valu_wave_insts_per_block_lane": round(total * 64.0 / (pixels_per_launch / (32.0 if codec == 3 else 16.0)), 1),
              "profile": "measured in this run (rocprofv3 --pmc SQ_INSTS_VALU over a child process launching this workload three times)"}
    vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
    if vi is None and os.path.exists(vpath):
        with open(vpath) as f:
            table = json.load(f)
            # a profile of EXACTLY the preset's launch shape and content first (c4: the ETC1 search is content- and size-dependent)
            vi = (table.get("preset:%s/%s/s%d" % (preset, content, etc_strategy if codec == 2 else 0)) if preset else None) or \
                table.get("%s/%s/s%d" % (workload, content, etc_strategy if codec == 2 else 0))
    if not vi:
        return None
    isa = {}
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("_isa.json"):
            with open(os.path.join(ROOT, "profiles", name)) as f:
                isa = json.load(f).get("kernels", {})
            break
    per_kernel = vi.get("per_kernel_valu_wave_insts_per_Mpixel") or {}
    clocks, detail = 0.0, {}
    if per_kernel:
        for k, ipm in per_kernel.items():
            cpi = (isa.get(k) or {}).get("valu_issue_clk_per_inst_static", 4.0)
            clocks += ipm * mpix * cpi
            detail[k] = {"valu_wave_insts": round(ipm * mpix), "issue_clk_per_inst": cpi}
    else:
        clocks = vi["valu_wave_insts_per_Mpixel"] * mpix * 4.0
    mhz = clock_mhz or 2400.0
    if live_insts and live_cycles and sorted(live_cycles) == sorted(live_insts):
        # r06: numerator AND denominator from ONE profiled pass over the same launches -- executed VALU
        # wave-instructions (SQ_INSTS_VALU) x model issue clocks over 1 024 SIMDs x the shader cycles those launches took
        # (GRBM_GUI_ACTIVE / 8 XCDs): no clock probe, no kernel time of another run, nothing that can drift apart.
        # simd_cycles_per_valu_inst is the model-free part: what the SIMDs spent per VALU instruction (profiles/r06_valu_counters.txt:
        # 3.8 - 4.0 for the VALU-bound kernels at four waves per SIMD).
        cyc = sum(live_cycles.values())
        return {"valu_frac": round(clocks / (1024.0 * cyc), 3),
                "simd_cycles_per_valu_inst": round(1024.0 * cyc / sum(live_insts.values()), 3),
                "model_issue_clk_per_valu_inst": round(clocks / sum(live_insts.values()), 3),
                "valu_wave_insts_per_block": vi.get("valu_wave_insts_per_block_lane"),
                "valu_profile": vi.get("profile") + "; shader cycles of the same launches from GRBM_GUI_ACTIVE / 8 in the same pass",
                "valu_kernels": detail, "valu_clock_MHz": "not needed: cycles counted (GRBM_GUI_ACTIVE)",
                "valu_frac_note": "executed VALU wave-instructions x model issue clocks (4; 2 for the add / logic / right-shift / mov "
                                  "ops, static mix of the kernel) / (1024 SIMDs x shader cycles of the same launches)"}
    return {"valu_frac": round(clocks / (1024 * mhz * 1e6) / (kernel_ms * 1e-3), 3),
            "valu_wave_insts_per_block": vi.get("valu_wave_insts_per_block_lane"),
            "valu_profile": vi.get("profile"), "valu_kernels": detail,
            "valu_clock_MHz": mhz if clock_mhz else "2400 nominal (no probe)",
            "valu_frac_note": "executed VALU wave-instructions x issue clocks (4; 2 for full-rate ops, static mix) / "
                              "(1024 SIMDs x measured shader clock x sustained kernel time)"}


def host_api_batch_leg(pkg, T, n_images=32, size=2048):
    """icamd_compress_batch (C callers without torch.distributed): 32 x 2048^2 kRGB -> DXT1 from pageable host
    buffers, one host worker thread (own stream + staging buffers) per device-list entry, on device lists [0] and [0, 0].
    The C entry point itself is timed (pointer lists and touched output buffers prepared once), not the Python convenience
    wrapper around it, whose fresh numpy outputs page-fault underneath the D2H copies."""
    import ctypes
    import numpy as np
    batch = [T.s_noise(size, size, 3, index=300 + i).reshape(-1) for i in range(n_images)]
    want = T.oracle_compress(T.DXTC, T.RGB, batch[0].reshape(size, size, 3), size, size, 0, 2)
    out_size = pkg.compute_compressed_data_size(pkg.COMPRESSOR_DXTC, pkg.RGB, size, size)
    outs = [np.ones(out_size, np.uint8) for _ in range(n_images)]
    in_ptrs = (ctypes.c_void_p * n_images)(*[b.ctypes.data for b in batch])
    out_ptrs = (ctypes.c_void_p * n_images)(*[o.ctypes.data for o in outs])
    statuses = (ctypes.c_int * n_images)()
    res = {"entry_point": "icamd_compress_batch", "images": n_images, "texture": [size, size], "format": "kRGB", "codec": "DXT1"}
    for devices in ([0], [0, 0]):
        devs = (ctypes.c_int * len(devices))(*devices)

        def call():
            st = pkg.lib().icamd_compress_batch(pkg.COMPRESSOR_DXTC, 2, pkg.RGB, size, size, 0, n_images, in_ptrs, out_ptrs,
                                                out_size, devs, len(devices), statuses)
            assert st == 0 and all(s == 0 for s in statuses)
        call()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        res["devices_%s" % "_".join(str(d) for d in devices)] = {
            "ms_per_batch": round(dt * 1e3, 3), "value": round(n_images * size * size / dt / 1e6, 1), "unit": "Mpixels/s",
            "source_GBps": round(n_images * size * size * 3 / dt / 1e9, 2),
            "parity": "bit-exact vs oracle (image 0)" if outs[0].tobytes() == want else "MISMATCH vs oracle"}
    return res


def clock_under_load(torch, pkg, step, kernel_ms, launches, stream=None):
    """Effective shader clock of `launches` launches of `step` AND the mean duration of exactly those launches -> (MHz, ms per
    launch), either None when the probe is unavailable.  r06: the probe wave (second stream) used to start
    before the first of its launches was even enqueued, after seconds of idle time spent in the traffic passes -- it averaged the
    chip's idle / ramping clock into a 10-launch window (c4: 2 269 MHz on one box, 2 403 on another, the kernel time the same)
    while `valu_frac` divided by the kernel time of the EARLIER timed region.  Now: >= 150 ms of untimed launches first, the probe
    wave is released by an event recorded behind them on the launch stream, the measured launches follow without a gap, the probe
    sleeps for 80 % of their expected duration -- it only ever samples while this leg's kernels run -- and the kernel time that
    goes with the clock is the one of the same launches (one HIP event on either side)."""
    try:
        s = stream if stream is not None else torch.cuda.current_stream()
        buf = pkg.clock_probe_buffer()
        ps = torch.cuda.Stream()
        for _ in range(max(launches, int(150.0 / max(kernel_ms, 1e-3)))):
            step()
        go = torch.cuda.Event()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        go.record(s)
        ps.wait_event(go)
        res = pkg.clock_probe(max(200, int(kernel_ms * 1e3 * launches * 0.8)), ps, buf)
        e0.record(s)
        for _ in range(launches):
            step()
        e1.record(s)
        torch.cuda.synchronize()
        r = res()
        return (round(r["shader_MHz"], 1) if r else None), e0.elapsed_time(e1) / launches
    except Exception:
        return None, None


def link_probe(ctx, mib=64, reps=3):
    """What a gather into rank 0 can reach on this node, measured: every peer alone sending `mib` MiB to rank 0 (one xGMI
    link under RCCL; host-staged under gloo) and all peers at once (rank 0's inbound links together).  The gathers of the
    `value_with_gather` figures cannot beat the second number whatever the encoders do."""
    if not ctx.distributed or ctx.world < 2:
        return None
    torch, dist, world, rank = ctx.torch, ctx.dist, ctx.world, ctx.rank
    staged = ctx.backend == "gloo"
    dev = "cpu" if staged else ctx.device
    n = mib << 20
    buf = torch.full((n,), rank & 255, dtype=torch.uint8, device=dev)
    recv = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None

    def timed(fn):
        fn()  # connection set-up
        ctx.sync(); ctx.barrier(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync(); ctx.barrier(); ctx.sync()
        return ctx.max_over_ranks(time.perf_counter() - t0) / reps
    alone = []
    for peer in range(1, world):
        def one(peer=peer):
            if rank == 0:
                dist.recv(recv[peer], src=peer)
            elif rank == peer:
                dist.send(buf, dst=0)
        alone.append(round(n / timed(one) / 1e9, 2))

    def everyone():
        dist.gather(buf, recv, dst=0)
    together = (world - 1) * n / timed(everyone) / 1e9
    ok = True
    if rank == 0:
        ok = all(int(recv[r][0].item()) == (r & 255) and int(recv[r][-1].item()) == (r & 255) for r in range(1, world))
    return {"backend": ctx.backend + (" (tensors staged through host memory: not an xGMI figure)" if staged else " (RCCL)"),
            "payload_MiB_per_peer": mib, "per_peer_alone_GBps": alone, "all_peers_at_once_GBps_into_rank0": round(together, 2),
            "peers": world - 1, "xgmi_links_into_rank0": min(world - 1, 7), "payload_intact": ok}


def gather_bound(probe, out_bytes_all, world, pixels_per_step_all):
    """Fields that make a value_with_gather figure interpretable: how many ranks gather, how many bytes enter rank 0 per step,
    the measured inbound rate that bounds it, and the Mpixels/s ceiling that rate implies."""
    inbound = out_bytes_all * (world - 1) / world
    d = {"gather_ranks": world, "gather_bytes_into_rank0_per_step": int(inbound)}
    if probe:
        rate = probe["all_peers_at_once_GBps_into_rank0"]
        d["gather_bound_GBps"] = rate
        d["value_with_gather_ceiling"] = round(pixels_per_step_all / (inbound / (rate * 1e9)) / 1e6, 1) if inbound and rate else None
    else:
        d["gather_bound_GBps"] = None
        d["value_with_gather_ceiling"] = None
    return d


class Ctx:
    """Rank / world / device of this process and the collectives the legs need.  `sync` is a no-op on the CPU, so that the
    gloo tier of tests/ can drive the multi-rank legs (with a stand-in encoder) where there is no GPU."""

    def __init__(self, torch, dist, rank, world, device, distributed, backend):
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.distributed, self.backend = distributed, backend
        self.on_gpu = getattr(device, "type", str(device)) == "cuda"
        self.rccl = None      # image_compression_amd.RcclGather: the library's own gather (icamd_gather_blocks_rccl), nccl backend only
        self.rccl_note = None  # why it is absent when it was asked for
        self.leg = "start"    # what is running now: what the watchdog names when it fires

    def gather_impl(self):
        if self.rccl is not None:
            return "icamd_gather_blocks_rccl (the library's C entry point: one grouped ncclSend / ncclRecv exchange on its own ncclComm_t)"
        return "torch.distributed (dist.gather / batched isend-irecv), backend %s%s" % (
            self.backend, "; the C entry point was not used: %s" % self.rccl_note if self.rccl_note else "")

    def sync(self):
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.distributed:
            self.dist.barrier()

    def _reduce(self, x, op):
        if not self.distributed:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64,
                              device=self.device if (self.backend == "nccl" and self.on_gpu) else "cpu")
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX)

    def min_over_ranks(self, x):
        return self._reduce(x, self.dist.ReduceOp.MIN)

    def sum_over_ranks(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM)

    def current_stream(self):
        return self.torch.cuda.current_stream() if self.on_gpu else None


def timed_steps(ctx, step, steps, warmup, precondition_seconds=0.0, stream=None):
    """The contract's timed region: W warm-up steps, then exactly K steps between barrier + synchronize on both sides.
    Returns (wall seconds, MAX over ranks; mean ms per step from HIP events on the launch stream -- None on the CPU)."""
    torch = ctx.torch
    if precondition_seconds > 0:
        tp = time.perf_counter()
        while time.perf_counter() - tp < precondition_seconds:
            for _ in range(50):
                step()
            ctx.sync()
    for _ in range(warmup):
        step()
    ctx.sync()
    ctx.barrier()
    ctx.sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if ctx.on_gpu else None
    t0 = time.perf_counter()
    for i in range(steps):
        if marks:
            marks[i].record(stream)
        step()
    if marks:
        marks[steps].record(stream)
    ctx.sync()
    ctx.barrier()
    ctx.sync()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    kernel_ms = marks[0].elapsed_time(marks[steps]) / steps if marks else None
    return elapsed, kernel_ms


LIVE_GUI_CYCLES = {}    # the same key -> {kernel: shader cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs)} of the SAME profiled launches
LIVE_VALU_INSTS = {}    # (workload, size, batch, content, etc_strategy) -> {kernel: SQ_INSTS_VALU per launch}, filled by live_traffic()
_LIVE_TRAFFIC_OFF = []  # set to [reason] by the first failed live measurement: the remaining legs go straight to the committed profile


def live_traffic(workload, size, batch, content, etc_strategy, timeout_s=75):
    """roofline.traffic measured IN THIS RUN: HBM bytes per launch = FETCH_SIZE * 2 + WRITE_SIZE (KiB -> bytes; the x 2 is the
    gfx950 correction for wide streaming reads, MI355X_MICROARCH.md section HBM), each counter from its own rocprofv3 --pmc pass
    (they cannot share one) over a child process that launches exactly this workload three times (`--traffic-child`); kernels
    of a multi-kernel step are summed.  Only --kernel-trace accompanies --pmc.  Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if _LIVE_TRAFFIC_OFF:
        return None, _LIVE_TRAFFIC_OFF[0]
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested rocprofv3"
    csv.field_size_limit(1 << 30)
    kib = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        d = tempfile.mkdtemp(prefix="icamd_pmc_", dir="/tmp")
        try:
            # (r06: the third pass also reads GRBM_GUI_ACTIVE -- another block, same pass: the shader cycles of the very launches
            # whose instructions are counted, summed over the 8 XCDs -- so that valu_frac needs neither a clock nor a time)
            ctrs = [ctr, "GRBM_GUI_ACTIVE"] if ctr == "SQ_INSTS_VALU" else [ctr]
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--workload", workload, "--size", str(size), "--batch", str(batch),
                   "--content", content, "--etc-strategy", str(etc_strategy)]
            # own process group: a pass that overruns is killed WITH the child it started (rocprofv3 is a wrapper)
            r = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp", start_new_session=True)
            try:
                r.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(r.pid, signal.SIGKILL)
                r.wait()
                raise
            per_kernel, gui = {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Kernel_Name"].startswith("icamd_") and row["Counter_Name"] == ctr:
                            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
                        elif row["Kernel_Name"].startswith("icamd_") and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                            gui.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if r.returncode != 0 or not per_kernel:
                _LIVE_TRAFFIC_OFF.append("rocprofv3 --pmc %s pass failed (rc %s)" % (ctr, r.returncode))
                return None, _LIVE_TRAFFIC_OFF[0]
            kib[ctr] = sum(sum(v) / len(v) for v in per_kernel.values())
            if ctr == "SQ_INSTS_VALU":  # executed VALU wave-instructions per launch, per kernel: valu_fraction()'s input
                LIVE_VALU_INSTS[(workload, size, batch, content, etc_strategy)] = {k: sum(v) / len(v) for k, v in per_kernel.items()}
                if gui and sorted(gui) == sorted(per_kernel):
                    LIVE_GUI_CYCLES[(workload, size, batch, content, etc_strategy)] = {k: sum(v) / len(v) / 8.0 for k, v in gui.items()}
        except Exception as e:  # a diagnostic: never fatal, and never paid for twice
            _LIVE_TRAFFIC_OFF.append("rocprofv3 --pmc %s pass: %s: %s" % (ctr, type(e).__name__, str(e)[:200]))
            return None, _LIVE_TRAFFIC_OFF[0]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(kib["FETCH_SIZE"] * 2048 + kib["WRITE_SIZE"] * 1024), \
        "measured in this run: FETCH_SIZE * 2 + WRITE_SIZE from two separate rocprofv3 --kernel-trace --pmc passes over a child " \
        "process launching this workload three times (per launch, kernels of a step summed); a third pass counts SQ_INSTS_VALU"


def preset_traffic(preset, workload, size, batch):
    """HBM bytes per launch from the committed PMC profile of EXACTLY this launch shape (profiles/traffic.json; FETCH_SIZE
    and WRITE_SIZE from separate rocprofv3 --pmc passes, scripts/summarize_profiles.py), else None."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as f:
        t = json.load(f)
    e = (t.get("presets") or {}).get(preset or "")
    if e and e.get("workload") == workload and e.get("size") == size and e.get("textures_per_launch") == batch:
        return e["hbm_bytes_per_launch"], "profiles/traffic.json presets.%s (%s): FETCH_SIZE * 2 + WRITE_SIZE of the committed rocprofv3 " \
            "--pmc passes of this launch shape (not measured in this run)" % (preset, e.get("profile", "?"))
    if (size, batch) == (4096, 16) and workload in t:
        return t[workload], "profiles/traffic.json: FETCH_SIZE * 2 + WRITE_SIZE of the committed rocprofv3 --pmc passes of this " \
            "workload (not measured in this run)"
    return None, None


def measured_or_committed_traffic(live, preset, workload, size, batch, content, strategy):
    """(traffic, source, committed): measured in this run where allowed and possible, else the committed profile's figure."""
    committed, csource = preset_traffic(preset, workload, size, batch) if content == "noise" else (None, None)
    if live:
        t, source = live_traffic(workload, size, batch, content, strategy)
        if t is not None:
            return t, source, committed
        return committed, "%s; live measurement unavailable (%s)" % (csource, source) if csource else "unavailable: %s" % source, committed
    return committed, csource, committed


def preset_leg(ctx, pkg, sharding, name, steps, content="noise", verify=True, gather=True, probe=None, live=False):
    """One BASELINE configuration other than the headline one, as a compact object for the `configs` field of the line:
    the same timed region (K steps between barriers, MAX over ranks), roofline of the dominant kernel from HIP events,
    parity of texture 0 against the oracle, and for N > 1 the gather of the compressed output to rank 0."""
    torch = ctx.torch
    cfg = CONFIGS[name]
    codec, comps, bytes_per_px, label, limiting_unit = WORKLOADS[cfg["workload"]]
    size, strategy = cfg["size"], cfg.get("etc_strategy", 2)
    if cfg.get("total_textures"):
        t_begin, t_end = sharding.texture_range(cfg["total_textures"], ctx.world, ctx.rank)
        batch, scaling = t_end - t_begin, "strong"
    else:
        batch, scaling = cfg["batch"], "weak"
    src = make_batch(torch, content, batch, size, comps, ctx.device, seed=1000 + ctx.rank)
    per = pkg.encoded_size(codec, size, size)
    outs = [torch.empty((batch, per), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
    stream = ctx.current_stream()

    def step(out=None):
        r = pkg.encode_device(codec, src, size, size, comps, etc_strategy=strategy, n_images=batch,
                              out=outs[0] if out is None else out, stream=stream)
        assert r is not None
    elapsed, kernel_ms = timed_steps(ctx, step, steps, 3, 0.3, stream)
    px_rank = batch * size * size
    px_all = ctx.sum_over_ranks(float(px_rank))
    res = {"workload": "BASELINE config %s: %s" % (name, cfg["text"]), "codec": cfg["workload"], "texture": [size, size],
           "textures_per_gpu_per_step": batch, "etc_strategy": strategy if codec == 2 else None, "scaling": scaling,
           "steps": steps, "value": round(px_all * steps / elapsed / 1e6, 1), "unit": "Mpixels/s",
           "ms_per_step": round(elapsed / steps * 1e3, 4), "data": "synthetic (%s)" % content}
    if ctx.distributed and gather:
        try:
            counts = [e - b for b, e in (sharding.texture_range(cfg["total_textures"], ctx.world, r) for r in range(ctx.world))] \
                if cfg.get("total_textures") else [batch] * ctx.world
            res.update(gather_region(ctx, sharding, lambda slot: step(outs[slot]), outs, counts, steps, stream, px_all, codec,
                                     probe))
        except Exception as e:
            res.update({"value_with_gather": None, "gather_error": "%s: %s" % (type(e).__name__, e)})
    if ctx.rank == 0:
        algo = px_rank * bytes_per_px
        achieved = algo / (kernel_ms * 1e-3) / 1e9
        traffic, source, committed = measured_or_committed_traffic(live and ctx.world == 1 and ctx.on_gpu, name, cfg["workload"], size,
                                                                   batch, content, strategy)
        res["roofline"] = {"bound": limiting_unit, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": source,
                           "traffic_committed_profile": committed,
                           "kernel": pkg.kernel_name(codec, comps), "kernel_ms": round(kernel_ms, 4),
                           "algorithmic_bytes_per_launch": int(algo)}
        # the clock this figure was taken at (DXT5 moves 835 <-> 1 264 Gpix/s with it, r04) and the VALU issue fraction
        mhz, window_ms = clock_under_load(torch, pkg, step, kernel_ms, max(steps, int(30.0 / max(kernel_ms, 1e-3))), stream) \
            if ctx.on_gpu else (None, None)
        res["roofline"]["effective_clock_MHz"] = mhz
        res["roofline"]["clock_window_kernel_ms"] = None if window_ms is None else round(window_ms, 4)
        # valu_frac: instructions x issue clocks over (SIMDs x clock x time) with clock AND time of the same launches
        vf = valu_fraction(cfg["workload"], content, strategy, codec, px_rank, window_ms if (mhz and window_ms) else kernel_ms, mhz,
                           preset=name, live_insts=LIVE_VALU_INSTS.get((cfg["workload"], size, batch, content, strategy)),
                           live_cycles=LIVE_GUI_CYCLES.get((cfg["workload"], size, batch, content, strategy)))
        if vf:
            res["roofline"].update({k: vf[k] for k in ("valu_frac", "valu_wave_insts_per_block", "valu_profile", "valu_clock_MHz",
                                                       "simd_cycles_per_valu_inst", "model_issue_clk_per_valu_inst") if k in vf})
        if verify:
            import ic_testlib as T
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
            step(outs[0])
            ctx.sync()
            host0 = src[0].cpu().numpy()
            want = T.oracle_encode(codec, host0, size, size, comps, 0, strategy, threads=1 if codec in (3, 4) else cores)
            res["parity"] = "bit-exact vs oracle (texture 0 of the batch)" if outs[0][0].cpu().numpy().tobytes() == want \
                else "MISMATCH vs oracle"
            if codec == 4:  # no reference exists: the oracle is a second form of the same rules, the decode the only outside check
                import numpy as np
                dec = T.oracle_decode(4, want, size, size).reshape(size, size, 4).astype(np.float64)
                mse = float(((dec - host0.astype(np.float64)) ** 2).mean())
                res["parity"] += " -- EXTENSION, parity UNPINNED (no reference implementation of PVRTC 4 bpp)"
                res["round_trip_psnr_db"] = None if mse == 0 else round(10.0 * math.log10(255.0 * 255.0 / mse), 2)
    del src, outs
    if ctx.on_gpu:
        torch.cuda.empty_cache()
    return res


def gather_region(ctx, sharding, step_into, outs, counts, steps, stream, pixels_per_step_all, codec, probe=None):
    """encode -> gather of the compressed output on rank 0, the gather of batch k on a second stream underneath
    the encode of batch k + 1 (double-buffered).  step_into(slot) encodes into outs[slot]."""
    torch = ctx.torch
    comm = torch.cuda.Stream(device=ctx.device)
    gathered = [sharding.alloc_gather_buffers(outs[0], counts, ctx.rank) for _ in range(2)]
    enc_done = [torch.cuda.Event() for _ in range(2)]
    gat_done = [torch.cuda.Event() for _ in range(2)]

    def gather_async(slot):
        enc_done[slot].record(stream)
        with torch.cuda.stream(comm):
            comm.wait_event(enc_done[slot])
            sharding.gather_to_root(outs[slot], gathered[slot], counts, ctx.rank, host_staged=ctx.backend == "gloo", rccl=ctx.rccl)
            gat_done[slot].record(comm)

    for slot in range(2):  # warm-up: communicator set-up, both buffers touched
        step_into(slot)
        gather_async(slot)
    ctx.sync()
    ctx.barrier()
    g0 = time.perf_counter()
    gather_async(0)
    ctx.sync()
    ctx.barrier()
    gather_ms = ctx.max_over_ranks((time.perf_counter() - g0) * 1e3)
    ctx.sync()
    ctx.barrier()
    ctx.sync()
    t1 = time.perf_counter()
    for i in range(steps):
        slot = i & 1
        stream.wait_event(gat_done[slot])  # the gather that last read this output buffer has finished
        step_into(slot)
        gather_async(slot)                 # ... overlaps the encode of the next batch
    ctx.sync()
    ctx.barrier()
    ctx.sync()
    elapsed_g = ctx.max_over_ranks(time.perf_counter() - t1)
    ok = True
    if ctx.rank == 0:
        last = gathered[(steps - 1) & 1]
        ok = bool(torch.equal(last[0], outs[(steps - 1) & 1]))
    out_bytes_all = pixels_per_step_all / 4.0 if codec == 3 else pixels_per_step_all / 16.0 * (16 if codec == 1 else 8)  # (4 bpp = DXT1's rate)
    world = ctx.world
    bound = gather_bound(probe, out_bytes_all, world, pixels_per_step_all)
    return {**bound, "value_with_gather": round(pixels_per_step_all * steps / elapsed_g / 1e6, 1),
            "ms_per_step_with_gather": round(elapsed_g / steps * 1e3, 4), "gather_ms": round(gather_ms, 4),
            "gather_GBps_into_rank0": round(out_bytes_all * (world - 1) / world / (gather_ms * 1e-3) / 1e9, 2),
            "gather": "gather of the compressed output to rank 0 on a second stream, double-buffered, overlapping the "
                      "next batch's encode: %s" % ctx.gather_impl(),
            "rank0_copy_matches": ok}


def slab_leg(ctx, pkg, sharding, workload, size, steps, content="noise", verify=True, oracle_encode=None, probe=None,
             rotate_bytes=320 << 20, rotate_min=5, rotate_max=64):
    """ONE size x size image split into contiguous slabs of block rows over the ranks (blocks are stored
    row-major, compressor4x4_helper.h:202-214, so every slab's blocks are one contiguous byte range of the final buffer):
    strong scaling.  Every rank holds only its slab of the source -- of `n_rot` DISTINCT images (>= 5 and >= 320 MiB of slabs
    per rank, r05: one 64 MiB image, or a rank's 8 MiB share of it, re-encoded every step would be read from the 256 MiB
    Infinity Cache, not from HBM), step i encoding the slab of image i mod n_rot.  Timed: (1) K encode steps between barriers
    -> value; (2) K steps of encode -> gather of the slabs into rank 0's final buffer (views of ONE contiguous allocation;
    equal slabs: one dist.gather, unequal: batched isend / irecv), each step waiting for its gather -> value_with_gather, the
    latency-shaped figure for one image.  Parity: every rank checks its slab of image 0 against the oracle's encoding of the
    same rows (blocks are independent), rank 0 additionally checks the gathered image."""
    torch = ctx.torch
    codec, comps, bytes_per_px, label, limiting_unit = WORKLOADS[workload]
    block_bytes = 16 if codec == 1 else 8
    stride = size * comps
    geos = [sharding.slab_geometry(size, size, comps, stride, block_bytes, ctx.world, r) for r in range(ctx.world)]
    geo = geos[ctx.rank]
    rows, brows, cols = geo["pixel_rows"], geo["block_rows"], (size + 3) // 4
    # the same number of distinct images on every rank (rank 0's slab sets it; slabs differ by at most one block row)
    slab_bytes = max(1, geos[0]["pixel_rows"] * stride)
    n_rot = int(max(rotate_min, min(rotate_max, -(-rotate_bytes // slab_bytes))))
    src = make_batch(torch, content, n_rot, size, comps, ctx.device, seed=2000 + ctx.rank, height=max(rows, 4), row0=geo["pixel_row0"])
    out = torch.empty((max(brows, 1), cols * block_bytes), dtype=torch.uint8, device=ctx.device)[:brows]
    stream = ctx.current_stream()
    turn = [0]

    def step(k=None):
        if rows == 0:
            return
        if k is None:
            k = turn[0] % n_rot
            turn[0] += 1
        r = pkg.encode_device(codec, src[k], rows, size, comps, n_images=1, out=out.view(1, -1), stream=stream)
        assert r is not None
    elapsed, kernel_ms = timed_steps(ctx, step, steps, 3, 0.1, stream)
    px = float(size) * size
    res = {"image": [size, size], "codec": workload, "shard": "block-row slabs (sharding.slab_geometry), one per rank",
           "slab_block_rows": [g["block_rows"] for g in geos], "scaling": "strong", "steps": steps,
           "distinct_images_rotated": n_rot, "distinct_source_MiB_per_rank": (n_rot * rows * stride) >> 20,
           "value": round(px * steps / elapsed / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(elapsed / steps * 1e3, 4)}
    counts = [g["block_rows"] for g in geos]
    final = torch.empty((sum(counts), cols * block_bytes), dtype=torch.uint8, device=ctx.device) if ctx.rank == 0 else None
    bufs = None
    if ctx.rank == 0:
        offs = [g["block_row0"] for g in geos]
        bufs = [final[o:o + c] for o, c in zip(offs, counts)]  # contiguous row ranges of the final image

    def encode_and_gather(k=None):
        step(k)
        sharding.gather_to_root(out, bufs, counts, ctx.rank, host_staged=(ctx.backend == "gloo" and ctx.on_gpu), rccl=ctx.rccl)
    try:
        elapsed_g, _ = timed_steps(ctx, encode_and_gather, steps, 2, 0.0, stream)
        res.update(gather_bound(probe, float(sum(counts)) * cols * block_bytes, ctx.world, px))
        res.update({"value_with_gather": round(px * steps / elapsed_g / 1e6, 1),
                    "ms_per_step_with_gather": round(elapsed_g / steps * 1e3, 4),
                    "gather": "slabs -> rank 0's final buffer (contiguous views), %s, not overlapped: one image's latency"
                              % ("one rank: device copy" if (ctx.world == 1 and ctx.rccl is None) else ctx.gather_impl())})
    except Exception as e:
        res.update({"value_with_gather": None, "gather_error": "%s: %s" % (type(e).__name__, e)})
    if ctx.rank == 0 and kernel_ms:
        algo = float(rows) * size * bytes_per_px
        res["roofline_rank0"] = {"achieved": round(algo / (kernel_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                 "frac": round(algo / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "kernel_ms": round(kernel_ms, 5)}
    if verify:
        ok = 1.0
        gathered = res.get("value_with_gather") is not None
        try:  # image 0 once more, untimed: what the checks below look at (every rank takes part in the gather)
            if gathered:
                encode_and_gather(0)
            else:
                step(0)
            ctx.sync()
        except Exception:
            ok = 0.0
        if rows and ok:
            if oracle_encode is None:
                import ic_testlib as T
                cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
                oracle_encode = lambda a, h, w: T.oracle_encode(codec, a, h, w, comps, threads=max(1, cores // max(1, ctx.world)))  # noqa: E731
            want = oracle_encode(src[0, :rows].cpu().numpy(), rows, size)
            ok = 1.0 if out.cpu().numpy().tobytes() == want else 0.0
            if ctx.rank == 0 and final is not None and ok and gathered:
                ok = 1.0 if final[:brows].cpu().numpy().tobytes() == want else 0.0
        # the slabs of the OTHER ranks inside rank 0's gathered image: position-weighted byte checksums, compared on rank 0
        def checksum(t):
            v = t.reshape(-1).to(torch.int64)
            return float(((v + 1) * (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 8191 + 1)).sum().item() % (1 << 52))
        mine = checksum(out) if brows else 0.0
        if ctx.distributed and gathered:
            sums = torch.zeros(ctx.world, dtype=torch.float64, device=ctx.device if (ctx.backend == "nccl" and ctx.on_gpu) else "cpu")
            sums[ctx.rank] = mine
            ctx.dist.all_reduce(sums, op=ctx.dist.ReduceOp.SUM)
            if ctx.rank == 0:
                for r in range(ctx.world):
                    if counts[r] and checksum(bufs[r]) != float(sums[r].item()):
                        ok = 0.0
        ok = ctx.min_over_ranks(ok)
        res["parity"] = "bit-exact vs oracle (every rank's slab of image 0); gathered image on rank 0 checked slab by slab (checksums)" \
            if ok == 1.0 else "MISMATCH"
    del src, out, final, bufs
    if ctx.on_gpu:
        torch.cuda.empty_cache()
    return res


def next_rows_leg(timeout_s=300):
    """The SURVEY 8f "next" rows at 16 x 4096^2, device-resident, in the driver's line: scripts/bench_next_rows.py
    --core in a child process (its own buffers and failures), every leg with its algorithmic GB/s, fraction of 8 TB/s and the parity
    of image 0 of exactly the timed call against the oracle; `bars` are the round's stated targets and whether each leg meets them."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="icamd_next_rows_", suffix=".json", dir="/tmp")
    os.close(fd)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_next_rows.py"), "--core", "--json", path],
                           capture_output=True, text=True, timeout=timeout_s)
        with open(path) as f:
            d = json.load(f)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    legs = {x["leg"]: {k: x[k] for k in ("algorithmic_GBps", "frac", "ms_per_call", "parity", "one_launch")} for x in d["legs"]}
    bars = {"downsample dxt1 batched": 0.45, "downsample dxt5 batched": 0.45, "downsample dxt1": 0.30, "downsample dxt5": 0.30,
            "transcode dxt1->etc1": 0.25}
    return {"shape": d.get("shape"), "all_bit_exact": d.get("all_bit_exact"), "child_rc": r.returncode, "legs": legs,
            "bars": {k: {"target_frac": v, "frac": (legs.get(k) or {}).get("frac"),
                         "met": bool(legs.get(k) and legs[k]["frac"] >= v)} for k, v in bars.items()},
            "note": "frac = algorithmic bytes (blocks in + blocks / pixels out) / time / 8 TB/s; `downsample <codec>` without "
                    "`batched` is one icamd_downsample_device call per image, 16 calls per timed step"}


def setup_rccl_gather(ctx, args, pkg, sharding):
    """The library's own communicator for the gather legs (ctx.rccl).  Called after the headline measurement and under the
    watchdog: its set-up is itself a collective that has never run between two real GPUs.  All ranks or none: a rank that failed
    must not leave the others waiting in a grouped receive."""
    if args.gather_impl != "c":
        ctx.rccl_note = "--gather-impl torch"
        return
    if args.backend != "nccl":
        ctx.rccl_note = "backend %s (ranks share a GPU or run on the CPU: RCCL wants one GPU per rank)" % args.backend
        return
    ctx.leg = "rccl communicator set-up (icamd_rccl_comm_init)"
    try:
        ctx.rccl = sharding.make_rccl_gather(pkg, ctx.rank, ctx.world, ctx.device)
    except Exception as e:
        ctx.rccl, ctx.rccl_note = None, "%s: %s" % (type(e).__name__, str(e)[:200])
    ok_everywhere = ctx.min_over_ranks(1.0 if ctx.rccl is not None else 0.0)
    if ok_everywhere != 1.0 and ctx.rccl is not None:
        ctx.rccl.destroy()
        ctx.rccl, ctx.rccl_note = None, "another rank could not create its communicator"


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_distributed(args)
    # (the host driver only supports dmabuf IPC: without this RCCL's peer mappings fail with hipIpcGetMemHandle: invalid argument;
    # exported already by the boxes this runs on -- a launcher that scrubbed the environment must not cost the N > 1 run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import ic_amd_loader
    pkg = ic_amd_loader.load_package()
    from image_compression_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using the launcher's world size" % (args.gpus, world),
              file=sys.stderr)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the backend has no CPU path")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        sys.exit("bench.py: %d ranks but only %d GPU(s) visible (RCCL needs one GPU per rank)" % (world, n_dev))
    device = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(device)
    # --force-distributed: initialise the process group (RCCL) and run the collective paths even with ONE rank -- the only
    # way to execute them on a 1-GPU box
    distributed = world > 1 or args.force_distributed
    if args.force_distributed:
        os.environ["ICAMD_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:  # a single forced rank, not launched by torch.distributed.run: any free port will do
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            else:  # ranks started by hand (RANK / WORLD_SIZE set, no launcher) must agree on one port
                os.environ["MASTER_PORT"] = "29500"
        import datetime
        tmo = datetime.timedelta(seconds=300)  # a wedged collective fails the run instead of hanging it
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, timeout=tmo)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo", timeout=tmo)

    ctx = Ctx(torch, dist, rank, world, device, distributed, args.backend)
    barrier = ctx.barrier

    if args.shard == "slab":  # ONE large image over the ranks: the headline line of this mode
        probe = link_probe(ctx) if distributed else None
        if distributed:
            setup_rccl_gather(ctx, args, pkg, sharding)
        res = slab_leg(ctx, pkg, sharding, args.workload, args.size, args.steps, args.content, verify=not args.no_verify, probe=probe)
        res["link_probe"] = probe
        if rank == 0:
            codec, comps, bytes_per_px, label, limiting_unit = WORKLOADS[args.workload]
            tex = "%dx%d %s" % (args.size, args.size, "RGBA8" if comps == 4 else "RGB888")
            line = {"metric": "Mpixels/s encode (%s, one %s image in block-row slabs)" % (label, tex), "value": res["value"],
                    "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": 3, "ms_per_step": res["ms_per_step"],
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32",
                    "data": "synthetic (%s, seeded, generated on device)" % args.content,
                    "config": {"workload": "%s encode of ONE %s image, block-row slabs over %d rank(s), device-resident" % (label, tex, world),
                               "codec": args.workload, "texture": [args.size, args.size], "parallelism": res["shard"],
                               "world_size": world, "visible_gpus": n_dev, "kernel": pkg.kernel_name(codec, comps)}}
            line.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step", "unit", "steps", "scaling")})
            print(json.dumps(line))
        if ctx.rccl is not None:
            ctx.rccl.destroy()
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    codec, comps, bytes_per_px, label, limiting_unit = WORKLOADS[args.workload]
    size = args.size
    if args.total_textures:  # the configuration fixes the whole job: this rank's contiguous share of it
        t_begin, t_end = sharding.texture_range(args.total_textures, world, rank)
        batch, scaling = t_end - t_begin, "strong"
    else:
        t_begin, batch, scaling = rank * args.batch, args.batch, "weak"
    src = make_batch(torch, args.content, batch, size, comps, device, seed=rank)
    per_image_out = pkg.encoded_size(codec, size, size)
    outs = [torch.empty((batch, per_image_out), dtype=torch.uint8, device=device) for _ in range(2)]
    stream = torch.cuda.current_stream()

    def step(out):
        r = pkg.encode_device(codec, src, size, size, comps, etc_strategy=args.etc_strategy, n_images=batch, out=out,
                              stream=stream)
        assert r is not None

    if args.traffic_child:  # live_traffic()'s child under rocprofv3 --pmc: three launches of exactly this workload, no output
        for _ in range(3):
            step(outs[0])
        torch.cuda.synchronize()
        return

    # ---- pre-conditioning (untimed, disclosed in the output line): the chip needs some hundred milliseconds of load
    # to leave its idle power state -- measured r03: the first launches after an idle period run 10-30 % slower than the
    # steady state of the same kernel (`sustained.median_ms_ramp`).  A caller streaming textures lives in the steady
    # state, so the K timed steps are taken there; the ramp itself is reported by the sustained leg.
    precondition_launches = 0
    if args.precondition_seconds > 0:
        tp = time.perf_counter()
        while time.perf_counter() - tp < args.precondition_seconds:
            for _ in range(50):
                step(outs[0])
            precondition_launches += 50
            torch.cuda.synchronize()
    # ---- timed region 1 (the contract): exactly K steps of the hot path between barriers
    for _ in range(args.warmup):
        step(outs[0])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # ONE event between consecutive steps (K + 1 in all): a step's duration is the distance between its event and the
    # next one, i.e. its kernel(s) plus the sub-microsecond dispatch gap.  (An event PAIR around every launch, as r01 / r02
    # had it, puts two timestamp packets between any two kernels and costs the timed region ~5 us per step.)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record(stream)
        step(outs[0])
    marks[args.steps].record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = marks[0].elapsed_time(marks[args.steps]) / args.steps

    elapsed = ctx.max_over_ranks(elapsed)
    pixels_per_step_rank = batch * size * size
    pixels_per_step_all = ctx.sum_over_ranks(float(pixels_per_step_rank))
    value = pixels_per_step_all * args.steps / elapsed / 1e6

    tex = "%dx%d %s" % (size, size, "RGBA8" if comps == 4 else "RGB888")
    result = {
        "metric": "Mpixels/s encode (%s, %s)" % (label, tex),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "int32", "data": "synthetic (%s, seeded, generated on device)" % args.content,
        "preconditioning": {"seconds": args.precondition_seconds, "untimed_launches_before_warmup": precondition_launches},
        "config": {"workload": ("BASELINE config %s: %s; " % (args.preset, CONFIGS[args.preset]["text"]) if args.preset else "")
                               + "%s encode of %d x %s textures per GPU per step (one launch), device-resident"
                               % (label, batch, tex),
                   "preset": args.preset, "codec": args.workload, "textures_per_gpu_per_step": batch,
                   "textures_per_step_all_gpus": int(round(pixels_per_step_all / (size * size))),
                   "texture": [size, size], "src_bytes_per_pixel": comps,
                   "etc_strategy": args.etc_strategy if codec == 2 else None,
                   "parallelism": "independent textures per GPU, texture_range per rank (no data-path collective)",
                   "world_size": dist.get_world_size() if distributed else 1, "visible_gpus": n_dev,
                   "kernel": pkg.kernel_name(codec, comps)},
    }
    if rank == 0:  # (completed below; present from here on so that a line printed by the watchdog carries it)
        a0 = pixels_per_step_rank * bytes_per_px / (kernel_ms * 1e-3) / 1e9
        t0_, src0_ = preset_traffic(args.preset, args.workload, size, batch) if args.content == "noise" else (None, None)
        result["roofline"] = {"bound": limiting_unit, "achieved": round(a0, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(a0 / HBM_PEAK_GBPS, 4), "traffic": t0_, "traffic_source": src0_,
                              "kernel": pkg.kernel_name(codec, comps), "kernel_ms": round(kernel_ms, 4)}
    # ---- watchdog: the contract's measurement is done; nothing below may cost the run its line.  Exceptions are caught leg by
    # leg, but a collective that HANGS (the multi-rank paths have only ever run over gloo / one forced RCCL rank) would hold the
    # line back until the launcher's own limit.  After --watchdog-seconds rank 0 prints what it has, every rank exits 0.
    import threading
    printed = threading.Lock()

    def watchdog_fire():
        if printed.acquire(False):
            # every rank says on stderr which leg it was in (a deadlocked collective must leave a trace besides the
            # `watchdog` field); the exit status stays 0 on purpose -- a non-zero rank would make the launcher tear the group down,
            # possibly before rank 0's line is out, and the line itself says that it is incomplete
            print("bench.py: rank %d: watchdog fired %.0f s after the headline measurement, still in leg '%s'"
                  % (rank, args.watchdog_seconds, ctx.leg), file=sys.stderr, flush=True)
            if rank == 0:
                result["watchdog"] = ("extra legs still running %.0f s after the headline measurement (rank 0 was in leg '%s'): line printed "
                                      "without the unfinished ones" % (args.watchdog_seconds, ctx.leg))
                text = None
                for _ in range(5):  # the main thread may be adding a key at this very moment: try again rather than lose the legs
                    try:
                        text = json.dumps(dict(result), default=str)
                        break
                    except Exception:
                        time.sleep(0.05)
                if text is None:
                    text = json.dumps({k: v for k, v in result.items() if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                                      "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "watchdog")},
                                      default=str)
                print(text, flush=True)
            os._exit(0)
    watchdog = None
    if args.watchdog_seconds > 0:
        watchdog = threading.Timer(args.watchdog_seconds + (0.0 if rank == 0 else 5.0), watchdog_fire)
        watchdog.daemon = True
        watchdog.start()
    # ---- timed region 2 (N > 1): encode -> RCCL gather of the compressed output on rank 0, overlapped
    gather = None
    probe = None
    if distributed and not args.no_gather:
        setup_rccl_gather(ctx, args, pkg, sharding)
    if distributed and not args.no_gather:
        ctx.leg = "link probe"
        try:
            probe = link_probe(ctx)
        except Exception as e:
            probe = None
            print("bench.py: link probe failed: %s: %s" % (type(e).__name__, e), file=sys.stderr)
        ctx.leg = "headline gather region"
        try:
            counts = [sharding.texture_range(args.total_textures, world, r) for r in range(world)] if args.total_textures \
                else [(r * batch, (r + 1) * batch) for r in range(world)]
            counts = [e - b for b, e in counts]
            gather = gather_region(ctx, sharding, lambda slot: step(outs[slot]), outs, counts, args.steps, stream,
                                   pixels_per_step_all, codec, probe)
        except Exception as e:  # the encode-only line must survive a failing gather (it is reported, not hidden)
            gather = {"value_with_gather": None, "gather_error": "%s: %s" % (type(e).__name__, e)}

    if gather is not None:
        result.update(gather)
    if distributed and not args.no_gather:
        result["gather_impl"] = ctx.gather_impl()
    ctx.leg = "rank 0's single-GPU legs (traffic, parity, cpu baseline, host api, sustained, single image)"
    result["link_probe"] = probe
    result["scaling_headline"] = (
        "`value` (every rank's compressed output stays in the HBM of the GPU that made it -- the texture pipeline's normal case) "
        "is the figure to compute scaling efficiency from.  `value_with_gather` additionally moves every rank's blocks into "
        "rank 0 (SURVEY 8d): (N-1)/N of the output enters ONE GPU through its min(N-1, 7) inbound xGMI links, so it is bound by "
        "`gather_bound_GBps` (measured here by link_probe; null on one rank) at `value_with_gather_ceiling` whatever the encoders "
        "do, and for HBM-bound DXT1 that ceiling lies BELOW the 1-GPU `value`.")

    if rank == 0:
        algo_bytes = pixels_per_step_rank * bytes_per_px
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_source, traffic_committed = measured_or_committed_traffic(
            world == 1 and not args.no_live_traffic, args.preset, args.workload, size, batch, args.content, args.etc_strategy)
        hbm_frac = round(achieved / HBM_PEAK_GBPS, 4)
        result["roofline"] = {
            "bound": limiting_unit, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": hbm_frac, "hbm_frac": hbm_frac, "valu_frac": None, "traffic": traffic,
            "kernel": pkg.kernel_name(codec, comps), "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_pixel": bytes_per_px, "algorithmic_bytes_per_launch": int(algo_bytes),
            "read_roofline_frac": round((pixels_per_step_rank * comps / (kernel_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS, 4),
        }
        result["roofline"]["traffic_source"] = traffic_source
        result["roofline"]["traffic_committed_profile"] = traffic_committed
        import ic_testlib as T
        host0 = src[0].cpu().numpy()
        if not args.no_verify:
            got = outs[0][0].cpu().numpy().tobytes()
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
            want = T.oracle_encode(codec, host0, size, size, comps, 0, args.etc_strategy,
                                   threads=1 if codec in (3, 4) else cores)
            result["parity"] = "bit-exact vs oracle (texture 0 of the batch)" if got == want else "MISMATCH vs oracle"
            if got != want:
                print(json.dumps(result))
                sys.exit("parity check failed")
            # informational: quality of the encoded texture 0, decoded again on the device
            try:
                dcomps = 4 if codec in (1, 3, 4) else 3
                if codec == 4:
                    result["parity"] += " -- EXTENSION, parity UNPINNED (no reference implementation of PVRTC 4 bpp)"
                dec = pkg.decode_device(codec, outs[0][0].contiguous(), size, size)
                a = dec.view(size, size, dcomps).to(torch.float64)
                b = src[0][..., :dcomps].to(torch.float64)
                mse = float(((a - b) ** 2).mean())
                result["psnr_db"] = None if mse == 0 else round(10.0 * math.log10(255.0 * 255.0 / mse), 2)
            except Exception as e:  # never let the informational figure take the benchmark line down
                result["psnr_db"] = "unavailable: %s" % e
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(T, codec, comps, size, args.etc_strategy, host0)
        if not args.no_host_api and world == 1:
            try:
                result["host_api"] = host_api_leg(pkg, T, codec, size, args.etc_strategy)
            except Exception as e:
                result["host_api"] = "unavailable: %s" % e
            if isinstance(result["host_api"], dict):
                try:
                    result["host_api"]["batch"] = host_api_batch_leg(pkg, T)
                except Exception as e:
                    result["host_api"]["batch"] = "unavailable: %s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_sustained:
            try:
                sus, clock = sustained_leg(torch, pkg, lambda i: step(outs[0]), stream, kernel_ms, args.sustained_seconds,
                                           algo_bytes)
                result["sustained"] = sus
                result["clock"] = clock
                result["roofline"]["sustained_frac"] = sus["frac_last_20pct"]
                result["roofline"]["sustained_kernel_ms"] = sus["median_ms_last_20pct"]
                mhz = (clock.get("last_25pct") or {}).get("shader_MHz")
                if mhz:
                    result["roofline"]["effective_clock_MHz"] = mhz
                vf = valu_fraction(args.workload, args.content, args.etc_strategy, codec, pixels_per_step_rank,
                                   sus["median_ms_last_20pct"], mhz,
                                   live_insts=LIVE_VALU_INSTS.get((args.workload, size, batch, args.content, args.etc_strategy)),
                                   live_cycles=LIVE_GUI_CYCLES.get((args.workload, size, batch, args.content, args.etc_strategy)))
                if vf:
                    result["roofline"].update(vf)
            except Exception as e:
                result["sustained"] = "unavailable: %s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_single_image:
            try:
                result["single_image"] = single_image_leg(torch, pkg, codec, comps, src, size, batch, args.etc_strategy,
                                                          stream, bytes_per_px)
            except Exception as e:
                result["single_image"] = "unavailable: %s: %s" % (type(e).__name__, e)
        if world == 1 and args.preset == "c2" and args.content == "noise" and not args.no_next_rows:
            ctx.leg = "next_rows"
            result["next_rows"] = next_rows_leg()
        result["library"] = {"path": os.path.relpath(pkg.LIB_PATH, ROOT), "overridden": bool(pkg.LIB_OVERRIDDEN),
                             "version": pkg.lib().icamd_version().decode()}
    # ---- the other BASELINE configurations and the one-large-image slab legs, in the same line (every rank takes part).
    # Only the default line (preset c2, no overrides) carries them; each leg frees its buffers before the next.
    default_line = args.preset == "c2" and args.content == "noise"
    if default_line and not (args.no_extra_configs and args.no_slab):
        del src, outs
        torch.cuda.empty_cache()
        if not args.no_extra_configs:
            configs = {}
            for name in ("c3", "c4", "c5", "c5_8192", "c5_4bpp"):
                ctx.leg = "configs." + name
                try:
                    configs[name] = preset_leg(ctx, pkg, sharding, name, args.extra_steps, verify=not args.no_verify,
                                               gather=not args.no_gather, probe=probe, live=not args.no_live_traffic)
                except Exception as e:  # a failing extra leg is reported, it does not take the headline down
                    configs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                if name == "c4" and "error" not in configs[name]:
                    # the ETC1 search is content-dependent (smooth gradients clamp at base +- b: 27 % slower than the noise the
                    # configuration is quoted on; flat tiles take the one-colour forms): the same leg on the other two contents
                    other = {}
                    for content in ("smooth", "flat"):
                        ctx.leg = "configs.c4.other_contents." + content
                        try:
                            r = preset_leg(ctx, pkg, sharding, name, max(2, args.extra_steps // 2), content=content,
                                           verify=not args.no_verify, gather=False, live=not args.no_live_traffic)
                            other[content] = {k: r.get(k) for k in ("value", "unit", "ms_per_step", "steps", "parity", "data")}
                            rf = r.get("roofline") or {}
                            other[content].update({k: rf.get(k) for k in ("frac", "valu_frac", "effective_clock_MHz", "kernel_ms", "clock_window_kernel_ms",
                                                                           "valu_wave_insts_per_block", "valu_profile", "traffic",
                                                                           "simd_cycles_per_valu_inst")})
                        except Exception as e:
                            other[content] = {"error": "%s: %s" % (type(e).__name__, e)}
                    configs[name]["other_contents"] = other
            result["configs"] = configs
        if not args.no_slab:
            slabs = {}
            for key, wl, sz in (("c2_one_4096", "dxt1_rgba8", 4096), ("c3_one_8192", "dxt5_rgba8", 8192), ("one_16384", "dxt1_rgba8", 16384)):
                ctx.leg = "slab." + key
                try:
                    slabs[key] = slab_leg(ctx, pkg, sharding, wl, sz, args.extra_steps, verify=not args.no_verify, probe=probe)
                except Exception as e:
                    slabs[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            result["slab"] = slabs
    ctx.leg = "done"
    if not printed.acquire(False):  # the watchdog is printing: let it finish
        time.sleep(30)
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.rccl is not None:
        try:
            ctx.rccl.destroy()
        except Exception as e:
            print("bench.py: rank %d: icamd_rccl_comm_destroy: %s" % (rank, e), file=sys.stderr)
    if distributed:
        if args.watchdog_seconds > 0:  # the final barrier must not hang the exit either
            bye = threading.Timer(60.0, lambda: os._exit(0))
            bye.daemon = True
            bye.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:  # a peer that already left (its watchdog): the line is out, leaving is all that remains
            print("bench.py: rank %d: final barrier: %s: %s" % (rank, type(e).__name__, str(e)[:200]), file=sys.stderr)


if __name__ == "__main__":
    main()
