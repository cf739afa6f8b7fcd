#!/usr/bin/env python3
"""Summarises gpurun_out/prof/<workload>/ (written on the GPU box by scripts/gpu_profile.sh) into
profiles/<round>_<workload>_{kernel_stats.csv,summary.json} and profiles/traffic.json.

HBM traffic = FETCH_SIZE * 2 + WRITE_SIZE (KiB -> bytes): on gfx950 this rocprofv3 reports exactly half
of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, section HBM), so the read side is
doubled; WRITE_SIZE is taken as reported.  FETCH_SIZE and WRITE_SIZE come from separate --pmc passes.
"""
import collections
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)  # torch's kernel names run to several KiB
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def counters(path):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith("icamd_"):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def source_version():
    """The version string the tree's library reports (icamd_version in csrc/ic_capi.hip)."""
    import re
    m = re.search(r'icamd_version\(void\)\s*\{\s*return\s*"([^"]+)"', open(os.path.join(ROOT, "image-compression_amd", "csrc", "ic_capi.hip")).read())
    return m.group(1) if m else None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rnd = args[0] if args else "r03"
    # r05: the tracked evidence must be of the shipped code object -- every profiled bench line carries the
    # version of the library that ran; a summary of any other build is refused
    want_version = source_version()
    stale = []
    for f in sorted(os.listdir(SRC)) if os.path.isdir(SRC) else []:
        if f.endswith(".bench.json"):
            try:
                got = (json.load(open(os.path.join(SRC, f))).get("library") or {}).get("version")
            except Exception:
                got = None
            if got != want_version:
                stale.append((f, got))
    if stale and "--allow-version-mismatch" not in sys.argv:
        sys.exit("summarize_profiles.py: profiles of another build (tree is %r): %s -- re-run scripts/gpu_profile.sh with the "
                 "current library" % (want_version, stale))
    os.makedirs(DST, exist_ok=True)
    tpath = os.path.join(DST, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    per_step = {}
    for wl in sorted(os.listdir(SRC)):
        d = os.path.join(SRC, wl)
        if not os.path.isdir(d) or wl.startswith("preset_"):
            continue
        stats_csv = os.path.join(d, "trace", wl + "_kernel_stats.csv")
        if not os.path.exists(stats_csv):
            continue
        rows = list(csv.DictReader(open(stats_csv)))
        out_csv = os.path.join(DST, "%s_%s_kernel_stats.csv" % (rnd, wl))
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([r["Name"][:96], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        summary = {"version": want_version, "workload": wl, "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 5 "
                   "--precondition-seconds 0.5 --workload %s --no-cpu-baseline --no-verify --no-host-api --no-sustained "
                   "--no-single-image" % wl, "kernels": {}}
        for r in rows:
            if r["Name"].startswith("icamd_"):
                summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                 "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                                 "note": "rocprofv3 --stats over ALL launches incl. preconditioning and warm-up"}
        # the timed launches only: the last 100 of each kernel in the per-launch timeline
        tl = os.path.join(d, "trace", wl + "_timeline.csv")
        if os.path.exists(tl):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(tl)):
                per[r["kernel"]].append(float(r["duration_us"]))
            with open(os.path.join(DST, "%s_%s_timeline_last100.csv" % (rnd, wl)), "w") as f:
                f.write("kernel,launch_from_end,duration_us\n")
                for k, v in per.items():
                    for i, x in enumerate(v[-100:]):
                        f.write("%s,%d,%.2f\n" % (k, len(v[-100:]) - i, x))
            for k, v in per.items():
                last = sorted(v[-100:])
                e = summary["kernels"].setdefault(k, {})
                e["timed_launches"] = {"n": len(last), "avg_us": round(sum(last) / len(last), 2), "median_us": last[len(last) // 2],
                                       "min_us": last[0], "max_us": last[-1], "launches_before_them": len(v) - len(last)}
        fetch = counters(os.path.join(d, "pmc_fetch", wl + "_counter_collection.csv"))
        write = counters(os.path.join(d, "pmc_write", wl + "_counter_collection.csv"))
        sq = counters(os.path.join(d, "pmc_sq", wl + "_counter_collection.csv"))
        for (k, c), v in list(fetch.items()) + list(write.items()) + list(sq.items()):
            summary["kernels"].setdefault(k, {})[c] = v
        for k, e in summary["kernels"].items():
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2  # gfx950 half-count correction
                e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
                e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
                # every kernel of a workload is launched once per bench step (PVRTC: morph + encode)
                per_step[wl] = per_step.get(wl, 0.0) + e["hbm_bytes_per_launch"]
            if "SQ_INSTS_VALU" in e and "SQ_WAVES" in e:
                e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        bj = os.path.join(SRC, wl + ".bench.json")
        if os.path.exists(bj):
            try:
                summary["bench_line_under_profiler"] = json.load(open(bj))
            except Exception:
                pass
        with open(os.path.join(DST, "%s_%s_summary.json" % (rnd, wl)), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        print(wl, json.dumps({k: {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in e.items()}
                              for k, e in summary["kernels"].items()})[:600])
    for wl, b in per_step.items():  # bytes per bench step (= per launch for the single-kernel workloads)
        traffic[wl] = int(b)
    # per BASELINE preset (bench.py's `configs` legs look their launch shape up here): c2 / c5 have the workload passes'
    # shape (16 x 4096^2); c3 / c4 come from their own passes (gpurun_out/prof/preset_cN)
    presets = traffic.setdefault("presets", {})
    shapes = {"c2": ("dxt1_rgba8", 4096, 16), "c3": ("dxt5_rgba8", 8192, 4), "c4": ("etc1_rgb888", 1024, 1024),
              "c5": ("pvrtc2_rgba8", 4096, 16), "c5_8192": ("pvrtc2_rgba8", 8192, 4)}  # (c5_8192, r06: the one-pass kernel's halo form)
    for cfg, (wl, size, n) in shapes.items():
        if cfg in ("c2", "c5"):
            if wl in per_step:
                presets[cfg] = {"workload": wl, "size": size, "textures_per_launch": n, "hbm_bytes_per_launch": int(per_step[wl]),
                                "profile": "%s, bench.py --workload %s" % (rnd, wl)}
            continue
        d = os.path.join(SRC, "preset_" + cfg)
        fetch = counters(os.path.join(d, "pmc_fetch", "preset_%s_counter_collection.csv" % cfg))
        write = counters(os.path.join(d, "pmc_write", "preset_%s_counter_collection.csv" % cfg))
        total = sum(v * 2048 for (k, c), v in fetch.items() if c == "FETCH_SIZE") + sum(v * 1024 for (k, c), v in write.items() if c == "WRITE_SIZE")
        if total > 0:
            presets[cfg] = {"workload": wl, "size": size, "textures_per_launch": n, "hbm_bytes_per_launch": int(total),
                            "profile": "%s, bench.py --config %s" % (rnd, cfg)}
    with open(tpath, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    # executed VALU instructions per workload / content / ETC strategy: what bench.py's roofline.valu_frac is built on
    # (it only reports the fraction when the key of the run matches a profile taken with exactly that configuration)
    vpath = os.path.join(DST, "valu_insts.json")
    valu = json.load(open(vpath)) if os.path.exists(vpath) else {}
    mpix = 16 * 4096 * 4096 / 1e6  # pixels of one bench step
    for tag in sorted(os.listdir(SRC)):
        d = os.path.join(SRC, tag)
        if not os.path.isdir(d) or tag.startswith("preset_"):
            continue
        if tag.startswith("presetsq_"):  # SQ pass of a BASELINE preset's own launch shape: presetsq_c4__<content>
            cfg, content = tag[len("presetsq_"):].split("__")
            wl, size, n = shapes[cfg]
            sq = counters(os.path.join(d, "pmc_sq", tag + "_counter_collection.csv"))
            insts = sum(v for (k, c), v in sq.items() if c == "SQ_INSTS_VALU")
            if insts > 0:
                pm = n * size * size / 1e6
                valu["preset:%s/%s/s2" % (cfg, content)] = {
                    "valu_wave_insts_per_Mpixel": insts / pm,
                    "per_kernel_valu_wave_insts_per_Mpixel": {k: v / pm for (k, c), v in sq.items() if c == "SQ_INSTS_VALU"},
                    "valu_wave_insts_per_block_lane": round(insts * 64.0 / (pm * 1e6 / 16.0), 1),
                    "profile": "%s: rocprofv3 --pmc SQ_INSTS_VALU, bench.py --config %s --content %s" % (rnd, cfg, content)}
            continue
        wl, content, strat = (tag.split("__") + ["noise", "s2"])[:3] if "__" in tag else (tag, "noise", "s2")
        sq = counters(os.path.join(d, "pmc_sq", tag + "_counter_collection.csv"))
        insts = sum(v for (k, c), v in sq.items() if c == "SQ_INSTS_VALU")
        if insts <= 0:
            continue
        per_kernel = {k: v / mpix for (k, c), v in sq.items() if c == "SQ_INSTS_VALU"}
        calls_per_step = 1.0
        if wl.startswith("pvrtc"):
            pass  # morph + encode: one launch each per step, summed
        key = "%s/%s/%s" % (wl, content, strat if wl.startswith("etc1") else "s0")
        lanes_per_block = 32.0 if wl.startswith("pvrtc") else 16.0  # pixels per block
        valu[key] = {"valu_wave_insts_per_Mpixel": insts * calls_per_step / mpix,
                     "per_kernel_valu_wave_insts_per_Mpixel": per_kernel,
                     "valu_wave_insts_per_block_lane": round(insts * 64.0 / (mpix * 1e6 / lanes_per_block), 1),
                     "profile": "%s: rocprofv3 --pmc SQ_INSTS_VALU, bench.py --workload %s --content %s%s" % (
                         rnd, wl, content, " --etc-strategy %s" % strat[1:] if wl.startswith("etc1") else "")}
    with open(vpath, "w") as f:
        json.dump(valu, f, indent=1, sort_keys=True)
    print("valu_insts.json:", {k: v["valu_wave_insts_per_block_lane"] for k, v in valu.items()})


if __name__ == "__main__":
    main()
