#!/usr/bin/env python3
"""Summarises gpurun_out/prof_next/ (scripts/gpu_profile_next_rows.sh) into profiles/<round>_next_rows_summary.json:
per icamd_* kernel of the "next" rows -- launches, mean duration, HBM bytes (FETCH_SIZE * 2 + WRITE_SIZE in KiB, the
gfx950 correction of MI355X_MICROARCH.md's HBM section, as scripts/summarize_profiles.py), executed VALU wave-instructions
per lane, SQ busy / wait fractions."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_next")
csv.field_size_limit(1 << 30)


def counters(sub):
    path = os.path.join(SRC, sub, "next_counter_collection.csv")
    agg = collections.defaultdict(list)
    grid = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"].startswith("icamd_"):
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
                grid[r["Kernel_Name"]] = int(r["Grid_Size"])
    return {k: sum(v) / len(v) for k, v in agg.items()}, grid


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    stats = {}
    for r in csv.DictReader(open(os.path.join(SRC, "trace", "next_kernel_stats.csv"))):
        if r["Name"].startswith("icamd_"):
            stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                "min_us": round(float(r["MinNs"]) / 1e3, 2)}
    fetch, _ = counters("pmc_fetch")
    write, _ = counters("pmc_write")
    sq, grid = counters("pmc_sq")
    out = {"command": "scripts/gpu_profile_next_rows.sh (rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE / SQ_* "
                      "in separate passes) -- python scripts/bench_next_rows.py (16 x 4096^2, noise)", "kernels": {}}
    for k, s in sorted(stats.items()):
        e = dict(s)
        f, w = fetch.get((k, "FETCH_SIZE")), write.get((k, "WRITE_SIZE"))
        if f is not None and w is not None:
            e["hbm_read_bytes"] = int(f * 1024 * 2)
            e["hbm_write_bytes"] = int(w * 1024)
            e["hbm_GBps"] = round((e["hbm_read_bytes"] + e["hbm_write_bytes"]) / (s["avg_us"] * 1e-6) / 1e9, 1)
        v, waves = sq.get((k, "SQ_INSTS_VALU")), sq.get((k, "SQ_WAVES"))
        if v and waves:
            e["valu_wave_insts_per_lane"] = round(v / waves, 1)
            e["grid_lanes"] = grid.get(k)
            busy, wc = sq.get((k, "SQ_BUSY_CYCLES")), sq.get((k, "SQ_WAVE_CYCLES"))
            if wc:
                e["sq_wait_any_frac_of_wave_cycles"] = round(sq.get((k, "SQ_WAIT_ANY"), 0) / wc, 3)
                e["sq_active_valu_frac_of_wave_cycles"] = round(sq.get((k, "SQ_ACTIVE_INST_VALU"), 0) / wc, 3)
        out["kernels"][k] = e
    dst = os.path.join(ROOT, "profiles", "%s_next_rows_summary.json" % rnd)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, e in out["kernels"].items():
        print("%-36s %3d calls %8.1f us  read %6.1f MB write %6.1f MB  %7.1f GB/s  valu/lane %s wait %s" % (
            k, e["calls"], e["avg_us"], e.get("hbm_read_bytes", 0) / 1e6, e.get("hbm_write_bytes", 0) / 1e6,
            e.get("hbm_GBps", 0), e.get("valu_wave_insts_per_lane"), e.get("sq_wait_any_frac_of_wave_cycles")))


if __name__ == "__main__":
    main()
