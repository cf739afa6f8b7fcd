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
    """Averages per (kernel @ grid size, counter): the same kernel is launched per image and batched, which must not be mixed.
    Also returns the mean dispatch duration (ns) of each key under this pass."""
    path = os.path.join(SRC, sub, "next_counter_collection.csv")
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"].startswith("icamd_"):
                key = "%s @ %s lanes" % (r["Kernel_Name"], r["Grid_Size"])
                agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: (sum(v) / len(v), len(v)) for k, v in dur.items()}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    stats = {}
    for r in csv.DictReader(open(os.path.join(SRC, "trace", "next_kernel_stats.csv"))):
        if r["Name"].startswith("icamd_"):
            stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                "min_us": round(float(r["MinNs"]) / 1e3, 2)}
    fetch, dur = counters("pmc_fetch")
    write, _ = counters("pmc_write")
    sq, _ = counters("pmc_sq")
    out = {"command": "scripts/gpu_profile_next_rows.sh (rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE / SQ_* "
                      "in separate passes) -- python scripts/bench_next_rows.py (16 x 4096^2, noise)",
           "kernel_stats_unprofiled_passes": stats,
           "note": "per kernel AND grid size (the same kernel runs once per image and once per batch); avg_us here is the dispatch "
                   "duration under the FETCH_SIZE pass, the unperturbed wall numbers are scripts/bench_next_rows.py's own lines "
                   "(profiles/<round>_next_rows.txt)", "kernels": {}}
    for k in sorted(dur):
        d, n = dur[k]
        e = {"calls": n, "avg_us": round(d / 1e3, 2)}
        f, w = fetch.get((k, "FETCH_SIZE")), write.get((k, "WRITE_SIZE"))
        if f is not None and w is not None:
            e["hbm_read_bytes"] = int(f * 1024 * 2)
            e["hbm_write_bytes"] = int(w * 1024)
            e["hbm_GBps"] = round((e["hbm_read_bytes"] + e["hbm_write_bytes"]) / (e["avg_us"] * 1e-6) / 1e9, 1)
        v, waves = sq.get((k, "SQ_INSTS_VALU")), sq.get((k, "SQ_WAVES"))
        if v and waves:
            e["valu_wave_insts_per_lane"] = round(v / waves, 1)
            wc = sq.get((k, "SQ_WAVE_CYCLES"))
            if wc:
                e["sq_wait_any_frac_of_wave_cycles"] = round(sq.get((k, "SQ_WAIT_ANY"), 0) / wc, 3)
                e["sq_active_valu_frac_of_wave_cycles"] = round(sq.get((k, "SQ_ACTIVE_INST_VALU"), 0) / wc, 3)
                busy = sq.get((k, "SQ_BUSY_CYCLES"))
                if busy:
                    e["sq_busy_cycles"] = busy
        out["kernels"][k] = e
    dst = os.path.join(ROOT, "profiles", "%s_next_rows_summary.json" % rnd)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, e in out["kernels"].items():
        print("%-60s %3d calls %8.1f us  read %6.1f MB write %6.1f MB  %7.1f GB/s  valu/lane %s wait %s" % (
            k, e["calls"], e["avg_us"], e.get("hbm_read_bytes", 0) / 1e6, e.get("hbm_write_bytes", 0) / 1e6,
            e.get("hbm_GBps", 0), e.get("valu_wave_insts_per_lane"), e.get("sq_wait_any_frac_of_wave_cycles")))


if __name__ == "__main__":
    main()
