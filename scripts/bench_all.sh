#!/bin/bash
# Runs ON THE GPU BOX: one clean (unprofiled) bench.py line per workload / content / ETC strategy and per BASELINE
# config preset -> gpurun_out/bench_all.jsonl
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/bench_all.jsonl; : > $O
for wl in dxt1_rgba8 dxt1_rgb888 dxt5_rgba8 etc1_rgb888 pvrtc2_rgba8; do
  for c in noise smooth flat; do
    python bench.py --steps 40 --warmup 5 --workload $wl --content $c --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-live-traffic 2>/dev/null | tail -1 >> $O
  done
done
for s in 0 1 3; do python bench.py --steps 40 --warmup 5 --workload etc1_rgb888 --etc-strategy $s --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-live-traffic 2>/dev/null | tail -1 >> $O; done
# the BASELINE presets exactly as the driver runs them (all legs: sustained, single image, clock, cpu baseline, host API)
# (c5_8192, r06: config 5's codec on 8192^2 textures = the one-pass kernel's halo form; --no-next-rows: the full next-row run follows below)
for cfg in c2 c3 c4 c5 c5_8192 c5_4bpp; do python bench.py --config $cfg --steps 20 --warmup 5 --no-next-rows 2>/dev/null | tail -1 >> $O; done
python - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(l)
    print("%-4s %-12s %-6s strat=%s  %9.0f Mpix/s  %.4f ms/step  kernel %.4f ms  %7.1f GB/s  valu_frac %s psnr %s %s" % (
        d["config"].get("preset"), d["config"]["codec"], d["data"].split("(")[1].split(",")[0], d["config"].get("etc_strategy"), d["value"],
        d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"].get("valu_frac"), d.get("psnr_db"), d["parity"][:9]),
        "| sustained", (d.get("sustained") or {}).get("median_ms_last_20pct"), (d.get("sustained") or {}).get("frac_last_20pct"),
        "clock", ((d.get("clock") or {}).get("last_25pct") or {}).get("shader_MHz"), "single", (d.get("single_image") or {}).get("median_ms_per_call"))
PY
# the "next"-row kernels, unprofiled and in the steady state (0.25 s of untimed calls per leg) -> profiles/<round>_next_rows.txt
python scripts/bench_next_rows.py > gpurun_out/next_rows.txt 2>&1; cat gpurun_out/next_rows.txt
