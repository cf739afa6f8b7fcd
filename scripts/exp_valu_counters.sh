#!/bin/bash
# Runs ON THE GPU BOX (r06): which SQ / GRBM counters give a MEASURED VALU-busy fraction per kernel?
# One rocprofv3 pass per launch shape with SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
# GRBM_GUI_ACTIVE over bench.py's --traffic-child (three launches of exactly the workload), kernel trace alongside.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/valu_counters
mkdir -p $OUT
run() {  # name workload size batch content strategy
  d=/tmp/pmc_$1; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${CTRS:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} \
      --output-format csv -d $d -o t -- python $OLDPWD/bench.py --traffic-child --workload $2 --size $3 --batch $4 --content $5 --etc-strategy $6 ) > $OUT/$1.log 2>&1
  python - "$1" $d <<'PY'
import csv, glob, sys, os
csv.field_size_limit(1 << 30)
name, d = sys.argv[1], sys.argv[2]
vals, dur = {}, {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("icamd_"):
            vals.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("icamd_"):
            dur.setdefault(row["Kernel_Name"], []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k in sorted(dur):
    c = {cn: sum(v) / len(v) for (kn, cn), v in vals.items() if kn == k}
    ns = sum(dur[k]) / len(dur[k])
    line = "%s %s: %.1f us" % (name, k, ns / 1e3)
    for cn in sorted(c):
        line += " | %s %.4g" % (cn, c[cn])
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        g = c["GRBM_GUI_ACTIVE"] / 8
        line += " || clock %.0f MHz | SIMD cycles per VALU inst %.3f | resident waves per SIMD %.2f" % (
            g / ns * 1e3, 1024 * g / c["SQ_INSTS_VALU"], c.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * g))
    if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
        g = c["GRBM_GUI_ACTIVE"]
        line += " || clock(GRBM/ns) %.0f MHz (or /8: %.0f) | VALU busy = ACTIVE_INST_VALU*4/(1024*GRBM) %.3f (GRBM/8: %.3f) | ANY %.3f | vs time*2.4GHz: %.3f" % (
            g / ns * 1e3, g / 8 / ns * 1e3, c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g), c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g / 8),
            c.get("SQ_ACTIVE_INST_ANY", 0) * 4 / (1024 * g / 8), c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * ns * 2.4))
    print(line)
PY
}
if [ "$1" = "pvrtc" ]; then  # the two PVRTC one-pass kernels side by side: 2 bpp runs two waves per SIMD, 4 bpp four
  CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  run c5 pvrtc2_rgba8 4096 16 noise 2
  run c5_4bpp pvrtc4_rgba8 4096 16 noise 2
  run c4 etc1_rgb888 1024 1024 noise 2
  exit 0
fi
run c2 dxt1_rgba8 4096 16 noise 2
run c3 dxt5_rgba8 8192 4 noise 2
run c4 etc1_rgb888 1024 1024 noise 2
run c4_smooth etc1_rgb888 1024 1024 smooth 2
run c5 pvrtc2_rgba8 4096 16 noise 2
run etc1_heur etc1_rgb888 4096 16 noise 3
