#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in $LIBS; do
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$(basename $lib .so) -o x -- python bench.py --steps 40 --warmup 5 --precondition-seconds 0.5 --workload pvrtc2_rgba8 --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-verify > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/abl_$(basename $lib .so)/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("icamd_"): print("$lib", r["Name"], r["Calls"], float(r["AverageNs"])/1e3)
PY
done
