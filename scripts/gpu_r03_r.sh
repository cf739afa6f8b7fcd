#!/bin/bash
# ETC1 kSmallerError, rocprofv3 kernel durations per library in $LIBS and content in $CONTENTS (alternating rounds)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2 3; do
for c in ${CONTENTS:-noise}; do
for lib in $LIBS; do
  n=$(basename $lib .so)
  rm -rf gpurun_out/abe_$n
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abe_$n -o x -- python bench.py --steps 40 --warmup 5 --precondition-seconds 0.5 --workload etc1_rgb888 --content $c ${EXTRA:-} --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-verify > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/abe_$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("icamd_etc1"): print("$c $lib round$round", r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
  rm -rf gpurun_out/abe_$n
done; done; done
