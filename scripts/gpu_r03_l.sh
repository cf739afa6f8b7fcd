#!/bin/bash
# A/B of PVRTC encode variants: kernel durations (rocprofv3 --stats) and HBM fetch bytes (separate --pmc pass) per library in $LIBS
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ARGS="--workload pvrtc2_rgba8 --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-verify"
for round in 1 2; do
for lib in $LIBS; do
  n=$(basename $lib .so)
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$n -o x -- python bench.py --steps 40 --warmup 5 --precondition-seconds 0.5 $ARGS > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/abl_$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("icamd_"): print("$lib round$round", r["Name"], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done; done
for lib in $LIBS; do
  n=$(basename $lib .so)
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/abf_$n -o x -- python bench.py --steps 20 --warmup 3 --precondition-seconds 0 $ARGS > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/abf_$n/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("icamd_") and r["Counter_Name"]=="FETCH_SIZE": per[(r["Kernel_Name"],r["Dispatch_Id"])]+=float(r["Counter_Value"])
    for (k,d),v in per.items(): acc[k].append(v)
for k,v in acc.items(): print("$lib", k, "FETCH_SIZE*2 per launch = %.4f GB" % (sum(v)/len(v)*2*1024/1e9), len(v))
PY
done
