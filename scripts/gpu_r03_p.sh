#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rocprofv3 --list-avail 2>/dev/null | grep -i -E "LDS|SQ_INSTS_|SQ_WAIT|SQ_ACTIVE|SQ_INST_CYCLES|VALU_BUSY|SQ_THREAD" | head -60
ARGS="--workload ${WL:-pvrtc2_rgba8} --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-verify --steps 20 --warmup 3 --precondition-seconds 0"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  rm -rf gpurun_out/pmcx
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcx -o x -- python bench.py $ARGS > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcx/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("icamd_"): per[(r["Kernel_Name"],r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
done
rm -rf gpurun_out/pmcx
