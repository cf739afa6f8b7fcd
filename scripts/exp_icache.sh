#!/bin/bash
# Runs ON THE GPU BOX (r06): does the 280 KB ETC1 kSmallerError kernel miss its instruction cache?  Lists the counters the box has,
# then SQC instruction-cache requests / hits / misses and the SQ instruction-fetch counters for config c4 on noise and smooth content.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/icache; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | sed 's/^[ \t]*//' | cut -c1-160 | sort -u | head -60 > $OUT/counters.txt
cat $OUT/counters.txt
run() {  # tag counters... -- workload args
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  d=/tmp/ic_$tag; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $d -o t -- python $OLDPWD/bench.py --traffic-child "$@" ) > $OUT/$tag.log 2>&1
  python - "$tag" $d <<'PY'
import csv, glob, sys, os
csv.field_size_limit(1 << 30)
tag, d = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("icamd_"):
            vals.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
print(tag, " | ".join("%s %s %.4g" % (k[0][6:30], k[1], sum(v) / len(v)) for k, v in sorted(vals.items())) or "NO DATA (see log)")
PY
}
for content in noise smooth; do
  run c4_${content}_a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU GRBM_GUI_ACTIVE -- --workload etc1_rgb888 --size 1024 --batch 1024 --content $content --etc-strategy 2
  run c4_${content}_b SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- --workload etc1_rgb888 --size 1024 --batch 1024 --content $content --etc-strategy 2
done
run c5_a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU GRBM_GUI_ACTIVE -- --workload pvrtc2_rgba8 --size 4096 --batch 16 --content noise --etc-strategy 2
run c3_a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU GRBM_GUI_ACTIVE -- --workload dxt5_rgba8 --size 8192 --batch 4 --content noise --etc-strategy 2
