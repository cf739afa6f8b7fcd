#!/bin/bash
# Runs ON THE GPU BOX (r06): why are only 3.09 waves per SIMD resident on smooth content (3.86 on noise) in the ETC1 kSmallerError
# kernel?  The SPI's resource-allocation counters of config c4 per content.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/spi; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -E "SPI_" | sed 's/^[ \t]*//' | sort -u > $OUT/spi_counters.txt
wc -l $OUT/spi_counters.txt; tr '\n' ' ' < $OUT/spi_counters.txt | cut -c1-3000; echo
run() {
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  d=/tmp/spi_$tag; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $d -o t -- python $OLDPWD/bench.py --traffic-child "$@" ) > $OUT/$tag.log 2>&1
  python - "$tag" $d <<'PY'
import csv, glob, sys, os
csv.field_size_limit(1 << 30)
tag, d = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("icamd_"):
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
print(tag, " | ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(vals.items())) or "NO DATA")
PY
}
for content in noise smooth flat; do
  A="--workload etc1_rgb888 --size 1024 --batch 1024 --content $content --etc-strategy 2"
  run ${content}_1 SPI_RA_REQ_NO_ALLOC SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_TMP_STALL_CSN GRBM_GUI_ACTIVE -- $A
  run ${content}_2 SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN GRBM_GUI_ACTIVE -- $A
  run ${content}_3 SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN SPI_CSN_BUSY SPI_CSN_WAVE GRBM_GUI_ACTIVE -- $A
done
