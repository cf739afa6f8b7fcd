#!/bin/bash
# Runs ON THE GPU BOX: gpu tests, then interleaved A/B of the DXT colour index search (r02 scan vs r03 thresholds)
# and of the RGB888 horizontal-pair variant; then the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
ab() {  # workload content lib...
  wl=$1; c=$2; shift 2
  for round in 1 2 3; do
    for lib in "$@"; do
      ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 200 --warmup 5 --workload $wl --content $c \
        --no-cpu-baseline --no-host-api --no-sustained --no-single-image 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $c $lib round$round', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity','')[:9])"
    done
  done
}
{
for wl in dxt5_rgba8 dxt1_rgba8 dxt1_rgb888; do
  for c in noise smooth flat; do
    ab $wl $c ab_libs/lib_r02idx.so image-compression_amd/libic_amd.so
  done
done
} 2>&1 | tee $O/ab_dxt_index.log
{
for c in noise smooth flat; do ab dxt1_rgb888 $c image-compression_amd/libic_amd.so ab_libs/lib_hpair.so; done
} 2>&1 | tee $O/ab_rgb888_hpair.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for wl in dxt1_rgb888 dxt5_rgba8 etc1_rgb888 pvrtc2_rgba8; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-host-api > $O/bench_$wl.json 2> $O/bench_$wl.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    s=d.get("sustained",{}); r=d["roofline"]
    print(f.split("/")[-1], d["value"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "| sustained", s.get("median_ms_ramp"), s.get("median_ms_last_20pct"), s.get("frac_last_20pct"), "| clock", d.get("clock"), "| single", d.get("single_image",{}).get("median_ms_per_call"))
PY
