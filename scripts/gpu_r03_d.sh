#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "pvrtc or golden or smoke or region" > $O/pytest_pvrtc.txt 2>&1; tail -3 $O/pytest_pvrtc.txt
ab() {  # workload content lib...
  wl=$1; c=$2; shift 2
  for round in 1 2 3; do
    for lib in "$@"; do
      ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 100 --warmup 5 --workload $wl --content $c \
        --no-cpu-baseline --no-host-api --no-sustained --no-single-image 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $c $lib round$round', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity','')[:9])"
    done
  done
}
{
for c in noise smooth flat; do ab pvrtc2_rgba8 $c ab_libs/lib_r02idx.so image-compression_amd/libic_amd.so; done
} 2>&1 | tee $O/ab_pvrtc_exchange.log
