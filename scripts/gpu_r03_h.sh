#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03h; rm -rf $O; mkdir -p $O
ab() {  # workload content lib...
  wl=$1; c=$2; shift 2
  for round in 1 2; do
    for lib in "$@"; do
      ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 100 --warmup 5 --workload $wl --content $c \
        --no-cpu-baseline --no-host-api --no-sustained --no-single-image 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $c $lib round$round', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity','')[:9])"
    done
  done
}
{ for c in noise smooth; do ab pvrtc2_rgba8 $c $LIBS; done; } 2>&1 | tee $O/ab.log
