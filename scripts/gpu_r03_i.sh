#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03i; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "etc or ETC or golden or encode_matches or c4 or smoke or transcode or pad or downsample" > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
ab() {  # workload content strategy lib...
  wl=$1; c=$2; st=$3; shift 3
  for round in 1 2; do
    for lib in "$@"; do
      ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 60 --warmup 5 --workload $wl --content $c --etc-strategy $st \
        --no-cpu-baseline --no-host-api --no-sustained --no-single-image 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $c s$st $lib round$round', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity','')[:9])"
    done
  done
}
{
for c in noise smooth flat; do ab etc1_rgb888 $c 2 $LIBS; done
for st in 0 3; do ab etc1_rgb888 smooth $st $LIBS; done
} 2>&1 | tee $O/ab.log
