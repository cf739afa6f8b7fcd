#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in $LIBS; do
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-host-api --no-sustained 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['single_image']; print('$lib round$round batch', d['roofline']['kernel_ms'], 'single', s['median_ms_per_call'], s['min_ms'], s['back_to_back_ms_per_call_wall'], d['parity'][:9])"
  for size in 1024 2048; do
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --workload pvrtc2_rgba8 --size $size --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-host-api --no-sustained 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['single_image']; print('$lib round$round size $size batch64', d['roofline']['kernel_ms'], 'single', s['median_ms_per_call'], s['min_ms'], d['parity'][:9])"
  done
done
done
