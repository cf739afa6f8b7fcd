#!/bin/bash
# ETC1 per content / strategy, libraries in $LIBS alternating (bench.py numbers, event-timed)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
for c in noise smooth flat; do
for lib in $LIBS; do
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 30 --warmup 5 --workload etc1_rgb888 --content $c --no-cpu-baseline --no-host-api --no-sustained --no-single-image 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('etc1 $c s2 $lib round$round %.4f ms' % d['ms_per_step'], d.get('parity', d.get('verify')))
"
done; done; done
