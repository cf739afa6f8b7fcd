#!/bin/bash
# A/B of the single-image leg (one config-sized image per launch) for the libraries in $LIBS, workload $WL
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WL=${WL:-pvrtc2_rgba8}
for round in 1 2 3; do
for lib in $LIBS; do
  ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --workload $WL --no-cpu-baseline --no-host-api --no-sustained --no-verify 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib round$round', d['ms_per_step'], json.dumps(d.get('single_image'))[:400])
"
done; done
