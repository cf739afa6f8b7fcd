#!/bin/bash
# PVRTC: per-image time as a function of the images per launch (wave rounds: 1 024 encode waves per 4096^2 image, 3 072 wave slots)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for b in 3 6 9 12 15 16 18 21 24; do
  python bench.py --steps 60 --warmup 5 --workload pvrtc2_rgba8 --batch $b --no-cpu-baseline --no-host-api --no-sustained --no-single-image --no-verify 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $b ms_per_step %.4f  per image %.2f us  kernel_ms %.4f' % (d['ms_per_step'], d['ms_per_step']*1e3/$b, d['roofline']['kernel_ms']))
"
done
