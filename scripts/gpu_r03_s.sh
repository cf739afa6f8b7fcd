#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in c2 c3 c4 c5; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-host-api --no-sustained --no-verify 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); si=d.get('single_image') or {}
        print('$cfg', 'step', d['ms_per_step'], 'single', si.get('median_ms_per_call'), 'wall', si.get('back_to_back_ms_per_call_wall'), 'graph', json.dumps(si.get('graph_replay')))
    elif 'Error' in l or 'Traceback' in l: print(l.strip())
"
done
