#!/bin/bash
# A/B on the GPU box: bench.py against several builds of the library, interleaved rounds.
# usage: scripts/ab_bench.sh <workload> <lib1> <lib2> ...   (paths relative to the repo root)
cd ${GRAFT_REPO_ROOT:-/root/repo}
WL=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 5 --workload $WL --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'round$round', d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms'])"
  done
done
