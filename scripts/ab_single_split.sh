#!/bin/bash
# Runs ON THE GPU BOX (r06): one config-sized image per call with and without the in-library two-band split
# (ICAMD_SPLIT_SINGLE), interleaved rounds of scripts/exp_single_image.py.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for split in 0 1; do
    for codec in dxt1 dxt5; do
      echo "== round $round ICAMD_SPLIT_SINGLE=$split $codec"
      ICAMD_SPLIT_SINGLE=$split ICAMD_SPLIT_SINGLE_MIN_BLOCKS=${MINB:-262144} CODEC=$codec CALLS=512 python scripts/exp_single_image.py 2>&1 | grep -v "^fit"
    done
  done
done
