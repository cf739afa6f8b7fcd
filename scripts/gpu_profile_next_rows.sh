#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) for the
# SURVEY 8f "next"-row kernels (decoders, Downsample, DXT1->ETC1 transcode, PVRTC decode) driven by
# scripts/bench_next_rows.py.  Output: gpurun_out/prof_next/..., summarised by scripts/summarize_next_rows.py.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_next
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export ICAMD_NEXT_ROWS_QUICK=1   # 3 + 20 launches per leg instead of the 0.25 s of preconditioning of the unprofiled run (bench_all.sh)
CMD="python scripts/bench_next_rows.py"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o next -- $CMD > "$O/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o next -- $CMD > "$O/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o next -- $CMD > "$O/write.log" 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$O/pmc_sq" -o next -- $CMD > "$O/sq.log" 2>&1
find "$O" -name '*kernel_trace.csv' -delete
find "$O" -name '*.csv' | wc -l
grep -E 'Mpix/s' "$O/trace.log"
