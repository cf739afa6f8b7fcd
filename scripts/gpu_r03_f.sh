#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r03f.txt 2>&1; tail -3 gpurun_out/pytest_gpu_r03f.txt
bash scripts/gpu_profile.sh > gpurun_out/gpu_profile.log 2>&1; tail -2 gpurun_out/gpu_profile.log
bash scripts/bench_all.sh > gpurun_out/bench_all.txt 2>&1; tail -40 gpurun_out/bench_all.txt
du -sh gpurun_out/prof
