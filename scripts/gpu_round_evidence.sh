#!/bin/bash
# Runs ON THE GPU BOX: the round's evidence from ONE box and one binary -- encoder profiles, next-row profiles, bench_all, the GPU test
# tier, the default bench line, the parity soak -> gpurun_out/evidence/ (+ gpurun_out/prof, prof_next, bench_all.jsonl).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/evidence
python -c "import ic_amd_loader as l; print(l.load_package().lib().icamd_version().decode())" > gpurun_out/evidence/version.txt
( time scripts/gpu_profile.sh ) > gpurun_out/evidence/profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/evidence/profile.log
( time scripts/gpu_profile_next_rows.sh ) > gpurun_out/evidence/profile_next.log 2>&1; echo "profile_next rc=$?"
( time scripts/bench_all.sh ) > gpurun_out/evidence/bench_all.txt 2>&1; echo "bench_all rc=$?"; tail -40 gpurun_out/evidence/bench_all.txt | cut -c1-260
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/evidence/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/evidence/pytest_gpu.txt
( time python bench.py ) > gpurun_out/evidence/bench_default.log 2>&1; echo "bench rc=$?"
( time timeout 1500 python scripts/parity_soak.py 300 ) > gpurun_out/evidence/parity_soak.txt 2>&1; echo "soak rc=$?"; tail -8 gpurun_out/evidence/parity_soak.txt
