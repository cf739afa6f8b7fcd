cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "pvrtc or PVRTC or golden or cxx" 2>&1 | tail -3
for round in 1 2; do for lib in base lerp; do
ICAMD_LIB_PATH=$PWD/build_ab/libic_amd_$lib.so python bench.py --steps 40 --warmup 5 --workload pvrtc2_rgba8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pvrtc $lib', d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms'], d.get('parity'))"
done; done
