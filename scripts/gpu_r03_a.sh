#!/bin/bash
# Runs ON THE GPU BOX: first round-3 trip -- gpu tests, the new bench legs, clock sources, a 1 200-launch timeline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03a; rm -rf $O; mkdir -p $O
( ls -la /sys/class/drm/ 2>&1; ls /sys/class/drm/card*/device/ 2>&1 | head -80; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1;
  ls /sys/class/drm/card*/device/hwmon/*/ 2>&1; cat /sys/class/drm/card*/device/hwmon/*/freq1_input 2>&1;
  time rocm-smi --showclocks 2>&1; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head ) > $O/clock_sources.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
for wl in dxt1_rgb888 dxt5_rgba8 etc1_rgb888 pvrtc2_rgba8; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-host-api > $O/bench_$wl.json 2> $O/bench_$wl.err
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="python $R/bench.py --steps 1200 --warmup 5 --workload dxt1_rgba8 --no-cpu-baseline --no-verify --no-host-api --no-sustained --no-single-image"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace1200 -o dxt1_rgba8 -- $B > $R/$O/trace1200.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_grbm -o dxt1_rgba8 -- $B > $R/$O/pmc_grbm.log 2>&1
cd $R
python - <<'PY'
import csv, glob, json
O = "gpurun_out/r03a"
for f in glob.glob(O + "/trace1200/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("icamd_")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    with open(O + "/timeline_dxt1_rgba8.csv", "w") as out:
        out.write("launch,start_us,duration_us\n")
        for i, r in enumerate(rows):
            out.write("%d,%.1f,%.2f\n" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    print("timeline:", len(d), "launches; first20 med", sorted(d[:20])[10], "last 20% med", sorted(d[-len(d)//5:])[len(d)//10])
for f in glob.glob(O + "/pmc_grbm/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("icamd_")]
    print(f, len(rows), rows[0].keys() if rows else None)
    for r in rows[-3:]:
        print({k: r[k] for k in r if k in ("Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp", "Dispatch_Id")})
PY
ls $O
