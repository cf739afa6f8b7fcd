#!/bin/bash
# per-kernel durations of the single-image launches (grid sizes tell them from the 16-image steps)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WL=${WL:-pvrtc2_rgba8}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/single_$WL -o x -- env ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=$PWD/${LIB:-image-compression_amd/libic_amd.so} python bench.py --steps 20 --warmup 5 --workload $WL --no-cpu-baseline --no-host-api --no-sustained --no-verify > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/single_$WL/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("icamd_")]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    prev_end=None
    for r in rows:
        k=(r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("Workgroup_Size_X",""))
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        gap=(int(r["Start_Timestamp"])-prev_end)/1e3 if prev_end else 0
        acc[k].append((d,gap)); prev_end=int(r["End_Timestamp"])
for k,v in acc.items():
    ds=sorted(x[0] for x in v); gs=sorted(x[1] for x in v)
    print(k, len(v), "median %.1f us" % ds[len(ds)//2], "gap-before median %.1f us" % gs[len(gs)//2])
PY
rm -rf gpurun_out/single_$WL
