#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel traces for every bench workload, plus separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950) for the HBM traffic of each kernel, plus SQ-only passes for
# the other contents / ETC1 strategies (executed VALU instructions per workload/content/strategy).
# The trace pass runs the bench the way the driver does plus 0.5 s of untimed preconditioning launches, 100 timed
# steps; scripts/summarize_profiles.py takes its statistics from the LAST 100 launches of each kernel (the timed
# ones), so warm-up and preconditioning launches are excluded.  PMC passes: 20 + 3 launches, no preconditioning.
# Output: gpurun_out/prof/<tag>/..., summarised by scripts/summarize_profiles.py into profiles/.
# Usage: scripts/gpu_profile.sh [workload ...]
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd "$R"
WLS=${@:-dxt1_rgba8 dxt1_rgb888 dxt5_rgba8 etc1_rgb888 pvrtc2_rgba8}
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
COMMON="--no-cpu-baseline --no-verify --no-host-api --no-sustained --no-single-image --no-extra-configs --no-slab --no-live-traffic"
for wl in $WLS; do
  T="python bench.py --steps 100 --warmup 5 --precondition-seconds 0.5 --workload $wl $COMMON"
  B="python bench.py --steps 20 --warmup 3 --precondition-seconds 0 --workload $wl $COMMON"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$wl/trace" -o "$wl" -- $T > "$O/$wl.trace.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/$wl/pmc_fetch" -o "$wl" -- $B > "$O/$wl.fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/$wl/pmc_write" -o "$wl" -- $B > "$O/$wl.write.log" 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$O/$wl/pmc_sq" -o "$wl" -- $B > "$O/$wl.sq.log" 2>&1
  grep -h '^{' "$O/$wl.trace.log" | tail -1 > "$O/$wl.bench.json"
  for c in smooth flat; do
    tag=${wl}__${c}__s2
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$O/$tag/pmc_sq" -o "$tag" -- $B --content $c > "$O/$tag.sq.log" 2>&1
  done
done
case " $WLS " in *" etc1_rgb888 "*)
  for s in 0 1 3; do
    tag=etc1_rgb888__noise__s$s
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$O/$tag/pmc_sq" -o "$tag" -- \
      python bench.py --steps 20 --warmup 3 --precondition-seconds 0 --workload etc1_rgb888 --etc-strategy $s $COMMON > "$O/$tag.sq.log" 2>&1
  done;;
esac
# HBM traffic of the BASELINE presets whose launch shape differs from the workload defaults (c3: 4 x 8192^2 DXT5, c4: 1024 x
# 1024^2 ETC1 kSmallerError): FETCH_SIZE / WRITE_SIZE in separate passes -> profiles/traffic.json "presets"
if [ -z "${SKIP_PRESETS:-}" ]; then
  for cfg in c3 c4 c5_8192; do
    B="python bench.py --steps 10 --warmup 2 --precondition-seconds 0 --config $cfg $COMMON"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/preset_$cfg/pmc_fetch" -o "preset_$cfg" -- $B > "$O/preset_$cfg.fetch.log" 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/preset_$cfg/pmc_write" -o "preset_$cfg" -- $B > "$O/preset_$cfg.write.log" 2>&1
  done
fi
# executed VALU instructions of config c4's own launch shape (1 024 x 1024^2 in one launch) per content: the ETC1 search depends
# on the content AND on the texture size (the smooth ramp is four times steeper at 1024^2) -> valu_insts.json "preset:c4/..."
if [ -z "${SKIP_PRESETS:-}" ]; then
  for c in noise smooth flat; do
    tag=presetsq_c4__${c}
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$O/$tag/pmc_sq" -o "$tag" -- \
      python bench.py --steps 5 --warmup 2 --precondition-seconds 0 --config c4 --content $c $COMMON > "$O/$tag.sq.log" 2>&1
  done
fi
# keep the merge small: the raw kernel traces of the preconditioned runs are reduced to per-launch duration lists
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof/*/trace/*kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("icamd_")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
    with open(f.replace("kernel_trace.csv", "timeline.csv"), "w") as out:
        out.write("kernel,start_us,duration_us\n")
        for r in rows:
            out.write("%s,%.1f,%.2f\n" % (r["Kernel_Name"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    os.remove(f)
PY
find "$O" -name '*.csv' | wc -l
