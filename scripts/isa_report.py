#!/usr/bin/env python3
"""Per-kernel resource and static instruction report from the gfx950 code objects: profiles/<round>_isa.json.

Compiles every HIP translation unit of the library to assembly (hipcc -S --cuda-device-only, same flags as the
Makefile; no GPU needed) and reads, per __global__ kernel, the code-object metadata (VGPRs, SGPRs, LDS, scratch) and
the static instruction mix of its body.  Occupancy = waves per SIMD allowed by the VGPR allocation (512 / VGPRs,
granule 8, at most 8) and by LDS (160 KiB per CU, 4-wave workgroups).  DESIGN.md's register / occupancy statements
are checkable against this file.
Usage: python scripts/isa_report.py r02"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image-compression_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
             "v_ashrrev_i32", "v_mov_b32", "v_not_b32"}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    report = {"command": "hipcc %s -S --cuda-device-only <tu>.hip" % " ".join(f for f in FLAGS if not f.startswith("-I")),
              "kernels": {}}
    for tu in sorted(f for f in os.listdir(CSRC) if f.endswith("_kernels.hip")):
        with tempfile.TemporaryDirectory() as tmp:
            asm = os.path.join(tmp, "k.s")
            subprocess.check_call(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, tu)],
                                  stderr=subprocess.DEVNULL)
            text = open(asm).read()
        meta = {}
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
            blk = m.group(0)
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            meta[name] = {"vgprs": g("vgpr_count"), "sgprs": g("sgpr_count"), "agprs": g("agpr_count"),
                          "lds_bytes": g("group_segment_fixed_size"), "scratch_bytes": g("private_segment_fixed_size"),
                          "workgroup_lanes": g("max_flat_workgroup_size")}
        lines = text.splitlines()
        for name, e in meta.items():
            start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
            end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # not the first s_endpgm
            mix = collections.Counter()
            for l in lines[start:end + 1]:
                mm = re.match(r"\s+([vs]_\w+|global_\w+|ds_\w+|buffer_\w+|flat_\w+|scratch_\w+)", l)
                if mm:
                    op = mm.group(1)
                    mix["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                        "lds" if op.startswith("ds_") else "vmem"] += 1
                    # the integer ops that issue at twice the base rate (scripts/ubench_valu.hip, r01: 73 vs 38.6 T
                    # lane-ops/s): add/sub, and/or/xor, right shifts, mov -- in their plain VOP1/VOP2 encodings
                    if re.sub(r"_e32$", "", op) in FULL_RATE:
                        mix["valu_full_rate"] += 1
            granule = (e["vgprs"] + e["agprs"] + 7) // 8 * 8
            by_vgpr = min(8, 512 // max(granule, 8))
            waves_per_wg = max(1, e["workgroup_lanes"] // 64)
            by_lds = 8 if e["lds_bytes"] == 0 else min(8, (160 * 1024 // e["lds_bytes"]) * waves_per_wg // 4)
            valu = max(1, mix.get("valu", 0))
            e["valu_full_rate_fraction_static"] = round(mix.get("valu_full_rate", 0) / valu, 4)
            # issue clocks per executed VALU wave-instruction if the executed mix equals the static one: 4 at the base
            # rate (16 lanes / clk / SIMD), 2 for the full-rate ops
            e["valu_issue_clk_per_inst_static"] = round(4.0 - 2.0 * mix.get("valu_full_rate", 0) / valu, 4)
            e.update({"static_instructions": dict(mix), "waves_per_simd_by_vgprs": by_vgpr,
                      "waves_per_simd_by_lds": by_lds, "waves_per_simd": min(by_vgpr, by_lds), "translation_unit": tu})
            report["kernels"][name] = e
    out = os.path.join(ROOT, "profiles", "%s_isa.json" % rnd)
    with open(out, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    for n, e in sorted(report["kernels"].items()):
        print("%-40s vgpr %3d sgpr %3d lds %6d scratch %d  waves/SIMD %d  valu %d" % (
            n, e["vgprs"], e["sgprs"], e["lds_bytes"], e["scratch_bytes"], e["waves_per_simd"],
            e["static_instructions"].get("valu", 0)))
    print("wrote", os.path.relpath(out))


if __name__ == "__main__":
    main()
