#!/usr/bin/env python3
"""After scripts/gpu_round_evidence.sh (or its scratch twin) has been merged back into gpurun_out/: summarise the profiles and copy the
round's evidence into profiles/<round>_*.  usage: collect_evidence.py r05 [evidence_dir = gpurun_out/evidence]"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
ev = os.path.join(ROOT, sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/evidence")
P = os.path.join(ROOT, "profiles")
for script in ("isa_report.py", "summarize_profiles.py", "summarize_next_rows.py"):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", script), rnd], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(ROOT, "gpurun_out", "bench_all.jsonl"), os.path.join(P, rnd + "_bench_all.jsonl"))
lines = open(os.path.join(ev, "bench_all.txt")).read().splitlines()
open(os.path.join(P, rnd + "_bench_all.txt"), "w").write("\n".join(l for l in lines if re.match(r"^(None|c\d)", l)) + "\n")
for l in open(os.path.join(ROOT, "gpurun_out", "bench_all.jsonl")):
    d = json.loads(l)
    p = d["config"].get("preset")
    if p and isinstance(d.get("sustained"), dict):
        json.dump({"config": d["config"], "value": d["value"], "ms_per_step": d["ms_per_step"], "roofline": d["roofline"],
                   "sustained": d["sustained"], "clock": d.get("clock")}, open(os.path.join(P, "%s_sustained_%s.json" % (rnd, p)), "w"),
                  indent=1, sort_keys=True)
line = [x for x in open(os.path.join(ev, "bench_default.log")) if x.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(P, rnd + "_bench_default_line.json"), "w"), indent=1)
shutil.copy(os.path.join(ROOT, "gpurun_out", "next_rows.txt"), os.path.join(P, rnd + "_next_rows.txt"))
shutil.copy(os.path.join(ev, "parity_soak.txt"), os.path.join(P, rnd + "_parity_soak.txt"))
shutil.copy(os.path.join(ev, "pytest_gpu.txt"), os.path.join(P, rnd + "_pytest_gpu.txt"))
d = json.loads(line)
print(open(os.path.join(ev, "version.txt")).read().strip())
print("default line: %.0f %s, frac %.4f, traffic %s (%s)" % (d["value"], d["unit"], d["roofline"]["frac"], d["roofline"]["traffic"],
                                                            d["roofline"]["traffic_source"][:24]))
for k, v in (d.get("configs") or {}).items():
    print("  %-8s %10.0f  frac %.4f  valu_frac %s" % (k, v["value"], v["roofline"]["frac"], v["roofline"].get("valu_frac")))
print([l for l in open(os.path.join(ev, "pytest_gpu.txt")) if "passed" in l or "failed" in l][-1].strip())
