import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--no-cpu-baseline", "--no-host-api", "--no-sustained", "--no-single-image", "--no-extra-configs", "--no-slab", "--steps", "10", "--warmup", "3", "--precondition-seconds", "0.3"]
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ICAMD_ALLOW_LIB_OVERRIDE="1", ICAMD_LIB_PATH=os.path.join(ROOT, lib))
        row = []
        for args in (["--config", "c4", "--content", "noise"], ["--config", "c4", "--content", "smooth"], ["--config", "c4", "--content", "flat"],
                     ["--workload", "etc1_rgb888", "--content", "noise"], ["--workload", "etc1_rgb888", "--content", "smooth"], ["--workload", "etc1_rgb888", "--content", "flat"],
                     ["--workload", "etc1_rgb888", "--etc-strategy", "0"]):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + COMMON, env=env, capture_output=True, text=True)
            try:
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                row.append("%s %.0f %s" % ("/".join(a for a in args if not a.startswith("--")), d["value"] / 1e3, d["parity"][:3]))
            except Exception as e:
                row.append("ERR " + r.stderr[-200:])
        print("%-34s r%d " % (lib, rnd) + " | ".join(row), flush=True)
