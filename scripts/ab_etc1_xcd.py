"""Runs ON THE GPU BOX (r06): the XCD-aware tile-column mappings of the RGB888 ETC1 kernels (ICAMD_ETC1_XCD_COLUMNS = 0 plain, 1 halves,
2 pairs, 4 ABBA quads) -- Mpixels/s, kernel time and (first round) the HBM traffic counted in the run, per content.
usage: python scripts/ab_etc1_xcd.py <lib> <lib> ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--no-cpu-baseline", "--no-host-api", "--no-sustained", "--no-single-image", "--no-extra-configs", "--no-slab", "--no-next-rows",
          "--steps", "20", "--warmup", "3", "--precondition-seconds", "0.3"]
CASES = (["--config", "c4"], ["--config", "c4", "--content", "smooth"], ["--config", "c4", "--content", "flat"],
         ["--workload", "etc1_rgb888", "--content", "smooth"], ["--workload", "etc1_rgb888", "--etc-strategy", "3"])
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ICAMD_ALLOW_LIB_OVERRIDE="1", ICAMD_LIB_PATH=os.path.join(ROOT, lib))
        row = []
        for args in CASES:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + COMMON + (["--no-live-traffic"] if rnd else []),
                               env=env, capture_output=True, text=True)
            try:
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                rf = d["roofline"]
                row.append("%s: %.1f Gpix/s %.4f ms%s %s" % (" ".join(a.lstrip("-") for a in args[1:]), d["value"] / 1e3, rf["kernel_ms"],
                           "" if rnd else " traffic x%.3f" % ((rf["traffic"] or 0) / rf["algorithmic_bytes_per_launch"]), d["parity"][:3]))
            except Exception as e:
                row.append("ERR %s %s" % (e, r.stderr[-200:]))
        print("%-30s r%d " % (lib, rnd) + " | ".join(row), flush=True)
