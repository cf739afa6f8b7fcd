"""A/B of the ETC1 kSmallerError encode on SMALL launches: one lane per block (ICAMD_ETC1_QUAD_MAX_BLOCKS=0) vs four lanes per block
(always: a huge threshold) vs the shipped threshold.  Each setting in its own subprocess, three interleaved rounds; every timed
result is compared with the oracle (texture 0)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, torch
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(11)
res = []
for (size, n, comps) in [(64, 1, 3), (256, 1, 3), (512, 1, 3), (512, 2, 3), (768, 1, 3), (1024, 1, 3), (256, 16, 3), (2048, 1, 3), (256, 1, 4)]:
    src = torch.randint(0, 256, (n, size, size, comps), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((n, size * size // 2), dtype=torch.uint8, device=dev)
    f = lambda: pkg.encode_device(T.ETC1, src, size, size, comps, n_images=n, out=out)
    for _ in range(200): f()
    torch.cuda.synchronize()
    ok = out[0].cpu().numpy().tobytes() == T.oracle_encode(T.ETC1, src[0].cpu().numpy(), size, size, comps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): f()
    e1.record(); torch.cuda.synchronize()
    lat = []
    for _ in range(200):   # one call at a time between an event pair, synchronised: the call's own duration on the device
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        lat.append(a.elapsed_time(b) * 1e3)
    lat.sort()
    res.append("%%dx%%d^2c%%d %%.1f us/call back to back, %%.1f us alone %%s" %% (n, size, comps, e0.elapsed_time(e1) / 400 * 1e3, lat[len(lat) // 2],
                                                                              "ok" if ok else "MISMATCH"))
print(" | ".join(res))
'''
for rnd in range(3):
    for name, v in (("one-lane", "0"), ("quad", str(1 << 40)), ("shipped", None)):
        env = dict(os.environ)
        env.pop("ICAMD_ETC1_QUAD_MAX_BLOCKS", None)
        if v is not None:
            env["ICAMD_ETC1_QUAD_MAX_BLOCKS"] = v
        r = subprocess.run([sys.executable, "-c", child % ROOT], env=env, capture_output=True, text=True)
        print("%-8s r%d %s" % (name, rnd, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-600:]), flush=True)
