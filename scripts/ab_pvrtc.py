"""A/B of library builds on PVRTC: usage r05_ab.py lib1 lib2 ... ; each timed in its own subprocess, 3 interleaved rounds"""
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
g = torch.Generator(device="cuda"); g.manual_seed(5)
res = []
for (size, n, mode, sb) in %s:
    src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((n, size * size // 4), dtype=torch.uint8, device=dev)
    pkg.pvrtc_tune(1, -1)
    ref = pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n).clone()
    pkg.pvrtc_tune(mode, sb)
    for _ in range(80): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, ref))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n, out=out)
    e1.record(); torch.cuda.synchronize()
    res.append("%%dx%%d m%%d k%%d %%.4f %%s" %% (n, size, mode, sb, e0.elapsed_time(e1) / 200, "ok" if ok else "MISMATCH"))
print(" | ".join(res))
'''
cases = os.environ.get("CASES", "[(4096,16,2,6),(4096,16,2,5),(2048,64,2,5),(1024,256,2,6),(1024,256,2,4)]")
for rnd in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ICAMD_ALLOW_LIB_OVERRIDE="1", ICAMD_LIB_PATH=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, "-c", child % (ROOT, cases)], env=env, capture_output=True, text=True)
        print("%-34s r%d %s" % (lib, rnd, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-300:]), flush=True)
