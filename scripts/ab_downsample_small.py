"""A/B of the ETC1 kSmallerError Downsample on small grids (a mip chain's low levels): one lane vs four lanes per output block
(ICAMD_PAD_BORDER_QUAD=0 / 1, the switch shared with the Pad border).  A whole chain 1024^2 -> 4^2 per 'chain' figure."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, ctypes, torch
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package(); L = pkg.lib()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(3)
sh = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 1024
src = torch.randint(0, 256, (n, n, 3), dtype=torch.uint8, device=dev, generator=g)
top = pkg.encode_device(T.ETC1, src, n, n, 3).reshape(-1)
levels = [top]
s = n
while s > 4:
    levels.append(torch.empty(((s // 2 + 3) // 4) ** 2 * 8, dtype=torch.uint8, device=dev)); s //= 2
def chain():
    s = n
    for i in range(len(levels) - 1):
        assert L.icamd_downsample_device(T.ETC, 2, T.RGB, s, s, ctypes.c_void_p(levels[i].data_ptr()), ctypes.c_void_p(levels[i + 1].data_ptr()), levels[i + 1].numel(), sh) == 0
        s //= 2
for _ in range(50): chain()
torch.cuda.synchronize()
ok = True
s = n; cur = top.cpu().numpy().tobytes()
for i in range(len(levels) - 1):
    want = T.oracle_downsample(T.ETC, T.RGB, cur, s, s, 2)
    ok = ok and levels[i + 1].cpu().numpy().tobytes() == want
    cur = want; s //= 2
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): chain()
e1.record(); torch.cuda.synchronize()
res = ["chain 1024->4 (8 levels): %%.1f us %%s" %% (e0.elapsed_time(e1) / 200 * 1e3, "ok" if ok else "MISMATCH")]
for lvl in (1, 2, 3):   # the 512^2, 256^2, 128^2 source levels on their own, each call alone
    s = n >> lvl
    lat = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.icamd_downsample_device(T.ETC, 2, T.RGB, s, s, ctypes.c_void_p(levels[lvl].data_ptr()), ctypes.c_void_p(levels[lvl + 1].data_ptr()), levels[lvl + 1].numel(), sh)
        b.record(); torch.cuda.synchronize(); lat.append(a.elapsed_time(b) * 1e3)
    lat.sort(); res.append("%%d^2 -> %%d^2 alone %%.1f us" %% (s, s // 2, lat[100]))
print(" | ".join(res))
'''
for rnd in range(3):
    for quad in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", child % ROOT], env=dict(os.environ, ICAMD_PAD_BORDER_QUAD=quad), capture_output=True, text=True)
        print("quad=%s r%d %s" % (quad, rnd, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-600:]), flush=True)
