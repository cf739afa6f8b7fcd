"""A/B of the ETC1 Pad border kernel: one lane per pad block (ICAMD_PAD_BORDER_QUAD=0, r04) vs four lanes per pad block (r05 default).
Each setting in its own subprocess, three interleaved rounds; every timed result checked against the oracle (images 0 and last)."""
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
g = torch.Generator(device="cuda"); g.manual_seed(7)
sh = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = []
for (n, batch, ph, pw, content) in [(4096, 16, 4104, 4104, "noise"), (4096, 16, 4104, 4104, "smooth"), (1024, 64, 1028, 1032, "noise"), (256, 64, 260, 260, "noise")]:
    if content == "noise":
        src = torch.randint(0, 256, (batch, n, n, 3), dtype=torch.uint8, device=dev, generator=g)
    else:
        x = torch.arange(n, device=dev, dtype=torch.int32).view(1, 1, n); y = torch.arange(n, device=dev, dtype=torch.int32).view(1, n, 1)
        nz = torch.randint(0, 32, (batch, n, n), dtype=torch.int32, device=dev, generator=g)
        src = torch.stack([(255 * x // n + nz) & 255, (255 * y // n + nz) & 255, (255 * (x + y) // (2 * n) + nz) & 255], dim=-1).to(torch.uint8).contiguous()
    blocks = pkg.encode_device(T.ETC1, src, n, n, 3, n_images=batch)
    torch.cuda.synchronize(); del src
    per_in = blocks.shape[1]
    pout = torch.zeros((batch, (ph // 4) * (pw // 4) * 8), dtype=torch.uint8, device=dev)
    def one():
        for i in range(batch):
            assert L.icamd_pad_device(T.ETC, 2, T.RGB, n, n, ctypes.c_void_p(blocks[i].data_ptr()), ph, pw, ctypes.c_void_p(pout[i].data_ptr()), pout.shape[1], sh) == 0
    def bat():
        assert L.icamd_pad_batch_device(T.ETC, 2, T.RGB, n, n, batch, ctypes.c_void_p(blocks.data_ptr()), per_in, ph, pw, ctypes.c_void_p(pout.data_ptr()), pout.shape[1], pout.shape[1], sh) == 0
    t1 = timeit(one, 10)
    ok = all(pout[i].cpu().numpy().tobytes() == T.oracle_pad(T.ETC, T.RGB, blocks[i].cpu().numpy().tobytes(), n, n, ph, pw, 2) for i in (0, batch - 1))
    pout.zero_()
    tb = timeit(bat, 20)
    ok = ok and all(pout[i].cpu().numpy().tobytes() == T.oracle_pad(T.ETC, T.RGB, blocks[i].cpu().numpy().tobytes(), n, n, ph, pw, 2) for i in (0, batch - 1))
    res.append("%%dx%%d^2->%%dx%%d %%s: %%.1f us/call, batched %%.1f us %%s" %% (batch, n, ph, pw, content, t1 * 1e3 / batch, tb * 1e3, "ok" if ok else "MISMATCH"))
print(" | ".join(res))
'''
for rnd in range(3):
    for quad in ("0", "1"):
        env = dict(os.environ, ICAMD_PAD_BORDER_QUAD=quad)
        r = subprocess.run([sys.executable, "-c", child % ROOT], env=env, capture_output=True, text=True)
        print("quad=%s r%d %s" % (quad, rnd, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-600:]), flush=True)
