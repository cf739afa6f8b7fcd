"""A/B of library builds on the block decoders / transcoder: usage ab_decode.py lib1 lib2 ...; each library timed in its own
subprocess, 3 interleaved rounds; outputs compared against the oracle (image 0) and between libraries (checksum)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, zlib, torch, numpy as np
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ic_amd_loader
pkg = ic_amd_loader.load_package()
import ic_testlib as T
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda"); g.manual_seed(5)
res = []
def timed(f, reps=100):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (what, size, n) in %s:
    if what == "pvrtc_enc":      # decode real encoder output
        src = torch.randint(0, 256, (n, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
        blocks = pkg.encode_device(T.PVRTC2, src, size, size, 4, n_images=n)
        codec = T.PVRTC2
    elif what == "pvrtc_rand":   # arbitrary block words: every mode / flag combination
        blocks = torch.randint(0, 256, (n, size * size // 4), dtype=torch.uint8, device=dev, generator=g)
        codec = T.PVRTC2
    elif what in ("dxt1_rand", "dxt5_rand", "etc1_rand"):
        codec = {"dxt1_rand": T.DXT1, "dxt5_rand": T.DXT5, "etc1_rand": T.ETC1}[what]
        blocks = torch.randint(0, 256, (n, pkg.encoded_size(codec, size, size)), dtype=torch.uint8, device=dev, generator=g)
    out = pkg.decode_device(codec, blocks, size, size, n_images=n)
    torch.cuda.synchronize()
    want = T.oracle_decode(codec, blocks[0].cpu().numpy().tobytes(), size, size).tobytes()
    ok = out[0].cpu().numpy().tobytes() == want
    crc = zlib.crc32(out.cpu().numpy().tobytes())
    import ctypes
    L, per_in, per_out = pkg.lib(), blocks.shape[1], out.shape[1]
    ms = timed(lambda: L.icamd_decode_device(codec, 0, size, size, 0, n, per_in, per_out, ctypes.c_void_p(blocks.data_ptr()),
                                             ctypes.c_void_p(out.data_ptr()), None))
    res.append("%%s %%dx%%d %%.4f ms %%s crc %%08x" %% (what, n, size, ms, "ok" if ok else "MISMATCH", crc))
print(" | ".join(res))
'''
cases = os.environ.get("CASES", '[("pvrtc_enc",4096,16),("pvrtc_rand",4096,16),("pvrtc_rand",1024,64),("pvrtc_rand",256,64)]')
for rnd in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ICAMD_ALLOW_LIB_OVERRIDE="1", ICAMD_LIB_PATH=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, "-c", child % (ROOT, cases)], env=env, capture_output=True, text=True)
        print("%-44s r%d %s" % (lib, rnd, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-400:]), flush=True)
