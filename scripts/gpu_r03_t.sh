#!/bin/bash
# icamd_encode_batch_sharded_device on 128 evenly spaced 1024^2 textures (BASELINE c4's per-GPU share): wall time per call
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import importlib, time, torch
pkg = importlib.import_module("image-compression_amd")
n, size = 128, 1024
src = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, device="cuda")
per = pkg.encoded_size(pkg.ETC1, size, size)
out = torch.zeros((n, per), dtype=torch.uint8, device="cuda")
for label, srcs in (("evenly spaced (batched launches)", [src[i] for i in range(n)]),
                    ("every other image elsewhere (one launch per image)", [src[i] if i % 2 else src[i].clone() for i in range(n)])):
    outs = [out[i] for i in range(n)]
    for _ in range(3):
        pkg.encode_batch_sharded_device(pkg.ETC1, srcs, size, size, 3, [0], outs=outs)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        pkg.encode_batch_sharded_device(pkg.ETC1, srcs, size, size, 3, [0], outs=outs)
    dt = (time.perf_counter() - t0) / reps
    print("%-55s %.3f ms per call = %.1f Gpix/s" % (label, dt * 1e3, n * size * size / dt / 1e9))
PY
