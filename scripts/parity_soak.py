#!/usr/bin/env python3
"""Long seeded random parity soak on the GPU box: the GPU tier's test_random_soak_matches_oracle with more seeds.
usage: python scripts/parity_soak.py [n_seeds]   (each seed = 400 block-codec cases + 60 PVRTC cases)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ic_amd_loader, ic_testlib as T
pkg = ic_amd_loader.load_package()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t0 = time.time(); cases = 0; bad = 0
for seed in range(n_seeds):
    for (codec, comps, swap, strategy, h, w, pad, img) in T.soak_cases(0xA000 + seed, 400, 60, max_h=260, max_w=400, max_log2_pvrtc=6):
        src = T.with_row_padding(img, pad)
        stride = w * comps + pad
        want = T.oracle_encode(codec, src, h, w, comps, swap, strategy, stride=stride)
        out = pkg.encode_device(codec, torch.from_numpy(np.ascontiguousarray(src)).cuda(), h, w, comps, swap_rb=bool(swap),
                                etc_strategy=strategy, row_stride_bytes=stride)
        torch.cuda.synchronize()
        cases += 1
        if out.cpu().numpy().tobytes() != want:
            bad += 1
            print("MISMATCH", seed, codec, comps, swap, strategy, h, w, pad)
print("parity soak: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
