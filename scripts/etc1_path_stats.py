#!/usr/bin/env python3
"""Diagnostics (GPU box): which evaluation the ETC1 codeword searches take, per content.  Needs a library built with
-DICAMD_ETC1_STATS (see the ICAMD_ETC1_COUNT macro in csrc/etc1_block.h); counts are per WAVE.
usage: ICAMD_ALLOW_LIB_OVERRIDE=1 ICAMD_LIB_PATH=<stats lib> python scripts/etc1_path_stats.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_amd_loader
import bench
pkg = ic_amd_loader.load_package()
L = pkg.lib()
L.icamd_debug_etc1_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
buf = (ctypes.c_uint * 16)()
for strategy in (2, 0):
    for content in ("noise", "smooth", "flat"):
        src = bench.make_batch(torch, content, 4, 4096, 3, dev, seed=0)
        L.icamd_debug_etc1_stats(buf, 1)
        pkg.encode_device(pkg.ETC1, src, 4096, 4096, 3, etc_strategy=strategy, n_images=4)
        torch.cuda.synchronize()
        L.icamd_debug_etc1_stats(buf, 1)
        c = list(buf)
        n = max(c[4], 1)
        print("strategy %d %-6s searches(waves) %8d | per search: shortcut %.2f tier %.2f exact %.2f pruned %.2f | fast@cw0 %.2f prunable %.2f tier-instantiation %.2f | exact by cw %s"
              % (strategy, content, c[4], c[0] / n, c[1] / n, c[2] / n, c[3] / n, c[5] / n, c[6] / n, c[7] / n,
                 " ".join("%.2f" % (x / n) for x in c[8:16])))
