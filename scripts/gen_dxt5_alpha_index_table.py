#!/usr/bin/env python3
"""Generates image-compression_amd/csrc/dxt5_alpha_index_table.inc (BUILD INPUT of the DXT5 alpha encoder).

ComputeAlphaBits (reference dxtc_compressor.cc:427-479) picks, per pixel, the nearest of 8 table values with ties to
the lowest index.  For a block's (alpha0, alpha1) that choice is a step function of the pixel's distance x from
alpha0 that depends only on the mode (8-value: alpha0 > alpha1, 6-value: alpha0 <= alpha1) and on D = |alpha0 -
alpha1|.  The kernel evaluates it as   k = LUT[ max(x * R + B, force) >> 13 ]   in 16-bit lanes (dxt_block.h); this
script finds, for every (mode, D), integers R, B (and the LUT) for which that is EXACT:

  * every x in [0, D] where the reference's choice changes must be a point where (x*R + B) >> 13 changes, all cells
    stay <= 6 (8-value mode) / <= 5 (6-value mode), and D*R + B < 2^16;
  * 8-value mode: the last step (towards alpha1, index 1, which wins its tie against index 7) does not sit on the
    uniform grid; pixels with x >= P are forced into cell 7 by the kernel (one packed subtract + max), LUT[7] = 1;
  * 6-value mode: cells 6 / 7 are reserved for the special pixel values 255 (index 7) and 0 (index 6); P is an x6
    with ((x6*R + B) mod 2^16) >> 13 == 6, where the kernel parks alpha = 255 pixels.

Row (mode8 ? 0 : 256) + D = { R | B << 16, P, LUT bytes 0..3, LUT bytes 4..7 }.  tests/test_kernel_math_host.py checks
the kernel math built on this table against the oracle for every (alpha0, alpha1, alpha)."""
import os

import numpy as np

S = 13
CELL = 1 << S


def scan(t, a):  # dxtc.cc:459-468: strict '<' scan, lowest index wins ties
    best = None
    for k in range(8):
        d = (t[k] - a) ** 2
        if best is None or d < best[0]:
            best = (d, k)
    return best[1]


def g8(D):  # choice as a function of x = alpha0 - alpha, 8-value table (dxtc.cc:446-455)
    a1, a0 = 0, D
    t = [a0, a1] + [(a0 * (7 - p) + a1 * p) // 7 for p in range(1, 7)]
    return [scan(t, a0 - x) for x in range(D + 1)]


def g6(D):  # x = alpha - alpha0, 6-value table (dxtc.cc:436-445)
    a0 = 1 if D <= 253 else 255 - D
    a1 = a0 + D
    t = [a0, a1] + [(a0 * (5 - p) + a1 * p) // 5 for p in range(1, 5)] + [0, 255]
    return [scan(t, a0 + x) for x in range(D + 1)]


def feasible(bps, xmax, maxcell, rs):
    b = np.arange(CELL)
    for r in rs:
        ok = (xmax * r + b) < (maxcell + 1) * CELL
        for th in bps:
            ok &= ((th - 1) * r + b) // CELL < (th * r + b) // CELL
            if not ok.any():
                break
        if ok.any():
            idx = np.nonzero(ok)[0]
            return r, int(idx[len(idx) // 2])
    return None


def order(n, D):  # candidate slopes, nearest to the nominal n * 2^13 / D first
    r0 = n * CELL / max(D, 1)
    c, out = int(round(r0)), []
    for d in range(0, int(r0 * 0.35) + 3):
        for r in (c + d, c - d):
            if 1 <= r <= 65535 and r not in out:
                out.append(r)
    return out


def build():
    rows = []
    for mode8 in (1, 0):
        for D in range(256):
            if mode8 and D == 0:
                rows.append((0, 0, 0, [0] * 8))
                continue
            g = g8(D) if mode8 else g6(D)
            runs = [x for x in range(1, D + 1) if g[x] != g[x - 1]]
            lut = [None] * 8
            if mode8:
                sol = None
                for th in ([runs[-1]] if runs and g[D] == 1 else []) + [D + 1]:
                    r = feasible([x for x in runs if x < th], th - 1, 6, order(7, D))
                    if r:
                        sol = (r[0], r[1], th)
                        break
                assert sol, (mode8, D)
                R, B, P = sol
                for x in range(D + 1):
                    assert x * R + B < 65536
                    i = 7 if x >= P else (x * R + B) >> S
                    assert lut[i] in (None, g[x]), (D, x)
                    lut[i] = g[x]
            else:
                sol = None
                for R2 in order(5, D):
                    rr = feasible(runs, D, 5, [R2])
                    if not rr:
                        continue
                    x6 = next((x for x in range(65536) if ((x * rr[0] + rr[1]) & 0xffff) >> S == 6), None)
                    if x6 is not None:
                        sol = (rr[0], rr[1], x6)
                        break
                assert sol, (mode8, D)
                R, B, P = sol
                for x in range(D + 1):
                    i = (x * R + B) >> S
                    assert i <= 5 and lut[i] in (None, g[x])
                    lut[i] = g[x]
                lut[6], lut[7] = 7, 6
            assert lut[0] == 0  # alpha == alpha0 -> index 0; also what the unused selector bytes of v_perm pick up
            rows.append((R, B, P, [v or 0 for v in lut]))
    return rows


def main():
    rows = build()
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "image-compression_amd", "csrc", "dxt5_alpha_index_table.inc")
    with open(out, "w") as f:
        f.write("/* GENERATED by scripts/gen_dxt5_alpha_index_table.py -- do not edit.\n"
                " * row (alpha0 > alpha1 ? 0 : 256) + |alpha0 - alpha1| = { R | B << 16, P, LUT[0..3], LUT[4..7] } */\n")
        for R, B, P, lut in rows:
            lo = sum(lut[j] << (8 * j) for j in range(4))
            hi = sum(lut[4 + j] << (8 * j) for j in range(4))
            f.write("{ 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu },\n" % (R | B << 16, P, lo, hi))
    print("wrote", os.path.relpath(out), len(rows), "rows")


if __name__ == "__main__":
    main()
