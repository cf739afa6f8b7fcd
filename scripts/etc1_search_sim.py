#!/usr/bin/env python3
"""CPU simulation (numpy) of what the ETC1 kSmallerError codeword searches face on the bench's synthetic contents -- the
evidence behind r04's decisions for VERDICT r03 item 2 (profiles/r04_etc1_search_sim.txt):

  * which codeword wins;
  * how many codewords per search cannot take the unclamped shortcut (base +/- b leaves 0..255) -- per lane, per 16x4-block
    wave as launched today, and per wave after the blocks of a 16x16-block workgroup are sorted by their room;
  * how the non-shortcut codewords split into clamp classes (which of +a, +b, -a, -b still fit) per wave;
  * what partial-distortion elimination (abandon a codeword once its partial error exceeds the best total) would add on top
    of the existing lower-bound pruning.

The encoder itself is not involved: candidates, errors and bases are restated here in a few numpy lines
(etc_compressor.cc:101-125, 299-312, 350-409)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

A = [2, 5, 9, 13, 18, 24, 33, 47]
B = [8, 17, 29, 42, 60, 80, 106, 183]


def subblocks(content, size):
    img = bench.make_batch(torch, content, 1, size, 3, "cpu", seed=0)[0].numpy().astype(np.int64)
    by = bx = size // 4
    blk = img.reshape(by, 4, bx, 4, 3).transpose(0, 2, 1, 3, 4)
    for S in (0, 1):  # the left | right partition; the other one behaves the same
        P = blk[:, :, :, 2 * S:2 * S + 2, :].reshape(by, bx, 8, 3)
        q5 = P.sum(2) >> 6
        yield P, (q5 << 3) | (q5 >> 2)


def run(content, size):
    lane = wave = regrouped = 0.0
    n = 0
    winners = np.zeros(8)
    classes = {}
    tot = dict(nonshortcut=0, lb_pruned=0, pde_pruned=0, full=0)
    nw = 0
    for P, Bs in subblocks(content, size):
        by, bx = Bs.shape[:2]
        room = np.minimum(Bs.min(-1), 255 - Bs.max(-1))
        lane += sum((room < B[cw]).mean() for cw in range(8))
        r = room.reshape(by // 4, 4, bx // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 64)
        wave += sum((r.min(1) < B[cw]).mean() for cw in range(8))
        t = np.sort(room.reshape(by // 16, 16, bx // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256), axis=1).reshape(-1, 4, 64)
        regrouped += sum((t.min(2) < B[cw]).mean() for cw in range(8))
        n += 1
        perr = np.zeros((by, bx, 8, 8))
        for cw in range(8):
            m = np.array([A[cw], B[cw], -A[cw], -B[cw]])
            v = np.clip(Bs[:, :, None, :] + m[None, None, :, None], 0, 255)
            perr[:, :, :, cw] = ((P[:, :, :, None, :] - v[:, :, None, :, :]) ** 2).sum(-1).min(-1)
        errs = perr.sum(2)
        win = errs.argmin(-1)
        for cw in range(8):
            winners[cw] += (win == cw).sum()
        d = P - Bs[:, :, None, :]
        dev = np.abs(d).max(2)
        d1 = np.abs(d).sum(-1).max(-1)
        up_all, dn_all = 255 - Bs.max(-1), Bs.min(-1)
        for wy in range(0, by, 4):
            for wx in range(0, bx, 16):
                sl = (slice(wy, wy + 4), slice(wx, wx + 16))
                E, PE, R = errs[sl].reshape(64, 8), perr[sl].reshape(64, 8, 8), room[sl].reshape(64)
                BB, DV = Bs[sl].reshape(64, 3), dev[sl].reshape(64, 3)
                U, D = up_all[sl].min(), dn_all[sl].min()
                nw += 1
                prunable = np.all(d1[sl] < 3 * 47)
                best = np.full(64, np.inf)
                for cw in range(8):
                    pa, pb, na, nb = A[cw] <= U, B[cw] <= U, A[cw] <= D, B[cw] <= D
                    k = ("all four fit (shortcut)" if pa and pb and na and nb else
                         "both a fit, one b clamps" if pa and na and (pb or nb) else
                         "both a fit, both b clamp (mixed tier)" if pa and na else
                         "one side fits, the other clamps" if (pa and pb) or (na and nb) else
                         "one a fits only" if pa or na else "everything clamps")
                    classes[k] = classes.get(k, 0) + 1
                    if not np.all(R >= B[cw]):
                        tot["nonshortcut"] += 1
                        lbp = False
                        if cw > 0 and prunable:  # the shipped lower bound (search_codewords)
                            up = np.maximum(np.minimum(A[cw], 255 - BB) - DV, 0)
                            dn = np.maximum(np.minimum(A[cw], BB) - DV, 0)
                            lbp = np.all(8 * np.minimum((up ** 2).sum(1), (dn ** 2).sum(1)) > best)
                        if lbp:
                            tot["lb_pruned"] += 1
                        elif cw > 0 and np.all(PE[:, :4, cw].sum(1) > best):
                            tot["pde_pruned"] += 1
                        else:
                            tot["full"] += 1
                    best = np.minimum(best, E[:, cw])
    print("%-6s %5d^2 | winners by codeword %s" % (content, size, np.round(winners / winners.sum(), 3)))
    print("   codewords per search that cannot take the shortcut: per lane %.2f | per 16x4-block wave %.2f | per wave after sorting "
          "the 16x16-block workgroup by room %.2f" % (lane / n, wave / n, regrouped / n))
    print("   clamp classes per search (wave-uniform): %s" % {k: round(v / nw, 2) for k, v in sorted(classes.items())})
    print("   of the %.2f non-shortcut codewords per search: %.2f fall to the shipped lower bound, partial-distortion elimination "
          "after 4 of 8 pixels would stop %.2f more, %.2f are evaluated in full" % (
              tot["nonshortcut"] / nw, tot["lb_pruned"] / nw, tot["pde_pruned"] / nw, tot["full"] / nw))


if __name__ == "__main__":
    for c in ("smooth", "noise", "flat"):
        for size in (1024,):
            run(c, size)
