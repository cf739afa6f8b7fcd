#!/usr/bin/env python3
"""Experiment (r04): co-residency of PVRTC's memory-bound morph kernel and its VALU-bound encode
kernel, using nothing but the public device entry point: the batch of 16 x 4096^2 is split over TWO streams whose
morph / encode phases are offset against each other (different first-group sizes), each stream with its own
caller-owned workspace (icamd_pvrtc2_set_workspace) so that the library's workspace events do not serialise them.
Prints the time per 16-image pass for the one-call baseline and for every split pattern."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_amd_loader

pkg = ic_amd_loader.load_package()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
size, batch = 4096, 16
reps = int(os.environ.get("REPS", "30"))
g = torch.Generator(device=dev)
g.manual_seed(7)
src = torch.randint(0, 256, (batch, size, size, 4), dtype=torch.uint8, device=dev, generator=g)
per = pkg.encoded_size(pkg.PVRTC2, size, size)
ref = torch.empty((batch, per), dtype=torch.uint8, device=dev)
out = torch.empty((batch, per), dtype=torch.uint8, device=dev)


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def baseline():
    s = torch.cuda.current_stream()
    for _ in range(200):
        pkg.encode_device(pkg.PVRTC2, src, size, size, 4, n_images=batch, out=ref, stream=s)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(s)
        pkg.encode_device(pkg.PVRTC2, src, size, size, 4, n_images=batch, out=ref, stream=s)
    ev[reps].record(s)
    torch.cuda.synchronize()
    return median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])


def two_streams(split_a, split_b, prio_a=0, prio_b=0):
    """split_a / split_b: group sizes (images) issued on stream A / B; together they cover the batch, A's images first."""
    assert sum(split_a) + sum(split_b) == batch
    sa, sb = torch.cuda.Stream(priority=prio_a), torch.cuda.Stream(priority=prio_b)
    wa = torch.empty(pkg.pvrtc_workspace_size(size, max(split_a)), dtype=torch.uint8, device=dev)
    wb = torch.empty(pkg.pvrtc_workspace_size(size, max(split_b)), dtype=torch.uint8, device=dev)
    out.zero_()
    torch.cuda.synchronize()

    def one_pass():
        # interleave the host-side submission so that both queues fill at the same pace
        qa, qb = [], []
        i0 = 0
        for n in split_a:
            qa.append((i0, n)); i0 += n
        for n in split_b:
            qb.append((i0, n)); i0 += n
        for k in range(max(len(qa), len(qb))):
            for q, s, w in ((qa, sa, wa), (qb, sb, wb)):
                if k < len(q):
                    b, n = q[k]
                    pkg.pvrtc_set_workspace(w)
                    pkg.encode_device(pkg.PVRTC2, src[b:b + n], size, size, 4, n_images=n, out=out[b:b + n], stream=s)
        pkg.pvrtc_set_workspace(None)

    times = []
    for r in range(reps + 5):
        e0, e1, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        e0.record(sa)
        sb.wait_event(e0)
        one_pass()
        eb.record(sb)
        sa.wait_event(eb)
        e1.record(sa)
        torch.cuda.synchronize()
        if r >= 5:
            times.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(out, ref))
    return median(times), ok


print("baseline: one call, one stream: %.4f ms per 16 x 4096^2" % baseline())
patterns = [
    ([8], [8]),
    ([2, 6], [4, 4]),
    ([1, 4, 3], [3, 3, 2]),
    ([2, 4, 2], [4, 4]),
    ([4, 4, 4, 4], []),          # one stream, four groups: the cost of the launch boundaries alone
    ([2, 2, 2, 2], [1, 2, 2, 2, 1]),
    ([1, 2, 2, 2, 1], [2, 2, 2, 2]),
]
for a, b in patterns:
    if not b:
        sa = torch.cuda.current_stream()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        for i in range(reps):
            ev[i].record(sa)
            i0 = 0
            for n in a:
                pkg.encode_device(pkg.PVRTC2, src[i0:i0 + n], size, size, 4, n_images=n, out=out[i0:i0 + n], stream=sa)
                i0 += n
        ev[reps].record(sa)
        torch.cuda.synchronize()
        print("one stream, groups %s: %.4f ms" % (a, median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])))
        continue
    for pa, pb in ((0, 0), (-1, 0)):
        t, ok = two_streams(a, b, pa, pb)
        print("two streams A=%s B=%s prio=(%d,%d): %.4f ms  %s" % (a, b, pa, pb, t, "bit-identical" if ok else "MISMATCH"))
