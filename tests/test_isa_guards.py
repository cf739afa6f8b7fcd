"""CPU tier: build-time guards on the generated gfx950 code of kernels whose correctness leans on hand-counted waits.

icamd_pvrtc2_encode_kernel fetches its pixel rows by LDS-DMA and its colour rows from inline asm, with `s_waitcnt vmcnt(4)`
counted by hand (pvrtc_kernels.hip): the colour registers of block row j >= 2 count as loaded only after the
third row wait that follows their request.  That holds only while hipcc neither spills those registers nor touches them
(a v_mov, a use scheduled early) before that wait.  The GPU tier would catch a miscompile as a parity failure; this test
catches it at build time, on the assembly hipcc emits today: no scratch, and no instruction between the in-loop colour
loads and the third `s_waitcnt vmcnt(4)` after them names one of their destination registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image-compression_amd", "csrc")


def _asm(tu, tmp_path, extra=()):
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = os.path.join(str(tmp_path), "k.s")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, tu)] + list(extra),
                          stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


def _kernel_meta(text, name):
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        if re.search(r"\.name:\s+%s\s" % re.escape(name), blk):
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
            return {"vgprs": g("vgpr_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}
    raise AssertionError("kernel %s not found in the code-object metadata" % name)


def _body(text, name):
    lines = text.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def _regs(operand_text):
    """VGPR numbers named in an instruction's operand text: v12, v[12:15]."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", operand_text))
    return out


def test_pvrtc_encode_kernel_hand_counted_waits_still_hold(tmp_path):
    text = _asm("pvrtc_kernels.hip", tmp_path)
    meta = _kernel_meta(text, "icamd_pvrtc2_encode_kernel")
    assert meta["scratch"] == 0, "icamd_pvrtc2_encode_kernel spills: in-flight colour registers could be stored to scratch"
    body = _body(text, "icamd_pvrtc2_encode_kernel")
    loop = next(i for i, l in enumerate(body) if "Inner Loop Header" in l)
    loads = [i for i in range(loop, len(body)) if re.match(r"\s+global_load_dwordx2\s", body[i])][:3]
    assert len(loads) == 3 and loads[2] - loads[0] <= 4, "the strip loop's three colour loads were not found together"
    in_flight = set()
    for i in loads:
        in_flight |= _regs(body[i].split("global_load_dwordx2")[1].split(",")[0])
    assert len(in_flight) == 6
    waits, checked = 0, 0
    for l in body[loads[2] + 1:]:
        if re.match(r"\s+s_waitcnt vmcnt\(0\)", l):
            continue  # the j <= 1 path waits for everything right away (branched around for j >= 2)
        if re.match(r"\s+s_waitcnt vmcnt\(4\)", l):
            waits += 1
            if waits == 3:
                break
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s+(.*?)(;.*)?$", l)
        if not m or m.group(1).startswith("s_") or l.strip().startswith((";", ".")):
            continue
        touched = _regs(m.group(2)) & in_flight
        assert not touched, "v%s used before its load is known to have landed: %s" % (sorted(touched), l.strip())
        checked += 1
    assert waits == 3 and checked > 300, (waits, checked)


def test_pvrtc_register_path_build_still_compiles(tmp_path):
    """-DICAMD_PVRTC_NO_ROW_DMA (the register path without any hand-counted wait) is the fallback if a compiler change
    ever breaks the guard above: keep it building and spill-free."""
    text = _asm("pvrtc_kernels.hip", tmp_path, ["-DICAMD_PVRTC_NO_ROW_DMA"])
    assert _kernel_meta(text, "icamd_pvrtc2_encode_kernel")["scratch"] == 0


def test_pvrtc_onepass_kernel_ring_protocol_as_compiled(tmp_path):
    """icamd_pvrtc2_onepass_kernel (r05) counts its waits by hand too: every tick of the strip walk issues exactly ONE pixel
    row (two global_load_lds) and waits with `s_waitcnt vmcnt(4)` -- the two youngest rows may be in flight, everything
    older has landed.  That only holds while hipcc adds no memory operation of its own to the walk: any LDS access or load
    it could see would come with a vmcnt(0) that drains the ring (a performance bug), a hoisted or duplicated DMA would
    break the count (a correctness bug).  Checked on the emitted assembly: no scratch; the 176-VGPR claim that caps a SIMD
    at two waves is in the descriptor; the loop body holds exactly 4 ticks' worth of DMA and waits, one barrier, and no
    other vmcnt wait; every LDS / memory instruction of the kernel sits inside an inline-asm region or is a DMA."""
    text = _asm("pvrtc_kernels.hip", tmp_path)
    meta = _kernel_meta(text, "icamd_pvrtc2_onepass_kernel")
    assert meta["scratch"] == 0
    assert 171 <= meta["vgprs"] <= 256, meta
    assert meta["lds"] == 0  # the ring is dynamic LDS, sized per workgroup width by the launcher
    body = _body(text, "icamd_pvrtc2_onepass_kernel")
    loop = next(i for i, l in enumerate(body) if "Inner Loop Header" in l)
    back = max(i for i, l in enumerate(body) if re.match(r"\s+s_cbranch_\w+\s+\.LBB\d+_\d+", l))
    in_asm, stray = False, []
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif re.match(r"\s+(ds_|global_|buffer_|flat_|scratch_)", l) and not in_asm and "global_load_lds" not in l:
            stray.append(l.strip())
    assert not stray, "memory / LDS instructions outside the inline asm of the walk: %s" % stray[:4]
    pre = [l for l in body[:loop] if "global_load_lds_dwordx4" in l]
    assert len(pre) == 6 + 3 * 2, "prologue: three rows up front + the three morph-only ticks (got %d DMA instructions)" % len(pre)
    walk = body[loop:back + 1]
    dma = sum("global_load_lds_dwordx4" in l for l in walk)
    waits = [re.sub(r"\s+", " ", l.strip()) for l in walk if "s_waitcnt" in l and "vmcnt" in l]
    assert dma == 8, dma
    assert waits == ["s_waitcnt vmcnt(4)"] * 4, waits
    assert sum(bool(re.match(r"\s+s_barrier", l)) for l in walk) == 1
    tail = [re.sub(r"\s+", " ", l.strip()) for l in body[back + 1:] if "s_waitcnt" in l and "vmcnt" in l]
    assert tail == ["s_waitcnt vmcnt(0)"], tail


def test_pvrtc_onepass_halo_kernel_keeps_the_ring_protocol(tmp_path):
    """r06: icamd_pvrtc2_onepass_halo_kernel is the same walk with a prologue in front (the workgroup's share of the halo table
    into LDS, the 4 K column-0 values right of its last lane) and two scalar address selects in the exchange.  The prologue is
    plain code (the reduction of the three halo columns with the pair path's routine, the edge values) -- its loads and LDS
    accesses sit in front of the first DMA and end in a barrier, so they may be visible to hipcc --
    but from the first row request on the hand-counted protocol must be exactly the plain kernel's: the loop holds 4 ticks' worth
    of DMA and `vmcnt(4)` waits, one barrier, no other vmcnt wait, no memory / LDS instruction outside the inline asm."""
    text = _asm("pvrtc_kernels.hip", tmp_path)
    meta = _kernel_meta(text, "icamd_pvrtc2_onepass_halo_kernel")
    assert meta["scratch"] == 0 and 171 <= meta["vgprs"] <= 256 and meta["lds"] == 0, meta
    body = _body(text, "icamd_pvrtc2_onepass_halo_kernel")
    first_dma = next(i for i, l in enumerate(body) if "global_load_lds_dwordx4" in l)
    loop = max(i for i, l in enumerate(body) if "Inner Loop Header" in l)   # (the prologue's copy loop comes first)
    assert loop > first_dma
    back = max(i for i, l in enumerate(body) if re.match(r"\s+s_cbranch_\w+\s+\.LBB\d+_\d+", l))
    assert any(re.match(r"\s+s_barrier", l) for l in body[:first_dma]), "the prologue must end in a barrier before the first row request"
    in_asm, stray = False, []
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif i > first_dma and re.match(r"\s+(ds_|global_|buffer_|flat_|scratch_)", l) and not in_asm and "global_load_lds" not in l:
            stray.append(l.strip())
    assert not stray, "memory / LDS instructions outside the inline asm of the walk: %s" % stray[:4]
    pre = [l for l in body[first_dma:loop] if "global_load_lds_dwordx4" in l]
    assert len(pre) == 6 + 3 * 2, len(pre)
    walk = body[loop:back + 1]
    waits = [re.sub(r"\s+", " ", l.strip()) for l in walk if "s_waitcnt" in l and "vmcnt" in l]
    assert sum("global_load_lds_dwordx4" in l for l in walk) == 8 and waits == ["s_waitcnt vmcnt(4)"] * 4, waits
    assert sum(bool(re.match(r"\s+s_barrier", l)) for l in walk) == 1
    tail = [re.sub(r"\s+", " ", l.strip()) for l in body[back + 1:] if "s_waitcnt" in l and "vmcnt" in l]
    assert tail == ["s_waitcnt vmcnt(0)"], tail


def test_pvrtc_onepass_kernel_plain_scan_build_still_compiles(tmp_path):
    """-DICAMD_PVRTC_NO_SCAN_SDWA (the early-exit scan as plain C++ instead of the VCC / SDWA sequence) stays buildable."""
    text = _asm("pvrtc_kernels.hip", tmp_path, ["-DICAMD_PVRTC_NO_SCAN_SDWA"])
    assert _kernel_meta(text, "icamd_pvrtc2_onepass_kernel")["scratch"] == 0


def test_etc1_encode_kernels_keep_four_waves_per_simd(tmp_path):
    """The ETC1 encoders are VALU-bound at 0.98 of the issue ceiling WITH four waves per SIMD: the exhaustive
    search is pinned at 128 VGPRs by amdgpu_waves_per_eu(4) and pays for it with two folded spills (8 bytes of scratch) in
    the kSmallerError kernels.  Nothing else guards that balance: a compiler update or a source change that spills more --
    or drops below four waves -- would cost the headline C4 figure silently.  Pinned on the emitted code objects: every
    encode kernel <= 128 VGPRs; scratch <= 8 bytes for kSmallerError, none for the split and heuristic kernels."""
    text = _asm("etc1_kernels.hip", tmp_path)
    for src in ("rgb888", "rgba8"):
        for variant, max_scratch in (("", 8), ("_split_h", 0), ("_split_v", 0), ("_heuristic", 0)):
            name = "icamd_etc1_%s%s_kernel" % (src, variant)
            meta = _kernel_meta(text, name)
            assert meta["vgprs"] <= 128, (name, meta)  # 512 VGPRs per SIMD lane / 4 waves
            assert meta["scratch"] <= max_scratch, (name, meta)
            assert meta["lds"] == 0, (name, meta)
        # r05: the four-lanes-per-block form of small launches -- a latency play, it must stay light (8 waves per SIMD) and
        # keep its quad exchange
        name = "icamd_etc1_%s_quad_kernel" % src
        meta = _kernel_meta(text, name)
        assert meta["vgprs"] <= 64 and meta["scratch"] == 0 and meta["lds"] == 0, (name, meta)
        assert sum("quad_perm" in l for l in _body(text, name)) >= 3, name


@pytest.mark.parametrize("flags", [[], ["-DICAMD_PVRTC_SCALAR_ROW"]])
def test_pvrtc_encode_kernel_colour_rows_are_not_used_before_their_wait(tmp_path, flags):
    """ADVICE r04 / r05 root cause: colour rows -1, 0 and 1 of a strip are loaded by one inline asm and waited for by a
    SECOND asm (`s_waitcnt vmcnt(0)`).  Until r05 that wait named no operands, so nothing stopped hipcc from scheduling a
    VALU use of a just-requested register in front of it -- which is exactly what r04's scalar-row build did (wrong first
    blocks of strips).  The wait now carries the registers as "+v" operands; this pins the consequence on the emitted code of
    the shipped build AND of the build that used to break: between such a load group and the vmcnt(0) that follows it, no
    instruction names one of its destination registers."""
    text = _asm("pvrtc_kernels.hip", tmp_path, flags)
    body = _body(text, "icamd_pvrtc2_encode_kernel")
    groups = 0
    i = 0
    while i < len(body):
        if not re.match(r"\s+global_load_dwordx2\s", body[i]):
            i += 1
            continue
        dests, j = set(), i
        while j < len(body) and (re.match(r"\s+global_load_dwordx2\s", body[j]) or body[j].strip().startswith(";")):
            if "global_load_dwordx2" in body[j]:
                dests |= _regs(body[j].split("global_load_dwordx2")[1].split(",")[0])
            j += 1
        wait = next((k for k in range(j, min(j + 60, len(body))) if re.match(r"\s+s_waitcnt vmcnt\(0\)", body[k])), None)
        nxt_branch = next((k for k in range(j, min(j + 60, len(body))) if re.match(r"\s+s_cbranch", body[k])), None)
        if len(dests) == 6 and wait is not None and (nxt_branch is None or True):
            # (the in-loop group for j >= 2 branches around its vmcnt(0); only the straight-line distance to the wait is
            # checked here, the branched case is test_pvrtc_encode_kernel_hand_counted_waits_still_hold's)
            for l in body[j:wait]:
                m = re.match(r"\s+([a-z_0-9]+)\s+(.*?)(;.*)?$", l)
                if not m or m.group(1).startswith("s_") or l.strip().startswith((";", ".")):
                    continue
                if nxt_branch is not None and nxt_branch < wait:
                    break
                assert not (_regs(m.group(2)) & dests), "v%s used before the wait for its load: %s" % (sorted(_regs(m.group(2)) & dests), l.strip())
            groups += 1
        i = j
    assert groups >= 2, groups  # colour rows -1 and 0 in front of the strip loop


def test_pvrtc4_onepass_kernel_ring_protocol_as_compiled(tmp_path):
    """The 4 bpp one-pass kernel (extension) uses the 2 bpp kernel's ring protocol with ONE DMA instruction per row: a tick
    issues one row and waits with `s_waitcnt vmcnt(2)`.  A 1 024-lane workgroup caps it at 128 VGPRs; no scratch."""
    text = _asm("pvrtc_kernels.hip", tmp_path)
    meta = _kernel_meta(text, "icamd_pvrtc4_onepass_kernel")
    assert meta["scratch"] == 0 and meta["vgprs"] <= 128 and meta["lds"] == 0, meta
    body = _body(text, "icamd_pvrtc4_onepass_kernel")
    loop = next(i for i, l in enumerate(body) if "Inner Loop Header" in l)
    back = max(i for i, l in enumerate(body) if re.match(r"\s+s_cbranch_\w+\s+\.LBB\d+_\d+", l))
    in_asm, stray = False, []
    for l in body:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif re.match(r"\s+(ds_|global_|buffer_|flat_|scratch_)", l) and not in_asm and "global_load_lds" not in l:
            stray.append(l.strip())
    assert not stray, stray[:4]
    assert sum("global_load_lds_dwordx4" in l for l in body[:loop]) == 3 + 3
    walk = body[loop:back + 1]
    waits = [re.sub(r"\s+", " ", l.strip()) for l in walk if "s_waitcnt" in l and "vmcnt" in l]
    assert sum("global_load_lds_dwordx4" in l for l in walk) == 4 and waits == ["s_waitcnt vmcnt(2)"] * 4, waits
    assert sum(bool(re.match(r"\s+s_barrier", l)) for l in walk) == 1


def test_pvrtc_decode_tile_kernel_writes_whole_lines(tmp_path):
    """r05: the decoder's bound was its store pattern.  As compiled: each pixel row of a wave goes through the
    per-wave LDS turn (two 16-byte writes, two 16-byte reads 1 KiB apart) and leaves as two NON-TEMPORAL 16-byte stores -- eight
    per block, none plain --, the LDS read of a row follows its writes in program order, nothing spills, and the kernel keeps
    at least 7 waves per SIMD (72 VGPRs)."""
    text = _asm("decode_kernels.hip", tmp_path)
    name = "icamd_pvrtc2_decode_tile_kernel"
    meta = _kernel_meta(text, name)
    assert meta["scratch"] == 0 and meta["vgprs"] <= 72, meta
    ops = [l.split(";")[0].strip() for l in _body(text, name)]
    stores = [l for l in ops if l.startswith("global_store")]
    assert len(stores) == 8 and all(l.startswith("global_store_dwordx4") and l.endswith(" nt") for l in stores), stores
    seq = [("w" if l.startswith("ds_write_b128") else "r" if l.startswith("ds_read_b128") else "s")
           for l in ops if l.startswith(("ds_write_b128", "ds_read_b128", "global_store_dwordx4"))]
    tail = "".join(seq)[-24:]   # the four rows: (write, write, read, read, store, store) each; the colour pairs' reads come first
    assert tail == "wwrrss" * 4, "".join(seq)
    reads = [l for l in ops if l.startswith("ds_read_b128")][-8:]
    assert sum("offset:1024" in l for l in reads) == 4, reads


def test_etc1_pad_quad_kernel_keeps_the_copy_at_full_occupancy(tmp_path):
    """r05: the kSmallerError Pad is one launch -- pad blocks (four lanes each) and the copy of the image's own blocks in the same
    kernel.  That only works while the split search stays within 64 VGPRs (8 waves per SIMD for the copy workgroups; r04's
    whole search took 121 and was split off into its own launch for that reason), without scratch."""
    text = _asm("blockops_kernels.hip", tmp_path)
    meta = _kernel_meta(text, "icamd_pad_etc1_quad_kernel")
    assert meta["scratch"] == 0 and meta["vgprs"] <= 64, meta
    body = [l.split(";")[0].strip() for l in _body(text, "icamd_pad_etc1_quad_kernel")]
    assert sum(l.startswith("v_mov_b32_dpp") and "quad_perm" in l for l in body) >= 3, "the quad exchange is gone"
