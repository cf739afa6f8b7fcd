"""CPU tier: bench.py's command line -- BASELINE.json config presets and the explicit-flag overrides -- resolves to
the workloads DESIGN.md section 5 names, without needing a GPU."""
import importlib.util
import os

import ic_testlib as T


def _bench():
    spec = importlib.util.spec_from_file_location("icamd_bench", os.path.join(T.ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_default_is_config_c2_and_presets_match_baseline_json():
    b = _bench()
    a = b.parse_args([])
    assert (a.preset, a.workload, a.size, a.batch, a.total_textures) == ("c2", "dxt1_rgba8", 4096, 16, None)
    a = b.parse_args(["--config", "c3"])
    assert (a.workload, a.size, a.batch) == ("dxt5_rgba8", 8192, 4)
    a = b.parse_args(["--config", "c4", "--gpus", "8"])
    assert (a.workload, a.size, a.total_textures, a.etc_strategy) == ("etc1_rgb888", 1024, 1024, 2)
    a = b.parse_args(["--config", "c5"])
    assert (a.workload, a.size, a.batch) == ("pvrtc2_rgba8", 4096, 16)
    import json
    cfgs = json.load(open(os.path.join(T.ROOT, "BASELINE.json")))["configs"]
    assert "4096" in cfgs[1] and "DXT1" in cfgs[1] and "8192" in cfgs[2] and "1024" in cfgs[3] and "ETC1" in cfgs[3]


def test_explicit_flags_leave_the_presets():
    b = _bench()
    a = b.parse_args(["--workload", "etc1_rgb888", "--size", "256", "--batch", "5", "--etc-strategy", "3"])
    assert (a.preset, a.workload, a.size, a.batch, a.total_textures, a.etc_strategy) == (None, "etc1_rgb888", 256, 5, None, 3)
    a = b.parse_args(["--config", "c4", "--batch", "7"])  # an explicit batch turns config 4 into a weak-scaling run
    assert (a.total_textures, a.batch) == (None, 7)
    for wl, (codec, comps, bpp, label, unit) in b.WORKLOADS.items():
        assert unit in ("hbm", "valu") and bpp == comps + {0: 0.5, 1: 1.0, 2: 0.5, 3: 0.25, 4: 0.5}[codec]  # 4: PVRTC 4 bpp (extension)


def test_live_counter_passes_degrade_to_the_committed_profile(monkeypatch, tmp_path):
    """bench.py counts roofline.traffic / valu_frac in the run itself (rocprofv3 --pmc child passes).  Where that cannot
    work -- no rocprofv3, the run itself under a profiler, a failing pass (this container: no GPU) -- it must say so, fall
    back to the committed profile's figure with its own label, and not try again for the next leg."""
    b = _bench()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "x")      # as inside `rocprofv3 -- python bench.py`
    t, why = b.live_traffic("dxt1_rgba8", 256, 1, "noise", 2)
    assert t is None and "being profiled" in why
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER"))]:
        monkeypatch.delenv(k)
    monkeypatch.setattr(b.shutil if hasattr(b, "shutil") else __import__("shutil"), "which", lambda name: None)
    t, why = b.live_traffic("dxt1_rgba8", 256, 1, "noise", 2)
    assert t is None and "not on PATH" in why
    # a failing pass: a stand-in "rocprofv3" that exits 1 at once
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 1\n")
    fake.chmod(0o755)
    monkeypatch.undo()
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER"))]:
        monkeypatch.delenv(k)
    b2 = _bench()
    t, why = b2.live_traffic("dxt1_rgba8", 256, 1, "noise", 2)
    assert t is None and "failed" in why and b2._LIVE_TRAFFIC_OFF
    fake.write_text("#!/bin/sh\nsleep 600\n")            # would hang: must not even be started again
    t2, why2 = b2.live_traffic("dxt5_rgba8", 256, 1, "noise", 2)
    assert t2 is None and why2 == why
    traffic, source, committed = b2.measured_or_committed_traffic(True, "c2", "dxt1_rgba8", 4096, 16, "noise", 2)
    assert traffic == committed and traffic and "not measured in this run" in source and "live measurement unavailable" in source
