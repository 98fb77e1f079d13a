"""bench.py's own counter passes (live_counters): the child command line, the summary and the fall-back, against a stand-in for
rocprofv3 that writes the csv a real pass would (CPU; the real passes run in the default `python bench.py` on the GPU box)."""
import json
import os
import stat
import time
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")]
import bench  # noqa: E402

STANDIN = r'''#!%(py)s
import json, os, sys
a = sys.argv[1:]
assert "--kernel-trace" in a and "--pmc" in a, a
for bad in ("--sys-trace", "-s", "--hip-trace", "--hsa-trace", "--runtime-trace", "-r", "--memory-copy-trace", "--marker-trace"):
    assert bad not in a, "a trace domain beside --pmc: %%s" %% bad
cut = a.index("--")
prog = a[cut + 1:]
assert prog[0] == sys.executable and prog[1].endswith("tools/run_workload.py") and prog[2:] == ["macro", "3"], prog
ctr = a[a.index("--pmc") + 1:a.index("--output-format")]
if "FETCH_SIZE" in ctr or "WRITE_SIZE" in ctr:
    assert len(ctr) == 1, ctr             # each alone
d = a[a.index("-d") + 1]
os.makedirs(os.path.join(d, "box"), exist_ok=True)
val = {"FETCH_SIZE": 4.0e6, "WRITE_SIZE": 8.0e6, "GRBM_GUI_ACTIVE": 8 * 1.0e6, "SQ_ACTIVE_INST_VALU": 1024 * 1.0e6 / 4 * 0.5,
       "SQ_WAVE_CYCLES": 1024 * 1.0e6 / 4 * 4.0, "SQ_WAIT_ANY": 1024 * 1.0e6 / 4 * 2.0, "SQ_INSTS_VALU": 1.0e6}
with open(os.path.join(d, "box", "1_counter_collection.csv"), "w") as f:
    f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n")
    for disp in range(3):
        for kern in ("void dhts::macro_rollout_fwd3_kernel<1, true>(int)", "void dhts::macro_rollout_bwd_fast_kernel<512, 2>(int)"):
            for c in ctr:
                # (the first dispatch is the first touch of the tape: dropped by the summary)
                f.write('%%d,"%%s",%%s,%%g\n' %% (2 * disp + ("bwd" in kern), kern, c, val.get(c, 1.0) * (3.0 if disp == 0 else 1.0)))
print("WORKLOAD " + json.dumps({"key": "macro", "name": "macro_straight_1024x512x1000", "units": 524288000, "unit": "cell-steps/s",
                                "moved_bytes": 9000000000, "fwd_ms": 2.5, "bwd_ms": 1.7, "library_code_sha16": %(sha)r}))
'''


def _standin(tmp_path, monkeypatch, sha):
    p = tmp_path / "rocprofv3"
    p.write_text(STANDIN % {"py": sys.executable, "sha": sha})
    p.chmod(p.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("DHTS_ROCPROFV3", str(p))
    monkeypatch.setitem(bench._LIVE, "issue", None)
    monkeypatch.setitem(bench._LIVE, "traffic", None)
    monkeypatch.setitem(bench._LIVE, "seconds", None)


class _W:
    name = "macro_straight_1024x512x1000"


def test_live_counters_summarise_three_passes(tmp_path, monkeypatch):
    _standin(tmp_path, monkeypatch, bench.library_code_sha16())
    assert bench.live_counters("macro", time.time() + 60.0) is True
    fwd = bench._LIVE["traffic"][_W.name]["rollout_fwd"]
    assert fwd["write_bytes"] == 8.0e6 * 1024 and fwd["fetch_bytes_corrected"] == 4.0e6 * 1024 * 2      # KiB; FETCH doubled on gfx950
    assert bench.pmc_traffic(_W, "rollout_fwd", moved=fwd["hbm_bytes"] * 1.01) == fwd["hbm_bytes"]
    assert bench.pmc_traffic(_W, "rollout_fwd", moved=fwd["hbm_bytes"] * 1.5) is None                   # disagrees with the tape: not quoted
    side = bench.issue_counters(_W, "rollout_fwd")
    assert side["vector_alu_busy"] == 0.5 and side["waves_per_simd"] == 4.0 and side["wait_any_frac"] == 0.5
    assert side["source"].startswith("measured by this run")
    assert json.dumps(side)


def test_live_counters_of_another_build_are_refused(tmp_path, monkeypatch):
    _standin(tmp_path, monkeypatch, "0123456789abcdef")
    assert bench.live_counters("macro", time.time() + 60.0) is False and bench._LIVE["traffic"] is None
    side = bench.issue_counters(_W, "rollout_fwd")             # falls back to the committed passes (or to nothing), never to a live label
    assert side is None or "not measured by this run" in side["source"]


def test_live_counters_are_skipped_under_a_profiler(monkeypatch):
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert bench.under_profiler()
    monkeypatch.delenv("ROCPROFILER_LIBRARY_CTOR")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()


def test_live_counters_give_up_at_the_deadline(tmp_path, monkeypatch):
    _standin(tmp_path, monkeypatch, bench.library_code_sha16())
    assert bench.live_counters("macro", time.time() + 5.0) is False and bench._LIVE["issue"] is None
