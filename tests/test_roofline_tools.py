"""tools/roofline.py: the arithmetic behind bench.py's `roofline` block, and the committed evidence it is re-derived from.

No GPU: the committed bench line of the round (profiles/r02_v16_bench.json) must follow from the committed counter sets
(profiles/roofline_inputs.json) — every fraction at most 1, recomputable to the last digit."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import roofline as RF  # noqa: E402


def _latest_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_v*_bench.json")), key=lambda p: int(os.path.basename(p).split("_")[1][1:]))
    line = [ln for ln in open(files[-1]).read().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_peaks_are_the_microarchitecture_guides():
    assert RF.PEAK_VALU == 1024 * 2.4e9 / 2 and RF.PEAK_L1_LINES == 256 * 2.4e9 and RF.PEAK_HBM_GBS == 8000.0


def test_valu_and_l1_rooflines_are_fractions():
    e = {"SQ_INSTS_VALU": 6.0e11, "SQ_THREAD_CYCLES_VALU": 64 * 6.0e11 * 0.5}
    r = RF.valu_roofline(e, 1000.0)
    assert r["bound"] == "valu" and abs(r["frac"] - 6.0e11 / RF.PEAK_VALU) < 1e-12 and abs(r["valu_lanes_frac"] - 0.5) < 1e-12
    assert RF.valu_roofline(None, 1000.0) is None and RF.valu_roofline(e, 0.0) is None
    l1 = RF.l1_roofline(10, 20, 1.0)
    assert l1["line_accesses_per_launch"] == 4 * 10 + 3 * 20
    assert RF.ta_busy({"TA_TA_BUSY_sum": 95.0, "TCP_GATE_EN1_sum": 100.0}) == 0.95 and RF.ta_busy({"TA_TA_BUSY_sum": 1.0}) is None
    assert RF.traffic_bytes({"FETCH_SIZE": 1.0, "WRITE_SIZE": 2.0}) == 4 * 1024 and RF.traffic_bytes({}) is None


def test_committed_bench_line_follows_from_committed_inputs(capsys):
    b = _latest_bench()
    assert RF.check(b)
    rows = [b] + b["secondary"]
    assert all(0.0 < r["roofline"]["frac"] <= 1.0 for r in rows)
    assert b["metric"] == "Mray/s" and b["config"]["workload"] == "cornell_1920x1080_1024spp_depth8_mis" and b["frame_equals_counted_frame"]
    assert b["roofline"]["kernel_ms"] <= b["ms_per_step"]
    for s in b["secondary"]:
        assert s["roofline"]["bound"] == "l1_lines" and 0.0 < s["roofline"]["ta_busy_frac"] <= 1.0
    assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1


def test_every_committed_build_of_the_round_rendered_the_same_full_frames():
    """bench.py hashes the timed frame (and fails if it differs from the counting kernel's frame): the full 1080p frames of the
    three workloads are the same bits in every build whose bench line is committed — kernels, layouts and options changed, the
    image did not."""
    seen = {}
    for f in glob.glob(os.path.join(ROOT, "profiles", "r02_v*_bench.json")):
        b = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        for row in [b] + b.get("secondary", []):
            if row.get("frame_sha"):
                seen.setdefault(row["config"]["workload"], set()).add(row["frame_sha"])
    assert len(seen) == 3 and all(len(v) == 1 for v in seen.values()), seen
