"""tools/roofline.py: the arithmetic behind bench.py's `roofline` block, and the committed evidence it is re-derived from.

No GPU: the committed bench line of the round (the newest profiles/rNN_v*_bench.json) must follow from the committed counter
sets (profiles/roofline_inputs.json) — every fraction at most 1, recomputable to the last digit — and those counter sets must
have been measured on the very kernels the built library holds (code_sha256): editing a kernel without re-running
tools/profile_round.sh + tools/collect_round.py turns this file red."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import roofline as RF  # noqa: E402


def _bench_files():
    def key(p):
        r, v = os.path.basename(p).split("_")[:2]
        return (int(r[1:]), int(v[1:]))
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_v*_bench.json")), key=key)


def _latest_bench():
    line = [ln for ln in open(_bench_files()[-1]).read().splitlines() if ln.startswith("{")][-1]
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
    assert os.path.basename(_bench_files()[-1]).startswith("r03_")
    assert RF.check(b)
    rows = [b] + b["secondary"]
    megak = [r for r in rows if r["config"]["variant"] == "megakernel"]
    assert len(megak) == 5 and all(0.0 < r["roofline"]["frac"] <= 1.0 and not r["roofline"].get("stale_profile") for r in megak)
    assert b["metric"] == "Mray/s" and b["config"]["workload"] == "cornell_1920x1080_1024spp_depth8_mis" and b["frame_equals_counted_frame"]
    assert b["roofline"]["kernel_ms"] <= b["ms_per_step"] and b["roofline"]["bound"] == "valu"
    names = [s["config"]["workload"].split(" ")[0] for s in b["secondary"]]
    assert names == ["blob82k_1920x1080_1024spp_depth8_mis", "atrium262k_1920x1080_4096spp_depth16_mis", "cornell_mixed_1920x1080_1024spp_depth8_mis",
                     "blob82k_glass_1920x1080_1024spp_depth8_mis", "atrium262k_1920x1080_4096spp_depth16_mis"]
    for s in b["secondary"]:
        rf = s["roofline"]
        if rf["bound"] == "l1_lines" and s["config"]["variant"] == "megakernel":
            assert list(rf)[0] == "ta_busy_frac" and 0.0 < rf["ta_busy_frac"] <= 1.0 and "micro-benchmark" in rf["peak_source"]
    # the general bounce (material dispatch, medium stack, dielectric / GGX / mirror arms) is in the driver's line, through its LEAN kernels
    assert b["secondary"][2]["roofline"]["kernel"] == "pt::megakernel_flat2<0, false, true>" and b["secondary"][2]["config"]["kernel_flags"]["lean"]
    assert b["secondary"][3]["roofline"]["kernel"] == "pt::megakernel_hbm<0, false, false, true, false, true>"
    assert b["secondary"][4]["config"]["variant"] == "wavefront"                      # BASELINE configs[4]: the divergence A/B
    assert set(b["projected_scaling"]) >= {"2", "4", "8"} and 1.0 < b["projected_scaling"]["8"] <= 8.0
    assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1
    # round 3: every instruction class per clock and CU next to what a CU was measured to issue (tools/issue_rate): the headline kernel's
    # VALU share of the nominal rate is `frac`; its total is close to that ceiling — which is why fewer instructions, not hidden latency, pay
    iss = b["roofline"]["issue"]
    assert iss["waves_per_simd"] == 4 and 0.85 < iss["frac_of_ceiling"] <= 1.0 and iss["valu"] < iss["ceiling_valu_alone"] and iss["salu"] < iss["ceiling_salu_alone"]
    assert abs(iss["valu"] / 2.0 - b["roofline"]["frac"]) < 0.03            # the same VALU count against the profiled launch's own clocks


def test_committed_counters_were_measured_on_the_built_kernels(api):
    """profiles/roofline_inputs.json carries the hash of each profiled kernel's machine code; the newest entry of every
    (workload, spp, kernel) the bench line uses must be the code libptamd.so holds NOW — otherwise bench.py would report
    `frac: null, stale_profile: true` on the driver's box (and this test says so first)."""
    hashes = RF.kernel_code_hashes(api.LIB_PATH)
    assert len(hashes) >= 30 and all(len(v) == 64 for v in hashes.values())
    b = _latest_bench()
    inputs = RF.load_inputs()
    checked = 0
    for row in [b] + b["secondary"]:
        rf = row["roofline"]
        if row["config"]["variant"] != "megakernel":
            continue
        e = RF.find_entry(inputs, row["config"]["workload"].split(" ")[0], row["config"]["spp"], rf["kernel"])
        assert e is not None, rf["kernel"]
        assert RF.profile_is_current(e, hashes), "%s was edited after its PMC pass (%s): re-run tools/profile_round.sh + tools/collect_round.py" % (rf["kernel"], e.get("source"))
        assert rf["code_sha256"] == e["code_sha256"]
        checked += 1
    assert checked == 5
    stale = dict(e, code_sha256="0" * 64)
    assert not RF.profile_is_current(stale, hashes) and not RF.profile_is_current(None, hashes)


def test_every_committed_build_of_the_round_rendered_the_same_full_frames():
    """bench.py hashes the timed frame (and fails if it differs from the counting kernel's frame): the full 1080p frames of the
    three workloads are the same bits in every build whose bench line is committed — kernels, layouts and options changed, the
    image did not."""
    seen = {}
    for f in _bench_files():
        b = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        for row in [b] + b.get("secondary", []):
            if row.get("frame_sha"):
                seen.setdefault(row["config"]["workload"], set()).add(row["frame_sha"])
    assert len(seen) == 5 and all(len(v) == 1 for v in seen.values()), seen       # rounds 2 and 3: kernels, layouts and node numbering changed, the frames did not


def test_rocprof_average_of_the_headline_kernel_agrees_with_the_bench_line():
    """profiles/<tag>_kernel_stats_headline.csv: rocprofv3 --kernel-trace --stats of `bench.py --no-secondary` (the headline's kernel
    launched on whole frames only — the default command also launches it on 1/2, 1/4, 1/8 of the tiles for `projected_scaling`, which
    pulls that summary's AVERAGE away from a frame's duration; its MAX is the frame). Both must agree with the line's kernel_ms."""
    import csv
    bench = _bench_files()[-1]
    tag = os.path.basename(bench)[:-len("_bench.json")]
    b = _latest_bench()
    name, ms = b["roofline"]["kernel"], b["roofline"]["kernel_ms"]
    head = os.path.join(os.path.dirname(bench), tag + "_kernel_stats_headline.csv")
    full = os.path.join(os.path.dirname(bench), tag + "_kernel_stats.csv")
    assert os.path.exists(head) and os.path.exists(full), (head, full)
    row = [r for r in csv.DictReader(open(head)) if name in r["Name"]]
    assert len(row) == 1 and int(row[0]["Calls"]) >= 3
    assert abs(float(row[0]["AverageNs"]) * 1e-6 - ms) < 0.01 * ms, (row[0]["AverageNs"], ms)
    row = [r for r in csv.DictReader(open(full)) if name in r["Name"]]
    assert len(row) == 1 and abs(float(row[0]["MaxNs"]) * 1e-6 - ms) < 0.01 * ms, (row[0]["MaxNs"], ms)
