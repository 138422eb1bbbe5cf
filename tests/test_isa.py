"""Build-time check on the gfx950 code of the megakernels: the tile hand-over ordering.

A wave that yields a tile writes the tile's state (RNG words, accumulator, samples left: sc1 dword stores) and then
publishes the tile in the queue (a 64-bit sc1 slot store inside queue_push). Another wave, possibly on another XCD,
continues from that state, so every state store must have been acknowledged before the slot store is issued. That is
an explicit `s_waitcnt vmcnt(0)` (pt_kernels.hip, marker PT_YIELD_STATE_STORED): no fence emits it by itself. This test
disassembles the device code (hipcc -S, no GPU needed) and checks the instruction order in every instantiation that
can hand tiles over (the counting ones never do: they run without time slices).
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "cudapathtracer_amd", "csrc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    """Device code of the two translation units that instantiate the megakernel, with the Makefile's flags."""
    d = tmp_path_factory.mktemp("isa")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
    procs = []
    for tu, extra in (("pt_mk_lds", ["-fno-slp-vectorize"]), ("pt_mk_hbm", ["-fno-slp-vectorize"])):
        out = str(d / (tu + ".s"))
        procs.append((out, subprocess.Popen(["hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, tu + ".hip")],
                                            stderr=subprocess.DEVNULL)))
    text = ""
    for out, pr in procs:
        assert pr.wait() == 0, out
        text += open(out).read()
    return text


def _functions(text):
    """{mangled name: body} of the megakernel entry points."""
    out = {}
    for m in re.finditer(r"^(_ZN2pt\d+megakernel\w*):.*?^\s*s_endpgm", text, flags=re.S | re.M):
        out[m.group(1)] = m.group(0)
    return out


def _is_counting(name):
    # megakernel<INTEG, COUNT, ...> / megakernel_hbm<INTEG, COUNT, ...>: the second template argument
    return re.search(r"megakernel(?:_hbm)?ILi\dELb1", name) is not None


def test_state_stores_are_waited_for_before_the_queue_entry(isa):
    fns = _functions(isa)
    assert len(fns) >= 28, sorted(fns)
    checked = 0
    for name, body in fns.items():
        lines = [ln.strip() for ln in body.split("\n")]
        marks = [i for i, ln in enumerate(lines) if "PT_YIELD_STATE_STORED" in ln]
        if _is_counting(name):
            continue
        assert marks, "%s: the hand-over marker is missing" % name
        for i in marks:
            # (1) the explicit wait follows the marker directly
            nxt = next(ln for ln in lines[i + 1:] if ln and not ln.startswith(";"))
            assert re.match(r"s_waitcnt\s+vmcnt\(0\)", nxt), (name, nxt)
            # (2) before it, in the same straight-line run of code: the tile's state stores (sc1 dword stores)
            back = []
            for ln in reversed(lines[:i]):
                if re.match(r"(s_cbranch|s_branch|s_endpgm|\.LBB)", ln):
                    break
                back.append(ln)
            assert any(re.match(r"global_store_dword\s.*\bsc1\b", ln) for ln in back), "%s: no state store ahead of the wait" % name
            # (3) after it: queue_push — the next vector-memory instruction is its claim on q[1] (returning atomic add at
            #     byte offset 4), and the 64-bit slot store comes later still
            after = [ln for ln in lines[i + 2:] if re.match(r"(global|flat|buffer)_", ln)]    # (scratch reloads are the wave's own spill slots)
            assert re.match(r"global_atomic_add\s+v\d+,.*offset:4\b", after[0]), (name, after[0])
            assert any(re.match(r"global_store_dwordx2\s.*\bsc1\b", ln) for ln in after[1:]), "%s: no slot store after the hand-over wait" % name
            checked += 1
    assert checked >= 15, checked


def test_no_agent_scope_fences_in_the_megakernels(isa):
    """An agent-scope release / acquire writes back / invalidates the whole L2 on gfx950 (buffer_wbl2 / buffer_inv sc1):
    measured 3x slower on the 263 k-triangle scene. The tile queue must not need one."""
    for name, body in _functions(isa).items():
        assert "buffer_wbl2" not in body and not re.search(r"buffer_inv\s+sc1", body), name


def _metadata(text):
    """{mangled kernel name: {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}} from the code-object notes."""
    out = {}
    cur = None
    for ln in text.split("\n"):
        s = ln.strip()
        m = re.match(r"\.name:\s+(_ZN2pt\d+megakernel\w+)$", s)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)$", s)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return {k: v for k, v in out.items() if len(v) == 4}


def test_register_budgets_and_private_segments_of_the_timed_kernels(isa):
    """What the timed instantiations cost in registers and scratch, pinned so that a change that pushes state back across
    the traversal loops shows up here and not as a slower frame (the DEFER path state was cut to what a
    path needs ACROSS a traversal — no pending-shadow ray, one NEE term instead of three factors, one Li for both the
    finished and the new path, no reciprocal direction, no min_t, no pixel coordinates):
      * the headline kernels (LDS-resident, 4 waves per SIMD, 128 VGPRs) and the 4-wave SIMPLE kernel an 8-GPU share runs:
        no private segment at all;
      * the production kernel for scenes in HBM (8 waves per SIMD): 64 VGPRs, and a private segment that may not grow — its
        logic step peaks at ~125 live registers (no spill at a cap of 128, 31 spill sites at 96, 86 at 80, 159 at 64), the
        price of eight waves per SIMD that measures 2 % FASTER than six with 86 (profiles/r03_ab_hbm_simple_shapes_*.log). Without the
        SLP vectorizer's packed pairs (Makefile, FLAGS_pt_mk_hbm) the segment is 220 B instead of 272 and the frame 3.4-3.8 % faster."""
    md = _metadata(isa)
    flat2 = md["_ZN2pt16megakernel_flat2ILi0ELb1ELb0EEEvNS_7KParamsE"]
    assert flat2["vgpr_count"] <= 128 and flat2["private_segment_fixed_size"] == 0 and flat2["vgpr_spill_count"] == 0, flat2
    share = md["_ZN2pt10megakernelILi0ELb0ELb0ELb0ELb1ELb0ELb1ELi1ELb0EEEvNS_7KParamsE"]    # megakernel<0, false, false, false, true, false, true, 1, false>: 4-wave REFILL SIMPLE
    assert share["vgpr_count"] <= 128 and share["private_segment_fixed_size"] == 0 and share["vgpr_spill_count"] == 0, share
    hbm = md["_ZN2pt21megakernel_hbm_simpleILi0EEEvNS_7KParamsE"]
    assert hbm["vgpr_count"] == 64 and hbm["private_segment_fixed_size"] <= 224, hbm
    gen = md["_ZN2pt14megakernel_hbmILi0ELb0ELb0ELb1ELb0ELb0EEEvNS_7KParamsE"]                # generic bounce (all arms), REFILL, 6 waves per SIMD
    assert gen["vgpr_count"] == 80 and gen["private_segment_fixed_size"] <= 320, gen
    # the LEAN generic bounce (no leaf arms, no texture fetches: scenes of glass, mirrors, metals) against the all-arms one
    lean = md["_ZN2pt14megakernel_hbmILi0ELb0ELb0ELb1ELb0ELb1EEEvNS_7KParamsE"]
    assert lean["vgpr_count"] == 80 and lean["private_segment_fixed_size"] <= 240, lean
    lean4 = md["_ZN2pt10megakernelILi0ELb0ELb0ELb0ELb1ELb0ELb0ELi1ELb1EEEvNS_7KParamsE"]
    assert lean4["private_segment_fixed_size"] <= 56, lean4
    pair = md["_ZN2pt16megakernel_flat2ILi0ELb0ELb1EEEvNS_7KParamsE"]
    assert pair["vgpr_count"] <= 128 and pair["private_segment_fixed_size"] <= 52, pair      # (12 registers around the logic step since the branch-free NEE triangle test: still +0.4 %)


def _hot_loops(body):
    """Innermost loops of a kernel that fetch a node or a triangle from global memory (>= 3 global_load_dwordx4, < 200 instructions):
    (instructions, VALU, v_readlane / v_writelane, scratch instructions) per loop."""
    lines = body.split("\n")
    pos = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            pos[m.group(1)] = i
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            ins = [x.strip() for x in lines[pos[m.group(1)]:i + 1] if x.strip() and not x.strip().startswith((";", ".L"))]
            if sum(1 for x in ins if x.startswith("global_load_dwordx4")) >= 3 and len(ins) < 200:
                out.append((len(ins), sum(1 for x in ins if x.startswith("v_")), sum(1 for x in ins if "readlane" in x or "writelane" in x),
                            sum(1 for x in ins if x.startswith("scratch_"))))
    return out


def test_traversal_loops_of_the_kernels_for_scenes_in_hbm_hold_no_spill_code(isa):
    """The node loop and the triangle loop are where a scene in HBM spends its time (~65 VALU instructions per trip at a third of
    the lanes). Round 3 learnt the hard way that a lane-varying `bool` kept across them becomes an SGPR pair, and that two more
    live SGPR pairs made the compiler spill SGPRs INSIDE the loops (8 v_readlane / v_writelane per trip: -3 % on both scenes,
    profiles/r03_ab_state_shrink.log) — state that crosses the loops lives in VGPR bits. No spill code of either kind in there."""
    fns = _functions(isa)
    for name in ("_ZN2pt21megakernel_hbm_simpleILi0EEEvNS_7KParamsE", "_ZN2pt14megakernel_hbmILi0ELb0ELb0ELb1ELb0ELb0EEEvNS_7KParamsE",
                 "_ZN2pt14megakernel_hbmILi0ELb0ELb0ELb1ELb0ELb1EEEvNS_7KParamsE", "_ZN2pt10megakernelILi0ELb0ELb0ELb0ELb1ELb0ELb1ELi1ELb0EEEvNS_7KParamsE"):
        loops = _hot_loops(fns[name])
        assert len(loops) >= 2, (name, loops)
        assert all(ls == 0 and sc == 0 for (_, _, ls, sc) in loops), (name, loops)
        assert min(n for (n, _, _, _) in loops) <= 120, (name, loops)          # the node loop: 116 instructions, 62 of them VALU


def test_node_fetch_keeps_lds_and_global_reads_in_flight_together(isa):
    """pt_trace.h: load_node (PT_NODE_OVERLAP): the lanes whose node is in HBM issue their four global loads, the lanes whose node
    is in the LDS copy of the tree top their four LDS reads into the SAME registers, and only then the wave waits — the compiler's own
    if/else puts `s_waitcnt vmcnt(0)` between the two groups. Checked on the code as built: no wait between the first global load
    and the last LDS read of a fetch, one wait for both counters after it."""
    fns = _functions(isa)
    checked = 0
    for name, body in fns.items():
        if "megakernel_hbm" not in name:
            continue
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";")]
        for i, ln in enumerate(lines):
            if not re.match(r"\.Lng\d+:", ln):
                continue
            # back to the fetch's first instruction, forward to its wait
            j = i
            while not lines[j].startswith("s_mov_b64") or "exec" not in lines[j]:
                j -= 1
            k = i
            while not lines[k].startswith("s_waitcnt"):
                k += 1
            block = lines[j:k + 1]
            assert sum(1 for x in block if x.startswith("global_load_dwordx4")) == 4, (name, block)
            assert sum(1 for x in block if x.startswith("ds_read_b128")) == 4, (name, block)
            assert sum(1 for x in block if x.startswith("s_waitcnt")) == 1 and re.match(r"s_waitcnt\s+vmcnt\(0\)\s+lgkmcnt\(0\)", block[-1]), (name, block)
            checked += 1
    assert checked >= 8, checked


def test_f32_denormals_are_preserved_in_every_megakernel(isa):
    """pt_trace.h: slab_hit folds `tmax > 0` into `tmax >= FLT_TRUE_MIN`; a build that flushed f32 denormals (-fgpu-flush-denormals-to-zero,
    -Ofast) would turn that constant into 0 and accept tmax == 0. The kernel descriptors say what the waves run with."""
    modes = re.findall(r"\.amdhsa_float_denorm_mode_32\s+(\d+)", isa)
    assert len(modes) >= 28 and set(modes) == {"3"}, sorted(set(modes))
