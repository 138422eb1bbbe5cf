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
    for tu, extra in (("pt_mk_lds", ["-fno-slp-vectorize"]), ("pt_mk_hbm", [])):
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
