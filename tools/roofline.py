#!/usr/bin/env python3
"""Roofline arithmetic of bench.py, in one place — and a CLI that re-derives a bench line from the committed profiles.

    python tools/roofline.py collect  OUTDIR WORKLOAD SPP [--kernel-ms MS]   # rocprofv3 pass CSVs (tools/pmc_passes.sh) -> one entry (JSON on stdout)
    python tools/roofline.py add      ENTRY.json                             # merge an entry into profiles/roofline_inputs.json
    python tools/roofline.py check    BENCH.json                             # recompute every roofline number of a bench line from the inputs

Which roofline binds which kernel ("Roofline"):

  * LDS-resident scenes (Cornell, BASELINE C2: the headline). The packed scene is 5 KB and sits in LDS; HBM sees 4 GB/s.
    The kernel is bound by VALU ISSUE: a SIMD issues one wave64 VALU instruction per 2 cycles (32 lanes per cycle,
    MI355X_MICROARCH.md "vector-instruction ISSUE cost" / the 157.3 TFLOP/s f32 vector peak), i.e. 1024 SIMDs x 2.4 GHz / 2
    = 1228.8 G wave-instructions/s. achieved = SQ_INSTS_VALU of one launch (rocprofv3 PMC pass on this workload, committed)
    / the launch's duration measured live. The fraction is an upper bound on usefulness: `valu_lanes_frac` (active
    lanes per VALU instruction, SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)) says how much of each issued instruction
    did work.
  * Scenes in HBM (82 k / 263 k triangles, C3-C5). 79 % of the node fetches hit the 32 KB L1; what limits them is the L1's
    tag rate: a divergent 16-byte load costs one cache-line access per lane, and a CU retires ONE line access per clock
    (tools/ta_rate/quad_fetch.hip: 152 G 64-byte records/s = 609 G line accesses/s chip-wide). peak = 256 CUs x 2.4 GHz =
    614.4 G accesses/s. achieved = (4 per internal-node fetch that missed the LDS scene cache + 3 per triangle test)
    / duration — counted live by the counting pass of bench.py (pt_debug_stamps()[0], tri_tests). `ta_busy_frac` next to it
    is TA_TA_BUSY / TCP_GATE_EN1 of the committed PMC pass of the same workload: how busy the unit behind the bound was.

SURVEY §8(d)'s ALGORITHMIC bytes (32 B/box test + 16 B/node + 52 B/triangle test + 96 B/hit + 16 B/pixel) stay in the bench
line as a labelled secondary figure: divided by the HBM peak they exceed 1 on every scene here, because those bytes are served
by LDS (Cornell) or L1/L2 — that ratio is not a roofline fraction and is no longer reported as one.
"""
import csv
import glob
import re
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "profiles", "roofline_inputs.json")

CUS, SIMDS, CLOCK_HZ = 256, 1024, 2.4e9                      # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
PEAK_VALU = SIMDS * CLOCK_HZ / 2.0                            # wave64 VALU instructions / s (one per 2 cycles per SIMD)
PEAK_L1_LINES = CUS * CLOCK_HZ                                # cache-line accesses / s (one per clock per CU)
PEAK_HBM_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_LDS_GBS = CUS * 128 * CLOCK_HZ / 1e9                     # 128 B/clk/CU


# ---- which build were the committed counters measured on? --------------------------------------------------------
# Every entry of profiles/roofline_inputs.json carries `code_sha256`: the hash of the profiled kernel's machine code
# (the bytes of its symbol in the gfx950 code object inside libptamd.so). bench.py recomputes it from the library it is
# about to time; a different hash means the kernel was edited after the PMC pass, and the line then reports
# `"frac": null, "stale_profile": true` instead of dividing an old instruction count by a new time.
LIB = os.path.join(ROOT, "cudapathtracer_amd", "csrc", "libptamd.so")
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_symbols(elf):
    """(name, bytes) of every FUNC symbol of a little-endian ELF64 image (the AMDGPU code object)."""
    import struct
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]     # name, type, flags, addr, offset, size, link, info, align, entsize
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ != 2 or entsize == 0:                                   # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for k in range(size // entsize):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + k * entsize)
            if (st_info & 0xf) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= shnum:           # STT_FUNC, defined
                continue
            end = elf.index(b"\0", stroff + st_name)
            sec = secs[st_shndx]
            start = sec[4] + (st_value - sec[3])
            yield elf[stroff + st_name:end].decode(), elf[start:start + st_size]


def kernel_code_hashes(lib_path=LIB):
    """{demangled kernel name (as rocprofv3 prints it, without `void ` and the parameter list): sha256 of its code} for every
    megakernel / wavefront kernel in the gfx950 code objects bundled in libptamd.so. {} if the library is missing."""
    import hashlib
    import struct
    import subprocess
    try:
        blob = open(lib_path, "rb").read()
    except OSError:
        return {}
    found = {}
    pos = 0
    while True:
        i = blob.find(_BUNDLE_MAGIC, pos)
        if i < 0:
            break
        pos = i + len(_BUNDLE_MAGIC)
        n, = struct.unpack_from("<Q", blob, i + 24)
        o = i + 32
        if n > 16:
            continue
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tlen]
            o += 24 + tlen
            if b"gfx950" in triple and size:
                for name, code in _elf_symbols(blob[i + off:i + off + size]):
                    if "megakernel" in name or "wf_" in name:
                        found[name] = hashlib.sha256(code).hexdigest()
    if not found:
        return {}
    names = sorted(found)
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            break
        except (OSError, subprocess.CalledProcessError):
            out = None
    if not out or len(out) < len(names):
        return dict(found)                                             # mangled names only: nothing will match, every profile reads as stale
    pretty = {}
    for m, d in zip(names, out):
        d = re.sub(r"^void ", "", d.strip())
        d = re.sub(r"\(pt::KParams\)$", "", d)
        d = re.sub(r"\(.*\)$", "", d) if d.endswith(")") else d
        pretty[d] = found[m]
    return pretty


def profile_is_current(entry, hashes):
    """True if the committed counters were measured on the very code the loaded library holds for that kernel."""
    return bool(entry) and bool(entry.get("code_sha256")) and hashes.get(entry.get("kernel")) == entry.get("code_sha256")


def alg_bytes(c, n_px):
    """32 B/box test, 16 B/node pop, 52 B/triangle test, 96 B/accepted hit, 16 B/pixel."""
    return 32 * c["box_tests"] + 16 * c["node_pops"] + 52 * c["tri_tests"] + 96 * c["hits"] + 16 * n_px


def load_inputs(path=INPUTS):
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return {"entries": []}


def find_entry(inputs, workload, spp, kernel=None):
    best = None
    for e in inputs.get("entries", []):
        if e.get("workload") == workload and e.get("spp") == spp and (kernel is None or e.get("kernel") == kernel):
            best = e                                              # the last matching entry is the newest
    return best


def traffic_bytes(entry):
    """HBM-side bytes per launch from the PMC passes: (2 x FETCH_SIZE + WRITE_SIZE) KB — the x2 is the gfx950 correction
    of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 bytes). None without both counters."""
    if not entry or "FETCH_SIZE" not in entry or "WRITE_SIZE" not in entry:
        return None
    return int((2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024)


def valu_roofline(entry, kernel_ms):
    """bound 'valu': wave-instructions issued per second against the SIMDs' issue rate."""
    if not entry or not entry.get("SQ_INSTS_VALU") or kernel_ms <= 0:
        return None
    achieved = entry["SQ_INSTS_VALU"] / (kernel_ms * 1e-3)
    out = {"bound": "valu", "achieved": achieved / 1e9, "peak": PEAK_VALU / 1e9, "unit": "Gwave-instr/s", "frac": achieved / PEAK_VALU,
           "valu_insts_per_launch": entry["SQ_INSTS_VALU"]}
    if entry.get("SQ_THREAD_CYCLES_VALU"):
        out["valu_lanes_frac"] = entry["SQ_THREAD_CYCLES_VALU"] / (64.0 * entry["SQ_INSTS_VALU"])
    if entry.get("GRBM_GUI_ACTIVE") and entry.get("kernel_ms"):
        out["clock_ghz_under_load"] = entry["GRBM_GUI_ACTIVE"] / 8.0 / (entry["kernel_ms"] * 1e-3) / 1e9     # sum over 8 XCDs
    return out


def l1_roofline(global_node_fetches, global_tri_tests, kernel_ms):
    """bound 'l1_lines': divergent 16-byte loads, one cache-line access per lane per load."""
    if kernel_ms <= 0:
        return None
    lines = 4 * global_node_fetches + 3 * global_tri_tests
    achieved = lines / (kernel_ms * 1e-3)
    return {"bound": "l1_lines", "achieved": achieved / 1e9, "peak": PEAK_L1_LINES / 1e9, "unit": "Gline-access/s", "frac": achieved / PEAK_L1_LINES,
            "line_accesses_per_launch": lines}


# What one CU issued per clock in tools/issue_rate/issue_rate.hip (profiles/r03_issue_rate_microbench.log; rates at the nominal 2.4 GHz):
# the best instruction stream we could construct, by waves per SIMD. VALU alone: 1.68 (4 waves) / 1.75 (8); the scalar unit alone 0.95.
ISSUE_CEILING = {4: 1.97, 6: 2.13, 8: 2.23}          # (those streams are three v_add per s_nop / s_waitcnt; of v_add + s_add + branches: 1.81 / 1.90 / 2.06)
ISSUE_VALU_ALONE = {4: 1.68, 6: 1.78, 8: 1.80}
ISSUE_SALU_ALONE = 0.95


def issue_block(entry, waves_per_simd):
    """Instructions of every class the profiled launch issued per clock and CU (SQ_INSTS*, clocks = GRBM_GUI_ACTIVE / 8 XCDs) next to
    the ceilings of the issue-rate micro-benchmark at the kernel's waves per SIMD. None without the counters of that PMC pass."""
    need = ("SQ_INSTS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE")
    if not entry or any(not entry.get(k) for k in need):
        return None
    clk = entry["GRBM_GUI_ACTIVE"] / 8.0 * CUS
    per = lambda k: entry.get(k, 0.0) / clk
    w = 8 if waves_per_simd >= 8 else (6 if waves_per_simd >= 6 else 4)
    out = {"unit": "instructions / clock / CU", "total": per("SQ_INSTS"), "valu": per("SQ_INSTS_VALU"), "salu": per("SQ_INSTS_SALU"),
           "branch": per("SQ_INSTS_BRANCH"), "lds": per("SQ_INSTS_LDS"), "vmem": per("SQ_INSTS_VMEM"), "smem": per("SQ_INSTS_SMEM"),
           "waves_per_simd": waves_per_simd, "ceiling_total": ISSUE_CEILING[w], "ceiling_valu_alone": ISSUE_VALU_ALONE[w],
           "ceiling_salu_alone": ISSUE_SALU_ALONE, "ceiling_source": "tools/issue_rate/issue_rate.hip, profiles/r03_issue_rate_microbench.log"}
    out["frac_of_ceiling"] = out["total"] / out["ceiling_total"]
    return out


def ta_busy(entry):
    """Scenes in HBM: the fraction of its cycles a CU's texture addresser (the unit the L1 line rate belongs to) was busy in the
    profiled launch — TA_TA_BUSY summed over the CUs / TCP_GATE_EN1 summed over the CUs (the L1's clock). None without both."""
    if not entry or not entry.get("TA_TA_BUSY_sum") or not entry.get("TCP_GATE_EN1_sum"):
        return None
    return entry["TA_TA_BUSY_sum"] / entry["TCP_GATE_EN1_sum"]


def algorithmic(own_bytes, kernel_ms):
    g = own_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return {"bytes_per_launch": own_bytes, "gbps": g, "over_hbm_peak": g / PEAK_HBM_GBS, "over_lds_peak": g / PEAK_LDS_GBS,
            "note": "SURVEY 8(d) bytes / kernel time; served from LDS / L1 / L2, so the ratio to the HBM peak is not a roofline fraction"}


# ---- CLI ---------------------------------------------------------------------------------------------------------
def collect(outdir, workload, spp, kernel_ms=None):
    """Counters of the TIMED megakernel instantiation (second template argument COUNT = false) of one launch; a pass
    profiles bench.py --steps 1 --warmup 0, i.e. one counting launch (ignored here) and one timed launch."""
    per = {}
    for f in sorted(glob.glob(os.path.join(outdir, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0].strip()
                if "megakernel" not in name or re.search(r"megakernel(_hbm)?<\d, ?true", name):      # counting instantiations: COUNT = true
                    continue
                d = per.setdefault(name, {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if len(per) != 1:
        raise SystemExit("expected exactly one timed megakernel instantiation in %s, found %s" % (outdir, sorted(per)))
    name, tot = next(iter(per.items()))
    e = {"workload": workload, "spp": int(spp), "kernel": name.replace("void ", ""), "source": os.path.relpath(outdir, ROOT)}
    e["code_sha256"] = kernel_code_hashes().get(e["kernel"])       # the library the passes ran on (they run it in-tree)
    if not e["code_sha256"]:
        raise SystemExit("no code hash for %s in %s" % (e["kernel"], LIB))
    e.update(tot)
    if kernel_ms:
        e["kernel_ms"] = float(kernel_ms)
    return e


def check(bench_line):
    """Recompute every roofline number of a bench line from profiles/roofline_inputs.json. A row whose `frac` is null
    (stale profile: the kernel was edited after its PMC pass; or no pass committed) is reported and counts as NOT ok
    unless it says why (`stale_profile`, a `note`, or no `profile` block at all)."""
    b = json.loads(bench_line) if isinstance(bench_line, str) else bench_line
    inputs = load_inputs()
    rows = [("headline", b)] + [("secondary %d" % i, s) for i, s in enumerate(b.get("secondary", []))]
    ok = True
    for label, line in rows:
        rf, cfg = line["roofline"], line["config"]
        ms = rf["kernel_ms"]
        if rf.get("frac") is None:
            explained = bool(rf.get("stale_profile") or rf.get("note") or "profile" not in rf)
            ok &= explained
            print("%-12s %-45s bound %-8s frac null (%s)" % (label, cfg["workload"][:45], rf["bound"],
                                                             "stale profile" if rf.get("stale_profile") else (rf.get("note") or "no PMC pass committed")[:60]))
            continue
        if rf["bound"] == "valu":
            e = find_entry(inputs, rf["profile"]["workload"], rf["profile"]["spp"], rf["kernel"])
            want = valu_roofline(e, ms)
            if e is not None and rf.get("code_sha256") and e.get("code_sha256") and e["code_sha256"] != rf["code_sha256"]:
                want = None                                            # the line used counters of other code
        else:
            want = l1_roofline(rf["global_node_fetches_per_launch"], rf["global_tri_tests_per_launch"], ms)
        same = want is not None and abs(want["frac"] - rf["frac"]) <= 1e-9 * max(1.0, rf["frac"]) and rf["frac"] <= 1.0
        ok &= same
        print("%-12s %-45s bound %-8s frac %.4f (recomputed %.4f) %s" % (label, cfg["workload"][:45], rf["bound"], rf["frac"], want["frac"] if want else float("nan"),
                                                                          "ok" if same else "MISMATCH"))
    return ok


def main(argv):
    if len(argv) >= 4 and argv[0] == "collect":
        ms = float(argv[argv.index("--kernel-ms") + 1]) if "--kernel-ms" in argv else None
        print(json.dumps(collect(argv[1], argv[2], argv[3], ms), indent=1))
    elif len(argv) == 2 and argv[0] == "add":
        inputs = load_inputs()
        inputs.setdefault("entries", []).append(json.load(open(argv[1])))
        json.dump(inputs, open(INPUTS, "w"), indent=1)
    elif len(argv) == 2 and argv[0] == "check":
        lines = [ln for ln in open(argv[1]).read().splitlines() if ln.startswith("{")]
        sys.exit(0 if check(lines[-1]) else 1)
    else:
        print(__doc__)
        sys.exit(2)


if __name__ == "__main__":
    main(sys.argv[1:])
