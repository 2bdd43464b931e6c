#!/usr/bin/env python
"""Folds the rocprofv3 --pmc passes of one round into ONE json that bench.py reads (profiles/rNN_pmc.json):
    python tools/pmc_collect.py out.json [--pairs pairs.json] entry:kernel_substr:fetch_csv:write_csv:sq_csv[:min_ns] [...]
(any csv may be '-'). Per entry: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes:
both in KiB, FETCH_SIZE doubled on gfx950), and from the SQ pass: VALU instructions per wave, VALU-busy / waiting fractions of the
wave cycles, and the chip-level VALU ISSUE utilisation  SQ_INSTS_VALU * 2 / (launch duration x clock x #SIMDs): the share of the peak issue
rate of one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32; 157.3 TFLOP/s
= 1024 SIMDs x 64 lanes x 2 flop / 2 cycles x 2.4 GHz). No instruction issues faster, so this is <= 1 by construction.
The denominator is the kernel's OWN duration (kernel trace) times the clock of the pass -- the median of GRBM_GUI_ACTIVE / 8 XCDs / duration
over the launches of at least CLOCK_MIN_NS of that pass, capped at MAX_CLOCK_GHZ. (Round 5 divided by GRBM_GUI_ACTIVE itself: that counter
spans more than a short kernel -- it implied clocks of 3.4-7.0 GHz for every kernel under 20 us, i.e. understated their utilisation by that
ratio, VERDICT r5 weak #9. `grbm_clock_ghz` keeps the raw ratio so that the discrepancy stays visible; tests/test_tools_cpu.py pins the rule.) (Round 4 divided
SQ_ACTIVE_INST_VALU x 4 by the SIMD cycles: that counter is summed per WAVE in quad-cycles and waves of one SIMD overlap in the pipe -- it
read 1.04 on the 1M-point k-NN. It is still recorded, as `valu_active_quadcycles_over_simd_cycles`, and not called a utilisation.)
`--pairs file.json` (tools/pair_counts.py) adds what the culled searches actually evaluate, for the flop fraction of the VALU stages.
The file is stamped with the git commit and a hash of fast_gicp_amd/csrc/: bench.py refuses to quote it once the kernels changed."""
import csv
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4
XCDS = 8  # GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (2.6 M "cycles" for a 135 us kernel = 8 x 2.44 GHz): one XCD's count is the chip's clock
MAX_CLOCK_GHZ = 2.5     # MI355X peak engine clock is 2.4 GHz: a derived clock above this is a counter window wider than the kernel, not a clock
NOMINAL_CLOCK_GHZ = 2.4
CLOCK_MIN_NS = 50000    # launches at least this long calibrate the clock of a pass (start / stop skew of the counter window is < 2 % of them)


def pass_clock_ghz(path):
    """(clock in GHz, how it was found) of one SQ pass: the median GRBM_GUI_ACTIVE / XCDS / duration over its LONG launches of any kernel."""
    ratios = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if d >= CLOCK_MIN_NS:
                ratios.append(float(r["Counter_Value"]) / XCDS / d)
    ratios = sorted(x for x in ratios if 0.5 <= x <= MAX_CLOCK_GHZ)
    if not ratios:
        return NOMINAL_CLOCK_GHZ, "nominal (no launch of >= %d us with a plausible GRBM_GUI_ACTIVE in this pass)" % (CLOCK_MIN_NS // 1000)
    return ratios[len(ratios) // 2], "median over %d launches of >= %d us of this pass" % (len(ratios), CLOCK_MIN_NS // 1000)


def issue_utilisation(insts_valu, dur_ns, clock_ghz):
    """SQ_INSTS_VALU (wave instructions per launch) x 2 cycles / (SIMD cycles the launch lasted)"""
    return insts_valu * 2.0 / (dur_ns * clock_ghz * SIMDS)


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fast_gicp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def per_launch(path, kern, min_ns=8000):
    acc, n, dur = defaultdict(float), defaultdict(int), []
    seen = set()
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if d >= min_ns:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
                key = (r.get("Dispatch_Id"), r["Start_Timestamp"])
                if key not in seen:
                    seen.add(key)
                    dur.append(d)
    return {k: acc[k] / n[k] for k in acc}, (max(n.values()) if n else 0), (sum(dur) / len(dur) if dur else None)


def main(out, *specs):
    pairs = {}
    if specs and specs[0] == "--pairs":
        if os.path.exists(specs[1]):
            pairs = json.load(open(specs[1]))
        specs = specs[2:]
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = os.environ.get("FVH_COMMIT", "unknown")  # (the GPU box has no .git: the artifacts script passes the commit in)
    res = {"_meta": {"commit": commit, "csrc_sha": csrc_sha(),
                     "method": "rocprofv3 --kernel-trace --pmc <one counter group per pass>; launches >= 8 us; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                               "(gfx950: FETCH_SIZE reports half of wide coalesced reads); profiled passes run at lower clocks than un-profiled ones"}}
    clocks = {}
    for spec in specs:
        parts = spec.split(":")
        name, kern, fcsv, wcsv, scsv = parts[:5]
        min_ns = int(parts[5]) if len(parts) > 5 else 8000  # (launches shorter than this are the early-exit / no-op ones of a multi-launch route)
        e = {"kernel": kern}
        if fcsv != "-" and wcsv != "-" and os.path.exists(fcsv) and os.path.exists(wcsv):
            f, nf, _ = per_launch(fcsv, kern, min_ns)
            w, nw, _ = per_launch(wcsv, kern, min_ns)
            if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
                e.update(fetch_size_kib_raw=round(f["FETCH_SIZE"], 1), write_size_kib_raw=round(w["WRITE_SIZE"], 1), launches_fetch_pass=nf, launches_write_pass=nw,
                         hbm_bytes_per_launch=int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024))
        if scsv != "-" and os.path.exists(scsv):
            m, ns, dur = per_launch(scsv, kern, min_ns)
            wc = m.get("SQ_WAVE_CYCLES") or float("nan")
            e["sq"] = {"launches": ns, "avg_launch_us_in_this_pass": None if dur is None else round(dur / 1e3, 2), "counters_per_launch": {k: round(v, 1) for k, v in m.items()},
                       "valu_insts_per_wave": round(m.get("SQ_INSTS_VALU", float("nan")) / max(m.get("SQ_WAVES", 1), 1), 1),
                       "valu_busy_frac_of_wave_cycles": round(m.get("SQ_ACTIVE_INST_VALU", float("nan")) / wc, 4),
                       "waiting_frac_of_wave_cycles": round(m.get("SQ_WAIT_ANY", float("nan")) / wc, 4),
                       "issue_stall_frac_of_wave_cycles": round(m.get("SQ_WAIT_INST_ANY", float("nan")) / wc, 4)}
            if dur and "SQ_INSTS_VALU" in m:
                if scsv not in clocks:
                    clocks[scsv] = pass_clock_ghz(scsv)
                clock, how = clocks[scsv]
                assert clock <= MAX_CLOCK_GHZ
                simd_cycles = dur * clock * SIMDS
                util = issue_utilisation(m["SQ_INSTS_VALU"], dur, clock)
                assert not util > 1.0, "a VALU issue utilisation above 1: the normalisation is wrong (%s: %r)" % (name, util)
                e["sq"]["valu_issue_utilisation"] = round(util, 4)
                e["sq"]["valu_active_quadcycles_over_simd_cycles"] = round(m.get("SQ_ACTIVE_INST_VALU", float("nan")) * 4.0 / simd_cycles, 4)  # (per-wave sum: NOT bounded by 1)
                e["sq"]["effective_clock_ghz"] = round(clock, 3)
                e["sq"]["clock_source"] = how
                if m.get("GRBM_GUI_ACTIVE"):
                    e["sq"]["grbm_clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / XCDS / dur, 3)  # raw: > 2.5 means the counter window was wider than this (short) kernel
                e["sq"]["valu_wave_instructions_per_sec"] = round(m.get("SQ_INSTS_VALU", 0.0) / (dur * 1e-9), 1)
        if name in pairs:
            e["pairs"] = pairs[name]
        res[name] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
