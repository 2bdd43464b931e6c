#!/usr/bin/env python
"""Static check of the device code for memory operations the compiler put behind their OWN `s_waitcnt vmcnt(0)` -- one dependent round trip each
(how the cooperative sort lost 7 us per pass in round 4, and the LM kernel ~6 round trips per item: loads that were meant to be in flight together).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -disable-machine-licm --cuda-device-only -S -o /tmp/fvh.s fast_gicp_amd/csrc/fvh_capi.hip
    python tools/scan_serial_loads.py /tmp/fvh.s [min_single_waits]
Prints, per kernel: loads + returning atomics, FLAT loads (their wait is vmcnt AND lgkmcnt: they drain everything in flight), full waits, full waits that
cover exactly ONE operation, the largest batch behind one wait.  tests/test_isa_cpu.py pins a few of these numbers."""
import re
import sys


def scan(text):
    """{mangled kernel name: {ops, flat, full_waits, one_op_waits, max_batch}}"""
    out = {}
    for m in re.finditer(r"\n(_Z\w+): +; @", text):
        name = m.group(1)
        i = m.end()
        j = text.index(".Lfunc_end", i)
        seq, flat = [], 0
        for l in text[i:j].split("\n"):
            l = l.strip()
            if re.match(r"(global|flat|buffer|scratch)_load", l):
                seq.append("L")
                flat += l.startswith("flat_load")
            elif re.match(r"(global|flat)_atomic", l) and " sc0" in l.split(";")[0]:
                seq.append("A")
            elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                seq.append("W")
        s = "".join(seq)
        groups = s.split("W")
        out[name] = {"ops": s.count("L") + s.count("A"), "flat": flat, "full_waits": s.count("W"), "one_op_waits": sum(1 for g in groups[:-1] if len(g) == 1),
                     "max_batch": max((len(g) for g in groups), default=0)}
    return out


if __name__ == "__main__":
    least = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for name, v in scan(open(sys.argv[1]).read()).items():
        if v["one_op_waits"] >= least:
            print("%-72s ops %3d  flat %2d  full waits %3d  one-op waits %3d  max batch %2d" % (name[:72], v["ops"], v["flat"], v["full_waits"], v["one_op_waits"], v["max_batch"]))
