#!/usr/bin/env python
"""Static check of the device code for memory operations the compiler put behind their OWN `s_waitcnt vmcnt(0)` -- one dependent round trip each
(how the cooperative sort lost 7 us per pass in round 4: agent-scope loads that were meant to be in flight together).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -disable-machine-licm --cuda-device-only -S -o /tmp/fvh.s fast_gicp_amd/csrc/fvh_capi.hip
    python tools/scan_serial_loads.py /tmp/fvh.s [min_single_waits]
Prints, per kernel: loads + returning atomics, full waits, full waits that cover exactly ONE operation, the largest batch behind one wait."""
import re
import sys

t = open(sys.argv[1]).read()
least = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for m in re.finditer(r"\n(_Z\w+): +; @", t):
    name = m.group(1)
    i = m.end()
    j = t.index(".Lfunc_end", i)
    seq = []
    for l in t[i:j].split("\n"):
        l = l.strip()
        if re.match(r"(global|flat|buffer|scratch)_load", l):
            seq.append("L")
        elif re.match(r"(global|flat)_atomic", l) and " sc0" in l.split(";")[0]:
            seq.append("A")
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            seq.append("W")
    s = "".join(seq)
    groups = s.split("W")
    singles = sum(1 for g in groups[:-1] if len(g) == 1)
    big = max((len(g) for g in groups), default=0)
    if singles >= least:
        print("%-72s ops %3d  full waits %3d  one-op waits %3d  max batch %2d" % (name[:72], s.count("L") + s.count("A"), s.count("W"), singles, big))
