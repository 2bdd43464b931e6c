#!/usr/bin/env python
"""Per-section listing of loads and waits of an LM kernel instantiation (build with -DFVH_ASM_MARKS, see tools/count_isa.py):
L global load, F flat load, d LDS read, Wn = s_waitcnt vmcnt(n), kn = lgkmcnt(n). A `W0` right behind a group of loads that the
arithmetic of the section was meant to hide is what to look for.   python tools/section_waits.py /tmp/fvh.s [mangled name ...]"""
import re
import sys

t = open(sys.argv[1]).read()
names = sys.argv[2:] or ["_ZN3fvh11cost_kernelIdLi0ELb1ELi4ELb0EEEvNS_10CostParamsE", "_ZN3fvh11cost_kernelIdLi2ELb1ELi1ELb0EEEvNS_10CostParamsE"]
for name in names:
    i = t.index(name + ":")
    j = t.index(".Lfunc_end", i)
    sec, out, order = "pre", {}, []
    for l in t[i:j].split("\n"):
        l = l.strip()
        m = re.match(r"; FVH_MARK (\d+)", l)
        if m:
            sec = "m" + m.group(1)
            continue
        if sec not in out:
            out[sec] = []
            order.append(sec)
        if re.match(r"(global|buffer)_load", l):
            out[sec].append("L")
        elif re.match(r"flat_load", l):
            out[sec].append("F")
        elif re.match(r"ds_(read|load)", l):
            out[sec].append("d")
        elif l.startswith("s_waitcnt"):
            mm, ll = re.search(r"vmcnt\((\d+)\)", l), re.search(r"lgkmcnt\((\d+)\)", l)
            out[sec].append((" W" + mm.group(1) if mm else " ") + ("k" + ll.group(1) if ll else "") + " ")
    print(name)
    for s in order:
        print("  %-5s %s" % (s, "".join(out[s])[:300]))
