#!/usr/bin/env python
"""Static instruction mix of the persistent LM kernel between the FVH_MARK comments (-DFVH_ASM_MARKS):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm --cuda-device-only -S -DFVH_ASM_MARKS -o /tmp/fvh.s fast_gicp_amd/csrc/fvh_capi.hip
    python tools/count_isa.py /tmp/fvh.s [mangled kernel name]"""
import collections
import re
import sys

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "_ZN3fvh11cost_kernelIdLi0ELb1ELi4ELb0EEEvNS_10CostParamsE"
t = open(path).read()
i = t.index(name + ":")
body = t[i:t.index(".Lfunc_end", i)].split("\n")
sec = "pre"
counts = collections.OrderedDict()
for l in body:
    l = l.strip()
    m = re.match(r"; FVH_MARK (\d+)", l)
    if m:
        sec = "after mark " + m.group(1)
        continue
    if not l or l.startswith((";", ".", "_")) or l.endswith(":"):
        continue
    op = l.split()[0]
    c = counts.setdefault(sec, collections.Counter())
    if op.startswith("v_"):
        kind = "valu_f64" if "f64" in op else ("valu_trans" if any(x in op for x in ("rcp", "rsq", "sqrt", "sin", "cos", "exp", "log")) else "valu_other")
        if any(x in op for x in ("rcp_f64", "rsq_f64", "sqrt_f64")):
            kind = "valu_trans64"
    elif op.startswith("s_"):
        kind = "salu"
    elif op.startswith("scratch_"):
        kind = "scratch"
    elif op.startswith(("global_", "flat_", "buffer_")):
        kind = "vmem"
    elif op.startswith("ds_"):
        kind = "lds"
    else:
        kind = "other"
    c[kind] += 1
    c["all"] += 1
print("%-16s %6s %8s %8s %10s %6s %5s %5s %7s" % ("section", "all", "valu_f64", "trans64", "valu_other", "salu", "vmem", "lds", "scratch"))
for k, c in counts.items():
    print("%-16s %6d %8d %8d %10d %6d %5d %5d %7d" % (k, c["all"], c["valu_f64"], c["valu_trans64"], c["valu_other"] + c["valu_trans"], c["salu"], c["vmem"], c["lds"], c["scratch"]))
