#!/usr/bin/env python
"""Register / LDS / occupancy report of every gfx950 kernel of the engine, from hipcc's own resource-usage remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed). The LM kernel sits exactly at the 256-VGPR limit of
two workgroups per CU: one more live value costs half the occupancy, which this report (and tests/test_capi_cpu.py) catch
before a GPU run does.   python tools/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool] + names, capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            pass
    return {n: n for n in names}


def kernel_resources():
    """{demangled kernel name: {vgprs, sgprs, scratch, occupancy, sgpr_spill, vgpr_spill, lds}}"""
    src = os.path.join(ROOT, "fast_gicp_amd", "csrc", "fvh_capi.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        sys.path.insert(0, ROOT)
        from fast_gicp_amd import build as _build
        cmd = [hipcc] + _build.HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "t.so"), src, "-ldl"] + os.environ.get("FVH_EXTRA_HIPCC_FLAGS", "").split()
        err = subprocess.run(cmd, capture_output=True, text=True, check=True, cwd=tmp).stderr
    res, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for label, key in FIELDS.items():
            m = re.search(re.escape(label) + r": (\d+)", line)
            if m:
                cur[key] = int(m.group(1))
    names = demangle(list(res)) if res else {}
    return {names[k]: dict(v, mangled=k) for k, v in res.items()}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    r = kernel_resources()
    print("%-96s %5s %5s %7s %4s %9s %9s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "VGPRspill", "SGPRspill", "LDS"))
    for k in sorted(r):
        if pat in k:
            v = r[k]
            print("%-96s %5d %5d %7d %4d %9d %9d %7d" % (k[:96], v.get("vgprs", -1), v.get("sgprs", -1), v.get("scratch", -1), v.get("occupancy", -1), v.get("vgpr_spill", -1),
                                                             v.get("sgpr_spill", -1), v.get("lds", -1)))
