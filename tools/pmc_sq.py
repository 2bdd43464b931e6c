#!/usr/bin/env python
"""SQ counters of a kernel from one rocprofv3 --pmc pass (per dispatch, averaged over the launches >= 8 us):
VALU busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES (quad-cycle counters, MI355X_MICROARCH.md),
VALU instructions per wave = SQ_INSTS_VALU / SQ_WAVES.   python tools/pmc_sq.py out.json workload:kernel_substr:csv ..."""
import csv
import json
import sys
from collections import defaultdict


def main(out, *specs):
    res = {}
    for spec in specs:
        wl, kern, path = spec.split(":")
        acc, n = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 8000:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
        m = {k: acc[k] / n[k] for k in acc}
        wc = m.get("SQ_WAVE_CYCLES") or float("nan")
        res[wl] = {"kernel": kern, "launches": max(n.values()) if n else 0, "counters_per_launch": {k: round(v, 1) for k, v in m.items()},
                   "valu_busy_frac_of_wave_cycles": round(m.get("SQ_ACTIVE_INST_VALU", float("nan")) / wc, 4),
                   "any_inst_active_frac": round(m.get("SQ_ACTIVE_INST_ANY", float("nan")) / wc, 4),
                   "waiting_frac_of_wave_cycles": round(m.get("SQ_WAIT_ANY", float("nan")) / wc, 4),
                   "issue_stall_frac": round(m.get("SQ_WAIT_INST_ANY", float("nan")) / wc, 4),
                   "valu_insts_per_wave": round(m.get("SQ_INSTS_VALU", float("nan")) / max(m.get("SQ_WAVES", 1), 1), 1)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
