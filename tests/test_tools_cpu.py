"""CPU tests of the measurement tooling (tools/): numbers the judge reads must not come out of a wrong denominator."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write_pass(path, launches):
    """a rocprofv3 counter_collection.csv: one row per (launch, counter)"""
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Counter_Name", "Counter_Value"])
        w.writeheader()
        t = 1000
        for i, (kern, dur_ns, counters) in enumerate(launches):
            for k, v in counters.items():
                w.writerow({"Dispatch_Id": i, "Kernel_Name": kern, "Start_Timestamp": t, "End_Timestamp": t + dur_ns, "Counter_Name": k, "Counter_Value": v})
            t += dur_ns + 5000


def test_pmc_utilisation_uses_the_kernels_own_duration_and_a_physical_clock(tmp_path):
    """GRBM_GUI_ACTIVE spans more than a short kernel (it implied 3.4-7.0 GHz for every kernel under 20 us), so dividing by it
    understated their VALU issue utilisation by that ratio. The rule now: duration of the launch (kernel trace) x the clock of the pass, calibrated
    on the pass's LONG launches and never above 2.5 GHz on a 2.4 GHz part."""
    pc = _load("pmc_collect")
    sq = str(tmp_path / "sq.csv")
    long_k = ("cost_kernel<double>", 120000, {"GRBM_GUI_ACTIVE": 8 * 2.3 * 120000, "SQ_INSTS_VALU": 4.0e6, "SQ_WAVES": 1896, "SQ_WAVE_CYCLES": 1e9, "SQ_ACTIVE_INST_VALU": 1e8, "SQ_WAIT_ANY": 6e8, "SQ_WAIT_INST_ANY": 1e8})
    # a 10 us kernel whose counter window was 3x as wide as the kernel (what round 5 recorded for the sort / the filter chain)
    short_k = ("sort_coop_kernel", 10000, {"GRBM_GUI_ACTIVE": 8 * 2.3 * 30000, "SQ_INSTS_VALU": 2.0e5, "SQ_WAVES": 256, "SQ_WAVE_CYCLES": 1e7, "SQ_ACTIVE_INST_VALU": 1e6, "SQ_WAIT_ANY": 6e6, "SQ_WAIT_INST_ANY": 1e6})
    _write_pass(sq, [long_k] * 3 + [short_k] * 5)
    clock, how = pc.pass_clock_ghz(sq)
    assert abs(clock - 2.3) < 1e-9 and "3 launches" in how
    out = str(tmp_path / "pmc.json")
    pc.main(out, "long:cost_kernel:-:-:%s" % sq, "short:sort_coop_kernel:-:-:%s:1000" % sq)
    d = json.load(open(out))
    for name in ("long", "short"):
        e = d[name]["sq"]
        assert e["effective_clock_ghz"] <= pc.MAX_CLOCK_GHZ, e  # the rule of the verdict: no derived clock above 2.5 GHz
    assert abs(d["short"]["sq"]["grbm_clock_ghz"] - 6.9) < 1e-6          # the raw ratio stays visible ...
    want = 2.0e5 * 2.0 / (10000 * 2.3 * 1024)
    assert abs(d["short"]["sq"]["valu_issue_utilisation"] - round(want, 4)) < 1e-9  # ... and no longer divides the utilisation by three
    assert abs(d["long"]["sq"]["valu_issue_utilisation"] - round(4.0e6 * 2.0 / (120000 * 2.3 * 1024), 4)) < 1e-9
    # a pass without any long launch: the nominal clock, said so
    sq2 = str(tmp_path / "sq2.csv")
    _write_pass(sq2, [short_k] * 4)
    clock2, how2 = pc.pass_clock_ghz(sq2)
    assert clock2 == pc.NOMINAL_CLOCK_GHZ and "nominal" in how2
