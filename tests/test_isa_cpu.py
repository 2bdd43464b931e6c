"""The compiled device code keeps the loads that are meant to be in flight together in flight together (CPU: hipcc cross-compiles gfx950 to
assembly; tools/scan_serial_loads.py counts, per kernel, the memory operations behind their own full wait and the FLAT loads).

Round 5 found three kernels whose source said "N independent loads" and whose ISA said "N dependent round trips": the LM kernel's pose reads
were FLAT loads (generic pointer into LDS: their wait drains every load in flight), its per-slot "cached ? LDS : global" selects ended at
joins with a full wait each, the covariance kernel's `if (j < k) p = pts[nbr[j]]` was ten round trips, and the cooperative sort read its
histogram matrix as 22 batched + 10 serialised loads. These assertions pin the repaired shapes against a compiler or source change that
silently brings the serialisation back (the numbers move with the toolchain: loosen them only after looking at the ISA)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    import scan_serial_loads
    from fast_gicp_amd import build
    out = str(tmp_path_factory.mktemp("isa") / "fvh.s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["--cuda-device-only", "-S", "-o", out, build.SOURCES[0]], stderr=subprocess.DEVNULL)
    return scan_serial_loads.scan(open(out).read())


def _one(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return kernels[hits[0]]


def test_persistent_lm_kernels_read_lds_with_ds_read_not_flat(kernels):
    import re
    names = [k for k in kernels if re.search(r"cost_kernelI[fd]Li[012]ELb1ELi[14]E", k)]  # PERSIST = true
    assert len(names) == 24  # 2 precisions x 3 modes x CH 1/4 x LM/GN
    for k in names:  # (one FLAT load remains in the VGICP instantiations' prologue: the peer view; none in the item loop)
        assert kernels[k]["flat"] <= 1, (k, kernels[k])


def test_lm_kernel_first_round_trip_is_one_batch(kernels):
    # four-lookup item of a cloud too large for sticky items: point, covariance (2), four stored ids, (offsets from LDS) -- and the twelve
    # record loads of round trip 2 -- are batches, not a chain of single loads
    v = _one(kernels, "cost_kernelIdLi0ELb1ELi4ELb0EE")
    assert v["max_batch"] >= 12, v
    assert v["one_op_waits"] <= 24, v  # (round 4: 40)


def test_covariance_gather_is_two_round_trips(kernels):
    v = _one(kernels, "cov_from_neighbors_kernelILi5E")
    assert v["max_batch"] >= 10 and v["full_waits"] <= 3, v  # (round 4: 11 loads, 11 full waits)


def test_cooperative_sort_reads_its_matrix_in_one_batch(kernels):
    v = _one(kernels, "sort_coop_kernelE")
    assert v["max_batch"] >= 8, v
