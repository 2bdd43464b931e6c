timeout 600 python -m pytest "tests/test_gpu_parity.py::test_gauss_newton_on_the_device_matches_the_oracle_on_both_routes" -x -q 2>&1 | grep -v "^$" | tail -30 | cut -c1-250
