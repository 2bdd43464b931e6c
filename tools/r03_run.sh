set -x
O=gpurun_out/r03p
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
tail -12 $O/pytest.txt
