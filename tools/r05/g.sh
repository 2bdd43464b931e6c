timeout 600 python -m pytest tests/test_gpu_ndt.py tests/test_gpu_peer.py tests/test_gpu_spatial_order.py -x -q 2>&1 | tail -15
