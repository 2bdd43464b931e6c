"""Debug: the pre-LM chain of the 17k headline in a loop (set_source_cloud -> Morton sort -> k-NN -> covariances), for
`rocprofv3 --kernel-trace --stats -- python tools/sort_bench.py` A/B runs of the sort / k-NN kernels (FVH_LIB_PATH picks the build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import capi, preprocess  # noqa: E402

tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
c = capi.VGICPCore(0)
dev = None
if "--device" in sys.argv:  # device-resident input (the headline's hand-over), else a host array per call (align.cpp's)
    import torch
    dev = torch.from_numpy(src).to("cuda:0").contiguous()
    torch.cuda.synchronize()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    if dev is not None:
        c.set_source_cloud_device(dev.data_ptr(), len(src), 3)
    else:
        c.set_source_cloud(src)
    c.find_source_neighbors(20)
    c.calculate_source_covariances()
c.synchronize()
