#!/bin/bash
# round 4, GPU call 13: the two-launch radix passes after tuning the scatter (rows in flight, LDS layout, gather preload)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spatial_order.py tests/test_gpu_parity_large.py -m gpu -q -x 2>&1 | tail -3 > $O/tests2.txt
cat $O/tests2.txt
timeout 500 python tools/ab_bench.py --workload synth100k --cov rbf --steps 40 default batch64 default batch64 > $O/ab_100k_2.txt 2>&1
timeout 500 python tools/ab_bench.py --workload synth1m --steps 40 default batch64 default batch64 > $O/ab_1m_2.txt 2>&1
cat $O/ab_100k_2.txt $O/ab_1m_2.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -o h -- python bench.py --workload synth1m --steps 25 --warmup 3 --no-cpu-baseline --streams 1 --configs none --no-host-leg > /dev/null 2> $O/prof.log < /dev/null
f=$(find $O/prof -name "*.db" | head -1); [ -n "$f" ] && timeout 30 python tools/rocpd_stats.py $f | grep -E "radix|tile_super|pack" | cut -c1-60,118-160
rm -rf $O/prof
