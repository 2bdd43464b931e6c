#!/bin/bash
# Debug A/B: avg duration of the pre-LM kernels per build of the library.  usage: tools/ab_sort.sh variant... ("product" = fast_gicp_amd/lib)
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $R
for v in "$@"; do
  L=$R/fast_gicp_amd/lib/variants/$v/libfast_vgicp_hip.so
  [ "$v" = "product" ] && L=$R/fast_gicp_amd/lib/libfast_vgicp_hip.so
  rm -rf /tmp/prof_$v
  FVH_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o h -- python tools/sort_bench.py 200 $AB_ARGS > /tmp/prof_$v.log 2>&1 < /dev/null
  f=$(find /tmp/prof_$v -name "*.db" | head -1)
  echo "== $v"
  if [ -n "$f" ]; then timeout 30 python tools/rocpd_stats.py $f 2>/dev/null | grep -i "sort_coop\|knn_tiled\|pack_points\|cov_from" | python -c "
import sys
for l in sys.stdin:
    p = [x.strip() for x in l.split('|')]
    print('   %-40s calls %s avg_us %s' % (p[1][:40], p[2], p[4]))
"; else tail -5 /tmp/prof_$v.log; fi
done
