#!/bin/bash
# per-kernel times of the ApproximateVoxelGrid chain on the LiDAR stream: library variants (points per wave of the counting sort) x fused / six launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for V in ${VARIANTS:-default avg256}; do
  [ $V = default ] && unset FVH_LIB_PATH || export FVH_LIB_PATH=$R/fast_gicp_amd/lib/variants/$V/libfast_vgicp_hip.so
  timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_$V -o h -- python bench.py --workload lidar_stream --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$V.json 2> $O/prof_$V.log < /dev/null
  f=$(find $O/prof_$V -name "*.db" | head -1)
  [ -n "$f" ] && timeout 30 python tools/rocpd_stats.py $f > $O/stats_$V.md 2>/dev/null < /dev/null
  rm -rf $O/prof_$V
  echo "== $V"; grep -E "avg_" $O/stats_$V.md | awk -F'|' '{print substr($2,1,40), $3, $4, $5}'
  python -c "
import json; d=json.loads([l for l in open('$O/bench_$V.json') if l.startswith('{')][-1]); print('value', d['value'], 'downsample', d['roofline_downsample']['avg_launch_us'])"
done
