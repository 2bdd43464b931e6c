#!/bin/bash
# PMC traffic of the LM kernel on the 100k workload + rocprofv3 summary of the lidar_stream workload -> gpurun_out/extra/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/extra
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 40 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --workload synth100k --steps 20 --warmup 3 --no-cpu-baseline --no-profile --streams 1 > $O/pmc_$c.log 2>&1 < /dev/null
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  timeout 20 python tools/pmc_traffic.py $O/pmc_synth100k.json synth100k_persistent:cost_kernel:$F:$W > /dev/null 2>$O/pmc_traffic.err < /dev/null
fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
timeout 45 rocprofv3 --kernel-trace --stats -d $O/prof_s -o s -- python bench.py --workload lidar_stream --steps 60 --warmup 5 --no-cpu-baseline --no-profile > $O/stream_under_rocprof.json 2> $O/prof_s.log < /dev/null
ls $O
cat $O/pmc_synth100k.json 2>/dev/null | head -12
