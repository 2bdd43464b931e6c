#!/bin/bash
# Round-2 artifacts on the GPU box (everything lands in gpurun_out/r02/; copy what is to be judged into profiles/):
#   1. PMC passes of the LM kernel on the three VGICP configurations: FETCH_SIZE and WRITE_SIZE in separate runs (they do not
#      fit one pass), plus one SQ pass (VALU busy / waiting) -- rocprofv3 with --kernel-trace only, as gpurun requires
#   2. rocprofv3 --kernel-trace --stats summaries of the default bench command, the 100k RBF config and the LiDAR stream
#   3. the default bench line
# Every command has its own timeout and no stdin.  Usage: tools/r02_artifacts.sh [pmc|stats|bench|all]
set -u
WHAT=${1:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
COMMON="--no-cpu-baseline --no-profile --streams 1 --configs none --no-host-leg"
declare -A WL
WL[bundled17k]="--steps 40 --warmup 5"
WL[synth100k_rbf]="--workload synth100k --cov rbf --steps 25 --warmup 3"
WL[synth1m]="--workload synth1m --steps 25 --warmup 3"
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  SPECS=""
  for w in bundled17k synth100k_rbf synth1m; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python bench.py ${WL[$w]} $COMMON > $O/pmc_${w}_$c.log 2>&1 < /dev/null
    done
    timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_${w}_SQ -o p -- python bench.py ${WL[$w]} $COMMON > $O/pmc_${w}_SQ.log 2>&1 < /dev/null
    F=$(find $O/pmc_${w}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
    W=$(find $O/pmc_${w}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    if [ -n "$F" ] && [ -n "$W" ]; then SPECS="$SPECS ${w}_persistent:cost_kernel:$F:$W"; fi
  done
  timeout 60 python tools/pmc_traffic.py $O/pmc_cost_kernel.json $SPECS > $O/pmc_traffic.out 2> $O/pmc_traffic.err < /dev/null
  timeout 60 python tools/pmc_sq.py $O/pmc_sq.json $(for w in bundled17k synth100k_rbf synth1m; do f=$(find $O/pmc_${w}_SQ -name "*counter_collection.csv" | head -1); [ -n "$f" ] && echo "$w:cost_kernel:$f"; done) > $O/pmc_sq.out 2> $O/pmc_sq.err < /dev/null
  rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/pmc_*_SQ
  cat $O/pmc_cost_kernel.json | head -40; cat $O/pmc_sq.out | tail -30
fi
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof17k -o h -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 --configs none > $O/bench_under_rocprof.json 2> $O/prof17k.log < /dev/null
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof100k -o h -- python bench.py --workload synth100k --cov rbf --steps 40 --warmup 5 --no-cpu-baseline --streams 1 --configs none > $O/synth100k_rbf_under_rocprof.json 2> $O/prof100k.log < /dev/null
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof1m -o h -- python bench.py --workload synth1m --steps 40 --warmup 5 --no-cpu-baseline --streams 1 --configs none > $O/synth1m_under_rocprof.json 2> $O/prof1m.log < /dev/null
  timeout 150 rocprofv3 --kernel-trace --stats -d $O/profstream -o h -- python bench.py --workload lidar_stream --steps 60 --warmup 5 --no-cpu-baseline --no-profile > $O/stream_under_rocprof.json 2> $O/profstream.log < /dev/null
  for d in prof17k prof100k prof1m profstream; do
    f=$(find $O/$d -name "*.db" | head -1)
    [ -n "$f" ] && timeout 30 python tools/rocpd_stats.py $f > $O/${d}_kernel_stats.md 2>/dev/null < /dev/null
    rm -rf $O/$d
  done
  head -12 $O/prof17k_kernel_stats.md
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  timeout 400 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
  echo "bench rc=$?"; tail -3 $O/bench.err
  python - <<PY < /dev/null
import json
d = json.load(open("$O/bench.json"))
print("headline", d["value"], d["ms_per_step"], "host", (d.get("host_clouds_in") or {}).get("value"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"], "wall", d.get("bench_wall_s"))
for k, v in (d.get("configs") or {}).items():
    if "value" in v:
        print(k, v["value"], v["ms_per_step"], "host", (v.get("host_clouds_in") or {}).get("value"), "roofline", (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("avg_launch_us"), "cpu", (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("cores"))
    else:
        print(k, v)
PY
fi
