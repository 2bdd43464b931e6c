#!/bin/bash
# Round-6 artifacts on the GPU box (everything lands in gpurun_out/r06/; copy what is to be judged into profiles/):
#   pmc    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ group: one counter group per pass, --kernel-trace only) of the five
#          bench workloads -> gpurun_out/r06/pmc.json with the LM kernel of each (VGICP and NDT instantiations), the exact k-NN
#          kernel and the RBF sweep
#   stats  rocprofv3 --kernel-trace --stats summaries of the same four commands
#   bench  the default bench line
# Every command has its own timeout and no stdin.  Usage: FVH_COMMIT=<sha> tools/r06_artifacts.sh [pmc|stats|bench|all]
# Before sending it to the GPU box: python tools/build_variants.py knn_timing="-DFVH_KNN_TIMING"  (the pair counters of the pmc leg; ~2 GPU-minutes for `all`)
set -u
WHAT=${1:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
COMMON="--no-cpu-baseline --no-profile --streams 1 --configs none --no-host-leg --no-pipelined-leg"
declare -A WL
WL[bundled17k]="--steps 40 --warmup 5 $COMMON"
WL[synth100k_rbf]="--workload synth100k --cov rbf --steps 25 --warmup 3 $COMMON"
WL[synth1m]="--workload synth1m --steps 25 --warmup 3 $COMMON"
WL[lidar_stream]="--workload lidar_stream --steps 40 --warmup 5 --no-cpu-baseline --no-profile"
WL[fgicp17k]="--workload fgicp17k --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-pipelined-leg"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  SPECS=""
  for w in bundled17k synth100k_rbf synth1m lidar_stream fgicp17k; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python bench.py ${WL[$w]} > $O/pmc_${w}_$c.log 2>&1 < /dev/null
    done
    timeout 150 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_${w}_SQ -o p -- python bench.py ${WL[$w]} > $O/pmc_${w}_SQ.log 2>&1 < /dev/null
    F=$(find $O/pmc_${w}_FETCH_SIZE -name "*counter_collection.csv" | head -1); F=${F:--}
    W=$(find $O/pmc_${w}_WRITE_SIZE -name "*counter_collection.csv" | head -1); W=${W:--}
    S=$(find $O/pmc_${w}_SQ -name "*counter_collection.csv" | head -1); S=${S:--}
    SPECS="$SPECS ${w}_cost:cost_kernel:$F:$W:$S"
    case $w in
      bundled17k) SPECS="$SPECS ${w}_knn:knn_tiled1_kernel:$F:$W:$S ${w}_sort:sort_coop_kernel:$F:$W:$S ${w}_cov:cov_from_neighbors_kernel:$F:$W:$S:3000" ;;
      fgicp17k) SPECS="$SPECS ${w}_nn1:nn1_rows_kernel:$F:$W:$S:5000" ;;
      synth1m) SPECS="$SPECS ${w}_knn:knn_tiled1_kernel:$F:$W:$S" ;;
      synth100k_rbf) SPECS="$SPECS ${w}_rbf:cov_rbf1_kernel:$F:$W:$S ${w}_radix_hist:radix_hist_fused_kernel:$F:$W:$S:3000 ${w}_radix_scatter:radix_scatter_fused_kernel:$F:$W:$S:3000" ;;
      lidar_stream) SPECS="$SPECS ${w}_downsample_emit:avg_emit_kernel:$F:$W:$S:1000 ${w}_downsample_hist:avg_keys_hist_kernel:$F:$W:$S:1000 ${w}_downsample_scatter:avg_scatter_kernel:$F:$W:$S:1000 ${w}_downsample_mark:avg_mark_kernel:$F:$W:$S:1000 ${w}_vm_accumulate:vm_accumulate_kernel:$F:$W:$S:1000 ${w}_vm_finalize:vm_finalize_kernel:$F:$W:$S:1000" ;;
    esac
  done
  # what the culled k-NN / RBF searches evaluate (debug build of the library: counters in the kernels) -> flop fractions
  FVH_LIB_PATH=fast_gicp_amd/lib/variants/knn_timing/libfast_vgicp_hip.so timeout 120 python tools/pair_counts.py $O/pairs.json > $O/pairs.out 2>&1 < /dev/null
  timeout 60 python tools/pmc_collect.py $O/pmc.json --pairs $O/pairs.json $SPECS > $O/pmc_collect.out 2> $O/pmc_collect.err < /dev/null
  rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/pmc_*_SQ
  tail -5 $O/pmc_collect.err; python -c "
import json; d = json.load(open('$O/pmc.json'))
for k, v in d.items():
    if k != '_meta': print(k, v.get('hbm_bytes_per_launch'), (v.get('sq') or {}).get('valu_busy_frac_of_wave_cycles'), (v.get('sq') or {}).get('valu_issue_utilisation'), (v.get('sq') or {}).get('avg_launch_us_in_this_pass'))
"
fi
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
  for w in bundled17k synth100k_rbf synth1m lidar_stream fgicp17k; do
    ARGS=$(echo "${WL[$w]}" | sed 's/--no-profile//')
    timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o h -- python bench.py $ARGS > $O/${w}_under_rocprof.json 2> $O/prof_$w.log < /dev/null
    f=$(find $O/prof_$w -name "*.db" | head -1)
    [ -n "$f" ] && timeout 30 python tools/rocpd_stats.py $f > $O/prof_${w}_kernel_stats.md 2>/dev/null < /dev/null
    rm -rf $O/prof_$w
  done
  head -14 $O/prof_bundled17k_kernel_stats.md
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  # (bench.py quotes profiles/r06_pmc.json when its stamp matches the kernels: in an "all" run that is the file made above)
  [ "$WHAT" = "all" ] && [ -s $O/pmc.json ] && cp $O/pmc.json profiles/r06_pmc.json
  timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
  echo "bench rc=$?"; cp bench_detail.json $O/bench_detail.json; wc -c $O/bench_line.json; cat $O/bench_line.json
fi
