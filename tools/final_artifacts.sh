#!/bin/bash
# Round-end artifacts on the GPU box: PMC traffic of the persistent LM kernel, the default bench line, the rocprofv3 kernel
# summary of the same workload, and the bench lines of the other configs. Everything lands in gpurun_out/final/.
# Every command has its own timeout and no stdin.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile --streams 1 > $O/pmc_$c.log 2>&1 < /dev/null
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  timeout 30 python tools/pmc_traffic.py $O/pmc_new.json bundled17k_persistent:cost_kernel:$F:$W > /dev/null 2>$O/pmc_traffic.err < /dev/null
  timeout 30 python - <<PY
import json
p = "$R/profiles/r01_pmc_cost_kernel.json"
d = json.load(open(p)); n = json.load(open("$O/pmc_new.json"))
e = n["bundled17k_persistent"]
if e.get("hbm_bytes_per_launch"):
    e["note"] = "one launch = the whole LM loop of a registration; run: bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile --streams 1 (end-of-round build, voxel keys in their own array)"
    d["bundled17k_persistent"] = e
    json.dump(d, open(p, "w"), indent=1)
    json.dump(d, open("$O/r01_pmc_cost_kernel.json", "w"), indent=1)
PY
fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
timeout 150 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 60 python bench.py --no-profile --no-cpu-baseline --streams 1 > $O/bench_no_profile.json 2>/dev/null < /dev/null
timeout 90 rocprofv3 --kernel-trace --stats -d $O/prof -o h -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --streams 1 > $O/bench_under_rocprof.json 2> $O/prof.log < /dev/null
timeout 60 python bench.py --workload synth100k --no-cpu-baseline --streams 1 > $O/synth100k.json 2>/dev/null < /dev/null
timeout 60 python bench.py --workload synth100k --cov rbf --no-cpu-baseline --streams 1 > $O/synth100k_rbf.json 2>/dev/null < /dev/null
timeout 60 python bench.py --workload synth1m --no-cpu-baseline --streams 1 > $O/synth1m.json 2>/dev/null < /dev/null
timeout 60 python bench.py --cov rbf --no-cpu-baseline --streams 1 > $O/bundled_rbf.json 2>/dev/null < /dev/null
timeout 90 python bench.py --workload lidar_stream > $O/stream.json 2>/dev/null < /dev/null
ls -la $O | head -30
for f in bench bench_no_profile bench_under_rocprof synth100k synth100k_rbf synth1m bundled_rbf stream; do python -c "
import json
try:
    d = json.load(open('$O/$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic'))
except Exception as ex: print('$f', 'ERR', ex)
" < /dev/null; done
# the N > 1 launch path of bench.py (two ranks sharing the one GPU of this box; the spatially sharded leg needs one GPU per rank and is skipped)
FVH_BENCH_SHARE_GPU=1 FVH_BENCH_BACKEND=gloo timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/share2.json 2> $O/share2.err < /dev/null
echo "share2 rc=$?"; cut -c1-200 $O/share2.json
