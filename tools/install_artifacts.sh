#!/bin/bash
# Copies what tools/r06_artifacts.sh left under gpurun_out/r06/ into profiles/ (tracked) and tests/golden/, and stamps profiles/README.md with the commit.
# Usage: tools/install_artifacts.sh <commit the artifacts were made at>
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r06; C=${1:?commit}
cp $O/pmc.json $R/profiles/r06_pmc.json
cp $O/pairs.json $R/profiles/r06_pair_counts.json
cp $O/bench_line.json $R/profiles/r06_bench.json
cp $O/bench_detail.json $R/profiles/r06_bench_detail.json
cp $O/bench_detail.json $R/tests/golden/bench_detail_n1.json
for w in bundled17k synth100k_rbf synth1m lidar_stream fgicp17k; do
  cp $O/prof_${w}_kernel_stats.md $R/profiles/r06_prof_${w}_kernel_stats.md
  cp $O/${w}_under_rocprof.json $R/profiles/r06_${w}_under_rocprof.json
done
sed -i -E "s/at commit \`[0-9a-f]{7}\`/at commit \`${C:0:7}\`/; s/The refresh at \`[0-9a-f]{7}\`/The refresh at \`${C:0:7}\`/" $R/profiles/README.md
python - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
m = json.load(open("$R/profiles/r06_pmc.json"))["_meta"]
print("tree csrc", bench.csrc_sha(), "pmc stamp", m["csrc_sha"], m["commit"])
assert bench.csrc_sha() == m["csrc_sha"], "the kernels changed since the artifacts were made"
PY
