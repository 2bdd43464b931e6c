#!/bin/bash
# Kernel trace of the pipelined VGICP loop against the sequential one (tools/hiccup_probe.py): per-kernel averages of both, and how much of the
# pipelined loop's wall time has two kernels of the handle's two streams in flight at once. Output: gpurun_out/overlap/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/overlap; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for m in ${MODES:-seq pipe}; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/t_$m -o h -- python tools/hiccup_probe.py 600 $m > $O/$m.out 2> $O/$m.err < /dev/null
  f=$(find $O/t_$m -name "*.db" | head -1)
  if [ -n "$f" ]; then
    echo "## $m" >> $O/summary.md; cat $O/$m.out >> $O/summary.md
    python tools/rocpd_stats.py $f >> $O/summary.md 2>/dev/null
    python - "$f" >> $O/summary.md <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t]
name = "kernels" if "kernels" in tabs else (kd[0] if kd else None)
rows = []
try:
    rows = list(cur.execute("select start, end from kernels"))
except Exception:
    try:
        rows = list(cur.execute("select start, end from %s" % kd[0]))
    except Exception as ex:
        print("(no dispatch intervals: %r; tables: %s)" % (ex, tabs[:12]))
if rows:
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = sorted([(s, 1) for s, e in rows] + [(e, -1) for s, e in rows])
    depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    print("wall %.1f ms, some kernel in flight %.1f %%, two or more in flight %.1f %% of the wall time" % ((t1 - t0) / 1e6, 100.0 * busy1 / (t1 - t0), 100.0 * busy2 / (t1 - t0)))
PY
  fi
  rm -rf $O/t_$m
done
cat $O/summary.md
