set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gap; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 200 rocprofv3 --kernel-trace -d $O/t -o h -- python tools/hiccup_probe.py 300 ${MODE:-pipe} > $O/out.txt 2>&1 < /dev/null
f=$(find $O/t -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start")) if 'stream_id' in cols else list(cur.execute("select name, start, end, queue_id from kernels order by start"))
lm = [i for i, r in enumerate(rows) if 'cost_kernel' in r[0]]
import collections
gaps = []
for a, b in zip(lm[100:-1], lm[101:]):
    ea, sb = rows[a][2], rows[b][1]
    between = [(r[0][:28], (r[1] - ea) / 1e3, (r[2] - ea) / 1e3, r[3]) for r in rows[a + 1:b]]
    gaps.append(((sb - ea) / 1e3, between))
gaps.sort(key=lambda g: g[0])
g = gaps[len(gaps) // 2]
print("median gap between two LM kernels: %.1f us" % g[0])
for b in g[1]: print("   %-28s start +%.1f end +%.1f  (stream %s)" % b)
# kernels overlapping the LM kernel: relative to the LM start
a = lm[150]
sa, ea = rows[a][1], rows[a][2]
print("LM kernel %.1f us; kernels starting within it:" % ((ea - sa) / 1e3))
for r in rows[a + 1:a + 14]:
    if r[1] < ea: print("   %-28s start +%.1f end +%.1f (stream %s)" % (r[0][:28], (r[1] - sa) / 1e3, (r[2] - sa) / 1e3, r[3]))
PY
rm -rf $O/t
