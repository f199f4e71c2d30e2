"""Kernel timeline of a rocprofv3 --kernel-trace run: per queue, the kernels of a window in the
middle of the run with start/end relative to the window.  python tools/kt_timeline.py db [us_window]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tab = lambda p: [t for t in tabs if t.startswith(p)][0]
ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from '%s'" % tab("rocpd_info_kernel_symbol"))}
rows = list(cur.execute("select kernel_id, queue_id, stream_id, start, end from '%s' order by start" % tab("rocpd_kernel_dispatch")))
rows = [r for r in rows if "sieve" in ks[r[0]]]
window = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
mid = rows[len(rows) * 2 // 3][3]
sel = [r for r in rows if mid <= r[3] < mid + window * 1000]
short = lambda n: "K1" if "k_sieveI" in n else "K2" if "check" in n else "K3"
byq = defaultdict(list)
for kid, q, st, s, e in sel:
    byq[(q, st)].append((short(ks[kid]), (s - mid) / 1e3, (e - mid) / 1e3))
for key in sorted(byq):
    print("queue %s stream %s:" % key, " ".join("%s[%.1f-%.1f]" % x for x in byq[key]))
# durations
dur = defaultdict(list)
for kid, q, st, s, e in rows:
    dur[short(ks[kid])].append((e - s) / 1e3)
for k, v in sorted(dur.items()):
    v.sort()
    print(k, "n=%d p50 %.1f p90 %.1f max %.1f us" % (len(v), v[len(v) // 2], v[len(v) * 9 // 10], v[-1]))
