"""Per-kernel duration statistics of a rocprofv3 --kernel-trace database.  python tools/kt_stats.py db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tab = lambda p: [t for t in tabs if t.startswith(p)][0]
ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from '%s'" % tab("rocpd_info_kernel_symbol"))}
dur = defaultdict(list)
for kid, s, e in cur.execute("select kernel_id, start, end from '%s'" % tab("rocpd_kernel_dispatch")):
    dur[ks[kid]].append((e - s) / 1e3)
tot = sum(sum(v) for v in dur.values())
print("%-70s %6s %9s %9s %9s %6s" % ("kernel", "calls", "avg us", "p50 us", "max us", "%"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-70s %6d %9.2f %9.2f %9.2f %6.1f" % (k.split("(")[0][-70:], len(v), sum(v) / len(v), v[len(v) // 2], v[-1],
                                                 100 * sum(v) / tot))
