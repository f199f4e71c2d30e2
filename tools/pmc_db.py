"""Summarise a rocprofv3 rocpd sqlite database: per kernel, average duration and the sum of
each PMC counter per dispatch.  python tools/pmc_db.py <results.db> [kernel substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from '%s'" % tab("rocpd_info_kernel_symbol"))}
pm = {r[0]: r[1] for r in cur.execute("select id, name from '%s'" % tab("rocpd_info_pmc"))}
disp = list(cur.execute("select id, kernel_id, start, end, event_id from '%s'" % tab("rocpd_kernel_dispatch")))
ev = defaultdict(lambda: defaultdict(float))
for event_id, pmc_id, value in cur.execute("select event_id, pmc_id, value from '%s'" % tab("rocpd_pmc_event")):
    ev[event_id][pm[pmc_id]] += value
agg = defaultdict(lambda: {"n": 0, "dur": 0.0, "c": defaultdict(float)})
for _id, kid, st, en, event_id in disp:
    name = ks[kid]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    a = agg[name]
    a["n"] += 1
    a["dur"] += (en - st) / 1e3
    for k, v in ev.get(event_id, {}).items():
        a["c"][k] += v
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    short = name.split("(")[0][-60:]
    print("%-60s n=%4d avg %.2f us" % (short, a["n"], a["dur"] / a["n"]))
    for k, v in sorted(a["c"].items()):
        print("      %-24s %16.1f per dispatch" % (k, v / a["n"]))
