"""Print a window of a rocprofv3 kernel trace as a per-queue timeline (start, duration, gap).
usage: python tests/trace_timeline.py <kernel_trace.csv> [skip_us] [window_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else -400.0
win = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
t_end = int(rows[-1]["End_Timestamp"])
t0 = int(rows[0]["Start_Timestamp"])
base = (t_end + skip * 1e3) if skip < 0 else (t0 + skip * 1e3)
short = lambda n: n.split("::")[-1].split("(")[0].split("<")[0][:18]
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < base or s > base + win * 1e3:
        continue
    print("q%-2s %9.2f +%7.2f  %s" % (r["Queue_Id"], (s - base) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])))
