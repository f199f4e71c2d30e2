import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ("walk" if "k_lds_walk" in r["Kernel_Name"] else "scatter"), r.get("Queue_Id", "?")) for r in rows if "k_lds" in r["Kernel_Name"]]
ks.sort()
t0 = ks[0][0]
last = ks[-60:]
for s, e, n, q in last:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  q{q} {n}")
