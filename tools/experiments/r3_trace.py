import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "k_lds"
def short(n):
    for k in ("k_lds_walk", "k_lds_scatter", "k_sieve_check", "k_sieve_emit", "k_sieve", "k_carry"):
        if k in n:
            return k
    return n[:30]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows if key in r["Kernel_Name"]]
ks.sort()
t0 = ks[0][0]
for s, e, n, q in ks[-75:]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  q{q} {n}")
# occupancy of the last 2 ms: how long 0, 1, 2, ... kernels of each kind were running
end = ks[-1][1]
lo = end - 1_800_000
for kind in sorted(set(k[2] for k in ks)):
    ev = []
    for s, e, n, q in ks:
        if n == kind and e > lo:
            ev.append((max(s, lo), 1))
            ev.append((e, -1))
    ev.sort()
    hist, cur, last = {}, 0, lo
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    hist[cur] = hist.get(cur, 0) + (end - last)
    print(kind, {k: round(v / 1e3, 1) for k, v in sorted(hist.items())}, "us with k kernels of the kind in flight (last 1.8 ms)")
