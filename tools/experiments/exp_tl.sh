cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify --steps 20 --warmup 20 --repeats 3 > gpurun_out/bench_prof_t.json 2> gpurun_out/bench_prof_t.err || exit 1
python3 -c "import json; d=json.loads(open('gpurun_out/bench_prof_t.json').read()); print(d['value'], d['ms_per_step'], d['blocks_ms'])"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_t/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_sieve" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 3 blocks: 3 blocks x (3 workers x 3 kernels) = 27 launches; print the last block
last = rows[-9:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"]
    short = "emit" if "emit" in n else "check" if "check" in n else "sieve"
    print("%-6s q%-3s start %8.1f us  dur %7.1f us  grid %s" % (short, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
