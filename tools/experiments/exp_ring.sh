cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python3 bench.py --workload sentiment --texts 4 --no-cpu-baseline --no-e2e --sub= 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['parity'][:40])" || { tail -3 gpurun_out/bg.err; }
done
for i in 1 2; do
rm -rf gpurun_out/prof_x
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python3 bench.py --workload sentiment --texts 4 --no-cpu-baseline --no-e2e --no-extra --sub= 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('profiled', d['value'], d['parity'][:40])" || { grep -i mismatch gpurun_out/bg.err | head -3; }
done
python3 bench.py --workload sentiment --texts 4 --no-cpu-baseline --no-e2e --sub= --steps 20 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps 20', d['value'], d['parity'][:40])"
python3 bench.py --workload sentiment --texts 4 --no-cpu-baseline --no-e2e --sub= --issue threads --workers 4 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads', d['value'], d['parity'][:40])"
