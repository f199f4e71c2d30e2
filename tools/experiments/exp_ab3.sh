timeout -k 10 300 python3 -m pytest tests/test_gpu_sparse.py -x -q -p no:cacheprovider 2>&1 | tail -2
bash tools/experiments/exp_ab.sh
