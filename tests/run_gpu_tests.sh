#!/bin/bash
# helper for gpurun: GPU test-suite with a progress log under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 ${1:-900} python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
exit ${PIPESTATUS[0]}
