#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4i
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_arbitrary_roots.py tests/test_integer_roots.py tests/test_cpp_api.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_bench.py -m gpu -x -q > $OUT/tests2.log 2>&1; echo "rc=$?" >> $OUT/tests2.log
