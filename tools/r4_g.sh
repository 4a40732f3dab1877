#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4g
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $OUT/tests_bench.log 2>&1; echo "rc=$?" >> $OUT/tests_bench.log
timeout 1500 python -m pytest tests/test_arbitrary_roots.py -m gpu -x -q > $OUT/tests_arb.log 2>&1; echo "rc=$?" >> $OUT/tests_arb.log
timeout 900 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "mgpu or exchange or one_rank or sharded" > $OUT/tests_mgpu.log 2>&1; echo "rc=$?" >> $OUT/tests_mgpu.log
