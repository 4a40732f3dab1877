#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mg
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_prove.py -x -q -m gpu -k "window_sharded_full or mgpu_pipeline or multi_gpu_partials or scalar_exchange" > $OUT/pytest.txt 2>&1
python -m pytest tests/test_cpp_api.py tests/test_integer_roots.py tests/test_gpu_bench.py -x -q -m gpu > $OUT/pytest2.txt 2>&1
ZK_COMM_FORCE_RCCL=1 python tools/rccl_starvation.py > $OUT/rccl_starvation.txt 2>&1
{
for W in 2 4 8; do python bench.py --emulate-world $W --steps 40 --warmup 6; done
for W in 2 8; do python bench.py --emulate-world $W --steps 40 --warmup 6 --transport zk-gloo; done
python bench.py --no-cpu-baseline --steps 60 --warmup 5 | cut -c1-160
} > $OUT/emulate.txt 2>&1
