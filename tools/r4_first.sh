#!/bin/bash
# round 4, first GPU call: parity of the asm multipliers, then instruction budgets and a same-box A/B of the loop shapes
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_lazy29.py tests/test_gpu_blocks.py tests/test_golden.py -m gpu -x -q > gpurun_out/r4a/tests1.log 2>&1
echo "tests1 rc=$?" >> gpurun_out/r4a/tests1.log
timeout 1200 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "not largest" > gpurun_out/r4a/tests2.log 2>&1
echo "tests2 rc=$?" >> gpurun_out/r4a/tests2.log
bash tools/valu_variants.sh default noasm g1a g1b g1c g1d g2a g2b g2c
cp gpurun_out/b10/valu_*.txt gpurun_out/r4a/
bash tools/ab.sh r4a_ab 2 60 -- "default" "noasm" "g1a" "g1b" "g1c" "g1d" "g2a" "g2b" "g2c"
