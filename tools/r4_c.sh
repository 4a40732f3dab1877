#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_lazy29.py tests/test_gpu_blocks.py -m gpu -x -q > $OUT/tests1.log 2>&1; echo "rc=$?" >> $OUT/tests1.log
timeout 900 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "not largest" > $OUT/tests2.log 2>&1; echo "rc=$?" >> $OUT/tests2.log
bash tools/valu_variants.sh default g1b g1c g1w4 g2b g2c
cp gpurun_out/b10/valu_*.txt $OUT/
bash tools/ab.sh r4c_ab 3 60 -- "default" "asm1" "r3" "g1b" "g1c" "g1w4" "g2b" "g2c"
cp gpurun_out/r4c_ab/ab.txt $OUT/
