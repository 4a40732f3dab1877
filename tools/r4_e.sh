#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4e
mkdir -p $OUT
bash tools/ab.sh r4e_ab 3 60 -- "asm1" "w4a" "w4b" "w4c" "w4d" "w3u1"
cp gpurun_out/r4e_ab/ab.txt $OUT/
bash tools/valu_variants.sh w4a w4c
cp gpurun_out/b10/valu_w4*.txt $OUT/
for v in w4a w4c; do ZKGPU_LIB=$REPO/zksnark_rs_amd/libzkgpu_$v.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_$v.json 2>/dev/null; done
