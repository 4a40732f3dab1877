#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4d
mkdir -p $OUT
bash tools/ab.sh r4d_ab 3 60 -- "default" "asm1" "g2old" "g2late" "g2u2"
cp gpurun_out/r4d_ab/ab.txt $OUT/
python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
