#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ZKGPU_LIB=$REPO/zksnark_rs_amd/libzkgpu_measure.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/arb -- python $REPO/bench.py --roots arbitrary --steps 6 --warmup 2 --serialize > $OUT/arb.log 2>&1
find $OUT/arb -name "*kernel_stats.csv" -exec cp {} $OUT/arb_kernel_stats.csv \;
rm -rf $OUT/arb
cd $REPO; python bench.py --roots arbitrary --steps 30 --warmup 4 2>/dev/null | tail -1 > $OUT/arb_bench.json
