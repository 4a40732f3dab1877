#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/b10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # label lib extra...
  label=$1; lib=$2; shift 2
  ZKGPU_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p_$label -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/p_$label.log 2>&1
  python $REPO/tools/valu_budget.py $OUT/p_$label $label > $OUT/valu_$label.txt 2>&1
  rm -rf $OUT/p_$label
}
for v in "$@"; do if [ $v = default ]; then run default $REPO/zksnark_rs_amd/libzkgpu.so; else run $v $REPO/zksnark_rs_amd/libzkgpu_$v.so; fi; done
