#!/bin/bash
# socket power and shader clock (rocm-smi, every 0.25 s) while bench.py runs pipelined and serialized; outputs in gpurun_out/power/
# the serialized leg needs the measurement build (make -C zksnark_rs_amd/csrc measure)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/power
mkdir -p $OUT
cd $REPO
sample() { while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" ; sleep 0.25; done; }
for mode in pipe ser; do
  EXTRA=""; [ $mode = ser ] && EXTRA="--serialize" && export ZKGPU_LIB=$REPO/zksnark_rs_amd/libzkgpu_measure.so
  sample > $OUT/smi_$mode.txt &
  SP=$!
  python bench.py --no-cpu-baseline --steps 800 --warmup 5 $EXTRA > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  kill $SP
done
