#!/bin/bash
# same-box A/B of two builds (default against libzkgpu_r3.so, built from an older commit) with issue / wait counters of the accumulations and
# stand-alone kernel times: what round 4 used to see that -9 % instructions gave +4 % rate at an unchanged clock (profiles/r4_pmc_counters_*.txt)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4b
mkdir -p $OUT
bash tools/ab.sh r4b_ab 3 60 -- "default" "r3" "g1a"
cp gpurun_out/r4b_ab/ab.txt $OUT/
python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
ZKGPU_LIB=$REPO/zksnark_rs_amd/libzkgpu_r3.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_r3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
CMD3="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for v in default r3; do
  lib=$REPO/zksnark_rs_amd/libzkgpu.so; [ $v = r3 ] && lib=$REPO/zksnark_rs_amd/libzkgpu_r3.so
  ZKGPU_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$v -- $CMD3 > $OUT/pmc_$v.log 2>&1
  python $REPO/tools/pmc_counters.py $OUT/pmc_$v k_msm_accumulate k_ntt_tile k_msm_fold > $OUT/pmc_counters_$v.txt
  ZKGPU_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $OUT/pmc2_$v -- $CMD3 > $OUT/pmc2_$v.log 2>&1
  python $REPO/tools/pmc_counters.py $OUT/pmc2_$v k_msm_accumulate > $OUT/pmc_counters2_$v.txt
  rm -rf $OUT/pmc_$v $OUT/pmc2_$v
  m=$REPO/zksnark_rs_amd/libzkgpu_measure.so; [ $v = r3 ] && m=$REPO/zksnark_rs_amd/libzkgpu_measure_r3.so
  ZKGPU_LIB=$m rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ser_$v -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --serialize > $OUT/ser_$v.log 2>&1
  find $OUT/ser_$v -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serialized_$v.csv \;
  rm -rf $OUT/ser_$v
done
ls $OUT
