set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5k/pytest.txt
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
ZKGPU_LIB=$GRAFT_REPO_ROOT/zksnark_rs_amd/libzkgpu_measure.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/arb_ser -- python $GRAFT_REPO_ROOT/bench.py --roots arbitrary --steps 12 --warmup 3 --no-cpu-baseline --serialize > $OUT/arb_ser.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/arb_pmc -- python $GRAFT_REPO_ROOT/bench.py --roots arbitrary --steps 3 --warmup 1 --no-cpu-baseline > $OUT/arb_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/arb_ser -name "*kernel_stats.csv" -exec cp {} $OUT/arbitrary_roots_kernel_stats_serialized.csv \;
python tools/valu_budget.py $OUT/arb_pmc "round-5 build, arbitrary roots, 2^20 gates" > $OUT/arbitrary_roots_valu_budget.txt
rm -rf $OUT/arb_ser $OUT/arb_pmc
cat gpurun_out/r5k/pytest.txt; tail -3 gpurun_out/profile_round.log; head -12 $OUT/arbitrary_roots_valu_budget.txt
