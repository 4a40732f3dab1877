set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
( timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r5i/pytest.txt
bash tools/ab.sh r5i_ab 4 60 -- "nounit" "default"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5i/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5i/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/valu_budget.py gpurun_out/r5i/pmc "r5i: second point peeled" > gpurun_out/r5i/valu_budget.txt
rm -rf gpurun_out/r5i/pmc
cat gpurun_out/r5i/pytest.txt gpurun_out/r5i_ab/ab.txt; head -8 gpurun_out/r5i/valu_budget.txt
