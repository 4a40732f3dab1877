set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5a/pytest.txt
bash tools/ab.sh r5a_ab 3 60 -- "r4" "default"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5a/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5a/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/valu_budget.py gpurun_out/r5a/pmc "r5a: lds_inc test, static digits, lazy element-wise" > gpurun_out/r5a/valu_budget.txt
rm -rf gpurun_out/r5a/pmc
cat gpurun_out/r5a/pytest.txt gpurun_out/r5a_ab/ab.txt; head -40 gpurun_out/r5a/valu_budget.txt
