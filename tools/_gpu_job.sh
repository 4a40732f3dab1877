set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5b/pytest.txt
bash tools/ab.sh r5b_ab 3 60 -- "measure --opt merge_lh=0" "measure" "default"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5b/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5b/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/valu_budget.py gpurun_out/r5b/pmc "r5b: L and H merged" > gpurun_out/r5b/valu_budget.txt
rm -rf gpurun_out/r5b/pmc
python bench.py --steps 20 --warmup 5 > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
cat gpurun_out/r5b/pytest.txt gpurun_out/r5b_ab/ab.txt; head -30 gpurun_out/r5b/valu_budget.txt; tail -c 1500 gpurun_out/r5b/bench.json
