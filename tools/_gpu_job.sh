set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
( timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_bench.py tests/test_integer_roots.py tests/test_arbitrary_roots.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5j/pytest.txt
bash tools/ab.sh r5j_ab 4 60 -- "measure --opt alt_stream=0" "measure" "default" "default --depth 3"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5j/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5j/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_csv.py "$(find gpurun_out/r5j/trace -name '*kernel_trace.csv' | head -1)" 36 16 > gpurun_out/r5j/timeline.txt 2>&1
rm -rf gpurun_out/r5j/trace
cat gpurun_out/r5j/pytest.txt gpurun_out/r5j_ab/ab.txt; head -18 gpurun_out/r5j/timeline.txt
