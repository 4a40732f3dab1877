set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
( timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_golden.py tests/test_gpu_blocks.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r5o/pytest.txt
bash tools/ab.sh r5o_ab 5 60 -- "measure --opt ntt_fuse=0" "measure" "measure --opt tail_stream=1" "measure --opt tail_stream=1 --depth 3"
cat gpurun_out/r5o/pytest.txt gpurun_out/r5o_ab/ab.txt gpurun_out/r5o_ab/ab_raw.txt
