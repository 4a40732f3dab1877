set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5p/pytest.txt
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --steps 100 --warmup 5 > gpurun_out/bench_final_100.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final_20.json 2>> gpurun_out/bench_final.err
cat gpurun_out/r5p/pytest.txt; tail -3 gpurun_out/profile_round.log
