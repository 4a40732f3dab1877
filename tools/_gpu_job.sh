set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5t/pytest.txt
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --steps 100 --warmup 5 > gpurun_out/bench_final_100.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_final_20.json 2>> gpurun_out/bench_final.err
bash tools/ab.sh r5t_ab 3 60 -- "r4" "default"
cat gpurun_out/r5t/pytest.txt gpurun_out/r5t_ab/ab.txt; tail -2 gpurun_out/profile_round.log
