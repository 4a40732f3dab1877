set -u
cd $GRAFT_REPO_ROOT
bash tools/soak.sh > /dev/null 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --steps 100 --warmup 5 > gpurun_out/bench_final_100.json 2>> gpurun_out/bench_final.err
cat gpurun_out/soak.txt; tail -c 400 gpurun_out/bench_final.json
