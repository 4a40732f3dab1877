set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
for lib in default r4 default r4 default r4; do
  if [ $lib = default ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$GRAFT_REPO_ROOT/zksnark_rs_amd/libzkgpu_$lib.so; fi
  ZK_BENCH_DUMP_INTERVALS=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> gpurun_out/r5s/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['steady_state_ms_per_proof'])"
  grep first_completion gpurun_out/r5s/err.txt | cut -c1-400
done > gpurun_out/r5s/runs.txt 2>&1
cat gpurun_out/r5s/runs.txt
