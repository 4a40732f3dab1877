# needs the measurement build: make -C zksnark_rs_amd/csrc measure; export ZKGPU_LIB=$PWD/zksnark_rs_amd/libzkgpu_measure.so
one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms')" "$@"; }
for n in 16 17 18; do
for u in 65536 131072 262144; do
one --latency --log-n $n --steps 40 --warmup 5 --opt msm_unchain_lanes=$u
one --no-cpu-baseline --log-n $n --steps 100 --opt msm_unchain_lanes=$u
done; done
