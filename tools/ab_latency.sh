# needs the measurement build: make -C zksnark_rs_amd/csrc measure; export ZKGPU_LIB=$PWD/zksnark_rs_amd/libzkgpu_measure.so
one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
for n in 4 12 16 20; do one --latency --log-n $n --steps 30 --warmup 5; one --latency --log-n $n --steps 30 --warmup 5 --opt msm_small_lanes=0; done
one --no-cpu-baseline --steps 40
one --no-cpu-baseline --steps 40 --opt msm_small_lanes=0
one --no-cpu-baseline --log-n 16 --steps 200
one --no-cpu-baseline --log-n 16 --steps 200 --opt msm_small_lanes=0
one --no-cpu-baseline --log-n 16 --steps 320 --batch 32
one --no-cpu-baseline --log-n 16 --steps 320 --batch 32 --opt msm_small_lanes=0
