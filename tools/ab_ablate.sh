# needs the measurement build: make -C zksnark_rs_amd/csrc measure; export ZKGPU_LIB=$PWD/zksnark_rs_amd/libzkgpu_measure.so
one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
for i in 1 2; do
one --no-cpu-baseline --steps 40
one --no-cpu-baseline --steps 40 --opt ablate=1
one --no-cpu-baseline --steps 40 --window-bits 17
one --no-cpu-baseline --steps 40 --window-bits 17 --opt ablate=1
done
