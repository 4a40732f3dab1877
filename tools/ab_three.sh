one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_proof']; print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', k.get('msm_sort_bins'), d['config']['proof_sha'])" "$@"; }
for i in 1 2; do
for l in 6 8 9 7; do one --no-cpu-baseline --steps 40 --opt msm_three_level_low=$l; done
one --no-cpu-baseline --steps 40 --opt msm_three_level_bits=0
done
for l in 6 8 9; do one --no-cpu-baseline --steps 12 --serialize --opt msm_three_level_low=$l; done
one --no-cpu-baseline --steps 12 --serialize --opt msm_three_level_bits=0
