one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_proof']; print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', {x: k[x] for x in ('msm_hist','msm_scatter','msm_sort_bins','msm_offsets')})" "$@"; }
for b in 8 9 10 11 8; do one --no-cpu-baseline --steps 40 --opt msm_sort_bins_log=$b; done
for b in 8 10; do one --no-cpu-baseline --steps 12 --serialize --opt msm_sort_bins_log=$b; done
