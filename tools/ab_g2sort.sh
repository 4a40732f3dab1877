one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
one --no-cpu-baseline --steps 40
one --no-cpu-baseline --steps 40 --opt g2_sort_main=0
one --no-cpu-baseline --steps 40
one --no-cpu-baseline --steps 40 --opt g2_sort_main=0
one --no-cpu-baseline --steps 40 --log-n 19
one --no-cpu-baseline --steps 40 --log-n 19 --opt g2_sort_main=0
one --no-cpu-baseline --steps 40 --log-n 18
one --no-cpu-baseline --steps 40 --log-n 18 --opt g2_sort_main=0
one --latency --steps 20
one --latency --steps 20 --opt g2_sort_main=0
