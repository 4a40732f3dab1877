one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms')" "$@"; }
for i in 1 2; do for n in 4 12 16; do
one --latency --log-n $n --steps 40 --warmup 5
one --latency --log-n $n --steps 40 --warmup 5 --opt split_assembly=0
done; done
one --no-cpu-baseline --log-n 16 --steps 200
one --no-cpu-baseline --log-n 16 --steps 200 --opt split_assembly=0
