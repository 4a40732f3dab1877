one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], 'ms')" "$@"; }
for n in 4 8; do
one --latency --log-n $n --steps 40 --warmup 5
for w in 4 5 6 7 10; do one --latency --log-n $n --steps 40 --warmup 5 --window-bits $w; done
done
