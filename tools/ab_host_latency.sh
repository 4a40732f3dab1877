one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms')" "$@"; }
one --latency --steps 30
one --latency --steps 30 --witness-from pinned
one --latency --steps 30 --witness-from pageable
