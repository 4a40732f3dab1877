for w in 2 4 8; do python bench.py --emulate-world $w --steps 12 --warmup 4 2>/dev/null | tail -1; done
python bench.py --emulate-world 8 --transport torch --steps 12 --warmup 4 2>/dev/null | tail -1
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('single', d['value'])"
