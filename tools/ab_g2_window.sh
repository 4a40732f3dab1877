one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
one --no-cpu-baseline --steps 40
for w in 2017 1917 2117 2217 170020 170019 172019 172018 2017; do one --no-cpu-baseline --steps 40 --window-bits $w; done
one --no-cpu-baseline --steps 40
