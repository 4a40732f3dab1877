one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
one --no-cpu-baseline --steps 40
one --no-cpu-baseline --steps 40 --window-bits 17
one --latency --steps 20
one --latency --steps 20 --window-bits 17
one --no-cpu-baseline --steps 40 --roots integers
one --no-cpu-baseline --steps 40 --roots integers --window-bits 17
one --no-cpu-baseline --steps 40 --log-n 19
one --no-cpu-baseline --steps 40 --log-n 19 --window-bits 17
one --no-cpu-baseline --steps 40 --witness boolean
one --no-cpu-baseline --steps 40 --witness boolean --window-bits 17
bash tools/emulate_world.sh
