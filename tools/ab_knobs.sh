one() { python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(' '.join(sys.argv[1:]), '->', d['value'], 'proofs/s', d['ms_per_step'], 'ms', d['config']['proof_sha'])" "$@"; }
B="--no-cpu-baseline --steps 40"
one $B
one $B --depth 3
one $B --depth 4
one $B --fold 8
one $B --fold 2
one $B --lane-entries 48
one $B --lane-entries 24
one $B
one $B --opt g2_sort_main=1
one $B --opt msm_acc_stream=1
one $B --opt defer_msm=0
one $B --batch 2
