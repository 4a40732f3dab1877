#!/bin/bash
# same-box A/B of library builds: tools/r3_ab.sh <out name> <repeats> <steps> -- "<lib or default> [bench args]" ...
# Every configuration is run <repeats> times, interleaved (A B C A B C ...), and the spread is printed.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; reps=$2; steps=$3; shift 4
OUT=$REPO/gpurun_out/$name
mkdir -p $OUT
cd $REPO
: > $OUT/ab_raw.txt
cfgs=("$@")
for r in $(seq 1 $reps); do
  for cfg in "${cfgs[@]}"; do
    read -r -a parts <<< "$cfg"
    lib=${parts[0]}
    if [ $lib = default ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$REPO/zksnark_rs_amd/libzkgpu_$lib.so; fi
    v=$(python bench.py --no-cpu-baseline --steps $steps --warmup 5 "${parts[@]:1}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['proof_sha'])")
    echo "$cfg | $v" >> $OUT/ab_raw.txt
  done
done
python - <<PY > $OUT/ab.txt
import collections
d = collections.OrderedDict()
for line in open("$OUT/ab_raw.txt"):
    cfg, v = line.rsplit("|", 1)
    val, sha = v.split()
    d.setdefault(cfg.strip(), []).append((float(val), sha))
for cfg, vs in d.items():
    xs = sorted(x for x, _ in vs)
    print("%-60s n=%d  median %.2f  min %.2f  max %.2f  proofs/s  sha %s" % (cfg, len(xs), xs[len(xs) // 2], xs[0], xs[-1], ",".join(sorted(set(s for _, s in vs)))))
PY
cat $OUT/ab.txt
