#!/bin/bash
# Soak of the round's build (through gpurun): many proofs over every form and witness kind, 8 distinct (witness, r, s) sets cycled; bench.py
# compares EVERY timed proof with the synchronous single-GPU proof of the same inputs and fails the run on a mismatch.
#   tools/gpu.sh -- 'bash tools/soak.sh' ; copy gpurun_out/soak.txt to profiles/rN_soak.txt
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/soak.txt
echo "Soak ($(cat .build_commit 2>/dev/null)): bench.py with 8 distinct (witness, r, s) sets; every timed proof is compared with the synchronous zk_prove_dev of the same inputs (bench.py asserts)." > $OUT
echo "columns: proofs/s, timed steps, proof sha per set" >> $OUT
run() {
  python bench.py --no-cpu-baseline --sets 8 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['steps'], d['config']['proof_shas'], {k: d[k] for k in ('degraded',) if k in d})" >> $OUT
  echo "rc=${PIPESTATUS[0]} $*" >> $OUT
}
run --steps 3000 --warmup 4
run --steps 600 --warmup 4 --witness boolean
run --steps 600 --warmup 4 --witness small
run --log-n 16 --steps 10000 --warmup 8
run --roots arbitrary --steps 600 --warmup 4
run --roots integers --steps 1000 --warmup 4
run --log-n 16 --batch 32 --steps 300 --warmup 4
run --log-n 12 --roots integers --batch 16 --steps 300 --warmup 4
cat $OUT
