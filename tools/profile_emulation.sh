#!/bin/bash
# Runs on the GPU box (through gpurun), in a call of its own: rank 0's share of the multi-GPU protocols on this one GPU.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
# multi-GPU protocols timed as rank 0's share on this one GPU (local copies in place of the collectives).  Run this block in a
# gpurun call of its own: after the ~70 s of continuous profiling above the GPU clocks sag and the rounds come out 5 % slower.
python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'same_box_single_gpu_proofs_per_s': d['value'], 'ms_per_proof': d['ms_per_step']}))" > $OUT/emulation.jsonl
for W in 2 4 8; do python bench.py --no-cpu-baseline --steps 16 --warmup 4 --emulate-world $W --mode exchange 2>/dev/null | tail -1 >> $OUT/emulation.jsonl; done
for W in 2 4 8; do python bench.py --no-cpu-baseline --steps 32 --warmup 8 --emulate-world $W --mode shard 2>/dev/null | tail -1 >> $OUT/emulation.jsonl; done
