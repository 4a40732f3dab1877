#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "captured_graph" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
for n in 4 12 16 20; do
  for g in "" "--lone-graph"; do
    python bench.py --log-n $n --latency --steps 60 --warmup 8 $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$n gates lone proof $g:', d['ms_per_step'], 'ms')"
  done
done > $OUT/lone_graph.txt 2>&1
