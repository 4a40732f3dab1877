#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4h
mkdir -p $OUT
M=$REPO/zksnark_rs_amd/libzkgpu_measure.so
for w in 2 4 8; do ZKGPU_LIB=$M python bench.py --emulate-world $w --steps 20 --warmup 4 2>$OUT/emul_$w.err | tail -1; done > $OUT/emul.txt
for r in 0 1 2; do ZKGPU_LIB=$M ZK_COMM_FORCE_RCCL=1 python tools/rccl_starvation.py 20 $r 2>$OUT/starv_$r.err | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; done > $OUT/rccl_starvation.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['hbm_measured_GBps_whole_proof'], d['roofline']['traffic'], d['roofline']['traffic_source_commit'], d['roofline']['valu'])" > $OUT/line_check.txt 2>&1
