#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r4f
mkdir -p $OUT
timeout 60 tools/_bin/ubench_waitvalue > $OUT/waitvalue.txt 2>&1; echo "rc=$?" >> $OUT/waitvalue.txt
bash tools/ab.sh r4f_ab 3 60 -- "default" "measure" "measure --opt msm_unchain_lanes=1000000000"
cp gpurun_out/r4f_ab/ab.txt $OUT/
