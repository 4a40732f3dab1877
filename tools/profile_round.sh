#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics and the two PMC passes of the bench command.
# Outputs land in gpurun_out/prof_*; copy the summaries into profiles/ afterwards.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
# the bench line first: after a minute of continuous profiling the GPU clocks sag by ~4 %
cd $REPO && python bench.py --steps 30 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
CMD3="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD3 > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
find $OUT/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw per-dispatch traces are large; keep only the summaries
rm -rf $OUT/pmc_fetch $OUT/pmc_write
find $OUT/prof_stats -name "*kernel_trace.csv" -delete
# the multi-GPU emulation block lives in tools/profile_emulation.sh: run it in a gpurun call of its own (after the ~70 s of
# continuous profiling above the GPU clocks sag and the rounds come out 5 % slower)
