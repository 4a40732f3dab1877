#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines, kernel-trace statistics, the PMC passes (traffic at 2^20 and 2^16, VALU /
# clock counters of the accumulation) of the bench command.  Outputs land in gpurun_out/; copy the summaries into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
# the bench lines first: after a minute of continuous profiling the GPU clocks sag by ~4 %
cd $REPO && python bench.py --steps 30 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
{
  python bench.py --log-n 16 --steps 200 --warmup 10 --no-cpu-baseline
  python bench.py --log-n 16 --steps 320 --warmup 64 --batch 32 --no-cpu-baseline
  python bench.py --log-n 16 --latency --steps 30 --warmup 5
  python bench.py --log-n 4 --latency --steps 30 --warmup 5
  python bench.py --log-n 4 --roots integers --steps 640 --warmup 64 --batch 32
  python bench.py --roots integers --steps 30 --warmup 5
  python bench.py --latency --steps 20 --warmup 4
} > $OUT/bench_other.jsonl 2> $OUT/bench_other.err
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_ser -- $CMD --serialize > $OUT/prof_stats_ser.log 2>&1
CMD3="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD3 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch16 -- $CMD3 --log-n 16 > $OUT/pmc_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write16 -- $CMD3 --log-n 16 > $OUT/pmc_write16.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_acc -- $CMD3 --serialize > $OUT/pmc_acc.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
python tools/pmc_summary.py $OUT/pmc_fetch16 $OUT/pmc_write16 > $OUT/pmc_traffic_2p16.json
python tools/pmc_counters.py $OUT/pmc_acc k_msm > $OUT/pmc_acc.txt
find $OUT/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof_stats_ser -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serialized.csv \;
# the raw per-dispatch traces are large; keep only the summaries
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch16 $OUT/pmc_write16 $OUT/pmc_acc $OUT/prof_stats_ser
find $OUT/prof_stats -name "*kernel_trace.csv" -delete
