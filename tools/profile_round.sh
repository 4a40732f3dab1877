#!/bin/bash
# Runs on the GPU box (through gpurun; build tools/_bin/ubench_assemble first: see tools/ubench_assemble.hip): the bench lines, kernel-trace statistics, the PMC passes (traffic at 2^20 and 2^16, the counters
# of the accumulation, the VALU budget of a proof) and the timelines of the bench command.  Outputs land in gpurun_out/prof/; copy
# the summaries into profiles/ as rN_*.  The serialized legs need the measurement build (make -C zksnark_rs_amd/csrc measure).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
M=$REPO/zksnark_rs_amd/libzkgpu_measure.so
mkdir -p $OUT
# the bench lines first: after a minute of continuous profiling the GPU clocks sag by ~4 %
cd $REPO && python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 100 --warmup 5 > $OUT/bench_100steps.json 2>> $OUT/bench.err
{
  python bench.py --log-n 16 --steps 300 --warmup 10 --no-cpu-baseline
  python bench.py --log-n 16 --steps 320 --warmup 64 --batch 32 --no-cpu-baseline
  python bench.py --log-n 16 --latency --steps 40 --warmup 8
  python bench.py --log-n 4 --latency --steps 40 --warmup 8
  python bench.py --log-n 4 --roots integers --steps 640 --warmup 64 --batch 32
  python bench.py --roots integers --steps 60 --warmup 5
  python bench.py --roots arbitrary --steps 30 --warmup 4
  python bench.py --roots arbitrary --log-n 16 --steps 100 --warmup 8
  python bench.py --latency --steps 20 --warmup 4
  python bench.py --log-n 21 --steps 30 --warmup 4 --no-cpu-baseline
  python bench.py --log-n 22 --steps 20 --warmup 4 --no-cpu-baseline
} > $OUT/bench_other.jsonl 2> $OUT/bench_other.err
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
ZKGPU_LIB=$M rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_ser -- $CMD --serialize > $OUT/prof_stats_ser.log 2>&1
CMD3="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD3 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch16 -- $CMD3 --log-n 16 > $OUT/pmc_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write16 -- $CMD3 --log-n 16 > $OUT/pmc_write16.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_acc -- $CMD3 > $OUT/pmc_acc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_acc16 -- $CMD3 --log-n 16 > $OUT/pmc_acc16.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_ntt -- $CMD3 > $OUT/pmc_ntt.log 2>&1
python $REPO/tools/pmc_counters.py $OUT/pmc_ntt k_ntt_tile > $OUT/pmc_ntt.txt; rm -rf $OUT/pmc_ntt
# the timed leg of this run is proofs 16 .. 55 (4 synchronous set-up proofs, warm-up + priming 8, warm-up 4, then 40 steps): the window
# cut below starts 20 proofs into it, so that no synchronisation of bench.py falls inside
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace16 -- python $REPO/bench.py --latency --log-n 16 --steps 6 --warmup 3 > $OUT/trace16.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
python tools/pmc_summary.py $OUT/pmc_fetch16 $OUT/pmc_write16 > $OUT/pmc_traffic_2p16.json
python tools/pmc_acc_summary.py $OUT/pmc_acc 20 > $OUT/pmc_acc.json
python tools/pmc_acc_summary.py $OUT/pmc_acc16 16 > $OUT/pmc_acc_2p16.json
python tools/valu_budget.py $OUT/pmc_acc "round-6 build, 2^20 gates" > $OUT/valu_budget.txt
python tools/pmc_counters.py $OUT/pmc_acc k_msm k_ntt > $OUT/pmc_counters.txt
find $OUT/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof_stats_ser -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serialized.csv \;
python tools/trace_csv.py "$(find $OUT/trace -name '*kernel_trace.csv' | head -1)" 36 16 > $OUT/timeline_pipelined_2p20.txt 2>&1
python tools/trace_one_proof.py "$(find $OUT/trace16 -name '*kernel_trace.csv' | head -1)" > $OUT/timeline_lone_2p16.txt 2>&1
python tools/trace_kernels.py "$(find $OUT/prof_stats_ser -name '*kernel_trace.csv' | head -1)" > $OUT/kernels_serialized_by_grid.txt 2>&1
# lone-proof latency split (host enqueue / wait), the closing kernel's parts, once-per-root-set and once-per-CRS costs, the multi-GPU emulation
{
  for n in 16 14 12 4 20; do python tools/lone_breakdown.py --log-n $n; done
  [ -x tools/_bin/ubench_assemble ] && timeout 60 tools/_bin/ubench_assemble
  python tools/time_root_tables.py 16 18 20 22
  python tools/time_change_of_basis.py 12 14 16
  python tools/time_first_proof.py 20 16 22 | grep -v amdgpu.ids
  python tools/host_pacing.py --log-n 16 --depth 4 --proofs 24 | grep -v amdgpu.ids | head -3
} > $OUT/lone.txt 2>/dev/null
# one GPU doing rank 0's share of every N > 1 leg on the round's build: the scalar exchange (copies in place of the all-to-alls),
# the window shard (windows w = 0 mod N of every product) and the point-range shard
{
  for w in 2 4 8; do ZKGPU_LIB=$M python bench.py --emulate-world $w --steps 20 --warmup 4 2>/dev/null | tail -1; done
  for w in 2 4 8; do ZKGPU_LIB=$M python bench.py --emulate-world $w --mode shard --shard windows --steps 20 --warmup 4 2>/dev/null | tail -1; done
  for w in 2 4 8; do ZKGPU_LIB=$M python bench.py --emulate-world $w --mode shard --shard points --steps 20 --warmup 4 2>/dev/null | tail -1; done
  for w in 2 4 8; do ZKGPU_LIB=$M python bench.py --emulate-world $w --mode shard --shard buckets --steps 20 --warmup 4 2>/dev/null | tail -1; done
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps({'diagnostic': 'one whole prover on the same box (replicas = N times this)', 'proofs_per_s': d['value'], 'steady_state_ms_per_proof': d.get('steady_state_ms_per_proof')}))"
} > $OUT/emul.txt
# same-round inputs of the bench line: the sustained issue rates (tools/ubench_valu.hip) and BASELINE config 4's window sweep, both regimes
# of SURVEY 8(d) (buckets in LDS c = 6 .. 10, buckets in HBM c = 11 .. 22)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu > /dev/null 2>&1 && timeout 300 /tmp/ubench_valu > $OUT/ubench_valu.txt 2>/dev/null
timeout 900 python tools/window_sweep.py --log-n 20 --windows 11,12,13,14,15,16,17,18,19,20,21,22 > $OUT/window_sweep_2p20.jsonl 2> $OUT/window_sweep.err
ZK_COMM_FORCE_RCCL=1 python tools/rccl_starvation.py > $OUT/rccl_starvation.txt 2>&1
# the raw per-dispatch traces are large; keep only the summaries
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch16 $OUT/pmc_write16 $OUT/pmc_acc $OUT/pmc_acc16 $OUT/prof_stats_ser $OUT/prof_stats $OUT/trace $OUT/trace16
ls -la $OUT
