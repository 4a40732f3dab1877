#!/bin/bash
# Copies the summaries a profile round left under gpurun_out/prof/ (tools/profile_round.sh) into profiles/ as rN_*:  tools/install_profiles.sh 5 [commit]
# [commit]: the profiled snapshot was an uncommitted working tree (files say "<HEAD>+uncommitted", tools/gpu.sh) that was committed
# UNCHANGED as <commit> afterwards -- the files then name that commit.
set -eu
N=${1:?round number}
cd "$(dirname "$0")/.."
P=gpurun_out/prof
for f in bench_other.jsonl kernel_stats.csv kernel_stats_serialized.csv kernels_serialized_by_grid.txt pmc_acc.json pmc_acc_2p16.json pmc_counters.txt \
         pmc_ntt.txt pmc_traffic.json pmc_traffic_2p16.json rccl_starvation.txt timeline_lone_2p16.txt timeline_pipelined_2p20.txt valu_budget.txt \
         ubench_valu.txt window_sweep_2p20.jsonl; do
  [ -s $P/$f ] && cp $P/$f profiles/r${N}_$f
done
[ -s $P/emul.txt ] && cp $P/emul.txt profiles/r${N}_multi_gpu_emulation.txt
[ -s $P/lone.txt ] && cp $P/lone.txt profiles/r${N}_lone_latency.txt
for s in "" _100 _20; do
  [ -s gpurun_out/bench_final$s.json ] && tail -1 gpurun_out/bench_final$s.json > profiles/r${N}_bench$( [ -z "$s" ] && echo "" || echo "${s}steps" ).json
done
if [ -n "${2:-}" ]; then sed -i -E "s/[0-9a-f]{12}\+uncommitted/$2/" $(grep -lE "[0-9a-f]{12}\+uncommitted" profiles/r${N}_* || echo /dev/null); fi
ls -la profiles/r${N}_* | wc -l
