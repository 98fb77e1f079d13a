#!/bin/bash
# Copies what one measurement round (tools/profile_round.sh + tools/pmc_workloads.sh, prefix P) left under gpurun_out/ into
# profiles/ (the tracked copies the docs and bench.py cite).  Usage: tools/collect_profiles.sh r04z
set -eu
P=$1
cd "$(dirname "$0")/.."
G=gpurun_out
for WL in macro micro itscp_hybrid itscp_macro itscp_stepwise; do
  cp $G/${P}_${WL}_kernel_stats.csv profiles/
  tail -n 1 $G/${P}_${WL}_bench.json > profiles/${P}_${WL}_bench.json
done
tail -n 1 $G/${P}_macro_bench_unprofiled.json > profiles/${P}_macro_bench_unprofiled.json
cp $G/${P}_pmc_counters/summary.csv profiles/${P}_pmc_workload_counters.csv
cp $G/${P}_pmc_counters/issue_counters.json profiles/issue_counters.json
cp $G/${P}_pmc_counters/pmc_traffic.json profiles/pmc_traffic.json
[ -f $G/${P}_gputest.log ] && cp $G/${P}_gputest.log profiles/
[ -f $G/${P}_slow_mirror_test.log ] && cp $G/${P}_slow_mirror_test.log profiles/
ls -la profiles | grep "${P}_"
