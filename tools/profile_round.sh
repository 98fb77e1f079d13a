#!/bin/bash
# One measurement round on the GPU box: rocprofv3 kernel stats + the bench line of the same run for the four workloads, an
# un-profiled default bench run (with the CPU baseline leg).
#   tools/profile_round.sh <prefix>        ->  gpurun_out/<prefix>_*   (copy what is to be kept into profiles/)
set -u
P=$1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for WL in macro micro itscp_hybrid itscp_macro itscp_stepwise; do
  D=$OUT/${P}_prof_$WL
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-also --workload $WL > "$OUT/${P}_${WL}_bench.json" 2> "$OUT/${P}_${WL}_bench.err"
  S=$(find "$D" -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp "$S" "$OUT/${P}_${WL}_kernel_stats.csv"
  tail -c 400 "$OUT/${P}_${WL}_bench.json"; echo
done
python3 "$REPO/bench.py" > "$OUT/${P}_macro_bench_unprofiled.json" 2> "$OUT/${P}_macro_bench_unprofiled.err"
# (the PMC passes -- FETCH_SIZE / WRITE_SIZE and the SQ groups, every workload -- are tools/pmc_workloads.sh)
