#!/bin/bash
# One measurement round on the GPU box: rocprofv3 kernel stats + the bench line of the same run for the four workloads, an
# un-profiled default bench run (with the CPU baseline leg), and the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate
# runs, --kernel-trace only) for the two straight-lane workloads.
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
PM=$OUT/${P}_pmc
for WL in macro micro; do
  for C in FETCH_SIZE WRITE_SIZE; do
    T=fetch; [ $C = WRITE_SIZE ] && T=write
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$PM/${WL}_$T" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload $WL > "$PM.${WL}_$T.log" 2>&1
  done
done
python3 "$REPO/tools/pmc_summary.py" "$PM" "$OUT/${P}_pmc_traffic.json" "$OUT/${P}_pmc_rollout_kernels.csv" | tail -30
