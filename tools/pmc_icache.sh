#!/bin/bash
# Instruction-cache counters of a workload's rollout kernels (GPU box).  Usage: tools/pmc_icache.sh <out-dir> <workload>
set -u
OUT=$1; WL=${2:-itscp_hybrid}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d "$OUT/ic" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload $WL > "$OUT/ic.log" 2>&1
tail -n 2 "$OUT/ic.log"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "dhts::" not in k:
            continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s %.4g  (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
