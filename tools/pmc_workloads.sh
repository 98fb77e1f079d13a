#!/bin/bash
# Counter passes over the rollout kernels of bench.py's workloads, forward and reverse (GPU box).
#   tools/pmc_workloads.sh <out-dir> [workload ...]        (default: macro micro itscp_hybrid itscp_stepwise)
# One rocprofv3 run per counter group and workload (SQ has 8 slots per pass; FETCH_SIZE and WRITE_SIZE each alone) with
# --kernel-trace only -- never a --sys-trace / hip / hsa domain beside --pmc.  tools/pmc_workloads_summary.py turns the passes into
# <out-dir>/summary.csv, issue_counters.json (what bench.py quotes as issue_side, per workload and kernel) and pmc_traffic.json.
set -u
OUT=$1; shift
WLS=${*:-macro micro itscp_hybrid itscp_stepwise}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT"
G4="GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
G7="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
for WL in $WLS; do
  i=0
  for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/$WL/g$i" -- python3 "$REPO/tools/run_workload.py" $WL 3 > "$OUT/$WL.g$i.log" 2>&1
    grep -h "^WORKLOAD" "$OUT/$WL.g$i.log" | tail -n 1 | cut -c1-200
  done
done
python3 "$REPO/tools/pmc_workloads_summary.py" "$OUT"
