#!/bin/bash
# Counter passes over the macro rollout kernels, forward and reverse (GPU box).  Usage: tools/pmc_macro_fwd.sh <out-dir> <variant> <waves>
# One rocprofv3 run per counter group (SQ has 8 slots per pass); --kernel-trace only, never with a --sys-trace domain.
set -u
OUT=$1; V=${2:-0}; W=${3:-0}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT"
G4="GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
G5="FETCH_SIZE"
G6="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
G7="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/g$i" -- python3 "$REPO/tools/run_macro_fwd.py" "$V" "$W" 2 > "$OUT/g$i.log" 2>&1
  tail -n 1 "$OUT/g$i.log"
done
python3 - "$OUT" "$REPO" <<'PY'
import csv, glob, json, os, sys, collections
out, repo = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "rollout" not in k:
            continue
        name = "fwd2" if "fwd2" in k else ("fwd" if "fwd" in k else "bwd")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write("%s,%s,%.6g,%d\n" % (k, c, sum(v) / len(v), len(v)))
print(open(os.path.join(out, "summary.csv")).read())
# what bench.py quotes as roofline.issue_side: derived per-cell-step figures of the two rollout kernels + the fingerprint of
# the library the passes ran on (copy issue_counters.json and summary.csv into profiles/)
sys.path.insert(0, repo)
sys.path.insert(0, os.path.join(repo, "diff-hybrid-traffic-sim_amd"))
import bench
units = 1024 * 512 * 1000
simds = 1024
m = lambda k, c: (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else None
rec = {}
for key, name in (("fwd2" if acc.get("fwd2") else "fwd", "rollout_fwd"), ("bwd", "rollout_bwd")):
    if not acc.get(key):
        continue
    e = {}
    if m(key, "SQ_INSTS_VALU"): e["vector_per_cell_step"] = round(m(key, "SQ_INSTS_VALU") * 64 / units, 1)
    if m(key, "SQ_INSTS_SALU"): e["scalar_per_cell_step"] = round(m(key, "SQ_INSTS_SALU") * 64 / units, 1)
    if m(key, "SQ_INSTS_LDS"): e["lds_per_cell_step"] = round(m(key, "SQ_INSTS_LDS") * 64 / units, 1)
    if m(key, "SQ_INSTS_VALU_CVT"): e["conversions_per_cell_step"] = round(m(key, "SQ_INSTS_VALU_CVT") * 64 / units, 1)
    if m(key, "SQ_ACTIVE_INST_VALU") and m(key, "GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["vector_alu_busy"] = round(m(key, "SQ_ACTIVE_INST_VALU") * 4 / (simds * m(key, "GRBM_GUI_ACTIVE") / 8), 3)
    if m(key, "SQ_WAIT_ANY") and m(key, "SQ_WAVE_CYCLES"): e["wait_any_frac"] = round(m(key, "SQ_WAIT_ANY") / m(key, "SQ_WAVE_CYCLES"), 3)
    if m(key, "SQ_LDS_BANK_CONFLICT") and m(key, "SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(m(key, "SQ_LDS_BANK_CONFLICT") / m(key, "SQ_LDS_IDX_ACTIVE"), 3)
    rec[name] = e
json.dump({"macro_straight_1024x512x1000": rec, "library_code_sha16": bench.library_code_sha16(),
           "source": "profiles/%s_pmc_macro_counters.csv" % os.path.basename(out.rstrip("/")).split("_")[0]},
          open(os.path.join(out, "issue_counters.json"), "w"), indent=1)
print(open(os.path.join(out, "issue_counters.json")).read())
PY
