#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_workloads.sh: <dir>/<workload>/g*/**/counter_collection.csv + <dir>/<workload>.g*.log ->
<dir>/summary.csv (mean per dispatch of every counter, per workload and kernel), <dir>/issue_counters.json (per-unit instruction
counts, vector-ALU busy fraction, wait fraction, LDS bank conflicts: what bench.py quotes next to each `limiter`) and
<dir>/pmc_traffic.json (HBM bytes per launch: FETCH_SIZE doubled -- gfx950 tallies 128-byte read requests at 64 B,
MI355X_MICROARCH.md, HBM section -- plus WRITE_SIZE, counter unit KiB).  The first dispatch of each kernel is dropped (first touch
of the tape) when there are more."""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 1024          # 256 CUs x 4


def classify(kernel):
    k = kernel
    if "rollout_fwd" in k or "net_hybrid_fwd_kernel" in k or "ns_persist_fwd_kernel" in k or "net_macro_fwd_kernel" in k:
        return "rollout_fwd"
    if "rollout_bwd" in k or "net_hybrid_bwd_kernel" in k or "ns_persist_bwd_kernel" in k or "net_macro_bwd_kernel" in k:
        return "rollout_bwd"
    return None


def summarise(out, quiet=False):
    """(issue, traffic) of the passes under `out`; also written there as issue_counters.json / pmc_traffic.json / summary.csv."""
    issue, traffic, lines = {}, {}, ["workload,kernel,counter,mean_per_dispatch,dispatches"]
    sha = None
    for wl_dir in sorted(d for d in glob.glob(os.path.join(out, "*")) if os.path.isdir(d)):
        key = os.path.basename(wl_dir)
        info = None
        for log in sorted(glob.glob(os.path.join(out, key + ".g*.log"))):
            for line in open(log, errors="replace"):
                if line.startswith("WORKLOAD "):
                    info = json.loads(line[len("WORKLOAD "):])
        if info is None:
            if not quiet:
                print("no WORKLOAD line for %s: skipped" % key)
            continue
        sha = sha or info.get("library_code_sha16")
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        names = collections.defaultdict(set)
        for path in glob.glob(os.path.join(wl_dir, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = classify(r["Kernel_Name"])
                if k is None:
                    continue
                names[k].add(r["Kernel_Name"].split("(")[0][:120])
                acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))

        def m(k, c):
            v = sorted(acc[k].get(c, []))
            if not v:
                return None
            use = [x for _, x in v[1:]] or [x for _, x in v]
            return sum(use) / len(use)
        for k in sorted(acc):
            for c in sorted(acc[k]):
                lines.append("%s,%s,%s,%.6g,%d" % (info["name"].replace(",", ";"), k, c, m(k, c), len(acc[k][c])))
        units = float(info["units"])
        rec, tr = {}, {}
        for k in ("rollout_fwd", "rollout_bwd"):
            if k not in acc:
                continue
            e = {"kernel": sorted(names[k])[0] if names[k] else None}
            per = lambda c, digits=1: None if m(k, c) is None else round(m(k, c) * 64 / units, digits)      # noqa: E731
            for tag, c in (("vector_per_unit", "SQ_INSTS_VALU"), ("scalar_per_unit", "SQ_INSTS_SALU"), ("lds_per_unit", "SQ_INSTS_LDS"),
                           ("conversions_per_unit", "SQ_INSTS_VALU_CVT"), ("vmem_writes_per_unit", "SQ_INSTS_VMEM_WR"),
                           ("vmem_reads_per_unit", "SQ_INSTS_VMEM_RD")):
                if per(c) is not None:
                    e[tag] = per(c, 2)
            if m(k, "SQ_ACTIVE_INST_VALU") and m(k, "GRBM_GUI_ACTIVE"):
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                e["vector_alu_busy"] = round(m(k, "SQ_ACTIVE_INST_VALU") * 4 / (SIMDS * m(k, "GRBM_GUI_ACTIVE") / 8), 3)
            if m(k, "SQ_WAIT_ANY") and m(k, "SQ_WAVE_CYCLES"):
                e["wait_any_frac"] = round(m(k, "SQ_WAIT_ANY") / m(k, "SQ_WAVE_CYCLES"), 3)
            if m(k, "SQ_WAIT_INST_ANY") and m(k, "SQ_WAVE_CYCLES"):
                e["wait_inst_frac"] = round(m(k, "SQ_WAIT_INST_ANY") / m(k, "SQ_WAVE_CYCLES"), 3)
            if m(k, "SQ_WAVE_CYCLES") and m(k, "GRBM_GUI_ACTIVE"):
                # resident wavefronts per SIMD, averaged over the kernel: wave-cycles (quad-cycles x 4) / (SIMDs x kernel cycles)
                e["waves_per_simd"] = round(m(k, "SQ_WAVE_CYCLES") * 4 / (SIMDS * m(k, "GRBM_GUI_ACTIVE") / 8), 2)
            if m(k, "SQ_LDS_BANK_CONFLICT") and m(k, "SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = round(m(k, "SQ_LDS_BANK_CONFLICT") / m(k, "SQ_LDS_IDX_ACTIVE"), 3)
            if m(k, "SQ_ACTIVE_INST_LDS") and m(k, "GRBM_GUI_ACTIVE"):
                e["lds_busy"] = round(m(k, "SQ_ACTIVE_INST_LDS") * 4 / (SIMDS * m(k, "GRBM_GUI_ACTIVE") / 8), 3)
            t = {}
            if m(k, "FETCH_SIZE") is not None:
                t["fetch_bytes_corrected"] = m(k, "FETCH_SIZE") * 1024.0 * 2.0
            if m(k, "WRITE_SIZE") is not None:
                t["write_bytes"] = m(k, "WRITE_SIZE") * 1024.0
            if t:
                t["hbm_bytes"] = t.get("fetch_bytes_corrected", 0.0) + t.get("write_bytes", 0.0)
                t["moved_bytes_counted_by_bench"] = info["moved_bytes"]
                tr[k] = t
                # (the same passes, under the same fingerprint: what bench.py quotes per kernel in the also records)
                e["hbm_bytes_per_launch"] = t["hbm_bytes"]
                e["hbm_fetch_write_bytes"] = [t.get("fetch_bytes_corrected"), t.get("write_bytes")]
                ms = info.get("fwd_ms" if k == "rollout_fwd" else "bwd_ms")
                if ms:
                    e["hbm_GBps_under_profiler"] = round(t["hbm_bytes"] / ms / 1e6, 1)
            rec[k] = e
        issue[info["name"]] = rec
        traffic[info["name"]] = tr
    open(os.path.join(out, "summary.csv"), "w").write("\n".join(lines) + "\n")
    tag = os.path.basename(out.rstrip("/")).split("_")[0]
    issue.update(library_code_sha16=sha, source="profiles/%s_pmc_workload_counters.csv" % tag)
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_workloads.sh, profiles/%s_pmc_workload_counters.csv); "
                        "counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B read requests at 64 B)" % tag)
    json.dump(issue, open(os.path.join(out, "issue_counters.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    return issue, traffic


def main():
    out = sys.argv[1]
    summarise(out)
    print(open(os.path.join(out, "issue_counters.json")).read())
    print(open(os.path.join(out, "pmc_traffic.json")).read())


if __name__ == "__main__":
    main()
