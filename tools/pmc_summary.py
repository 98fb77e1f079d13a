#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json.

Usage (on the GPU box, from /tmp with TMPDIR=/tmp; one counter per pass, --kernel-trace only):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/macro_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir>/macro_write -- python3 bench.py ...
    (same with --workload micro into micro_fetch / micro_write)
    python3 tools/pmc_summary.py <dir> profiles/pmc_traffic.json profiles/<prefix>_pmc_rollout_kernels.csv

Counter unit: KiB.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, MI355X_MICROARCH.md, HBM section).
The first dispatch of each kernel (first touch of the tape) is dropped; the rest are averaged.
"""
import csv
import glob
import json
import os
import sys

NAMES = {"macro": "macro_straight_1024x512x1000", "micro": "micro_idm_4096x256x1000"}


def rows(pass_dir):
    for path in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r


def main():
    src, out_json, out_csv = sys.argv[1:4]
    table, per = [], {}
    for wl in ("macro", "micro"):
        for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            d = os.path.join(src, "%s_%s" % (wl, tag))
            if not os.path.isdir(d):
                continue
            for r in rows(d):
                name = r["Kernel_Name"]
                if "rollout_fwd" in name:
                    k = "rollout_fwd"
                elif "rollout_bwd" in name:
                    k = "rollout_bwd"
                else:
                    continue
                if r["Counter_Name"] != counter:
                    continue
                v = float(r["Counter_Value"])
                table.append((wl, counter, k, int(r["Dispatch_Id"]), v))
                per.setdefault((wl, k, counter), []).append((int(r["Dispatch_Id"]), v))
    res = {}
    for (wl, k, counter), vals in per.items():
        vals.sort()
        use = [v for _, v in vals[1:]] or [v for _, v in vals]
        mean_kib = sum(use) / len(use)
        e = res.setdefault(NAMES[wl], {}).setdefault(k, {})
        if counter == "FETCH_SIZE":
            e["fetch_bytes_corrected"] = mean_kib * 1024.0 * 2.0
        else:
            e["write_bytes"] = mean_kib * 1024.0
    for wl in res.values():
        for e in wl.values():
            e["hbm_bytes"] = e.get("fetch_bytes_corrected", 0.0) + e.get("write_bytes", 0.0)
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (%s); counter unit KiB; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B read requests at 64 B)" % os.path.basename(out_csv))
    json.dump(res, open(out_json, "w"), indent=1)
    with open(out_csv, "w") as f:
        f.write("workload,counter,kernel,dispatch,value_KiB\n")
        for t in sorted(table):
            f.write("%s,%s,%s,%d,%r\n" % t)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
