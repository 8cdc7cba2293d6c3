#!/usr/bin/env python3
"""Reduce rocprofv3 counter_collection CSVs to per-kernel, per-launch means (JSON on stdout).

    python scripts/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rN/xxx_pmc_summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("f20_order", "fused20", "seg_kernel", "linearize", "condense", "qp_dense", "expand", "rowqp_sort", "rowqp", "argmin", "epilogue", "waypoints", "shoot"):
        if key in name:
            return key
    return None


def main(dirs):
    workload = None
    if dirs and dirs[0].startswith("--workload="):            # --workload=N,B,dtype : tags the summary so that bench.py can match it
        parts = dirs[0].split("=", 1)[1].split(",")            # N,B,dtype[,variant]   (variant: "gp" for the GP-residual workload)
        n, b, dt = parts[:3]
        workload = {"horizon": int(n), "batch": int(b), "dtype": dt}
        if len(parts) > 3 and parts[3]:
            workload["variant"] = parts[3]
        dirs = dirs[1:]
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> counter -> dispatch -> value
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out = {}
    for k, counters in acc.items():
        out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in sorted(counters.items())}
    # HBM traffic of one step = sum over the step's kernels, per launch: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # counts 32-byte requests as 16 for wide coalesced reads and is doubled (MI355X_MICROARCH.md, HBM section)
    step = [k for k in ("f20_order", "fused20", "seg_kernel", "linearize", "condense", "qp_dense", "expand", "rowqp") if k in out and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]]
    if step:
        # launches per step, relative to the kernel that runs exactly once per step: the fused N = 20 kernel, else the linearisation kernel
        # (a split batch launches the row kernel twice + its sort kernel, whose traffic -- two passes over B keys -- is not counted)
        once = "fused20" if "fused20" in step else ("seg_kernel" if "seg_kernel" in step else ("linearize" if "linearize" in step else None))
        per = {k: (out[k]["FETCH_SIZE"]["launches"] / out[once]["FETCH_SIZE"]["launches"] if once else 1.0) for k in step}
        fetch = sum(out[k]["FETCH_SIZE"]["mean_per_launch"] * per[k] for k in step) * 1024.0 * 2.0
        write = sum(out[k]["WRITE_SIZE"]["mean_per_launch"] * per[k] for k in step) * 1024.0
        out["_step_traffic"] = {"kernels": step, "launches_per_step": per, "fetch_bytes_corrected": fetch, "write_bytes": write, "bytes": fetch + write}
    if workload:
        out["_workload"] = workload
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
