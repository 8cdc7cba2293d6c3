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
    for key in ("linearize", "condense", "qp_dense", "qp_kernel", "argmin", "epilogue", "waypoints", "shoot"):
        if key in name:
            return key
    return None


def main(dirs):
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
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
