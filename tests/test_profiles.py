"""Every committed profile must reproduce its bench line.  For each profiles/rN/X_kernel_stats.csv (N >= 4) with
a sibling X_bench.json -- the two files scripts/profile_r4.sh writes from ONE command line -- the kernels of a step cannot take longer than
the step (sum over the admpc kernels of average duration x launches per step <= 1.03 x ms_per_step), and the HBM traffic of the PMC
summary X_pmc_summary.json is the traffic the bench line measured live (within 10 %).  Rounds 1-3 are history: two of their summaries were
taken at other stop levels than their bench lines and are superseded by profiles/r4/."""
import csv
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sets():
    out = []
    for ks in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_kernel_stats.csv"))):
        m = re.search(r"profiles[/\\]r(\d+)[/\\](.+)_kernel_stats\.csv$", ks)
        if not m or int(m.group(1)) < 4:
            continue
        bench = ks.replace("_kernel_stats.csv", "_bench.json")
        if os.path.exists(bench):
            out.append((ks, bench, ks.replace("_kernel_stats.csv", "_pmc_summary.json")))
    return out


def _kernels(ks):
    rows = [r for r in csv.DictReader(open(ks)) if "admpc_" in r["Name"]]
    calls = {r["Name"]: int(r["Calls"]) for r in rows}
    # the kernel that runs exactly once per step: the work-order pre-pass of the persistent kernels, else the linearisation kernel
    once = [n for n in calls if "f20_order" in n] or [n for n in calls if "linearize" in n]
    assert once, "no once-per-step kernel in " + ks
    steps = calls[once[0]]
    return [(r["Name"], float(r["AverageNs"]), int(r["Calls"]) / steps) for r in rows]


@pytest.mark.parametrize("ks,bench,pmc", _sets() or [pytest.param(None, None, None, marks=pytest.mark.skip(reason="no round >= 4 profile committed yet"))])
def test_profile_reproduces_its_bench_line(ks, bench, pmc):
    b = json.load(open(bench))
    step_ms = b["ms_per_step"]
    kern_ms = sum(avg * per for (_, avg, per) in _kernels(ks)) / 1e6
    assert kern_ms <= 1.03 * step_ms, "%s: kernels of a step %.4f ms > 1.03 x step %.4f ms" % (os.path.basename(ks), kern_ms, step_ms)
    assert kern_ms >= 0.80 * step_ms, "%s: kernels of a step %.4f ms explain less than 80 %% of the step %.4f ms" % (os.path.basename(ks), kern_ms, step_ms)
    live = b["roofline"].get("traffic"); src = b["roofline"].get("traffic_source") or ""
    if os.path.exists(pmc) and live and src.startswith("live"):
        t = json.load(open(pmc))["_step_traffic"]["bytes"]
        assert abs(t - live) <= 0.10 * live, "%s: PMC summary %.0f B vs live %.0f B" % (os.path.basename(pmc), t, live)
