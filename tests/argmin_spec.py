"""Table-driven specification of the scenario arg-min (BASELINE.json north_star): ONE table that every
implementation is checked against --

    device, second level   admpc_argmin_pairs   (tests/test_gpu_parity.py)          } both compiled from
    host,   second level   admpc_argmin_pairs_host (tests/test_dist_gloo.py, CPU)   } csrc/argmin_rule.h
    device, first level    admpc_argmin         (tests/test_gpu_parity.py; cases with consecutive indices)
    torch restatement      ad_mpc_amd.dist.pairs_min_torch (tests/test_dist_gloo.py, CPU)

Rules: a NaN cost is read as +inf and never beats a finite cost; the lower cost wins; equal costs (including -0.0 == +0.0 and
+inf == +inf) -> the lower global index wins; the winner's cost is reported as read (NaN -> +inf).

Each case: (name, [(cost, global index), ...], (expected cost, expected index)).
"""
import math

NAN, INF = float("nan"), float("inf")
BIG = 7 * 8192 + 8191          # last global index of config 4's last shard

CASES = [
    ("single record", [(3.25, 17)], (3.25, 17)),
    ("plain minimum, last", [(5.0, 0), (4.0, 1), (3.0, 2)], (3.0, 2)),
    ("plain minimum, first", [(1.0, 8192), (4.0, 16384), (3.0, 24576)], (1.0, 8192)),
    ("tie -> lowest index, listed first", [(1.5, 10), (1.5, 20), (2.0, 5)], (1.5, 10)),
    ("tie -> lowest index, listed last", [(1.5, 20), (2.0, 5), (1.5, 10)], (1.5, 10)),
    ("tie across all ranks", [(0.25, 8192 * r + 5) for r in (7, 3, 6, 0, 2, 5, 1, 4)], (0.25, 5)),
    ("NaN never wins", [(NAN, 0), (2.5, 8192)], (2.5, 8192)),
    ("NaN in the middle", [(7.0, 3), (NAN, 1), (6.0, 9)], (6.0, 9)),
    ("+inf loses against finite", [(INF, 0), (1e300, 1)], (1e300, 1)),
    ("all +inf -> lowest index, cost +inf", [(INF, 12), (INF, 4), (INF, 8)], (INF, 4)),
    ("all NaN -> read as +inf, lowest index", [(NAN, 9), (NAN, 2)], (INF, 2)),
    ("NaN ties with +inf by index", [(INF, 6), (NAN, 3)], (INF, 3)),
    ("-inf wins", [(0.0, 0), (-INF, 5), (-1e308, 2)], (-INF, 5)),
    ("negative costs", [(-1.0, 4), (-2.0, 6), (-2.0, 5)], (-2.0, 5)),
    ("-0.0 ties with +0.0", [(0.0, 9), (-0.0, 11)], (0.0, 9)),
    ("denormal beats zero only if smaller", [(5e-324, 1), (0.0, 2)], (0.0, 2)),
    ("index beyond 32 bits", [(1.0, (1 << 40) + 3), (1.0, (1 << 40) + 2), (2.0, 1)], (1.0, (1 << 40) + 2)),
    ("last instance of the last shard", [(9.0, 0), (8.0, BIG)], (8.0, BIG)),
    ("more records than a wave", [(100.0 - (i % 37), 3 * i) for i in range(150)], (64.0, 3 * 36)),
    ("more records than a wave, tie in the tail", [(1.0 if i in (70, 140) else 2.0, 1000 - i) for i in range(150)], (1.0, 860)),
]


def reference(records):
    """The rules, spelled out once more in plain Python (the table's expectations were written by hand; this guards the table)."""
    best = None
    for c, i in records:
        c = INF if math.isnan(c) else c
        if best is None or c < best[0] or (c == best[0] and i < best[1]):
            best = (c, i)
    return best


def consecutive(case):
    """Cases whose indices are offset + 0, 1, 2, ... in listing order can also be fed to the first-level admpc_argmin."""
    _, recs, _ = case
    off = recs[0][1]
    return all(i == off + k for k, (_, i) in enumerate(recs))


# first-level cases: cost arrays with an index offset (admpc_argmin); expected index = offset + position
ARRAY_CASES = [
    ("array: plain", [4.0, 2.0, 3.0], 100, (2.0, 101)),
    ("array: tie -> first position", [2.0, 1.0, 1.0, 5.0], 8192, (1.0, 8193)),
    ("array: NaN skipped", [NAN, NAN, 3.0, NAN], 0, (3.0, 2)),
    ("array: all NaN", [NAN, NAN], 16, (INF, 16)),
    ("array: all +inf (no valid candidate)", [INF] * 5, 24576, (INF, 24576)),
    ("array: longer than the block", [1000.0 - (i % 613) for i in range(3000)], 7 * 8192, (388.0, 7 * 8192 + 612)),
    ("array: -0.0 and 0.0", [0.0, -0.0], 4, (0.0, 4)),
]
