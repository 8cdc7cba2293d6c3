#!/bin/bash
# Same-box A/B of builds of the library (ad_mpc_amd/libadmpc*.so) on short bench lines without side measurements but two_in_flight.
#   bash scripts/quick_ab.sh <tag> <test|notest> "<workload args>;<workload args>;..." lib1.so lib2.so ...
# `test` runs the GPU suite (against libadmpc.so) first.  Every (workload, library) pair is benched twice, interleaved, best kept.
tag=$1; mode=$2; IFS=';' read -ra WL <<< "$3"; shift 3
out=gpurun_out/$tag; mkdir -p $out
if [ "$mode" = test ]; then timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $out/gputest.log 2>&1; tail -3 $out/gputest.log; fi
F="--steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --no-tight-stop"
for a in "${WL[@]}"; do
  for rep in 1 2; do
    for lib in "$@"; do
      f="$out/b_$(echo $a | tr -d ' -')_${lib%.so}_$rep.json"
      timeout -k 10 200 python3 scripts/bench_lib.py $lib $F $a > $f 2>/dev/null
    done
  done
  for lib in "$@"; do
    python3 - "$out/b_$(echo $a | tr -d ' -')_${lib%.so}" "$a" $lib <<'P'
import json, sys
best = None
for rep in (1, 2):
    try: d = json.loads(open(sys.argv[1] + "_%d.json" % rep).read().strip().splitlines()[-1])
    except Exception: continue
    if best is None or d["value"] > best["value"]: best = d
if best is None: print("%-40s %-22s FAILED" % (sys.argv[2] or "default", sys.argv[3])); sys.exit(0)
t = best.get("two_in_flight", {})
print("%-40s %-22s value %.3f M  ms %.4f  two %.3f M  iters %.3f" % (sys.argv[2] or "default", sys.argv[3], best["value"] / 1e6, best["ms_per_step"], t.get("solves_per_s", 0) / 1e6, best.get("mean_ipm_iters", 0)))
P
  done
done
