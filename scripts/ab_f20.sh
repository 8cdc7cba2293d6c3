#!/bin/bash
# A/B of builds of the library on the default bench: scripts/ab_f20.sh libadmpc_X.so libadmpc_Y.so ...   (ms per step, one batch in flight / two)
for lib in "$@"; do
  timeout -k 10 120 python scripts/bench_lib.py $lib --steps 30 --warmup 5 --no-cpu-baseline --no-tight-stop --no-live-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s kernel %.4f ms  step %.4f ms  %.2f M/s   two-in-flight %.4f ms' % ('$lib', d['roofline']['kernel_ms'], d['ms_per_step'], d['value']/1e6, d['two_in_flight']['ms_per_step']))"
done
