#!/bin/bash
# HBM traffic of one step of the default bench with another build of the library: scripts/pmc_traffic.sh libadmpc_X.so [bench args]
# (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
lib=$1; shift; export TMPDIR=/tmp
out=gpurun_out/traffic_$lib; rm -rf $out; mkdir -p $out
ARGS="scripts/bench_lib.py $lib --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-tight-stop --no-live-traffic $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $ARGS > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -- python3 $ARGS > $out/pw.log 2>&1
python3 scripts/pmc_summary.py $out/pf $out/pw | python3 -c "
import json,sys
d=json.load(sys.stdin); t=d.get('_step_traffic')
print('$lib', {k: {c: round(v['mean_per_launch']) for c, v in d[k].items()} for k in d if not k.startswith('_')}, 'step MB: fetch(x2) %.1f write %.1f total %.1f' % (t['fetch_bytes_corrected']/1e6, t['write_bytes']/1e6, t['bytes']/1e6) if t else None)"
rm -rf $out
