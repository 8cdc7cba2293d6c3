#!/bin/bash
# per-kernel average durations with the GP residual active: scripts/kstats_gp.sh VARIANT...
export TMPDIR=/tmp
cp ad_mpc_amd/libadmpc.so /tmp/libadmpc_keep.so
for v in "$@"; do
  cp ad_mpc_amd/libadmpc_$v.so ad_mpc_amd/libadmpc.so
  rm -rf gpurun_out/ksg_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksg_$v -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --gp > /dev/null 2>&1
  python3 - <<PY
import csv,glob
for f in glob.glob('gpurun_out/ksg_$v/*/*kernel_stats.csv'):
    print('$v', ' | '.join('%s %.1f us' % (r['Name'].split('admpc_')[-1][:18], float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f)) if 'admpc' in r['Name']))
PY
done
cp /tmp/libadmpc_keep.so ad_mpc_amd/libadmpc.so
