#!/bin/bash
for v in MFMA VALU; do
  cp ad_mpc_amd/libadmpc_$v.so ad_mpc_amd/libadmpc.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['mean_ipm_iters'], d['status_nonzero'])"
done
cp ad_mpc_amd/libadmpc_MFMA.so ad_mpc_amd/libadmpc.so
export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_a.log 2>&1; cat gpurun_out/prof_a/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-70,150-
