#!/bin/bash
# A/B of kernel build variants on the GPU box: scripts/ablate.sh VARIANT...   (expects ad_mpc_amd/libadmpc_<VARIANT>.so, git-ignored)
cp ad_mpc_amd/libadmpc.so /tmp/libadmpc_keep.so
for v in "$@"; do
  cp ad_mpc_amd/libadmpc_$v.so ad_mpc_amd/libadmpc.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['value']), d['mean_ipm_iters'], d['status_nonzero'])"
done
cp /tmp/libadmpc_keep.so ad_mpc_amd/libadmpc.so
