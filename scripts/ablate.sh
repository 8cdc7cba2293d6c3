#!/bin/bash
# timing-only ablation: duplicate one phase of the dense QP kernel and report the step time
for v in BASE DUP_CHOL DUP_SUBST; do
  cp ad_mpc_amd/libadmpc_$v.so ad_mpc_amd/libadmpc.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['mean_ipm_iters'], d['status_nonzero'])"
done
cp ad_mpc_amd/libadmpc_BASE.so ad_mpc_amd/libadmpc.so
