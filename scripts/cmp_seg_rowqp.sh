#!/bin/bash
# kernel S (ADMPC_QP=seg) against kernel R (ADMPC_QP=riccati) at the horizons where both exist:  scripts/cmp_seg_rowqp.sh [horizons] [batches]
mkdir -p gpurun_out
for H in ${1:-60 80}; do for B in ${2:-4096 16384}; do
for Q in seg riccati; do
ADMPC_QP=$Q timeout -k 10 300 python bench.py --horizon $H --batch-per-gpu $B --no-live-traffic --no-cpu-baseline --no-tight-stop --no-two-in-flight > gpurun_out/cmp_${H}_${B}_$Q.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/cmp_${H}_${B}_$Q.json')); print('N $H B $B $Q', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],3), 'ms')"
done; done; done
