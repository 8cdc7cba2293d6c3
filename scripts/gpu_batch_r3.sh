#!/bin/bash
# Round-3 evidence batch on the GPU box: profiles of the headline shapes, bench lines of the other configs, parity census.
set -x
bash scripts/profile_r3.sh r3_n20 20 4096 f64 > gpurun_out/r3_n20_profile.log 2>&1
bash scripts/profile_r3.sh r3_gp 20 4096 f64 --gp > gpurun_out/r3_gp_profile.log 2>&1
bash scripts/profile_r3.sh r3_n40 40 4096 f64 > gpurun_out/r3_n40_profile.log 2>&1
bash scripts/profile_r3.sh r3_cfg5 80 16384 f32 > gpurun_out/r3_cfg5_profile.log 2>&1
for a in "--batch-per-gpu 8192" "--dynamic" "--batch-per-gpu 16384" "--horizon 80 --batch-per-gpu 2048" "--horizon 40 --batch-per-gpu 16384"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $a > "gpurun_out/r3_bench_$(echo $a | tr -d ' -').json" 2>/dev/null
done
ADMPC_N20=split timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_split_pipeline.json 2>/dev/null
timeout -k 10 900 python3 scripts/gpu_parity_census.py 5 > gpurun_out/r3_census.txt 2>&1
timeout -k 10 300 python3 scripts/bench_quad.py 4096 > gpurun_out/r3_bench_quad.txt 2>&1
echo done
