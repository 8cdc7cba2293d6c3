# A/B of kernel R with and without the two-phase path (ADMPC_ROWQP_SPLIT) over batch sizes: bash scripts/ab_split.sh
set -e
mkdir -p gpurun_out/split
for cfg in "40 4096 f64" "40 8192 f64" "40 16384 f64" "80 16384 f32" "80 8192 f32" "80 2048 f64" "80 8192 f64"; do
  set -- $cfg
  for m in 0 1; do
    ADMPC_ROWQP_SPLIT=$m timeout -k 10 200 python3 scripts/run_rowqp.py $1 $2 5 $3 2>&1 | grep "^N " | sed "s/^/split=$m $3 /"
  done
done
