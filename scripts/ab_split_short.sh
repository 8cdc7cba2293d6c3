# A/B of the two-phase path on short horizons and fp32, where LDS would allow more than one wave per SIMD but registers do not: bash scripts/ab_split_short.sh
for cfg in "40 8192 f32" "40 16384 f32" "20 8192 f32" "24 8192 f64" "24 16384 f64" "40 4096 f32"; do
  set -- $cfg
  for m in 0 1 x; do
    if [ $m = x ]; then unset ADMPC_ROWQP_SPLIT; else export ADMPC_ROWQP_SPLIT=$m; fi
    timeout -k 10 200 python3 scripts/run_rowqp.py $1 $2 8 $3 2>&1 | grep "^N " | sed "s/^/split=$m $3 /"
  done
done
