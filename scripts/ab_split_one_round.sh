# A/B of the two-phase path of kernel R at about one round of waves (where the default threshold sits): bash scripts/ab_split_one_round.sh
for cfg in "40 3000 f64" "40 4096 f64" "40 5000 f64" "80 4096 f32" "80 4096 f64" "64 4096 f64" "24 4096 f64"; do
  set -- $cfg
  for m in 0 1; do
    ADMPC_ROWQP_SPLIT=$m timeout -k 10 200 python3 scripts/run_rowqp.py $1 $2 8 $3 2>&1 | grep "^N " | sed "s/^/split=$m $3 /"
  done
done
