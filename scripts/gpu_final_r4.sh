#!/bin/bash
# Evidence batch of round 4 on the GPU box: the whole GPU suite, smoke, the default bench line as the driver runs it, and for EVERY workload
# that is benched a profile of the same command (kernel trace + PMC passes) -- scripts/profile_r4.sh -- whose three files go to profiles/r4/.
set -x
out=gpurun_out/final4; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
bash scripts/profile_r4.sh final4/n20 20 4096 f64 > $out/n20_profile.log 2>&1
bash scripts/profile_r4.sh final4/n40 40 4096 f64 > $out/n40_profile.log 2>&1
bash scripts/profile_r4.sh final4/gp 20 4096 f64 --gp > $out/gp_profile.log 2>&1
bash scripts/profile_r4.sh final4/cfg5 80 16384 f32 > $out/cfg5_profile.log 2>&1
bash scripts/profile_r4.sh final4/b8192 20 8192 f64 > $out/b8192_profile.log 2>&1
for t in n20 n40 gp cfg5 b8192; do for f in bench.json kernel_stats.csv pmc_summary.json; do cp $out/$t/$f $out/${t}_$f; done; done
for a in "--dynamic" "--batch-per-gpu 16384" "--horizon 40 --batch-per-gpu 16384" "--horizon 40 --dynamic"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-live-traffic $a > "$out/bench_$(echo $a | tr -d ' -').json" 2>/dev/null
done
echo done
