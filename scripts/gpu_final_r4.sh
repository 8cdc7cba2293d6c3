#!/bin/bash
# Evidence batch of round 4 on the GPU box: the whole GPU suite, smoke, the default bench line as the driver runs it, and for EVERY workload
# that is benched a profile of the same command (kernel trace + PMC passes) -- scripts/profile_r4.sh -- whose three files go to profiles/r4/.
# Two parts (a gpurun call is at most 20 minutes):  bash scripts/gpu_final_r4.sh 1 | 2
set -x
out=gpurun_out/final4; mkdir -p $out; export TMPDIR=/tmp
prof() { t=$1; shift; bash scripts/profile_r4.sh final4/$t "$@" > $out/${t}_profile.log 2>&1; for f in bench.json kernel_stats.csv pmc_summary.json; do cp $out/$t/$f $out/${t}_$f; done; }
if [ "${1:-1}" = 1 ]; then
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
prof n20 20 4096 f64
prof n40 40 4096 f64
prof n60 60 4096 f64
prof n80 80 4096 f64
else
prof gp 20 4096 f64 --gp
prof cfg5 80 16384 f32
prof b8192 20 8192 f64
for a in "--dynamic" "--batch-per-gpu 16384" "--horizon 40 --batch-per-gpu 16384" "--horizon 40 --dynamic" "--horizon 40 --gp" "--horizon 80 --batch-per-gpu 16384"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-live-traffic $a > "$out/bench_$(echo $a | tr -d ' -').json" 2>/dev/null
done
timeout -k 10 300 python3 scripts/bench_quad.py 4096 20 > $out/bench_quad_n20.txt 2>&1
timeout -k 10 300 python3 scripts/bench_quad.py 4096 10 > $out/bench_quad_n10.txt 2>&1
fi
echo done
