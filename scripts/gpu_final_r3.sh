#!/bin/bash
# Final evidence batch of round 3 on the GPU box: the whole GPU suite, smoke, the parity census, the default bench line as the driver runs it
# (live PMC traffic, CPU baseline), the other configs, the launched one-rank path, a refreshed profile of the headline shape.
set -x
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -3 $out/gputest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout -k 10 300 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
for a in "--gp" "--dynamic" "--batch-per-gpu 8192" "--batch-per-gpu 16384" "--horizon 40" "--horizon 40 --batch-per-gpu 16384" "--dtype f32 --horizon 80 --batch-per-gpu 16384"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $a > "$out/bench_$(echo $a | tr -d ' -').json" 2>/dev/null
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_launched_1rank.json 2> $out/bench_launched.err
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --batch-per-gpu 8192 > $out/bench_launched_1rank_b8192.json 2>> $out/bench_launched.err
ADMPC_N20=split timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $out/bench_split_pipeline.json 2>/dev/null
timeout -k 10 300 python3 scripts/bench_quad.py 4096 > $out/bench_quad.txt 2>&1
timeout -k 10 900 python3 scripts/gpu_parity_census.py 5 > $out/census.txt 2>&1
timeout -k 10 900 python3 scripts/gpu_parity_census.py 3 tight > $out/census_tight.txt 2>&1
bash scripts/profile_r3.sh final/n20 20 4096 f64 > $out/n20_profile.log 2>&1
python3 - <<'PY' > gpurun_out/final/gp40_diag.txt 2>&1
# the one census family above the test tolerance: N = 40 with the GP residual -- where does max |dx| come from?
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config, set_gp
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios, grid_gp
cfg = default_config(N=40); set_gp(cfg, grid_gp())
eng = BatchSolver(cfg, device=0); o = Oracle(omp=True)
for seed in range(100, 105):
    s = random_scenarios(4096, N=40, seed=seed)
    g = eng.solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"])
    r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=32)
    dxi = np.abs(g[0] - r[0]).max(axis=(1, 2)); dui = np.abs(g[1] - r[1]).max(axis=(1, 2))
    b = int(np.argmax(dxi))
    print("seed %d: worst instance %d |dx| %.2e |du| %.2e iters %d/%d status %d; max |x| of it %.2e, max |u| %.2e; instances with |dx| > 1e-7: %d; its state component of the worst entry: %s"
          % (seed, b, dxi[b], dui[b], g[4][b], r[4][b], g[3][b], np.abs(r[0][b]).max(), np.abs(r[1][b]).max(), int((dxi > 1e-7).sum()),
             np.unravel_index(np.argmax(np.abs(g[0][b] - r[0][b])), g[0][b].shape)))
PY
echo done
