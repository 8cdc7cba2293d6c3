import sys, numpy as np, heapq
sys.path.insert(0, '/root/repo')
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
N, B, W = 80, 16384, 1024
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234)
o = Oracle(omp=True) if 'omp' in Oracle.__init__.__code__.co_varnames else Oracle()
r = o.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
it = r[4]
print("iters mean %.2f max %d, zero frac %.3f" % (it.mean(), it.max(), (it == 0).mean()))
c = it[it > 0].astype(int)
rng = np.random.default_rng(0)
def makespan(groups_cost):          # list scheduling of wave jobs on W slots in the given order
    h = [0.0] * W; heapq.heapify(h)
    for g in groups_cost:
        t = heapq.heappop(h); heapq.heappush(h, t + g)
    return max(h)
def quads(order):
    k = len(order); pad = (-k) % 4
    a = np.concatenate([order, np.zeros(pad, dtype=int)]).reshape(-1, 4)
    return a
base = makespan(quads(rng.permutation(c)).max(1))
ideal = makespan(quads(np.sort(c)[::-1]).max(1))
print("launch-2 makespan (iteration units): random grouping %.1f, sorted by true count %.1f (x%.2f), lower bound sum/4/W %.1f" % (base, ideal, base / ideal, c.sum() / 4 / W))
for noise in (0.0, 0.3):
  for E in (2, 3, 4, 6):
    rem = c.copy().astype(float); total = 0.0; first = True; nep = 0
    while (rem > 0).any():
        live = rem[rem > 0]
        key = live if not first else rng.permutation(live)               # first epoch: no knowledge; later: remaining (with noise)
        if not first and noise > 0: key = live * np.exp(noise * rng.standard_normal(len(live)))
        order = live[np.argsort(-key)] if not first else key
        q = quads(order)
        total += makespan(np.minimum(q.max(1), E)); nep += 1
        rem_new = np.maximum(order - E, 0); rem = rem_new; first = False
    print("epochs of %d iterations, key noise %.1f: %.1f units in %d epochs (x%.2f vs random one-launch)" % (E, noise, total, nep, base / total))
