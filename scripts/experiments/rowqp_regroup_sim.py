import numpy as np, heapq
d = np.load('/tmp/its_trace_N80.npy'); its = d[:, 0].astype(int); mu = d[:, 1:9]; al = d[:, 9:17]
rng = np.random.default_rng(1)
idx = rng.choice(np.nonzero(its > 0)[0], 9490)          # config[4]: about 9490 deferred instances
c = its[idx]; MU = mu[idx]; W = 1024
def makespan(g):
    h = [0.0] * W; heapq.heapify(h)
    for x in g: heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)
def quads(o):
    pad = (-len(o)) % 4
    return np.concatenate([o, np.zeros(pad, dtype=o.dtype)]).reshape(-1, 4)
base = makespan(quads(rng.permutation(c)).max(1)); ideal = makespan(quads(np.sort(c)[::-1]).max(1))
print("one launch, random grouping %.1f; sorted by true count %.1f; lower bound %.1f" % (base, ideal, c.sum() / 4 / W))
for k in (1, 2, 3, 4):
    first = makespan(np.full(len(quads(c)), float(k)))
    live = c > k
    key = np.log(MU[live, k]); rem = (c[live] - k)
    order = rem[np.argsort(-key)]
    second = makespan(quads(order).max(1))
    second_ideal = makespan(quads(np.sort(rem)[::-1]).max(1))
    print("k = %d: first launch %.1f + second (sorted by log mu_k) %.1f = %.1f  (x%.2f);  with the true remaining count as key %.1f" % (k, first, second, first + second, base / (first + second), first + second_ideal))
