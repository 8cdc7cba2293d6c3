import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from ad_mpc_amd.config import default_config, AdmpcConfig
from ad_mpc_amd.scenarios import random_scenarios
from scipy.stats import spearmanr
L = C.CDLL('/root/repo/oracle/liboracle.so')
dp = C.POINTER(C.c_double)
L.oracle_ipm_trace.argtypes = [C.POINTER(AdmpcConfig), dp, dp, dp, C.c_double, dp, dp, dp, C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80; B = 3000
cfg = default_config(N=N); s = random_scenarios(B, N=N, seed=1234)
P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
its = np.zeros(B, int); tr = np.zeros((B, 80, 2))
for b in range(B):
    t = np.zeros((80, 2))
    its[b] = L.oracle_ipm_trace(C.byref(cfg), P(s["x0"][b]), P(s["yref"][b]), P(s["yref_e"][b]), float(s["p"][b]), P(s["xbar"][b]), P(s["ubar"][b]), t.ctypes.data_as(dp), 80)
    tr[b] = t
m = its > 0
print("N %d: iterating %d of %d, mean its %.2f max %d" % (N, m.sum(), B, its[m].mean(), its.max()))
for k in (1, 2, 3, 4):
    sel = its > k
    mu_k = tr[sel, k, 0]; rem = its[sel] - k
    amin = tr[sel, :k, 1].min(1); aprod = np.prod(1 - tr[sel, :k, 1] + 1e-9, axis=1)
    print(" after %d its (%d still iterating): spearman(remaining, log mu_k) %.3f, (remaining, min alpha) %.3f, (remaining, prod(1-alpha)) %.3f" % (k, sel.sum(), spearmanr(rem, np.log(mu_k))[0], spearmanr(rem, amin)[0], spearmanr(rem, aprod)[0]))
np.save('/tmp/its_trace_N%d.npy' % N, np.concatenate([its[:, None], tr[:, :8, 0], tr[:, :8, 1]], axis=1))
