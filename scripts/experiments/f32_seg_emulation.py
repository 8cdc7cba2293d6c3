import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import seg_spec
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
orc = Oracle(omp=True)
N = 80; B = int(sys.argv[1]) if len(sys.argv) > 1 else 60; NS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234)
o64 = orc.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
# fp32 stop levels of the device path (rq_make_params): comp 1e-3, res 1e-2, step 1e-3
c32 = cfg.copy(); c32.ipm_tol_comp = 1e-3; c32.ipm_tol_res = 1e-2; c32.ipm_tol_step = 1e-3
seg_spec.IPM_FLOOR = 1e-8
seg_spec.MU_FLOOR = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # kernel R in fp32 has no centring floor (rowqp_core.h, rq_make_params)
MODE = sys.argv[4] if len(sys.argv) > 4 else 'f32'
F64 = MODE == 'f64'      # control: fp64 arithmetic at the fp32 stop levels
# MODE 'mixed': problem data and condensed blocks rounded to float32 (factorisations, substitutions, products with them run in float32), numpy's own
# promotion elsewhere (the vectors of the interior point that pass through a float64 array stay float64); 'f32': every array float32
f32 = np.float64 if F64 else np.float32
class Q32(seg_spec.SegQP):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for nm in ("Qd", "Rd", "Qe", "A", "B", "b", "q", "r", "dlu", "duu", "dld", "dud", "dx0"):
            setattr(self, nm, getattr(self, nm).astype(f32))
        self.h = f32(self.h); self.rho_l = f32(self.rho_l); self.rho_u = f32(self.rho_u)
        self.segs = [self._condense(t) for t in range(self.S)]
        for sg in self.segs:
            for k2, v in sg.items():
                if isinstance(v, np.ndarray): sg[k2] = v.astype(f32)
class _NP32:
    """numpy with float32 as the default floating type, so that no float64 temporary leaks into the emulated arithmetic"""
    def __getattr__(self, k): return getattr(np, k)
    @staticmethod
    def _d(k): return k.setdefault("dtype", f32)
    def zeros(self, *a, **k): self._d(k); return np.zeros(*a, **k)
    def ones(self, *a, **k): self._d(k); return np.ones(*a, **k)
    def empty(self, *a, **k): self._d(k); return np.empty(*a, **k)
    def eye(self, *a, **k): self._d(k); return np.eye(*a, **k)
    def full(self, *a, **k): self._d(k); return np.full(*a, **k)
    def array(self, x, *a, **k):
        r = np.array(x, *a, **k); return r.astype(f32) if r.dtype == np.float64 else r
    def asarray(self, x, *a, **k):
        r = np.asarray(x, *a, **k); return r.astype(f32) if r.dtype == np.float64 else r
    def concatenate(self, xs, *a, **k):
        r = np.concatenate(xs, *a, **k); return r.astype(f32) if r.dtype == np.float64 else r
if MODE == 'f32': seg_spec.np = _NP32()
err = []; its = []; fails = 0; leaks = 0
for i in range(B):
    q = orc.qp_debug(cfg, s['x0'][i], s['yref'][i], s['yref_e'][i], s['p'][i], s['xbar'][i], s['ubar'][i])
    qp = Q32(cfg, q["A"], q["B"], q["b"], s["x0"][i], s["yref"][i], s["yref_e"][i], s["xbar"][i], s["ubar"][i], Ns=NS)
    try:
        res, it = seg_spec.seg_ipm(c32, qp)
    except Exception as e:
        fails += 1; print("instance", i, "exception", repr(e)[:80]); continue
    if res is None: fails += 1; continue
    du, dx = res
    leaks += int(du.dtype != f32)
    e = np.abs(du.astype(np.float64) - q['du']).max()
    err.append(e); its.append(it)
err = np.array(err)
print("float64 results (leaks):", leaks)
print(MODE, "mu floor", seg_spec.MU_FLOOR, "Ns", NS); print("fp32 segmented (numpy emulation): B %d, failures %d, iterations mean %.2f max %d (fp64 oracle mean %.2f); |du - du64| quantiles 50/90/99/max: %.2e %.2e %.2e %.2e" % (B, fails, np.mean(its), max(its), o64[4].mean(), *np.quantile(err, [0.5, 0.9, 0.99, 1.0])))
