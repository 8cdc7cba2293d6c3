import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import seg_spec
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
orc = Oracle(omp=True)
N = 80; B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = default_config(N=N)
s = random_scenarios(B, N=N, seed=1234)
o64 = orc.solve_batch(cfg, s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], nthreads=8)
# fp32 stop levels of the device path (rq_make_params): comp 1e-3, res 1e-2, step 1e-3
c32 = cfg.copy(); c32.ipm_tol_comp = 1e-3; c32.ipm_tol_res = 1e-2; c32.ipm_tol_step = 1e-3
seg_spec.IPM_FLOOR = 1e-8
f32 = np.float32
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
err = []; its = []; fails = 0
for i in range(B):
    q = orc.qp_debug(cfg, s['x0'][i], s['yref'][i], s['yref_e'][i], s['p'][i], s['xbar'][i], s['ubar'][i])
    qp = Q32(cfg, q['A'], q['B'], q['b'], s['x0'][i], s['yref'][i], s['yref_e'][i], s['xbar'][i], s['ubar'][i])
    try:
        res, it = seg_spec.seg_ipm(c32, qp)
    except Exception as e:
        fails += 1; print("instance", i, "exception", repr(e)[:80]); continue
    if res is None: fails += 1; continue
    du, dx = res
    e = np.abs(du.astype(np.float64) - q['du']).max()
    err.append(e); its.append(it)
err = np.array(err)
print("fp32 segmented (numpy emulation): B %d, failures %d, iterations mean %.2f max %d (fp64 oracle mean %.2f); |du - du64| quantiles 50/90/99/max: %.2e %.2e %.2e %.2e" % (B, fails, np.mean(its), max(its), o64[4].mean(), *np.quantile(err, [0.5, 0.9, 0.99, 1.0])))
