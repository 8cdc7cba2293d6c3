"""Two builds of the library on the same fp32 batch: how many instances differ, and by how much?  python3 scripts/experiments/f32_cmp_libs.py libA libB [seed]
(each library in a child process: one process cannot load two builds)"""
import os, sys, subprocess, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    import ad_mpc_amd._lib as _lib
    _lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", sys.argv[2])
    from ad_mpc_amd.config import default_config
    from ad_mpc_amd.engine import BatchSolver
    from ad_mpc_amd.scenarios import random_scenarios
    N, B = 80, 16384
    s = random_scenarios(B, N=N, seed=int(sys.argv[3]), blend=(3.0, 5.0))
    g = BatchSolver(default_config(N=N), device=0).solve_numpy(s["x0"], s["yref"], s["yref_e"], s["p"], s["xbar"], s["ubar"], dtype=np.float32)
    pickle.dump(g, open(sys.argv[4], "wb")); sys.exit(0)
seed = sys.argv[3] if len(sys.argv) > 3 else "110"
out = []
for i, lib in enumerate(sys.argv[1:3]):
    f = "/tmp/f32_cmp_%d.pkl" % i
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, seed, f], check=True, stderr=subprocess.DEVNULL)
    out.append(pickle.load(open(f, "rb")))
a, b = out
d = np.abs(a[1].astype(np.float64) - b[1]).max(axis=(1, 2))
print("seed %s: instances with any different input: %d of %d; different iteration counts: %d; |du_A - du_B| quantiles 50/90/99/max of the differing ones: %s"
      % (seed, (d > 0).sum(), len(d), (a[4] != b[4]).sum(), np.quantile(d[d > 0], [0.5, 0.9, 0.99, 1.0]) if (d > 0).any() else "-"))
w = np.argsort(d)[-5:][::-1]
for i in w: print("  instance %d: |du_A - du_B| %.3e, iterations %d / %d, status %d / %d" % (i, d[i], a[4][i], b[4][i], a[3][i], b[3][i]))
