import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from oracle import actuation_oracle as ao
eng = BatchSolver(default_config(), device=0)
rng = np.random.default_rng(5)
B, H = 300, 21
vx = rng.uniform(0, 15, B); vy = rng.uniform(-1, 1, B); vref = rng.uniform(0, 20, (B, H))
t = eng.to_device(vref.copy())
eng.resample_vel(t, eng.to_device(vx), eng.to_device(vy), 5.0, 0.05)
torch.cuda.synchronize()
out = t.cpu().numpy()
exp = np.array([ao.resample_vel(vref[b], vx[b], vy[b], 5.0, 0.05) for b in range(B)])
bad = np.argwhere(out != exp)
print("mismatches", len(bad), "rows", len(set(bad[:, 0])))
for b, i in bad[:12]:
    print(b, i, repr(out[b, i]), repr(exp[b, i]), "in", repr(vref[b, i]), "diff ulps", (out[b, i] - exp[b, i]) / np.spacing(exp[b, i]))
b = bad[0][0]
import math
bd = math.sqrt(vx[b] * vx[b] + vy[b] * vy[b]); seq = []
for i in range(H): seq.append(bd); bd = bd + 5.0 * 0.05 * 0.8
print("row", b, "first mismatch col", bad[0][1], "bounds", [repr(x) for x in seq[:4]])
