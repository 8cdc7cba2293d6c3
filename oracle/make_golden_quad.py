"""Golden vectors of the quadrotor model from the reference's own compiled CasADi code (oracle/_ref/libquad_ref.so, built in place
by `make -C oracle ref`): (x, u) -> xdot, and the ERK4 step phi, A, B as acados integrates it (one step of h = 0.1 s).
    python oracle/make_golden_quad.py     (development container only; writes tests/golden/quad_shooting.json)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.quad_oracle import RefQuadModel  # noqa: E402

ref = RefQuadModel()
rng = np.random.default_rng(2024)
cases = []
for i in range(60):
    x = rng.standard_normal(13) * np.r_[2, 2, 2, 1, 1, 1, 1, 3, 3, 3, 2, 2, 2]
    x[3:7] /= np.linalg.norm(x[3:7])
    if i % 5 == 0:
        x[3:7] *= 1.05                       # off the unit sphere: the model is polynomial in q, the iterate need not be normalised
    u = rng.uniform(0, 1, 4)
    phi, A, B = ref.rk4_sens(x, u, 0.1)
    cases.append(dict(x=x.tolist(), u=u.tolist(), h=0.1, xdot=ref.ode(x, u).tolist(), phi=phi.tolist(), A=A.tolist(), B=B.tolist()))
with open(os.path.join(ROOT, "tests", "golden", "quad_shooting.json"), "w") as f:
    json.dump(dict(source="my_quad_expl_ode_fun.c / my_quad_expl_vde_forw.c of the reference, ERK4 one step", cases=cases), f)
print("wrote", len(cases), "cases")
