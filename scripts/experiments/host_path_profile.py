"""Where the 0.58 ms of the reference-shaped single-vehicle path go: cProfile over 400 calls of AD3DMPC.set_reference + optimize."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from ad_mpc_amd.ad_3d import AD3D
from ad_mpc_amd.ad_3d_mpc import AD3DMPC
from ad_mpc_amd.scenarios import straight_scenario
car = AD3D(); mpc = AD3DMPC(car, t_horizon=1.0, n_nodes=20)
x0, xref, uref = straight_scenario(N=20, Ts=0.05, v=5.0)
car.set_state(list(x0))
def once():
    mpc.set_reference(xref, uref)
    return mpc.optimize(use_model=0, return_x=True)
for _ in range(50): once()
pr = cProfile.Profile(); pr.enable()
for _ in range(400): once()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
