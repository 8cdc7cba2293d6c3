import sys, numpy as np
sys.path.insert(0,'/root/repo')
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
o=Oracle(omp=True)
cfg=default_config(N=20)
eng=BatchSolver(cfg,device=0)
for B,seed in ((2049,2149),(2500,2600),(4096,1234),(4096,7)):
    s=random_scenarios(B,N=20,seed=seed,blend=(3.0,5.0) if seed!=1234 else (100.0,110.0))
    g=eng.solve_numpy(s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"])
    r=o.solve_batch(cfg,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=16)
    d=np.abs(g[1]-r[1]).reshape(B,-1).max(1)
    i=np.argsort(-d)[:3]
    print(B,seed,'max|du| %.2e'%d.max(),'worst',i,d[i],'iters',g[4][i],r[4][i], 'quantiles 99%% %.1e 99.9%% %.1e'%(np.quantile(d,0.99),np.quantile(d,0.999)))
