# usage (repository root): patch -p0 -o /tmp/exp_oracle.c < scripts/experiments/finishing_step.patch && cp oracle/admpc_oracle.h /tmp/ && sed -i 's#"../include/admpc.h"#"'$PWD'/include/admpc.h"#' /tmp/exp_oracle.c /tmp/admpc_oracle.h && gcc -O3 -fPIC -std=gnu11 -fopenmp -shared -o /tmp/libexp.so /tmp/exp_oracle.c -lm
#        FIN_MU=1e-3 FIN_M=1e10 FIN_EPS=1e-8 python3 scripts/experiments/finishing_step_run.py N B [instance indices]
import numpy as np, sys, os, ctypes as C
sys.path.insert(0,'.')
import oracle.oracle as oo
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
class Exp(oo.Oracle):
    def __init__(self):
        self.lib=C.CDLL(os.path.abspath('/tmp/libexp.so')); L=self.lib; cp=C.POINTER(oo.AdmpcConfig); _dp=oo._dp; _ip=oo._ip
        L.oracle_solve_batch.argtypes=[cp,C.c_int,_dp,_dp,_dp,_dp,_dp,_dp,_dp,_ip,_ip,C.c_int]; L.oracle_solve_batch.restype=C.c_int
        L.oracle_max_threads.restype=C.c_int
N=int(sys.argv[1]); B=int(sys.argv[2]); idx=[int(a) for a in sys.argv[3:]]
cfg=default_config(N=N); s=random_scenarios(B,N=N,seed=1234)
if idx: s={k:v[idx] for k,v in s.items()}
ref=oo.Oracle(omp=True).solve_batch(cfg,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=8)
r=Exp().solve_batch(cfg,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=1 if (idx or os.environ.get("FIN_DBG")) else 8)
print("ref  hist",np.bincount(ref[4]),"mean %.2f"%ref[4].mean())
print("exp  hist",np.bincount(r[4]),"mean %.2f"%r[4].mean(), "status!=0",(r[3]!=0).sum())
print("max|du| exp-ref %.2e  max|dx| %.2e"%(np.abs(r[1]-ref[1]).max(),np.abs(r[0]-ref[0]).max()))
top=np.argsort(-r[4])[:12]; print("slowest exp:",[(int(i),int(r[4][i]),int(ref[4][i])) for i in top])
top=np.argsort(-ref[4])[:12]; print("slowest ref:",[(int(i),int(ref[4][i]),int(r[4][i])) for i in top])
