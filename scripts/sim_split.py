"""Analysis script (uses oracle/ -- not part of the product): what does the two-phase path of kernel R buy, and which sort key?
Lock-step wave model over the oracle's iteration counts of one bench batch:  python3 scripts/sim_split.py N B"""
import numpy as np, sys, heapq
sys.path.insert(0,'.')
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
N=int(sys.argv[1]); B=int(sys.argv[2])
o=Oracle(omp=True); cfg=default_config(N=N)
s=random_scenarios(B,N=N,seed=1234)
r=o.solve_batch(cfg,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=8)
it=r[4]
free=cfg.copy()
for j in range(2): free.lbu[j]=-1e9; free.ubu[j]=1e9
free.lbx_delta=-1e9; free.ubx_delta=1e9
f=o.solve_batch(free,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=8)
assert (f[4]==0).all()
x,u=f[0],f[1]
lbu=np.array(cfg.lbu[:]); ubu=np.array(cfg.ubu[:])
vu=np.maximum(np.maximum(lbu-u,u-ubu),0)/(ubu-lbu)          # [B,N,2]
d=x[:,1:,6]; vd=np.maximum(np.maximum(cfg.lbx_delta-d,d-cfg.ubx_delta),0)/(cfg.ubx_delta-cfg.lbx_delta)
feats={"max_vu":vu.max((1,2)),"max_vd":vd.max(1),"n_viol":(vu>0).sum((1,2))+(vd>0).sum(1),"sum_v":vu.sum((1,2))+vd.sum(1),
       "max_all":np.maximum(vu.max((1,2)),vd.max(1))}
print("iters mean %.2f max %d; solved by trial %.3f"%(it.mean(),it.max(),(it==0).mean()))
for k,v in feats.items():
    m=it>0
    print(k,"spearman-ish corr with iters (iterating instances):",np.corrcoef(np.argsort(np.argsort(v[m])),np.argsort(np.argsort(it[m])))[0,1])
W=1024; a,b,c,F=0.25,0.55,0.30,0.25
def lock(order, two_kernel=False):
    its=it[order]
    if two_kernel: its=its[its>0]
    nq=(len(its)+3)//4; pad=np.zeros(nq*4,int); pad[:len(its)]=its
    qm=pad.reshape(-1,4).max(1)
    h=[0.0]*min(W,nq); heapq.heapify(h)
    for q in qm:
        t=heapq.heappop(h); heapq.heappush(h,t+(a+b+c if not two_kernel else a+c)+q+F)   # two-kernel: second kernel redoes the roll-out and init, not the trial
    T=max(h)
    if two_kernel: T+= (B/4/W)*(a+b+0.1)+0.0          # first kernel: roll-out + trial + check for every quad, lock-step, no tail
    return T
base=lock(np.arange(B)); print("shipped (ticket order, one kernel): %.1f"%base)
print("two kernels, unsorted remainder: %.1f (%.2fx)"%(lock(np.arange(B),True),base/lock(np.arange(B),True)))
print("two kernels, perfect longest-first: %.2fx"%(base/lock(np.argsort(-it,kind='stable'),True)))
for k,v in feats.items():
    od=np.argsort(-v,kind='stable')
    print("two kernels, remainder sorted by %-8s: %.1f (%.2fx)"%(k,lock(od,True),base/lock(od,True)))
# lower bounds
work=(it[it>0].sum()/4+ (it>0).sum()/4*(a+c+F))/W
print("perfect packing bound (two kernels): %.1f (%.2fx)"%(work+(B/4/W)*(a+b+0.1),base/(work+(B/4/W)*(a+b+0.1))))
