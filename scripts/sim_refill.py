"""What would row-level refill or a two-kernel split buy kernel R?  Event simulation over the oracle's iteration counts of one
bench batch (1024 waves of four rows, ticket-drawn) with the measured phase costs in units of one interior-point iteration:
    python scripts/sim_refill.py N B      (uses oracle/ -- analysis script, not part of the product)
"""
import numpy as np, sys, heapq
sys.path.insert(0,'.')
from oracle.oracle import Oracle
from ad_mpc_amd.config import default_config
from ad_mpc_amd.scenarios import random_scenarios
N=int(sys.argv[1]); B=int(sys.argv[2])
o=Oracle(omp=True); cfg=default_config(N=N)
s=random_scenarios(B,N=N,seed=1234)
it=o.solve_batch(cfg,s["x0"],s["yref"],s["yref_e"],s["p"],s["xbar"],s["ubar"],nthreads=8)[4]
W=1024
# costs in units of one IPM iteration (M): S0+setup a, trial sweeps (S1+S4) b, check/init/E1 c, finish F
a,b,c,F=0.25,0.55,0.30,0.25
def lockstep():
    # waves draw quads by ticket; time per quad = a+b+c + max_its + F
    t=np.zeros(W); h=[(0.0,w) for w in range(W)]; heapq.heapify(h); q=0; nq=(B+3)//4
    while q<nq:
        tw,w=heapq.heappop(h); its=it[4*q:4*q+4].max(); q+=1
        heapq.heappush(h,(tw+a+b+c+its+F,w))
    return max(x[0] for x in h)
def refill(thresh):
    # per wave: rows with states; global ticket of instances; slot loop
    nxt=[0]
    def draw():
        if nxt[0]<B: nxt[0]+=1; return nxt[0]-1
        return None
    # event-driven per wave is complex because of the shared ticket: approximate by processing waves in time order slot by slot
    rows=[[None]*4 for _ in range(W)]   # remaining its (>=0) or None idle; state: ('new',inst)/('iter',rem)/('done',)
    h=[(0.0,w) for w in range(W)]; heapq.heapify(h); tend=0
    st=[[('empty',)]*4 for _ in range(W)]
    while h:
        tw,w=heapq.heappop(h)
        r=st[w]
        # rows wanting refill: empty or done
        want=[i for i in range(4) if r[i][0] in ('empty','done')]
        niter=sum(1 for i in range(4) if r[i][0]=='iter')
        cost=0.0
        do_refill = len(want)>=thresh or niter==0
        if do_refill and want:
            if any(r[i][0]=='done' for i in want): cost+=F
            got=False
            for i in want:
                k=draw()
                if k is None: r[i]=('idle',)
                else: r[i]=('new',k); got=True
            if got: cost+=a
        nnew=sum(1 for i in range(4) if r[i][0]=='new'); niter=sum(1 for i in range(4) if r[i][0]=='iter')
        if nnew==0 and niter==0:
            tend=max(tend,tw+cost); continue
        # one slot: sweeps shared
        cost+= 1.0 if niter>0 else b
        if nnew>0: cost+=c
        for i in range(4):
            if r[i][0]=='iter':
                rem=r[i][1]-1; r[i]=('iter',rem) if rem>0 else ('done',)
            elif r[i][0]=='new':
                n=it[r[i][1]]; r[i]=('iter',n) if n>0 else ('done',)
        heapq.heappush(h,(tw+cost,w))
    return tend
print("N",N,"B",B,"mean its",it.mean(),"max",it.max())
L=lockstep(); print("lockstep", L)
for th in (1,2,3,4):
    R=refill(th); print("refill thresh",th,R,"speedup %.2f"%(L/R))
def twopass(restart=0.6, r1=1.1, sort=False):
    need=np.nonzero(it>0)[0]
    its=it[need]
    if sort: its=np.sort(its)[::-1]
    nq=(len(its)+3)//4
    h=[(0.0,w) for w in range(W)]; heapq.heapify(h)
    for q in range(nq):
        tw,w=heapq.heappop(h); heapq.heappush(h,(tw+restart+its[4*q:4*q+4].max()+F,w))
    t2=max(x[0] for x in h)
    t1=np.ceil(((B+3)//4)/W)*r1
    return t1+t2
print("two-pass", twopass(), "speedup %.2f"%(L/twopass()), " (oracle-sorted: %.2f)"%(L/twopass(sort=True)))
