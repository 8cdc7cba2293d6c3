"""Experiment: does splitting one batch over 2 (or 4) HIP streams shorten a step?  (Each part has its own engine:
own scheduler header and workspaces.)  Usage on the GPU box: python scripts/overlap_proto.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = default_config(N=20, Ts=0.05)
sc = random_scenarios(B, N=20, Ts=0.05, seed=1234, start=0, blend=(100.0, 110.0))
for parts in (1, 2, 3, 4):
    engs = [BatchSolver(cfg, device=0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    d = engs[0].to_device
    cuts = [B * i // parts for i in range(parts + 1)]
    data = []
    for i in range(parts):
        s = slice(cuts[i], cuts[i + 1])
        data.append([d(sc[k][s]) for k in ("x0", "yref", "yref_e", "p", "xbar", "ubar")])
    main = torch.cuda.current_stream()
    bufs = [(a[4].clone(), a[5].clone()) for a in data]
    def step():                      # fork from the caller's stream, join back: no overlap across steps
        ev = main.record_event()
        for e, st, a, b in zip(engs, streams, data, bufs):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                e.solve(a[0], a[1], a[2], a[3], b[0], b[1])
            main.wait_event(st.record_event())
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("parts", parts, "ms/step %.4f" % (dt * 1e3), "solves/s %.3e" % (B / dt), flush=True)
    for e in engs: e.close()
