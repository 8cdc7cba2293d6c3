"""python scripts/seg_debug.py [N seed inst]: one instance through the bring-up build of the segmented kernel (make -C ad_mpc_amd/csrc segdebug),
its LDS dumps against the quantities of tests/seg_spec.py."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ad_mpc_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "ad_mpc_amd", "libadmpc_segdbg.so")
from ad_mpc_amd.config import default_config
from ad_mpc_amd.engine import BatchSolver
from ad_mpc_amd.scenarios import random_scenarios
from oracle.oracle import Oracle
from seg_spec import SegQP, seg_ipm

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pick = int(sys.argv[3]) if len(sys.argv) > 3 else 0
S = N // 20
NB = 7 if S == 2 else 14
cfg = default_config(N=N)
sc = random_scenarios(pick + 1, N=N, seed=seed)
one = {k: v[pick:pick + 1] for k, v in sc.items()}
orc = Oracle()
if len(sys.argv) > 4 and sys.argv[4] == "iter2":      # the iterate after one RTI step: stage-dependent linearisation, non-zero inputs
    r1 = orc.solve_batch(cfg, one["x0"], one["yref"], one["yref_e"], one["p"], one["xbar"], one["ubar"])
    one["xbar"] = r1[0].copy(); one["ubar"] = r1[1].copy()
o = orc.qp_debug(cfg, one["x0"][0], one["yref"][0], one["yref_e"][0], one["p"][0], one["xbar"][0], one["ubar"][0])
qp = SegQP(cfg, o["A"], o["B"], o["b"], one["x0"][0], one["yref"][0], one["yref_e"][0], one["xbar"][0], one["ubar"][0])
dbg = []
res, it = seg_ipm(cfg, qp, dbg=dbg)
print("oracle iters", o["iters"], "spec iters", it)

bs = BatchSolver(cfg)
lib = bs.lib if hasattr(bs, "lib") else _lib.load()
lib.admpc_debug_seg.argtypes = [C.POINTER(C.c_double), C.c_int]
lib.admpc_debug_seg.restype = C.c_int
assert lib.admpc_debug_seg(None, 0) == 0, "not a SEG_DEBUG build"
g = bs.solve_numpy(one["x0"], one["yref"], one["yref_e"], one["p"], one["xbar"], one["ubar"])
W = 16384
buf = np.zeros(4 * 2 * W)
assert lib.admpc_debug_seg(buf.ctypes.data_as(C.POINTER(C.c_double)), 0) == 0
print("device iters", g[4][0], "status", g[3][0], "max|du - oracle|", np.abs(g[1][0] - (one["ubar"][0] + o["du"])).max())

oH, oHb = 0, 820; oL = oHb + NB * 40; oLb = oL + 820
# interface block layout of SegLds<S> (admpc_seg.hip)
SCS = NB
IF = {}
IF["SC"] = 0; IF["HZZ"] = (NB * NB + 1) & ~1; IF["C"] = IF["HZZ"] + 56; IF["ZB"] = IF["C"] + 8; IF["Z"] = IF["ZB"] + 16; IF["ZC"] = IF["Z"] + 8
IF["DZ"] = IF["ZC"] + 8; IF["NU"] = IF["DZ"] + 8; IF["RED"] = IF["NU"] + 8; IF["BU"] = IF["RED"] + 32; IF["PI"] = IF["BU"] + 8; IF["G56"] = IF["PI"] + 56
IF["ABAR"] = IF["G56"] + 2
IFS = IF["ABAR"] + (56 + 8 + 56 + 8 if S > 2 else 0)

def tri(v):
    M = np.zeros((40, 40))
    for i in range(40):
        M[i, :i + 1] = v[i * (i + 1) // 2: i * (i + 1) // 2 + i + 1]
    return M

def rep(name, got, want):
    e = np.abs(got - want).max() / max(1e-300, np.abs(want).max())
    print("   %-28s rel err %.2e   (|want| max %.3e)" % (name, e, np.abs(want).max()))

trial = dbg[0]
for s in range(S):
    sg = qp.segs[s]; first, last = sg["first"], sg["last"]
    bslot = 0 if first else 7
    D1 = buf[(s * 2 + 0) * W:(s * 2 + 1) * W]; D2 = buf[(s * 2 + 1) * W:(s * 2 + 2) * W]
    print("segment", s, "P1 (behind the condensing)")
    rep("Huu (lower)", tri(D1[oH:oH + 820]), np.tril(sg["Huu"]))
    Hb = D1[oHb:oHb + NB * 40].reshape(NB, 40)
    if not first: rep("Hzu", Hb[:7], sg["Hzu"])
    if not last: rep("Bbar", Hb[bslot:bslot + 7], sg["Bbar"])
    I1 = D1[4000:4000 + IFS]
    if not first:
        rep("Hzz", I1[IF["HZZ"]:IF["HZZ"] + 56].reshape(7, 8)[:, :7], sg["Hzz"])
    if not last:
        if not first: rep("Abar", I1[IF["ABAR"]:IF["ABAR"] + 56].reshape(7, 8)[:, :7], sg["Abar"])
        rep("c", I1[IF["C"]:IF["C"] + 7], sg["c"])
    g0 = D1[5000:5064]
    rep("g0 (inputs)", g0[:40], sg["gu"])
    if not first: rep("g0 (z lanes)", g0[40:47], sg["gz"])
    rep("xh6", D1[5064:5084], sg["xh6"])
    print("      rsx", D1[5128])
    print("segment", s, "P2 (behind the trial)")
    f = trial["fac"]["seg"][s]
    Ld = tri(D2[oL:oL + 820]); Lw = np.tril(f["L"], -1)
    rep("L (strictly lower)", np.tril(Ld, -1), Lw)
    rep("1/d", D2[oLb + NB * 40 + 320 + 64: oLb + NB * 40 + 320 + 64 + 40], 1.0 / f["d"])
    Lb = D2[oLb:oLb + NB * 40].reshape(NB, 40)
    want_Lb = np.zeros((NB, 40)); k = 0
    if not first: want_Lb[:7] = f["Lb"][:7]; k = 7
    if not last: want_Lb[bslot:bslot + 7] = f["Lb"][k:k + 7]
    rep("Lb", Lb, want_Lb)
    I2 = D2[4000:4000 + IFS]
    Sc = I2[IF["SC"]:IF["SC"] + NB * NB].reshape(NB, NB)
    idx = []
    if not first: idx += list(range(7))
    if not last: idx += list(range(bslot, bslot + 7))
    rep("Sc", Sc[np.ix_(idx, idx)], f["Sc"])
    zb = I2[IF["ZB"]:IF["ZB"] + 14]
    idx = idx if NB == 14 or first else list(range(7))
    rep("zb", zb[idx], trial["zb"][s])
    if not first:
        rep("dz", I2[IF["DZ"]:IF["DZ"] + 7], trial["dz"][s])
        if S > 2: rep("nu", I2[IF["NU"]:IF["NU"] + 7], trial["nu"][s])
        rep("Pi", I2[IF["PI"]:IF["PI"] + 56].reshape(7, 8)[:, :7], trial["fac"]["Pi"][s])
        rep("zc (cold z)", I2[IF["ZC"]:IF["ZC"] + 7], trial["dz"][s] * 0 + I2[IF["ZC"]:IF["ZC"] + 7])
    rep("trial dU", D2[5000:5040], trial["dU"][s])
    print("      trial rhs: zg", D2[5064:5064 + 47][[0, 1, 39, 40, 46]])
