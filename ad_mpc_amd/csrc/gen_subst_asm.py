#!/usr/bin/env python3
"""Generates subst_asm.inc: the two triangular substitutions of the condensed interior-point kernel as hand-scheduled
gfx950 assembly (n = 2N unknowns, unit lower-triangular factor packed by rows in LDS).

Why assembly: one substitution step is  z_j = y[lane j] (v_readlane x2) ; y[i] -= L[i][j] * z_j  for the lanes below the
diagonal.  hipcc keeps one address register per step (2 x 39 VGPRs) and waits for every ds_read right before its FMA
(~64 cycles exposed per step).  Here the factor entries are prefetched D steps ahead into a ring of D register pairs, the
addresses are one base register + immediate offsets, and the "lanes below / above the diagonal" predicate is an EXEC mask
written by one SALU instruction (s_bfm_b64) -- no per-step VALU work besides the two v_readlane and the FMA.

Register contract (fixed physical registers, see fwd_subst / bwd_subst in admpc_kernels.hip):
  v[100:101] y (in/out)   v102 LDS byte address (per lane)   v[104 : 104+2D) prefetch ring   s[40:43] broadcast values
  s[46:47] saved EXEC
"""
import sys

D = 8
Y, ADDR, RING, SB, SAVE = 100, 102, 104, 40, 46


def _subst(n, steps, offset_of, exec_of):
    """Common schedule of both substitutions.  Ring slot s % D holds the factor entries of step s and is refilled (for step
    s + D) one step later.  HAZARD (measured on MI355X, scripts/probes/subst_probe.hip: wrong and non-repeatable results
    without it): the v_readlane_b32 pair of the next step reads the register the v_fma_f64 has just written; one wait state
    is required between them and an LDS instruction does NOT count as one -- hence the s_nop 0 closing every step.  (The
    assembler cannot check hazards inside inline assembly; hipcc's own recogniser never sees these instructions.)"""
    out = ["s_mov_b64 s[%d:%d], exec" % (SAVE, SAVE + 1)]
    issued = []                                   # step indices in issue order
    def load(s):
        t = RING + 2 * (s % D)
        out.append("ds_read_b64 v[%d:%d], v%d offset:%d" % (t, t + 1, ADDR, offset_of(steps[s])))
        issued.append(s)
    for s in range(min(D, len(steps))):
        load(s)
    for s, j in enumerate(steps):
        sp = SB + 2 * (s & 1); t = RING + 2 * (s % D)
        out.append("v_readlane_b32 s%d, v%d, %d" % (sp, Y, j))
        out.append("v_readlane_b32 s%d, v%d, %d" % (sp + 1, Y + 1, j))
        out.append(exec_of(j))
        out.append("s_waitcnt lgkmcnt(%d)" % (len(issued) - 1 - issued.index(s)))
        out.append("v_fma_f64 v[%d:%d], -s[%d:%d], v[%d:%d], v[%d:%d]" % (Y, Y + 1, sp, sp + 1, t, t + 1, Y, Y + 1))
        if s >= 1 and s - 1 + D < len(steps):
            load(s - 1 + D)                       # refill the slot of the previous step
        out.append("s_nop 0")                     # hazard pad: v_fma_f64 result -> v_readlane of it (see _subst.__doc__)
    out.append("s_mov_b64 exec, s[%d:%d]" % (SAVE, SAVE + 1))
    return out


def fwd(n):
    """L z = y, unit lower L: lane i reads L[i][j] at base_i + 8 j (base_i = &Lp[i(i+1)/2]); lanes j+1 .. n-1 take part in step j."""
    return _subst(n, list(range(n - 1)), lambda j: 8 * j, lambda j: "s_bfm_b64 exec, %d, %d" % (n - 1 - j, j + 1))


def bwd(n):
    """L' x = z: lane i reads L[j][i] at base_i + 8 j(j+1)/2 (base_i = &Lp[i]); lanes 0 .. j-1 take part in step j = n-1 .. 1."""
    return _subst(n, list(range(n - 1, 0, -1)), lambda j: 8 * (j * (j + 1) // 2), lambda j: "s_bfm_b64 exec, %d, 0" % j)


def rowbuild(n, lo, hi):
    """Newton-matrix row of this lane, columns lo..hi-1: a[c] = (c <= lane ? H[lane][c] : 0) + (c odd ? s_odd : 0) + (c == lane ? dbar : 0).
    Operands: %0..%(hi-lo-1) the row entries (outputs), then the lane's row address in LDS, dbar, s_odd.  Predicates are
    EXEC masks (s_bfm_b64), all loads are in flight together, one wait."""
    cnt = hi - lo; A, DB, SO = cnt, cnt + 1, cnt + 2
    out = ["s_mov_b64 s[%d:%d], exec" % (SAVE, SAVE + 1)]
    for q in range(cnt):
        out.append("v_mov_b64 %%%d, 0" % q)
    for q in range(cnt):
        c = lo + q
        out.append("s_bfm_b64 exec, %d, %d" % (n - c, c))                      # lanes c .. n-1
        out.append("ds_read_b64 %%%d, %%%d offset:%d" % (q, A, 8 * c))
    out.append("s_mov_b64 exec, s[%d:%d]" % (SAVE, SAVE + 1))
    out.append("s_waitcnt lgkmcnt(0)")
    for q in range(cnt):
        if (lo + q) & 1:
            out.append("v_add_f64 %%%d, %%%d, %%%d" % (q, q, SO))
    for q in range(cnt):
        c = lo + q
        out.append("s_bfm_b64 exec, 1, %d" % c)                                # lane c
        out.append("v_add_f64 %%%d, %%%d, %%%d" % (q, q, DB))
    out.append("s_mov_b64 exec, s[%d:%d]" % (SAVE, SAVE + 1))
    return out


def rowstore(n, lo, hi):
    """Packed lower-triangular row of this lane, columns lo..hi-1, to LDS: lanes c .. n-1 store column c (EXEC mask), one base
    address + immediate offsets.  Operands: %0..%(hi-lo-1) the row entries, then the lane's row address."""
    cnt = hi - lo
    out = ["s_mov_b64 s[%d:%d], exec" % (SAVE, SAVE + 1)]
    for q in range(cnt):
        c = lo + q
        out.append("s_bfm_b64 exec, %d, %d" % (n - c, c))
        out.append("ds_write_b64 %%%d, %%%d offset:%d" % (cnt, q, 8 * c))
    out.append("s_mov_b64 exec, s[%d:%d]" % (SAVE, SAVE + 1))
    return out


def symrow(n, lo, hi):
    """Row of the symmetric matrix stored as packed lower-triangular rows, columns lo..hi-1, for the mat-vec: lanes >= c read
    H[lane][c] (row address + 8 c), lanes < c read H[c][lane] (column address + 8 c (c + 1) / 2) -- two EXEC-masked loads per
    column into the same register, all in flight together.  Operands: %0..%(hi-lo-1) outputs, then row address, column address.
    Lanes >= n keep 0."""
    cnt = hi - lo; RA, CA = cnt, cnt + 1
    out = []
    for q in range(cnt):
        out.append("v_mov_b64 %%%d, 0" % q)
    for q in range(cnt):
        c = lo + q
        out.append("s_bfm_b64 exec, %d, %d" % (n - c, c))
        out.append("ds_read_b64 %%%d, %%%d offset:%d" % (q, RA, 8 * c))
        if c > 0:
            out.append("s_bfm_b64 exec, %d, 0" % c)
            out.append("ds_read_b64 %%%d, %%%d offset:%d" % (q, CA, 8 * (c * (c + 1) // 2)))
    out.append("s_mov_b64 exec, -1")
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def emit(name, lines):
    body = " \\\n".join('    "%s\\n\\t"' % l for l in lines)
    return "#define %s \\\n%s\n" % (name, body)


def main():
    n = 40
    clob = ", ".join('"v%d"' % r for r in range(RING, RING + 2 * D)) + ", " + ", ".join('"s%d"' % r for r in (40, 41, 42, 43, 46, 47))
    txt = "// GENERATED by gen_subst_asm.py (n = %d, prefetch depth %d) -- do not edit\n" % (n, D)
    txt += emit("ADMPC_FWD_SUBST_ASM_%d" % n, fwd(n)) + "\n" + emit("ADMPC_BWD_SUBST_ASM_%d" % n, bwd(n)) + "\n"
    txt += emit("ADMPC_ROWBUILD_ASM_%d_A" % n, rowbuild(n, 0, n // 2)) + "\n" + emit("ADMPC_ROWBUILD_ASM_%d_B" % n, rowbuild(n, n // 2, n)) + "\n"
    txt += emit("ADMPC_ROWSTORE_ASM_%d_A" % n, rowstore(n, 0, n // 2)) + "\n" + emit("ADMPC_ROWSTORE_ASM_%d_B" % n, rowstore(n, n // 2, n)) + "\n"
    txt += emit("ADMPC_SYMROW_ASM_%d_A" % n, symrow(n, 0, n // 2)) + "\n" + emit("ADMPC_SYMROW_ASM_%d_B" % n, symrow(n, n // 2, n)) + "\n"
    txt += "#define ADMPC_SUBST_CLOBBERS %s, \"memory\"\n" % clob
    open(sys.argv[1] if len(sys.argv) > 1 else "subst_asm.inc", "w").write(txt)


if __name__ == "__main__":
    main()
