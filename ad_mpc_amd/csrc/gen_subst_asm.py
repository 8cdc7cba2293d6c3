#!/usr/bin/env python3
"""Generates subst_asm.inc: the two triangular substitutions of the condensed interior-point kernel as hand-scheduled
gfx950 assembly (n = 2N unknowns, unit lower-triangular factor packed by rows in LDS).

Why assembly: one substitution step is  z_j = y[lane j] (v_readlane x2) ; y[i] -= L[i][j] * z_j  for the lanes below the
diagonal.  hipcc keeps one address register per step (2 x 39 VGPRs) and waits for every ds_read right before its FMA
(~64 cycles exposed per step).  Here the factor entries are prefetched D steps ahead into a ring of D register pairs, the
addresses are one base register + immediate offsets, and the "lanes below / above the diagonal" predicate is an EXEC mask
written by one SALU instruction (s_bfm_b64) -- no per-step VALU work besides the two v_readlane and the FMA.

Register contract (fixed physical registers, see fwd_subst / bwd_subst in admpc_kernels.hip):
  v[100:101] y (in/out)   v102 LDS byte address of the lane's row (forward) / column (backward) of the packed factor
  v103 LDS byte address of publish buffer + 8 (lane % 16)   v[104:120) in-block ring   v[120:136) cross-block ring
  v[136:137] published block   s[46:47] saved EXEC
"""
import sys

D = 8
Y, ADDR, RING, SB, SAVE = 100, 102, 104, 40, 46


PUB, RING2, ZB = 103, 120, 136


def _rows(n):
    return [(16 * r, min(16 * r + 15, n - 1)) for r in range((n + 15) // 16)]


def _subst_blocked(n, forward, nrows=None):
    """Blocked triangular substitution without v_readlane (see the module docstring for the register contract).

    The wave's 16-lane rows are the blocks.  Inside a block a step is ONE instruction: the multiplier y_j reaches the lanes of
    its own row through DPP (v_fmac_f64_dpp y, y, -L row_newbcast:j%16), a plain VALU -> VALU dependency (measured with
    scripts/probes/dppchain_probe.hip: no wait state needed) instead of the v_readlane -> SGPR -> v_fma round trip that cost a
    lone wave ~54 cycles per step.  The DPP source lane must be enabled (no fetch-inactive on this DPP), so lane j takes part
    with the multiplier 0.0 that the packed factor keeps in its (otherwise unused) diagonal slot.  After a block its 16 finished
    values are published in LDS, read back into every row (ZB) and applied to the lanes of the remaining rows, again one DPP
    FMAC per column.  Factor entries: ring of D ds_read_b64 for the in-block steps, a second ring for the cross-block updates,
    EXEC masks (s_bfm_b64) as predicates, immediate offsets off one base register.
      forward : L z = y   (unit lower L, lane i reads L[i][j] at row base + 8 j)
      backward: L' x = z  (lane i reads L[j][i] at column base + 8 j (j + 1) / 2)"""
    # nrows > n (forward only): BORDER rows ride along -- lanes n .. nrows-1 hold rows of a second matrix C (row address in the same
    # register, entries at row base + 8 j) and receive y_b -= sum_j C[b][j] z_j, i.e. on return they hold y_b - C L^-T... (C L^-1-applied):
    # the reduced right-hand side of the bordered system (admpc_seg.hip)
    nrows = n if nrows is None else nrows
    assert forward or nrows == n
    rows = _rows(nrows)
    off = (lambda j: 8 * j) if forward else (lambda j: 8 * (j * (j + 1) // 2))
    out = ["s_mov_b64 s[%d:%d], exec" % (SAVE, SAVE + 1)]
    order = rows if forward else rows[::-1]
    # in-block steps, in execution order: (j, lo, hi) with lanes lo..hi enabled (targets + the source lane j)
    inblk = []
    nin = []                                       # in-block steps per row block
    for (r0, r1) in order:
        js = [j for j in range(r0, r1) if j < n] if forward else list(range(r1, r0, -1))
        for j in js:
            inblk.append((j, j, r1) if forward else (j, r0, j))
        nin.append(len(js))
    issued = []
    cur_exec = [None]
    def set_exec(lo, hi):
        if cur_exec[0] != (lo, hi):
            out.append("s_bfm_b64 exec, %d, %d" % (hi - lo + 1, lo)); cur_exec[0] = (lo, hi)
    def covers(lo, hi):
        return cur_exec[0] is not None and cur_exec[0][0] <= lo and hi <= cur_exec[0][1]
    def load_in(s):
        # a load needs its lanes enabled, extra enabled lanes only fetch values nobody uses: keep the current mask when it covers
        j, lo, hi = inblk[s]
        t = RING + 2 * (s % D)
        if not covers(lo, hi):
            set_exec(lo, hi)
        out.append("ds_read_b64 v[%d:%d], v%d offset:%d" % (t, t + 1, ADDR, off(j)))
        issued.append(("in", s))
    for s in range(min(D, len(inblk))):
        load_in(s)
    def wait_for(tag):
        out.append("s_waitcnt lgkmcnt(%d)" % min(15, len(issued) - 1 - issued.index(tag)))      # the counter is 4 bits wide
    s = 0
    WAIT_EVERY = 4
    for bi, (r0, r1) in enumerate(order):
        nsteps = nin[bi]
        waited_through = s - 1
        for _ in range(nsteps):
            j, lo, hi = inblk[s]
            t = RING + 2 * (s % D)
            if s > waited_through:                               # one wait covers the next WAIT_EVERY steps (their loads are older than D steps)
                upto = s
                while upto + 1 < len(inblk) and upto + 1 < s + WAIT_EVERY and ("in", upto + 1) in issued:
                    upto += 1
                wait_for(("in", upto)); waited_through = upto
            set_exec(lo, hi)
            out.append("v_fmac_f64_dpp v[%d:%d], v[%d:%d], -v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (Y, Y + 1, Y, Y + 1, t, t + 1, j % 16))
            if s + D < len(inblk):
                load_in(s + D)
            s += 1
        rest = order[bi + 1:]
        if not rest or r0 >= n:
            break
        # publish the block, read it back into every row, update the remaining rows
        tlo, thi = min(r[0] for r in rest), max(r[0] for r in rest) + 15      # WHOLE rows: a DPP source lane must be enabled, also beyond n
        c1 = min(r1, n - 1)                                                    # last COLUMN of the block (border rows have no columns)
        cols = list(range(r0, c1 + 1)) if forward else list(range(r1, r0 - 1, -1))
        out.append("s_bfm_b64 exec, %d, %d" % (c1 - r0 + 1, r0))
        out.append("s_nop 0")
        out.append("ds_write_b64 v%d, v[%d:%d] offset:%d" % (PUB, Y, Y + 1, 8 * r0))
        out.append("s_bfm_b64 exec, %d, %d" % (thi - tlo + 1, tlo))
        out.append("ds_read_b64 v[%d:%d], v%d offset:%d" % (ZB, ZB + 1, PUB, 8 * r0))
        issued.append(("zb", bi))
        def load_x(q):
            t = RING2 + 2 * (q % D)
            out.append("ds_read_b64 v[%d:%d], v%d offset:%d" % (t, t + 1, ADDR, off(cols[q])))
            issued.append(("x", bi, q))
        for q in range(min(D, len(cols))):
            load_x(q)
        for q, j in enumerate(cols):
            t = RING2 + 2 * (q % D)
            wait_for(("x", bi, q))       # in-order returns: the ZB read was issued earlier and is complete as well
            out.append("v_fmac_f64_dpp v[%d:%d], v[%d:%d], -v[%d:%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (Y, Y + 1, ZB, ZB + 1, t, t + 1, j % 16))
            if q + D < len(cols):
                load_x(q + D)
    out.append("s_waitcnt lgkmcnt(0)")
    out.append("s_mov_b64 exec, s[%d:%d]" % (SAVE, SAVE + 1))
    return out


def fwd(n, nrows=None):
    return _subst_blocked(n, True, nrows)


def bwd(n):
    return _subst_blocked(n, False)


def rowbuild(n, lo, hi, nrows=None):
    """Newton-matrix row of this lane, columns lo..hi-1: a[c] = (c <= lane ? H[lane][c] : 0) + (c odd ? s_odd : 0) + (c == lane ? dbar : 0).
    Operands: %0..%(hi-lo-1) the row entries (outputs), then the lane's row address in LDS, dbar, s_odd.  Predicates are
    EXEC masks (s_bfm_b64), all loads are in flight together, one wait."""
    cnt = hi - lo; A, DB, SO = cnt, cnt + 1, cnt + 2
    out = ["s_mov_b64 s[%d:%d], exec" % (SAVE, SAVE + 1)]
    for q in range(cnt):
        out.append("v_mov_b64 %%%d, 0" % q)
    for q in range(cnt):
        c = lo + q
        out.append("s_bfm_b64 exec, %d, %d" % ((nrows or n) - c, c))           # lanes c .. n-1 (.. nrows-1: border rows are full rows)
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


def symrow(n, lo, hi, nrows=None):
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
        out.append("s_bfm_b64 exec, %d, %d" % ((nrows or n) - c, c))           # border rows (lanes n .. nrows-1) read their full rows
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
    clob = ", ".join('"v%d"' % r for r in list(range(RING, RING + 2 * D)) + list(range(RING2, RING2 + 2 * D)) + [ZB, ZB + 1]) + ", " + ", ".join('"s%d"' % r for r in (46, 47))
    txt = "// GENERATED by gen_subst_asm.py (n = %d, prefetch depth %d) -- do not edit\n" % (n, D)
    txt += emit("ADMPC_FWD_SUBST_ASM_%d" % n, fwd(n)) + "\n" + emit("ADMPC_BWD_SUBST_ASM_%d" % n, bwd(n)) + "\n"
    txt += emit("ADMPC_ROWBUILD_ASM_%d_A" % n, rowbuild(n, 0, n // 2)) + "\n" + emit("ADMPC_ROWBUILD_ASM_%d_B" % n, rowbuild(n, n // 2, n)) + "\n"
    txt += emit("ADMPC_ROWSTORE_ASM_%d_A" % n, rowstore(n, 0, n // 2)) + "\n" + emit("ADMPC_ROWSTORE_ASM_%d_B" % n, rowstore(n, n // 2, n)) + "\n"
    txt += emit("ADMPC_SYMROW_ASM_%d_A" % n, symrow(n, 0, n // 2)) + "\n" + emit("ADMPC_SYMROW_ASM_%d_B" % n, symrow(n, n // 2, n)) + "\n"
    # bordered variants (admpc_seg.hip): lanes n .. nrows-1 carry full rows of a border matrix through the row build and the forward substitution
    for nr in (47, 54):
        txt += emit("ADMPC_FWD_SUBST_ASM_%d_R%d" % (n, nr), fwd(n, nr)) + "\n"
        txt += emit("ADMPC_ROWBUILD_ASM_%d_A_R%d" % (n, nr), rowbuild(n, 0, n // 2, nr)) + "\n" + emit("ADMPC_ROWBUILD_ASM_%d_B_R%d" % (n, nr), rowbuild(n, n // 2, n, nr)) + "\n"
        txt += emit("ADMPC_SYMROW_ASM_%d_A_R%d" % (n, nr), symrow(n, 0, n // 2, nr)) + "\n" + emit("ADMPC_SYMROW_ASM_%d_B_R%d" % (n, nr), symrow(n, n // 2, n, nr)) + "\n"
    txt += "#define ADMPC_SUBST_CLOBBERS %s, \"memory\"\n" % clob
    open(sys.argv[1] if len(sys.argv) > 1 else "subst_asm.inc", "w").write(txt)


if __name__ == "__main__":
    main()
