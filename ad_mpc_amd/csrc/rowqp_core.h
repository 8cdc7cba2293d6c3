// rowqp_core.h -- the QP of one RTI step (H2-H6 of SURVEY 8a), one MPC instance per 16-lane DPP row.
//
// Backend-generic source: the same text is compiled
//   * by hipcc for gfx950 with the device backend (rowqp_dev.h): a wavefront carries FOUR instances, one per 16-lane
//     row; every cross-lane exchange inside an instance is a DPP row broadcast folded into the FMA
//     (v_fmac_f64_dpp / v_fmac_f32_dpp row_newbcast:n) or a DPP move, never an LDS round trip;
//   * by g++ with the lane emulator of tests/emu (one row = one instance, 16 emulated lanes) so that the CPU test suite runs
//     the product's own algorithm source against the oracle.  The emulator is test infrastructure, not a fallback.
//
// What is computed (reference: data_driven_mpc/ros_gp_mpc/src/ad_mpc/ad_3d_optimizer.py:146-205 formulation, acados SQP_RTI
// step + HPIPM solve behind ad_3d_optimizer.py:456):  a Mehrotra predictor-corrector primal-dual interior
// point method on the stage QP, every Newton system solved by a stage-wise Riccati recursion.  Same Newton steps as
// oracle/admpc_oracle.c (ipm_solve), reorganised:
//   * the dynamics multipliers are not iterated.  The primal Newton step does not depend on them (their contribution to the
//     stationarity residual telescopes over any direction that satisfies the linearised dynamics), so the right-hand sides are
//     built with pi = 0.  Every linear residual of a Newton iteration in residual form shrinks by exactly (1 - alpha) per step:
//     the stationarity residual of the stopping test is tracked from its start value instead of being re-evaluated; the
//     multipliers themselves (iterate snapshot) are the exact adjoint of the returned point (sweep SA).
//   * sl == t[2], su == t[3] (they start equal and receive identical steps), so they are not stored.
//   * per-inequality work is organised by SIDE: a lane owns one bound of one input together with its slack pair
//     (t_b, lam_b, t_s, lam_s), or one steering bound (t_b, lam_b); two stages per pass step (lanes 0-5 and 8-13).
//   * the iterate is carried in absolute form (xa = xbar + dx, ua = ubar + du).
//
// Lane roles inside a row (lane l = 0..15):
//   sweeps   l < 7: state component l (column l of P, K, A);  l = 7, 8: input 0, 1 (columns of B);  l >= 9 idle
//   passes   l = 8*sp + side: stage parity sp, side 0..3 = (u0 lower, u0 upper, u1 lower, u1 upper), 4, 5 = steering lower, upper
//
// Data placement per instance
//   LDS, pass-private, RQ_RS = 31 values per stage k = 0..N-1 behind a 16-value dump header (N = 40 fp64 / N = 80 fp32: 16 instances
//   fill the 160 KB of a CU to 98.8 %):
//     T[10] LAM[10]  slack / multiplier of the bound pairs (side 0..5) and slack pairs (6 + side, side 0..3)
//     UA[2]          input of stage k (absolute)        X6   steering angle of stage k (copy of XA_k[6])
//     U[2]           gu -> kff -> ddu -> (corrector) gA -> kff -> ddu
//     A[2]           Rt (barrier-augmented R) -> predictor ddu
//     Q1             Qt -> predictor ddx6              X     steering-barrier term of gx6 -> (corrector) xA -> ddx6
//     UR[2]          input references of the stage (copied once: the passes of an iteration touch no global memory but E2's store)
//   workspace in global memory (L2-resident, streamed one stage ahead by the sweeps), RQ_RW values per record r = 0..N:
//     XA[7] state of stage r (absolute)   D[7] Newton step of the state of stage r
//     KK[9][2]  of stage r-1, one pair per lane: (K0[c], K1[c]) for c < 7 (feedback gains, column c), then (i00, i01), (i01, i11)
//               (inverse of Huu as lanes 7, 8 use it) -- ONE paired load / store per stage and sweep: a vector-memory
//               instruction costs the CU 16 cycles whatever its lanes do, and the four waves of a CU share that pipe
//     UB[2] XB  sigma-mu coefficients of the corrector right-hand side of stage r-1
#pragma once

#ifndef RQ_FN
#define RQ_FN static inline
#endif
#ifndef RQ_UNROLL
#define RQ_UNROLL
#define RQ_NOUNROLL
#endif
#ifndef RQ_DBG
#define RQ_DBG(...) do { } while (0)
#endif

// stages in flight in the light sweeps: 3 (2 -> 3 gained 1 % at N = 40).  4 costs fp64 1 % (246 AGPRs of spill space) and gains fp32 3.9 % at configs[4]
// (3.04 -> 3.16 M solves/s: the kernel waits for its workspace half of the time, SQ_WAIT_ANY 49 % -- 200 MB of stage records live in the MALL, not in
// L2), but the other build contracts its multiply-adds differently, and in fp32 that moved three instances of 393 216 to 2 - 4e-3 off the fp64
// minimiser, outside the documented 2.5e-3 (with 3: none above 1.3e-3): not taken (profiles/r4/rowqp_prefetch_depth.txt).  -DRQ_PF=k sets both.
#ifdef RQ_PF
#define RQ_PF64 RQ_PF
#define RQ_PF32 RQ_PF
#else
#define RQ_PF64 3
#define RQ_PF32 3
#endif
#define RQ_HDR 16
#define RQ_RS 31
enum { RQ_T = 0, RQ_LAM = 10, RQ_UA = 20, RQ_X6 = 22, RQ_X = 23, RQ_U = 24, RQ_A = 26, RQ_Q1 = 28, RQ_UR = 29 };
#define RQ_RW 38
enum { RW_XA = 0, RW_D = 8, RW_KK = 16, RW_UB = 34, RW_XB = 36 };
#define RQ_GTS 42      // packed linearisation per stage: stored columns c' = 0..6 <-> (A[:,2..6], B[:,0..1]), rows 0..5 of each

template <class T>
struct RqParams {              // wave-uniform scalars
    int N, itmax, try_unc;
    T h;                       // Ts
    T Qd[7], Qe[7], Rd[2];     // Ts*q, W_e, Ts*r
    T lbu[2], ubu[2], lbd, ubd;
    T rho_l, rho_u;            // Ts*zl, Ts*zu
    T thr, thw, mu0, tol_comp, tol_res, tol_step;
    T inv_nineq;               // 1 / (8N + 2(N-1))
    T big;                     // |value| above this (or NaN) = failed step
    T floor_;                  // lower clamp of t, lam
    T mu_floor;                // lower bound of the centring target sigma * mu (ADMPC_IPM_MU_FLOOR, admpc.h)
    T blocked;                 // centring safeguard: step length below which the next iteration centres (ADMPC_IPM_BLOCKED_STEP)
    T wrest;                   // a warm start whose first step is shorter than this is abandoned for the cold start (0: never)
    int fbit;                  // fallback: a row still iterating after this many iterations restarts without the second-order term (0: never)
};

template <class T>
struct RqArrays {              // wave-uniform array bases (device: kernel arguments, SGPR-addressed accesses with 32-bit lane offsets)
    const T *x0, *yref, *yref_e, *GT, *bl;
    T *xbar, *ubar;
    T *pi;                     // optional snapshot: [B][N+1][7] dynamics multipliers pi_0..pi_{N-1}, row N = multiplier of the x0 equality
    T *ineq;                   // optional snapshot: [B][N][20] slacks T[10] and multipliers LAM[10] of every stage (record order)
    T *ws;                     // [B][N+1][RQ_RW]
    T *dump;                   // split batches: [B][RQ_HDR + N RQ_RS] the LDS region of a deferred instance after its trial (phase 1 -> phase 2)
};

template <class X>
struct RowQp {
    typedef typename X::T T;
    typedef typename X::V V;
    typedef typename X::I I;
    typedef typename X::M M;
    typedef typename X::Lds Lds;
    static constexpr int PF = sizeof(T) == 4 ? RQ_PF32 : RQ_PF64;

    const RqParams<T>& q;
    const RqArrays<T>& io;
    I ix0, iyr, iye, igt, ibl, ixb, iub, iws, ipi, iiq, idp;   // element offsets of the row's instance in the arrays
    Lds lds;
    const int N;
    M owns;                    // the row works on an instance of its own (rows that only shadow another row's instance never write its workspace)
    M wrows;                   // rows whose LDS records / workspace may be written right now (all, except while single rows are re-initialised)
    V w2;                      // weight of the second-order corrector term of the row: 1, or 0 in fallback mode (cfg.ipm_fallback_iter)

    // ---- lane constants -------------------------------------------------------------------------------------------
    I lane;
    M is_x, is_u, is6, is7, lt2;
    V E[9];                    // E[r] = (lane == r)
    V keepc, unitc[2];         // column-layout fix-up of lanes 0, 1 (unit columns of A)
    V g6c;                     // G[6][c] of the lane's column: 1 (c = 6), h (c = 8), else 0
    V lt2f;                    // 1 on lanes 0, 1
    V wq, wqe;                 // state weights of the lane (0 on lanes >= 7)
    V rjm;                     // Rd[j] on lanes 7, 8
    I o_gc, o_gr, o_x, o_y, o_u, o_kk, o_ub;
    I l_u, l_a, l_ua, l_l0;
    // passes
    I sp, side, jin, o_tb, o_ts, o_vl, o_sa, o_sc, o_o1, o_o2, o_dxu;
    M e_valid, e_isd, e_even, e_in, dxl;
    V sgn, bound, rho, rjin, wsd, wesd, lbj, ubj, q6orr;

    RQ_FN V splat(T x) { return X::splat(x); }

    // inst: index of the row's instance (row-uniform); owns_: it is this row's own instance
    RQ_FN RowQp(const RqParams<T>& q_, const RqArrays<T>& io_, Lds lds_, I inst, M owns_) : q(q_), io(io_), lds(lds_), N(q_.N), owns(owns_)
    {
        wrows = X::mtrue();
        ix0 = inst * 7; iye = ix0; iyr = inst * (N * 9); igt = inst * (N * RQ_GTS); ibl = inst * (N * 7); ixb = inst * ((N + 1) * 7);
        iub = inst * (N * 2); iws = inst * ((N + 1) * RQ_RW); ipi = ixb; iiq = inst * (N * 20); idp = inst * (RQ_HDR + N * RQ_RS);
        lane = X::lane();
        is_x = lane < 7; is_u = (lane == 7) | (lane == 8); is6 = lane == 6; is7 = lane == 7; lt2 = lane < 2;
        const V one = splat((T)1), zero = splat((T)0);
        RQ_UNROLL
        for (int r = 0; r < 9; ++r) E[r] = X::sel(lane == r, one, zero);
        keepc = X::sel(lt2, zero, one);
        unitc[0] = E[0]; unitc[1] = E[1];
        lt2f = X::sel(lt2, one, zero);
        g6c = X::sel(is6, one, X::sel(lane == 8, splat(q.h), zero));
        wq = zero; wqe = zero;
        RQ_UNROLL
        for (int i = 0; i < 7; ++i) { wq = X::sel(lane == i, splat(q.Qd[i]), wq); wqe = X::sel(lane == i, splat(q.Qe[i]), wqe); }
        rjm = X::sel(is7, splat(q.Rd[0]), X::sel(lane == 8, splat(q.Rd[1]), zero));
        const I zi = X::isplat(0);
        const I ju = X::isel(is_u, lane - 7, zi);
        o_gc = X::isel((lane >= 2) & (lane < 9), (lane - 2) * 6, zi);
        o_gr = X::isel(lane < 6, lane, zi);
        o_x = X::isel(is_x, lane, zi);
        o_y = X::isel(lane < 9, lane, zi);
        o_u = ju;
        o_kk = X::isel(lane < 9, lane * 2, zi) + RW_KK;                              // the lane's pair of the gain record
        o_ub = X::isel(is_u, ju + RW_UB, X::isplat(RW_XB));                          // S3: lanes 7, 8 UB[j], lane 6 XB
        l_u = ju + RQ_U;                                                            // LDS: U[j] (lanes 7, 8)
        l_a = X::isel(is_u, ju + RQ_A, X::isplat(RQ_X));                            // LDS: A[j] (lanes 7, 8), X (lane 6)
        l_ua = ju + RQ_UA;
        l_l0 = X::isel(is_u, ju * 2 + RQ_LAM, X::isplat(RQ_LAM + 4));               // SA: lower-bound multiplier; upper = +1
        // pass lanes
        sp = lane >> 3; side = lane & 7;
        e_valid = side < 6; e_isd = (side >= 4) & e_valid; e_in = side < 4; e_even = (side & 1) == 0;
        dxl = side < 7;
        jin = X::isel(e_in, side >> 1, zi);
        const M up = (side & 1) == 1;
        sgn = X::sel(up, splat((T)-1), one);
        bound = X::sel(e_isd, X::sel(up, splat(q.ubd), splat(q.lbd)),
                       X::sel(jin == 0, X::sel(up, splat(q.ubu[0]), splat(q.lbu[0])), X::sel(up, splat(q.ubu[1]), splat(q.lbu[1]))));
        rho = X::sel(up, splat(q.rho_u), splat(q.rho_l));
        rjin = X::sel(jin == 0, splat(q.Rd[0]), splat(q.Rd[1]));
        q6orr = X::sel(e_isd, splat(q.Qd[6]), rjin);
        wsd = zero; wesd = zero;
        RQ_UNROLL
        for (int i = 0; i < 7; ++i) { wsd = X::sel(side == i, splat(q.Qd[i]), wsd); wesd = X::sel(side == i, splat(q.Qe[i]), wesd); }
        lbj = X::sel(jin == 0, splat(q.lbu[0]), splat(q.lbu[1])); ubj = X::sel(jin == 0, splat(q.ubu[0]), splat(q.ubu[1]));
        const I sidec = X::isel(e_valid, side, zi);
        o_tb = sidec;                                                               // bound pair of the side
        o_ts = X::isel(e_in, side + 6, sidec);                                      // slack pair (inputs), own pair otherwise
        o_vl = X::isel(e_isd, X::isplat(RQ_X6), jin + RQ_UA);                        // absolute value the side bounds: delta_k / u_kj
        o_sa = X::isel(e_isd, X::isplat(RQ_Q1), jin + RQ_A);                         // predictor step slot
        o_sc = X::isel(e_isd, X::isplat(RQ_X), jin + RQ_U);                          // (corrector / trial) step slot
        o_o1 = o_sc;                                                                // pass outputs: X | U
        o_o2 = o_sa;                                                                //               Q1 | A
        o_dxu = X::isel(dxl, side, zi);
    }

    // ---- small helpers ----------------------------------------------------------------------------------------------
    RQ_FN V ld(I off, int imm) { return X::lds_ld(lds, off, imm); }
    RQ_FN void st(I off, int imm, V v, M m) { X::lds_st(lds, off, imm, v, m & wrows); }
    RQ_FN static V fma(V a, V b, V c) { return X::fma(a, b, c); }
    RQ_FN static int rec(int k) { return RQ_HDR + k * RQ_RS; }
    // workspace: written by one sweep / pass and read back by a later one of the same wave -> accessors of their own (the device
    // reads it past the CU's L1, whose lines do not follow the wave's own stores)
    RQ_FN V wld(I off, int imm) { return X::wld(io.ws, iws + off + imm); }
    // masked-off lanes store into the unused gain slots of record 0 (there is no stage -1): no EXEC branch around the store
    RQ_FN void wst(I off, int imm, V v, M m) { X::wst(io.ws, iws + off + imm, iws + lane + RW_KK, v, m & owns & wrows); }
    RQ_FN void wld2(I off, int imm, V& a, V& b) { X::wld2(io.ws, iws + off + imm, a, b); }      // two consecutive values (even offset)
    RQ_FN void wst2(I off, int imm, V a, V b, M m) { X::wst2(io.ws, iws + off + imm, iws + (lane & 7) * 2 + RW_KK, a, b, m & owns & wrows); }

    // column layout of stage k: Gc[l] = G[l][c] for the lane's column c (rows 0..5)
    RQ_FN void load_gc(int k, V Gc[6]) { X::gld6(io.GT, igt + o_gc + k * RQ_GTS, Gc); }
    RQ_FN void fix_gc(V Gc[6]) {          // lanes 0, 1: unit columns e0, e1 of A
        Gc[0] = fma(Gc[0], keepc, unitc[0]); Gc[1] = fma(Gc[1], keepc, unitc[1]);
        RQ_UNROLL
        for (int l = 2; l < 6; ++l) Gc[l] = Gc[l] * keepc;
    }
    // row layout: Gr[c'] = G[lane][2 + c'] for c' = 0..6 (lanes 0..5; other lanes load a valid row and ignore it)
    RQ_FN void load_gr(int k, V Gr[7]) {
        RQ_UNROLL
        for (int c = 0; c < 7; ++c) Gr[c] = X::gld(io.GT, igt + o_gr + (k * RQ_GTS + c * 6));
    }
    // zn = G_k zz (zz = (z_0..6, u_0, u_1) on lanes 0..8), rows 0..5 from Gr, row 6 structural (delta' = delta + h u1)
    RQ_FN V apply_g(const V Gr[7], V zz) {
        V zn = splat((T)0);
        X::template dotbc<7, 2>(Gr, zz, zn);                     // columns 2..8
        zn = fma(zz, lt2f, zn);                                  // unit columns 0, 1
        const V u1 = X::template bc<8>(zz);
        return X::sel(is6, fma(u1, splat(q.h), zz), zn);
    }

    // =================================================================================================================
    // sweeps (every loop loads the data of the next stage before it works on the current one)
    // =================================================================================================================

    // S0 / SF: states rolled out through the linearised dynamics, dx_{k+1} = A dx_k + B du_k + b_k, xa = xbar + dx (S0: du = 0)
    RQ_FN void sweep_rollout(bool with_du) {
        const V zero = splat((T)0);
        V z;
        {
            const V x0v = X::gld(io.x0, ix0 + o_x);
            z = x0v - X::gld(io.xbar, ixb + o_x);
            wst(o_x + RW_XA, 0, x0v, is_x);
            wst(o_x + RW_D, 0, zero, is_x);
        }
        V Gr[7], Gn[7];
        load_gr(0, Gr);
        V bk = X::gld(io.bl, ibl + o_x), xk1 = X::gld(io.xbar, ixb + o_x + 7);
        V uk = zero;
        if (with_du) uk = ld(l_ua, rec(0)) - X::gld(io.ubar, iub + o_u);
        RQ_NOUNROLL
        for (int k = 0; k < N; ++k) {
            const int kn = k + 1 < N ? k + 1 : k;
            load_gr(kn, Gn);
            const V bn = X::gld(io.bl, ibl + o_x + kn * 7), xn1 = X::gld(io.xbar, ixb + o_x + (kn + 1) * 7);
            V un = zero;
            if (with_du) un = ld(l_ua, rec(kn)) - X::gld(io.ubar, iub + o_u + kn * 2);
            const V zz = X::sel(is_x, z, uk);
            const V zn = apply_g(Gr, zz) + bk;
            const V xa = xk1 + zn;
            wst(o_x + RW_XA, (k + 1) * RQ_RW, xa, is_x);
            if (k + 1 < N) st(X::isplat(RQ_X6), rec(k + 1), xa, is6);
            z = zn;
            RQ_UNROLL
            for (int c = 0; c < 7; ++c) Gr[c] = Gn[c];
            bk = bn; xk1 = xn1; uk = un;
        }
    }

    // terminal values of the backward sweeps: gx_N = W_e (xa_N - yref_e) on lanes < 7
    RQ_FN V terminal_gx() {
        const V xN = wld(o_x + RW_XA, N * RQ_RW), rN = X::gld(io.yref_e, iye + o_x);
        return X::sel(is_x, wqe * (xN - rN), splat((T)0));
    }

    // S1: backward Riccati sweep for the matrices, fused with the backward sweep of the gradient (gx from the point + X; gu = U).
    //     Leaves K0, K1, LI (workspace) and kff (U).
    RQ_FN void sweep_factor() {
        const V zero = splat((T)0);
        V P[7], p = terminal_gx();
        RQ_UNROLL
        for (int i = 0; i < 7; ++i) P[i] = E[i] * wqe;
        V Gc[6], Gn[6];
        load_gc(N - 1, Gc);
        V xk = wld(o_x + RW_XA, (N - 1) * RQ_RW), rk = X::gld(io.yref, iyr + o_y + (N - 1) * 9);
        V vu = ld(l_u, rec(N - 1)), va = ld(l_a, rec(N - 1)), vq = ld(X::isplat(RQ_Q1), rec(N - 1));
        RQ_NOUNROLL
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            load_gc(kn, Gn);
            const V xn = wld(o_x + RW_XA, kn * RQ_RW), rn = X::gld(io.yref, iyr + o_y + kn * 9);
            const V vun = ld(l_u, rec(kn)), van = ld(l_a, rec(kn)), vqn = ld(X::isplat(RQ_Q1), rec(kn));
            fix_gc(Gc);
            const V g = X::sel(is_x, fma(wq, xk - rk, X::sel(is6, va, zero)), vu);
            const V wd = X::sel(is_x, X::sel(is6, vq, wq), va);
            V G7[7];
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) G7[l] = Gc[l];
            G7[6] = g6c;
            V Mm[7];
            X::pg(P, G7, Mm);
            V H[9];
            X::gtm(Gc, Mm, q.h, H);
            RQ_UNROLL
            for (int r = 0; r < 9; ++r) H[r] = fma(E[r], wd, H[r]);
            const V h00 = X::template bc<7>(H[7]), h01 = X::template bc<7>(H[8]), h11 = X::template bc<8>(H[8]);
            const V idet = X::rcp(h00 * h11 - h01 * h01);
            const V i00 = h11 * idet, i01 = -(h01 * idet), i11 = h00 * idet;
            const V K0 = X::sel(is_x, -(i00 * H[7] + i01 * H[8]), zero);
            const V K1 = X::sel(is_x, -(i01 * H[7] + i11 * H[8]), zero);
            X::schur(H, K0, K1);                                   // H[i] += bc7(H[i]) K0 + bc8(H[i]) K1, i < 7
            V hv = g;
            X::template dotbc<7, 0>(G7, p, hv);
            const V hu0 = X::template bc<7>(hv), hu1 = X::template bc<8>(hv);
            const V pn = fma(K1, hu1, fma(K0, hu0, hv));
            const V kff = X::sel(is7, -(i00 * hu0 + i01 * hu1), -(i01 * hu0 + i11 * hu1));
            wst2(o_kk, (k + 1) * RQ_RW, X::sel(is_x, K0, X::sel(is7, i00, i01)), X::sel(is_x, K1, X::sel(is7, i01, i11)), lane < 9);
            st(l_u, rec(k), kff, is_u);
            RQ_UNROLL
            for (int i = 0; i < 7; ++i) P[i] = H[i];
            p = X::sel(is_x, pn, zero);
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) Gc[l] = Gn[l];
            xk = xn; rk = rn; vu = vun; va = van; vq = vqn;
        }
    }

    // S3: backward sweep of the gradient alone (corrector right-hand side: gu = U - smu * UB, gx6 term = X - smu * XB).
    //     A light loop (about 60 instructions per stage) behind L2-latency loads: PF stages are kept in flight.
    struct BwdIn { V Gc[6], xk, rk, vu, va, vb, ka, kb; };
    RQ_FN void bwd_load(int k, BwdIn& B) {
        load_gc(k, B.Gc);
        B.xk = wld(o_x + RW_XA, k * RQ_RW); B.rk = X::gld(io.yref, iyr + o_y + k * 9);
        B.vu = ld(l_u, rec(k)); B.va = ld(l_a, rec(k));
        B.vb = wld(o_ub, (k + 1) * RQ_RW); wld2(o_kk, (k + 1) * RQ_RW, B.ka, B.kb);
    }
    RQ_FN void bwd_stage(int k, BwdIn& c, V smu, V& p) {
        const V zero = splat((T)0);
        fix_gc(c.Gc);
        const V K0 = X::sel(is_x, c.ka, zero), K1 = X::sel(is_x, c.kb, zero);
        const V g = X::sel(is_x, fma(wq, c.xk - c.rk, X::sel(is6, c.va - smu * c.vb, zero)), c.vu - smu * c.vb);
        V G7[7];
        RQ_UNROLL
        for (int l = 0; l < 6; ++l) G7[l] = c.Gc[l];
        G7[6] = g6c;
        V hv = g;
        X::template dotbc<7, 0>(G7, p, hv);
        const V hu0 = X::template bc<7>(hv), hu1 = X::template bc<8>(hv);
        const V pn = fma(K1, hu1, fma(K0, hu0, hv));
        const V kff = -(c.ka * hu0 + c.kb * hu1);                 // lanes 7, 8: their rows of the inverse of Huu
        st(l_u, rec(k), kff, is_u);
        p = X::sel(is_x, pn, zero);
    }
    RQ_FN void sweep_backward(V smu) {
        V p = terminal_gx();
        BwdIn buf[PF];
        RQ_UNROLL
        for (int j = 0; j < PF; ++j) bwd_load(N - 1 - j > 0 ? N - 1 - j : 0, buf[j]);
        int k0 = N - 1;
        RQ_NOUNROLL
        for (; k0 - (PF - 1) >= 0; k0 -= PF) {                // full groups: straight-line code, the ring slots are consumed in
            RQ_UNROLL                                               // place and reloaded (no register rotation, exact wait counts)
            for (int j = 0; j < PF; ++j) {
                const int k = k0 - j;
                bwd_stage(k, buf[j], smu, p);
                bwd_load(k - PF > 0 ? k - PF : 0, buf[j]);
            }
        }
        RQ_UNROLL
        for (int j = 0; j < PF - 1; ++j)                         // the last N mod PF stages (their data is in the first slots)
            if (k0 - j >= 0) bwd_stage(k0 - j, buf[j], smu, p);
    }

    // SA: exact adjoint of the current point: returns the max-norm of the reduced gradient and, with want_pi, writes the adjoint
    //     (= dynamics multipliers pi_k of the iterate snapshot) to io.pi [B][N][7] on the rows of pim.
    //     Raw gradients: gu = R (u - uref) - lam_l + lam_u, gx = w (x - ref), steering multipliers on component 6.
    RQ_FN V sweep_adjoint(bool want_pi, M pim) {
        const V zero = splat((T)0);
        V lam = terminal_gx();
        V rg = zero;
        V Gc[6], Gn[6];
        load_gc(N - 1, Gc);
        V xk = wld(o_x + RW_XA, (N - 1) * RQ_RW), rk = X::gld(io.yref, iyr + o_y + (N - 1) * 9);
        RQ_NOUNROLL
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            load_gc(kn, Gn);
            const V xn = wld(o_x + RW_XA, kn * RQ_RW), rn = X::gld(io.yref, iyr + o_y + kn * 9);
            if (want_pi) X::gst(io.pi, ipi + o_x + k * 7, lam, is_x & pim);   // pi_k multiplies dx_{k+1} = A dx_k + B du_k + b_k
            const V ua = ld(l_ua, rec(k));
            const V l0 = ld(l_l0, rec(k)), l1 = ld(l_l0, rec(k) + 1);        // lane 6: steering pair; lanes 7, 8: bound pair of input j
            fix_gc(Gc);
            const V g = X::sel(is_x, fma(wq, xk - rk, X::sel(is6, l1 - l0, zero)), fma(rjm, ua - rk, l1 - l0));
            V G7[7];
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) G7[l] = Gc[l];
            G7[6] = g6c;
            V lv = g;
            X::template dotbc<7, 0>(G7, lam, lv);
            rg = X::vmaxnan(rg, X::sel(is_u, X::vabs(lv), zero));
            lam = X::sel(is_x, lv, zero);
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) Gc[l] = Gn[l];
            xk = xn; rk = rn;
        }
        if (want_pi) X::gst(io.pi, ipi + o_x + N * 7, lam, is_x & pim);       // adjoint at stage 0 = multiplier of dx_0 = x0 - xbar_0
        return X::row_maxnan(rg);
    }

    // S2 / S4: forward roll-out of the Newton step: ddu_k = K_k ddx_k + kff_k (into U), ddx_{k+1} = A ddx_k + B ddu_k.
    //     full = false (predictor): only ddx6_{k+1} is kept (Q1 of stage k+1);  full = true: ddx_{k+1} into D, ddx6 also into X.
    //     A light loop (about 50 instructions per stage) behind L2-latency loads: PF stages are kept in flight.
    struct FwdIn { V Gr[7], ka, kb, kff; };
    RQ_FN void fwd_load(int k, FwdIn& F) {
        load_gr(k, F.Gr);
        wld2(o_kk, (k + 1) * RQ_RW, F.ka, F.kb);
        F.kff = ld(l_u, rec(k));
    }
    RQ_FN void fwd_stage(int k, FwdIn& c, bool full, V& z) {
        V ddu = c.kff;                                             // ddu_j = kff_j + sum_c K_j[c] z_c: the products are formed where
        X::sumbc2(c.ka * z, c.kb * z, E[7], E[8], ddu);            // column c lives (lane c) and gathered on lanes 7, 8
        const V zz = X::sel(is_x, z, ddu);
        const V zn = apply_g(c.Gr, zz);
        st(l_u, rec(k), ddu, is_u);
        if (full) wst(o_x + RW_D, (k + 1) * RQ_RW, zn, is_x);
        st(X::isplat(full ? RQ_X : RQ_Q1), rec(k + 1 < N ? k + 1 : k), zn, is6 & X::mfrom(k + 1 < N));
        z = X::sel(is_x, zn, splat((T)0));
    }
    RQ_FN void sweep_forward(bool full) {
        V z = splat((T)0);
        FwdIn buf[PF];
        RQ_UNROLL
        for (int j = 0; j < PF; ++j) fwd_load(j < N ? j : N - 1, buf[j]);
        int k0 = 0;
        RQ_NOUNROLL
        for (; k0 + PF <= N; k0 += PF) {                      // full groups: straight-line code (see sweep_backward)
            RQ_UNROLL
            for (int j = 0; j < PF; ++j) {
                const int k = k0 + j;
                fwd_stage(k, buf[j], full, z);
                fwd_load(k + PF < N ? k + PF : N - 1, buf[j]);
            }
        }
        RQ_UNROLL
        for (int j = 0; j < PF - 1; ++j)
            if (k0 + j < N) fwd_stage(k0 + j, buf[j], full, z);
    }

    // =================================================================================================================
    // passes over the inequalities (lane = side, two stages per step; the loads of step s+1 are issued before step s is worked on)
    // =================================================================================================================
    struct Side {                 // what a side lane holds for its stage
        V tb, lb, ts, ls;         // slack / multiplier of the bound pair and of the slack pair (inputs)
        V vabs, uref;             // u_kj (inputs) or delta_k (steering); uref of the input
        V sa, sc;                 // raw step slots: predictor (A | Q1) and corrector / trial (U | X)
        M act;                    // this side exists (stage < N, steering only on stages >= 1)
        M stv;                    // the even lane of the pair writes the stage outputs (stage < N)
        M inb;                    // valid side lane of a stage < N (whether or not the side exists)
        I kl;                     // LDS offset of the stage's record (clamped stage)
        I kc;                     // clamped stage index
    };

    RQ_FN void side_load(int s, Side& S, bool with_state, bool with_steps) {
        const I k = sp + 2 * s;
        const M in = k < N;
        S.kc = X::isel(in, k, X::isplat(N - 1));
        S.kl = S.kc * RQ_RS + RQ_HDR;
        S.act = e_valid & in & ((!e_isd) | (k >= 1));
        S.stv = e_valid & in & e_even;
        S.inb = e_valid & in;
        S.uref = ld(S.kl + jin, RQ_UR);
        S.vabs = ld(S.kl + o_vl, 0);
        if (with_state) {
            S.tb = ld(S.kl + o_tb, RQ_T); S.lb = ld(S.kl + o_tb, RQ_LAM);
            S.ts = ld(S.kl + o_ts, RQ_T); S.ls = ld(S.kl + o_ts, RQ_LAM);
        }
        if (with_steps) { S.sa = ld(S.kl + o_sa, 0); S.sc = ld(S.kl + o_sc, 0); }
    }

    struct Bar {                  // barrier quantities of a side
        V itb, its, Gb, Gs, is, rd, e, psi, Rtc;
    };
    RQ_FN void side_barrier(const Side& S, Bar& B) {
        const V zero = splat((T)0);
        const V qv = sgn * (S.vabs - bound);
        B.rd = qv + X::sel(e_isd, zero, S.ts) - S.tb;
        B.itb = X::rcp(S.tb); B.Gb = S.lb * B.itb;
        B.its = X::sel(e_isd, zero, X::rcp(S.ts)); B.Gs = S.ls * B.its;       // steering sides have no slack pair
        B.is = X::sel(e_isd, zero, X::rcp(B.Gb + B.Gs));
        B.Rtc = X::sel(e_isd, B.Gb, B.Gb * B.Gs * B.is);
        B.e = fma(B.Gb, B.rd, rho);
        B.psi = B.Gb * (B.rd - B.e * B.is);
    }

    struct Red { V mu, cmax, rmax; };     // per-lane partial sums of a pass

    // E1 on in-register state: complementarity sums, primal residuals, predictor right-hand side (U, A, X, Q1)
    RQ_FN void e1_core(const Side& S, Red& R) {
        const V zero = splat((T)0);
        Bar B; side_barrier(S, B);
        const V cb = S.tb * S.lb, cs = X::sel(e_isd, zero, S.ts * S.ls);
        R.mu = R.mu + X::sel(S.act, cb + cs, zero);
        R.cmax = X::vmax(R.cmax, X::sel(S.act, X::vmax(cb, cs), zero));
        const V rs = X::sel(e_isd, zero, X::vabs(rho - S.lb - S.ls));
        R.rmax = X::vmaxnan(R.rmax, X::sel(S.act, X::vmaxnan(X::vabs(B.rd), rs), zero));
        const V ps = X::sel(S.act, sgn * B.psi, zero), rt = X::sel(S.act, B.Rtc, zero);
        const V gs = ps + X::swap1(ps), Rs = rt + X::swap1(rt);
        // inputs: U = R (u - uref) + gs, A = R + Rs ; steering: X = gs, Q1 = Qd6 + Rs
        st(S.kl + o_o1, 0, X::sel(e_isd, gs, fma(rjin, S.vabs - S.uref, gs)), S.stv);
        st(S.kl + o_o2, 0, q6orr + Rs, S.stv);
    }

    RQ_FN void pass_e1(Red& R) {
        R.mu = splat((T)0); R.cmax = splat((T)0); R.rmax = splat((T)0);
        Side S, Sn;
        side_load(0, S, true, false);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            side_load(2 * (s + 1) < N ? s + 1 : s, Sn, true, false);
            X::sched_fence(S.tb, S.lb, S.ts, S.ls);                 // the loads of the next step stay ahead of this step's work
            e1_core(S, R);
            S = Sn;
        }
        R.mu = X::row_sum(R.mu); R.cmax = X::row_max(R.cmax); R.rmax = X::row_maxnan(R.rmax);
    }

    // predictor step of a side from uua = sgn * (ddu or ddx6): dt, dlam of both pairs
    struct Step { V dtb, dts, dlb, dls; };
    RQ_FN void side_step_pred(const Side& S, const Bar& B, V uua, Step& P) {
        const V ds = -((B.e + B.Gb * uua) * B.is);
        P.dtb = uua + ds + B.rd; P.dts = ds;
        P.dlb = -S.lb - B.Gb * P.dtb; P.dls = -S.ls - B.Gs * P.dts;
    }

    // E2: after the predictor solve (ddu in U, ddx6 in Q1).  Ratio test + sum dt*dlam (per-lane partials), moves the predictor
    //     step of the inputs to A and leaves the corrector right-hand side split as  gA - smu * gB  (U / UB, X / XB).
    RQ_FN void pass_e2(V& rr, V& s2) {
        const V zero = splat((T)0), one = splat((T)1);
        rr = zero; s2 = zero;
        Side S, Sn;
        side_load(0, S, true, true);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            side_load(2 * (s + 1) < N ? s + 1 : s, Sn, true, true);
            X::sched_fence(S.tb, S.lb, S.ts, S.ls);
            Bar B; side_barrier(S, B);
            const V stp = X::sel(e_isd, S.sa, S.sc);               // ddx6_a (Q1) | ddu_a (U)
            const V uua = sgn * stp;
            Step P; side_step_pred(S, B, uua, P);
            const V xb = P.dtb * B.itb, xs = P.dts * B.its;
            V r = X::vmax(-xb, one + xb);
            r = X::vmax(r, X::sel(e_isd, zero, X::vmax(-xs, one + xs)));
            rr = X::vmax(rr, X::sel(S.act, r, zero));
            const V mb = P.dtb * P.dlb, ms = X::sel(e_isd, zero, P.dts * P.dls);
            s2 = s2 + X::sel(S.act, mb + ms, zero);
            const V mbi = w2 * mb * B.itb, msi = w2 * ms * B.its;      // second-order term of the corrector (dropped in fallback mode)
            const V P1 = mbi - B.Gb * B.is * (mbi + msi);
            const V P2 = B.itb - B.Gb * B.is * (B.itb + B.its);
            const V pa = X::sel(S.act, sgn * (B.psi + P1), zero), pb = X::sel(S.act, sgn * P2, zero);
            const V sa = pa + X::swap1(pa), sb = pb + X::swap1(pb);
            st(S.kl + jin + RQ_A, 0, stp, S.stv & e_in);                                              // predictor ddu
            st(S.kl + o_o1, 0, X::sel(e_isd, sa, fma(rjin, S.vabs - S.uref, sa)), S.stv);
            wst((S.kc + 1) * RQ_RW + X::isel(e_isd, X::isplat(RW_XB), jin + RW_UB), 0, sb, S.stv);
            S = Sn;
        }
        rr = X::row_max(rr); s2 = X::row_sum(s2);
    }

    // corrector step of a side (predictor recomputed from its stored inputs)
    RQ_FN void side_step_corr(const Side& S, const Bar& B, V smu, Step& C) {
        const V zero = splat((T)0);
        const V uua = sgn * S.sa, uu = sgn * S.sc;
        Step P; side_step_pred(S, B, uua, P);
        const V mb = w2 * (P.dtb * P.dlb), ms = w2 * (P.dts * P.dls);
        const V cb = (mb - smu) * B.itb, cs = X::sel(e_isd, zero, (ms - smu) * B.its);
        const V ec = B.e + cb + cs;
        const V ds = -((ec + B.Gb * uu) * B.is);
        C.dtb = uu + ds + B.rd; C.dts = ds;
        C.dlb = -(S.lb + cb) - B.Gb * C.dtb; C.dls = -(S.ls + cs) - B.Gs * C.dts;
    }

    // E3a: ratio test of the corrector step (ddu in U, ddx6 in X; predictor steps in A, Q1)
    RQ_FN V pass_e3a(V smu) {
        const V zero = splat((T)0);
        V rr = zero;
        Side S, Sn;
        side_load(0, S, true, true);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            side_load(2 * (s + 1) < N ? s + 1 : s, Sn, true, true);
            X::sched_fence(S.tb, S.lb, S.ts, S.ls);
            Bar B; side_barrier(S, B);
            Step C; side_step_corr(S, B, smu, C);
            V r = X::vmax(-(C.dtb * B.itb), -(C.dlb * X::rcp(S.lb)));
            r = X::vmax(r, X::sel(e_isd, zero, X::vmax(-(C.dts * B.its), -(C.dls * X::rcp(S.ls)))));
            rr = X::vmax(rr, X::sel(S.act, r, zero));
            S = Sn;
        }
        return X::row_max(rr);
    }

    // xa += alpha ddx for every stage (lanes (sp, i < 7) <-> record 2s + sp + 1); the steering angle is mirrored into the LDS record.
    // Four steps per group: their eight loads are in flight together (a step-by-step loop waits a full memory latency per step).
    RQ_FN void pass_dx_update(V alpha, M rowact) {
        const M go = rowact & (alpha > splat((T)0));
        RQ_NOUNROLL
        for (int s0 = 0; 2 * s0 < N; s0 += 4) {
            V xa[4], dd[4]; I kc[4];
            RQ_UNROLL
            for (int j = 0; j < 4; ++j) {
                const I k = sp + 2 * (s0 + j);
                kc[j] = X::isel(k < N, k, X::isplat(N - 1));
                const I wo = (kc[j] + 1) * RQ_RW + o_dxu;
                xa[j] = wld(wo, RW_XA); dd[j] = wld(wo, RW_D);
            }
            RQ_UNROLL
            for (int j = 0; j < 4; ++j) {
                const I k = sp + 2 * (s0 + j);
                const M in = (k < N) & dxl & go;
                const V xn = fma(alpha, dd[j], xa[j]);
                wst((kc[j] + 1) * RQ_RW + o_dxu, RW_XA, xn, in);
                st((kc[j] + 1) * RQ_RS + RQ_HDR, RQ_X6, xn, in & (side == 6) & (k < N - 1));
            }
        }
    }

    // E3b + E1: apply the corrector step with step length alpha on the rows of rowact, then the quantities of the next iteration
    //     from the updated state, still in registers.  Returns the row's max |alpha ddu| (fp32: max |ddu|).
    RQ_FN V pass_e3b_e1(V smu, V alpha, M rowact, Red& R) {
        const V zero = splat((T)0);
        V stp = zero;
        R.mu = zero; R.cmax = zero; R.rmax = zero;
        Side S, Sn;
        side_load(0, S, true, true);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            side_load(2 * (s + 1) < N ? s + 1 : s, Sn, true, true);   // old state of the next step (its records are not written here)
            X::sched_fence(S.tb, S.lb, S.ts, S.ls);
            Bar B; side_barrier(S, B);
            Step C; side_step_corr(S, B, smu, C);
            const V fl = splat(q.floor_);
            // rows that are not iterating keep their state bit for bit (their recomputed steps are never applied)
            S.tb = X::sel(rowact, X::vmax(fma(alpha, C.dtb, S.tb), fl), S.tb); S.lb = X::sel(rowact, X::vmax(fma(alpha, C.dlb, S.lb), fl), S.lb);
            S.ts = X::sel(rowact, X::vmax(fma(alpha, C.dts, S.ts), fl), S.ts); S.ls = X::sel(rowact, X::vmax(fma(alpha, C.dls, S.ls), fl), S.ls);
            const V au = X::sel(rowact, alpha * S.sc, zero);                 // alpha * ddu_j (inputs), alpha * ddx6 (steering)
            // what the step test sees.  fp64: the step taken, alpha * ddu (the oracle's rule).  fp32: the Newton step itself -- its looser
            // complementarity / residual levels are no safety net, and a BLOCKED step (alpha ~ 1e-3) is short without being converged: one
            // instance of 196 608 then stopped 0.3 off the minimiser with status 0 (found when a different FMA contraction moved the
            // rounding: profiles/r4/f32_step_test.txt)
            const V sv = sizeof(T) == 4 ? X::sel(rowact, S.sc, zero) : au;
            stp = X::vmax(stp, X::sel(S.act & e_in, X::vabs(sv), zero));
            S.vabs = S.vabs + au;
            st(S.kl + jin + RQ_UA, 0, S.vabs, S.stv & e_in);
            st(S.kl + o_tb, RQ_T, S.tb, S.act); st(S.kl + o_tb, RQ_LAM, S.lb, S.act);
            st(S.kl + o_ts, RQ_T, S.ts, S.act & e_in); st(S.kl + o_ts, RQ_LAM, S.ls, S.act & e_in);
            e1_core(S, R);
            S = Sn;
        }
        pass_dx_update(alpha, rowact);
        R.mu = X::row_sum(R.mu); R.cmax = X::row_max(R.cmax); R.rmax = X::row_maxnan(R.rmax);
        return X::row_max(stp);
    }

    // trial set-up: the QP without its inequalities from the start point (ua = ubar, rolled-out states): U = R (ubar - uref), A = R,
    // X = 0, Q1 = Qd6
    RQ_FN void pass_trial_setup() {
        const V zero = splat((T)0);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            const I k = sp + 2 * s;
            const M in = k < N;
            const I kc = X::isel(in, k, X::isplat(N - 1));
            const I kl = kc * RQ_RS + RQ_HDR;
            const M stv = e_valid & in & e_even;
            const V ub = X::gld(io.ubar, iub + kc * 2 + jin), ur = X::gld(io.yref, iyr + kc * 9 + 7 + jin);
            st(kl + o_o1, 0, X::sel(e_isd, zero, rjin * (ub - ur)), stv);
            st(kl + o_o2, 0, q6orr, stv);
            st(kl + jin + RQ_UA, 0, ub, stv & e_in);
            st(kl + jin + RQ_UR, 0, ur, stv & e_in);
        }
    }

    // does the step in U / X respect every bound?
    RQ_FN M pass_trial_check(V& nviol) {
        M ok = X::mtrue();
        V nv = splat((T)0);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, false, true);
            const V qv = sgn * (S.vabs + S.sc - bound);
            const M in = (!S.act) | (qv >= splat((T)0));
            ok = ok & in;
            nv = nv + X::sel(in, splat((T)0), splat((T)1));
        }
        nviol = X::row_sum(nv);
        return X::row_and(ok);
    }

    // start of the interior point: ua += a0 * ddu, xa += a0 * ddx (a0 = 1: from the trial's minimiser, 0: zero step), then
    // slacks and multipliers (th = clip level; warm: a violated input bound is absorbed by its slack).
    // Returns the max-norm of the stationarity residual of this start point, as the oracle's ipm_residuals() sees it: with the
    // zero step its dynamics multipliers are zero (ru = r - lam_l + lam_u, rx = Q dx + q + steering multipliers); from the
    // trial's minimiser they are that minimiser's exact multipliers, which leaves the inequality multipliers alone.
    RQ_FN V pass_init(V a0, V th, M warm, M solved) {
        const V zero = splat((T)0);
        const M step = a0 > zero;
        const V gsc = X::sel(step, zero, splat((T)1));            // weight of the plain gradient in the residual
        V rs0 = zero;
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, false, true);
            const V vabs = S.vabs + X::sel(step, S.sc, zero);      // steering: X6 still holds the start point, X the trial's ddx6
            st(S.kl + jin + RQ_UA, 0, vabs, S.stv & e_in);
            const V qv = sgn * (vabs - bound);
            const V ts = X::sel(warm, X::vmax(-qv, zero), zero) + th;
            const V tb = X::vmax(qv + X::sel(e_isd, zero, ts), th);
            const V mu0 = splat(q.mu0);
            // rows whose trial minimiser is feasible are done: no bound is active (multiplier 0, slack = distance to the bound), the
            // slack pairs carry the L1 weight (multiplier rho, slack 0) -- what the iterate snapshot reports for them
            const V lb = X::sel(S.act & !solved, mu0 * X::rcp(tb), zero);
            st(S.kl + o_tb, RQ_T, X::sel(S.act, X::sel(solved, qv, tb), splat((T)1)), S.inb);
            st(S.kl + o_tb, RQ_LAM, lb, S.inb);
            st(S.kl + o_ts, RQ_T, X::sel(S.act, X::sel(solved, zero, ts), splat((T)1)), S.inb & e_in);
            st(S.kl + o_ts, RQ_LAM, X::sel(S.act, X::sel(solved, rho, mu0 * X::rcp(ts)), zero), S.inb & e_in);
            // stationarity row of the pair: plain gradient (zero step only) - lam_lower + lam_upper
            const V ml = -(sgn * lb);
            const V ref6 = X::gld(io.yref, iyr + S.kc * 9 + 6);
            const V gpl = X::sel(e_isd, splat(q.Qd[6]) * (vabs - ref6), rjin * (vabs - S.uref));
            const V row = fma(gsc, gpl, ml + X::swap1(ml));
            rs0 = X::vmaxnan(rs0, X::sel(S.stv & (e_in | S.act), X::vabs(row), zero));
        }
        // the other state rows (zero step only): w (xa - ref), stages 1..N; component 6 of stages 1..N-1 is covered above
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            const I k1 = sp + (2 * s + 1);                         // stage of the state row, 1..N
            const M in = (k1 <= N) & dxl;
            const I kc = X::isel(k1 <= N, k1, X::isplat(N));
            const M term = kc == N;
            const V xa = wld(kc * RQ_RW + o_dxu, RW_XA);
            const V rr = X::gld(io.yref, iyr + X::isel(term, X::isplat(0), kc) * 9 + o_dxu), re = X::gld(io.yref_e, iye + o_dxu);
            const V gx = X::sel(term, wesd, wsd) * (xa - X::sel(term, re, rr));
            rs0 = X::vmaxnan(rs0, X::sel(in & (term | !(side == 6)), X::vabs(gsc * gx), zero));
        }
        pass_dx_update(a0, X::mtrue());                            // after the loops above: they read the start point
        return X::row_maxnan(rs0);
    }

    // iterate snapshot: slacks and multipliers of every stage, record order (lanes 0..9 <-> pairs)
    RQ_FN void pass_snapshot(M pim) {
        const M m = (lane < 10) & pim;
        const I l10 = X::isel(lane < 10, lane, X::isplat(0));
        RQ_NOUNROLL
        for (int k = 0; k < N; ++k) {
            X::gst(io.ineq, iiq + l10 + k * 20, ld(l10 + RQ_T, rec(k)), m);
            X::gst(io.ineq, iiq + l10 + (k * 20 + 10), ld(l10 + RQ_LAM, rec(k)), m);
        }
    }

    // =================================================================================================================
    // one instance
    // =================================================================================================================
    // deferred / nviol (solve mode 1 only): the trial's minimiser leaves a bound -- nothing of this row has been written; nviol = number
    // of violated bounds (sort key of the second phase)
    struct Result { M failed; I iters; V rmax; M deferred; V nviol; };

    // The rows of `rows` start over from the cold start (zero input step, rolled-out states, slacks at thr); the other rows of the wave
    // keep their records bit for bit (write mask).  Leaves the quantities of the next iteration (R, rstat, step, alpha_prev) for them.
    RQ_FN void restart_cold(M rows, Red& R, V& rstat, V& step, V& alpha_prev) {
        const V zero = splat((T)0);
        wrows = rows;
        sweep_rollout(false);
        X::fence();
        pass_trial_setup();
        X::fence();
        const V rs0 = pass_init(zero, splat(q.thr), X::mfalse(), X::mfalse());
        X::fence();
        Red R2;
        pass_e1(R2);
        X::fence();
        wrows = X::mtrue();
        R.mu = X::sel(rows, R2.mu, R.mu); R.cmax = X::sel(rows, R2.cmax, R.cmax); R.rmax = X::sel(rows, R2.rmax, R.rmax);
        rstat = X::sel(rows, rs0, rstat);
        step = X::sel(rows, splat((T)1e30), step);
        alpha_prev = X::sel(rows, splat((T)1), alpha_prev);
    }

    // split batches: the row's LDS region (after the trial: inputs, their references, the trial's steps, gains ...) to and from io.dump
    RQ_FN void lds_dump(M rows) {
        const int used = RQ_HDR + N * RQ_RS;
        RQ_NOUNROLL
        for (int j = 0; j * 16 < used; ++j) {
            const I o = lane + X::isplat(16 * j);
            const M in = o < X::isplat(used);
            const I oc = X::isel(in, o, X::isplat(0));
            X::gst(io.dump, idp + oc, ld(oc, 0), in & rows & owns);
        }
    }
    RQ_FN void lds_restore() {
        const int used = RQ_HDR + N * RQ_RS;
        RQ_NOUNROLL
        for (int j = 0; j * 16 < used; ++j) {
            const I o = lane + X::isplat(16 * j);
            const M in = o < X::isplat(used);
            const I oc = X::isel(in, o, X::isplat(0));
            st(oc, 0, X::gld(io.dump, idp + oc), in);
        }
    }

    // valid: the row carries an instance to solve.  want_pi (wave-uniform): also write the multipliers of the returned iterate
    // (io.pi, io.ineq) on the rows of pim.
    // mode (wave-uniform) 0: the whole solve.  1: first phase of a split batch -- roll-out and unconstrained trial only; rows whose
    // trial fails are reported in res.deferred and must not be finished: their LDS region goes to io.dump, their workspace records
    // (absolute states, the trial's state steps) stay as they are, nothing else of them is written.  2: second phase -- every row is
    // such an instance: the region is restored and the solve continues behind the trial, exactly where mode 0 would be (rows of one
    // wave now come packed with their like: a wave iterates until its slowest row has converged).
    RQ_FN void solve(M valid, Result& res, bool want_pi, M pim, int mode) {
        const V zero = splat((T)0), one = splat((T)1);
        X::stamp(0);
        M active = valid, failed = X::mfalse(), warmrow = X::mfalse();
        I iters = X::isplat(0);
        res.deferred = X::mfalse(); res.nviol = zero;
        V rstat;                                                   // stationarity residual of the interior point's iterate (tracked)
        if (mode == 2) {
            lds_restore();
            X::fence();
            const M warm = X::mfrom(q.thw > (T)0);
            rstat = pass_init(X::sel(warm, one, zero), X::sel(warm, splat(q.thw), splat(q.thr)), warm, X::mfalse());
            warmrow = warm;
            X::stamp(2);
        } else {
        sweep_rollout(false);
        X::fence();
        X::stamp(1);
        pass_trial_setup();
        X::fence();
        if (q.try_unc) {
            sweep_factor();
            X::fence();
            sweep_forward(true);
            X::fence();
            const M ok = pass_trial_check(res.nviol);
            const M warm = (!ok) & X::mfrom(q.thw > (T)0);
            const V a0 = X::sel(ok | warm, one, zero);
            if (mode == 1) {                                       // deferred rows: region to io.dump, then frozen (LDS and workspace) for the rest of this launch
                res.deferred = active & !ok;
                if (X::any(res.deferred)) lds_dump(res.deferred);
                wrows = !res.deferred;
                pim = pim & !res.deferred;
            }
            rstat = pass_init(a0, X::sel(warm, splat(q.thw), splat(q.thr)), warm, ok);
            active = active & !ok;
            if (mode == 1) active = X::mfalse();
            warmrow = warm;
            X::stamp(2);
        } else {
            rstat = pass_init(zero, splat(q.thr), X::mfalse(), X::mfalse());
        }
        }
        X::fence();
        Red R;
        R.mu = zero; R.cmax = zero; R.rmax = zero;
        if (mode != 1) {                                           // (phase 1 of a split batch iterates nothing: no statistics, no predictor right-hand side)
            pass_e1(R);
            X::fence();
        }
        X::stamp(3);
        V step = splat((T)1e30), rmax_prev = zero, rmax_last = zero, alpha_prev = one;
        M cons = X::mfalse();                                      // rows in fallback mode
        w2 = one;
        RQ_NOUNROLL
        for (int guard = 0; guard <= q.itmax + q.fbit; ++guard) {
            // every linear residual of a Newton iteration in residual form shrinks by (1 - alpha) per step; the inequality rows are
            // re-evaluated (R.rmax), the stationarity rows (which would need the dynamics multipliers) are tracked: rstat
            V mu = R.mu * splat(q.inv_nineq);
            const V rmax = X::vmaxnan(R.rmax, rstat);
            const M nan = (!(mu == mu)) | (!(rmax == rmax));
            failed = failed | (active & nan);
            active = active & !nan;
            const M conv = active & (R.cmax <= splat(q.tol_comp)) & (step <= splat(q.tol_step)) &
                           ((rmax <= splat(q.tol_res)) | (X::mfrom(guard > 0) & (rmax > splat((T)0.1) * rmax_prev) & (rmax <= splat((T)ADMPC_IPM_FLOOR_CAP * q.tol_res))));      // admpc.h: stopping test
            rmax_last = X::sel(active, rmax, rmax_last);
            rmax_prev = rmax;
            active = active & !conv & (iters < X::isel(cons, X::isplat(q.itmax + q.fbit), X::isplat(q.itmax)));
            if (!X::any(active)) break;
            if (q.fbit > 0) {
                // cfg.ipm_fallback_iter: a row that is still iterating has most likely fallen into a limit cycle of the centring
                // heuristic: it starts over and finishes with plain predictor-centring steps (no second-order term), on a budget of its own
                const M fb = active & !cons & (iters >= q.fbit);
                if (X::any(fb)) {
                    restart_cold(fb, R, rstat, step, alpha_prev);
                    cons = cons | fb;
                    w2 = X::sel(cons, zero, one);
                    warmrow = warmrow & !fb;
                    mu = R.mu * splat(q.inv_nineq);
                }
            }
            // ---- predictor
            X::stamp(4);
            sweep_factor();
            X::fence();
            X::stamp(5);
            sweep_forward(false);
            X::fence();
            X::stamp(6);
            V rr, s2;
            pass_e2(rr, s2);
            X::fence();
            X::stamp(7);
            const V a_aff = X::sel(rr > one, X::rcp(rr), one);
            const V munq = R.mu;                                    // = mu * nineq
            const V mu_aff = ((one - a_aff) * munq + a_aff * a_aff * s2) * splat(q.inv_nineq);
            V sigma = mu_aff * X::rcp(mu); sigma = sigma * sigma * sigma;
            sigma = X::sel(alpha_prev < splat(q.blocked), one, sigma);   // after a blocked step: centre
            const V smu = X::vmax(sigma * mu, splat(q.mu_floor));      // admpc.h: centring target floor
            RQ_DBG("[emu] it mu=%.6e cmax=%.3e rmax=%.3e a_aff=%.6e mu_aff=%.6e sigma=%.6e\n", X::first(mu), X::first(R.cmax), X::first(rmax), X::first(a_aff), X::first(mu_aff), X::first(sigma));
            // ---- corrector
            sweep_backward(smu);
            X::fence();
            X::stamp(8);
            sweep_forward(true);
            X::fence();
            X::stamp(9);
            const V rc = pass_e3a(smu);
            X::stamp(10);
            const V amax = X::sel(rc > one, X::rcp(rc), one);
            V tau = one - mu_aff; tau = X::vmax(tau, splat((T)0.995)); tau = X::vmin(tau, splat((T)0.999999));
            const V alpha0 = X::sel(active, X::vmin(tau * amax, one), zero);
            // a warm start whose very first step is blocked (cfg.ipm_warm_restart) is abandoned: the row starts over from the cold
            // start (zero input step, rolled-out states), the iteration counts.  From a minimiser far outside the hard steering
            // box the method otherwise creeps for ten iterations (N = 80: 24 iterations warm, 17 cold)
            M rst = X::mfalse();
            if (guard == 0 && q.wrest > (T)0) rst = active & warmrow & (alpha0 < splat(q.wrest));
            const M upd = active & !rst;
            const V alpha = X::sel(rst, zero, alpha0);
            RQ_DBG("[emu]      alpha=%.6e\n", X::first(alpha));
            const V stn = pass_e3b_e1(smu, alpha, upd, R);
            X::fence();
            X::stamp(11);
            step = X::sel(upd, stn, step);
            alpha_prev = X::sel(upd, alpha, alpha_prev);
            rstat = X::sel(upd, (one - alpha) * rstat, rstat);
            iters = iters + X::isel(active, X::isplat(1), X::isplat(0));
            if (X::any(rst)) restart_cold(rst, R, rstat, step, alpha_prev);
        }
        res.failed = failed; res.iters = iters; res.rmax = rmax_last;
        // ---- H6: expand the states from the input step through the linearised dynamics (as acados' expand step)
        X::stamp(12);
        sweep_rollout(true);
        X::fence();
        X::stamp(13);
        if (want_pi) { (void)sweep_adjoint(true, pim); pass_snapshot(pim); }
    }

    // full step, cost, outputs.  Call after solve(); `write`: rows whose iterate may be overwritten when the step is finite.
    // (The stop of an SQP solve with a tolerance is decided in front of the QP, on acados' four residuals: admpc_nlp_res_kernel.)
    RQ_FN void finish(M write, M& failed, V& cost) {
        const V zero = splat((T)0), half = splat((T)0.5);
        M bad = X::mfalse();
        V J = zero;
        RQ_NOUNROLL
        for (int pass = 0; pass < 2; ++pass) {
            const M wr = write & !failed;
            if (pass == 1 && !X::any(wr)) break;
            // states: lanes (sp, i < 7) <-> x_k, k = 2s + sp over 0..N
            RQ_NOUNROLL
            for (int s = 0; 2 * s < N + 1; ++s) {
                const I k = sp + 2 * s;
                const M in = (k <= N) & dxl;
                const I kc = X::isel(k <= N, k, X::isplat(N));
                const V xn = wld(kc * RQ_RW + o_dxu, RW_XA);
                if (pass == 0) {
                    const M term = kc == N;
                    const I kr = X::isel(term, X::isplat(0), kc);
                    const V rr = X::gld(io.yref, iyr + kr * 9 + o_dxu), re = X::gld(io.yref_e, iye + o_dxu);
                    const V e = xn - X::sel(term, re, rr);
                    J = J + X::sel(in, half * X::sel(term, wesd, wsd) * e * e, zero);
                    bad = bad | (in & !(X::vabs(xn) <= splat(q.big)));
                } else {
                    X::gst(io.xbar, ixb + kc * 7 + o_dxu, xn, in & wr);
                }
            }
            RQ_NOUNROLL
            for (int s = 0; 2 * s < N; ++s) {
                const I k = sp + 2 * s;
                const M in = (k < N) & e_in & e_even;
                const I kc = X::isel(k < N, k, X::isplat(N - 1));
                const V un = ld(kc * RQ_RS + RQ_HDR + jin, RQ_UA);
                if (pass == 0) {
                    const V ur = X::gld(io.yref, iyr + kc * 9 + 7 + jin);
                    const V e = un - ur;
                    V j = half * rjin * e * e;
                    j = j + X::sel(un < lbj, splat(q.rho_l) * (lbj - un), zero) + X::sel(un > ubj, splat(q.rho_u) * (un - ubj), zero);
                    J = J + X::sel(in, j, zero);
                    bad = bad | (in & !(X::vabs(un) <= splat(q.big)));
                } else {
                    X::gst(io.ubar, iub + kc * 2 + jin, un, in & wr);
                }
            }
            if (pass == 0) { failed = failed | X::row_or(bad); }
        }
        cost = X::row_sum(J);
    }
};

// Scalars of the kernel from the problem description (include/admpc.h).
// T = float: an interior point cannot be driven to the fp64 levels in 24-bit arithmetic -- with complementarity 1e-7 the barrier
// ratios lam / t reach 1e6..1e7 and the Riccati factorisation of H + diag(lam / t) loses every digit (measured with the lane
// emulator: 10-20 % of the instances run to iter_max with garbage steps).  The fp32 instantiation therefore stops at
// complementarity 1e-3, residual 1e-2, step 1e-3 and clamps slacks / multipliers at 1e-8: every instance of the config-5
// scenarios converges (<= 19 iterations at N = 80) to within 4e-4 (absolute, inputs of size 10) of the fp64 oracle.
// Tolerances of the config that are looser than these are kept.
template <class T, class Cfg>
RQ_FN void rq_make_params(const Cfg& c, RqParams<T>& q)
{
    const bool f32 = sizeof(T) == 4;
    q.N = c.N; q.itmax = c.ipm_iter_max; q.try_unc = c.ipm_try_unconstrained != 0.0 ? 1 : 0;
    q.h = (T)c.Ts;
    for (int i = 0; i < 7; ++i) { q.Qd[i] = (T)(c.Ts * c.W[i]); q.Qe[i] = (T)c.We[i]; }
    for (int j = 0; j < 2; ++j) { q.Rd[j] = (T)(c.Ts * c.W[7 + j]); q.lbu[j] = (T)c.lbu[j]; q.ubu[j] = (T)c.ubu[j]; }
    q.lbd = (T)c.lbx_delta; q.ubd = (T)c.ubx_delta;
    q.rho_l = (T)(c.Ts * c.zl); q.rho_u = (T)(c.Ts * c.zu);
    q.thr = (T)c.ipm_thr0; q.thw = (T)c.ipm_warm_thr; q.mu0 = (T)c.ipm_mu0;
    q.tol_comp = (T)(f32 && c.ipm_tol_comp < 1e-3 ? 1e-3 : c.ipm_tol_comp);
    q.tol_res = (T)(f32 && c.ipm_tol_res < 1e-2 ? 1e-2 : c.ipm_tol_res);
    // fp32 ALWAYS tests the step (its complementarity / residual levels are stop levels, not accuracy targets: the accuracy of the float
    // path comes from the step test) -- at the configured level when that is looser than 1e-3 but a real level (<= 1), else at 1e-3;
    // fp64: as configured (default: off, 1e30 -- HPIPM, whose BALANCE levels the defaults are, has none)
    q.tol_step = (T)(f32 ? ((c.ipm_tol_step < 1e-3 || c.ipm_tol_step > 1.0) ? 1e-3 : c.ipm_tol_step) : c.ipm_tol_step);
    q.inv_nineq = (T)(1.0 / (double)(8 * c.N + 2 * (c.N - 1)));
    q.big = f32 ? (T)1e30 : (T)1e300;
    q.floor_ = f32 ? (T)1e-8 : (T)1e-40;
    // fp64: MU_FLOOR * tol_comp (admpc.h).  fp32: none -- its tol_comp (1e-3) is a stop level, not an accuracy target (the accuracy of
    // the float path comes from the steps that overshoot it), and slacks / multipliers are clamped at 1e-8 already.  (A floor of 1e-8
    // was tried: one of 49 152 census instances then ends 2.6e-2 off the fp64 minimiser instead of 1.8e-3.)
    q.mu_floor = f32 ? (T)0 : (T)(ADMPC_IPM_MU_FLOOR * c.ipm_tol_comp);
    q.blocked = (T)ADMPC_IPM_BLOCKED_STEP;
    q.wrest = (T)c.ipm_warm_restart;
    q.fbit = (int)c.ipm_fallback_iter;
}
