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
//     built with pi = 0; where a stationarity residual is needed (stopping test, iterate snapshot) the exact adjoint of the
//     current point is swept backwards (sweep SA): residual = reduced gradient, as in the condensed N = 20 kernel.
//   * sl == t[2], su == t[3] (they start equal and receive identical steps), so they are not stored.
//   * per-inequality work is organised by SIDE: a lane owns one bound of one input together with its slack pair
//     (t_b, lam_b, t_s, lam_s), or one steering bound (t_b, lam_b); two stages per pass step (lanes 0-5 and 8-13).
//
// Lane roles inside a row (lane l = 0..15):
//   sweeps   l < 7: state component l (column l of P, K, A);  l = 7, 8: input 0, 1 (columns of B);  l >= 9 idle
//   passes   l = 8*sp + side: stage parity sp, side 0..3 = (u0 lower, u0 upper, u1 lower, u1 upper), 4, 5 = steering lower, upper
//
// Per-stage record in LDS (RQ_RS values of T; record r = k + 1, k = -1 .. N-1):
//   T[10] LAM[10]  slack / multiplier of bound pairs (side 0..5) and slack pairs (6 + side, side 0..3)      stage k
//   DU[2]          input step of stage k                     DX[7]   state step of stage k+1
//   K[7][2] LI[3]  feedback gains (column c: K0c, K1c), inverse of Huu of stage k
//   U[2]           gu -> kff -> ddu of stage k               A[2]    Rt (barrier-augmented R) -> predictor ddu
//   Q1             Qt of stage k -> predictor ddx6 of stage k        X   steering-barrier term of gx6 of stage k
//   D[7]           ddx of stage k+1; between pass E2 and sweep S4: D[0..1] = sigma-mu coefficient of gu, D[2] = of gx6
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
#define RQ_RS 60
enum { RQ_T = 0, RQ_LAM = 10, RQ_DU = 20, RQ_DX = 22, RQ_X = 29, RQ_K = 30, RQ_LI = 44, RQ_Q1 = 47, RQ_U = 48, RQ_A = 50, RQ_D = 52 };
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
};

template <class T>
struct RqArrays {              // wave-uniform array bases (device: kernel arguments, SGPR-addressed loads with 32-bit lane offsets)
    const T *x0, *yref, *yref_e, *GT, *bl;
    T *xbar, *ubar, *pi;
};

template <class X>
struct RowQp {
    typedef typename X::T T;
    typedef typename X::V V;
    typedef typename X::I I;
    typedef typename X::M M;
    typedef typename X::Lds Lds;

    const RqParams<T>& q;
    const RqArrays<T>& io;
    I ix0, iyr, iye, igt, ibl, ixb, iub;   // element offsets of the row's instance in the arrays
    Lds lds;
    const int N;

    // ---- lane constants -------------------------------------------------------------------------------------------
    I lane;
    M is_x, is_u, is6, is7, lt2;
    V E[9];                    // E[r] = (lane == r)
    V keepc, unitc[2];         // column-layout fix-up of lanes 0, 1 (unit columns of A)
    V g6c;                     // G[6][c] of the lane's column: 1 (c = 6), h (c = 8), else 0
    V lt2f;                    // 1 on lanes 0, 1
    V wq, wqe;                 // state weights of the lane (0 on lanes >= 7)
    V rjm;                     // Rd[j] on lanes 7, 8
    I o_gc, o_gr, o_x, o_y, o_u, o_k;
    I o1, o1sa, o2s1, o2s3, o2sa, o3sa;
    // passes
    I sp, side, jin, o_tb, o_ts, o_vl, o_st, o_dxu;
    M e_valid, e_isd, e_even, e_in, dxl;
    V sgn, bound, rho, rjin, isdf;
    V wsd, wesd, lbj, ubj;     // pass lanes (sp, i < 7): weights of state i; input lanes: bounds of input jin

    RQ_FN V splat(T x) { return X::splat(x); }

    // inst: index of the row's instance (row-uniform)
    RQ_FN RowQp(const RqParams<T>& q_, const RqArrays<T>& io_, Lds lds_, I inst) : q(q_), io(io_), lds(lds_), N(q_.N)
    {
        ix0 = inst * 7; iye = ix0; iyr = inst * (N * 9); igt = inst * (N * RQ_GTS); ibl = inst * (N * 7); ixb = inst * ((N + 1) * 7); iub = inst * (N * 2);
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
        o_k = X::isel(is_x, lane * 2, X::isel(is_u, ju, zi));                       // K: lanes < 7 their pair, lanes 7, 8 row j (stride 2)
        // sweep LDS offsets relative to k * RS
        o1   = X::isel(is_x, lane + RQ_DX, X::isel(is_u, ju + (RQ_RS + RQ_U), X::isplat(RQ_DX)));
        o1sa = X::isel(is_x, lane + RQ_DX, X::isel(is_u, ju + (RQ_RS + RQ_DU), X::isplat(RQ_DX)));
        o2s1 = X::isel(is_u, ju + (RQ_RS + RQ_A), X::isplat(RQ_RS + RQ_X));
        o2s3 = X::isel(is_u, ju + (RQ_RS + RQ_D), X::isplat(RQ_RS + RQ_X));
        o2sa = X::isel(is_u, ju * 2 + (RQ_RS + RQ_LAM), X::isplat(RQ_RS + RQ_LAM + 4));
        o3sa = o2sa + 1;
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
        isdf = X::sel(e_isd, one, zero);
        wsd = zero; wesd = zero;
        for (int i = 0; i < 7; ++i) { wsd = X::sel(side == i, splat(q.Qd[i]), wsd); wesd = X::sel(side == i, splat(q.Qe[i]), wesd); }
        lbj = X::sel(jin == 0, splat(q.lbu[0]), splat(q.lbu[1])); ubj = X::sel(jin == 0, splat(q.ubu[0]), splat(q.ubu[1]));
        const I sidec = X::isel(e_valid, side, zi);
        o_tb = sidec;                                                               // bound pair of the side
        o_ts = X::isel(e_in, side + 6, sidec);                                      // slack pair (inputs), own pair otherwise
        o_vl = X::isel(e_isd, X::isplat(RQ_DX + 6 - RQ_RS), jin + RQ_DU);           // steering: dx6 of stage k = rec(k-1).DX[6]
        o_st = X::isel(e_isd, X::isplat(RQ_Q1), jin + RQ_U);                        // step slot: ddx6 predictor / U
        o_dxu = X::isel(dxl, side, zi);
    }

    // ---- small helpers ----------------------------------------------------------------------------------------------
    RQ_FN V ld(I off, int imm) { return X::lds_ld(lds, off, imm); }
    RQ_FN void st(I off, int imm, V v, M m) { X::lds_st(lds, off, imm, v, m); }
    RQ_FN static V fma(V a, V b, V c) { return X::fma(a, b, c); }

    // column layout of stage k: Gc[l] = G[l][c] for the lane's column c (rows 0..5)
    RQ_FN void load_gc(int k, V Gc[6]) {
        X::gld6(io.GT, igt + o_gc + k * RQ_GTS, Gc);
    }
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
    // =================================================================================================================
    // sweeps
    // =================================================================================================================

    // gradient of the GN model at the point (lanes < 7: state k), stage k: w (dx_k + xbar_k - ref_k)
    RQ_FN V gx_plain(V dxk, V xk, V rk, bool terminal) { return (terminal ? wqe : wq) * (dxk + xk - rk); }

    // S0 / SF: states rolled out through the linearised dynamics, dx_{k+1} = A dx_k + B du_k + b_k (S0: du = 0)
    RQ_FN void sweep_rollout(bool with_du) {
        V z = ld(o_x + RQ_DX, 0);                                  // dx_0 = rec(-1).DX
        V Gr[7], Gn[7];
        load_gr(0, Gr);
        V bk = X::gld(io.bl, ibl + o_x);
        RQ_NOUNROLL
        for (int k = 0; k < N; ++k) {
            const int kn = k + 1 < N ? k + 1 : k;
            load_gr(kn, Gn);
            const V bn = X::gld(io.bl, ibl + o_x + kn * 7);
            V zz = X::sel(is_x, z, splat((T)0));
            if (with_du) { const V du = ld(o_u + RQ_DU, (k + 1) * RQ_RS); zz = X::sel(is_x, z, du); }
            V zn = splat((T)0);
            X::template dotbc<7, 2>(Gr, zz, zn);
            zn = fma(zz, lt2f, zn);
            const V u1 = X::template bc<8>(zz);
            zn = X::sel(is6, fma(u1, splat(q.h), zz), zn) + bk;
            st(o_x + RQ_DX, (k + 1) * RQ_RS, zn, is_x);
            z = zn;
            RQ_UNROLL
            for (int c = 0; c < 7; ++c) Gr[c] = Gn[c];
            bk = bn;
        }
    }

    // S1: backward Riccati sweep for the matrices, fused with the backward sweep of the gradient (gx from the point, X; gu = U).
    //     Leaves K, LI, kff (in U).
    RQ_FN void sweep_factor() {
        V P[7], p;
        {
            const V dxN = ld(o_x + RQ_DX, N * RQ_RS);             // dx_N = rec(N-1).DX
            const V xN = X::gld(io.xbar, ixb + o_x + N * 7), rN = X::gld(io.yref_e, iye + o_x);
            p = X::sel(is_x, wqe * (dxN + xN - rN), splat((T)0));
            RQ_UNROLL
            for (int i = 0; i < 7; ++i) P[i] = E[i] * wqe;
        }
        V Gc[6], Gn[6];
        load_gc(N - 1, Gc);
        V xk = X::gld(io.xbar, ixb + o_x + (N - 1) * 7), rk = X::gld(io.yref, iyr + o_y + (N - 1) * 9);
        RQ_NOUNROLL
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            load_gc(kn, Gn);
            const V xn = X::gld(io.xbar, ixb + o_x + kn * 7), rn = X::gld(io.yref, iyr + o_y + kn * 9);
            const V v1 = ld(o1, k * RQ_RS);                        // lanes < 7: dx_k ; lanes 7, 8: gu
            const V v2 = ld(o2s1, k * RQ_RS);                      // lane 6: X ; lanes 7, 8: Rt
            const V v3 = ld(X::isplat(0), (k + 1) * RQ_RS + RQ_Q1);  // Qt
            fix_gc(Gc);
            const V g = X::sel(is_x, fma(wq, v1 + xk - rk, X::sel(is6, v2, splat((T)0))), v1);
            const V wd = X::sel(is_x, X::sel(is6, v3, wq), v2);
            V G7[7]; for (int l = 0; l < 6; ++l) G7[l] = Gc[l]; G7[6] = g6c;
            V Mm[7];
            X::pg(P, G7, Mm);
            V H[9];
            X::gtm(Gc, Mm, q.h, H);
            RQ_UNROLL
            for (int r = 0; r < 9; ++r) H[r] = fma(E[r], wd, H[r]);
            const V h00 = X::template bc<7>(H[7]), h01 = X::template bc<7>(H[8]), h11 = X::template bc<8>(H[8]);
            const V idet = X::rcp(h00 * h11 - h01 * h01);
            const V i00 = h11 * idet, i01 = -(h01 * idet), i11 = h00 * idet;
            const V K0 = X::sel(is_x, -(i00 * H[7] + i01 * H[8]), splat((T)0));
            const V K1 = X::sel(is_x, -(i01 * H[7] + i11 * H[8]), splat((T)0));
            X::schur(H, K0, K1);                                   // H[i] += bc7(H[i]) K0 + bc8(H[i]) K1, i < 7
            V hv = g;
            X::template dotbc<7, 0>(G7, p, hv);
            const V hu0 = X::template bc<7>(hv), hu1 = X::template bc<8>(hv);
            const V pn = fma(K1, hu1, fma(K0, hu0, hv));
            const V kff = X::sel(is7, -(i00 * hu0 + i01 * hu1), -(i01 * hu0 + i11 * hu1));
            X::lds_st2(lds, o_k, (k + 1) * RQ_RS + RQ_K, K0, K1, is_x);
            st(X::isplat(0), (k + 1) * RQ_RS + RQ_LI, i00, is7);
            st(X::isplat(1), (k + 1) * RQ_RS + RQ_LI, i01, is7);
            st(X::isplat(2), (k + 1) * RQ_RS + RQ_LI, i11, lane == 8);
            st(o_u + RQ_U, (k + 1) * RQ_RS, kff, is_u);
            RQ_UNROLL
            for (int i = 0; i < 7; ++i) P[i] = H[i];
            p = X::sel(is_x, pn, splat((T)0));
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) Gc[l] = Gn[l];
            xk = xn; rk = rn;
        }
    }

    // S3: backward sweep of the gradient alone (corrector right-hand side: gu = U - smu * D[j], gx6 term = X - smu * D[2])
    RQ_FN void sweep_backward(V smu) {
        V p;
        {
            const V dxN = ld(o_x + RQ_DX, N * RQ_RS);
            const V xN = X::gld(io.xbar, ixb + o_x + N * 7), rN = X::gld(io.yref_e, iye + o_x);
            p = X::sel(is_x, wqe * (dxN + xN - rN), splat((T)0));
        }
        V Gc[6], Gn[6];
        load_gc(N - 1, Gc);
        V xk = X::gld(io.xbar, ixb + o_x + (N - 1) * 7), rk = X::gld(io.yref, iyr + o_y + (N - 1) * 9);
        RQ_NOUNROLL
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            load_gc(kn, Gn);
            const V xn = X::gld(io.xbar, ixb + o_x + kn * 7), rn = X::gld(io.yref, iyr + o_y + kn * 9);
            const V v1 = ld(o1, k * RQ_RS);                        // dx_k | gA
            const V v2 = ld(o2s3, k * RQ_RS);                      // lane 6: xA ; lanes 7, 8: gB
            const V v3 = ld(X::isplat(0), (k + 1) * RQ_RS + RQ_D + 2);   // xB
            V K0, K1;
            X::lds_ld2(lds, o_k, (k + 1) * RQ_RS + RQ_K, K0, K1);
            K0 = X::sel(is_x, K0, splat((T)0)); K1 = X::sel(is_x, K1, splat((T)0));
            const V i00 = ld(X::isplat(0), (k + 1) * RQ_RS + RQ_LI), i01 = ld(X::isplat(1), (k + 1) * RQ_RS + RQ_LI),
                    i11 = ld(X::isplat(2), (k + 1) * RQ_RS + RQ_LI);
            fix_gc(Gc);
            const V g = X::sel(is_x, fma(wq, v1 + xk - rk, X::sel(is6, v2 - smu * v3, splat((T)0))), v1 - smu * v2);
            V G7[7]; for (int l = 0; l < 6; ++l) G7[l] = Gc[l]; G7[6] = g6c;
            V hv = g;
            X::template dotbc<7, 0>(G7, p, hv);
            const V hu0 = X::template bc<7>(hv), hu1 = X::template bc<8>(hv);
            const V pn = fma(K1, hu1, fma(K0, hu0, hv));
            const V kff = X::sel(is7, -(i00 * hu0 + i01 * hu1), -(i01 * hu0 + i11 * hu1));
            st(o_u + RQ_U, (k + 1) * RQ_RS, kff, is_u);
            p = X::sel(is_x, pn, splat((T)0));
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) Gc[l] = Gn[l];
            xk = xn; rk = rn;
        }
    }

    // SA: exact adjoint of the current point -> max-norm of the reduced gradient (the stationarity residual of the stopping
    //     test).  Raw gradients: gu = R (u - uref) - lam_l + lam_u, gx = w (x - ref), steering multipliers on component 6.
    //     With want_pi the adjoint (= dynamics multipliers pi_k of the iterate snapshot) is written there, [N][7].
    RQ_FN V sweep_adjoint(bool want_pi, M pim) {
        V lam;
        {
            const V dxN = ld(o_x + RQ_DX, N * RQ_RS);
            const V xN = X::gld(io.xbar, ixb + o_x + N * 7), rN = X::gld(io.yref_e, iye + o_x);
            lam = X::sel(is_x, wqe * (dxN + xN - rN), splat((T)0));
        }
        V rg = splat((T)0);
        V Gc[6], Gn[6];
        load_gc(N - 1, Gc);
        V xk = X::gld(io.xbar, ixb + o_x + (N - 1) * 7), rk = X::gld(io.yref, iyr + o_y + (N - 1) * 9), uk = X::gld(io.ubar, iub + o_u + (N - 1) * 2);
        RQ_NOUNROLL
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            load_gc(kn, Gn);
            const V xn = X::gld(io.xbar, ixb + o_x + kn * 7), rn = X::gld(io.yref, iyr + o_y + kn * 9), un = X::gld(io.ubar, iub + o_u + kn * 2);
            if (want_pi) X::gst(io.pi, ibl + o_x + k * 7, lam, is_x & pim);   // pi_k multiplies dx_{k+1} = A dx_k + B du_k + b_k
            const V v1 = ld(o1sa, k * RQ_RS);                      // dx_k | du_k
            const V l0 = ld(o2sa, k * RQ_RS), l1 = ld(o3sa, k * RQ_RS);   // lane 6: steering pair; lanes 7, 8: bound pair of input j
            fix_gc(Gc);
            const V g = X::sel(is_x, fma(wq, v1 + xk - rk, X::sel(is6, l1 - l0, splat((T)0))), fma(rjm, v1 + uk - rk, l1 - l0));
            V G7[7]; for (int l = 0; l < 6; ++l) G7[l] = Gc[l]; G7[6] = g6c;
            V lv = g;
            X::template dotbc<7, 0>(G7, lam, lv);
            rg = X::vmaxnan(rg, X::sel(is_u, X::vabs(lv), splat((T)0)));
            lam = X::sel(is_x, lv, splat((T)0));
            RQ_UNROLL
            for (int l = 0; l < 6; ++l) Gc[l] = Gn[l];
            xk = xn; rk = rn; uk = un;
        }
        return X::row_maxnan(rg);
    }

    // S2 / S4: forward roll-out of the Newton step: ddu_k = K_k ddx_k + kff_k (into U), ddx_{k+1} = A ddx_k + B ddu_k.
    //     full = false (predictor): only ddx6_{k+1} is kept (Q1);  full = true: ddx_{k+1} into D.
    RQ_FN void sweep_forward(bool full) {
        V z = splat((T)0);
        V Gr[7], Gn[7];
        load_gr(0, Gr);
        RQ_NOUNROLL
        for (int k = 0; k < N; ++k) {
            const int kn = k + 1 < N ? k + 1 : k;
            load_gr(kn, Gn);
            V Kr[7];
            RQ_UNROLL
            for (int c = 0; c < 7; ++c) Kr[c] = ld(o_u, (k + 1) * RQ_RS + RQ_K + 2 * c);     // lanes 7, 8: row j of K
            const V kff = ld(o_u + RQ_U, (k + 1) * RQ_RS);
            V ddu = kff;
            X::template dotbc<7, 0>(Kr, z, ddu);
            const V zz = X::sel(is_x, z, ddu);
            V zn = splat((T)0);
            X::template dotbc<7, 2>(Gr, zz, zn);
            zn = fma(zz, lt2f, zn);
            const V u1 = X::template bc<8>(zz);
            zn = X::sel(is6, fma(u1, splat(q.h), zz), zn);
            st(o_u + RQ_U, (k + 1) * RQ_RS, ddu, is_u);
            if (full) st(o_x + RQ_D, (k + 1) * RQ_RS, zn, is_x);
            else if (k + 1 < N) st(X::isplat(0), (k + 2) * RQ_RS + RQ_Q1, zn, is6);     // predictor ddx6 of stage k+1, in ITS record
            z = X::sel(is_x, zn, splat((T)0));
            RQ_UNROLL
            for (int c = 0; c < 7; ++c) Gr[c] = Gn[c];
        }
    }

    // =================================================================================================================
    // passes over the inequalities (lane = side, two stages per step)
    // =================================================================================================================
    struct Side {                 // what a side lane holds for its stage
        V tb, lb, ts, ls;         // slack / multiplier of the bound pair and of the slack pair (inputs)
        V vabs, uref;             // ubar + du (inputs) or xbar6 + dx6 (steering); uref of the input
        M act;                    // this side exists (stage < N, steering only on stages >= 1)
        M stv;                    // the even lane of the pair writes the stage outputs (stage < N)
        M inb;                    // valid side lane of a stage < N (whether or not the side exists)
        int koff;                 // (k + 1) * RS of the EVEN stage of the step; lanes add sp * RS through kl
        I kl;                     // per-lane record offset (clamped stage)
    };

    RQ_FN void side_load(int s, Side& S, bool with_state) {
        const I k = sp + 2 * s;
        const M in = k < N;
        const I kc = X::isel(in, k, X::isplat(N - 1));
        S.kl = (kc + 1) * RQ_RS;
        S.act = e_valid & in & ((!e_isd) | (k >= 1));
        S.stv = e_valid & in & e_even;
        S.inb = e_valid & in;
        const V ub = X::gld(io.ubar, iub + kc * 2 + jin), xb = X::gld(io.xbar, ixb + kc * 7 + 6);
        S.uref = X::gld(io.yref, iyr + kc * 9 + 7 + jin);
        const V vl = ld(S.kl + o_vl, 0);
        S.vabs = X::sel(e_isd, xb, ub) + vl;
        if (with_state) {
            S.tb = ld(S.kl + o_tb, RQ_T); S.lb = ld(S.kl + o_tb, RQ_LAM);
            S.ts = ld(S.kl + o_ts, RQ_T); S.ls = ld(S.kl + o_ts, RQ_LAM);
        }
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
        const V o1v = X::sel(e_isd, gs, fma(rjin, S.vabs - S.uref, gs));
        const V o2v = X::sel(e_isd, splat(q.Qd[6]), rjin) + Rs;
        st(S.kl + X::isel(e_isd, X::isplat(RQ_X), jin + RQ_U), 0, o1v, S.stv);
        st(S.kl + X::isel(e_isd, X::isplat(RQ_Q1), jin + RQ_A), 0, o2v, S.stv);
    }

    RQ_FN void pass_e1(Red& R) {
        R.mu = splat((T)0); R.cmax = splat((T)0); R.rmax = splat((T)0);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) { Side S; side_load(s, S, true); e1_core(S, R); }
        R.mu = X::row_sum(R.mu); R.cmax = X::row_max(R.cmax); R.rmax = X::row_maxnan(R.rmax);
    }

    // predictor step of a side from uua = sgn * (ddu or ddx6): dt, dlam of both pairs
    struct Step { V dtb, dts, dlb, dls; };
    RQ_FN void side_step_pred(const Side& S, const Bar& B, V uua, Step& P) {
        const V ds = -((B.e + B.Gb * uua) * B.is);
        P.dtb = uua + ds + B.rd; P.dts = ds;
        P.dlb = -S.lb - B.Gb * P.dtb; P.dls = -S.ls - B.Gs * P.dts;
    }

    // E2: after the predictor solve.  Ratio test + sum dt*dlam (per-lane partials), moves the predictor step out of U and
    //     leaves the corrector right-hand side split as  gA - smu * gB  (U / D[0..1], X / D[2]).
    RQ_FN void pass_e2(V& rr, V& s2) {
        const V zero = splat((T)0), one = splat((T)1);
        rr = zero; s2 = zero;
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, true);
            Bar B; side_barrier(S, B);
            const V stp = ld(S.kl + o_st, 0);                      // ddu_a (U) or ddx6_a (Q1)
            const V uua = sgn * stp;
            Step P; side_step_pred(S, B, uua, P);
            const V xb = P.dtb * B.itb, xs = P.dts * B.its;
            V r = X::vmax(-xb, one + xb);
            r = X::vmax(r, X::sel(e_isd, zero, X::vmax(-xs, one + xs)));
            rr = X::vmax(rr, X::sel(S.act, r, zero));
            const V mb = P.dtb * P.dlb, ms = X::sel(e_isd, zero, P.dts * P.dls);
            s2 = s2 + X::sel(S.act, mb + ms, zero);
            const V mbi = mb * B.itb, msi = ms * B.its;
            const V P1 = mbi - B.Gb * B.is * (mbi + msi);
            const V P2 = B.itb - B.Gb * B.is * (B.itb + B.its);
            const V pa = X::sel(S.act, sgn * (B.psi + P1), zero), pb = X::sel(S.act, sgn * P2, zero);
            const V sa = pa + X::swap1(pa), sb = pb + X::swap1(pb);
            st(S.kl + jin + RQ_A, 0, stp, S.stv & e_in);                                              // predictor ddu
            st(S.kl + X::isel(e_isd, X::isplat(RQ_X), jin + RQ_U), 0,
               X::sel(e_isd, sa, fma(rjin, S.vabs - S.uref, sa)), S.stv);
            st(S.kl + X::isel(e_isd, X::isplat(RQ_D + 2), jin + RQ_D), 0, sb, S.stv);
        }
        rr = X::row_max(rr); s2 = X::row_sum(s2);
    }

    // corrector step of a side (predictor recomputed from its stored inputs)
    RQ_FN void side_step_corr(const Side& S, const Bar& B, V uua, V uu, V smu, Step& C) {
        const V zero = splat((T)0);
        Step P; side_step_pred(S, B, uua, P);
        const V mb = P.dtb * P.dlb, ms = P.dts * P.dls;
        const V cb = (mb - smu) * B.itb, cs = X::sel(e_isd, zero, (ms - smu) * B.its);
        const V ec = B.e + cb + cs;
        const V ds = -((ec + B.Gb * uu) * B.is);
        C.dtb = uu + ds + B.rd; C.dts = ds;
        C.dlb = -(S.lb + cb) - B.Gb * C.dtb; C.dls = -(S.ls + cs) - B.Gs * C.dts;
    }
    RQ_FN void side_steps_load(const Side& S, V& uua, V& uu) {
        const V sa = ld(S.kl + X::isel(e_isd, X::isplat(RQ_Q1), jin + RQ_A), 0);               // predictor ddu / ddx6
        const V sc = ld(S.kl + X::isel(e_isd, X::isplat(RQ_D + 6 - RQ_RS), jin + RQ_U), 0);    // corrector ddu / ddx6
        uua = sgn * sa; uu = sgn * sc;
    }

    // E3a: ratio test of the corrector step
    RQ_FN V pass_e3a(V smu) {
        const V zero = splat((T)0);
        V rr = zero;
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, true);
            Bar B; side_barrier(S, B);
            V uua, uu; side_steps_load(S, uua, uu);
            Step C; side_step_corr(S, B, uua, uu, smu, C);
            V r = X::vmax(-(C.dtb * B.itb), -(C.dlb * X::rcp(S.lb)));
            r = X::vmax(r, X::sel(e_isd, zero, X::vmax(-(C.dts * B.its), -(C.dls * X::rcp(S.ls)))));
            rr = X::vmax(rr, X::sel(S.act, r, zero));
        }
        return X::row_max(rr);
    }

    // dx += alpha ddx for every stage (lanes (sp, i < 7) <-> record 2s + sp)
    RQ_FN void pass_dx_update(V alpha, M rowact) {
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            const I k = sp + 2 * s;
            const M in = (k < N) & dxl;
            const I kl = (X::isel(k < N, k, X::isplat(N - 1)) + 1) * RQ_RS + o_dxu;
            const V dx = ld(kl, RQ_DX), dd = ld(kl, RQ_D);
            st(kl, RQ_DX, fma(alpha, dd, dx), in & rowact & (alpha > splat((T)0)));
        }
    }

    // E3b + E1: apply the corrector step with step length alpha (0 on rows that are not iterating), then the quantities of
    //     the next iteration from the updated state, still in registers.  Returns the per-lane max |alpha ddu| partial.
    RQ_FN V pass_e3b_e1(V smu, V alpha, M rowact, Red& R) {
        const V zero = splat((T)0);
        V stp = zero;
        R.mu = zero; R.cmax = zero; R.rmax = zero;
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, true);                         // old state, old dx6 (dx is stepped after this loop)
            Bar B; side_barrier(S, B);
            V uua, uu; side_steps_load(S, uua, uu);
            Step C; side_step_corr(S, B, uua, uu, smu, C);
            const V fl = splat(q.floor_);
            // rows that are not iterating keep their state bit for bit (their recomputed steps are never applied)
            S.tb = X::sel(rowact, X::vmax(fma(alpha, C.dtb, S.tb), fl), S.tb); S.lb = X::sel(rowact, X::vmax(fma(alpha, C.dlb, S.lb), fl), S.lb);
            S.ts = X::sel(rowact, X::vmax(fma(alpha, C.dts, S.ts), fl), S.ts); S.ls = X::sel(rowact, X::vmax(fma(alpha, C.dls, S.ls), fl), S.ls);
            const V au = X::sel(rowact, alpha * (sgn * uu), zero);                               // alpha * ddu_j (inputs), alpha * ddx6 (steering)
            stp = X::vmax(stp, X::sel(S.act & e_in, X::vabs(au), zero));
            S.vabs = S.vabs + au;
            const V dun = ld(S.kl + jin + RQ_DU, 0) + au;
            st(S.kl + jin + RQ_DU, 0, dun, S.stv & e_in);
            st(S.kl + o_tb, RQ_T, S.tb, S.act); st(S.kl + o_tb, RQ_LAM, S.lb, S.act);
            st(S.kl + o_ts, RQ_T, S.ts, S.act & e_in); st(S.kl + o_ts, RQ_LAM, S.ls, S.act & e_in);
            e1_core(S, R);
        }
        pass_dx_update(alpha, rowact);
        R.mu = X::row_sum(R.mu); R.cmax = X::row_max(R.cmax); R.rmax = X::row_maxnan(R.rmax);
        return X::row_max(stp);
    }

    // trial set-up: the QP without its inequalities from the start point (du = 0, rolled-out dx): U = R (ubar - uref), A = R,
    // X = 0, Q1 = Qd6; also zeroes du
    RQ_FN void pass_trial_setup() {
        const V zero = splat((T)0);
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            const I k = sp + 2 * s;
            const M in = k < N;
            const I kc = X::isel(in, k, X::isplat(N - 1));
            const I kl = (kc + 1) * RQ_RS;
            const M stv = e_valid & in & e_even;
            const V ub = X::gld(io.ubar, iub + kc * 2 + jin), ur = X::gld(io.yref, iyr + kc * 9 + 7 + jin);
            st(kl + X::isel(e_isd, X::isplat(RQ_X), jin + RQ_U), 0, X::sel(e_isd, zero, rjin * (ub - ur)), stv);
            st(kl + X::isel(e_isd, X::isplat(RQ_Q1), jin + RQ_A), 0, X::sel(e_isd, splat(q.Qd[6]), rjin), stv);
            st(kl + jin + RQ_DU, 0, zero, stv & e_in);
        }
    }

    // does the step in U / D respect every bound?  (trial: du = 0 before it)
    RQ_FN M pass_trial_check() {
        M ok = X::mtrue();
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, false);
            const V stp = ld(S.kl + X::isel(e_isd, X::isplat(RQ_D + 6 - RQ_RS), jin + RQ_U), 0);
            const V qv = sgn * (S.vabs + stp - bound);
            ok = ok & ((!S.act) | (qv >= splat((T)0)));
        }
        return X::row_and(ok);
    }

    // start of the interior point: du += a0 * ddu, dx += a0 * ddx (a0 = 1: from the trial's minimiser, 0: zero step), then
    // slacks and multipliers (th = clip level; warm: a violated input bound is absorbed by its slack).
    // Returns the max-norm of the stationarity residual of this start point, as the oracle's ipm_residuals() sees it: with the
    // zero step its dynamics multipliers are zero (ru = r - lam_l + lam_u, rx = Q dx + q + steering multipliers); from the
    // trial's minimiser they are that minimiser's exact multipliers, which leaves the inequality multipliers alone.
    RQ_FN V pass_init(V a0, V th, M warm) {
        const V zero = splat((T)0);
        const V gsc = X::sel(a0 > zero, zero, splat((T)1));        // weight of the plain gradient in the residual
        pass_dx_update(a0, X::mtrue());
        V rs0 = zero;
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            Side S; side_load(s, S, false);
            const V stp = ld(S.kl + X::isel(e_isd, X::isplat(RQ_D + 6 - RQ_RS), jin + RQ_U), 0);
            const V au = X::sel(a0 > zero, stp, zero);
            const V vabs = S.vabs + X::sel(e_isd, zero, au);       // steering: dx6 already updated
            st(S.kl + jin + RQ_DU, 0, ld(S.kl + jin + RQ_DU, 0) + au, S.stv & e_in);
            const V qv = sgn * (vabs - bound);
            const V ts = X::sel(warm, X::vmax(-qv, zero), zero) + th;
            const V tb = X::vmax(qv + X::sel(e_isd, zero, ts), th);
            const V mu0 = splat(q.mu0);
            const V lb = X::sel(S.act, mu0 * X::rcp(tb), zero);
            st(S.kl + o_tb, RQ_T, X::sel(S.act, tb, splat((T)1)), S.inb);
            st(S.kl + o_tb, RQ_LAM, lb, S.inb);
            st(S.kl + o_ts, RQ_T, X::sel(S.act, ts, splat((T)1)), S.inb & e_in);
            st(S.kl + o_ts, RQ_LAM, X::sel(S.act, mu0 * X::rcp(ts), zero), S.inb & e_in);
            // stationarity row of the pair: plain gradient (zero step only) - lam_lower + lam_upper
            const V ml = -(sgn * lb);
            const V ref6 = X::gld(io.yref, iyr + X::isel(S.inb, sp + 2 * s, X::isplat(N - 1)) * 9 + 6);
            const V gpl = X::sel(e_isd, splat(q.Qd[6]) * (vabs - ref6), rjin * (vabs - S.uref));
            const V row = fma(gsc, gpl, ml + X::swap1(ml));
            rs0 = X::vmaxnan(rs0, X::sel(S.stv & (e_in | S.act), X::vabs(row), zero));
        }
        // the other state rows (zero step only): w (dx + xbar - ref), k = 1..N; component 6 of stages 1..N-1 is covered above
        RQ_NOUNROLL
        for (int s = 0; 2 * s < N; ++s) {
            const I k1 = sp + (2 * s + 1);                         // stage index of the state row, 1..N
            const M in = (k1 <= N) & dxl;
            const I kc = X::isel(k1 <= N, k1, X::isplat(N));
            const M term = kc == N;
            const V xb = X::gld(io.xbar, ixb + kc * 7 + o_dxu), dx = ld(kc * RQ_RS + o_dxu, RQ_DX);
            const V rr = X::gld(io.yref, iyr + X::isel(term, X::isplat(0), kc) * 9 + o_dxu), re = X::gld(io.yref_e, iye + o_dxu);
            const V gx = X::sel(term, wesd, wsd) * (xb + dx - X::sel(term, re, rr));
            rs0 = X::vmaxnan(rs0, X::sel(in & (term | !(side == 6)), X::vabs(gsc * gx), zero));
        }
        return X::row_maxnan(rs0);
    }

    // =================================================================================================================
    // one instance: returns status / iterations / cost through the references (row-uniform)
    // =================================================================================================================
    struct Result { M failed; I iters; V cost; V rmax; };

    // valid: the row carries an instance to solve.  want_pi (wave-uniform): also write the dynamics multipliers of the returned
    // iterate, [B][N][7], to io.pi on the rows of pim.
    RQ_FN void solve(M valid, Result& res, bool want_pi, M pim) {
        const V zero = splat((T)0), one = splat((T)1);
        // dx_0 = x0 - xbar_0 into rec(-1).DX ; rec(-1).D = 0
        st(o_x + RQ_DX, 0, X::gld(io.x0, ix0 + o_x) - X::gld(io.xbar, ixb + o_x), is_x);
        st(o_x + RQ_D, 0, zero, is_x);
        X::lds_fence();
        sweep_rollout(false);
        X::lds_fence();
        M active = valid, failed = X::mfalse();
        I iters = X::isplat(0);
        V rstat;                                                   // stationarity residual of the interior point's iterate (tracked)
        if (q.try_unc) {
            pass_trial_setup();
            X::lds_fence();
            sweep_factor();
            X::lds_fence();
            sweep_forward(true);
            X::lds_fence();
            const M ok = pass_trial_check();
            const M warm = (!ok) & X::mfrom(q.thw > (T)0);
            const V a0 = X::sel(ok | warm, one, zero);
            rstat = pass_init(a0, X::sel(warm, splat(q.thw), splat(q.thr)), warm);
            active = active & !ok;
        } else {
            pass_trial_setup();                                    // zeroes du
            X::lds_fence();
            st(o_x + RQ_D, 0, zero, is_x);
            for (int k = 0; k < N; ++k) { st(o_x + RQ_D, (k + 1) * RQ_RS, zero, is_x); st(o_u + RQ_U, (k + 1) * RQ_RS, zero, is_u); }
            X::lds_fence();
            rstat = pass_init(zero, splat(q.thr), X::mfalse());
        }
        X::lds_fence();
        Red R;
        pass_e1(R);
        X::lds_fence();
        V step = splat((T)1e30), rmax_prev = zero, rmax_last = zero;
        RQ_NOUNROLL
        for (int guard = 0; guard <= q.itmax; ++guard) {
            // every linear residual of a Newton iteration in residual form shrinks by (1 - alpha) per step; the inequality rows are
            // re-evaluated (R.rmax), the stationarity rows (which would need the dynamics multipliers) are tracked: rstat
            const V mu = R.mu * splat(q.inv_nineq);
            const V rmax = X::vmaxnan(R.rmax, rstat);
            const M nan = (!(mu == mu)) | (!(rmax == rmax));
            failed = failed | (active & nan);
            active = active & !nan;
            const M conv = active & (R.cmax <= splat(q.tol_comp)) & (step <= splat(q.tol_step)) &
                           ((rmax <= splat(q.tol_res)) | (X::mfrom(guard > 0) & (rmax > splat((T)0.1) * rmax_prev)));
            rmax_last = X::sel(active, rmax, rmax_last);
            rmax_prev = rmax;
            active = active & !conv & (iters < q.itmax);
            if (!X::any(active)) break;
            // ---- predictor
            sweep_factor();
            X::lds_fence();
            sweep_forward(false);
            X::lds_fence();
            V rr, s2;
            pass_e2(rr, s2);
            X::lds_fence();
            const V a_aff = X::sel(rr > one, X::rcp(rr), one);
            const V munq = R.mu;                                    // = mu * nineq
            const V mu_aff = ((one - a_aff) * munq + a_aff * a_aff * s2) * splat(q.inv_nineq);
            V sigma = mu_aff * X::rcp(mu); sigma = sigma * sigma * sigma;
            const V smu = sigma * mu;
            RQ_DBG("[emu] it mu=%.6e cmax=%.3e rmax=%.3e a_aff=%.6e mu_aff=%.6e sigma=%.6e\n", X::first(mu), X::first(R.cmax), X::first(rmax), X::first(a_aff), X::first(mu_aff), X::first(sigma));
            // ---- corrector
            sweep_backward(smu);
            X::lds_fence();
            sweep_forward(true);
            X::lds_fence();
            const V rc = pass_e3a(smu);
            const V amax = X::sel(rc > one, X::rcp(rc), one);
            V tau = one - mu_aff; tau = X::vmax(tau, splat((T)0.995)); tau = X::vmin(tau, splat((T)0.999999));
            const V alpha = X::sel(active, X::vmin(tau * amax, one), zero);
            RQ_DBG("[emu]      alpha=%.6e\n", X::first(alpha));
            const V stn = pass_e3b_e1(smu, alpha, active, R);
            X::lds_fence();
            RQ_DBG("[emu]      step=%.6e\n", X::first(stn));
            step = X::sel(active, stn, step);
            rstat = X::sel(active, (one - alpha) * rstat, rstat);
            iters = iters + X::isel(active, X::isplat(1), X::isplat(0));
        }
        res.failed = failed; res.iters = iters; res.rmax = rmax_last;
        // ---- H6: expand the states from the input step through the linearised dynamics (as acados' expand step)
        sweep_rollout(true);
        X::lds_fence();
        if (want_pi) (void)sweep_adjoint(true, pim);
    }

    // full step, cost, outputs.  Call after solve(); `write`: rows whose iterate may be overwritten when the step is finite.
    RQ_FN void finish(M write, M& failed, V& cost) {
        const V zero = splat((T)0), half = splat((T)0.5);
        M bad = X::mfalse();
        V J = zero;
        // states: lanes (sp, i < 7) <-> x_{k}, k = 2s + sp over 0..N (record k - 1)
        RQ_NOUNROLL
        for (int pass = 0; pass < 2; ++pass) {
            const M wr = write & !failed;
            if (pass == 1 && !X::any(wr)) break;
            RQ_NOUNROLL
            for (int s = 0; 2 * s < N + 1; ++s) {
                const I k = sp + 2 * s;
                const M in = (k <= N) & dxl;
                const I kc = X::isel(k <= N, k, X::isplat(N));
                const V xb = X::gld(io.xbar, ixb + kc * 7 + o_dxu), dx = ld(kc * RQ_RS + o_dxu, RQ_DX);
                const V xn = xb + dx;
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
                const V ub = X::gld(io.ubar, iub + kc * 2 + jin), du = ld((kc + 1) * RQ_RS + jin, RQ_DU);
                const V un = ub + du;
                if (pass == 0) {
                    const V ur = X::gld(io.yref, iyr + kc * 9 + 7 + jin);
                    const V e = un - ur;
                    const V lb = lbj, ubd = ubj;
                    V j = half * rjin * e * e;
                    j = j + X::sel(un < lb, splat(q.rho_l) * (lb - un), zero) + X::sel(un > ubd, splat(q.rho_u) * (un - ubd), zero);
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

// Scalars of the kernel from the problem description (include/admpc.h).  T = float: the interior point cannot resolve the
// fp64 tolerances; they are clipped to what fp32 arithmetic reaches (see DESIGN.md, fp32 path).
template <class T, class Cfg>
RQ_FN void rq_make_params(const Cfg& c, RqParams<T>& q)
{
    const bool f32 = sizeof(T) == 4;
    q.N = c.N; q.itmax = c.ipm_iter_max; q.try_unc = c.ipm_try_unconstrained != 0.0 ? 1 : 0;
    q.h = (T)c.Ts;
    RQ_UNROLL
    for (int i = 0; i < 7; ++i) { q.Qd[i] = (T)(c.Ts * c.W[i]); q.Qe[i] = (T)c.We[i]; }
    for (int j = 0; j < 2; ++j) { q.Rd[j] = (T)(c.Ts * c.W[7 + j]); q.lbu[j] = (T)c.lbu[j]; q.ubu[j] = (T)c.ubu[j]; }
    q.lbd = (T)c.lbx_delta; q.ubd = (T)c.ubx_delta;
    q.rho_l = (T)(c.Ts * c.zl); q.rho_u = (T)(c.Ts * c.zu);
    q.thr = (T)c.ipm_thr0; q.thw = (T)c.ipm_warm_thr; q.mu0 = (T)c.ipm_mu0;
    q.tol_comp = (T)(f32 && c.ipm_tol_comp < 1e-7 ? 1e-7 : c.ipm_tol_comp);
    q.tol_res = (T)(f32 && c.ipm_tol_res < 1e-4 ? 1e-4 : c.ipm_tol_res);
    q.tol_step = (T)(f32 && c.ipm_tol_step < 1e-4 ? 1e-4 : c.ipm_tol_step);
    q.inv_nineq = (T)(1.0 / (double)(8 * c.N + 2 * (c.N - 1)));
    q.big = f32 ? (T)1e30 : (T)1e300;
    q.floor_ = f32 ? (T)1e-30 : (T)1e-40;
}
