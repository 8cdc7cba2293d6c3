// seg_cut.h -- the coupling of TWO condensed segments through the D states at their cut (D = 7: the car, admpc_seg.hip; D = 13: the quadrotor,
// admpc_quad.hip), and the bordered 40 x 40 factorisation both build on (dense40.h): ONE text for both vehicle models.
// Mathematics and notation: DESIGN.md section 4, kernel S; tests/seg_spec.py.  Include inside the translation unit's anonymous namespace after
// dense40.h.
//
// Segment 0 delivers  Pbb = Bbar M0^-1 Bbar'  and  dhat = Bbar M0^-1 y0  (= -zb_0: the reduced right-hand side of its Bbar rows), segment 1
// Pi = Qzz - Qzu M1^-1 Quz  and  eta = yz - Qzu M1^-1 y1 (= zb_1).  With Lambda = I + Pbb Pi:
//     dz_1 = Lambda^-1 (dhat + Pbb eta) = Y1 dhat + Y2 eta,      nu_1 = eta - Pi dz_1 = Y3 dhat + Y4 eta
//     Y1 = Lambda^-1,  Y2 = Lambda^-1 Pbb,  Y3 = -Pi Y1,  Y4 = I - Pi Y2
// The four operators depend on the factorisations only: built once per interior-point iteration, applied to every right-hand side.
#pragma once

template <int D> struct ColD { double v[D]; };

// out = A col, A[r][k] at A + r * sa + k (wave-uniform LDS reads)
template <int D>
__device__ __forceinline__ ColD<D> cut_mat_col(const double* A, const int sa, const ColD<D>& x) {
    ColD<D> o;
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) a = fma(A[r * sa + k], x.v[k], a);
        o.v[r] = a;
    }
    return o;
}

// Gaussian elimination with partial pivoting: lanes 0 .. D-1 hold the columns of a D x D matrix, any other lane a right-hand side; on return every
// right-hand-side lane holds its solution.  Row operations are lane-parallel, pivots and multipliers wave-uniform (v_readlane); the row exchange
// is written as selects (as conditional swaps hipcc turns the registers into a dynamically indexed array in scratch).
template <int D>
__device__ __forceinline__ void cut_ge_solve(ColD<D>& c) {
    static_for<0, D>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        double best = fabs(rdlane(c.v[p], p));
        int bi = p;
        static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const double v = fabs(rdlane(c.v[r], p));
            if (v > best) { best = v; bi = r; }
        });
        static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const bool sw = bi == r;
            const double vp = c.v[p], vr = c.v[r];
            c.v[p] = sw ? vr : vp; c.v[r] = sw ? vp : vr;
        });
        const double pinv = rcp_nr(rdlane(c.v[p], p));
        static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const double m = rdlane(c.v[r], p) * pinv;
            c.v[r] = fma(-m, c.v[p], c.v[r]);
        });
    });
    static_for<0, D>([&](auto qc) __attribute__((always_inline)) {
        constexpr int r = D - 1 - decltype(qc)::value;
        double a = c.v[r];
        static_for<r + 1, D>([&](auto cc) __attribute__((always_inline)) {
            constexpr int k = decltype(cc)::value;
            a = fma(-rdlane(c.v[r], k), c.v[k], a);
        });
        c.v[r] = a * rcp_nr(rdlane(c.v[r], r));
    });
}

// The same elimination without a v_readlane: every cross-lane operand -- pivot index, pivot, multipliers, the rows of U in the back substitution -- comes from
// the matrix lane by DPP row broadcast (v_mov_*_dpp / v_fmac_f64_dpp row_newbcast), the pivot search runs on the vector unit in every lane (the one of
// matrix lane p counts).  PRECONDITION: a DPP broadcast stays inside its 16-lane row, so every row that holds right-hand sides holds an identical replica of
// the matrix in its lanes 0 .. D-1 (D <= 8 with groups of eight lanes: matrix | right-hand sides per row).  ~380 instructions against ~570 at D = 7.
template <int L> __device__ __forceinline__ double cut_bc(double v) {
    double r; asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L)); return r;
}
template <int L> __device__ __forceinline__ int cut_bc_i(int v) {
    int r; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L)); return r;
}
template <int D>
__device__ __forceinline__ void cut_ge_solve_dpp(ColD<D>& c) {
    static_assert(D <= 8, "matrix and right-hand sides share a 16-lane row");
    double pinv[D];
    static_for<0, D>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p + 1 < D) {
            double best = fabs(c.v[p]);
            int bi = p;
            static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                const double v = fabs(c.v[r]);
                const bool gt = v > best;
                best = gt ? v : best; bi = gt ? r : bi;
            });
            const int bb = cut_bc_i<p>(bi);
            static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                const bool sw = bb == r;
                const double vp = c.v[p], vr = c.v[r];
                c.v[p] = sw ? vr : vp; c.v[r] = sw ? vp : vr;
            });
        }
        pinv[p] = rcp_nr(cut_bc<p>(c.v[p]));
        double m[D];
        static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) { constexpr int r = decltype(rc)::value; m[r] = -(c.v[r] * pinv[p]); });
        static_for<p + 1, D>([&](auto rc) __attribute__((always_inline)) { constexpr int r = decltype(rc)::value; fmac_rowbc<p>(c.v[r], m[r], c.v[p]); });
    });
    static_for<0, D>([&](auto qc) __attribute__((always_inline)) {
        constexpr int r = D - 1 - decltype(qc)::value;
        double sacc = 0.0;
        static_for<r + 1, D>([&](auto cc) __attribute__((always_inline)) { constexpr int k = decltype(cc)::value; fmac_rowbc<k>(sacc, c.v[r], c.v[k]); });
        c.v[r] = (c.v[r] - sacc) * pinv[r];
    });
}

// The four operators of the cut (one wave).  GS = lanes per group (>= D, 3 GS <= 64): lanes [0, GS) hold the columns of Lambda, [GS, 2 GS) those
// of Pbb, [2 GS, 3 GS) those of I.  Pbb [r * sb + c], Pi [r * GS + c] (written here from Qzz [r * sq + c] + e6 e6' g66 - Szz [r * ss + c]),
// Y [4][D][GS] (Y1, Y2, Y3, Y4; row-major).
template <int D, int GS, bool DPPGE = false>
__device__ __forceinline__ void cut_operators2(const double* Pbb, const int sb, const double* Qzz, const int sq, const double g66, const int i66,
                                               const double* Szz, const int ss, double* Pi, double* Y, const int lane) {
    static_assert(3 * GS <= 64 && GS >= D, "three groups of GS lanes");
    static_assert(!DPPGE || (GS == 8 && D <= 8), "DPP elimination: groups of eight lanes, matrix | right-hand sides in every 16-lane row");
    // roles of the groups of GS lanes.  Plain: Lambda | Pbb | I.  DPPGE: Lambda | Pbb | Lambda (replica for the second 16-lane row) | I, and again
    const int cl = lane % GS, g = lane / GS;
    const int role = DPPGE ? ((g & 3) == 0 || (g & 3) == 2 ? 0 : ((g & 3) == 1 ? 1 : 2)) : (g < 2 ? g : 2);      // 0: Lambda, 1: Pbb, 2: I
    const bool owner = DPPGE ? (g == 1 || g == 3) : (g == 1 || g == 2);                                           // the one copy of a role that stores
    const int cc = cl < D ? cl : 0;
    ColD<D> pi;
#pragma unroll
    for (int r = 0; r < D; ++r) pi.v[r] = Qzz[r * sq + cc] + ((r == i66 && cc == i66) ? g66 : 0.0) - Szz[r * ss + cc];
    if (lane < D) {
#pragma unroll
        for (int r = 0; r < D; ++r) Pi[r * GS + lane] = pi.v[r];
    }
    ColD<D> T = cut_mat_col<D>(Pbb, sb, pi);                         // Pbb Pi[:, c]
#pragma unroll
    for (int r = 0; r < D; ++r) {
        const double pb = Pbb[r * sb + cc];
        const double id = r == cc ? 1.0 : 0.0;
        T.v[r] = role == 0 ? T.v[r] + id : (role == 1 ? pb : id);
    }
    if constexpr (DPPGE) cut_ge_solve_dpp<D>(T); else cut_ge_solve<D>(T);      // role 1: columns of Y2, role 2: columns of Y1
    WSYNC();                                                         // Pi is in LDS
    ColD<D> P = cut_mat_col<D>(Pi, GS, T);                           // Pi X
    if (owner && cl < D) {
        double* Ya = Y + (role == 2 ? 0 : D * GS);                   // Y1 / Y2
        double* Yb = Y + (role == 2 ? 2 * D * GS : 3 * D * GS);      // Y3 / Y4
#pragma unroll
        for (int r = 0; r < D; ++r) {
            Ya[r * GS + cl] = T.v[r];
            Yb[r * GS + cl] = role == 2 ? -P.v[r] : (r == cl ? 1.0 : 0.0) - P.v[r];
        }
    }
}

// one right-hand side through the operators: first segment (its border unknown is -nu_1): returns nu_1[r]; second: dz_1[r]  (lane r < D)
template <int D, int GS>
__device__ __forceinline__ double cut_apply2(const double* Y, const bool first, const double* zb0, const double* zb1, const int lane) {
    const int r = lane < D ? lane : 0;
    const double* Ya = Y + (first ? 2 * D * GS : 0) + r * GS;
    const double* Yb = Ya + D * GS;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) a = fma(Yb[k], zb1[k], fma(-Ya[k], zb0[k], a));
    return a;
}

// ---- the bordered 40 x 40 factorisation:
// M = H + diag(dbar) = L D L' with row i in the registers of lane i and the NR - 40 border rows (full rows of Hb, lanes 40 .. NR-1) riding
// along: Lb = C L^-T D^-1.  LDS: Hp / Lp packed lower rows [820] (the diagonal slots of Lp hold 0.0), Hb / Lb [NR - 40][40], cb: exchange
// buffers in the layout col_head assumes (cb [64], invd [64] = 1 / D_jj, second buffer at cb + 128).
struct Dense40bLds { double *Hp, *Hb, *Lp, *Lb, *cb, *invd; };
// BS = row stride of Hb / Lb in doubles (>= 40).  40 puts border rows b and b + 4 on the same LDS banks (80 dwords = 16 banks apart): every row-wise
// access of the factorisation and the substitutions then carries a two-way conflict (SQ_LDS_BANK_CONFLICT: 44 % of the LDS-active cycles of the car's
// kernel S against 4 % of kernel F's, which has no border); 42 spreads up to 14 rows over distinct bank pairs and keeps the rows 16-byte aligned.
template <int NR, int BS = 40>
__device__ __forceinline__ unsigned dense40b_row_addr(double* tri_, double* brd, const int lz_) {
    return lds_byte_addr(lz_ < 40 ? tri_ + lz_ * (lz_ + 1) / 2 : (lz_ < NR ? brd + (lz_ - 40) * BS : tri_));
}
struct Dense40bNoFix { __device__ __forceinline__ void operator()(double (&)[40]) const {} };
// fix(a): hook behind the row build (the rows are in registers, H's buffer is free): the car's kernel zeroes the factor's diagonal slots there
// (H and L share one buffer) and adds the steering-box barrier to the z6 border row; s_odd: added to the odd columns of this lane's row
template <int NR, int BS = 40, class Fix = Dense40bNoFix>
__device__ __forceinline__ void dense40b_factorise(const Dense40bLds& W, const double dbar_, const int lz_, const double sodd_ = 0.0, const Fix& fix = Fix())
{
    constexpr int n = 40;
    double a[n];
    newton_row_40_b<NR>(a, dense40b_row_addr<NR, BS>(W.Hp, W.Hb, lz_), dbar_, sodd_);
    fix(a);
    const unsigned lrow = dense40b_row_addr<NR, BS>(W.Lp, W.Lb, lz_);
    const unsigned pub_wr = lds_byte_addr(W.cb + lz_), pub_rd = lds_byte_addr(W.cb + (lz_ & 15));
    double* const invd = W.invd;
    auto chain = [&](auto jc, double& nln) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const double dj = rdlane(a[j], j);
        const double dinv = rcp_nr(dj);
        const double lu = a[j] * dinv;
        invd[j] = dinv;
        asm volatile("s_bfm_b64 exec, %2, %3\n\tds_write_b64 %0, %1 offset:%4\n\ts_mov_b64 exec, -1"
                     : : "v"(lrow), "v"(lu), "n"(NR - 1 - j), "n"(j + 1), "n"(8 * j) : "memory");
        nln = -lu;
    };
    double Rb[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, nlb[2] = {0.0, 0.0};
    W.cb[lz_] = a[0];
#pragma unroll
    for (int m = 0; m < 3; ++m) Rb[0][m] = W.cb[16 * m + (lz_ & 15)];
    chain(std::integral_constant<int, 0>{}, nlb[0]);
    static_for<0, n - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr bool wide = NR > 48;                               // rows in the fourth 16-lane row: every column goes through LDS
        constexpr bool own = !wide && (j + 1) / 16 == 2;
        constexpr bool pub = wide ? (j + 2 < n) : (j + 2 < n && (j + 2) / 16 < 2);
        constexpr int mlo = (j + 2) / 16 < 2 ? (j + 2) / 16 : 2;
        double (&R)[3] = Rb[j & 1];
        double (&Rn)[3] = Rb[(j + 1) & 1];
        double& nl = nlb[j & 1];
        double& nln = nlb[(j + 1) & 1];
        col_head<j + 1, mlo, pub, own>(a[j + 1], R, nl, Rn, pub_wr, pub_rd);
        if constexpr (!pub) Rn[2] = a[j + 1];
        chain(std::integral_constant<int, j + 1>{}, nln);
        constexpr int j4 = ((j + 2 + 3) / 4) * 4 < n ? ((j + 2 + 3) / 4) * 4 : n;
        static_for<j + 2, j4>([&](auto c) __attribute__((always_inline)) {
            constexpr int jj = decltype(c)::value;
            if constexpr (own) fmac_rowbc<jj % 16>(a[jj], R[jj / 16], nl);
            else fmac_rowbc_ld<jj % 16>(a[jj], R[jj / 16], nl);
        });
        static_for<j4 / 4, n / 4>([&](auto c) __attribute__((always_inline)) {
            constexpr int jj = 4 * decltype(c)::value;
            if constexpr (own) fmac_rowbc4<jj % 16>(a[jj], a[jj + 1], a[jj + 2], a[jj + 3], R[jj / 16], nl);
            else fmac_rowbc4_ld<jj % 16>(a[jj], a[jj + 1], a[jj + 2], a[jj + 3], R[jj / 16], nl);
        });
    });
    WSYNC();
}
// Schur blocks of the border: out [i * so + j] = (Lb D Lb')[i][j], i, j < NB <= 16, in ten v_mfma_f64_16x16x4_f64
template <int NB, int BS = 40>
__device__ __forceinline__ void dense40b_schur(const Dense40bLds& W, double* out, const int so, const int lane)
{
    typedef double d4_ __attribute__((ext_vector_type(4)));
    const int r16 = lane & 15, kq = lane >> 4;
    const int rb = r16 < NB ? r16 : 0;
    d4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const int c = 4 * t + kq;
        const double lb = W.Lb[rb * BS + c];
        const double dc = rcp_nr(W.invd[c]);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lb * dc, lb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = kq + 4 * v;                                    // C/D layout: register v of lane (kq, r16) is element (kq + 4 v, r16)
        if (i < NB && r16 < NB) out[i * so + r16] = acc[v];
    }
}
// (H v)_lane for the bordered matrix: lanes < 40 the symmetric H, border lanes their full rows of Hb (cb is the exchange buffer)
template <int NR>
__device__ __forceinline__ double dense40b_symv(const Dense40bLds& W, const int lane, const double v, const int lz_)
{
    constexpr int n = 40;
    W.cb[lane] = v;
    WSYNC();
    double Rd3[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) Rd3[m] = W.cb[16 * m + (lane & 15)];
    double hv[n];
    sym_row_40_b<NR>(hv, dense40b_row_addr<NR>(W.Hp, W.Hb, lz_), lds_byte_addr(W.Hp + (lz_ < n ? lz_ : 0)));
    double acc = 0.0;
    static_for<0, n>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        fmac_rowbc_ld<c % 16>(acc, Rd3[c / 16], hv[c]);
    });
    WSYNC();
    return acc;
}
