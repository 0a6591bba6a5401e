// xinv_fusedbih.h -- one pass per sweep for the biharmonic form (gfx950).
//
// numbas.invert_general_bih_2D (reference numbas.py:1204-1586): radius-2 stencil, 9 colours
// (j % 3 major, i % 3 minor).  The colour order decouples into a streaming schedule: with rows
// entering a register window top to bottom, row j of class 0 can take its three column colours as
// soon as row j+2 is in (it reads only old values), class 1 once row j+4 is in (it needs the new
// class-0 rows j-1 and j+2), class 2 once row j+6 is in (new rows j-2, j-1, j+1, j+2).  All three
// conditions fall on the steps r = 2 (mod 3): such a step updates rows r-2, r-4, r-6 in that order
// and retires rows r-8, r-7, r-6 -- a nine-row window, one read and one write of S per sweep instead
// of the row-class kernel's three passes that re-read five rows per updated row.
//
// A wavefront owns a strip of 192 columns -- three adjacent columns per lane = the three column
// colours, neighbours by DPP -- and RB rows.  Updated rows stay in the window, so what the edge lanes
// get wrong feeds the next class.  Colours run west to east, so the error zone is asymmetric: on
// the left a class-0 / 1 / 2 update is wrong in one / two / three lanes; on the right in columns
// E, E-1, E-3 (class 0), up to E-9 (class 1), up to E-15 (class 2), E = the wave's last column.
// THREE halo lanes on the left and SIX on the right (165 owned columns) cover the chain; class-0
// rows read only source values, so nothing accumulates from sweep to sweep.  With periodic x the
// two east columns read five columns to the west (the stale index), and their wrapped copies sit in
// the left halo next to the owned west columns: two more halo lanes on the left (159 owned).  RB is a multiple
// of 3, so every tile starts on a class-0 row and needs no recomputation above it: two source rows
// suffice); below the tile rows y1, y1+1, y1+3 are recomputed from source rows up to y1+5.  Tiles
// do not communicate: S ping-pongs between two buffers; the 'extend' pre-pass (k_extend_bih) runs
// on the source buffer before the sweep.  Periodic x needs xc % 3 == 0 (see k_bih_rowclass).  Same
// ordering as nine colour launches: bitwise equal results.
//
// Where the coefficients come from (template VM):
//   VM = 0  A..I constant along x (Munk / Stommel-Munk with constant A4, R on Cartesian and lat-lon grids): per-row
//           records through the scalar unit (k_row_factor_bih);
//   VM = 1  A, C, D, F vary along x (A4(x, y), R(x, y): apps.py:1793-1836 puts A4 into A and C, R / D into D and F),
//           B == E == 0, G, H, I per row: four vector streams + the point-factor stream Q (round 6);
//   VM = 3  VM = 1 where A and C hold the same numbers everywhere, and D and F do (Cartesian Munk: apps.py:1823-1829 puts A4
//           into both A and C, -R / D into both D and F): C is read out of A's registers, F out of D's -- two streams less;
//   VM = 2  anything: all nine as vector streams + Q.
// Q (k_point_factor_bih, once per coefficient stack): the point's relaxation factor -optArg / denominator
// (numbas.py:1474-1477, same expression, same bits), 0 where the reference's predicate on A..I (or the row) forbids the
// update.  Every coefficient is sampled at the updated point only, so a row's coefficient values are requested a group
// ahead of its update, like its forcing, and used once.
#pragma once
#include "xinv_fused.h"
#include "xinv_pipe2d.h"      /* xinv_cdouble_ptr: loads through the scalar unit */

#define XINV_BIH_OWN(per) ((per) ? 159 : 165)   /* owned columns: lanes 3..57 (5..57 when periodic) x 3 */

// Point update of the one-pass kernel: every coefficient is a per-row scalar here, so the divide
// -optArg / denominator (numbas.py:1474-1477) is the same for the whole row and comes in as `rq`.
// ZBE: B and E are identically zero (Munk / Stommel-Munk on a Cartesian or lat-lon grid: no mixed
// derivatives) -- their terms are exact zeros and are left out, with the diagonal operands they
// would have needed (only the sign of a zero sum could differ; DESIGN.md section 2, "Arithmetic").
template <bool ZBE>
__device__ __forceinline__ double xinv_upd_bih2d_rq(
    double p2_0, double p2_p2, double p2_b, double p1_0, double p1_p1, double p1_m1,
    double r0_0, double r0_p1, double r0_m1, double r0_p2, double r0_m2, double r0_b,
    double m1_0, double m1_p1, double m1_m1, double m2_0, double m2_p2, double m2_b,
    double A, double B, double C, double D, double E, double F, double G, double H, double I,
    double J, double rq, bool cond, bool edge, const XinvScal &sc)
{
    double gterm = G * (
                       (p1_0 - m1_0)
                   );
    gterm = edge ? gterm * sc.delxTr / 2.0 * sc.ratio : gterm * sc.delxTr * sc.ratio / 2.0;
    double temp = A * (
            (p2_0 - 4.0*p1_0 + 6.0*r0_0 - 4.0*m1_0 + m2_0)
        ) * sc.ratioSSr;
    if (!ZBE)
        temp = temp + B * (
            (    p2_p2 - 2.0*p2_0 +     p2_b +
            -2.0*r0_p2 + 4.0*r0_0 - 2.0*r0_b +
                 m2_p2 - 2.0*m2_0 +     m2_b)
        ) * sc.ratioSqr / 16.0;
    temp = temp + C * (
            (r0_p2 - 4.0*r0_p1 + 6.0*r0_0 - 4.0*r0_m1 + r0_m2)
        );
    temp = temp + D * (
            (p1_0 - r0_0)-(r0_0 - m1_0)
        ) * sc.ratioSqr * sc.delxSqr;
    if (!ZBE)
        temp = temp + E * (
            (p1_p1 - m1_p1)-(p1_m1 - m1_m1)
        ) * sc.ratioQtr * sc.delxSqr;
    temp = temp + F * (
            (r0_p1 - r0_0)-(r0_0 - r0_m1)
        ) * sc.delxSqr;
    temp = temp + gterm;
    temp = temp + H * (
            (r0_p1 - r0_m1)
        ) * sc.delxTr / 2.0;
    temp = temp + (
        I * r0_0 - J) * sc.delxSSr;
    temp *= rq;
    return cond ? r0_0 + temp : r0_0;
}

struct FusedBihArgs {
    const double *src;
    double *dst;
    const double *c[10];       // A..I, J
    int64_t sS, sc[10];
    int64_t yc, xc;
    int per;
    int nstrip, nrb, RB;       // x strips, row blocks, rows per block (RB % 3 == 0)
    int nwg;
    int force, no_ctl;
    int64_t member0;
    XinvScal sc_;
    XinvCtl *ctl;
    XinvStop stop;
    unsigned long long *psum;  // [nbatch][XINV_KMAX][NB][3] sequence-tagged norm partials
    const int *tile_list;      // masked-tile skipping, as FusedArgs
    int ntl;
    const double *xsum;
    const long long *xcnt;
    // lagged norm, as FusedArgs
    int lag;
    unsigned tag;
    unsigned long long *lagp_psum;
    const double *lagp_xsum;
    const long long *lagp_xcnt;
    int lagp_NB, lagp_K;
    unsigned lagp_tag;
    const double *rowf;        // per-row records [nbatch][yc][XINV_BIH_RW] (k_row_factor_bih), read through the scalar unit
    const double *q;           // VM > 0: the point-factor stream [nbatch][yc][xc] (k_point_factor_bih)
};

// Per-row record of the one-pass kernel (round 3): A..I of the row, the row's relaxation factor
// -optArg / denominator (numbas.py:1474-1477: every operand is a per-row value here) and the row part of the update
// predicate as an all-ones / zero word -- evaluated once per solve by k_row_factor_bih (the expression the kernel used
// per row and sweep: same bits) and read through the scalar unit one row ahead of its use.  Until then every row
// update began with nine dependent vector loads of the coefficients and an IEEE divide, and the VALU sat idle 60 %
// of the time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.40 at one wavefront per SIMD).
#define XINV_BIH_RW 12

struct RowFactorBihArgs {
    const double *c[9];
    int64_t sc[9];
    int64_t yc, xc;
    XinvScal sc_;
    double *rowf;
};

struct PointFactorBihArgs {
    const double *c[9];           // A..I
    int64_t sc[9];
    int64_t yc, xc, n;            // n = yc * xc
    XinvScal sc_;
    double *q;                    // [nbatch][yc][xc]
    int *flag;                    // bit 0: an updatable point's factor is +-0 (Q == 0 means "skip": the variants are not used);
                                  // bit 1: A and C differ somewhere (bitwise); bit 2: D and F do
};

#ifdef XINV_AUX_KERNELS
__global__ __launch_bounds__(256) void k_row_factor_bih(RowFactorBihArgs a)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (j >= a.yc) return;
    const double u = a.sc_.undef;
    double cs[9];
    bool rowok = (j >= 2 && j <= a.yc - 3);              // rows 0, 1, yc-2, yc-1 are never updated
#pragma unroll
    for (int q = 0; q < 9; q++) { cs[q] = a.c[q][m * a.sc[q] + j * a.xc]; rowok = rowok && (cs[q] != u); }
    const double rq = -a.sc_.optArg / ((cs[0]*a.sc_.ratioSSr + cs[2]) * 6.0 +
                                         cs[1]*a.sc_.ratioSqr / 4.0 +
                                       -(cs[3]*a.sc_.ratioSqr + cs[5]) * 2.0 * a.sc_.delxSqr +
                                         cs[8]*a.sc_.delxSSr);
    double *f = a.rowf + (m * a.yc + j) * XINV_BIH_RW;
#pragma unroll
    for (int q = 0; q < 9; q++) f[q] = cs[q];
    f[9] = rq;
    f[10] = rowok ? __longlong_as_double(-1LL) : 0.0;
    f[11] = 0.0;
}

// once per coefficient stack: Q[j,i] = -optArg / denominator (numbas.py:1474-1477: the expression of the reference, of
// k_row_factor_bih and of the colour kernels, evaluated under the same -ffp-contract=off: the same bits) where the
// reference's predicate on A..I lets the point be updated (numbas.py:1437-1442 without J; rows 2 .. yc-3), 0 elsewhere
__global__ __launch_bounds__(256) void k_point_factor_bih(PointFactorBihArgs a)
{
    const int64_t m = blockIdx.y;
    const double u = a.sc_.undef;
    bool zero = false, dac = false, ddf = false;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < a.n; t += (int64_t)gridDim.x * 256) {
        const int64_t j = t / a.xc;
        double cs[9];
        bool ok = (j >= 2) && (j <= a.yc - 3);
#pragma unroll
        for (int q = 0; q < 9; q++) { cs[q] = a.c[q][m * a.sc[q] + t]; ok = ok && (cs[q] != u); }
        dac = dac || (__double_as_longlong(cs[0]) != __double_as_longlong(cs[2]));
        ddf = ddf || (__double_as_longlong(cs[3]) != __double_as_longlong(cs[5]));
        double qv = 0.0;
        if (ok) {
            qv = -a.sc_.optArg / ((cs[0]*a.sc_.ratioSSr + cs[2]) * 6.0 +
                                    cs[1]*a.sc_.ratioSqr / 4.0 +
                                  -(cs[3]*a.sc_.ratioSqr + cs[5]) * 2.0 * a.sc_.delxSqr +
                                    cs[8]*a.sc_.delxSSr);
            zero = zero || (qv == 0.0);
        }
        a.q[m * a.n + t] = qv;
    }
    if (__any(zero) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 1);
    if (__any(dac) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 2);
    if (__any(ddf) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 4);
}
#endif

template <bool PER, bool ZBE, int VM = 0>
__global__ __launch_bounds__(256, VM == 2 ? 1 : 2) void k_fusedbih(FusedBihArgs a)
{
    static_assert((VM != 1 && VM != 3) || ZBE, "VM = 1, 3: B and E are identically zero");
    constexpr int D = 9;
    constexpr int NV = VM == 0 ? 0 : (VM == 1 ? 4 : (VM == 3 ? 2 : 9));  // coefficient streams read as vectors
    xinv_fresh_scalar_cache();                         // (the per-row records come through the scalar unit: DESIGN.md 4.8)
    const int64_t m = a.member0 + blockIdx.y;
    XinvCtl *ctl = a.ctl + m;
    if (!a.force && xinv_ctl_done(ctl)) return;
    if (a.lag && (int)blockIdx.x == a.nwg) { xinv_lag_reduce_prev(a, ctl, m); return; }
    const unsigned tag = a.lag ? a.tag : xinv_ctl_seq(ctl);

    const int NB = a.nwg;
    int T;
    {
        const int L = blockIdx.x, q = NB >> 3, rem = NB & 7, xcd = L & 7, idx = L >> 3;
        T = xcd * q + (xcd < rem ? xcd : rem) + idx;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int wt = T * 4 + wave;
    bool active = wt < a.nstrip * a.nrb;
    if (a.tile_list) {
        wt = a.tile_list[m * a.ntl + wt];
        active = wt >= 0;
    }
    // the tile is the wavefront's: its index in an SGPR makes every row quantity below scalar (row bases as SGPR
    // pairs, the march's compares on the scalar unit, the record loads s_loads)
    wt = __builtin_amdgcn_readfirstlane(active ? wt : 0);
    active = __builtin_amdgcn_readfirstlane((int)active) != 0;
    const int rb = active ? wt / a.nstrip : 0, strip = active ? wt - rb * a.nstrip : 0;
    const int64_t xc = a.xc, yc = a.yc;
    const int64_t y0 = (int64_t)rb * a.RB;
    const int64_t y1 = (rb + 1 == a.nrb) ? yc : y0 + a.RB;
    const double u = a.sc_.undef;
    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;

    double acc[1] = {0.0};
    int cnt[1] = {0};

    if (active) {
        constexpr int LH = PER ? 5 : 3;                              // halo lanes on the left (6 on the right)
        const int64_t c = (int64_t)strip * XINV_BIH_OWN(PER) - 3 * LH + 3 * lane;   // unwrapped first column (c % 3 == 0)
        int64_t lcol[3];
        bool upd[3], own[3], edge[3], east[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int64_t cc = c + k;
            if (PER) {
                int64_t w = cc % xc; if (w < 0) w += xc;
                lcol[k] = w;
                upd[k] = true;
                edge[k] = (w < 2) || (w >= xc - 2);
                east[k] = (w >= xc - 2);
                own[k] = (cc >= 0) && (cc < xc) && (lane >= LH) && (lane < 58);
            } else {
                lcol[k] = cc < 0 ? 0 : (cc > xc - 1 ? xc - 1 : cc);
                upd[k] = (cc >= 2) && (cc <= xc - 3);
                edge[k] = false; east[k] = false;
                own[k] = (cc >= 0) && (cc < xc) && (lane >= LH) && (lane < 58);
            }
        }
        const double *pJ = a.c[9] + m * a.sc[9];

        // rows as 32-bit scalars (clamps on the scalar unit: 64-bit compares are VALU instructions on this target, and
        // each one a round trip VALU -> scalar unit), the lane's columns as 32-bit byte offsets: every access is
        // `uniform row base (SGPR pair) + 32-bit lane offset`, no 64-bit address arithmetic per load
        typedef int row_t;
        const int yci = (int)yc;
        unsigned boff[3];
#pragma unroll
        for (int k = 0; k < 3; k++) boff[k] = (unsigned)lcol[k] * 8u;
        // (a raw buffer resource per row -- base = the row, range = its bytes -- makes the access
        //  `buffer_load_dwordx2 v, v_off, s[rsrc], 0 offen`: three scalar instructions per row instead of a 64-bit
        //  vector add per load; no limit on the size of the slice)
        const int rowbytes = (int)(xc * 8);
        auto load_row = [&](const double *base, int r) {
            const int rr = min(max(r, 0), yci - 1);
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void *)(base + (int64_t)rr * xc), 0, rowbytes, 0x00020000);
            Tri t;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const auto w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)boff[k], 0, 0);
                t.v[k] = __hiloint2double((int)w[1], (int)w[0]);
            }
            return t;
        };

        // The solve's scalars in VECTOR registers (every lane the same value): with them, the records, the lane masks and
        // the row pointers in SGPRs the compiler ran out (106) and re-read kernel arguments from memory, waiting for
        // each, a dozen times per group of three rows.
        XinvScal scl = a.sc_;
        asm("" : "+v"(scl.ratioSSr), "+v"(scl.ratioSqr), "+v"(scl.delxSqr), "+v"(scl.delxTr), "+v"(scl.ratio),
                 "+v"(scl.delxSSr), "+v"(scl.ratioQtr));
        double uv = u;
        asm("" : "+v"(uv));
#define u uv
        Tri W[D];
#pragma unroll
        for (int t = 0; t < D; t++) { W[t].v[0] = 0.0; W[t].v[1] = 0.0; W[t].v[2] = 0.0; }
        Tri Jst[3];
#pragma unroll
        for (int t = 0; t < 3; t++) Jst[t] = W[0];

        struct Rec { double cs[9]; double rq; double rok; };
        const xinv_cdouble_ptr rowf = (xinv_cdouble_ptr)(uintptr_t)(a.rowf + m * yc * XINV_BIH_RW);
        // the record of row j (clamped: rows outside 2 .. yc-3 carry a zero predicate and are left alone)
        // (VM = 2 reads nothing from it: the loads fall away)
        auto ldrec = [&](int j) {
            Rec R;
            if constexpr (VM == 2) {
#pragma unroll
                for (int q = 0; q < 9; q++) R.cs[q] = 0.0;
                R.rq = 0.0; R.rok = 0.0;
            } else {
                const int jj = min(max(j, 0), (int)yc - 1);
                const xinv_cdouble_ptr pr = rowf + (int64_t)jj * XINV_BIH_RW;
#pragma unroll
                for (int q = 0; q < 9; q++) R.cs[q] = pr[q];
                R.rq = pr[9]; R.rok = pr[10];
            }
            return R;
        };
        // VM > 0: the row's coefficient values at the lane's columns and its point factors, requested a group ahead of
        // the row's update like its forcing (one request per stream and row; every value is used exactly once)
        constexpr int NVA = NV > 0 ? NV : 1;
        struct CoefV { Tri a[NVA]; Tri q; };
        const double *cpv[NVA];
        const double *pQ = nullptr;
        if constexpr (VM > 0) {
            constexpr int idx1[4] = {0, 2, 3, 5};            // A, C, D, F
            constexpr int idx3[2] = {0, 3};                  // A (= C), D (= F)
#pragma unroll
            for (int t = 0; t < NV; t++) {
                const int q = (VM == 1) ? idx1[t < 4 ? t : 0] : (VM == 3 ? idx3[t < 2 ? t : 0] : t);
                cpv[t] = a.c[q] + m * a.sc[q];
            }
            pQ = a.q + m * yc * xc;
        } else {
            cpv[0] = nullptr;
        }
        auto ldcoef = [&](int j) {
            CoefV v;
            if constexpr (VM > 0) {
#pragma unroll
                for (int t = 0; t < NV; t++) v.a[t] = load_row(cpv[t], j);
                v.q = load_row(pQ, j);
            } else {
                v.a[0] = W[0]; v.q = W[0];
            }
            return v;
        };
        // two sets live: `cvc` -- the row about to be updated -- and `cvn`, requested one update earlier for the update
        // after it (a set per class, requested a whole group ahead like the forcing, is 90 VGPRs: A, C, D, F, Q x three
        // columns x three classes -- the variant spilled at 256)
        CoefV cvc, cvn;

        // the three column colours of row j (window slot SJ), forcing row Jt
        auto upd_row = [&](auto sjtag, int, const Tri &Jt, const Rec &R, const CoefV &V) {
            constexpr int SJ = decltype(sjtag)::value;
            constexpr int SM2 = (SJ + 7) % D, SM1 = (SJ + 8) % D, SP1 = (SJ + 1) % D, SP2 = (SJ + 2) % D;
            const double (&cs)[9] = R.cs;
            const bool rowok = __double_as_longlong(R.rok) != 0;
            double em2[7], em1[7], ep1[7], ep2[7];
            bih_ext(W[SM2], em2); bih_ext(W[SM1], em1); bih_ext(W[SP1], ep1); bih_ext(W[SP2], ep2);
            double fm2[2] = {0.0, 0.0}, fp2[2] = {0.0, 0.0};
            if (PER && !ZBE) { bih_far(W[SM2], fm2[0], fm2[1]); bih_far(W[SP2], fp2[0], fp2[1]); }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double e0[7];
                bih_ext(W[SJ], e0);
                const int o = k + 2;
                double p2_b = ep2[o - 2], r0_b = e0[o - 2], m2_b = em2[o - 2];
                if (PER && !ZBE && k > 0) {
                    double f0[2];
                    bih_far(W[SJ], f0[0], f0[1]);
                    if (east[k]) { p2_b = fp2[k - 1]; r0_b = f0[k - 1]; m2_b = fm2[k - 1]; }
                }
                // the coefficients of the point: per-row values out of the record, or the lane's own out of the streams
                double cA, cB, cC, cD, cE, cF, cG, cH, cI, rq;
                bool cok;
                if constexpr (VM == 0) {
                    cA = cs[0]; cB = cs[1]; cC = cs[2]; cD = cs[3]; cE = cs[4]; cF = cs[5]; cG = cs[6]; cH = cs[7]; cI = cs[8];
                    rq = R.rq; cok = rowok;
                } else if constexpr (VM == 1) {
                    cA = V.a[0].v[k]; cC = V.a[1].v[k]; cD = V.a[2].v[k]; cF = V.a[3].v[k];
                    cB = 0.0; cE = 0.0; cG = cs[6]; cH = cs[7]; cI = cs[8];
                    rq = V.q.v[k]; cok = (rq != 0.0);
                } else if constexpr (VM == 3) {
                    cA = V.a[0].v[k]; cC = cA; cD = V.a[1].v[k]; cF = cD;
                    cB = 0.0; cE = 0.0; cG = cs[6]; cH = cs[7]; cI = cs[8];
                    rq = V.q.v[k]; cok = (rq != 0.0);
                } else {
                    cA = V.a[0].v[k]; cB = V.a[1].v[k]; cC = V.a[2].v[k]; cD = V.a[3].v[k]; cE = V.a[4].v[k];
                    cF = V.a[5].v[k]; cG = V.a[6].v[k]; cH = V.a[7].v[k]; cI = V.a[8].v[k];
                    rq = V.q.v[k]; cok = (rq != 0.0);
                }
                W[SJ].v[k] = xinv_upd_bih2d_rq<ZBE>(
                    ep2[o], ep2[o + 2], p2_b, ep1[o], ep1[o + 1], ep1[o - 1],
                    e0[o], e0[o + 1], e0[o - 1], e0[o + 2], e0[o - 2], r0_b,
                    em1[o], em1[o + 1], em1[o - 1], em2[o], em2[o + 2], m2_b,
                    cA, cB, cC, cD, cE, cF, cG, cH, cI,
                    Jt.v[k], rq, upd[k] && cok && (Jt.v[k] != u), edge[k], scl);
            }
        };
        unsigned sof[3];                                     // store offsets: a column the lane does not own lies beyond the
#pragma unroll                                               // row's range and the store is dropped (no branch)
        for (int k = 0; k < 3; k++) sof[k] = own[k] ? boff[k] : 0xffffffffu;
        auto retire = [&](auto stag, row_t j) {
            constexpr int SL = decltype(stag)::value;
            if (j < (row_t)y0 || j >= (row_t)y1) return;
            const __amdgpu_buffer_rsrc_t rsd =
                __builtin_amdgcn_make_buffer_rsrc((void *)(dstS + (int64_t)j * xc), 0, rowbytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double v = W[SL].v[k];
                typedef unsigned xinv_v2u_ __attribute__((__vector_size__(8)));
                const xinv_v2u_ tv = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
                __builtin_amdgcn_raw_buffer_store_b64(tv, rsd, (int)sof[k], 0, 0);
                const bool c = own[k] && (v != u);
                acc[0] += c ? fabs(v) : 0.0;                 // (+0.0 leaves a sum of magnitudes unchanged bit for bit)
                cnt[0] += c ? 1 : 0;
            }
        };

        // Rows y0-3 .. y1+7 stream through the window three at a time; with r = base + 2 = 2 (mod 3), rows r-2 / r-4 / r-6
        // are logical slots 6 / 4 / 2 and rows r-8..r-6 (slots 0..2) retire.  The next group's three rows are requested
        // before the updates of this one, the forcing row (and, VM > 0, the coefficient rows) of each class right after
        // their use.
        // The window ROTATES instead of being shifted: group g keeps logical slot s in register slot (s + 3 g) mod 9,
        // the march is unrolled three groups so that every slot is a compile-time register name, and what moves per
        // group is only the three prefetched rows into their slots (they are the slots of the rows retiring in the
        // group before, which the class-2 update still reads): 55 -> 9 register moves per group of ~600 vector
        // instructions.
        const row_t rstart = (row_t)y0 - 3, rlast = (row_t)y1 + 7;   // row j retires by step j + 8
        Tri N0 = load_row(srcS, rstart), N1 = load_row(srcS, rstart + 1), N2 = load_row(srcS, rstart + 2);
        Jst[0] = load_row(pJ, rstart + 2 - 2); Jst[1] = load_row(pJ, rstart + 2 - 4); Jst[2] = load_row(pJ, rstart + 2 - 6);
        cvc = ldcoef(rstart + 2 - 2); cvn = ldcoef(rstart + 2 - 4);
        Rec R0 = ldrec(rstart + 2 - 2);
        auto group = [&](auto phtag, row_t base) {
            constexpr int PH = decltype(phtag)::value;
#define XINV_BIH_P(sl) std::integral_constant<int, ((sl) + 3 * PH) % D>{}
            const row_t r = base + 2;
            W[((6) + 3 * PH) % D] = N0; W[((7) + 3 * PH) % D] = N1; W[((8) + 3 * PH) % D] = N2;
            N0 = load_row(srcS, base + 3); N1 = load_row(srcS, base + 4); N2 = load_row(srcS, base + 5);
            const Rec R1 = ldrec(r - 4);
            upd_row(XINV_BIH_P(6), r - 2, Jst[0], R0, cvc);     // class 0
            Jst[0] = load_row(pJ, r + 3 - 2);
            if constexpr (VM > 0) { cvc = cvn; cvn = ldcoef(r - 6); }
            const Rec R2 = ldrec(r - 6);
            upd_row(XINV_BIH_P(4), r - 4, Jst[1], R1, cvc);     // class 1
            Jst[1] = load_row(pJ, r + 3 - 4);
            if constexpr (VM > 0) { cvc = cvn; cvn = ldcoef(r + 3 - 2); }
            R0 = ldrec(r + 3 - 2);
            upd_row(XINV_BIH_P(2), r - 6, Jst[2], R2, cvc);     // class 2
            Jst[2] = load_row(pJ, r + 3 - 6);
            if constexpr (VM > 0) { cvc = cvn; cvn = ldcoef(r + 3 - 4); }
            retire(XINV_BIH_P(0), r - 8);
            retire(XINV_BIH_P(1), r - 7);
            retire(XINV_BIH_P(2), r - 6);
#undef XINV_BIH_P
        };
        for (row_t base = rstart; base <= rlast; base += 9) {
            group(std::integral_constant<int, 0>{}, base);
            if (base + 3 > rlast) break;
            group(std::integral_constant<int, 1>{}, base + 3);
            if (base + 6 > rlast) break;
            group(std::integral_constant<int, 2>{}, base + 6);
        }
    }

#undef u
    if (a.no_ctl) return;
    xinv_norm_tail<1>(a, acc, cnt, wave, lane, NB, T, tag, ctl, m);
}
