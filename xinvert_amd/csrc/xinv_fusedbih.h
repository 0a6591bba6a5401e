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
// on the source buffer before the sweep.  Requires the coefficient arrays A..I to be constant
// along x (per-row scalars: Munk / Stommel-Munk on Cartesian and lat-lon grids); otherwise the
// row-class kernel runs.  Periodic x needs xc % 3 == 0 (see k_bih_rowclass).  Same ordering as
// nine colour launches: bitwise equal results.
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
    const double *c[10];       // A..I (x-uniform), J
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
};

// Per-row record of the one-pass kernel (round 3): A..I of the row, the row's relaxation factor
// -optArg / denominator (numbas.py:1474-1477: every operand is a per-row value here) and the row part of the update
// predicate as an all-ones / zero word -- evaluated once per solve by k_row_factor_bih (the expression the kernel used
// per row and sweep: same bits) and read through the scalar unit one row ahead of its use.  Until then every row
// update began with nine dependent vector loads of the coefficients and an IEEE divide, and the VALU sat idle 60 %
// of the time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.40 at one wavefront per SIMD).
#define XINV_BIH_RW 12
#ifndef XINV_BIH_REC
#define XINV_BIH_REC 1
#endif

struct RowFactorBihArgs {
    const double *c[9];
    int64_t sc[9];
    int64_t yc, xc;
    XinvScal sc_;
    double *rowf;
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
#endif

#ifndef XINV_BIH_MINWAVES
#define XINV_BIH_MINWAVES 2
#endif
template <bool PER, bool ZBE>
__global__ __launch_bounds__(256, XINV_BIH_MINWAVES) void k_fusedbih(FusedBihArgs a)
{
    constexpr int D = 9;
#if XINV_BIH_REC
    xinv_fresh_scalar_cache();                         // (the per-row records come through the scalar unit: DESIGN.md 4.8)
#endif
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
#if XINV_BIH_REC
    // the tile is the wavefront's: its index in an SGPR makes every row quantity below scalar (row bases as SGPR
    // pairs, the march's compares on the scalar unit, the record loads s_loads)
    wt = __builtin_amdgcn_readfirstlane(active ? wt : 0);
    active = __builtin_amdgcn_readfirstlane((int)active) != 0;
#endif
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
        const double *cp[9];
#pragma unroll
        for (int q = 0; q < 9; q++) cp[q] = a.c[q] + m * a.sc[q];
        const double *pJ = a.c[9] + m * a.sc[9];

#if XINV_BIH_REC
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
#else
        typedef int64_t row_t;
        auto load_row = [&](const double *base, int64_t r) {
            const int64_t rr = r < 0 ? 0 : (r > yc - 1 ? yc - 1 : r);
            const double *row = base + rr * xc;
            Tri t;
#pragma unroll
            for (int k = 0; k < 3; k++) t.v[k] = row[lcol[k]];
            return t;
        };
#endif

#ifndef XINV_BIH_VSCAL
#define XINV_BIH_VSCAL XINV_BIH_REC
#endif
        // The solve's scalars in VECTOR registers (every lane the same value): with them, the records, the lane masks and
        // the row pointers in SGPRs the compiler ran out (106) and re-read kernel arguments from memory, waiting for
        // each, a dozen times per group of three rows.
        XinvScal scl = a.sc_;
#if XINV_BIH_VSCAL
        asm("" : "+v"(scl.ratioSSr), "+v"(scl.ratioSqr), "+v"(scl.delxSqr), "+v"(scl.delxTr), "+v"(scl.ratio),
                 "+v"(scl.delxSSr), "+v"(scl.ratioQtr));
        double uv = u;
        asm("" : "+v"(uv));
#define u uv
#endif
        Tri W[D];
#pragma unroll
        for (int t = 0; t < D; t++) { W[t].v[0] = 0.0; W[t].v[1] = 0.0; W[t].v[2] = 0.0; }
        Tri Jst[3];
#pragma unroll
        for (int t = 0; t < 3; t++) Jst[t] = W[0];

#if XINV_BIH_REC
        struct Rec { double cs[9]; double rq; double rok; };
        const xinv_cdouble_ptr rowf = (xinv_cdouble_ptr)(uintptr_t)(a.rowf + m * yc * XINV_BIH_RW);
        // the record of row j (clamped: rows outside 2 .. yc-3 carry a zero predicate and are left alone)
        auto ldrec = [&](int j) {
            const int jj = min(max(j, 0), (int)yc - 1);
            const xinv_cdouble_ptr pr = rowf + (int64_t)jj * XINV_BIH_RW;
            Rec R;
#pragma unroll
            for (int q = 0; q < 9; q++) R.cs[q] = pr[q];
            R.rq = pr[9]; R.rok = pr[10];
            return R;
        };
#endif
        // the three column colours of row j (window slot SJ), forcing row Jt
#if XINV_BIH_REC
        auto upd_row = [&](auto sjtag, int, const Tri &Jt, const Rec &R) {
            constexpr int SJ = decltype(sjtag)::value;
            constexpr int SM2 = (SJ + 7) % D, SM1 = (SJ + 8) % D, SP1 = (SJ + 1) % D, SP2 = (SJ + 2) % D;
            const double (&cs)[9] = R.cs;
            const double rq = R.rq;
            const bool rowok = __double_as_longlong(R.rok) != 0;
#else
        auto upd_row = [&](auto sjtag, int64_t j, const Tri &Jt) {
            constexpr int SJ = decltype(sjtag)::value;
            constexpr int SM2 = (SJ + 7) % D, SM1 = (SJ + 8) % D, SP1 = (SJ + 1) % D, SP2 = (SJ + 2) % D;
            if (j < 2 || j > yc - 3) return;                       // rows 0, 1, yc-2, yc-1 are never updated
            double cs[9];
#pragma unroll
            for (int q = 0; q < 9; q++) cs[q] = cp[q][j * xc];     // one value per row
            bool rowok = true;
#pragma unroll
            for (int q = 0; q < 9; q++) rowok = rowok && (cs[q] != u);
            const double rq = -a.sc_.optArg / ((cs[0]*a.sc_.ratioSSr + cs[2]) * 6.0 +
                                                 cs[1]*a.sc_.ratioSqr / 4.0 +
                                               -(cs[3]*a.sc_.ratioSqr + cs[5]) * 2.0 * a.sc_.delxSqr +
                                                 cs[8]*a.sc_.delxSSr);
#endif
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
                W[SJ].v[k] = xinv_upd_bih2d_rq<ZBE>(
                    ep2[o], ep2[o + 2], p2_b, ep1[o], ep1[o + 1], ep1[o - 1],
                    e0[o], e0[o + 1], e0[o - 1], e0[o + 2], e0[o - 2], r0_b,
                    em1[o], em1[o + 1], em1[o - 1], em2[o], em2[o + 2], m2_b,
                    cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], cs[6], cs[7], cs[8],
                    Jt.v[k], rq, upd[k] && rowok && (Jt.v[k] != u), edge[k], scl);
            }
        };
#if XINV_BIH_REC
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
#else
        auto retire = [&](auto stag, row_t j) {
            constexpr int SL = decltype(stag)::value;
            if (j < (row_t)y0 || j >= (row_t)y1) return;
            double *row = dstS + j * xc;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double v = W[SL].v[k];
                if (own[k]) {
                    row[lcol[k]] = v;
                    if (v != u) { acc[0] += fabs(v); cnt[0] += 1; }
                }
            }
        };
#endif

        // Rows y0-3 .. y1+7 stream through the window three at a time.  The window is shifted by
        // three rows per group (18 register moves against ~900 instructions of updates), so the
        // newest row always sits in slot 8 and every slot index below is a constant: with
        // r = base + 2 = 2 (mod 3), rows r-2 / r-4 / r-6 are slots 6 / 4 / 2 and rows r-8..r-6
        // (slots 0..2) retire.  The next group's three rows and forcing rows are requested before
        // the updates of this one.
        const row_t rstart = (row_t)y0 - 3, rlast = (row_t)y1 + 7;   // row j retires by step j + 8
#ifndef XINV_BIH_PREFETCH
#define XINV_BIH_PREFETCH XINV_BIH_REC
#endif
#if XINV_BIH_PREFETCH
        Tri N0 = load_row(srcS, rstart), N1 = load_row(srcS, rstart + 1), N2 = load_row(srcS, rstart + 2);
#endif
        Jst[0] = load_row(pJ, rstart + 2 - 2); Jst[1] = load_row(pJ, rstart + 2 - 4); Jst[2] = load_row(pJ, rstart + 2 - 6);
#if XINV_BIH_REC
        Rec R0 = ldrec(rstart + 2 - 2);
#endif
#ifndef XINV_BIH_ROT
#define XINV_BIH_ROT XINV_BIH_REC
#endif
#if XINV_BIH_ROT
        // The window ROTATES instead of being shifted: group g keeps logical slot s in register slot (s + 3 g) mod 9,
        // the march is unrolled three groups so that every slot is a compile-time register name, and what moves per
        // group is only the three prefetched rows into their slots (they are the slots of the rows retiring in the
        // group before, which the class-2 update still reads).  The forcing row of each class is re-requested right
        // after its use.  55 -> 9 register moves per group of ~600 vector instructions.
        auto group = [&](auto phtag, row_t base) {
            constexpr int PH = decltype(phtag)::value;
#define XINV_BIH_P(sl) std::integral_constant<int, ((sl) + 3 * PH) % D>{}
            const row_t r = base + 2;
            W[((6) + 3 * PH) % D] = N0; W[((7) + 3 * PH) % D] = N1; W[((8) + 3 * PH) % D] = N2;
            N0 = load_row(srcS, base + 3); N1 = load_row(srcS, base + 4); N2 = load_row(srcS, base + 5);
            const Rec R1 = ldrec(r - 4);
            upd_row(XINV_BIH_P(6), r - 2, Jst[0], R0);          // class 0
            Jst[0] = load_row(pJ, r + 3 - 2);
            const Rec R2 = ldrec(r - 6);
            upd_row(XINV_BIH_P(4), r - 4, Jst[1], R1);          // class 1
            Jst[1] = load_row(pJ, r + 3 - 4);
            R0 = ldrec(r + 3 - 2);
            upd_row(XINV_BIH_P(2), r - 6, Jst[2], R2);          // class 2
            Jst[2] = load_row(pJ, r + 3 - 6);
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
#else
        for (row_t base = rstart; base <= rlast; base += 3) {
            const row_t r = base + 2;
#pragma unroll
            for (int t = 0; t < 6; t++) W[t] = W[t + 3];
#if XINV_BIH_PREFETCH
            W[6] = N0; W[7] = N1; W[8] = N2;
            const Tri J0 = Jst[0], J1 = Jst[1], J2 = Jst[2];
            N0 = load_row(srcS, base + 3); N1 = load_row(srcS, base + 4); N2 = load_row(srcS, base + 5);
#else
            W[6] = load_row(srcS, base); W[7] = load_row(srcS, base + 1); W[8] = load_row(srcS, base + 2);
            const Tri J0 = Jst[0], J1 = Jst[1], J2 = Jst[2];
#endif
            Jst[0] = load_row(pJ, r + 3 - 2); Jst[1] = load_row(pJ, r + 3 - 4); Jst[2] = load_row(pJ, r + 3 - 6);
#if XINV_BIH_REC
            // each row's record is asked for one row update (~1000 cycles of arithmetic) before it is used
            const Rec R1 = ldrec(r - 4);
            upd_row(std::integral_constant<int, 6>{}, r - 2, J0, R0);   // class 0
            const Rec R2 = ldrec(r - 6);
            upd_row(std::integral_constant<int, 4>{}, r - 4, J1, R1);   // class 1
            R0 = ldrec(r + 3 - 2);
            upd_row(std::integral_constant<int, 2>{}, r - 6, J2, R2);   // class 2
#else
            upd_row(std::integral_constant<int, 6>{}, r - 2, J0);   // class 0
            upd_row(std::integral_constant<int, 4>{}, r - 4, J1);   // class 1
            upd_row(std::integral_constant<int, 2>{}, r - 6, J2);   // class 2
#endif
            retire(std::integral_constant<int, 0>{}, r - 8);
            retire(std::integral_constant<int, 1>{}, r - 7);
            retire(std::integral_constant<int, 2>{}, r - 6);
        }
#endif
    }

#undef u
    if (a.no_ctl) return;
    xinv_norm_tail<1>(a, acc, cnt, wave, lane, NB, T, tag, ctl, m);
}
