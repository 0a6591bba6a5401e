// xinv_fused9.h -- streaming fused 4-colour SOR sweep(s) for the 2-D 9-point forms (gfx950).
//
// The cross-derivative coefficient B couples the diagonal neighbours, which share a red-black
// colour; the engine then orders a sweep by the four colours (j&1, i&1):
//     c0 even row / even column, c1 even / odd, c2 odd / even, c3 odd / odd.
// With two adjacent columns per lane (.x even, .y odd) an even row carries c0 (.x) and c1 (.y),
// an odd row c2 (.x) and c3 (.y).  Rows stream through the same rotating register window as the
// 5-point kernel, two rows per step: with r the odd row just loaded, sweep s updates
//     E_s: even row r-2s+1   (c0 then c1; its odd neighbours r-2s, r-2s+2 are still "old")
//     O_s: odd  row r-2s     (c2 then c3; its even neighbours have finished E_s)
// so rows r-2K+1 and r-2K leave with K complete sweeps.  Every colour stage reads the 3x3
// neighbourhood: the centre column from the lane's own registers, one side column from its own
// other component and the other side column from the neighbouring lane with three DPP wave
// shifts (rows j-1, j, j+1).  Each colour consumes one halo column per side (c0..c3: 4 per sweep)
// and each sweep two halo rows; S ping-pongs between two buffers; norm and stopping rule as in
// the 5-point kernel.  Bitwise equal to the oracle's 4-colour ordering (tests).
#pragma once
#include "xinv_fused.h"

// 9-point updates with the mask folded into a select (cf. xinv_upd_std2d_9 / xinv_upd_gen2d_9).
__device__ __forceinline__ double xinv_upd_gen2d_9_sel(
    double sC, double sP, double sM, double sW, double sE,
    double sPE, double sPW, double sME, double sMW,
    double A, double B, double C, double D, double E, double F, double G, bool inr,
    const XinvScal &sc)
{
    const double u = sc.undef;
    const bool cond = inr && (G != u) && (A != u) && (B != u) && (C != u) && (D != u) &&
                      (E != u) && (F != u);
    double temp = (
        A * (
            (sP - sC) - (sC - sM)
        ) * sc.ratioSqr +
        B * (
            (sPE - sME) - (sPW - sMW)
        ) * sc.ratioQtr +
        C * (
            (sE - sC) - (sC - sW)
        ) + (
        D * (
            (sP - sM)
        ) * sc.ratio +
        E * (
            (sE - sW)
        )) * sc.delx / 2.0 + (
        F * sC - G) * sc.delxSqr
    );
    temp *= sc.optArg / ((A * sc.ratioSqr + C) * 2.0
                         - F * sc.delxSqr);
    return cond ? sC + temp : sC;
}

__device__ __forceinline__ double xinv_upd_std2d_9_sel(
    double sC, double sP, double sM, double sW, double sE,
    double sPE, double sPW, double sME, double sMW, double sM_q,
    double aP, double a0, double bE, double bW, double bP_chk, double bP_use, double bM,
    double cE, double c0, double f, bool inr, const XinvScal &sc)
{
    const double u = sc.undef;
    const bool cond = inr && (f != u) && (aP != u) && (a0 != u) && (bE != u) && (bW != u) &&
                      (bP_chk != u) && (bM != u) && (cE != u) && (c0 != u);
    double temp = (
        (
            aP * (sP - sC) -
            a0 * (sC - sM)
        ) * sc.ratioSqr + (
            bP_use * (sPE - sPW) -
            bM * (sM_q - sMW)
        ) * sc.ratioQtr + (
            bE * (sPE - sME) -
            bW * (sPW - sMW)
        ) * sc.ratioQtr + (
            cE * (sE - sC) -
            c0 * (sC - sW)
        )
    ) - f * sc.delxSqr;
    temp *= sc.optArg / ((aP + a0) * sc.ratioSqr + (cE + c0));
    return cond ? sC + temp : sC;
}

// the 3x3 neighbourhood of component X on the row in slot sj
struct Nbr9 { double c, p, m, w, e, pe, pw, me, mw; };

// (FIX: the seam lanes' pass of the ring layout -- X == 0, the east column is the NEXT lane's .x, column 0, instead of the
//  lane's own .y, the phantom column: xinv_fused.h RING)
template <int X, bool FIX = false>
__device__ __forceinline__ Nbr9 nbr9(const double2 &Rm, const double2 &R0, const double2 &Rp)
{
    static_assert(!FIX || X == 0, "column xc-1 sits in an .x slot");
    Nbr9 n;
    n.c = comp<X>(R0); n.p = comp<X>(Rp); n.m = comp<X>(Rm);
    if (X == 0) {
        n.w = xinv_lane_up(R0.y); n.pw = xinv_lane_up(Rp.y); n.mw = xinv_lane_up(Rm.y);
        if (FIX) { n.e = xinv_lane_down(R0.x); n.pe = xinv_lane_down(Rp.x); n.me = xinv_lane_down(Rm.x); }
        else { n.e = R0.y; n.pe = Rp.y; n.me = Rm.y; }
    } else {
        n.w = R0.x; n.pw = Rp.x; n.mw = Rm.x;
        n.e = xinv_lane_down(R0.x); n.pe = xinv_lane_down(Rp.x); n.me = xinv_lane_down(Rm.x);
    }
    return n;
}

struct Fused9Gen {                  // numbas.invert_general_2D, B != 0
    static constexpr int NC = 7;    // A, B, C, D, E, F, G
    template <int X, int D, bool FIX = false>
    static __device__ __forceinline__ double upd(const double2 (&cw)[NC][D], const double2 (&sw)[D],
                                                 int sj, int sjp, int sjm, bool inr, bool,
                                                 const XinvScal &sc)
    {
        const Nbr9 n = nbr9<X, FIX>(sw[sjm], sw[sj], sw[sjp]);
        return xinv_upd_gen2d_9_sel(n.c, n.p, n.m, n.w, n.e, n.pe, n.pw, n.me, n.mw,
                                    comp<X>(cw[0][sj]), comp<X>(cw[1][sj]), comp<X>(cw[2][sj]),
                                    comp<X>(cw[3][sj]), comp<X>(cw[4][sj]), comp<X>(cw[5][sj]),
                                    comp<X>(cw[6][sj]), inr, sc);
    }
};

struct Fused9Std {                  // numbas.invert_standard_2D, B != 0
    static constexpr int NC = 4;    // A, B, C, F
    // `west`: this lane's .x is (wrapped) column 0 of a periodic row -- the reference's i == 0
    // branch multiplies by B[j+1,1] and differences S[j-1,0] - S[j-1,-1] (numbas.py:327-328).
    template <int X, int D, bool FIX = false>
    static __device__ __forceinline__ double upd(const double2 (&cw)[NC][D], const double2 (&sw)[D],
                                                 int sj, int sjp, int sjm, bool inr, bool west,
                                                 const XinvScal &sc)
    {
        const Nbr9 n = nbr9<X, FIX>(sw[sjm], sw[sj], sw[sjp]);
        const double aP = comp<X>(cw[0][sjp]), a0 = comp<X>(cw[0][sj]);
        const double bP_chk = comp<X>(cw[1][sjp]), bM = comp<X>(cw[1][sjm]);
        const double c0 = comp<X>(cw[2][sj]), f = comp<X>(cw[3][sj]);
        double bE, bW, cE, bP_use = bP_chk, sM_q = n.me;
        if (X == 0) {
            bW = xinv_lane_up(cw[1][sj].y); bE = cw[1][sj].y; cE = cw[2][sj].y;
            if (FIX) { bE = xinv_lane_down(cw[1][sj].x); cE = xinv_lane_down(cw[2][sj].x); }   // (column 0's, not the phantom's)
            if (west) { bP_use = cw[1][sjp].y; sM_q = n.m; }
        } else {
            bW = cw[1][sj].x; bE = xinv_lane_down(cw[1][sj].x); cE = xinv_lane_down(cw[2][sj].x);
            // (odd xc: wrapped column 0 can sit in .y; column 1 is then the next lane's .x -- fetched by every lane: a
            //  cross-lane move under a divergent branch would read lanes that are switched off)
            const double bP1 = xinv_lane_down(cw[1][sjp].x);
            if (west) { bP_use = bP1; sM_q = n.m; }
        }
        return xinv_upd_std2d_9_sel(n.c, n.p, n.m, n.w, n.e, n.pe, n.pw, n.me, n.mw, sM_q,
                                    aP, a0, bE, bW, bP_chk, bP_use, bM, cE, c0, f, inr, sc);
    }
};

// SEAM (periodic x, ODD xc; unaligned strips only).  Columns 0 and xc-1 are both even: the coloured ordering runs the
// seam colours right after the colour they split from -- c0, c0' (column xc-1 on even rows), c1, c2, c2' (column xc-1 on
// odd rows), c3 (oracle: seq_colour).  Round 5: the row is laid out as an even ring with a phantom column (xinv_fused.h:
// RING) -- .x slots hold even, .y slots odd virtual columns across any wrap; a row stage of a tile that holds the seam is
// three passes (the .x slots without the seam lanes; the seam lanes alone, their east column taken from the next lane's
// .x -- rows j-1, j, j+1 and, in the standard form, the coefficients -- after which the phantom column mirrors column
// xc-1 again; the .y slots), every other tile's the usual two.  (Round 4 classed the lanes by their wrapped columns and
// ran up to six lane-masked passes: 2000 x 2001 at 0.8 against 1.5e11.)  Both halos hold a column pair more.
template <class M, int K, bool AL, bool EXT, bool SEAM = false>
__global__ __launch_bounds__(256) void k_fused9(FusedArgs a)
{
    static_assert(!SEAM || !AL, "odd xc: strips are never aligned");
    constexpr int NC = M::NC;
    constexpr int HX = 4 * K;           // halo columns: one per colour per sweep
    constexpr int HY = 2 * K;           // halo rows
    constexpr int D = 2 * K + 2;

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
    if constexpr (SEAM) {                                    // (the edge strips' tiles first: xinv_heavy_first, as k_fused2d)
        if (!a.tile_list) {
            const int nh = (a.nstrip == 1 ? 1 : 2) * a.nrb;
            wt = xinv_heavy_first((int)blockIdx.x, NB, nh >> 2) * 4 + wave;
            active = wt < a.nstrip * a.nrb;
            wt = xinv_seam_tile(active ? wt : 0, a.nstrip, a.nrb);
        }
    }
    if (a.tile_list) {                                       // masked-tile skipping, as k_fused2d
        wt = a.tile_list[m * a.ntl + wt];
        active = wt >= 0;
        wt = active ? wt : 0;
    }
    const int rb = wt / a.nstrip, strip = wt - rb * a.nstrip;
    const int64_t xc = a.xc, yc = a.yc;
    const int UW = SEAM ? xinv_ring_uw(xc, HX) : 128 - 2 * HX, HW = SEAM ? xinv_ring_hw(xc, HX, strip) : HX;   // (SEAM: xinv_tiles.h)
    const int64_t xu0 = (int64_t)strip * UW;
    int64_t yu0, yu1;
    if (a.RY > 0) {
        yu0 = (int64_t)rb * a.RY;
        yu1 = (yu0 + a.RY < yc) ? yu0 + a.RY : yc;
    } else {
        yu0 = (((int64_t)rb * yc) / a.nrb) & ~(int64_t)1;
        yu1 = (rb + 1 == a.nrb) ? yc : ((((int64_t)(rb + 1) * yc) / a.nrb) & ~(int64_t)1);
    }
    const double u = a.sc_.undef;
    RingSeam rs = {0ull, false};
    LaneCols lc;
    if constexpr (SEAM) lc = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
    else lc = make_lanecols<AL>(xu0, HX, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - HW + 2 * lane;
    const bool west = (a.per != 0) && (lc.l0 == 0);          // .x sits on wrapped column 0
    const bool seam_x = SEAM && (lc.l0 == xc - 1);           // .x sits on column xc-1 (its .y is the phantom column)

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *cp[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) cp[q] = a.c[q] + m * a.sc[q];

    double acc[K];
    int cnt[K];
#pragma unroll
    for (int s = 0; s < K; s++) { acc[s] = 0.0; cnt[s] = 0; }

    // SEAM: only a tile that holds a seam lane marches with the extra pass (as k_fused2d: with it behind uniform
    // branches inside ONE march every tile of the launch paid for them)
    auto march = [&](auto smtag) {
        constexpr bool SM = decltype(smtag)::value;
        struct Pack { double2 s; double2 c[NC]; };
        auto load = [&](int64_t r) {
            Pack p;
            const int64_t rr = r < 0 ? 0 : (r > yc - 1 ? yc - 1 : r);
            const int64_t off = rr * xc;
            p.s = ld2<AL>(srcS, off, lc);
#pragma unroll
            for (int q = 0; q < NC; q++) p.c[q] = ld2<AL>(cp[q], off, lc);
            return p;
        };

        double2 sw[D];
        double2 cw[NC][D];
#pragma unroll
        for (int t = 0; t < D; t++) {
            sw[t] = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < NC; q++) cw[q][t] = make_double2(0.0, 0.0);
        }

        auto tally = [&](int s, int64_t j, const double2 &t) {      // row j now holds sweep s+1
            if ((j >= yu0) && (j < yu1)) {                          // wave-uniform: an owned row
                const bool cx = lc.use_x & (t.x != u);
                const bool cy = lc.use_y & (t.y != u);
                acc[s] += (cx ? fabs(t.x) : 0.0);
                acc[s] += (cy ? fabs(t.y) : 0.0);
                cnt[s] += (cx ? 1 : 0) + (cy ? 1 : 0);
            }
        };
        auto store = [&](int64_t j, const double2 &t) {
            if (j >= yu0 && j < yu1) {
                if (AL) { if (lc.use_x) *reinterpret_cast<double2 *>(dstS + j * xc + st0) = t; }
                else { if (lc.use_x) dstS[j * xc + st0] = t.x; if (lc.use_y) dstS[j * xc + st0 + 1] = t.y; }
            }
        };

        // the two colours of a row (even row: c0 then c1; odd row: c2 then c3); SEAM: one more pass for the seam lanes
        auto row_stage = [&](int sj, int sjp, int sjm, bool rowok) {
            if constexpr (!SM) {
                double v = M::template upd<0, D>(cw, sw, sj, sjp, sjm, rowok && lc.ok_x, west, a.sc_);
                sw[sj].x = v;
                v = M::template upd<1, D>(cw, sw, sj, sjp, sjm, rowok && lc.ok_y, false, a.sc_);
                sw[sj].y = v;
            } else {
                double v = M::template upd<0, D>(cw, sw, sj, sjp, sjm, rowok && !seam_x, west, a.sc_);     // c0 / c2
                sw[sj].x = v;
                v = M::template upd<0, D, true>(cw, sw, sj, sjp, sjm, rowok && seam_x, false, a.sc_);      // c0' / c2': column xc-1
                sw[sj].x = v;
                sw[sj].y = seam_x ? v : sw[sj].y;                                                          // (the phantom column mirrors it)
                v = M::template upd<1, D>(cw, sw, sj, sjp, sjm, rowok && lc.ok_y, false, a.sc_);           // c1 / c3
                sw[sj].y = v;
            }
        };

        // one double step: even row r-1 sits in slot U, odd row r in slot U+1 (U even)
        auto step2 = [&](int64_t r, auto utag) {
            constexpr int U1 = decltype(utag)::value + 1;            // slot of row r
#define SLOT(w) ((U1 - (w) + 4 * D) % D)                             /* slot of row r - w */
#pragma unroll
            for (int s = 1; s <= K; s++) {
                {   // E_s: even row je = r-2s+1, colours c0 (.x) then c1 (.y)
                    const int64_t je = r - 2 * s + 1;
                    const int sj = SLOT(2 * s - 1), sjp = SLOT(2 * s - 2), sjm = SLOT(2 * s);
                    if (EXT) {       // 'extend' rows take their neighbour's state of sweep s-1
                        if (je == 0) fused_extend_fix(sw[sj], sw[sjp], lc, a.tall, u);
                        if (je == yc - 1) fused_extend_fix(sw[sj], sw[sjm], lc, a.tall, u);
                        if (je == yc - 2) fused_extend_fix(sw[sjp], sw[sj], lc, a.tall, u);
                    }
                    const bool rowok = (je >= 1) && (je <= yc - 2);
                    row_stage(sj, sjp, sjm, rowok);
                    tally(s - 1, je, sw[sj]);
                }
                {   // O_s: odd row jo = r-2s, colours c2 (.x) then c3 (.y)
                    const int64_t jo = r - 2 * s;
                    const int sj = SLOT(2 * s), sjp = SLOT(2 * s - 1), sjm = SLOT(2 * s + 1);
                    const bool rowok = (jo >= 1) && (jo <= yc - 2);
                    row_stage(sj, sjp, sjm, rowok);
                    tally(s - 1, jo, sw[sj]);
                }
            }
            store(r - 2 * K + 1, sw[SLOT(2 * K - 1)]);
            store(r - 2 * K, sw[SLOT(2 * K)]);
#undef SLOT
        };

        // rows enter in (even, odd) pairs; r0 is even.  The first pair's even row has no sweep
        // stage of its own yet -- E_s needs the following odd row, which arrives with it.
        const int64_t r0 = yu0 - HY;
        const int64_t rlast = yu1 - 1 + HY + 1;                      // through an odd row
        Pack p0 = load(r0), p1 = load(r0 + 1);
        for (int64_t rb_ = r0; rb_ <= rlast; rb_ += D) {
            xinv_unroll_steps([&](auto htag) {
                constexpr int U = 2 * decltype(htag)::value;         // slot of the even row
                sw[U] = p0.s; sw[U + 1] = p1.s;
#pragma unroll
                for (int q = 0; q < NC; q++) { cw[q][U] = p0.c[q]; cw[q][U + 1] = p1.c[q]; }
                p0 = load(rb_ + U + 2); p1 = load(rb_ + U + 3);
                step2(rb_ + U + 1, std::integral_constant<int, U>{});
            }, std::make_integer_sequence<int, D / 2>{});
        }
    };
    if (active) {
        bool wraps = false;
        if constexpr (SEAM) wraps = rs.any;
        if (wraps) march(std::integral_constant<bool, SEAM>{}); else march(std::false_type{});
    }

    if (a.no_ctl) return;
    xinv_norm_tail<K>(a, acc, cnt, wave, lane, NB, T, tag, ctl, m);
}
