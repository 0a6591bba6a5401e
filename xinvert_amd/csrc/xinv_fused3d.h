// xinv_fused3d.h -- streaming fused red-black SOR sweep for the 3-D standard form (gfx950).
//
// numbas.invert_standard_3D (reference numbas.py:15-212): 7-point, colour (k+j+i)&1.
//
// Decomposition.  A workgroup of NW wavefronts owns a cross-section of NW consecutive j rows x
// one 128-column strip and marches it through all k planes.  Wave w holds ITS row of the last
// four planes in registers (two adjacent columns per lane, as in the 2-D kernel).  With `r` the
// plane just loaded, the red points of plane r-1 and then the black points of plane r-2 are
// updated; plane r-2 leaves with a complete sweep.  Neighbours:
//   k +/- 1  same lane, own registers (other planes of the window);
//   i +/- 1  DPP wave shift inside the row;
//   j +/- 1  the adjacent waves of the workgroup, through LDS: after loading / red-updating its
//            row a wave publishes the one component (8 B per lane) its neighbours will need in
//            the NEXT step, double-buffered by step parity -> ONE workgroup barrier per plane.
// The two edge rows on each side of the cross-section and two columns on each side of the strip
// are halo (recomputed by the neighbouring tile, never exchanged); S ping-pongs between buffers.
// HBM traffic per sweep: one read of S, A, B, C, F (+ halo re-reads, + B[j+1] rows that hit
// cache) and one write of S: the algorithmic 48 B per point instead of the colour-pass path's
// two passes.  Coefficient rows that are constant along x (every lat-lon omega coefficient:
// apps.py:2033-2035) are read as one scalar per row and the divide is hoisted per row (UM).
//
// 'extend' (BCy): rows 0 / yc-1 of planes 1..zc-2 take rows 1 / yc-2 of the same plane at the
// start of the sweep (numbas.py:87-115); folded into the load of those rows.
// Norm and stopping rule: as the 2-D kernel (xinv_norm_finalize: tagged partials, one reducer).
#pragma once
#include "xinv_fused.h"

struct Fused3Args {
    const double *src;
    double *dst;
    const double *c[4];        // A, B, C, F
    int64_t sS, sc[4];
    int64_t zc, yc, xc;
    int per;
    int nstrip, njb;           // x strips, j blocks
    int nkc, KC;               // k chunks of KC planes (KC % 4 == 0); nkc == 1: the whole column
    int force, no_ctl;
    int64_t member0;
    XinvScal sc_;
    XinvCtl *ctl;
    XinvStop stop;
    unsigned long long *psum;  // [nbatch][NB]
    const double *rowf;        // k_pipe3d: per-(plane, row) records [nb][zc][yc][8] (xinv_pipe3d.h)
    int64_t srowf;             // member stride of the table in doubles (0: one table for the batch)
    // k_pipe3d: a FLAT grid over the launch's members.  With g the tile's index in the launch (member-major; a member has
    // nstrip * njb tiles), workgroups L < nfull march tile g = L through the whole column; the tiles behind them are cut
    // into the nkc chunks: workgroup nfull + q marches chunk q % nkc of tile nfull + q / nkc (xinv_pipe3d.h: the tail of
    // a launch whose tile count is not a multiple of the compute units).  nfull == 0: every tile is cut.
    int64_t nfull;
    int joff;                  // k_pipe3d: rows the row blocks are shifted up by (0, or 2 with BCy = 'extend' where that puts rows
                               // yc-2 / yc-1 into one wavefront: xinv_tiles.h, xinv_p3_extend_joff); block jb owns rows [jb RJ - joff, ..)
};

// 7-point update with the mask folded into a select (numbas.py:146-169).
__device__ __forceinline__ double xinv_upd_std3d_sel(
    double sC, double sKP, double sKM, double sJP, double sJM, double sE, double sW,
    double aP, double a0, double bP, double b0, double cE, double c0, double f, bool inr,
    const XinvScal &sc)
{
    const double u = sc.undef;
    const bool cond = inr && (f != u) && (aP != u) && (a0 != u) && (bP != u) && (b0 != u) &&
                      (cE != u) && (c0 != u);
    double temp = (
        (
            aP * (sKP - sC) -
            a0 * (sC - sKM)
        ) * sc.ratio2Sqr + (
            bP * (sJP - sC) -
            b0 * (sC - sJM)
        ) * sc.ratio1Sqr + (
            cE * (sE - sC) -
            c0 * (sC - sW)
        )
    ) - f * sc.delxSqr;
    temp *= sc.optArg / ((aP + a0) * sc.ratio2Sqr +
                         (bP + b0) * sc.ratio1Sqr +
                         (cE + c0));
    return cond ? sC + temp : sC;
}

template <bool UNI> struct Coef3Pack;
template <> struct Coef3Pack<false> { double2 A, B, B1, C; };
template <> struct Coef3Pack<true>  { double A, B, B1, C; };

// FMA: the opt-in contracted arithmetic of XINV_FLAG_FMA (x-uniform coefficients only; oracle: XO_FMA, bit for bit).
// SEAM: periodic x with ODD xc (unaligned strips only): column xc-1 is updated inside the half-sweep of its own colour right
// after column 0 (oracle: seq_colour; 3-D colours (k+j+i)&1, seam colours (k+j)&1 on column xc-1).  Round 5: the row as an
// even ring with a phantom column (xinv_fused.h: RING) -- the half-sweeps that update the .x slots leave the seam lanes out
// of their pass and run one more for them alone (east operand and, with full coefficient arrays, east coefficient from the
// next lane's .x), then the phantom column mirrors column xc-1 again; only the cross-sections that hold a seam lane march
// with that code.  (Round 4: lane classes, up to three passes, both components of a row exchanged through LDS.)
template <int NW, bool AL, bool UNI, bool EXT, bool FMA = false, bool SEAM = false>
__global__ __launch_bounds__(NW * 64) void k_fused3d(Fused3Args a)
{
    static_assert(!FMA || UNI, "contracted arithmetic: x-uniform coefficients only");
    static_assert(!SEAM || !AL, "odd xc: strips are never aligned");
    constexpr int H = 2, D = 4, RJ = NW - 4;

    const int64_t m = a.member0 + blockIdx.y;
    XinvCtl *ctl = a.ctl + m;
    if (!a.force && xinv_ctl_done(ctl)) return;
    const unsigned tag = xinv_ctl_seq(ctl);

    const int NB = a.nstrip * a.njb * a.nkc;
    int T;
    {
        const int L = blockIdx.x, q = NB >> 3, rem = NB & 7, xcd = L & 7, idx = L >> 3;
        T = xcd * q + (xcd < rem ? xcd : rem) + idx;
    }
    // tile = (k chunk, j block, x strip).  A chunk owns planes [k0, k1); two planes on each side
    // are halo (loaded, recomputed, not stored), exactly like the rows of the 2-D kernel: tall
    // volumes and small batches then give every CU a workgroup.
    const int kc = T / (a.nstrip * a.njb), Tj = T - kc * (a.nstrip * a.njb);
    const int jb = Tj / a.nstrip, st = Tj - jb * a.nstrip;
    const int64_t k0 = (int64_t)kc * a.KC;
    const int64_t k1 = (kc + 1 == a.nkc) ? a.zc : k0 + a.KC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t xc = a.xc, yc = a.yc, zc = a.zc;
    const int UW = SEAM ? xinv_ring_uw(xc, H) : 128 - 2 * H, HW = SEAM ? xinv_ring_hw(xc, H, st) : H;   // (SEAM: xinv_tiles.h)
    const int64_t xu0 = (int64_t)st * UW;
    const double u = a.sc_.undef;
    RingSeam rs = {0ull, false};
    LaneCols lc;
    if constexpr (SEAM) lc = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
    else lc = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - HW + 2 * lane;
    const bool seam_x = SEAM && (lc.l0 == xc - 1);           // .x holds column xc-1 (its .y is the phantom column)

    const int64_t j = (int64_t)jb * RJ - 2 + wave;             // this wave's row (may be outside)
    const int64_t jr = j < 0 ? 0 : (j > yc - 1 ? yc - 1 : j);
    const int64_t jr1 = (j + 1) < 0 ? 0 : ((j + 1) > yc - 1 ? yc - 1 : (j + 1));
    const bool row_upd = (j >= 1) && (j <= yc - 2);
    const bool row_use = (wave >= 2) && (wave < NW - 2) && (j < yc);
    const int wm = wave > 0 ? wave - 1 : 0, wp = wave < NW - 1 ? wave + 1 : NW - 1;
    const int64_t fixrow = (j == 0) ? 1 : ((j == yc - 1) ? yc - 2 : -1);   // 'extend' source row

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *pA = a.c[0] + m * a.sc[0], *pB = a.c[1] + m * a.sc[1];
    const double *pC = a.c[2] + m * a.sc[2], *pF = a.c[3] + m * a.sc[3];

    __shared__ double xch[2][2][NW][64];       // [step parity][as-loaded | red-updated][wave][lane]

    struct Pack { double2 s, f, sfix; Coef3Pack<UNI> c; };
    auto load = [&](int64_t r) {
        Pack p;
        const int64_t pr = r > zc - 1 ? zc - 1 : (r < 0 ? 0 : r);
        const int64_t off = (pr * yc + jr) * xc, off1 = (pr * yc + jr1) * xc;
        p.s = ld2<AL>(srcS, off, lc);
        p.f = ld2<AL>(pF, off, lc);
        if constexpr (UNI) {
            p.c.A = pA[off]; p.c.B = pB[off]; p.c.B1 = pB[off1]; p.c.C = pC[off];
        } else {
            p.c.A = ld2<AL>(pA, off, lc); p.c.B = ld2<AL>(pB, off, lc);
            p.c.B1 = ld2<AL>(pB, off1, lc); p.c.C = ld2<AL>(pC, off, lc);
        }
        p.sfix = p.s;
        if (EXT) {
            if (fixrow >= 0 && pr >= 1 && pr <= zc - 2)
                p.sfix = ld2<AL>(srcS, (pr * yc + fixrow) * xc, lc);
        }
        return p;
    };

    double acc = 0.0;
    int cnt = 0;

    double2 sw[D], fw[D];
    Coef3Pack<UNI> cw[D];
    double rq[D];
    bool rok[D];
#pragma unroll
    for (int t = 0; t < D; t++) {
        sw[t] = make_double2(0.0, 0.0); fw[t] = sw[t]; rq[t] = 0.0; rok[t] = false;
        if constexpr (UNI) { cw[t].A = cw[t].B = cw[t].B1 = cw[t].C = 0.0; }
        else { cw[t].A = cw[t].B = cw[t].B1 = cw[t].C = make_double2(0.0, 0.0); }
    }

    // coefficient access: vector rows pick the lane component, uniform rows are scalars
    auto gA = [&](int slot, auto xt) { constexpr int X = decltype(xt)::value;
        if constexpr (UNI) return cw[slot].A; else return comp<X>(cw[slot].A); };
    auto gB = [&](int slot, auto xt) { constexpr int X = decltype(xt)::value;
        if constexpr (UNI) return cw[slot].B; else return comp<X>(cw[slot].B); };
    auto gB1 = [&](int slot, auto xt) { constexpr int X = decltype(xt)::value;
        if constexpr (UNI) return cw[slot].B1; else return comp<X>(cw[slot].B1); };
    auto gC = [&](int slot, auto xt) { constexpr int X = decltype(xt)::value;
        if constexpr (UNI) return cw[slot].C; else return comp<X>(cw[slot].C); };
    auto gCE = [&](int slot, auto xt, auto fixt) { constexpr int X = decltype(xt)::value;
        if constexpr (UNI) return cw[slot].C;
        else { if (X == 0 && !decltype(fixt)::value) return cw[slot].C.y; else return xinv_lane_down(cw[slot].C.x); } };

    // one point update of component X on the plane held in slot `sk` (k+1 in `skp`, k-1 in `skm`)
    // (SM marches: the pass of the .x slots leaves the seam lanes out; fixt: the seam lanes' own pass, east = the next lane's .x)
    auto update = [&](int sk, int skp, int skm, int64_t kk, double jP, double jM, auto xt, auto smt, auto fixt) {
        constexpr int X = decltype(xt)::value;
        constexpr bool SM = decltype(smt)::value, FIX = decltype(fixt)::value;
        static_assert(!FIX || (SM && X == 0), "column xc-1 sits in an .x slot");
        const bool okc = FIX ? seam_x : ((X ? lc.ok_y : lc.ok_x) && !(SM && X == 0 && seam_x));
        const bool inr = okc && row_upd && (kk >= 1) && (kk <= zc - 2);
        double w, e;
        row_neighbours<X>(sw[sk], w, e);
        if constexpr (FIX) e = xinv_lane_down(sw[sk].x);
        const double sC = comp<X>(sw[sk]), sKP = comp<X>(sw[skp]), sKM = comp<X>(sw[skm]);
        const double aP = gA(skp, xt), a0 = gA(sk, xt), bP = gB1(sk, xt), b0 = gB(sk, xt);
        const double cE = gCE(sk, xt, fixt), c0 = gC(sk, xt), f = comp<X>(fw[sk]);
        double v;
        if constexpr (FMA) {
            const bool cond = inr && rok[sk] && (f != u);
            const double ya = __builtin_fma(aP, sKP - sC, -(a0 * (sC - sKM)));
            const double yb = __builtin_fma(bP, jP - sC, -(b0 * (sC - jM)));
            const double yc_ = __builtin_fma(cE, e - sC, -(c0 * (sC - w)));
            double t = __builtin_fma(ya, a.sc_.ratio2Sqr, __builtin_fma(yb, a.sc_.ratio1Sqr, yc_));
            t = __builtin_fma(-f, a.sc_.delxSqr, t);
            v = cond ? __builtin_fma(t, rq[sk], sC) : sC;
        } else if constexpr (UNI) {
            const bool cond = inr && rok[sk] && (f != u);
            double temp = (
                (
                    aP * (sKP - sC) -
                    a0 * (sC - sKM)
                ) * a.sc_.ratio2Sqr + (
                    bP * (jP - sC) -
                    b0 * (sC - jM)
                ) * a.sc_.ratio1Sqr + (
                    cE * (e - sC) -
                    c0 * (sC - w)
                )
            ) - f * a.sc_.delxSqr;
            temp *= rq[sk];
            v = cond ? sC + temp : sC;
        } else {
            v = xinv_upd_std3d_sel(sC, sKP, sKM, jP, jM, e, w, aP, a0, bP, b0, cE, c0, f, inr, a.sc_);
        }
        setc<X>(sw[sk], v);
        return v;
    };
    // a half-sweep on the plane in slot sk; SM marches: column xc-1 behind the pass of the .x slots, the phantom column after it
    auto half = [&](int sk, int skp, int skm, int64_t kk, double jP, double jM, auto xt, auto smt) {
        constexpr int X = decltype(xt)::value;
        constexpr bool SM = decltype(smt)::value;
        double v = update(sk, skp, skm, kk, jP, jM, xt, smt, std::false_type{});
        if constexpr (SM && X == 0) {
            v = update(sk, skp, skm, kk, jP, jM, xt, smt, std::true_type{});
            sw[sk].y = seam_x ? v : sw[sk].y;
        }
        return v;
    };

    // SEAM: only a cross-section that holds a seam lane marches with the extra pass (SM); every other workgroup of the launch
    // runs the plain march (profiles/r05_seam_rates.txt: with extra passes behind uniform branches in ONE march every
    // tile lost its instruction interleaving)
    // one pipeline step: plane r (= rbase + U) enters slot U; JP = parity of this wave's row
    auto step = [&](int64_t r, const Pack &p, auto utag, auto jtag, auto smt) {
        constexpr int U = decltype(utag)::value;
        constexpr int JP = decltype(jtag)::value;
        constexpr int X = (1 + (U & 1) + JP) & 1;              // component touched in this step
        constexpr int S1 = (U + 3) % D, S2 = (U + 2) % D, S3 = (U + 1) % D;
        using XT = std::integral_constant<int, X>;
        const int bw = U & 1, br = (U + 1) & 1;

        double2 sn = p.s;
        if (EXT) {                                             // numbas.py:87-115
            if (fixrow >= 0 && r >= 1 && r <= zc - 2) fused_extend_fix(sn, p.sfix, lc, false, u);
        }
        sw[U] = sn; fw[U] = p.f; cw[U] = p.c;
        if constexpr (UNI) {                                   // plane r-1: A[r], A[r-1] known now
            const double aP = cw[U].A, a0 = cw[S1].A, bP = cw[S1].B1, b0 = cw[S1].B, c = cw[S1].C;
            rq[S1] = a.sc_.optArg / ((aP + a0) * a.sc_.ratio2Sqr +
                                     (bP + b0) * a.sc_.ratio1Sqr +
                                     (c + c));
            rok[S1] = (aP != u) && (a0 != u) && (bP != u) && (b0 != u) && (c != u);
        }
        xch[bw][0][wave][lane] = comp<X>(sw[U]);               // as loaded: neighbours' next red

        {   // red half-sweep on plane r-1
            const double jM = xch[br][0][wm][lane], jP = xch[br][0][wp][lane];
            const double v = half(S1, U, S2, r - 1, jP, jM, XT{}, smt);
            xch[bw][1][wave][lane] = v;                        // red-updated: neighbours' next black
        }
        {   // black half-sweep on plane r-2
            const int64_t kk = r - 2;
            const double jM = xch[br][1][wm][lane], jP = xch[br][1][wp][lane];
            half(S2, S1, S3, kk, jP, jM, XT{}, smt);
            const bool pin = row_use && (kk >= k0) && (kk < k1);
            const double2 t = sw[S2];
            if (pin) {                                         // wave-uniform: an owned row of an owned plane
                const bool cx = lc.use_x & (t.x != u);
                const bool cy = lc.use_y & (t.y != u);
                acc += (cx ? fabs(t.x) : 0.0);
                acc += (cy ? fabs(t.y) : 0.0);
                cnt += (cx ? 1 : 0) + (cy ? 1 : 0);
                double *d = dstS + (kk * yc + j) * xc + st0;
                if (AL) { if (lc.use_x) *reinterpret_cast<double2 *>(d) = t; }
                else { if (lc.use_x) d[0] = t.x; if (lc.use_y) d[1] = t.y; }
            }
        }
        __syncthreads();
    };

    auto march = [&](auto jtag, auto smt) {
        // start a multiple of D planes below k0 - 2 (slot indices are compile-time), run until the
        // black half-sweep of plane k1 - 1 (step k1 + 1)
        const int64_t rstart = (k0 >= D) ? k0 - D : 0;
#ifndef XINV_3D_PF
#define XINV_3D_PF 2              /* planes in flight per wavefront (2 or 4: divides the window depth) */
#endif
        constexpr int PF = XINV_3D_PF;
        Pack pf[PF];
#pragma unroll
        for (int t = 0; t < PF; t++) pf[t] = load(rstart + t);
        const int64_t rlast = k1 - 1 + 2;
        for (int64_t rb_ = rstart; rb_ <= rlast; rb_ += D) {
            xinv_unroll_steps([&](auto utag) {
                constexpr int U = decltype(utag)::value;
                step(rb_ + U, pf[U % PF], utag, jtag, smt);
                pf[U % PF] = load(rb_ + U + PF);
            }, std::make_integer_sequence<int, D>{});
        }
    };
    bool wraps = false;
    if constexpr (SEAM) wraps = rs.any;                      // (the same for every wavefront of the workgroup: one strip)
    if (wraps) {
        if (j & 1) march(std::integral_constant<int, 1>{}, std::integral_constant<bool, SEAM>{});
        else       march(std::integral_constant<int, 0>{}, std::integral_constant<bool, SEAM>{});
    } else {
        if (j & 1) march(std::integral_constant<int, 1>{}, std::false_type{});
        else       march(std::integral_constant<int, 0>{}, std::false_type{});
    }

    if (a.no_ctl) return;

    // ---- norm partials: sequence-tagged words, the last-dispatched workgroup finalises
    //      (xinv_norm_finalize, xinv_fused.h)
    const double acc1[1] = {acc};
    const int cnt1[1] = {cnt};
    xinv_norm_finalize<1, NW>(acc1, cnt1, wave, lane, NB, T, tag, a.psum + (size_t)m * NB * XINV_PW,
                              ctl, a.stop);
}
