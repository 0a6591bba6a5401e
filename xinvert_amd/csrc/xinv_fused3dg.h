// xinv_fused3dg.h -- streaming fused red-black sweep for the GENERAL 3-D form (gfx950).
//
// numbas.invert_general_3D (reference numbas.py:745-984; apps.invert_3DOcean): 7-point,
//   A S_zz + B S_yy + C S_xx + D S_z + E S_y + F S_x + G S = H,  colour (k+j+i)&1.
// Same decomposition as k_fused3d (xinv_fused3d.h): a workgroup of NW wavefronts owns NW
// consecutive j rows x one 128-column strip x one k chunk, marches it plane by plane with a
// register window of four planes, j-neighbours through LDS (double-buffered by step parity, one
// barrier per plane), i-neighbours by DPP, two recomputed halo rows / columns / planes a side.
// This variant requires every coefficient array A..G to be constant along x (true for every
// 3DOcean coefficient, apps.py:2074-2081: functions of level and latitude): they are read as
// one scalar per (plane, row) and the divide is hoisted per row; only S and the forcing H stream.
// Anything else runs the colour-pass kernel (k_colour_gen3d).
// The reference's west-periodic branch never tests H against undef (numbas.py:849-852): kept.
#pragma once
#include "xinv_fused3d.h"

struct Fused3GArgs {
    const double *src;
    double *dst;
    const double *c[8];        // A..G (x-uniform), H
    int64_t sS, sc[8];
    int64_t zc, yc, xc;
    int per;
    int nstrip, njb;
    int nkc, KC;
    int force, no_ctl;
    int64_t member0;
    XinvScal sc_;
    XinvCtl *ctl;
    XinvStop stop;
    unsigned long long *psum;  // [nbatch][NB]
};

// SEAM: the odd-xc periodic seam inside the kernel, exactly as in k_fused3d (xinv_fused3d.h): the even ring with a phantom
// column, one more pass for the seam lanes in the half-sweeps of their colour, only in cross-sections that hold them.
template <int NW, bool AL, bool EXT, bool SEAM = false>
__global__ __launch_bounds__(NW * 64) void k_fused3dg(Fused3GArgs a)
{
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
    const int kc = T / (a.nstrip * a.njb), Tj = T - kc * (a.nstrip * a.njb);
    const int jb = Tj / a.nstrip, st = Tj - jb * a.nstrip;
    const int64_t k0 = (int64_t)kc * a.KC;
    const int64_t k1 = (kc + 1 == a.nkc) ? a.zc : k0 + a.KC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t xc = a.xc, yc = a.yc, zc = a.zc;
    const int UW = SEAM ? xinv_ring_uw(xc, H) : 128 - 2 * H, HW = SEAM ? xinv_ring_hw(xc, H, st) : H;   // (SEAM: xinv_tiles.h)
    const int64_t xu0 = (int64_t)st * UW;
    const double u = a.sc_.undef;
    const bool tall = yc > xc;             // the general kernel's pre-pass loops over range(1, yc-1)
    RingSeam rs = {0ull, false};
    LaneCols lc;
    if constexpr (SEAM) lc = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
    else lc = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - HW + 2 * lane;
    const bool seam_x = SEAM && (lc.l0 == xc - 1);           // .x holds column xc-1 (its .y is the phantom column)
    // the reference's i == 0 periodic branch does not test H (numbas.py:849-852)
    const bool noH_x = (a.per != 0) && (lc.l0 == 0), noH_y = (a.per != 0) && (lc.l1 == 0);

    const int64_t j = (int64_t)jb * RJ - 2 + wave;
    const int64_t jr = j < 0 ? 0 : (j > yc - 1 ? yc - 1 : j);
    const bool row_upd = (j >= 1) && (j <= yc - 2);
    const bool row_use = (wave >= 2) && (wave < NW - 2) && (j < yc);
    const int wm = wave > 0 ? wave - 1 : 0, wp = wave < NW - 1 ? wave + 1 : NW - 1;
    const int64_t fixrow = (j == 0) ? 1 : ((j == yc - 1) ? yc - 2 : -1);

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *pc[7];
#pragma unroll
    for (int q = 0; q < 7; q++) pc[q] = a.c[q] + m * a.sc[q];
    const double *pH = a.c[7] + m * a.sc[7];

    __shared__ double xch[2][2][NW][64];

    struct Pack { double2 s, h, sfix; double c[7]; };
    auto load = [&](int64_t r) {
        Pack p;
        const int64_t pr = r > zc - 1 ? zc - 1 : (r < 0 ? 0 : r);
        const int64_t off = (pr * yc + jr) * xc;
        p.s = ld2<AL>(srcS, off, lc);
        p.h = ld2<AL>(pH, off, lc);
#pragma unroll
        for (int q = 0; q < 7; q++) p.c[q] = pc[q][off];
        p.sfix = p.s;
        if (EXT) {
            if (fixrow >= 0 && pr >= 1 && pr <= zc - 2)
                p.sfix = ld2<AL>(srcS, (pr * yc + fixrow) * xc, lc);
        }
        return p;
    };

    double acc = 0.0;
    int cnt = 0;

    double2 sw[D], hw[D];
    double cw[D][7];
    double rq[D];
    bool rok[D];
#pragma unroll
    for (int t = 0; t < D; t++) {
        sw[t] = make_double2(0.0, 0.0); hw[t] = sw[t]; rq[t] = 0.0; rok[t] = false;
#pragma unroll
        for (int q = 0; q < 7; q++) cw[t][q] = 0.0;
    }

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
        const double cA = cw[sk][0], cB = cw[sk][1], cC = cw[sk][2], cD = cw[sk][3];
        const double cE = cw[sk][4], cF = cw[sk][5], cG = cw[sk][6], h = comp<X>(hw[sk]);
        const bool noH = X ? noH_y : noH_x;
        const bool cond = inr && rok[sk] && (noH || (h != u));
        double temp = (
            cA * (
                (sKP - sC)-(sC - sKM)
            ) * a.sc_.ratio2Sqr +
            cB * (
                (jP - sC)-(sC - jM)
            ) * a.sc_.ratio1Sqr +
            cC * (
                (e - sC)-(sC - w)
            ) + (
            cD * (
                (sKP - sKM)
            ) * a.sc_.ratio2 +
            cE * (
                (jP - jM)
            ) * a.sc_.ratio1 +
            cF * (
                (e - w)
            )) * a.sc_.delx / 2.0 + (
            cG * sC - h) * a.sc_.delxSqr
        );
        temp *= rq[sk];
        const double v = cond ? sC + temp : sC;
        setc<X>(sw[sk], v);
        return v;
    };
    auto half = [&](int sk, int skp, int skm, int64_t kk, double jP, double jM, auto xt, auto smt) {
        constexpr int X = decltype(xt)::value;
        constexpr bool SM = decltype(smt)::value;
        double v = update(sk, skp, skm, kk, jP, jM, xt, smt, std::false_type{});
        if constexpr (SM && X == 0) {                        // column xc-1 behind column 0; the phantom column mirrors it again
            v = update(sk, skp, skm, kk, jP, jM, xt, smt, std::true_type{});
            sw[sk].y = seam_x ? v : sw[sk].y;
        }
        return v;
    };

    auto step = [&](int64_t r, const Pack &p, auto utag, auto jtag, auto smt) {
        constexpr int U = decltype(utag)::value;
        constexpr int JP = decltype(jtag)::value;
        constexpr int X = (1 + (U & 1) + JP) & 1;
        constexpr int S1 = (U + 3) % D, S2 = (U + 2) % D, S3 = (U + 1) % D;
        using XT = std::integral_constant<int, X>;
        const int bw = U & 1, br = (U + 1) & 1;

        double2 sn = p.s;
        if (EXT) {
            if (fixrow >= 0 && r >= 1 && r <= zc - 2) fused_extend_fix(sn, p.sfix, lc, tall, u);
        }
        sw[U] = sn; hw[U] = p.h;
#pragma unroll
        for (int q = 0; q < 7; q++) cw[U][q] = p.c[q];
        {   // coefficients are given at the point: the hoisted divide of plane r is known at once
            const double cA = p.c[0], cB = p.c[1], cC = p.c[2], cG = p.c[6];
            rq[U] = a.sc_.optArg / ((
                cA*a.sc_.ratio2Sqr + cB*a.sc_.ratio1Sqr + cC
            ) * 2.0 - cG*a.sc_.delxSqr);
            rok[U] = (cG != u) && (cA != u) && (cB != u) && (cC != u) && (p.c[3] != u) &&
                     (p.c[4] != u) && (p.c[5] != u);
        }
        xch[bw][0][wave][lane] = comp<X>(sw[U]);

        {   // red half-sweep on plane r-1
            const double jM = xch[br][0][wm][lane], jP = xch[br][0][wp][lane];
            const double v = half(S1, U, S2, r - 1, jP, jM, XT{}, smt);
            xch[bw][1][wave][lane] = v;
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
        const int64_t rstart = (k0 >= D) ? k0 - D : 0;
        Pack p0 = load(rstart), p1 = load(rstart + 1);
        const int64_t rlast = k1 - 1 + 2;
        for (int64_t rb_ = rstart; rb_ <= rlast; rb_ += D) {
            xinv_unroll_steps([&](auto utag) {
                constexpr int U = decltype(utag)::value;
                if (U & 1) { step(rb_ + U, p1, utag, jtag, smt); p1 = load(rb_ + U + 2); }
                else       { step(rb_ + U, p0, utag, jtag, smt); p0 = load(rb_ + U + 2); }
            }, std::make_integer_sequence<int, D>{});
        }
    };
    // (only a cross-section that holds a seam lane marches with the extra pass: the same for every wavefront of the workgroup)
    if (SEAM && rs.any) {
        if (j & 1) march(std::integral_constant<int, 1>{}, std::integral_constant<bool, SEAM>{});
        else       march(std::integral_constant<int, 0>{}, std::integral_constant<bool, SEAM>{});
    } else {
        if (j & 1) march(std::integral_constant<int, 1>{}, std::false_type{});
        else       march(std::integral_constant<int, 0>{}, std::false_type{});
    }

    if (a.no_ctl) return;

    // norm partials: sequence-tagged words, the last-dispatched workgroup finalises
    // (xinv_norm_finalize, xinv_fused.h)
    const double acc1[1] = {acc};
    const int cnt1[1] = {cnt};
    xinv_norm_finalize<1, NW>(acc1, cnt1, wave, lane, NB, T, tag, a.psum + (size_t)m * NB * XINV_PW,
                              ctl, a.stop);
}
