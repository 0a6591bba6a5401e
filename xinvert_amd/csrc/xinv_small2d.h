// xinv_small2d.h -- register-resident solver for SMALL 2-D slices (gfx950).
//
// The regime of the reference's own cases: 73 x 144 (2.5 degree) fields, a few hundred time
// slices (reference tests/test_GillMatsuno.py, test_Poisson.py on Data/Helmholtz_atmos.nc).  There
// the streaming kernels are bound by launch latency and halo re-reads.  Here ONE workgroup of eight
// wavefronts keeps a whole slice in registers for the whole solve:
//
//   - wavefront w owns the band of rows [w*RW, (w+1)*RW); a lane holds two adjacent columns
//     (one of each colour) of every row of the band, in NSEG segments of 128 columns -- S lives in
//     VGPRs (RW*NSEG*4 of them), the update predicate as one bit per point, the forcing (already
//     multiplied by delxSqr where the form allows) in LDS, one 8-byte read per point update;
//   - per-row coefficients (the form requires every coefficient array to be constant along x:
//     lat-lon Poisson, Gill-Matsuno -- detected on the device, as for the streaming kernels) and
//     the hoisted relaxation factor optArg/denominator sit in an LDS table;
//   - a half-sweep updates one colour of every row of the band from registers, the east/west
//     neighbour by a DPP wave shift (segment seams and the periodic wrap by v_readlane +
//     a one-lane select), the rows above and below the band from LDS, where every wavefront publishes
//     the component it has just updated in its first and last row: ONE workgroup barrier per
//     half-sweep, no HBM traffic between sweeps;
//   - mean|S| (numbas.py:1710-1728) is reduced wave -> LDS in a fixed order and every wavefront
//     applies the reference's stop rule (numbas.py:401-414) to the same total: the whole solve --
//     every sweep, the norm, the stop decision -- is ONE launch, one slice per CU.
//
// Same red-black ordering, same point arithmetic and association as k_fused2d (bitwise equal to it
// and to the oracle's coloured ordering).  RW is even, so the colour component of a row is a
// compile-time property of its slot; rows >= yc and columns >= xc are phantoms (S = 0, predicate 0).
#pragma once
#include "xinv_device.h"
#include "xinv_fused.h"

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- every register
// array below is indexed with constants only, so that it lives in VGPRs
template <int N, class F> __device__ __forceinline__ void small_for(F &&f)
{
    xinv_unroll_steps(f, std::make_integer_sequence<int, N>{});
}
#define SFOR(var, N) small_for<N>([&](auto var##_t) { constexpr int var = decltype(var##_t)::value;
#define SEND });

#ifndef XINV_SMALL_SCHED_BARRIER
#define XINV_SMALL_SCHED_BARRIER 1
#endif


struct SmallArgs {
    double *S; int64_t sS;
    const double *c[6];        // std: A, C, F ; gen: A, C, D, E, F, G   (B == 0; all but the forcing x-uniform)
    int64_t sc[6];
    int64_t yc, xc;
    int per, ext, tall;
    int64_t member0;
    XinvScal sc_;
    XinvStop stop;
    XinvCtl *ctl;
};

__device__ __forceinline__ double small_readlane(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// v with the lanes of `sel` replaced by the (wave-uniform) value s   (two v_cndmask_b32; this
// compiler has no v_writelane builtin)
__device__ __forceinline__ double small_setlane(double v, double s, bool sel)
{
    return sel ? s : v;
}

// ---- models: row tables and the point update (expressions of FusedStd2D / FusedGen2D with the
// x-uniform hoist; same operands, same association) ------------------------------------------
struct SmallStd {                   // numbas.invert_standard_2D, B == 0, A and C constant along x
    static constexpr int NR = 4;    // row table: aP = A[j+1], a0 = A[j], c = C[j], rq
    static constexpr int FQ = 2;    // index of the forcing in SmallArgs::c
    static __device__ __forceinline__ bool row_setup(const SmallArgs &a, int64_t m, int64_t j, double *t)
    {
        const double u = a.sc_.undef;
        const int64_t jp = (j + 1 < a.yc) ? j + 1 : j;
        const double aP = a.c[0][m * a.sc[0] + jp * a.xc], a0 = a.c[0][m * a.sc[0] + j * a.xc];
        const double c = a.c[1][m * a.sc[1] + j * a.xc];
        t[0] = aP; t[1] = a0; t[2] = c;
        t[3] = a.sc_.optArg / ((aP + a0) * a.sc_.ratioSqr + (c + c));
        return (aP != u) && (a0 != u) && (c != u);
    }
    static __device__ __forceinline__ double prep(double f, const XinvScal &sc) { return f * sc.delxSqr; }
    static __device__ __forceinline__ double upd(const double *t, double fd, double sC, double sP,
                                                 double sM, double sW, double sE, const XinvScal &sc)
    {
        const double aP = t[0], a0 = t[1], cE = t[2], c0 = t[2];
        double temp = (
            (
                aP * (sP - sC) -
                a0 * (sC - sM)
            ) * sc.ratioSqr + (
                cE * (sE - sC) -
                c0 * (sC - sW)
            )
        ) - fd;
        temp *= t[3];
        return sC + temp;
    }
};

struct SmallGen {                   // numbas.invert_general_2D, B == 0, A C D E F constant along x
    static constexpr int NR = 6;    // row table: A, C, D, E, F, rq
    static constexpr int FQ = 5;
    static __device__ __forceinline__ bool row_setup(const SmallArgs &a, int64_t m, int64_t j, double *t)
    {
        const double u = a.sc_.undef;
        bool ok = true;
        for (int q = 0; q < 5; q++) { t[q] = a.c[q][m * a.sc[q] + j * a.xc]; ok = ok && (t[q] != u); }
        t[5] = a.sc_.optArg / ((t[0] * a.sc_.ratioSqr + t[1]) * 2.0
                               - t[4] * a.sc_.delxSqr);
        return ok;
    }
    static __device__ __forceinline__ double prep(double f, const XinvScal &) { return f; }
    static __device__ __forceinline__ double upd(const double *t, double G, double sC, double sP,
                                                 double sM, double sW, double sE, const XinvScal &sc)
    {
        const double A = t[0], C = t[1], Dd = t[2], E = t[3], F = t[4];
        double temp = (
            A * (
                (sP - sC) - (sC - sM)
            ) * sc.ratioSqr +
            C * (
                (sE - sC) - (sC - sW)
            ) + (
            Dd * (
                (sP - sM)
            ) * sc.ratio +
            E * (
                (sE - sW)
            )) * sc.delx / 2.0 + (
            F * sC - G) * sc.delxSqr
        );
        temp *= t[5];
        return sC + temp;
    }
};

template <class M, int NW, int RW, int NSEG, bool EXT>
__global__ __launch_bounds__(NW * 64, NW / 4) void k_small2d(SmallArgs a)
{
    constexpr int NR = M::NR;
    static_assert((RW & 1) == 0, "rows per wavefront must be even: the colour component of a slot is a compile-time constant");
    static_assert(2 * RW * NSEG <= 64, "one predicate bit per point");
    __shared__ double rowc[NW * RW][NR];
    __shared__ int rowok[NW * RW];
    __shared__ double edge[NW + 2][2][NSEG][2][64];  // [wave + 1][first / last row][segment][component][lane]; [0] and [NW+1] stay zero
    __shared__ double nsum[NW];
    __shared__ long long ncnt[NW];
    extern __shared__ double Fl[];                   // forcing: [row][component][column pair], real rows / pairs only

    const int64_t m = a.member0 + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t yc = a.yc, xc = a.xc;
    const double u = a.sc_.undef;
    double *Sg = a.S + m * a.sS;
    const double *Fg = a.c[M::FQ] + m * a.sc[M::FQ];

    // ---- per-row table: coefficients, hoisted relaxation factor, row part of the predicate ----
    for (int j = tid; j < NW * RW; j += NW * 64) {
        double t[NR];
        bool ok = false;
        if (j < yc) ok = M::row_setup(a, m, j, t);
        else for (int q = 0; q < NR; q++) t[q] = 0.0;
        for (int q = 0; q < NR; q++) rowc[j][q] = t[q];
        rowok[j] = (ok && j >= 1 && j <= yc - 2) ? 1 : 0;
    }
    __syncthreads();

    // ---- the band of this wavefront: S, forcing, predicate bits ------------------------------
    double2 S[RW][NSEG];
    const int xp = (int)((xc + 1) >> 1);             // column pairs per row
    unsigned long long mb = 0ull;
    int npts = 0;                                    // real points this lane holds
    bool anyu = false;                               // an S == undef somewhere: the norm must test for it
    SFOR(r, RW)
        SFOR(k, NSEG)
            const int64_t j = (int64_t)wave * RW + r;
            const int64_t c0 = 128 * k + 2 * lane, c1 = c0 + 1;
            const bool in0 = (j < yc) && (c0 < xc), in1 = (j < yc) && (c1 < xc);
            double2 s, f;
            s.x = in0 ? Sg[j * xc + c0] : 0.0;  s.y = in1 ? Sg[j * xc + c1] : 0.0;
            f.x = in0 ? Fg[j * xc + c0] : u;    f.y = in1 ? Fg[j * xc + c1] : u;
            const bool rok = (j < yc) && rowok[j < yc ? j : 0];
            const bool ok0 = a.per ? in0 : (c0 >= 1 && c0 <= xc - 2);
            const bool ok1 = a.per ? in1 : (c1 >= 1 && c1 <= xc - 2);
            const bool b0 = in0 && ok0 && rok && (f.x != u), b1 = in1 && ok1 && rok && (f.y != u);
            mb |= (unsigned long long)(b0 ? 1 : 0) << ((r * NSEG + k) * 2);
            mb |= (unsigned long long)(b1 ? 1 : 0) << ((r * NSEG + k) * 2 + 1);
            npts += (in0 ? 1 : 0) + (in1 ? 1 : 0);
            anyu = anyu || (in0 && s.x == u) || (in1 && s.y == u);
            S[r][k] = s;
            if (in0) Fl[((int)j * 2 + 0) * xp + 64 * k + lane] = M::prep(f.x, a.sc_);
            if (in1) Fl[((int)j * 2 + 1) * xp + 64 * k + lane] = M::prep(f.y, a.sc_);
        SEND
    SEND
    const long long wave_pts = xinv_wave_sum_ll((long long)npts);
    // seam lanes: column xc-1 sits in lane `le` of the last segment
    const int le = (int)(((xc - 1) & 127) >> 1);

    auto publish = [&](int X0, int XL) {             // component X0 of the first row, XL of the last row
        SFOR(k, NSEG)
            edge[wave + 1][0][k][X0][lane] = X0 ? S[0][k].y : S[0][k].x;
            edge[wave + 1][1][k][XL][lane] = XL ? S[RW - 1][k].y : S[RW - 1][k].x;
        SEND
    };
    if (wave == 0)
        for (int q = lane; q < 2 * NSEG * 2 * 64; q += 64) { (&edge[0][0][0][0][0])[q] = 0.0; (&edge[NW + 1][0][0][0][0])[q] = 0.0; }
    publish(0, 0); publish(1, 1);
    __syncthreads();

    // colour C (0 = red: j + i even, 1 = black) on every row of the band
    auto half = [&](auto ctag) {
        constexpr int C = decltype(ctag)::value;
        constexpr int XF = C, XL = ((RW - 1) & 1) ^ C;           // component updated in the first / last slot
        // the row table is re-read from LDS every half-sweep (two ds_read_b128 per row): hoisted out of
        // the sweep loop it would hold 8-12 VGPRs per row for the whole solve
        int rbase = wave * RW;
        asm volatile("" : "+s"(rbase));
        // the predicate bits stay ONE word pair per lane: opaque here, or the optimiser expands them
        // into 2*RW*NSEG lane masks in SGPRs ahead of the loop (and spills those)
        const int flast = (int)yc * 2 * xp - 1;
        unsigned mlo = (unsigned)mb, mhi = (unsigned)(mb >> 32);
        asm volatile("" : "+v"(mlo), "+v"(mhi));
        double eA[NSEG], eB[NSEG];
        SFOR(k, NSEG)
            eA[k] = edge[wave][1][k][XF][lane];            // last row of the band above (zeros above band 0)
            eB[k] = edge[wave + 2][0][k][XL][lane];        // first row of the band below
        SEND
        // row table: the next row's entries are requested before this row's arithmetic; the compiler
        // fence keeps the scheduler from pulling every row's reads to the top of the half-sweep
        // (8-12 VGPRs per row: the big variants would spill)
        double tn[NR];
#pragma unroll
        for (int q = 0; q < NR; q++) tn[q] = rowc[rbase][q];
        SFOR(r, RW)
            constexpr int X = (r & 1) ^ C;                         // the band starts on an even row (RW even)
            double t[NR];
#pragma unroll
            for (int q = 0; q < NR; q++) t[q] = tn[q];
            if constexpr (r + 1 < RW) {
#pragma unroll
                for (int q = 0; q < NR; q++) tn[q] = rowc[rbase + r + 1][q];
            }
#if XINV_SMALL_SCHED_BARRIER
            asm volatile("" ::: "memory");
#endif
            SFOR(k, NSEG)
                const double sC = comp<X>(S[r][k]);
                double sP, sM;
                if constexpr (r + 1 < RW) sP = comp<X>(S[r + 1][k]); else sP = eB[k];
                if constexpr (r > 0) sM = comp<X>(S[r - 1][k]); else sM = eA[k];
                double w, e;
                if (X == 0) {
                    e = S[r][k].y;
                    w = xinv_lane_up(S[r][k].y);
                    // lane 0: the column west of this segment -- the last column of the previous
                    // segment, or (periodic wrap; harmless otherwise: column 0 is then never updated)
                    // column xc-1
                    double src;
                    if constexpr (k > 0) src = small_readlane(S[r][k > 0 ? k - 1 : 0].y, 63);
                    else                 src = small_readlane(S[r][NSEG - 1].y, le);
                    w = small_setlane(w, src, lane == 0);
                } else {
                    w = S[r][k].x;
                    e = xinv_lane_down(S[r][k].x);
                    if constexpr (k < NSEG - 1) e = small_setlane(e, small_readlane(S[r][k < NSEG - 1 ? k + 1 : 0].x, 0), lane == 63);
                    else                        e = small_setlane(e, small_readlane(S[r][0].x, 0), lane == le);
                }
                // the forcing of this component: phantom lanes / rows read a clamped (unused) entry
                int fi = ((rbase + r) * 2 + X) * xp + 64 * k + lane;
                fi = fi < flast ? fi : flast;
                const double v = M::upd(t, Fl[fi], sC, sP, sM, w, e, a.sc_);
                constexpr int bit = (r * NSEG + k) * 2 + X;
                unsigned msk = (unsigned)((int)(((bit < 32) ? mlo : mhi) << (31 - (bit & 31))) >> 31);   // bit -> all-ones / zero (v_bfe_i32)
                asm("" : "+v"(msk));
                setc<X>(S[r][k], xinv_bitsel(msk, v, sC));
#if XINV_SMALL_SCHED_BARRIER
                // one update at a time: all updates of a half-sweep are independent, and left alone the
                // scheduler hoists every neighbour shift / seam read to the top (2-4 registers per row
                // segment, spills in the large variants); two wavefronts per SIMD provide the overlap
                __builtin_amdgcn_sched_barrier(0);
#endif
            SEND
        SEND
        publish(XF, XL);
    };

    XinvCtl lc;
    lc.normPrev = DBL_MAX; lc.flag1 = 0.0; lc.flag2 = 0.0; lc.loop = 0; lc.sweeps = 0;
    lc.done = 0; lc.overflow = 0; lc.wrote = 0; lc.ticket = 0; lc.seq = 1; lc.pad_ = 0;

    // 'extend' pre-pass of a sweep (numbas.py:284-310): rows 0 / yc-1 <- rows 1 / yc-2.  The host
    // picks RW so that row yc-1 is not the first row of its band: both copies are local to one
    // wavefront and neither target row is ever read by another one.
    const int wlast = (int)((yc - 1) / RW), rlast = (int)((yc - 1) % RW);
    auto extend = [&]() {
        SFOR(r, RW)
            SFOR(k, NSEG)
                const bool top = (wave == 0 && r == 0), bot = (wave == wlast && r == rlast);
                const LaneCols lcol = make_lanecols<false>(128 * k, 0, 128, lane, xc, a.per != 0);
                double2 e = S[r][k];                        // (a local copy: S must never have its address taken)
                if constexpr (r + 1 < RW) { if (top) { const double2 in = S[r + 1][k]; fused_extend_fix(e, in, lcol, a.tall != 0, u); } }
                if constexpr (r > 0)      { if (bot) { const double2 in = S[r - 1][k]; fused_extend_fix(e, in, lcol, a.tall != 0, u); } }
                S[r][k] = e;
            SEND
        SEND
    };

    for (;;) {
        if (EXT) extend();                               // BCy == 'extend': compiled out otherwise
        half(std::integral_constant<int, 0>{});
        __syncthreads();
        half(std::integral_constant<int, 1>{});
        // ---- this band's share of mean|S| ----------------------------------------------------
        double acc = 0.0;
        bool hit = anyu;
        SFOR(r, RW)
            SFOR(k, NSEG)
                acc += fabs(S[r][k].x); acc += fabs(S[r][k].y);        // phantoms hold 0
                hit = hit || (S[r][k].x == u) || (S[r][k].y == u);
            SEND
        SEND
        long long cnt = wave_pts;
        if (__any(hit)) {                                              // rare: some S equals undef
            acc = 0.0; int c = 0;
            int lane3 = lane;
            asm volatile("" : "+v"(lane3));
            SFOR(r, RW)
                SFOR(k, NSEG)
                    const int64_t j = (int64_t)wave * RW + r;
                    const int64_t c0 = 128 * k + 2 * lane3;
                    const bool in0 = (j < yc) && (c0 < xc) && (S[r][k].x != u);
                    const bool in1 = (j < yc) && (c0 + 1 < xc) && (S[r][k].y != u);
                    acc += in0 ? fabs(S[r][k].x) : 0.0; acc += in1 ? fabs(S[r][k].y) : 0.0;
                    c += (in0 ? 1 : 0) + (in1 ? 1 : 0);
                SEND
            SEND
            cnt = xinv_wave_sum_ll((long long)c);
        }
        acc = xinv_wave_sum(acc);
        if (lane == 0) { nsum[wave] = acc; ncnt[wave] = cnt; }
        __syncthreads();
        double ts = 0.0; long long tc = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) { ts += nsum[q]; tc += ncnt[q]; }
        xinv_ctl_update(&lc, ts, tc, a.stop);
        if (lc.done) break;
    }

    // ---- the slice goes back to HBM once --------------------------------------------------------
    // (addresses and in-range predicates are re-derived from laundered values: shared with the load
    // phase they would stay live -- two to three registers per row segment -- through the whole solve)
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    SFOR(r, RW)
        SFOR(k, NSEG)
            const int64_t j = (int64_t)wave * RW + r;
            const int64_t c0 = 128 * k + 2 * lane2;
            if (j < yc && c0 < xc) Sg[j * xc + c0] = S[r][k].x;
            if (j < yc && c0 + 1 < xc) Sg[j * xc + c0 + 1] = S[r][k].y;
        SEND
    SEND
    if (tid == 0) a.ctl[m] = lc;
}
