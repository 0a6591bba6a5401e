// xinv_fused3d2.h -- TWO red-black sweeps per pass for the 3-D standard form with x-uniform
// coefficients (every lat-lon omega coefficient, apps.py:2033-2035): BASELINE configs[4].
//
// k_fused3d (one sweep per pass) sits at what the fabric delivers with ~27 bytes per point-sweep;
// only fewer bytes help.  Here a pass applies two sweeps -- S and the forcing are read once and S
// written once per TWO sweeps -- and a wavefront owns TWO adjacent j rows, so that the deeper halo
// (four rows per side instead of two) costs no more than before: NW wavefronts = 2*NW rows per
// cross-section, 2*NW - 8 of them owned (12 wavefronts: 24 rows, 16 owned).
//
// Pipeline.  With `r` the plane just loaded, stage q = 1..4 runs on plane r-q: the red and black
// half-sweeps of sweep 1 (planes r-1, r-2) and of sweep 2 (planes r-3, r-4); plane r-4 leaves with
// two complete sweeps.  All four stages of a step touch the same lane component of a row (the
// colour is (k+j+i)&1 and the planes step by one with the stages), and the two rows of a wavefront
// touch opposite components.  Neighbours:
//   k +/- 1   own registers (six-plane rotating window per row);
//   i +/- 1   DPP wave shift;
//   j +/- 1   between the two rows of a wavefront: the other row's registers -- the component read is
//             the one that row does not touch in this step, so it is exactly one stage behind, as
//             required; across wavefronts: through LDS, where a wavefront publishes after every
//             stage the component it has just produced in its first and in its second row, for the
//             neighbours' NEXT step (double-buffered by step parity): ONE workgroup barrier per plane.
// Coefficients are one value per (plane, row): a wavefront parks them -- with the hoisted relaxation
// factor and the row part of the predicate -- in a private LDS table when a plane enters and reads
// them back (broadcast) at each stage: 2 rows x 6 planes x 6 values would not fit its registers.
// Halo: four rows / columns / planes per side, recomputed, never exchanged; S ping-pongs.
// Same point arithmetic as k_fused3d<UNI> (hoisted per-row relaxation factor), same ordering:
// bitwise equal to two passes of it and to the oracle.
#pragma once
#include "xinv_fused3d.h"

template <int NW, bool AL>
__global__ __launch_bounds__(NW * 64) void k_fused3d2(Fused3Args a)
{
    constexpr int K = 2, H = 2 * K, UW = 128 - 2 * H, D = 2 * K + 2, RJ = 2 * NW - 2 * H;

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
    const int64_t xu0 = (int64_t)st * UW;
    const double u = a.sc_.undef;
    const LaneCols lc = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - H + 2 * lane;

    // the two rows of this wavefront (RJ and H are even: row 0 of a wavefront is an even row)
    const int64_t j0 = (int64_t)jb * RJ - H + 2 * wave;
    int64_t jr[2], jr1[2];
    bool row_upd[2], row_use[2];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int64_t j = j0 + rr;
        jr[rr] = j < 0 ? 0 : (j > yc - 1 ? yc - 1 : j);
        jr1[rr] = (j + 1) < 0 ? 0 : ((j + 1) > yc - 1 ? yc - 1 : (j + 1));
        row_upd[rr] = (j >= 1) && (j <= yc - 2);
        row_use[rr] = (2 * wave + rr >= H) && (2 * wave + rr < 2 * NW - H) && (j < yc);
    }
    const int wm = wave > 0 ? wave - 1 : 0, wp = wave < NW - 1 ? wave + 1 : NW - 1;

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *pA = a.c[0] + m * a.sc[0], *pB = a.c[1] + m * a.sc[1];
    const double *pC = a.c[2] + m * a.sc[2], *pF = a.c[3] + m * a.sc[3];

    // [step parity][level 0..3: as loaded, after stages 1, 2, 3][wave][row][lane]
    __shared__ double xch[2][4][NW][2][64];
    // per-wavefront private: [wave][row][plane slot]{A, B[j], B[j+1], C, optArg/denominator, row predicate}
    __shared__ double ctab[NW][2][2 * 2 + 2][8];

    // plane r is requested one step ahead: S and the forcing into a register pack, its coefficients
    // straight into the table slot it will occupy (the plane that slot held, r-6, is no longer read)
    struct Pack { double2 s[2], f[2]; };
    auto load = [&](int64_t r, int slot) {
        Pack p;
        const int64_t pr = r > zc - 1 ? zc - 1 : (r < 0 ? 0 : r);
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int64_t off = (pr * yc + jr[rr]) * xc, off1 = (pr * yc + jr1[rr]) * xc;
            p.s[rr] = ld2<AL>(srcS, off, lc);
            p.f[rr] = ld2<AL>(pF, off, lc);
            double *en = ctab[wave][rr][slot];       // every lane writes the same values (see below)
            en[0] = pA[off]; en[1] = pB[off]; en[2] = pB[off1]; en[3] = pC[off];
        }
        return p;
    };

    double acc[K] = {0.0, 0.0};
    int cnt[K] = {0, 0};

    double2 sw[2][D], fw[2][D];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
#pragma unroll
        for (int t = 0; t < D; t++) {
            sw[rr][t] = make_double2(0.0, 0.0); fw[rr][t] = sw[rr][t];
            for (int q = 0; q < 8; q++) ctab[wave][rr][t][q] = 0.0;
        }
    }

    // one point update of component X of row rr on the plane in slot sk (k+1 in skp, k-1 in skm)
    auto update = [&](auto rtag, int sk, int skp, int skm, int64_t kk, double jP, double jM, auto xt) {
        constexpr int rr = decltype(rtag)::value;
        constexpr int X = decltype(xt)::value;
        const bool okc = X ? lc.ok_y : lc.ok_x;
        const bool inr = okc && row_upd[rr] && (kk >= 1) && (kk <= zc - 2);
        double w, e;
        row_neighbours<X>(sw[rr][sk], w, e);
        const double sC = comp<X>(sw[rr][sk]), sKP = comp<X>(sw[rr][skp]), sKM = comp<X>(sw[rr][skm]);
        const double *ce = ctab[wave][rr][sk];
        const double aP = ctab[wave][rr][skp][0], a0 = ce[0], b0 = ce[1], bP = ce[2];
        const double cE = ce[3], c0 = ce[3], f = comp<X>(fw[rr][sk]);
        const bool cond = inr && (ce[5] != 0.0) && (f != u);
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
        temp *= ce[4];
        const double v = cond ? sC + temp : sC;
        setc<X>(sw[rr][sk], v);
        return v;
    };

    // one pipeline step: plane r (= rbase + U) enters slot U.  X0 = component row 0 touches.
    auto step = [&](int64_t r, const Pack &p, auto utag) {
        constexpr int U = decltype(utag)::value;
        constexpr int X0 = (1 + (U & 1)) & 1, X1 = 1 - X0;          // j0 is even
        using XT0 = std::integral_constant<int, X0>;
        using XT1 = std::integral_constant<int, X1>;
        using R0 = std::integral_constant<int, 0>;
        using R1 = std::integral_constant<int, 1>;
#define SL(w) ((U - (w) + 4 * D) % D)                            /* slot of plane r - w */
        const int bw = U & 1, br = (U + 1) & 1;

#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            sw[rr][U] = p.s[rr]; fw[rr][U] = p.f[rr];
            // plane r-1: A[r] and A[r-1] are known now -- hoisted relaxation factor and row predicate.
            // Every lane writes the same values: each lane's later reads are then ordered after ITS OWN
            // writes (a lane-0-only write would be a cross-lane dependency the compiler does not see).
            double *e1 = ctab[wave][rr][SL(1)];
            const double aP = ctab[wave][rr][U][0], a0 = e1[0], b0 = e1[1], bP = e1[2], c = e1[3];
            e1[4] = a.sc_.optArg / ((aP + a0) * a.sc_.ratio2Sqr +
                                    (bP + b0) * a.sc_.ratio1Sqr +
                                    (c + c));
            e1[5] = ((aP != u) && (a0 != u) && (bP != u) && (b0 != u) && (c != u)) ? 1.0 : 0.0;
        }
        xch[bw][0][wave][0][lane] = comp<X0>(sw[0][U]);              // as loaded: the neighbours' next stage 1
        xch[bw][0][wave][1][lane] = comp<X1>(sw[1][U]);

        // stage q on plane r-q; the rows above / below the pair come from the previous step's LDS
        xinv_unroll_steps([&](auto qtag) {
            constexpr int q = decltype(qtag)::value + 1;              // 1..4
            const int64_t kk = r - q;
            const double jM0 = xch[br][q - 1][wm][1][lane];           // row j0 - 1: second row of the wavefront above
            const double jP1 = xch[br][q - 1][wp][0][lane];           // row j0 + 2: first row of the wavefront below
            const double jP0 = comp<X0>(sw[1][SL(q)]);               // row j0 + 1, the component it does not touch now
            const double jM1 = comp<X1>(sw[0][SL(q)]);               // row j0, likewise (read before it is ... untouched: X1)
            const double v0 = update(R0{}, SL(q), SL(q - 1), SL(q + 1), kk, jP0, jM0, XT0{});
            const double v1 = update(R1{}, SL(q), SL(q - 1), SL(q + 1), kk, jP1, jM1, XT1{});
            if constexpr (q < 4) {
                xch[bw][q][wave][0][lane] = v0;
                xch[bw][q][wave][1][lane] = v1;
            }
            __builtin_amdgcn_sched_barrier(0);                        // one stage at a time: registers are the scarce resource here
            if constexpr ((q & 1) == 0) {                             // a black half-sweep completes sweep q/2
                const bool pin = (kk >= k0) && (kk < k1);
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    if (pin && row_use[rr]) {                         // wave-uniform: an owned row of an owned plane
                        const double2 t = sw[rr][SL(q)];
                        const bool cx = lc.use_x & (t.x != u);
                        const bool cy = lc.use_y & (t.y != u);
                        acc[q / 2 - 1] += (cx ? fabs(t.x) : 0.0);
                        acc[q / 2 - 1] += (cy ? fabs(t.y) : 0.0);
                        cnt[q / 2 - 1] += (cx ? 1 : 0) + (cy ? 1 : 0);
                        if constexpr (q == 4) {
                            double *d = dstS + (kk * yc + (j0 + rr)) * xc + st0;
                            if (AL) { if (lc.use_x) *reinterpret_cast<double2 *>(d) = t; }
                            else { if (lc.use_x) d[0] = t.x; if (lc.use_y) d[1] = t.y; }
                        }
                    }
                }
            }
        }, std::make_integer_sequence<int, 4>{});
#undef SL
        __syncthreads();
    };

    {
        // start a multiple of D planes below k0 - H (slot indices are compile-time; k0 is even, so
        // the parity of a plane is the parity of its slot), run until the last stage of plane k1 - 1
        const int64_t rstart = (k0 >= D) ? k0 - D : 0;
        // (one plane of prefetch: a second pack would push the 170-register budget of twelve
        // wavefronts into scratch; three wavefronts per SIMD cover the rest of the latency)
        Pack p0 = load(rstart, 0);
        const int64_t rlast = k1 - 1 + 2 * K;
        for (int64_t rb_ = rstart; rb_ <= rlast; rb_ += D) {
            xinv_unroll_steps([&](auto utag) {
                constexpr int U = decltype(utag)::value;
                const Pack pc = p0;
                p0 = load(rb_ + U + 1, (U + 1) % D);       // (that table slot is not read during this step)
                step(rb_ + U, pc, utag);
            }, std::make_integer_sequence<int, D>{});
        }
    }

    if (a.no_ctl) return;
    xinv_norm_finalize<K, NW>(acc, cnt, wave, lane, NB, T, tag,
                              a.psum + (size_t)m * XINV_KMAX * NB * XINV_PW, ctl, a.stop);
}
