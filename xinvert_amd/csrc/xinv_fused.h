// xinv_fused.h -- streaming fused red-black SOR sweep(s) for the 2-D 5-point forms (gfx950).
//
// The hot kernel of the engine (configs: Poisson / Stommel / Gill-Matsuno, B == 0).
//
// Decomposition.  One wavefront owns a tile: a strip of 128 columns (two adjacent columns per
// lane, one of each colour, so every lane works in every half-sweep) marched top to bottom over
// RY rows.  A 256-thread workgroup is four independent wavefronts on four consecutive tiles; the
// only workgroup-level step is the final reduction of the norm partials.  Neighbouring tiles
// overlap by a halo of 2K columns / 2K rows that is recomputed (never exchanged), so a launch
// needs no inter-workgroup synchronisation and S ping-pongs between two buffers.
//
// Pipeline.  Rows stream through a register window (no LDS).  With `r` the row just loaded,
// sweep s (s = 1..K) updates its red points on row r-2s+1 and its black points on row r-2s; row
// r-2K leaves the window with K complete sweeps applied.  All stages of one step update the same
// lane component ((r+1)&1), and each needs exactly one cross-lane operand (the other colour's
// value one column over), taken with a DPP wave shift.  The window rotates: the march is unrolled
// 2K+2 steps so every register index is a compile-time constant and rows never move.  HBM traffic
// per launch is one read of S and of each streamed coefficient array plus one write of S
// (+ halo re-reads), for K sweeps.  Loads are 16 B per lane (1 KiB per wavefront per array row),
// issued two rows ahead of use.
//
// x-uniform coefficients (template mask UM).  On lat-lon grids the reference materialises
// coefficient fields that depend on latitude only as full 2-D arrays (`zero + cos(lat)`,
// apps.py:1406-1408, 1630-1635).  The host detects arrays whose rows are constant along x
// (bitwise, one pass per solve) and the kernel then reads ONE value per row through the scalar
// unit instead of a 1 KiB vector row: no HBM stream, no vector registers, and the per-point
// divide `optArg / denom` -- now uniform along the row -- is evaluated once per row.  The
// arithmetic applied to each point is unchanged (same operands, same order), so results stay
// bitwise equal to the full-array path and to the oracle.
//
// Norm.  mean|S| of the reference (numbas.py:1710-1728) is accumulated per sweep for the rows
// and columns a tile owns, reduced wave -> workgroup -> sequence-tagged partials in a fixed
// order, and the workgroup dispatched last adds them in a fixed order (xinv_norm_finalize)
// and applies the stopping rule (numbas.py:401-414) on the device.  No floating-point atomics:
// the norm is run-to-run reproducible.
#pragma once
#include "xinv_device.h"
#include <type_traits>
#include <utility>

#define XINV_KMAX 4             /* most sweeps fused into one pass (standard form with per-row A, C); 2 elsewhere */
#ifndef XINV_TEST_HOOKS
#define XINV_TEST_HOOKS 0       /* 1: build/libxinv_hooks.so only (xinvert_amd/build.py: build_hooks) -- a tile can be told to
                                   withhold its norm partial, the reducer's watchdog is 30 ms instead of 2 s, and the host
                                   driver reads the XINV_EXP_WATCHDOG / XINV_HOOK_SKIP_PUBLISH switches.  The shipped library
                                   has none of it. */
#endif
#if XINV_TEST_HOOKS
#define XINV_WATCHDOG_TICKS 3000000ull          /* s_memrealtime ticks at 100 MHz: 30 ms */
#else
#define XINV_WATCHDOG_TICKS 200000000ull        /* 2 s */
#endif
// test-hooks build: does this tile of this launch withhold its partial?  (hook = FusedArgs::dbg as int[3])
__device__ __forceinline__ bool xinv_hook_withhold(const void *hook, int T, unsigned tag, int64_t m)
{
#if XINV_TEST_HOOKS
    const int *h = reinterpret_cast<const int *>(hook);
    return h && __hip_atomic_load(h + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == T &&
           (unsigned)__hip_atomic_load(h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag &&
           (int64_t)__hip_atomic_load(h + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m;
#else
    (void)hook; (void)T; (void)tag; (void)m;
    return false;
#endif
}
typedef int64_t row_t;
// Stage skipping: half-sweep h of a halo row d rows outside the owned block can only reach an owned
// row's final value if h + d <= 2K; the other half-sweep stages of the pipeline (a triangle at each
// end of the tile, and the stages that run on empty window slots while the pipeline fills) are
// jumped over with a wave-uniform branch.  What they would have written is never read by a stage
// that matters, so results are unchanged bit for bit.  MEASURED AND NOT KEPT (A/B switch, default off):
// 24 % fewer stage executions at 3600x1800, K = 4 -- and 46.3 -> 61.5 us per launch: the branches
// cut each pipeline step into eight basic blocks, the scheduler can no longer interleave
// independent stages, and every block boundary waits on its operands.

struct FusedArgs {
    const double *src;
    double *dst;
    const double *c[7];        // 5-point: std A, C, F ; gen A, C, D, E, F, G (B == 0)
                               // 9-point: std A, B, C, F ; gen A, B, C, D, E, F, G
    int64_t sS, sc[7];         // batch strides (elements)
    int64_t yc, xc;
    int per, ext, tall;
    int nstrip, nrb, RY;       // x strips, row blocks, rows per tile (0: even split)
    int nwg;                   // workgroups per member = ceil(nstrip * nrb / 4)
    int force, no_ctl;
    int lag;                   // 1: publish the norm partials with `tag` and return; the NEXT launch's extra
                               //    workgroup (or k_norm_reduce_lag) evaluates them
    unsigned tag;
    // the previous launch's partials, evaluated by workgroup `nwg` of this launch (lagp_tag == 0: none)
    unsigned long long *lagp_psum;
    const double *lagp_xsum;
    const long long *lagp_xcnt;
    int lagp_NB, lagp_K;
    unsigned lagp_tag;
    int64_t member0;
    XinvScal sc_;
    XinvCtl *ctl;
    XinvStop stop;
    unsigned long long *psum;  // [nbatch][XINV_KMAX][NB][3] sequence-tagged norm partials (xinv_norm_finalize)
    // Masked-tile skipping (5-point kernel): wave-tiles whose owned points are all masked never
    // change, so the launch runs only the listed ones and adds the skipped tiles' constant share
    // of the norm.  nullptr = every tile.
    const int *tile_list;      // [nbatch][ntl] wave-tile ids, -1 = idle wavefront
    int ntl;                   // entries per member (multiple of 4); nwg = ntl / 4
    const double *xsum;        // [nbatch] sum |S| over the skipped tiles (S != undef)
    const long long *xcnt;     // [nbatch] their sample count
    const void *rowf;          // k_pipe2d: [nbatch][yc] per-row records (M::PIPE_RW doubles each, xinv_pipe2d.h)
    double *dbg;               // test-hooks build only: the hook record {tile, launch tag, member} (xinv_hook_withhold);
                               // test-hooks build (XINV_TEST_HOOKS): three ints {tile, launch tag, member} -- that tile of
                               // that launch withholds its norm partial, so that the reducer REALLY times out
};

#include "xinv_tiles.h"                              // tile ids, the seam launches' dispatch order

template <class F, int... U>
__device__ __forceinline__ void xinv_unroll_steps(F &&f, std::integer_sequence<int, U...>)
{
    (f(std::integral_constant<int, U>{}), ...);
}

template <int X> __device__ __forceinline__ double comp(const double2 &v) { return X ? v.y : v.x; }
template <int X> __device__ __forceinline__ void setc(double2 &v, double t) { if (X) v.y = t; else v.x = t; }

// value of the neighbouring column: X == 0 -> west neighbour lives in lane-1's .y, east is own .y
//                                   X == 1 -> west is own .x, east neighbour lives in lane+1's .x
template <int X> __device__ __forceinline__ void row_neighbours(const double2 &row, double &w, double &e)
{
    if (X == 0) { w = xinv_lane_up(row.y); e = row.y; }
    else        { w = row.x; e = xinv_lane_down(row.x); }
}

// Per-row register window of the coefficient streams.  Stream q is either a vector row
// (v[q][slot], two columns per lane) or, when bit q of UM is set, one scalar per row (s[q][slot]).
// rq / rok: per-row relaxation factor optArg/denom and uniform part of the mask predicate, used
// when the model's denominator is x-uniform (M::hoist<UM>()).
// mx / my: the complete update predicate of the row's two columns (row and column in range and
// every operand the reference's `cond` lists defined), evaluated ONCE when the row below it has
// entered and kept as all-ones / zero words, so that each of the 2K half-sweeps that touch the
// row selects with two v_bfi instead of re-deriving the predicate (compares + lane-mask logic).
template <int NC, int D> struct CoefWin {
    double2 v[NC][D];
    double s[NC][D];
    double rq[D];
    double2 rqv[D];            // general form, coefficients varying along x: the point's relaxation factor, divided
                               // ONCE when the row enters the window (derive) instead of in each of the 2K half-sweeps
    unsigned mx[D], my[D];
};

// all-ones / zero word per lane, opaque to the optimiser: it would otherwise fold the word back
// into a lane mask in SGPRs and re-create the scalar logic this representation is there to avoid
__device__ __forceinline__ unsigned xinv_lane_word(bool b)
{
    unsigned m = b ? ~0u : 0u;
    return m;
}

// bitwise select: m == ~0u -> a, m == 0 -> b   (v_bfi_b32 on each half; exact, no arithmetic)
__device__ __forceinline__ double xinv_bitsel(unsigned m, double a, double b)
{
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a);
    const unsigned long long ub = (unsigned long long)__double_as_longlong(b);
    const unsigned lo = ((unsigned)ua & m) | ((unsigned)ub & ~m);
    const unsigned hi = ((unsigned)(ua >> 32) & m) | ((unsigned)(ub >> 32) & ~m);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// select by a wave-uniform 64-bit lane mask (an SGPR pair): lanes of m -> a, the others -> b (two v_cndmask_b32)
__device__ __forceinline__ double xinv_bitsel64(unsigned long long m, double a, double b)
{
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %2, %3, %6\n\tv_cndmask_b32_e64 %1, %4, %5, %6"
        : "=&v"(lo), "=&v"(hi)
        : "v"((unsigned)__double2loint(b)), "v"((unsigned)__double2loint(a)),
          "v"((unsigned)__double2hiint(b)), "v"((unsigned)__double2hiint(a)), "s"(m));
    return __hiloint2double((int)hi, (int)lo);
}

template <int X, unsigned UM, int Q, int NC, int D>
__device__ __forceinline__ double cget(const CoefWin<NC, D> &w, int slot)
{
    if ((UM >> Q) & 1u) return w.s[Q][slot];
    return comp<X>(w.v[Q][slot]);
}

// ---- models: which coefficient streams exist and how a point is updated -------------------
struct FusedStd2D {                 // numbas.invert_standard_2D, B == 0
    static constexpr int NC = 3;    // A, C, F
    template <unsigned UM> static constexpr bool hoist() { return (UM & 3u) == 3u; }   // A and C uniform
    // wave-pipelined pass (xinv_pipe2d.h): the update of row j reads A[j+1], so the per-row record is asked for
    // two steps ahead
    static constexpr int PIPE_PFR = 2;

    // called once per step after row r entered slot `sr`; `s1` = slot of row r-1, whose operands
    // (A[r], A[r-1], C[r-1], F[r-1]) are all in the window now: relaxation factor when it is
    // x-uniform, the update predicate (numbas.py:344-348), and F*delxSqr in place of F.
    // PRE: F * delxSqr replaces F in the window here (each row is touched by 2K half-sweeps: k_fused2d); false:
    // the window keeps F and `upd<.., false>` multiplies at use (k_pipe2d: one sweep per wavefront, the same
    // two multiplications per row, and the original F can ride the LDS ring to the next wavefront)
    template <unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ void derive(CoefWin<NC, D> &w, int sr, int s1, bool okx, bool oky,
                                                  const XinvScal &sc)
    {
        const double u = sc.undef;
        const double fx = cget<0, UM, 2>(w, s1), fy = cget<1, UM, 2>(w, s1);
        bool bx = okx && (fx != u), by = oky && (fy != u);
        if (hoist<UM>()) {
            const double aP = w.s[0][sr], a0 = w.s[0][s1], c = w.s[1][s1];
            w.rq[s1] = sc.optArg / ((aP + a0) * sc.ratioSqr + (c + c));
            const bool rok = (aP != u) && (a0 != u) && (c != u);
            bx = bx && rok; by = by && rok;
        } else {
            const double aPx = cget<0, UM, 0>(w, sr), aPy = cget<1, UM, 0>(w, sr);
            const double a0x = cget<0, UM, 0>(w, s1), a0y = cget<1, UM, 0>(w, s1);
            const double c0x = cget<0, UM, 1>(w, s1), c0y = cget<1, UM, 1>(w, s1);
            const double cEy = ((UM >> 1) & 1u) ? w.s[1][s1] : xinv_lane_down(w.v[1][s1].x);
            bx = bx && (aPx != u) && (a0x != u) && (c0y != u) && (c0x != u);      // east of .x is .y
            by = by && (aPy != u) && (a0y != u) && (cEy != u) && (c0y != u);
        }
        w.mx[s1] = xinv_lane_word(bx);
        w.my[s1] = xinv_lane_word(by);
        if (PRE) {
            if (!((UM >> 2) & 1u)) { w.v[2][s1].x = fx * sc.delxSqr; w.v[2][s1].y = fy * sc.delxSqr; }
            else                   w.s[2][s1] = fx * sc.delxSqr;
        }
    }

    // sj = slot of row j, sjp = slot of row j+1.  inc: the increment of the point (numbas.py:350-369 without the
    // final `S[j,i] += temp`); upd: the point's new value, or the old one where the predicate fails.
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double inc(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double aP = cget<X, UM, 0>(w, sjp);
        const double a0 = cget<X, UM, 0>(w, sj);
        const double c0 = cget<X, UM, 1>(w, sj);
        double cE;
        if ((UM >> 1) & 1u) cE = w.s[1][sj];
        else if (X == 0)    cE = w.v[1][sj].y;
        else                cE = xinv_lane_down(w.v[1][sj].x);
        const double fd = PRE ? cget<X, UM, 2>(w, sj)        // F * delxSqr (see derive)
                              : cget<X, UM, 2>(w, sj) * sc.delxSqr;
        double temp = (
            (
                aP * (sP - sC) -
                a0 * (sC - sM)
            ) * sc.ratioSqr + (
                cE * (sE - sC) -
                c0 * (sC - sW)
            )
        ) - fd;
        if (hoist<UM>()) temp *= w.rq[sj];
        else             temp *= sc.optArg / ((aP + a0) * sc.ratioSqr + (cE + c0));
        return temp;
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double temp = inc<X, UM, D, PRE>(w, sj, sjp, sC, sP, sM, sW, sE, sc);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], sC + temp, sC);
    }
};

struct FusedStd2DT {                // numbas.invert_standard_2D_test, B == 0 and C == 0
    static constexpr int NC = 4;    // A, D, E, F
    template <unsigned UM> static constexpr bool hoist() { return (UM & 7u) == 7u; }   // A, D, E uniform

    template <unsigned UM, int D>
    static __device__ __forceinline__ void derive(CoefWin<NC, D> &w, int sr, int s1, bool okx, bool oky,
                                                  const XinvScal &sc)
    {
        const double u = sc.undef;
        bool bx = okx && (cget<0, UM, 3>(w, s1) != u), by = oky && (cget<1, UM, 3>(w, s1) != u);
        if (hoist<UM>()) {          // row r-1
            const double aP = w.s[0][sr], a0 = w.s[0][s1], d = w.s[1][s1], e = w.s[2][s1];
            w.rq[s1] = sc.optArg / ((aP + a0) * sc.ratioSqr +
                                    (d + d) - e * sc.delxSqr);
            const bool rok = (aP != u) && (a0 != u) && (d != u) && (e != u);
            bx = bx && rok; by = by && rok;
        } else {
            const double d0x = cget<0, UM, 1>(w, s1), d0y = cget<1, UM, 1>(w, s1);
            const double dEy = ((UM >> 1) & 1u) ? w.s[1][s1] : xinv_lane_down(w.v[1][s1].x);
            bx = bx && (cget<0, UM, 0>(w, sr) != u) && (cget<0, UM, 0>(w, s1) != u) && (d0y != u) &&
                 (d0x != u) && (cget<0, UM, 2>(w, s1) != u);
            by = by && (cget<1, UM, 0>(w, sr) != u) && (cget<1, UM, 0>(w, s1) != u) && (dEy != u) &&
                 (d0y != u) && (cget<1, UM, 2>(w, s1) != u);
        }
        w.mx[s1] = xinv_lane_word(bx);
        w.my[s1] = xinv_lane_word(by);
    }

    template <int X, unsigned UM, int D>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double aP = cget<X, UM, 0>(w, sjp);
        const double a0 = cget<X, UM, 0>(w, sj);
        const double d0 = cget<X, UM, 1>(w, sj);
        double dE;
        if ((UM >> 1) & 1u) dE = w.s[1][sj];
        else if (X == 0)    dE = w.v[1][sj].y;
        else                dE = xinv_lane_down(w.v[1][sj].x);
        const double e = cget<X, UM, 2>(w, sj);
        const double f = cget<X, UM, 3>(w, sj);
        double temp = (
            (
                aP * (sP - sC) -
                a0 * (sC - sM)
            ) * sc.ratioSqr + (
                dE * (sE - sC) -
                d0 * (sC - sW)
            )
        ) + (e * sC - f) * sc.delxSqr;
        if (hoist<UM>()) temp *= w.rq[sj];
        else             temp *= sc.optArg / ((aP + a0) * sc.ratioSqr +
                                              (dE + d0) - e * sc.delxSqr);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], sC + temp, sC);
    }
};

struct FusedGen2D {                 // numbas.invert_general_2D, B == 0
    static constexpr int NC = 6;    // A, C, D, E, F, G
    template <unsigned UM> static constexpr bool hoist() { return (UM & 0x13u) == 0x13u; }  // A, C, F uniform
    // wave-pipelined pass: the update of a row reads only its own per-row record, so it is asked for one step
    // ahead (fewer records live in SGPRs)
    static constexpr int PIPE_PFR = 1;

    // every operand of the predicate (numbas.py:1126-1129) sits on the point itself: row r-1 is
    // handled here like in the other models (its first half-sweep runs in this very step)
    template <unsigned UM, int D, bool PRE = true>       // (PRE: nothing is premultiplied in this form)
    static __device__ __forceinline__ void derive(CoefWin<NC, D> &w, int, int s1, bool okx, bool oky,
                                                  const XinvScal &sc)
    {
        const double u = sc.undef;
        bool bx = okx && (cget<0, UM, 5>(w, s1) != u), by = oky && (cget<1, UM, 5>(w, s1) != u);
        if (hoist<UM>()) {
            const double A = w.s[0][s1], C = w.s[1][s1], F = w.s[4][s1];
            w.rq[s1] = sc.optArg / ((A * sc.ratioSqr + C) * 2.0
                                    - F * sc.delxSqr);
            const bool rok = (A != u) && (C != u) && (F != u);
            bx = bx && rok; by = by && rok;
            bx = bx && (cget<0, UM, 2>(w, s1) != u) && (cget<0, UM, 3>(w, s1) != u);
            by = by && (cget<1, UM, 2>(w, s1) != u) && (cget<1, UM, 3>(w, s1) != u);
        } else {
            bx = bx && (cget<0, UM, 0>(w, s1) != u) && (cget<0, UM, 1>(w, s1) != u) &&
                 (cget<0, UM, 2>(w, s1) != u) && (cget<0, UM, 3>(w, s1) != u) && (cget<0, UM, 4>(w, s1) != u);
            by = by && (cget<1, UM, 0>(w, s1) != u) && (cget<1, UM, 1>(w, s1) != u) &&
                 (cget<1, UM, 2>(w, s1) != u) && (cget<1, UM, 3>(w, s1) != u) && (cget<1, UM, 4>(w, s1) != u);
            // (same expression as inc()'s: same bits; every operand sits on the point itself)
            w.rqv[s1].x = sc.optArg / ((cget<0, UM, 0>(w, s1) * sc.ratioSqr + cget<0, UM, 1>(w, s1)) * 2.0
                                       - cget<0, UM, 4>(w, s1) * sc.delxSqr);
            w.rqv[s1].y = sc.optArg / ((cget<1, UM, 0>(w, s1) * sc.ratioSqr + cget<1, UM, 1>(w, s1)) * 2.0
                                       - cget<1, UM, 4>(w, s1) * sc.delxSqr);
        }
        w.mx[s1] = xinv_lane_word(bx);
        w.my[s1] = xinv_lane_word(by);
    }

    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double inc(const CoefWin<NC, D> &w, int sj, int, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double A = cget<X, UM, 0>(w, sj), C = cget<X, UM, 1>(w, sj);
        const double Dd = cget<X, UM, 2>(w, sj), E = cget<X, UM, 3>(w, sj);
        const double F = cget<X, UM, 4>(w, sj), G = cget<X, UM, 5>(w, sj);
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
        if (hoist<UM>()) temp *= w.rq[sj];
        else if (PRE)    temp *= comp<X>(w.rqv[sj]);     // (k_fused2d: divided once per row entry; k_pipe2d: PRE = false)
        else             temp *= sc.optArg / ((A * sc.ratioSqr + C) * 2.0
                                              - F * sc.delxSqr);
        return temp;
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double temp = inc<X, UM, D, PRE>(w, sj, sjp, sC, sP, sM, sW, sE, sc);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], sC + temp, sC);
    }
};

// ---- opt-in contracted arithmetic (XINV_FLAG_FMA): the per-row-coefficient forms with explicit fma at fixed positions
// of the update -- the oracle's XO_FMA restatement, bit for bit; NOT the reference's arithmetic (tied to it by tests:
// <= 1e-12 relative after tens of sweeps, <= 1e-6 rel-L2 converged).  The relaxation factor and the predicate are the
// plain models' (same per-row records).  inc() returns the bracket BEFORE the relaxation factor: the callers finish
// with one fma(bracket, rq, S) (k_pipe2d: under the EXEC mask, xinv_fma_where_ne).
struct FusedStd2DF : FusedStd2D {
    static constexpr bool FMA = true;
    template <unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ void derive(CoefWin<NC, D> &w, int sr, int s1, bool okx, bool oky,
                                                  const XinvScal &sc)
    { FusedStd2D::derive<UM, D, false>(w, sr, s1, okx, oky, sc); }     // (F stays F: multiplied inside an fma at use)
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double inc(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        static_assert(hoist<UM>(), "contracted arithmetic: per-row A and C only");
        const double aP = w.s[0][sjp], a0 = w.s[0][sj], c = w.s[1][sj];
        const double y = __builtin_fma(aP, sP - sC, -(a0 * (sC - sM)));
        const double x = __builtin_fma(c, sE - sC, -(c * (sC - sW)));
        double t = __builtin_fma(y, sc.ratioSqr, x);
        t = __builtin_fma(-cget<X, UM, 2>(w, sj), sc.delxSqr, t);
        return t;
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double t = inc<X, UM, D, PRE>(w, sj, sjp, sC, sP, sM, sW, sE, sc);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], __builtin_fma(t, w.rq[sj], sC), sC);
    }
};

struct FusedGen2DF : FusedGen2D {
    static constexpr bool FMA = true;
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double inc(const CoefWin<NC, D> &w, int sj, int, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        static_assert((UM & 0x1fu) == 0x1fu, "contracted arithmetic: per-row A, C, D, E, F only");
        const double A = w.s[0][sj], C = w.s[1][sj], Dd = w.s[2][sj], E = w.s[3][sj], F = w.s[4][sj];
        const double G = cget<X, UM, 5>(w, sj);
        double t = (A * ((sP - sC) - (sC - sM))) * sc.ratioSqr;
        t = __builtin_fma(C, (sE - sC) - (sC - sW), t);
        double v = (Dd * (sP - sM)) * sc.ratio;
        v = __builtin_fma(E, sE - sW, v);
        t = __builtin_fma(v * sc.delx, 0.5, t);
        t = __builtin_fma(__builtin_fma(F, sC, -G), sc.delxSqr, t);
        return t;
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double t = inc<X, UM, D, PRE>(w, sj, sjp, sC, sP, sM, sW, sE, sc);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], __builtin_fma(t, w.rq[sj], sC), sC);
    }
};

// ---- general form, coefficients that vary along x (round 5; BASELINE configs[2]: Stommel with R(x, y)).  One more
// stream Q carries everything of a point that does not change during a solve: its relaxation factor
// optArg / ((A ratioSqr + C) 2 - F delxSqr) -- the expression FusedGen2D divides by when a row enters the window,
// evaluated ONCE per coefficient stack (k_point_factor: same expression, same bits; kept by a resident plan) -- and, as
// Q == 0, the verdict of the update predicate (numbas.py:1126-1129: some operand undefined).  A pass then multiplies and
// compares instead of dividing and testing six operands per point.  Streams: A, C, D, E, F, Q, G (the forcing last).
// (On the wave-pipelined pass the same model was measured and not kept: bit-exact, 175-181 VGPRs -- four vector
//  streams x eight row records --, two workgroups per CU, C3 Stommel 2.6e11 against 3.08e11 for k_fused2d with three
//  sweeps per pass: profiles/r05_point_factor.txt.)
// AC: A and C are the same numbers everywhere (an isotropic operator on a Cartesian grid: Stommel, apps.py:1733-1735,
// found by k_point_factor): C is read out of A's registers, one stream and its window less (the variant runs with bit 1
// of UM set, so that nothing of C is loaded as a vector).
template <bool AC> struct FusedGen2DQ_ {
    static constexpr int NC = 7;
    static constexpr bool PQ = true;
    static constexpr int QI = 5;                         // stream index of Q
    template <unsigned UM> static constexpr bool hoist() { return false; }
    static constexpr int PIPE_PFR = 1;
    template <unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ void derive(CoefWin<NC, D> &w, int, int s1, bool okx, bool oky, const XinvScal &)
    {
        w.mx[s1] = xinv_lane_word(okx && (w.v[QI][s1].x != 0.0));
        w.my[s1] = xinv_lane_word(oky && (w.v[QI][s1].y != 0.0));
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double upd(const CoefWin<NC, D> &w, int sj, int sjp, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double temp = inc<X, UM, D, PRE>(w, sj, sjp, sC, sP, sM, sW, sE, sc);
        return xinv_bitsel(X ? w.my[sj] : w.mx[sj], sC + temp, sC);
    }
    template <int X, unsigned UM, int D, bool PRE = true>
    static __device__ __forceinline__ double inc(const CoefWin<NC, D> &w, int sj, int, double sC,
                                                 double sP, double sM, double sW, double sE,
                                                 const XinvScal &sc)
    {
        const double A = cget<X, UM, 0>(w, sj), C = AC ? A : cget<X, UM, 1>(w, sj);
        const double Dd = cget<X, UM, 2>(w, sj), E = cget<X, UM, 3>(w, sj);
        const double F = cget<X, UM, 4>(w, sj), G = cget<X, UM, 6>(w, sj);
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
        temp *= cget<X, UM, QI>(w, sj);
        return temp;
    }
};
using FusedGen2DQ = FusedGen2DQ_<false>;
using FusedGen2DQA = FusedGen2DQ_<true>;
template <class M, class = void> struct ModelPQ { static constexpr bool value = false; };
template <class M> struct ModelPQ<M, std::void_t<decltype(M::PQ)>> { static constexpr bool value = M::PQ; };

template <class M, class = void> struct ModelFma { static constexpr bool value = false; };
template <class M> struct ModelFma<M, std::void_t<decltype(M::FMA)>> { static constexpr bool value = M::FMA; };

template <int NC> struct RowPack { double2 s; double2 c[NC]; double cs[NC]; };

struct LaneCols {
    int64_t l0, l1;            // load columns of .x / .y (wrapped or clamped)
    bool ok_x, ok_y;           // column may be updated (interior, or any column when periodic)
    bool use_x, use_y;         // column is owned (stored / counted) by this lane
    int cls_x, cls_y;          // extend fix: 0 none, 1 straight copy, 2 column 0, 3 column xc-1
};

template <bool AL>
__device__ __forceinline__ double2 ld2(const double *p, int64_t row_off, const LaneCols &lc)
{
    if (AL) return *reinterpret_cast<const double2 *>(p + row_off + lc.l0);
    double2 v; v.x = p[row_off + lc.l0]; v.y = p[row_off + lc.l1]; return v;
}

// Lane -> column map of one wavefront's strip: lane owns columns c0 = xu0 - H + 2*lane and c0+1.
// (np column pairs per lane, this is pair q: k_pipe2d's wide strips; np = 1, q = 0 everywhere else)
template <bool AL>
__device__ __forceinline__ LaneCols make_lanecols(int64_t xu0, int H, int UW, int lane, int64_t xc,
                                                  bool per, int np = 1, int q = 0)
{
    LaneCols lc;
    const int64_t c0 = xu0 - H + 2 * np * lane + 2 * q, c1 = c0 + 1;
    if (per) {
        int64_t w0 = c0 % xc; if (w0 < 0) w0 += xc;
        int64_t w1 = c1 % xc; if (w1 < 0) w1 += xc;
        lc.l0 = w0; lc.l1 = w1;
        lc.ok_x = lc.ok_y = true;
        lc.cls_x = lc.cls_y = 1;
    } else {
        if (AL) {
            int64_t p = c0 < 0 ? 0 : (c0 > xc - 2 ? xc - 2 : c0);
            lc.l0 = p; lc.l1 = p + 1;
        } else {
            lc.l0 = c0 < 0 ? 0 : (c0 > xc - 1 ? xc - 1 : c0);
            lc.l1 = c1 < 0 ? 0 : (c1 > xc - 1 ? xc - 1 : c1);
        }
        lc.ok_x = (c0 >= 1 && c0 <= xc - 2);
        lc.ok_y = (c1 >= 1 && c1 <= xc - 2);
        lc.cls_x = lc.ok_x ? 1 : (c0 == 0 ? 2 : (c0 == xc - 1 ? 3 : 0));
        lc.cls_y = lc.ok_y ? 1 : (c1 == xc - 1 ? 3 : 0);
    }
    lc.use_x = (c0 >= xu0 && c0 < xu0 + UW && c0 < xc);
    lc.use_y = (c1 >= xu0 && c1 < xu0 + UW && c1 < xc);
    return lc;
}

// RING (periodic x with ODD xc; round 5, k_pipe3d): the row as an EVEN ring of xc + 1 virtual columns -- the xc real ones
// and a phantom column P between xc-1 and 0.  Strips start on even virtual columns, so a lane's .x slot always holds an
// even virtual column and its .y slot an odd one: the lane slots keep ONE parity across a wrap (the SeamLanes scheme
// above flips it and runs a half-sweep of a wrapping tile as two or three lane-masked passes with both components of the
// j neighbours).  Column xc-1 (even) always sits in an .x slot and P in the .y slot of the same lane ("seam lanes").
//   * P is never updated (ok_y = false there), never owned; it is loaded from column xc-1's address and refreshed from the
//     lane's .x after every update of column xc-1: it always MIRRORS column xc-1.  Column 0 reads it as its west neighbour.
//   * The coloured order of this case (oracle: seq_colour) updates column xc-1 inside the half-sweep of its own colour --
//     the one that updates the .x slots of its row -- right AFTER column 0: that half-sweep masks the seam lanes out of
//     its pass and runs ONE more pass for them alone, whose east operand is the next lane's .x (the new column 0)
//     instead of the lane's own .y.  The other half-sweep of the row is the plain single pass: 1.5 passes on average,
//     and only in tiles that hold a seam lane.
//   * Information crosses the seam westwards one virtual column pair faster (0 -> xc-1 inside one half-sweep, P skipped)
//     and the west halo of a strip loses a slot to P: both halos get one column pair more (HW = H + 2 virtual columns
//     a side, the strip owns 128 - 2 H - 4).
struct RingSeam { unsigned long long lanes; bool any; };     // lanes whose .x slot holds column xc-1 (wave-uniform mask)
__device__ __forceinline__ LaneCols make_lanecols_ring(int64_t xu0, int HW, int UW, int lane, int64_t xc, RingSeam &rs)
{
    LaneCols lc;
    const int64_t c0 = xu0 - HW + 2 * lane, c1 = c0 + 1, xv = xc + 1;
    int64_t w0 = c0 % xv; if (w0 < 0) w0 += xv;              // (even: never P)
    int64_t w1 = c1 % xv; if (w1 < 0) w1 += xv;
    lc.l0 = w0; lc.l1 = (w1 == xc) ? xc - 1 : w1;
    lc.ok_x = true; lc.ok_y = (w1 != xc);
    lc.cls_x = lc.cls_y = 1;
    lc.use_x = (c0 >= xu0 && c0 < xu0 + UW && c0 < xc);
    lc.use_y = (c1 >= xu0 && c1 < xu0 + UW && c1 < xc);
    rs.lanes = __builtin_amdgcn_ballot_w64(w0 == xc - 1);
    rs.any = rs.lanes != 0ull;
    return lc;
}

// 'extend' pre-pass for one boundary row held in the window (numbas.py:284-310).
// edge = row 0 (or yc-1), inner = row 1 (or yc-2), both in their state at the start of sweep s.
__device__ __forceinline__ void fused_extend_fix(double2 &edge, const double2 &inner,
                                                 const LaneCols &lc, bool tall, double u)
{
    const double inner_w = xinv_lane_up(inner.y);              // column c0-1
    // component x (column c0)
    if (lc.cls_x == 1) { if (inner.x != u) edge.x = inner.x; }
    else if (lc.cls_x == 2) { if (inner.y != u) edge.x = inner.y; }           // (0,0) <- (1,1)
    else if (lc.cls_x == 3) {
        if (tall && inner.x != u) edge.x = inner.x;
        if (inner_w != u) edge.x = inner_w;                                    // <- column xc-2
    }
    // component y (column c0+1)
    if (lc.cls_y == 1) { if (inner.y != u) edge.y = inner.y; }
    else if (lc.cls_y == 3) {
        if (tall && inner.y != u) edge.y = inner.y;
        if (inner.x != u) edge.y = inner.x;                                    // <- column xc-2
    }
}

// ---- norm partials: wave -> workgroup -> global, then one workgroup finalises ----------------
// acc/cnt: this lane's sum |S| and count over S != undef for each of the K fused sweeps.
// No workgroup waits on memory here except the reducer: a workgroup publishes its partial of
// sweep s as three 64-bit words -- low half of the sum, high half, count -- each carrying the
// launch's sequence number (ctl->seq, read at kernel start) in its upper 32 bits, with relaxed
// agent-scope stores (write-through, each word single-copy atomic, any arrival order).  The
// workgroup dispatched last (blockIdx.x == gridDim.x - 1: every other one of the member is
// resident or finished by then, so waiting on them cannot deadlock) polls the words until all
// carry the current number, adds the partials in a fixed order (deterministic; no floating-point
// atomics), applies the reference's stop rule once per fused sweep and advances ctl->seq.
// The buffer is cleared per solve (sequence numbers restart at 1).  Against an arrival ticket
// (store, wait, atomic, wait, last one reads) this takes two memory round trips off the tail of
// EVERY workgroup: 32.8 -> 28.5 us per launch at 3600x1800.  NWV = wavefronts per workgroup.
#define XINV_PW 3          /* words per partial */
// publish this workgroup's partial of each of the K fused sweeps (no wait, any arrival order)
// (EXTS: the LDS scratch -- NWV * K * 16 bytes, + 8 for the reducer -- is the caller's `scr` instead of static
//  arrays of its own: k_pipe3d's rings fill the 160 KiB to the byte and lend a finished buffer)
template <int K, int NWV, bool EXTS = false>
__device__ __forceinline__ void xinv_norm_publish(const double (&acc)[K], const int (&cnt)[K],
                                                  int wave, int lane, int NB, int T, unsigned tag,
                                                  unsigned long long *pw, char *scr = nullptr, bool withhold = false)
{
    double (*ls)[K]; long long (*lcn)[K];
    if constexpr (EXTS) {
        ls = reinterpret_cast<double (*)[K]>(scr);
        lcn = reinterpret_cast<long long (*)[K]>(scr + sizeof(double) * NWV * K);
    } else {
        __shared__ double ls_[NWV][K];
        __shared__ long long lcn_[NWV][K];
        ls = ls_; lcn = lcn_;
    }
#pragma unroll
    for (int s = 0; s < K; s++) {
        double ws = xinv_wave_sum(acc[s]);
        long long wc = xinv_wave_sum_ll((long long)cnt[s]);
        if (lane == 0) { ls[wave][s] = ws; lcn[wave][s] = wc; }
    }
    __syncthreads();
    const unsigned long long hi = (unsigned long long)tag << 32;
    if (threadIdx.x < K && !withhold) {
        const int s = threadIdx.x;
        double ts = 0.0; long long tc = 0;
        for (int q = 0; q < NWV; q++) { ts += ls[q][s]; tc += lcn[q][s]; }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(ts);
        unsigned long long *q = pw + ((size_t)s * NB + T) * XINV_PW;
        __hip_atomic_store(q + 0, hi | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, hi | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, hi | (unsigned long long)(unsigned)tc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one workgroup of NWV wavefronts: wait until the K * NB partials carry `tag`, add them in a fixed
// order, apply the reference's stop rule once per fused sweep, advance ctl->seq
template <int K, int NWV, bool EXTS = false>
__device__ __forceinline__ void xinv_norm_reduce(int wave, int lane, int NB, unsigned tag,
                                                 unsigned long long *pw, XinvCtl *ctl, const XinvStop &stop,
                                                 double xsum, long long xcnt, char *scr = nullptr)
{
    double (*ls)[K]; long long (*lcn)[K]; unsigned *s_timeout_p;
    if constexpr (EXTS) {
        ls = reinterpret_cast<double (*)[K]>(scr);
        lcn = reinterpret_cast<long long (*)[K]>(scr + sizeof(double) * NWV * K);
        s_timeout_p = reinterpret_cast<unsigned *>(scr + 2 * sizeof(double) * NWV * K);
    } else {
        __shared__ double ls_[NWV][K];
        __shared__ long long lcn_[NWV][K];
        __shared__ unsigned s_timeout_;
        ls = ls_; lcn = lcn_; s_timeout_p = &s_timeout_;
    }
#define s_timeout (*s_timeout_p)
    if (threadIdx.x == 0) s_timeout = 0u;
    const unsigned long long hi = (unsigned long long)tag << 32;
    // reducer: item i = s * NB + t ; thread tid takes i = tid, tid + NT, ... four at a time so
    // that the twelve loads of a pass are in flight together
    constexpr int NT = NWV * XINV_WAVE, C = 4;
    const int tid = threadIdx.x, nitem = K * NB;
    double ps[K]; long long pc[K];
#pragma unroll
    for (int s = 0; s < K; s++) { ps[s] = 0.0; pc[s] = 0; }
    // Watchdog: a partial that never arrives would otherwise spin until the driver's reset.  After
    // ~2 s (s_memrealtime ticks at 100 MHz) the member is stopped with overflow = 2, which the
    // host reports as an internal error.
    unsigned long long t0 = 0;                         // taken lazily: only a pass that found nothing costs a clock read
    bool timed_out = false;
    for (int base = 0; base < nitem && !timed_out; base += NT * C) {
        unsigned long long w[C][XINV_PW];
        bool ready;
        do {
            ready = true;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int i = base + c * NT + tid;
                if (i < nitem) {
#pragma unroll
                    for (int k = 0; k < XINV_PW; k++)
                        w[c][k] = __hip_atomic_load(pw + (size_t)i * XINV_PW + k, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                } else {
#pragma unroll
                    for (int k = 0; k < XINV_PW; k++) w[c][k] = hi;
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int k = 0; k < XINV_PW; k++) ready = ready && ((unsigned)(w[c][k] >> 32) == tag);
            if (!ready) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > XINV_WATCHDOG_TICKS) { timed_out = true; s_timeout = 1u; }
            }
        } while (!ready && !timed_out);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int i = base + c * NT + tid;
            const double v = __longlong_as_double((long long)((w[c][1] << 32) | (w[c][0] & 0xffffffffull)));
            const long long n = (long long)(unsigned)w[c][2];
#pragma unroll
            for (int s = 0; s < K; s++) {
                const bool mine = (i < nitem) && (i >= s * NB) && (i < (s + 1) * NB);
                ps[s] += mine ? v : 0.0;
                pc[s] += mine ? n : 0;
            }
        }
    }
    __syncthreads();
    if (s_timeout) {
        if (tid == 0) {
            ctl->overflow = 2; ctl->done = 1; ctl->sweeps = ctl->loop + 1;
            ctl->seq = tag + 1u;
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < K; s++) {
        double ws = xinv_wave_sum(ps[s]);
        long long wc = xinv_wave_sum_ll(pc[s]);
        if (lane == 0) { ls[wave][s] = ws; lcn[wave][s] = wc; }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int s = 0; s < K; s++) {
            double ts = 0.0; long long tc = 0;
            for (int q = 0; q < NWV; q++) { ts += ls[q][s]; tc += lcn[q][s]; }
            xinv_ctl_update(ctl, ts + xsum, tc + xcnt, stop);     // + the skipped tiles' constant share
        }
        ctl->seq = tag + 1u;
    }
#undef s_timeout
}

template <int K, int NWV, bool EXTS = false>
__device__ __forceinline__ void xinv_norm_finalize(const double (&acc)[K], const int (&cnt)[K],
                                                   int wave, int lane, int NB, int T, unsigned tag,
                                                   unsigned long long *pw, XinvCtl *ctl,
                                                   const XinvStop &stop,
                                                   double xsum = 0.0, long long xcnt = 0, char *scr = nullptr,
                                                   bool withhold = false)
{
    xinv_norm_publish<K, NWV, EXTS>(acc, cnt, wave, lane, NB, T, tag, pw, scr, withhold);
    if (blockIdx.x != gridDim.x - 1) return;
    __syncthreads();                                   // (the publish step's LDS scratch is reused by the reducer)
    xinv_norm_reduce<K, NWV, EXTS>(wave, lane, NB, tag, pw, ctl, stop, xsum, xcnt, scr);
}

// The extra workgroup of a lagged launch (blockIdx.x == nwg): norm + stop rule of the PREVIOUS pass,
// whose partials are complete (that kernel has finished), while this pass's tiles run.  `A` is any
// argument struct with the lagp_* fields and `stop` (FusedArgs, FusedBihArgs).
template <class A>
__device__ __forceinline__ void xinv_lag_reduce_prev(const A &a, XinvCtl *ctl, int64_t m)
{
    if (!a.lagp_tag) return;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    unsigned long long *pw = a.lagp_psum + (size_t)m * XINV_KMAX * a.lagp_NB * XINV_PW;
    const double xs = a.lagp_xsum ? a.lagp_xsum[m] : 0.0;
    const long long xn = a.lagp_xcnt ? a.lagp_xcnt[m] : 0;
    switch (a.lagp_K) {
    case 1: xinv_norm_reduce<1, 4>(wv, ln, a.lagp_NB, a.lagp_tag, pw, ctl, a.stop, xs, xn); break;
    case 2: xinv_norm_reduce<2, 4>(wv, ln, a.lagp_NB, a.lagp_tag, pw, ctl, a.stop, xs, xn); break;
    case 3: xinv_norm_reduce<3, 4>(wv, ln, a.lagp_NB, a.lagp_tag, pw, ctl, a.stop, xs, xn); break;
    default: xinv_norm_reduce<4, 4>(wv, ln, a.lagp_NB, a.lagp_tag, pw, ctl, a.stop, xs, xn); break;
    }
}

// (test-hooks build: the hook record travels in FusedArgs::dbg; the other argument structs have none)
template <class A> __device__ __forceinline__ const void *xinv_hook_of(const A &) { return nullptr; }
__device__ __forceinline__ const void *xinv_hook_of(const FusedArgs &a) { return a.dbg; }

// end of a sweep kernel: lagged -> publish only; else publish + in-kernel reducer
template <int K, class A>
__device__ __forceinline__ void xinv_norm_tail(const A &a, const double (&acc)[K], const int (&cnt)[K], int wave,
                                               int lane, int NB, int T, unsigned tag, XinvCtl *ctl, int64_t m)
{
    unsigned long long *pw = a.psum + (size_t)m * XINV_KMAX * NB * XINV_PW;
    const bool wh = xinv_hook_withhold(xinv_hook_of(a), T, tag, m);
    if (a.lag) { xinv_norm_publish<K, 4>(acc, cnt, wave, lane, NB, T, tag, pw, nullptr, wh); return; }
    xinv_norm_finalize<K, 4>(acc, cnt, wave, lane, NB, T, tag, pw, ctl, a.stop,
                             a.xsum ? a.xsum[m] : 0.0, a.xcnt ? a.xcnt[m] : 0, nullptr, wh);
}


// Lagged evaluation (k_fused2d with FusedArgs::lag): the sweep kernel only publishes; the partials are
// added and the stop rule applied by an extra workgroup of the NEXT launch, while that pass's tiles
// run -- the reduction (a global round trip after the last tile) leaves the critical path between
// launches.  This one-workgroup kernel does the same for the last launch of a chunk, before the host
// reads the control blocks.  Same order of summation as the in-kernel reducer.
struct NormLagArgs {
    unsigned long long *psum;  // this launch's partial buffer: [nbatch][XINV_KMAX][NB][XINV_PW]
    XinvCtl *ctl;
    XinvStop stop;
    const double *xsum;        // skipped tiles' share (nullptr: none)
    const long long *xcnt;
    int NB, K;
    unsigned tag;
    int64_t member0;
};

#ifdef XINV_AUX_KERNELS
__global__ __launch_bounds__(256) void k_norm_reduce_lag(NormLagArgs a)
{
    const int64_t m = a.member0 + blockIdx.x;
    XinvCtl *ctl = a.ctl + m;
    if (xinv_ctl_done(ctl)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long *pw = a.psum + (size_t)m * XINV_KMAX * a.NB * XINV_PW;
    const double xs = a.xsum ? a.xsum[m] : 0.0;
    const long long xc = a.xcnt ? a.xcnt[m] : 0;
    switch (a.K) {
    case 1: xinv_norm_reduce<1, 4>(wave, lane, a.NB, a.tag, pw, ctl, a.stop, xs, xc); break;
    case 2: xinv_norm_reduce<2, 4>(wave, lane, a.NB, a.tag, pw, ctl, a.stop, xs, xc); break;
    case 3: xinv_norm_reduce<3, 4>(wave, lane, a.NB, a.tag, pw, ctl, a.stop, xs, xc); break;
    default: xinv_norm_reduce<4, 4>(wave, lane, a.NB, a.tag, pw, ctl, a.stop, xs, xc); break;
    }
}

#endif /* XINV_AUX_KERNELS */

#ifndef XINV_MINWAVES
#define XINV_MINWAVES 1
#endif
// SEAM (periodic x with ODD xc; strips are never aligned then).  Columns 0 and xc-1 are neighbours of the same colour; the
// coloured ordering for this case (oracle: seq_colour) updates column xc-1 inside the half-sweep of its own colour, right
// after column 0.  The row is laid out as an even ring with a phantom column (RING above): the half-sweeps that update the
// .x slots leave the seam lanes out of their pass and run one more for them alone, after which the phantom column mirrors
// column xc-1 again; tiles that hold no seam lane run the plain march.  With coefficient arrays that vary along x the
// phantom slot's COEFFICIENTS are read from column 0: the models take a point's east coefficient from the lane's own .y,
// which is then right for column xc-1 (nothing else reads them: the phantom column is never updated).
// (Round 4 classed the lanes by wrap -- the slots' parity flipped beyond one -- and ran two or three lane-masked passes.)
// The planner admits xc >= 64.
template <class M, int K, bool AL, unsigned UM, bool EXT, int PFD = 0, bool SEAM = false>
__global__ __launch_bounds__(256, XINV_MINWAVES) void k_fused2d(FusedArgs a)
{
    static_assert(!SEAM || !AL, "odd xc: strips are never aligned");
    constexpr int NC = M::NC;
    constexpr int H = 2 * K;            // halo (rows and columns) consumed by K sweeps
    // (columns owned by one wavefront: 128 - 2 H; SEAM: the ring layout's strips, xinv_tiles.h)
    constexpr int D = 2 * K + 2;        // rows held in the register window
    constexpr int PF = PFD > 0 ? PFD : 2;        // rows in flight
    static_assert(D % PF == 0, "prefetch depth must divide the window depth");

    const int64_t m = a.member0 + blockIdx.y;
    XinvCtl *ctl = a.ctl + m;
    if (!a.force && xinv_ctl_done(ctl)) return;
    if (a.lag && (int)blockIdx.x == a.nwg) { xinv_lag_reduce_prev(a, ctl, m); return; }
    const unsigned tag = a.lag ? a.tag : xinv_ctl_seq(ctl);

    // ---- tile of this wavefront; workgroup -> tile map keeps each XCD on a band of rows ----
    // A member's wave-tiles are numbered strip-fastest, then row block; workgroup T takes four
    // consecutive ones (so narrow grids do not leave wavefronts idle).
    const int NB = a.nwg;
    int T;
    {
        const int L = blockIdx.x, q = NB >> 3, rem = NB & 7, xcd = L & 7, idx = L >> 3;
        T = xcd * q + (xcd < rem ? xcd : rem) + idx;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int wt = T * 4 + wave;
    bool active = wt < a.nstrip * a.nrb;
    if constexpr (SEAM) {                            // (the edge strips' tiles first, four to a workgroup: xinv_heavy_first)
        if (!a.tile_list) {
            const int nh = (a.nstrip == 1 ? 1 : 2) * a.nrb;
            wt = xinv_heavy_first((int)blockIdx.x, NB, nh >> 2) * 4 + wave;
            active = wt < a.nstrip * a.nrb;
            wt = xinv_seam_tile(active ? wt : 0, a.nstrip, a.nrb);
        }
    }
    if (a.tile_list) {
        wt = a.tile_list[m * a.ntl + wt];
        active = wt >= 0;
        wt = active ? wt : 0;
    }
    int rb = wt / a.nstrip, strip = wt - rb * a.nstrip;
    const int64_t xc = a.xc, yc = a.yc;
    const row_t ycr = (row_t)yc;
    // rows owned by this tile: fixed height RY, or (RY == 0) the yc rows split evenly over the
    // nrb row blocks, boundaries rounded to even rows
    row_t yu0, yu1;
    if (a.RY > 0) {
        yu0 = (row_t)rb * a.RY;
        yu1 = (yu0 + a.RY < ycr) ? yu0 + a.RY : ycr;
    } else {
        yu0 = (row_t)((((int64_t)rb * yc) / a.nrb) & ~(int64_t)1);
        yu1 = (rb + 1 == a.nrb) ? ycr : (row_t)((((int64_t)(rb + 1) * yc) / a.nrb) & ~(int64_t)1);
    }
    const double u = a.sc_.undef;
    const int UW = SEAM ? xinv_ring_uw(xc, H) : 128 - 2 * H, HW = SEAM ? xinv_ring_hw(xc, H, strip) : H;
    const int64_t xu0 = (int64_t)strip * UW;

    RingSeam rs = {0ull, false};
    LaneCols lc;
    if constexpr (SEAM) lc = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
    else lc = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - HW + 2 * lane;         // unwrapped store column of .x
    const bool seam_x = SEAM && (lc.l0 == xc - 1);   // .x holds column xc-1 (its .y is the phantom column)
    LaneCols lcc = lc;                               // coefficient loads: the phantom slot reads column 0's (see SEAM above)
    if (seam_x) lcc.l1 = 0;

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *cp[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) cp[q] = a.c[q] + m * a.sc[q];

    double acc[K];
    int cnt[K];
#pragma unroll
    for (int s = 0; s < K; s++) { acc[s] = 0.0; cnt[s] = 0; }

    // SEAM: only a tile that holds a seam lane marches with the extra pass; every other tile of the launch runs the
    // plain march (with the passes behind wave-uniform branches inside ONE march every tile lost its instruction
    // interleaving: 3601 columns took 1.9x the time of 3600 -- profiles/r05_seam_rates.txt)
    auto march = [&](auto smtag) {
        constexpr bool SM = decltype(smtag)::value;
        (void)SM;
        // rows are requested in increasing order: keep the clamped row offset incrementally
        row_t lrow = yu0 - H;
        int64_t loff = (int64_t)(lrow < 0 ? 0 : (lrow > ycr - 1 ? ycr - 1 : lrow)) * xc;
        auto load = [&](row_t r) {
            RowPack<NC> p;
            const int64_t off = loff;
            loff += (lrow >= 0 && lrow < ycr - 1) ? xc : 0;
            lrow += 1;
            (void)r;
            p.s = ld2<AL>(srcS, off, lc);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                if ((UM >> q) & 1u) {
                    double t = cp[q][off];               // scalar load: one value per row
                    p.cs[q] = t; p.c[q] = make_double2(0.0, 0.0);
                }
                else                { p.c[q] = ld2<AL>(cp[q], off, lcc); p.cs[q] = 0.0; }
            }
            return p;
        };

        // Register window: row r lives in slot (r - r0) mod D.  The march is unrolled D steps so
        // that every slot index below is a compile-time constant: rows never move between
        // registers, a new row simply overwrites the slot of the row that left the pipeline.
        double2 sw[D];
        CoefWin<NC, D> cw;
#pragma unroll
        for (int t = 0; t < D; t++) {
            sw[t] = make_double2(0.0, 0.0);
            cw.rq[t] = 0.0; cw.rqv[t] = make_double2(0.0, 0.0); cw.mx[t] = 0u; cw.my[t] = 0u;
#pragma unroll
            for (int q = 0; q < NC; q++) { cw.v[q][t] = make_double2(0.0, 0.0); cw.s[q][t] = 0.0; }
        }

        // row r (= rbase + U) enters slot U: the prefetched pack is consumed here, so that its
        // registers can be re-loaded BEFORE this step's arithmetic (prefetch distance = PF steps)
        auto enter = [&](const RowPack<NC> &p, auto utag) {
            constexpr int U = decltype(utag)::value;
            sw[U] = p.s;
#pragma unroll
            for (int q = 0; q < NC; q++) { cw.v[q][U] = p.c[q]; cw.s[q][U] = p.cs[q]; }
        };

        // one pipeline step on the window whose newest row r sits in slot U
        auto step = [&](row_t r, auto utag) {
            constexpr int U = decltype(utag)::value;
            constexpr int X = (U & 1) ? 0 : 1;       // r even -> .y ; all stages of a step share it
#define SLOT(w) ((U - (w) + 4 * D) % D)              /* slot of row r - w */
            {   // row r-1 is complete in the window: its predicate, once for all 2K half-sweeps
                const bool rv = (r - 1 >= 1) && (r - 1 <= ycr - 2);
                M::template derive<UM, D>(cw, U, SLOT(1), rv && lc.ok_x, rv && lc.ok_y, a.sc_);
            }
            // SM marches: a half-sweep on the .x slots leaves the seam lanes out of its pass and updates them alone behind it
            // (east operand: the next lane's .x, the new column 0); then the phantom column mirrors column xc-1 again
            auto seam_half = [&](int sj, int sjp, int sjm) {
                double w, e;
                row_neighbours<X>(sw[sj], w, e);
                const double old = comp<X>(sw[sj]);
                double v = M::template upd<X, UM, D>(cw, sj, sjp, old, comp<X>(sw[sjp]), comp<X>(sw[sjm]), w, e, a.sc_);
                if constexpr (X == 0) {
                    sw[sj].x = seam_x ? old : v;
                    e = xinv_lane_down(sw[sj].x);
                    v = M::template upd<X, UM, D>(cw, sj, sjp, old, comp<X>(sw[sjp]), comp<X>(sw[sjm]), w, e, a.sc_);
                    sw[sj].x = seam_x ? v : sw[sj].x;
                    sw[sj].y = seam_x ? v : sw[sj].y;
                } else {
                    sw[sj].y = v;
                }
            };
            (void)seam_half;
#pragma unroll
            for (int s = 1; s <= K; s++) {
                {   // red half-sweep of sweep s on row ja = r-2s+1
                    const row_t ja = r - 2 * s + 1;
                    const int sj = SLOT(2 * s - 1), sjp = SLOT(2 * s - 2), sjm = SLOT(2 * s);
                    if (EXT) {           // BCy == 'extend': compiled out otherwise
                        if (ja == 1) fused_extend_fix(sw[sjm], sw[sj], lc, a.tall, u);
                        if (ja == ycr - 2) fused_extend_fix(sw[sjp], sw[sj], lc, a.tall, u);
                    }
                    if constexpr (!SM) {
                    double w, e;
                    row_neighbours<X>(sw[sj], w, e);
                    const double v = M::template upd<X, UM, D>(cw, sj, sjp, comp<X>(sw[sj]),
                                                               comp<X>(sw[sjp]), comp<X>(sw[sjm]),
                                                               w, e, a.sc_);
                    setc<X>(sw[sj], v);
                    } else {
                        seam_half(sj, sjp, sjm);
                    }
                }
                {   // black half-sweep of sweep s on row jb = r-2s
                    const row_t jb = r - 2 * s;
                    const int sj = SLOT(2 * s), sjp = SLOT(2 * s - 1), sjm = SLOT(2 * s + 1);
                    if constexpr (!SM) {
                    double w, e;
                    row_neighbours<X>(sw[sj], w, e);
                    const double v = M::template upd<X, UM, D>(cw, sj, sjp, comp<X>(sw[sj]),
                                                               comp<X>(sw[sjp]), comp<X>(sw[sjm]),
                                                               w, e, a.sc_);
                    setc<X>(sw[sj], v);
                    } else {
                        seam_half(sj, sjp, sjm);
                    }
                    // row jb now holds sweep s: its share of mean|S| (branch-free)
                    if ((jb >= yu0) && (jb < yu1)) {           // wave-uniform: an owned row
                        const double2 t = sw[sj];
                        const bool cx = lc.use_x & (t.x != u);
                        const bool cy = lc.use_y & (t.y != u);
                        acc[s - 1] += (cx ? fabs(t.x) : 0.0);
                        acc[s - 1] += (cy ? fabs(t.y) : 0.0);
                        cnt[s - 1] += (cx ? 1 : 0) + (cy ? 1 : 0);
                    }
                }
            }
            const row_t jo = r - 2 * K;                  // row leaving the pipeline
            if (jo >= yu0 && jo < yu1) {
                const double2 t = sw[SLOT(2 * K)];
                if (AL) {
                    if (lc.use_x) *reinterpret_cast<double2 *>(dstS + (int64_t)jo * xc + st0) = t;
                } else {
                    if (lc.use_x) dstS[(int64_t)jo * xc + st0] = t.x;
                    if (lc.use_y) dstS[(int64_t)jo * xc + st0 + 1] = t.y;
                }
            }
#undef SLOT
        };

        // Prefetch ring: PF rows in flight per wavefront (PF divides D so ring indices are
        // compile-time constants).  Bytes in flight, not issue rate, bound this kernel: a wave
        // keeps PF x (1 + vector streams) KiB outstanding.
        const row_t r0 = yu0 - H;                        // even: RY and H are even
        const row_t rlast = yu1 - 1 + H;
        RowPack<NC> pf[PF];
#pragma unroll
        for (int t = 0; t < PF; t++) pf[t] = load(r0 + t);
        for (row_t rb_ = r0; rb_ <= rlast; rb_ += D) {
            xinv_unroll_steps([&](auto utag) {
                constexpr int U = decltype(utag)::value;
                enter(pf[U % PF], utag);
                step(rb_ + U, utag);
                pf[U % PF] = load(rb_ + U + PF);
            }, std::make_integer_sequence<int, D>{});
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

#ifdef XINV_AUX_KERNELS   /* non-template helper kernels: compiled into the main translation unit only */
// ---- masked-tile skipping: activity map and the skipped tiles' share of the norm -------------
// act[m][row][strip] = 1 when the forcing has a defined point in that row of that strip's owned
// columns.  Every mask predicate of the reference tests the forcing (numbas.py:344, 1126, 530),
// so a tile without such a point can never change: a superset of the updatable tiles.
struct StripActArgs {
    const double *f;           // forcing (last coefficient array)
    int64_t sf;                // its batch stride (0 = shared)
    int64_t yc, xc;
    int nstrip, UW;
    double undef;
    unsigned char *act;        // [nbatch][yc][nstrip]
};

__global__ __launch_bounds__(256) void k_strip_active(StripActArgs a)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t cell = (int64_t)blockIdx.x * 4 + wave;       // (row, strip) of member blockIdx.y
    if (cell >= a.yc * a.nstrip) return;
    const int64_t row = cell / a.nstrip;
    const int strip = (int)(cell - row * a.nstrip);
    const double *f = a.f + (int64_t)blockIdx.y * a.sf + row * a.xc;
    const int64_t c0 = (int64_t)strip * a.UW;
    const int64_t c1 = (c0 + a.UW < a.xc) ? c0 + a.UW : a.xc;
    bool any = false;
    for (int64_t i = c0 + lane; i < c1; i += 64) any |= (f[i] != a.undef);
    const bool w = __ballot(any) != 0ull;
    if (lane == 0) a.act[((int64_t)blockIdx.y * a.yc + row) * a.nstrip + strip] = w ? 1 : 0;
}

// One wavefront per skipped tile: sum |S| and count over its owned points with S != undef.
struct SkipNormArgs {
    const double *S;
    int64_t sS, yc, xc;
    int nstrip, nrb, UW;
    int RB;                    // > 0: row blocks of exactly RB rows (biharmonic kernel); 0: even split
    double undef;
    const int *skip_list;      // [nbatch][nskip_max] wave-tile ids, -1 = none
    int nskip_max;
    double *tsum;              // [nbatch][nskip_max]
    long long *tcnt;
    double *xsum;              // [nbatch]
    long long *xcnt;
    unsigned *ticket;          // [nbatch], zero between solves: k_skip_tiles' arrival counter (the last block of a member sums)
};

// One wavefront per skipped tile: its sum |S| and count over its owned points with S != undef; its owned region copied
// into the buffers S rotates through (the skipped tiles are never written by the sweep launches: every buffer must hold
// the caller's S there); and the block that arrives last at the member's ticket adds the tiles' shares (fixed order)
// into xsum / xcnt and leaves the ticket at zero for the next solve.  ONE launch since round 5 (k_skip_norm_tile,
// k_skip_norm_sum and k_copy_skipped until round 4: three dependent dispatches, ~32 us + the gaps between them, on the
// path to a solve's second sweep launch; same loads, same order of additions, same bits).
__global__ __launch_bounds__(64) void k_skip_tiles(SkipNormArgs a, double *D1, double *D2)
{
    const int lane = threadIdx.x;
    const int64_t m = blockIdx.y;
    const int wt = a.skip_list[m * a.nskip_max + blockIdx.x];
    double acc = 0.0; long long cnt = 0;
    if (wt >= 0) {
        const TileRows tr = xinv_tile_rows(wt, a.nstrip, a.nrb, a.yc, a.RB);
        const int64_t yu0 = tr.y0, yu1 = tr.y1;
        const int64_t c0 = (int64_t)tr.strip * a.UW;
        const int64_t c1 = (c0 + a.UW < a.xc) ? c0 + a.UW : a.xc;
        const double *S = a.S + m * a.sS;
        double *d1 = D1 + m * a.sS, *d2 = D2 ? D2 + m * a.sS : nullptr;
        for (int64_t j = yu0; j < yu1; j += 4) {
            double v[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t jj = (j + q < yu1) ? j + q : yu1 - 1;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const int64_t i = c0 + lane + 64 * h;
                    v[q][h] = (i < c1) ? S[jj * a.xc + i] : a.undef;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (j + q < yu1) {
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const int64_t i = c0 + lane + 64 * h;
                        if (i < c1) { d1[(j + q) * a.xc + i] = v[q][h]; if (d2) d2[(j + q) * a.xc + i] = v[q][h]; }
                        if (v[q][h] != a.undef) { acc += fabs(v[q][h]); cnt++; }
                    }
                }
        }
    }
    acc = xinv_wave_sum(acc);
    cnt = xinv_wave_sum_ll(cnt);
    __shared__ unsigned s_last;
    if (lane == 0) {
        a.tsum[m * a.nskip_max + blockIdx.x] = acc; a.tcnt[m * a.nskip_max + blockIdx.x] = cnt;
        __threadfence();                                   // (the shares are visible before the ticket counts this block)
        const unsigned t = atomicAdd(a.ticket + m, 1u);
        s_last = (t == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double ps = 0.0; long long pc = 0;
    for (int t = lane; t < a.nskip_max; t += 64) {
        ps += __hip_atomic_load(a.tsum + m * a.nskip_max + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pc += __hip_atomic_load(a.tcnt + m * a.nskip_max + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ps = xinv_wave_sum(ps);
    pc = xinv_wave_sum_ll(pc);
    if (lane == 0) { a.xsum[m] = ps; a.xcnt[m] = pc; a.ticket[m] = 0u; }
}

// ---- detection of x-uniform coefficient rows (once per solve) -----------------------------
// flag[q] |= 1 when array q has a row whose elements are not all bitwise equal to its first.
struct XUniArgs {
    const double *c[10];
    int64_t stride[10];        // batch stride; 0 = shared (checked once)
    int64_t nbatch, yc, xc;
    int nstream;
    int *flag;                 // [16]
};

__global__ __launch_bounds__(256) void k_xuniform(XUniArgs a)
{
    const int q = blockIdx.y;
    if (q >= a.nstream) return;
    volatile int *fl = a.flag + q;
    const int64_t nm = (a.stride[q] == 0) ? 1 : a.nbatch;
    const int64_t nrow = nm * a.yc;
    const unsigned long long *base = reinterpret_cast<const unsigned long long *>(a.c[q]);
    bool bad = false;
    // one wavefront per row at a time, grid-stride over rows
    const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * wpb + wave; row < nrow; row += (int64_t)gridDim.x * wpb) {
        if (*fl) break;                                  // someone already found a varying row
        const int64_t mm = row / a.yc, j = row - mm * a.yc;
        const unsigned long long *p = base + mm * a.stride[q] + j * a.xc;
        const unsigned long long first = p[0];
        // (four loads in flight per lane: one wavefront walks a row, and the loop was a chain of memory round trips --
        //  53 us per solve for the two 52 MB coefficient arrays of a 3600x1800 lat-lon Poisson problem)
        //  round 4: 47 us still -- a row of 3600 was fourteen dependent rounds of four 8-byte loads; 16-byte loads where
        //  the row allows, eight in flight)
        if (((a.xc & 1) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            const ulonglong2 *p2 = reinterpret_cast<const ulonglong2 *>(p);
            const int64_t n2 = a.xc >> 1;
            for (int64_t i = lane; i < n2; i += 512) {
                ulonglong2 v[8];
#pragma unroll
                for (int h = 0; h < 8; h++) v[h] = (i + 64 * h < n2) ? p2[i + 64 * h] : make_ulonglong2(first, first);
#pragma unroll
                for (int h = 0; h < 8; h++) bad |= (v[h].x != first) | (v[h].y != first);
            }
        } else
        for (int64_t i = lane; i < a.xc; i += 256) {
            unsigned long long v[4];
#pragma unroll
            for (int h = 0; h < 4; h++) v[h] = (i + 64 * h < a.xc) ? p[i + 64 * h] : first;
#pragma unroll
            for (int h = 0; h < 4; h++) bad |= (v[h] != first);
        }
        if (__any(bad)) break;
    }
    if (__any(bad) && lane == 0) atomicOr(a.flag + q, 1);
}
#endif /* XINV_AUX_KERNELS */
