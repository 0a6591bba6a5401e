// xinv_device.h -- device-side helpers shared by the colour-pass and fused kernels (gfx950).
//
// Point arithmetic follows the reference expression by expression (reference
// xinvert/numbas.py:343-369, 1125-1153, 146-169); the translation unit is compiled with
// -ffp-contract=off so that no FMA is formed and results are bitwise those of the
// CPU restatement of the same ordering.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>

#define XINV_WAVE 64

// The scalar data cache is not reliably invalidated between kernels: on the first launch of a kernel variant
// k_pipe2d relaxed a whole slice with the previous solve's row factors (read with s_load from a reused workspace
// address) in 4 of 14 runs of the GPU suite, 0 of 14 with an explicit invalidate (profiles/r02_pipe2d_bringup.txt).
// So: a kernel that reads solver state through the scalar unit on purpose starts with xinv_fresh_scalar_cache()
// (k_pipe2d only: invalidating in every workgroup of every kernel cost 3-7 % on the many-round launches, C4 64
// members 3.94 -> 3.67e11), and the two words every kernel reads of state written by its predecessor -- the stop
// flag and the sequence number, at a uniform address, hence an s_load if left to the compiler -- are read with
// agent-scope atomic loads (vector loads through the coherent L2; a stale `done` would silently skip a pass).
// The invalidate returns on the LGKM counter like a scalar load: wait for it before anything else is issued
// (the compiler barrier keeps later loads below it; tools/smem_audit.py checks the order in the binary).
__device__ __forceinline__ void xinv_fresh_scalar_cache()
{
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// One control block per batch member, resident in HBM for the whole solve.  Written only by
// the reducing workgroup of a sweep launch (fused path) or by k_norm_final (colour path);
// read by every later launch (`done` => the launch is a no-op for that member) and by the host
// every `check_every` launches.  Mirrors the reference's loop variables (numbas.py:278-283,
// 401-414): loop, normPrev, flags[1], flags[2], overflow.
struct XinvCtl {
    double normPrev;
    double flag1;          // last relative change of mean|S|
    double flag2;          // last loop index
    long long loop;
    long long sweeps;      // valid once done: sweeps the returned S must contain (= loop + 1)
    int done;
    int overflow;
    int wrote;             // flag1/flag2 have been written at least once
    unsigned ticket;       // (spare)
    unsigned seq;          // fused kernels: sequence number of the next launch (tags its norm partials)
    unsigned pad_;
};

struct XinvStop {
    long long mxLoop;
    double tolerance;
    int stop_on_zero_norm;  // standard_2D only (numbas.py:410)
};

__device__ __forceinline__ int xinv_ctl_done(const XinvCtl *c)
{
    return __hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned xinv_ctl_seq(const XinvCtl *c)
{
    return __hip_atomic_load(&c->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// numbas.py:401-414 / 1186-1199 / 197-210, one sweep's worth.
__device__ __forceinline__ void xinv_ctl_update(XinvCtl *c, double sum, long long count,
                                                const XinvStop &st)
{
    if (c->done) return;
    double norm = (count != 0) ? sum / (double)count : NAN;
    if (isnan(norm) || norm > 1e100) {
        c->overflow = 1;
        c->done = 1;
        c->sweeps = c->loop + 1;
        return;
    }
    c->flag1 = fabs(norm - c->normPrev) / c->normPrev;
    c->flag2 = (double)c->loop;
    c->wrote = 1;
    if (c->flag1 < st.tolerance || c->loop >= st.mxLoop ||
        (st.stop_on_zero_norm && norm == 0.0)) {
        c->done = 1;
        c->sweeps = c->loop + 1;
        return;
    }
    c->normPrev = norm;
    c->loop += 1;
}

// Deterministic sum over the 64 lanes of a wavefront, returned in every lane.  DPP row shifts
// (1, 2, 4, 8 inside the rows of 16 lanes, zero shifted in), then row_bcast:15 / row_bcast:31 carry
// the row totals across: the total lands in lane 63 and is broadcast with v_readlane.  18 VALU
// instructions per double instead of six dependent ds_bpermute round trips (__shfl_xor): the norm
// partials of K fused sweeps are reduced at the very end of every wavefront, on the critical path.
// The order of the additions is fixed (run-to-run reproducible), though not the butterfly's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double xinv_dpp_add(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    // lanes the row mask leaves out (or that shift in nothing) receive +0.0: v + 0.0 == v
    return v + __hiloint2double(hi, lo);
}

__device__ __forceinline__ double xinv_wave_sum(double v)
{
    v = xinv_dpp_add<0x111, 0xf>(v);       // row_shr:1
    v = xinv_dpp_add<0x112, 0xf>(v);       // row_shr:2
    v = xinv_dpp_add<0x114, 0xf>(v);       // row_shr:4
    v = xinv_dpp_add<0x118, 0xf>(v);       // row_shr:8   -> lane 15 of each row holds the row total
    v = xinv_dpp_add<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v = xinv_dpp_add<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long xinv_dpp_add_ll(long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned long long)v, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)((unsigned long long)v >> 32), CTRL, ROW_MASK, 0xf, true);
    return v + (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ long long xinv_wave_sum_ll(long long v)
{
    v = xinv_dpp_add_ll<0x111, 0xf>(v);
    v = xinv_dpp_add_ll<0x112, 0xf>(v);
    v = xinv_dpp_add_ll<0x114, 0xf>(v);
    v = xinv_dpp_add_ll<0x118, 0xf>(v);
    v = xinv_dpp_add_ll<0x142, 0xa>(v);
    v = xinv_dpp_add_ll<0x143, 0xc>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Neighbour-lane moves on the VALU (DPP wave shifts, gfx9 family): no LDS round trip, unlike
// __shfl_up/__shfl_down (ds_bpermute_b32).  Edge lanes keep their own value, as __shfl does.
// Verified on gfx950: wave_shr:1 == __shfl_up(v, 1), wave_shl:1 == __shfl_down(v, 1).
__device__ __forceinline__ double xinv_lane_up(double v)       // lane i <- lane i-1
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl: the edge lane reads 0 instead of keeping its own value, so the destination
    // needs no prior copy of the source (one v_mov_b32 less per half).  Edge lanes are halo.
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);     // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double xinv_lane_down(double v)     // lane i <- lane i+1
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);     // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// a + b in the lanes of `mask`, a in the others: the add runs with the mask as EXEC -- two scalar instructions around one
// vector instruction, where `cond ? a + b : a` costs the add and two v_cndmask_b32 on a unit that is the bound of
// the pipelined pass (the same bits: a masked lane keeps a).  (+|b| for the norm share: a lane that does not count
// adds nothing instead of +0.0 -- the accumulator is a sum of magnitudes, never -0.0, so the same bits again.)
__device__ __forceinline__ double xinv_add_where(double a, double b, unsigned long long mask)
{
    unsigned long long sv;
    asm("s_and_saveexec_b64 %1, %3\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, %1"
        : "+v"(a), "=&s"(sv) : "v"(b), "s"(mask) : "scc");
    return a;
}
__device__ __forceinline__ double xinv_add_abs_where(double a, double b, unsigned long long mask)
{
    unsigned long long sv;
    asm("s_and_saveexec_b64 %1, %3\n\tv_add_f64 %0, %0, |%2|\n\ts_mov_b64 exec, %1"
        : "+v"(a), "=&s"(sv) : "v"(b), "s"(mask) : "scc");
    return a;
}

// The same with the compare folded in: a + b in the lanes of `rowmask` whose f differs from u (the reference's
// `F[j,i] != undef` of the update predicate), a in the others -- v_cmpx writes the compare's result into EXEC, so
// the predicate costs no scalar logic at all: s_and_saveexec, v_cmpx, v_add, s_mov.
__device__ __forceinline__ double xinv_add_where_ne(double a, double b, double f, double u, unsigned long long rowmask)
{
    unsigned long long sv;
    asm("s_and_saveexec_b64 %1, %5\n\tv_cmpx_neq_f64_e32 vcc, %4, %3\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, %1"
        : "+v"(a), "=&s"(sv) : "v"(b), "v"(f), "s"(u), "s"(rowmask) : "scc", "vcc");
    return a;
}
// the same with the contracted finish of XINV_FLAG_FMA: a = fma(t, rq, a) in those lanes (rq: the row's relaxation
// factor, wave-uniform, read through the scalar unit)
__device__ __forceinline__ double xinv_fma_where_ne(double a, double t, double rq, double f, double u, unsigned long long rowmask)
{
    unsigned long long sv;
    asm("s_and_saveexec_b64 %1, %6\n\tv_cmpx_neq_f64_e32 vcc, %5, %4\n\tv_fma_f64 %0, %2, %3, %0\n\ts_mov_b64 exec, %1"
        : "+v"(a), "=&s"(sv) : "v"(t), "s"(rq), "v"(f), "s"(u), "s"(rowmask) : "scc", "vcc");
    return a;
}
// a row's share of mean|S|: sum += |x| and n += 1 in the lanes whose x differs from u, for both columns of the lane
// (per-lane accumulators; the caller discards the lanes that do not own their column).  Nine instructions.
__device__ __forceinline__ void xinv_norm_row(double &sx, double &sy, int &nx, int &ny, double x, double y, double u)
{
    unsigned long long sv;
    asm("s_mov_b64 %4, exec\n\t"
        "v_cmpx_neq_f64_e32 vcc, %7, %5\n\tv_add_f64 %0, %0, |%5|\n\tv_add_u32 %2, %2, 1\n\ts_mov_b64 exec, %4\n\t"
        "v_cmpx_neq_f64_e32 vcc, %7, %6\n\tv_add_f64 %1, %1, |%6|\n\tv_add_u32 %3, %3, 1\n\ts_mov_b64 exec, %4"
        : "+v"(sx), "+v"(sy), "+v"(nx), "+v"(ny), "=&s"(sv) : "v"(x), "v"(y), "s"(u) : "vcc");
}

// Scalars of one solve (same for every member), passed by value to the kernels.
struct XinvScal {
    double delx, delxSqr, ratio, ratioQtr, ratioSqr;   // 2-D
    double ratio2Sqr, ratio1Sqr;                       // 3-D
    double ratio2, ratio1;                             // general 3-D
    double delxSSr, delxTr, ratioSSr;                  // biharmonic 2-D
    double optArg, undef;
};

// ---- point updates: return the new value of S at the point, or sC when masked -----------

// standard 2-D, full 9-point form (numbas.py:343-369).  `west` selects the two irregular
// operands of the reference's i == 0 periodic branch (numbas.py:327-328).
__device__ __forceinline__ double xinv_upd_std2d_9(
    double sC, double sP, double sM, double sW, double sE,
    double sPE, double sPW, double sME, double sMW, double sM_q,
    double aP, double a0, double bE, double bW, double bP_chk, double bP_use, double bM,
    double cE, double c0, double f, const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = (f != u) && (aP != u) && (a0 != u) && (bE != u) && (bW != u) &&
                (bP_chk != u) && (bM != u) && (cE != u) && (c0 != u);
    if (!cond) return sC;
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
    return sC + temp;
}

// standard 2-D with B == 0 everywhere (5-point coupling).
__device__ __forceinline__ double xinv_upd_std2d_5(
    double sC, double sP, double sM, double sW, double sE,
    double aP, double a0, double cE, double c0, double f, bool inrange, const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = inrange && (f != u) && (aP != u) && (a0 != u) && (cE != u) && (c0 != u);
    double temp = (
        (
            aP * (sP - sC) -
            a0 * (sC - sM)
        ) * sc.ratioSqr + (
            cE * (sE - sC) -
            c0 * (sC - sW)
        )
    ) - f * sc.delxSqr;
    temp *= sc.optArg / ((aP + a0) * sc.ratioSqr + (cE + c0));
    return cond ? sC + temp : sC;
}

// general 2-D, full 9-point form (numbas.py:1125-1153).
__device__ __forceinline__ double xinv_upd_gen2d_9(
    double sC, double sP, double sM, double sW, double sE,
    double sPE, double sPW, double sME, double sMW,
    double A, double B, double C, double D, double E, double F, double G, const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = (G != u) && (A != u) && (B != u) && (C != u) && (D != u) && (E != u) && (F != u);
    if (!cond) return sC;
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
    return sC + temp;
}

// general 2-D with B == 0 everywhere (5-point coupling).
__device__ __forceinline__ double xinv_upd_gen2d_5(
    double sC, double sP, double sM, double sW, double sE,
    double A, double C, double D, double E, double F, double G, bool inrange,
    const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = inrange && (G != u) && (A != u) && (C != u) && (D != u) && (E != u) && (F != u);
    double temp = (
        A * (
            (sP - sC) - (sC - sM)
        ) * sc.ratioSqr +
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

// standard 3-D, 7-point (numbas.py:146-169).  P/M = k+1/k-1, N/S = j+1/j-1, E/W = i+1/i-1.
__device__ __forceinline__ double xinv_upd_std3d(
    double sC, double sKP, double sKM, double sJP, double sJM, double sE, double sW,
    double aP, double a0, double bP, double b0, double cE, double c0, double f,
    const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = (f != u) && (aP != u) && (a0 != u) && (bP != u) && (b0 != u) &&
                (cE != u) && (c0 != u);
    if (!cond) return sC;
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
    return sC + temp;
}

// general 3-D, 7-point (numbas.py:899-930).  `testH` is false in the reference's west-periodic
// branch, which tests G twice and never H (numbas.py:849-852).
__device__ __forceinline__ double xinv_upd_gen3d(
    double sC, double sKP, double sKM, double sJP, double sJM, double sE, double sW,
    double A, double B, double C, double D, double E, double F, double G, double H, bool testH,
    const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = (!testH || H != u) && (G != u) && (A != u) && (B != u) && (C != u) && (D != u) &&
                (E != u) && (F != u);
    if (!cond) return sC;
    double temp = (
        A * (
            (sKP - sC)-(sC - sKM)
        ) * sc.ratio2Sqr +
        B * (
            (sJP - sC)-(sC - sJM)
        ) * sc.ratio1Sqr +
        C * (
            (sE - sC)-(sC - sW)
        ) + (
        D * (
            (sKP - sKM)
        ) * sc.ratio2 +
        E * (
            (sJP - sJM)
        ) * sc.ratio1 +
        F * (
            (sE - sW)
        )) * sc.delx / 2.0 + (
        G * sC - H) * sc.delxSqr
    );
    temp *= sc.optArg / ((
        A*sc.ratio2Sqr + B*sc.ratio1Sqr + C
    ) * 2.0 - G*sc.delxSqr);
    return sC + temp;
}

// biharmonic 2-D (numbas.py:1437-1479 inner loop; 1347-1434, 1482-1570 periodic branches).
// r0/p1/p2/m1/m2 = rows j, j+1, j+2, j-1, j-2 of S.  `edge` and `bm2` carry the reference's
// branch irregularities (see oracle header): the G-term association of the periodic branches
// and the stale loop index in the east branches' B term.
__device__ __forceinline__ double xinv_upd_bih2d(
    const double *r0, const double *p1, const double *p2, const double *m1, const double *m2,
    int64_t i, int64_t im2, int64_t im1, int64_t ip1, int64_t ip2, int64_t bm2, bool edge,
    double A, double B, double C, double D, double E, double F, double G, double H, double I,
    double J, const XinvScal &sc)
{
    const double u = sc.undef;
    const double sC = r0[i];
    bool cond = (A != u) && (B != u) && (C != u) && (D != u) && (E != u) && (F != u) &&
                (G != u) && (H != u) && (I != u) && (J != u);
    if (!cond) return sC;
    double gterm = G * (
                       (p1[i] - m1[i])
                   );
    if (edge) gterm = gterm * sc.delxTr / 2.0 * sc.ratio;
    else      gterm = gterm * sc.delxTr * sc.ratio / 2.0;
    double temp = (
        A * (
            (p2[i] - 4.0*p1[i] + 6.0*r0[i] - 4.0*m1[i] + m2[i])
        ) * sc.ratioSSr +
        B * (
            (    p2[ip2] - 2.0*p2[i] +     p2[bm2] +
            -2.0*r0[ip2] + 4.0*r0[i] - 2.0*r0[bm2] +
                 m2[ip2] - 2.0*m2[i] +     m2[bm2])
        ) * sc.ratioSqr / 16.0 +
        C * (
            (r0[ip2] - 4.0*r0[ip1] + 6.0*r0[i] - 4.0*r0[im1] + r0[im2])
        ) +
        D * (
            (p1[i] - r0[i])-(r0[i] - m1[i])
        ) * sc.ratioSqr * sc.delxSqr +
        E * (
            (p1[ip1] - m1[ip1])-(p1[im1] - m1[im1])
        ) * sc.ratioQtr * sc.delxSqr +
        F * (
            (r0[ip1] - r0[i])-(r0[i] - r0[im1])
        ) * sc.delxSqr +
        gterm +
        H * (
            (r0[ip1] - r0[im1])
        ) * sc.delxTr / 2.0 + (
        I * r0[i] - J) * sc.delxSSr
    );
    temp *= -sc.optArg / ((A*sc.ratioSSr + C) * 6.0 +
                           B*sc.ratioSqr / 4.0 +
                         -(D*sc.ratioSqr + F) * 2.0 * sc.delxSqr +
                           I*sc.delxSSr);
    return sC + temp;
}

// standard 2-D "test" form (numbas.py:563-589): d/dy(A Sy + B Sx) + d/dx(C Sy + D Sx) + E S = F.
__device__ __forceinline__ double xinv_upd_std2dt_9(
    double sC, double sP, double sM, double sW, double sE,
    double sPE, double sPW, double sME, double sMW, double sM_q,
    double aP, double a0, double bP_chk, double bP_use, double bM, double cE, double cW,
    double dE, double d0, double e, double f, const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = (f != u) && (aP != u) && (a0 != u) && (bP_chk != u) && (bM != u) &&
                (cE != u) && (cW != u) && (dE != u) && (d0 != u) && (e != u);
    if (!cond) return sC;
    double temp = (
        (
            aP * (sP - sC) -
            a0 * (sC - sM)
        ) * sc.ratioSqr + (
            bP_use * (sPE - sPW) -
            bM * (sM_q - sMW)
        ) * sc.ratioQtr + (
            cE * (sPE - sME) -
            cW * (sPW - sMW)
        ) * sc.ratioQtr + (
            dE * (sE - sC) -
            d0 * (sC - sW)
        )
    ) + (e * sC - f) * sc.delxSqr;
    temp *= sc.optArg / ((aP + a0) * sc.ratioSqr +
                         (dE + d0) - e * sc.delxSqr);
    return sC + temp;
}

// the same with B == 0 and C == 0 everywhere (5-point coupling)
__device__ __forceinline__ double xinv_upd_std2dt_5(
    double sC, double sP, double sM, double sW, double sE,
    double aP, double a0, double dE, double d0, double e, double f, bool inrange,
    const XinvScal &sc)
{
    const double u = sc.undef;
    bool cond = inrange && (f != u) && (aP != u) && (a0 != u) && (dE != u) && (d0 != u) && (e != u);
    double temp = (
        (
            aP * (sP - sC) -
            a0 * (sC - sM)
        ) * sc.ratioSqr + (
            dE * (sE - sC) -
            d0 * (sC - sW)
        )
    ) + (e * sC - f) * sc.delxSqr;
    temp *= sc.optArg / ((aP + a0) * sc.ratioSqr +
                         (dE + d0) - e * sc.delxSqr);
    return cond ? sC + temp : sC;
}

// the biharmonic update on register operands (numbas.py:1437-1479 inner loop; 1347-1434,
// 1482-1570 periodic branches); XY naming: first letter = row (p2, p1, r0, m1, m2), suffix =
// column offset (m2, m1, 0, p1, p2).  p2_b / r0_b / m2_b are the west operands of the B term: column
// i-2, except in the reference's two east-periodic branches where the stale loop index makes it
// column i-5; `edge` selects the G-term association of the periodic branches.
__device__ __forceinline__ double xinv_upd_bih2d_v(
    double p2_0, double p2_p2, double p2_b, double p1_0, double p1_p1, double p1_m1,
    double r0_0, double r0_p1, double r0_m1, double r0_p2, double r0_m2, double r0_b,
    double m1_0, double m1_p1, double m1_m1, double m2_0, double m2_p2, double m2_b,
    double A, double B, double C, double D, double E, double F, double G, double H, double I,
    double J, bool inr, bool edge, const XinvScal &sc)
{
    const double u = sc.undef;
    const bool cond = inr && (A != u) && (B != u) && (C != u) && (D != u) && (E != u) &&
                      (F != u) && (G != u) && (H != u) && (I != u) && (J != u);
    double gterm = G * (
                       (p1_0 - m1_0)
                   );
    gterm = edge ? gterm * sc.delxTr / 2.0 * sc.ratio : gterm * sc.delxTr * sc.ratio / 2.0;
    double temp = (
        A * (
            (p2_0 - 4.0*p1_0 + 6.0*r0_0 - 4.0*m1_0 + m2_0)
        ) * sc.ratioSSr +
        B * (
            (    p2_p2 - 2.0*p2_0 +     p2_b +
            -2.0*r0_p2 + 4.0*r0_0 - 2.0*r0_b +
                 m2_p2 - 2.0*m2_0 +     m2_b)
        ) * sc.ratioSqr / 16.0 +
        C * (
            (r0_p2 - 4.0*r0_p1 + 6.0*r0_0 - 4.0*r0_m1 + r0_m2)
        ) +
        D * (
            (p1_0 - r0_0)-(r0_0 - m1_0)
        ) * sc.ratioSqr * sc.delxSqr +
        E * (
            (p1_p1 - m1_p1)-(p1_m1 - m1_m1)
        ) * sc.ratioQtr * sc.delxSqr +
        F * (
            (r0_p1 - r0_0)-(r0_0 - r0_m1)
        ) * sc.delxSqr +
        gterm +
        H * (
            (r0_p1 - r0_m1)
        ) * sc.delxTr / 2.0 + (
        I * r0_0 - J) * sc.delxSSr
    );
    temp *= -sc.optArg / ((A*sc.ratioSSr + C) * 6.0 +
                           B*sc.ratioSqr / 4.0 +
                         -(D*sc.ratioSqr + F) * 2.0 * sc.delxSqr +
                           I*sc.delxSSr);
    return cond ? r0_0 + temp : r0_0;
}

// three adjacent columns of one row held by a lane (biharmonic kernels: the three column colours)
struct Tri { double v[3]; };

__device__ __forceinline__ void bih_ext(const Tri &t, double (&e)[7])   // e[k+2] = column c+k, k=-2..4
{
    e[2] = t.v[0]; e[3] = t.v[1]; e[4] = t.v[2];
    e[1] = xinv_lane_up(t.v[2]); e[0] = xinv_lane_up(t.v[1]);
    e[5] = xinv_lane_down(t.v[0]); e[6] = xinv_lane_down(t.v[1]);
}
// columns c-4 and c-3: the stale-index operands of components 1 and 2
__device__ __forceinline__ void bih_far(const Tri &t, double &cm4, double &cm3)
{
    cm4 = xinv_lane_up(xinv_lane_up(t.v[2]));
    cm3 = xinv_lane_up(t.v[0]);
}

