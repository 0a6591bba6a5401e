// xinv_pipe2d.h -- wave-pipelined streaming red-black SOR pass: four sweeps per pass over HBM, one sweep
// per WAVEFRONT (gfx950).  Standard form with per-row A and C (lat-lon Poisson) and general form with per-row
// A, C, D, E, F (lat-lon Gill-Matsuno), B == 0: the model M supplies the update and its per-row record.
//
// k_fused2d applies K sweeps inside one wavefront; its tile count is tied to the number of wavefront slots
// (256 CUs x 4 SIMDs x 2), so at 3600x1800 a tile owns ~23 rows and marches 40: half of the point updates
// it executes are recomputed halo, and the kernel is bound by the fp64 VALU.  Here a workgroup of four
// wavefronts owns ONE tile (128 columns x RY rows) and the sweeps are pipelined ACROSS the wavefronts:
// wavefront p applies sweep p+1 to the rows as they stream by and hands every finished row to wavefront
// p+1 through a ring of rows in LDS.  The same number of wavefronts now needs a quarter of the tiles, each
// four times as tall (~90 rows at 3600x1800), so the 2K halo rows per side weigh 1.2 instead of 1.7 -- and
// because every wavefront runs its own loop, sweep s only marches the rows that can still reach an owned
// row (RY + 2(2K - 2(s-1)) entering rows: the triangle k_fused2d could not skip without breaking its
// schedule falls away by construction).
//
// Schedule.  With r the row that enters wavefront p's four-row register window at a step, the wavefront
// updates its red points on row r-1 and its black points on row r-2 (sweep p+1), and row r-2 leaves: into
// the ring (p < 3) or to HBM (p = 3, owned rows only).  Wavefront p+1 asks for that row one step later and
// enters it the step after (LDS latency hidden behind a step of arithmetic), so it runs LAG = 6 steps
// behind wavefront p.  One `s_waitcnt lgkmcnt(0); s_barrier` per step keeps the four in step (no vmcnt
// wait: the global prefetch of wavefront 0 stays in flight across barriers).  Wavefront 0 reads S and every
// wavefront reads the forcing rows it needs from global memory (the later ones hit in L2) PF rows ahead.
//
// The relaxation factor optArg / ((A[j+1] + A[j]) ratioSqr + 2 C[j]) is uniform along a row and the same
// for every sweep of the solve: k_row_factor evaluates it once per solve (same expression, same bits) with
// the row's share of the update predicate, and the kernel loads it per row instead of dividing.
//
// Operands and operation order per point are those of k_fused2d (FusedStd2D::upd): results are bitwise
// equal to it and to the oracle.  Norm partials: wavefront p publishes the tile's share of sweep p+1.
#pragma once
#include "xinv_fused.h"

#define XINV_PIPE_P 4             /* wavefronts per workgroup = sweeps per pass */
#ifndef XINV_PIPE_B
#define XINV_PIPE_B 2             /* steps per workgroup barrier (1 or 2; the four-row LDS ring allows no more) */
#endif
#define XINV_PIPE_LAG (XINV_PIPE_B + 5)   /* steps wavefront p+1 runs behind wavefront p */
#define XINV_PIPE_NS 4            /* ring rows per hand-over */
#define XINV_PIPE_UW(np) (128 * (np) - 4 * XINV_PIPE_P)   /* columns a tile owns with np column pairs per lane */
#ifndef XINV_PIPE_PF0
#define XINV_PIPE_PF0 4           /* rows in flight from HBM, wavefront 0 (S and F) */
#endif
#ifndef XINV_PIPE_PF
#define XINV_PIPE_PF 4            /* rows of F in flight, wavefronts 1..3 */
#endif

// What a row needs besides S and the vector streams: one record per row, read through the scalar unit --
// the values of the coefficient streams that are constant along x (bit q of UM), in stream order, then, when the
// model's denominator is x-uniform (M::hoist<UM>()), the row's relaxation factor and the row part of the update
// predicate (all ones / zero: all 64 bits are read, so no half of a load in flight is ever reused).  Padded to 4 or 8
// doubles.  Standard form, A and C per row: {A[j], C[j], rq, rok}; general form, A C D E F per row:
// {A, C, D, E, F, rq, rok, -}; general form, D E F per row (A, C, G streamed): {D, E, F, -}; no record when
// nothing is uniform.
template <class M, unsigned UM> struct PipeRec {
    static constexpr int NCO = M::NC - 1;                                    // coefficient streams (the forcing is last)
    static constexpr int NUNI = __builtin_popcount(UM & ((1u << NCO) - 1u));
    static constexpr bool HOIST = M::template hoist<UM>();
    static constexpr int NW = NUNI + (HOIST ? 2 : 0);
    static constexpr int RW = NW == 0 ? 0 : (NW <= 4 ? 4 : 8);
};

struct RowFactorArgs {
    const double *c[5];           // the coefficient streams in FusedArgs order (std: A, C; gen: A, C, D, E, F)
    int64_t sc[5];                // batch strides (0 = shared)
    int64_t yc, xc;
    int gen;                      // 0: standard form, 1: general form
    unsigned um;                  // which streams are per-row values
    int hoist, rw;                // record carries rq / rok; doubles per record
    XinvScal sc_;
    double *rowf;                 // [nbatch][yc][rw]
};

#ifdef XINV_AUX_KERNELS
// once per solve: the per-row records (numbas.py:343-369 / 1125-1153 with the coefficients constant along x;
// FusedStd2D::derive / FusedGen2D::derive, hoisted branch: same expressions, same bits)
__global__ __launch_bounds__(256) void k_row_factor(RowFactorArgs a)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (j >= a.yc) return;
    const double u = a.sc_.undef;
    const bool inner = (j >= 1 && j <= a.yc - 2);
    const int nco = a.gen ? 5 : 2;
    // the row predicate as a 64-bit word, all ones or zero: the kernel ANDs it into its lane masks (one s_and_b64;
    // as a double the all-ones pattern is a NaN, which also compares != 0.0)
    const double ones = __longlong_as_double(-1LL);
    double *f = a.rowf + (m * a.yc + j) * a.rw;
    double v[5];
    int k = 0;
    bool def = true;
    for (int q = 0; q < nco; q++) {
        v[q] = 0.0;
        if ((a.um >> q) & 1u) { v[q] = a.c[q][m * a.sc[q] + j * a.xc]; f[k++] = v[q]; def = def && (v[q] != u); }
    }
    if (a.hoist) {
        double rq = 0.0, rok = 0.0;
        if (inner) {
            if (!a.gen) {
                const double aP = a.c[0][m * a.sc[0] + (j + 1) * a.xc], a0 = v[0], c = v[1];
                rq = a.sc_.optArg / ((aP + a0) * a.sc_.ratioSqr + (c + c));
                rok = (def && (aP != u)) ? ones : 0.0;
            } else {
                const double A = v[0], C = v[1], F = v[4];
                rq = a.sc_.optArg / ((A * a.sc_.ratioSqr + C) * 2.0
                                     - F * a.sc_.delxSqr);
                rok = def ? ones : 0.0;
            }
        }
        f[k++] = rq; f[k++] = rok;
    }
    for (; k < a.rw; k++) f[k] = 0.0;
}
#endif

// ---- the point-factor stream Q of FusedGen2DQ (xinv_fused.h; k_fused2d) ------------------------------------------
struct PointFactorArgs {
    const double *c[6];           // A, C, D, E, F, G (FusedArgs order of the general form)
    int64_t sc[6];
    int64_t yc, xc, n;            // n = yc * xc
    XinvScal sc_;
    double *q;                    // [nbatch][yc][xc]
    int *flag;                    // bit 0: an updatable point's factor is +-0 (Q == 0 means "skip": the variant is not used);
                                  // bit 1: A and C differ somewhere (bitwise)
};
#ifdef XINV_AUX_KERNELS
// once per coefficient stack: Q[j,i] = optArg / ((A ratioSqr + C) 2 - F delxSqr) where numbas.py:1126-1129 lets the point
// be updated (every operand on the point defined, an interior row), 0 elsewhere.  The expression is inc()'s of
// FusedGen2D, evaluated in this translation unit under the same -ffp-contract=off: the same bits.
__global__ __launch_bounds__(256) void k_point_factor(PointFactorArgs a)
{
    const int64_t m = blockIdx.y;
    const double u = a.sc_.undef;
    bool zero = false, differ = false;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < a.n; t += (int64_t)gridDim.x * 256) {
        const int64_t j = t / a.xc;
        const double A = a.c[0][m * a.sc[0] + t], C = a.c[1][m * a.sc[1] + t], Dd = a.c[2][m * a.sc[2] + t];
        const double E = a.c[3][m * a.sc[3] + t], F = a.c[4][m * a.sc[4] + t], G = a.c[5][m * a.sc[5] + t];
        differ = differ || (__double_as_longlong(A) != __double_as_longlong(C));
        const bool ok = (j >= 1) && (j <= a.yc - 2) && (G != u) && (A != u) && (C != u) && (Dd != u) && (E != u) && (F != u);
        double q = 0.0;
        if (ok) {
            q = a.sc_.optArg / ((A * a.sc_.ratioSqr + C) * 2.0
                                - F * a.sc_.delxSqr);
            zero = zero || (q == 0.0);
        }
        a.q[m * a.n + t] = q;
    }
    if (__any(zero) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 1);
    if (__any(differ) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 2);
}
#endif

__device__ __forceinline__ void xinv_pipe_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// per-row factors through the scalar unit: the constant address space makes a uniform load an s_load
// (one s_load_dwordx8 per row into SGPRs: no vector registers, no VALU)
typedef const double __attribute__((address_space(4))) *xinv_cdouble_ptr;
// global address space spelled out: a pointer that has passed through an asm constraint is otherwise
// treated as generic (flat_load, which also counts on lgkmcnt)
typedef const char __attribute__((address_space(1))) *xinv_gcptr;
typedef char __attribute__((address_space(1))) *xinv_gptr;
typedef double xinv_v2d __attribute__((ext_vector_type(2)));          // (double2 is a class: no address-space overloads)
typedef const xinv_v2d __attribute__((address_space(1))) *xinv_gcd2ptr;
typedef const double __attribute__((address_space(1))) *xinv_gcdptr;
typedef xinv_v2d __attribute__((address_space(1))) *xinv_gd2ptr;
typedef double __attribute__((address_space(1))) *xinv_gdptr;

// 'extend' pre-pass for one boundary row held in the window (numbas.py:284-310), as fused_extend_fix but with
// the value of column c0-1 of the inner row handed in (with two column pairs per lane it is the lane's own)
__device__ __forceinline__ void pipe_extend_fix(double2 &edge, const double2 &inner, double inner_w,
                                                const LaneCols &lc, bool tall, double u)
{
    if (lc.cls_x == 1) { if (inner.x != u) edge.x = inner.x; }
    else if (lc.cls_x == 2) { if (inner.y != u) edge.x = inner.y; }           // (0,0) <- (1,1)
    else if (lc.cls_x == 3) {
        if (tall && inner.x != u) edge.x = inner.x;
        if (inner_w != u) edge.x = inner_w;                                    // <- column xc-2
    }
    if (lc.cls_y == 1) { if (inner.y != u) edge.y = inner.y; }
    else if (lc.cls_y == 3) {
        if (tall && inner.y != u) edge.y = inner.y;
        if (inner.x != u) edge.y = inner.x;                                    // <- column xc-2
    }
}

// one wavefront of the pipeline: sweep PW+1 on the rows of the tile [yu0, yu1).
// A lane holds NP adjacent column pairs (NP = 2: four consecutive columns, strips of 256 columns of which 240
// are owned): the inner neighbours of a point are then the lane's own registers and only one operand per PAIR
// of updates crosses lanes (DPP), the row bookkeeping, the barriers and the scalar work are shared by twice the
// arithmetic, and the column halo weighs 1.07 instead of 1.14.
// Registers: ONE ring of R = PF + 4 row records (S, F, per-row factors).  Row r is loaded straight into
// record r mod R, PF steps before it enters the window (the per-row factors two steps before), and stays
// there until it has left the window four steps later; the march is unrolled R steps, so every record index
// is a compile-time register name and no row is ever copied (a separate prefetch ring made the compiler
// rotate registers across the loop back-edge: ~80 v_mov and a full `s_waitcnt vmcnt(0)` per iteration).
// Row arithmetic is 32-bit (scalar compares; the 64-bit ones are VALU instructions on this target) and
// every global access is `uniform row base + 32-bit lane offset` (no per-load address arithmetic).
// FR: the forcing row travels with S through the ring -- wavefront 0 reads it from HBM, the others take it out of
// LDS instead of asking the L2 for it again.  For launches whose arrays exceed the caches (a batch of members) the
// later wavefronts' requests, ~20 rows behind the first, had fallen out of the L2 by then: C4, 64 members: 8.3 B
// per point-sweep measured against 6 B the variant must move (profiles/r03_pmc_*_c4.txt).
// SEAM (periodic x, odd xc): the row as an even ring with a phantom column (xinv_fused.h: RING).  `seam_lanes`: the lanes
// whose .x slot holds column xc-1; the half-sweeps that update the .x slots leave them out of their pass and run one more
// pass for them alone (east operand: the next lane's .x, the new column 0), after which the phantom column mirrors
// column xc-1 again.  Only the tiles that hold a seam lane are marched with SEAM = true (k_pipe2d below).
template <class M, unsigned UM, bool FR, int NP, bool AL, bool EXT, int PW, int PF, bool SEAM = false>
__device__ __forceinline__ void xinv_pipe_wave(const FusedArgs &a, int64_t m, int yu0, int yu1,
                                               const LaneCols (&lc)[NP], const int64_t (&st0)[NP], int lane,
                                               double2 (*ring)[XINV_PIPE_NS][NP * (FR ? 2 : 1)][XINV_WAVE], int gtot,
                                               double &acc, int &cnt,
                                               unsigned long long seam_lanes = 0ull)
{
    constexpr int P = XINV_PIPE_P, H = 2 * P, D = 4, LAG = XINV_PIPE_LAG, B = XINV_PIPE_B;
    constexpr int R = PF + D;                            // row records; also the unroll period
    constexpr int PFR = M::PIPE_PFR;                     // steps the per-row record is requested ahead
    using REC = PipeRec<M, UM>;
    constexpr int NC = M::NC, FQ = M::NC - 1, RW = REC::RW;
    constexpr bool HOIST = REC::HOIST;
    constexpr int RS = FR ? 2 : 1;                       // ring entries per column pair: S (, forcing)
    static_assert(R % D == 0 && R % B == 0 && R % XINV_PIPE_NS == 0,
                  "the unroll period must keep row parity, ring slots and barriers compile-time");
    const int ycr = (int)a.yc;
    const unsigned rowbytes = (unsigned)a.xc * 8u;       // (a row is shorter than 4 GiB)
    const double u = a.sc_.undef;
    // Every array of the member is addressed through a raw buffer resource (base, size of the slice) with the ROW as
    // the instruction's scalar offset and the lane's columns as its vector offset: `buffer_load_dwordx4 v, v_lane,
    // s[rsrc], s_row offen` -- no address arithmetic per load, and the row offset of all streams is ONE running
    // 32-bit value (the planner admits slices below 2 GiB).
    const int slice_bytes = (int)(rowbytes * (unsigned)ycr);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)(a.src + m * a.sS), 0, slice_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dst + m * a.sS), 0, slice_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsC[NC];                      // vector streams (x-uniform ones come through the record)
#pragma unroll
    for (int q = 0; q < NC; q++)
        rsC[q] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.c[q] + m * a.sc[q]), 0, slice_bytes, 0x00020000);
    const xinv_cdouble_ptr rowf = (xinv_cdouble_ptr)(uintptr_t)(reinterpret_cast<const double *>(a.rowf) + m * a.yc * RW);
    unsigned lo0[NP], lo1[NP], so0[NP], so1[NP];         // byte offsets of the lane's columns (loads / owned stores)
#pragma unroll
    for (int q = 0; q < NP; q++) {
        lo0[q] = (unsigned)lc[q].l0 * 8u; lo1[q] = (unsigned)lc[q].l1 * 8u;
        so0[q] = lc[q].use_x ? (unsigned)st0[q] * 8u : 0xffffffffu;      // (a column the lane does not own: beyond the
        so1[q] = lc[q].use_y ? (unsigned)(st0[q] + 1) * 8u : 0xffffffffu;  //  resource's range -- the store is dropped)
    }

    unsigned long long okx64[NP], oky64[NP];             // "column may be updated" as 64-bit lane masks (SGPR pairs)
    double nsx[NP], nsy[NP];                             // norm share per lane and column
    int nnx[NP], nny[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        okx64[q] = __builtin_amdgcn_ballot_w64(lc[q].ok_x) & ~seam_lanes; oky64[q] = __builtin_amdgcn_ballot_w64(lc[q].ok_y);
        nsx[q] = nsy[q] = 0.0; nnx[q] = nny[q] = 0;
    }
    static_assert(!SEAM || (NP == 1 && !AL && PipeRec<M, UM>::HOIST),
                  "seam variants: one column pair per lane, per-row records, updates under EXEC masks");

    const int in_lo = yu0 - H + 2 * PW;                  // first / last row entering this wavefront's window
    const int in_hi = yu1 - 1 + H - 2 * PW;

    double2 sw[NP][R];
    CoefWin<NC, R> cw[NP];
    double rokw[R];
#pragma unroll
    for (int t = 0; t < R; t++) {
        rokw[t] = 0.0;
#pragma unroll
        for (int q = 0; q < NP; q++) {
            sw[q][t] = make_double2(0.0, 0.0);
            cw[q].rq[t] = 0.0; cw[q].mx[t] = 0u; cw[q].my[t] = 0u;
#pragma unroll
            for (int c = 0; c < NC; c++) { cw[q].v[c][t] = make_double2(0.0, 0.0); cw[q].s[c][t] = 0.0; }
        }
    }

    // request row r into record `slot` (S from HBM for wavefront 0 only; later wavefronts get it through LDS).
    // Rows are requested in order: nxo is the byte offset of the NEXT row in the slice, unclamped (negative above
    // the slice); the rows of a tile at the top or bottom of the slice are clamped to [0, yc-1] as before (their
    // updates are switched off by the row predicate).  Three scalar instructions per step for all streams.
    const int maxo = (int)(rowbytes * (unsigned)(ycr - 1));
    int nxo = (yu0 - H + 2 * PW) * (int)rowbytes;
    int nxr = (yu0 - H + 2 * PW) * (RW * 8);             // the same for the per-row records
    // a row of one of the lane's column pairs
    auto ldrow = [&](__amdgpu_buffer_rsrc_t rs, int soff, auto qtag) {
        constexpr int q = decltype(qtag)::value;
        double2 v;
        if (AL) {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo0[q], soff, 0);
            v.x = __hiloint2double((int)t[1], (int)t[0]); v.y = __hiloint2double((int)t[3], (int)t[2]);
        } else {
            const auto t0 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lo0[q], soff, 0);
            const auto t1 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lo1[q], soff, 0);
            v.x = __hiloint2double((int)t0[1], (int)t0[0]); v.y = __hiloint2double((int)t1[1], (int)t1[0]);
        }
        return v;
    };
    auto request = [&](int, auto stag) {
        constexpr int slot = decltype(stag)::value;
        const int soff = min(max(nxo, 0), maxo);
        nxo += (int)rowbytes;
        asm("" : "+s"(nxo));                             // (kept as ONE running value: the unrolled steps would otherwise
                                                         //  hold eight multiples of the row pitch in SGPRs)
        xinv_unroll_steps([&](auto qtag) {
            constexpr int q = decltype(qtag)::value;
            if (PW == 0) sw[q][slot] = ldrow(rsS, soff, qtag);
            xinv_unroll_steps([&](auto ctag) {
                constexpr int c = decltype(ctag)::value;
                if (!((UM >> c) & 1u) && !(FR && c == FQ && PW > 0)) cw[q].v[c][slot] = ldrow(rsC[c], soff, qtag);
            }, std::make_integer_sequence<int, NC>{});
        }, std::make_integer_sequence<int, NP>{});
    };
    auto request_rf = [&](int, auto stag) {
        constexpr int slot = decltype(stag)::value;
        if constexpr (RW > 0) {                          // (rows 0 and yc-1 carry rok = 0: so do the clamped ones)
            const unsigned roff = (unsigned)min(max(nxr, 0), (ycr - 1) * (RW * 8));
            nxr += RW * 8;
            const xinv_cdouble_ptr pr = (xinv_cdouble_ptr)((const char __attribute__((address_space(4))) *)rowf + roff);
            double rec[RW];
#pragma unroll
            for (int k = 0; k < RW; k++) rec[k] = pr[k];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                int k = 0;
#pragma unroll
                for (int c = 0; c < NC - 1; c++)
                    if ((UM >> c) & 1u) cw[q].s[c][slot] = rec[k++];
                if (HOIST) { cw[q].rq[slot] = rec[k]; rokw[slot] = rec[k + 1]; }
            }
        }
    };
    // (in row order, as in the loop: the vmcnt waits of the loop are computed against the worst path into it)
    xinv_unroll_steps([&](auto ttag) { request(in_lo + decltype(ttag)::value, ttag); asm volatile("" ::: "memory"); },
                      std::make_integer_sequence<int, PF>{});
    xinv_unroll_steps([&](auto ttag) { request_rf(in_lo + decltype(ttag)::value, ttag); },
                      std::make_integer_sequence<int, PFR>{});

    // one half-sweep of row record sj (sjp / sjm: the rows below / above) on lane components X
    // (fixt: the seam lanes' pass -- X == 0, the east operand is the next lane's .x)
    auto half_pass = [&](auto xtag, auto jtag, auto ptag, auto mtag, auto fixt) {
        constexpr int X = decltype(xtag)::value, sj = decltype(jtag)::value, sjp = decltype(ptag)::value,
                      sjm = decltype(mtag)::value;
        constexpr bool FIX = decltype(fixt)::value;
        static_assert(!FIX || (X == 0 && NP == 1), "column xc-1 sits in an .x slot");
        // the one operand that crosses lanes: west of the first column / east of the last
        const double edge = (X == 0) ? xinv_lane_up(sw[NP - 1][sj].y) : xinv_lane_down(sw[0][sj].x);
        double nv[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            double w, e;
            if (X == 0) { w = (q == 0) ? edge : sw[q > 0 ? q - 1 : 0][sj].y; e = sw[q][sj].y; }
            else        { w = sw[q][sj].x; e = (q == NP - 1) ? edge : sw[q < NP - 1 ? q + 1 : q][sj].x; }
            if constexpr (FIX) e = xinv_lane_down(sw[q][sj].x);
            // the increment, added under the update predicate as EXEC (xinv_add_where: no select on the VALU)
            const double t = M::template inc<X, UM, R, false>(cw[q], sj, sjp, comp<X>(sw[q][sj]), comp<X>(sw[q][sjp]),
                                              comp<X>(sw[q][sjm]), w, e, a.sc_);
            if constexpr (HOIST) {
                // the predicate: (column may be updated) & (row may be updated: the record's predicate word is all
                // ones or zero, k_row_factor) as a 64-bit lane mask on the scalar unit, (forcing defined) by the
                // compare that writes EXEC
                const unsigned long long rm = (FIX ? seam_lanes : (X ? oky64[q] : okx64[q])) &
                                              (unsigned long long)__double_as_longlong(rokw[sj]);
                if constexpr (ModelFma<M>::value)          // (t is the bracket before the relaxation factor: one fma finishes)
                    nv[q] = xinv_fma_where_ne(comp<X>(sw[q][sj]), t, cw[q].rq[sj], comp<X>(cw[q].v[FQ][sj]), u, rm);
                else
                    nv[q] = xinv_add_where_ne(comp<X>(sw[q][sj]), t, comp<X>(cw[q].v[FQ][sj]), u, rm);
            } else {
                const unsigned long long pm = __builtin_amdgcn_ballot_w64((X ? cw[q].my[sj] : cw[q].mx[sj]) != 0u);
                nv[q] = xinv_add_where(comp<X>(sw[q][sj]), t, pm);
            }
        }
#pragma unroll
        for (int q = 0; q < NP; q++) setc<X>(sw[q][sj], nv[q]);
    };
    auto half_sweep = [&](auto xtag, auto jtag, auto ptag, auto mtag) {
        constexpr int X = decltype(xtag)::value;
        half_pass(xtag, jtag, ptag, mtag, std::false_type{});
        if constexpr (SEAM && X == 0) {                  // column xc-1 behind column 0; the phantom column mirrors it again
            constexpr int sj = decltype(jtag)::value;
            half_pass(xtag, jtag, ptag, mtag, std::true_type{});
            sw[0][sj].y = xinv_bitsel64(seam_lanes, sw[0][sj].x, sw[0][sj].y);
        }
    };

    // global step g of the workgroup = local step + LAG * PW; a barrier closes every B-th global step.
    // Row in_lo is taken out of the ring in global step LAG * PW - 1 -- the step before this wavefront's first,
    // like every later row (one step before it enters) -- and BEFORE the barrier that may close that step: the
    // producer wrote it two steps earlier and reuses the slot two steps later, so the read has to sit in the
    // barrier interval between.  (Read after the pre-loop, as this code did until round 3, it fell into the
    // NEXT interval for even PW -- LAG * PW - 1 odd -- which is the interval of the producer's overwrite: under
    // load the third wavefront sometimes got row in_lo + 4 for row in_lo, and the one point of the tile whose
    // dependency cone touches it, the first owned row's last half-sweep, came out wrong by a few ulps to 1e-4.
    // Found on 64 x 1440 x 720 Gill-Matsuno members; profiles/r03_pipe2d_ring_race.txt.)
    int g = 0;
#pragma unroll
    for (; g < LAG * PW; g++) {
        if (PW > 0 && g == LAG * PW - 1) {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                sw[q][0] = ring[PW - 1][(2 * PW) % XINV_PIPE_NS][q * RS][lane];      // row in_lo
                if (FR) cw[q].v[FQ][0] = ring[PW - 1][(2 * PW) % XINV_PIPE_NS][q * RS + 1][lane];
            }
        }
        if ((g + 1) % B == 0) xinv_pipe_barrier();
    }

    for (int rb_ = in_lo; rb_ <= in_hi; rb_ += R) {
        xinv_unroll_steps([&](auto utag) {
            constexpr int U = decltype(utag)::value;     // record of the entering row r
            constexpr int X = (U & 1) ? 0 : 1;
#define SLOT(w) ((U - (w) + 4 * R) % R)
#define RSLOT(w) ((2 * PW + U - (w) + 64 * XINV_PIPE_NS) % XINV_PIPE_NS)  /* LDS ring slot of row r - w */
#define ITAG(v) std::integral_constant<int, (v)>{}
            const int r = rb_ + U;
            request(r + PF, ITAG((U + PF) % R));
            request_rf(r + PFR, ITAG((U + PFR) % R));
            if (PW > 0) {
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    sw[q][(U + 1) % R] = ring[PW - 1][RSLOT(-1)][q * RS][lane];   // row r+1: written B+1 steps ago
                    if (FR) cw[q].v[FQ][(U + 1) % R] = ring[PW - 1][RSLOT(-1)][q * RS + 1][lane];
                }
            }
            if constexpr (!HOIST) {          // coefficient arrays that vary along x: the model's own predicate
                const bool rv = (r - 1 >= 1) && (r - 1 <= ycr - 2);
#pragma unroll
                for (int q = 0; q < NP; q++)
                    M::template derive<UM, R, false>(cw[q], SLOT(0), SLOT(1), rv && lc[q].ok_x, rv && lc[q].ok_y, a.sc_);
            }
            {   // red half-sweep on row r-1
                const int ja = r - 1;
                constexpr int sj = SLOT(1), sjp = SLOT(0), sjm = SLOT(2);
                if (EXT) {
                    if (ja == 1 || ja == ycr - 2) {
                        const double wl = xinv_lane_up(sw[NP - 1][sj].y);      // column c0-1 of the first pair
#pragma unroll
                        for (int q = 0; q < NP; q++) {
                            const double iw = (q == 0) ? wl : sw[q > 0 ? q - 1 : 0][sj].y;
                            if (ja == 1) pipe_extend_fix(sw[q][sjm], sw[q][sj], iw, lc[q], a.tall, u);
                            if (ja == ycr - 2) pipe_extend_fix(sw[q][sjp], sw[q][sj], iw, lc[q], a.tall, u);
                        }
                    }
                }
                half_sweep(ITAG(X), ITAG(sj), ITAG(sjp), ITAG(sjm));
            }
            {   // black half-sweep on row r-2
                const int jb = r - 2;
                constexpr int sj = SLOT(2), sjp = SLOT(1), sjm = SLOT(3);
                half_sweep(ITAG(X), ITAG(sj), ITAG(sjp), ITAG(sjm));
                if ((unsigned)(jb - yu0) < (unsigned)(yu1 - yu0)) {   // an owned row: its share of mean|S| of this sweep
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        const double2 t = sw[q][sj];
                        // magnitudes and samples per lane and column under `S != undef` as EXEC; the lanes that
                        // do not own their column are discarded after the march
                        xinv_norm_row(nsx[q], nsy[q], nnx[q], nny[q], t.x, t.y, u);
                    }
                }
                // ---- row r-2 leaves
                if (PW < P - 1) {
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        ring[PW][RSLOT(2)][q * RS][lane] = sw[q][sj];
                        if (FR) ring[PW][RSLOT(2)][q * RS + 1][lane] = cw[q].v[FQ][sj];
                    }
                } else if ((unsigned)(jb - yu0) < (unsigned)(yu1 - yu0)) {
                    const int doff = jb * (int)rowbytes;
                    xinv_unroll_steps([&](auto qtag) {
                        constexpr int q = decltype(qtag)::value;
                        const double2 t = sw[q][sj];
                        typedef unsigned xinv_v4u __attribute__((__vector_size__(16)));
                        typedef unsigned xinv_v2u __attribute__((__vector_size__(8)));
                        if (AL) {                            // (use_x == use_y on aligned strips; un-owned lanes are out of range)
                            const xinv_v4u tv = {(unsigned)__double2loint(t.x), (unsigned)__double2hiint(t.x),
                                                 (unsigned)__double2loint(t.y), (unsigned)__double2hiint(t.y)};
                            __builtin_amdgcn_raw_buffer_store_b128(tv, rsD, (int)so0[q], doff, 0);
                        } else {
                            const xinv_v2u tx = {(unsigned)__double2loint(t.x), (unsigned)__double2hiint(t.x)};
                            const xinv_v2u ty = {(unsigned)__double2loint(t.y), (unsigned)__double2hiint(t.y)};
                            __builtin_amdgcn_raw_buffer_store_b64(tx, rsD, (int)so0[q], doff, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(ty, rsD, (int)so1[q], doff, 0);
                        }
                    }, std::make_integer_sequence<int, NP>{});
                }
            }
            if ((LAG * PW + U + 1) % B == 0) xinv_pipe_barrier();
#undef SLOT
#undef RSLOT
#undef ITAG
        }, std::make_integer_sequence<int, R>{});
        g += R;
    }
    for (; g < gtot; g++) if ((g + 1) % B == 0) xinv_pipe_barrier();
#pragma unroll
    for (int q = 0; q < NP; q++) {                       // only the columns this lane owns count
        acc += (lc[q].use_x ? nsx[q] : 0.0);
        acc += (lc[q].use_y ? nsy[q] : 0.0);
        cnt += (lc[q].use_x ? nnx[q] : 0) + (lc[q].use_y ? nny[q] : 0);
    }
}

template <class M, unsigned UM, bool FR, int NP, bool AL, bool EXT, bool SEAM = false>
__global__ __launch_bounds__(64 * XINV_PIPE_P) void k_pipe2d(FusedArgs a_)
{
    xinv_fresh_scalar_cache();
    constexpr int P = XINV_PIPE_P, K = P, H = 2 * K, LAG = XINV_PIPE_LAG, B = XINV_PIPE_B;
    __shared__ double2 ring[P - 1][XINV_PIPE_NS][NP * (FR ? 2 : 1)][XINV_WAVE];

    unsigned tag;
    int T;
    double acc = 0.0;
    int cnt = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int pwi;
    {
    const FusedArgs &a = a_;
    const int64_t m = a.member0 + blockIdx.y;
    XinvCtl *ctl = a.ctl + m;
    if (!a.force && xinv_ctl_done(ctl)) return;
    if (a.lag && (int)blockIdx.x == a.nwg) { xinv_lag_reduce_prev(a, ctl, m); return; }
    tag = a.lag ? a.tag : xinv_ctl_seq(ctl);

    const int NB = a.nwg;
    {
        const int L = blockIdx.x, q = NB >> 3, rem = NB & 7, xcd = L & 7, idx = L >> 3;
        T = xcd * q + (xcd < rem ? xcd : rem) + idx;
    }
    // which sweep this wavefront applies: rotated from workgroup to workgroup, so that the workgroups sharing
    // a CU do not all start (and drain) their pipelines on the same SIMD
    pwi = (wave + (int)(blockIdx.x >> 8)) & (P - 1);       // (dispatch order: 8 XCDs x 32 CUs, then the next round)
    int wt = T;
    bool active = wt < a.nstrip * a.nrb;
    if constexpr (SEAM) {                                // (the edge strips' tiles first: xinv_heavy_first)
        if (!a.tile_list && NB == a.nstrip * a.nrb)
            wt = xinv_seam_tile(xinv_heavy_first((int)blockIdx.x, NB, (a.nstrip == 1 ? 1 : 2) * a.nrb), a.nstrip, a.nrb);
    }
    if (a.tile_list) {
        wt = a.tile_list[m * a.ntl + T];
        active = wt >= 0;
        wt = active ? wt : 0;
    }
    int rb = wt / a.nstrip, strip = wt - rb * a.nstrip;
    const int64_t xc = a.xc, yc = a.yc;
    int yu0, yu1;
    if (a.RY > 0) {
        yu0 = rb * a.RY;
        yu1 = (yu0 + a.RY < (int)yc) ? yu0 + a.RY : (int)yc;
    } else {
        yu0 = (int)((((int64_t)rb * yc) / a.nrb) & ~(int64_t)1);
        yu1 = (rb + 1 == a.nrb) ? (int)yc : (int)((((int64_t)(rb + 1) * yc) / a.nrb) & ~(int64_t)1);
    }
    // (SEAM: the ring layout's strips and halos, xinv_tiles.h)
    const int UW = SEAM ? xinv_ring_uw(xc, H) : XINV_PIPE_UW(NP), HW = SEAM ? xinv_ring_hw(xc, H, strip) : H;
    const int64_t xu0 = (int64_t)strip * UW;
    LaneCols lc[NP];
    int64_t st0[NP];
    RingSeam rs = {0ull, false};
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if constexpr (SEAM) lc[q] = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
        else lc[q] = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0, NP, q);
        st0[q] = xu0 - HW + 2 * NP * lane + 2 * q;       // unwrapped store column of the pair's .x
    }

    if (active) {
        // global steps every wavefront goes through: the longest of the four schedules, whole barrier periods
        const int ry = yu1 - yu0;
        int gtot = 0;
#pragma unroll
        for (int pw = 0; pw < P; pw++) {
            const int per = (pw == 0 ? XINV_PIPE_PF0 : XINV_PIPE_PF) + 4;      // unroll period of that wavefront
            const int n = ry + 2 * H - 4 * pw;
            const int g = LAG * pw + ((n + per - 1) / per) * per;
            gtot = g > gtot ? g : gtot;
        }
        gtot = ((gtot + B - 1) / B) * B;
        // SEAM: only the tiles that hold a seam lane take the march with the extra pass; every other tile of the launch
        // runs the plain one.  (One march with the extra passes behind wave-uniform branches cost EVERY tile its
        // instruction interleaving: 3601 columns ran 1.45x the time of 3600 -- profiles/r05_seam_rates.txt.)
        const bool wraps = rs.any;
#define XINV_PIPE_MARCH(SM) \
        switch (pwi) { \
        case 0: xinv_pipe_wave<M, UM, FR, NP, AL, EXT, 0, XINV_PIPE_PF0, SM>(a, m, yu0, yu1, lc, st0, lane, ring, gtot, acc, cnt, rs.lanes); break; \
        case 1: xinv_pipe_wave<M, UM, FR, NP, AL, EXT, 1, XINV_PIPE_PF, SM>(a, m, yu0, yu1, lc, st0, lane, ring, gtot, acc, cnt, rs.lanes); break; \
        case 2: xinv_pipe_wave<M, UM, FR, NP, AL, EXT, 2, XINV_PIPE_PF, SM>(a, m, yu0, yu1, lc, st0, lane, ring, gtot, acc, cnt, rs.lanes); break; \
        default: xinv_pipe_wave<M, UM, FR, NP, AL, EXT, 3, XINV_PIPE_PF, SM>(a, m, yu0, yu1, lc, st0, lane, ring, gtot, acc, cnt, rs.lanes); break; \
        }
        if (wraps) { XINV_PIPE_MARCH(SEAM) } else { XINV_PIPE_MARCH(false) }
#undef XINV_PIPE_MARCH
    }
    }
    // What follows needs a dozen kernel arguments the march does not: read them again from the argument
    // block here (through a pointer the optimiser cannot see through) instead of carrying them in SGPRs
    // across the march, where they pushed loop values out into VGPR lanes (v_readlane in every step).
    const FusedArgs __attribute__((address_space(4))) *kp =
        (const FusedArgs __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp) :: "memory");
    if (kp->no_ctl) return;
    const int64_t m = kp->member0 + blockIdx.y;
    XinvCtl *ctl = kp->ctl + m;
    const int NB = kp->nwg;
    struct { unsigned long long *psum; const double *xsum; const long long *xcnt; int lag; XinvStop stop; const void *hook; } a;
    a.psum = kp->psum; a.xsum = kp->xsum; a.xcnt = kp->xcnt; a.lag = kp->lag;
    a.hook = XINV_TEST_HOOKS ? (const void *)kp->dbg : nullptr;
    a.stop.mxLoop = kp->stop.mxLoop; a.stop.tolerance = kp->stop.tolerance;
    a.stop.stop_on_zero_norm = kp->stop.stop_on_zero_norm;

    // wavefront `pwi` holds the tile's share of sweep pwi+1: publish it (three tagged words, see xinv_norm_publish)
    {
        unsigned long long *pw = a.psum + (size_t)m * XINV_KMAX * NB * XINV_PW;
        const double ws = xinv_wave_sum(acc);
        const long long wc = xinv_wave_sum_ll((long long)cnt);
        if (lane == 0 && !xinv_hook_withhold(a.hook, T, tag, m)) {
            const unsigned long long hi = (unsigned long long)tag << 32;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(ws);
            unsigned long long *q = pw + ((size_t)pwi * NB + T) * XINV_PW;
            __hip_atomic_store(q + 0, hi | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 1, hi | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 2, hi | (unsigned long long)(unsigned)wc, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.lag || blockIdx.x != gridDim.x - 1) return;
        __syncthreads();
        xinv_norm_reduce<K, P>(wave, lane, NB, tag, pw, ctl, a.stop, a.xsum ? a.xsum[m] : 0.0,
                               a.xcnt ? a.xcnt[m] : 0);
    }
}
