// xinv_pipe3d.h -- TWO red-black sweeps per pass for the 3-D standard form with x-uniform coefficients
// (every lat-lon omega coefficient, apps.py:2033-2035; numbas.invert_standard_3D, numbas.py:15-212), the
// sweeps pipelined ACROSS two groups of wavefronts of a workgroup (gfx950).  BASELINE configs[4].
//
// k_fused3d (one sweep per pass) moves ~27 B per point-sweep and sits at what HBM delivers; only fewer bytes
// help.  k_fused3d2 applied both sweeps inside every wavefront (four dependent stages with an LDS exchange
// between each: bound by its own latency chain, slower).  Here the idea of k_pipe2d is carried to the k march:
// a workgroup of 2 G wavefronts owns a cross-section of NR = G * RR rows x one 128-column strip and marches it
// through the planes;
//   group 0 (wavefronts 0..G-1)  streams S and the forcing from HBM and applies sweep 1 -- the K = 1 step of
//                                k_fused3d: with r the plane entering, red points of plane r-1, black points of
//                                plane r-2 -- and hands every finished plane on through a two-slot ring in LDS;
//   group 1 (wavefronts G..2G-1) takes the plane out of the ring one workgroup barrier later (three planes
//                                behind), applies sweep 2 the same way and writes the owned rows back.
// S and the forcing are read once and S written once per TWO sweeps.  A wavefront owns RR adjacent rows (RR = 3:
// 24 rows per cross-section, 16 owned -- the halo is four rows / columns / planes a side, recomputed, never
// exchanged), so two of its four j neighbours per row pair are its own registers; the rows above / below its block
// come from the adjacent wavefronts of its group through LDS exactly as in k_fused3d (published for the NEXT step,
// double-buffered by step parity): ONE workgroup barrier per plane.
//
// Coefficients are one value per (plane, row); with them the relaxation factor
//   optArg / ((A[k+1,j] + A[k,j]) ratio2Sqr + (B[k,j+1] + B[k,j]) ratio1Sqr + 2 C[k,j])
// and the (plane, row) part of the update predicate are the same for every point of a row and every sweep of
// the solve: k_row_factor3d evaluates them once per solve (same expression as k_fused3d<UNI>, same bits) into
// 64-byte records {A[k+1], A[k], B[j+1], B[j], C, factor, predicate word, -} and the wavefronts read one record
// per (plane, row) through the scalar unit (behind s_dcache_inv: the scalar-cache rule).
// The update and the norm share are added under EXEC masks (xinv_add_where_ne / xinv_norm_row).
// Same point arithmetic and ordering as two passes of k_fused3d: bitwise equal to it and to the oracle.
#pragma once
#include "xinv_fused3d.h"
#include "xinv_pipe2d.h"

#ifndef XINV_P3_RR
#define XINV_P3_RR 3              /* rows per wavefront */
#endif
#ifndef XINV_P3_G
#define XINV_P3_G 8               /* wavefronts per group (two groups per workgroup) */
#endif

struct RowFactor3Args {
    const double *c[3];           // A, B, C
    int64_t sc[3];                // batch strides (0 = shared)
    int64_t zc, yc, xc;
    XinvScal sc_;
    double *rowf;                 // [nb][zc][yc][8]
    int64_t srowf;                // member stride of the table in doubles (0: one table for the batch)
};

#ifdef XINV_AUX_KERNELS
// once per solve: the per-(plane, row) records (numbas.py:146-169 with the coefficients constant along x;
// the hoisted branch of k_fused3d<UNI>: same expressions, same bits)
__global__ __launch_bounds__(256) void k_row_factor3d(RowFactor3Args a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (idx >= a.zc * a.yc) return;
    const int64_t k = idx / a.yc, j = idx - k * a.yc;
    const double u = a.sc_.undef;
    const int64_t kp = k + 1 > a.zc - 1 ? a.zc - 1 : k + 1, jp = j + 1 > a.yc - 1 ? a.yc - 1 : j + 1;
    const double *pA = a.c[0] + m * a.sc[0], *pB = a.c[1] + m * a.sc[1], *pC = a.c[2] + m * a.sc[2];
    const double aP = pA[(kp * a.yc + j) * a.xc], a0 = pA[(k * a.yc + j) * a.xc];
    const double bP = pB[(k * a.yc + jp) * a.xc], b0 = pB[(k * a.yc + j) * a.xc];
    const double c = pC[(k * a.yc + j) * a.xc];
    const bool inner = (k >= 1) && (k <= a.zc - 2) && (j >= 1) && (j <= a.yc - 2);
    double rq = 0.0, rok = 0.0;
    if (inner) {
        rq = a.sc_.optArg / ((aP + a0) * a.sc_.ratio2Sqr +
                             (bP + b0) * a.sc_.ratio1Sqr +
                             (c + c));
        rok = ((aP != u) && (a0 != u) && (bP != u) && (b0 != u) && (c != u)) ? __longlong_as_double(-1LL) : 0.0;
    }
    double *f = a.rowf + m * a.srowf + idx * 8;
    f[0] = aP; f[1] = a0; f[2] = bP; f[3] = b0; f[4] = c; f[5] = rq; f[6] = rok; f[7] = 0.0;
}
#endif

// SEAM (periodic x, ODD xc): the even ring with a phantom column (xinv_fused.h: RING) -- one more pass, for the seam lanes
// alone, in the half-sweeps that update the .x slots, and only in the cross-sections that hold a seam lane.
// EXT (BCy = 'extend', numbas.py:87-115: at the start of every sweep rows 0 / yc-1 of planes 1 .. zc-2 take the values of rows
// 1 / yc-2 where those are defined): the pre-pass of the pass's FIRST sweep is applied to the source buffer by k_extend
// before this launch (idempotent; launch_fused3d) -- group 0 finds it in what it loads; the pre-pass of the SECOND sweep is
// applied by group 1 to the plane it takes out of the ring, before anything reads it, out of the wavefront's own registers:
// rows 0 / 1 and rows yc-2 / yc-1 each have to sit in one wavefront of every cross-section that needs the boundary row
// right: true for five row counts in eight as the blocks lie, for all of them with the blocks shifted up by two rows where
// needed (Fused3Args::joff; xinv_p3_extend_joff, xinv_tiles.h).  (Round 5 loaded the partner rows from HBM; round 6 first read row yc-2 out of the neighbouring
// wavefront's ring slot where the pair is split: both bit-exact, both spilled 26-34 registers and ran at a third of the
// rate -- the kernel has 128 registers and uses 123-126.)
template <int G, int RR, bool AL, bool FMA = false, bool SEAM = false, bool EXT = false>
__global__ __launch_bounds__(2 * G * 64) void k_pipe3d(Fused3Args a)
{
    static_assert(!SEAM || !AL, "odd xc: strips are never aligned");
    static_assert(!EXT || !SEAM, "'extend' with the odd-xc periodic seam keeps the one-sweep kernel");
    xinv_fresh_scalar_cache();
    constexpr int K = 2, H = 2 * K, NW = 2 * G, NR = G * RR, RJ = NR - 2 * H, D = 4;

    // ---- which tile, which planes.  The grid is FLAT over the launch's members (Fused3Args::nfull): the workgroups below
    // nfull march one tile each through the whole column; the tiles behind them -- the remainder of a launch whose tile
    // count is not a multiple of the compute units: at one workgroup per CU they would march alone in a last round while
    // the other CUs idle (15 volumes of 50 x 360 x 720: 2070 tiles = 8.09 rounds) -- are cut into the nkc chunks of KC
    // planes, pieces of a third to a half of a march that finish together.  A member's partials are laid out for the cut
    // (nkc slots per tile); a workgroup that marches the whole column publishes its share in the first and zeros in the rest.
    const int NT = a.nstrip * a.njb;                     // tiles of a member
    const int NB = NT * a.nkc;                           // partial slots of a member
    const int L = (int)blockIdx.x, nfull = (int)a.nfull; // (a launch has fewer than 2^31 workgroups)
    const bool whole = L < nfull;                        // (slot -> tile, chunk, member: xinv_tiles.h, shared with the CPU suite's check)
    int kc;
    const int g = xinv_p3_slot_tile(L, nfull, a.nkc, kc);
    const int ml = g / NT;                               // member of the launch
    const int64_t m = a.member0 + ml;
    XinvCtl *ctl = a.ctl + m;
    if (!a.force && xinv_ctl_done(ctl)) return;
    const unsigned tag = xinv_ctl_seq(ctl);
    // the member's workgroup dispatched last reduces its norm (every other one is resident or finished by then)
    const bool reducer = xinv_p3_slot_reduces(L, nfull, a.nkc, NT, ml);
    const int Tj = xinv_p3_slot_member_tile(L, nfull, NT, ml, g);             // tile of the member (XCD-aware order)
    const int T = kc * NT + Tj;                          // the partial slot
    const int jb = Tj / a.nstrip, st = Tj - jb * a.nstrip;
    const int zc = (int)a.zc, yc = (int)a.yc;
    const int k0 = whole ? 0 : kc * a.KC;
    const int k1 = (whole || kc + 1 == a.nkc) ? zc : k0 + a.KC;
    // (the wavefront index as a scalar: rows, record addresses and row predicates are then wave-uniform values)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int grp = wave >= G ? 1 : 0, gw = wave - grp * G;
    const int64_t xc = a.xc;
    const int UW = SEAM ? xinv_ring_uw(xc, H) : 128 - 2 * H, HW = SEAM ? xinv_ring_hw(xc, H, st) : H;   // (SEAM: xinv_tiles.h)
    const int64_t xu0 = (int64_t)st * UW;
    const double u = a.sc_.undef;
    RingSeam rs = {0ull, false};
    LaneCols lc;
    if constexpr (SEAM) lc = make_lanecols_ring(xu0, HW, UW, lane, xc, rs);
    else lc = make_lanecols<AL>(xu0, H, UW, lane, xc, a.per != 0);
    const int64_t st0 = xu0 - HW + 2 * lane;
    // (SEAM: the seam lanes' .x -- column xc-1 -- is left out of the pass of its half-sweep and updated alone behind it)
    const unsigned long long okx64 = __builtin_amdgcn_ballot_w64(lc.ok_x) & ~rs.lanes, oky64 = __builtin_amdgcn_ballot_w64(lc.ok_y);

    // the RR rows of this wavefront (the same rows in both groups)
    const int j0 = jb * RJ - H - a.joff + gw * RR;       // (joff: the 'extend' variant's tiling offset, 0 elsewhere)
    int jr[RR];
    bool row_use[RR];
#pragma unroll
    for (int rr = 0; rr < RR; rr++) {
        const int j = j0 + rr;
        jr[rr] = j < 0 ? 0 : (j > yc - 1 ? yc - 1 : j);
        row_use[rr] = (gw * RR + rr >= H) && (gw * RR + rr < NR - H) && (j < yc) && (j >= 0);
    }
    // neighbouring wavefronts of the group (the first / last one reads itself: those rows are halo)
    const int wm = gw > 0 ? wave - 1 : wave, wp = gw < G - 1 ? wave + 1 : wave;

    const double *srcS = a.src + m * a.sS;
    double *dstS = a.dst + m * a.sS;
    const double *pF = a.c[3] + m * a.sc[3];
    const xinv_cdouble_ptr rowf = (xinv_cdouble_ptr)(uintptr_t)(a.rowf + m * a.srowf);

    __shared__ double xch[2][2][NW][2][64];              // [step parity][as loaded | red-updated][wave][top | bottom row][lane]
    // [wave of the group][row][slot][S (| forcing)][lane]: planes with sweep 1 complete.
    // (laid out [wave][row][slot][S | forcing][lane]: every access of a wavefront is ONE base register + an immediate
    //  offset below 64 KiB -- with the slot outermost the last quarter lay beyond the 16-bit offset and cost a second
    //  base register, which the 128-register budget did not have: spills)
    __shared__ double2 ring[G][RR][2][2][64];
    // (the first and the last row of the cross-section are updated with a j neighbour missing: whatever they become is
    //  never read by a row that is kept -- their forcing is not loaded: 2 of 24 rows, 4 % of the bytes read)
    //  (all ones ORed into the lane offset: out of the resource's range, the request returns zero and fetches nothing;
    //   a branch around the load cost 31 spilled registers)
    unsigned f_drop[RR];
#pragma unroll
    for (int rr = 0; rr < RR; rr++) f_drop[rr] = ((gw == 0 && rr == 0) || (gw == G - 1 && rr == RR - 1)) ? ~0u : 0u;

    double nsx = 0.0, nsy = 0.0;                         // norm share per lane and column (un-owned lanes discarded below)
    int nnx = 0, nny = 0;

    // (ring variant: the forcing of the plane in its red stage as both components, of the plane in its black stage as
    //  the one component that stage still reads: six registers less than two full planes)
    double2 sw[RR][D], fwr[RR], pfS[RR], pfF[RR];
    double fwb[RR];
#pragma unroll
    for (int rr = 0; rr < RR; rr++) {
#pragma unroll
        for (int t = 0; t < D; t++) sw[rr][t] = make_double2(0.0, 0.0);
        fwr[rr] = pfS[rr] = pfF[rr] = make_double2(0.0, 0.0);
        fwb[rr] = 0.0;
    }

    // Every array of the member is addressed through a raw buffer resource (base, bytes of the volume) with the row as
    // the instruction's scalar offset and the lane's columns as its 32-bit vector offset -- no 64-bit vector address per
    // load in flight (twelve VGPRs of the 128 this kernel has), stores of columns a lane does not own dropped by range
    // (the planner admits volumes below 2 GiB).  S is read once and written once: streamed (non-temporal), so that the
    // L2 keeps what is asked for again (measured: -7.5 % fetched bytes, +3 %).
    const int vol_bytes = (int)((int64_t)zc * yc * xc * 8);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)srcS, 0, vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)pF, 0, vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)dstS, 0, vol_bytes, 0x00020000);
    const unsigned lo0 = (unsigned)lc.l0 * 8u;
    const unsigned so0 = lc.use_x ? (unsigned)st0 * 8u : 0xffffffffu;
    // (unaligned strips: a lane owns both of its columns or -- the last column of an odd row -- only the first: one 16-byte
    //  store, one 8-byte)
    const unsigned so01 = (lc.use_x && lc.use_y) ? (unsigned)st0 * 8u : 0xffffffffu;
    const unsigned sox = (lc.use_x && !lc.use_y) ? (unsigned)st0 * 8u : 0xffffffffu;
    (void)so01; (void)sox;
    constexpr int NTAUX = 2;                             // cache-policy operand of S loads / stores: bit 1 = nt on this target
    auto ldrow = [&](__amdgpu_buffer_rsrc_t rs, int soff, auto auxtag, unsigned drop = 0u) {
        constexpr int AUX = decltype(auxtag)::value;
        double2 v;
        // ONE 16-byte request per lane also where the strips are not 16-byte aligned (odd xc, odd strides: the hardware
        // takes 8-byte-aligned addresses; round 4 split these into two 8-byte requests: fixed x, 721 columns 2.34 against
        // 2.74e11 for 720).  A lane's .y is the element behind its .x wherever .y is a column that is ever read: even xc
        // (periodic or not), odd xc with fixed x (the last real column sits in an .x slot, what follows it is never
        // read), and the ring layout of the odd-xc periodic seam, whose seam lanes' .y -- column 0 of the next row, or
        // nothing -- is replaced by the mirror of column xc-1 where the plane enters the window.
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lo0 | drop), soff, AUX);
        v.x = __hiloint2double((int)t[1], (int)t[0]); v.y = __hiloint2double((int)t[3], (int)t[2]);
        return v;
    };
    auto ldS = [&](int soff) { return ldrow(rsS, soff, std::integral_constant<int, NTAUX>{}); };
    auto ldF = [&](int soff, unsigned drop = 0u) { return ldrow(rsF, soff, std::integral_constant<int, 0>{}, drop); };
    const int rowbytes = (int)(xc * 8);
    auto plane_off = [&](int p, int rr) {                // byte offset of the lane's row in plane p (clamped), a scalar
        const int pr = p > zc - 1 ? zc - 1 : (p < 0 ? 0 : p);
        return (pr * yc + jr[rr]) * rowbytes;
    };
    // the (plane, row) record through the scalar unit: {A[k+1], A[k], B[j+1], B[j], C, factor, predicate, -}
    struct Rec { double aP, a0, bP, b0, c, rq; unsigned long long rok; };
    auto record = [&](int p, int rr) {
        const int pr = p > zc - 1 ? zc - 1 : (p < 0 ? 0 : p);
        const xinv_cdouble_ptr q = rowf + (unsigned)((pr * yc + jr[rr]) * 8);
        Rec e;
        e.aP = q[0]; e.a0 = q[1]; e.bP = q[2]; e.b0 = q[3]; e.c = q[4]; e.rq = q[5];
        e.rok = (unsigned long long)__double_as_longlong(q[6]);
        return e;
    };

    // one point update of component X of row rr on the plane in slot sk (k+1 in skp, k-1 in skm): the expression
    // of k_fused3d<UNI>, the increment added under the predicate as EXEC
    // (fixt: the seam lanes' pass -- X == 0, east operand = the next lane's .x, the new column 0; lanes: rs.lanes)
    auto update = [&](auto rtag, auto xt, int sk, int skp, int skm, const Rec &e, double jP, double jM, double f,
                      auto fixt) {
        constexpr int rr = decltype(rtag)::value;
        constexpr int X = decltype(xt)::value;
        constexpr bool FIX = decltype(fixt)::value;
        static_assert(!FIX || X == 0, "column xc-1 sits in an .x slot");
        double w, ee;
        row_neighbours<X>(sw[rr][sk], w, ee);
        if constexpr (FIX) ee = xinv_lane_down(sw[rr][sk].x);
        const unsigned long long lanes64 = FIX ? rs.lanes : (X ? oky64 : okx64);
        const double sC = comp<X>(sw[rr][sk]), sKP = comp<X>(sw[rr][skp]), sKM = comp<X>(sw[rr][skm]);
        if constexpr (FMA) {                             // XINV_FLAG_FMA: the oracle's XO_FMA form, finished by one fma under EXEC
            const double ya = __builtin_fma(e.aP, sKP - sC, -(e.a0 * (sC - sKM)));
            const double yb = __builtin_fma(e.bP, jP - sC, -(e.b0 * (sC - jM)));
            const double yc_ = __builtin_fma(e.c, ee - sC, -(e.c * (sC - w)));
            double t = __builtin_fma(ya, a.sc_.ratio2Sqr, __builtin_fma(yb, a.sc_.ratio1Sqr, yc_));
            t = __builtin_fma(-f, a.sc_.delxSqr, t);
            const double v = xinv_fma_where_ne(sC, t, e.rq, f, u, lanes64 & e.rok);
            setc<X>(sw[rr][sk], v);
            return v;
        }
        double temp = (
            (
                e.aP * (sKP - sC) -
                e.a0 * (sC - sKM)
            ) * a.sc_.ratio2Sqr + (
                e.bP * (jP - sC) -
                e.b0 * (sC - jM)
            ) * a.sc_.ratio1Sqr + (
                e.c * (ee - sC) -
                e.c * (sC - w)
            )
        ) - f * a.sc_.delxSqr;
        temp *= e.rq;
        const double v = xinv_add_where_ne(sC, temp, f, u, lanes64 & e.rok);
        setc<X>(sw[rr][sk], v);
        return v;
    };
    // SM marches: column xc-1 after its half-sweep's pass, then P mirrors it again
    auto seam_fix = [&](auto rtag, int sk, int skp, int skm, const Rec &e, double jP, double jM, double f) {
        constexpr int rr = decltype(rtag)::value;
        const double v = update(rtag, std::integral_constant<int, 0>{}, sk, skp, skm, e, jP, jM, f, std::true_type{});
        sw[rr][sk].y = xinv_bitsel64(rs.lanes, v, sw[rr][sk].y);
        return v;
    };

    // one pipeline step of group GRP: plane r enters slot U; JP = parity of the wavefront's first row; SM: this
    // cross-section holds a seam lane
    auto step = [&](int r, auto gtag, auto utag, auto jtag, auto smt) {
        constexpr int GRP = decltype(gtag)::value, U = decltype(utag)::value, JP = decltype(jtag)::value;
        constexpr bool SM = decltype(smt)::value;
        constexpr int S1 = (U + 3) % D, S2 = (U + 2) % D, S3 = (U + 1) % D;
        constexpr int bw = U & 1, br = (U + 1) & 1;
#define XROW(rr) ((1 + (U & 1) + JP + (rr)) & 1)          /* component row rr touches in this step */

        // ---- plane r enters; the forcing of plane r-1 arrives (it is first read in this step); requests one plane ahead
#pragma unroll
        for (int rr = 0; rr < RR; rr++) {
            if (GRP == 0) {
                sw[rr][U] = pfS[rr];
                if constexpr (SM) sw[rr][U].y = xinv_bitsel64(rs.lanes, sw[rr][U].x, sw[rr][U].y);     // P mirrors column xc-1
                pfS[rr] = ldS(plane_off(r + 1, rr));
            } else {
                sw[rr][U] = ring[gw][rr][U & 1][0][lane];
            }
            // plane r-2 goes from its red to its black stage: its forcing travels on to group 1 (slot U & 1 was read by
            // group 1 in the previous step and is rewritten with S at the end of this one) and keeps one component
            if (GRP == 0) ring[gw][rr][U & 1][1][lane] = fwr[rr];
            fwb[rr] = XROW(rr) ? fwr[rr].y : fwr[rr].x;
            fwr[rr] = pfF[rr];
            if (GRP == 0) pfF[rr] = ldF(plane_off(r, rr), f_drop[rr]);
            else pfF[rr] = ring[gw][rr][U & 1][1][lane];
        }
        if constexpr (EXT && GRP == 1) {
            // the 'extend' pre-pass of sweep 2 on the entering plane (planes 1 .. zc-2): row 0 <- row 1, row yc-1 <- row yc-2
            // where the source is defined; with fixed x the corners take the diagonal neighbours (numbas.py:109-115) --
            // fused_extend_fix's rules, with the column classes worked out here from the lane's load offset instead of
            // living in two registers through the march (the whole kernel has 128)
            auto ext_fix = [&](double2 &edge, const double2 inner) {
                double sx = inner.x, sy = inner.y;
                bool doy = true;
                if (!a.per) {
                    const int cx = (int)(lo0 >> 3), xl = (int)xc - 1;
                    const double iw = xinv_lane_up(inner.y);                 // column cx - 1
                    sx = (cx == 0) ? inner.y : ((cx == xl) ? iw : inner.x);  // (0, 0) <- (1, 1); (0, xc-1) <- (1, xc-2)
                    sy = (cx + 1 == xl) ? inner.x : inner.y;
                    doy = cx + 1 <= xl;
                }
                if (sx != u) edge.x = sx;
                if (doy && sy != u) edge.y = sy;
            };
            if (r >= 1 && r <= zc - 2 && j0 <= 0 && j0 + RR > 0) {            // (wave-uniform: this wavefront holds row 0)
                xinv_unroll_steps([&](auto rtag) {
                    constexpr int rr = decltype(rtag)::value;
                    if (j0 + rr == 0) {                          // (rows 0 and 1 share a wavefront: H mod RR = 1)
                        const double2 inner = sw[rr + 1 < RR ? rr + 1 : rr][U];
                        ext_fix(sw[rr][U], inner);
                    }
                }, std::make_integer_sequence<int, RR>{});
            }
            if (r >= 1 && r <= zc - 2 && j0 <= yc - 1 && j0 + RR > yc - 1) {  // (... row yc-1)
                xinv_unroll_steps([&](auto rtag) {
                    constexpr int rr = decltype(rtag)::value;
                    if (j0 + rr == yc - 1) {                     // (rr > 0 wherever this row matters: xinv_p3_extend_ok)
                        const double2 inner = sw[rr > 0 ? rr - 1 : 0][U];
                        ext_fix(sw[rr][U], inner);
                    }
                }, std::make_integer_sequence<int, RR>{});
            }
        }
        // as loaded: what the neighbouring wavefronts' next red half-sweep reads of the first / last row
        xch[bw][0][wave][0][lane] = XROW(0) ? sw[0][U].y : sw[0][U].x;
        xch[bw][0][wave][1][lane] = XROW(RR - 1) ? sw[RR - 1][U].y : sw[RR - 1][U].x;

        // ---- red half-sweep on plane r-1
        {
            const double jMe = xch[br][0][wm][1][lane], jPe = xch[br][0][wp][0][lane];
            xinv_unroll_steps([&](auto rtag) {
                constexpr int rr = decltype(rtag)::value;
                constexpr int X = XROW(rr);
                using XT = std::integral_constant<int, X>;
                const Rec e = record(r - 1, rr);
                const double jM = (rr > 0) ? comp<X>(sw[rr > 0 ? rr - 1 : 0][S1]) : jMe;
                const double jP = (rr < RR - 1) ? comp<X>(sw[rr < RR - 1 ? rr + 1 : rr][S1]) : jPe;
                const double fX = comp<X>(fwr[rr]);
                double v = update(rtag, XT{}, S1, U, S2, e, jP, jM, fX, std::false_type{});
                if constexpr (SM && X == 0) v = seam_fix(rtag, S1, U, S2, e, jP, jM, fX);
                if (rr == 0) xch[bw][1][wave][0][lane] = v;          // red-updated: the neighbours' next black half-sweep
                if (rr == RR - 1) xch[bw][1][wave][1][lane] = v;
            }, std::make_integer_sequence<int, RR>{});
        }
        // ---- black half-sweep on plane r-2; plane r-2 leaves
        {
            const int kk = r - 2;
            const bool pin = (kk >= k0) && (kk < k1);
            const double jMe = xch[br][1][wm][1][lane], jPe = xch[br][1][wp][0][lane];
            xinv_unroll_steps([&](auto rtag) {
                constexpr int rr = decltype(rtag)::value;
                constexpr int X = XROW(rr);
                using XT = std::integral_constant<int, X>;
                const Rec e = record(kk, rr);
                const double jM = (rr > 0) ? comp<X>(sw[rr > 0 ? rr - 1 : 0][S2]) : jMe;
                const double jP = (rr < RR - 1) ? comp<X>(sw[rr < RR - 1 ? rr + 1 : rr][S2]) : jPe;
                const double fX = fwb[rr];
                update(rtag, XT{}, S2, S1, S3, e, jP, jM, fX, std::false_type{});
                if constexpr (SM && X == 0) seam_fix(rtag, S2, S1, S3, e, jP, jM, fX);
            }, std::make_integer_sequence<int, RR>{});
#pragma unroll
            for (int rr = 0; rr < RR; rr++) {
                const double2 t = sw[rr][S2];
                if (GRP == 0) {
                    ring[gw][rr][U & 1][0][lane] = t;
                }
                if (pin && row_use[rr]) {                // wave-uniform: an owned row of an owned plane
                    xinv_norm_row(nsx, nsy, nnx, nny, t.x, t.y, u);
                    if (GRP == 1) {
                        const int doff = (kk * yc + (j0 + rr)) * rowbytes;
                        typedef unsigned xinv_v4u_ __attribute__((__vector_size__(16)));
                        typedef unsigned xinv_v2u_ __attribute__((__vector_size__(8)));
                        if (AL) {                            // (use_x == use_y on aligned strips)
                            const xinv_v4u_ tv = {(unsigned)__double2loint(t.x), (unsigned)__double2hiint(t.x),
                                                  (unsigned)__double2loint(t.y), (unsigned)__double2hiint(t.y)};
                            __builtin_amdgcn_raw_buffer_store_b128(tv, rsD, (int)so0, doff, NTAUX);
                        } else {
                            const xinv_v4u_ tv = {(unsigned)__double2loint(t.x), (unsigned)__double2hiint(t.x),
                                                  (unsigned)__double2loint(t.y), (unsigned)__double2hiint(t.y)};
                            const xinv_v2u_ tx = {(unsigned)__double2loint(t.x), (unsigned)__double2hiint(t.x)};
                            __builtin_amdgcn_raw_buffer_store_b128(tv, rsD, (int)so01, doff, NTAUX);
                            // (a last real column in an .x slot: odd xc -- with the seam only in the cross-sections that hold it)
                            if constexpr (SM || !SEAM) __builtin_amdgcn_raw_buffer_store_b64(tx, rsD, (int)sox, doff, NTAUX);
                        }
                    }
                }
            }
        }
#undef XROW
        xinv_pipe_barrier();
    };

    // Global steps g = 0, 1, ...: group 0 enters plane rstart + g, group 1 plane rstart + g - 3 (the plane group 0
    // finished in step g - 1).  rstart is a multiple of D planes below k0 - H (slot indices are compile-time; k0 is
    // a multiple of D); the last owned plane, k1 - 1, leaves group 1 in step k1 + 4 - rstart.
    const int rstart = (k0 >= D) ? k0 - D : 0;
    const int gend = k1 + 4 - rstart;
    auto march = [&](auto gtag, auto jtag, auto smt) {
        constexpr int GRP = decltype(gtag)::value;
        if (GRP == 0) {
#pragma unroll
            for (int rr = 0; rr < RR; rr++) pfS[rr] = ldS(plane_off(rstart, rr));
        }
#pragma unroll
        for (int rr = 0; rr < RR; rr++) pfF[rr] = ldF(plane_off(rstart - 1 - 3 * GRP, rr), f_drop[rr]);
        for (int gb = 0; gb <= gend; gb += D) {
            xinv_unroll_steps([&](auto utag) {
                constexpr int Ug = decltype(utag)::value;                // global step mod D
                constexpr int U = (Ug + (GRP ? 1 : 0)) % D;              // slot of the entering plane: (g - 3) mod D for group 1
                step(rstart + gb + Ug - 3 * GRP, gtag, std::integral_constant<int, U>{}, jtag, smt);
            }, std::make_integer_sequence<int, D>{});
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using SMF = std::false_type;
    using SMT = std::integral_constant<bool, SEAM>;
    if (SEAM && rs.any) {                                    // (the same for every wavefront of the workgroup: one strip)
        if (grp == 0) { if (j0 & 1) march(I0{}, I1{}, SMT{}); else march(I0{}, I0{}, SMT{}); }
        else          { if (j0 & 1) march(I1{}, I1{}, SMT{}); else march(I1{}, I0{}, SMT{}); }
    } else {
        if (grp == 0) { if (j0 & 1) march(I0{}, I1{}, SMF{}); else march(I0{}, I0{}, SMF{}); }
        else          { if (j0 & 1) march(I1{}, I1{}, SMF{}); else march(I1{}, I0{}, SMF{}); }
    }

    if (a.no_ctl) return;
    // wavefronts of group g hold the tile's share of sweep g+1 (only the columns a lane owns count)
    double acc[K] = {0.0, 0.0};
    int cnt[K] = {0, 0};
    {
        double s = 0.0;
        s += (lc.use_x ? nsx : 0.0);
        s += (lc.use_y ? nsy : 0.0);
        const int n = (lc.use_x ? nnx : 0) + (lc.use_y ? nny : 0);
        acc[0] = grp == 0 ? s : 0.0; acc[1] = grp == 1 ? s : 0.0;
        cnt[0] = grp == 0 ? n : 0;   cnt[1] = grp == 1 ? n : 0;
    }
    __syncthreads();                                     // (every wavefront is done with xch: the norm tail's scratch)
    unsigned long long *pw = a.psum + (size_t)m * XINV_KMAX * NB * XINV_PW;
    char *scr = reinterpret_cast<char *>(&xch[0][0][0][0][0]);
    xinv_norm_publish<K, NW, true>(acc, cnt, wave, lane, NB, T, tag, pw, scr);
    if (whole && a.nkc > 1) {                            // the slots of the chunks this march covered: nothing to add
        const unsigned long long hi = (unsigned long long)tag << 32;
        const int i = (int)threadIdx.x - 64;             // (K * (nkc - 1) <= 30 words: the second wavefront's lanes)
        if (i >= 0 && i < K * (a.nkc - 1)) {
            const int s = i / (a.nkc - 1), c = 1 + i - s * (a.nkc - 1);
            unsigned long long *q = pw + ((size_t)s * NB + (size_t)c * NT + Tj) * XINV_PW;
            __hip_atomic_store(q + 0, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + 2, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!reducer) return;
    __syncthreads();                                     // (the publish step's LDS scratch is reused by the reducer)
    xinv_norm_reduce<K, NW, true>(wave, lane, NB, tag, pw, ctl, a.stop, 0.0, 0, scr);
}
