// xinv_tiles.h -- tile ids of the 2-D streaming kernels (k_fused2d / k_pipe2d) and the dispatch order of seam launches:
// integer arithmetic shared by the kernels and the planner (xinv_launch.h), and compiled on its own by the CPU suite
// (tests/test_tiles.py builds tests/csrc/tiles_check.cpp with g++ against this header).
#pragma once
#include <cstdint>
#ifndef __HIPCC__
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif

// Tile id -> strip and owned rows [y0, y1): row block rb = id / nstrip of strip id % nstrip (fixed height RY, or RY == 0:
// the yc rows split evenly, boundaries rounded to even rows).
struct TileRows { int strip; int64_t y0, y1; };
__host__ __device__ inline TileRows xinv_tile_rows(int wt, int nstrip, int nrb, int64_t yc, int RY)
{
    TileRows t;
    const int rb = wt / nstrip;
    t.strip = wt - rb * nstrip;
    if (RY > 0) { t.y0 = (int64_t)rb * RY; t.y1 = (t.y0 + RY < yc) ? t.y0 + RY : yc; }
    else {
        t.y0 = (((int64_t)rb * yc) / nrb) & ~(int64_t)1;
        t.y1 = (rb + 1 == nrb) ? yc : ((((int64_t)(rb + 1) * yc) / nrb) & ~(int64_t)1);
    }
    return t;
}

// Dispatch order of a seam launch (periodic x, odd xc).  The edge strips' tiles -- the ones that hold the seam: an extra pass
// in every other half-sweep -- are the slow ones and the launch ends with the last of them: they go FIRST, spread over the
// XCDs (round 4 cut their row blocks in pieces instead, whose ids at the end of the id space the chunk mapping handed to
// the one XCD dispatched last: profiles/r05_seam_rates.txt; whole row blocks, dispatched first, are faster).
// xinv_heavy_first: dispatch position L of `n` (blockIdx.x: XCD L & 7, round L >> 3) -> index in a sequence whose first
// `nh` entries are the heavy ones: rounds below nh / 8 take them eight at a time, the rest keeps the contiguous range per
// XCD of the plain mapping.  A bijection of [0, n).
__host__ __device__ inline int xinv_heavy_first(int L, int n, int nh)
{
    const int hq = (nh < n ? nh : n) >> 3, xcd = L & 7, idx = L >> 3;
    if (idx < hq) return idx * 8 + xcd;
    const int nr = n - 8 * hq, q = nr >> 3, rem = nr & 7;
    return 8 * hq + xcd * q + (xcd < rem ? xcd : rem) + (idx - hq);
}
// xinv_seam_tile: index in the heavy-first sequence -> tile id (heavy: the tiles of the last strip, then of strip 0, row
// block fastest; light: the other strips' tiles in id order).
__host__ __device__ inline int xinv_seam_tile(int s, int nstrip, int nrb)
{
    const int edges = nstrip == 1 ? 1 : 2, nh = edges * nrb;
    if (s < nh) {
        const int e = s / nrb, rb = s - e * nrb;
        return rb * nstrip + (e == 0 ? nstrip - 1 : 0);
    }
    const int k = s - nh, nl = nstrip > edges ? nstrip - edges : 1, rb = k / nl;
    return rb * nstrip + (k - rb * nl) + 1;
}

// Strips of the even-ring layout of the odd-xc periodic seam (xinv_fused.h: RING; H = halo columns a side the kernel needs
// without a seam).  A halo that holds the seam needs a column pair more: on the west because the phantom column takes a
// slot, on the east because information crosses the seam one pair faster.
//   symmetric:  every strip H + 2 | 128 - 2H - 4 owned | H + 2;
//   asymmetric: strip 0 -- its west halo wraps -- H + 2 | 128 - 2H - 2 | H, every other strip H | 128 - 2H - 2 | H + 2: valid
//               when no strip's halos hold the seam on both sides, i.e. strip 0's east halo -- H slots -- holds column xc-1
//               at most as its outermost slot (a column xc-1 further in is updated with the NEW column 0, which such a
//               halo does not hold: it would go stale two half-sweeps early): xc >= 128 - 2H - 2 + H; there are then at
//               least two strips.  3601 columns are then 33 strips of the pipelined pass, as 3600 are (34 symmetric).
// The kernels and the planner both call these with the kernel's H: no argument travels.
__host__ __device__ inline bool xinv_ring_asym(int64_t xc, int H) { return xc >= (int64_t)(128 - 2 * H - 2) + H; }
__host__ __device__ inline int xinv_ring_uw(int64_t xc, int H) { return 128 - 2 * H - (xinv_ring_asym(xc, H) ? 2 : 4); }
__host__ __device__ inline int xinv_ring_hw(int64_t xc, int H, int strip)      // west halo of a strip
{
    return xinv_ring_asym(xc, H) ? (strip == 0 ? H + 2 : H) : H + 2;
}

// ---- k_pipe3d (xinv_pipe3d.h): a FLAT grid over the members of a launch, one workgroup per CU -----------------------------
// Dispatch slot L of a launch -> which tile of which member, which planes.  With g the tile's index in the launch
// (member-major; a member has NT tiles), the slots below nfull march tile g = L through the WHOLE column; the tiles behind
// them are cut into nkc chunks: slot nfull + q marches chunk q % nkc of tile nfull + q / nkc.  `reducer`: the slot is the
// member's last in dispatch order (it adds the member's norm partials: every other slot of the member is resident or
// finished by then).  Whole-column slots take the member's tiles in an XCD-aware order: slot L lands on XCD L & 7, and the
// member's slots of one XCD take a contiguous band of its tiles -- Tj = rank of (L & 7, L) among the member's whole-column
// slots [L0, L0 + n) (with L0 a multiple of 8: xcd * (n / 8) + min(xcd, n % 8) + (L - L0) / 8).
// (Three pieces, so that the kernel can place them around its early exit; xinv_p3_slot puts them together.)
__host__ __device__ inline int xinv_p3_slot_tile(int L, int nfull, int nkc, int &kc)       // -> g; kc: the chunk (0 for a whole column)
{
    int g;                                               // tile index in the launch, member-major
    kc = 0;
    if (L < nfull) g = L;
    else { const int q = L - nfull; g = nfull + q / nkc; kc = q - (g - nfull) * nkc; }
    return g;
}
__host__ __device__ inline bool xinv_p3_slot_reduces(int L, int nfull, int nkc, int NT, int ml)
{
    const int gl = (ml + 1) * NT - 1;                    // the member's last tile
    return (gl < nfull) ? (L == gl) : (L == nfull + (gl - nfull) * nkc + nkc - 1);
}
__host__ __device__ inline int xinv_p3_slot_member_tile(int L, int nfull, int NT, int ml, int g)
{
    if (L < nfull) {
        const int L0 = ml * NT;
        const int n = (L0 + NT <= nfull) ? NT : (nfull - L0);                 // (the member's whole-column slots)
        const int xcd = L & 7;
        auto below = [](int e, int x) { return (e >> 3) * x + ((e & 7) < x ? (e & 7) : x); };       // v in [0, e): (v & 7) < x
        auto same = [](int e, int x) { return e <= x ? 0 : ((e - x + 7) >> 3); };                   // v in [0, e): (v & 7) == x
        return (below(L0 + n, xcd) - below(L0, xcd)) + (same(L, xcd) - same(L0, xcd));
    }
    return g - ml * NT;
}
struct P3Slot { int ml, Tj, kc; bool whole, reducer; };
__host__ __device__ inline P3Slot xinv_p3_slot(int L, int nfull, int nkc, int NT)
{
    P3Slot s;
    s.whole = L < nfull;
    const int g = xinv_p3_slot_tile(L, nfull, nkc, s.kc);
    s.ml = g / NT;
    s.reducer = xinv_p3_slot_reduces(L, nfull, nkc, NT, s.ml);
    s.Tj = xinv_p3_slot_member_tile(L, nfull, NT, s.ml, g);
    return s;
}

// How many tiles of a launch march the whole column (they come first).  One workgroup per CU: `tiles` tiles take
// ceil(tiles / cus) rounds of a whole march -- and when the last round holds only a few tiles (15 volumes of 50 x 360 x 720 on
// 256 CUs: 2070 tiles, 8.09 rounds) they march alone while the other CUs idle.  The plan fixes a cut of the column into nk
// chunks of KC planes; per launch: none of the tiles is cut, all are (small batches: more workgroups than tiles), or the
// remainder of the last round -- pieces that start together when the whole-column rounds end and finish in a fraction of a
// march.  Costs in pipeline steps: a whole march zc + 4, a chunk KC + 14 (four halo planes a side and the pipeline's fill).
// cus < 0: -cus compute units, never the remainder cut (xinv_options.cu_count = -1: A/B comparisons).
__host__ inline int64_t xinv_p3_whole_tiles(int64_t tiles, int nk, int64_t KC, int64_t zc, int cus_, double *cost_out = nullptr)
{
    const bool no_rem = cus_ < 0;
    const int64_t cus = cus_ < 0 ? -cus_ : cus_;
    auto cdiv64 = [](int64_t a, int64_t b) { return (a + b - 1) / b; };
    const double cf = (double)(zc + 4), cs = (double)(KC + 14);
    double best = (double)cdiv64(tiles, cus) * cf;
    int64_t nfull = tiles;
    if (nk > 1) {
        const double call = (double)cdiv64(tiles * nk, cus) * cs;
        if (call < best * 0.97) { best = call; nfull = 0; }
        const int64_t r = tiles % cus, R = tiles / cus;
        if (r && R && !no_rem) {
            const double crem = (double)R * cf + (double)cdiv64(r * nk, cus) * cs;
            if (crem < best * 0.985) { best = crem; nfull = tiles - r; }
        }
    }
    if (cost_out) *cost_out = best;
    return nfull;
}

// BCy = 'extend' on k_pipe3d (EXT): the second sweep's pre-pass is applied out of a wavefront's own registers, so rows 0 / 1
// and rows yc-2 / yc-1 each have to sit in ONE wavefront (RR adjacent rows each; the cross-section of row block jb starts at
// row jb * RJ - H - joff) -- the first pair in block 0, the second in the block that owns row yc-1 and in the block before
// it when that one owns row yc-2 or yc-3, whose second sweep reads row yc-1 through row yc-2.  xinv_p3_extend_joff: the
// shift of the row blocks that achieves it -- 0 for five row counts in eight, 2 for the others (-1: none; not reached).
__host__ __device__ inline bool xinv_p3_extend_ok(int64_t yc, int joff, int RJ, int H, int RR)
{
    if ((H + joff) % RR == RR - 1) return false;                   // row 0 would be a wavefront's last row
    const int64_t jbo = (yc - 1 + joff) / RJ;                      // the block that owns row yc-1
    auto together = [&](int64_t jb) { return ((yc - 2) - (jb * RJ - H - joff)) % RR != RR - 1; };   // row yc-2 is not a wavefront's last row
    if (!together(jbo)) return false;
    if (jbo > 0 && (yc - 1 + joff) - jbo * RJ <= 1 && !together(jbo - 1)) return false;
    return true;
}
__host__ __device__ inline int xinv_p3_extend_joff(int64_t yc, int RJ, int H, int RR)
{
    return xinv_p3_extend_ok(yc, 0, RJ, H, RR) ? 0 : (xinv_p3_extend_ok(yc, 2, RJ, H, RR) ? 2 : -1);
}
