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

// Tile id -> strip and owned rows [y0, y1).  Ids [0, nstrip nrb): row block rb = id / nstrip of strip id % nstrip (fixed
// height RY, or RY == 0: the yc rows split evenly, boundaries rounded to even rows).  With the odd-xc periodic seam the
// tiles of the EDGE strips run an extra pass in every other half-sweep (the seam lanes') and a launch of one round of workgroups
// ends with them; their row blocks are therefore cut in `parts` pieces (nsplit = edge strips x (parts - 1) = the extra
// tile groups; the edge strips are the last one and, with more than one strip, strip 0): id (rb, edge strip) is the
// first piece, ids nstrip nrb + (e (parts - 1) + piece - 1) nrb + rb the later pieces of edge strip e (e = 0: the last
// strip, 1: strip 0).  Pieces start on even rows like every tile; a piece without rows is an idle tile (y0 >= y1).
struct TileRows { int strip; int64_t y0, y1; };
__host__ __device__ inline TileRows xinv_tile_rows(int wt, int nstrip, int nrb, int nsplit, int64_t yc, int RY)
{
    TileRows t;
    int rb, part = -1;
    const int edges = nstrip == 1 ? 1 : 2, parts = nsplit > 0 ? nsplit / edges + 1 : 1;
    if (wt < nstrip * nrb) {
        rb = wt / nstrip; t.strip = wt - rb * nstrip;
        if (nsplit > 0 && (t.strip == nstrip - 1 || t.strip == 0)) part = 0;
    } else {
        const int q = wt - nstrip * nrb, g = q / nrb, e = g / (parts - 1);
        rb = q - g * nrb; t.strip = (e == 0) ? nstrip - 1 : 0; part = 1 + g - e * (parts - 1);
    }
    if (RY > 0) { t.y0 = (int64_t)rb * RY; t.y1 = (t.y0 + RY < yc) ? t.y0 + RY : yc; }
    else {
        t.y0 = (((int64_t)rb * yc) / nrb) & ~(int64_t)1;
        t.y1 = (rb + 1 == nrb) ? yc : ((((int64_t)(rb + 1) * yc) / nrb) & ~(int64_t)1);
    }
    if (part >= 0) {
        const int64_t y0 = t.y0, y1 = t.y1, len = y1 - y0;
        if (part > 0) { const int64_t c = y0 + ((((len * part) / parts) + 1) & ~(int64_t)1); t.y0 = c < y1 ? c : y1; }
        if (part < parts - 1) { const int64_t c = y0 + ((((len * (part + 1)) / parts) + 1) & ~(int64_t)1); t.y1 = c < y1 ? c : y1; }
    }
    return t;
}

// Dispatch order of a seam launch.  The edge strips' tiles are the slow ones (two passes per half-sweep) and the launch
// ends with the last of them: they go FIRST, spread over the XCDs -- with the later pieces' ids at the end of the id
// space (above) the chunk mapping handed all of them to ONE XCD, the one dispatched last (measured: cutting the blocks
// made a launch slower, profiles/r05_seam_rates.txt).
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
// xinv_seam_tile: index in the heavy-first sequence -> tile id of xinv_tile_rows (heavy: the pieces of the edge strips'
// row blocks, row block fastest; light: the other strips' tiles in id order).
__host__ __device__ inline int xinv_seam_tile(int s, int nstrip, int nrb, int nsplit)
{
    const int edges = nstrip == 1 ? 1 : 2, parts = nsplit / edges + 1, nh = edges * parts * nrb;
    if (s < nh) {
        const int g = s / nrb, rb = s - g * nrb, e = g / parts, part = g - e * parts;
        return part == 0 ? rb * nstrip + (e == 0 ? nstrip - 1 : 0) : nstrip * nrb + (e * (parts - 1) + part - 1) * nrb + rb;
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
