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
