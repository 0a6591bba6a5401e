// CPU check of xinvert_amd/csrc/xinv_tiles.h (built and run by tests/test_tiles.py): for every geometry tried,
//  - the tiles of a strip partition the rows [0, yc), every tile starts on an even row;
//  - dispatch position -> xinv_heavy_first -> xinv_seam_tile visits every tile id exactly once, the edge strips' tiles in the
//    first rounds, eight consecutive positions (one per XCD) at a time.
#include "xinv_tiles.h"
#include <cstdio>
#include <vector>

static int fail(const char *what, int nstrip, int nrb, int parts, long yc)
{
    std::printf("FAIL %s nstrip %d nrb %d parts %d yc %ld\n", what, nstrip, nrb, parts, yc);
    return 1;
}

int main()
{
    long cases = 0;
    const int strips[] = {1, 2, 3, 4, 8, 31}, blocks[] = {1, 2, 3, 16, 31, 40};
    const long rows[] = {16, 17, 73, 180, 721};
    for (int nstrip : strips) for (int nrb : blocks) for (long yc : rows) {
        if (nrb * 2 > yc) continue;
        const int edges = nstrip == 1 ? 1 : 2, n = nstrip * nrb, nh = edges * nrb;
        for (int RY : {0, (int)((yc + nrb - 1) / nrb + 1) & ~1}) {
            if (RY && (long)RY * nrb < yc) continue;
            // rows of every strip covered once
            std::vector<std::vector<int>> cover((size_t)nstrip, std::vector<int>((size_t)yc, 0));
            for (int id = 0; id < n; id++) {
                const TileRows t = xinv_tile_rows(id, nstrip, nrb, yc, RY);
                if (t.strip < 0 || t.strip >= nstrip) return fail("strip", nstrip, nrb, 1, yc);
                if (t.y0 >= t.y1) continue;
                if (t.y0 & 1) return fail("odd first row", nstrip, nrb, 1, yc);
                if (t.y0 < 0 || t.y1 > yc) return fail("range", nstrip, nrb, 1, yc);
                for (long r = t.y0; r < t.y1; r++) cover[(size_t)t.strip][(size_t)r]++;
            }
            for (int s = 0; s < nstrip; s++) for (long r = 0; r < yc; r++)
                if (cover[(size_t)s][(size_t)r] != 1) return fail("cover", nstrip, nrb, 1, yc);
            cases++;
        }
        // dispatch order: a bijection, the edge strips' tiles in the first rounds
        std::vector<int> seen((size_t)n, 0);
        for (int L = 0; L < n; L++) {
            const int sq = xinv_heavy_first(L, n, nh);
            if (sq < 0 || sq >= n) return fail("sequence range", nstrip, nrb, 1, yc);
            const int id = xinv_seam_tile(sq, nstrip, nrb);
            if (id < 0 || id >= n || seen[(size_t)id]++) return fail("bijection", nstrip, nrb, 1, yc);
            const int strip = id % nstrip;
            const bool heavy = strip == 0 || strip == nstrip - 1;
            if ((L >> 3) < (nh >> 3) && !heavy) return fail("light tile in a heavy round", nstrip, nrb, 1, yc);
            if (heavy != (sq < nh)) return fail("heavy tiles first in the sequence", nstrip, nrb, 1, yc);
        }
        // workgroups of four tiles (k_fused2d, k_fused9): positions of workgroups, tiles sq * 4 + wave
        const int nwg = (n + 3) / 4;
        std::vector<int> seen4((size_t)(4 * nwg), 0);
        for (int L = 0; L < nwg; L++) {
            const int sq = xinv_heavy_first(L, nwg, nh >> 2);
            if (sq < 0 || sq >= nwg || seen4[(size_t)sq]++) return fail("workgroup bijection", nstrip, nrb, 1, yc);
        }
    }
    // the ring layout's strips (odd xc, periodic x): every strip has H halo columns a side, H + 2 on a side whose halo holds
    // the seam (west: the phantom column takes a slot; east: when column xc-1 sits in it anywhere but in the outermost slot)
    for (int H : {2, 4, 8, 12, 16})
        for (long xc = 65; xc < 1500; xc += 2) {
            const int uw = xinv_ring_uw(xc, H), nstrip = (int)((xc + uw - 1) / uw);
            if (uw != 128 - 2 * H - 2 && uw != 128 - 2 * H - 4) return fail("ring uw", nstrip, 0, H, xc);
            for (int s_ = 0; s_ < nstrip; s_++) {
                const int hw = xinv_ring_hw(xc, H, s_), he = 128 - hw - uw;
                const long lo = (long)s_ * uw - hw, hi = lo + 127;              // virtual columns of the strip's 128 slots
                if (hw < H || he < H) return fail("ring halo", nstrip, s_, H, xc);
                if (lo < 0 && hw < H + 2) return fail("ring west halo with the phantom column", nstrip, s_, H, xc);
                bool seam_east = false;                                         // column xc-1 (virtual xc-1 + n (xc+1)) in the east halo, not outermost
                for (long v = (long)s_ * uw + uw; v < hi; v++) seam_east = seam_east || ((v % (xc + 1)) == xc - 1);
                if (seam_east && he < H + 2) return fail("ring east halo with column xc-1", nstrip, s_, H, xc);
            }
            cases++;
        }
    std::printf("OK %ld geometries\n", cases);
    return 0;
}
