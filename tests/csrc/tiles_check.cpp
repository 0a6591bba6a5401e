// CPU check of xinvert_amd/csrc/xinv_tiles.h (built and run by tests/test_tiles.py): for every geometry tried,
//  - the tiles of a strip partition the rows [0, yc), every tile starts on an even row;
//  - dispatch position -> xinv_heavy_first -> xinv_seam_tile visits every tile id exactly once, the edge strips' tiles in the
//    first rounds, eight consecutive positions (one per XCD) at a time;
//  - k_pipe3d's flat grid: dispatch slot -> (member, tile, chunk) covers every tile of every member once (whole column) or
//    once per chunk (cut tiles), each member has exactly one reducer and it is the member's last slot, the whole-column
//    slots of one XCD take a contiguous band of the member's tiles; the launch shape xinv_p3_whole_tiles picks is the
//    cheapest of the three it considers; the 'extend' tiling offset puts the row pairs 0 / 1 and yc-2 / yc-1 into one
//    wavefront in every row block whose second sweep reads them (checked by laying the rows out, for every row count).
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
    // ---- k_pipe3d: slot -> (member, tile, chunk, reducer) ----
    for (int NT : {1, 3, 7, 8, 9, 16, 23, 138}) for (int nmem : {1, 2, 3, 5}) for (int nkc : {1, 2, 3, 4}) {
        const int tiles = NT * nmem;
        std::vector<int> cuts = {tiles, 0};                          // nfull: all whole, all cut, and remainders
        for (int cus : {4, 8, 13, 256}) if (tiles % cus && tiles / cus) cuts.push_back(tiles - tiles % cus);
        for (int r = 1; r < tiles; r += 5) cuts.push_back(tiles - r);
        for (int nfull : cuts) {
            if (nkc == 1 && nfull != tiles) continue;
            const int nslots = nfull + (tiles - nfull) * nkc;
            std::vector<int> seen((size_t)tiles * nkc, 0), last((size_t)nmem, -1), red((size_t)nmem, -1), nred((size_t)nmem, 0);
            std::vector<int> band_lo((size_t)nmem * 8, 1 << 30), band_hi((size_t)nmem * 8, -1), band_n((size_t)nmem * 8, 0), prev((size_t)nmem * 8, -1);
            for (int L = 0; L < nslots; L++) {
                const P3Slot sl = xinv_p3_slot(L, nfull, nkc, NT);
                if (sl.ml < 0 || sl.ml >= nmem || sl.Tj < 0 || sl.Tj >= NT || sl.kc < 0 || sl.kc >= nkc) return fail("p3 slot range", NT, nmem, nkc, nfull);
                if (sl.whole != (L < nfull) || (sl.whole && sl.kc)) return fail("p3 whole", NT, nmem, nkc, nfull);
                const size_t t = (size_t)(sl.ml * NT + sl.Tj) * nkc;
                if (sl.whole) { for (int c = 0; c < nkc; c++) if (seen[t + c]++) return fail("p3 tile twice", NT, nmem, nkc, nfull); }
                else if (seen[t + sl.kc]++) return fail("p3 chunk twice", NT, nmem, nkc, nfull);
                last[(size_t)sl.ml] = L;
                if (sl.reducer) { red[(size_t)sl.ml] = L; nred[(size_t)sl.ml]++; }
                if (sl.whole) {                                      // XCD bands: contiguous, ascending with the slot
                    const size_t b = (size_t)sl.ml * 8 + (L & 7);
                    if (sl.Tj <= prev[b]) return fail("p3 band order", NT, nmem, nkc, nfull);
                    prev[b] = sl.Tj;
                    band_lo[b] = sl.Tj < band_lo[b] ? sl.Tj : band_lo[b];
                    band_hi[b] = sl.Tj > band_hi[b] ? sl.Tj : band_hi[b];
                    band_n[b]++;
                }
            }
            for (size_t i = 0; i < seen.size(); i++) if (seen[i] != 1) return fail("p3 cover", NT, nmem, nkc, nfull);
            for (int m = 0; m < nmem; m++) {
                if (nred[(size_t)m] != 1 || red[(size_t)m] != last[(size_t)m]) return fail("p3 reducer is the member's last slot", NT, nmem, nkc, nfull);
                int next = 0;                                        // the bands follow each other in XCD order from tile 0
                for (int x = 0; x < 8; x++) {
                    const size_t b = (size_t)m * 8 + x;
                    if (!band_n[b]) continue;
                    if (band_lo[b] != next || band_hi[b] - band_lo[b] + 1 != band_n[b]) return fail("p3 band contiguous", NT, nmem, nkc, nfull);
                    next = band_hi[b] + 1;
                }
            }
            cases++;
        }
    }
    // ---- k_pipe3d: launch shape ----
    for (long tiles : {1L, 7L, 138L, 256L, 257L, 300L, 512L, 1104L, 2070L, 2208L}) for (int nk : {1, 2, 3, 4}) for (long zc : {10L, 50L, 120L})
        for (int cus : {256, -256, 64, 8}) {
            const long KC = (zc + nk - 1) / nk, c = cus < 0 ? -cus : cus;
            double cost = -1;
            const long nf = xinv_p3_whole_tiles(tiles, nk, KC, zc, cus, &cost);
            const double cf = (double)(zc + 4), cs = (double)(KC + 14);
            const double c_whole = (double)((tiles + c - 1) / c) * cf, c_all = (double)((tiles * nk + c - 1) / c) * cs;
            const long r = tiles % c;
            const double c_rem = (double)(tiles / c) * cf + (double)((r * nk + c - 1) / c) * cs;
            double want;
            if (nf == tiles) want = c_whole;
            else if (nf == 0) want = c_all;
            else if (nf == tiles - r && r && tiles / c && cus > 0) want = c_rem;
            else return fail("p3 launch shape is none of the three", (int)tiles, nk, cus, zc);
            if (nk == 1 && nf != tiles) return fail("p3 cut without chunks", (int)tiles, nk, cus, zc);
            if (cost != want) return fail("p3 launch shape cost", (int)tiles, nk, cus, zc);
            if (cost > c_whole) return fail("p3 launch shape dearer than whole columns", (int)tiles, nk, cus, zc);
            // a cut is taken only when it wins by its margin; never lose more than the margin against the cheapest
            double cheapest = c_whole;
            if (nk > 1) { cheapest = c_all < cheapest ? c_all : cheapest; if (r && tiles / c && cus > 0) cheapest = c_rem < cheapest ? c_rem : cheapest; }
            if (cost > cheapest / 0.97 + 1e-9) return fail("p3 launch shape far from the cheapest", (int)tiles, nk, cus, zc);
            cases++;
        }
    // ---- k_pipe3d, BCy = 'extend': the tiling offset, for every row count ----
    {
        const int G = 8, RR = 3, H = 4, NR = G * RR, RJ = NR - 2 * H;
        for (long yc = 4; yc <= 3000; yc++) {
            const int joff = xinv_p3_extend_joff(yc, RJ, H, RR);
            if (joff != 0 && joff != 2) return fail("p3 extend: no tiling offset", 0, 0, joff, yc);
            if (!xinv_p3_extend_ok(yc, joff, RJ, H, RR)) return fail("p3 extend: joff not ok", 0, 0, joff, yc);
            const long njb = (yc + joff + RJ - 1) / RJ;
            auto wave_of = [&](long jb, long j) { const long li = j - (jb * RJ - H - joff); return (li < 0 || li >= NR) ? -1L : li / RR; };
            for (long jb = 0; jb < njb; jb++) {
                const long own0 = jb * RJ - joff, own1 = own0 + RJ;
                // the second sweep of this block: first colour on rows [own0-1, own1+1), second on [own0, own1), rows 1..yc-2 only
                for (long j = own0 - 1; j < own1 + 1; j++) {
                    if (j < 1 || j > yc - 2) continue;
                    if (j == 1 && (wave_of(jb, 0) < 0 || wave_of(jb, 0) != wave_of(jb, 1)))
                        return fail("p3 extend: rows 0 / 1 in two wavefronts", (int)jb, 0, joff, yc);
                    if (j == yc - 2 && (wave_of(jb, yc - 1) < 0 || wave_of(jb, yc - 1) != wave_of(jb, yc - 2)))
                        return fail("p3 extend: rows yc-2 / yc-1 in two wavefronts", (int)jb, 0, joff, yc);
                }
            }
            cases++;
        }
    }
    std::printf("OK %ld geometries\n", cases);
    return 0;
}
