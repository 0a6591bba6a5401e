// xinv_plan.h -- the planner of libxinv_hip.so: colouring -> path -> tiling of the chosen kernel family, the once-per-solve
// detection passes, per-row records, point-factor streams and tile lists (everything a solve derives from the
// coefficient stack and the forcing's mask; nothing from S).  Included by xinv_hip.hip only, after xinv_launch.h.
#pragma once

// ------------------------------------------------------------------ planning
// solve_dev = plan (colouring -> path -> tiling of the chosen kernel family) -> sweep loop -> finalise.
// Every plan_* step fills `Plan`; the once-per-solve detection passes (is B zero? which arrays are
// constant along x? which tiles are fully masked?) run on the caller's stream and are synchronous.

// red-black when the cross coefficient vanishes, else 4 colours; 9 for the biharmonic form; +seam colours
static int plan_colouring(const Problem &p, Workspace *ws, hipStream_t st, Plan &pl)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
    if (is3d(p.kind)) {
        pl.base = 2;
    } else if (p.kind == KIND_BIH2D) {
        pl.base = 9;                                   // radius-2 stencil: (j%3, i%3)
    } else {
        bool bzero = (p.c[1] == nullptr);
        if (p.kind == KIND_STD2DT && p.sc_.undef != 0.0) {          // cross coefficients B and C
            HIPCHK(hipMemsetAsync(ws->dflag, 0, sizeof(int), st));
            for (int q = 1; q <= 2; q++) {
                const int64_t nb = (p.sc[q] == 0) ? n : (p.nbatch - 1) * p.sc[q] + n;
                hipLaunchKernelGGL(k_any_nonzero, dim3(1024), dim3(256), 0, st, p.c[q], nb, ws->dflag);
            }
            HIPCHK(hipMemcpyAsync(ws->hflag, ws->dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            bzero = (*ws->hflag == 0);
        } else if (!bzero && p.sc_.undef != 0.0) {
            HIPCHK(hipMemsetAsync(ws->dflag, 0, sizeof(int), st));
            const int64_t nb = (p.sc[1] == 0) ? n : (p.nbatch - 1) * p.sc[1] + n;
            hipLaunchKernelGGL(k_any_nonzero, dim3(1024), dim3(256), 0, st, p.c[1], nb, ws->dflag);
            HIPCHK(hipMemcpyAsync(ws->hflag, ws->dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            bzero = (*ws->hflag == 0);
        }
        pl.base = bzero ? 2 : 4;
    }
    if (p.kind == KIND_BIH2D) {
        pl.seam = (p.BCx == XINV_BC_PERIODIC) ? (int)(p.xc % 3) : 0;     // trailing columns
        pl.ncol = 9 + 3 * pl.seam;
    } else {
        pl.seam = (p.BCx == XINV_BC_PERIODIC) && (p.xc & 1);
        pl.xc = p.xc;
        pl.ncol = pl.base + (pl.seam ? 2 : 0);
    }

    return XINV_OK;
}

// biharmonic one-pass kernel: row blocks of RB rows (multiple of 3) x strips, four wave-tiles per workgroup
static int plan_fusedbih(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
        // biharmonic: one pass per sweep; row blocks of RB rows (multiple of 3) x 180-column strips,
        // four consecutive wave-tiles per workgroup; RB from the (workgroups per CU) x (steps) model
        pl.K = 1;
        pl.aligned = false;
        const int nstrip = (int)cdiv(p.xc, XINV_BIH_OWN(p.BCx == XINV_BC_PERIODIC));
        int occ = 1;
        {   // no mixed derivatives (B == E == 0 everywhere)?  One flag pass over the two arrays.
            HIPCHK(hipMemsetAsync(ws->dflag, 0, sizeof(int), st));
            for (int q = 1; q <= 4; q += 3) {
                const int64_t nb = (p.sc[q] == 0) ? n : (p.nbatch - 1) * p.sc[q] + n;
                hipLaunchKernelGGL(k_any_nonzero, dim3(1024), dim3(256), 0, st, p.c[q], nb, ws->dflag);
            }
            HIPCHK(hipMemcpyAsync(ws->hflag, ws->dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            pl.bih_zbe = (*ws->hflag == 0) && (p.sc_.undef != 0.0);
        }
        // Where the coefficients come from (xinv_fusedbih.h): per-row records when A..I are constant along x; else the
        // vector-stream variants (round 6): A, C, D, F as streams when only they vary (A4(x, y), R(x, y) of
        // apps.py:1793-1836) and there are no mixed derivatives, all nine otherwise -- with the point-factor stream Q
        // (relaxation factor, 0 = the reference's predicate forbids the update), evaluated here, once per coefficient stack.
        pl.bih_vm = ((pl.umask & 0x1ffu) == 0x1ffu) ? 0 : ((pl.bih_zbe && (pl.umask & 0x1d2u) == 0x1d2u) ? 1 : 2);
        if (pl.bih_vm) {
            rc = ensure_dev(&ws->d_pfac, &ws->d_pfac_cap, (size_t)p.nbatch * p.yc * p.xc * sizeof(double));
            if (rc) return rc;
            PointFactorBihArgs fa;
            memset(&fa, 0, sizeof fa);
            for (int q = 0; q < 9; q++) { fa.c[q] = p.c[q]; fa.sc[q] = p.sc[q]; }
            fa.yc = p.yc; fa.xc = p.xc; fa.n = p.yc * p.xc; fa.sc_ = p.sc_; fa.q = ws->d_pfac; fa.flag = ws->dflag;
            HIPCHK(hipMemsetAsync(ws->dflag, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_point_factor_bih, dim3((unsigned)std::min<int64_t>(2048, cdiv(fa.n, 256)), (unsigned)p.nbatch, 1),
                               dim3(256), 0, st, fa);
            HIPCHK(hipMemcpyAsync(ws->hflag, ws->dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (pl.bih_vm == 1 && !(*ws->hflag & 6)) pl.bih_vm = 3;      // A == C and D == F everywhere (bitwise): two streams less
            if (*ws->hflag & 1) {                        // (a factor of exactly zero somewhere: Q == 0 could not mean "skip")
                if (opt.path == XINV_PATH_FUSED)
                    return fail_arg("biharmonic form: a relaxation factor of exactly zero on an updatable point -- the colour launches handle it");
                pl.path = XINV_PATH_COLOUR;
                return XINV_OK;
            }
        }
        {
            FusedBihArgs dummy; memset(&dummy, 0, sizeof dummy);
            xinv_launch_fusedbih(false, pl.bih_zbe, pl.bih_vm, dim3(1), st, dummy, &occ);
        }
        {   // per-row records (A..I, relaxation factor, row predicate), once per solve: xinv_fusedbih.h
            rc = ensure_dev(&ws->d_rowf, &ws->d_rowf_cap, (size_t)p.nbatch * p.yc * XINV_BIH_RW * sizeof(double));
            if (rc) return rc;
            RowFactorBihArgs ra;
            memset(&ra, 0, sizeof ra);
            for (int q = 0; q < 9; q++) { ra.c[q] = p.c[q]; ra.sc[q] = p.sc[q]; }
            ra.yc = p.yc; ra.xc = p.xc; ra.sc_ = p.sc_; ra.rowf = (double *)ws->d_rowf;
            hipLaunchKernelGGL(k_row_factor_bih, dim3((unsigned)cdiv(p.yc, 256), (unsigned)p.nbatch, 1), dim3(256), 0, st, ra);
        }
        // Time of a launch ~ (steps per tile) x f(workgroups per CU).  f measured on the round-3 kernel at 2000 x 2000
        // (profiles/r03_bih_rework.txt: rows 9 .. 33): one workgroup per CU 1.0; the second costs little (the
        // wavefronts fill each other's dependency stalls): 1.2 just above one per CU, 1.35 at two; a third 1.6 .. 1.75.
        // (the vector-stream variants, round 6, move 64 / 104 bytes per point and sweep and sit at what the fabric delivers
        //  -- 6.4 TB/s at 2000 x 2000 whatever the tile height: a second workgroup on a CU takes as long again, so the launch
        //  is planned in whole rounds of ONE workgroup per CU and what counts are the ten halo rows per tile: 15-row tiles
        //  67.5 us, 24-row tiles -- 273 workgroups -- 80 us, 27-row tiles 55 us, profiles/r06_bih_vector_streams.txt)
        const bool streams = pl.bih_vm != 0;
        auto wg_cost = [](double x) {
            if (x <= 1.0) return 1.0;
            if (x <= 2.0) return 1.15 + 0.10 * x;
            return 1.30 + 0.15 * x;
        };
        int bestRB = 3; double best = 1e300;
        for (int RB = 3; RB <= 192; RB += 3) {
            if (opt.rows_per_tile > 0 && RB != std::max(3, (opt.rows_per_tile / 3) * 3)) continue;
            const int64_t nrb = cdiv(p.yc, RB);
            const int64_t wgs = (int64_t)cdiv((int64_t)nstrip * nrb, 4) * p.nbatch;
            const int n = streams ? 1 : std::min(occ, 3);
            const int64_t cap = 256 * (int64_t)n;
            const int64_t rounds = cdiv(wgs, cap);
            const int64_t w_last = wgs - (rounds - 1) * cap;
            const double cost = ((double)(rounds - 1) * wg_cost((double)n) + wg_cost((double)w_last / 256.0)) * (double)(RB + 11 + 8);
            if (cost < best) { best = cost; bestRB = RB; }
        }
        pl.RY = bestRB;
        pl.nrb = (int)cdiv(p.yc, bestRB);
        pl.nsg = (int)cdiv((int64_t)nstrip * pl.nrb, 4) + 1;
        if (!(opt.flags & XINV_FLAG_NO_TILE_SKIP)) {
            rc = plan_tile_skip(p, pl, ws, st, opt, bestRB, XINV_BIH_OWN(p.BCx == XINV_BC_PERIODIC));
            if (rc) return rc;
        }
    return XINV_OK;
}

// 9-point forms: 4-colour fused kernel, all coefficient arrays streamed
static int plan_fused9(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
        // 9-point forms: 4-colour fused kernel, all coefficient arrays streamed
        pl.um = pl.umask = 0;
        {
            const int k9max = (p.kind == KIND_STD2D) ? 3 : 2;
            const int k9def = 2;      // bandwidth-bound: 2000x2000 general 0.74 -> 1.24e11, standard 1.07 -> 1.42e11 against K = 1
            pl.K = opt.sweeps_per_launch > 0 ? std::min(opt.sweeps_per_launch, k9max) : k9def;
        }
        pl.aligned = !(p.xc & 1) && !(p.sS & 1) && ptr_al16(p.S);
        for (int q = 0; q < p.ncoef; q++) pl.aligned = pl.aligned && ptr_al16(p.c[q]) && !(p.sc[q] & 1);
        pl.even_split = false;
        if (opt.rows_per_tile > 0) {
            pl.RY = (opt.rows_per_tile + 1) & ~1;
            pl.nrb = (int)cdiv(p.yc, pl.RY);
        } else {
            if (opt.rows_per_tile < 0) {
                pl.nrb = (int)std::max<int64_t>(1, std::min<int64_t>(-opt.rows_per_tile, p.yc / 2));
            } else {
                int occ = 1;
                FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
                fused9_dispatch(p.kind, pl.K, pl.aligned, p.BCy == XINV_BC_EXTEND, dim3(1), st, dummy, &occ, pl.seam != 0);
                // the 9-point kernels stream every coefficient array and sit at the fabric's bandwidth
                // (6+ TB/s): halo re-reads cost more than occupancy gives, so one workgroup per CU
                // (tall tiles) is the target -- measured +25 % (standard, K=1) / +21 % (general) at 2000x2000
                pl.lone = 1.0;
                pl.nrb = (int)choose_row_blocks(p.yc, cdiv(p.xc, strip9_uw(pl, pl.K)), p.nbatch, pl.K, occ, pl.lone);
            }
            pl.even_split = true;
            pl.RY = (int)cdiv(p.yc, pl.nrb);
        }
        pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, strip9_uw(pl, XINV_KMAX)) * pl.nrb, 4) + 1;
        if (pl.even_split && !(opt.flags & XINV_FLAG_NO_TILE_SKIP)) {
            int occ9 = 1;
            FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
            fused9_dispatch(p.kind, pl.K, pl.aligned, p.BCy == XINV_BC_EXTEND, dim3(1), st, dummy, &occ9, pl.seam != 0);
            rc = plan_tile_skip(p, pl, ws, st, opt, 0, strip9_uw(pl, pl.K), occ9);
            if (rc) return rc;
        }
    return XINV_OK;
}

// 3-D forms: cross-sections of NW rows marched through the planes, k chunks
static int plan_fused3d(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
        // 3-D: one sweep per launch; cross-section of NW rows per workgroup (rows_per_tile = NW)
        pl.K = 1;
        pl.RY = (opt.rows_per_tile == 8 || opt.rows_per_tile == 12 || opt.rows_per_tile == 16)
                    ? opt.rows_per_tile : 0;     // 0: decided below, once the variant is known
        pl.nsg = (int)cdiv(p.xc, pl.seam ? xinv_ring_uw(p.xc, 2) : 124);   // x strips (seam: the ring layout's, xinv_tiles.h)
        pl.aligned = !(p.xc & 1) && !(p.sS & 1) && ptr_al16(p.S);
        // only S and the forcing are read as vectors when the coefficients are per-row scalars
        pl.aligned = pl.aligned && ptr_al16(p.c[p.ncoef - 1]) && !(p.sc[p.ncoef - 1] & 1);
        if (p.kind == KIND_STD3D) {
            for (int q = 0; q < 3; q++) pl.aligned = pl.aligned && ptr_al16(p.c[q]) && !(p.sc[q] & 1);
            pl.umask = 0;
            if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
                const int idx3[3] = {0, 1, 2};
                rc = detect_xuniform_of(p, ws, st, idx3, 3, p.zc * p.yc, &pl.umask);
                if (rc) return rc;
            }
            pl.um = (pl.umask == 7u) ? 7u : 0u;
            // 16 waves x 64 lanes leaves 128 VGPRs per lane: enough only for the x-uniform variant (the others
            // spilled and are not instantiated: a request for sixteen gets twelve)
            const bool nw16_ok = (pl.um == 7u && p.BCy != XINV_BC_EXTEND);
            const bool nw16 = nw16_ok && !pl.seam;            // (seam variants: 8 or 12 wavefronts)
            if (pl.RY == 0) pl.RY = nw16 ? 16 : 12;
            if (pl.RY == 16 && !nw16) pl.RY = 12;
        } else {
            pl.um = pl.umask;                               // 0x7f: A..G are per-row scalars
            if (pl.RY == 0 || pl.RY == 16) pl.RY = 12;      // seven coefficient windows: 12 waves x 170 VGPRs
        }
        pl.nrb = (int)cdiv(p.yc, pl.RY - 4);
        // k chunks: one workgroup per CU is resident (16 / 12 waves); pick the chunk count that
        // minimises (rounds of 256 workgroups) x (planes marched per workgroup, incl. 4 halo + 4 warm-up)
        {
            const int64_t wg1 = (int64_t)pl.nsg * pl.nrb * p.nbatch;
            int best = 1; double best_cost = 1e300;
            for (int nk = 1; nk <= 16; nk++) {
                const int64_t KC = (int64_t)cdiv(cdiv(p.zc, nk), 4) * 4;
                if (nk > 1 && (KC < 16 || (int64_t)(nk - 1) * KC >= p.zc)) break;
                const int64_t rounds = cdiv(wg1 * nk, 256);
                const double cost = (double)rounds * (double)(KC + (nk > 1 ? 10 : 2));
                if (cost < best_cost * 0.97) { best_cost = cost; best = nk; }
            }
            pl.nkc = best;
            pl.KC = (int)(cdiv(cdiv(p.zc, best), 4) * 4);
        }
        // Two sweeps per pass, x-uniform coefficients, no 'extend': k_pipe3d (the two sweeps pipelined across two
        // groups of eight wavefronts, xinv_pipe3d.h) -- the one-sweep kernel sits at what HBM delivers, so halving the
        // bytes per sweep pays: 15 volumes of 50 x 360 x 720: 2.83e11 against 1.88e11, 2 volumes 1.85 against 1.46,
        // 601 x 300 x 300 2.40 against 1.57 (profiles/r03_pipe3d_first.txt); sweeps_per_launch = 1 keeps the one-sweep
        // kernel.  (Round 2's k_fused3d2, both sweeps inside every wavefront, was bound by its own latency chain
        // -- 1.45e11 -- and is gone.)
        pl.K2 = false;
        const int k2_auto = XINV_ENV_INT("XINV_3D_K2", 1);
        const bool ext3 = (p.BCy == XINV_BC_EXTEND);      // ('extend': the EXT variant -- not with the seam or the contracted arithmetic)
        if (p.kind == KIND_STD3D && pl.um == 7u && !(ext3 && (pl.seam || pl.fma || p3_extend_joff(p.yc) < 0)) && !(pl.seam && (pl.fma || p.xc < 64)) &&
            (opt.sweeps_per_launch == 2 || (opt.sweeps_per_launch == 0 && k2_auto)) &&
            opt.rows_per_tile == 0 && p.stop.mxLoop >= 1 &&
            p.zc * p.yc * 64 < ((int64_t)1 << 31) &&                      // (32-bit offsets into the record table)
            n * 8 < ((int64_t)1 << 31)) {                                 // (k_pipe3d addresses a volume through buffer resources: below 2 GiB)
            pl.K2 = true;
            pl.K = 2;
            pl.nsg2 = (int)cdiv(p.xc, pl.seam ? xinv_ring_uw(p.xc, 4) : 120);  // (odd-xc periodic seam: the ring variant's strips, xinv_tiles.h)
            pl.joff2 = ext3 ? p3_extend_joff(p.yc) : 0;
            pl.nrb2 = (int)cdiv(p.yc + pl.joff2, XINV_P3_G * XINV_P3_RR - 8);
            {
                // per-(plane, row) records of the x-uniform coefficients, relaxation factor and predicate: once per solve
                const bool shared = (p.sc[0] == 0 && p.sc[1] == 0 && p.sc[2] == 0);
                const int64_t tab = p.zc * p.yc * 8;
                rc = ensure_dev(&ws->d_rowf, &ws->d_rowf_cap, (size_t)(shared ? 1 : p.nbatch) * tab * sizeof(double));
                if (rc) return rc;
                RowFactor3Args ra;
                memset(&ra, 0, sizeof ra);
                for (int q = 0; q < 3; q++) { ra.c[q] = p.c[q]; ra.sc[q] = p.sc[q]; }
                ra.zc = p.zc; ra.yc = p.yc; ra.xc = p.xc; ra.sc_ = p.sc_;
                ra.rowf = (double *)ws->d_rowf; ra.srowf = shared ? 0 : tab;
                pl.srowf2 = ra.srowf;
                hipLaunchKernelGGL(k_row_factor3d, dim3((unsigned)cdiv(p.zc * p.yc, 256), (unsigned)(shared ? 1 : p.nbatch), 1),
                                   dim3(256), 0, st, ra);
            }
            // the cut of the column into k chunks (p3_whole_tiles: which tiles of a launch are cut is decided per launch):
            // the count that makes the launch of the whole batch cheapest
            const int64_t wg1 = (int64_t)pl.nsg2 * pl.nrb2 * p.nbatch;
            int best = 1; double best_cost = 1e300;
            for (int nk = 1; nk <= 16; nk++) {
                const int64_t KC = (int64_t)cdiv(cdiv(p.zc, nk), 4) * 4;
                if (nk > 1 && (KC < 16 || (int64_t)(nk - 1) * KC >= p.zc)) break;
                double cost;
                p3_whole_tiles(wg1, nk, KC, p.zc, pl.cus, &cost);
                if (cost < best_cost * 0.97) { best_cost = cost; best = nk; }
            }
            pl.nkc2 = best;
            pl.KC2 = (int)(cdiv(cdiv(p.zc, best), 4) * 4);
        }
    return XINV_OK;
}

// 2-D 5-point forms: x-uniform streams, sweeps per pass, rows per tile, masked-tile skipping
static int plan_fused5(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
        // which coefficient streams are constant along x (lat-lon grids: functions of latitude)
        {
            // (the forcing -- the last stream of every model -- is not looked at: no variant reads it per row)
            const int cmapS[3] = {0, 2, 3}, cmapG[6] = {0, 2, 3, 4, 5, 6}, cmapT[4] = {0, 3, 4, 5};
            const int ns = (p.kind == KIND_STD2D) ? 2 : (p.kind == KIND_STD2DT ? 3 : 5);
            const int *cmap = (p.kind == KIND_STD2D) ? cmapS : (p.kind == KIND_STD2DT ? cmapT : cmapG);
            pl.umask = 0;
            ws->act_ready = false;
            if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
                // (the activity map plan_tile_skip will ask for, with the pipelined pass's strips -- what a lat-lon problem
                //  of this size gets --, rides the same host round trip as the detection's flags)
                if (p.kind != KIND_STD2DT && opt.rows_per_tile == 0 && opt.sweeps_per_launch == 0 &&
                    !(opt.flags & (XINV_FLAG_NO_TILE_SKIP | XINV_FLAG_NO_PIPE)) && p.nbatch <= 64 &&
                    p.nbatch * p.yc * p.xc >= (int64_t)2000000) {
                    const int uw_pipe = pl.seam ? xinv_ring_uw(p.xc, 2 * XINV_PIPE_P) : XINV_PIPE_UW(1);   // (one column pair per lane: what ships)
                    if (p.xc >= uw_pipe && p.nbatch * cdiv(p.xc, uw_pipe) * (p.yc + 1) <= (int64_t)50000000) {
                        rc = issue_strip_active(p, ws, st, uw_pipe, p.kind == KIND_STD2D ? 3 : 6);
                        if (rc) return rc;
                    }
                }
                rc = detect_xuniform_of(p, ws, st, cmap, ns, p.yc, &pl.umask);
                if (rc) return rc;
            }
            pl.um = pick_um(p.kind, pl.umask);
        }
        if (opt.sweeps_per_launch > XINV_KMAX) return fail_arg("sweeps_per_launch must be 1 to 4");
        pl.aligned = !(p.xc & 1) && !(p.sS & 1) && ptr_al16(p.S);
        const int cmap3[3] = {0, 2, 3}, cmap6[6] = {0, 2, 3, 4, 5, 6}, cmap4[4] = {0, 3, 4, 5};
        const int nc = (p.kind == KIND_STD2D) ? 3 : (p.kind == KIND_STD2DT ? 4 : 6);
        for (int q = 0; q < nc; q++) {
            const int s = (p.kind == KIND_STD2D) ? cmap3[q] : (p.kind == KIND_STD2DT ? cmap4[q] : cmap6[q]);
            pl.aligned = pl.aligned && ptr_al16(p.c[s]) && !(p.sc[s] & 1);
        }
        // Sweeps per pass over HBM.  Each one costs two more window rows of registers and 2 more
        // halo rows/columns per side, and saves a pass and a launch.  Vector streams per row step
        // (S plus every coefficient array that is not x-uniform) tell the two regimes apart:
        //  - one or two (lat-lon Poisson, Gill-Matsuno): issue-bound, needs two wavefronts per
        //    SIMD.  The standard form still has them at K = 4 (3600x1800: 14.2 / 12.8 / 11.7 us
        //    per sweep for K = 2 / 3 / 4); the general form gains nothing from K = 3 (C4);
        //  - four or more (full coefficient arrays): bandwidth-bound, one workgroup per CU is as
        //    fast as two, so K = 3 pays even at one wavefront per SIMD (2000x2000 general form:
        //    2.20 -> 2.38e11 with A, C, G streamed, 1.53 -> 2.21e11 with all seven; standard form
        //    3600x1800: 2.65 -> 3.9e11), K = 4 does not (one wavefront per SIMD: 3.4e11).
        const int nvec = 1 + nc - __builtin_popcount(pl.um & ((1u << nc) - 1u));
        pl.lone = nvec <= 2 ? 1.6 : (nvec == 3 ? 1.3 : 1.0);
        // Four sweeps per pass pipelined across the four wavefronts of a workgroup (xinv_pipe2d.h) -- a quarter of
        // the tiles, four times as tall, half the recomputed halo -- for the forms whose coefficients are per-row
        // records: the standard form with per-row A and C (lat-lon Poisson) and the general form with per-row
        // A, C, D, E, F (lat-lon Gill-Matsuno).
        const int pipe_mode = XINV_ENV_INT("XINV_PIPE", 1);
        // (Only the variants whose relaxation factor is a per-row record.  With coefficient arrays that vary along
        // x every wavefront of the pipeline streams them and divides per point: built, bit-exact, and slower than
        // k_fused2d at three sweeps per pass -- C3 Stommel 2.09 against 2.56e11, C2 with every array streamed 2.95
        // against 4.06e11, profiles/r03_pipe_vector_streams.txt -- so those forms stay on k_fused2d.)
        const bool pipe_form = (p.kind == KIND_STD2D && pl.um == 3u) || (p.kind == KIND_GEN2D && pl.um == 0x1fu);
        // At every size (round 3, profiles/r03_pipe_size_crossover.txt).  Until the pass lost a quarter of its
        // instructions per step and the forcing rode the LDS ring, k_fused2d (VALU 96 % busy against ~75 %) won on
        // launches of several rounds of workgroups and the standard form switched at 1e7 points; re-measured on
        // 1 / 2 / 3 / 4 / 8 slices of 3600x1800: 6.17 / 6.76 / 7.25 / 7.21 / 7.72e11 pipelined (forcing through the
        // ring from two slices on) against 5.31 / 6.33 / 6.53 / 6.64 / 7.31e11.  The general form stops at two sweeps
        // per pass on k_fused2d (registers) and is bound by HBM at C4.  XINV_PIPE=3 restores the old crossover.
        const bool pipe_size_ok = pipe_mode != 3 || p.kind == KIND_GEN2D || p.nbatch * p.yc * p.xc <= (int64_t)10000000;
        // (k_pipe2d addresses a slice through buffer resources with signed 32-bit row offsets: slices below 2 GiB)
        const bool pipe_want = pipe_mode != 0 && pipe_form && pipe_size_ok && !(opt.flags & XINV_FLAG_NO_PIPE) &&
                               (opt.sweeps_per_launch == 0 || opt.sweeps_per_launch == XINV_PIPE_P) &&
                               (p.yc + 16) * p.xc * 8 < ((int64_t)1 << 31);
        {
            const bool hoisted_gen = (p.kind == KIND_GEN2D && nvec <= 2);
            const int ksup = (p.kind == KIND_STD2D && nvec <= 2) ? XINV_KMAX : (hoisted_gen ? 2 : 3);
            const int occ_needed = nvec >= 4 ? 1 : 2;
            if (pipe_want)
                pl.K = XINV_PIPE_P;
            else if (opt.sweeps_per_launch > 0)
                pl.K = std::min(opt.sweeps_per_launch, (p.kind == KIND_STD2D) ? XINV_KMAX : 3);
            else {
                pl.K = 2;
                for (int k = ksup; k > 2; k--) {
                    int o = 0;
                    FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
                    if (fused_dispatch(p.kind, pl.aligned, p.BCy == XINV_BC_EXTEND, pl.um, k, dim3(1), dim3(256),
                                       st, dummy, &o, pl.seam != 0, pl.fma) == 0 && o >= occ_needed) { pl.K = k; break; }
                }
            }
        }
        pl.pipe = pipe_want && pl.K == XINV_PIPE_P;
        // Round 5: the general form with A and C varying along x (Stommel with R(x, y), BASELINE configs[2]) reads the
        // relaxation factor and the update predicate of every point from one more stream (FusedGen2DQ) instead of dividing
        // and testing six operands whenever a row enters a window.  XINV_FLAG_NO_POINT_FACTOR keeps FusedGen2D.
        pl.pq = p.kind == KIND_GEN2D && (pl.um == 0x1cu || pl.um == 0u) && !pl.pipe && !pl.seam && !pl.fma &&
                !(opt.flags & XINV_FLAG_NO_POINT_FACTOR) && p.sc_.optArg != 0.0;
        if (pl.pq) {                                     // the point-factor stream, once per coefficient stack
            rc = ensure_dev(&ws->d_pfac, &ws->d_pfac_cap, (size_t)p.nbatch * p.yc * p.xc * sizeof(double));
            if (rc) return rc;
            PointFactorArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.c[0] = p.c[0]; fa.sc[0] = p.sc[0];
            for (int q = 2; q < 7; q++) { fa.c[q - 1] = p.c[q]; fa.sc[q - 1] = p.sc[q]; }
            fa.yc = p.yc; fa.xc = p.xc; fa.n = p.yc * p.xc; fa.sc_ = p.sc_; fa.q = ws->d_pfac; fa.flag = ws->dflag;
            HIPCHK(hipMemsetAsync(ws->dflag, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_point_factor, dim3((unsigned)std::min<int64_t>(2048, cdiv(fa.n, 256)), (unsigned)p.nbatch, 1),
                               dim3(256), 0, st, fa);
            HIPCHK(hipMemcpyAsync(ws->hflag, ws->dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (*ws->hflag & 1) pl.pq = false;           // (a factor of exactly zero somewhere: Q == 0 could not mean "skip")
            pl.alias_ac = pl.pq && !(*ws->hflag & 2);    // A and C bitwise equal everywhere: C is read out of A
        }
        pl.tpw = pl.pipe ? 1 : 4;
        // one column pair per lane (two -- strips of 240 owned columns -- were measured slower, 45.2 against 40.0 us at
        // 3600x1800, and are no longer instantiated: round 5)
        pl.npair = 1;
        // the forcing through the LDS ring where the launch's streams (S read + write + forcing, every member) no
        // longer fit the caches and the later wavefronts' forcing requests would go back to HBM; XINV_PIPE_FR=0|1 forces
        {
            const int fr_env = opt.pipe_fr ? (opt.pipe_fr > 0 ? 1 : 0) : XINV_ENV_INT("XINV_PIPE_FR", -1);
            const bool fr_form = (p.kind == KIND_STD2D && pl.um == 3u) || (p.kind == KIND_GEN2D && pl.um == 0x1fu);
            const bool fr_size = (double)p.nbatch * (double)p.yc * (double)p.xc * 24.0 > 2.0e8;
            pl.pipe_fr = pl.pipe && fr_form && pl.npair == 1 && (fr_env < 0 ? fr_size : fr_env != 0);
        }
        if (pl.pipe) {
            // per-row records of the x-uniform streams (+ relaxation factor and row predicate when the model hoists)
            const bool gen = (p.kind == KIND_GEN2D);
            const int nco = gen ? 5 : 2;
            const bool hoist = gen ? ((pl.um & 0x13u) == 0x13u) : ((pl.um & 3u) == 3u);
            const int nw = __builtin_popcount(pl.um & ((1u << nco) - 1u)) + (hoist ? 2 : 0);
            const int rw = nw == 0 ? 0 : (nw <= 4 ? 4 : 8);
            if (rw) {
                rc = ensure_dev(&ws->d_rowf, &ws->d_rowf_cap, (size_t)p.nbatch * p.yc * rw * sizeof(double));
                if (rc) return rc;
                RowFactorArgs ra;
                memset(&ra, 0, sizeof ra);
                ra.c[0] = p.c[0]; ra.sc[0] = p.sc[0];                                     // A
                for (int q = 2; q < (gen ? 6 : 3); q++) { ra.c[q - 1] = p.c[q]; ra.sc[q - 1] = p.sc[q]; }   // C (, D, E, F)
                ra.gen = gen ? 1 : 0; ra.um = pl.um; ra.hoist = hoist ? 1 : 0; ra.rw = rw;
                ra.yc = p.yc; ra.xc = p.xc; ra.sc_ = p.sc_; ra.rowf = (double *)ws->d_rowf;
                hipLaunchKernelGGL(k_row_factor, dim3(cdiv(p.yc, 256), (unsigned)p.nbatch, 1), dim3(256), 0, st, ra);
            }
        }
        // Rows per tile (see the cost model below).
        pl.even_split = false;
        if (opt.rows_per_tile > 0) {
            pl.RY = (opt.rows_per_tile + 1) & ~1;
            pl.nrb = (int)cdiv(p.yc, pl.RY);
        } else if (opt.rows_per_tile < 0) {              // -n: exactly n row blocks, even split
            pl.nrb = (int)std::max<int64_t>(1, std::min<int64_t>(-opt.rows_per_tile, p.yc / 2));
            pl.even_split = true;
            pl.RY = (int)cdiv(p.yc, pl.nrb);
        } else {
            // Tall tiles amortise the 4K recomputed halo rows, but the launch should put the same
            // number of workgroups on every CU: with the K = 2 kernels two workgroups fit per CU
            // (register-limited), so the target is a multiple of 512 workgroups.  Pick the row-block
            // count that minimises (workgroups per CU) x (steps per tile); rows are then split
            // evenly (measured at 3600x1800: 64 blocks of ~28 rows beat 53 blocks of 34).
            int occ = 2;                                   // workgroups of the chosen variant per CU
            {
                FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
                if (pl.pipe) xinv_launch_pipe2d(p.kind == KIND_GEN2D, pl.um, pl.npair, pl.pipe_fr, pl.aligned, p.BCy == XINV_BC_EXTEND, dim3(1), st, dummy, &occ, 0, pl.seam != 0, pl.fma);
                else fused_dispatch(p.kind, pl.aligned, p.BCy == XINV_BC_EXTEND, pl.um | (pl.alias_ac ? 2u : 0u), pl.K, dim3(1), dim3(256),
                                    st, dummy, &occ, pl.seam != 0, pl.fma, pl.pq);
            }
            // (pl.lone, set with K above: with one or two vector streams a second workgroup per CU
            // fills idle issue slots; with four or more a pair runs no faster than one, and tall
            // tiles (less halo) win -- 2000x2000 general form, A C G streamed: 40-row tiles 38.3 us,
            // 17-row tiles 44.0 us)
            const int64_t best = choose_row_blocks(p.yc, cdiv(p.xc, strip_uw(pl, pl.K, pl.pipe)),
                                                   p.nbatch, pl.K, occ, pl.lone, pl.pipe);
            pl.nrb = (int)best;
            pl.even_split = true;
            pl.RY = (int)cdiv(p.yc, pl.nrb);
        }
        // workgroups per member with the narrowest strips any K uses: sizes the partials
        // (the shorter tail / redo passes of a pipelined plan run k_fused2d: four 112-column tiles per workgroup)
        pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, strip_uw(pl, XINV_KMAX, false)) * pl.nrb, 4) + 1;
        if (pl.pipe) pl.nsg = std::max(pl.nsg, (int)cdiv(p.xc, strip_uw(pl, pl.K, true)) * pl.nrb + 1);
        if (pl.even_split && !(opt.flags & XINV_FLAG_NO_TILE_SKIP)) {
            rc = plan_tile_skip(p, pl, ws, st, opt);
            if (rc) return rc;
        }
    return XINV_OK;
}

// which path, then the tiling of its kernel family
static int plan_path(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    ws->act_ready = false;                               // (an activity map issued ahead belongs to ONE plan: issue_strip_active)
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
    // ---- path ------------------------------------------------------------------------------
    // (the odd-xc periodic seam runs inside the 2-D 5-point streaming kernels -- xinv_fused.h: SEAM -- when a strip
    //  spans at most three wraps of the row: xc >= 64; the 3-D, 9-point and biharmonic forms keep the colour launches)
    const bool seam5_ok = !pl.seam || p.xc >= 64;        // (3-D forms: the SEAM variants of k_fused3d / k_fused3dg)
    const bool fused5_ok = pl.base == 2 && seam5_ok && p.kind != KIND_BIH2D && p.kind != KIND_GEN3D;
    const bool fused9_ok = pl.base == 4 && seam5_ok && p.c[1] &&      // (seam: k_fused9's SEAM variants)
                           (p.kind == KIND_STD2D || p.kind == KIND_GEN2D);
    // biharmonic: the one-pass kernel needs A..I as per-row scalars (and xc % 3 == 0 when periodic)
    bool fusedbih_ok = false;
    if (p.kind == KIND_BIH2D) {
        pl.umask = 0;
        if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
            const int idx10[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
            rc = detect_xuniform_of(p, ws, st, idx10, 10, p.yc, &pl.umask);
            if (rc) return rc;
        }
        pl.um = pl.umask;
        // (coefficients that vary along x: the vector-stream variants of the one-pass kernel, xinv_fusedbih.h)
        fusedbih_ok = (p.BCx != XINV_BC_PERIODIC || p.xc % 3 == 0) && !(opt.flags & XINV_FLAG_NO_XUNIFORM) &&
                      p.sc_.optArg != 0.0;
    }
    // general 3-D: the fused kernel exists for x-uniform coefficients only (every 3DOcean array)
    bool fused3g_ok = false;
    if (p.kind == KIND_GEN3D && seam5_ok && opt.path != XINV_PATH_COLOUR && !(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
        const int idx7[7] = {0, 1, 2, 3, 4, 5, 6};
        rc = detect_xuniform_of(p, ws, st, idx7, 7, p.zc * p.yc, &pl.umask);
        if (rc) return rc;
        fused3g_ok = (pl.umask == 0x7fu);
    }
    const bool fused_ok = fused5_ok || fused9_ok || fused3g_ok || fusedbih_ok;
    pl.path = XINV_PATH_COLOUR;
    pl.nine = false;
    if (fused_ok && opt.path != XINV_PATH_COLOUR) { pl.path = XINV_PATH_FUSED; pl.nine = fused9_ok && !fused5_ok; }
    if (opt.path == XINV_PATH_FUSED && !fused_ok)
        return fail_arg("no fused kernel for this form (odd-xc periodic seam with xc < 64; 9-point test form; biharmonic with periodic x and xc % 3 != 0; general 3-D with coefficients that vary along x)");
    if (pl.path == XINV_PATH_COLOUR && !is3d(p.kind) && !p.c[1] && pl.base == 4)
        return fail_arg("internal: 9-point form without B");

    if (pl.path != XINV_PATH_FUSED) return XINV_OK;
    if (p.kind == KIND_BIH2D) return plan_fusedbih(p, opt, ws, st, pl);
    if (pl.nine) return plan_fused9(p, opt, ws, st, pl);
    if (is3d(p.kind)) return plan_fused3d(p, opt, ws, st, pl);
    return plan_fused5(p, opt, ws, st, pl);
}

