// xinv_hip.hip -- host driver and C-ABI of the MI355X SOR inversion engine (include/xinv.h).
//
// Replaces the reference's numba kernels behind the call boundary of xinvert/core.py
// (core.py:60-69, 130-139, 419-428).  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//
// Control flow of one solve (all batch members together, one stream):
//   k_solve_init -> [ sweep launches ... ] x check_every -> async read-back of the per-member
//   control blocks -> repeat until every member has stopped.  The stopping rule runs on the
//   device after every sweep (reducing workgroup / k_norm_final); once a member is done
//   every later launch is a no-op for it, so S holds exactly the sweep the reference stops at.
//
// Threading: solves on one device are serialised by a per-device lock (they share the cached
// workspace); different devices may be driven concurrently from different host threads.
// Statistics and the last error text are thread-local.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <exception>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/xinv.h"
#include "xinv_device.h"
#include "xinv_colour.h"
#define XINV_AUX_KERNELS            /* the detection / skip-norm helper kernels live in this unit */
#include "xinv_dispatch.h"          /* argument structs + launchers of the sweep kernels (xinv_tu_*.hip) */

#define XINV_VERSION 500
#define XINV_MEMBER_CHUNK 32768     /* members per launch: grid.y / grid.z are limited to 65535 */

// The shipped library reads NO environment variable: the planner's choices are overridden through xinv_options
// (lanes, norm_lag, pipe_fr, graph, sweeps_per_launch, flags).  The test-hooks build (build/libxinv_hooks.so) and A/B
// variant builds (-DXINV_EXPERIMENTS=1) additionally honour the XINV_* switches of rounds 2-4 where the option is 0.
#if XINV_TEST_HOOKS || XINV_EXPERIMENTS
#define XINV_ENV_INT(name, dflt) ([&] { const char *e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define XINV_ENV_INT(name, dflt) (dflt)
#endif

// The control-block mirror and the mailbox word the host SPINS on (k_ctl_mail) must be fine-grained, coherent host
// memory whatever HIP_HOST_COHERENT says: the device's stores of the blocks have to be visible before its store of the
// sequence word.
#define XINV_HOST_COHERENT (hipHostMallocCoherent | hipHostMallocMapped)
static inline void xinv_cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

#include "xinv_host.h"
#include "xinv_launch.h"

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
        if (p.kind == KIND_STD3D && pl.um == 7u && p.BCy != XINV_BC_EXTEND && !(pl.seam && (pl.fma || p.xc < 64)) &&
            (opt.sweeps_per_launch == 2 || (opt.sweeps_per_launch == 0 && k2_auto)) &&
            opt.rows_per_tile == 0 && p.stop.mxLoop >= 1 &&
            p.zc * p.yc * 64 < ((int64_t)1 << 31) &&                      // (32-bit offsets into the record table)
            n * 8 < ((int64_t)1 << 31)) {                                 // (k_pipe3d addresses a volume through buffer resources: below 2 GiB)
            pl.K2 = true;
            pl.K = 2;
            pl.nsg2 = (int)cdiv(p.xc, pl.seam ? xinv_ring_uw(p.xc, 4) : 120);  // (odd-xc periodic seam: the ring variant's strips, xinv_tiles.h)
            pl.nrb2 = (int)cdiv(p.yc, XINV_P3_G * XINV_P3_RR - 8);
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

// ------------------------------------------------------------------ the sweep loop
// What the loop leaves for finalise(): where each launch started, the final control blocks.
struct SweepRun {
    double *S2 = nullptr;
    double *buf[3] = {nullptr, nullptr, nullptr};
    int nbuf = 2;                                        // 3 with the lagged norm
    std::vector<signed char> srcb, dstb;                 // launch i swept buf[srcb[i]] into buf[dstb[i]] (see launch_idx)
    bool lag = false;
    std::vector<int64_t> bound;                          // bound[i] = sweeps before launch i (fused path)
    int64_t launched = 0, nlaunch = 0;
    double ms_total = 0.0;
    const XinvCtl *hc = nullptr;                         // the slot holding the final control blocks
    std::vector<int> rec_where;                          // watchdog recovery: buffer index of a recovered member's final state (-1: not recovered)
    int Kf = 1;
    int lanes = 1;                                       // independent launch chains the batch was cut into
    // the replayed chunk of small problems: lives until finalise() has drained the stream (replays
    // queued after the last poll may still be executing when run_sweeps returns)
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t stream = nullptr;
    // xinv_options.timing == 2 (one lane, plain launches): an event before the first sweep launch and one behind each
    // of them, on the launches' own stream -- per-launch durations (launch_us_min / avg / max) without a profiler
    std::vector<hipEvent_t> lev;
    ~SweepRun()
    {
        if (graph_exec) { (void)hipStreamSynchronize(stream); (void)hipGraphExecDestroy(graph_exec); }
        if (!lev.empty()) { (void)hipStreamSynchronize(stream); for (hipEvent_t e : lev) (void)hipEventDestroy(e); }
    }
};
#define XINV_MAX_LAUNCH_EVENTS 8192

// one sweep launch of the planned kernel (fused: k sweeps from src into dst; colour path: one sweep in place)
static int launch_planned(const Problem &p, const Plan &pl, Workspace *ws, hipStream_t s, int k,
                          const double *src, double *dst, int64_t member0, int64_t nmem, int force, int no_ctl,
                          unsigned lag_tag = 0, NormLagArgs *lag_out = nullptr, const NormLagArgs *lag_prev = nullptr,
                          bool prepass = true)
{
    if (pl.path != XINV_PATH_FUSED) return launch_colour_sweep(p, pl, ws, s);
    return (p.kind == KIND_BIH2D)   ? launch_fusedbih(p, pl, src, dst, ws, s, member0, nmem, force, no_ctl, lag_tag, lag_out, lag_prev, prepass)
         : (p.kind == KIND_GEN3D)   ? launch_fused3dg(p, pl, src, dst, ws, s, member0, nmem, force, no_ctl)
         : (p.kind == KIND_STD3D)   ? launch_fused3d(p, pl, k, src, dst, ws, s, member0, nmem, force, no_ctl)
         : pl.nine                  ? launch_fused9(p, pl, k, src, dst, ws, s, member0, nmem, force, no_ctl, lag_tag, lag_out, lag_prev)
                                    : launch_fused(p, pl, k, src, dst, ws, s, member0, nmem, force, no_ctl, lag_tag, lag_out, lag_prev);
}

// A stream-ordered plan solve returns with its redo pass and the copy of the final state out of S2 / S3 still queued
// (finalise); the workspace's own event sits behind them.  Whoever writes those buffers or a plan's records next -- the
// next solve, a plan build / refresh -- makes ITS stream wait for the event (no host wait, no handle of the earlier
// caller's stream: that stream may be gone by now); plan_free waits on the host before it frees.
static int tail_wait(Workspace *ws, hipStream_t st, bool host = false)
{
    if (!ws->tail_pending) return XINV_OK;
    if (host) HIPCHK(hipEventSynchronize(ws->ev_tail));
    else HIPCHK(hipStreamWaitEvent(st, ws->ev_tail, 0));
    if (host) ws->tail_pending = false;                  // (a stream wait orders only `st`: another stream must wait again;
    return XINV_OK;                                      //  waiting on an event that has completed costs nothing)
}

// sweep loop in lanes (run_sweeps): how many independent launch chains the batch is cut into.
// Measured with 1 and 2 lanes on one box (profiles/r04_lanes.txt; XINV_LANES=n forces n, 0 or 1 = off):
//   3600x1800 x 2/3/4/6/8/12/16/32 members  +6 +7 +6 +8 +10 +5 +8 +1.5 %      (5 members: 0)
//   1440x720 general form x 4/8/12/16/24/32/64/128   0 +9 +6 +7 +8.5 +7 +5.5 +2 %
//   360x180 x 8/16/32/64/100/200/365/1000    -1 -6 +1 +11 +13 +8 +5..12 +3.5 %      73x144 x 365/3650  +5 +5 %
//   720x360x50 x 2/3/4/6/8/15/16/30 volumes  +33 -3 +17 +6 +4 +1 -2 -2 %   (one workgroup per CU: the gain is the tail of a
//                                             launch of one or two rounds; with eight rounds there is none to win)
// With the lagged norm (a launch of at most one round; every lane keeps its own pending evaluation):
//   1440x720 general form x 2/3/4  -5 +9 +4 %
// Three or four lanes were no better than two; lanes on streams of the lowest priority were erratic (-30 % on small
// batches).  A pass of a few microseconds is bound by the host's launch rate, which lanes double: the rule wants an
// estimated 20 us (64 slices of 360x180, 365 of 144x73).
static int lane_rule(const Problem &p, double est_pass_us)
{
    if (p.nbatch < 2 || est_pass_us < 20.0) return 1;
    if (is3d(p.kind)) return p.nbatch <= 8 ? 2 : 1;
    return 2;
}

// workspace, then chunks of launches with pipelined polling of the device-side stop flags
static int run_sweeps(const Problem &p, const Plan &pl, const xinv_options &opt, Workspace *ws, hipStream_t st,
                      SweepRun &R)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
    // ---- workspace ---------------------------------------------------------------------------
    rc = tail_wait(ws, st);                              // (the previous plan solve's copy into its caller's S reads S2 / S3)
    if (rc) return rc;
    rc = ensure_dev(&ws->ctl, &ws->ctl_cap, (size_t)p.nbatch * sizeof(XinvCtl));
    if (rc) return rc;
    if (ws->hctl_cap < (size_t)p.nbatch) {             // two slots: polling is pipelined
        if (ws->hctl) HIPCHK(hipHostFree(ws->hctl));
        HIPCHK(hipHostMalloc((void **)&ws->hctl, 2 * (size_t)p.nbatch * sizeof(XinvCtl), XINV_HOST_COHERENT));
        ws->hctl_cap = (size_t)p.nbatch;
    }
    size_t pbytes;
    if (pl.path == XINV_PATH_FUSED)
        pbytes = (size_t)p.nbatch * XINV_KMAX *
                 (is3d(p.kind) ? std::max((size_t)pl.nsg * pl.nrb * std::max(1, pl.nkc),
                                          pl.K2 ? (size_t)pl.nsg2 * pl.nrb2 * std::max(1, pl.nkc2) : (size_t)0)
                               : (size_t)pl.nsg) *
                 (3 * sizeof(unsigned long long));      // three tagged words per partial
    else
        pbytes = (size_t)p.nbatch * XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long));
    // Lagged norm (5-point 2-D kernels): the sweep kernel only publishes its partials; an extra workgroup
    // of the NEXT launch adds them and applies the stop rule while that pass's tiles run.  Measured at
    // 3600x1800, K = 4 (profiles/r02_norm_lag_experiment.txt): 47.6 us per launch with the in-kernel
    // reducer (a global round trip after the last tile), 43.7 us publishing only, 41.7 us without any
    // norm; a reducer kernel on a second stream (events both ways) was slower than either: 50.6 us.
    // The decision about pass i arrives while pass i+1 runs, so S rotates through THREE buffers: pass
    // i+2 -- the first that could overwrite the source of pass i -- starts after reducer i has finished,
    // finds the member stopped and does nothing, and finalise() re-sweeps from the intact source.
    const bool lag_env = opt.norm_lag ? opt.norm_lag > 0 : XINV_ENV_INT("XINV_LAG", 1) != 0;
    // Only where a member has many workgroups: the reducing workgroup is one more per member and launch,
    // and with one or two tile workgroups per member (365 slices of 73x144) it would double the launch.
    const int64_t own_cols = (p.kind == KIND_BIH2D) ? XINV_BIH_OWN(p.BCx == XINV_BC_PERIODIC)
                           : (pl.path == XINV_PATH_FUSED && pl.pipe) ? strip_uw(pl, pl.K, true)
                                                    : (pl.nine ? strip9_uw(pl, std::max(1, pl.K)) : strip_uw(pl, std::max(1, pl.K), false));
    const int tpw = (pl.path == XINV_PATH_FUSED && pl.pipe) ? 1 : 4;
    const int64_t wg_member = pl.skip ? pl.ntl / tpw : (int64_t)cdiv((int64_t)cdiv(p.xc, own_cols) * pl.nrb, tpw);
    // ... and only where a launch is one or two rounds of workgroups: with many rounds (64 Gill-Matsuno
    // members: 3.97e11 without, 3.73e11 with) the in-kernel reducer's wait already hides behind other tiles.
    const bool lag_cand = lag_env && pl.path == XINV_PATH_FUSED && wg_member >= 32 &&
                          wg_member * p.nbatch <= 1024 && !is3d(p.kind);     // every 2-D streaming kernel
    pbytes = (pbytes + 255) & ~(size_t)255;
    ws->partials_half = pbytes;
    rc = ensure_dev(&ws->partials, &ws->partials_cap, lag_cand ? 2 * pbytes : pbytes);
    if (rc) return rc;
    const size_t pclear = (pl.path == XINV_PATH_FUSED) ? (lag_cand ? 2 * pbytes : pbytes) : 0;   // tagged partials: no stale sequence numbers
    double *&S2 = R.S2;
    if (pl.path == XINV_PATH_COLOUR && p.kind == KIND_BIH2D) {       // side buffer of the row-class kernel
        rc = ensure_dev(&ws->S2, &ws->S2_cap, (size_t)((p.nbatch - 1) * p.sS + n) * sizeof(double));
        if (rc) return rc;
    }
    if (pl.path == XINV_PATH_FUSED) {
        const size_t need = (size_t)((p.nbatch - 1) * p.sS + n) * sizeof(double);
        rc = ensure_dev(&ws->S2, &ws->S2_cap, need);
        if (rc) return rc;
        S2 = ws->S2;
    }

    // (control blocks and partials in ONE launch: a dispatch less on the way to the first sweep launch.  Measured and not
    //  kept: the NEXT solve's initialisation queued behind a plan solve -- the sweep launch waits for it either way)
    hipLaunchKernelGGL(k_solve_init, dim3((unsigned)std::max<int64_t>(cdiv(p.nbatch, 256), std::min<int64_t>(256, cdiv((int64_t)(pclear / 16), 256)))),
                       dim3(256), 0, st, ws->ctl, p.nbatch, (uint4 *)ws->partials, (int64_t)(pclear / 16));

    // ---- sweep loop ----------------------------------------------------------------------------
    const int64_t max_sweeps = p.stop.mxLoop + 1;       // numbas.py:410: loop >= mxLoop stops
    const int Kf = R.Kf = (pl.path == XINV_PATH_FUSED) ? pl.K : 1;
    int check_every = opt.check_every;
    const double sweep_rate = (pl.path != XINV_PATH_FUSED) ? 4.0e4 : (pl.pipe ? 6.0e5 : (is3d(p.kind) ? 2.5e5 : 3.0e5));   // points per us
    const double est_pass_us = (double)p.nbatch * (double)n * Kf / sweep_rate;
    if (check_every <= 0) {
        // poll the device stop flags about every 2 ms of sweeping (fused kernels run at roughly
        // 2e5 points per microsecond, the colour path at a quarter of that); launches issued after
        // a member has stopped are no-ops of a few microseconds each
        // (round 3: the pipelined 2-D pass runs at 6-7e5 points per microsecond; with the round-1 constant a 500-sweep
        //  solve at 3600x1800 was polled 18 times -- each poll ends a chunk: one-workgroup norm reduction of the lagged
        //  launch + control-block copy, ~10 us of idle GPU -- 4 % of the solve)
        const double est_us = std::max(4.0, est_pass_us);
        check_every = (int)std::min(256.0, std::max(4.0, 2000.0 / est_us));
        // (a non-positive tolerance can never stop a solve -- the reference tests' idiom for a fixed number of sweeps,
        //  tests/test_GeoAdjustment.py:31 -- only an overflow or, in the standard form, a zero norm can: nothing worth a
        //  poll every 2 ms, each of which holds the next launch back for ~10 us)
        if (p.stop.tolerance <= 0.0 && pl.path == XINV_PATH_FUSED) check_every = 256;
    }
    R.buf[0] = p.S; R.buf[1] = S2; R.buf[2] = nullptr;
    double **buf = R.buf;
    std::vector<int64_t> &bound = R.bound;
    int64_t &launched = R.launched, &nlaunch = R.nlaunch;
    double &ms_total = R.ms_total;
    bool all_done = false;
    // A chunk = `check_every` launches followed by an asynchronous copy of the control blocks.
    // Polling is pipelined: chunk c+1 is queued BEFORE the host waits for chunk c's copy, so the
    // GPU never idles on the host's reaction time; once every member has stopped, the launches
    // already queued are no-ops (each kernel returns on ctl.done).
    // one sweep launch (fused: K sweeps from buf[cur] into buf[cur^1]; colour path: one sweep in place)
#if XINV_EXPERIMENTS
    static const int exp_noctl = XINV_ENV_INT("XINV_EXP_NOCTL", 0);   // timing experiment (variant builds only): launches without norm / stop rule
#else
    constexpr int exp_noctl = 0;
#endif
    auto launch_one = [&](hipStream_t s, int cur, int k) -> int {
        return launch_planned(p, pl, ws, s, k, buf[cur], buf[cur ^ 1], 0, p.nbatch, exp_noctl, exp_noctl);
    };
    // Small problems are bound by the host's launch rate (a 151x251 coloured sweep is six launches
    // of 2-3 us each): a full chunk is captured once into a hipGraph on an engine-owned stream and
    // replayed into the caller's stream.  The chunk has an even number of launches, so the
    // ping-pong parity at its start is always 0.
    bool use_graph = false;
    {
        const int graph_env = opt.graph ? (opt.graph > 0 ? 1 : 0) : XINV_ENV_INT("XINV_GRAPH", -1);
        const double est_launch_us = (double)p.nbatch * (double)n * Kf /
                                     ((pl.path == XINV_PATH_FUSED) ? 2.0e5 : 4.0e4);
        // Replay pays on the colour path only (six or more tiny launches per sweep: 25.8 -> 22.2 us per sweep
        // at 151x251); for the fused kernels it gained nothing (round 1; C1: 2.5 ms replayed against 1.9 ms
        // per 500 sweeps with plain launches and the lagged norm, which excludes replay).  XINV_GRAPH=1 forces it.
        const bool want = graph_env >= 0 ? (graph_env != 0) : (est_launch_us < 12.0 && pl.path != XINV_PATH_FUSED);
        if (want && max_sweeps >= 2 * (int64_t)check_every * Kf) {
            check_every = (check_every + 1) & ~1;
            if (!ws->gstream) HIPCHK(hipStreamCreateWithFlags(&ws->gstream, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(ws->gstream, hipStreamCaptureModeRelaxed) == hipSuccess) {
                int r = XINV_OK;
                for (int i = 0; i < check_every && r == XINV_OK; i++) r = launch_one(ws->gstream, i & 1, Kf);
                const hipError_t ce = hipStreamEndCapture(ws->gstream, &g);
                if (r == XINV_OK && ce == hipSuccess && g &&
                    hipGraphInstantiate(&R.graph_exec, g, nullptr, nullptr, 0) == hipSuccess)
                    use_graph = true;
                if (g) (void)hipGraphDestroy(g);
            }
            (void)hipGetLastError();                       // a failed capture falls back to plain launches
        }
    }
    // (a solve of ONE launch -- the frames of apps.animate_iteration -- has nothing to overlap the reduction with: its own
    //  last workgroup reduces, one kernel launch less per frame)
    const bool lag = R.lag = lag_cand && !use_graph && !exp_noctl && max_sweeps > (int64_t)Kf;
    NormLagArgs lag_pending[XINV_MAX_LANES];               // per lane (one lane: [0])
    memset(lag_pending, 0, sizeof lag_pending);
    if (lag) {
        const size_t need = (size_t)((p.nbatch - 1) * p.sS + n) * sizeof(double);
        rc = ensure_dev(&ws->S3, &ws->S3_cap, need);
        if (rc) return rc;
        R.buf[2] = ws->S3; R.nbuf = 3;
    }
    // Masked-tile skipping: the skipped tiles' constant share of the norm, and their copy into every buffer S rotates
    // through (they are never written by the sweep launches).  Nobody needs either before the SECOND launch when the norm
    // is lagged -- launch 0 reads the caller's S and writes the active tiles of S2, its norm is evaluated in launch 1 --
    // so for one slice (one chain) the four small kernels (~40 us) run on a side stream beside launch 0.
    bool side_pending = false;
    struct SideGuard { Workspace *w; bool *pending; ~SideGuard() { if (*pending) (void)hipStreamSynchronize(w->s_side); } } side_guard{ws, &side_pending};
    if (pl.path == XINV_PATH_FUSED && pl.skip) {
        hipStream_t sk = st;
        if (lag && p.nbatch == 1) {
            if (!ws->s_side) {
                HIPCHK(hipStreamCreateWithFlags(&ws->s_side, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&ws->ev_side0, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ws->ev_side1, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(ws->ev_side0, st));    // (behind the planner's uploads and the workspace set-up)
            HIPCHK(hipStreamWaitEvent(ws->s_side, ws->ev_side0, 0));
            sk = ws->s_side;
        }
        // (one launch: every skipped tile's share of the norm, its copy into the other buffers, and -- by the block that
        //  arrives last -- the member's sum; k_skip_norm_tile / k_skip_norm_sum / k_copy_skipped until round 4)
        hipLaunchKernelGGL(k_skip_tiles, dim3((unsigned)pl.nskip, (unsigned)p.nbatch, 1), dim3(64), 0, sk,
                           pl.skipna, S2, lag ? ws->S3 : (double *)nullptr);
        HIPCHK(hipGetLastError());
        if (sk != st) { HIPCHK(hipEventRecord(ws->ev_side1, sk)); side_pending = true; }
    }
    auto side_join = [&]() -> int {                      // before the first reader: launch 1, or a chunk's closing reduction
        if (side_pending) { HIPCHK(hipStreamWaitEvent(st, ws->ev_side1, 0)); side_pending = false; }
        return XINV_OK;
    };
    // Lanes.  Every launch boundary synchronises the chip: the last round of workgroups drains, the reducers
    // wait for their last tile, and the next launch of the SAME members starts with every workgroup in the same phase.  The
    // members are independent, so the batch is cut into halves whose launches form independent chains -- the caller's
    // stream and the engine's own --; one chain's boundary is covered by the other's launch.  The control blocks are copied
    // for the host on a third stream behind both chains; everything is joined back into the caller's stream below.
    const int lanes_env = opt.lanes > 0 ? opt.lanes : XINV_ENV_INT("XINV_LANES", -1);
    int nlane = 1;
    if (!use_graph && !exp_noctl && pl.path == XINV_PATH_FUSED)
        nlane = (int)std::min<int64_t>(p.nbatch, lanes_env >= 0 ? std::max(1, std::min(lanes_env, XINV_MAX_LANES)) : lane_rule(p, est_pass_us));
    const bool two = nlane > 1;
    R.lanes = nlane;
    auto lane_first = [&](int l) { return p.nbatch * l / nlane; };   // members [lane_first(l), lane_first(l+1))
    // With several chains the host's polls of the control blocks are copies on a stream of their own, behind an event of
    // every chain.  (ONE chain keeps its polls on its own stream: with the copies on a second stream -- built in round 5 to
    // spare the ~10 us a copy holds the next launch back -- every sweep launch of the chain took 1 us longer, 35.0 ->
    // 36.0 us at 3600x1800, profiles/r05_solve_overhead.txt: a second active queue costs more than three polls.)
    const bool side_poll = two;
    struct LaneGuard {                                   // no return path leaves the side streams running
        Workspace *w; int n; bool poll;
        ~LaneGuard() { for (int l = 1; l < n; l++) (void)hipStreamSynchronize(w->s_lane[l]); if (poll) (void)hipStreamSynchronize(w->s_poll); }
    } lane_guard{ws, nlane, side_poll};
    if (side_poll) {
        if (!ws->s_poll) {
            for (int l = 1; l < XINV_MAX_LANES; l++) HIPCHK(hipStreamCreateWithFlags(&ws->s_lane[l], hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&ws->s_poll, hipStreamNonBlocking));
            for (int l = 0; l < XINV_MAX_LANES; l++)
                for (int q = 0; q < 2; q++) HIPCHK(hipEventCreateWithFlags(&ws->ev_lane[l][q], hipEventDisableTiming));
            HIPCHK(hipEventCreate(&ws->ev_s));
        }
    }
    if (two) {
        HIPCHK(hipEventRecord(ws->ev_s, st));            // fork: everything queued so far (workspace set-up) precedes every chain
        for (int l = 1; l < nlane; l++) HIPCHK(hipStreamWaitEvent(ws->s_lane[l], ws->ev_s, 0));
    }
    // Launch number i of the solve (fused path): k sweeps from buf[srcb[i]] into buf[dstb[i]]; srcb[0] = 0 (the caller's
    // S), srcb[i] = dstb[i-1].  Two buffers (no lagged norm): ping-pong.  Three (lagged norm): the decision about pass
    // i-1 arrives while pass i runs, so pass i must leave the source of pass i-1 intact (finalise() redoes a pass the
    // stop rule fired in from it): dstb[i] is the buffer that is neither srcb[i] nor srcb[i-1] -- a rotation.  Where
    // the rotation ends decides whether finalise() has to copy the result back into the caller's array (52 MB at
    // 3600x1800: ~30 us of a 4.3 ms solve).  Evaluating the pending pass BEFORE launch i (flush_lag: one small kernel)
    // lifts the constraint for that launch -- pass i is then a no-op for a member that stopped in pass i-1 -- and it
    // may write into srcb[i-1], which REVERSES the rotation: with nl launches to the sweep budget, forward for f and
    // backward for nl - f ends in buffer (2 f - nl) mod 3, so one reversal at f = nl - 1 (nl mod 3 == 2) or nl - 2
    // (nl mod 3 == 1) brings an un-converged solve home to buffer 0.  A solve that stops earlier copies, as before.
    const int64_t nl_budget = (max_sweeps + Kf - 1) / Kf;
    const int64_t flip_at = (!lag || nl_budget % 3 == 0) ? -1 : (nl_budget % 3 == 2 ? nl_budget - 1 : nl_budget - 2);
    int64_t wd_at = -1, wd_member = 0;
#if XINV_TEST_HOOKS
    // TEST-HOOKS BUILD ONLY (build/libxinv_hooks.so; the shipped library reads neither switch):
    // XINV_EXP_WATCHDOG="i[,m]" leaves member m (default 0), before launch i, in the state a reducer that timed out leaves
    // behind; XINV_HOOK_SKIP_PUBLISH="i,tile[,m]" makes that tile of launch i withhold its norm partial, so that the
    // reducer of launch i -- the launch's last workgroup, or with the lagged norm the extra workgroup of launch i+1 /
    // k_norm_reduce_lag -- REALLY runs into its (30 ms) watchdog while the later launches are queued behind it.
    if (const char *e = getenv("XINV_EXP_WATCHDOG")) {
        wd_at = atoll(e);
        if (const char *c = strchr(e, ',')) wd_member = atoll(c + 1);
        if (wd_member < 0 || wd_member >= p.nbatch) wd_at = -1;
    }
    struct HookGuard { ~HookGuard() { t_hook_record = nullptr; } } hook_guard;
    t_hook_record = nullptr;
    if (const char *e = getenv("XINV_HOOK_SKIP_PUBLISH")) {
        long long li = -1, tile = -1, mem = 0;
        if (sscanf(e, "%lld,%lld,%lld", &li, &tile, &mem) >= 2 && li >= 0 && tile >= 0 && mem >= 0 && mem < p.nbatch) {
            if (!ws->d_hook) HIPCHK(hipMalloc((void **)&ws->d_hook, 3 * sizeof(int)));
            const int rec[3] = {(int)tile, (int)(li + 1), (int)mem};      // (launch i publishes with tag i + 1)
            HIPCHK(hipMemcpyAsync(ws->d_hook, rec, sizeof rec, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            t_hook_record = ws->d_hook;
        }
    }
#endif
    std::function<int()> flush_lag;                      // (defined below; launch_idx flushes before a rotation reversal)
    auto launch_idx = [&](int64_t i, int k) -> int {
#if XINV_TEST_HOOKS
        if (i == wd_at) {                                // (on the stream of the member's lane: ordered before ITS launch i)
            int l = 0;
            while (l + 1 < nlane && lane_first(l + 1) <= wd_member) l++;
            hipLaunchKernelGGL(k_ctl_fake_timeout, dim3(1), dim3(1), 0, l ? ws->s_lane[l] : st, ws->ctl + wd_member);
        }
#endif
        const int sb = (i == 0) ? 0 : R.dstb[(size_t)i - 1];
        int db;
        if (R.nbuf == 2) db = sb ^ 1;
        else if (i == 0) db = 1;
        else if (i == flip_at) {                         // (the pending pass is evaluated first: its source is free)
            const int r = flush_lag(); if (r) return r;
            db = R.srcb[(size_t)i - 1];
        } else db = 3 - sb - R.srcb[(size_t)i - 1];
        R.srcb.push_back((signed char)sb); R.dstb.push_back((signed char)db);
        const double *src = buf[sb];
        double *dst = buf[db];
        if (i >= 1) { const int r = side_join(); if (r) return r; }
        if (exp_noctl == 2)                              // (timing experiment: publish only, nobody reduces)
            return launch_fused(p, pl, k, src, dst, ws, st, 0, p.nbatch, 1, 0, (unsigned)(i + 1), nullptr);
        if (!lag && !two) return launch_planned(p, pl, ws, st, k, src, dst, 0, p.nbatch, exp_noctl, exp_noctl);
        for (int l = 0; l < nlane; l++) {                // (one lane: the whole batch on the caller's stream)
            hipStream_t sl = l ? ws->s_lane[l] : st;
            const int64_t m0 = lane_first(l), nm = lane_first(l + 1) - m0;
            if (!lag) {
                const int r = launch_planned(p, pl, ws, sl, k, src, dst, m0, nm, 0, 0);
                if (r) return r;
                continue;
            }
            NormLagArgs la;
            const int r = launch_planned(p, pl, ws, sl, k, src, dst, m0, nm, 0, 0, (unsigned)(i + 1), &la, &lag_pending[l]);
            if (r) return r;
            lag_pending[l] = la;                         // evaluated by the lane's next launch, or by flush_lag()
        }
        return XINV_OK;
    };
    // the last launch of a chunk has no successor yet: its norm is evaluated by a one-workgroup kernel
    // before the control blocks are copied for the host
    flush_lag = [&]() -> int {
        if (!lag) return XINV_OK;
        { const int r = side_join(); if (r) return r; }
        for (int l = 0; l < nlane; l++) {
            if (!lag_pending[l].tag) continue;
            const int64_t mend = lane_first(l + 1);
            for (int64_t m0 = lane_first(l); m0 < mend; m0 += (int64_t)1 << 30) {
                lag_pending[l].member0 = m0;
                hipLaunchKernelGGL(k_norm_reduce_lag, dim3((unsigned)std::min<int64_t>((int64_t)1 << 30, mend - m0)),
                                   dim3(256), 0, l ? ws->s_lane[l] : st, lag_pending[l]);
            }
            lag_pending[l].tag = 0;
        }
        return XINV_OK;
    };
    int last_slot = 0;
    const bool per_launch_events = opt.timing == 2 && !two && !use_graph && pl.path == XINV_PATH_FUSED;
    // (k_ctl_mail, below: one chain on the caller's stream, the fused path, no timing events, a small batch)
    const bool mail_ok = !side_poll && !opt.timing && pl.path == XINV_PATH_FUSED && p.nbatch <= 64 &&
                         (int64_t)p.nbatch * n <= ((int64_t)1 << 21);
    unsigned mail_val[2] = {0u, 0u};
    if (mail_ok && !ws->hmail) {
        HIPCHK(hipHostMalloc((void **)&ws->hmail, 64, XINV_HOST_COHERENT));
        *ws->hmail = 0u;
    }
    auto issue_chunk = [&](int slot) -> int {
        if (opt.timing && !two) HIPCHK(hipEventRecord(ws->ev0[slot], st));
        if (use_graph && max_sweeps - launched >= (int64_t)check_every * Kf &&
            (pl.path != XINV_PATH_FUSED || (bound.size() & 1) == 0)) {
            HIPCHK(hipGraphLaunch(R.graph_exec, st));
            for (int i = 0; i < check_every; i++) {
                if (pl.path == XINV_PATH_FUSED) {       // (the captured chunk ping-pongs from buffer 0: launch_one)
                    bound.push_back(launched);
                    R.srcb.push_back((signed char)(i & 1)); R.dstb.push_back((signed char)((i & 1) ^ 1));
                }
                launched += Kf;
                nlaunch++;
            }
        } else
        for (int i = 0; i < check_every && launched < max_sweeps; i++) {
            int r;
            if (pl.path == XINV_PATH_FUSED) {
                const int k = (int)std::min<int64_t>(Kf, max_sweeps - launched);   // the tail: one shorter pass
                const bool tev = per_launch_events && R.lev.size() < XINV_MAX_LAUNCH_EVENTS;
                if (tev && R.lev.empty()) {
                    hipEvent_t e0; HIPCHK(hipEventCreate(&e0)); R.lev.push_back(e0);
                    HIPCHK(hipEventRecord(e0, st));
                }
                r = launch_idx((int64_t)bound.size(), k);
                if (r) return r;
                if (tev) {
                    hipEvent_t e1; HIPCHK(hipEventCreate(&e1)); R.lev.push_back(e1);
                    HIPCHK(hipEventRecord(e1, st));
                }
                bound.push_back(launched);
                launched += k;
            } else {
                r = launch_one(st, 0, 1);
                if (r) return r;
                launched += 1;
            }
            nlaunch++;
        }
        // (the last pass of a chunk has no successor yet to evaluate its norm: a one-workgroup kernel does -- at the end of
        //  the sweep budget only; the last pass of an earlier chunk is evaluated by the first launch of the next chunk like
        //  any other, and the host sees its decision one poll later)
        if (launched >= max_sweeps) { int r = flush_lag(); if (r) return r; }
        if (side_poll) {                                 // no chain waits for the copy (or for another chain): it has its own stream
            if (opt.timing && !two) HIPCHK(hipEventRecord(ws->ev1[slot], st));
            for (int l = 0; l < nlane; l++) {
                HIPCHK(hipEventRecord(ws->ev_lane[l][slot], l ? ws->s_lane[l] : st));
                HIPCHK(hipStreamWaitEvent(ws->s_poll, ws->ev_lane[l][slot], 0));
            }
            if (opt.timing && two) HIPCHK(hipEventRecord(ws->ev1[slot], ws->s_poll));
            HIPCHK(hipMemcpyAsync(ws->hctl + (size_t)slot * p.nbatch, ws->ctl, (size_t)p.nbatch * sizeof(XinvCtl),
                                  hipMemcpyDeviceToHost, ws->s_poll));
            HIPCHK(hipEventRecord(ws->evc[slot], ws->s_poll));
            last_slot = slot;
            return XINV_OK;
        }
        if (opt.timing) HIPCHK(hipEventRecord(ws->ev1[slot], st));
        // A short solve's last chunk (a launch or two on a small problem: the frames of apps.animate_iteration): the device
        // writes the control blocks into the pinned mirror itself and the host spins on a sequence word -- no copy engine,
        // no stream synchronisation (its wake-up was a quarter of such a solve).  Anything longer keeps the copy + event.
        if (mail_ok && launched >= max_sweeps && nlaunch <= 2) {
            mail_val[slot] = ++ws->mail_seq ? ws->mail_seq : ++ws->mail_seq;
            hipLaunchKernelGGL(k_ctl_mail, dim3(1), dim3(64), 0, st, ws->ctl, p.nbatch, ws->hctl + (size_t)slot * p.nbatch,
                               ws->hmail, mail_val[slot]);
            return XINV_OK;
        }
        HIPCHK(hipMemcpyAsync(ws->hctl + (size_t)slot * p.nbatch, ws->ctl, (size_t)p.nbatch * sizeof(XinvCtl),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(ws->evc[slot], st));
        return XINV_OK;
    };
    const XinvCtl *&hc = R.hc;
    hc = ws->hctl;
    bool more_at_break = false;
    rc = issue_chunk(0);
    if (rc) return rc;
    for (int c = 0;; c++) {
        const int slot = c & 1;
        const bool more = launched < max_sweeps;
        if (more) { rc = issue_chunk(slot ^ 1); if (rc) return rc; }
        if (mail_val[slot]) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0; __atomic_load_n(ws->hmail, __ATOMIC_ACQUIRE) != mail_val[slot]; spin++) {
                xinv_cpu_relax();
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                    HIPCHK(hipStreamSynchronize(st));    // (not short after all: wait the ordinary way; the mail has landed then)
                    break;
                }
            }
            mail_val[slot] = 0;
        } else
        HIPCHK(hipEventSynchronize(ws->evc[slot]));
        if (opt.timing) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, two ? ws->ev_s : ws->ev0[slot], ws->ev1[slot]));
            if (two) ms_total = ms; else ms_total += ms; // (two lanes: chunks overlap -- from the fork to the end of this chunk)
        }
        hc = ws->hctl + (size_t)slot * p.nbatch;
        all_done = true;
        for (int64_t m = 0; m < p.nbatch; m++) all_done = all_done && hc[m].done;
        if (all_done || !more) { more_at_break = more; break; }
    }
    if (side_poll && (two || more_at_break)) {
        // join: everything below runs on the caller's stream.  The copies above were taken while later launches ran (a
        // block caught in the middle of a reducer's update may be torn); a stopped member's block no longer changes, and
        // the final blocks are read again behind every chain.  (One chain that ran to its sweep budget: the last copy sits
        // behind the last launch and its closing reduction -- nothing to read again.)
        HIPCHK(hipStreamWaitEvent(st, ws->evc[last_slot], 0));   // (recorded behind every lane's last chunk)
        HIPCHK(hipMemcpyAsync(ws->hctl, ws->ctl, (size_t)p.nbatch * sizeof(XinvCtl), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        hc = ws->hctl;
        all_done = true;
        for (int64_t m = 0; m < p.nbatch; m++) all_done = all_done && hc[m].done;
    }
    if (pl.path != XINV_PATH_FUSED)                      // drain the queued no-op tail (the fused path syncs below)
        HIPCHK(hipStreamSynchronize(st));
    if (!all_done && exp_noctl) {                        // (experiment: no norm, no stop rule -- report the timing only)
        HIPCHK(hipStreamSynchronize(st));
        t_stats.sweep_launches = nlaunch; t_stats.sweep_ms = ms_total; t_stats.sweeps_per_launch = Kf;
        t_err = "XINV_EXP_NOCTL: timing experiment, no result";
        return XINV_ERR_ARG;
    }
    if (!all_done) { t_err = "internal: sweep budget exhausted before the stop rule fired"; return XINV_ERR_HIP; }
    // A member whose in-kernel norm reduction gave up waiting for a partial (watchdog, overflow == 2; never seen in
    // a run so far) is finished here instead of failing the call: the reducer stops the member BEFORE applying the
    // stop rule to any sweep of its launch, so the control block still describes the state at the start of that
    // launch and the launch's source buffer is intact (every later launch was a no-op for the member).  From there:
    // one sweep per launch without in-kernel norm, then the two separate norm kernels of the colour path
    // (k_norm_partial / k_norm_final: no waiting on other workgroups) -- the same sweeps and the same stop rule; the
    // partial sums are added in another order than the tiles' (flags[1] agrees to rounding).
    for (int64_t m = 0; m < p.nbatch; m++)
        if (hc[m].overflow == 2) {
            if (pl.path != XINV_PATH_FUSED) { t_err = "internal: watchdog flag outside the fused path"; return XINV_ERR_HIP; }
            HIPCHK(hipStreamSynchronize(st));            // (queued no-op launches)
            XinvCtl *hcm = const_cast<XinvCtl *>(hc) + m;
            const int64_t L = hcm->loop;
            const size_t i = std::lower_bound(bound.begin(), bound.end(), L) - bound.begin();
            if (i >= bound.size() || bound[i] != L) {
                t_err = "internal: norm partials of a sweep launch never arrived (watchdog) and the control block is not at a launch boundary";
                return XINV_ERR_HIP;
            }
            rc = ensure_dev(&ws->wd_part, &ws->wd_part_cap, (size_t)XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long)));
            if (rc) return rc;
            hipLaunchKernelGGL(k_ctl_resume, dim3(1), dim3(1), 0, st, ws->ctl + m);
            NormArgs na;
            memset(&na, 0, sizeof na);
            na.sS = p.sS; na.n = n; na.undef = p.sc_.undef;
            na.psum = (double *)ws->wd_part - m * XINV_NORM_BLOCKS;          // (the kernels index by member)
            na.pcnt = (long long *)((char *)ws->wd_part + XINV_NORM_BLOCKS * sizeof(double)) - m * XINV_NORM_BLOCKS;
            na.ctl = ws->ctl; na.stop = p.stop; na.force = 0; na.member0 = m;
            const int nblk = (int)std::min<int64_t>(XINV_NORM_BLOCKS, std::max<int64_t>(1, n / 2048));
            const int b0 = R.srcb[i], b1 = R.dstb[i];
            int a = b0, b = b1;
            int64_t s = L;
            bool fin = false;
            while (!fin && s < max_sweeps) {
                const int64_t burst = std::min<int64_t>(32, max_sweeps - s);
                for (int64_t q = 0; q < burst; q++, s++) {
                    // (biharmonic form, 'extend': the in-place pre-pass of the launch being redone has already run on
                    //  its source -- k_extend_bih precedes the sweep kernel whose reducer timed out -- and the periodic
                    //  one is not idempotent: the first recovery sweep skips it.  The test-hooks switch
                    //  XINV_EXP_WATCHDOG stops the member BEFORE that launch: no hooks case combines it with this form.)
                    const bool prepass = !(s == L && p.kind == KIND_BIH2D && p.BCy == XINV_BC_EXTEND);
                    rc = launch_planned(p, pl, ws, st, 1, buf[a], buf[b], m, 1, 0, 1, 0, nullptr, nullptr, prepass);
                    if (rc) return rc;
                    na.S = buf[b];
                    hipLaunchKernelGGL(k_norm_partial, dim3(nblk, 1, 1), dim3(256, 1, 1), 0, st, na);
                    hipLaunchKernelGGL(k_norm_final, dim3(1, 1, 1), dim3(64, 1, 1), 0, st, na, nblk);
                    std::swap(a, b);
                }
                HIPCHK(hipMemcpyAsync(hcm, ws->ctl + m, sizeof(XinvCtl), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                fin = hcm->done != 0;
            }
            if (!fin || hcm->overflow == 2) {
                t_err = "internal: norm partials of a sweep launch never arrived (watchdog) and the recovery did not finish";
                return XINV_ERR_HIP;
            }
            if (R.rec_where.empty()) R.rec_where.assign((size_t)p.nbatch, -1);
            // sweeps the recovery applied before the stop rule fired (launches after that were no-ops): parity = buffer
            R.rec_where[(size_t)m] = ((hcm->sweeps - L) & 1) ? b1 : b0;
            t_stats.recovered_members++;
        }

    return XINV_OK;
}

// fused path: put each member's final state into S (redo of a pass the stop rule fired inside); flags, stats
static int finalise(const Problem &p, const Plan &pl, Workspace *ws, hipStream_t st, double *flags, SweepRun &R,
                    bool stream_ordered = false)
{
    const int64_t n = p.zc * p.yc * p.xc;
    int rc = XINV_OK;
    (void)n; (void)rc;
    std::vector<int64_t> &bound = R.bound;
    double **buf = R.buf;
    const XinvCtl *hc = R.hc;
    int64_t sweeps_max = 0;
    if (pl.path == XINV_PATH_FUSED) {
        bound.push_back(R.launched);
        for (int64_t m = 0; m < p.nbatch; m++) {
            if (!R.rec_where.empty() && R.rec_where[(size_t)m] >= 0) {       // finished by the watchdog recovery
                const int where = R.rec_where[(size_t)m];
                if (where != 0)
                    HIPCHK(hipMemcpyAsync(p.S + m * p.sS, buf[where] + m * p.sS, (size_t)n * sizeof(double),
                                          hipMemcpyDeviceToDevice, st));
                continue;
            }
            const int64_t sw = hc[m].sweeps;
            // launch i covers sweeps (bound[i], bound[i+1]]; find the one holding sweep `sw`
            size_t i = std::upper_bound(bound.begin(), bound.end(), sw - 1) - bound.begin() - 1;
            const int nbuf = R.nbuf;
            if (i >= R.dstb.size()) { t_err = "internal: final sweep outside the launches issued"; return XINV_ERR_HIP; }
            int where;                                   // buffer index holding the final state
            // The biharmonic kernel's 'extend' pre-pass (k_extend_bih) works IN PLACE on the source buffer of its launch.
            // With the lagged norm the decision about pass i arrives while pass i+1 runs: that pass's pre-pass has then
            // already copied interior rows into the boundary rows of pass i's OUTPUT -- the final state -- which the
            // reference leaves as sweep i's own pre-pass made them (found by the extended fuzz at the end of round 4:
            // rows 0, 1, yc-2, yc-1 of a tolerance stop).  Pass i is redone from its source, which nothing has touched
            // but pass i's own pre-pass -- not applied again: the periodic one (r0 <- r1, then r1 <- r2) is not idempotent.
            const bool prepass_hit = R.lag && p.kind == KIND_BIH2D && p.BCy == XINV_BC_EXTEND && i + 2 < bound.size();
            if (bound[i + 1] == sw && !prepass_hit) {
                where = R.dstb[i];
            } else {                                     // stopped inside a K-sweep launch: redo from its source
                const int src0 = R.srcb[i];
                int cur = src0;                          // (intact: with the lagged norm the passes after i+1 did nothing)
                int nxt = R.dstb[i];                     // the pass's own output: free to overwrite
                const int spare = (nbuf == 3) ? 3 - cur - nxt : cur;
                for (int64_t s = bound[i]; s < sw; s++) {
                    rc = launch_planned(p, pl, ws, st, 1, buf[cur], buf[nxt], m, 1, 1, 1, 0, nullptr, nullptr, !prepass_hit);
                    if (rc) return rc;
                    const int t = cur; cur = nxt; nxt = (nbuf == 3 && t == src0) ? spare : t;
                }
                where = cur;
            }
            if (where != 0)
                HIPCHK(hipMemcpyAsync(p.S + m * p.sS, buf[where] + m * p.sS, (size_t)n * sizeof(double),
                                      hipMemcpyDeviceToDevice, st));
        }
        // (run_sweeps has synchronised behind the last launch and its control blocks; what may be queued behind that is
        //  the copy of the final state into S -- and a redone pass.  A plan solve leaves them in flight: S completes in
        //  stream order, 15-25 us of host wake-up less per solve; the workspace remembers the stream)
        if (stream_ordered && R.lev.size() <= 1) {
            if (!ws->ev_tail) HIPCHK(hipEventCreateWithFlags(&ws->ev_tail, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ws->ev_tail, st));
            ws->tail_pending = true;
        } else HIPCHK(hipStreamSynchronize(st));
        if (R.lev.size() > 1) {                          // timing == 2: the launches that did work (not the no-op tail)
            double mn = 1e300, mx = 0.0, sum = 0.0; int cnt = 0;
            for (size_t i = 0; i + 1 < R.lev.size() && i + 1 < bound.size(); i++) {
                bool live = false;                       // (some member still sweeping when launch i started)
                for (int64_t m = 0; m < p.nbatch && !live; m++) live = hc[m].sweeps > bound[i];
                if (!live) break;
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, R.lev[i], R.lev[i + 1]));
                mn = std::min(mn, (double)ms); mx = std::max(mx, (double)ms); sum += ms; cnt++;
            }
            if (cnt) { t_stats.launch_us_min = mn * 1e3; t_stats.launch_us_max = mx * 1e3; t_stats.launch_us_avg = sum * 1e3 / cnt; }
        }
    }
    for (int64_t m = 0; m < p.nbatch; m++) {
        const XinvCtl &c = hc[m];
        if (c.overflow) flags[3 * m + 0] = 1.0;
        if (c.wrote) { flags[3 * m + 1] = c.flag1; flags[3 * m + 2] = c.flag2; }
        sweeps_max = std::max<int64_t>(sweeps_max, c.sweeps);
    }
    t_stats.path = pl.path;
    t_stats.colours = pl.ncol;
    t_stats.sweeps_per_launch = R.Kf;
    t_stats.rows_per_tile = pl.RY;
    t_stats.xuniform_mask = (pl.path == XINV_PATH_FUSED || p.kind == KIND_BIH2D) ? (int32_t)pl.um : 0;
    t_stats.masked_tile_pct = (pl.path == XINV_PATH_FUSED && pl.skip) ? pl.skip_pct : 0;
    t_stats.masked_tile_ppm = (pl.path == XINV_PATH_FUSED && pl.skip) ? pl.skip_ppm : 0;
    t_stats.pipelined = (pl.path == XINV_PATH_FUSED && pl.pipe) ? pl.npair : 0;
    t_stats.lanes = R.lanes;
    t_stats.point_factor = (pl.path == XINV_PATH_FUSED && pl.pq) ? (pl.alias_ac ? 2 : 1) : 0;
    if (pl.path == XINV_PATH_FUSED && p.kind == KIND_BIH2D) t_stats.point_factor = pl.bih_vm;
    if (pl.path == XINV_PATH_FUSED && p.kind == KIND_STD3D && pl.K2) {
        const int64_t nm = (p.nbatch * 1 / R.lanes) - (p.nbatch * 0 / R.lanes);      // (members of the first lane's launches)
        const int64_t tiles = (int64_t)pl.nsg2 * pl.nrb2 * nm;
        t_stats.k_chunks = std::max(1, pl.nkc2);
        t_stats.cut_tiles = (int32_t)(tiles - p3_whole_tiles(tiles, std::max(1, pl.nkc2), pl.KC2, p.zc, pl.cus));
    }
    t_stats.sweep_launches = R.nlaunch;
    t_stats.sweeps_max = sweeps_max;
    t_stats.sweep_ms = R.ms_total;
    return XINV_OK;
}

// ------------------------------------------------------------------ the solve (device ptrs)
static int ws_ready(Workspace *ws)
{
    if (!ws->ev0[0])
        for (int q = 0; q < 2; q++) {
            HIPCHK(hipEventCreate(&ws->ev0[q])); HIPCHK(hipEventCreate(&ws->ev1[q]));
            HIPCHK(hipEventCreateWithFlags(&ws->evc[q], hipEventDisableTiming));
        }
    if (!ws->dflag) {
        HIPCHK(hipMalloc((void **)&ws->dflag, sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&ws->hflag, sizeof(int), hipHostMallocDefault));
    }
    return XINV_OK;
}

// colouring -> path -> tiling, per-row records, tile lists: everything a solve derives from the coefficient stack and the
// forcing's mask (nothing from S).  Detection passes run on `st` and are synchronous.
static int make_plan(const Problem &p, const xinv_options &opt, Workspace *ws, hipStream_t st, Plan &pl)
{
    memset(&pl, 0, sizeof pl);
    t_detected_um = 0;
    if (ws->cus <= 0) {
        int dev = 0, cus = 0;
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        ws->cus = cus > 0 ? cus : 256;
    }
    pl.cus = opt.cu_count > 0 ? opt.cu_count : (opt.cu_count < 0 ? (opt.cu_count == -1 ? -ws->cus : opt.cu_count) : ws->cus);
    int rc = plan_colouring(p, ws, st, pl);
    if (rc) return rc;
    if (opt.path == 3)                                  // (the value of round 2-3's XINV_PATH_SMALL)
        return fail_arg("path 3 (the register-resident small-slice solver) was removed in version 400: it never beat the "
                        "streaming kernels; use XINV_PATH_AUTO");
    pl.fma = (opt.flags & XINV_FLAG_FMA) != 0;
    rc = plan_path(p, opt, ws, st, pl);
    if (rc) return rc;
    if (pl.fma) {
        // contracted arithmetic exists for the per-row-coefficient variants of the standard 2-D, general 2-D and
        // standard 3-D streaming kernels (every lat-lon Poisson / Gill-Matsuno / omega problem): say so instead of
        // silently running the plain arithmetic
        const bool ok = pl.path == XINV_PATH_FUSED && !pl.nine && !pl.seam &&
                        ((p.kind == KIND_STD2D && pl.um == 3u) || (p.kind == KIND_GEN2D && pl.um == 0x1fu) ||
                         (p.kind == KIND_STD3D && pl.um == 7u));
        if (!ok)
            return fail_arg("XINV_FLAG_FMA: contracted arithmetic is available for the streaming kernels' per-row-coefficient "
                            "variants only (standard 2-D with A, C constant along x; general 2-D with A, C, D, E, F constant "
                            "along x; standard 3-D with A, B, C constant along x; B == 0; no odd-xc periodic seam)");
    }
    return XINV_OK;
}

static int solve_dev(Problem &p, double *flags, const xinv_options *opt_in, hipStream_t st, int slot = 0)
{
    int rc = validate(p, flags);
    if (rc) return rc;
    xinv_options opt;
    fill_options(opt, opt_in);

    DeviceGuard dg;
    HIPCHK(dg.select(opt.device));
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    Workspace *ws = get_ws(device, slot);
    std::lock_guard<std::recursive_mutex> solve_lock(ws->busy);
    rc = ws_ready(ws);
    if (rc) return rc;

    memset(&t_stats, 0, sizeof t_stats);
    const auto t_plan0 = std::chrono::steady_clock::now();
    Plan pl;
    rc = make_plan(p, opt, ws, st, pl);
    if (rc) return rc;
    const double plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
    SweepRun R;
    R.stream = st;
    rc = run_sweeps(p, pl, opt, ws, st, R);
    if (rc) return rc;
    rc = finalise(p, pl, ws, st, flags, R);
    t_stats.plan_ms = plan_ms;
    return rc;
}

// ------------------------------------------------------------------ resident plans (xinv_plan_*)
// The reference calls its kernel again and again on one coefficient stack: apps.animate_iteration (apps.py:1031-1044,
// one `invt_func(*coeffs, maskF, initS, dims, iParams)` per frame), a restart of an un-converged solve, a new first guess
// -- and every call of the *_dev entries re-derives what only depends on that stack: which arrays are constant along x
// (a pass over each), the per-row records, the forcing's activity map (a pass + a host round trip), the row split and
// the tile lists (host time), ~0.26 ms of a 4.8 ms headline solve and ALL of a two-sweep frame.  A plan holds them:
// built once by xinv_plan_create_*, used by every xinv_plan_solve_f64_dev; its device buffers (per-row records, tile
// lists, the skipped tiles' norm slots, expanded row-constant coefficients) are its own, swapped into the per-device
// workspace for the duration of a solve (under the workspace lock).
struct PlanBufs {
    void *d_rowf = nullptr; size_t d_rowf_cap = 0;
    int *d_list = nullptr; size_t d_list_cap = 0;
    double *d_tsum = nullptr; size_t d_tsum_cap = 0;
    double *d_pfac = nullptr; size_t d_pfac_cap = 0;
};
struct BufSwap {                                         // the plan's buffers sit in the workspace while this lives
    Workspace *ws; PlanBufs *b;
    static void sw(Workspace *w, PlanBufs *q)
    {
        std::swap(w->d_rowf, q->d_rowf); std::swap(w->d_rowf_cap, q->d_rowf_cap);
        std::swap(w->d_list, q->d_list); std::swap(w->d_list_cap, q->d_list_cap);
        std::swap(w->d_tsum, q->d_tsum); std::swap(w->d_tsum_cap, q->d_tsum_cap);
        std::swap(w->d_pfac, q->d_pfac); std::swap(w->d_pfac_cap, q->d_pfac_cap);
    }
    BufSwap(Workspace *w, PlanBufs *q) : ws(w), b(q) { sw(ws, b); }
    ~BufSwap() { sw(ws, b); }
};

#define XINV_PLAN_MAGIC 0x58504c4eu
struct xinv_plan {
    unsigned magic = XINV_PLAN_MAGIC;
    int device = 0;
    Problem p;                                           // S = a placeholder; stop = the kind's stop_on_zero_norm only
    xinv_options opt;
    Plan pl;
    PlanBufs bufs;
    std::vector<void *> owned;                           // row-constant coefficients expanded into HBM copies of the plan
    int64_t solves = 0;
};

static double *const kPlanS = (double *)(uintptr_t)4096;  // (never dereferenced: planning reads no S)

static int plan_build(xinv_plan *h, hipStream_t st)
{
    DeviceGuard dg;
    HIPCHK(dg.select(h->device));
    Workspace *ws = get_ws(h->device);
    std::lock_guard<std::recursive_mutex> lock(ws->busy);
    int rc = ws_ready(ws);
    if (rc) return rc;
    Problem p = h->p;
    p.S = kPlanS;
    p.stop.mxLoop = (long long)1 << 40; p.stop.tolerance = 0.0;
    rc = tail_wait(ws, st);                              // (a queued redo pass may still read this plan's records and lists)
    if (rc) return rc;
    BufSwap sw(ws, &h->bufs);
    rc = make_plan(p, h->opt, ws, st, h->pl);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));                    // records and lists are complete: any stream may solve on them
    return XINV_OK;
}

static void plan_free(xinv_plan *h)
{
    if (!h) return;
    DeviceGuard dg;
    (void)dg.select(h->device);
    {                                                    // (a stream-ordered solve's redo pass may still read the buffers)
        Workspace *ws = get_ws(h->device);
        std::lock_guard<std::recursive_mutex> lock(ws->busy);
        (void)tail_wait(ws, nullptr, true);
    }
    if (h->bufs.d_rowf) (void)hipFree(h->bufs.d_rowf);
    if (h->bufs.d_list) (void)hipFree(h->bufs.d_list);
    if (h->bufs.d_tsum) (void)hipFree(h->bufs.d_tsum);
    if (h->bufs.d_pfac) (void)hipFree(h->bufs.d_pfac);
    for (void *q : h->owned) (void)hipFree(q);
    h->magic = 0;
    delete h;
}

static int plan_create(xinv_plan **out, Problem &p, const xinv_options *opt_in, hipStream_t st)
{
    if (!out) return fail_arg("null plan pointer");
    *out = nullptr;
    xinv_options opt;
    fill_options(opt, opt_in);
    p.rowconst = (unsigned)opt.rowconst_mask & ((1u << p.ncoef) - 1u);
    if ((p.rowconst >> (p.ncoef - 1)) & 1u) return fail_arg("xinv_plan_create: the forcing cannot be row-constant");
    p.S = kPlanS;
    p.stop.mxLoop = 0; p.stop.tolerance = 0.0;
    double dummy_flags[3];
    int rc = validate(p, dummy_flags);
    if (rc) return rc;
    DeviceGuard dg;
    HIPCHK(dg.select(opt.device));
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    xinv_plan *h = new xinv_plan();
    h->device = device;
    h->opt = opt;
    h->opt.device = device;
    struct Undo { xinv_plan *h; ~Undo() { if (h) plan_free(h); } } undo{h};
    // coefficients handed over as one value per row (lat-lon grids: functions of latitude, apps.py:1406-1408,
    // 1630-1635): the plan expands them into its own HBM copies -- the caller's row vectors are not referenced after
    // this call -- and knows without a detection pass that they are constant along x
    const int64_t n = p.zc * p.yc * p.xc, rows = p.zc * p.yc;
    for (int q = 0; q < p.ncoef; q++) {
        if (!((p.rowconst >> q) & 1u) || !p.c[q]) continue;
        if (p.sc[q] != 0 && p.sc[q] != rows)
            return fail_arg("xinv_plan_create: a row-constant coefficient has batch stride 0 or exactly rows");
        const int64_t members = (p.sc[q] == 0) ? 1 : p.nbatch;
        void *full = nullptr;
        HIPCHK(hipMalloc(&full, (size_t)members * n * sizeof(double)));
        h->owned.push_back(full);
        hipLaunchKernelGGL(k_expand_rows, dim3(cdiv(rows * members, 4)), dim3(256), 0, st, p.c[q], (double *)full, rows,
                           p.xc, members);
        p.c[q] = (const double *)full;
        p.sc[q] = (members == 1) ? 0 : n;
        p.known_um |= 1u << q;
    }
    HIPCHK(hipGetLastError());
    p.rowconst = 0;
    h->p = p;
    rc = plan_build(h, st);
    if (rc) return rc;
    undo.h = nullptr;
    *out = h;
    return XINV_OK;
}

static int plan_solve(xinv_plan *h, double *S, double *flags, int64_t mxLoop, double tolerance, hipStream_t st)
{
    if (!h || h->magic != XINV_PLAN_MAGIC) return fail_arg("xinv_plan_solve: not a live plan");
    Problem p = h->p;
    p.S = S;
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tolerance;
    int rc = validate(p, flags);
    if (rc) return rc;
    if (h->pl.aligned && !ptr_al16(S))
        return fail_arg("xinv_plan_solve: this plan's kernels use 16-byte accesses: S must be 16-byte aligned");
    DeviceGuard dg;
    HIPCHK(dg.select(h->device));
    Workspace *ws = get_ws(h->device);
    std::lock_guard<std::recursive_mutex> lock(ws->busy);
    rc = ws_ready(ws);
    if (rc) return rc;
    BufSwap sw(ws, &h->bufs);
    memset(&t_stats, 0, sizeof t_stats);
    Plan pl = h->pl;
    pl.skipna.S = S;                                     // (the skipped tiles' norm share and copies read THIS solve's S)
    SweepRun R;
    R.stream = st;
    rc = run_sweeps(p, pl, h->opt, ws, st, R);
    if (rc) return rc;
    rc = finalise(p, pl, ws, st, flags, R, true);
    t_stats.planned = 1;
    h->solves++;
    return rc;
}

// ------------------------------------------------------------------ the solve (host ptrs)
// One device: upload -> solve -> download, pipelined over chunks of members on three streams.
// Every upload is queued at once on the `up` stream (shared coefficient arrays first, then S and
// the per-member arrays chunk by chunk, an event after each chunk); the solve of chunk c waits only
// for ITS event, so chunk c+1 travels while chunk c sweeps, and the download of chunk c (queued on
// the `down` stream when its solve returns) overlaps the sweeps of chunk c+1.  Both DMA directions
// and the CUs are busy at once; what stays exposed is the first chunk's upload and the last
// chunk's download.  Members are independent (reference core.py:129: no cross-slice state), so the
// chunking cannot change any result.
struct HostEvents {                                   // destroyed on every return path
    std::vector<hipEvent_t> e;
    ~HostEvents() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    int make(hipEvent_t *out, bool timing)
    {
        hipEvent_t x = nullptr;
        HIPCHK(timing ? hipEventCreate(&x) : hipEventCreateWithFlags(&x, hipEventDisableTiming));
        e.push_back(x);
        *out = x;
        return XINV_OK;
    }
};

// Member chunks of the upload / solve / download pipeline: sizes in members, in batch order.
// Chunking hides PCIe time behind sweeps (chunk c+1 travels and chunk c-1 returns while chunk c sweeps) but
// costs twice: every chunk repeats the once-per-solve detection / planning passes (~0.5 ms), and a chunk fills
// the 256 CUs less evenly than the whole batch (the 3-D kernels run ceil(workgroups / 256) rounds of one
// workgroup per CU).
static std::vector<int64_t> host_chunks(const Problem &p, const xinv_options &opt)
{
    const int64_t nb = p.nbatch;
    std::vector<int64_t> out;
    if (nb <= 1) { out.push_back(nb); return out; }
    if (opt.host_chunk > 0) {
        const int64_t mc = std::min<int64_t>(opt.host_chunk, nb);
        for (int64_t m0 = 0; m0 < nb; m0 += mc) out.push_back(std::min(mc, nb - m0));
        return out;
    }
    const int64_t n = p.zc * p.yc * p.xc;
    int per_member = 1;                                   // S
    for (int q = 0; q < p.ncoef; q++) per_member += (p.c[q] && p.sc[q] != 0 && !((p.rowconst >> q) & 1u)) ? 1 : 0;
    const double member_bytes = (double)n * 8.0 * per_member;
    const double total = member_bytes * (double)nb;
    // Round 5: TWO chunk solves are in flight on the device at a time (solve_host_one), so the holes a small chunk leaves
    // on the 256 CUs are filled by its neighbour's launches, and what remains to be minimised is the exposed first
    // upload / last download against the fixed cost of a chunk (~0.5 ms of planning, launches of few workgroups).
    // Measured (profiles/r05_host_pipeline.txt): C5, 15 volumes -- 1 chunk 212 ms, [4, 7, 4] (round 4's split) 172,
    // chunks of 2 volumes 161, of 1 volume 200; C4, 8 members -- 1 chunk 15.4 ms, chunks of 2 members 14.1, of 1: 17.2.
    if (total < 100663296.0 || nb < 4) { out.push_back(nb); return out; }
    if (is3d(p.kind)) {                                   // up to eight chunks of at least two volumes, the remainder LAST
        const int64_t nch = std::min<int64_t>(8, (nb + 1) / 2), per = (nb + nch - 1) / nch;
        for (int64_t m0 = 0; m0 < nb; m0 += per) out.push_back(std::min(per, nb - m0));
        return out;
    }
    const int64_t nch = std::min<int64_t>(std::min<int64_t>(4, nb / 2), std::max<int64_t>(2, (int64_t)((total + 33554431.0) / 33554432.0)));
    for (int64_t c = 0; c < nch; c++) out.push_back(nb / nch + (c < nb % nch ? 1 : 0));
    return out;
}

// One device: upload -> solve -> download, pipelined over member chunks by three actors:
//   the UPLOADER thread stages every upload through the library's pinned ring (xinv_host.h) in batch order --
//     shared coefficient arrays first, then S and the per-member arrays chunk by chunk, an event after each chunk;
//   the CALLING thread solves chunk c as soon as its event is recorded (the compute stream waits for it);
//   the DOWNLOADER thread brings each solved chunk's S back through its own ring.
// Both DMA directions and the CUs are busy at once; what stays exposed is the first chunk's upload and the
// last chunk's download.  Members are independent (reference core.py:129: no cross-slice state), so the
// chunking cannot change any result.
struct HostActors {                                   // joins the helper threads and drains the streams on EVERY return path
    std::thread up, down, solver2;                    // (solver2: the odd chunks' solves, beside the calling thread's)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> chunk_ready;                    // set by the uploader once chunk c's event is recorded
    std::deque<std::function<int()>> dq;              // download jobs
    bool d_closed = false, abort = false;
    int u_rc = 0, d_rc = 0, s2_rc = 0;                // (s2: the helper solver thread's verdict -- kept here: this object
    std::string u_err, d_err, s2_err;                 //  outlives the thread on every return path)
    std::vector<hipStream_t> streams;
    void close_downloads() { { std::lock_guard<std::mutex> lk(mu); d_closed = true; } cv.notify_all(); }
    ~HostActors()
    {
        { std::lock_guard<std::mutex> lk(mu); abort = true; d_closed = true; }
        cv.notify_all();
        if (solver2.joinable()) solver2.join();       // (before the downloader: it still queues download jobs)
        if (up.joinable()) up.join();
        if (down.joinable()) down.join();
        for (hipStream_t s : streams) (void)hipStreamSynchronize(s);      // nothing of this call stays in flight
    }
};

static int solve_host_one(Problem &p, double *flags, const xinv_options &opt, const Pinned *outer)
{
    const bool may_register = (outer == nullptr);     // a per-device call of a multi-device solve uses the parent's registrations
    const auto wall0 = std::chrono::steady_clock::now();
    DeviceGuard dg;
    HIPCHK(dg.select(opt.device));
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    const int64_t n = p.zc * p.yc * p.xc;
    // the staging rings, the device pool and the solver workspace are per device: hold the device for the whole
    // upload -> solve -> download sequence
    Workspace *ws = get_ws(device);
    std::lock_guard<std::recursive_mutex> host_lock(ws->busy);
    for (hipStream_t *sp : { &ws->s_up, &ws->s_down, &ws->s_compute })
        if (!*sp) HIPCHK(hipStreamCreateWithFlags(sp, hipStreamNonBlocking));
    hipStream_t sup = ws->s_up, sdn = ws->s_down, scp = ws->s_compute;
    DevPool *pool = get_pool(device);
    pool->reset();
    g_copy_pool.start();
    Pinned pin;                                       // opt-in registration of the caller's arrays (off by default)
    pin.enabled = may_register && (Pinned::env_allowed() || (opt.flags & XINV_FLAG_PIN_HOST));
    pin.outer = outer;
    pin.streams = { sup, sdn, scp };
    // a previous call that returned on an error may have left slots of the staging rings marked in flight, with
    // `dst` pointing into ITS host array: drain the (normally idle) copy streams and forget them
    HIPCHK(hipStreamSynchronize(sup));
    HIPCHK(hipStreamSynchronize(sdn));
    ws->ring_up.reset();
    ws->ring_down.reset();
    HostEvents ev;
    hipEvent_t e_up0, e_up1, e_dn0, e_dn1;
    int rc;
    if ((rc = ev.make(&e_up0, true)) || (rc = ev.make(&e_up1, true)) || (rc = ev.make(&e_dn0, true)) ||
        (rc = ev.make(&e_dn1, true))) return rc;

    const int64_t hsS = p.nbatch > 1 ? p.sS : n;
    const std::vector<int64_t> chunks = host_chunks(p, opt);
    const int64_t nchunk = (int64_t)chunks.size();
    std::vector<int64_t> first((size_t)nchunk + 1, 0);
    for (int64_t c = 0; c < nchunk; c++) first[(size_t)c + 1] = first[(size_t)c] + chunks[(size_t)c];

    // ---- device buffers now; what travels is queued for the uploader ----------------------------
    std::vector<std::function<int()>> shared_ops;     // before the first chunk
    std::vector<std::vector<std::function<int()>>> chunk_ops((size_t)nchunk);
    // host range -> device, `members` pieces of `len` elements of `esz` bytes (host stride hstride, device stride len)
    auto h2d_raw = [&](void *dev, const void *host, int64_t members, int64_t hstride, int64_t len, int esz) -> int {
        auto one = [&](char *d, const char *h, size_t bytes) -> int {
            if (pin.covers(h, bytes)) {                // registered in place: the DMA reads the caller's memory
                HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sup));
                return XINV_OK;
            }
            return stage_h2d(ws->ring_up, sup, (double *)d, (const double *)h, bytes);
        };
        char *dv = (char *)dev; const char *hs = (const char *)host;
        if (members == 1 || hstride == len) return one(dv, hs, (size_t)members * len * esz);
        for (int64_t m = 0; m < members; m++) {
            int r = one(dv + (size_t)m * len * esz, hs + (size_t)m * hstride * esz, (size_t)len * esz);
            if (r) return r;
        }
        return XINV_OK;
    };
    // float64 host array, or (tmp != nullptr) a float32 one: uploaded as it is -- half the bytes over PCIe -- into
    // `tmp` and promoted on the device (exact), in stream order
    auto h2d = [&](double *dev, const double *host, int64_t members, int64_t hstride, int64_t len, float *tmp = nullptr) -> int {
        if (!tmp) return h2d_raw(dev, host, members, hstride, len, 8);
        int r = h2d_raw(tmp, host, members, hstride, len, 4);
        if (r) return r;
        const int64_t cnt = members * len;
        hipLaunchKernelGGL(k_promote_f32, dim3((unsigned)std::min<int64_t>(4096, (cnt + 255) / 256)), dim3(256), 0, sup,
                           (const float *)tmp, dev, cnt);
        return XINV_OK;
    };
    auto is_f32 = [&](int arr) { return ((p.f32 >> arr) & 1u) != 0; };       // arr: 0 = S, q + 1 = coefficient q
    auto esz_of = [&](int arr) { return is_f32(arr) ? (size_t)4 : (size_t)8; };
    // (scratch for the float32 uploads: one buffer per array, as large as its largest piece; pieces follow each other
    //  in stream order on `sup`, so the buffer is free again when the next one lands)
    auto f32_tmp = [&](int arr, int64_t elems, float **out) -> int {
        *out = nullptr;
        if (!is_f32(arr)) return XINV_OK;
        double *t;
        int r = pool_alloc(pool, (size_t)elems * sizeof(float), &t);
        if (r) return r;
        *out = (float *)t;
        return XINV_OK;
    };
    Problem d = p;
    d.rowconst = 0;
    d.sS = n;
    rc = pool_alloc(pool, (size_t)p.nbatch * n * sizeof(double), &d.S);
    if (rc) return rc;
    pin.try_pin(p.S, (size_t)((p.nbatch - 1) * hsS + n) * esz_of(0));
    pin.note_pinned(p.S, (size_t)((p.nbatch - 1) * hsS + n) * esz_of(0));
    const int64_t mmax_chunk = *std::max_element(chunks.begin(), chunks.end());
    float *tmpS_up = nullptr, *tmpS_dn = nullptr;
    if (!(opt.prep_flags & XINV_PREP_S_ZERO)) { rc = f32_tmp(0, mmax_chunk * n, &tmpS_up); if (rc) return rc; }
    rc = f32_tmp(0, p.nbatch * n, &tmpS_dn);             // (downloads trail the solves: every chunk its own piece)
    if (rc) return rc;
    bool per_member[10];
    float *tmpC[10];
    for (int q = 0; q < p.ncoef; q++) {
        per_member[q] = false; tmpC[q] = nullptr;
        if (!p.c[q]) { d.c[q] = nullptr; d.sc[q] = 0; continue; }
        const int64_t hst = p.nbatch > 1 ? p.sc[q] : 0;
        const double *hq = p.c[q];
        double *dc;
        if ((p.rowconst >> q) & 1u) {                 // one value per row: upload rows, expand on the device
            const int64_t rows = p.zc * p.yc;
            const int64_t members = (hst == 0) ? 1 : p.nbatch;
            double *drow;
            rc = pool_alloc(pool, (size_t)members * rows * sizeof(double), &drow);
            if (rc) return rc;
            rc = pool_alloc(pool, (size_t)members * n * sizeof(double), &dc);
            if (rc) return rc;
            rc = f32_tmp(q + 1, members * rows, &tmpC[q]);
            if (rc) return rc;
            float *tq = tmpC[q];
            const int64_t xc = p.xc;
            shared_ops.push_back([=, &h2d]() -> int {
                int r = h2d(drow, hq, members, hst, rows, tq);
                if (r) return r;
                hipLaunchKernelGGL(k_expand_rows, dim3(cdiv(rows * members, 4)), dim3(256), 0, sup,
                                   (const double *)drow, dc, rows, xc, members);
                return XINV_OK;
            });
            d.sc[q] = (hst == 0) ? 0 : n;
            d.known_um |= 1u << q;                    // (expanded from one value per row: constant along x by construction)
        } else if (hst == 0) {
            rc = pool_alloc(pool, (size_t)n * sizeof(double), &dc);
            if (rc) return rc;
            pin.try_pin(hq, (size_t)n * esz_of(q + 1));
            pin.note_pinned(hq, (size_t)n * esz_of(q + 1));
            rc = f32_tmp(q + 1, n, &tmpC[q]);
            if (rc) return rc;
            float *tq = tmpC[q];
            shared_ops.push_back([=, &h2d]() -> int { return h2d(dc, hq, 1, 0, n, tq); });
            d.sc[q] = 0;
        } else {                                      // per member: travels with its chunk
            rc = pool_alloc(pool, (size_t)p.nbatch * n * sizeof(double), &dc);
            if (rc) return rc;
            pin.try_pin(hq, (size_t)((p.nbatch - 1) * hst + n) * esz_of(q + 1));
            pin.note_pinned(hq, (size_t)((p.nbatch - 1) * hst + n) * esz_of(q + 1));
            rc = f32_tmp(q + 1, mmax_chunk * n, &tmpC[q]);
            if (rc) return rc;
            d.sc[q] = n;
            per_member[q] = true;
        }
        d.c[q] = dc;
    }
    // front-end passes on the device (xinv_options.prep_flags): the forcing is the last array
    const int fq = p.ncoef - 1;
    const bool do_prep = (opt.prep_flags & (XINV_PREP_MASK_NAN | XINV_PREP_MASK_VALUE)) != 0;
    double *d_rowscale = nullptr;
    if (do_prep && (opt.prep_flags & XINV_PREP_ROWSCALE)) {
        if (!opt.prep_rowscale) return fail_arg("XINV_PREP_ROWSCALE without prep_rowscale");
        rc = pool_alloc(pool, (size_t)p.yc * sizeof(double), &d_rowscale);
        if (rc) return rc;
        const double *hrs = opt.prep_rowscale;
        const int64_t yc = p.yc;
        shared_ops.push_back([=, &h2d]() -> int { return h2d(d_rowscale, hrs, 1, 0, yc); });
    }
    const int prep_nan = (opt.prep_flags & XINV_PREP_MASK_NAN) ? 1 : 0;
    const double prep_undef = opt.prep_undef, undef_tmp = p.sc_.undef;
    const int64_t pyc = p.yc, pxc = p.xc;
    auto prep_forcing = [=](double *dF, int64_t nelem) {
        const unsigned nblk = (unsigned)std::min<int64_t>(4096, (nelem + 255) / 256);
        hipLaunchKernelGGL(k_prep_forcing, dim3(nblk), dim3(256), 0, sup, dF, nelem, pyc, pxc, (const double *)d_rowscale,
                           prep_nan, prep_undef, undef_tmp);
    };
    if (do_prep && !per_member[fq]) {
        double *dF = const_cast<double *>(d.c[fq]);
        shared_ops.push_back([=]() -> int { prep_forcing(dF, n); return XINV_OK; });      // one shared forcing
    }
    std::vector<hipEvent_t> e_chunk((size_t)nchunk);
    for (int64_t c = 0; c < nchunk; c++) {
        const int64_t m0 = first[(size_t)c], nm = chunks[(size_t)c];
        if ((rc = ev.make(&e_chunk[(size_t)c], false))) return rc;
        auto &ops = chunk_ops[(size_t)c];
        double *dS = d.S;
        const double *hS = p.S;
        if (opt.prep_flags & XINV_PREP_S_ZERO)
            ops.push_back([=]() -> int { HIPCHK(hipMemsetAsync(dS + m0 * n, 0, (size_t)nm * n * sizeof(double), sup)); return XINV_OK; });
        else
            ops.push_back([=, &h2d]() -> int {
                return h2d(dS + m0 * n, (const double *)((const char *)hS + (size_t)m0 * hsS * (tmpS_up ? 4 : 8)), nm, hsS, n, tmpS_up);
            });
        for (int q = 0; q < p.ncoef; q++)
            if (per_member[q]) {
                double *dq_ = const_cast<double *>(d.c[q]);
                const double *hq = p.c[q];
                const int64_t hst = p.sc[q];
                const bool prep_here = do_prep && q == fq;
                float *tq = tmpC[q];
                ops.push_back([=, &h2d]() -> int {
                    int r = h2d(dq_ + m0 * n, (const double *)((const char *)hq + (size_t)m0 * hst * (tq ? 4 : 8)), nm, hst, n, tq);
                    if (r) return r;
                    if (prep_here) prep_forcing(dq_ + m0 * n, nm * n);
                    return XINV_OK;
                });
            }
    }

    // ---- the actors -----------------------------------------------------------------------------
    HostActors act;
    act.streams = { sup, sdn, scp };
    act.chunk_ready.assign((size_t)nchunk, 0);
    act.up = std::thread([&]() {
        int r = (hipSetDevice(device) == hipSuccess) ? XINV_OK : XINV_ERR_HIP;
        auto run = [&](std::vector<std::function<int()>> &ops) {
            for (auto &f : ops) {
                { std::lock_guard<std::mutex> lk(act.mu); if (act.abort) r = r ? r : XINV_ERR_HIP; }
                if (r) return;
                try { r = f(); } catch (const std::exception &e) { t_err = e.what(); r = XINV_ERR_HIP; }
            }
        };
        if (!r && hipEventRecord(e_up0, sup) != hipSuccess) r = XINV_ERR_HIP;
        if (!r) run(shared_ops);
        for (int64_t c = 0; c < nchunk; c++) {
            if (!r) run(chunk_ops[(size_t)c]);
            if (!r && hipEventRecord(e_chunk[(size_t)c], sup) != hipSuccess) r = XINV_ERR_HIP;
            if (!r && c == nchunk - 1 && hipEventRecord(e_up1, sup) != hipSuccess) r = XINV_ERR_HIP;
            { std::lock_guard<std::mutex> lk(act.mu); act.chunk_ready[(size_t)c] = 1; if (r) { act.u_rc = r; act.u_err = t_err; } }
            act.cv.notify_all();
        }
    });
    act.down = std::thread([&]() {
        int r = (hipSetDevice(device) == hipSuccess) ? XINV_OK : XINV_ERR_HIP;
        bool first_job = true;
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(act.mu);
                act.cv.wait(lk, [&] { return act.d_closed || !act.dq.empty(); });
                if (act.dq.empty()) break;
                job = std::move(act.dq.front()); act.dq.pop_front();
                if (act.abort) continue;
            }
            if (r) continue;
            if (first_job) { if (hipEventRecord(e_dn0, sdn) != hipSuccess) r = XINV_ERR_HIP; first_job = false; }
            if (!r) { try { r = job(); } catch (const std::exception &e) { t_err = e.what(); r = XINV_ERR_HIP; } }
        }
        if (!r && first_job && hipEventRecord(e_dn0, sdn) != hipSuccess) r = XINV_ERR_HIP;
        if (!r && hipEventRecord(e_dn1, sdn) != hipSuccess) r = XINV_ERR_HIP;
        if (!r && hipStreamSynchronize(sdn) != hipSuccess) r = XINV_ERR_HIP;
        std::lock_guard<std::mutex> lk(act.mu);
        act.d_rc = r; if (r) act.d_err = t_err;
    });

    // ---- solve chunk by chunk; downloads trail on their own thread ------------------------------
    // Two chunk solves are in flight at a time (round 5): the even chunks on the calling thread (the device's workspace),
    // the odd ones on a helper thread with a workspace and a compute stream of its own.  A chunk fills the 256 CUs less
    // evenly than the whole batch -- the 3-D kernels run ceil(workgroups / 256) rounds, every 2-D launch ends with a
    // tail --; with the next chunk's launches already queued on the device those holes are filled, as the two launch
    // chains of a device-resident batch fill each other's (the lanes of run_sweeps).
    Workspace *ws1 = (nchunk > 1) ? get_ws(device, 1) : nullptr;
    hipStream_t scp1 = nullptr;
    if (ws1) {
        if (!ws1->s_compute) HIPCHK(hipStreamCreateWithFlags(&ws1->s_compute, hipStreamNonBlocking));
        scp1 = ws1->s_compute;
        act.streams.push_back(scp1);
    }
    // the workspaces grow on demand: size them for the LARGEST chunk now, so that a later, larger chunk does not pay a
    // free + malloc of the ping-pong buffer (or of the pinned control-block mirror) mid-pipeline
    {
        const int64_t mmax = *std::max_element(chunks.begin(), chunks.end());
        if (nchunk > 1 && p.kind != KIND_BIH2D)
            for (Workspace *w : { ws, ws1 }) {
                if ((rc = ensure_dev(&w->S2, &w->S2_cap, (size_t)mmax * n * sizeof(double)))) return rc;
                if ((rc = ensure_dev(&w->ctl, &w->ctl_cap, (size_t)mmax * sizeof(XinvCtl)))) return rc;
                if (w->hctl_cap < (size_t)mmax) {
                    if (w->hctl) HIPCHK(hipHostFree(w->hctl));
                    w->hctl = nullptr; w->hctl_cap = 0;
                    HIPCHK(hipHostMalloc((void **)&w->hctl, 2 * (size_t)mmax * sizeof(XinvCtl), XINV_HOST_COHERENT));
                    w->hctl_cap = (size_t)mmax;
                }
            }
    }
    xinv_stats acc;
    memset(&acc, 0, sizeof acc);
    bool acc_set = false;
    unsigned shared_um = 0;
    xinv_options o1 = opt;
    o1.device = device; o1.ndev = 0;
    // one chunk: wait for its upload, solve it on `cs` (workspace `slot`), run the output passes, hand it to the downloader
    auto do_chunk = [&](int64_t c, hipStream_t cs, int slot) -> int {
        const int64_t m0 = first[(size_t)c], nm = chunks[(size_t)c];
        {
            std::unique_lock<std::mutex> lk(act.mu);
            act.cv.wait(lk, [&] { return act.chunk_ready[(size_t)c] != 0 || act.abort; });
            if (act.u_rc) { t_err = act.u_err; return act.u_rc; }
            if (act.abort) { t_err = "host-pointer solve aborted"; return XINV_ERR_HIP; }
        }
        HIPCHK(hipStreamWaitEvent(cs, e_chunk[(size_t)c], 0));
        Problem dc = d;
        { std::lock_guard<std::mutex> lk(act.mu); dc.known_um |= shared_um; }     // (what an earlier chunk's plan found out)
        dc.nbatch = nm;
        dc.S = d.S + m0 * n;
        for (int q = 0; q < p.ncoef; q++)
            if (d.c[q] && d.sc[q] != 0) dc.c[q] = d.c[q] + m0 * d.sc[q];
        int r = solve_dev(dc, flags + 3 * m0, &o1, cs, slot);
        if (r) return r;
        {
            std::lock_guard<std::mutex> lk(act.mu);
            for (int q = 0; q < p.ncoef; q++)            // shared arrays found constant along x: the same for every chunk
                if (d.c[q] && d.sc[q] == 0 && ((t_detected_um >> q) & 1u)) shared_um |= 1u << q;
            if (!acc_set) { acc = t_stats; acc_set = true; }
            else {
                acc.sweep_launches += t_stats.sweep_launches;
                acc.sweeps_max = std::max(acc.sweeps_max, t_stats.sweeps_max);
                acc.sweep_ms += t_stats.sweep_ms;
                acc.recovered_members += t_stats.recovered_members;
            }
        }
        // solve_dev has returned: the chunk's S is final on the device
        if (opt.prep_flags & XINV_PREP_DEMASK) {
            for (int64_t m = 0; m < nm; m++) {
                const double *dF = d.c[fq] + (per_member[fq] ? (m0 + m) * n : 0);
                hipLaunchKernelGGL(k_demask, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, cs,
                                   d.S + (m0 + m) * n, dF, n, p.sc_.undef, opt.demask_value);
            }
            HIPCHK(hipStreamSynchronize(cs));
        }
        if (tmpS_dn) {                                   // float32 S: rounded on the device, half the bytes back
            hipLaunchKernelGGL(k_demote_f64, dim3((unsigned)std::min<int64_t>(4096, (nm * n + 255) / 256)), dim3(256), 0, cs,
                               (const double *)(d.S + m0 * n), tmpS_dn + m0 * n, nm * n);
            HIPCHK(hipStreamSynchronize(cs));
        }
        {
            char *hS = (char *)p.S;
            const char *dS = tmpS_dn ? (const char *)tmpS_dn : (const char *)d.S;
            const size_t es = tmpS_dn ? 4 : 8;
            const Pinned *pinp = &pin;
            std::lock_guard<std::mutex> lk(act.mu);
            act.dq.push_back([=]() -> int {
                auto one = [&](char *h, const char *dv, size_t bytes) -> int {
                    if (pinp->covers(h, bytes)) { HIPCHK(hipMemcpyAsync(h, dv, bytes, hipMemcpyDeviceToHost, sdn)); return XINV_OK; }
                    return stage_d2h(ws->ring_down, sdn, (double *)h, (const double *)dv, bytes);
                };
                if (hsS == n || nm == 1) return one(hS + (size_t)m0 * hsS * es, dS + (size_t)m0 * n * es, (size_t)nm * n * es);
                for (int64_t m = m0; m < m0 + nm; m++) {
                    int rr = one(hS + (size_t)m * hsS * es, dS + (size_t)m * n * es, (size_t)n * es);
                    if (rr) return rr;
                }
                return XINV_OK;
            });
        }
        act.cv.notify_all();
        return XINV_OK;
    };
    if (nchunk > 1)
        act.solver2 = std::thread([&]() {
            int r = (hipSetDevice(device) == hipSuccess) ? XINV_OK : XINV_ERR_HIP;
            for (int64_t c = 1; c < nchunk && !r; c += 2) {
                try { r = do_chunk(c, scp1, 1); } catch (const std::exception &e) { t_err = e.what(); r = XINV_ERR_HIP; }
            }
            act.s2_rc = r; if (r) act.s2_err = t_err;
        });
    for (int64_t c = 0; c < nchunk; c += 2) {
        rc = do_chunk(c, scp, 0);
        if (rc) return rc;                               // (HostActors' destructor stops and joins the helper)
    }
    if (act.solver2.joinable()) act.solver2.join();
    if (act.s2_rc) { t_err = act.s2_err; return act.s2_rc; }
    act.close_downloads();
    act.up.join();
    act.down.join();
    if (act.u_rc) { t_err = act.u_err; return act.u_rc; }
    if (act.d_rc) { t_err = act.d_err; return act.d_rc; }
    HIPCHK(hipStreamSynchronize(sup));
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventElapsedTime(&a, e_up0, e_up1));
    HIPCHK(hipEventElapsedTime(&b, e_dn0, e_dn1));
    t_stats = acc;
    t_stats.h2d_ms = a; t_stats.d2h_ms = b;
    t_stats.host_chunks = (int32_t)nchunk;
    t_stats.devices = 1;
    t_stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return XINV_OK;
}

// Host-pointer entry: one device, or the batch axis split in contiguous blocks over a device list
// (the reference loops slices in ONE process, core.py:129-139; so does this --
// one host thread per GPU, no collective, S and flags land in the caller's arrays).
static int solve_host(Problem &p, double *flags, const xinv_options *opt_in)
{
    xinv_options opt;
    fill_options(opt, opt_in);
    p.rowconst = (unsigned)opt.rowconst_mask & ((1u << p.ncoef) - 1u);
    p.f32 = (unsigned)opt.f32_mask & ((2u << p.ncoef) - 1u);
    int rc = validate(p, flags);
    if (rc) return rc;
    int nvis = 0;
    if (hipGetDeviceCount(&nvis) != hipSuccess || nvis < 1) {
        (void)hipGetLastError();
        t_err = "no HIP device available";
        return XINV_ERR_NODEV;
    }
    std::vector<int> devs;
    if (opt.ndev < 0) {                                // every visible GPU
        for (int i = 0; i < nvis; i++) devs.push_back(i);
    } else if (opt.ndev > 0) {
        if (opt.ndev > XINV_MAX_DEVICES) return fail_arg("ndev exceeds XINV_MAX_DEVICES");
        for (int i = 0; i < opt.ndev; i++) {
            if (opt.device_ids[i] < 0 || opt.device_ids[i] >= nvis) return fail_arg("device_ids: no such device");
            devs.push_back(opt.device_ids[i]);
        }
    }
    if ((int64_t)devs.size() > p.nbatch) devs.resize((size_t)p.nbatch);
    if (devs.size() <= 1) {
        if (devs.size() == 1) opt.device = devs[0];
        return solve_host_one(p, flags, opt, nullptr);
    }

    const auto wall0 = std::chrono::steady_clock::now();
    const int nd = (int)devs.size();
    const int64_t n = p.zc * p.yc * p.xc;
    // host ranges pinned ONCE for every device (portable registration); the per-device threads
    // then copy straight out of / into the caller's arrays
    Pinned pin;                                        // (opt-in: the per-device calls stage through their own rings otherwise)
    pin.enabled = Pinned::env_allowed() || (opt.flags & XINV_FLAG_PIN_HOST);
    pin.flags = hipHostRegisterPortable;
    auto esz = [&](int arr) { return ((p.f32 >> arr) & 1u) ? (size_t)4 : (size_t)8; };
    pin.try_pin(p.S, (size_t)((p.nbatch - 1) * p.sS + n) * esz(0));
    pin.note_pinned(p.S, (size_t)((p.nbatch - 1) * p.sS + n) * esz(0));
    for (int q = 0; q < p.ncoef; q++) {
        if (!p.c[q]) continue;
        const int64_t len = ((p.rowconst >> q) & 1u) ? p.zc * p.yc : n;
        pin.try_pin(p.c[q], (size_t)((p.sc[q] == 0 ? 0 : (p.nbatch - 1) * p.sc[q]) + len) * esz(q + 1));
        pin.note_pinned(p.c[q], (size_t)((p.sc[q] == 0 ? 0 : (p.nbatch - 1) * p.sc[q]) + len) * esz(q + 1));
    }
    struct Result { int rc = 0; std::string err; xinv_stats st; };
    std::vector<Result> res((size_t)nd);
    std::vector<std::thread> th;
    const int64_t q0 = p.nbatch / nd, r0 = p.nbatch % nd;
    for (int i = 0; i < nd; i++) {
        const int64_t lo = i * q0 + std::min<int64_t>(i, r0), hi = lo + q0 + (i < r0 ? 1 : 0);
        th.emplace_back([&, i, lo, hi]() {
            Problem sub = p;
            sub.nbatch = hi - lo;
            sub.S = (double *)((char *)p.S + (size_t)lo * p.sS * esz(0));          // (strides count elements of the array's type)
            for (int q = 0; q < p.ncoef; q++)
                if (p.c[q]) sub.c[q] = (const double *)((const char *)p.c[q] + (size_t)lo * p.sc[q] * esz(q + 1));
            xinv_options o1 = opt;
            o1.device = devs[(size_t)i]; o1.ndev = 0;
            int r;
            try { r = solve_host_one(sub, flags + 3 * lo, o1, &pin); }
            catch (const std::exception &e) { t_err = e.what(); r = XINV_ERR_HIP; }
            catch (...) { t_err = "unknown C++ exception"; r = XINV_ERR_HIP; }
            res[(size_t)i].rc = r; res[(size_t)i].err = t_err; res[(size_t)i].st = t_stats;
        });
    }
    for (auto &t : th) t.join();
    t_stats = res[0].st;
    for (int i = 0; i < nd; i++) {
        if (res[(size_t)i].rc) { t_err = res[(size_t)i].err; return res[(size_t)i].rc; }
        if (i == 0) continue;
        const xinv_stats &s = res[(size_t)i].st;
        t_stats.sweep_launches += s.sweep_launches;
        t_stats.sweeps_max = std::max(t_stats.sweeps_max, s.sweeps_max);
        t_stats.sweep_ms = std::max(t_stats.sweep_ms, s.sweep_ms);
        t_stats.h2d_ms = std::max(t_stats.h2d_ms, s.h2d_ms);
        t_stats.d2h_ms = std::max(t_stats.d2h_ms, s.d2h_ms);
        t_stats.host_chunks += s.host_chunks;
        t_stats.recovered_members += s.recovered_members;
    }
    t_stats.devices = nd;
    t_stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return XINV_OK;
}

// ------------------------------------------------------------------ problem builders
static void set_scal2d(Problem &p, double delx, double delxSqr, double ratio, double ratioQtr,
                       double ratioSqr, double optArg, double undef)
{
    memset(&p.sc_, 0, sizeof p.sc_);
    p.sc_.delx = delx; p.sc_.delxSqr = delxSqr; p.sc_.ratio = ratio;
    p.sc_.ratioQtr = ratioQtr; p.sc_.ratioSqr = ratioSqr; p.sc_.optArg = optArg;
    p.sc_.undef = undef;
}

static Problem mk_std2d(double *S, const double *A, const double *B, const double *C,
                        const double *F, int64_t nbatch, const int64_t *st, int64_t yc,
                        int64_t xc, double delx, int BCy, int BCx, double delxSqr,
                        double ratioQtr, double ratioSqr, double optArg, double undef,
                        int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_STD2D; p.nbatch = nbatch; p.zc = 1; p.yc = yc; p.xc = xc;
    p.S = S; p.c[0] = A; p.c[1] = B; p.c[2] = C; p.c[3] = F; p.ncoef = 4;
    const int64_t n = yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 4; q++) p.sc[q] = st ? st[1 + q] : n;
    p.BCz = 0; p.BCy = BCy; p.BCx = BCx;
    set_scal2d(p, delx, delxSqr, 0.0, ratioQtr, ratioSqr, optArg, undef);
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 1;
    return p;
}

static Problem mk_gen2d(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, const double *G,
                        int64_t nbatch, const int64_t *st, int64_t yc, int64_t xc, double delx,
                        int BCy, int BCx, double delxSqr, double ratio, double ratioQtr,
                        double ratioSqr, double optArg, double undef, int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_GEN2D; p.nbatch = nbatch; p.zc = 1; p.yc = yc; p.xc = xc;
    p.S = S;
    p.c[0] = A; p.c[1] = B; p.c[2] = C; p.c[3] = D; p.c[4] = E; p.c[5] = F; p.c[6] = G;
    p.ncoef = 7;
    const int64_t n = yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 7; q++) p.sc[q] = st ? st[1 + q] : n;
    p.BCz = 0; p.BCy = BCy; p.BCx = BCx;
    set_scal2d(p, delx, delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef);
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 0;
    return p;
}

static Problem mk_std2dt(double *S, const double *const *co, int64_t nbatch, const int64_t *st,
                         int64_t yc, int64_t xc, double delx, int BCy, int BCx, double delxSqr,
                         double ratioQtr, double ratioSqr, double optArg, double undef,
                         int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_STD2DT; p.nbatch = nbatch; p.zc = 1; p.yc = yc; p.xc = xc;
    p.S = S; p.ncoef = 6;
    const int64_t n = yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 6; q++) { p.c[q] = co[q]; p.sc[q] = st ? st[1 + q] : n; }
    p.BCz = 0; p.BCy = BCy; p.BCx = BCx;
    set_scal2d(p, delx, delxSqr, 0.0, ratioQtr, ratioSqr, optArg, undef);
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 1;
    return p;
}

static Problem mk_bih2d(double *S, const double *const *co, int64_t nbatch, const int64_t *st,
                        int64_t yc, int64_t xc, int BCy, int BCx, double delxSSr, double delxTr,
                        double delxSqr, double ratio, double ratioSSr, double ratioQtr,
                        double ratioSqr, double optArg, double undef, int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_BIH2D; p.nbatch = nbatch; p.zc = 1; p.yc = yc; p.xc = xc;
    p.S = S; p.ncoef = 10;
    const int64_t n = yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 10; q++) { p.c[q] = co[q]; p.sc[q] = st ? st[1 + q] : n; }
    p.BCz = 0; p.BCy = BCy; p.BCx = BCx;
    memset(&p.sc_, 0, sizeof p.sc_);
    p.sc_.delxSSr = delxSSr; p.sc_.delxTr = delxTr; p.sc_.delxSqr = delxSqr; p.sc_.ratio = ratio;
    p.sc_.ratioSSr = ratioSSr; p.sc_.ratioQtr = ratioQtr; p.sc_.ratioSqr = ratioSqr;
    p.sc_.optArg = optArg; p.sc_.undef = undef;
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 0;
    return p;
}

static Problem mk_std3d(double *S, const double *A, const double *B, const double *C,
                        const double *F, int64_t nbatch, const int64_t *st, int64_t zc,
                        int64_t yc, int64_t xc, int BCz, int BCy, int BCx, double delxSqr,
                        double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                        int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_STD3D; p.nbatch = nbatch; p.zc = zc; p.yc = yc; p.xc = xc;
    p.S = S; p.c[0] = A; p.c[1] = B; p.c[2] = C; p.c[3] = F; p.ncoef = 4;
    const int64_t n = zc * yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 4; q++) p.sc[q] = st ? st[1 + q] : n;
    p.BCz = BCz; p.BCy = BCy; p.BCx = BCx;
    memset(&p.sc_, 0, sizeof p.sc_);
    p.sc_.delxSqr = delxSqr; p.sc_.ratio2Sqr = ratio2Sqr; p.sc_.ratio1Sqr = ratio1Sqr;
    p.sc_.optArg = optArg; p.sc_.undef = undef;
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 0;
    return p;
}

static Problem mk_gen3d(double *S, const double *const *c, int64_t nbatch, const int64_t *st,
                        int64_t zc, int64_t yc, int64_t xc, double delx, int BCz, int BCy, int BCx,
                        double delxSqr, double ratio2, double ratio1, double ratio2Sqr,
                        double ratio1Sqr, double optArg, double undef, int64_t mxLoop, double tol)
{
    Problem p;
    memset(&p, 0, sizeof p);
    p.kind = KIND_GEN3D; p.nbatch = nbatch; p.zc = zc; p.yc = yc; p.xc = xc;
    p.S = S; p.ncoef = 8;
    const int64_t n = zc * yc * xc;
    p.sS = st ? st[0] : n;
    for (int q = 0; q < 8; q++) { p.c[q] = c[q]; p.sc[q] = st ? st[1 + q] : n; }
    p.BCz = BCz; p.BCy = BCy; p.BCx = BCx;
    memset(&p.sc_, 0, sizeof p.sc_);
    p.sc_.delx = delx; p.sc_.delxSqr = delxSqr; p.sc_.ratio2 = ratio2; p.sc_.ratio1 = ratio1;
    p.sc_.ratio2Sqr = ratio2Sqr; p.sc_.ratio1Sqr = ratio1Sqr;
    p.sc_.optArg = optArg; p.sc_.undef = undef;
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tol; p.stop.stop_on_zero_norm = 0;
    return p;
}

// ------------------------------------------------------------------ C-ABI
extern "C" {

void xinv_default_options(xinv_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->device = -1;
    o->path = XINV_PATH_AUTO;
}

int xinv_last_stats(xinv_stats *out)
{
    if (!out) return XINV_ERR_ARG;
    *out = t_stats;
    return XINV_OK;
}

const char *xinv_last_error(void) { return t_err.c_str(); }

int xinv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int xinv_version(void) { return XINV_VERSION; }
void xinv_abi_sizes(int32_t *options_bytes, int32_t *stats_bytes)
{
    if (options_bytes) *options_bytes = (int32_t)sizeof(xinv_options);
    if (stats_bytes) *stats_bytes = (int32_t)sizeof(xinv_stats);
}

#define GUARD(expr) try { return (expr); } catch (const std::exception &e) { t_err = e.what(); return XINV_ERR_HIP; } catch (...) { t_err = "unknown C++ exception"; return XINV_ERR_HIP; }

int xinv_standard_2d_f64(double *S, const double *A, const double *B, const double *C,
                         const double *F, int64_t yc, int64_t xc, double dely, double delx,
                         int BCy, int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                         double optArg, double undef, double *flags, int64_t mxLoop,
                         double tolerance)
{
    (void)dely;
    Problem p = mk_std2d(S, A, B, C, F, 1, nullptr, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr,
                         ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_general_2d_f64(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, const double *G,
                        int64_t yc, int64_t xc, double dely, double delx, int BCy, int BCx,
                        double delxSqr, double ratio, double ratioQtr, double ratioSqr,
                        double optArg, double undef, double *flags, int64_t mxLoop,
                        double tolerance)
{
    (void)dely;
    Problem p = mk_gen2d(S, A, B, C, D, E, F, G, 1, nullptr, yc, xc, delx, BCy, BCx, delxSqr,
                         ratio, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_standard_3d_f64(double *S, const double *A, const double *B, const double *C,
                         const double *F, int64_t zc, int64_t yc, int64_t xc, double delz,
                         double dely, double delx, int BCz, int BCy, int BCx, double delxSqr,
                         double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                         double *flags, int64_t mxLoop, double tolerance)
{
    (void)delz; (void)dely; (void)delx;
    Problem p = mk_std3d(S, A, B, C, F, 1, nullptr, zc, yc, xc, BCz, BCy, BCx, delxSqr,
                         ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_standard_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                 const double *F, int64_t nbatch, const int64_t *strides,
                                 int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                 int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                                 double optArg, double undef, double *flags, int64_t mxLoop,
                                 double tolerance, const xinv_options *opt)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std2d(S, A, B, C, F, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr,
                         ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_general_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, int64_t nbatch, const int64_t *strides,
                                int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                int BCx, double delxSqr, double ratio, double ratioQtr,
                                double ratioSqr, double optArg, double undef, double *flags,
                                int64_t mxLoop, double tolerance, const xinv_options *opt)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_gen2d(S, A, B, C, D, E, F, G, nbatch, strides, yc, xc, delx, BCy, BCx,
                         delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_standard_3d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                 const double *F, int64_t nbatch, const int64_t *strides,
                                 int64_t zc, int64_t yc, int64_t xc, double delz, double dely,
                                 double delx, int BCz, int BCy, int BCx, double delxSqr,
                                 double ratio2Sqr, double ratio1Sqr, double optArg,
                                 double undef, double *flags, int64_t mxLoop, double tolerance,
                                 const xinv_options *opt)
{
    (void)delz; (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std3d(S, A, B, C, F, nbatch, strides, zc, yc, xc, BCz, BCy, BCx, delxSqr,
                         ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_standard_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t nbatch, const int64_t *strides,
                             int64_t yc, int64_t xc, double dely, double delx, int BCy,
                             int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                             double optArg, double undef, double *flags, int64_t mxLoop,
                             double tolerance, const xinv_options *opt, void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std2d(S, A, B, C, F, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr,
                         ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

int xinv_general_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc,
                            double dely, double delx, int BCy, int BCx, double delxSqr,
                            double ratio, double ratioQtr, double ratioSqr, double optArg,
                            double undef, double *flags, int64_t mxLoop, double tolerance,
                            const xinv_options *opt, void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_gen2d(S, A, B, C, D, E, F, G, nbatch, strides, yc, xc, delx, BCy, BCx,
                         delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

int xinv_standard_3d_f64_dev(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t nbatch, const int64_t *strides,
                             int64_t zc, int64_t yc, int64_t xc, double delz, double dely,
                             double delx, int BCz, int BCy, int BCx, double delxSqr,
                             double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                             double *flags, int64_t mxLoop, double tolerance,
                             const xinv_options *opt, void *stream)
{
    (void)delz; (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std3d(S, A, B, C, F, nbatch, strides, zc, yc, xc, BCz, BCy, BCx, delxSqr,
                         ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

int xinv_general_3d_f64(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, const double *G,
                        const double *H, int64_t zc, int64_t yc, int64_t xc, double delz,
                        double dely, double delx, int BCz, int BCy, int BCx, double delxSqr,
                        double ratio2, double ratio1, double ratio2Sqr, double ratio1Sqr,
                        double optArg, double undef, double *flags, int64_t mxLoop,
                        double tolerance)
{
    (void)delz; (void)dely;
    const double *c[8] = { A, B, C, D, E, F, G, H };
    Problem p = mk_gen3d(S, c, 1, nullptr, zc, yc, xc, delx, BCz, BCy, BCx, delxSqr, ratio2, ratio1,
                         ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_general_3d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, const double *H, int64_t nbatch,
                                const int64_t *strides, int64_t zc, int64_t yc, int64_t xc,
                                double delz, double dely, double delx, int BCz, int BCy, int BCx,
                                double delxSqr, double ratio2, double ratio1, double ratio2Sqr,
                                double ratio1Sqr, double optArg, double undef, double *flags,
                                int64_t mxLoop, double tolerance, const xinv_options *opt)
{
    (void)delz; (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *c[8] = { A, B, C, D, E, F, G, H };
    Problem p = mk_gen3d(S, c, nbatch, strides, zc, yc, xc, delx, BCz, BCy, BCx, delxSqr, ratio2,
                         ratio1, ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_general_3d_f64_dev(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            const double *H, int64_t nbatch, const int64_t *strides, int64_t zc,
                            int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                            int BCy, int BCx, double delxSqr, double ratio2, double ratio1,
                            double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                            double *flags, int64_t mxLoop, double tolerance,
                            const xinv_options *opt, void *stream)
{
    (void)delz; (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *c[8] = { A, B, C, D, E, F, G, H };
    Problem p = mk_gen3d(S, c, nbatch, strides, zc, yc, xc, delx, BCz, BCy, BCx, delxSqr, ratio2,
                         ratio1, ratio2Sqr, ratio1Sqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

int xinv_general_bih_2d_f64(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            const double *H, const double *I, const double *J, int64_t yc,
                            int64_t xc, double dely, double delx, int BCy, int BCx,
                            double delxSSr, double delxTr, double delxSqr, double ratio,
                            double ratioSSr, double ratioQtr, double ratioSqr, double optArg,
                            double undef, double *flags, int64_t mxLoop, double tolerance)
{
    (void)dely; (void)delx;
    const double *co[10] = { A, B, C, D, E, F, G, H, I, J };
    Problem p = mk_bih2d(S, co, 1, nullptr, yc, xc, BCy, BCx, delxSSr, delxTr, delxSqr, ratio,
                         ratioSSr, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_general_bih_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                    const double *D, const double *E, const double *F,
                                    const double *G, const double *H, const double *I,
                                    const double *J, int64_t nbatch, const int64_t *strides,
                                    int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                    int BCx, double delxSSr, double delxTr, double delxSqr,
                                    double ratio, double ratioSSr, double ratioQtr,
                                    double ratioSqr, double optArg, double undef, double *flags,
                                    int64_t mxLoop, double tolerance, const xinv_options *opt)
{
    (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    const double *co[10] = { A, B, C, D, E, F, G, H, I, J };
    Problem p = mk_bih2d(S, co, nbatch, strides, yc, xc, BCy, BCx, delxSSr, delxTr, delxSqr,
                         ratio, ratioSSr, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_general_bih_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, const double *H, const double *I,
                                const double *J, int64_t nbatch, const int64_t *strides,
                                int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                int BCx, double delxSSr, double delxTr, double delxSqr,
                                double ratio, double ratioSSr, double ratioQtr, double ratioSqr,
                                double optArg, double undef, double *flags, int64_t mxLoop,
                                double tolerance, const xinv_options *opt, void *stream)
{
    (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    const double *co[10] = { A, B, C, D, E, F, G, H, I, J };
    Problem p = mk_bih2d(S, co, nbatch, strides, yc, xc, BCy, BCx, delxSSr, delxTr, delxSqr,
                         ratio, ratioSSr, ratioQtr, ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

int xinv_standard_2d_test_f64(double *S, const double *A, const double *B, const double *C,
                              const double *D, const double *E, const double *F, int64_t yc,
                              int64_t xc, double dely, double delx, int BCy, int BCx,
                              double delxSqr, double ratioQtr, double ratioSqr, double optArg,
                              double undef, double *flags, int64_t mxLoop, double tolerance)
{
    (void)dely;
    const double *co[6] = { A, B, C, D, E, F };
    Problem p = mk_std2dt(S, co, 1, nullptr, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr, ratioSqr,
                          optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, nullptr))
}

int xinv_standard_2d_test_f64_batched(double *S, const double *A, const double *B, const double *C,
                                      const double *D, const double *E, const double *F,
                                      int64_t nbatch, const int64_t *strides, int64_t yc,
                                      int64_t xc, double dely, double delx, int BCy, int BCx,
                                      double delxSqr, double ratioQtr, double ratioSqr,
                                      double optArg, double undef, double *flags, int64_t mxLoop,
                                      double tolerance, const xinv_options *opt)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *co[6] = { A, B, C, D, E, F };
    Problem p = mk_std2dt(S, co, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr,
                          ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_host(p, flags, opt))
}

int xinv_standard_2d_test_f64_dev(double *S, const double *A, const double *B, const double *C,
                                  const double *D, const double *E, const double *F,
                                  int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc,
                                  double dely, double delx, int BCy, int BCx, double delxSqr,
                                  double ratioQtr, double ratioSqr, double optArg, double undef,
                                  double *flags, int64_t mxLoop, double tolerance,
                                  const xinv_options *opt, void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *co[6] = { A, B, C, D, E, F };
    Problem p = mk_std2dt(S, co, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr,
                          ratioSqr, optArg, undef, mxLoop, tolerance);
    GUARD(solve_dev(p, flags, opt, (hipStream_t)stream))
}

// ---- resident plans (include/xinv.h: "resident plans") ---------------------------------------------------------
int xinv_plan_create_standard_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                         const double *F, int64_t nbatch, const int64_t *strides, int64_t yc,
                                         int64_t xc, double dely, double delx, int BCy, int BCx, double delxSqr,
                                         double ratioQtr, double ratioSqr, double optArg, double undef,
                                         const xinv_options *opt, void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std2d(nullptr, A, B, C, F, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr,
                         ratioSqr, optArg, undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_create_general_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                        const double *D, const double *E, const double *F, const double *G,
                                        int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc, double dely,
                                        double delx, int BCy, int BCx, double delxSqr, double ratio, double ratioQtr,
                                        double ratioSqr, double optArg, double undef, const xinv_options *opt,
                                        void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_gen2d(nullptr, A, B, C, D, E, F, G, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr, ratio,
                         ratioQtr, ratioSqr, optArg, undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_create_standard_3d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                         const double *F, int64_t nbatch, const int64_t *strides, int64_t zc,
                                         int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                                         int BCy, int BCx, double delxSqr, double ratio2Sqr, double ratio1Sqr,
                                         double optArg, double undef, const xinv_options *opt, void *stream)
{
    (void)delz; (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    Problem p = mk_std3d(nullptr, A, B, C, F, nbatch, strides, zc, yc, xc, BCz, BCy, BCx, delxSqr, ratio2Sqr,
                         ratio1Sqr, optArg, undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_create_general_3d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                        const double *D, const double *E, const double *F, const double *G,
                                        const double *H, int64_t nbatch, const int64_t *strides, int64_t zc,
                                        int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                                        int BCy, int BCx, double delxSqr, double ratio2, double ratio1,
                                        double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                                        const xinv_options *opt, void *stream)
{
    (void)delz; (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *c[8] = { A, B, C, D, E, F, G, H };
    Problem p = mk_gen3d(nullptr, c, nbatch, strides, zc, yc, xc, delx, BCz, BCy, BCx, delxSqr, ratio2, ratio1,
                         ratio2Sqr, ratio1Sqr, optArg, undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_create_general_bih_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                            const double *D, const double *E, const double *F, const double *G,
                                            const double *H, const double *I, const double *J, int64_t nbatch,
                                            const int64_t *strides, int64_t yc, int64_t xc, double dely, double delx,
                                            int BCy, int BCx, double delxSSr, double delxTr, double delxSqr,
                                            double ratio, double ratioSSr, double ratioQtr, double ratioSqr,
                                            double optArg, double undef, const xinv_options *opt, void *stream)
{
    (void)dely; (void)delx;
    if (!strides) return fail_arg("null strides");
    const double *co[10] = { A, B, C, D, E, F, G, H, I, J };
    Problem p = mk_bih2d(nullptr, co, nbatch, strides, yc, xc, BCy, BCx, delxSSr, delxTr, delxSqr, ratio, ratioSSr,
                         ratioQtr, ratioSqr, optArg, undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_create_standard_2d_test_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                              const double *D, const double *E, const double *F, int64_t nbatch,
                                              const int64_t *strides, int64_t yc, int64_t xc, double dely,
                                              double delx, int BCy, int BCx, double delxSqr, double ratioQtr,
                                              double ratioSqr, double optArg, double undef, const xinv_options *opt,
                                              void *stream)
{
    (void)dely;
    if (!strides) return fail_arg("null strides");
    const double *co[6] = { A, B, C, D, E, F };
    Problem p = mk_std2dt(nullptr, co, nbatch, strides, yc, xc, delx, BCy, BCx, delxSqr, ratioQtr, ratioSqr, optArg,
                          undef, 0, 0.0);
    GUARD(plan_create(plan, p, opt, (hipStream_t)stream))
}

int xinv_plan_solve_f64_dev(xinv_plan *plan, double *S, double *flags, int64_t mxLoop, double tolerance, void *stream)
{
    GUARD(plan_solve(plan, S, flags, mxLoop, tolerance, (hipStream_t)stream))
}

int xinv_plan_refresh(xinv_plan *plan, void *stream)
{
    if (!plan || plan->magic != XINV_PLAN_MAGIC) return fail_arg("xinv_plan_refresh: not a live plan");
    GUARD(plan_build(plan, (hipStream_t)stream))
}

int xinv_plan_destroy(xinv_plan *plan)
{
    if (!plan) return XINV_OK;
    if (plan->magic != XINV_PLAN_MAGIC) return fail_arg("xinv_plan_destroy: not a live plan");
    {   // (not while a solve on this device is using the plan's buffers)
        Workspace *ws = get_ws(plan->device);
        std::lock_guard<std::recursive_mutex> lock(ws->busy);
        plan_free(plan);
    }
    return XINV_OK;
}

// Gill-Matsuno (u, v) from the mass field; every array argument is a DEVICE pointer.
static int gm_flow_dev(const double *S, double *u, double *v, int64_t nbatch, int64_t yc, int64_t xc,
                       const double *ytab, const double *xtab, int yuniform, int xuniform,
                       const double *rowtab, double deg2m, int latlon, hipStream_t st)
{
    if (!S || !u || !v || !ytab || !xtab || !rowtab || nbatch < 1 || yc < 2 || xc < 2)
        return fail_arg("bad arguments to xinv_gm_flow_f64_dev");
    FlowArgs a;
    memset(&a, 0, sizeof a);
    a.S = S; a.u = u; a.v = v; a.nbatch = nbatch; a.yc = yc; a.xc = xc;
    // ytab / xtab: [3][n] interior weights a, b, c followed by {dx, dx0, dxn}
    a.gy.a = ytab; a.gy.b = ytab + yc; a.gy.c = ytab + 2 * yc; a.gy.uniform = yuniform;
    a.gx.a = xtab; a.gx.b = xtab + xc; a.gx.c = xtab + 2 * xc; a.gx.uniform = xuniform;
    double hy[3], hx[3];
    HIPCHK(hipMemcpyAsync(hy, ytab + 3 * yc, sizeof hy, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hx, xtab + 3 * xc, sizeof hx, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    a.gy.dx = hy[0]; a.gy.dx0 = hy[1]; a.gy.dxn = hy[2];
    a.gx.dx = hx[0]; a.gx.dx0 = hx[1]; a.gx.dxn = hx[2];
    a.coef1 = rowtab; a.coef2 = rowtab + yc; a.cosl = rowtab + 2 * yc;
    a.deg2m = deg2m; a.latlon = latlon;
    for (int64_t m0 = 0; m0 < nbatch; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nbatch - m0);
        FlowArgs b = a;
        b.S = S + m0 * yc * xc; b.u = u + m0 * yc * xc; b.v = v + m0 * yc * xc;
        if (yc > 65535) return fail_arg("xinv_gm_flow_f64_dev: yc > 65535 not supported");
        hipLaunchKernelGGL(k_gm_flow, dim3(cdiv(xc, 256), (unsigned)yc, (unsigned)nm), dim3(256), 0, st, b);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return XINV_OK;
}

int xinv_gm_flow_f64_dev(const double *S, double *u, double *v, int64_t nbatch, int64_t yc,
                         int64_t xc, const double *ytab, const double *xtab, int yuniform,
                         int xuniform, const double *rowtab, double deg2m, int latlon, void *stream)
{
    GUARD(gm_flow_dev(S, u, v, nbatch, yc, xc, ytab, xtab, yuniform, xuniform, rowtab, deg2m, latlon,
                      (hipStream_t)stream))
}

static int abs_norm_dev(const double *S, int64_t n, double undef, double *out, hipStream_t st)
{
    if (!S || !out || n < 1) return fail_arg("bad arguments to xinv_abs_norm_f64_dev");
    int device;
    HIPCHK(hipGetDevice(&device));
    Workspace *ws = get_ws(device);
    // shares the solver's partials / ctl buffers: one user of a device's workspace at a time
    std::lock_guard<std::recursive_mutex> ws_lock(ws->busy);
    int rc = ensure_dev(&ws->partials, &ws->partials_cap,
                        (size_t)XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long)) + 64);
    if (rc) return rc;
    rc = ensure_dev(&ws->ctl, &ws->ctl_cap, sizeof(XinvCtl));
    if (rc) return rc;
    NormArgs a;
    memset(&a, 0, sizeof a);
    a.S = S; a.sS = 0; a.n = n; a.undef = undef;
    a.psum = (double *)ws->partials;
    a.pcnt = (long long *)((char *)ws->partials + XINV_NORM_BLOCKS * sizeof(double));
    a.ctl = ws->ctl; a.force = 1; a.member0 = 0;
    int nblk = (int)std::min<int64_t>(XINV_NORM_BLOCKS, std::max<int64_t>(1, n / 2048));
    double *dout = (double *)((char *)ws->partials + XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long)));
    hipLaunchKernelGGL(k_norm_partial, dim3(nblk, 1, 1), dim3(256, 1, 1), 0, st, a);
    hipLaunchKernelGGL(k_norm_out, dim3(1), dim3(64), 0, st, a.psum, a.pcnt, nblk, dout);
    HIPCHK(hipMemcpyAsync(out, dout, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return XINV_OK;
}

int xinv_abs_norm_f64_dev(const double *S, int64_t n, double undef, double *out, void *stream)
{
    GUARD(abs_norm_dev(S, n, undef, out, (hipStream_t)stream))
}

} // extern "C"
