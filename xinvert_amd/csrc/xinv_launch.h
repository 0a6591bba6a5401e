// xinv_launch.h -- how a solve is laid out on the GPU: the plan (path, colours, tiling, x-uniform
// streams, masked-tile skipping), the template dispatch of the fused kernels, the colour-pass
// launches and the once-per-solve detection passes.  Included by xinv_hip.hip only.
#pragma once

// ------------------------------------------------------------------ launch helpers
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

#if XINV_TEST_HOOKS
static thread_local int *t_hook_record = nullptr;    // device int[3] {tile, launch tag, member}, or nullptr
#endif

struct Plan {
    int path, base, seam, ncol;
    int K, RY, nsg, nrb;     // nsg: 2-D = workgroups per member (partials sizing); 3-D = x strips
    int nkc, KC;             // 3-D: k chunks and planes per chunk
    bool K2;                 // 3-D standard form: passes of two sweeps (k_pipe3d) with the tiling below
    int nsg2, nrb2, nkc2, KC2;
    int joff2;               // k_pipe3d: rows its row blocks are shifted up by ('extend': p3_extend_joff; 0 elsewhere)
    int cus;                 // compute units the planner fills (xinv_options.cu_count, or the device's)
    int64_t srowf2;          // k_pipe3d: member stride of the record table (0: shared by the batch)
    bool bih_zbe;            // biharmonic one-pass kernel: B and E identically zero (terms left out)
    int bih_vm;              // ... where its coefficients come from: 0 per-row records (A..I constant along x), 1 A C D F
                             // as vector streams + the point-factor stream, 2 all nine (xinv_fusedbih.h); Q in ws->d_pfac
    bool aligned;
    unsigned umask;          // fused streams whose rows are constant along x (bit = stream index)
    unsigned um;             // the kernel variant's mask (subset of umask)
    bool even_split;         // rows split evenly over nrb row blocks (RY = average height)
    bool nine;               // 9-point form on the fused 4-colour kernel
    bool skip;               // masked-tile skipping: launches of K == Plan::K run the listed tiles only
    SkipNormArgs skipna;     // (the skipped tiles' geometry and list: k_skip_tiles in run_sweeps)
    int ntl, nskip;          // list entries per member (active, multiple of 4 / skipped)
    int skip_pct;            // share of wave-tiles skipped, percent
    int skip_ppm;            // ... per million
    double lone = 1.6;       // planner: cost of a workgroup alone on its CU relative to one of a pair
    bool pipe;               // K == 4 passes run the wave-pipelined kernel (k_pipe2d): one tile per workgroup
    int tpw;                 // wave-tiles per workgroup of the planned kernel: 4, or 1 with `pipe`
    int npair;               // `pipe`: column pairs per lane (1 or 2: strips of 112 or 240 owned columns)
    bool pipe_fr;            // `pipe`: the forcing rides the LDS ring (launches whose arrays exceed the caches)
    int64_t xc;              // the problem's row length (the ring layout's strip width depends on it: xinv_tiles.h)
    bool fma;                // XINV_FLAG_FMA: the contracted-arithmetic kernel variants (per-row-coefficient forms only)
    bool alias_ac;           // `pq`: A and C hold the same numbers everywhere -- C is read out of A (FusedGen2DQA, kernel mask um | 2)
    bool pq;                 // general form with A, C varying along x: the point-factor stream Q (FusedGen2DQ: relaxation
                             // factor and update predicate of every point, evaluated once per coefficient stack);
                             // Q lives in ws->d_pfac (a plan's own buffer while it solves)
    bool lag;                // 5-point 2-D kernels: norm + stop rule evaluated by k_norm_reduce_lag on a second stream,
                             // one pass behind the sweeps (three S buffers); see run_sweeps
};

// kernel variants instantiated per model: mask of streams read as one scalar per row
static unsigned pick_um(int kind, unsigned umask)
{
    if (kind == KIND_STD2D) return ((umask & 3u) == 3u) ? 3u : 0u;            // A, C
    if (kind == KIND_STD2DT) return ((umask & 7u) == 7u) ? 7u : 0u;           // A, D, E
    if ((umask & 0x1fu) == 0x1fu) return 0x1fu;                                // A, C, D, E, F
    if ((umask & 0x1cu) == 0x1cu) return 0x1cu;                                // D, E, F
    return 0u;
}

// edge strips whose row blocks are cut in two for a tiling of `nstrip` strips (one strip spans the row: it is both)
// columns a tile owns: 128 minus the 2K halo columns a side; the wave-pipelined pass has K = 4 and np column pairs per
// lane; the odd-xc periodic seam variants own one pair less (k_fused2d: SEAM)
static inline int strip_uw(const Plan &pl, int K, bool pipe)
{
    // (odd-xc periodic seam: the ring layout's strips, xinv_tiles.h)
    if (pl.seam) return xinv_ring_uw(pl.xc, pipe ? 2 * XINV_PIPE_P : 2 * K);
    return pipe ? XINV_PIPE_UW(pl.npair) : 128 - 4 * K;
}

// k_pipe3d's launch shape (which tiles of a launch march the whole column, which are cut into the plan's k chunks) and the
// 'extend' variant's tiling offset: xinv_tiles.h (xinv_p3_whole_tiles, xinv_p3_extend_joff; checked on the CPU).
static int64_t p3_whole_tiles(int64_t tiles, int nk, int64_t KC, int64_t zc, int cus, double *cost_out = nullptr)
{
    return xinv_p3_whole_tiles(tiles, nk, KC, zc, cus, cost_out);
}

static int p3_extend_joff(int64_t yc)
{
    static_assert(XINV_P3_RR == 3 && XINV_P3_G * XINV_P3_RR - 8 > 0, "k_pipe3d: three rows per wavefront");
    return xinv_p3_extend_joff(yc, XINV_P3_G * XINV_P3_RR - 8, 4, XINV_P3_RR);
}

static int fused_dispatch(int kind, bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                          hipStream_t st, const FusedArgs &a, int *occ, bool seam = false, bool fma = false, bool pq = false)
{
    if (pq) return (kind != KIND_GEN2D || seam || fma) ? 1 : xinv_launch_fused2d_genq(al, ext, um, K, grid, block, st, a, occ);   // (um | 2: the A == C variant)
    if (fma) {                                           // contracted arithmetic (xinv_fused.h: FusedStd2DF / FusedGen2DF)
        if (seam || kind == KIND_STD2DT) return 1;
        return kind == KIND_GEN2D ? xinv_launch_fused2d_genf(al, ext, um, K, grid, block, st, a, occ)
                                  : xinv_launch_fused2d_stdf(al, ext, um, K, grid, block, st, a, occ);
    }
    if (seam) {                                          // odd-xc periodic seam variants (xinv_fused.h: SEAM)
        if (kind == KIND_GEN2D) return xinv_launch_fused2d_gen_seam(al, ext, um, K, grid, block, st, a, occ);
        if (kind == KIND_STD2DT) return xinv_launch_fused2d_std2dt_seam(al, ext, um, K, grid, block, st, a, occ);
        return xinv_launch_fused2d_std_seam(al, ext, um, K, grid, block, st, a, occ);
    }
    if (kind == KIND_GEN2D) return xinv_launch_fused2d_gen(al, ext, um, K, grid, block, st, a, occ);
    if (kind == KIND_STD2DT) return xinv_launch_fused2d_std2dt(al, ext, um, K, grid, block, st, a, occ);
    return xinv_launch_fused2d_std(al, ext, um, K, grid, block, st, a, occ);
}

static int pipe_occ_cap();
static bool ptr_al16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// Lagged norm (run_sweeps): this launch publishes with `lag_tag` into the partial buffer of its parity;
// its extra workgroup evaluates `lag_prev`; `lag_out` receives what the evaluation of THIS launch needs.
// `a` is a FusedArgs / FusedBihArgs whose psum / xsum / xcnt / nwg are already final.
template <class A>
static void set_lag(A &a, const Workspace *ws, const Problem &p, int K, unsigned lag_tag, NormLagArgs *lag_out,
                    const NormLagArgs *lag_prev)
{
    if (lag_tag) {
        a.lag = 1; a.tag = lag_tag;
        a.psum = (unsigned long long *)((char *)ws->partials + (size_t)((lag_tag - 1u) & 1u) * ws->partials_half);
        if (lag_prev && lag_prev->tag) {
            a.lagp_psum = lag_prev->psum; a.lagp_xsum = lag_prev->xsum; a.lagp_xcnt = lag_prev->xcnt;
            a.lagp_NB = lag_prev->NB; a.lagp_K = lag_prev->K; a.lagp_tag = lag_prev->tag;
        }
    }
    if (lag_out) {
        memset(lag_out, 0, sizeof *lag_out);
        lag_out->psum = a.psum; lag_out->ctl = ws->ctl; lag_out->stop = p.stop;
        lag_out->xsum = a.xsum; lag_out->xcnt = a.xcnt; lag_out->NB = a.nwg; lag_out->K = K; lag_out->tag = lag_tag;
    }
}

static int launch_fused(const Problem &p, const Plan &pl, int K, const double *src, double *dst,
                        Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                        int no_ctl, unsigned lag_tag = 0, NormLagArgs *lag_out = nullptr,
                        const NormLagArgs *lag_prev = nullptr)
{
    FusedArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst;
    a.sS = p.sS;
    if (p.kind == KIND_STD2D) {
        a.c[0] = p.c[0]; a.sc[0] = p.sc[0];      // A
        a.c[1] = p.c[2]; a.sc[1] = p.sc[2];      // C
        a.c[2] = p.c[3]; a.sc[2] = p.sc[3];      // F
    } else if (p.kind == KIND_STD2DT) {
        a.c[0] = p.c[0]; a.sc[0] = p.sc[0];      // A
        for (int q = 3; q < 6; q++) { a.c[q - 2] = p.c[q]; a.sc[q - 2] = p.sc[q]; }   // D, E, F
    } else {
        a.c[0] = p.c[0]; a.sc[0] = p.sc[0];      // A
        for (int q = 2; q < 7; q++) { a.c[q - 1] = p.c[q]; a.sc[q - 1] = p.sc[q]; }   // C..G
        if (pl.pq) {                             // A, C, D, E, F, Q, G (FusedGen2DQ)
            a.c[6] = a.c[5]; a.sc[6] = a.sc[5];
            a.c[5] = (const double *)ws->d_pfac; a.sc[5] = p.yc * p.xc;
        }
    }
    a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    a.ext = (p.BCy == XINV_BC_EXTEND);
    a.tall = (p.yc > p.xc);
    a.RY = pl.even_split ? 0 : pl.RY;
    const bool pipe = pl.pipe && K == pl.K;          // wave-pipelined pass: one tile per workgroup
    const int UW = strip_uw(pl, K, pipe);
    a.nstrip = (int)cdiv(p.xc, UW);
    a.nrb = pl.even_split ? pl.nrb : (int)cdiv(p.yc, pl.RY);
    const int64_t ntiles = (int64_t)a.nstrip * a.nrb;
    a.nwg = (int)cdiv(ntiles, 4);
    a.force = force; a.no_ctl = no_ctl;
    a.member0 = member0;
    a.sc_ = p.sc_;
    a.ctl = ws->ctl;
    a.stop = p.stop;
    a.psum = (unsigned long long *)ws->partials;
    if (pipe) { a.nwg = (int)ntiles; a.rowf = ws->d_rowf; }
#if XINV_TEST_HOOKS
    a.dbg = (double *)t_hook_record;                 // (run_sweeps: XINV_HOOK_SKIP_PUBLISH; nullptr otherwise)
#endif
    if (pl.skip && K == pl.K) {                      // the lists were built for this K's strips
        a.tile_list = ws->d_list;
        a.ntl = pl.ntl;
        a.nwg = pl.ntl / pl.tpw;
        char *base = (char *)ws->d_tsum;
        const size_t nt = (size_t)p.nbatch * pl.nskip;
        a.xsum = (const double *)(base + nt * (sizeof(double) + sizeof(long long)));
        a.xcnt = (const long long *)(base + nt * (sizeof(double) + sizeof(long long)) + p.nbatch * sizeof(double));
    }
    set_lag(a, ws, p, K, lag_tag, lag_out, lag_prev);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {      // grid.y is limited to 65535
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)a.nwg + (lag_tag ? 1u : 0u), (unsigned)nm, 1), block(256, 1, 1);
        if (pipe) {
            // (XINV_PIPE_LDSPAD: unused dynamic LDS per workgroup, to cap the workgroups per CU in experiments;
            //  capping at the planned count changed nothing: the dispatcher already spreads them evenly)
            const int pad = std::max(0, XINV_ENV_INT("XINV_PIPE_LDSPAD", 0));
            xinv_launch_pipe2d(p.kind == KIND_GEN2D, pl.um, pl.npair, pl.pipe_fr, pl.aligned, a.ext != 0, grid, st, a, nullptr, pad, pl.seam != 0, pl.fma);
            continue;
        }
        if (fused_dispatch(p.kind, pl.aligned, a.ext != 0, pl.um | (pl.alias_ac ? 2u : 0u), K, grid, block, st, a, nullptr, pl.seam != 0, pl.fma, pl.pq))
            return fail_arg("unsupported sweeps_per_launch for this kernel variant");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

static int fused9_dispatch(int kind, int K, bool al, bool ext, dim3 grid, hipStream_t st,
                           const FusedArgs &a, int *occ, bool seam = false)
{
    return xinv_launch_fused9(kind == KIND_GEN2D, K, al, ext, grid, st, a, occ, seam);
}
// columns a wavefront of the 9-point kernel owns (one halo column per colour and sweep; the seam variants one pair less)
static inline int strip9_uw(const Plan &pl, int K) { return pl.seam ? xinv_ring_uw(pl.xc, 4 * K) : 128 - 8 * K; }   // (seam: the ring layout, xinv_tiles.h)

static int launch_fused9(const Problem &p, const Plan &pl, int K, const double *src, double *dst,
                         Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                         int no_ctl, unsigned lag_tag = 0, NormLagArgs *lag_out = nullptr,
                         const NormLagArgs *lag_prev = nullptr)
{
    FusedArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst; a.sS = p.sS;
    const int nc = (p.kind == KIND_GEN2D) ? 7 : 4;
    for (int q = 0; q < nc; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
    a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    a.ext = (p.BCy == XINV_BC_EXTEND);
    a.tall = (p.yc > p.xc);
    a.RY = pl.even_split ? 0 : pl.RY;
    a.nstrip = (int)cdiv(p.xc, strip9_uw(pl, K));
    a.nrb = pl.even_split ? pl.nrb : (int)cdiv(p.yc, pl.RY);
    a.nwg = (int)cdiv((int64_t)a.nstrip * a.nrb, 4);
    a.force = force; a.no_ctl = no_ctl;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    const size_t NBmax = (size_t)pl.nsg;
    a.psum = (unsigned long long *)ws->partials;
    if (pl.skip && K == pl.K) {                      // fully masked tiles are left out (plan_tile_skip)
        a.tile_list = ws->d_list;
        a.ntl = pl.ntl;
        a.nwg = pl.ntl / 4;
        char *base = (char *)ws->d_tsum;
        const size_t nt = (size_t)p.nbatch * pl.nskip;
        a.xsum = (const double *)(base + nt * (sizeof(double) + sizeof(long long)));
        a.xcnt = (const long long *)(base + nt * (sizeof(double) + sizeof(long long)) + p.nbatch * sizeof(double));
    }
    set_lag(a, ws, p, K, lag_tag, lag_out, lag_prev);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)a.nwg + (lag_tag ? 1u : 0u), (unsigned)nm, 1);
        if (fused9_dispatch(p.kind, K, pl.aligned, a.ext != 0, grid, st, a, nullptr, pl.seam != 0))
            return fail_arg("unsupported sweeps_per_launch for the 9-point kernel");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

static int launch_fused3d(const Problem &p, const Plan &pl, int K, const double *src, double *dst,
                          Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                          int no_ctl)
{
    Fused3Args a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst; a.sS = p.sS;
    for (int q = 0; q < 4; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
    a.zc = p.zc; a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    if (K == 2) {                                        // two sweeps per pass: its own tiling
        if (!pl.K2) return fail_arg("internal: two-sweep 3-D pass without its plan");
        a.nstrip = pl.nsg2; a.njb = pl.nrb2; a.joff = pl.joff2;
        a.nkc = std::max(1, pl.nkc2); a.KC = pl.KC2;
        a.force = force; a.no_ctl = no_ctl;
        a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
        a.psum = (unsigned long long *)ws->partials;
        const int64_t NT2 = (int64_t)a.nstrip * a.njb;
        // a flat grid over the members of the launch (k_pipe3d); at most 2^30 workgroups per launch
        const int64_t mstep = std::max<int64_t>(1, std::min<int64_t>(XINV_MEMBER_CHUNK, ((int64_t)1 << 30) / (NT2 * a.nkc)));
        const bool ext2 = (p.BCy == XINV_BC_EXTEND);
        for (int64_t m0 = 0; m0 < nmem; m0 += mstep) {
            const int64_t nm = std::min<int64_t>(mstep, nmem - m0);
            a.member0 = member0 + m0;
            a.rowf = (const double *)ws->d_rowf; a.srowf = pl.srowf2;
            if (ext2) {
                // 'extend': the pre-pass of the pass's first sweep, in place on the source (numbas.py:87-115; idempotent: a
                // pass redone from this source by the one-sweep kernel applies it again to the same effect); the second
                // sweep's is applied inside the kernel (xinv_pipe3d.h: EXT).  A no-op for a member that has stopped.
                ExtendArgs e;
                e.S = const_cast<double *>(src); e.sS = p.sS; e.yc = p.yc; e.xc = p.xc;
                e.kfirst = 1; e.nk = p.zc - 2;
                e.per = a.per; e.tall = 0; e.force = force;
                e.undef = p.sc_.undef; e.ctl = ws->ctl; e.member0 = a.member0;
                for (int64_t q0 = 0; q0 < nm; q0 += 32768) {     // (grid.z is limited to 65535)
                    const int64_t nq = std::min<int64_t>(32768, nm - q0);
                    e.member0 = a.member0 + q0;
                    hipLaunchKernelGGL(k_extend, dim3(cdiv(p.xc, 256), (unsigned)e.nk, (unsigned)nq), dim3(256, 1, 1), 0, st, e);
                }
            }
            // which tiles march the whole column, which are cut into the plan's k chunks (p3_whole_tiles)
            a.nfull = p3_whole_tiles(NT2 * nm, a.nkc, a.KC, p.zc, pl.cus);
            const int64_t nwg = a.nfull + (NT2 * nm - a.nfull) * a.nkc;
            if (pl.fma) xinv_launch_pipe3d_fma(pl.aligned, dim3((unsigned)nwg, 1, 1), st, a);
            else        xinv_launch_pipe3d(pl.aligned, dim3((unsigned)nwg, 1, 1), st, a, pl.seam != 0, ext2);
        }
        HIPCHK(hipGetLastError());
        return XINV_OK;
    }
    a.nstrip = pl.nsg; a.njb = pl.nrb;
    a.nkc = std::max(1, pl.nkc); a.KC = pl.KC;
    a.force = force; a.no_ctl = no_ctl; a.member0 = member0;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    const size_t NB = (size_t)pl.nsg * pl.nrb * a.nkc;
    a.psum = (unsigned long long *)ws->partials;
    const bool ext = (p.BCy == XINV_BC_EXTEND), uni = (pl.um == 7u);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)NB, (unsigned)nm, 1);
        if (pl.seam ? xinv_launch_fused3d_seam(pl.RY, uni, ext, grid, st, a)
            : pl.fma ? xinv_launch_fused3d_fma(pl.RY, pl.aligned, ext, grid, st, a)
                     : xinv_launch_fused3d(pl.RY, pl.aligned, uni, ext, grid, st, a))
            return fail_arg("internal: no 3-D kernel variant for this cross-section");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

// ---- biharmonic one-pass launch (A..I x-uniform) ----------------------------------------------
static int launch_fusedbih(const Problem &p, const Plan &pl, const double *src, double *dst,
                           Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                           int no_ctl, unsigned lag_tag = 0, NormLagArgs *lag_out = nullptr,
                           const NormLagArgs *lag_prev = nullptr, bool prepass = true)
{
    const bool per = (p.BCx == XINV_BC_PERIODIC);
    // (prepass = false: finalise() redoing a pass whose source already carries that pass's pre-pass -- the periodic
    //  pre-pass, r0 <- r1 then r1 <- r2, is not idempotent)
    if (p.BCy == XINV_BC_EXTEND && prepass) {        // the kernel's own pre-pass, on the source buffer
        ExtendArgs e;
        e.S = const_cast<double *>(src); e.sS = p.sS; e.yc = p.yc; e.xc = p.xc; e.kfirst = 0; e.nk = 1;
        e.per = per; e.tall = (p.yc > p.xc); e.force = force;
        e.undef = p.sc_.undef; e.ctl = ws->ctl;
        for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
            const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
            e.member0 = member0 + m0;
            hipLaunchKernelGGL(k_extend_bih, dim3(cdiv(p.xc, 256), 1, (unsigned)nm), dim3(256, 1, 1), 0, st, e);
        }
    }
    FusedBihArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst; a.sS = p.sS;
    for (int q = 0; q < 10; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
    a.yc = p.yc; a.xc = p.xc; a.per = per;
    a.nstrip = (int)cdiv(p.xc, XINV_BIH_OWN(per)); a.nrb = pl.nrb; a.RB = pl.RY;
    a.nwg = (int)cdiv((int64_t)a.nstrip * a.nrb, 4);
    a.force = force; a.no_ctl = no_ctl;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    a.rowf = (const double *)ws->d_rowf;
    a.q = pl.bih_vm ? (const double *)ws->d_pfac : nullptr;
    const size_t NBmax = (size_t)pl.nsg;
    a.psum = (unsigned long long *)ws->partials;
    if (pl.skip) {                                   // fully masked tiles are left out (plan_tile_skip)
        a.tile_list = ws->d_list;
        a.ntl = pl.ntl;
        a.nwg = pl.ntl / 4;
        char *base = (char *)ws->d_tsum;
        const size_t nt = (size_t)p.nbatch * pl.nskip;
        a.xsum = (const double *)(base + nt * (sizeof(double) + sizeof(long long)));
        a.xcnt = (const long long *)(base + nt * (sizeof(double) + sizeof(long long)) + p.nbatch * sizeof(double));
    }
    set_lag(a, ws, p, 1, lag_tag, lag_out, lag_prev);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)a.nwg + (lag_tag ? 1u : 0u), (unsigned)nm, 1), block(256, 1, 1);
        (void)block;
        xinv_launch_fusedbih(per, pl.bih_zbe, pl.bih_vm, grid, st, a, nullptr);
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

static int launch_fused3dg(const Problem &p, const Plan &pl, const double *src, double *dst,
                           Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                           int no_ctl)
{
    Fused3GArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst; a.sS = p.sS;
    for (int q = 0; q < 8; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
    a.zc = p.zc; a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    a.nstrip = pl.nsg; a.njb = pl.nrb;
    a.nkc = std::max(1, pl.nkc); a.KC = pl.KC;
    a.force = force; a.no_ctl = no_ctl; a.member0 = member0;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    const size_t NB = (size_t)pl.nsg * pl.nrb * a.nkc;
    a.psum = (unsigned long long *)ws->partials;
    const bool ext = (p.BCy == XINV_BC_EXTEND);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)NB, (unsigned)nm, 1);
        if (pl.seam ? xinv_launch_fused3dg_seam(pl.RY, ext, grid, st, a) : xinv_launch_fused3dg(pl.RY, pl.aligned, ext, grid, st, a))
            return fail_arg("internal: no general 3-D kernel variant for this cross-section");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

// one full coloured sweep (+ norm + stop rule) in place on p.S
// Order of the colour launches of one sweep (the oracle's seq_colour): with the odd-xc periodic seam each seam colour
// (column xc-1, rows of one parity) runs right after the base colour its points would otherwise belong to.
static inline int seam_sequence(int pos, int base, int seam)
{
    static const int s2[4] = {0, 2, 1, 3}, s4[6] = {0, 4, 1, 2, 5, 3};
    if (!seam) return pos;
    return base == 2 ? s2[pos] : s4[pos];
}

static int launch_colour_chunk(const Problem &p, const Plan &pl, Workspace *ws, hipStream_t st,
                               int64_t m0, int64_t nm)
{
    const int per = (p.BCx == XINV_BC_PERIODIC);
    if (p.kind == KIND_BIH2D) {
        if (p.BCy == XINV_BC_EXTEND) {
            ExtendArgs e;
            e.S = p.S; e.sS = p.sS; e.yc = p.yc; e.xc = p.xc; e.kfirst = 0; e.nk = 1;
            e.per = per; e.tall = (p.yc > p.xc); e.force = 0;
            e.undef = p.sc_.undef; e.ctl = ws->ctl; e.member0 = m0;
            hipLaunchKernelGGL(k_extend_bih, dim3(cdiv(p.xc, 256), 1, (unsigned)nm), dim3(256, 1, 1), 0, st, e);
        }
        ColourArgsBih a;
        memset(&a, 0, sizeof a);
        a.S = p.S; a.sS = p.sS;
        for (int q = 0; q < 10; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
        a.yc = p.yc; a.xc = p.xc; a.per = per; a.trail = pl.seam; a.force = 0; a.member0 = m0;
        a.sc_ = p.sc_; a.ctl = ws->ctl; a.umask = pl.umask;
        dim3 b(64, 4, 1);
        dim3 g(cdiv(cdiv(p.xc, 3) + 1, 64), cdiv(cdiv(p.yc, 3) + 1, 4), (unsigned)nm);
        if (!per || p.xc % 3 == 0) {
            // one launch per row class does its three column colours in registers (periodic x needs
            // xc % 3 == 0 so that the wrap keeps colour == column % 3; otherwise nine + trailing colours)
            dim3 gb(cdiv(cdiv(p.xc, 180), 4), cdiv(p.yc, 3) + 1, (unsigned)nm), bb(256, 1, 1);
            a.Y = ws->S2; a.sY = p.sS;                     // side buffer (allocated by solve_dev for this path)
            for (int cj = 0; cj < 3; cj++) {
                a.colour = cj;
                if (per) {
                    if (pl.umask) hipLaunchKernelGGL((k_bih_rowclass<true, true>), gb, bb, 0, st, a);
                    else          hipLaunchKernelGGL((k_bih_rowclass<false, true>), gb, bb, 0, st, a);
                } else {
                    if (pl.umask) hipLaunchKernelGGL((k_bih_rowclass<true, false>), gb, bb, 0, st, a);
                    else          hipLaunchKernelGGL((k_bih_rowclass<false, false>), gb, bb, 0, st, a);
                }
            }
            hipLaunchKernelGGL(k_rows_copy_back, dim3(256, (unsigned)nm, 1), dim3(256), 0, st,
                               (const double *)ws->S2, p.sS, p.S, p.sS, p.yc, p.xc,
                               (const XinvCtl *)ws->ctl, m0, 0);
        } else
        for (int cc = 0; cc < pl.ncol; cc++) {
            a.colour = cc;
            if (pl.umask) hipLaunchKernelGGL(k_colour_bih2d<true>, g, b, 0, st, a);
            else          hipLaunchKernelGGL(k_colour_bih2d<false>, g, b, 0, st, a);
        }
    } else if (p.BCy == XINV_BC_EXTEND) {
        ExtendArgs e;
        e.S = p.S; e.sS = p.sS; e.yc = p.yc; e.xc = p.xc;
        e.kfirst = is3d(p.kind) ? 1 : 0;
        e.nk = is3d(p.kind) ? p.zc - 2 : 1;
        // the standard 3-D kernel's second loop stays inside the row (numbas.py:104-108)
        e.per = per; e.tall = (p.kind != KIND_STD3D) && (p.yc > p.xc); e.force = 0;
        e.undef = p.sc_.undef; e.ctl = ws->ctl; e.member0 = m0;
        dim3 g(cdiv(p.xc, 256), (unsigned)e.nk, (unsigned)nm), b(256, 1, 1);
        hipLaunchKernelGGL(k_extend, g, b, 0, st, e);
    }
    if (p.kind == KIND_BIH2D) {
        // sweeps launched above
    } else if (is3d(p.kind)) {
        ColourArgs3D a;
        memset(&a, 0, sizeof a);
        a.S = p.S; a.sS = p.sS;
        for (int q = 0; q < p.ncoef; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
        a.zc = p.zc; a.yc = p.yc; a.xc = p.xc;
        a.per = per; a.seam = pl.seam; a.force = 0; a.sc_ = p.sc_; a.ctl = ws->ctl;
        a.nbatch = p.nbatch; a.member0 = m0;
        dim3 b(64, 4, 1);
        dim3 g(cdiv(cdiv(p.xc, 2) + 1, 64), cdiv(p.yc - 2, 4), (unsigned)(nm * (p.zc - 2)));
        for (int pos = 0; pos < pl.ncol; pos++) {
            a.colour = seam_sequence(pos, 2, pl.seam);
            if (p.kind == KIND_GEN3D) hipLaunchKernelGGL(k_colour_gen3d, g, b, 0, st, a);
            else                      hipLaunchKernelGGL(k_colour_std3d, g, b, 0, st, a);
        }
    } else {
        ColourArgs2D a;
        memset(&a, 0, sizeof a);
        a.S = p.S; a.sS = p.sS;
        for (int q = 0; q < p.ncoef; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
        a.yc = p.yc; a.xc = p.xc;
        a.per = per; a.base = pl.base; a.seam = pl.seam; a.force = 0;
        a.sc_ = p.sc_; a.ctl = ws->ctl; a.member0 = m0;
        dim3 b(64, 4, 1);
        dim3 g(cdiv(cdiv(p.xc, 2) + 1, 64), cdiv(p.yc - 2, 4), (unsigned)nm);
        const bool nine = (pl.base == 4);
        for (int pos = 0; pos < pl.ncol; pos++) {
            a.colour = seam_sequence(pos, pl.base, pl.seam);
            if (p.kind == KIND_STD2D) {
                if (nine) hipLaunchKernelGGL(k_colour_std2d<true>, g, b, 0, st, a);
                else      hipLaunchKernelGGL(k_colour_std2d<false>, g, b, 0, st, a);
            } else if (p.kind == KIND_STD2DT) {
                if (nine) hipLaunchKernelGGL(k_colour_std2dt<true>, g, b, 0, st, a);
                else      hipLaunchKernelGGL(k_colour_std2dt<false>, g, b, 0, st, a);
            } else {
                if (nine) hipLaunchKernelGGL(k_colour_gen2d<true>, g, b, 0, st, a);
                else      hipLaunchKernelGGL(k_colour_gen2d<false>, g, b, 0, st, a);
            }
        }
    }
    NormArgs n;
    n.S = p.S; n.sS = p.sS; n.n = p.zc * p.yc * p.xc; n.undef = p.sc_.undef;
    n.psum = (double *)ws->partials;
    n.pcnt = (long long *)((char *)ws->partials + p.nbatch * XINV_NORM_BLOCKS * sizeof(double));
    n.ctl = ws->ctl; n.stop = p.stop; n.force = 0; n.member0 = m0;
    int nblk = (int)std::min<int64_t>(XINV_NORM_BLOCKS, std::max<int64_t>(1, n.n / 2048));
    hipLaunchKernelGGL(k_norm_partial, dim3(nblk, (unsigned)nm, 1), dim3(256, 1, 1), 0, st, n);
    hipLaunchKernelGGL(k_norm_final, dim3((unsigned)nm, 1, 1), dim3(64, 1, 1), 0, st, n, nblk);
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

static int launch_colour_sweep(const Problem &p, const Plan &pl, Workspace *ws, hipStream_t st)
{
    // grid.z carries members (x planes in 3-D) and is limited to 65535
    int64_t chunk = XINV_MEMBER_CHUNK;
    if (is3d(p.kind)) chunk = std::max<int64_t>(1, 65535 / std::max<int64_t>(1, p.zc - 2));
    for (int64_t m0 = 0; m0 < p.nbatch; m0 += chunk) {
        int rc = launch_colour_chunk(p, pl, ws, st, m0, std::min<int64_t>(chunk, p.nbatch - m0));
        if (rc) return rc;
    }
    return XINV_OK;
}

// Number of row blocks for the fused 2-D kernels.  Tall tiles amortise the 4K recomputed halo
// rows, but every CU should hold the same number of workgroups: `occ` of the chosen variant fit
// per CU (register-limited, queried from the runtime).  Minimise (workgroups per CU, in rounds of
// 256*occ resident ones) x (steps per tile); rows are then split evenly over the blocks.
static int64_t choose_row_blocks(int64_t yc, int64_t nstrip, int64_t nbatch, int K, int occ, double lone = 1.6,
                                 bool pipe = false)
{
    occ = std::max(1, std::min(occ, pipe ? pipe_occ_cap() : 3));
    const int64_t cap = 256 * (int64_t)occ, period = pipe ? 4 : 2 * K + 2;
    int64_t best = 1; double best_cost = 1e300;
    const int64_t nmin = std::max<int64_t>(1, cdiv(yc, pipe ? 512 : 128)), nmax = std::max<int64_t>(nmin, yc / 4);
    for (int64_t nr = nmin; nr <= nmax; nr++) {
        const int64_t rows = cdiv(yc, nr) + 1;                     // +1: even rounding
        const int64_t steps = pipe ? cdiv(rows + 4 + 3 * XINV_PIPE_LAG, 8) * 8
                                   : cdiv(rows + 4 * K, period) * period;
        const int64_t wgs = (int64_t)cdiv(nstrip * nr, pipe ? 1 : 4) * nbatch;
        // rounds of `cap` resident workgroups; inside a round a CU holds ceil(w/256) of them,
        // and a lone workgroup on a CU leaves issue slots idle (charged like `lone`: 1.6 for the
        // issue-bound variants with one or two vector streams, ~1 for the bandwidth-bound ones)
        const int64_t rounds = cdiv(wgs, cap);
        const int64_t w_last = wgs - (rounds - 1) * cap;
        // (pipelined kernel, measured at 3600x1800: a step of n workgroups on a CU costs ~1.5 + n -- 2, 3, 4
        //  per CU: 0.346, 0.445, 0.543 us -- the wavefronts wait for each other at the step barriers, and more
        //  of them per SIMD fill the gaps)
        const double full = pipe ? 1.5 + occ : ((occ == 1) ? lone : (double)occ);
        const double last = pipe ? 1.5 + (double)cdiv(w_last, 256) : ((w_last <= 256) ? lone : (double)cdiv(w_last, 256));
        const double cost = ((double)(rounds - 1) * full + last) * (double)steps;
        if (cost <= best_cost * 1.0001) { best_cost = std::min(cost, best_cost); best = nr; }   // ties: more, shorter tiles
    }
    return best;
}

// Cost of a fused 2-D launch in (workgroups per CU) x (steps per tile) units -- the model behind
// choose_row_blocks, shared with the masked-tile planner.
// (pipelined kernel: up to XINV_PIPE_OCC workgroups -- one wavefront each per SIMD -- share a CU)
static int pipe_occ_cap()
{
    return std::max(1, XINV_ENV_INT("XINV_PIPE_OCC", 5));
}

static double tile_cost(int64_t wgs, int64_t rows, int K, int occ, double lone = 1.6, bool pipe = false)
{
    occ = std::max(1, std::min(occ, pipe ? pipe_occ_cap() : 3));
    const int64_t cap = 256 * (int64_t)occ, period = pipe ? 4 : 2 * K + 2;
    // (pipelined: the last wavefront starts 3 x LAG steps late and enters RY + 4 rows)
    const int64_t steps = pipe ? cdiv(rows + 1 + 4 + 3 * XINV_PIPE_LAG, 8) * 8
                               : cdiv(rows + 1 + 4 * K, period) * period;
    const int64_t rounds = std::max<int64_t>(1, cdiv(wgs, cap));
    const int64_t w_last = wgs - (rounds - 1) * cap;
    const double full = pipe ? 1.5 + occ : ((occ == 1) ? lone : (double)occ);
    const double last = pipe ? 1.5 + (double)cdiv(w_last, 256) : ((w_last <= 256) ? lone : (double)cdiv(w_last, 256));
    return ((double)(rounds - 1) * full + last) * (double)steps;
}

// Masked-tile skipping for the 5-point fused kernels.  Wave-tiles whose forcing is undefined at
// every owned point (land, topography, polar caps) can never change (every mask predicate of the
// reference tests the forcing), so launches run the other tiles only; the row split is re-chosen
// so that the ACTIVE tiles fill the CUs evenly, and the skipped tiles' constant share of the norm
// is computed once.  Decided per solve from one pass over the forcing.
// `fixedRB` > 0 (biharmonic one-pass kernel): tiles are row blocks of exactly fixedRB rows x `UW_` owned
// columns and the split is kept; 0: the 5-point kernels' even split, re-planned for the active tiles.
// The activity map of the forcing for strips of UW owned columns: kernel + copy to the host, NOT synchronised.  The planner
// of the lat-lon forms issues it ahead of the x-uniform detection with the strip width those forms end up with, so that
// ONE host round trip brings both answers (25 us of a 4.6 ms headline solve); plan_tile_skip issues it itself otherwise.
static int issue_strip_active(const Problem &p, Workspace *ws, hipStream_t st, int UW, int fi)
{
    const int64_t yc = p.yc, nb = p.nbatch;
    const int nstrip = (int)cdiv(p.xc, UW);
    const int64_t cells = yc * nstrip;
    int rc = ensure_dev(&ws->d_act, &ws->d_act_cap, (size_t)(nb * cells));
    if (rc) return rc;
    if (ws->h_act_cap < (size_t)(nb * cells)) {
        if (ws->h_act) HIPCHK(hipHostFree(ws->h_act));
        HIPCHK(hipHostMalloc((void **)&ws->h_act, (size_t)(nb * cells), hipHostMallocDefault));
        ws->h_act_cap = (size_t)(nb * cells);
    }
    StripActArgs sa;
    sa.f = p.c[fi]; sa.sf = p.sc[fi]; sa.yc = yc; sa.xc = p.xc; sa.nstrip = nstrip; sa.UW = UW;
    sa.undef = p.sc_.undef; sa.act = ws->d_act;
    hipLaunchKernelGGL(k_strip_active, dim3(cdiv(cells, 4), (unsigned)nb, 1), dim3(256), 0, st, sa);
    HIPCHK(hipMemcpyAsync(ws->h_act, ws->d_act, (size_t)(nb * cells), hipMemcpyDeviceToHost, st));
    ws->act_ready = true; ws->act_uw = UW; ws->act_f = p.c[fi]; ws->act_synced = false;
    return XINV_OK;
}

static int plan_tile_skip(const Problem &p, Plan &pl, Workspace *ws, hipStream_t st,
                          const xinv_options &opt, int fixedRB = 0, int UW_ = 0, int occ_ = 0)
{
    pl.skip = false; pl.ntl = pl.nskip = 0; pl.skip_pct = 0; pl.skip_ppm = 0;
    const bool forced = (opt.flags & XINV_FLAG_FORCE_TILE_SKIP) != 0;
    const int tpw = pl.pipe ? 1 : 4;                              // wave-tiles per workgroup
    const int K = pl.K, UW = UW_ ? UW_ : strip_uw(pl, K, pl.pipe);   // 9-point kernel: 128 - 8K owned columns
    const int nstrip = (int)cdiv(p.xc, UW);
    const int64_t wscale = 4 / tpw;                       // (thresholds in wavefronts: a pipelined tile has four)
    if (!forced && ((int64_t)nstrip * pl.nrb * p.nbatch * wscale < 1024 || (int64_t)nstrip * pl.nrb * wscale < 64 ||
                    p.nbatch > 64))
        return XINV_OK;                                   // small problems: nothing to balance
    const int64_t yc = p.yc, nb = p.nbatch;
    const int64_t cells = yc * nstrip;
    if (nb * nstrip * (yc + 1) > (int64_t)50000000) return XINV_OK;   // host-side prefix table would exceed 200 MB
    const int fi = (p.kind == KIND_STD2D) ? 3 : (p.kind == KIND_GEN2D ? 6 : (p.kind == KIND_BIH2D ? 9 : 5));   // the forcing

    int rc = XINV_OK;
    if (!(ws->act_ready && ws->act_uw == UW && ws->act_f == p.c[fi])) {     // (not already there: issue_strip_active below)
        rc = issue_strip_active(p, ws, st, UW, fi);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(st));
    } else if (!ws->act_synced)                           // (issued ahead, and no detection pass synchronised behind it)
        HIPCHK(hipStreamSynchronize(st));
    ws->act_ready = false;

    // prefix counts of active rows per (member, strip), laid out [member][row][strip]: building them and looking up the
    // two rows of a row block for every strip both run along contiguous memory (this planning is host time inside every
    // solve: 155 us of a 4.9 ms headline solve with the [member][strip][row] layout and per-tile id decoding)
    std::vector<int> &pre = ws->h_pre;
    pre.resize((size_t)(nb * nstrip * (yc + 1)));
    int64_t nact = 0;
    for (int64_t m = 0; m < nb; m++) {
        int *q = &pre[(size_t)(m * (yc + 1) * nstrip)];
        for (int s = 0; s < nstrip; s++) q[s] = 0;
        const unsigned char *ha = ws->h_act + m * yc * nstrip;
        for (int64_t r = 0; r < yc; r++, q += nstrip, ha += nstrip)
            for (int s = 0; s < nstrip; s++) q[nstrip + s] = q[s] + ha[s];
        for (int s = 0; s < nstrip; s++) nact += q[s];
    }
    if (!forced && (double)nact > 0.92 * (double)(nb * cells)) return XINV_OK;     // little to skip
    auto rows_active = [&](int64_t m, int strip, int64_t y0, int64_t y1) {
        const int *q = &pre[(size_t)(m * (yc + 1) * nstrip)];
        return q[y1 * nstrip + strip] - q[y0 * nstrip + strip] > 0;
    };

    const bool ext = (p.BCy == XINV_BC_EXTEND);
    // (tile ids and their rows: xinv_tile_rows -- with the odd-xc periodic seam the edge strips' row blocks are two tiles)
    auto ntile_ids = [&](int nrb) { return nstrip * nrb; };
    auto id_rb = [&](int, int id) { return id / nstrip; };
    auto tile_active = [&](int64_t m, int nrb, int id) {
        const int rb = id_rb(nrb, id);
        const TileRows t = xinv_tile_rows(id, nstrip, nrb, yc, fixedRB);
        if (t.y0 >= t.y1) return false;
        if (ext && (rb == 0 || rb >= nrb - (fixedRB ? 2 : 1))) return true;    // the boundary rows get their copy
                                                               // (a last block of one row: yc-2 sits in the one before)
        return rows_active(m, t.strip, t.y0, t.y1);
    };
    auto active_wgs = [&](int nrb, int64_t *maxact) {
        int64_t wgs = 0, mx = 0;
        for (int64_t m = 0; m < nb; m++) {
            int64_t c = 0;
            const int *q = &pre[(size_t)(m * (yc + 1) * nstrip)];                // (the rows of a block once, then its strips)
            for (int rb = 0; rb < nrb; rb++) {
                const TileRows t = xinv_tile_rows(rb * nstrip, nstrip, nrb, yc, fixedRB);
                if (t.y0 >= t.y1) continue;
                if (ext && (rb == 0 || rb >= nrb - (fixedRB ? 2 : 1))) { c += nstrip; continue; }
                const int *q0 = q + t.y0 * nstrip, *q1 = q + t.y1 * nstrip;
                for (int st_ = 0; st_ < nstrip; st_++) c += (q1[st_] - q0[st_]) > 0;
            }
            wgs += cdiv(c, tpw); mx = std::max(mx, c);
        }
        if (maxact) *maxact = mx;
        return wgs;
    };
    int occ = occ_ > 0 ? occ_ : 2;
    if (!fixedRB && occ_ <= 0) {
        FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
        if (pl.pipe) xinv_launch_pipe2d(p.kind == KIND_GEN2D, pl.um, pl.npair, pl.pipe_fr, pl.aligned, ext, dim3(1), st, dummy, &occ, 0, pl.seam != 0, pl.fma);
        else fused_dispatch(p.kind, pl.aligned, ext, pl.um | (pl.alias_ac ? 2u : 0u), K, dim3(1), dim3(256), st, dummy, &occ, pl.seam != 0, pl.fma, pl.pq);
    }
    const bool pp = pl.pipe;
    const double cost0 = tile_cost((int64_t)cdiv((int64_t)nstrip * pl.nrb, tpw) * nb, cdiv(yc, pl.nrb), K, occ, pl.lone, pp);
    // candidates: the row split that brings the ACTIVE workgroups back to the default count sits
    // near nrb / (active share); search a window around it
    int best = pl.nrb; double best_cost = 1e300;
    int lo = pl.nrb, hi = pl.nrb;
    if (!forced && opt.rows_per_tile == 0 && !fixedRB) {
        const int64_t w0 = active_wgs(pl.nrb, nullptr);
        const double share = std::max(0.05, (double)w0 / (double)((int64_t)cdiv((int64_t)nstrip * pl.nrb, tpw) * nb));
        const double centre = (double)pl.nrb / share;
        const int64_t cap_rows = std::max<int64_t>(pl.nrb, yc / 4);
        lo = (int)std::min<int64_t>(cap_rows, std::max<int64_t>(pl.nrb, (int64_t)(centre * 0.85)));
        hi = (int)std::min<int64_t>(cap_rows, std::max<int64_t>(lo, (int64_t)(centre * 1.10) + 1));
    }
    best_cost = tile_cost(active_wgs(pl.nrb, nullptr), cdiv(yc, pl.nrb), K, occ, pl.lone, pp);   // keep the split, skip only
    for (int nrb = lo; nrb <= hi; nrb++) {
        const double c = tile_cost(active_wgs(nrb, nullptr), cdiv(yc, nrb), K, occ, pl.lone, pp);
        if (c < best_cost) { best_cost = c; best = nrb; }
    }
    if (!forced && best_cost > 0.95 * cost0) return XINV_OK;            // (fixed split: skipping must save 5 % of the workgroups)

    // lists for the chosen split
    int64_t maxact = 0;
    active_wgs(best, &maxact);
    const int64_t ntiles = (int64_t)ntile_ids(best);
    const int ntl = (int)(tpw * std::max<int64_t>(1, cdiv(maxact, tpw)));
    int64_t maxskip = 0, nskipped = 0;
    std::vector<std::vector<int>> act((size_t)nb), skp((size_t)nb);
    for (int64_t m = 0; m < nb; m++) {
        for (int id = 0; id < (int)ntiles; id++) {
            const TileRows t = xinv_tile_rows(id, nstrip, best, yc, fixedRB);
            if (t.y0 >= t.y1) continue;
            (tile_active(m, best, id) ? act[(size_t)m] : skp[(size_t)m]).push_back(id);
        }
        maxskip = std::max<int64_t>(maxskip, (int64_t)skp[(size_t)m].size());
        nskipped += (int64_t)skp[(size_t)m].size();
    }
    if (nskipped == 0) return XINV_OK;
    const int nskip = (int)maxskip;
    const size_t nints = (size_t)nb * ((size_t)ntl + nskip);
    rc = ensure_dev(&ws->d_list, &ws->d_list_cap, nints * sizeof(int));
    if (rc) return rc;
    if (ws->h_list_cap < nints * sizeof(int)) {
        if (ws->h_list) HIPCHK(hipHostFree(ws->h_list));
        HIPCHK(hipHostMalloc((void **)&ws->h_list, nints * sizeof(int), hipHostMallocDefault));
        ws->h_list_cap = nints * sizeof(int);
    }
    int *hl = ws->h_list, *hs = ws->h_list + (size_t)nb * ntl;
    for (int64_t m = 0; m < nb; m++) {
        std::vector<int> &am = act[(size_t)m];
        if (pl.seam && !fixedRB) {
            // odd-xc periodic seam: the edge strips' tiles (two passes per half-sweep) are dispatched first, spread over
            // the XCDs (xinv_heavy_first; the kernels do the same arithmetic when there is no list)
            auto heavy = [&](int id) { const int s_ = id % nstrip; return s_ == 0 || s_ == nstrip - 1; };
            const int nh = (int)(std::stable_partition(am.begin(), am.end(), heavy) - am.begin());
            const int nwg = ntl / tpw, q = nwg >> 3, rem = nwg & 7;
            for (int L = 0; L < nwg; L++) {
                const int xcd = L & 7, T = xcd * q + std::min(xcd, rem) + (L >> 3);
                const int sq = xinv_heavy_first(L, nwg, nh / tpw);
                for (int w = 0; w < tpw; w++)
                    hl[m * ntl + T * tpw + w] = sq * tpw + w < (int)am.size() ? am[(size_t)(sq * tpw + w)] : -1;
            }
        } else
        for (int t = 0; t < ntl; t++) hl[m * ntl + t] = t < (int)am.size() ? am[(size_t)t] : -1;
        for (int t = 0; t < nskip; t++) hs[m * nskip + t] = t < (int)skp[(size_t)m].size() ? skp[(size_t)m][t] : -1;
    }
    HIPCHK(hipMemcpyAsync(ws->d_list, ws->h_list, nints * sizeof(int), hipMemcpyHostToDevice, st));
    const size_t nt = (size_t)nb * nskip;
    const size_t tsum_bytes = (nt + (size_t)nb) * (sizeof(double) + sizeof(long long));
    rc = ensure_dev(&ws->d_tsum, &ws->d_tsum_cap, tsum_bytes + (size_t)nb * sizeof(unsigned));
    if (rc) return rc;
    HIPCHK(hipMemsetAsync((char *)ws->d_tsum + tsum_bytes, 0, (size_t)nb * sizeof(unsigned), st));   // k_skip_tiles' tickets
    SkipNormArgs na;
    na.S = p.S; na.sS = p.sS; na.yc = yc; na.xc = p.xc; na.nstrip = nstrip; na.nrb = best; na.UW = UW; na.RB = fixedRB;
    na.undef = p.sc_.undef; na.skip_list = ws->d_list + (size_t)nb * ntl; na.nskip_max = nskip;
    char *base = (char *)ws->d_tsum;
    na.tsum = (double *)base;
    na.tcnt = (long long *)(base + nt * sizeof(double));
    na.xsum = (double *)(base + nt * (sizeof(double) + sizeof(long long)));
    na.xcnt = (long long *)(base + nt * (sizeof(double) + sizeof(long long)) + (size_t)nb * sizeof(double));
    na.ticket = (unsigned *)(base + tsum_bytes);
    pl.skipna = na;                                      // (the skipped tiles' norm share and their copies: run_sweeps)

    pl.skip = true; pl.ntl = ntl; pl.nskip = nskip;
    pl.skip_pct = (int)((100 * nskipped) / (ntiles * nb));
    pl.skip_ppm = (int)((1000000 * nskipped) / (ntiles * nb));
    if (!fixedRB) {
        pl.nrb = best; pl.even_split = true; pl.RY = (int)cdiv(yc, best);
        pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, 128 - (pl.nine ? 8 : 4) * XINV_KMAX - (pl.seam ? 4 : 0)) * pl.nrb, pl.pipe ? 4 : tpw) + 1;
        if (pl.pipe) pl.nsg = std::max(pl.nsg, (int)cdiv(p.xc, UW) * pl.nrb + 1);
    }
    return XINV_OK;
}

// Which of `nstream` arrays have rows (of xc elements, `rows` per member) that are bitwise constant
// along x?  One pass over each array on the device; *mask gets bit q set for uniform array q.
static int detect_xuniform(Workspace *ws, hipStream_t st, const double *const *arr,
                           const int64_t *stride, int nstream, int64_t nbatch, int64_t rows,
                           int64_t xc, unsigned *mask)
{
    XUniArgs xa;
    memset(&xa, 0, sizeof xa);
    xa.nstream = nstream;
    for (int q = 0; q < nstream; q++) { xa.c[q] = arr[q]; xa.stride[q] = stride[q]; }
    xa.nbatch = nbatch; xa.yc = rows; xa.xc = xc;
    if (!ws->dflags16) {
        HIPCHK(hipMalloc((void **)&ws->dflags16, 16 * sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&ws->hflags16, 16 * sizeof(int), hipHostMallocDefault));
    }
    xa.flag = ws->dflags16;
    HIPCHK(hipMemsetAsync(ws->dflags16, 0, 16 * sizeof(int), st));
    hipLaunchKernelGGL(k_xuniform, dim3(512, (unsigned)nstream, 1), dim3(256), 0, st, xa);
    HIPCHK(hipMemcpyAsync(ws->hflags16, ws->dflags16, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ws->act_synced = true;                                // (an activity map issued ahead on this stream has landed too)
    *mask = 0;
    for (int q = 0; q < nstream; q++) if (!ws->hflags16[q]) *mask |= (1u << q);
    return XINV_OK;
}

static thread_local unsigned t_detected_um = 0;       // coefficient arrays (bit = coefficient index) the calling thread's last
                                                      // plan found constant along x (or knew to be): solve_host_one hands the
                                                      // shared ones to the later chunks of the same call
// The same over coefficient arrays idx[0..ns) of a problem; bit q of *mask = array idx[q].  Arrays the caller declared
// row-constant to a resident plan (Problem::known_um: the plan expanded them itself) are not read again.
static int detect_xuniform_of(const Problem &p, Workspace *ws, hipStream_t st, const int *idx, int ns, int64_t rows,
                              unsigned *mask)
{
    const double *arr[10]; int64_t strd[10]; int pos[10];
    int nu = 0;
    unsigned m = 0;
    for (int q = 0; q < ns; q++) {
        if ((p.known_um >> idx[q]) & 1u) { m |= 1u << q; continue; }
        arr[nu] = p.c[idx[q]]; strd[nu] = p.sc[idx[q]]; pos[nu] = q; nu++;
    }
    if (nu) {
        unsigned mm = 0;
        const int rc = detect_xuniform(ws, st, arr, strd, nu, p.nbatch, rows, p.xc, &mm);
        if (rc) return rc;
        for (int k = 0; k < nu; k++) if ((mm >> k) & 1u) m |= 1u << pos[k];
    }
    for (int q = 0; q < ns; q++) if ((m >> q) & 1u) t_detected_um |= 1u << idx[q];
    *mask = m;
    return XINV_OK;
}
