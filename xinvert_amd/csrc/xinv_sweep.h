// xinv_sweep.h -- the sweep loop of libxinv_hip.so: launch chains with pipelined polling of the device-side stop
// flags, lanes, the lagged norm and its three-buffer rotation, watchdog recovery, finalise() (where each member's
// final state lives; the redo of a pass the stop rule fired in), the device-pointer solve and the resident plans
// (xinv_plan_*).  Included by xinv_hip.hip only, after xinv_plan.h.
#pragma once

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
// bytes of the norm partials of one solve (every member, every tiling its launches may use)
static size_t partial_bytes(const Problem &p, const Plan &pl)
{
    if (pl.path == XINV_PATH_FUSED)
        return (size_t)p.nbatch * XINV_KMAX *
               (is3d(p.kind) ? std::max((size_t)pl.nsg * pl.nrb * std::max(1, pl.nkc),
                                        pl.K2 ? (size_t)pl.nsg2 * pl.nrb2 * std::max(1, pl.nkc2) : (size_t)0)
                             : (size_t)pl.nsg) *
               (3 * sizeof(unsigned long long));        // three tagged words per partial
    return (size_t)p.nbatch * XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long));
}

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
    size_t pbytes = partial_bytes(p, pl);
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


// ------------------------------------------------------------------ several restarts of a plan without a host round trip
// apps.animate_iteration (reference apps.py:1031-1044; tests/test_AnimateConverge.py:13-31: 40 frames of 2 sweeps on 73 x
// 144) calls its kernel once per frame, continuing from the previous frame's S.  Through xinv_plan_solve_f64_dev a frame is
// one C-ABI call: control blocks reset, one or two launches, the control blocks back to the host (k_ctl_mail), 36-38 us of
// which 18 are the launch.  Here nframes restarts are QUEUED behind each other: every frame has its own control blocks,
// the state ping-pongs between S and its twin ACROSS the frames (no copy back per frame), a snapshot of the state goes into
// the caller's frame buffer behind each frame, and the host reads all the control blocks once, at the end.  What the host
// cannot do without looking -- redo a pass the stop rule fired INSIDE of (finalise) -- does not happen while every frame runs
// its whole budget; if some frame stopped early (the field has converged) the state before the first such frame is
// restored from the frame buffer and the remaining frames take the ordinary road, one xinv_plan_solve each.
// One launch chain, the launch's own last workgroup reduces the norm (no lagged evaluation: its three-buffer rotation is
// what the per-frame bookkeeping was for).  Bit for bit nframes calls of xinv_plan_solve_f64_dev.
static int plan_solve_frames(xinv_plan *h, double *S, double *frames, int64_t nframes, int64_t frame_stride, double *flags,
                             int64_t mxLoop, double tolerance, hipStream_t st)
{
    if (!h || h->magic != XINV_PLAN_MAGIC) return fail_arg("xinv_plan_solve_frames: not a live plan");
    if (nframes < 1 || !frames || !flags) return fail_arg("xinv_plan_solve_frames: no frames");
    Problem p = h->p;
    p.S = S;
    p.stop.mxLoop = mxLoop; p.stop.tolerance = tolerance;
    int rc = validate(p, flags);
    if (rc) return rc;
    const int64_t n = p.zc * p.yc * p.xc, nb = p.nbatch;
    const int64_t span = (nb - 1) * p.sS + n;            // elements of S
    if (frame_stride < span) return fail_arg("xinv_plan_solve_frames: frame stride smaller than S");
    const int64_t max_sweeps = mxLoop + 1;
    auto slow_from = [&](int64_t f0) -> int {             // frames f0 .. nframes-1 the ordinary way (S holds the state before f0)
        for (int64_t f = f0; f < nframes; f++) {
            double *fl = flags + 3 * nb * f;
            for (int64_t m = 0; m < nb; m++) { fl[3 * m] = 0.0; fl[3 * m + 1] = 1.0; fl[3 * m + 2] = 0.0; }
            int r = plan_solve(h, S, fl, mxLoop, tolerance, st);
            if (r) return r;
            HIPCHK(hipMemcpyAsync(frames + f * frame_stride, S, (size_t)span * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipStreamSynchronize(st));
        return XINV_OK;
    };
    // the queued road: the streaming path, a batch small enough for one chain; anything else frame by frame
    if (h->pl.path != XINV_PATH_FUSED || nb > 64 || nframes * nb > 65536 || (h->pl.aligned && !ptr_al16(S)))
        return slow_from(0);
    DeviceGuard dg;
    HIPCHK(dg.select(h->device));
    Workspace *ws = get_ws(h->device);
    std::unique_lock<std::recursive_mutex> lock(ws->busy);
    rc = ws_ready(ws);
    if (rc) return rc;
    memset(&t_stats, 0, sizeof t_stats);
    const Plan &pl0 = h->pl;
    const int Kf = pl0.K;
    const int64_t L = (max_sweeps + Kf - 1) / Kf;
    XinvCtl *ctl_all = nullptr;
    std::vector<XinvCtl> hc((size_t)(nframes * nb));
    int64_t bad = -1;
    {
        BufSwap sw(ws, &h->bufs);
        Plan pl = pl0;
        pl.skipna.S = S;
        rc = tail_wait(ws, st);
        if (rc) return rc;
        // control blocks of EVERY frame (the workspace's block is theirs for the duration), partials, the twin of S
        if ((rc = ensure_dev(&ws->ctl, &ws->ctl_cap, (size_t)(nframes * nb) * sizeof(XinvCtl)))) return rc;
        ctl_all = ws->ctl;
        const size_t pbytes = (partial_bytes(p, pl) + 255) & ~(size_t)255;
        ws->partials_half = pbytes;
        if ((rc = ensure_dev(&ws->partials, &ws->partials_cap, pbytes))) return rc;
        if ((rc = ensure_dev(&ws->S2, &ws->S2_cap, (size_t)span * sizeof(double)))) return rc;
        if ((rc = ensure_dev(&ws->S3, &ws->S3_cap, (size_t)span * sizeof(double)))) return rc;
        HIPCHK(hipMemcpyAsync(ws->S3, S, (size_t)span * sizeof(double), hipMemcpyDeviceToDevice, st));   // (the state before frame 0)
        struct CtlRestore { Workspace *w; XinvCtl *base; ~CtlRestore() { w->ctl = base; } } restore{ws, ctl_all};
        if (pl.skip) {                                   // the skipped tiles' share of the norm and their copy into the twin: once
            hipLaunchKernelGGL(k_skip_tiles, dim3((unsigned)pl.nskip, (unsigned)nb, 1), dim3(64), 0, st, pl.skipna, ws->S2,
                               (double *)nullptr);
            HIPCHK(hipGetLastError());
        }
        double *buf[2] = { S, ws->S2 };
        int cur = 0;
        int64_t nlaunch = 0;
        for (int64_t f = 0; f < nframes; f++) {
            ws->ctl = ctl_all + f * nb;                  // (the launchers take the control blocks from the workspace)
            hipLaunchKernelGGL(k_solve_init, dim3((unsigned)std::max<int64_t>(cdiv(nb, 256), std::min<int64_t>(256, cdiv((int64_t)(pbytes / 16), 256)))),
                               dim3(256), 0, st, ws->ctl, nb, (uint4 *)ws->partials, (int64_t)(pbytes / 16));
            for (int64_t i = 0; i < L; i++) {
                const int k = (int)std::min<int64_t>(Kf, max_sweeps - i * Kf);
                rc = launch_planned(p, pl, ws, st, k, buf[cur], buf[cur ^ 1], 0, nb, 0, 0);
                if (rc) return rc;
                cur ^= 1; nlaunch++;
            }
            HIPCHK(hipMemcpyAsync(frames + f * frame_stride, buf[cur], (size_t)span * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        ws->ctl = ctl_all;
        HIPCHK(hipMemcpyAsync(hc.data(), ctl_all, (size_t)(nframes * nb) * sizeof(XinvCtl), hipMemcpyDeviceToHost, st));
        if (cur != 0) HIPCHK(hipMemcpyAsync(S, ws->S2, (size_t)span * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        // every frame must have run its whole budget (or have stopped exactly at the end of its last pass)
        for (int64_t f = 0; f < nframes && bad < 0; f++)
            for (int64_t m = 0; m < nb; m++) {
                const XinvCtl &c = hc[(size_t)(f * nb + m)];
                if (!c.done || c.overflow == 2 || c.sweeps != max_sweeps) { bad = f; break; }
            }
        const int64_t good = bad < 0 ? nframes : bad;
        for (int64_t f = 0; f < good; f++)
            for (int64_t m = 0; m < nb; m++) {
                const XinvCtl &c = hc[(size_t)(f * nb + m)];
                double *fl = flags + 3 * (nb * f + m);
                fl[0] = c.overflow ? 1.0 : 0.0; fl[1] = 1.0; fl[2] = 0.0;
                if (c.wrote) { fl[1] = c.flag1; fl[2] = c.flag2; }
            }
        t_stats.path = pl.path; t_stats.colours = pl.ncol; t_stats.sweeps_per_launch = Kf; t_stats.rows_per_tile = pl.RY;
        t_stats.xuniform_mask = (int32_t)pl.um; t_stats.lanes = 1; t_stats.sweep_launches = nlaunch; t_stats.planned = 1;
        t_stats.sweeps_max = max_sweeps; t_stats.pipelined = pl.pipe ? pl.npair : 0;
        t_stats.masked_tile_pct = pl.skip ? pl.skip_pct : 0; t_stats.masked_tile_ppm = pl.skip ? pl.skip_ppm : 0;
        h->solves += good;
        if (bad >= 0)                                    // the state before the first frame that stopped early, back into S
            HIPCHK(hipMemcpyAsync(S, bad > 0 ? frames + (bad - 1) * frame_stride : ws->S3, (size_t)span * sizeof(double),
                                  hipMemcpyDeviceToDevice, st));
    }
    if (bad < 0) return XINV_OK;
    lock.unlock();
    return slow_from(bad);
}
