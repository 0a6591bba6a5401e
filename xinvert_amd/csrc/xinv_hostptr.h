// xinv_hostptr.h -- the host-pointer entries of libxinv_hip.so: upload -> solve -> download pipelined over member
// chunks (uploader / downloader threads through the library's pinned staging rings, two chunk solves in flight),
// and the in-call split of the batch axis over several GPUs (one host thread per device, bound to the GPU's NUMA
// node).  Included by xinv_hip.hip only, after xinv_sweep.h.
#pragma once

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
    if (opt.host_chunk < 0) {                              // -k: a ramp 1, 2, 4, .. up to k members, then chunks of k
        const int64_t kmax = -(int64_t)opt.host_chunk;
        int64_t left = nb, c = 1;
        while (left > 0) { const int64_t t = std::min(std::min(c, kmax), left); out.push_back(t); left -= t; c *= 2; }
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
    if (is3d(p.kind)) {
        if (total < 100663296.0 || nb < 4) { out.push_back(nb); return out; }
        // Round 6 (profiles/r06_host_pipeline.txt; C5 x 15, ms): chunks of three volumes, THREE chunk solves in flight, each
        // one launch chain (no lanes inside a chunk): 148; chunks of two, two in flight -- round 5 -- 160-167; three
        // in flight 155; chunks of four 154-168; ramps 1, 2, 4, .. 163-173.  The remainder LAST.
        const int64_t per = nb >= 6 ? 3 : 2;
        for (int64_t m0 = 0; m0 < nb; m0 += per) out.push_back(std::min(per, nb - m0));
        return out;
    }
    // 2-D forms, end of round 6 (TWO chunk solves in flight, the copy streams at high priority; wall ms of 500 sweeps,
    // profiles/r06_host_pipeline.txt): 1440 x 720 general form (17 MB a member) x 3 members -- chunks of 1: 8.5, one chunk 7.5;
    // x 6 -- 1: 13.3, 2: 12.0, 3: 12.9; x 8 -- 1: 17.2, 2: 14.0, 4: 15.9; x 12 -- 2: 19.0, 3: 22.4, 4: 20.2, 6: 19.3; x 16 --
    // 2: 23.9, 4: 26.2, 8: 23.6; x 32 -- 2: 42.8, 4: 49.0, 8: 41.5, 16: 40.6; x 64 -- 8: 82.0, 16: 71.7, 32: 72.8.  3600 x 1800
    // standard form (104 MB a member) x 2 -- 1: 16.7, one chunk 18.1; x 3 -- 1: 20.5, one chunk 25.1; x 8 -- 1: 43.7, 2: 45.8,
    // 4: 48.4.  Odd chunks lose (their two lanes are uneven).  Large members travel one by one; small ones in pairs up to 16
    // members, in four chunks beyond.  (FOUR chunk solves in flight are faster in a fresh process -- C4 x 8 13.1 against
    // 13.7, x 16 21.6 --, but in a process that has run resident solves before -- bench.py -- every second process lands on
    // 15-16 ms: the chains then share the runtime's hardware queues unevenly; two in flight give 12.8-13.1 every time.)
    if (total < 50331648.0) { out.push_back(nb); return out; }
    int64_t per;
    if (member_bytes >= 67108864.0) per = 1;
    else if (nb < 4) { out.push_back(nb); return out; }
    else per = nb <= 16 ? 2 : ((std::max<int64_t>(2, nb / 4) + 1) & ~(int64_t)1);
    for (int64_t m0 = 0; m0 < nb; m0 += per) out.push_back(std::min(per, nb - m0));
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
    std::thread up, down;
    std::vector<std::thread> solvers;                 // helper threads: chunk solves in flight beside the calling thread's
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> chunk_ready;                    // set by the uploader once chunk c's event is recorded
    std::deque<std::function<int()>> dq;              // download jobs
    bool d_closed = false, abort = false;
    int u_rc = 0, d_rc = 0, s2_rc = 0;                // (s2: the first failing helper solver thread's verdict -- kept here: this
    std::string u_err, d_err, s2_err;                 //  object outlives the threads on every return path)
    std::vector<hipStream_t> streams;
    void close_downloads() { { std::lock_guard<std::mutex> lk(mu); d_closed = true; } cv.notify_all(); }
    ~HostActors()
    {
        { std::lock_guard<std::mutex> lk(mu); abort = true; d_closed = true; }
        cv.notify_all();
        for (auto &t : solvers) if (t.joinable()) t.join();      // (before the downloader: they still queue download jobs)
        if (up.joinable()) up.join();
        if (down.joinable()) down.join();
        for (hipStream_t s : streams) (void)hipStreamSynchronize(s);      // nothing of this call stays in flight
    }
};

static int solve_host_one(Problem &p, double *flags, const xinv_options &opt, const Pinned *outer)
{
    const bool may_register = (outer == nullptr);     // a per-device call of a multi-device solve uses the parent's registrations
    const auto wall0 = std::chrono::steady_clock::now();
    // XINV_HOST_TRACE=1: host-side time stamps of the call's phases on stderr (ms since entry; diagnosis only)
    static const bool trace_on = XINV_ENV_INT("XINV_HOST_TRACE", 0) != 0;
    auto trace = [&](const char *what, long long k = -1) {
        if (!trace_on) return;
        const double ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        if (k < 0) fprintf(stderr, "[xinv host %8.3f ms] %s\n", ms_, what);
        else fprintf(stderr, "[xinv host %8.3f ms] %s #%lld\n", ms_, what, k);
    };
    DeviceGuard dg;
    HIPCHK(dg.select(opt.device));
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    const int64_t n = p.zc * p.yc * p.xc;
    // the staging rings, the device pool and the solver workspace are per device: hold the device for the whole
    // upload -> solve -> download sequence
    Workspace *ws = get_ws(device);
    std::lock_guard<std::recursive_mutex> host_lock(ws->busy);
    // The copy streams take the highest stream priority: the runtime keeps its hardware queues per priority, so the staged
    // copies (blit kernels on this runtime) no longer queue behind a chunk solve's launch chain that happens to share
    // their hardware queue -- C4 x 8 with four chunk solves in flight: 14.0 -> 12.8 ms, uploads no longer stretched to
    // 8 ms (profiles/r06_host_pipeline.txt; XINV_COPY_PRIO=0 in a hooks build: the round-5 streams).
    for (hipStream_t *sp : { &ws->s_up, &ws->s_down, &ws->s_compute })
        if (!*sp) {
            if (sp != &ws->s_compute && XINV_ENV_INT("XINV_COPY_PRIO", 1)) {
                int lo_ = 0, hi_ = 0;
                if (hipDeviceGetStreamPriorityRange(&lo_, &hi_) != hipSuccess || hipStreamCreateWithPriority(sp, hipStreamNonBlocking, hi_) != hipSuccess) {
                    (void)hipGetLastError();             // (no priorities on this device / runtime: a plain stream)
                    *sp = nullptr;
                    HIPCHK(hipStreamCreateWithFlags(sp, hipStreamNonBlocking));
                }
            } else
                HIPCHK(hipStreamCreateWithFlags(sp, hipStreamNonBlocking));
        }
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
    // The rolling batch (round 6; roll_3d below): ONE chain of launches over the members that have arrived and are not done
    // yet, instead of one solve per chunk.  For the standard 3-D form with shared coefficient arrays (its plan reads nothing
    // of a member's own), on the streaming path, where the planner is left to itself.
    // (xinv_options.host_inflight = -1 takes it for any batch of two or more: the tests' small volumes)
    const bool roll2d = (p.kind == KIND_STD2D || p.kind == KIND_GEN2D);     // (2-D: only where no tile is fully masked, see roll)
    bool rolling = (p.kind == KIND_STD3D || roll2d) && (opt.host_chunk == 0 || opt.host_inflight < 0) && opt.host_inflight <= 0 &&
                   opt.path != XINV_PATH_COLOUR && !(p.BCx == XINV_BC_PERIODIC && (p.xc & 1) && p.xc < 64) &&
                   ((!roll2d && p.nbatch >= 4 && (double)n * 16.0 * (double)p.nbatch >= 100663296.0) || (opt.host_inflight < 0 && p.nbatch >= 2));
    // (2-D forms roll on request only -- host_inflight = -1 --: C4 x 8, 500 sweeps: 15.0 ms rolling against 13.4 in chunks of
    //  two members, two chunk solves in flight (7.6 resident).  The 2-D tiling is chosen for the whole batch -- 240 workgroups
    //  per member where the chip holds a thousand --, so a launch over two members of eight takes two thirds of the time of
    //  one over all eight, there are no lanes, and the two plans cost a millisecond: profiles/r06_host_pipeline.txt)
    for (int q = 0; q + 1 < p.ncoef; q++) rolling = rolling && (p.c[q] ? (p.nbatch == 1 || p.sc[q] == 0) : (roll2d && q == 1));
    // (3-D: a volume per upload event; 2-D: the chunk scheme's chunks -- it takes over when the forcing has masked tiles)
    const std::vector<int64_t> chunks = (rolling && !roll2d) ? std::vector<int64_t>((size_t)p.nbatch, 1) : host_chunks(p, opt);
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

    trace("set up: device buffers, ops queued for the uploader");
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
        trace("uploader: shared arrays queued");
        for (int64_t c = 0; c < nchunk; c++) {
            if (!r) run(chunk_ops[(size_t)c]);
            trace("uploader: chunk queued", c);
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
            trace("downloader: job queued / staged");
        }
        if (!r && first_job && hipEventRecord(e_dn0, sdn) != hipSuccess) r = XINV_ERR_HIP;
        if (!r && hipEventRecord(e_dn1, sdn) != hipSuccess) r = XINV_ERR_HIP;
        if (!r && hipStreamSynchronize(sdn) != hipSuccess) r = XINV_ERR_HIP;
        trace("downloader: drained");
        std::lock_guard<std::mutex> lk(act.mu);
        act.d_rc = r; if (r) act.d_err = t_err;
    });

    // ---- solve chunk by chunk; downloads trail on their own thread ------------------------------
    // Two chunk solves are in flight at a time (round 5): the even chunks on the calling thread (the device's workspace),
    // the odd ones on a helper thread with a workspace and a compute stream of its own.  A chunk fills the 256 CUs less
    // evenly than the whole batch -- the 3-D kernels run ceil(workgroups / 256) rounds, every 2-D launch ends with a
    // tail --; with the next chunk's launches already queued on the device those holes are filled, as the two launch
    // chains of a device-resident batch fill each other's (the lanes of run_sweeps).
    // How many chunk solves are in flight: every one is a chain of dependent launches, and a launch of a two-volume chunk
    // (276 tiles on 256 CUs) ends in a tail during which only the OTHER chains' launches keep the CUs busy.  Two chains
    // (round 5) left the chip 1.68 launches deep on average -- C5 x 15: 160 ms of which the GPU is busy 160, at 107 us per
    // volume and launch against 77 for the resident batch (profiles/r06_host_pipeline.txt) --; three / four fill the tails.
    // 2-D forms: two.  (Four -- possible since the copy streams have queues of their own; before, a third and fourth chain
    // stretched the uploads behind them to 8 ms -- are faster in a fresh process, C4 x 8 14.4 -> 13.3 ms, 3600 x 1800 x 8
    // 47.6 -> 43.6, and bimodal in one that has run resident solves before: bench.py's end_to_end leg 12.3-12.9 or 14.7-16.1
    // ms, process by process, against 12.8-13.1 with two; host_inflight = 4 asks for them.)  What is left against the
    // resident batch (C4 x 8, 2000 sweeps: 37.1 ms against 28.4 + 5 of copies) is kernel time of small launches: four
    // chains of two-member launches run 8.7 us per member and pass, the resident batch's two lanes of four 7.2
    // (tools/kernel_groups.sh; GPU_MAX_HW_QUEUES=8 makes the resident two-lane solve itself 7.6 -> 11.7 ms).
    const int ninfl = (int)std::min<int64_t>(nchunk, std::max(1, opt.host_inflight > 0 ? std::min(opt.host_inflight, XINV_MAX_INFLIGHT)
                                                                                         : (is3d(p.kind) ? 3 : XINV_ENV_INT("XINV_INFLIGHT_2D", XINV_DEFAULT_INFLIGHT))));
    std::vector<Workspace *> wss((size_t)ninfl, nullptr);
    std::vector<hipStream_t> scps((size_t)ninfl, nullptr);
    wss[0] = ws; scps[0] = scp;
    for (int k = 1; k < ninfl; k++) {
        wss[(size_t)k] = get_ws(device, k);
        if (!wss[(size_t)k]->s_compute) HIPCHK(hipStreamCreateWithFlags(&wss[(size_t)k]->s_compute, hipStreamNonBlocking));
        scps[(size_t)k] = wss[(size_t)k]->s_compute;
        act.streams.push_back(scps[(size_t)k]);
    }
    // the workspaces grow on demand: size them for the LARGEST chunk now, so that a later, larger chunk does not pay a
    // free + malloc of the ping-pong buffer (or of the pinned control-block mirror) mid-pipeline
    {
        const int64_t mmax = *std::max_element(chunks.begin(), chunks.end());
        if (nchunk > 1 && p.kind != KIND_BIH2D)
            for (Workspace *w : wss) {
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
    if (is3d(p.kind) && nchunk > 1 && o1.lanes == 0) o1.lanes = 1;      // (several chunk solves in flight already: one chain each)
    // members [m0, m0 + nm): S is final on the device in stream order of `cs` -- the output passes (de-mask, float32), then
    // the hand-over to the downloader.  `after`: an event to record behind them for the downloader to wait on (the rolling
    // batch: no host synchronisation of the compute stream); nullptr: synchronise `cs` here.
    auto finish_members = [&](int64_t m0, int64_t nm, hipStream_t cs, hipEvent_t after) -> int {
        if (opt.prep_flags & XINV_PREP_DEMASK) {
            for (int64_t m = 0; m < nm; m++) {
                const double *dF = d.c[fq] + (per_member[fq] ? (m0 + m) * n : 0);
                hipLaunchKernelGGL(k_demask, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, cs,
                                   d.S + (m0 + m) * n, dF, n, p.sc_.undef, opt.demask_value);
            }
        }
        if (tmpS_dn)                                     // float32 S: rounded on the device, half the bytes back
            hipLaunchKernelGGL(k_demote_f64, dim3((unsigned)std::min<int64_t>(4096, (nm * n + 255) / 256)), dim3(256), 0, cs,
                               (const double *)(d.S + m0 * n), tmpS_dn + m0 * n, nm * n);
        if (after) HIPCHK(hipEventRecord(after, cs));
        else if ((opt.prep_flags & XINV_PREP_DEMASK) || tmpS_dn) HIPCHK(hipStreamSynchronize(cs));
        {
            char *hS = (char *)p.S;
            const char *dS = tmpS_dn ? (const char *)tmpS_dn : (const char *)d.S;
            const size_t es = tmpS_dn ? 4 : 8;
            const Pinned *pinp = &pin;
            std::lock_guard<std::mutex> lk(act.mu);
            act.dq.push_back([=]() -> int {
                if (after) HIPCHK(hipStreamWaitEvent(sdn, after, 0));
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
        trace("solver: chunk's upload queued, solve starts", c);
        Problem dc = d;
        { std::lock_guard<std::mutex> lk(act.mu); dc.known_um |= shared_um; }     // (what an earlier chunk's plan found out)
        dc.nbatch = nm;
        dc.S = d.S + m0 * n;
        for (int q = 0; q < p.ncoef; q++)
            if (d.c[q] && d.sc[q] != 0) dc.c[q] = d.c[q] + m0 * d.sc[q];
        int r = solve_dev(dc, flags + 3 * m0, &o1, cs, slot);
        if (r) return r;
        if (trace_on) { char b_[96]; snprintf(b_, sizeof b_, "solver: chunk solved (plan %.3f ms, sweeps %.3f ms)", t_stats.plan_ms, t_stats.sweep_ms); trace(b_, c); }
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
        return finish_members(m0, nm, cs, nullptr);
    };
    // ---- the rolling batch (standard 3-D form, shared coefficients) ---------------------------------------------------
    // Every chunk solve above is a chain of small launches -- a two-volume launch of k_pipe3d is 276 tiles on 256 CUs -- and
    // with two or three chains in flight the chip still ran at 107 us per volume and launch against 77 for the resident
    // batch (profiles/r06_host_pipeline.txt).  Here ONE chain of launches sweeps the members [lo, hi) that have arrived and
    // still have sweeps to do: a volume joins at the next even launch after its upload event (its S sits in buffer 0, the
    // launches ping-pong), runs its L = ceil(sweeps / K) launches -- the device-side stop rule counts its sweeps, whatever
    // the launch index -- and retires (FIFO: every member runs the same budget; a member the tolerance stopped earlier
    // idles through its remaining launches as a no-op).  Members are independent (reference core.py:129), so what a
    // launch covers cannot change any result.  The host stays two launches ahead of the GPU, so that a join is decided
    // when the launch is about to run; a retired member's control block travels behind its last launch, its final state
    // is put into S as finalise() does (the redo of a pass the stop rule fired in: from that pass's source, intact since),
    // and the downloader takes it from there behind an event.
    constexpr int ROLL_FALLBACK = 0x7fff0001;            // (not an error: the chunk scheme takes the call)
    auto roll = [&]() -> int {
        const int64_t nb = p.nbatch;
        {   // the plan needs the shared coefficient arrays: they travel ahead of member 0
            std::unique_lock<std::mutex> lk(act.mu);
            act.cv.wait(lk, [&] { return act.chunk_ready[0] != 0 || act.abort; });
            if (act.u_rc) { t_err = act.u_err; return act.u_rc; }
            if (act.abort) { t_err = "host-pointer solve aborted"; return XINV_ERR_HIP; }
        }
        HIPCHK(hipStreamWaitEvent(scp, e_chunk[0], 0));
        int r = ws_ready(ws);
        if (r) return r;
        memset(&t_stats, 0, sizeof t_stats);
        const auto t_plan0 = std::chrono::steady_clock::now();
        Plan pl;
        xinv_options oroll = o1;
        if (roll2d) {
            // 2-D: the plan of a batch reads every member's forcing (the lists of fully masked tiles) -- the rolling batch plans
            // before they have arrived.  The first chunk is planned alone: if IT has masked tiles to skip, the chunk scheme
            // takes the call; else the batch is planned without tile lists (a later member's masked tiles are swept like any
            // other: the update leaves masked points alone, the result is the same).
            Problem d1 = d;
            d1.nbatch = chunks[0];
            Plan pa;
            r = make_plan(d1, o1, ws, scp, pa);
            if (r) return r;
            // (the point-factor stream of the general form with coefficients that vary along x folds the forcing's mask into the
            //  factors: it reads every member's forcing too)
            if (pa.path != XINV_PATH_FUSED || pa.skip || pa.pq) return ROLL_FALLBACK;
            for (int q = 0; q < p.ncoef; q++)                // (what that plan found constant along x is not tested again)
                if (d.c[q] && d.sc[q] == 0 && ((t_detected_um >> q) & 1u)) d.known_um |= 1u << q;
            oroll.flags |= XINV_FLAG_NO_TILE_SKIP;
        }
        r = make_plan(d, oroll, ws, scp, pl);
        if (r) return r;
        if (pl.path != XINV_PATH_FUSED || pl.skip || pl.pq) return roll2d ? ROLL_FALLBACK : (t_err = "internal: rolling batch without a streaming kernel", XINV_ERR_HIP);
        const double plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
        // workspace (run_sweeps' own, without the lagged norm: never in 3-D)
        r = tail_wait(ws, scp);
        if (r) return r;
        if ((r = ensure_dev(&ws->ctl, &ws->ctl_cap, (size_t)nb * sizeof(XinvCtl)))) return r;
        if (ws->hctl_cap < (size_t)nb) {
            if (ws->hctl) HIPCHK(hipHostFree(ws->hctl));
            ws->hctl = nullptr; ws->hctl_cap = 0;
            HIPCHK(hipHostMalloc((void **)&ws->hctl, 2 * (size_t)nb * sizeof(XinvCtl), XINV_HOST_COHERENT));
            ws->hctl_cap = (size_t)nb;
        }
        const size_t pbytes = (partial_bytes(d, pl) + 255) & ~(size_t)255;
        ws->partials_half = pbytes;
        if ((r = ensure_dev(&ws->partials, &ws->partials_cap, pbytes))) return r;
        if ((r = ensure_dev(&ws->S2, &ws->S2_cap, (size_t)nb * n * sizeof(double)))) return r;
        hipLaunchKernelGGL(k_solve_init, dim3((unsigned)std::max<int64_t>(cdiv(nb, 256), std::min<int64_t>(256, cdiv((int64_t)(pbytes / 16), 256)))),
                           dim3(256), 0, scp, ws->ctl, nb, (uint4 *)ws->partials, (int64_t)(pbytes / 16));
        double *buf[2] = { d.S, ws->S2 };
        const int64_t max_sweeps = d.stop.mxLoop + 1;
        const int Kf = pl.K;
        const int64_t L = (max_sweeps + Kf - 1) / Kf;               // launches of a member
        const int klast = (int)(max_sweeps - (L - 1) * Kf);           // sweeps of its last one
        std::vector<int64_t> join((size_t)nb, -1);
        struct Retired { int64_t a, b; hipEvent_t ctl_done; hipStream_t s; };
        std::deque<Retired> fin;
        // How far the host runs ahead of the GPU: far enough that a wake-up of the pacing wait (20-50 us) never starves the
        // queue -- ~400 us of launches, two at least (a 3-D launch is 0.1-1 ms, a 2-D one 30-60 us) --, not so far that a
        // member that has just arrived waits long for the next join.
        constexpr int NQ = 16;
        const double est_launch_us = std::max(20.0, (double)nb * (double)n * pl.K / (pl.pipe ? 6.0e5 : (is3d(p.kind) ? 2.5e5 : 3.0e5)) * 0.6);
        const int depth = (int)std::min(12.0, std::max(2.0, 400.0 / est_launch_us));
        const int pstep = (roll2d && est_launch_us < 100.0) ? 4 : 1;   // launches per pacing event
        const int dsteps = std::max(1, (depth + pstep - 1) / pstep);     // pacing events the host runs ahead
        hipEvent_t ev_l[NQ];
        for (int q = 0; q < NQ; q++) if ((r = ev.make(&ev_l[q], false))) return r;
        hipEvent_t ev_t0, ev_t1;
        if ((r = ev.make(&ev_t0, true)) || (r = ev.make(&ev_t1, true))) return r;
        HIPCHK(hipEventRecord(ev_t0, scp));
        XinvCtl *hc = ws->hctl;
        int64_t nlaunch = 0, sweeps_max = 0;
        // a retired group whose control blocks have arrived: final state into S, flags, output passes, download
        auto finish = [&](const Retired &g) -> int {
            for (int64_t m = g.a; m < g.b; m++) {
                const XinvCtl &c = hc[m];
                if (!c.done) { t_err = "internal: rolling batch: a member retired before its stop rule fired"; return XINV_ERR_HIP; }
                if (c.overflow == 2) { t_err = "internal: norm partials of a sweep launch never arrived (watchdog) in the rolling batch"; return XINV_ERR_HIP; }
                const int64_t sw = c.sweeps;
                const int64_t rl = (sw - 1) / Kf;                   // the member's launch that holds sweep sw (0-based)
                const int64_t lend = std::min<int64_t>((rl + 1) * Kf, max_sweeps);
                int where;
                if (sw == lend) where = (int)((join[(size_t)m] + rl + 1) & 1);
                else {                                              // stopped inside a pass: redo from its source, sweep by sweep
                    int cur = (int)((join[(size_t)m] + rl) & 1);
                    for (int64_t q = rl * Kf; q < sw; q++) {
                        int rr = launch_planned(d, pl, ws, g.s, 1, buf[cur], buf[cur ^ 1], m, 1, 1, 1);
                        if (rr) return rr;
                        cur ^= 1;
                    }
                    where = cur;
                }
                if (where != 0)
                    HIPCHK(hipMemcpyAsync(d.S + m * n, ws->S2 + m * n, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g.s));
                if (c.overflow) flags[3 * m + 0] = 1.0;
                if (c.wrote) { flags[3 * m + 1] = c.flag1; flags[3 * m + 2] = c.flag2; }
                sweeps_max = std::max<int64_t>(sweeps_max, sw);
            }
            hipEvent_t after;
            int rr = ev.make(&after, false);
            if (rr) return rr;
            return finish_members(g.a, g.b - g.a, g.s, after);
        };
        // Lanes (2-D forms): the members are cut into two halves at a chunk boundary, each half a rolling chain of its own on
        // its own stream, the two chains' launches issued alternately by this thread -- the lanes of a resident solve
        // (run_sweeps): one chain's launch boundary is covered by the other's launch.  (Chains issued by DIFFERENT host
        // threads -- the chunk scheme -- do not alternate in the runtime's hardware queues: C4 x 8, 2000 sweeps, 37 ms in
        // chunks against 28.4 resident + 5 of copies.)  The 3-D form keeps one chain: its launches fill the chip in rounds.
        struct Lane { int64_t lo, hi, cj, cend, mend; hipStream_t s; };
        const int nl = (roll2d && nchunk >= 2 && ninfl >= 2) ? 2 : 1;
        const int64_t csplit = nl == 2 ? (nchunk + 1) / 2 : nchunk;
        Lane lanes[2] = { {0, 0, 0, csplit, first[(size_t)csplit], scp},
                          {first[(size_t)csplit], first[(size_t)csplit], csplit, nchunk, nb, nl == 2 ? scps[1] : scp} };
        if (nl == 2) {                                               // the second chain starts behind the plan and k_solve_init
            hipEvent_t e_init;
            if ((r = ev.make(&e_init, false))) return r;
            HIPCHK(hipEventRecord(e_init, scp));
            HIPCHK(hipStreamWaitEvent(lanes[1].s, e_init, 0));
        }
        hipEvent_t ev_l1[NQ];
        if (nl == 2) for (int q = 0; q < NQ; q++) if ((r = ev.make(&ev_l1[q], false))) return r;
        auto active = [&]() { for (int l = 0; l < nl; l++) if (lanes[l].lo < lanes[l].mend) return true; return false; };
        for (int64_t i = 0; active(); i++) {
            if (!(i & 1)) {                                          // a join point: buffer 0 is the source of this launch
                std::unique_lock<std::mutex> lk(act.mu);
                bool idle = true;
                for (int l = 0; l < nl; l++) idle = idle && lanes[l].hi == lanes[l].lo;
                if (idle) {                                          // nobody active: wait for the next arrival (uploads come in batch order)
                    int64_t cw = -1;
                    for (int l = nl - 1; l >= 0; l--) if (lanes[l].cj < lanes[l].cend) cw = lanes[l].cj;
                    if (cw >= 0) act.cv.wait(lk, [&] { return act.chunk_ready[(size_t)cw] != 0 || act.abort; });
                }
                if (act.u_rc) { t_err = act.u_err; return act.u_rc; }
                if (act.abort) { t_err = "host-pointer solve aborted"; return XINV_ERR_HIP; }
                int64_t cn[2];
                for (int l = 0; l < nl; l++) {                       // (chunks cj .. cn-1 of the lane have arrived)
                    cn[l] = lanes[l].cj;
                    while (cn[l] < lanes[l].cend && act.chunk_ready[(size_t)cn[l]]) cn[l]++;
                }
                lk.unlock();
                for (int l = 0; l < nl; l++) {
                    Lane &ln = lanes[l];
                    for (; ln.cj < cn[l]; ln.cj++) {
                        HIPCHK(hipStreamWaitEvent(ln.s, e_chunk[(size_t)ln.cj], 0));
                        for (int64_t m = first[(size_t)ln.cj]; m < first[(size_t)ln.cj + 1]; m++) join[(size_t)m] = i;
                        ln.hi = first[(size_t)ln.cj + 1];
                    }
                }
            }
            for (int l = 0; l < nl; l++) {
                Lane &ln = lanes[l];
                const int64_t lo = ln.lo, hi = ln.hi;
                if (hi > lo) {
                    int64_t f = lo;                                  // [lo, f): their last launch (klast sweeps)
                    while (f < hi && i - join[(size_t)f] == L - 1) f++;
                    const double *src = buf[i & 1];
                    double *dst = buf[(i + 1) & 1];
                    if (klast == Kf) {                               // (a budget that is whole passes: one launch for everybody)
                        r = launch_planned(d, pl, ws, ln.s, Kf, src, dst, lo, hi - lo, 0, 0); if (r) return r; nlaunch++;
                    } else {
                        if (f > lo) { r = launch_planned(d, pl, ws, ln.s, klast, src, dst, lo, f - lo, 0, 0); if (r) return r; nlaunch++; }
                        if (hi > f) { r = launch_planned(d, pl, ws, ln.s, Kf, src, dst, f, hi - f, 0, 0); if (r) return r; nlaunch++; }
                    }
                    if (f > lo) {
                        HIPCHK(hipMemcpyAsync(hc + lo, ws->ctl + lo, (size_t)(f - lo) * sizeof(XinvCtl), hipMemcpyDeviceToHost, ln.s));
                        Retired g{lo, f, nullptr, ln.s};
                        if ((r = ev.make(&g.ctl_done, false))) return r;
                        HIPCHK(hipEventRecord(g.ctl_done, ln.s));
                        fin.push_back(g);
                        ln.lo = f;
                    }
                }
                // `depth` launches ahead of the GPU, no more.  (An event behind EVERY launch of a 2-D chain -- 30-60 us -- holds the
                //  next launch back by a few microseconds: the pacing events of those chains sit behind every fourth launch.)
                if (i % pstep == 0) {
                    hipEvent_t *evq = l ? ev_l1 : ev_l;
                    const int64_t e = i / pstep;
                    HIPCHK(hipEventRecord(evq[e % NQ], ln.s));
                    if (e >= dsteps) HIPCHK(hipEventSynchronize(evq[(e - dsteps) % NQ]));
                }
            }
            // (the retired groups of two chains do not finish in queue order: take whichever has arrived)
            for (size_t q = 0; q < fin.size(); ) {
                if (hipEventQuery(fin[q].ctl_done) == hipSuccess) {
                    r = finish(fin[q]); if (r) return r;
                    fin.erase(fin.begin() + (std::ptrdiff_t)q);
                } else q++;
            }
            (void)hipGetLastError();                                 // (hipEventQuery: hipErrorNotReady is not an error)
        }
        while (!fin.empty()) {
            HIPCHK(hipEventSynchronize(fin.front().ctl_done));
            r = finish(fin.front()); if (r) return r;
            fin.pop_front();
        }
        if (nl == 2) HIPCHK(hipStreamSynchronize(lanes[1].s));
        HIPCHK(hipEventRecord(ev_t1, scp));
        HIPCHK(hipStreamSynchronize(scp));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev_t0, ev_t1));
        t_stats.path = pl.path; t_stats.colours = pl.ncol; t_stats.sweeps_per_launch = Kf; t_stats.rows_per_tile = pl.RY;
        t_stats.xuniform_mask = (int32_t)pl.um; t_stats.lanes = nl; t_stats.sweep_launches = nlaunch;
        t_stats.sweeps_max = sweeps_max; t_stats.sweep_ms = ms; t_stats.plan_ms = plan_ms;
        t_stats.k_chunks = pl.K2 ? std::max(1, pl.nkc2) : 0;
        t_stats.pipelined = pl.pipe ? pl.npair : 0;
        t_stats.rolling = 1;
        acc = t_stats; acc_set = true;
        return XINV_OK;
    };
    bool rolled = false;
    if (rolling) {
        rc = roll();
        if (rc && rc != ROLL_FALLBACK) return rc;
        rolled = (rc == XINV_OK);
    }
    if (!rolled) {
    for (int k = 1; k < ninfl; k++)
        act.solvers.emplace_back([&, k]() {
            int r = (hipSetDevice(device) == hipSuccess) ? XINV_OK : XINV_ERR_HIP;
            for (int64_t c = k; c < nchunk && !r; c += ninfl) {
                try { r = do_chunk(c, scps[(size_t)k], k); } catch (const std::exception &e) { t_err = e.what(); r = XINV_ERR_HIP; }
            }
            if (r) { std::lock_guard<std::mutex> lk(act.mu); if (!act.s2_rc) { act.s2_rc = r; act.s2_err = t_err; } }
        });
    for (int64_t c = 0; c < nchunk; c += ninfl) {
        rc = do_chunk(c, scp, 0);
        if (rc) return rc;                               // (HostActors' destructor stops and joins the helpers)
    }
    for (auto &t : act.solvers) if (t.joinable()) t.join();
    if (act.s2_rc) { t_err = act.s2_err; return act.s2_rc; }
    }
    trace("solves done");
    act.close_downloads();
    act.up.join();
    act.down.join();
    trace("uploader and downloader joined");
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
            (void)bind_thread_to_device_node(o1.device);         // (this thread only lives for the call: nothing to undo)
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

