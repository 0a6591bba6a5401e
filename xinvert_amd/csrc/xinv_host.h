// xinv_host.h -- host-side state of libxinv_hip.so: error/stat slots of the calling thread, the
// per-device workspace, the problem description handed from the C-ABI to the solver, and the
// staging of host arrays (pinning in place, pooled device buffers).  Included by xinv_hip.hip only.
#pragma once

// ------------------------------------------------------------------ errors / thread state
static thread_local std::string t_err;
static thread_local xinv_stats t_stats;

#define HIPCHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            char b_[512];                                                              \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                              \
            t_err = b_;                                                                \
            return (e_ == hipErrorOutOfMemory) ? XINV_ERR_NOMEM : XINV_ERR_HIP;        \
        }                                                                              \
    } while (0)

static int fail_arg(const char *msg) { t_err = msg; return XINV_ERR_ARG; }

// The entry points select `opt.device` for the duration of the call and put the caller's current
// device back on every return path (the caller's torch tensors / streams live on ITS device).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t select(int dev)                         // dev < 0: keep the current device
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { prev = -1; return e; }
        if (dev >= 0 && dev != prev) {
            e = hipSetDevice(dev);
            if (e != hipSuccess) return e;
            changed = true;
        }
        return hipSuccess;
    }
    ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};

// ------------------------------------------------------------------ library-owned pinned staging
// Host <-> HBM traffic of the host-pointer entries goes through pinned buffers the LIBRARY owns (hipHostMalloc,
// a ring of slots per direction and device): a helper thread copies the caller's pageable memory into a slot
// (memcpy split over a few worker threads), queues the DMA out of the slot, and moves on to the next slot while
// the DMA runs; downloads mirror it.  The caller's memory is never registered (round 2: registering and
// unregistering caller ranges left the runtime in a state that aborted a LATER pageable copy of another array,
// profiles/r02_pinning_abort.txt), the thread that drives the solves is never blocked by a copy, and chunk c+1
// travels while chunk c sweeps ("staged through hipHostMalloc pinned buffers").
struct CopyPool {                                   // process-wide memcpy workers
    // `left` is only touched under `mu`: the waiter owns the Batch (its stack frame) and may destroy it the moment it
    // sees left == 0, so the last worker must still hold the mutex when it publishes the zero and notifies.
    struct Batch { int left = 0; std::mutex mu; std::condition_variable cv; };
    struct Task { char *d; const char *s; size_t n; Batch *b; };
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv;
    std::deque<Task> q;
    bool stop = false;
    int nworker = 0;
    void start()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (nworker) return;
        unsigned hc = std::thread::hardware_concurrency();
        // (three workers + the caller.  Round 6, EPYC 9575F host, 3600 x 1800 host to host / C4 x 8 through host pointers, ms:
        //  no worker 8.4 / 15.0, one 7.4-7.9 / 13.4-13.8, two 7.5 / 13.1, three 7.5 / 13.3, four 7.6 / 13.0, eight -- until
        //  then -- 7.7 / 13.4, sixteen 7.9 / 13.4, twenty-four 8.0 / 14.4: one core moves ~25 GB/s, more hands only add wake-ups)
        nworker = (int)std::max(1u, std::min(3u, hc / 2u));
        { const int e_ = XINV_ENV_INT("XINV_COPY_THREADS", 0); if (e_ > 0) nworker = std::min(e_, 32); if (e_ < 0) nworker = 0; }
        for (int i = 0; i < nworker; i++)
            th.emplace_back([this] {
                for (;;) {
                    Task t;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !q.empty(); });
                        if (stop && q.empty()) return;
                        t = q.front(); q.pop_front();
                    }
                    memcpy(t.d, t.s, t.n);
                    {
                        std::lock_guard<std::mutex> lk(t.b->mu);
                        if (--t.b->left == 0) t.b->cv.notify_all();
                    }                                    // (nothing of *t.b is touched after the unlock)
                }
            });
    }
    // copy n bytes with the workers + the calling thread; returns when every byte has landed
    void copy(void *d, const void *s, size_t n)
    {
        if (n < ((size_t)4 << 20) || nworker < 1 || th.empty()) { memcpy(d, s, n); return; }
        const int parts = nworker + 1;
        const size_t piece = ((n / parts) + 4095) & ~(size_t)4095;
        Batch b;
        std::vector<Task> mine;
        size_t off = 0;
        int k = 0;
        for (; off < n; off += piece, k++) {
            Task t{(char *)d + off, (const char *)s + off, std::min(piece, n - off), &b};
            mine.push_back(t);
        }
        b.left = (int)mine.size() - 1;               // (before the tasks are visible to the workers)
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 1; i < mine.size(); i++) q.push_back(mine[i]);
        }
        cv.notify_all();
        memcpy(mine[0].d, mine[0].s, mine[0].n);
        if (mine.size() > 1) {
            std::unique_lock<std::mutex> lk(b.mu);
            b.cv.wait(lk, [&] { return b.left == 0; });
        }
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) if (t.joinable()) t.join();
    }
};
static CopyPool g_copy_pool;

struct StageRing {                                  // pinned slots of one direction on one device
    // (round 6, measured and not kept: eight slots of 8 MiB -- less of the first / last slot's memcpy exposed: 3600 x 1800 host
    //  to host 8.4 -> 8.2 ms, but C5 x 15 through host pointers 160 -> 209 ms: four times the DMAs and event waits)
    static constexpr int NSLOT = 4;
    static constexpr size_t SLOT = (size_t)32 << 20;
    char *buf[NSLOT] = {};
    hipEvent_t ev[NSLOT] = {};
    bool inflight[NSLOT] = {};
    char *dst[NSLOT] = {};                                        // downloads: where the slot's bytes go on the host
    size_t len[NSLOT] = {};
    int next = 0;
    // Forget every slot in flight (after the stream they were queued on has been drained): an error return from a
    // staged copy must not leave `dst` / `len` pointing into that call's host array for the next call to retire.
    void reset()
    {
        for (int k = 0; k < NSLOT; k++) { inflight[k] = false; dst[k] = nullptr; len[k] = 0; }
        next = 0;
    }
    int ensure()
    {
        for (int k = 0; k < NSLOT; k++) {
            if (!buf[k]) HIPCHK(hipHostMalloc((void **)&buf[k], SLOT, hipHostMallocDefault));
            if (!ev[k]) HIPCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        return XINV_OK;
    }
};

// host -> device through the ring (called on the uploader thread)
static int stage_h2d(StageRing &r, hipStream_t s, double *dev, const double *host, size_t bytes)
{
    if (bytes < ((size_t)1 << 20)) {                 // small: the runtime's own staged copy
        HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        return XINV_OK;
    }
    int rc = r.ensure();
    if (rc) return rc;
    // An idle ring (the first array of a call): the DMA engine waits for the first slot's memcpy -- 32 MiB, half a
    // millisecond nothing overlaps.  Start with short pieces (4, 8, 16 MiB), so that the engine starts after 4 MiB;
    // a ring with copies in flight keeps whole slots (more, smaller DMAs cost a long batch a third: StageRing).
    bool idle = true;
    for (int k = 0; k < StageRing::NSLOT && idle; k++)
        if (r.inflight[k]) { if (hipEventQuery(r.ev[k]) == hipSuccess) r.inflight[k] = false; else idle = false; }
    (void)hipGetLastError();                         // (hipErrorNotReady of the query)
    size_t piece = idle ? ((size_t)4 << 20) : StageRing::SLOT;
    for (size_t off = 0; off < bytes; ) {
        const size_t n = std::min(piece, bytes - off);
        const int k = r.next;
        if (r.inflight[k]) { HIPCHK(hipEventSynchronize(r.ev[k])); r.inflight[k] = false; }
        g_copy_pool.copy(r.buf[k], (const char *)host + off, n);
        HIPCHK(hipMemcpyAsync((char *)dev + off, r.buf[k], n, hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(r.ev[k], s));
        r.inflight[k] = true;
        r.next = (k + 1) % StageRing::NSLOT;
        off += n;
        piece = std::min(StageRing::SLOT, piece * 2);
    }
    return XINV_OK;
}

static int stage_d2h_retire(StageRing &r, int k)
{
    if (!r.inflight[k]) return XINV_OK;
    HIPCHK(hipEventSynchronize(r.ev[k]));
    g_copy_pool.copy(r.dst[k], r.buf[k], r.len[k]);
    r.inflight[k] = false;
    return XINV_OK;
}

// device -> host through the ring (called on the downloader thread); returns when the bytes are in `host`
static int stage_d2h(StageRing &r, hipStream_t s, double *host, const double *dev, size_t bytes)
{
    if (bytes < ((size_t)1 << 20)) {
        HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return XINV_OK;
    }
    int rc = r.ensure();
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += StageRing::SLOT) {
        const size_t n = std::min(StageRing::SLOT, bytes - off);
        const int k = r.next;
        if ((rc = stage_d2h_retire(r, k))) return rc;
        HIPCHK(hipMemcpyAsync(r.buf[k], (const char *)dev + off, n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(r.ev[k], s));
        r.inflight[k] = true; r.dst[k] = (char *)host + off; r.len[k] = n;
        r.next = (k + 1) % StageRing::NSLOT;
    }
    for (int i = 0; i < StageRing::NSLOT; i++)       // oldest first
        if ((rc = stage_d2h_retire(r, (r.next + i) % StageRing::NSLOT))) return rc;
    return XINV_OK;
}

// ------------------------------------------------------------------ NUMA placement of a device's host threads
// In-process multi-GPU (xinv_options.ndev > 1): the host thread that drives a GPU -- and the uploader / downloader threads
// it starts, which inherit its affinity, and the pinned staging rings they allocate (first touch) -- is bound to the CPUs of
// the NUMA node the GPU hangs off (sysfs: /sys/bus/pci/devices/<bus id>/numa_node -> /sys/devices/system/node/nodeN/cpulist),
// so that eight GPUs' staging copies do not all run out of one socket's memory.  Best effort: no sysfs entry, node -1 or a
// cpulist that does not parse leave the thread where it is.  Returns the node, or -1.
#include <sched.h>
static int bind_thread_to_device_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return -1;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return -1;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return -1;
    cpu_set_t set;
    CPU_ZERO(&set);
    int count = 0;
    char *save = nullptr;                                 // (several device threads parse at once: strtok_r)
    for (char *tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {     // "0-63,128-191"
        int a = -1, b = -1;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) { CPU_SET(c, &set); count++; } }
        else if (sscanf(tok, "%d", &a) == 1 && a >= 0 && a < CPU_SETSIZE) { CPU_SET(a, &set); count++; }
    }
    if (count == 0) return -1;
    // (only CPUs the process may run on: a cgroup / taskset restriction of the caller stays in force)
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) == 0) {
        cpu_set_t both;
        CPU_AND(&both, &set, &cur);
        if (CPU_COUNT(&both) == 0) return -1;
        set = both;
    }
    return sched_setaffinity(0, sizeof set, &set) == 0 ? node : -1;
}

// ------------------------------------------------------------------ per-device workspace
// Grown on demand, reused across solves (no hipMalloc in steady state).
#define XINV_MAX_LANES 4
#define XINV_MAX_INFLIGHT 6          /* host-pointer entries: chunk solves in flight on one device (workspace slots 0 .. 5) */
#define XINV_DEFAULT_INFLIGHT 2          /* ... of the 2-D forms (three for the 3-D ones: xinv_hostptr.h) */
struct Workspace {
    int device = -1;
    int cus = 0;                                        // the device's compute units (make_plan)
    int slot = 0;                                       // 0: the device's workspace; 1: a second one, so that two solves of ONE
                                                        // host-pointer call can be in flight on the device at once (solve_host_one)
    std::recursive_mutex busy;                          // one solve at a time per device
    double *S2 = nullptr; size_t S2_cap = 0;            // ping-pong twin of S (fused path)
    hipEvent_t ev_tail = nullptr; bool tail_pending = false;   // a plan solve's redo pass and last copy out of S2 / S3 may still
                                                        // be in flight (xinv_plan_solve: S completes in stream order): an event
                                                        // the WORKSPACE owns sits behind them -- the caller may destroy its stream
                                                        // (tail_wait: whoever touches S2 / S3 or a plan's buffers next waits on it)
    XinvCtl *ctl = nullptr; size_t ctl_cap = 0;
    void *partials = nullptr; size_t partials_cap = 0;  // norm partials
    size_t partials_half = 0;                           // lagged norm: byte offset of the odd launches' buffer
    double *S3 = nullptr; size_t S3_cap = 0;            // lagged norm: third S buffer
    int *dflag = nullptr;
    int *d_hook = nullptr;                              // test-hooks build: {tile, launch tag, member} (FusedArgs::dbg)
    int *dflags16 = nullptr, *hflags16 = nullptr;      // x-uniform detection flags
    XinvCtl *hctl = nullptr; size_t hctl_cap = 0;       // pinned mirror of ctl
    unsigned *hmail = nullptr; unsigned mail_seq = 0;   // pinned sequence word of k_ctl_mail (short solves: the host spins on it)
    int *hflag = nullptr;
    hipEvent_t ev0[2] = {nullptr, nullptr}, ev1[2] = {nullptr, nullptr}, evc[2] = {nullptr, nullptr};
    hipStream_t gstream = nullptr;                      // capture stream for the small-problem hipGraph
    hipStream_t s_side = nullptr; hipEvent_t ev_side0 = nullptr, ev_side1 = nullptr;   // skipped tiles' work beside the first sweep launch
    hipStream_t s_lane[XINV_MAX_LANES] = {}, s_poll = nullptr;   // sweep loop in lanes: lanes 1.. of the batch; the control-block copies
    hipEvent_t ev_lane[XINV_MAX_LANES][2] = {}, ev_s = nullptr;  // [0]: the caller's stream
    hipStream_t s_up = nullptr, s_down = nullptr, s_compute = nullptr;   // host-pointer entries: copy / sweep overlap
    // masked-tile skipping
    unsigned char *d_act = nullptr; size_t d_act_cap = 0;
    unsigned char *h_act = nullptr; size_t h_act_cap = 0;      // pinned
    int *d_list = nullptr; size_t d_list_cap = 0;               // [nbatch][ntl] then [nbatch][nskip]
    int *h_list = nullptr; size_t h_list_cap = 0;               // pinned
    bool act_ready = false; int act_uw = 0; const double *act_f = nullptr;   // activity map issued ahead (issue_strip_active)
    bool act_synced = false;                                    // ... and its copy to the host is known to have completed
    std::vector<int> h_pre;                                     // plan_tile_skip: prefix counts (kept: no allocation per solve)
    double *d_tsum = nullptr; size_t d_tsum_cap = 0;            // tsum | tcnt | xsum | xcnt
    void *d_rowf = nullptr; size_t d_rowf_cap = 0;              // k_pipe2d: per-row records [nbatch][yc][PIPE_RW]
    double *d_pfac = nullptr; size_t d_pfac_cap = 0;            // k_pipe2d<FusedGen2DQ>: the point-factor stream Q [nbatch][yc][xc]
    void *wd_part = nullptr; size_t wd_part_cap = 0;            // watchdog recovery: partials of the separate norm kernels
    StageRing ring_up, ring_down;                               // host-pointer entries: the library's pinned staging
};

static std::mutex g_ws_mutex;
static std::vector<Workspace *> g_ws;

static Workspace *get_ws(int device, int slot = 0)
{
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    for (auto *w : g_ws) if (w->device == device && w->slot == slot) return w;
    Workspace *w = new Workspace();
    w->device = device;
    w->slot = slot;
    g_ws.push_back(w);
    return w;
}

template <class T>
static int ensure_dev(T **p, size_t *cap, size_t need_bytes)
{
    if (*cap >= need_bytes && *p) return XINV_OK;
    if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(hipMalloc((void **)p, need_bytes));
    *cap = need_bytes;
    return XINV_OK;
}

// ------------------------------------------------------------------ problem description
enum { KIND_STD2D = 0, KIND_GEN2D = 1, KIND_STD3D = 2, KIND_BIH2D = 3, KIND_STD2DT = 4, KIND_GEN3D = 5 };
static inline bool is3d(int kind) { return kind == KIND_STD3D || kind == KIND_GEN3D; }

struct Problem {
    int kind;
    int64_t nbatch, zc, yc, xc;
    double *S;
    const double *c[10];         // std2d/std3d: A,B,C,F ; gen2d: A..G ; bih2d: A..J ; std2dt: A..F ; gen3d: A..H
    int64_t sS, sc[10];
    int ncoef;
    unsigned rowconst;           // host entries: arrays given as one value per row (see xinv.h)
    unsigned f32;                // host entries: bit 0 = S, bit q+1 = coefficient q is FLOAT32 on the host (xinv_options.f32_mask)
    unsigned known_um;           // bit q: coefficient q is known to be constant along x (a resident plan expanded it from one
                                 // value per row itself: xinv_plan_create_*, rowconst_mask) -- the detection pass does not read it
    int BCz, BCy, BCx;
    XinvScal sc_;
    XinvStop stop;
};

static int bc_ok(int b) { return b == XINV_BC_FIXED || b == XINV_BC_EXTEND || b == XINV_BC_PERIODIC; }

static int validate(const Problem &p, const double *flags)
{
    if (!p.S || !flags) return fail_arg("null S or flags");
    for (int q = 0; q < p.ncoef; q++)
        if (!p.c[q] && !(q == 1 && (p.kind == KIND_STD2D || p.kind == KIND_GEN2D)))   // B may be NULL: identically 0
            return fail_arg("null coefficient array");
    if (p.nbatch < 1) return fail_arg("nbatch < 1");
    if (p.yc < 3 || p.xc < 3 || (is3d(p.kind) && p.zc < 3))
        return fail_arg("every core dimension needs at least 3 points");
    if (!bc_ok(p.BCy) || !bc_ok(p.BCx) || (is3d(p.kind) && !bc_ok(p.BCz)))
        return fail_arg("unknown boundary-condition code");
    if (p.kind == KIND_BIH2D && (p.yc < 5 || p.xc < 7))
        return fail_arg("the biharmonic form needs yc >= 5 and xc >= 7");
    if (p.stop.mxLoop < 0) return fail_arg("mxLoop < 0");
    const int64_t n = p.zc * p.yc * p.xc;
    if (p.nbatch > 1 && p.sS < n) return fail_arg("S batch stride smaller than one slice");
    for (int q = 0; q < p.ncoef; q++) {
        const int64_t need = ((p.rowconst >> q) & 1u) ? p.zc * p.yc : n;
        if (p.c[q] && p.sc[q] != 0 && p.sc[q] < need)
            return fail_arg("coefficient batch stride must be 0 (shared) or >= slice size");
    }
    return XINV_OK;
}

static void fill_options(xinv_options &o, const xinv_options *in)
{
    xinv_default_options(&o);
    if (in) o = *in;
}


// ------------------------------------------------------------------ host-pointer staging
// Host <-> HBM path of the *_f64 / *_batched entry points.  Device buffers come from a
// per-device pool that is kept across calls (the coefficient stack of a repeated solve is
// re-uploaded but never re-allocated).  Large host arrays are pinned IN PLACE for the duration
// of the call (hipHostRegister) so the DMA engines read them directly at PCIe rate and all
// uploads are queued asynchronously on one stream; small arrays, or hosts where registration
// fails, take the runtime's staged copy.
struct DevPool {
    std::vector<std::pair<void *, size_t>> bufs;   // (ptr, capacity)
    size_t next = 0;
    void reset() { next = 0; }
};
static std::mutex g_pool_mutex;
static std::vector<std::pair<int, DevPool *>> g_pools;

static DevPool *get_pool(int device)
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (auto &e : g_pools) if (e.first == device) return e.second;
    DevPool *p = new DevPool();
    g_pools.push_back({device, p});
    return p;
}

static int pool_alloc(DevPool *pool, size_t bytes, double **out)
{
    if (pool->next < pool->bufs.size()) {
        auto &b = pool->bufs[pool->next];
        if (b.second < bytes) {
            HIPCHK(hipFree(b.first));
            b.first = nullptr; b.second = 0;
            HIPCHK(hipMalloc(&b.first, bytes));
            b.second = bytes;
        }
        *out = (double *)b.first;
        pool->next++;
        return XINV_OK;
    }
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, bytes));
    pool->bufs.push_back({d, bytes});
    pool->next++;
    *out = (double *)d;
    return XINV_OK;
}

// In-place pinning of the CALLER's arrays (hipHostRegister for the duration of the call: the DMA
// engines then read them at PCIe rate, 55 GB/s against ~25 GB/s for the runtime's staged copy) is OFF
// by default since round 2: registering and unregistering ranges of memory the caller's allocator
// later unmaps and maps again leaves this runtime (ROCm 7.2) in a state in which a LATER pageable
// copy of another array at the same virtual address aborts the process -- about one run in seven of
// tests/test_gpu_frontend.py + test_gpu_fullsize.py, never without the registration (28 runs,
// profiles/r02_pinning_abort.txt).  XINV_FLAG_PIN_HOST (or XINV_PIN=1 in the environment) turns it on
// for callers whose buffers stay mapped.
struct Pinned {                                     // host ranges registered for this call (opt-in, see above)
    static bool env_allowed()
    {
        return XINV_ENV_INT("XINV_PIN", 0) != 0;         // (variant / hooks builds only: the shipped library takes XINV_FLAG_PIN_HOST)
    }
    std::vector<std::pair<char *, size_t>> regs;    // (base, bytes) of every registered range
    const Pinned *outer = nullptr;                  // multi-device call: the parent's portable registrations
    std::vector<hipStream_t> streams;               // streams that may still hold copies of these ranges
    bool enabled = false;                           // this call registers ranges (XINV_FLAG_PIN_HOST / XINV_PIN=1)
    unsigned flags = hipHostRegisterDefault;
    bool try_pin(const void *h, size_t bytes)
    {
        if (!enabled || bytes < (1u << 20)) return false;
        if (hipHostRegister((void *)h, bytes, flags) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        regs.push_back({(char *)h, bytes});
        return true;
    }
    // Host arrays the CALLER allocated as pinned memory (hipHostMalloc / a framework's pinned allocator: the front end's
    // result array comes out of torch's pinned pool, xinvert_amd/core.py): the DMA engines reach them directly, no staging
    // copy through the library's ring.  Asked once per array (note_pinned); nothing is registered or unregistered.
    std::vector<std::pair<char *, size_t>> native;
    void note_pinned(const void *h, size_t bytes)
    {
        if (!h || bytes < (1u << 20)) return;
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, h) != hipSuccess) { (void)hipGetLastError(); return; }   // (pageable memory: an error, cleared)
        if (at.type != hipMemoryTypeHost) return;
        // the allocation must cover the whole range (the last byte belongs to a pinned allocation too)
        hipPointerAttribute_t a2;
        memset(&a2, 0, sizeof a2);
        if (hipPointerGetAttributes(&a2, (const char *)h + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return; }
        if (a2.type != hipMemoryTypeHost) return;
        native.push_back({(char *)h, bytes});
    }
    // [h, h + bytes) lies inside a range registered by this call (or by the parent of a per-device call) or inside an
    // array the caller pinned itself: every chunk and member of it copies straight out of / into the caller's memory
    bool covers(const void *h, size_t bytes) const
    {
        const char *c = (const char *)h;
        for (const auto &r : regs) if (c >= r.first && c + bytes <= r.first + r.second) return true;
        for (const auto &r : native) if (c >= r.first && c + bytes <= r.first + r.second) return true;
        return outer ? outer->covers(h, bytes) : false;
    }
    // Every return path -- error paths included -- drains the copy streams before the ranges are
    // unregistered: an async copy still in flight must not lose its pinning.
    ~Pinned()
    {
        for (hipStream_t s : streams) (void)hipStreamSynchronize(s);
        for (const auto &r : regs) (void)hipHostUnregister(r.first);
    }
};

