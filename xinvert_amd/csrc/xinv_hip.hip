// xinv_hip.hip -- host driver and C-ABI of the MI355X SOR inversion engine (include/xinv.h).
//
// Replaces the reference's numba kernels behind the call boundary of xinvert/core.py
// (core.py:60-69, 130-139, 419-428).  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//
// Control flow of one solve (all batch members together, one stream):
//   k_ctl_init -> [ sweep launches ... ] x check_every -> async read-back of the per-member
//   control blocks -> repeat until every member has stopped.  The stopping rule runs on the
//   device after every sweep (last-arriving workgroup / k_norm_final); once a member is done
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
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/xinv.h"
#include "xinv_device.h"
#include "xinv_colour.h"
#include "xinv_fused.h"
#include "xinv_fused3d.h"
#include "xinv_fused9.h"

#define XINV_VERSION 100
#define XINV_MEMBER_CHUNK 32768     /* members per launch: grid.y / grid.z are limited to 65535 */

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

// ------------------------------------------------------------------ per-device workspace
// Grown on demand, reused across solves (no hipMalloc in steady state).
struct Workspace {
    int device = -1;
    std::recursive_mutex busy;                          // one solve at a time per device
    double *S2 = nullptr; size_t S2_cap = 0;            // ping-pong twin of S (fused path)
    XinvCtl *ctl = nullptr; size_t ctl_cap = 0;
    void *partials = nullptr; size_t partials_cap = 0;  // psum + pcnt
    int *dflag = nullptr;
    int *dflags16 = nullptr, *hflags16 = nullptr;      // x-uniform detection flags
    XinvCtl *hctl = nullptr; size_t hctl_cap = 0;       // pinned mirror of ctl
    int *hflag = nullptr;
    hipEvent_t ev0[2] = {nullptr, nullptr}, ev1[2] = {nullptr, nullptr}, evc[2] = {nullptr, nullptr};
    // masked-tile skipping
    unsigned char *d_act = nullptr; size_t d_act_cap = 0;
    unsigned char *h_act = nullptr; size_t h_act_cap = 0;      // pinned
    int *d_list = nullptr; size_t d_list_cap = 0;               // [nbatch][ntl] then [nbatch][nskip]
    int *h_list = nullptr; size_t h_list_cap = 0;               // pinned
    double *d_tsum = nullptr; size_t d_tsum_cap = 0;            // tsum | tcnt | xsum | xcnt
};

static std::mutex g_ws_mutex;
static std::vector<Workspace *> g_ws;

static Workspace *get_ws(int device)
{
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    for (auto *w : g_ws) if (w->device == device) return w;
    Workspace *w = new Workspace();
    w->device = device;
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

// ------------------------------------------------------------------ launch helpers
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

struct Plan {
    int path, base, seam, ncol;
    int K, RY, nsg, nrb;     // nsg: 2-D = workgroups per member (partials sizing); 3-D = x strips
    bool aligned;
    unsigned umask;          // fused streams whose rows are constant along x (bit = stream index)
    unsigned um;             // the kernel variant's mask (subset of umask)
    bool even_split;         // rows split evenly over nrb row blocks (RY = average height)
    bool nine;               // 9-point form on the fused 4-colour kernel
    bool skip;               // masked-tile skipping: launches of K == Plan::K run the listed tiles only
    int ntl, nskip;          // list entries per member (active, multiple of 4 / skipped)
    int skip_pct;            // share of wave-tiles skipped, percent
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

// Launch one instantiation -- or, when `occ` is given, only report how many of its workgroups
// fit on a CU (register-limited: 1 to 3), which the tiling heuristic needs.
template <class M, int K, bool AL, unsigned UM, bool EXT>
static int fused_one(dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a, int *occ)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fused2d<M, K, AL, UM, EXT>, 256, 0) != hipSuccess)
            n = 1;
        *occ = n < 1 ? 1 : n;
        return 0;
    }
    hipLaunchKernelGGL((k_fused2d<M, K, AL, UM, EXT>), grid, block, 0, st, a);
    return 0;
}

template <class M, bool AL, unsigned UM, bool EXT>
static int launch_fused_k(int K, dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a, int *occ)
{
    switch (K) {
    case 1: return fused_one<M, 1, AL, UM, EXT>(grid, block, st, a, occ);
    case 2: return fused_one<M, 2, AL, UM, EXT>(grid, block, st, a, occ);
    default: break;
    }
    return 1;
}

template <class M, bool AL, bool EXT>
static int launch_fused_um(unsigned um, int K, dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a,
                           int *occ)
{
    if constexpr (std::is_same<M, FusedStd2D>::value) {
        if (um == 3u) return launch_fused_k<M, AL, 3u, EXT>(K, grid, block, st, a, occ);
    } else if constexpr (std::is_same<M, FusedStd2DT>::value) {
        if (um == 7u) return launch_fused_k<M, AL, 7u, EXT>(K, grid, block, st, a, occ);
    } else {
        if (um == 0x1fu) return launch_fused_k<M, AL, 0x1fu, EXT>(K, grid, block, st, a, occ);
        if (um == 0x1cu) return launch_fused_k<M, AL, 0x1cu, EXT>(K, grid, block, st, a, occ);
    }
    return launch_fused_k<M, AL, 0u, EXT>(K, grid, block, st, a, occ);
}

template <class M>
static int launch_fused_m(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                          const FusedArgs &a, int *occ)
{
    if (al) return ext ? launch_fused_um<M, true, true>(um, K, grid, block, st, a, occ)
                       : launch_fused_um<M, true, false>(um, K, grid, block, st, a, occ);
    return ext ? launch_fused_um<M, false, true>(um, K, grid, block, st, a, occ)
               : launch_fused_um<M, false, false>(um, K, grid, block, st, a, occ);
}

static int fused_dispatch(int kind, bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                          hipStream_t st, const FusedArgs &a, int *occ)
{
    if (kind == KIND_GEN2D) return launch_fused_m<FusedGen2D>(al, ext, um, K, grid, block, st, a, occ);
    if (kind == KIND_STD2DT) return launch_fused_m<FusedStd2DT>(al, ext, um, K, grid, block, st, a, occ);
    return launch_fused_m<FusedStd2D>(al, ext, um, K, grid, block, st, a, occ);
}

static bool ptr_al16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

static int launch_fused(const Problem &p, const Plan &pl, int K, const double *src, double *dst,
                        Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                        int no_ctl)
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
    }
    a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    a.ext = (p.BCy == XINV_BC_EXTEND);
    a.tall = (p.yc > p.xc);
    a.RY = pl.even_split ? 0 : pl.RY;
    const int UW = 128 - 4 * K;
    a.nstrip = (int)cdiv(p.xc, UW);
    a.nrb = pl.even_split ? pl.nrb : (int)cdiv(p.yc, pl.RY);
    a.nwg = (int)cdiv((int64_t)a.nstrip * a.nrb, 4);
    a.force = force; a.no_ctl = no_ctl;
    a.member0 = member0;
    a.sc_ = p.sc_;
    a.ctl = ws->ctl;
    a.stop = p.stop;
    const size_t NBmax = (size_t)pl.nsg;             // workgroups per member, narrowest strips (K = XINV_KMAX)
    a.psum = (unsigned long long *)ws->partials;
    a.pcnt = (long long *)((char *)ws->partials + p.nbatch * XINV_KMAX * NBmax * sizeof(double));
    if (pl.skip && K == pl.K) {                      // the lists were built for this K's strips
        a.tile_list = ws->d_list;
        a.ntl = pl.ntl;
        a.nwg = pl.ntl / 4;
        char *base = (char *)ws->d_tsum;
        const size_t nt = (size_t)p.nbatch * pl.nskip;
        a.xsum = (const double *)(base + nt * (sizeof(double) + sizeof(long long)));
        a.xcnt = (const long long *)(base + nt * (sizeof(double) + sizeof(long long)) + p.nbatch * sizeof(double));
    }
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {      // grid.y is limited to 65535
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)a.nwg, (unsigned)nm, 1), block(256, 1, 1);
        if (fused_dispatch(p.kind, pl.aligned, a.ext != 0, pl.um, K, grid, block, st, a, nullptr))
            return fail_arg("unsupported sweeps_per_launch for this kernel variant");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

// ---- 9-point fused launch ---------------------------------------------------------------------
template <class M, int K>
static void launch_fused9_k(bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ)
{
    dim3 block(256, 1, 1);
#define L9(AL, EXT)                                                                              \
    do {                                                                                         \
        if (occ) {                                                                               \
            int n = 0;                                                                           \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fused9<M, K, AL, EXT>, 256, 0) != hipSuccess) n = 1; \
            *occ = n < 1 ? 1 : n;                                                                \
        } else hipLaunchKernelGGL((k_fused9<M, K, AL, EXT>), grid, block, 0, st, a);             \
    } while (0)
    if (al) { if (ext) L9(true, true); else L9(true, false); }
    else    { if (ext) L9(false, true); else L9(false, false); }
#undef L9
}

static int fused9_dispatch(int kind, int K, bool al, bool ext, dim3 grid, hipStream_t st,
                           const FusedArgs &a, int *occ)
{
    if (kind == KIND_GEN2D) {
        if (K == 1) launch_fused9_k<Fused9Gen, 1>(al, ext, grid, st, a, occ); else return 1;
    } else {
        if (K == 1) launch_fused9_k<Fused9Std, 1>(al, ext, grid, st, a, occ);
        else if (K == 2) launch_fused9_k<Fused9Std, 2>(al, ext, grid, st, a, occ);
        else return 1;
    }
    return 0;
}

static int launch_fused9(const Problem &p, const Plan &pl, int K, const double *src, double *dst,
                         Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                         int no_ctl)
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
    a.nstrip = (int)cdiv(p.xc, 128 - 8 * K);
    a.nrb = pl.even_split ? pl.nrb : (int)cdiv(p.yc, pl.RY);
    a.nwg = (int)cdiv((int64_t)a.nstrip * a.nrb, 4);
    a.force = force; a.no_ctl = no_ctl;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    const size_t NBmax = (size_t)pl.nsg;
    a.psum = (unsigned long long *)ws->partials;
    a.pcnt = (long long *)((char *)ws->partials + p.nbatch * XINV_KMAX * NBmax * sizeof(double));
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)a.nwg, (unsigned)nm, 1);
        if (fused9_dispatch(p.kind, K, pl.aligned, a.ext != 0, grid, st, a, nullptr))
            return fail_arg("unsupported sweeps_per_launch for the 9-point kernel");
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

// ---- 3-D fused launch ------------------------------------------------------------------------
template <int NW>
static int launch_fused3d_nw(bool al, bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    dim3 block(NW * 64, 1, 1);
#define L3(AL, UNI, EXT) hipLaunchKernelGGL((k_fused3d<NW, AL, UNI, EXT>), grid, block, 0, st, a)
    if (al) {
        if (uni) { if (ext) L3(true, true, true); else L3(true, true, false); }
        else     { if (ext) L3(true, false, true); else L3(true, false, false); }
    } else {
        if (uni) { if (ext) L3(false, true, true); else L3(false, true, false); }
        else     { if (ext) L3(false, false, true); else L3(false, false, false); }
    }
#undef L3
    return 0;
}

static int launch_fused3d(const Problem &p, const Plan &pl, const double *src, double *dst,
                          Workspace *ws, hipStream_t st, int64_t member0, int64_t nmem, int force,
                          int no_ctl)
{
    Fused3Args a;
    memset(&a, 0, sizeof a);
    a.src = src; a.dst = dst; a.sS = p.sS;
    for (int q = 0; q < 4; q++) { a.c[q] = p.c[q]; a.sc[q] = p.sc[q]; }
    a.zc = p.zc; a.yc = p.yc; a.xc = p.xc;
    a.per = (p.BCx == XINV_BC_PERIODIC);
    a.nstrip = pl.nsg; a.njb = pl.nrb;
    a.force = force; a.no_ctl = no_ctl; a.member0 = member0;
    a.sc_ = p.sc_; a.ctl = ws->ctl; a.stop = p.stop;
    const size_t NB = (size_t)pl.nsg * pl.nrb;
    a.psum = (unsigned long long *)ws->partials;
    a.pcnt = (long long *)((char *)ws->partials + p.nbatch * NB * sizeof(double));
    const bool ext = (p.BCy == XINV_BC_EXTEND), uni = (pl.um == 7u);
    for (int64_t m0 = 0; m0 < nmem; m0 += XINV_MEMBER_CHUNK) {
        const int64_t nm = std::min<int64_t>(XINV_MEMBER_CHUNK, nmem - m0);
        a.member0 = member0 + m0;
        dim3 grid((unsigned)NB, (unsigned)nm, 1);
        if (pl.RY == 8) launch_fused3d_nw<8>(pl.aligned, uni, ext, grid, st, a);
        else if (pl.RY == 12) launch_fused3d_nw<12>(pl.aligned, uni, ext, grid, st, a);
        else launch_fused3d_nw<16>(pl.aligned, uni, ext, grid, st, a);
    }
    HIPCHK(hipGetLastError());
    return XINV_OK;
}

// one full coloured sweep (+ norm + stop rule) in place on p.S
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
        if (!per) {
            // non-periodic x: one launch per row class does its three column colours in registers
            dim3 gb(cdiv(cdiv(p.xc, 180), 4), cdiv(p.yc, 3) + 1, (unsigned)nm), bb(256, 1, 1);
            for (int cj = 0; cj < 3; cj++) {
                a.colour = cj;
                if (pl.umask) hipLaunchKernelGGL(k_bih_rowclass<true>, gb, bb, 0, st, a);
                else          hipLaunchKernelGGL(k_bih_rowclass<false>, gb, bb, 0, st, a);
            }
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
        for (int cc = 0; cc < pl.ncol; cc++) {
            a.colour = cc;
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
        for (int cc = 0; cc < pl.ncol; cc++) {
            a.colour = cc;
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
static int64_t choose_row_blocks(int64_t yc, int64_t nstrip, int64_t nbatch, int K, int occ)
{
    occ = std::max(1, std::min(occ, 3));
    const int64_t cap = 256 * (int64_t)occ, period = 2 * K + 2;
    int64_t best = 1; double best_cost = 1e300;
    const int64_t nmin = std::max<int64_t>(1, cdiv(yc, 128)), nmax = std::max<int64_t>(nmin, yc / 4);
    for (int64_t nr = nmin; nr <= nmax; nr++) {
        const int64_t rows = cdiv(yc, nr) + 1;                     // +1: even rounding
        const int64_t steps = cdiv(rows + 4 * K, period) * period;
        const int64_t wgs = (int64_t)cdiv(nstrip * nr, 4) * nbatch;
        // rounds of `cap` resident workgroups; inside a round a CU holds ceil(w/256) of them,
        // and a lone workgroup on a CU leaves issue slots idle (charged like 1.6)
        const int64_t rounds = cdiv(wgs, cap);
        const int64_t w_last = wgs - (rounds - 1) * cap;
        const double full = (occ == 1) ? 1.6 : (double)occ;
        const double last = (w_last <= 256) ? 1.6 : (double)cdiv(w_last, 256);
        const double cost = ((double)(rounds - 1) * full + last) * (double)steps;
        if (cost <= best_cost * 1.0001) { best_cost = std::min(cost, best_cost); best = nr; }   // ties: more, shorter tiles
    }
    return best;
}

// Cost of a fused 2-D launch in (workgroups per CU) x (steps per tile) units -- the model behind
// choose_row_blocks, shared with the masked-tile planner.
static double tile_cost(int64_t wgs, int64_t rows, int K, int occ)
{
    occ = std::max(1, std::min(occ, 3));
    const int64_t cap = 256 * (int64_t)occ, period = 2 * K + 2;
    const int64_t steps = cdiv(rows + 1 + 4 * K, period) * period;
    const int64_t rounds = std::max<int64_t>(1, cdiv(wgs, cap));
    const int64_t w_last = wgs - (rounds - 1) * cap;
    const double full = (occ == 1) ? 1.6 : (double)occ;
    const double last = (w_last <= 256) ? 1.6 : (double)cdiv(w_last, 256);
    return ((double)(rounds - 1) * full + last) * (double)steps;
}

static int fused_dispatch(int kind, bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                          hipStream_t st, const FusedArgs &a, int *occ);

// Masked-tile skipping for the 5-point fused kernels.  Wave-tiles whose forcing is undefined at
// every owned point (land, topography, polar caps) can never change (every mask predicate of the
// reference tests the forcing), so launches run the other tiles only; the row split is re-chosen
// so that the ACTIVE tiles fill the CUs evenly, and the skipped tiles' constant share of the norm
// is computed once.  Decided per solve from one pass over the forcing.
static int plan_tile_skip(const Problem &p, Plan &pl, Workspace *ws, hipStream_t st,
                          const xinv_options &opt)
{
    pl.skip = false; pl.ntl = pl.nskip = 0; pl.skip_pct = 0;
    const bool forced = (opt.flags & XINV_FLAG_FORCE_TILE_SKIP) != 0;
    const int K = pl.K, UW = 128 - 4 * K;
    const int nstrip = (int)cdiv(p.xc, UW);
    if (!forced && ((int64_t)nstrip * pl.nrb * p.nbatch < 1024 || (int64_t)nstrip * pl.nrb < 64 || p.nbatch > 64))
        return XINV_OK;                                   // small problems: nothing to balance
    const int64_t yc = p.yc, nb = p.nbatch;
    const int64_t cells = yc * nstrip;
    const int fi = (p.kind == KIND_STD2D) ? 3 : (p.kind == KIND_GEN2D ? 6 : 5);     // the forcing

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
    HIPCHK(hipStreamSynchronize(st));

    // prefix counts of active rows per (member, strip)
    std::vector<int> pre((size_t)(nb * nstrip * (yc + 1)));
    int64_t nact = 0;
    for (int64_t m = 0; m < nb; m++)
        for (int s = 0; s < nstrip; s++) {
            int *q = &pre[(size_t)((m * nstrip + s) * (yc + 1))];
            q[0] = 0;
            for (int64_t r = 0; r < yc; r++) q[r + 1] = q[r] + ws->h_act[(m * yc + r) * nstrip + s];
            nact += q[yc];
        }
    if (!forced && (double)nact > 0.92 * (double)(nb * cells)) return XINV_OK;     // little to skip

    const bool ext = (p.BCy == XINV_BC_EXTEND);
    auto bounds = [&](int nrb, int rb, int64_t &y0, int64_t &y1) {
        y0 = (((int64_t)rb * yc) / nrb) & ~(int64_t)1;
        y1 = (rb + 1 == nrb) ? yc : ((((int64_t)(rb + 1) * yc) / nrb) & ~(int64_t)1);
    };
    auto tile_active = [&](int64_t m, int nrb, int rb, int s) {
        if (ext && (rb == 0 || rb == nrb - 1)) return true;    // the boundary rows get their copy
        int64_t y0, y1; bounds(nrb, rb, y0, y1);
        const int *q = &pre[(size_t)((m * nstrip + s) * (yc + 1))];
        return q[y1] - q[y0] > 0;
    };
    auto active_wgs = [&](int nrb, int64_t *maxact) {
        int64_t wgs = 0, mx = 0;
        for (int64_t m = 0; m < nb; m++) {
            int64_t c = 0;
            for (int rb = 0; rb < nrb; rb++) {
                if (ext && (rb == 0 || rb == nrb - 1)) { c += nstrip; continue; }
                int64_t y0, y1; bounds(nrb, rb, y0, y1);
                const int *q = &pre[(size_t)(m * nstrip * (yc + 1))];
                for (int s = 0; s < nstrip; s++, q += yc + 1) c += (q[y1] - q[y0] > 0) ? 1 : 0;
            }
            wgs += cdiv(c, 4); mx = std::max(mx, c);
        }
        if (maxact) *maxact = mx;
        return wgs;
    };
    int occ = 2;
    {
        FusedArgs dummy; memset(&dummy, 0, sizeof dummy);
        fused_dispatch(p.kind, pl.aligned, ext, pl.um, K, dim3(1), dim3(256), st, dummy, &occ);
    }
    const double cost0 = tile_cost((int64_t)cdiv((int64_t)nstrip * pl.nrb, 4) * nb, cdiv(yc, pl.nrb), K, occ);
    // candidates: the row split that brings the ACTIVE workgroups back to the default count sits
    // near nrb / (active share); search a window around it
    int best = pl.nrb; double best_cost = 1e300;
    int lo = pl.nrb, hi = pl.nrb;
    if (!forced && opt.rows_per_tile == 0) {
        const int64_t w0 = active_wgs(pl.nrb, nullptr);
        const double share = std::max(0.05, (double)w0 / (double)((int64_t)cdiv((int64_t)nstrip * pl.nrb, 4) * nb));
        const double centre = (double)pl.nrb / share;
        const int64_t cap_rows = std::max<int64_t>(pl.nrb, yc / 4);
        lo = (int)std::min<int64_t>(cap_rows, std::max<int64_t>(pl.nrb, (int64_t)(centre * 0.85)));
        hi = (int)std::min<int64_t>(cap_rows, std::max<int64_t>(lo, (int64_t)(centre * 1.10) + 1));
    }
    best_cost = tile_cost(active_wgs(pl.nrb, nullptr), cdiv(yc, pl.nrb), K, occ);   // keep the split, skip only
    for (int nrb = lo; nrb <= hi; nrb++) {
        const double c = tile_cost(active_wgs(nrb, nullptr), cdiv(yc, nrb), K, occ);
        if (c < best_cost) { best_cost = c; best = nrb; }
    }
    if (!forced && best_cost > 0.95 * cost0) return XINV_OK;

    // lists for the chosen split
    int64_t maxact = 0;
    active_wgs(best, &maxact);
    const int64_t ntiles = (int64_t)nstrip * best;
    const int ntl = (int)(4 * std::max<int64_t>(1, cdiv(maxact, 4)));
    int64_t maxskip = 0, nskipped = 0;
    std::vector<std::vector<int>> act((size_t)nb), skp((size_t)nb);
    for (int64_t m = 0; m < nb; m++) {
        for (int rb = 0; rb < best; rb++)
            for (int s = 0; s < nstrip; s++)
                (tile_active(m, best, rb, s) ? act[(size_t)m] : skp[(size_t)m]).push_back(rb * nstrip + s);
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
        for (int t = 0; t < ntl; t++) hl[m * ntl + t] = t < (int)act[(size_t)m].size() ? act[(size_t)m][t] : -1;
        for (int t = 0; t < nskip; t++) hs[m * nskip + t] = t < (int)skp[(size_t)m].size() ? skp[(size_t)m][t] : -1;
    }
    HIPCHK(hipMemcpyAsync(ws->d_list, ws->h_list, nints * sizeof(int), hipMemcpyHostToDevice, st));
    const size_t nt = (size_t)nb * nskip;
    rc = ensure_dev(&ws->d_tsum, &ws->d_tsum_cap,
                    (nt + (size_t)nb) * (sizeof(double) + sizeof(long long)));
    if (rc) return rc;
    SkipNormArgs na;
    na.S = p.S; na.sS = p.sS; na.yc = yc; na.xc = p.xc; na.nstrip = nstrip; na.nrb = best; na.UW = UW;
    na.undef = p.sc_.undef; na.skip_list = ws->d_list + (size_t)nb * ntl; na.nskip_max = nskip;
    char *base = (char *)ws->d_tsum;
    na.tsum = (double *)base;
    na.tcnt = (long long *)(base + nt * sizeof(double));
    na.xsum = (double *)(base + nt * (sizeof(double) + sizeof(long long)));
    na.xcnt = (long long *)(base + nt * (sizeof(double) + sizeof(long long)) + (size_t)nb * sizeof(double));
    hipLaunchKernelGGL(k_skip_norm_tile, dim3((unsigned)nskip, (unsigned)nb, 1), dim3(64), 0, st, na);
    hipLaunchKernelGGL(k_skip_norm_sum, dim3((unsigned)nb, 1, 1), dim3(64), 0, st, na);
    HIPCHK(hipGetLastError());

    pl.skip = true; pl.ntl = ntl; pl.nskip = nskip;
    pl.skip_pct = (int)((100 * nskipped) / (ntiles * nb));
    pl.nrb = best; pl.even_split = true; pl.RY = (int)cdiv(yc, best);
    pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, 128 - 4 * XINV_KMAX) * pl.nrb, 4) + 1;
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
    *mask = 0;
    for (int q = 0; q < nstream; q++) if (!ws->hflags16[q]) *mask |= (1u << q);
    return XINV_OK;
}

// ------------------------------------------------------------------ the solve (device ptrs)
static int solve_dev(Problem &p, double *flags, const xinv_options *opt_in, hipStream_t st)
{
    int rc = validate(p, flags);
    if (rc) return rc;
    xinv_options opt;
    fill_options(opt, opt_in);

    int device = opt.device;
    if (device < 0) HIPCHK(hipGetDevice(&device));
    else HIPCHK(hipSetDevice(device));
    Workspace *ws = get_ws(device);
    std::lock_guard<std::recursive_mutex> solve_lock(ws->busy);
    if (!ws->ev0[0])
        for (int q = 0; q < 2; q++) {
            HIPCHK(hipEventCreate(&ws->ev0[q])); HIPCHK(hipEventCreate(&ws->ev1[q]));
            HIPCHK(hipEventCreateWithFlags(&ws->evc[q], hipEventDisableTiming));
        }
    if (!ws->dflag) {
        HIPCHK(hipMalloc((void **)&ws->dflag, sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&ws->hflag, sizeof(int), hipHostMallocDefault));
    }

    const int64_t n = p.zc * p.yc * p.xc;
    memset(&t_stats, 0, sizeof t_stats);

    // ---- colouring: red-black when the cross coefficient vanishes, else 4 colours ----------
    Plan pl;
    memset(&pl, 0, sizeof pl);
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
        pl.ncol = pl.base + (pl.seam ? 2 : 0);
    }

    // ---- path ------------------------------------------------------------------------------
    const bool fused5_ok = pl.base == 2 && !pl.seam && p.kind != KIND_BIH2D && p.kind != KIND_GEN3D;
    const bool fused9_ok = pl.base == 4 && !pl.seam && p.c[1] &&
                           (p.kind == KIND_STD2D || p.kind == KIND_GEN2D);
    const bool fused_ok = fused5_ok || fused9_ok;
    pl.path = XINV_PATH_COLOUR;
    pl.nine = false;
    if (fused_ok && opt.path != XINV_PATH_COLOUR) { pl.path = XINV_PATH_FUSED; pl.nine = !fused5_ok; }
    if (opt.path == XINV_PATH_FUSED && !fused_ok)
        return fail_arg("no fused kernel for this form (odd-xc periodic seam, biharmonic, or 9-point test form)");
    if (pl.path == XINV_PATH_COLOUR && !is3d(p.kind) && !p.c[1] && pl.base == 4)
        return fail_arg("internal: 9-point form without B");

    if (pl.path == XINV_PATH_FUSED && pl.nine) {
        // 9-point forms: 4-colour fused kernel, all coefficient arrays streamed
        pl.um = pl.umask = 0;
        pl.K = (p.kind == KIND_STD2D && opt.sweeps_per_launch != 1) ? 2 : 1;    // general form: registers allow K = 1 only
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
                fused9_dispatch(p.kind, pl.K, pl.aligned, p.BCy == XINV_BC_EXTEND, dim3(1), st, dummy, &occ);
                pl.nrb = (int)choose_row_blocks(p.yc, cdiv(p.xc, 128 - 8 * pl.K), p.nbatch, pl.K, occ);
            }
            pl.even_split = true;
            pl.RY = (int)cdiv(p.yc, pl.nrb);
        }
        pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, 128 - 8 * XINV_KMAX) * pl.nrb, 4) + 1;
    } else if (pl.path == XINV_PATH_FUSED && p.kind == KIND_STD3D) {
        // 3-D: one sweep per launch; cross-section of NW rows per workgroup (rows_per_tile = NW)
        pl.K = 1;
        pl.RY = (opt.rows_per_tile == 8 || opt.rows_per_tile == 12 || opt.rows_per_tile == 16)
                    ? opt.rows_per_tile : 0;     // 0: decided below, once the variant is known
        pl.nsg = (int)cdiv(p.xc, 124);                      // x strips
        pl.nrb = (int)cdiv(p.yc, pl.RY - 4);                // j blocks
        pl.aligned = !(p.xc & 1) && !(p.sS & 1) && ptr_al16(p.S);
        for (int q = 0; q < 4; q++) pl.aligned = pl.aligned && ptr_al16(p.c[q]) && !(p.sc[q] & 1);
        pl.umask = 0;
        if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
            rc = detect_xuniform(ws, st, p.c, p.sc, 3, p.nbatch, p.zc * p.yc, p.xc, &pl.umask);
            if (rc) return rc;
        }
        pl.um = (pl.umask == 7u) ? 7u : 0u;
        // 16 waves x 64 lanes leaves 128 VGPRs per lane: enough only for the x-uniform variant
        if (pl.RY == 0) pl.RY = (pl.um == 7u && p.BCy != XINV_BC_EXTEND) ? 16 : 12;
        pl.nrb = (int)cdiv(p.yc, pl.RY - 4);
    } else
    if (pl.path == XINV_PATH_FUSED) {
        // which coefficient streams are constant along x (lat-lon grids: functions of latitude)
        {
            const int cmapS[3] = {0, 2, 3}, cmapG[6] = {0, 2, 3, 4, 5, 6}, cmapT[4] = {0, 3, 4, 5};
            const int ns = (p.kind == KIND_STD2D) ? 3 : (p.kind == KIND_STD2DT ? 4 : 6);
            const double *arr[6]; int64_t strd[6];
            for (int q = 0; q < ns; q++) {
                const int sidx = (p.kind == KIND_STD2D) ? cmapS[q] : (p.kind == KIND_STD2DT ? cmapT[q] : cmapG[q]);
                arr[q] = p.c[sidx]; strd[q] = p.sc[sidx];
            }
            pl.umask = 0;
            if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
                rc = detect_xuniform(ws, st, arr, strd, ns, p.nbatch, p.yc, p.xc, &pl.umask);
                if (rc) return rc;
            }
            pl.um = pick_um(p.kind, pl.umask);
        }
        const int kmax = XINV_KMAX;
        pl.K = opt.sweeps_per_launch > 0 ? opt.sweeps_per_launch : 2;   // 2 sweeps per pass over HBM
        if (pl.K > kmax) return fail_arg("sweeps_per_launch must be 1 or 2");
        // Rows per tile (see the cost model below).
        pl.aligned = !(p.xc & 1) && !(p.sS & 1) && ptr_al16(p.S);
        const int cmap3[3] = {0, 2, 3}, cmap6[6] = {0, 2, 3, 4, 5, 6}, cmap4[4] = {0, 3, 4, 5};
        const int nc = (p.kind == KIND_STD2D) ? 3 : (p.kind == KIND_STD2DT ? 4 : 6);
        for (int q = 0; q < nc; q++) {
            const int s = (p.kind == KIND_STD2D) ? cmap3[q] : (p.kind == KIND_STD2DT ? cmap4[q] : cmap6[q]);
            pl.aligned = pl.aligned && ptr_al16(p.c[s]) && !(p.sc[s] & 1);
        }
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
                fused_dispatch(p.kind, pl.aligned, p.BCy == XINV_BC_EXTEND, pl.um, pl.K, dim3(1), dim3(256),
                               st, dummy, &occ);
            }
            const int64_t best = choose_row_blocks(p.yc, cdiv(p.xc, 128 - 4 * pl.K), p.nbatch, pl.K, occ);
            pl.nrb = (int)best;
            pl.even_split = true;
            pl.RY = (int)cdiv(p.yc, pl.nrb);
        }
        // workgroups per member with the narrowest strips any K uses: sizes the partials
        pl.nsg = (int)cdiv((int64_t)cdiv(p.xc, 128 - 4 * XINV_KMAX) * pl.nrb, 4) + 1;
        if (pl.even_split && !(opt.flags & XINV_FLAG_NO_TILE_SKIP)) {
            rc = plan_tile_skip(p, pl, ws, st, opt);
            if (rc) return rc;
        }
    }

    if (p.kind == KIND_BIH2D) {                       // x-uniform coefficient rows -> scalar loads
        pl.umask = 0;
        if (!(opt.flags & XINV_FLAG_NO_XUNIFORM)) {
            rc = detect_xuniform(ws, st, p.c, p.sc, 10, p.nbatch, p.yc, p.xc, &pl.umask);
            if (rc) return rc;
        }
        pl.um = pl.umask;
    }

    // ---- workspace ---------------------------------------------------------------------------
    rc = ensure_dev(&ws->ctl, &ws->ctl_cap, (size_t)p.nbatch * sizeof(XinvCtl));
    if (rc) return rc;
    if (ws->hctl_cap < (size_t)p.nbatch) {             // two slots: polling is pipelined
        if (ws->hctl) HIPCHK(hipHostFree(ws->hctl));
        HIPCHK(hipHostMalloc((void **)&ws->hctl, 2 * (size_t)p.nbatch * sizeof(XinvCtl), hipHostMallocDefault));
        ws->hctl_cap = (size_t)p.nbatch;
    }
    size_t pbytes;
    if (pl.path == XINV_PATH_FUSED)
        pbytes = (size_t)p.nbatch * XINV_KMAX * (p.kind == KIND_STD3D ? (size_t)pl.nsg * pl.nrb : (size_t)pl.nsg) *
                 (sizeof(double) + sizeof(long long));
    else
        pbytes = (size_t)p.nbatch * XINV_NORM_BLOCKS * (sizeof(double) + sizeof(long long));
    rc = ensure_dev(&ws->partials, &ws->partials_cap, pbytes);
    if (rc) return rc;
    double *S2 = nullptr;
    if (pl.path == XINV_PATH_FUSED) {
        const size_t need = (size_t)((p.nbatch - 1) * p.sS + n) * sizeof(double);
        rc = ensure_dev(&ws->S2, &ws->S2_cap, need);
        if (rc) return rc;
        S2 = ws->S2;
        if (pl.skip)            // skipped tiles are never written: both buffers start from the caller's S
            HIPCHK(hipMemcpyAsync(S2, p.S, need, hipMemcpyDeviceToDevice, st));
    }

    hipLaunchKernelGGL(k_ctl_init, dim3(cdiv(p.nbatch, 256)), dim3(256), 0, st, ws->ctl, p.nbatch);

    // ---- sweep loop ----------------------------------------------------------------------------
    const int64_t max_sweeps = p.stop.mxLoop + 1;       // numbas.py:410: loop >= mxLoop stops
    const int Kf = (pl.path == XINV_PATH_FUSED) ? pl.K : 1;
    int check_every = opt.check_every;
    if (check_every <= 0) {
        // poll the device stop flags about every 2 ms of sweeping (fused kernels run at roughly
        // 2e5 points per microsecond, the colour path at a quarter of that); launches issued after
        // a member has stopped are no-ops of a few microseconds each
        const double rate = (pl.path == XINV_PATH_FUSED) ? 2.0e5 : 4.0e4;
        const double est_us = std::max(4.0, (double)p.nbatch * (double)n * Kf / rate);
        check_every = (int)std::min(256.0, std::max(4.0, 2000.0 / est_us));
    }
    double *buf[2] = { p.S, S2 };
    std::vector<int64_t> bound;                          // bound[i] = sweeps before launch i
    int64_t launched = 0;
    bool all_done = false;
    double ms_total = 0.0;
    int64_t nlaunch = 0;
    // A chunk = `check_every` launches followed by an asynchronous copy of the control blocks.
    // Polling is pipelined: chunk c+1 is queued BEFORE the host waits for chunk c's copy, so the
    // GPU never idles on the host's reaction time; once every member has stopped, the launches
    // already queued are no-ops (each kernel returns on ctl.done).
    auto issue_chunk = [&](int slot) -> int {
        if (opt.timing) HIPCHK(hipEventRecord(ws->ev0[slot], st));
        for (int i = 0; i < check_every && launched < max_sweeps; i++) {
            int r;
            if (pl.path == XINV_PATH_FUSED) {
                const int k = (max_sweeps - launched >= Kf) ? Kf : 1;
                const int cur = (int)(bound.size() & 1);
                r = (p.kind == KIND_STD3D)
                        ? launch_fused3d(p, pl, buf[cur], buf[cur ^ 1], ws, st, 0, p.nbatch, 0, 0)
                    : pl.nine
                        ? launch_fused9(p, pl, k, buf[cur], buf[cur ^ 1], ws, st, 0, p.nbatch, 0, 0)
                        : launch_fused(p, pl, k, buf[cur], buf[cur ^ 1], ws, st, 0, p.nbatch, 0, 0);
                if (r) return r;
                bound.push_back(launched);
                launched += k;
            } else {
                r = launch_colour_sweep(p, pl, ws, st);
                if (r) return r;
                launched += 1;
            }
            nlaunch++;
        }
        if (opt.timing) HIPCHK(hipEventRecord(ws->ev1[slot], st));
        HIPCHK(hipMemcpyAsync(ws->hctl + (size_t)slot * p.nbatch, ws->ctl, (size_t)p.nbatch * sizeof(XinvCtl),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(ws->evc[slot], st));
        return XINV_OK;
    };
    const XinvCtl *hc = ws->hctl;                        // the slot holding the final control blocks
    rc = issue_chunk(0);
    if (rc) return rc;
    for (int c = 0;; c++) {
        const int slot = c & 1;
        const bool more = launched < max_sweeps;
        if (more) { rc = issue_chunk(slot ^ 1); if (rc) return rc; }
        HIPCHK(hipEventSynchronize(ws->evc[slot]));
        if (opt.timing) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ws->ev0[slot], ws->ev1[slot]));
            ms_total += ms;
        }
        hc = ws->hctl + (size_t)slot * p.nbatch;
        all_done = true;
        for (int64_t m = 0; m < p.nbatch; m++) all_done = all_done && hc[m].done;
        if (all_done || !more) break;
    }
    if (pl.path != XINV_PATH_FUSED)                      // drain the queued no-op tail (the fused path syncs below)
        HIPCHK(hipStreamSynchronize(st));
    if (!all_done) { t_err = "internal: sweep budget exhausted before the stop rule fired"; return XINV_ERR_HIP; }

    // ---- fused path: put each member's final state into S ------------------------------------
    int64_t sweeps_max = 0;
    if (pl.path == XINV_PATH_FUSED) {
        bound.push_back(launched);
        for (int64_t m = 0; m < p.nbatch; m++) {
            const int64_t sw = hc[m].sweeps;
            // launch i covers sweeps (bound[i], bound[i+1]]; find the one holding sweep `sw`
            size_t i = std::upper_bound(bound.begin(), bound.end(), sw - 1) - bound.begin() - 1;
            int where;                                   // buffer index holding the final state
            if (bound[i + 1] == sw) {
                where = (int)((i + 1) & 1);
            } else {                                     // stopped inside a K-sweep launch: redo
                int cur = (int)(i & 1);
                for (int64_t s = bound[i]; s < sw; s++) {
                    rc = (p.kind == KIND_STD3D)
                             ? launch_fused3d(p, pl, buf[cur], buf[cur ^ 1], ws, st, m, 1, 1, 1)
                         : pl.nine
                             ? launch_fused9(p, pl, 1, buf[cur], buf[cur ^ 1], ws, st, m, 1, 1, 1)
                             : launch_fused(p, pl, 1, buf[cur], buf[cur ^ 1], ws, st, m, 1, 1, 1);
                    if (rc) return rc;
                    cur ^= 1;
                }
                where = cur;
            }
            if (where == 1)
                HIPCHK(hipMemcpyAsync(p.S + m * p.sS, S2 + m * p.sS, (size_t)n * sizeof(double),
                                      hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    for (int64_t m = 0; m < p.nbatch; m++) {
        const XinvCtl &c = hc[m];
        if (c.overflow) flags[3 * m + 0] = 1.0;
        if (c.wrote) { flags[3 * m + 1] = c.flag1; flags[3 * m + 2] = c.flag2; }
        sweeps_max = std::max<int64_t>(sweeps_max, c.sweeps);
    }
    t_stats.path = pl.path;
    t_stats.colours = pl.ncol;
    t_stats.sweeps_per_launch = Kf;
    t_stats.rows_per_tile = pl.RY;
    t_stats.xuniform_mask = (pl.path == XINV_PATH_FUSED || p.kind == KIND_BIH2D) ? (int32_t)pl.um : 0;
    t_stats.masked_tile_pct = (pl.path == XINV_PATH_FUSED && pl.skip) ? pl.skip_pct : 0;
    t_stats.sweep_launches = nlaunch;
    t_stats.sweeps_max = sweeps_max;
    t_stats.sweep_ms = ms_total;
    return XINV_OK;
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

struct Pinned {                                     // host ranges registered for this call
    std::vector<void *> regs;
    bool try_pin(const void *h, size_t bytes)
    {
        if (bytes < (1u << 20)) return false;
        if (hipHostRegister((void *)h, bytes, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        regs.push_back((void *)h);
        return true;
    }
    ~Pinned() { for (void *h : regs) (void)hipHostUnregister(h); }
};

static int upload(DevPool *pool, Pinned &pin, hipStream_t st, const double *h, int64_t nbatch,
                  int64_t stride, int64_t n, double **out, int64_t *dstride)
{
    if (!h) { *out = nullptr; *dstride = 0; return XINV_OK; }
    const int64_t members = (stride == 0) ? 1 : nbatch;
    double *d = nullptr;
    int rc = pool_alloc(pool, (size_t)members * n * sizeof(double), &d);
    if (rc) return rc;
    if (members == 1 || stride == n) {
        const size_t bytes = (size_t)members * n * sizeof(double);
        pin.try_pin(h, bytes);
        HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    } else {
        pin.try_pin(h, (size_t)((members - 1) * stride + n) * sizeof(double));
        for (int64_t m = 0; m < members; m++)
            HIPCHK(hipMemcpyAsync(d + m * n, h + m * stride, (size_t)n * sizeof(double),
                                  hipMemcpyHostToDevice, st));
    }
    *out = d;
    *dstride = (stride == 0) ? 0 : n;
    return XINV_OK;
}

static int solve_host(Problem &p, double *flags, const xinv_options *opt)
{
    p.rowconst = opt ? ((unsigned)opt->rowconst_mask & ((1u << p.ncoef) - 1u)) : 0u;
    int rc = validate(p, flags);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        t_err = "no HIP device available";
        return XINV_ERR_NODEV;
    }
    if (opt && opt->device >= 0) HIPCHK(hipSetDevice(opt->device));
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    const int64_t n = p.zc * p.yc * p.xc;
    // the staging pool and the solver workspace are per device: hold the device for the whole
    // upload -> solve -> download sequence
    std::lock_guard<std::recursive_mutex> host_lock(get_ws(device)->busy);
    DevPool *pool = get_pool(device);
    pool->reset();
    Pinned pin;
    Problem d = p;
    double *hS = p.S;
    const int64_t hsS = p.nbatch > 1 ? p.sS : n;
    hipStream_t st = 0;
    struct Events {                                   // destroyed on every return path
        hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
        ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (auto &x : ev.e) HIPCHK(hipEventCreate(&x));
    hipEvent_t e0 = ev.e[0], e1 = ev.e[1], e2 = ev.e[2], e3 = ev.e[3];
    HIPCHK(hipEventRecord(e0, st));
    int64_t ds;
    rc = upload(pool, pin, st, p.S, p.nbatch, hsS, n, &d.S, &ds);
    if (rc) return rc;
    d.sS = n;
    for (int q = 0; q < p.ncoef; q++) {
        double *dc;
        if (((p.rowconst >> q) & 1u) && p.c[q]) {        // one value per row: upload rows, expand on the device
            const int64_t rows = p.zc * p.yc;
            double *drow; int64_t rstride;
            rc = upload(pool, pin, st, p.c[q], p.nbatch, p.nbatch > 1 ? p.sc[q] : 0, rows, &drow, &rstride);
            if (rc) return rc;
            const int64_t members = (rstride == 0) ? 1 : p.nbatch;
            rc = pool_alloc(pool, (size_t)members * n * sizeof(double), &dc);
            if (rc) return rc;
            hipLaunchKernelGGL(k_expand_rows, dim3(cdiv(rows * members, 4)), dim3(256), 0, st,
                               (const double *)drow, dc, rows, p.xc, members);
            d.sc[q] = (rstride == 0) ? 0 : n;
        } else {
            rc = upload(pool, pin, st, p.c[q], p.nbatch, p.nbatch > 1 ? p.sc[q] : 0, n, &dc, &d.sc[q]);
            if (rc) return rc;
        }
        d.c[q] = dc;
    }
    d.rowconst = 0;
    HIPCHK(hipEventRecord(e1, st));
    rc = solve_dev(d, flags, opt, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e2, st));
    if (p.nbatch == 1 || hsS == n) {
        HIPCHK(hipMemcpyAsync(hS, d.S, (size_t)p.nbatch * n * sizeof(double), hipMemcpyDeviceToHost, st));
    } else {
        for (int64_t m = 0; m < p.nbatch; m++)
            HIPCHK(hipMemcpyAsync(hS + m * hsS, d.S + m * n, (size_t)n * sizeof(double),
                                  hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipEventRecord(e3, st));
    HIPCHK(hipEventSynchronize(e3));
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventElapsedTime(&a, e0, e1));
    HIPCHK(hipEventElapsedTime(&b, e2, e3));
    t_stats.h2d_ms = a; t_stats.d2h_ms = b;
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
