/*
 * xinv.h -- C-ABI of the MI355X SOR inversion engine (libxinv_hip.so).
 *
 * Drop-in boundary for the reference's hot path: every entry point replaces one call site of
 * the numba kernels in the reference's xinvert/core.py (file:line cited per function, relative
 * to the reference tree).  Plain pointers and sizes only; no C++/torch types cross this line.
 * Implemented in xinvert_amd/csrc/xinv_hip.hip (hand-written HIP for gfx950).
 *
 * Conventions
 *   - arrays are C-contiguous float64, x fastest: [yc][xc] or [zc][yc][xc]; a batch is
 *     [nbatch] of those with a per-array batch stride in ELEMENTS (0 = one copy shared by all
 *     members, kept once in HBM);
 *   - S is read (initial guess / boundary values) and written in place, as the reference does;
 *   - BC codes: XINV_BC_FIXED 0, XINV_BC_EXTEND 1, XINV_BC_PERIODIC 2 (the reference passes the
 *     strings 'fixed' / 'extend' / 'periodic'; 'extend' on x or z is a no-op there and here);
 *   - flags[3] per member = {overflow, last relative change of mean|S|, last loop index}
 *     (reference numbas.py:401-414): flags[0] is written only on overflow, sweeps executed =
 *     flags[2] + 1;
 *   - return 0 on success, negative XINV_ERR_* otherwise; nothing throws across the ABI.
 *
 * Sweep ordering.  The reference sweeps lexicographically (serial Gauss-Seidel).  The engine
 * sweeps red-black on (j+i)&1 when the cross coefficient B is identically zero and 4-colour on
 * (j&1, i&1) otherwise (3-D: (k+j+i)&1; biharmonic: 9 colours (j%3, i%3)); with periodic x and odd xc
 * the last column is its own pair of colours, each run right after the base colour it belongs to.  Point arithmetic, masking predicate, 'extend' pre-pass, norm (mean |S| over
 * S != undef) and the stopping rule are the reference's.
 */
#ifndef XINV_H
#define XINV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XINV_BC_FIXED    0
#define XINV_BC_EXTEND   1
#define XINV_BC_PERIODIC 2

#define XINV_OK          0
#define XINV_ERR_ARG    -1   /* bad argument (sizes < 3, null pointer, unknown BC code, ...) */
#define XINV_ERR_HIP    -2   /* a HIP runtime call failed; xinv_last_error() has the text     */
#define XINV_ERR_NODEV  -3   /* no usable GPU                                                   */
#define XINV_ERR_NOMEM  -4   /* device allocation failed                                        */

/* kernel paths (xinv_options.path / xinv_stats.path) */
#define XINV_PATH_AUTO   0
#define XINV_PATH_COLOUR 1   /* one launch per colour, in place (general fallback)              */
#define XINV_PATH_FUSED  2   /* streaming kernels: a whole sweep (or two) per pass, ping-pong buffers */
/* (3 was XINV_PATH_SMALL, a register-resident solver for small slices: removed in version 400, it
   never beat the streaming kernels on the shapes it was built for -- DESIGN.md 4.7) */

#define XINV_FLAG_NO_XUNIFORM 1  /* stream every coefficient array in full: do not look for rows
                                    that are constant along x                                    */
#define XINV_FLAG_NO_TILE_SKIP 2 /* run every tile even where the forcing is masked throughout    */
#define XINV_FLAG_FORCE_TILE_SKIP 4 /* testing aid: skip masked tiles whatever the grid size, keep
                                    the row split                                                */
#define XINV_FLAG_FMA 32         /* OPT-IN contracted arithmetic: the point update with fused multiply-adds at fixed
                                    positions (about a fifth fewer vector instructions).  NOT the reference's
                                    arithmetic -- numba evaluates the update without contraction, and so does the
                                    default path, bit for bit -- but within 1e-12 (relative) of it after tens of
                                    sweeps and within north_star's 1e-6 rel-L2 at convergence; the oracle restates
                                    it (XO_FMA) and the kernels with the flag are bitwise THAT.  Available for the
                                    per-row-coefficient variants of the streaming kernels (lat-lon Poisson,
                                    Gill-Matsuno, omega); XINV_ERR_ARG elsewhere.                              */
#define XINV_FLAG_PIN_HOST 8     /* host-pointer entries: register the caller's arrays in place for the
                                    call (DMA at PCIe rate).  Off by default: only for buffers that stay
                                    mapped afterwards (see xinv_host.h)                           */

#define XINV_FLAG_NO_PIPE 16     /* standard form with per-row A and C: keep the four sweeps of a pass inside
                                    one wavefront (k_fused2d) instead of pipelining them across the four
                                    wavefronts of a workgroup (k_pipe2d)                          */

#define XINV_FLAG_NO_POINT_FACTOR 64 /* general 2-D form with coefficients varying along x: divide and test the operands in
                                    the kernel whenever a row enters a window, as rounds 1-4 did, instead of reading the
                                    point's relaxation factor / update predicate from a stream evaluated once per
                                    coefficient stack (same expression, same bits)                                    */

#define XINV_MAX_DEVICES 16

typedef struct xinv_options {
    int32_t device;             /* HIP device ordinal; -1 = current device                      */
    int32_t path;               /* XINV_PATH_*                                                   */
    int32_t sweeps_per_launch;  /* fused path: sweeps fused in one launch (1..4, capped at what
                                   the kernel variant supports); 0 = auto                      */
    int32_t check_every;        /* launches between host polls of the device stop flags; 0=auto */
    int32_t rows_per_tile;      /* fused 2-D: rows per tile (n > 0) or exactly -n evenly split row
                                   blocks (n < 0); biharmonic one-pass kernel: rows per block (multiple
                                   of 3); fused 3-D: rows per workgroup (8, 12, 16); 0 = auto            */
    int32_t timing;             /* 1: bracket launch chunks with HIP events (xinv_last_stats)   */
    int32_t flags;              /* XINV_FLAG_* bits                                              */
    int32_t rowconst_mask;      /* host-pointer entries: bit q set = coefficient array q (argument order
                                   without S: A = bit 0) holds ONE value per row, [rows] per member with
                                   rows = yc (zc*yc in 3-D); its batch stride is 0 or >= rows.  The rows
                                   are uploaded and expanded on the device: lat-lon coefficients are
                                   functions of latitude only (apps.py:1406-1408, 1630-1635)             */
    int32_t host_chunk;         /* host-pointer entries: members per upload/solve/download chunk (the
                                   three overlap, chunk c+1 travelling while chunk c sweeps); 0 = auto  */
    int32_t ndev;               /* host-pointer batched entries: 0 = one device (`device`); n > 0 = split
                                   the batch axis in contiguous blocks over device_ids[0..n-1] (one host
                                   thread per GPU, no collective: slices are independent, reference
                                   core.py:129-139); -1 = every visible GPU.  Ignored by *_dev entries.  */
    int32_t device_ids[XINV_MAX_DEVICES];
    /* Front-end passes done on the device by the host-pointer batched entries (ignored by *_dev):
       what apps.__mask_FS, the builders' `F * cos(lat)` + re-mask and the output de-mask do with
       numpy in the reference (apps.py:2112-2159, 1409-1411, 1389-1392).                              */
    int32_t prep_flags;         /* XINV_PREP_* bits                                                  */
    int32_t f32_mask;           /* host-pointer entries: bit 0 = S, bit q+1 = coefficient array q (A = bit 1) is FLOAT32 on
                                   the host: the pointer (declared double*) points to floats, its batch stride counts
                                   floats.  The array travels as float32 -- half the bytes over PCIe -- and is promoted
                                   on the device (exact: the same float64 values a host-side promotion gives); S with
                                   bit 0 is also written back as float32 (the float64 result rounded to nearest, as
                                   an assignment into a float32 array does).  The reference's own datasets are float32
                                   (tests/test_Poisson.py:14-24).  Ignored by the *_dev entries.                    */
    double  prep_undef;         /* XINV_PREP_MASK_VALUE: the caller's undefined value                 */
    double  demask_value;       /* XINV_PREP_DEMASK: written to S where the forcing is masked         */
    const double *prep_rowscale;/* XINV_PREP_ROWSCALE: [yc] host doubles, defined forcing values of
                                   row j (of every plane / member) are multiplied by prep_rowscale[j] */
    /* Expert overrides of the planner's own choices (0 = the planner decides).  The library reads NO environment
       variable: what used to be XINV_LANES / XINV_LAG / XINV_PIPE_FR / XINV_GRAPH in rounds 2-4 are these fields.  */
    int32_t lanes;              /* device entries: independent launch chains the batch is cut into (n > 0: exactly
                                   min(n, nbatch, 4) chains; 1 = one chain)                                          */
    int32_t norm_lag;           /* 2-D streaming kernels: -1 = the launch's own last workgroup reduces the norm
                                   (no third S buffer); 1 / 0 = evaluated one pass behind where the planner sees a gain  */
    int32_t pipe_fr;            /* wave-pipelined pass: the forcing rides the LDS ring (1) or is read by every
                                   wavefront (-1); 0 = by the size of the launch's streams                           */
    int32_t graph;              /* colour path: replay chunks of launches from a hipGraph (1) or launch one by one (-1) */
    int32_t cu_count;           /* compute units the planner fills (0 = the device's own count).  The two-sweep 3-D pass
                                   runs one workgroup per CU: the tiles of a launch's last, partly filled round are cut
                                   into k chunks so that they do not march alone (xinv_pipe3d.h).  A smaller count makes
                                   small test problems take that path; it never changes a result.  -1: the device's count,
                                   without that cut (A/B comparisons); -n: n units, without it.                        */
    int32_t host_inflight;      /* host-pointer entries: chunk solves in flight on a device at a time (1 .. 6; 0 = the
                                   library's choice -- two for the 2-D forms, three for the 3-D ones): every one is a chain of dependent launches on a stream and a
                                   workspace of its own; their launches fill each other's tails.  The standard 3-D form with
                                   shared coefficient arrays runs a ROLLING batch instead where it is left at 0 (and
                                   host_chunk is): one chain of launches over the members that have arrived and are
                                   not done yet (xinv_hostptr.h); -1 takes the rolling batch for any batch of two or more */
} xinv_options;

#define XINV_PREP_MASK_NAN   1  /* the forcing (last coefficient array) marks masked points with NaN   */
#define XINV_PREP_MASK_VALUE 2  /* ... with prep_undef                                                 */
#define XINV_PREP_ROWSCALE   4
#define XINV_PREP_S_ZERO     8  /* the initial guess is zero: S is not read from the host              */
#define XINV_PREP_DEMASK    16

typedef struct xinv_stats {
    int32_t path;               /* path actually used                                           */
    int32_t colours;            /* colours per sweep (2, 4, +2 with the odd-periodic seam; biharmonic 9+) */
    int32_t sweeps_per_launch;
    int32_t rows_per_tile;
    int32_t xuniform_mask;      /* fused path: coefficient streams read as one scalar per row    */
    int32_t masked_tile_pct;    /* fused 2-D path: share of tiles skipped because fully masked   */
    int64_t sweep_launches;     /* sweep passes issued (incl. no-op tail launches); one kernel launch per lane each */
    int64_t sweeps_max;         /* max over members of sweeps executed                          */
    double  sweep_ms;           /* HIP-event time over all launch chunks (timing=1), ms         */
    double  h2d_ms, d2h_ms;     /* host-pointer entry points only: span of the upload / download
                                   streams (they overlap the sweeps when the batch has chunks)  */
    double  wall_ms;            /* host-pointer entry points: wall clock of the whole call      */
    int32_t host_chunks;        /* member chunks the call was pipelined over (all devices)      */
    int32_t devices;            /* GPUs the batch was split over                                */
    int32_t pipelined;          /* fused 2-D path: the full passes ran the wave-pipelined kernel (k_pipe2d:
                                   one tile per workgroup, one sweep per wavefront); value = column pairs
                                   per lane (1 or 2), 0 = k_fused2d                              */
    int32_t masked_tile_ppm;    /* masked_tile_pct at full resolution: skipped wave-tiles per million (bench.py prices
                                   its roofline on the tiles that ran)                           */
    int32_t recovered_members;  /* members whose in-kernel norm reduction timed out (watchdog) and that were finished
                                   sweep by sweep with the separate norm kernels; 0 in every run seen so far outside the
                                   test-hooks build of the library                                   */
    int32_t lanes;              /* device entries: independent launch chains the batch was cut into (1 or 2; each
                                   `sweep_launches` pass is then one kernel launch per lane, on its own stream, and
                                   kernel durations in a trace overlap)                                              */
    int32_t planned;            /* 1: the solve ran on a resident plan (xinv_plan_*): no detection / planning pass      */
    int32_t point_factor;       /* general 2-D form, coefficients varying along x: 1 = the sweeps read the relaxation factor /
                                   update predicate of every point from the stream k_point_factor evaluated once per
                                   coefficient stack; 2 = ... and C out of A (the two hold the same numbers); 0 = in-kernel.
                                   Biharmonic one-pass kernel: 0 = A..I per row (records); 1 = A, C, D, F as vector streams
                                   + the point-factor stream; 3 = ... with C read out of A and F out of D (the pairs hold the same
                                   numbers: Cartesian Munk); 2 = all nine coefficient arrays as vector streams + it        */
    double  plan_ms;            /* wall clock of the planning part of the call (detection passes, host round trips,
                                   tile lists, per-row records); ~0 for a solve on a resident plan                    */
    int32_t k_chunks;           /* two-sweep 3-D pass: chunks the plan cuts a tile's column into where it cuts (1: never)    */
    int32_t cut_tiles;          /* ... tiles of the solve's first sweep launch that were cut (the remainder of its last round
                                   of one workgroup per CU, or every tile of a small batch); the others march whole         */
    int32_t rolling;            /* host-pointer entries: 1 = the batch ran as a rolling batch (one launch chain over the members
                                   that had arrived and were not done: xinv_hostptr.h), 0 = chunk solves                    */
    int32_t reserved_;
    double  launch_us_min, launch_us_avg, launch_us_max;   /* timing = 2: every sweep launch bracketed by its own pair
                                   of HIP events on the stream it runs on (one lane only): per-launch durations without
                                   a profiler's per-dispatch overhead                                                  */
} xinv_stats;

void        xinv_default_options(xinv_options *opt);
int         xinv_last_stats(xinv_stats *out);       /* stats of the calling thread's last solve */
const char *xinv_last_error(void);
int         xinv_device_count(void);
int         xinv_version(void);
/* sizeof(xinv_options) and sizeof(xinv_stats) as this build of the library sees them: a binding compares them with its
 * own declarations before the first call (xinvert_amd/_lib.py does; a stale mirror would otherwise corrupt memory). */
void        xinv_abi_sizes(int32_t *options_bytes, int32_t *stats_bytes);

/* ---- single slice, HOST pointers: positional twins of the numba kernels -------------------
 * xinv_standard_2d_f64  replaces numbas.invert_standard_2D  called at core.py:130-139
 * xinv_general_2d_f64   replaces numbas.invert_general_2D   called at core.py:419-428
 * xinv_standard_3d_f64  replaces numbas.invert_standard_3D  called at core.py:60-69
 * Upload, solve on the current device, download S and flags. */
int xinv_standard_2d_f64(double *S, const double *A, const double *B, const double *C,
                         const double *F, int64_t yc, int64_t xc, double dely, double delx,
                         int BCy, int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                         double optArg, double undef, double *flags, int64_t mxLoop,
                         double tolerance);

int xinv_general_2d_f64(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, const double *G,
                        int64_t yc, int64_t xc, double dely, double delx, int BCy, int BCx,
                        double delxSqr, double ratio, double ratioQtr, double ratioSqr,
                        double optArg, double undef, double *flags, int64_t mxLoop,
                        double tolerance);

int xinv_standard_3d_f64(double *S, const double *A, const double *B, const double *C,
                         const double *F, int64_t zc, int64_t yc, int64_t xc, double delz,
                         double dely, double delx, int BCz, int BCy, int BCx, double delxSqr,
                         double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                         double *flags, int64_t mxLoop, double tolerance);

/* ---- batched over the outer (time / level / member) axis ----------------------------------
 * The loop `for selDict in loop_noncore(F, dims)` of core.py:129 / 418 / 59 becomes ONE call.
 * strides[]: batch stride in elements for each array, in argument order
 *            (2-D standard / 3-D: S,A,B,C,F ; 2-D general: S,A,B,C,D,E,F,G); 0 = shared.
 * flags:     [nbatch*3], caller-initialised (the reference passes {0,1,0}).
 * opt:       may be NULL (defaults).
 * *_batched  take HOST pointers (staged through pinned buffers, coefficients with stride 0
 *            uploaded once); *_dev take DEVICE pointers already resident in HBM on opt->device
 *            and run on `stream` (a hipStream_t, NULL = default stream; a batch may be split over
 *            `stream` and an engine-owned stream forked from / joined back into it: xinv_stats.lanes);
 *            flags stays a HOST pointer.  Both return after the solve has completed. */
int xinv_standard_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                 const double *F, int64_t nbatch, const int64_t *strides,
                                 int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                 int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                                 double optArg, double undef, double *flags, int64_t mxLoop,
                                 double tolerance, const xinv_options *opt);

int xinv_general_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, int64_t nbatch, const int64_t *strides,
                                int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                int BCx, double delxSqr, double ratio, double ratioQtr,
                                double ratioSqr, double optArg, double undef, double *flags,
                                int64_t mxLoop, double tolerance, const xinv_options *opt);

int xinv_standard_3d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                 const double *F, int64_t nbatch, const int64_t *strides,
                                 int64_t zc, int64_t yc, int64_t xc, double delz, double dely,
                                 double delx, int BCz, int BCy, int BCx, double delxSqr,
                                 double ratio2Sqr, double ratio1Sqr, double optArg,
                                 double undef, double *flags, int64_t mxLoop, double tolerance,
                                 const xinv_options *opt);

int xinv_standard_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t nbatch, const int64_t *strides,
                             int64_t yc, int64_t xc, double dely, double delx, int BCy,
                             int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                             double optArg, double undef, double *flags, int64_t mxLoop,
                             double tolerance, const xinv_options *opt, void *stream);

int xinv_general_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc,
                            double dely, double delx, int BCy, int BCx, double delxSqr,
                            double ratio, double ratioQtr, double ratioSqr, double optArg,
                            double undef, double *flags, int64_t mxLoop, double tolerance,
                            const xinv_options *opt, void *stream);

int xinv_standard_3d_f64_dev(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t nbatch, const int64_t *strides,
                             int64_t zc, int64_t yc, int64_t xc, double delz, double dely,
                             double delx, int BCz, int BCy, int BCx, double delxSqr,
                             double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                             double *flags, int64_t mxLoop, double tolerance,
                             const xinv_options *opt, void *stream);

/* ---- general 3-D form (3DOcean), SURVEY 8(f) rank 4 -----------------------------------------
 * xinv_general_3d_f64 replaces numbas.invert_general_3D called at core.py:345-356.
 * strides[]: S,A,B,C,D,E,F,G,H.  7-point stencil, red-black on (k+j+i)&1; streaming kernel k_fused3dg when A..G are
 * constant along x (every 3DOcean coefficient), colour-pass kernels otherwise.
 * The reference's west-periodic branch never tests the forcing H against undef
 * (numbas.py:849-852) -- kept.  BCz is accepted and never read. */
int xinv_general_3d_f64(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, const double *G,
                        const double *H, int64_t zc, int64_t yc, int64_t xc, double delz,
                        double dely, double delx, int BCz, int BCy, int BCx, double delxSqr,
                        double ratio2, double ratio1, double ratio2Sqr, double ratio1Sqr,
                        double optArg, double undef, double *flags, int64_t mxLoop,
                        double tolerance);

int xinv_general_3d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, const double *H, int64_t nbatch,
                                const int64_t *strides, int64_t zc, int64_t yc, int64_t xc,
                                double delz, double dely, double delx, int BCz, int BCy, int BCx,
                                double delxSqr, double ratio2, double ratio1, double ratio2Sqr,
                                double ratio1Sqr, double optArg, double undef, double *flags,
                                int64_t mxLoop, double tolerance, const xinv_options *opt);

int xinv_general_3d_f64_dev(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            const double *H, int64_t nbatch, const int64_t *strides, int64_t zc,
                            int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                            int BCy, int BCx, double delxSqr, double ratio2, double ratio1,
                            double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                            double *flags, int64_t mxLoop, double tolerance,
                            const xinv_options *opt, void *stream);

/* ---- biharmonic 2-D form (Munk / Stommel-Munk), SURVEY 8(f) rank 1 ----------------------------
 * xinv_general_bih_2d_f64 replaces numbas.invert_general_bih_2D called at core.py:503-516.
 * strides[]: S,A,B,C,D,E,F,G,H,I,J.  Radius-2 stencil, 9 colours (j%3, i%3) (+3 per trailing
 * column when x is periodic and xc % 3 != 0); one-pass streaming kernel k_fusedbih when A..I are constant along x,
 * one launch per row class otherwise (colour launches for periodic x with xc % 3 != 0); yc >= 5, xc >= 7.  The
 * reference's east-periodic branches index the B term with a stale loop variable
 * (numbas.py:1495-1497, 1540-1542); that behaviour is reproduced. */
int xinv_general_bih_2d_f64(double *S, const double *A, const double *B, const double *C,
                            const double *D, const double *E, const double *F, const double *G,
                            const double *H, const double *I, const double *J, int64_t yc,
                            int64_t xc, double dely, double delx, int BCy, int BCx,
                            double delxSSr, double delxTr, double delxSqr, double ratio,
                            double ratioSSr, double ratioQtr, double ratioSqr, double optArg,
                            double undef, double *flags, int64_t mxLoop, double tolerance);

int xinv_general_bih_2d_f64_batched(double *S, const double *A, const double *B, const double *C,
                                    const double *D, const double *E, const double *F,
                                    const double *G, const double *H, const double *I,
                                    const double *J, int64_t nbatch, const int64_t *strides,
                                    int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                    int BCx, double delxSSr, double delxTr, double delxSqr,
                                    double ratio, double ratioSSr, double ratioQtr,
                                    double ratioSqr, double optArg, double undef, double *flags,
                                    int64_t mxLoop, double tolerance, const xinv_options *opt);

int xinv_general_bih_2d_f64_dev(double *S, const double *A, const double *B, const double *C,
                                const double *D, const double *E, const double *F,
                                const double *G, const double *H, const double *I,
                                const double *J, int64_t nbatch, const int64_t *strides,
                                int64_t yc, int64_t xc, double dely, double delx, int BCy,
                                int BCx, double delxSSr, double delxTr, double delxSqr,
                                double ratio, double ratioSSr, double ratioQtr, double ratioSqr,
                                double optArg, double undef, double *flags, int64_t mxLoop,
                                double tolerance, const xinv_options *opt, void *stream);

/* ---- standard 2-D "test" form (Fofonoff, Bretherton-Haidvogel), SURVEY 8(f) rank 2 -------------
 * xinv_standard_2d_test_f64 replaces numbas.invert_standard_2D_test called at core.py:205-215:
 * d/dy(A Sy + B Sx) + d/dx(C Sy + D Sx) + E S = F.  strides[]: S,A,B,C,D,E,F.  Red-black fused
 * kernels when B and C are identically zero, 4-colour passes otherwise. */
int xinv_standard_2d_test_f64(double *S, const double *A, const double *B, const double *C,
                              const double *D, const double *E, const double *F, int64_t yc,
                              int64_t xc, double dely, double delx, int BCy, int BCx,
                              double delxSqr, double ratioQtr, double ratioSqr, double optArg,
                              double undef, double *flags, int64_t mxLoop, double tolerance);

int xinv_standard_2d_test_f64_batched(double *S, const double *A, const double *B, const double *C,
                                      const double *D, const double *E, const double *F,
                                      int64_t nbatch, const int64_t *strides, int64_t yc,
                                      int64_t xc, double dely, double delx, int BCy, int BCx,
                                      double delxSqr, double ratioQtr, double ratioSqr,
                                      double optArg, double undef, double *flags, int64_t mxLoop,
                                      double tolerance, const xinv_options *opt);

int xinv_standard_2d_test_f64_dev(double *S, const double *A, const double *B, const double *C,
                                  const double *D, const double *E, const double *F,
                                  int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc,
                                  double dely, double delx, int BCy, int BCx, double delxSqr,
                                  double ratioQtr, double ratioSqr, double optArg, double undef,
                                  double *flags, int64_t mxLoop, double tolerance,
                                  const xinv_options *opt, void *stream);

/* ---- resident plans: what a solve derives from the coefficient stack, built once -------------------------------
 * The reference calls its kernel again and again on ONE coefficient stack: apps.animate_iteration (apps.py:1031-1044:
 * `invt_func(*coeffs, maskF, initS, dims, iParams)` once per frame, tests/test_AnimateConverge.py:13-31: 40 frames of
 * 2 sweeps on 73x144), the restart of an un-converged solve, another first guess.  Every call of the *_dev entries above
 * re-derives what depends on that stack alone -- is B zero, which arrays are constant along x, the per-row records
 * (coefficients, relaxation factor, row predicate), the forcing's activity map, the row split and the lists of tiles
 * that hold a defined point: passes over the arrays, host round trips and host planning, ~0.26 ms of a 4.8 ms solve at
 * 3600x1800 and most of a two-sweep frame.  A plan holds all of it in HBM, next to the coefficient stack it describes.
 *
 * xinv_plan_create_<form>_f64_dev: the arguments of xinv_<form>_f64_dev without S, flags, mxLoop and tolerance (DEVICE
 *   pointers; strides[0] = batch stride of the S arrays the plan will be solved on); `opt` as for the *_dev entries
 *   (device, path, sweeps_per_launch, rows_per_tile, flags, timing, lanes, ...), and additionally opt->rowconst_mask:
 *   bit q set = coefficient array q holds ONE value per row ([rows] per member, rows = yc or zc*yc, batch stride 0 or
 *   exactly rows) -- lat-lon coefficients are functions of latitude (apps.py:1406-1408, 1630-1635) --, which the plan
 *   expands into HBM copies of its own (the caller's row vectors are not referenced after the call returns) and never
 *   has to test for uniformity.  Synchronous: returns with the plan complete.
 * xinv_plan_solve_f64_dev: one call of the hot path on the plan -- S (device pointer, read and written in place, 16-byte
 *   aligned), flags (host, [nbatch*3], caller-initialised), mxLoop, tolerance, on `stream` --, bit for bit what
 *   xinv_<form>_f64_dev returns for the same arrays.  Returns when the stop rule has decided every member: flags are
 *   final on return; S completes IN STREAM ORDER -- the copy of the final state into S may still be queued on `stream`
 *   (work queued on `stream` afterwards sees the result; a reader on another stream or on the host synchronises with
 *   `stream` first).  xinv_last_stats() has planned = 1.  Solves on one device are serialised (per-device lock),
 *   whatever thread or plan they come from.
 * CONTRACT while a plan lives: the coefficient arrays it was created on stay allocated and UNCHANGED -- values of the
 *   forcing may change as long as its set of undefined points does not (the plan's tile lists leave out tiles whose
 *   forcing is undefined throughout).  After any other change call xinv_plan_refresh (re-derives everything in place,
 *   synchronous) or create a new plan.  S is not part of the plan: any S of the planned shape and stride may be solved.
 * xinv_plan_destroy frees the plan's device memory (NULL is accepted). */
typedef struct xinv_plan xinv_plan;

int xinv_plan_create_standard_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                         const double *F, int64_t nbatch, const int64_t *strides, int64_t yc,
                                         int64_t xc, double dely, double delx, int BCy, int BCx, double delxSqr,
                                         double ratioQtr, double ratioSqr, double optArg, double undef,
                                         const xinv_options *opt, void *stream);

int xinv_plan_create_general_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                        const double *D, const double *E, const double *F, const double *G,
                                        int64_t nbatch, const int64_t *strides, int64_t yc, int64_t xc, double dely,
                                        double delx, int BCy, int BCx, double delxSqr, double ratio, double ratioQtr,
                                        double ratioSqr, double optArg, double undef, const xinv_options *opt,
                                        void *stream);

int xinv_plan_create_standard_3d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                         const double *F, int64_t nbatch, const int64_t *strides, int64_t zc,
                                         int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                                         int BCy, int BCx, double delxSqr, double ratio2Sqr, double ratio1Sqr,
                                         double optArg, double undef, const xinv_options *opt, void *stream);

int xinv_plan_create_general_3d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                        const double *D, const double *E, const double *F, const double *G,
                                        const double *H, int64_t nbatch, const int64_t *strides, int64_t zc,
                                        int64_t yc, int64_t xc, double delz, double dely, double delx, int BCz,
                                        int BCy, int BCx, double delxSqr, double ratio2, double ratio1,
                                        double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                                        const xinv_options *opt, void *stream);

int xinv_plan_create_general_bih_2d_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                            const double *D, const double *E, const double *F, const double *G,
                                            const double *H, const double *I, const double *J, int64_t nbatch,
                                            const int64_t *strides, int64_t yc, int64_t xc, double dely, double delx,
                                            int BCy, int BCx, double delxSSr, double delxTr, double delxSqr,
                                            double ratio, double ratioSSr, double ratioQtr, double ratioSqr,
                                            double optArg, double undef, const xinv_options *opt, void *stream);

int xinv_plan_create_standard_2d_test_f64_dev(xinv_plan **plan, const double *A, const double *B, const double *C,
                                              const double *D, const double *E, const double *F, int64_t nbatch,
                                              const int64_t *strides, int64_t yc, int64_t xc, double dely,
                                              double delx, int BCy, int BCx, double delxSqr, double ratioQtr,
                                              double ratioSqr, double optArg, double undef, const xinv_options *opt,
                                              void *stream);

int xinv_plan_solve_f64_dev(xinv_plan *plan, double *S, double *flags, int64_t mxLoop, double tolerance, void *stream);
/* nframes restarts of the plan's solve queued behind each other (apps.animate_iteration, reference apps.py:1031-1044: one
 * kernel call per frame, every frame continuing from the previous frame's S): frame f runs mxLoop + 1 sweeps at most from the
 * state frame f-1 left, its S is copied into frames + f * frame_stride (device pointer, frame_stride >= the elements S
 * spans) and its flags land in flags + 3 * nbatch * f (host, [nframes][nbatch][3]).  No host round trip between the frames
 * while every frame runs its whole budget; a frame the tolerance stops earlier sends the remaining frames down the road of
 * xinv_plan_solve_f64_dev, one call each.  Results are those of nframes calls of xinv_plan_solve_f64_dev, bit for bit.
 * Returns with S, the frames and the flags complete. */
int xinv_plan_solve_frames_f64_dev(xinv_plan *plan, double *S, double *frames, int64_t nframes, int64_t frame_stride,
                                   double *flags, int64_t mxLoop, double tolerance, void *stream);
int xinv_plan_refresh(xinv_plan *plan, void *stream);
int xinv_plan_destroy(xinv_plan *plan);

/* ---- Gill-Matsuno winds from the inverted mass field, SURVEY 8(f) rank 3 -----------------------
 * replaces apps.cal_flow(vtype='GillMatsuno') (apps.py:1277-1317) for device-resident fields, so
 * config 4 delivers (phi, u, v) without a host round trip.  All array arguments are DEVICE
 * pointers.  ytab[3*yc+3] / xtab[3*xc+3]: numpy.gradient's non-uniform interior weights a, b, c
 * per index followed by {dx, dx_first, dx_last}; y/xuniform select numpy's uniform-spacing form.
 * rowtab[3*yc]: coef1 = eps/(eps^2+f^2), coef2 = f/(eps^2+f^2), cos(lat) per row. */
int xinv_gm_flow_f64_dev(const double *S, double *u, double *v, int64_t nbatch, int64_t yc,
                         int64_t xc, const double *ytab, const double *xtab, int yuniform,
                         int xuniform, const double *rowtab, double deg2m, int latlon, void *stream);

/* mean |S| over S != undef of one device-resident slab of n elements (reference
 * numbas.absNorm2D/3D, numbas.py:1710-1728 / 1689-1708); *out is a host double. */
int xinv_abs_norm_f64_dev(const double *S, int64_t n, double undef, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XINV_H */
