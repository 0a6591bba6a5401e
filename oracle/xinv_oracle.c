/*
 * xinv_oracle.c -- CPU restatement of the reference SOR kernels.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from it.  The product (xinvert_amd/) never links, imports or calls it.
 *
 * What it restates (all file:line relative to /root/reference):
 *   xo_standard_2d  <- xinvert/numbas.py:215-416   invert_standard_2D
 *   xo_general_2d   <- xinvert/numbas.py:987-1201  invert_general_2D
 *   xo_standard_3d  <- xinvert/numbas.py:15-212    invert_standard_3D
 *   xo_general_bih_2d <- xinvert/numbas.py:1204-1586 invert_general_bih_2D (radius-2, Munk)
 *   xo_general_3d     <- xinvert/numbas.py:745-984 invert_general_3D (7-point, 3DOcean)
 *   xo_standard_2d_test <- xinvert/numbas.py:420-629 invert_standard_2D_test (Fofonoff, Bretherton)
 *   norm2d / norm3d <- xinvert/numbas.py:1710-1728 / 1689-1708  absNorm2D / absNorm3D
 *
 * Two orderings of the same point update:
 *   order == XO_LEX       the reference's lexicographic Gauss-Seidel SOR (row-major, in place).
 *                         Pinned bit-for-bit (S and flags) against the reference's own numbas.py
 *                         imported as plain Python in the build container
 *                         (tests/golden/gen_golden.py -> tests/golden/ fixtures) and against the
 *                         loop counts / known answers the reference's tests and notebooks print.
 *   order == XO_COLOUR_*  the coloured (red-black / 4-colour) ordering the HIP kernels use.
 *                         Same per-point arithmetic, same boundary pre-pass, same norm and
 *                         stopping rule; only the visiting order of the points inside one sweep
 *                         changes.  Parity of the HIP path is bitwise (S) against this ordering;
 *                         converged fields agree with XO_LEX to <= 1e-6 rel-L2 (tests).
 *
 * Arithmetic notes.  Every expression keeps the reference's association order; build with
 * -ffp-contract=off (no FMA), no fast-math.  BC codes: 0 fixed, 1 extend, 2 periodic.
 *
 * Deliberate, documented deviation: the 'extend' pre-pass second loop (numbas.py:297-301) runs
 * `for i in range(1, yc-1)` over a COLUMN index; for yc > xc the reference indexes out of bounds
 * (undefined under numba's unchecked indexing).  Here the index is clamped to i < xc.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>

#define XO_LEX 0
#define XO_COLOUR_AUTO 1
#define XO_COLOUR_2 2
#define XO_COLOUR_4 4
/* ORed into `order` (standard 2-D, general 2-D and standard 3-D forms; B must be identically zero in 2-D): the
 * contracted arithmetic of the HIP kernels' opt-in mode XINV_FLAG_FMA -- explicit fma() at fixed positions of the update,
 * the relaxation factor as before.  NOT the reference's arithmetic: tied to it by tests (<= 1e-12 relative after tens
 * of sweeps, <= 1e-6 rel-L2 converged); the HIP kernels with the flag are bitwise this. */
#define XO_FMA 0x100

#define BC_FIXED 0
#define BC_EXTEND 1
#define BC_PERIODIC 2

/* ---------------------------------------------------------------- norms */
/* numbas.py:1710-1728 */
static double norm2d(const double *S, int64_t yc, int64_t xc, double undef)
{
    double norm = 0.0;
    int64_t count = 0;
    for (int64_t j = 0; j < yc; j++)
        for (int64_t i = 0; i < xc; i++) {
            double v = S[j * xc + i];
            if (v != undef) { norm += fabs(v); count++; }
        }
    if (count != 0) norm /= (double)count; else norm = NAN;
    return norm;
}

/* numbas.py:1689-1708 */
static double norm3d(const double *S, int64_t zc, int64_t yc, int64_t xc, double undef)
{
    return norm2d(S, zc * yc, xc, undef);   /* same row-major serial order */
}

/* ------------------------------------------------- sweep-loop control (a5) */
/* numbas.py:401-414 (std 2D), 1186-1199 (gen 2D), 197-210 (3D).  Returns 1 to stop. */
typedef struct { int64_t loop; double normPrev; } xo_ctl;

static int ctl_step(xo_ctl *c, double norm, double *flags, int64_t mxLoop, double tol,
                    int stop_on_zero_norm)
{
    if (isnan(norm) || norm > 1e100) { flags[0] = 1.0; return 1; }
    flags[1] = fabs(norm - c->normPrev) / c->normPrev;
    flags[2] = (double)c->loop;
    if (flags[1] < tol || c->loop >= mxLoop || (stop_on_zero_norm && norm == 0.0)) return 1;
    c->normPrev = norm;
    c->loop += 1;
    return 0;
}

/* ------------------------------------------- 'extend' boundary pre-pass (a6) */
/* numbas.py:284-310 / 1064-1090: one 2-D slab of yc x xc. */
static void extend2d(double *S, int64_t yc, int64_t xc, int BCx, double undef)
{
    double *r0 = S, *r1 = S + xc, *rm2 = S + (yc - 2) * xc, *rm1 = S + (yc - 1) * xc;
    if (BCx == BC_PERIODIC) {
        for (int64_t i = 0; i < xc; i++) {
            if (r1[i] != undef) r0[i] = r1[i];
            if (rm2[i] != undef) rm1[i] = rm2[i];
        }
    } else {
        for (int64_t i = 1; i < xc - 1; i++) {
            if (r1[i] != undef) r0[i] = r1[i];
            if (rm2[i] != undef) rm1[i] = rm2[i];
        }
        int64_t lim = yc - 1 < xc ? yc - 1 : xc;          /* clamp: see header */
        for (int64_t i = 1; i < lim; i++) {
            if (r1[i] != undef) r0[i] = r1[i];
            if (rm2[i] != undef) rm1[i] = rm2[i];
        }
        if (r1[1] != undef) r0[0] = r1[1];
        if (r1[xc - 2] != undef) r0[xc - 1] = r1[xc - 2];
        if (rm2[1] != undef) rm1[0] = rm2[1];
        if (rm2[xc - 2] != undef) rm1[xc - 1] = rm2[xc - 2];
    }
}

/* numbas.py:87-115: planes k = 1 .. zc-2 only; the non-periodic branch repeats the same
 * i-loop twice (idempotent) and fixes the four corners of each plane. */
/* `tall`: the general 3-D kernel's second loop runs over range(1, yc-1) (numbas.py:872-876); inside
 * the array bounds it re-copies the same rows, plus column xc-1 when yc > xc (as extend2d). */
static void extend3d(double *S, int64_t zc, int64_t yc, int64_t xc, int BCx, double undef, int tall)
{
    for (int64_t k = 1; k < zc - 1; k++) {
        double *P = S + k * yc * xc;
        double *r0 = P, *r1 = P + xc, *rm2 = P + (yc - 2) * xc, *rm1 = P + (yc - 1) * xc;
        if (BCx == BC_PERIODIC) {
            for (int64_t i = 0; i < xc; i++) {
                if (r1[i] != undef) r0[i] = r1[i];
                if (rm2[i] != undef) rm1[i] = rm2[i];
            }
        } else {
            for (int64_t i = 1; i < xc - 1; i++) {
                if (r1[i] != undef) r0[i] = r1[i];
                if (rm2[i] != undef) rm1[i] = rm2[i];
            }
            if (tall && yc > xc) {
                if (r1[xc - 1] != undef) r0[xc - 1] = r1[xc - 1];
                if (rm2[xc - 1] != undef) rm1[xc - 1] = rm2[xc - 1];
            }
            if (r1[1] != undef) r0[0] = r1[1];
            if (r1[xc - 2] != undef) r0[xc - 1] = r1[xc - 2];
            if (rm2[1] != undef) rm1[0] = rm2[1];
            if (rm2[xc - 2] != undef) rm1[xc - 1] = rm2[xc - 2];
        }
    }
}

/* ------------------------------------------------------------ point updates */
/* numbas.py:343-369 (inner), 314-340 (west, im = xc-1, ip = 1), 373-399 (east, im = xc-2,
 * ip = 0).  `west` reproduces the two irregularities of the reference's i == 0 branch
 * (numbas.py:327-328): it multiplies by B[j+1,1] where it tested B[j+1,0], and it differences
 * S[j-1,0] - S[j-1,-1] where the inner loop has S[j-1,i+1] - S[j-1,i-1]. */
static inline void upd_std2d(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t xc, int64_t j, int64_t i, int64_t im,
                             int64_t ip, int west, double delxSqr, double ratioQtr,
                             double ratioSqr, double optArg, double undef, int fm)
{
    const int64_t r = j * xc, rp = (j + 1) * xc, rm = (j - 1) * xc;
    const int64_t bn = west ? ip : i, sq = west ? i : ip;
    int cond = (F[r + i] != undef &&
                A[rp + i] != undef && A[r + i] != undef &&
                B[r + ip] != undef && B[r + im] != undef &&
                B[rp + i] != undef && B[rm + i] != undef &&
                C[r + ip] != undef && C[r + i] != undef);
    if (!cond) return;
    if (fm) {           /* XO_FMA (B == 0, checked by the caller): the contracted form of the HIP kernels' opt-in mode */
        const double sC = S[r + i];
        const double y = fma(A[rp + i], S[rp + i] - sC, -(A[r + i] * (sC - S[rm + i])));
        const double x = fma(C[r + ip], S[r + ip] - sC, -(C[r + i] * (sC - S[r + im])));
        double t = fma(y, ratioSqr, x);
        t = fma(-F[r + i], delxSqr, t);
        const double rq = optArg / ((A[rp + i] + A[r + i]) * ratioSqr + (C[r + ip] + C[r + i]));
        S[r + i] = fma(t, rq, sC);
        return;
    }
    double temp = (
        (
            A[rp + i] * (S[rp + i] - S[r + i]) -
            A[r + i] * (S[r + i] - S[rm + i])
        ) * ratioSqr + (
            B[rp + bn] * (S[rp + ip] - S[rp + im]) -
            B[rm + i] * (S[rm + sq] - S[rm + im])
        ) * ratioQtr + (
            B[r + ip] * (S[rp + ip] - S[rm + ip]) -
            B[r + im] * (S[rp + im] - S[rm + im])
        ) * ratioQtr + (
            C[r + ip] * (S[r + ip] - S[r + i]) -
            C[r + i] * (S[r + i] - S[r + im])
        )
    ) - F[r + i] * delxSqr;
    temp *= optArg / ((A[rp + i] + A[r + i]) * ratioSqr + (C[r + ip] + C[r + i]));
    S[r + i] += temp;
}

/* numbas.py:1125-1153 (inner), 1094-1122 (west), 1156-1184 (east). */
static inline void upd_gen2d(double *S, const double *A, const double *B, const double *C,
                             const double *D, const double *E, const double *F, const double *G,
                             int64_t xc, int64_t j, int64_t i, int64_t im, int64_t ip,
                             double delx, double delxSqr, double ratio, double ratioQtr,
                             double ratioSqr, double optArg, double undef, int fm)
{
    const int64_t r = j * xc, rp = (j + 1) * xc, rm = (j - 1) * xc;
    int cond = (G[r + i] != undef &&
                A[r + i] != undef && B[r + i] != undef &&
                C[r + i] != undef && D[r + i] != undef &&
                E[r + i] != undef && F[r + i] != undef);
    if (!cond) return;
    if (fm) {           /* XO_FMA (B == 0, checked by the caller) */
        const double sC = S[r + i], sP = S[rp + i], sM = S[rm + i], sE = S[r + ip], sW = S[r + im];
        double t = (A[r + i] * ((sP - sC) - (sC - sM))) * ratioSqr;
        t = fma(C[r + i], (sE - sC) - (sC - sW), t);
        double v = (D[r + i] * (sP - sM)) * ratio;
        v = fma(E[r + i], sE - sW, v);
        t = fma(v * delx, 0.5, t);
        t = fma(fma(F[r + i], sC, -G[r + i]), delxSqr, t);
        const double rq = optArg / ((A[r + i] * ratioSqr + C[r + i]) * 2.0
                                    - F[r + i] * delxSqr);
        S[r + i] = fma(t, rq, sC);
        return;
    }
    double temp = (
        A[r + i] * (
            (S[rp + i] - S[r + i]) - (S[r + i] - S[rm + i])
        ) * ratioSqr +
        B[r + i] * (
            (S[rp + ip] - S[rm + ip]) - (S[rp + im] - S[rm + im])
        ) * ratioQtr +
        C[r + i] * (
            (S[r + ip] - S[r + i]) - (S[r + i] - S[r + im])
        ) + (
        D[r + i] * (
            (S[rp + i] - S[rm + i])
        ) * ratio +
        E[r + i] * (
            (S[r + ip] - S[r + im])
        )) * delx / 2.0 + (
        F[r + i] * S[r + i] - G[r + i]) * delxSqr
    );
    temp *= optArg / ((A[r + i] * ratioSqr + C[r + i]) * 2.0
                      - F[r + i] * delxSqr);
    S[r + i] += temp;
}

/* numbas.py:146-169 (inner), 120-143 (west), 172-195 (east).  P = yc*xc plane stride. */
static inline void upd_std3d(double *S, const double *A, const double *B, const double *C,
                             const double *F, int64_t P, int64_t xc, int64_t k, int64_t j,
                             int64_t i, int64_t im, int64_t ip, double delxSqr,
                             double ratio2Sqr, double ratio1Sqr, double optArg, double undef, int fm)
{
    const int64_t r = k * P + j * xc;
    const int64_t c = r + i;
    int cond = (F[c] != undef &&
                A[c + P] != undef && A[c] != undef &&
                B[c + xc] != undef && B[c] != undef &&
                C[r + ip] != undef && C[c] != undef);
    if (!cond) return;
    if (fm) {           /* XO_FMA */
        const double sC = S[c];
        const double ya = fma(A[c + P], S[c + P] - sC, -(A[c] * (sC - S[c - P])));
        const double yb = fma(B[c + xc], S[c + xc] - sC, -(B[c] * (sC - S[c - xc])));
        const double yc_ = fma(C[r + ip], S[r + ip] - sC, -(C[c] * (sC - S[r + im])));
        double t = fma(ya, ratio2Sqr, fma(yb, ratio1Sqr, yc_));
        t = fma(-F[c], delxSqr, t);
        const double rq = optArg / ((A[c + P] + A[c]) * ratio2Sqr +
                                    (B[c + xc] + B[c]) * ratio1Sqr +
                                    (C[r + ip] + C[c]));
        S[c] = fma(t, rq, sC);
        return;
    }
    double temp = (
        (
            A[c + P] * (S[c + P] - S[c]) -
            A[c] * (S[c] - S[c - P])
        ) * ratio2Sqr + (
            B[c + xc] * (S[c + xc] - S[c]) -
            B[c] * (S[c] - S[c - xc])
        ) * ratio1Sqr + (
            C[r + ip] * (S[r + ip] - S[c]) -
            C[c] * (S[c] - S[r + im])
        )
    ) - F[c] * delxSqr;
    temp *= optArg / ((A[c + P] + A[c]) * ratio2Sqr +
                      (B[c + xc] + B[c]) * ratio1Sqr +
                      (C[r + ip] + C[c]));
    S[c] += temp;
}

/* numbas.py:899-930 (inner), 848-893 (west: tests G twice and never H, numbas.py:849-852),
 * 933-966 (east).  7-point, no cross terms. */
static inline void upd_gen3d(double *S, const double *const *c, int64_t P, int64_t xc, int64_t k,
                             int64_t j, int64_t i, int64_t im, int64_t ip, int west,
                             double delx, double delxSqr, double ratio2, double ratio1,
                             double ratio2Sqr, double ratio1Sqr, double optArg, double undef)
{
    const int64_t r = k * P + j * xc, q = r + i;
    const double A = c[0][q], B = c[1][q], C = c[2][q], D = c[3][q], E = c[4][q];
    const double F = c[5][q], G = c[6][q], H = c[7][q];
    int cond = ((west || H != undef) && G != undef && A != undef && B != undef && C != undef &&
                D != undef && E != undef && F != undef);
    if (!cond) return;
    double temp = (
        A * (
            (S[q + P] - S[q])-(S[q] - S[q - P])
        ) * ratio2Sqr +
        B * (
            (S[q + xc] - S[q])-(S[q] - S[q - xc])
        ) * ratio1Sqr +
        C * (
            (S[r + ip] - S[q])-(S[q] - S[r + im])
        ) + (
        D * (
            (S[q + P] - S[q - P])
        ) * ratio2 +
        E * (
            (S[q + xc] - S[q - xc])
        ) * ratio1 +
        F * (
            (S[r + ip] - S[r + im])
        )) * delx / 2.0 + (
        G * S[q] - H) * delxSqr
    );
    temp *= optArg / ((
        A*ratio2Sqr + B*ratio1Sqr + C
    ) * 2.0 - G*delxSqr);
    S[q] += temp;
}

/* ------------------------------------------------------------------ colours */
/* Colour of point (j,i) (2-D) for the coloured ordering.  base = 2: red-black on (j+i)&1,
 * valid when the cross coefficient B is identically zero (5-point coupling); base = 4:
 * (j&1, i&1), valid for the full 9-point coupling.  With periodic x and odd xc, columns 0 and
 * xc-1 are neighbours of equal base colour, so column xc-1 forms two extra colours by row
 * parity (a "seam").  The HIP kernels use exactly this function. */
static inline int colour2d(int64_t j, int64_t i, int64_t xc, int base, int seam)
{
    if (seam && i == xc - 1) return base + (int)(j & 1);
    if (base == 2) return (int)((j + i) & 1);
    return (int)(2 * (j & 1) + (i & 1));
}

/* Order of the colour passes of one sweep.  Without a seam: 0 .. base-1.  With the seam (periodic x, odd xc) each
 * seam colour runs RIGHT AFTER the base colour its points would otherwise belong to -- column xc-1 is an even column,
 * so rows of parity p of it follow colour (p, 0): red, red', black, black' (base 2), c0, c0', c1, c2, c2', c3 (base
 * 4).  (Until round 3 the two seam colours ran after all base colours; the fused kernels update the seam column
 * inside the half-sweep of its own colour, right after column 0, which is this order.) */
static inline int seq_colour(int pos, int base, int seam)
{
    static const int s2[4] = { 0, 2, 1, 3 }, s4[6] = { 0, 4, 1, 2, 5, 3 };
    if (!seam) return pos;
    return base == 2 ? s2[pos] : s4[pos];
}

static int all_zero(const double *B, int64_t n)
{
    for (int64_t t = 0; t < n; t++) if (B[t] != 0.0) return 0;
    return 1;
}

/* ---------------------------------------------------------------- kernels */
int xo_standard_2d(double *S, const double *A, const double *B, const double *C,
                   const double *F, int64_t yc, int64_t xc, double dely, double delx,
                   int BCy, int BCx, double delxSqr, double ratioQtr, double ratioSqr,
                   double optArg, double undef, double *flags, int64_t mxLoop,
                   double tolerance, int order)
{
    (void)dely; (void)delx;
    if (yc < 3 || xc < 3) return -1;
    const int fm = (order & XO_FMA) != 0;
    order &= ~XO_FMA;
    if (fm && !all_zero(B, yc * xc)) return -2;
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    int base = 0, seam = 0;
    if (order != XO_LEX) {
        base = order == XO_COLOUR_AUTO ? (all_zero(B, yc * xc) ? 2 : 4) : order;
        seam = per && (xc & 1);
    }
    const int ncol = base + (seam ? 2 : 0);
    const int64_t i0 = per ? 0 : 1, i1 = per ? xc : xc - 1;

    for (;;) {
        if (BCy == BC_EXTEND) extend2d(S, yc, xc, BCx, undef);

        if (order == XO_LEX) {
            for (int64_t j = 1; j < yc - 1; j++) {
                if (per)
                    upd_std2d(S, A, B, C, F, xc, j, 0, xc - 1, 1, 1,
                              delxSqr, ratioQtr, ratioSqr, optArg, undef, fm);
                for (int64_t i = 1; i < xc - 1; i++)
                    upd_std2d(S, A, B, C, F, xc, j, i, i - 1, i + 1, 0,
                              delxSqr, ratioQtr, ratioSqr, optArg, undef, fm);
                if (per)
                    upd_std2d(S, A, B, C, F, xc, j, xc - 1, xc - 2, 0, 0,
                              delxSqr, ratioQtr, ratioSqr, optArg, undef, fm);
            }
        } else {
            for (int c = 0; c < ncol; c++)
                for (int64_t j = 1; j < yc - 1; j++)
                    for (int64_t i = i0; i < i1; i++) {
                        if (colour2d(j, i, xc, base, seam) != seq_colour(c, base, seam)) continue;
                        int64_t im = i == 0 ? xc - 1 : i - 1;
                        int64_t ip = i == xc - 1 ? 0 : i + 1;
                        upd_std2d(S, A, B, C, F, xc, j, i, im, ip, i == 0,
                                  delxSqr, ratioQtr, ratioSqr, optArg, undef, fm);
                    }
        }

        double norm = norm2d(S, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 1)) break;
    }
    return 0;
}

int xo_general_2d(double *S, const double *A, const double *B, const double *C,
                  const double *D, const double *E, const double *F, const double *G,
                  int64_t yc, int64_t xc, double dely, double delx, int BCy, int BCx,
                  double delxSqr, double ratio, double ratioQtr, double ratioSqr,
                  double optArg, double undef, double *flags, int64_t mxLoop,
                  double tolerance, int order)
{
    (void)dely;
    if (yc < 3 || xc < 3) return -1;
    const int fm = (order & XO_FMA) != 0;
    order &= ~XO_FMA;
    if (fm && !all_zero(B, yc * xc)) return -2;
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    int base = 0, seam = 0;
    if (order != XO_LEX) {
        base = order == XO_COLOUR_AUTO ? (all_zero(B, yc * xc) ? 2 : 4) : order;
        seam = per && (xc & 1);
    }
    const int ncol = base + (seam ? 2 : 0);
    const int64_t i0 = per ? 0 : 1, i1 = per ? xc : xc - 1;

    for (;;) {
        if (BCy == BC_EXTEND) extend2d(S, yc, xc, BCx, undef);

        if (order == XO_LEX) {
            for (int64_t j = 1; j < yc - 1; j++) {
                if (per)
                    upd_gen2d(S, A, B, C, D, E, F, G, xc, j, 0, xc - 1, 1,
                              delx, delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, fm);
                for (int64_t i = 1; i < xc - 1; i++)
                    upd_gen2d(S, A, B, C, D, E, F, G, xc, j, i, i - 1, i + 1,
                              delx, delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, fm);
                if (per)
                    upd_gen2d(S, A, B, C, D, E, F, G, xc, j, xc - 1, xc - 2, 0,
                              delx, delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, fm);
            }
        } else {
            for (int c = 0; c < ncol; c++)
                for (int64_t j = 1; j < yc - 1; j++)
                    for (int64_t i = i0; i < i1; i++) {
                        if (colour2d(j, i, xc, base, seam) != seq_colour(c, base, seam)) continue;
                        int64_t im = i == 0 ? xc - 1 : i - 1;
                        int64_t ip = i == xc - 1 ? 0 : i + 1;
                        upd_gen2d(S, A, B, C, D, E, F, G, xc, j, i, im, ip,
                                  delx, delxSqr, ratio, ratioQtr, ratioSqr, optArg, undef, fm);
                    }
        }

        double norm = norm2d(S, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 0)) break;
    }
    return 0;
}

/* BCz is accepted and never read, exactly as numbas.py:16-19. */
int xo_standard_3d(double *S, const double *A, const double *B, const double *C,
                   const double *F, int64_t zc, int64_t yc, int64_t xc, double delz,
                   double dely, double delx, int BCz, int BCy, int BCx, double delxSqr,
                   double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                   double *flags, int64_t mxLoop, double tolerance, int order)
{
    (void)delz; (void)dely; (void)delx; (void)BCz;
    if (zc < 3 || yc < 3 || xc < 3) return -1;
    const int fm = (order & XO_FMA) != 0;
    order &= ~XO_FMA;
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    const int seam = (order != XO_LEX) && per && (xc & 1);
    const int ncol = 2 + (seam ? 2 : 0);
    const int64_t i0 = per ? 0 : 1, i1 = per ? xc : xc - 1;
    const int64_t P = yc * xc;

    for (;;) {
        if (BCy == BC_EXTEND) extend3d(S, zc, yc, xc, BCx, undef, 0);

        if (order == XO_LEX) {
            for (int64_t k = 1; k < zc - 1; k++)
                for (int64_t j = 1; j < yc - 1; j++) {
                    if (per)
                        upd_std3d(S, A, B, C, F, P, xc, k, j, 0, xc - 1, 1,
                                  delxSqr, ratio2Sqr, ratio1Sqr, optArg, undef, fm);
                    for (int64_t i = 1; i < xc - 1; i++)
                        upd_std3d(S, A, B, C, F, P, xc, k, j, i, i - 1, i + 1,
                                  delxSqr, ratio2Sqr, ratio1Sqr, optArg, undef, fm);
                    if (per)
                        upd_std3d(S, A, B, C, F, P, xc, k, j, xc - 1, xc - 2, 0,
                                  delxSqr, ratio2Sqr, ratio1Sqr, optArg, undef, fm);
                }
        } else {
            for (int c = 0; c < ncol; c++)
                for (int64_t k = 1; k < zc - 1; k++)
                    for (int64_t j = 1; j < yc - 1; j++)
                        for (int64_t i = i0; i < i1; i++) {
                            int col = (seam && i == xc - 1) ? 2 + (int)((k + j) & 1)
                                                             : (int)((k + j + i) & 1);
                            if (col != seq_colour(c, 2, seam)) continue;
                            int64_t im = i == 0 ? xc - 1 : i - 1;
                            int64_t ip = i == xc - 1 ? 0 : i + 1;
                            upd_std3d(S, A, B, C, F, P, xc, k, j, i, im, ip,
                                      delxSqr, ratio2Sqr, ratio1Sqr, optArg, undef, fm);
                        }
        }

        double norm = norm3d(S, zc, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 0)) break;
    }
    return 0;
}


/* numbas.py:745-984 invert_general_3D (optional; apps.invert_3DOcean).
 * BCz is accepted and never read.  Same loop control and norm as the standard 3-D kernel. */
int xo_general_3d(double *S, const double *A, const double *B, const double *C, const double *D,
                  const double *E, const double *F, const double *G, const double *H,
                  int64_t zc, int64_t yc, int64_t xc, double delz, double dely, double delx,
                  int BCz, int BCy, int BCx, double delxSqr, double ratio2, double ratio1,
                  double ratio2Sqr, double ratio1Sqr, double optArg, double undef,
                  double *flags, int64_t mxLoop, double tolerance, int order)
{
    (void)delz; (void)dely; (void)BCz;
    if (zc < 3 || yc < 3 || xc < 3) return -1;
    const double *c[8] = { A, B, C, D, E, F, G, H };
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    const int seam = (order != XO_LEX) && per && (xc & 1);
    const int ncol = 2 + (seam ? 2 : 0);
    const int64_t i0 = per ? 0 : 1, i1 = per ? xc : xc - 1;
    const int64_t P = yc * xc;

    for (;;) {
        if (BCy == BC_EXTEND) extend3d(S, zc, yc, xc, BCx, undef, 1);

        if (order == XO_LEX) {
            for (int64_t k = 1; k < zc - 1; k++)
                for (int64_t j = 1; j < yc - 1; j++) {
                    if (per)
                        upd_gen3d(S, c, P, xc, k, j, 0, xc - 1, 1, 1, delx, delxSqr, ratio2,
                                  ratio1, ratio2Sqr, ratio1Sqr, optArg, undef);
                    for (int64_t i = 1; i < xc - 1; i++)
                        upd_gen3d(S, c, P, xc, k, j, i, i - 1, i + 1, 0, delx, delxSqr, ratio2,
                                  ratio1, ratio2Sqr, ratio1Sqr, optArg, undef);
                    if (per)
                        upd_gen3d(S, c, P, xc, k, j, xc - 1, xc - 2, 0, 0, delx, delxSqr, ratio2,
                                  ratio1, ratio2Sqr, ratio1Sqr, optArg, undef);
                }
        } else {
            for (int cc = 0; cc < ncol; cc++)
                for (int64_t k = 1; k < zc - 1; k++)
                    for (int64_t j = 1; j < yc - 1; j++)
                        for (int64_t i = i0; i < i1; i++) {
                            int col = (seam && i == xc - 1) ? 2 + (int)((k + j) & 1)
                                                             : (int)((k + j + i) & 1);
                            if (col != seq_colour(cc, 2, seam)) continue;
                            int64_t im = i == 0 ? xc - 1 : i - 1;
                            int64_t ip = i == xc - 1 ? 0 : i + 1;
                            upd_gen3d(S, c, P, xc, k, j, i, im, ip, per && i == 0, delx, delxSqr,
                                      ratio2, ratio1, ratio2Sqr, ratio1Sqr, optArg, undef);
                        }
        }

        double norm = norm3d(S, zc, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 0)) break;
    }
    return 0;
}


/* ===================================================================== biharmonic 2-D
 * numbas.py:1204-1586.  Radius-2 stencil; rows 2..yc-3 and columns 2..xc-3 are updated, plus
 * columns 0, 1, xc-2, xc-1 when x is periodic.  Irregularities of the reference reproduced here:
 *   - the periodic branches write the G term as `* delxTr / 2.0 * ratio`, the inner loop as
 *     `* delxTr * ratio / 2.0` (different rounding) -> `edge`;
 *   - the two east branches (numbas.py:1495-1497, 1540-1542) index the B term with the STALE inner
 *     loop variable: `S[.., i-4]` / `S[.., i-3]` with i == xc-3, i.e. columns xc-7 / xc-6 instead
 *     of xc-4 / xc-3 -> `bm2`;
 *   - the 'extend' pre-pass differs between periodic (row 0 <- old row 1, row 1 <- row 2) and
 *     non-periodic (rows 0, 1 <- row 2) x (numbas.py:1299-1343); its second loop is clamped to the
 *     array bounds as in extend2d. */
static inline void upd_bih2d(double *S, const double *const *c, int64_t xc, int64_t j,
                             int64_t i, int64_t im2, int64_t im1, int64_t ip1, int64_t ip2,
                             int64_t bm2, int edge, double delxSSr, double delxTr,
                             double delxSqr, double ratio, double ratioSSr, double ratioQtr,
                             double ratioSqr, double optArg, double undef)
{
    const int64_t r = j * xc, p = r + i;
    const double A = c[0][p], B = c[1][p], C = c[2][p], D = c[3][p], E = c[4][p];
    const double F = c[5][p], G = c[6][p], H = c[7][p], I = c[8][p], J = c[9][p];
    int cond = (A != undef && B != undef && C != undef && D != undef && E != undef &&
                F != undef && G != undef && H != undef && I != undef && J != undef);
    if (!cond) return;
    const double *r0 = S + r, *p1 = r0 + xc, *p2 = r0 + 2 * xc, *m1 = r0 - xc, *m2 = r0 - 2 * xc;
    double gterm = G * (
                       (p1[i] - m1[i])
                   );
    if (edge) gterm = gterm * delxTr / 2.0 * ratio;
    else      gterm = gterm * delxTr * ratio / 2.0;
    double temp = (
        A * (
            (p2[i] - 4.0*p1[i] + 6.0*r0[i] - 4.0*m1[i] + m2[i])
        ) * ratioSSr +
        B * (
            (    p2[ip2] - 2.0*p2[i] +     p2[bm2] +
            -2.0*r0[ip2] + 4.0*r0[i] - 2.0*r0[bm2] +
                 m2[ip2] - 2.0*m2[i] +     m2[bm2])
        ) * ratioSqr / 16.0 +
        C * (
            (r0[ip2] - 4.0*r0[ip1] + 6.0*r0[i] - 4.0*r0[im1] + r0[im2])
        ) +
        D * (
            (p1[i] - r0[i])-(r0[i] - m1[i])
        ) * ratioSqr * delxSqr +
        E * (
            (p1[ip1] - m1[ip1])-(p1[im1] - m1[im1])
        ) * ratioQtr * delxSqr +
        F * (
            (r0[ip1] - r0[i])-(r0[i] - r0[im1])
        ) * delxSqr +
        gterm +
        H * (
            (r0[ip1] - r0[im1])
        ) * delxTr / 2.0 + (
        I * r0[i] - J) * delxSSr
    );
    temp *= -optArg / ((A*ratioSSr + C) * 6.0 +
                        B*ratioSqr / 4.0 +
                      -(D*ratioSqr + F) * 2.0 * delxSqr +
                        I*delxSSr);
    S[p] += temp;
}

static void extend_bih(double *S, int64_t yc, int64_t xc, int BCx, double undef)
{
    double *r0 = S, *r1 = S + xc, *r2 = S + 2 * xc;
    double *b1 = S + (yc - 1) * xc, *b2 = S + (yc - 2) * xc, *b3 = S + (yc - 3) * xc;
    if (BCx == BC_PERIODIC) {
        for (int64_t i = 0; i < xc; i++) {
            if (r2[i] != undef) { r0[i] = r1[i]; r1[i] = r2[i]; }
            if (b3[i] != undef) { b1[i] = b3[i]; b2[i] = b3[i]; }
        }
    } else {
        for (int64_t i = 1; i < xc - 1; i++) {
            if (r2[i] != undef) { r0[i] = r2[i]; r1[i] = r2[i]; }
            if (b3[i] != undef) { b1[i] = b3[i]; b2[i] = b3[i]; }
        }
        int64_t lim = yc - 1 < xc ? yc - 1 : xc;
        for (int64_t i = 1; i < lim; i++) {
            if (r2[i] != undef) { r0[i] = r2[i]; r1[i] = r2[i]; }
            if (b3[i] != undef) { b1[i] = b3[i]; b2[i] = b3[i]; }
        }
        if (r2[2] != undef) { r0[0] = r2[2]; r0[1] = r2[2]; r1[0] = r2[2]; r1[1] = r2[2]; }
        if (r2[xc - 3] != undef) { r0[xc - 1] = r2[xc - 3]; r0[xc - 2] = r2[xc - 3];
                                   r1[xc - 1] = r2[xc - 3]; r1[xc - 2] = r2[xc - 3]; }
        if (b3[2] != undef) { b1[0] = b3[2]; b2[0] = b3[2]; b1[1] = b3[2]; b2[1] = b3[2]; }
        if (b3[xc - 3] != undef) { b1[xc - 1] = b3[xc - 3]; b1[xc - 2] = b3[xc - 3];
                                   b2[xc - 1] = b3[xc - 3]; b2[xc - 2] = b3[xc - 3]; }
    }
}

/* Colour of (j,i) for the radius-2 stencil: (j%3, i%3) -> 9 colours; with periodic x and
 * xc % 3 != 0 the trailing xc%3 columns get 3 colours each (by j%3). */
static inline int colour_bih(int64_t j, int64_t i, int64_t xc, int trail)
{
    if (trail && i >= xc - trail) return 9 + 3 * (int)(i - (xc - trail)) + (int)(j % 3);
    return 3 * (int)(j % 3) + (int)(i % 3);
}

/* operands of the reference's five x-branches for column i */
static inline void bih_cols(int64_t i, int64_t xc, int per, int64_t *im2, int64_t *im1,
                            int64_t *ip1, int64_t *ip2, int64_t *bm2, int *edge)
{
    *im2 = i - 2; *im1 = i - 1; *ip1 = i + 1; *ip2 = i + 2; *edge = 0;
    if (per) {
        if (*im2 < 0) *im2 += xc;
        if (*im1 < 0) *im1 += xc;
        if (*ip1 >= xc) *ip1 -= xc;
        if (*ip2 >= xc) *ip2 -= xc;
        if (i < 2 || i >= xc - 2) *edge = 1;
    }
    *bm2 = *im2;
    if (per && i >= xc - 2) {                 /* stale loop variable: (xc-3) - 4 or - 3 */
        int64_t b = (i == xc - 2) ? xc - 7 : xc - 6;
        if (b < 0) b += xc;                   /* Python negative index */
        *bm2 = b;
    }
}

int xo_general_bih_2d(double *S, const double *A, const double *B, const double *C,
                      const double *D, const double *E, const double *F, const double *G,
                      const double *H, const double *I, const double *J,
                      int64_t yc, int64_t xc, double dely, double delx, int BCy, int BCx,
                      double delxSSr, double delxTr, double delxSqr, double ratio,
                      double ratioSSr, double ratioQtr, double ratioSqr, double optArg,
                      double undef, double *flags, int64_t mxLoop, double tolerance, int order)
{
    (void)dely; (void)delx;
    if (yc < 5 || xc < 7) return -1;
    const double *c[10] = { A, B, C, D, E, F, G, H, I, J };
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    const int trail = (order != XO_LEX && per) ? (int)(xc % 3) : 0;
    const int ncol = 9 + 3 * trail;
    const int64_t i0 = per ? 0 : 2, i1 = per ? xc : xc - 2;

    for (;;) {
        if (BCy == BC_EXTEND) extend_bih(S, yc, xc, BCx, undef);
        const int npass = (order == XO_LEX) ? 1 : ncol;
        for (int pass = 0; pass < npass; pass++)
            for (int64_t j = 2; j < yc - 2; j++)
                for (int64_t i = i0; i < i1; i++) {
                    if (order != XO_LEX && colour_bih(j, i, xc, trail) != pass) continue;
                    int64_t im2, im1, ip1, ip2, bm2; int edge;
                    bih_cols(i, xc, per, &im2, &im1, &ip1, &ip2, &bm2, &edge);
                    upd_bih2d(S, c, xc, j, i, im2, im1, ip1, ip2, bm2, edge, delxSSr, delxTr,
                              delxSqr, ratio, ratioSSr, ratioQtr, ratioSqr, optArg, undef);
                }
        double norm = norm2d(S, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 0)) break;
    }
    return 0;
}

/* ===================================================================== standard 2-D "test" form
 * numbas.py:420-629:  d/dy(A Sy + B Sx) + d/dx(C Sy + D Sx) + E S = F.  Same sweep, pre-pass,
 * stop rule (incl. norm == 0) and west-branch irregularities as invert_standard_2D. */
static inline void upd_std2dt(double *S, const double *A, const double *B, const double *C,
                              const double *D, const double *E, const double *F, int64_t xc,
                              int64_t j, int64_t i, int64_t im, int64_t ip, int west,
                              double delxSqr, double ratioQtr, double ratioSqr, double optArg,
                              double undef)
{
    const int64_t r = j * xc, rp = (j + 1) * xc, rm = (j - 1) * xc;
    const int64_t bn = west ? ip : i, sq = west ? i : ip;
    int cond = (F[r + i] != undef &&
                A[rp + i] != undef && A[r + i] != undef &&
                B[rp + i] != undef && B[rm + i] != undef &&
                C[r + ip] != undef && C[r + im] != undef &&
                D[r + ip] != undef && D[r + i] != undef &&
                E[r + i] != undef);
    if (!cond) return;
    double temp = (
        (
            A[rp + i] * (S[rp + i] - S[r + i]) -
            A[r + i] * (S[r + i] - S[rm + i])
        ) * ratioSqr + (
            B[rp + bn] * (S[rp + ip] - S[rp + im]) -
            B[rm + i] * (S[rm + sq] - S[rm + im])
        ) * ratioQtr + (
            C[r + ip] * (S[rp + ip] - S[rm + ip]) -
            C[r + im] * (S[rp + im] - S[rm + im])
        ) * ratioQtr + (
            D[r + ip] * (S[r + ip] - S[r + i]) -
            D[r + i] * (S[r + i] - S[r + im])
        )
    ) + (E[r + i] * S[r + i] - F[r + i]) * delxSqr;
    temp *= optArg / ((A[rp + i] + A[r + i]) * ratioSqr +
                      (D[r + ip] + D[r + i]) - E[r + i] * delxSqr);
    S[r + i] += temp;
}

int xo_standard_2d_test(double *S, const double *A, const double *B, const double *C,
                        const double *D, const double *E, const double *F, int64_t yc,
                        int64_t xc, double dely, double delx, int BCy, int BCx, double delxSqr,
                        double ratioQtr, double ratioSqr, double optArg, double undef,
                        double *flags, int64_t mxLoop, double tolerance, int order)
{
    (void)dely; (void)delx;
    if (yc < 3 || xc < 3) return -1;
    xo_ctl ctl = { 0, DBL_MAX };
    const int per = (BCx == BC_PERIODIC);
    int base = 0, seam = 0;
    if (order != XO_LEX) {
        base = order == XO_COLOUR_AUTO
                   ? ((all_zero(B, yc * xc) && all_zero(C, yc * xc)) ? 2 : 4) : order;
        seam = per && (xc & 1);
    }
    const int ncol = base + (seam ? 2 : 0);
    const int64_t i0 = per ? 0 : 1, i1 = per ? xc : xc - 1;

    for (;;) {
        if (BCy == BC_EXTEND) extend2d(S, yc, xc, BCx, undef);
        const int npass = (order == XO_LEX) ? 1 : ncol;
        for (int c = 0; c < npass; c++)
            for (int64_t j = 1; j < yc - 1; j++)
                for (int64_t i = i0; i < i1; i++) {
                    if (order != XO_LEX && colour2d(j, i, xc, base, seam) != seq_colour(c, base, seam)) continue;
                    int64_t im = i == 0 ? xc - 1 : i - 1;
                    int64_t ip = i == xc - 1 ? 0 : i + 1;
                    upd_std2dt(S, A, B, C, D, E, F, xc, j, i, im, ip, i == 0,
                               delxSqr, ratioQtr, ratioSqr, optArg, undef);
                }
        double norm = norm2d(S, yc, xc, undef);
        if (ctl_step(&ctl, norm, flags, mxLoop, tolerance, 1)) break;
    }
    return 0;
}

/* Standalone norm entry points (for tests of the fused device-side norm). */
double xo_abs_norm_2d(const double *S, int64_t yc, int64_t xc, double undef)
{
    return norm2d(S, yc, xc, undef);
}

double xo_abs_norm_3d(const double *S, int64_t zc, int64_t yc, int64_t xc, double undef)
{
    return norm3d(S, zc, yc, xc, undef);
}
