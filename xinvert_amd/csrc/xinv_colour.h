// xinv_colour.h -- colour-pass kernels: one launch per colour, in place on S.
//
// General path of the engine: every model, every BC combination, the 9-point forms and the
// odd-xc periodic seam.  Each thread owns one point of the launch's colour; points of one
// colour never read each other, so the in-place update is race free and its result is
// independent of scheduling (bitwise equal to the CPU restatement of the same ordering).
// Companion kernels: the 'extend' boundary pre-pass, the mean|S| norm (two deterministic
// stages) with the device-side stopping rule.
#pragma once
#include "xinv_device.h"

struct ColourArgs2D {
    double *S;
    const double *c[7];        // std: A,B,C,F ; gen: A,B,C,D,E,F,G
    int64_t sS, sc[7];         // batch strides (elements)
    int64_t yc, xc;
    int per;                   // periodic x
    int base;                  // 2 or 4
    int seam;                  // odd xc with periodic x
    int colour;                // colour of this launch
    int force;                 // ignore ctl.done (never set in production)
    int64_t member0;           // first member of this launch (grids are chunked to <= 32768 members)
    XinvScal sc_;
    const XinvCtl *ctl;
};

// Row / column of the thread's point for `colour`; returns false when the thread has none.
__device__ __forceinline__ bool xinv_colour_point(const ColourArgs2D &a, int64_t tj, int64_t ti,
                                                  int64_t &j, int64_t &i)
{
    const int cc = a.colour;
    if (cc >= a.base) {                          // seam colours: column xc-1, row parity cc-base
        if (ti != 0) return false;
        int par = cc - a.base;
        j = ((par == 1) ? 1 : 2) + 2 * tj;
        i = a.xc - 1;
        return j <= a.yc - 2;
    }
    if (a.base == 2) {
        j = 1 + tj;
        if (j > a.yc - 2) return false;
        i = 2 * ti + ((j + cc) & 1);
    } else {
        int pj = cc >> 1, pi = cc & 1;
        j = ((pj == 1) ? 1 : 2) + 2 * tj;
        if (j > a.yc - 2) return false;
        i = 2 * ti + pi;
    }
    const int64_t ilo = a.per ? 0 : 1;
    const int64_t ihi = a.per ? a.xc - 1 : a.xc - 2;
    if (i < ilo || i > ihi) return false;
    if (a.seam && i == a.xc - 1) return false;   // belongs to the seam colours
    return true;
}

template <bool NINE>
__global__ __launch_bounds__(256) void k_colour_std2d(ColourArgs2D a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tj = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    int64_t j, i;
    if (!xinv_colour_point(a, tj, ti, j, i)) return;
    const int64_t xc = a.xc;
    double *S = a.S + m * a.sS;
    const double *A = a.c[0] + m * a.sc[0];
    const double *B = a.c[1] + m * a.sc[1];
    const double *C = a.c[2] + m * a.sc[2];
    const double *F = a.c[3] + m * a.sc[3];
    const int64_t im = (i == 0) ? xc - 1 : i - 1;
    const int64_t ip = (i == xc - 1) ? 0 : i + 1;
    const int64_t r = j * xc, rp = r + xc, rm = r - xc;
    const double sC = S[r + i];
    double v;
    if (NINE) {
        const bool west = (i == 0);
        const int64_t bn = west ? ip : i, sq = west ? i : ip;
        v = xinv_upd_std2d_9(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                             S[rp + ip], S[rp + im], S[rm + ip], S[rm + im], S[rm + sq],
                             A[rp + i], A[r + i], B[r + ip], B[r + im], B[rp + i], B[rp + bn],
                             B[rm + i], C[r + ip], C[r + i], F[r + i], a.sc_);
    } else {
        v = xinv_upd_std2d_5(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                             A[rp + i], A[r + i], C[r + ip], C[r + i], F[r + i], true, a.sc_);
    }
    S[r + i] = v;
}

template <bool NINE>
__global__ __launch_bounds__(256) void k_colour_gen2d(ColourArgs2D a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tj = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    int64_t j, i;
    if (!xinv_colour_point(a, tj, ti, j, i)) return;
    const int64_t xc = a.xc;
    double *S = a.S + m * a.sS;
    const int64_t im = (i == 0) ? xc - 1 : i - 1;
    const int64_t ip = (i == xc - 1) ? 0 : i + 1;
    const int64_t r = j * xc, rp = r + xc, rm = r - xc;
    const int64_t p = r + i;
    const double cA = a.c[0][m * a.sc[0] + p], cC = a.c[2][m * a.sc[2] + p];
    const double cD = a.c[3][m * a.sc[3] + p], cE = a.c[4][m * a.sc[4] + p];
    const double cF = a.c[5][m * a.sc[5] + p], cG = a.c[6][m * a.sc[6] + p];
    const double sC = S[p];
    double v;
    if (NINE) {
        const double cB = a.c[1][m * a.sc[1] + p];
        v = xinv_upd_gen2d_9(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                             S[rp + ip], S[rp + im], S[rm + ip], S[rm + im],
                             cA, cB, cC, cD, cE, cF, cG, a.sc_);
    } else {
        v = xinv_upd_gen2d_5(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                             cA, cC, cD, cE, cF, cG, true, a.sc_);
    }
    S[p] = v;
}


// standard 2-D "test" form: c[] = A, B, C, D, E, F (numbas.invert_standard_2D_test).
template <bool NINE>
__global__ __launch_bounds__(256) void k_colour_std2dt(ColourArgs2D a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tj = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    int64_t j, i;
    if (!xinv_colour_point(a, tj, ti, j, i)) return;
    const int64_t xc = a.xc;
    double *S = a.S + m * a.sS;
    const double *A = a.c[0] + m * a.sc[0];
    const double *B = a.c[1] + m * a.sc[1];
    const double *C = a.c[2] + m * a.sc[2];
    const double *D = a.c[3] + m * a.sc[3];
    const double *E = a.c[4] + m * a.sc[4];
    const double *F = a.c[5] + m * a.sc[5];
    const int64_t im = (i == 0) ? xc - 1 : i - 1;
    const int64_t ip = (i == xc - 1) ? 0 : i + 1;
    const int64_t r = j * xc, rp = r + xc, rm = r - xc;
    const double sC = S[r + i];
    double v;
    if (NINE) {
        const bool west = (i == 0);
        const int64_t bn = west ? ip : i, sq = west ? i : ip;
        v = xinv_upd_std2dt_9(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                              S[rp + ip], S[rp + im], S[rm + ip], S[rm + im], S[rm + sq],
                              A[rp + i], A[r + i], B[rp + i], B[rp + bn], B[rm + i],
                              C[r + ip], C[r + im], D[r + ip], D[r + i], E[r + i], F[r + i], a.sc_);
    } else {
        v = xinv_upd_std2dt_5(sC, S[rp + i], S[rm + i], S[r + im], S[r + ip],
                              A[rp + i], A[r + i], D[r + ip], D[r + i], E[r + i], F[r + i], true, a.sc_);
    }
    S[r + i] = v;
}

// ------------------------------------------------------------------------------- 3-D
struct ColourArgs3D {
    double *S;
    const double *c[8];        // standard: A,B,C,F ; general: A..H
    int64_t sS, sc[8];
    int64_t zc, yc, xc;
    int per, seam, colour, force;
    XinvScal sc_;
    const XinvCtl *ctl;
    int64_t nbatch, member0;
};

// grid: x over column pairs, y over rows 1..yc-2, z over (member, plane 1..zc-2).
__global__ __launch_bounds__(256) void k_colour_std3d(ColourArgs3D a)
{
    const int64_t nk = a.zc - 2;
    const int64_t m = a.member0 + blockIdx.z / nk;
    const int64_t k = 1 + blockIdx.z % nk;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = 1 + (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    if (j > a.yc - 2) return;
    const int64_t xc = a.xc;
    int64_t i;
    if (a.colour >= 2) {                           // seam: column xc-1, parity of (k+j)
        if (ti != 0 || ((k + j) & 1) != a.colour - 2) return;
        i = xc - 1;
    } else {
        i = 2 * ti + ((k + j + a.colour) & 1);
        const int64_t ilo = a.per ? 0 : 1;
        const int64_t ihi = a.per ? xc - 1 : xc - 2;
        if (i < ilo || i > ihi) return;
        if (a.seam && i == xc - 1) return;
    }
    const int64_t P = a.yc * xc;
    double *S = a.S + m * a.sS;
    const double *A = a.c[0] + m * a.sc[0];
    const double *B = a.c[1] + m * a.sc[1];
    const double *C = a.c[2] + m * a.sc[2];
    const double *F = a.c[3] + m * a.sc[3];
    const int64_t im = (i == 0) ? xc - 1 : i - 1;
    const int64_t ip = (i == xc - 1) ? 0 : i + 1;
    const int64_t r = k * P + j * xc;
    const int64_t p = r + i;
    S[p] = xinv_upd_std3d(S[p], S[p + P], S[p - P], S[p + xc], S[p - xc], S[r + ip], S[r + im],
                          A[p + P], A[p], B[p + xc], B[p], C[r + ip], C[p], F[p], a.sc_);
}

// general 3-D form (numbas.invert_general_3D, numbas.py:745-984): same colouring and grid.
__global__ __launch_bounds__(256) void k_colour_gen3d(ColourArgs3D a)
{
    const int64_t nk = a.zc - 2;
    const int64_t m = a.member0 + blockIdx.z / nk;
    const int64_t k = 1 + blockIdx.z % nk;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = 1 + (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    if (j > a.yc - 2) return;
    const int64_t xc = a.xc;
    int64_t i;
    if (a.colour >= 2) {
        if (ti != 0 || ((k + j) & 1) != a.colour - 2) return;
        i = xc - 1;
    } else {
        i = 2 * ti + ((k + j + a.colour) & 1);
        const int64_t ilo = a.per ? 0 : 1;
        const int64_t ihi = a.per ? xc - 1 : xc - 2;
        if (i < ilo || i > ihi) return;
        if (a.seam && i == xc - 1) return;
    }
    const int64_t P = a.yc * xc;
    double *S = a.S + m * a.sS;
    const int64_t im = (i == 0) ? xc - 1 : i - 1;
    const int64_t ip = (i == xc - 1) ? 0 : i + 1;
    const int64_t r = k * P + j * xc;
    const int64_t p = r + i;
    double cv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) cv[q] = a.c[q][m * a.sc[q] + p];
    S[p] = xinv_upd_gen3d(S[p], S[p + P], S[p - P], S[p + xc], S[p - xc], S[r + ip], S[r + im],
                          cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], i != 0, a.sc_);
}

// ---------------------------------------------------------------- 'extend' pre-pass
// numbas.py:284-310 (2-D) / 87-115 (3-D, planes 1..zc-2).  One thread per column; reads rows
// 1 and yc-2, writes rows 0 and yc-1.  `tall` (2-D, yc > xc) reproduces what the reference's
// second loop does inside the array bounds ("extend pre-pass").
struct ExtendArgs {
    double *S;
    int64_t sS, yc, xc;
    int64_t kfirst, nk;        // planes kfirst .. kfirst+nk-1 of each member (2-D: 0, 1)
    int per, tall, force;
    double undef;
    const XinvCtl *ctl;
    int64_t member0;
};

__global__ __launch_bounds__(256) void k_extend(ExtendArgs a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.xc) return;
    const int64_t k = a.kfirst + blockIdx.y;
    const int64_t xc = a.xc, yc = a.yc;
    double *P = a.S + m * a.sS + k * yc * xc;
    double *r0 = P, *r1 = P + xc, *rm2 = P + (yc - 2) * xc, *rm1 = P + (yc - 1) * xc;
    const double u = a.undef;
    if (a.per || (i >= 1 && i <= xc - 2)) {
        double t = r1[i], b = rm2[i];
        if (t != u) r0[i] = t;
        if (b != u) rm1[i] = b;
    } else if (i == 0) {
        double t = r1[1], b = rm2[1];
        if (t != u) r0[0] = t;
        if (b != u) rm1[0] = b;
    } else {                                       // i == xc-1
        if (a.tall) {
            double t = r1[i], b = rm2[i];
            if (t != u) r0[i] = t;
            if (b != u) rm1[i] = b;
        }
        double t = r1[xc - 2], b = rm2[xc - 2];
        if (t != u) r0[i] = t;
        if (b != u) rm1[i] = b;
    }
}


// ---------------------------------------------------------------- biharmonic 2-D (Munk)
// numbas.invert_general_bih_2D (numbas.py:1204-1586).  Radius-2 stencil: 9 colours (j%3, i%3);
// with periodic x and xc % 3 != 0 the trailing xc%3 columns are 3 extra colours each.
struct ColourArgsBih {
    double *S;
    const double *c[10];       // A..J
    int64_t sS, sc[10];
    int64_t yc, xc;
    int per, trail, colour, force;
    int64_t member0;
    XinvScal sc_;
    const XinvCtl *ctl;
    unsigned umask;            // bit q: coefficient array q is constant along x (scalar per row)
    double *Y;                 // row-class kernel: where the rows of this class are written (see there)
    int64_t sY;
};

// One wavefront per row (block = 64 x 4): with UNI the row index is made wave-uniform so the
// x-uniform coefficient arrays (bit set in umask) are fetched with one scalar load per row.
template <bool UNI>
__global__ __launch_bounds__(256) void k_colour_bih2d(ColourArgsBih a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tj = (int64_t)blockIdx.y * blockDim.y + threadIdx.y;
    const int64_t xc = a.xc, yc = a.yc;
    int64_t j, i;
    const int cc = a.colour;
    const int64_t ilo = a.per ? 0 : 2, ihi = a.per ? xc - 1 : xc - 3;
    int cj;
    if (cc >= 9) {                                     // trailing-column colours
        if (ti != 0) return;
        i = xc - a.trail + (cc - 9) / 3;
        cj = (cc - 9) % 3;
    } else {
        cj = cc / 3;
        const int ci = cc % 3;
        const int64_t first = ilo + ((ci - ilo % 3) + 3) % 3;
        i = first + 3 * ti;
        if (i > ihi) return;
        if (a.trail && i >= xc - a.trail) return;
    }
    j = 2 + ((cj - 2) % 3 + 3) % 3 + 3 * tj;           // first row >= 2 with j % 3 == cj
    if (UNI) j = __builtin_amdgcn_readfirstlane((int)j);   // blockDim.x == 64: one row per wave
    if (j > yc - 3) return;

    int64_t im2 = i - 2, im1 = i - 1, ip1 = i + 1, ip2 = i + 2;
    bool edge = false;
    if (a.per) {
        if (im2 < 0) im2 += xc;
        if (im1 < 0) im1 += xc;
        if (ip1 >= xc) ip1 -= xc;
        if (ip2 >= xc) ip2 -= xc;
        edge = (i < 2) || (i >= xc - 2);
    }
    int64_t bm2 = im2;
    if (a.per && i >= xc - 2) {                        // numbas.py:1495-1497, 1540-1542
        bm2 = (i == xc - 2) ? xc - 7 : xc - 6;
        if (bm2 < 0) bm2 += xc;
    }
    double *S = a.S + m * a.sS;
    const int64_t p = j * xc + i;
    double cv[10];
#pragma unroll
    for (int q = 0; q < 10; q++) {
        if (UNI && ((a.umask >> q) & 1u)) cv[q] = a.c[q][m * a.sc[q] + j * xc];   // wave-uniform address
        else cv[q] = a.c[q][m * a.sc[q] + p];
    }
    const double *r0 = S + j * xc;
    S[p] = xinv_upd_bih2d(r0, r0 + xc, r0 + 2 * xc, r0 - xc, r0 - 2 * xc, i, im2, im1, ip1, ip2,
                          bm2, edge, cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7],
                          cv[8], cv[9], a.sc_);
}


// ---- biharmonic: the three i-colours of one row class in ONE launch ---------------------------
// A wavefront owns one row j of the class (j % 3 == a.colour) and a strip of 192 columns, three
// adjacent columns per lane = the three colours i % 3.  Rows j+-1, j+-2 belong to other row
// classes and are constant during the launch; the lane's columns of row j are updated colour
// after colour in registers, the columns of the neighbouring lanes coming in by DPP shifts after
// every stage.  Two lanes on each side are halo (each later colour needs the earlier colours two
// columns away), so a strip owns 180 columns.  Same ordering as nine separate colour launches
// (bitwise equal), one third of the launches and of the passes over S.
// NOT in place: the halo lanes of a strip read columns of row j that the neighbouring strip's
// wavefront updates in the same launch, so the updated rows go to a side buffer Y (every owned
// column, updatable or not) and later launches of the sweep read a neighbour row from Y when its
// class has already run (rows 2..yc-3 of a lower class), from S otherwise; k_rows_copy_back moves
// rows 2..yc-3 of Y into S after the third launch.
// Periodic x (xc % 3 == 0, so that the wrap keeps colour == column % 3): the lane->column map
// wraps, the first/last two columns take the periodic branches' G-term association, and the two
// east columns read the B term's west operand FIVE columns away (the reference's stale loop index,
// numbas.py:1495-1497, 1540-1542) -- one and two lanes to the left, in the state the 9-colour
// order gives them (column xc-7 has colour 2: not yet updated; column xc-6 colour 0: updated).
template <bool UNI, bool PER>
__global__ __launch_bounds__(256) void k_bih_rowclass(ColourArgsBih a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t strip = (int64_t)blockIdx.x * 4 + wave;
    const int64_t xc = a.xc, yc = a.yc;
    const int cj = a.colour;
    int64_t j = 2 + ((cj - 2) % 3 + 3) % 3 + 3 * (int64_t)blockIdx.y;
    if (j > yc - 3) return;
    const int64_t c = strip * 180 - 6 + 3 * lane;              // unwrapped first column (c % 3 == 0)
    if (strip * 180 >= xc) return;
    const double *S = a.S + m * a.sS;
    double *Y = a.Y + m * a.sY;
    int64_t lcol[3];
    bool upd[3], own[3], edge[3], east[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int64_t cc = c + k;
        if (PER) {
            int64_t w = cc % xc; if (w < 0) w += xc;
            lcol[k] = w;
            upd[k] = true;
            edge[k] = (w < 2) || (w >= xc - 2);
            east[k] = (w >= xc - 2);
        } else {
            lcol[k] = cc < 0 ? 0 : (cc > xc - 1 ? xc - 1 : cc);
            upd[k] = (cc >= 2) && (cc <= xc - 3);
            edge[k] = false; east[k] = false;
        }
        own[k] = (cc >= 0) && (cc < xc) && (lane >= 2) && (lane < 62);
    }
    Tri R[5];                                                   // rows j-2 .. j+2
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int64_t jr = j - 2 + q;
        // a row of a lower class has been rewritten in this sweep: its current values are in Y
        const bool newer = (jr >= 2) && (jr <= yc - 3) && ((int)(jr % 3) < cj);
        const double *row = (newer ? (const double *)Y : S) + jr * xc;
#pragma unroll
        for (int k = 0; k < 3; k++) R[q].v[k] = row[lcol[k]];
    }
    double cv[10][3];
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const double *cr = a.c[q] + m * a.sc[q] + j * xc;
        if (UNI && ((a.umask >> q) & 1u)) {
            const double t = cr[0];
            cv[q][0] = t; cv[q][1] = t; cv[q][2] = t;
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) cv[q][k] = cr[lcol[k]];
        }
    }
    double em2[7], em1[7], ep1[7], ep2[7];
    bih_ext(R[0], em2); bih_ext(R[1], em1); bih_ext(R[3], ep1); bih_ext(R[4], ep2);
    double fm2[2] = {0.0, 0.0}, fp2[2] = {0.0, 0.0};            // columns c-4, c-3 of rows j-2, j+2
    if (PER) { bih_far(R[0], fm2[0], fm2[1]); bih_far(R[4], fp2[0], fp2[1]); }
#pragma unroll
    for (int k = 0; k < 3; k++) {                               // colour i % 3 == k
        double e0[7];
        bih_ext(R[2], e0);
        const int o = k + 2;                                    // index of column c+k in e[]
        double p2_b = ep2[o - 2], r0_b = e0[o - 2], m2_b = em2[o - 2];
        if (PER && k > 0) {
            double f0[2];
            bih_far(R[2], f0[0], f0[1]);
            if (east[k]) { p2_b = fp2[k - 1]; r0_b = f0[k - 1]; m2_b = fm2[k - 1]; }
        }
        R[2].v[k] = xinv_upd_bih2d_v(
            ep2[o], ep2[o + 2], p2_b, ep1[o], ep1[o + 1], ep1[o - 1],
            e0[o], e0[o + 1], e0[o - 1], e0[o + 2], e0[o - 2], r0_b,
            em1[o], em1[o + 1], em1[o - 1], em2[o], em2[o + 2], m2_b,
            cv[0][k], cv[1][k], cv[2][k], cv[3][k], cv[4][k], cv[5][k], cv[6][k], cv[7][k],
            cv[8][k], cv[9][k], upd[k], edge[k], a.sc_);
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (own[k]) Y[j * xc + lcol[k]] = R[2].v[k];
}

// rows 2..yc-3 of the side buffer back into S (after the three row-class launches of a sweep)
__global__ __launch_bounds__(256) void k_rows_copy_back(const double *Y, int64_t sY, double *S, int64_t sS,
                                                        int64_t yc, int64_t xc, const XinvCtl *ctl,
                                                        int64_t member0, int force)
{
    const int64_t m = member0 + blockIdx.y;
    if (!force && xinv_ctl_done(ctl + m)) return;
    const int64_t n = (yc - 4) * xc;
    const double *y = Y + m * sY + 2 * xc;
    double *s = S + m * sS + 2 * xc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        s[t] = y[t];
}

// 'extend' pre-pass of the biharmonic kernel (numbas.py:1299-1343): one thread per column.
__global__ __launch_bounds__(256) void k_extend_bih(ExtendArgs a)
{
    const int64_t m = a.member0 + blockIdx.z;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.xc) return;
    const int64_t xc = a.xc, yc = a.yc;
    double *P = a.S + m * a.sS;
    double *r0 = P, *r1 = P + xc, *r2 = P + 2 * xc;
    double *b1 = P + (yc - 1) * xc, *b2 = P + (yc - 2) * xc, *b3 = P + (yc - 3) * xc;
    const double u = a.undef;
    if (a.per) {
        if (r2[i] != u) { r0[i] = r1[i]; r1[i] = r2[i]; }
        if (b3[i] != u) { b1[i] = b3[i]; b2[i] = b3[i]; }
        return;
    }
    // non-periodic: loops over 1..xc-2 (and xc-1 when yc > xc), then the 2x2 corner blocks
    double t0 = r0[i], t1 = r1[i], q1 = b1[i], q2 = b2[i];
    const bool inloop = (i >= 1 && i <= xc - 2) || (i == xc - 1 && a.tall);
    if (inloop) {
        if (r2[i] != u) { t0 = r2[i]; t1 = r2[i]; }
        if (b3[i] != u) { q1 = b3[i]; q2 = b3[i]; }
    }
    if (i <= 1) {
        if (r2[2] != u) { t0 = r2[2]; t1 = r2[2]; }
        if (b3[2] != u) { q1 = b3[2]; q2 = b3[2]; }
    }
    if (i >= xc - 2) {
        if (r2[xc - 3] != u) { t0 = r2[xc - 3]; t1 = r2[xc - 3]; }
        if (b3[xc - 3] != u) { q1 = b3[xc - 3]; q2 = b3[xc - 3]; }
    }
    r0[i] = t0; r1[i] = t1; b1[i] = q1; b2[i] = q2;
}

// ------------------------------------------------------------------------------ norm
// Stage 1: per-workgroup (sum |S|, count) over S != undef in a fixed element -> thread map.
// Stage 2: one workgroup per member adds the partials in index order and applies the
// stopping rule.  Both stages are order-deterministic (no floating-point atomics).
#define XINV_NORM_BLOCKS 512

struct NormArgs {
    const double *S;
    int64_t sS, n;
    double undef;
    double *psum;              // [nbatch][XINV_NORM_BLOCKS]
    long long *pcnt;
    XinvCtl *ctl;
    XinvStop stop;
    int force;
    int64_t member0;
};

__global__ __launch_bounds__(256) void k_norm_partial(NormArgs a)
{
    const int64_t m = a.member0 + blockIdx.y;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    const double *S = a.S + m * a.sS;
    double s = 0.0;
    long long c = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.n;
         t += (int64_t)gridDim.x * blockDim.x) {
        double v = S[t];
        if (v != a.undef) { s += fabs(v); c += 1; }
    }
    s = xinv_wave_sum(s);
    c = xinv_wave_sum_ll(c);
    __shared__ double ls[4];
    __shared__ long long lc[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ls[w] = s; lc[w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0; long long tc = 0;
        for (int q = 0; q < 4; q++) { ts += ls[q]; tc += lc[q]; }
        a.psum[m * XINV_NORM_BLOCKS + blockIdx.x] = ts;
        a.pcnt[m * XINV_NORM_BLOCKS + blockIdx.x] = tc;
    }
}

__global__ __launch_bounds__(64) void k_norm_final(NormArgs a, int nblocks)
{
    const int64_t m = a.member0 + blockIdx.x;
    if (!a.force && xinv_ctl_done(a.ctl + m)) return;
    double s = 0.0;
    long long c = 0;
    for (int t = threadIdx.x; t < nblocks; t += 64) {
        s += a.psum[m * XINV_NORM_BLOCKS + t];
        c += a.pcnt[m * XINV_NORM_BLOCKS + t];
    }
    s = xinv_wave_sum(s);
    c = xinv_wave_sum_ll(c);
    if (threadIdx.x == 0) xinv_ctl_update(&a.ctl[m], s, c, a.stop);
}

// Stand-alone mean|S| (xinv_abs_norm_f64_dev): same two stages, result written to out[0..1].
__global__ __launch_bounds__(64) void k_norm_out(const double *psum, const long long *pcnt,
                                                 int nblocks, double *out)
{
    double s = 0.0;
    long long c = 0;
    for (int t = threadIdx.x; t < nblocks; t += 64) { s += psum[t]; c += pcnt[t]; }
    s = xinv_wave_sum(s);
    c = xinv_wave_sum_ll(c);
    if (threadIdx.x == 0) out[0] = (c != 0) ? s / (double)c : NAN;
}

// any(B != 0) over n elements -> *flag (int), used once per solve to pick the colouring.
__global__ __launch_bounds__(256) void k_any_nonzero(const double *B, int64_t n, int *flag)
{
    bool nz = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x)
        nz |= (B[t] != 0.0);
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// row-constant coefficient arrays (xinv_options.rowconst_mask): out[m][row][0..xc) = in[m][row]
__global__ __launch_bounds__(256) void k_expand_rows(const double *in, double *out, int64_t rows,
                                                     int64_t xc, int64_t nmem)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows * nmem) return;
    const double v = in[row];
    double *o = out + row * xc;
    for (int64_t i = threadIdx.x & 63; i < xc; i += 64) o[i] = v;
}

// float32 host arrays (xinv_options.f32_mask): promoted on the device after the upload (exact), S demoted before the
// download (round to nearest even: what assigning a float64 into a float32 array does)
__global__ __launch_bounds__(256) void k_promote_f32(const float *in, double *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (double)in[i];
}
__global__ __launch_bounds__(256) void k_demote_f64(const double *in, float *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i];
}

// ---- front-end passes on the device (xinv_options.prep_flags) --------------------------------
// The forcing as the caller holds it -> the forcing the kernels read: masked points (NaN, or equal
// to the caller's undefined value) become `undef_tmp`, defined ones are multiplied by a per-row
// factor (cos(lat)) -- reference apps.__mask_FS (apps.py:2112-2159) followed by the builders'
// `F * cos(lat)` and re-mask (apps.py:1409-1411, 2036-2038).  One pass at HBM rate instead of
// four numpy passes over the host array.
__global__ __launch_bounds__(256) void k_prep_forcing(double *F, int64_t n, int64_t yc, int64_t xc,
                                                      const double *rowscale, int mask_nan,
                                                      double undef_in, double undef_tmp)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = F[i];
        const bool masked = (mask_nan ? isnan(v) : (v == undef_in)) || (v == undef_tmp);
        double w = v;
        if (rowscale) w = v * rowscale[(i / xc) % yc];
        F[i] = masked ? undef_tmp : w;
    }
}

// output de-mask (apps.py:1389-1392): S = `value` where the forcing is masked
__global__ __launch_bounds__(256) void k_demask(double *S, const double *F, int64_t n, double undef_tmp,
                                                double value)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (F[i] == undef_tmp) S[i] = value;
}

// control blocks at the start of a solve + the norm partials cleared (n16 uint4 words; pbytes is a multiple of 256), grid-stride
__global__ __launch_bounds__(256) void k_solve_init(XinvCtl *ctl, int64_t nbatch, uint4 *part, int64_t n16)
{
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    for (int64_t m = t0; m < nbatch; m += nt) {
        XinvCtl c;
        c.normPrev = DBL_MAX;
        c.flag1 = 0.0; c.flag2 = 0.0;
        c.loop = 0; c.sweeps = 0;
        c.done = 0; c.overflow = 0; c.wrote = 0; c.ticket = 0; c.seq = 1; c.pad_ = 0;
        ctl[m] = c;
    }
    for (int64_t i = t0; i < n16; i += nt) part[i] = make_uint4(0u, 0u, 0u, 0u);
}

// The control blocks of a SHORT solve handed to the host without a copy engine and without a stream synchronisation
// (run_sweeps): written into the pinned host mirror by the device, then a sequence word with system-scope release order --
// the host spins on that word (a frame of apps.animate_iteration: 42 -> 36-38 us, profiles/r05_animate.txt).
__global__ __launch_bounds__(64) void k_ctl_mail(const XinvCtl *ctl, int64_t nbatch, XinvCtl *host, unsigned *seq, unsigned val)
{
    for (int64_t m = threadIdx.x; m < nbatch; m += 64) host[m] = ctl[m];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Watchdog recovery (run_sweeps): the member goes on from the control state the timed-out reducer left untouched.
__global__ void k_ctl_resume(XinvCtl *c) { c->overflow = 0; c->done = 0; }
#if XINV_TEST_HOOKS
// Test hook (XINV_EXP_WATCHDOG, test-hooks build only): the control state a reducer that timed out leaves behind -- it
// stops the member without having applied the stop rule to any sweep of its launch.
__global__ void k_ctl_fake_timeout(XinvCtl *c)
{
    if (!c->done) { c->overflow = 2; c->done = 1; c->sweeps = c->loop + 1; }
}
#endif

// ------------------------------------------------------------------ Gill-Matsuno flow (u, v)
// reference apps.cal_flow(vtype='GillMatsuno') (apps.py:1277-1317): pointwise combination of
// the two first derivatives of the inverted mass field, computed exactly as xarray's
// .differentiate / numpy.gradient(edge_order=1) does -- uniform-spacing form when the coordinate
// differences are all equal, 3-point non-uniform form otherwise -- so that the device result is
// bitwise the host restatement's.
struct GradAxis {
    const double *a, *b, *c;   // non-uniform interior weights (numpy: a*f[i-1] + b*f[i] + c*f[i+1])
    double dx, dx0, dxn;       // uniform spacing; first / last spacing for the one-sided edges
    int uniform;
};

struct FlowArgs {
    const double *S;
    double *u, *v;
    int64_t nbatch, yc, xc;
    GradAxis gy, gx;
    const double *coef1, *coef2, *cosl;   // per row
    double deg2m;
    int latlon;
};

__device__ __forceinline__ double xinv_grad1(const GradAxis &g, double fm, double f0, double fp,
                                              int64_t i, int64_t n)
{
    if (i == 0) return (fp - f0) / g.dx0;
    if (i == n - 1) return (f0 - fm) / g.dxn;
    if (g.uniform) return (fp - fm) / (2. * g.dx);
    return g.a[i] * fm + g.b[i] * f0 + g.c[i] * fp;
}

__global__ __launch_bounds__(256) void k_gm_flow(FlowArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = blockIdx.y;
    const int64_t m = blockIdx.z;
    if (i >= a.xc) return;
    const int64_t xc = a.xc, yc = a.yc;
    const double *S = a.S + m * yc * xc;
    const int64_t p = j * xc + i;
    const double f0 = S[p];
    const double fxm = S[j * xc + (i > 0 ? i - 1 : i)], fxp = S[j * xc + (i < xc - 1 ? i + 1 : i)];
    const double fym = S[(j > 0 ? j - 1 : j) * xc + i], fyp = S[(j < yc - 1 ? j + 1 : j) * xc + i];
    const double Sx = xinv_grad1(a.gx, fxm, f0, fxp, i, xc);
    const double Sy = xinv_grad1(a.gy, fym, f0, fyp, j, yc);
    const double c1 = a.coef1[j], c2 = a.coef2[j];
    double u, v;
    if (a.latlon) {
        const double cl = a.cosl[j];
        u = - c1 * Sx / a.deg2m / cl - c2 * Sy / a.deg2m;
        v = - c1 * Sy / a.deg2m + c2 * Sx / a.deg2m / cl;
    } else {
        u = - c1 * Sx - c2 * Sy;
        v = - c1 * Sy + c2 * Sx;
    }
    a.u[m * yc * xc + p] = u;
    a.v[m * yc * xc + p] = v;
}
