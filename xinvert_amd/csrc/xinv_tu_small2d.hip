// xinv_tu_small2d.hip -- instantiations of k_small2d (register-resident solver for small slices).
#include "xinv_dispatch.h"

size_t xinv_small2d_lds(bool gen, int NW, int RW, int NSEG, int64_t yc, int64_t xc)
{
    const size_t nr = gen ? SmallGen::NR : SmallStd::NR;
    return (size_t)(NW + 2) * 2 * NSEG * 2 * 64 * 8            // edge rows
         + (size_t)NW * RW * (nr * 8 + 4) + (size_t)NW * 16 + 64 // row table, norm partials
         + (size_t)yc * 2 * ((xc + 1) / 2) * 8;                  // forcing
}

template <class M>
static int launch_small_m(int NW, int RW, int NSEG, dim3 grid, hipStream_t st, const SmallArgs &a)
{
    const size_t lds = (size_t)a.yc * 2 * ((a.xc + 1) / 2) * sizeof(double);     // the forcing (dynamic part)
#define V1(W, R, N, E) do { \
        if (hipFuncSetAttribute((const void *)k_small2d<M, W, R, N, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2; \
        hipLaunchKernelGGL((k_small2d<M, W, R, N, E>), grid, dim3(W * 64, 1, 1), lds, st, a); } while (0)
#define V(W, R, N) if (NW == W && RW == R && NSEG == N) { if (a.ext) V1(W, R, N, true); else V1(W, R, N, false); return 0; }
    V(16, 2, 1) V(16, 4, 1) V(16, 6, 1) V(16, 2, 2) V(16, 4, 2) V(16, 6, 2) V(16, 2, 3) V(16, 4, 3)
    V(8, 4, 1) V(8, 6, 1) V(8, 8, 1) V(8, 10, 1) V(8, 12, 1)
    V(8, 4, 2) V(8, 6, 2) V(8, 8, 2) V(8, 10, 2)
    V(8, 4, 3) V(8, 6, 3)
#undef V
#undef V1
    return 1;
}

int xinv_launch_small2d(bool gen, int NW, int RW, int NSEG, dim3 grid, hipStream_t st, const SmallArgs &a)
{
    return gen ? launch_small_m<SmallGen>(NW, RW, NSEG, grid, st, a) : launch_small_m<SmallStd>(NW, RW, NSEG, grid, st, a);
}
