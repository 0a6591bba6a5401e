// xinv_tu_pipe2d.hip -- instantiations of k_pipe2d (wave-pipelined four-sweep pass, xinv_pipe2d.h) for ONE model
// (compiled twice: -DXINV_TU_MODEL=0 standard form, 1 general form; each once more with -DXINV_TU_SEAM=1: the odd-xc
// periodic seam variants -- unaligned strips, one column pair per lane).
#include "xinv_dispatch.h"

#ifndef XINV_TU_SEAM
#define XINV_TU_SEAM 0
#endif
constexpr bool SEAM = XINV_TU_SEAM != 0;

template <class M, unsigned UM, bool FR, int NP, bool AL, bool EXT>
static int pipe_one(dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (occ) {
        static int cached = 0;                           // (asked by the planner in every solve: a few microseconds per query)
        int n = cached;
        if (!n) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_pipe2d<M, UM, FR, NP, AL, EXT, SEAM>, 64 * XINV_PIPE_P, 0) != hipSuccess)
                n = 1;
            cached = n = n < 1 ? 1 : n;
        }
        *occ = n;
        return 0;
    }
    hipLaunchKernelGGL((k_pipe2d<M, UM, FR, NP, AL, EXT, SEAM>), grid, dim3(64 * XINV_PIPE_P, 1, 1), (size_t)lds_pad, st, a);
    return 0;
}

template <class M, unsigned UM, bool FR, int NP>
static int pipe_np(bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if constexpr (!SEAM) {
        if (al) return ext ? pipe_one<M, UM, FR, NP, true, true>(grid, st, a, occ, lds_pad) : pipe_one<M, UM, FR, NP, true, false>(grid, st, a, occ, lds_pad);
    } else if (al) return 1;
    return ext ? pipe_one<M, UM, FR, NP, false, true>(grid, st, a, occ, lds_pad) : pipe_one<M, UM, FR, NP, false, false>(grid, st, a, occ, lds_pad);
}

#if XINV_TU_SEAM
#define xinv_launch_pipe2d_std xinv_launch_pipe2d_std_seam
#define xinv_launch_pipe2d_gen xinv_launch_pipe2d_gen_seam
#endif
#if XINV_TU_MODEL == 2        /* contracted arithmetic (XINV_FLAG_FMA), both forms in one unit: one column pair per lane */
int xinv_launch_pipe2d_fma(bool gen, unsigned um, bool fr, bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (!gen && um == 3u) return fr ? pipe_np<FusedStd2DF, 3u, true, 1>(al, ext, grid, st, a, occ, lds_pad)
                                    : pipe_np<FusedStd2DF, 3u, false, 1>(al, ext, grid, st, a, occ, lds_pad);
    if (gen && um == 0x1fu) return fr ? pipe_np<FusedGen2DF, 0x1fu, true, 1>(al, ext, grid, st, a, occ, lds_pad)
                                      : pipe_np<FusedGen2DF, 0x1fu, false, 1>(al, ext, grid, st, a, occ, lds_pad);
    return 1;
}
#elif XINV_TU_MODEL == 0
int xinv_launch_pipe2d_std(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    // (two column pairs per lane -- NP = 2 -- were measured slower in round 2 and are no longer instantiated)
    if (um == 3u && np == 1) return fr ? pipe_np<FusedStd2D, 3u, true, 1>(al, ext, grid, st, a, occ, lds_pad)
                                       : pipe_np<FusedStd2D, 3u, false, 1>(al, ext, grid, st, a, occ, lds_pad);
    return 1;
}
#else
int xinv_launch_pipe2d_gen(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    (void)np;                                            // (one column pair per lane only)
    if (um == 0x1fu) return fr ? pipe_np<FusedGen2D, 0x1fu, true, 1>(al, ext, grid, st, a, occ, lds_pad)
                               : pipe_np<FusedGen2D, 0x1fu, false, 1>(al, ext, grid, st, a, occ, lds_pad);
    return 1;
}
#endif
