// xinv_tu_pipe2d.hip -- instantiations of k_pipe2d (wave-pipelined four-sweep pass, xinv_pipe2d.h).
#include "xinv_dispatch.h"

template <class M, int NP, bool AL, bool EXT>
static int pipe_one(dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_pipe2d<M, NP, AL, EXT>, 64 * XINV_PIPE_P, 0) != hipSuccess)
            n = 1;
        *occ = n < 1 ? 1 : n;
        return 0;
    }
    hipLaunchKernelGGL((k_pipe2d<M, NP, AL, EXT>), grid, dim3(64 * XINV_PIPE_P, 1, 1), (size_t)lds_pad, st, a);
    return 0;
}

template <class M, int NP>
static int pipe_np(bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (al) return ext ? pipe_one<M, NP, true, true>(grid, st, a, occ, lds_pad) : pipe_one<M, NP, true, false>(grid, st, a, occ, lds_pad);
    return ext ? pipe_one<M, NP, false, true>(grid, st, a, occ, lds_pad) : pipe_one<M, NP, false, false>(grid, st, a, occ, lds_pad);
}

int xinv_launch_pipe2d(bool gen, int np, bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (gen) return pipe_np<FusedGen2D, 1>(al, ext, grid, st, a, occ, lds_pad);      // (one column pair per lane only)
    return np == 2 ? pipe_np<FusedStd2D, 2>(al, ext, grid, st, a, occ, lds_pad) : pipe_np<FusedStd2D, 1>(al, ext, grid, st, a, occ, lds_pad);
}
