// xinv_tu_pipe2d.hip -- instantiations of k_pipe2d (wave-pipelined four-sweep pass, xinv_pipe2d.h).
#include "xinv_dispatch.h"

template <int NP, bool AL, bool EXT>
static int pipe_one(dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_pipe2d<NP, AL, EXT>, 64 * XINV_PIPE_P, 0) != hipSuccess)
            n = 1;
        *occ = n < 1 ? 1 : n;
        return 0;
    }
    hipLaunchKernelGGL((k_pipe2d<NP, AL, EXT>), grid, dim3(64 * XINV_PIPE_P, 1, 1), (size_t)lds_pad, st, a);
    return 0;
}

template <int NP>
static int pipe_np(bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    if (al) return ext ? pipe_one<NP, true, true>(grid, st, a, occ, lds_pad) : pipe_one<NP, true, false>(grid, st, a, occ, lds_pad);
    return ext ? pipe_one<NP, false, true>(grid, st, a, occ, lds_pad) : pipe_one<NP, false, false>(grid, st, a, occ, lds_pad);
}

int xinv_launch_pipe2d(int np, bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ, int lds_pad)
{
    return np == 2 ? pipe_np<2>(al, ext, grid, st, a, occ, lds_pad) : pipe_np<1>(al, ext, grid, st, a, occ, lds_pad);
}
