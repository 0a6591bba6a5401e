// xinv_tu_pipe2d.hip -- instantiations of k_pipe2d (wave-pipelined four-sweep pass, xinv_pipe2d.h).
#include "xinv_dispatch.h"

template <bool AL, bool EXT>
static int pipe_one(dim3 grid, hipStream_t st, const FusedArgs &a, int *occ)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_pipe2d<AL, EXT>, 64 * XINV_PIPE_P, 0) != hipSuccess)
            n = 1;
        *occ = n < 1 ? 1 : n;
        return 0;
    }
    hipLaunchKernelGGL((k_pipe2d<AL, EXT>), grid, dim3(64 * XINV_PIPE_P, 1, 1), 0, st, a);
    return 0;
}

int xinv_launch_pipe2d(bool al, bool ext, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ)
{
    if (al) return ext ? pipe_one<true, true>(grid, st, a, occ) : pipe_one<true, false>(grid, st, a, occ);
    return ext ? pipe_one<false, true>(grid, st, a, occ) : pipe_one<false, false>(grid, st, a, occ);
}
