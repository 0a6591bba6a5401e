// xinv_tu_bih.hip -- instantiations of k_fusedbih (one-pass biharmonic kernel).
#include "xinv_dispatch.h"

int xinv_launch_fusedbih(bool per, bool zbe, dim3 grid, hipStream_t st, const FusedBihArgs &a, int *occ)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fusedbih<false, false>, 256, 0) != hipSuccess || n < 1) n = 1;
        *occ = n;
        return 0;
    }
    dim3 block(256, 1, 1);
    if (zbe) {
        if (per) hipLaunchKernelGGL((k_fusedbih<true, true>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_fusedbih<false, true>), grid, block, 0, st, a);
    } else {
        if (per) hipLaunchKernelGGL((k_fusedbih<true, false>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_fusedbih<false, false>), grid, block, 0, st, a);
    }
    return 0;
}
