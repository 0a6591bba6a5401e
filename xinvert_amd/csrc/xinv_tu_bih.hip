// xinv_tu_bih.hip -- instantiations of k_fusedbih (one-pass biharmonic kernel).
#include "xinv_dispatch.h"

// vm: 0 = A..I per row (records); 1 = A, C, D, F vector streams (B == E == 0: zbe); 3 = ... with C read out of A, F out of D;
//     2 = all nine vector streams
template <int VM>
static int launch_vm(bool per, bool zbe, dim3 grid, hipStream_t st, const FusedBihArgs &a, int *occ)
{
    if (occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fusedbih<false, VM == 1 || VM == 3, VM>, 256, 0) != hipSuccess || n < 1) n = 1;
        *occ = n;
        return 0;
    }
    dim3 block(256, 1, 1);
    if (zbe || VM == 1 || VM == 3) {
        if (per) hipLaunchKernelGGL((k_fusedbih<true, true, VM>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_fusedbih<false, true, VM>), grid, block, 0, st, a);
    } else if constexpr (VM != 1 && VM != 3) {
        if (per) hipLaunchKernelGGL((k_fusedbih<true, false, VM>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_fusedbih<false, false, VM>), grid, block, 0, st, a);
    }
    return 0;
}

int xinv_launch_fusedbih(bool per, bool zbe, int vm, dim3 grid, hipStream_t st, const FusedBihArgs &a, int *occ)
{
    if (vm == 1) return launch_vm<1>(per, true, grid, st, a, occ);
    if (vm == 3) return launch_vm<3>(per, true, grid, st, a, occ);
    if (vm == 2) return launch_vm<2>(per, zbe, grid, st, a, occ);
    return launch_vm<0>(per, zbe, grid, st, a, occ);
}
