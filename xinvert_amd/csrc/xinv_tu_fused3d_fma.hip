// xinv_tu_fused3d_fma.hip -- the contracted-arithmetic variants (XINV_FLAG_FMA) of the 3-D standard-form kernels:
// k_fused3d with x-uniform coefficients (k_pipe3d's: xinv_tu_pipe3d.hip).
#include "xinv_dispatch.h"

template <int NW>
static int launch_fused3d_fma_nw(bool al, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    dim3 block(NW * 64, 1, 1);
    if (al) { if (ext) hipLaunchKernelGGL((k_fused3d<NW, true, true, true, true>), grid, block, 0, st, a);
              else     hipLaunchKernelGGL((k_fused3d<NW, true, true, false, true>), grid, block, 0, st, a); }
    else    { if (ext) hipLaunchKernelGGL((k_fused3d<NW, false, true, true, true>), grid, block, 0, st, a);
              else     hipLaunchKernelGGL((k_fused3d<NW, false, true, false, true>), grid, block, 0, st, a); }
    return 0;
}

int xinv_launch_fused3d_fma(int NW, bool al, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    if (NW == 8) return launch_fused3d_fma_nw<8>(al, ext, grid, st, a);
    if (NW == 16 && !ext) {                              // (as the plain kernel: sixteen wavefronts without 'extend' only)
        if (al) hipLaunchKernelGGL((k_fused3d<16, true, true, false, true>), grid, dim3(16 * 64, 1, 1), 0, st, a);
        else    hipLaunchKernelGGL((k_fused3d<16, false, true, false, true>), grid, dim3(16 * 64, 1, 1), 0, st, a);
        return 0;
    }
    if (NW != 12) return 1;
    return launch_fused3d_fma_nw<12>(al, ext, grid, st, a);
}
