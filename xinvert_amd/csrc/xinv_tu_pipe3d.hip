// xinv_tu_pipe3d.hip -- instantiations of k_pipe3d (two sweeps per pass, pipelined across two groups of wavefronts:
// xinv_pipe3d.h), plain and contracted (XINV_FLAG_FMA).  Compiled with -amdgpu-sched-strategy=iterative-minreg
// (xinvert_amd/build.py): sixteen wavefronts leave 128 VGPRs, and with the default scheduler the variant that carries
// the forcing through the LDS ring spilled 11-30 of them.
#include "xinv_dispatch.h"

int xinv_launch_pipe3d(bool al, dim3 grid, hipStream_t st, const Fused3Args &a, bool seam, bool ext)
{
    constexpr int G = XINV_P3_G, RR = XINV_P3_RR;
    if (ext) {                                           // BCy = 'extend' (never with the seam: the planner keeps the one-sweep kernel)
        if (al) hipLaunchKernelGGL((k_pipe3d<G, RR, true, false, false, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
        else    hipLaunchKernelGGL((k_pipe3d<G, RR, false, false, false, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
        return 0;
    }
    if (seam) hipLaunchKernelGGL((k_pipe3d<G, RR, false, false, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);   // (periodic x, odd xc)
    else if (al) hipLaunchKernelGGL((k_pipe3d<G, RR, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
    else    hipLaunchKernelGGL((k_pipe3d<G, RR, false>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
    return 0;
}

int xinv_launch_pipe3d_fma(bool al, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    constexpr int G = XINV_P3_G, RR = XINV_P3_RR;
    if (al) hipLaunchKernelGGL((k_pipe3d<G, RR, true, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
    else    hipLaunchKernelGGL((k_pipe3d<G, RR, false, true>), grid, dim3(2 * G * 64, 1, 1), 0, st, a);
    return 0;
}
