// xinv_tu_fused3d.hip -- instantiations of k_fused3d (standard 3-D form) and k_fused3dg (general 3-D form with x-uniform
// coefficients).  k_pipe3d (two sweeps per pass) has its own unit, xinv_tu_pipe3d.hip.
#include "xinv_dispatch.h"

// ---- 3-D fused launch ------------------------------------------------------------------------
template <int NW>
static int launch_fused3d_nw(bool al, bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    dim3 block(NW * 64, 1, 1);
#define L3(AL, UNI, EXT) hipLaunchKernelGGL((k_fused3d<NW, AL, UNI, EXT>), grid, block, 0, st, a)
    if (al) {
        if (uni) { if (ext) L3(true, true, true); else L3(true, true, false); }
        else     { if (ext) L3(true, false, true); else L3(true, false, false); }
    } else {
        if (uni) { if (ext) L3(false, true, true); else L3(false, true, false); }
        else     { if (ext) L3(false, false, true); else L3(false, false, false); }
    }
#undef L3
    return 0;
}


int xinv_launch_fused3d(int NW, bool al, bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    if (NW == 8) return launch_fused3d_nw<8>(al, uni, ext, grid, st, a);
    if (NW == 16 && uni && !ext) {
        // sixteen wavefronts leave 128 VGPRs per lane: enough for the x-uniform variant without 'extend' only (the
        // other sixteen-wavefront variants spilled 76-203 bytes per lane and are not instantiated: the planner
        // gives them twelve)
        if (al) hipLaunchKernelGGL((k_fused3d<16, true, true, false>), grid, dim3(16 * 64, 1, 1), 0, st, a);
        else    hipLaunchKernelGGL((k_fused3d<16, false, true, false>), grid, dim3(16 * 64, 1, 1), 0, st, a);
        return 0;
    }
    if (NW != 12) return 1;
    return launch_fused3d_nw<12>(al, uni, ext, grid, st, a);
}

// ---- general 3-D fused launch (every coefficient array x-uniform) --------------------------------
template <int NW>
static void launch_fused3dg_nw(bool al, bool ext, dim3 grid, hipStream_t st, const Fused3GArgs &a)
{
    dim3 block(NW * 64, 1, 1);
    if (al) { if (ext) hipLaunchKernelGGL((k_fused3dg<NW, true, true>), grid, block, 0, st, a);
              else     hipLaunchKernelGGL((k_fused3dg<NW, true, false>), grid, block, 0, st, a); }
    else    { if (ext) hipLaunchKernelGGL((k_fused3dg<NW, false, true>), grid, block, 0, st, a);
              else     hipLaunchKernelGGL((k_fused3dg<NW, false, false>), grid, block, 0, st, a); }
}


int xinv_launch_fused3dg(int NW, bool al, bool ext, dim3 grid, hipStream_t st, const Fused3GArgs &a)
{
    if (NW == 8) launch_fused3dg_nw<8>(al, ext, grid, st, a);      // (sixteen wavefronts: 128 VGPRs, spills -- not instantiated)
    else if (NW == 12) launch_fused3dg_nw<12>(al, ext, grid, st, a);
    else return 1;
    return 0;
}
