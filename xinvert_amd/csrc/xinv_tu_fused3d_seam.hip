// xinv_tu_fused3d_seam.hip -- k_fused3d / k_fused3dg with the odd-xc periodic seam inside the kernel (SEAM variants: unaligned strips,
// the even-ring layout; xinv_fused3d.h).
#include "xinv_dispatch.h"

template <int NW>
static int launch_fused3d_seam_nw(bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    dim3 block(NW * 64, 1, 1);
#define L3(UNI, EXT) hipLaunchKernelGGL((k_fused3d<NW, false, UNI, EXT, false, true>), grid, block, 0, st, a)
    if (uni) { if (ext) L3(true, true); else L3(true, false); }
    else     { if (ext) L3(false, true); else L3(false, false); }
#undef L3
    return 0;
}

int xinv_launch_fused3d_seam(int NW, bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a)
{
    if (NW == 8) return launch_fused3d_seam_nw<8>(uni, ext, grid, st, a);
    if (NW == 12) return launch_fused3d_seam_nw<12>(uni, ext, grid, st, a);
    return 1;
}

// general 3-D form (x-uniform coefficients)
template <int NW>
static void launch_fused3dg_seam_nw(bool ext, dim3 grid, hipStream_t st, const Fused3GArgs &a)
{
    dim3 block(NW * 64, 1, 1);
    if (ext) hipLaunchKernelGGL((k_fused3dg<NW, false, true, true>), grid, block, 0, st, a);
    else     hipLaunchKernelGGL((k_fused3dg<NW, false, false, true>), grid, block, 0, st, a);
}

int xinv_launch_fused3dg_seam(int NW, bool ext, dim3 grid, hipStream_t st, const Fused3GArgs &a)
{
    if (NW == 8) launch_fused3dg_seam_nw<8>(ext, grid, st, a);
    else if (NW == 12) launch_fused3dg_seam_nw<12>(ext, grid, st, a);
    else return 1;
    return 0;
}
