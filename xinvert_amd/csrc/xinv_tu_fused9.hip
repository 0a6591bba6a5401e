// xinv_tu_fused9.hip -- instantiations of k_fused9 (the 9-point forms, B != 0).
#include "xinv_dispatch.h"
#include "xinv_fused9.h"

// ---- 9-point fused launch ---------------------------------------------------------------------
template <class M, int K>
static void launch_fused9_k(bool al, bool ext, bool seam, dim3 grid, hipStream_t st, const FusedArgs &a, int *occ)
{
    dim3 block(256, 1, 1);
#define L9(AL, EXT, SEAM)                                                                        \
    do {                                                                                         \
        if (occ) {                                                                               \
            int n = 0;                                                                           \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fused9<M, K, AL, EXT, SEAM>, 256, 0) != hipSuccess) n = 1; \
            *occ = n < 1 ? 1 : n;                                                                \
        } else hipLaunchKernelGGL((k_fused9<M, K, AL, EXT, SEAM>), grid, block, 0, st, a);       \
    } while (0)
    if (seam) { if (ext) L9(false, true, true); else L9(false, false, true); }     // (odd xc: unaligned strips)
    else if (al) { if (ext) L9(true, true, false); else L9(true, false, false); }
    else    { if (ext) L9(false, true, false); else L9(false, false, false); }
#undef L9
}

int xinv_launch_fused9(bool gen, int K, bool al, bool ext, dim3 grid, hipStream_t st,
                       const FusedArgs &a, int *occ, bool seam)
{
    if (gen) {
        if (K == 1) launch_fused9_k<Fused9Gen, 1>(al, ext, seam, grid, st, a, occ);
        else if (K == 2) launch_fused9_k<Fused9Gen, 2>(al, ext, seam, grid, st, a, occ);
        else return 1;
    } else {
        if (K == 1) launch_fused9_k<Fused9Std, 1>(al, ext, seam, grid, st, a, occ);
        else if (K == 2) launch_fused9_k<Fused9Std, 2>(al, ext, seam, grid, st, a, occ);
        else if (K == 3) launch_fused9_k<Fused9Std, 3>(al, ext, seam, grid, st, a, occ);
        else return 1;
    }
    return 0;
}

