// xinv_tu_fused2d.hip -- instantiations of k_fused2d for ONE model (compiled three times:
// -DXINV_TU_MODEL=0 standard form, 1 general form, 2 standard "test" form; and each once more with -DXINV_TU_SEAM=1:
// the odd-xc periodic seam variants, unaligned strips only; 3 / 4: the contracted-arithmetic models of XINV_FLAG_FMA,
// per-row-coefficient variants only).
#include <type_traits>
#include "xinv_dispatch.h"

// Launch one instantiation -- or, when `occ` is given, only report how many of its workgroups
// fit on a CU (register-limited: 1 to 3), which the tiling heuristic needs.
#ifndef XINV_TU_SEAM
#define XINV_TU_SEAM 0
#endif
constexpr bool SEAM = XINV_TU_SEAM != 0;

template <class M, int K, bool AL, unsigned UM, bool EXT>
static int fused_one(dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a, int *occ)
{
    if (occ) {
        static int cached = 0;                           // (asked by the planner in every solve: a few microseconds per query)
        int n = cached;
        if (!n) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fused2d<M, K, AL, UM, EXT, 0, SEAM>, 256, 0) != hipSuccess)
                n = 1;
            cached = n = n < 1 ? 1 : n;
        }
        *occ = n;
        return 0;
    }
    hipLaunchKernelGGL((k_fused2d<M, K, AL, UM, EXT, 0, SEAM>), grid, block, 0, st, a);
    return 0;
}

template <class M, bool AL, unsigned UM, bool EXT>
static int launch_fused_k(int K, dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a, int *occ)
{
    switch (K) {
    case 1: return fused_one<M, 1, AL, UM, EXT>(grid, block, st, a, occ);
    case 2: return fused_one<M, 2, AL, UM, EXT>(grid, block, st, a, occ);
    // three and four sweeps per pass: only the standard form with per-row A and C keeps two
    // wavefronts per SIMD at that window depth (194 / 249 VGPRs; the general form does not gain)
    case 3:
        return fused_one<M, 3, AL, UM, EXT>(grid, block, st, a, occ);
        break;
    case 4:
        if constexpr (std::is_base_of<FusedStd2D, M>::value)
            return fused_one<M, 4, AL, UM, EXT>(grid, block, st, a, occ);
        break;
    default: break;
    }
    return 1;
}

template <class M, bool AL, bool EXT>
static int launch_fused_um(unsigned um, int K, dim3 grid, dim3 block, hipStream_t st, const FusedArgs &a,
                           int *occ)
{
    if constexpr (ModelFma<M>::value) {                  // contracted arithmetic: the per-row-coefficient variants only
        constexpr unsigned UMF = std::is_base_of<FusedStd2D, M>::value ? 3u : 0x1fu;
        if (um == UMF) return launch_fused_k<M, AL, UMF, EXT>(K, grid, block, st, a, occ);
        return 1;
    } else {
        if constexpr (std::is_same<M, FusedStd2D>::value) {
            if (um == 3u) return launch_fused_k<M, AL, 3u, EXT>(K, grid, block, st, a, occ);
        } else if constexpr (std::is_same<M, FusedStd2DT>::value) {
            if (um == 7u) return launch_fused_k<M, AL, 7u, EXT>(K, grid, block, st, a, occ);
        } else if constexpr (std::is_same<M, FusedGen2DQA>::value) {   // (C read out of A: bit 1 set, nothing of C loaded)
            if (um == 0x1eu) return launch_fused_k<M, AL, 0x1eu, EXT>(K, grid, block, st, a, occ);
            if (um == 0x02u) return launch_fused_k<M, AL, 0x02u, EXT>(K, grid, block, st, a, occ);
            return 1;
        } else if constexpr (ModelPQ<M>::value) {        // (A, C varying along x: D, E, F per row, or nothing)
            if (um == 0x1cu) return launch_fused_k<M, AL, 0x1cu, EXT>(K, grid, block, st, a, occ);
            if (um != 0u) return 1;
        } else {
            if (um == 0x1fu) return launch_fused_k<M, AL, 0x1fu, EXT>(K, grid, block, st, a, occ);
            if (um == 0x1cu) return launch_fused_k<M, AL, 0x1cu, EXT>(K, grid, block, st, a, occ);
        }
        if constexpr (!std::is_same<M, FusedGen2DQA>::value)      // (every stream a vector)
            return launch_fused_k<M, AL, 0u, EXT>(K, grid, block, st, a, occ);
        return 1;
    }
}

template <class M>
static int launch_fused_m(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                          const FusedArgs &a, int *occ)
{
    if constexpr (!SEAM) {
        if (al) return ext ? launch_fused_um<M, true, true>(um, K, grid, block, st, a, occ)
                           : launch_fused_um<M, true, false>(um, K, grid, block, st, a, occ);
    } else if (al) return 1;
    return ext ? launch_fused_um<M, false, true>(um, K, grid, block, st, a, occ)
               : launch_fused_um<M, false, false>(um, K, grid, block, st, a, occ);
}


#if XINV_TU_SEAM
#define xinv_launch_fused2d_std xinv_launch_fused2d_std_seam
#define xinv_launch_fused2d_gen xinv_launch_fused2d_gen_seam
#define xinv_launch_fused2d_std2dt xinv_launch_fused2d_std2dt_seam
#endif
#if XINV_TU_MODEL == 0
int xinv_launch_fused2d_std(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                            const FusedArgs &a, int *occ)
{ return launch_fused_m<FusedStd2D>(al, ext, um, K, grid, block, st, a, occ); }
#elif XINV_TU_MODEL == 1
int xinv_launch_fused2d_gen(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                            const FusedArgs &a, int *occ)
{ return launch_fused_m<FusedGen2D>(al, ext, um, K, grid, block, st, a, occ); }
#elif XINV_TU_MODEL == 2
int xinv_launch_fused2d_std2dt(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                               const FusedArgs &a, int *occ)
{ return launch_fused_m<FusedStd2DT>(al, ext, um, K, grid, block, st, a, occ); }
#elif XINV_TU_MODEL == 3
int xinv_launch_fused2d_stdf(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                             const FusedArgs &a, int *occ)
{ return launch_fused_m<FusedStd2DF>(al, ext, um, K, grid, block, st, a, occ); }
#elif XINV_TU_MODEL == 4
int xinv_launch_fused2d_genf(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                             const FusedArgs &a, int *occ)
{ return launch_fused_m<FusedGen2DF>(al, ext, um, K, grid, block, st, a, occ); }
#else
int xinv_launch_fused2d_genq(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block, hipStream_t st,
                             const FusedArgs &a, int *occ)
{
    if (um & 2u) return launch_fused_m<FusedGen2DQA>(al, ext, um, K, grid, block, st, a, occ);
    return launch_fused_m<FusedGen2DQ>(al, ext, um, K, grid, block, st, a, occ);
}
#endif
