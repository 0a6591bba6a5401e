// xinv_dispatch.h -- host-side launchers of the templated sweep kernels.  Each kernel family is
// instantiated in its own translation unit (xinv_tu_*.hip) so that the library builds in parallel
// and a change to one family recompiles one file; the host driver (xinv_hip.hip) sees only these
// plain functions.  With `occ` non-null a launcher reports how many workgroups of the variant fit
// on a CU (register-limited) instead of launching it.  Return 0, or 1 for an uninstantiated variant.
#pragma once
#include <hip/hip_runtime.h>
#include "xinv_fused.h"
#include "xinv_pipe2d.h"
#include "xinv_fused3d.h"
#include "xinv_fused3dg.h"
#include "xinv_pipe3d.h"
#include "xinv_fusedbih.h"

#define XINV_HIDDEN __attribute__((visibility("hidden")))

XINV_HIDDEN int xinv_launch_fused2d_std(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                        hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_fused2d_gen(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                        hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_fused2d_std2dt(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                           hipStream_t st, const FusedArgs &a, int *occ);
// the odd-xc periodic seam variants (unaligned strips only; xinv_tu_fused2d.hip with -DXINV_TU_SEAM=1)
XINV_HIDDEN int xinv_launch_fused2d_std_seam(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                             hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_fused2d_gen_seam(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                             hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_fused2d_std2dt_seam(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                                hipStream_t st, const FusedArgs &a, int *occ);
// contracted arithmetic (XINV_FLAG_FMA): per-row-coefficient variants of k_fused2d / k_pipe2d on the F models
XINV_HIDDEN int xinv_launch_fused2d_stdf(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                         hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_fused2d_genf(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                         hipStream_t st, const FusedArgs &a, int *occ);
// general form with coefficient arrays that vary along x and the point-factor stream Q (FusedGen2DQ: um = 0x1c or 0)
XINV_HIDDEN int xinv_launch_fused2d_genq(bool al, bool ext, unsigned um, int K, dim3 grid, dim3 block,
                                         hipStream_t st, const FusedArgs &a, int *occ);
XINV_HIDDEN int xinv_launch_pipe2d_fma(bool gen, unsigned um, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                       const FusedArgs &a, int *occ, int lds_pad);
// wave-pipelined four-sweep pass, one tile per 256-thread workgroup: standard form (um = 3: A and C per row, np = 1
// or 2 column pairs per lane) and general form (um = 0x1f: A C D E F per row).  Returns 1 for a variant that is
// not instantiated (coefficient arrays varying along x: measured slower than k_fused2d, see plan_fused5).
// lds_pad: unused dynamic LDS per workgroup, which caps how many of them share a CU (see launch_fused)
// fr: the forcing row rides the LDS ring with S (per-row-coefficient variants, one column pair per lane)
XINV_HIDDEN int xinv_launch_pipe2d_std(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                       const FusedArgs &a, int *occ, int lds_pad);
XINV_HIDDEN int xinv_launch_pipe2d_gen(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                       const FusedArgs &a, int *occ, int lds_pad);
XINV_HIDDEN int xinv_launch_pipe2d_std_seam(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                            const FusedArgs &a, int *occ, int lds_pad);
XINV_HIDDEN int xinv_launch_pipe2d_gen_seam(unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                            const FusedArgs &a, int *occ, int lds_pad);
static inline int xinv_launch_pipe2d(bool gen, unsigned um, int np, bool fr, bool al, bool ext, dim3 grid, hipStream_t st,
                                     const FusedArgs &a, int *occ, int lds_pad = 0, bool seam = false, bool fma = false)
{
    if (fma) return (seam || np != 1) ? 1 : xinv_launch_pipe2d_fma(gen, um, fr, al, ext, grid, st, a, occ, lds_pad);
    if (seam) return gen ? xinv_launch_pipe2d_gen_seam(um, np, fr, al, ext, grid, st, a, occ, lds_pad)
                         : xinv_launch_pipe2d_std_seam(um, np, fr, al, ext, grid, st, a, occ, lds_pad);
    return gen ? xinv_launch_pipe2d_gen(um, np, fr, al, ext, grid, st, a, occ, lds_pad)
               : xinv_launch_pipe2d_std(um, np, fr, al, ext, grid, st, a, occ, lds_pad);
}
XINV_HIDDEN int xinv_launch_fused9(bool gen, int K, bool al, bool ext, dim3 grid, hipStream_t st,
                                   const FusedArgs &a, int *occ, bool seam = false);
XINV_HIDDEN int xinv_launch_fused3d(int NW, bool al, bool uni, bool ext, dim3 grid, hipStream_t st,
                                    const Fused3Args &a);
// two sweeps per pass pipelined across two groups of eight wavefronts (xinv_pipe3d.h): x-uniform coefficients, no 'extend'
XINV_HIDDEN int xinv_launch_pipe3d(bool al, dim3 grid, hipStream_t st, const Fused3Args &a, bool seam = false, bool ext = false);
// odd-xc periodic seam inside the kernel (xinv_tu_fused3d_seam.hip): unaligned strips, NW = 8 or 12
XINV_HIDDEN int xinv_launch_fused3d_seam(int NW, bool uni, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a);
XINV_HIDDEN int xinv_launch_fused3dg_seam(int NW, bool ext, dim3 grid, hipStream_t st, const Fused3GArgs &a);
// contracted arithmetic (XINV_FLAG_FMA): x-uniform coefficients only (xinv_tu_fused3d_fma.hip)
XINV_HIDDEN int xinv_launch_fused3d_fma(int NW, bool al, bool ext, dim3 grid, hipStream_t st, const Fused3Args &a);
XINV_HIDDEN int xinv_launch_pipe3d_fma(bool al, dim3 grid, hipStream_t st, const Fused3Args &a);
XINV_HIDDEN int xinv_launch_fused3dg(int NW, bool al, bool ext, dim3 grid, hipStream_t st,
                                     const Fused3GArgs &a);
// vm: where the coefficients come from (xinv_fusedbih.h: 0 per-row records, 1 A C D F as vector streams, 2 all nine)
XINV_HIDDEN int xinv_launch_fusedbih(bool per, bool zbe, int vm, dim3 grid, hipStream_t st,
                                     const FusedBihArgs &a, int *occ);
