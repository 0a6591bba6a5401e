// fp64_latency.hip -- how fast does ONE wavefront issue DEPENDENT fp64 operations, and how many
// wavefronts per SIMD does it take to fill the pipe?  (k_fused2d runs two wavefronts per SIMD and its
// point update is a chain of ~10 dependent fp64 operations with little parallelism beside it.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fp64_latency.hip -o build/fp64_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 8192
template <int NACC>
__global__ void k_chain(double *out, double a, double b)
{
    double x[NACC];
    for (int i = 0; i < NACC; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < NACC; i++) { x[i] = x[i] * a; x[i] = x[i] + b; }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(double *d, int waves_per_simd)
{
    // one workgroup of 256 threads = one wavefront per SIMD of a CU; `waves_per_simd` workgroups per CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); k_chain<NACC><<<blocks, 256>>>(d, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double ops_per_wave = (double)ITER * NACC * 2;
    const double ns_per_op = ms * 1e6 / ops_per_wave / 1.0;          // wall time per op of ONE wave's stream
    printf("chains/lane %d, waves/SIMD %d: %.2f TFLOP/s, %.2f ns per op in a wave's stream (%.1f cycles at 2.1 GHz)\n",
           NACC, waves_per_simd, (double)blocks * 256 * ops_per_wave / ms / 1e9, ns_per_op, ns_per_op * 2.1);
}
int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    for (int w : {1, 2, 4, 8}) run<1>(d, w);
    for (int w : {1, 2, 4}) run<2>(d, w);
    for (int w : {1, 2}) run<4>(d, w);
    run<8>(d, 1); run<8>(d, 2);
    return 0;
}
