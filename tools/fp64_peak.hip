// fp64_peak.hip -- measured issue ceiling of the fp64 vector ALU on this GPU, the resource that
// bounds k_fused2d (bench.py's roofline "valu_fp64").  Three chains of independent operations per
// lane: mul+add pairs without contraction (what the engine may use: the bit-exactness contract
// forbids FMA), FMA (the datasheet's 2 flop/lane/op figure), and 32-bit moves (DPP / select class).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fp64_peak.hip -o build/fp64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 4096
#define NACC 8

__global__ void k_muladd(double *out, double a, double b)
{
    double x[NACC];
    for (int i = 0; i < NACC; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < NACC; i++) { x[i] = x[i] * a; x[i] = x[i] + b; }   // -ffp-contract=off: two instructions
    double s = 0;
    for (int i = 0; i < NACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma(double *out, double a, double b)
{
    double x[NACC];
    for (int i = 0; i < NACC; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < NACC; i++) { x[i] = __builtin_fma(x[i], a, b); x[i] = __builtin_fma(x[i], a, b); }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mov32(int *out, int a)
{
    int x[NACC];
    for (int i = 0; i < NACC; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < NACC; i++) { x[i] = (x[i] & a) | (x[(i + 1) % NACC] & ~a); x[i] ^= it; }
    int s = 0;
    for (int i = 0; i < NACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    double *d; int *di;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipMalloc(&di, sizeof(int) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); k_muladd<<<blocks, threads>>>(d, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * threads * ITER * NACC * 2;
        if (rep == 2) printf("fp64 mul+add (no FMA): %.2f TFLOP/s  (%.3f ms)\n", ops / ms / 1e9, ms);
        hipEventRecord(e0); k_fma<<<blocks, threads>>>(d, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("fp64 FMA             : %.2f TFLOP/s counting 2 flop per FMA (%.2f T instr-lanes/s)\n",
                             2 * ops / ms / 1e9, ops / ms / 1e9);
        hipEventRecord(e0); k_mov32<<<blocks, threads>>>(di, 0x0f0f0f0f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("32-bit logic         : %.2f T lane-ops/s\n", (double)blocks * threads * ITER * NACC * 2 / ms / 1e9);
    }
    return 0;
}
