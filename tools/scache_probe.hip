// scache_probe.hip -- what does the scalar data cache (K$) of gfx950 do with data that vector stores rewrite?
//   hipcc --offload-arch=gfx950 -O3 -o build/scache_probe tools/scache_probe.hip && build/scache_probe
//
// Part 1 (inside ONE kernel): a wavefront reads X through the scalar unit, rewrites it with a vector store,
//   waits for the store, and reads it again  (a) with a plain s_load, (b) with `s_load ... glc`,
//   (c) after s_dcache_inv.  Tells which of the three see the new value.
// Part 2 (ACROSS kernels, one stream): W(it) rewrites a table with vector stores, R(it) reads it through the
//   scalar unit from every workgroup and counts words that do not carry `it`.  Variants: back to back; with a
//   host synchronisation between; with other kernels between; R starting with s_dcache_inv.
// Part 3: the same with the table rewritten by a hipMemcpyAsync from the host and by hipMemsetAsync.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef const unsigned long long __attribute__((address_space(4))) *cptr;

__device__ __forceinline__ unsigned long long sload(const unsigned long long *p)
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long sload_glc(const unsigned long long *p)
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// ---- part 1
__global__ void k_inkernel(unsigned long long *X, unsigned long long *out)
{
    unsigned long long *p = X + blockIdx.x * 64;            // one 512-byte region per workgroup (one wavefront)
    const unsigned long long v0 = sload(p);                  // caches the line
    if (threadIdx.x == 0) *(volatile unsigned long long *)p = v0 + 1000;
    __builtin_amdgcn_s_waitcnt(0);                           // vmcnt(0): the store has left
    __threadfence();
    const unsigned long long a = sload(p);                   // plain: K$ hit?
    const unsigned long long b = sload_glc(p);               // glc
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c = sload(p);
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = v0; out[blockIdx.x * 4 + 1] = a; out[blockIdx.x * 4 + 2] = b; out[blockIdx.x * 4 + 3] = c; }
}

// ---- part 2
__global__ void k_write(unsigned long long *X, int n, unsigned long long it)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) X[i] = (it << 20) | (unsigned)i;
}

template <int MODE>     // 0 plain s_load, 1 s_dcache_inv first, 2 glc loads, 3 vector loads
__global__ void k_read(const unsigned long long *X, int n, unsigned long long it, unsigned *bad, unsigned *first_bad)
{
    if (MODE == 1) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    unsigned nb = 0;
    // every wavefront walks the whole table from a different starting row (like the tiles of a sweep kernel)
    const int wv = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    for (int k = 0; k < n; k += 4) {
        int i = (k + wv * 8) % n;
        i = __builtin_amdgcn_readfirstlane(i);
        unsigned long long v;
        if (MODE == 2) v = sload_glc(X + i);
        else if (MODE == 3) v = X[i + (threadIdx.x & 3)] - (threadIdx.x & 3);
        else v = sload(X + i);
        if (v != ((it << 20) | (unsigned)i)) { nb++; if ((threadIdx.x & 63) == 0) atomicMin(first_bad, (unsigned)(v >> 20)); }
    }
    if (nb && (threadIdx.x & 63) == 0) atomicAdd(bad, nb);
}

// graph variants: the generation counter lives in device memory
__global__ void k_write_gen(unsigned long long *X, int n, unsigned long long *gen)
{
    const unsigned long long it = *(volatile unsigned long long *)gen + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) X[i + 8] = (it << 20) | (unsigned)i;
    __syncthreads();
    // (every workgroup computed `it` from the old counter; the last one to finish would be needed for a clean
    //  bump -- instead the counter sits in X[0] and is bumped by the reader's single checker thread)
}
template <int MODE>
__global__ void k_read_gen(const unsigned long long *X, int n, unsigned long long *gen, unsigned *bad, unsigned *first_bad)
{
    if (MODE == 1) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    const unsigned long long it = *(volatile unsigned long long *)gen + 1;
    unsigned nb = 0;
    const int wv = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    for (int k = 0; k < n; k += 4) {
        int i = (k + wv * 8) % n;
        i = __builtin_amdgcn_readfirstlane(i);
        unsigned long long v;
        if (MODE == 2) v = sload_glc(X + 8 + i);
        else if (MODE == 3) v = X[8 + i + (threadIdx.x & 3)] - (threadIdx.x & 3);
        else v = sload(X + 8 + i);
        if (v != ((it << 20) | (unsigned)i)) { nb++; if ((threadIdx.x & 63) == 0) atomicMin(first_bad, (unsigned)(v >> 20)); }
    }
    if (nb && (threadIdx.x & 63) == 0) atomicAdd(bad, nb);
    // the last workgroup to finish bumps the generation
    __shared__ unsigned last;
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); last = (atomicAdd(bad + 2, 1u) == gridDim.x - 1); }
    __syncthreads();
    if (last && threadIdx.x == 0) { bad[2] = 0; *(volatile unsigned long long *)gen = it; __threadfence(); }
}

__global__ void k_other(double *Y, int n)                   // unrelated traffic between W and R
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) Y[i] = Y[i] * 1.0000001 + 1.0;
}

template <int MODE>
static void cross(const char *what, int variant, int iters, int n, int wgs)
{
    unsigned long long *X; unsigned *bad; double *Y;
    CHECK(hipMalloc(&X, (n + 8) * 8)); CHECK(hipMalloc(&bad, 16)); CHECK(hipMalloc(&Y, 1 << 22));
    CHECK(hipMemset(bad, 0, 16)); CHECK(hipMemset(Y, 0, 1 << 22));
    unsigned init[2] = {0u, 0xffffffffu};
    CHECK(hipMemcpy(bad, init, 8, hipMemcpyHostToDevice));
    std::vector<unsigned long long> h(n);
    hipStream_t st; CHECK(hipStreamCreate(&st));
    unsigned long long *hp; CHECK(hipHostMalloc(&hp, n * 8));
    int stale_iters = 0;
    unsigned prev = 0;
    // variants 4 / 5: the reader (4) or writer + reader (5) replayed from a hipGraph; the generation travels in a
    // device word the writer bumps, so that the captured arguments stay valid
    hipGraphExec_t gexec = nullptr;
    unsigned long long *gen; CHECK(hipMalloc(&gen, 8)); CHECK(hipMemset(gen, 0, 8));
    if (variant >= 4) {
        hipGraph_t g;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        if (variant == 5) hipLaunchKernelGGL(k_write_gen, dim3(8), dim3(256), 0, st, X, n, gen);
        hipLaunchKernelGGL(k_read_gen<MODE>, dim3(wgs), dim3(256), 0, st, (const unsigned long long *)X, n, gen, bad, bad + 1);
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
        CHECK(hipGraphDestroy(g));
        for (int it = 1; it <= iters; it++) {
            if (variant == 4) hipLaunchKernelGGL(k_write_gen, dim3(8), dim3(256), 0, st, X, n, gen);
            CHECK(hipGraphLaunch(gexec, st));
        }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipGraphExecDestroy(gexec));
    } else
    for (int it = 1; it <= iters; it++) {
        if (variant == 3) {                                   // host copy instead of a kernel
            for (int i = 0; i < n; i++) hp[i] = ((unsigned long long)it << 20) | (unsigned)i;
            CHECK(hipMemcpyAsync(X, hp, n * 8, hipMemcpyHostToDevice, st));
        } else
            hipLaunchKernelGGL(k_write, dim3(8), dim3(256), 0, st, X, n, (unsigned long long)it);
        if (variant == 1) CHECK(hipStreamSynchronize(st));
        if (variant == 2) hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, st, Y, (1 << 22) / 8);
        hipLaunchKernelGGL(k_read<MODE>, dim3(wgs), dim3(256), 0, st, (const unsigned long long *)X, n, (unsigned long long)it, bad, bad + 1);
        if (variant == 3) CHECK(hipStreamSynchronize(st));    // the pinned buffer is rewritten next iteration
        if ((it & 63) == 0) {
            unsigned b[2]; CHECK(hipMemcpyAsync(b, bad, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
            if (b[0] != prev) { stale_iters++; prev = b[0]; }
        }
    }
    unsigned b[2]; CHECK(hipMemcpy(b, bad, 8, hipMemcpyDeviceToHost));
    printf("  %-46s variant %d: %u stale words in %d iterations x %d workgroups (oldest generation seen: %s%u)\n",
           what, variant, b[0], iters, wgs, b[1] == 0xffffffffu ? "-" : "", b[1] == 0xffffffffu ? 0 : b[1]);
    CHECK(hipFree(X)); CHECK(hipFree(bad)); CHECK(hipFree(Y)); CHECK(hipHostFree(hp)); CHECK(hipStreamDestroy(st));
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    {
        const int nb = 512;
        unsigned long long *X, *out;
        CHECK(hipMalloc(&X, nb * 64 * 8)); CHECK(hipMalloc(&out, nb * 4 * 8));
        std::vector<unsigned long long> h(nb * 64, 7), o(nb * 4);
        CHECK(hipMemcpy(X, h.data(), nb * 64 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_inkernel, dim3(nb), dim3(64), 0, 0, X, out);
        CHECK(hipMemcpy(o.data(), out, nb * 4 * 8, hipMemcpyDeviceToHost));
        int plain = 0, glc = 0, inv = 0;
        for (int b = 0; b < nb; b++) { plain += o[b * 4 + 1] == 1007; glc += o[b * 4 + 2] == 1007; inv += o[b * 4 + 3] == 1007; }
        printf("part 1, re-read after a vector store inside one kernel, %d wavefronts: fresh with plain s_load %d, with glc %d, after s_dcache_inv %d\n",
               nb, plain, glc, inv);
    }
    printf("part 2/3, table rewritten between kernels (variant 0 back to back, 1 host sync between, 2 another kernel between, 3 rewritten by an H2D copy, 4 reader replayed from a hipGraph, 5 writer + reader in one hipGraph)\n");
    for (int n : {64, 2048}) {                               // 512 B (stays in the K$) and 16 KiB
        printf(" table of %d words\n", n);
        for (int wgs : {64, 2048}) {
            for (int v = 0; v < 6; v++) cross<0>("plain s_load", v, iters, n, wgs);
            cross<1>("s_dcache_inv at kernel start, then s_load", 4, iters, n, wgs);
            cross<1>("s_dcache_inv at kernel start, then s_load", 0, iters, n, wgs);
            cross<2>("s_load glc", 0, iters, n, wgs);
            cross<3>("vector loads", 0, iters, n, wgs);
        }
    }
    return 0;
}
