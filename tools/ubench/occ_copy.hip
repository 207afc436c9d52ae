// Microbenchmark: bandwidth of the FFT access pattern vs workgroups resident per CU (limited through
// dynamic LDS), with and without nontemporal accesses, with and without LDS exchanges + barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT, int XCH>  // XCH = number of LDS exchanges (each: barrier, 16 writes, barrier, 16 reads)
__global__ __launch_bounds__(256) void k(const f2 *__restrict__ in, f2 *__restrict__ out, int ngroups)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f2 *lds = (f2 *)smem;
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const f2 *src = in + (size_t)g * 4096;
        f2 *dst = out + (size_t)g * 4096;
        f2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = NT ? __builtin_nontemporal_load(src + tid + 256 * r) : src[tid + 256 * r];
#pragma unroll
        for (int x = 0; x < XCH; x++) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r++) lds[(tid * 16 + r) ^ (tid & 15)] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = lds[(tid + 256 * r) ^ ((tid >> 4) & 15)] * 1.0001f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (NT) __builtin_nontemporal_store(v[r], dst + ((tid + 256 * r) ^ 2048)); else dst[(tid + 256 * r) ^ 2048] = v[r];
        }
    }
}

template <bool NT, int XCH> float run(const f2 *in, f2 *out, int ngroups, int grid, int lds)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void *)k<NT, XCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<NT, XCH>), dim3(grid), dim3(256), lds, 0, in, out, ngroups);
    (void)hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<NT, XCH>), dim3(grid), dim3(256), lds, 0, in, out, ngroups);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 20;
}

int main()
{
    const int ngroups = 16384;
    const size_t bytes = (size_t)ngroups * 4096 * 8;
    f2 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 0x11, bytes));
    for (int percu : {1, 2, 3, 4, 5}) {
        int lds = 160 * 1024 / percu; lds -= lds % 256; if (lds < 32768) lds = 32768;
        int grid = 256 * percu;
        float t0 = run<false, 0>(in, out, ngroups, grid, lds), t1 = run<true, 0>(in, out, ngroups, grid, lds);
        float t2 = run<false, 2>(in, out, ngroups, grid, lds), t3 = run<true, 2>(in, out, ngroups, grid, lds);
        float t4 = run<true, 2>(in, out, ngroups, 16384, lds);
        printf("WG/CU %d: copy %.2f | copy nt %.2f | 2 LDS exch %.2f | 2 LDS exch nt %.2f | same, grid=ngroups %.2f TB/s\n", percu,
               2.0 * bytes / t0 / 1e9, 2.0 * bytes / t1 / 1e9, 2.0 * bytes / t2 / 1e9, 2.0 * bytes / t3 / 1e9, 2.0 * bytes / t4 / 1e9);
    }
    return 0;
}
