// Microbenchmark: streaming copy (512 MiB -> 512 MiB) with the cache-policy bits of the store (and of the load) varied.
// aux bits of the raw buffer builtins on gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4g __attribute__((vector_size(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LAUX, int SAUX>
__global__ __launch_bounds__(256) void k(const void *in, void *out, unsigned n16)
{
    // 1 GiB windows per buffer: rebuild the descriptor per block of work is not needed at this size (offsets < 2^31)
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(n16 * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(n16 * 16u), 0x00020000);
    for (unsigned i = blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += gridDim.x * 256 * 4) {
        u4g v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(ri, (i + q * 256) * 16u, 0, LAUX);
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_amdgcn_raw_buffer_store_b128(v[q], ro, (i + q * 256) * 16u, 0, SAUX);
    }
}

template <int LAUX, int SAUX> void run(const void *in, void *out, unsigned n16)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 128;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<LAUX, SAUX>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<LAUX, SAUX>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("load aux %2d  store aux %2d: %6.2f TB/s\n", LAUX, SAUX, 2.0 * n16 * 16 / (ms / 20 * 1e-3) / 1e12);
}

int main()
{
    const unsigned n16 = (512u << 20) / 16;
    void *in, *out;
    CK(hipMalloc(&in, (size_t)n16 * 16)); CK(hipMalloc(&out, (size_t)n16 * 16));
    CK(hipMemset(in, 1, (size_t)n16 * 16));
    run<2, 2>(in, out, n16); run<2, 0>(in, out, n16); run<2, 1>(in, out, n16); run<2, 3>(in, out, n16); run<2, 16>(in, out, n16);
    run<2, 17>(in, out, n16); run<2, 18>(in, out, n16); run<2, 19>(in, out, n16);
    run<0, 2>(in, out, n16); run<1, 2>(in, out, n16); run<3, 2>(in, out, n16); run<16, 2>(in, out, n16); run<18, 2>(in, out, n16); run<19, 2>(in, out, n16);
    run<2, 2>(in, out, n16);
    return 0;
}
