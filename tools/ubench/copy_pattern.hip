// Microbenchmark: does the FFT kernel's access pattern (8 B per lane, 16 row loads of a 32 KiB frame per
// workgroup, persistent grid) reach copy bandwidth, and what do 16 B per lane / nontemporal buy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0: 8B/lane plain, 1: 16B/lane plain, 2: 8B nontemporal, 3: 16B nontemporal, 4: 16B + lds roundtrip
__global__ __launch_bounds__(256) void k_copy(const f2 *__restrict__ in, f2 *__restrict__ out, int ngroups)
{
    __shared__ f2 lds[4096];
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const f2 *src = in + (size_t)g * 4096;
        f2 *dst = out + (size_t)g * 4096;
        if constexpr (MODE == 0 || MODE == 2) {
            f2 v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = MODE == 2 ? __builtin_nontemporal_load(src + tid + 256 * r) : src[tid + 256 * r];
#pragma unroll
            for (int r = 0; r < 16; r++) { v[r].x += 1.f; if (MODE == 2) __builtin_nontemporal_store(v[r], dst + ((tid + 256 * r) ^ 2048)); else dst[(tid + 256 * r) ^ 2048] = v[r]; }
        } else {
            f4 v[8];
            const f4 *s4 = (const f4 *)src;
            f4 *d4 = (f4 *)dst;
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = MODE == 3 ? __builtin_nontemporal_load(s4 + tid + 256 * r) : s4[tid + 256 * r];
            if constexpr (MODE == 4) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 8; r++) { lds[2 * tid + 512 * r] = (f2){v[r].x, v[r].y}; lds[2 * tid + 1 + 512 * r] = (f2){v[r].z, v[r].w}; }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 8; r++) { f2 a = lds[(tid + 256 * r)], b = lds[(tid + 256 * r + 2048)]; v[r] = (f4){a.x, a.y, b.x, b.y}; }
            }
#pragma unroll
            for (int r = 0; r < 8; r++) { v[r].x += 1.f; if (MODE == 3) __builtin_nontemporal_store(v[r], d4 + ((tid + 256 * r) ^ 1024)); else d4[(tid + 256 * r) ^ 1024] = v[r]; }
        }
    }
}

template <int MODE> float run(const f2 *in, f2 *out, int ngroups, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_copy<MODE>, dim3(grid), dim3(256), 0, 0, in, out, ngroups);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_copy<MODE>, dim3(grid), dim3(256), 0, 0, in, out, ngroups);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 20;
}

int main()
{
    const int ngroups = 16384;
    const size_t bytes = (size_t)ngroups * 4096 * 8;
    f2 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 0x11, bytes));
    const char *names[5] = {"8B/lane", "16B/lane", "8B/lane nt", "16B/lane nt", "16B/lane + LDS trip"};
    for (int grid : {768, 1024, 2048, 16384}) {
        float t[5] = {run<0>(in, out, ngroups, grid), run<1>(in, out, ngroups, grid), run<2>(in, out, ngroups, grid),
                      run<3>(in, out, ngroups, grid), run<4>(in, out, ngroups, grid)};
        for (int m = 0; m < 5; m++) printf("grid %5d  %-22s %7.1f us  %.2f TB/s\n", grid, names[m], t[m] * 1e3, 2.0 * bytes / (t[m] * 1e-3) / 1e12);
    }
    return 0;
}
