// Microbenchmark (round 5): issue rate of v_mfma_i32_16x16x64_i8 / 16x16x32 from ONE wave per SIMD, accumulators in AGPRs, in-place, inline asm --
// back to back over 16 accumulators, with "s_nop 1" in front, and with 2 / 4 / 8 v_perm_b32 between products.  Cycles per product (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R> __device__ __forceinline__ void mm64(const v4i &A, const v4i &B, bool nop)
{
    if (nop) asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(A), "v"(B), "n"(R), "n"(R + 3) : "a0");
    else asm volatile("v_mfma_i32_16x16x64_i8 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(A), "v"(B), "n"(R), "n"(R + 3) : "a0");
}
template <int R> __device__ __forceinline__ void mm32(const long &A, const long &B)
{
    asm volatile("v_mfma_i32_16x16x32_i8 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(A), "v"(B), "n"(R), "n"(R + 3) : "a0");
}
template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel)
{
    unsigned d;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "v"(lo), "s"(sel));
    return d;
}

// MODE 0: 16x16x64 back to back; 1: with s_nop 1; 2: 16x16x32; 10 + k: 16x16x64 with k perms after every product
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, unsigned long long *out, int *sink)
{
    v4i A = (v4i){(int)threadIdx.x, 2, 3, 4}, B = (v4i){5, 6, (int)threadIdx.x, 8};
    unsigned p0 = threadIdx.x, p1 = threadIdx.x * 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        sfor<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == 0) mm64<4 * i>(A, B, false);
            else if constexpr (MODE == 1) mm64<4 * i>(A, B, true);
            else if constexpr (MODE == 2) mm32<4 * i>(__builtin_bit_cast(long, (int __attribute__((ext_vector_type(2)))){A[0], A[1]}), __builtin_bit_cast(long, (int __attribute__((ext_vector_type(2)))){B[0], B[1]}));
            else {
                mm64<4 * i>(A, B, true);
#pragma unroll
                for (int q = 0; q < MODE - 10; q++) { p0 = perm(p0, p1, 0x05010400u); }
            }
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
    int v;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 0x12345678 && p0 == 77) sink[0] = v;
}
template <int MODE> int run(const char *name, unsigned long long *d, int *sink)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, iters, d, sink);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, iters, d, sink);
    hipEventRecord(b);
    CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h;
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%-40s %6.1f clock ticks per product (s_memtime), %6.2f ns per product and SIMD\n", name, (double)h / (iters * 16.0), ms * 1e6 / (iters * 16.0));
    return 0;
}
int main()
{
    unsigned long long *d; int *sink;
    CK(hipMalloc(&d, 8 * 256)); CK(hipMalloc(&sink, 64));
    run<0>("16x16x64 back to back", d, sink);
    run<1>("16x16x64, s_nop 1 in front", d, sink);
    run<2>("16x16x32 back to back", d, sink);
    run<12>("16x16x64 + 2 v_perm", d, sink);
    run<14>("16x16x64 + 4 v_perm", d, sink);
    run<18>("16x16x64 + 8 v_perm", d, sink);
    return 0;
}
