// Microbenchmark: do global loads make progress while the matrix pipes are saturated?  One 512-thread workgroup per CU
// (128 KiB of LDS requested so that only one fits), every wave repeats: issue LD 16-byte loads, run NM fp32 MFMAs
// (16x16x4, independent accumulators), wait for the loads, fold them into a checksum.  Reports the time of the loop with
// loads only, MFMAs only and both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LD, int NM, bool AGPR_HINT>
__global__ __launch_bounds__(512) void k(const v4i *__restrict__ in, float *__restrict__ out, int iters, size_t stride_v4)
{
    __shared__ unsigned char lds[128 * 1024];
    const int tid = threadIdx.x;
    lds[tid] = 0;
    v4f acc[10];
#pragma unroll
    for (int q = 0; q < 10; q++) acc[q] = (v4f){0.f, 0.f, 0.f, 0.f};
    v4i sum = (v4i){0, 0, 0, 0};
    const v4i *p = in + (size_t)blockIdx.x * 512 * 8 + tid;
    float a = (float)tid, b = 1.0f;
    for (int it = 0; it < iters; it++) {
        v4i st[LD > 0 ? LD : 1];
#pragma unroll
        for (int k = 0; k < LD; k++) st[k] = p[(size_t)it * stride_v4 + (size_t)k * 512];
#pragma unroll
        for (int m = 0; m < NM; m++) acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 10], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < LD; k++) sum += st[k];
        asm volatile("" : "+v"(a));
    }
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 10; q++) r += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    r += (float)(sum.x + sum.y + sum.z + sum.w) + lds[tid];
    if (r == 123.456f) out[blockIdx.x * 512 + tid] = r;
}

template <int LD, int NM> float run(const v4i *in, float *out, int iters, size_t stride)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<LD, NM, false>), dim3(256), dim3(512), 0, 0, in, out, iters, stride);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<LD, NM, false>), dim3(256), dim3(512), 0, 0, in, out, iters, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e3f;
}

int main()
{
    const int iters = 32;
    const size_t per_it_v4 = (size_t)256 * 512 * 8;  // 256 CUs x 512 threads x 8 loads of 16 B = 16 MiB per iteration
    v4i *in; float *out;
    CK(hipMalloc(&in, per_it_v4 * 16 * iters)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 1, per_it_v4 * 16 * iters));
    const float l = run<8, 0>(in, out, iters, per_it_v4), m = run<0, 160>(in, out, iters, per_it_v4), both = run<8, 160>(in, out, iters, per_it_v4);
    const float m40 = run<0, 40>(in, out, iters, per_it_v4), both40 = run<8, 40>(in, out, iters, per_it_v4);
    printf("32 iterations, 16 MiB loaded per iteration (512 MiB total), 2 waves per SIMD\n");
    printf("loads only            : %7.1f us  (%.2f TB/s)\n", l, 512.0 * 1.048576 / l);
    printf("160 MFMA/iter only    : %7.1f us\n", m);
    printf("loads + 160 MFMA/iter : %7.1f us   (sum %.1f, max %.1f)\n", both, l + m, l > m ? l : m);
    printf(" 40 MFMA/iter only    : %7.1f us\n", m40);
    printf("loads +  40 MFMA/iter : %7.1f us   (sum %.1f, max %.1f)\n", both40, l + m40, l > m40 ? l : m40);
    return 0;
}
