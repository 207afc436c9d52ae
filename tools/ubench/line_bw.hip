// Which 128-byte lines of a 2 KiB-strided image are slow?  (round 4: the X-engine's XCD that owns lines 3 and 11 of every row finishes 27 % late)
// Every workgroup reads ONE line index of a contiguous block of rows (plain 16-byte loads, 8 rows per wave instruction); per line index the
// time for all rows is printed.  Then: the same with the buffer base shifted, and with all line indices at once (the copy-like reference).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int INFL>
__global__ __launch_bounds__(512) void k_line(const char *__restrict__ in, int *out, int rows, int line, size_t stride)
{
    const int rows_per = rows / gridDim.x, r0 = blockIdx.x * rows_per;
    const char *base = in + (size_t)r0 * stride + (size_t)line * 128 + (threadIdx.x & 7) * 16;
    int acc = 0;
    for (int p0 = threadIdx.x >> 3; p0 < rows_per; p0 += 64 * INFL) {
        v4i v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) {
            const char *p = base + (size_t)(p0 + u * 64) * stride;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u]) : "v"(p));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFL; u++) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}

static hipEvent_t ea, eb;
template <typename F> float timeit(F f)
{
    for (int i = 0; i < 2; i++) f();
    (void)hipEventRecord(ea);
    for (int i = 0; i < 5; i++) f();
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms; (void)hipEventElapsedTime(&ms, ea, eb);
    return ms / 5 * 1e3f;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rows = 1024 * 64 * 64;  // 64 x config 5: 8 GiB image, one line index of it = 512 MiB (past the 256 MiB Infinity Cache)
    char *buf; int *out;
    CK(hipMalloc(&buf, (size_t)rows * 2048 + (1 << 20))); CK(hipMalloc(&out, 1 << 16));
    CK(hipMemset(buf, 1, (size_t)rows * 2048 + (1 << 20)));
    (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    printf("buffer at %p\n", (void *)buf);
    for (int shift = 0; shift <= 2; shift++) {
        const char *in = buf + shift * 128 * 3;  // base shifted by 0 / 3 / 6 lines
        printf("--- base + %d B, row stride 2048: us per line index (512 MiB each), 256 workgroups\n", shift * 384);
        for (int l = 0; l < 16; l++) {
            const float us = timeit([&] { hipLaunchKernelGGL((k_line<8>), dim3(256), dim3(512), 0, 0, in, out, rows, l, (size_t)2048); });
            printf("  line %2d (addr bits 7..10 = %2d): %6.1f us  %5.2f TB/s\n", l, (int)((((size_t)in >> 7) + l) & 15), us, (double)rows * 128 / us / 1e6);
        }
    }
    printf("--- row stride 2048 + 128 (lines walk through all offsets)\n");
    for (int l = 0; l < 4; l++) {
        const float us = timeit([&] { hipLaunchKernelGGL((k_line<8>), dim3(256), dim3(512), 0, 0, buf, out, rows / 2, l, (size_t)2176); });
        printf("  line %2d: %6.1f us  %5.2f TB/s\n", l, us, (double)(rows / 2) * 128 / us / 1e6);
    }
    printf("--- row stride 4096 / 8192 / 1024\n");
    for (size_t st : {(size_t)4096, (size_t)8192, (size_t)1024}) {
        const int nl = (int)(st / 128);
        for (int l = 0; l < (nl < 16 ? nl : 16); l++) {
            const int r = (int)((size_t)rows * 2048 / st);
            const float us = timeit([&] { hipLaunchKernelGGL((k_line<8>), dim3(256), dim3(512), 0, 0, buf, out, r, l, st); });
            printf("  stride %5zu line %2d: %6.1f us  %5.2f TB/s\n", st, l, us, (double)r * 128 / us / 1e6);
        }
    }
    return 0;
}
