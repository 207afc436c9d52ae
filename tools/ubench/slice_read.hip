// Microbenchmark for a fused int8 X-engine: column-slice reads of the [t][station][2048 B] input by cache policy.
// Workgroup = (W-byte slice of every (t, station) row, one of S time ranges); the 128/W workgroups that share a
// 128-byte line sit on one XCD (blockIdx % 8).  Per lane one W-byte (8 / 16) access per row, rows of consecutive
// lanes 2048 B apart, i.e. every lane of a wave instruction touches a different line.  Flavours: plain, nt,
// sc1, sc0 sc1 (L1 bypass), and LDS-DMA (global_load_lds_dwordx4, 16 B only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { PLAIN = 0, NT = 1, SC1 = 2, SC01 = 3, SC0 = 4 };

template <int FL> __device__ __forceinline__ void ld16(v4i &d, const void *p)
{
    if constexpr (FL == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p));
    if constexpr (FL == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC01) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=&v"(d) : "v"(p));
}
template <int FL> __device__ __forceinline__ void ld8(v2i &d, const void *p)
{
    if constexpr (FL == PLAIN) asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(d) : "v"(p));
    if constexpr (FL == NT) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC01) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(d) : "v"(p));
    if constexpr (FL == SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=&v"(d) : "v"(p));
}

// W = 8 or 16 bytes per row; S = time split; PIN = sharing workgroups on one XCD
template <int W, int FL, int NTH, int INFL, bool PIN>
__global__ __launch_bounds__(NTH) void k_slice(const char *__restrict__ in, int *__restrict__ out, int rows, int S)
{
    constexpr int SH = 128 / W, NSL = 2048 / W;
    int slice, ts;
    if (PIN) {
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, sector = within % SH, combo = xcd + 8 * (within / SH);
        const int lg = combo % (NSL / SH);
        ts = combo / (NSL / SH);
        slice = lg * SH + sector;
    } else {
        slice = blockIdx.x % NSL;
        ts = blockIdx.x / NSL;
    }
    const int rows_per = rows / S, r0 = ts * rows_per;
    int acc = 0;
    const char *base = in + (size_t)r0 * 2048 + slice * W;
    for (int p0 = threadIdx.x; p0 < rows_per; p0 += NTH * INFL) {
        if constexpr (W == 8) {
            v2i v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; u++) ld8<FL>(v[u], base + (size_t)(p0 + u * NTH) * 2048);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < INFL; u++) asm volatile("" : "+v"(v[u]));  // the values exist only after the wait
#pragma unroll
            for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y;
        } else {
            v4i v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; u++) ld16<FL>(v[u], base + (size_t)(p0 + u * NTH) * 2048);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < INFL; u++) asm volatile("" : "+v"(v[u]));
#pragma unroll
            for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}

// 32-byte slices: two lanes per row
template <int FL, int NTH, int INFL>
__global__ __launch_bounds__(NTH) void k_slice32(const char *__restrict__ in, int *__restrict__ out, int rows, int S)
{
    constexpr int SH = 4, NSL = 64;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, sector = within % SH, combo = xcd + 8 * (within / SH);
    const int lg = combo % (NSL / SH), ts = combo / (NSL / SH), slice = lg * SH + sector;
    const int rows_per = rows / S, r0 = ts * rows_per;
    int acc = 0;
    const char *base = in + (size_t)r0 * 2048 + slice * 32 + (threadIdx.x & 1) * 16;
    for (int p0 = threadIdx.x >> 1; p0 < rows_per; p0 += (NTH / 2) * INFL) {
        v4i v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) ld16<FL>(v[u], base + (size_t)(p0 + u * (NTH / 2)) * 2048);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFL; u++) asm volatile("" : "+v"(v[u]));
#pragma unroll
        for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}

// LDS-DMA: every wave gathers 64 rows x 16 B per instruction into its own 1 KiB LDS slots (ring of DEPTH), never reads them
template <int NTH, int DEPTH, bool NTF>
__global__ __launch_bounds__(NTH) void k_slice_dma(const char *__restrict__ in, int *__restrict__ out, int rows, int S)
{
    constexpr int SH = 8, NSL = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, sector = within % SH, combo = xcd + 8 * (within / SH);
    const int lg = combo % (NSL / SH), ts = combo / (NSL / SH), slice = lg * SH + sector;
    const int rows_per = rows / S, r0 = ts * rows_per;
    const char *base = in + (size_t)r0 * 2048 + slice * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lbase = (unsigned)(uintptr_t)lds + wave * DEPTH * 1024;
    int it = 0;
    for (int p0 = threadIdx.x; p0 < rows_per; p0 += NTH * DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            const char *src = base + (size_t)(p0 + u * NTH) * 2048;
            const unsigned dst = lbase + u * 1024;
            unsigned keep;
            if (NTF)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        it++;
    }
    if (it == 0x12345678) out[blockIdx.x] = lds[threadIdx.x];
}

static hipEvent_t ea, eb;
template <typename F> float timeit(F f)
{
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(ea);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(eb); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); fflush(stdout); }
    return ms / 10 * 1e3f;
}
static const char *fl_name[] = {"plain", "nt", "sc1", "sc0sc1", "sc0"};

template <int W, int FL, int NTH, int INFL, bool PIN> void run(const char *in, int *out, int rows, int S)
{
    const int grid = (2048 / W) * S;
    const float us = timeit([&] { hipLaunchKernelGGL((k_slice<W, FL, NTH, INFL, PIN>), dim3(grid), dim3(NTH), 0, 0, in, out, rows, S); });
    printf("W=%2d %-6s pin=%d thr=%4d infl=%2d S=%d grid=%4d rows=%6d: %7.1f us %6.0f GB/s\n", W, fl_name[FL], (int)PIN, NTH, INFL, S, grid, rows, us,
           (double)rows * 2048 / us / 1e3);
}
template <int FL, int NTH, int INFL> void run32(const char *in, int *out, int rows, int S)
{
    const int grid = 64 * S;
    const float us = timeit([&] { hipLaunchKernelGGL((k_slice32<FL, NTH, INFL>), dim3(grid), dim3(NTH), 0, 0, in, out, rows, S); });
    printf("W=32 %-6s pin=1 thr=%4d infl=%2d S=%d grid=%4d rows=%6d: %7.1f us %6.0f GB/s\n", fl_name[FL], NTH, INFL, S, grid, rows, us, (double)rows * 2048 / us / 1e3);
}
template <int NTH, int DEPTH, bool NTF> void run_dma(const char *in, int *out, int rows, int S)
{
    const int grid = 128 * S;
    const size_t lds = (size_t)(NTH / 64) * DEPTH * 1024;
    hipFuncSetAttribute((const void *)k_slice_dma<NTH, DEPTH, NTF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float us = timeit([&] { hipLaunchKernelGGL((k_slice_dma<NTH, DEPTH, NTF>), dim3(grid), dim3(NTH), lds, 0, in, out, rows, S); });
    printf("W=16 dma nt=%d thr=%4d depth=%2d S=%d grid=%4d rows=%6d lds=%zu: %7.1f us %6.0f GB/s\n", (int)NTF, NTH, DEPTH, S, grid, rows, lds, us, (double)rows * 2048 / us / 1e3);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rows = 1024 * 64;
    char *in; int *out;
    CK(hipMalloc(&in, (size_t)rows * 2048 * 2)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(in, 1, (size_t)rows * 2048 * 2));
    hipEventCreate(&ea); hipEventCreate(&eb);
    printf("--- 8-byte slices (4 channels per workgroup), no time split, 256 workgroups\n");
    run<8, PLAIN, 256, 16, true>(in, out, rows, 1);
    run<8, NT, 256, 16, true>(in, out, rows, 1);
    run<8, SC1, 256, 16, true>(in, out, rows, 1);
    run<8, SC01, 256, 16, true>(in, out, rows, 1);
    run<8, SC0, 256, 16, true>(in, out, rows, 1);
    run<8, NT, 256, 32, true>(in, out, rows, 1);
    run<8, NT, 512, 16, true>(in, out, rows, 1);
    run<8, NT, 1024, 8, true>(in, out, rows, 1);
    run<8, NT, 256, 16, false>(in, out, rows, 1);
    run<8, SC1, 512, 16, true>(in, out, rows, 1);
    run<8, NT, 256, 16, true>(in, out, rows, 2);
    printf("--- 16-byte slices (8 channels), S = 2 -> 256 workgroups, S = 4 -> 512\n");
    run<16, PLAIN, 512, 8, true>(in, out, rows, 2);
    run<16, NT, 512, 8, true>(in, out, rows, 2);
    run<16, SC1, 512, 8, true>(in, out, rows, 2);
    run<16, SC01, 512, 8, true>(in, out, rows, 2);
    run<16, NT, 512, 16, true>(in, out, rows, 2);
    run<16, NT, 256, 16, true>(in, out, rows, 2);
    run<16, NT, 1024, 8, true>(in, out, rows, 2);
    run<16, NT, 512, 8, true>(in, out, rows, 4);
    run<16, NT, 512, 8, false>(in, out, rows, 2);
    run<16, NT, 512, 8, true>(in, out, rows, 1);
    printf("--- 32-byte slices (16 channels), S = 4 -> 256 workgroups\n");
    run32<PLAIN, 1024, 8>(in, out, rows, 4);
    run32<NT, 1024, 8>(in, out, rows, 4);
    run32<SC1, 1024, 8>(in, out, rows, 4);
    run32<NT, 512, 16>(in, out, rows, 4);
    run32<NT, 1024, 8>(in, out, rows, 8);
    printf("--- 16-byte slices through LDS-DMA\n");
    run_dma<256, 16, false>(in, out, rows, 2);
    run_dma<256, 16, true>(in, out, rows, 2);
    run_dma<512, 8, true>(in, out, rows, 2);
    run_dma<512, 16, true>(in, out, rows, 2);
    run_dma<256, 32, true>(in, out, rows, 2);
    run_dma<256, 16, true>(in, out, rows, 4);
    printf("--- cache-resident input (16 MiB): the non-HBM limits of the same patterns\n");
    run<8, NT, 256, 16, true>(in, out, 8192, 1);
    run<8, PLAIN, 256, 16, true>(in, out, 8192, 1);
    run<16, NT, 512, 8, true>(in, out, 8192, 2);
    run<16, PLAIN, 512, 8, true>(in, out, 8192, 2);
    run32<NT, 1024, 8>(in, out, 8192, 4);
    return 0;
}
