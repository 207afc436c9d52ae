// Microbenchmark for a fused X-engine: how fast can workgroups stream a [rows][2048 B] matrix when each workgroup
// reads only a W-byte column slice (W = 32 / 64 / 128) of every row in its row range?  (rows = (t, station),
// 2048 B = 1024 channels x {I,Q} int8.)  MAP 0: neighbouring blockIdx share a 128-B line (they land on different
// XCDs); MAP 1: the workgroups sharing a line are placed on the same XCD (blockIdx % 8).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v2i __attribute__((ext_vector_type(2)));
// narrow slices, all rows per workgroup (no partial sums needed by the consumer): W = 8 or 16 bytes per row, NTH threads
template <int W, int NTH, int INFL>
__global__ __launch_bounds__(NTH) void k_read_narrow(const char *__restrict__ in, int *__restrict__ out, int rows)
{
    constexpr int NCG = 2048 / W, SH = 128 / W;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, sector = within % SH, lg = xcd + 8 * (within / SH);
    const int cg = lg * SH + sector;
    (void)NCG;
    int acc = 0;
    for (int p0 = threadIdx.x; p0 < rows; p0 += NTH * INFL) {
        if constexpr (W == 8) {
            v2i v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; u++) { const int row = p0 + u * NTH; v[u] = row < rows ? *(const v2i *)(in + (size_t)row * 2048 + cg * 8) : (v2i){0, 0}; }
#pragma unroll
            for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y;
        } else {
            v4i v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; u++) { const int row = p0 + u * NTH; v[u] = row < rows ? *(const v4i *)(in + (size_t)row * 2048 + cg * 16) : (v4i){0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}
template <int W, int NTH, int INFL> void run_narrow(const v4i *in, int *out, int rows)
{
    const int grid = 2048 / W;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_read_narrow<W, NTH, INFL>), dim3(grid), dim3(NTH), 0, 0, (const char *)in, out, rows);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_read_narrow<W, NTH, INFL>), dim3(grid), dim3(NTH), 0, 0, (const char *)in, out, rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 20;
    printf("narrow W=%2d threads=%4d inflight=%2d grid=%3d: %7.1f us  %6.0f GB/s\n", W, NTH, INFL, grid, ms * 1e3, (double)rows * 2048 / ms / 1e6);
}

template <int W, int MAP, int NT>
__global__ __launch_bounds__(1024) void k_read(const v4i *__restrict__ in, int *__restrict__ out, int rows, int tsplit)
{
    constexpr int NCG = 2048 / W, PPR = W / 16, SH = 128 / W;  // column groups, pieces per row, groups sharing a line
    int cg, tq;
    if (MAP == 0 || SH == 1) { cg = blockIdx.x % NCG; tq = blockIdx.x / NCG; }
    else {
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, sector = within % SH, combo = xcd + 8 * (within / SH);
        const int lg = combo % (NCG / SH);
        tq = combo / (NCG / SH);
        cg = lg * SH + sector;
    }
    const int rows_per = rows / tsplit, r0 = tq * rows_per;
    const int npieces = rows_per * PPR;
    v4i acc = (v4i){0, 0, 0, 0};
    for (int p0 = threadIdx.x; p0 < npieces; p0 += 1024 * 8) {
        v4i v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = p0 + u * 1024;
            const int row = r0 + p / PPR, piece = p % PPR;
            const v4i *src = in + (size_t)row * 128 + cg * PPR + piece;
            v[u] = (p < npieces) ? (NT ? __builtin_nontemporal_load(src) : *src) : (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    const int s = acc.x + acc.y + acc.z + acc.w;
    if (s == 0x12345678) out[blockIdx.x] = s;
}

template <int W, int MAP, int NT> void run(const v4i *in, int *out, int rows, int tsplit)
{
    const int grid = (2048 / W) * tsplit;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_read<W, MAP, NT>), dim3(grid), dim3(1024), 0, 0, in, out, rows, tsplit);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_read<W, MAP, NT>), dim3(grid), dim3(1024), 0, 0, in, out, rows, tsplit);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 20;
    printf("W=%3d map=%d nt=%d tsplit=%2d grid=%4d: %7.1f us  %6.0f GB/s\n", W, MAP, NT, tsplit, grid, ms * 1e3, (double)rows * 2048 / ms / 1e6);
}

int main()
{
    const int rows = 1024 * 64;
    v4i *in; int *out;
    CK(hipMalloc(&in, (size_t)rows * 2048)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(in, 1, (size_t)rows * 2048));
    run<128, 0, 0>(in, out, rows, 16); run<128, 0, 1>(in, out, rows, 16); run<128, 0, 0>(in, out, rows, 32);
    run<64, 0, 0>(in, out, rows, 8); run<64, 1, 0>(in, out, rows, 8); run<64, 1, 1>(in, out, rows, 8); run<64, 1, 0>(in, out, rows, 16);
    run<32, 0, 0>(in, out, rows, 4); run<32, 1, 0>(in, out, rows, 4); run<32, 1, 1>(in, out, rows, 4); run<32, 1, 0>(in, out, rows, 8); run<32, 0, 0>(in, out, rows, 8);
    run_narrow<8, 256, 16>(in, out, rows); run_narrow<8, 256, 32>(in, out, rows); run_narrow<8, 512, 16>(in, out, rows); run_narrow<8, 1024, 8>(in, out, rows);
    run_narrow<16, 512, 16>(in, out, rows); run_narrow<16, 1024, 8>(in, out, rows); run_narrow<16, 1024, 16>(in, out, rows);
    return 0;
}
