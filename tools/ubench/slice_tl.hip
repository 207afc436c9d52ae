// Per-line finish times of the X-engine's slice-read pattern (round 4): [rows][2 KiB] image, a workgroup reads a W-byte column slice of a
// contiguous quarter of the rows; the workgroups of a 128-byte line sit on one XCD.  Flavours: loads to registers, LDS-DMA; W = 32 / 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// W = 32: 2 lanes per row, 256 workgroups (64 slices x 4 ranges); W = 128: 8 lanes per row, 64 workgroups (16 lines x 4 ranges)
template <int W, int INFL>
__global__ __launch_bounds__(512) void k_slice(const char *__restrict__ in, unsigned long long *ts, int rows, size_t stride, int order)
{
    constexpr int LPR = W / 16, NSL = 2048 / W, SPL = 128 / W;  // lanes per row, slices per row, slices per line
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3, sector = within % SPL, combo = xcd + 8 * (within / SPL);
    const int line = combo % 16, q = combo / 16, slice = line * SPL + sector;
    (void)NSL;
    const int rows_per = rows / 4, r0 = q * rows_per;
    const char *base = in + (size_t)r0 * stride + (size_t)slice * W + (threadIdx.x % LPR) * 16;
    int acc = 0;
    const int RPI = 512 / LPR;  // rows per wave-set instruction
    for (int p0 = threadIdx.x / LPR; p0 < rows_per; p0 += RPI * INFL) {
        v4i v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) {
            int r = p0 + u * RPI;
            if (order == 1) r = (r & ~63) | ((r * 5) & 63);           // stations permuted within a time step
            if (order == 2) r = (r & ~4095) | ((r & 63) << 6) | ((r >> 6) & 63);  // station <-> time step (64 x 64 blocks)
            if (order == 3) r = (r * 33) & (rows_per - 1);            // stride 33 rows
            if (order == 4) r = (r * 1025) & (rows_per - 1);          // stride 1025 rows (2 MiB + 2 KiB)
            if (order == 5 && (q & 1)) r = rows_per - 1 - r;          // odd ranges backwards
            if (order == 6) r = (r & ~1023) | ((r & 15) << 6) | ((r >> 4) & 63);  // 16 time steps x 64 stations, time fastest
            const char *p = base + (size_t)r * stride;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u]) : "v"(p));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFL; u++) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    }
    __syncthreads();
    if (threadIdx.x == 0) ts[b] = wall_clock64();
    if (acc == 0x12345678) ts[b] = 0;
}
// W = 32 with scalar-load prefetch of the NEXT iteration's lines into L2: the four workgroups of a line share the work (each a quarter of the
// rows), every wave 64 s_load_dword per iteration of 2048 rows
template <int INFL, int AHEAD>
__global__ __launch_bounds__(512) void k_slice_pf(const char *__restrict__ in, unsigned long long *ts, int rows, size_t stride)
{
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3, sector = within % 4, combo = xcd + 8 * (within / 4);
    const int line = combo % 16, q = combo / 16, slice = line * 4 + sector;
    const int rows_per = rows / 4, r0 = q * rows_per;
    const char *base = in + (size_t)r0 * stride + (size_t)slice * 32 + (threadIdx.x % 2) * 16;
    const char *lbase = in + (size_t)r0 * stride + (size_t)line * 128;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int acc = 0;
    const int RPI = 256;
    for (int p0 = threadIdx.x / 2; p0 < rows_per; p0 += RPI * INFL) {
        const int it_row0 = __builtin_amdgcn_readfirstlane(p0 - (int)(threadIdx.x / 2)) + AHEAD * RPI * INFL;  // first row of the iteration to prefetch
        if (it_row0 < rows_per) {
            const int per_wg = RPI * INFL / 4, per_wave = per_wg / 8;
            const char *pf = lbase + (size_t)(it_row0 + sector * per_wg + wave * per_wave) * stride;
            for (int i = 0; i < per_wave; i++) {
                // (the result is never used; it lands asynchronously, so it goes to a register the allocator does not reach in this kernel)
                asm volatile("s_load_dword s100, %0, 0x0" : : "s"(pf + (size_t)i * stride) : "memory", "s100");
            }
        }
        v4i v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) {
            const char *p = base + (size_t)(p0 + u * RPI) * stride;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u]) : "v"(p));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < INFL; u++) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    }
    __syncthreads();
    if (threadIdx.x == 0) ts[b] = wall_clock64();
    if (acc == 0x12345678) ts[b] = 0;
}
// Mapping "M2" (every XCD reads all 16 lines of one quarter of the rows) with optional L2 prefetch of the slow line class by ALL 32 workgroups
// of the XCD: one lane per line, 64 rows per workgroup and iteration, one iteration ahead.
template <int INFL, int PF>
__global__ __launch_bounds__(512) void k_slice_m2(const char *__restrict__ in, unsigned long long *ts, int rows, size_t stride)
{
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3, sector = within & 3;
    const int q = xcd >> 1, line = (xcd & 1) * 8 + (within >> 2), slice = line * 4 + sector;
    const int rows_per = rows / 4, r0 = q * rows_per;
    const char *base = in + (size_t)r0 * stride + (size_t)slice * 32 + (threadIdx.x % 2) * 16;
    const char *slow = in + (size_t)r0 * stride + (size_t)((xcd & 1) * 8 + 3) * 128;  // this XCD's line of the slow class
    int acc = 0;
    const int RPI = 256;
    for (int p0 = threadIdx.x / 2; p0 < rows_per; p0 += RPI * INFL) {
        int pf = 0;
        if (PF && threadIdx.x < 64) {
            const int row = (p0 - (int)(threadIdx.x / 2)) + PF * RPI * INFL + within + 32 * (int)threadIdx.x;
            if (row < rows_per) asm volatile("global_load_dword %0, %1, off" : "=&v"(pf) : "v"(slow + (size_t)row * stride));
        }
        v4i v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) {
            const char *p = base + (size_t)(p0 + u * RPI) * stride;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u]) : "v"(p));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(pf));
        acc += pf;
#pragma unroll
        for (int u = 0; u < INFL; u++) { asm volatile("" : "+v"(v[u])); acc += v[u].x + v[u].w; }
    }
    __syncthreads();
    if (threadIdx.x == 0) ts[b] = wall_clock64();
    if (acc == 0x12345678) ts[b] = 0;
}
template <int INFL, int PF> int run_m2(const char *in, unsigned long long *ts, int rows, size_t stride, const char *what);
__global__ void k_t0(unsigned long long *ts) { ts[0] = wall_clock64(); }

template <int INFL, int AHEAD> int run_pf(const char *in, unsigned long long *ts, int rows, size_t stride, const char *what)
{
    const int grid = 256;
    std::vector<unsigned long long> h(grid + 1);
    double fin[16] = {0};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_t0, dim3(1), dim3(1), 0, 0, ts + grid);
        hipLaunchKernelGGL((k_slice_pf<INFL, AHEAD>), dim3(grid), dim3(512), 0, 0, in, ts, rows, stride);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), ts, (grid + 1) * 8, hipMemcpyDeviceToHost));
        for (int l = 0; l < 16; l++) fin[l] = 0;
        for (int b = 0; b < grid; b++) {
            const int xcd = b & 7, within = b >> 3, combo = xcd + 8 * (within / 4), line = combo % 16;
            fin[line] = std::max(fin[line], (double)(h[b] - h[grid]) * 0.01);
        }
    }
    printf("%-44s finish per line (us after a marker kernel):", what);
    for (int l = 0; l < 16; l++) printf(" %5.1f", fin[l]);
    printf("\n");
    return 0;
}

template <int INFL, int PF> int run_m2(const char *in, unsigned long long *ts, int rows, size_t stride, const char *what)
{
    const int grid = 256;
    std::vector<unsigned long long> h(grid + 1);
    double fin[16] = {0};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_t0, dim3(1), dim3(1), 0, 0, ts + grid);
        hipLaunchKernelGGL((k_slice_m2<INFL, PF>), dim3(grid), dim3(512), 0, 0, in, ts, rows, stride);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), ts, (grid + 1) * 8, hipMemcpyDeviceToHost));
        for (int l = 0; l < 16; l++) fin[l] = 0;
        for (int b = 0; b < grid; b++) {
            const int xcd = b & 7, within = b >> 3, line = (xcd & 1) * 8 + (within >> 2);
            fin[line] = std::max(fin[line], (double)(h[b] - h[grid]) * 0.01);
        }
    }
    printf("%-44s finish per line (us after a marker kernel):", what);
    for (int l = 0; l < 16; l++) printf(" %5.1f", fin[l]);
    printf("\n");
    return 0;
}

static size_t g_rotate = 0;  // bytes between the images of consecutive repetitions (0: the same image, i.e. served by the 256 MiB Infinity Cache)
template <int W, int INFL> int run(const char *in0, unsigned long long *ts, int rows, size_t stride, int order, const char *what)
{
    const int grid = (2048 / W) * 4;
    std::vector<unsigned long long> h(grid + 1);
    double fin[16] = {0};
    for (int rep = 0; rep < 3; rep++) {
        const char *in = in0 + (size_t)rep * g_rotate;
        hipLaunchKernelGGL(k_t0, dim3(1), dim3(1), 0, 0, ts + grid);
        hipLaunchKernelGGL((k_slice<W, INFL>), dim3(grid), dim3(512), 0, 0, in, ts, rows, stride, order);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), ts, (grid + 1) * 8, hipMemcpyDeviceToHost));
        for (int l = 0; l < 16; l++) fin[l] = 0;
        for (int b = 0; b < grid; b++) {
            constexpr int SPL = 128 / W;
            const int xcd = b & 7, within = b >> 3, combo = xcd + 8 * (within / SPL), line = combo % 16;
            fin[line] = std::max(fin[line], (double)(h[b] - h[grid]) * 0.01);
        }
    }
    printf("%-44s finish per line (us after a marker kernel):", what);
    for (int l = 0; l < 16; l++) printf(" %5.1f", fin[l]);
    printf("\n");
    return 0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rows = 1024 * 64;
    char *buf; unsigned long long *ts;
    const size_t big = (size_t)9 << 30;  // room for the far-offset cases below
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&ts, 1 << 16));
    CK(hipMemset(buf, 1, big));
    printf("buffer at %p\n", (void *)buf);
    run<32, 8>(buf, ts, rows, 2048, 0, "W=32 stride 2048");
    g_rotate = (size_t)512 << 20;
    run<32, 8>(buf, ts, rows, 2048, 0, "W=32 stride 2048, a fresh image every run (HBM)");
    run<32, 16>(buf + ((size_t)2 << 30), ts, rows, 2048, 0, "W=32 infl 16, fresh image (HBM)");
    run<32, 4>(buf + ((size_t)4 << 30), ts, rows, 2048, 0, "W=32 infl 4, fresh image (HBM)");
    run<128, 8>(buf + ((size_t)6 << 30), ts, rows, 2048, 0, "W=128 (64 workgroups), fresh image (HBM)");
    g_rotate = 0;
    run<32, 8>(buf + ((size_t)128 << 20), ts, rows, 2048, 0, "W=32 stride 2048, base + 128 MiB");
    run<32, 8>(buf + ((size_t)1 << 30), ts, rows, 2048, 0, "W=32 stride 2048, base + 1 GiB");
    run<32, 8>(buf + ((size_t)3 << 30) + ((size_t)192 << 20), ts, rows, 2048, 0, "W=32 stride 2048, base + 3.19 GiB");
    run<32, 8>(buf + ((size_t)8 << 30), ts, rows, 2048, 0, "W=32 stride 2048, base + 8 GiB");
    run_m2<8, 0>(buf, ts, rows, 2048, "W=32 map M2 (all lines per XCD)");
    run_m2<8, 1>(buf, ts, rows, 2048, "W=32 map M2 + slow-line prefetch 1 ahead");
    run_m2<8, 2>(buf, ts, rows, 2048, "W=32 map M2 + slow-line prefetch 2 ahead");
    run_m2<4, 2>(buf, ts, rows, 2048, "W=32 infl 4 map M2 + prefetch 2 ahead");
    run_pf<8, 1>(buf, ts, rows, 2048, "W=32 + scalar prefetch 1 iteration ahead");
    run_pf<8, 2>(buf, ts, rows, 2048, "W=32 + scalar prefetch 2 iterations ahead");
    run_pf<4, 2>(buf, ts, rows, 2048, "W=32 infl 4 + scalar prefetch 2 ahead");
    run<32, 4>(buf, ts, rows, 2048, 0, "W=32 infl 4");
    run<32, 16>(buf, ts, rows, 2048, 0, "W=32 infl 16");
    run<32, 8>(buf + 1024, ts, rows, 2048, 0, "W=32 stride 2048, base + 1024");
    run<32, 8>(buf + 4096, ts, rows, 2048, 0, "W=32 stride 2048, base + 4096");
    run<32, 8>(buf + (1 << 19), ts, rows, 2048, 0, "W=32 stride 2048, base + 512 KiB");
    run<32, 8>(buf, ts, rows, 2048, 1, "W=32 stride 2048, stations permuted");
    run<32, 8>(buf, ts, rows, 2048, 2, "W=32 stride 2048, station <-> time");
    run<32, 8>(buf, ts, rows, 2048, 3, "W=32 stride 2048, 33-row steps");
    run<32, 8>(buf, ts, rows, 2048, 4, "W=32 stride 2048, 1025-row steps");
    run<32, 8>(buf, ts, rows, 2048, 5, "W=32 stride 2048, odd ranges backwards");
    run<32, 8>(buf, ts, rows, 2048, 6, "W=32 stride 2048, 16 t x 64 s, time fastest");
    run<128, 8>(buf, ts, rows, 2048, 3, "W=128 stride 2048, 33-row steps");
    run<128, 8>(buf, ts, rows, 2048, 4, "W=128 stride 2048, 1025-row steps");
    run<32, 8>(buf, ts, rows, 4096, 0, "W=32 stride 4096 (first 2 KiB of each)");
    run<32, 8>(buf, ts, rows, 2048 + 256, 0, "W=32 stride 2304");
    run<128, 8>(buf, ts, rows, 2048, 0, "W=128 stride 2048 (64 workgroups)");
    return 0;
}
