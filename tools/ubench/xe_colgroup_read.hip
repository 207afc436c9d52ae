// Microbenchmark (round 5): the REQUEST STREAM of two decompositions of the fused int8 X-engine at BASELINE config 5
// (64 stations x 1024 channels x 1024 frames, rows of 2 KiB), inputs in rotation so that every launch reads HBM.
//   S32  (today's kernel): workgroup = (32-byte column slice = 16 channels, ALL 64 stations, a time range); the four workgroups of a
//         128-byte line sit on one XCD.  One request (two lanes x 16 B) per (t, station) row.
//   L128 (candidate): workgroup = (whole 128-byte line = 64 channels, a GROUP of row-tile pairs, a time range).  The 10 pairs of the
//         lower triangle of the 4 x 4 row-tile grid go to 4 workgroups: A = {00,10,11} needs stations 0-31, B = {22,32,33} 32-63,
//         C = {20,21} 0-47, D = {30,31} 0-31 + 48-63: 2.5 x the rows, a quarter of the requests per row (eight lanes x 16 B = one line).
//         The 4 workgroups of a (line, time range) sit on one XCD, so HBM should still see every line once.
//   L128x6: six workgroups per line with two row tiles each ({00,10,11} {22,32,33} {20} {31} {30} {21}): 3 x the rows, equal loads.
// No arithmetic: LDS-DMA (global_load_lds_dwordx4) into a ring of stages, one barrier per stage, exactly the kernel's loop skeleton.
// Unit tables are built on the host, so other maps can be tried without touching the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Unit {
    unsigned long long base;  // byte offset of (window, t0, station 0, column) in the input set
    int mask;                 // row tiles (16 stations each) this unit reads, bit b = stations 16b .. 16b+15
    int nt;                   // time steps
};

constexpr int kT = 1024, kN = 64, kRow = 2048;
constexpr size_t kTStride = (size_t)kN * kRow, kWindow = (size_t)kT * kTStride;

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BYTES per row and request (32 / 64 / 128), ROWS per stage, RING stages.  8 waves; IPW DMA instructions per wave and stage.
template <int BYTES, int ROWS, int RING>
__global__ __launch_bounds__(512) void k_stream(const unsigned char *__restrict__ in, const Unit *__restrict__ units, int items, int *__restrict__ sink, unsigned long long *__restrict__ ts)
{
    if (ts && threadIdx.x == 0) ts[blockIdx.x * 2] = wall_clock64();
    constexpr int LPR = BYTES / 16, RPI = 64 / LPR, IPW = ROWS / RPI / 8, STAGE = ROWS * BYTES;
    static_assert(ROWS % (RPI * 8) == 0, "");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int probe = 0;
    // one flat stream of stages over all units of this workgroup
    int iu = 0, is = 0, ns_iss = 0, kt_iss = 0, nstage_iss = 0;  // issue cursor
    size_t off[IPW];
    const unsigned char *ubase = nullptr;
    auto setup = [&](int u) {
        const Unit un = units[(size_t)blockIdx.x * items + u];
        const int nb = __builtin_popcount(un.mask);
        ns_iss = 16 * nb;
        kt_iss = ROWS / ns_iss;
        nstage_iss = (un.nt + kt_iss - 1) / kt_iss;
        ubase = in + un.base;
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const int row = (wave * IPW + i) * RPI + lane / LPR, tl = row / ns_iss, sl = row % ns_iss;
            int m = un.mask, b = 0;
            for (int k = 0; k < (sl >> 4); k++) m &= m - 1;
            b = __builtin_ctz(m);
            off[i] = (size_t)tl * kTStride + (size_t)(b * 16 + (sl & 15)) * kRow + (lane % LPR) * 16;
        }
    };
    setup(0);
    int total = 0;
    for (int u = 0; u < items; u++) {
        const Unit un = units[(size_t)blockIdx.x * items + u];
        const int kt = ROWS / (16 * __builtin_popcount(un.mask));
        total += (un.nt + kt - 1) / kt;
    }
    int issued = 0;
    auto issue = [&]() {
        const unsigned dst0 = lds0 + (issued % RING) * STAGE;
        const size_t so = (size_t)is * kt_iss * kTStride;
#pragma unroll
        for (int i = 0; i < IPW; i++) dma16(ubase + so + off[i], __builtin_amdgcn_readfirstlane(dst0 + (wave * IPW + i) * 1024));
        issued++;
        if (++is == nstage_iss) {
            is = 0;
            if (++iu < items) setup(iu);
        }
    };
    for (int k = 0; k < RING - 1 && k < total; k++) issue();
    for (int n = 0; n < total; n++) {
        if (n + RING - 1 <= total) wait_vm<(RING - 2) * IPW>();
        else wait_vm<0>();
        __syncthreads();
        probe += *(const int *)(lds + (n % RING) * STAGE + tid * 4);  // one LDS read per stage (keeps the data path honest)
        if (issued < total) issue();
    }
    if (probe == 0x12345678) sink[blockIdx.x] = probe;
    if (ts && threadIdx.x == 0) ts[blockIdx.x * 2 + 1] = wall_clock64();
}

struct Plan { std::vector<Unit> u; int grid, items; const char *name; };

// S32: slice units (all stations), TS time ranges, W windows; items = units per workgroup (persistent)
static Plan plan_s32(int TS, int W, int grid)
{
    Plan p; p.name = "S32";
    const int units = 64 * TS * W;
    p.grid = grid; p.items = units / grid;
    p.u.resize((size_t)units);
    for (int k = 0; k < p.items; k++)
        for (int b = 0; b < grid; b++) {
            const int g = b + k * grid, xcd = g & 7, within = g >> 3, sector = within & 3, combo = xcd + 8 * (within >> 2);
            const int line = combo % 16, rest = combo / 16, q = rest % TS, win = rest / TS;
            Unit un;
            un.base = (size_t)win * kWindow + (size_t)q * (kT / TS) * kTStride + line * 128 + sector * 32;
            un.mask = 15; un.nt = kT / TS;
            p.u[(size_t)b * p.items + k] = un;
        }
    return p;
}
// L128: NG groups per (line, time range): masks[]; workgroups of one (line, range, window) consecutive on one XCD.
// swap: in the persistent form, item k of a workgroup takes group (g + k * shift) % NG so that light and heavy groups alternate
static Plan plan_l128(int TS, int W, int grid, int NG, const int *masks, int shift, const char *name, unsigned only = 0xffff)
{
    Plan p; p.name = name;
    const int units = 16 * NG * TS * W;
    p.grid = grid; p.items = units / grid;
    p.u.resize((size_t)units);
    for (int k = 0; k < p.items; k++)
        for (int b = 0; b < grid; b++) {
            const int g = b + k * grid, xcd = g & 7, within = g >> 3, grp = within % NG, combo = xcd + 8 * (within / NG);
            const int line = combo % 16, rest = combo / 16, q = rest % TS, win = rest / TS;
            Unit un;
            un.base = (size_t)win * kWindow + (size_t)q * (kT / TS) * kTStride + line * 128;
            un.mask = masks[(grp + k * shift) % NG]; un.nt = ((only >> un.mask) & 1) ? kT / TS : 0;
            p.u[(size_t)b * p.items + k] = un;
        }
    return p;
}

template <int BYTES, int ROWS, int RING> static void run(const Plan &p, const unsigned char *in, size_t set_bytes, int nsets, int W, int *sink)
{
    Unit *du;
    CK(hipMalloc(&du, p.u.size() * sizeof(Unit)));
    CK(hipMemcpy(du, p.u.data(), p.u.size() * sizeof(Unit), hipMemcpyHostToDevice));
    const int lds_bytes = ROWS * BYTES * RING;
    auto kern = k_stream<BYTES, ROWS, RING>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 24;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), lds_bytes, 0, in + (size_t)(i % nsets) * set_bytes, du, p.items, sink, (unsigned long long *)nullptr);
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), lds_bytes, 0, in + (size_t)(i % nsets) * set_bytes, du, p.items, sink, (unsigned long long *)nullptr);
    hipEventRecord(b);
    CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    size_t rows = 0;
    for (auto &u : p.u) rows += (size_t)u.nt * 16 * __builtin_popcount(u.mask);
    printf("%-8s %3d B  grid %3d x %d items, %d window(s): %8.1f us per launch, %6.1f us per window  (%.2f M requests, %.0f MB into the CUs per window)\n", p.name, BYTES,
           p.grid, p.items, W, ms / reps * 1e3, ms / reps * 1e3 / W, rows / 1e6 / W, rows * (double)BYTES / 1e6 / W);
    {
        unsigned long long *dts;
        CK(hipMalloc(&dts, p.grid * 16));
        hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), lds_bytes, 0, in, du, p.items, sink, dts);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(p.grid * 2);
        CK(hipMemcpy(h.data(), dts, p.grid * 16, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < p.grid; b++) t0 = h[2 * b] < t0 ? h[2 * b] : t0;
        double dm[16] = {0}, em[16] = {0}; int nm[16] = {0};
        double dl[16] = {0}; int nl[16] = {0};
        double emax = 0;
        for (int b = 0; b < p.grid; b++) {
            const Unit &u = p.u[(size_t)b * p.items];
            const double d = (h[2 * b + 1] - h[2 * b]) * 0.01, e = (h[2 * b + 1] - t0) * 0.01;
            dm[u.mask] += d; em[u.mask] += e; nm[u.mask]++;
            const int line = (int)((u.base % kRow) / 128);
            dl[line] += d; nl[line]++;
            emax = e > emax ? e : emax;
        }
        printf("      stamped launch: last end %.1f us; by first unit's tiles:", emax);
        for (int m = 0; m < 16; m++) if (nm[m]) printf("  mask %x: dur %.1f end %.1f", m, dm[m] / nm[m], em[m] / nm[m]);
        printf("\n      by line:");
        for (int l = 0; l < 16; l++) if (nl[l]) printf(" %.0f", dl[l] / nl[l]);
        printf("\n");
        hipFree(dts);
    }
    hipFree(du);
}

int main(int argc, char **argv)
{
    const int W = 8;
    const size_t set_bytes = (size_t)W * kWindow;  // 1 GiB: eight windows
    const int nsets = 2;
    unsigned char *in; int *sink;
    CK(hipMalloc(&in, set_bytes * nsets + (64u << 20))); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(in, 1, set_bytes * nsets));
    const int m4[4] = {0x3, 0xc, 0x7, 0xb}, m6[6] = {0x3, 0xc, 0x5, 0xa, 0x9, 0x6};
    const int mh[2] = {0x7, 0xf};
    // ---- one window per launch, four time ranges (the windows of the two sets in rotation: a launch reads window i % 16)
    // (a set here = one window, 16 of them in rotation)
    printf("one window per launch (16 windows in rotation, 2 GiB)\n");
    run<32, 1024, 4>(plan_s32(4, 1, 256), in, kWindow, 16, 1, sink);
    run<128, 384, 3>(plan_l128(4, 1, 256, 4, m4, 0, "L128x4"), in, kWindow, 16, 1, sink);
    run<128, 256, 4>(plan_l128(4, 1, 256, 4, m4, 0, "L128x4r"), in, kWindow, 16, 1, sink);  // (A/B-type stages of 8 steps, C/D of 5.33: timing only)
    run<128, 384, 3>(plan_l128(4, 1, 256, 4, m4, 0, "L128 AB", (1u << 3) | (1u << 12)), in, kWindow, 16, 1, sink);
    run<128, 384, 3>(plan_l128(4, 1, 256, 4, m4, 0, "L128 CD", (1u << 7) | (1u << 11)), in, kWindow, 16, 1, sink);
    run<128, 384, 3>(plan_l128(4, 1, 256, 4, m4, 0, "L128 A", (1u << 3)), in, kWindow, 16, 1, sink);
    run<128, 256, 4>(plan_l128(2, 1, 192, 6, m6, 0, "L128x6"), in, kWindow, 16, 1, sink);
    run<128, 256, 4>(plan_l128(4, 1, 384, 6, m6, 0, "L128x6"), in, kWindow, 16, 1, sink);
    printf("eight windows per launch, persistent workgroups (two sets in rotation, 2 GiB)\n");
    run<32, 1024, 4>(plan_s32(1, W, 256), in, set_bytes, nsets, W, sink);
    run<128, 384, 3>(plan_l128(1, W, 256, 4, m4, 0, "L128x4"), in, set_bytes, nsets, W, sink);
    run<128, 384, 3>(plan_l128(1, W, 256, 4, m4, 2, "L128x4s"), in, set_bytes, nsets, W, sink);
    run<128, 256, 4>(plan_l128(1, W, 256, 6, m6, 0, "L128x6"), in, set_bytes, nsets, W, sink);
    run<128, 256, 4>(plan_l128(2, W, 256, 4, m4, 2, "L128x4s2"), in, set_bytes, nsets, W, sink);
    (void)mh;
    return 0;
}
