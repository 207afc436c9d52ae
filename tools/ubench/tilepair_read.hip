// Microbenchmark for a pair-partitioned fused int8 X-engine: workgroup (channel group of 64 channels = one 128-byte line per
// (t, station), row-tile pair (bi, bj)) streams the lines of the stations of its one or two row tiles for all T.  Every line is
// needed by the 4 workgroups whose pair contains its tile; all 10 pair-workgroups of a channel group are pinned to one XCD so
// that the re-reads can hit L2.  How long does the whole read take (134 MB unique, 537 MB requested)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TSPLIT, bool PIN>
__global__ __launch_bounds__(1024) void k(const v4i *__restrict__ in, int *__restrict__ out, int T, int N, int F)
{
    const int ncg = F / 64, npair = 10;
    int cg, pair, ts;
    if (PIN) {
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;  // all pairs (and time ranges) of a channel group on one XCD
        pair = within % npair;
        const int rest = within / npair;
        ts = rest % TSPLIT;
        cg = xcd + 8 * (rest / TSPLIT);
    } else {
        pair = blockIdx.x % npair;
        const int rest = blockIdx.x / npair;
        ts = rest % TSPLIT;
        cg = rest / TSPLIT;
    }
    if (cg >= ncg) return;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= pair) bi++;
    const int bj = pair - bi * (bi + 1) / 2;
    const int ntile = (bi == bj) ? 1 : 2;
    const int t0 = ts * (T / TSPLIT), t1 = t0 + T / TSPLIT;
    const size_t row_v4 = (size_t)F * 2 / 16;  // 16-byte pieces per (t, station) row
    v4i acc = (v4i){0, 0, 0, 0};
    for (int tb = t0; tb < t1; tb += 16) {
        // 16 t x (16 * ntile) stations x 8 pieces
        const int items = 16 * 16 * ntile * 8;
        v4i v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = threadIdx.x + 1024 * k;
            v[k] = (v4i){0, 0, 0, 0};
            if (idx < items) {
                const int piece = idx & 7, sl = (idx >> 3) % (16 * ntile), t = idx / (8 * 16 * ntile);
                const int s = (sl < 16 ? bi : bj) * 16 + (sl & 15);
                v[k] = in[((size_t)(tb + t) * N + s) * row_v4 + (size_t)cg * 8 + piece];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc += v[k];
    }
    const int s = acc.x + acc.y + acc.z + acc.w;
    if (s == 0x12345678) out[blockIdx.x] = s;
}

template <int TSPLIT, bool PIN> void run(const v4i *in, int *out)
{
    const int T = 1024, N = 64, F = 1024;
    const int grid = (F / 64) * 10 * TSPLIT;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<TSPLIT, PIN>), dim3(grid), dim3(1024), 0, 0, in, out, T, N, F);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<TSPLIT, PIN>), dim3(grid), dim3(1024), 0, 0, in, out, T, N, F);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("tsplit=%d pinned=%d grid=%3d: %7.1f us\n", TSPLIT, (int)PIN, grid, ms / 20 * 1e3);
}


// half-line variant: workgroup = (32-channel half of a line, pair), 512 threads, 8 pieces per thread and K block of 32 steps
template <bool PIN>
__global__ __launch_bounds__(512) void kh(const v4i *__restrict__ in, int *__restrict__ out, int T, int N, int F)
{
    const int nlines = F / 64, npair = 10, per_line = 2 * npair;
    int line, within;
    if (PIN) { const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3; line = xcd + 8 * (w / per_line); within = w % per_line; }
    else { line = blockIdx.x / per_line; within = blockIdx.x % per_line; }
    if (line >= nlines) return;
    const int half = within & 1, pair = within >> 1;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= pair) bi++;
    const int bj = pair - bi * (bi + 1) / 2;
    const int tid = threadIdx.x, o = tid & 3, sl = (tid >> 2) & 31, to = tid >> 7;
    const int station = ((sl >> 4) ? bj : bi) * 16 + (sl & 15);
    const bool loads = !(bi == bj && (sl >> 4));
    const size_t row_v4 = (size_t)F * 2 / 16, tstr = (size_t)N * row_v4;
    const v4i *src = in + ((size_t)(to * 8) * N + station) * row_v4 + (size_t)line * 8 + half * 4 + o;
    v4i acc = (v4i){0, 0, 0, 0};
    if (loads)
        for (int tb = 0; tb < T; tb += 32) {
            v4i v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = src[(size_t)(tb + i) * tstr];
#pragma unroll
            for (int i = 0; i < 8; i++) acc += v[i];
        }
    const int s = acc.x + acc.y + acc.z + acc.w;
    if (s == 0x12345678) out[blockIdx.x] = s;
}
template <bool PIN> void runh(const v4i *in, int *out)
{
    const int T = 1024, N = 64, F = 1024, grid = (F / 64) * 20;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((kh<PIN>), dim3(grid), dim3(512), 0, 0, in, out, T, N, F);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((kh<PIN>), dim3(grid), dim3(512), 0, 0, in, out, T, N, F);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("half lines, 512 threads, pinned=%d grid=%3d: %7.1f us\n", (int)PIN, grid, ms / 20 * 1e3);
}

int main()
{
    v4i *in; int *out;
    const size_t bytes = (size_t)1024 * 64 * 1024 * 2;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(in, 1, bytes));
    run<1, true>(in, out); run<1, false>(in, out); run<2, true>(in, out); run<2, false>(in, out); run<4, true>(in, out);
    runh<true>(in, out); runh<false>(in, out);
    return 0;
}
