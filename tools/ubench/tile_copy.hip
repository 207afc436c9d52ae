// The memory pattern of the two-pass tile FFT without the transform: a workgroup moves tiles of R rows x 128 bytes (sixteen 8-byte
// columns) whose rows are LD elements apart.  mode 0: tile -> the same tile of the output (pass B); mode 1: tile -> R x 16 contiguous
// values (pass A's k1-contiguous stores).  What does the pattern alone reach for R x LD = 256 x 256 ... 1024 x 1024, and how many
// workgroups per CU does it need?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TH, int MODE>
__global__ __launch_bounds__(TH) void k_tile(const f2 *__restrict__ in, f2 *__restrict__ out, int R, int ld, long long nitems)
{
    const int tiles = ld / 16, per = R * 16 / TH;  // elements per thread
    const size_t frame_elems = (size_t)R * ld;
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const long long frame = item / tiles;
        const int c0 = (int)(item - frame * tiles) * 16;
        for (int i0 = 0; i0 < per; i0 += 16) {
            f2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = threadIdx.x + TH * (i0 + i), col = e & 15, row = e >> 4;
                v[i] = __builtin_nontemporal_load(in + frame * frame_elems + (size_t)row * ld + c0 + col);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = threadIdx.x + TH * (i0 + i), col = e & 15, row = e >> 4;
                v[i].x += 1.0f;
                if (MODE == 0) __builtin_nontemporal_store(v[i], out + frame * frame_elems + (size_t)row * ld + c0 + col);
                else out[frame * frame_elems + (size_t)c0 * R + e] = v[i];
            }
        }
    }
}

int main()
{
    const size_t n = (size_t)1 << 26;
    f2 *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
    CK(hipMemset(in, 0, n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][2] = {{256, 256}, {256, 512}, {512, 256}, {512, 512}, {512, 1024}, {1024, 512}, {1024, 1024}, {256, 1024}, {1024, 256}};
    for (auto &sh : shapes) {
        const int R = sh[0], ld = sh[1];
        const long long items = (long long)(n / ((size_t)R * ld)) * (ld / 16);
        for (int mode = 0; mode < 2; mode++)
            for (int th : {256, 1024})
                for (int wpc : {1, 2, 4, 8}) {
                    if ((th == 1024 && wpc > 2) || th > R) continue;
                    const int grid = 256 * wpc;
                    float best = 1e9f;
                    for (int rep = 0; rep < 5; rep++) {
                        CK(hipEventRecord(e0));
                        if (th == 256) { if (mode == 0) k_tile<256, 0><<<grid, 256>>>(in, out, R, ld, items); else k_tile<256, 1><<<grid, 256>>>(in, out, R, ld, items); }
                        else           { if (mode == 0) k_tile<1024, 0><<<grid, 1024>>>(in, out, R, ld, items); else k_tile<1024, 1><<<grid, 1024>>>(in, out, R, ld, items); }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    printf("R %4d ld %4d mode %d threads %4d wg/CU %d: %6.1f us  %.2f TB/s\n", R, ld, mode, th, wpc, best * 1e3, 2.0 * n * 8 / best / 1e9);
                }
    }
    return 0;
}
