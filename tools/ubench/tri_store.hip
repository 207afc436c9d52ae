// Microbenchmark: writing the X-engine's output (per channel the lower triangle of a 256 x 256 matrix of 8-byte elements, rows packed
// back to back: row r1 starts at element r1 (r1 + 1) / 2) with the store shapes a 16 x 16 MFMA accumulator tile offers.
//   A: lane l of a 16-lane group writes 8 B, the 16 lanes one contiguous 128-byte run (C layout, row-tile operand first)
//   B: lane (r, q) writes 32 B = 2 x 16 B of row r (column-tile operand first): rows across lanes
//   C: as A after a pairwise lane exchange: 8 even lanes write 128 contiguous bytes as 16 B each, odd lanes another row
//   D: ideal: every lane 16 B, the wave 1 KiB contiguous (not a layout the accumulators offer; the ceiling)
// Every variant writes each byte of the 135 MB (512 channels) exactly once; one workgroup per channel pass, 8 waves, wave w rows w, 15 - w.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NT = 16, A = 256, NB = A * (A + 1) / 2;

template <int MODE> __global__ __launch_bounds__(512) void k(float *out, int chans_per_wg)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    for (int c = 0; c < chans_per_wg; c++) {
        float *chan = out + ((size_t)(blockIdx.x * chans_per_wg + c) * NB) * 2;
        if (MODE == 3) {  // contiguous: the channel's bytes split evenly over the 8 waves
            const size_t n16 = (size_t)NB * 8 / 16;
            for (size_t i = (size_t)wave * 64 + lane; i < n16; i += 512) *(f4 *)(chan + i * 4) = (f4){1.f, 2.f, 3.f, 4.f};
            continue;
        }
        for (int half = 0; half < 2; half++) {
            const int bi = half ? NT - 1 - wave : wave;
            for (int bj = 0; bj <= bi; bj++) {
                if (MODE == 0) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int r1 = bi * 16 + 4 * q + reg, r2 = bj * 16 + r;
                        if (r2 <= r1) *(f2 *)(chan + ((size_t)r1 * (r1 + 1) / 2 + r2) * 2) = (f2){1.f, 2.f};
                    }
                } else if (MODE == 1) {
                    const int r1 = bi * 16 + r, c0 = bj * 16 + 4 * q;
                    float *d = chan + ((size_t)r1 * (r1 + 1) / 2 + c0) * 2;
                    if (c0 + 3 <= r1) { *(f4 *)d = (f4){1.f, 2.f, 3.f, 4.f}; *(f4 *)(d + 4) = (f4){1.f, 2.f, 3.f, 4.f}; }
                    else for (int e = 0; e < 4; e++) if (c0 + e <= r1) *(f2 *)(d + 2 * e) = (f2){1.f, 2.f};
                } else {
                    // even lanes: row 4 q + {0, 2} columns (r, r + 1); odd lanes: rows 4 q + {1, 3} columns (r - 1, r)
#pragma unroll
                    for (int k2 = 0; k2 < 2; k2++) {
                        const int r1 = bi * 16 + 4 * q + 2 * k2 + (r & 1), c0 = bj * 16 + (r & ~1);
                        float *d = chan + ((size_t)r1 * (r1 + 1) / 2 + c0) * 2;
                        if (c0 + 1 <= r1) *(f4 *)d = (f4){1.f, 2.f, 3.f, 4.f};
                        else if (c0 <= r1) *(f2 *)d = (f2){1.f, 2.f};
                    }
                }
            }
        }
    }
}

template <int MODE> void run(float *out, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, 2);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: %7.1f us  %5.2f TB/s\n", name, ms / 20 * 1e3, 512.0 * NB * 8 / (ms / 20 * 1e-3) / 1e12);
}

int main()
{
    float *out;
    CK(hipMalloc(&out, (size_t)512 * NB * 8 + 4096));
    run<0>(out, "A  8 B per lane, 16 lanes contiguous    ");
    run<1>(out, "B  2 x 16 B per lane, rows across lanes  ");
    run<2>(out, "C  16 B per lane, alternate lanes        ");
    run<3>(out, "D  contiguous 16 B per lane (ceiling)    ");
    run<0>(out, "A  again                                 ");
    return 0;
}
