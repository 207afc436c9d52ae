// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 vs v_mov_b32 (wave64, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f;
    const f2 cc = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1\n"
                             "v_fma_f32 %5, %5, %1, %1\n v_fma_f32 %6, %6, %1, %1\n v_fma_f32 %7, %7, %1, %1\n v_fma_f32 %8, %8, %1, %1"
                             : "+v"(a0) : "v"(c), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %2, %2, %1, %1\n v_pk_fma_f32 %3, %3, %1, %1\n v_pk_fma_f32 %4, %4, %1, %1\n"
                             "v_pk_fma_f32 %5, %5, %1, %1\n v_pk_fma_f32 %6, %6, %1, %1\n v_pk_fma_f32 %7, %7, %1, %1\n v_pk_fma_f32 %8, %8, %1, %1"
                             : "+v"(p0) : "v"(cc), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
            }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %1\n v_pk_add_f32 %3, %3, %1\n v_pk_add_f32 %4, %4, %1\n"
                             "v_pk_add_f32 %5, %5, %1\n v_pk_add_f32 %6, %6, %1\n v_pk_add_f32 %7, %7, %1\n v_pk_add_f32 %8, %8, %1"
                             : "+v"(p0) : "v"(cc), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n"
                             "v_add_f32 %5, %5, %1\n v_add_f32 %6, %6, %1\n v_add_f32 %7, %7, %1\n v_add_f32 %8, %8, %1"
                             : "+v"(a0) : "v"(c), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            }
        } else if constexpr (MODE == 5) {  // v_pk_fma_f32 with three distinct 64-bit VGPR operands
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %6, %5, %1\n v_pk_fma_f32 %2, %7, %5, %2\n v_pk_fma_f32 %3, %8, %5, %3\n"
                             "v_pk_fma_f32 %0, %6, %5, %0\n v_pk_fma_f32 %1, %7, %5, %1\n v_pk_fma_f32 %2, %8, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                             : "+v"(p0), "+v"(p2), "+v"(p4), "+v"(p6) : "v"(p1), "v"(cc), "v"(p3), "v"(p5), "v"(p7));
            }
        } else if constexpr (MODE == 6) {  // v_fma_f32 with three distinct VGPR operands
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %6, %5, %1\n v_fma_f32 %2, %7, %5, %2\n v_fma_f32 %3, %8, %5, %3\n"
                             "v_fma_f32 %0, %6, %5, %0\n v_fma_f32 %1, %7, %5, %1\n v_fma_f32 %2, %8, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                             : "+v"(a0), "+v"(a2), "+v"(a4), "+v"(a6) : "v"(a1), "v"(c), "v"(a3), "v"(a5), "v"(a7));
            }
        } else if constexpr (MODE == 7) {  // v_fmac_f32 (2-operand encoding, accumulator in place)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %6, %5\n v_fmac_f32 %2, %7, %5\n v_fmac_f32 %3, %8, %5\n"
                             "v_fmac_f32 %0, %6, %5\n v_fmac_f32 %1, %7, %5\n v_fmac_f32 %2, %8, %5\n v_fmac_f32 %3, %4, %5"
                             : "+v"(a0), "+v"(a2), "+v"(a4), "+v"(a6) : "v"(a1), "v"(c), "v"(a3), "v"(a5), "v"(a7));
            }
        } else if constexpr (MODE == 8) {  // v_pk_mul_f32 distinct operands
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %6, %5\n v_pk_mul_f32 %2, %7, %5\n v_pk_mul_f32 %3, %8, %5\n"
                             "v_pk_mul_f32 %0, %6, %5\n v_pk_mul_f32 %1, %7, %5\n v_pk_mul_f32 %2, %8, %5\n v_pk_mul_f32 %3, %4, %5"
                             : "+v"(p0), "+v"(p2), "+v"(p4), "+v"(p6) : "v"(p1), "v"(cc), "v"(p3), "v"(p5), "v"(p7));
            }
        } else if constexpr (MODE == 9) {  // v_mul_f32 distinct
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %6, %5\n v_mul_f32 %2, %7, %5\n v_mul_f32 %3, %8, %5\n"
                             "v_mul_f32 %0, %6, %5\n v_mul_f32 %1, %7, %5\n v_mul_f32 %2, %8, %5\n v_mul_f32 %3, %4, %5"
                             : "+v"(a0), "+v"(a2), "+v"(a4), "+v"(a6) : "v"(a1), "v"(c), "v"(a3), "v"(a5), "v"(a7));
            }
        } else if constexpr (MODE == 10) {  // v_pk_fma_f32, tap in an SGPR pair, low half broadcast (the FIR inner loop's form)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %6, %5, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %7, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, %5, %3 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %0, %6, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %7, %5, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %8, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[1,0,1]"
                             : "+v"(p0), "+v"(p2), "+v"(p4), "+v"(p6) : "v"(p1), "s"(cc), "v"(p3), "v"(p5), "v"(p7));
            }
        } else if constexpr (MODE == 11) {  // same with the tap in a VGPR pair and op_sel broadcast
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %6, %5, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %7, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, %5, %3 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %0, %6, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %7, %5, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %8, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[1,0,1]"
                             : "+v"(p0), "+v"(p2), "+v"(p4), "+v"(p6) : "v"(p1), "v"(cc), "v"(p3), "v"(p5), "v"(p7));
            }
        } else if constexpr (MODE == 12) {  // two v_fma_f32 with an SGPR tap instead of one packed
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fmac_f32 %0, %5, %4\n v_fmac_f32 %1, %5, %6\n v_fmac_f32 %2, %5, %7\n v_fmac_f32 %3, %5, %8\n"
                             "v_fmac_f32 %0, %5, %6\n v_fmac_f32 %1, %5, %7\n v_fmac_f32 %2, %5, %8\n v_fmac_f32 %3, %5, %4"
                             : "+v"(a0), "+v"(a2), "+v"(a4), "+v"(a6) : "v"(a1), "s"(c), "v"(a3), "v"(a5), "v"(a7));
            }
        } else if constexpr (MODE == 13) {  // explicit registers: accumulator and x pairs in the SAME VGPR bank pair (index % 4 == 0)
#pragma unroll
            for (int u = 0; u < 8; u++)
                asm volatile("v_pk_fma_f32 v[32:33], v[48:49], %0, v[32:33] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[36:37], v[52:53], %0, v[36:37] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[40:41], v[56:57], %0, v[40:41] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[44:45], v[60:61], %0, v[44:45] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[32:33], v[52:53], %0, v[32:33] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[36:37], v[56:57], %0, v[36:37] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[40:41], v[60:61], %0, v[40:41] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[44:45], v[48:49], %0, v[44:45] op_sel_hi:[1,0,1]" :: "s"(cc) : "v32","v33","v36","v37","v40","v41","v44","v45","v48","v49","v52","v53","v56","v57","v60","v61");
        } else if constexpr (MODE == 14) {  // accumulators at index % 4 == 2, x pairs at index % 4 == 0
#pragma unroll
            for (int u = 0; u < 8; u++)
                asm volatile("v_pk_fma_f32 v[34:35], v[48:49], %0, v[34:35] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[38:39], v[52:53], %0, v[38:39] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[42:43], v[56:57], %0, v[42:43] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[46:47], v[60:61], %0, v[46:47] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[34:35], v[52:53], %0, v[34:35] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[38:39], v[56:57], %0, v[38:39] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[42:43], v[60:61], %0, v[42:43] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[46:47], v[48:49], %0, v[46:47] op_sel_hi:[1,0,1]" :: "s"(cc) : "v34","v35","v38","v39","v42","v43","v46","v47","v48","v49","v52","v53","v56","v57","v60","v61");
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n"
                             "v_mov_b32 %5, %1\n v_mov_b32 %6, %1\n v_mov_b32 %7, %1\n v_mov_b32 %8, %1"
                             : "+v"(a0) : "v"(c), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + p0.x + p1.y;
}

template <int MODE> void run(const char *name, float *out, int waves_per_simd)
{
    const int iters = 4096, grid = 256 * waves_per_simd;  // 256 threads = 1 wave per SIMD per block
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double insts_per_simd = (double)iters * 64 * waves_per_simd;  // wave-instructions issued on each SIMD
    printf("%-14s waves/SIMD=%d: %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}

int main()
{
    float *out; (void)hipMalloc(&out, 4 * 256 * 8 * 256);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", out, w); run<1>("v_pk_fma_f32", out, w); run<2>("v_pk_add_f32", out, w); run<3>("v_add_f32", out, w); run<4>("v_mov_b32", out, w);
        run<5>("pk_fma 3src", out, w); run<6>("fma 3src", out, w); run<7>("fmac", out, w); run<8>("pk_mul", out, w); run<9>("mul", out, w);
        run<10>("pk_fma sgpr bc", out, w); run<11>("pk_fma vgpr bc", out, w); run<12>("fmac sgpr", out, w); run<13>("pk_fma samebank", out, w); run<14>("pk_fma diffbank", out, w);
    }
    return 0;
}
