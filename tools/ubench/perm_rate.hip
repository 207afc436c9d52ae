// Microbenchmark (round 5): issue rate of the byte / halfword shuffles a register transpose can be built from, one wave per SIMD and two,
// independent (eight destinations in rotation) and as a dependent chain.  Shader clocks per instruction and wave (s_memtime).
//   v_perm_b32 (SGPR / VGPR selector), v_pack_b32_f16 (both op_sel forms), v_and_or_b32, v_bfi_b32, v_alignbit_b32, v_lshl_or_b32, v_xor_b32, v_not_b32
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define OP8(INS)                                                                                                                        \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                               \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])                       \
                 : "v"(a), "v"(b), "s"(sel), "v"(vsel))
#define I_PERM_S(n) "v_perm_b32 %" #n ", %8, %9, %10\n\t"
#define I_PERM_V(n) "v_perm_b32 %" #n ", %8, %9, %11\n\t"
#define I_PACK(n) "v_pack_b32_f16 %" #n ", %8, %9\n\t"
#define I_PACKH(n) "v_pack_b32_f16 %" #n ", %8, %9 op_sel:[1,1,0]\n\t"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %8, %10, %9\n\t"
#define I_BFI(n) "v_bfi_b32 %" #n ", %10, %8, %9\n\t"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %8, %9, 16\n\t"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %8, 16, %9\n\t"
#define I_XOR(n) "v_xor_b32 %" #n ", %8, %9\n\t"
#define I_NOT(n) "v_not_b32 %" #n ", %8\n\t"
// dependent chains: destination n is also the first source
#define D_PERM_S(n) "v_perm_b32 %0, %0, %9, %10\n\t"
#define D_PACK(n) "v_pack_b32_f16 %0, %0, %9\n\t"
#define D_XOR(n) "v_xor_b32 %0, %0, %9\n\t"

template <int MODE> __global__ void k(int iters, unsigned long long *out, unsigned *sink)
{
    unsigned d[8];
    for (int i = 0; i < 8; i++) d[i] = threadIdx.x * (i + 1);
    const unsigned a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5, vsel = 0x05010400u;
    const unsigned sel = 0x07030602u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if constexpr (MODE == 0) OP8(I_PERM_S);
            else if constexpr (MODE == 1) OP8(I_PERM_V);
            else if constexpr (MODE == 2) OP8(I_PACK);
            else if constexpr (MODE == 3) OP8(I_PACKH);
            else if constexpr (MODE == 4) OP8(I_ANDOR);
            else if constexpr (MODE == 5) OP8(I_BFI);
            else if constexpr (MODE == 6) OP8(I_ALIGN);
            else if constexpr (MODE == 7) OP8(I_LSHLOR);
            else if constexpr (MODE == 8) OP8(I_XOR);
            else if constexpr (MODE == 9) OP8(I_NOT);
            else if constexpr (MODE == 10) OP8(D_PERM_S);
            else if constexpr (MODE == 11) OP8(D_PACK);
            else OP8(D_XOR);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    unsigned x = 0;
    for (int i = 0; i < 8; i++) x ^= d[i];
    if (x == 0x12345679u) sink[0] = x;
}

template <int MODE> int run(const char *name, unsigned long long *dptr, unsigned *sink)
{
    const int iters = 4000;
    for (int threads = 256; threads <= 512; threads += 256) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, dptr, sink);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, dptr, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, dptr, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-44s %d wave(s) per SIMD: %6.2f clocks per instruction and wave\n", name, threads / 256, (double)h[0] / ((double)iters * 64));
    }
    return 0;
}

int main()
{
    unsigned long long *d;
    unsigned *sink;
    CK(hipMalloc(&d, 256 * 16 * 8));
    CK(hipMalloc(&sink, 64));
    run<0>("v_perm_b32, SGPR selector", d, sink);
    run<1>("v_perm_b32, VGPR selector", d, sink);
    run<2>("v_pack_b32_f16", d, sink);
    run<3>("v_pack_b32_f16 op_sel:[1,1,0]", d, sink);
    run<4>("v_and_or_b32", d, sink);
    run<5>("v_bfi_b32", d, sink);
    run<6>("v_alignbit_b32", d, sink);
    run<7>("v_lshl_or_b32", d, sink);
    run<8>("v_xor_b32", d, sink);
    run<9>("v_not_b32", d, sink);
    run<10>("v_perm_b32, dependent chain", d, sink);
    run<11>("v_pack_b32_f16, dependent chain", d, sink);
    run<12>("v_xor_b32, dependent chain", d, sink);
    return 0;
}
