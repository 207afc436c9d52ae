// Microbenchmark: combining the four time ranges' partial matrices of the fused int8 X-engine through L2 atomics instead of a
// partial-sum workspace + reduction kernel.  256 workgroups x 512 threads, workgroup b on XCD b % 8; the four workgroups
// {xcd + 8 * (4 u + r), r = 0..3} share accumulator region u * 8 + xcd (REG ints, 320 KiB = 16 channels x 20 records of 1 KiB) -- all four on one XCD, so
// atomics without a scope bit meet in that XCD's L2.
//   mode 0: nt dwordx4 stores of the region size to private regions (what the fused kernel does today)
//   mode 1: global_atomic_add_u32 (no scope bits) into the shared region
//   mode 2: the same with sc1 (agent scope)
//   mode 3: mode 1 + arrival counter; the last arriver reads the region (sc1 loads: L1 bypass), writes floats, zeroes the region
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REG = 80 * 1024;  // ints per region

template <int MODE> __global__ __launch_bounds__(512) void k(int *acc, int *priv, float *out, int *cnt, int val)
{
    const int b = blockIdx.x, xcd = b & 7, within = b >> 3, u = within >> 2, r = within & 3, region = u * 8 + xcd;
    const int tid = threadIdx.x;
    if (MODE == 0) {
        v4i *dst = (v4i *)(priv + (size_t)b * REG);
        for (int i = tid; i < REG / 4; i += 512) __builtin_nontemporal_store((v4i){val, val + i, r, tid}, dst + i);
        return;
    }
    int *a = acc + (size_t)region * REG;
    // a wave's 64 lanes on 64 consecutive dwords, 4 instructions cover one 1 KiB record
    for (int i = tid; i < REG; i += 512) {
        int *p = a + i;
        const int v = val + (i & 7);
        if (MODE == 2) asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(v) : "memory");
    }
    if (MODE != 3) return;
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int old;
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(cnt + region), "v"(1) : "memory");
        last = (old & 3) == 3;
    }
    __syncthreads();
    if (!last) return;
    float *o = out + (size_t)region * REG;
    const v4i zero = {0, 0, 0, 0};
    for (int i = tid; i < REG / 4; i += 512 * 4) {
        v4i x[4];
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++)
            if (i + k2 * 512 < REG / 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(x[k2]) : "v"((v4i *)a + i + k2 * 512) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++)
            if (i + k2 * 512 < REG / 4) {
                asm volatile("" : "+v"(x[k2]));
                __builtin_nontemporal_store((v4f){(float)x[k2][0], (float)x[k2][1], (float)x[k2][2], (float)x[k2][3]}, (v4f *)o + i + k2 * 512);
                ((v4i *)a)[i + k2 * 512] = zero;
            }
    }
}

int main()
{
    int *acc, *priv, *cnt;
    float *out;
    CK(hipMalloc(&acc, (size_t)64 * REG * 4));
    CK(hipMalloc(&priv, (size_t)256 * REG * 4));
    CK(hipMalloc(&out, (size_t)64 * REG * 4));
    CK(hipMalloc(&cnt, 64 * 4));
    CK(hipMemset(acc, 0, (size_t)64 * REG * 4));
    CK(hipMemset(cnt, 0, 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // something that streams 256 MB through the caches between launches (as the input of the next integration would)
    char *junk;
    CK(hipMalloc(&junk, 256 << 20));
    for (int mode = 0; mode < 4; mode++) {
        for (int flush = 0; flush < 2; flush++) {
            float tot = 0;
            const int it = 20;
            for (int i = 0; i < it + 3; i++) {
                if (flush) CK(hipMemsetAsync(junk, i, 256 << 20, 0));
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, acc, priv, out, cnt, 1);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, acc, priv, out, cnt, 1);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, acc, priv, out, cnt, 1);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, acc, priv, out, cnt, 1);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (i >= 3) tot += ms;
            }
            printf("mode %d flush %d: %.2f us per launch\n", mode, flush, tot / it * 1e3);
        }
        if (mode == 3) {
            std::vector<float> h((size_t)64 * REG);
            std::vector<int> ha((size_t)64 * REG);
            CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ha.data(), acc, ha.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, nz = 0;
            for (size_t i = 0; i < h.size(); i++) {
                bad += h[i] != 4.0f * (1 + ((i % REG) & 7));
                nz += ha[i] != 0;
            }
            printf("mode 3 check: %zu wrong outputs, %zu accumulators not zero\n", bad, nz);
        } else
            CK(hipMemset(acc, 0, (size_t)64 * REG * 4));
    }
    return 0;
}
