// Does a cluster of workgroups that share an XCD keep a 512 KiB exchange buffer in that XCD's L2?
// One "transform" = 512 KiB (256 rows x 2 KiB).  Phase A: member m reads column tile m of the input (256 rows x 128 B), writes 32 KiB
// contiguous of `out`.  Phase B: member m reads column tile m of `out` (written by all 16 members), writes it back in place.
//   mode 0: two launches (A over everything, then B)            -- what the two-pass FFT does today
//   mode 1: one persistent launch, cluster barrier, compiler agent-scope fences (buffer_wbl2 sc1 / buffer_inv sc1)
//   mode 2: one persistent launch, cluster barrier, s_waitcnt vmcnt(0) before / buffer_inv sc1 after (no L2 write-back)
//   mode 4: as 2 without the buffer_inv (phase B reads addresses this CU never read: nothing stale in its L1);  5: buffer_inv sc0;
//   mode 6: no barrier at all (wrong results; the cost of everything but the barrier)
//   mode 3: as 2 with the members of a cluster spread over all XCDs (control: nothing to find in the local L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CL = 16, ROWS = 256, ROWB = 2048, TRB = ROWS * ROWB;

__device__ __forceinline__ void tile_load(const char *base, int m, float4 (&v)[8])
{
    const int t = threadIdx.x, seg = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = *(const float4 *)(base + (size_t)(r0 + 32 * i) * ROWB + m * 128 + seg * 16);
}
__device__ __forceinline__ void tile_store(char *base, int m, const float4 (&v)[8])
{
    const int t = threadIdx.x, seg = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 8; i++) *(float4 *)(base + (size_t)(r0 + 32 * i) * ROWB + m * 128 + seg * 16) = v[i];
}
__device__ __forceinline__ void phase_a(const char *in, char *out, int tr, int m)
{
    float4 v[8];
    tile_load(in + (size_t)tr * TRB, m, v);
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i].x += 1.0f; *(float4 *)(out + (size_t)tr * TRB + (size_t)m * 32768 + i * 4096 + threadIdx.x * 16) = v[i]; }
}
__device__ __forceinline__ void phase_b(char *out, int tr, int m)
{
    float4 v[8];
    tile_load(out + (size_t)tr * TRB, m, v);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i].x *= 2.0f;
    tile_store(out + (size_t)tr * TRB, m, v);
}
__global__ void __launch_bounds__(256) k_a(const char *in, char *out) { phase_a(in, out, blockIdx.x / CL, blockIdx.x % CL); }
__global__ void __launch_bounds__(256) k_b(char *out) { phase_b(out, blockIdx.x / CL, blockIdx.x % CL); }

template <int MODE>
__global__ void __launch_bounds__(256) k_fused(const char *in, char *out, unsigned *bar, unsigned *xcc_bad, int ntrans, int wpx)
{
    // wpx workgroups per XCD; cluster = CL workgroups with consecutive local numbers on one XCD (mode 3: consecutive blockIdx)
    const int b = blockIdx.x;
    int cluster, m;
    if (MODE == 3) { cluster = b / CL; m = b % CL; }
    else { const int xcd = b & 7, local = b >> 3; cluster = xcd * (wpx / CL) + local / CL; m = local % CL; }
    const int nclusters = gridDim.x / CL;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && (xcc & 0xf) != (unsigned)(b & 7)) atomicAdd(xcc_bad, 1u);
    unsigned *cnt = bar + cluster * 32;
    unsigned step = 0;
    for (int tr = cluster; tr < ntrans; tr += nclusters, step++) {
        phase_a(in, out, tr, m);
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && MODE != 6) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (step + 1) * CL) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else if (MODE == 2 || MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
        else if (MODE == 5) asm volatile("buffer_inv sc0" ::: "memory");
        phase_b(out, tr, m);
    }
}

int main(int argc, char **argv)
{
    const int ntrans = argc > 1 ? atoi(argv[1]) : 2048, wpc = argc > 2 ? atoi(argv[2]) : 2;
    char *in, *out;
    unsigned *bar, *bad;
    CK(hipMalloc(&in, (size_t)ntrans * TRB));
    CK(hipMalloc(&out, (size_t)ntrans * TRB));
    CK(hipMalloc(&bar, 4096 * 128));
    CK(hipMalloc(&bad, 4));
    std::vector<float> h((size_t)ntrans * TRB / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 1000);
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wpx = 32 * wpc, grid = 8 * wpx;
    std::vector<float> r(h.size());
    for (int mode = 0; mode < 7; mode++) {
        if (mode == 1) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(bar, 0, 4096 * 128)); CK(hipMemset(bad, 0, 4));
            CK(hipMemset(out, 0, (size_t)ntrans * TRB));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 0) { k_a<<<ntrans * CL, 256>>>(in, out); k_b<<<ntrans * CL, 256>>>(out); }
            else if (mode == 1) k_fused<1><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            else if (mode == 2) k_fused<2><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            else if (mode == 3) k_fused<3><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            else if (mode == 4) k_fused<4><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            else if (mode == 5) k_fused<5><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            else k_fused<6><<<grid, 256>>>(in, out, bar, bad, ntrans, wpx);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost));
        // expected: element (row, col) of out = 2 * (value phase A put there); phase A wrote in-tile order: check through a full recompute
        size_t wrong = 0;
        for (int tr = 0; tr < ntrans; tr += 97)
            for (int m = 0; m < CL; m++)
                for (int i = 0; i < 8; i++)
                    for (int t = 0; t < 256; t++) {
                        const int seg = t & 7, r0 = t >> 3;
                        const size_t src = ((size_t)tr * TRB + (size_t)(r0 + 32 * i) * ROWB + m * 128 + seg * 16) / 4;
                        const size_t dst = ((size_t)tr * TRB + (size_t)m * 32768 + i * 4096 + t * 16) / 4;
                        for (int c = 0; c < 4; c++) {
                            const float a = h[src + c] + (c == 0 ? 1.0f : 0.0f);
                            // phase B doubles .x of every float4 it moves (every 4th float of out)
                            const float e = (dst + c) % 4 == 0 ? 2.0f * a : a;
                            if (r[dst + c] != e) wrong++;
                        }
                    }
        unsigned nbad; CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
        const double gb = (double)ntrans * TRB / 1e9;
        printf("mode %d: %.1f us  (%.2f TB/s if 2 x %.2f GB moved, %.2f TB/s if 4 x)  wrong %zu  xcc!=blockIdx%%8: %u\n", mode, best * 1e3,
               2 * gb / best, gb, 4 * gb / best, wrong, nbad);
    }
    return 0;
}
