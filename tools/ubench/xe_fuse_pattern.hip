// Data path of a FUSED corner turn + correlator for more than 64 rows (DESIGN_EXPERIMENTS.md R5.7), measured before it is built.
// 256 antennas x 512 channels x 1024 frames of int8 I/Q: today k_xe_turn_lds writes 268 MB of operand tiles to HBM and k_xe_corr_sb reads
// them back (939 MB moved for 403 MB algorithmic, 197 us).  A correlator workgroup owns its CU (8 waves x 232 registers: ONE channel's
// accumulators), and a channel of a raw (t, station) row is 2 bytes, so the corner turn has to be shared: the 32 workgroups of an XCD turn a
// 64-byte column slice (32 channels) of a K block together and hand each other 32 KiB of tiles per channel and K block THROUGH THE XCD's L2.
// This program runs exactly that traffic and its synchronisation, with a stand-in for the matrix products, and checks every byte that
// crosses between workgroups (a checksum per channel against the host's), so a stale L1 / L2 line shows up as a wrong sum:
//   * teams are formed at run time from HW_REG_XCC_ID + a per-XCD ticket (HIP promises no placement; plain stores + L1-bypassing loads are
//     only coherent inside one XCD), 32 workgroups per XCD or the run is reported invalid;
//   * producer share of workgroup j of an XCD per K block (64 frames): row tile j / 2 (16 stations) x frames 32 (j % 2) .. +31 = 512 rows of
//     64 bytes; staged through LDS, byte-transposed (v_perm) into the MFMA operand order of k_xe_corr_sb, written as 2 x 512-byte runs per
//     channel into a ring of R K-block slots (1 MiB each) in global memory;
//   * flags: produced[xcd][seq] / consumed[xcd][seq] counters (relaxed agent-scope atomics, stores drained first), bounded polls;
//   * consumer: LDS-DMA of its channel's 32 KiB per K block, a checksum of the landed bytes, `mfma` matrix products per wave and K block
//     (k_xe_corr_sb issues 68 x 4 of v_mfma_i32_16x16x64_i8 per wave and K block at 256 rows), and 263 KiB of matrix stores per channel.
// usage: xe_fuse_pattern [store flavour 0 plain | 1 sc1] [load flavour 0 plain | 1 sc1 | 2 sc0 sc1] [mfma per wave and K block] [matrix stores 0|1] [look-ahead P] [ring R]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kN = 256, kF = 512, kT = 1024, kRow = kF * 2, kKB = 64, kNKB = kT / kKB;   // 16 K blocks of 64 frames
constexpr int kTeam = 32, kChanBytes = 2 * 16 * 1024;                                    // 32 workgroups per XCD; 32 KiB of tiles per channel and K block
constexpr int kSlotBytes = kTeam * kChanBytes;                                           // 1 MiB per XCD and K block
constexpr int kSeq = 2 * kNKB;                                                           // two passes of 256 channels

struct Args {
    const unsigned char *in;
    unsigned char *ring;      // [xcd][R][32 channels][32 KiB]
    unsigned *ticket;         // [8]
    unsigned *produced, *consumed;  // [8][kSeq]
    unsigned long long *sums; // [512 channels]: checksum of the bytes the consumer received
    v4i *out;                 // stand-in for the matrices: 263 KiB per channel
    int *status;              // != 0: a team was not 32 workgroups, or a poll ran out
    int st_flavour, ld_flavour, mfma, out_stores, P, R;
    int skip;  // 1: no producer traffic (flags only), 2: no consumer traffic (polls and flags only)
};

__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ void transpose4x4(unsigned i0, unsigned i1, unsigned i2, unsigned i3, unsigned (&o)[4])
{
    const unsigned t0 = perm(i1, i0, 0x05010400u), t1 = perm(i1, i0, 0x07030602u);
    const unsigned t2 = perm(i3, i2, 0x05010400u), t3 = perm(i3, i2, 0x07030602u);
    o[0] = perm(t2, t0, 0x05040100u); o[1] = perm(t2, t0, 0x07060302u); o[2] = perm(t3, t1, 0x05040100u); o[3] = perm(t3, t1, 0x07060302u);
}
__device__ __forceinline__ void dma16(const void *g, unsigned lds_dst, int fl)
{
    unsigned keep;
    if (fl == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
    else if (fl == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void st16(v4i *p, v4i v, int fl)
{
    if (fl == 0) *p = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(512, 2) void k_fuse(Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [0, 32 KiB): raw staging (stride 80 per row) ; [48 KiB, +32 KiB): the consumer's K block
    __shared__ int s_xcd, s_idx, s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        const int xcd = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;  // HW_REG_XCC_ID
        s_xcd = xcd;
        s_idx = (int)atomicAdd(a.ticket + xcd, 1u);
        s_bad = 0;
    }
    __syncthreads();
    const int xcd = s_xcd, idx = s_idx;
    if (idx >= kTeam) { if (tid == 0) atomicExch(a.status, 1); return; }  // (a ninth-XCD-worth of workgroups on this XCD: no team for it)
    unsigned char *ring = a.ring + (size_t)xcd * a.R * kSlotBytes;
    unsigned *produced = a.produced + xcd * kSeq, *consumed = a.consumed + xcd * kSeq;
    constexpr int RS = 80;  // LDS stride of a staged 64-byte row (bank spread for the dword gather)
    const unsigned lds0 = (unsigned)(size_t)lds;
    v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long sum = 0;

    auto poll = [&](unsigned *flag, unsigned want) {  // one lane; bounded
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 2000000) { atomicExch(a.status, 2); s_bad = 1; break; }
            }
        }
        __syncthreads();
    };
    auto produce = [&](int seq) {
        const int pass = seq / kNKB, kb = seq % kNKB;
        if (seq >= a.R) poll(consumed + seq - a.R, kTeam);  // the slot's previous K block has been read by every consumer of the team
        // ---- 512 rows x 64 bytes: row = tid: frame kb * 64 + 32 (idx % 2) + tid / 16, station 16 (idx / 2) + tid % 16
        if (!(a.skip & 1)) {
            // four lanes per row (one 64-byte request instead of four 16-byte ones from lanes 1 KiB apart: 370 -> see the header), four rows per thread
            v4i r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = (tid >> 2) + 128 * k, seg = tid & 3;
                const int t = kb * kKB + 32 * (idx & 1) + (row >> 4), st = 16 * (idx >> 1) + (row & 15);
                r[k] = __builtin_nontemporal_load((const v4i *)(a.in + ((size_t)t * kN + st) * kRow + (size_t)(pass * 8 + xcd) * 64) + seg);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) *(v4i *)(lds + ((tid >> 2) + 128 * k) * RS + (tid & 3) * 16) = r[k];
        }
        __syncthreads();
        // ---- item = (station r = tid % 16, frame group tg = (tid / 16) % 2, channel pair cp = tid / 32): 16 frames x one 4-byte unit
        if (!(a.skip & 1)) {
            const int r = tid & 15, tg = (tid >> 4) & 1, cp = tid >> 5;
            unsigned w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = *(const unsigned *)(lds + ((tg * 16 + i) * 16 + r) * RS + cp * 4);
            unsigned col[4][4];  // [byte of the unit: a.I a.Q b.I b.Q][frames 4 q .. 4 q + 3]
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned o[4];
                transpose4x4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3], o);
#pragma unroll
                for (int j = 0; j < 4; j++) col[j][q] = o[j];
            }
            const int g = 2 * (idx & 1) + tg;  // 16-frame group of the K block = lane / 16 of the operand tile
            unsigned char *slot = ring + (size_t)(seq % a.R) * kSlotBytes;
#pragma unroll
            for (int smp = 0; smp < 2; smp++)
#pragma unroll
                for (int plane = 0; plane < 2; plane++) {
                    v4i *dst = (v4i *)(slot + (size_t)(2 * cp + smp) * kChanBytes + plane * (16 * 1024) + (idx >> 1) * 1024 + (g * 16 + r) * 16);
                    st16(dst, (v4i){(int)col[2 * smp + plane][0], (int)col[2 * smp + plane][1], (int)col[2 * smp + plane][2], (int)col[2 * smp + plane][3]}, a.st_flavour);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's tile stores have left the CU (plain: they are in the XCD's L2)
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(produced + seq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto consume = [&](int seq) {
        poll(produced + seq, kTeam);
        const unsigned char *src = ring + (size_t)(seq % a.R) * kSlotBytes + (size_t)idx * kChanBytes + tid * 16;
        const unsigned dst = lds0 + 48 * 1024 + (unsigned)wave * 1024;
        if (!(a.skip & 2))
#pragma unroll
        for (int k = 0; k < 4; k++) dma16(src + k * 8192, __builtin_amdgcn_readfirstlane(dst + k * 8192), a.ld_flavour);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(consumed + seq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the slot may be rewritten
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v4i v = *(const v4i *)(lds + 48 * 1024 + k * 8192 + tid * 16);
#pragma unroll
            for (int e = 0; e < 4; e++) sum += __builtin_amdgcn_sad_u8((unsigned)v[e] ^ 0x80808080u, 0u, 0u);
            if (k == 0) { acc[0][0] ^= v[0]; }
        }
        // stand-in for the K block's matrix products (operands: whatever the registers hold)
        v4i A = {tid, seq, lane, wave}, B = {wave, lane, seq, tid};
        for (int m = 0; m < a.mfma; m++) acc[m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[m & 3], 0, 0, 0);
        __syncthreads();  // the K block in LDS is overwritten by the next one
    };
    for (int s = 0; s < a.P && s < kSeq; s++) produce(s);
    for (int seq = 0; seq < kSeq; seq++) {
        if (s_bad) break;
        if (seq + a.P < kSeq) produce(seq + a.P);
        consume(seq);
        if ((seq + 1) % kNKB == 0) {  // a channel is complete: checksum, and the matrix stores' traffic
            const int chan = (seq / kNKB) * 256 + xcd * 32 + idx;
            if (sum) atomicAdd(a.sums + chan, sum);
            sum = 0;
            if (a.out_stores) {
                v4i *o = a.out + (size_t)chan * (263 * 1024 / 16);
                for (int i = tid; i < 263 * 1024 / 16; i += 512) o[i] = acc[i & 3];
            }
        }
    }
    if (acc[0][0] == 0x12345678 && acc[3][2] == 0x7654321) a.out[tid] = acc[1];
}

int main(int argc, char **argv)
{
    Args a;
    a.st_flavour = argc > 1 ? atoi(argv[1]) : 0;
    a.ld_flavour = argc > 2 ? atoi(argv[2]) : 1;
    a.mfma = argc > 3 ? atoi(argv[3]) : 272;
    a.out_stores = argc > 4 ? atoi(argv[4]) : 1;
    a.P = argc > 5 ? atoi(argv[5]) : 2;
    a.R = argc > 6 ? atoi(argv[6]) : 3;
    a.skip = argc > 7 ? atoi(argv[7]) : 0;
    if (a.P < 1 || a.R < a.P + 1) { printf("need R >= P + 1\n"); return 1; }
    const size_t in_bytes = (size_t)kT * kN * kRow;
    const int NBUF = 3;  // inputs in rotation: every launch reads HBM, not the Infinity Cache
    std::vector<unsigned char> h(in_bytes);
    unsigned lcg = 99991u;
    for (auto &b : h) { lcg = lcg * 1664525u + 1013904223u; b = (unsigned char)(lcg >> 24); }
    // expected checksum per channel: sum over (t, station) of the biased I and Q bytes
    std::vector<unsigned long long> want(kF, 0);
    for (size_t row = 0; row < (size_t)kT * kN; row++)
        for (int c = 0; c < kF; c++) want[c] += (unsigned)(h[row * kRow + 2 * c] ^ 0x80u) + (unsigned)(h[row * kRow + 2 * c + 1] ^ 0x80u);
    unsigned char *d_in[NBUF];
    for (int k = 0; k < NBUF; k++) { CK(hipMalloc(&d_in[k], in_bytes)); CK(hipMemcpy(d_in[k], h.data(), in_bytes, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&a.ring, (size_t)8 * a.R * kSlotBytes));
    CK(hipMalloc(&a.ticket, 8 * 4)); CK(hipMalloc(&a.produced, 8 * kSeq * 4)); CK(hipMalloc(&a.consumed, 8 * kSeq * 4));
    CK(hipMalloc(&a.sums, kF * 8)); CK(hipMalloc(&a.status, 4));
    CK(hipMalloc(&a.out, (size_t)kF * 263 * 1024));
    const int lds_bytes = 48 * 1024 + 32 * 1024 + 50 * 1024;  // + padding: one workgroup per CU, as the correlator's 129 KiB would have it
    CK(hipFuncSetAttribute((const void *)k_fuse, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int k) {
        (void)hipMemsetAsync(a.ticket, 0, 32, 0); (void)hipMemsetAsync(a.produced, 0, 8 * kSeq * 4, 0); (void)hipMemsetAsync(a.consumed, 0, 8 * kSeq * 4, 0);
        a.in = d_in[k % NBUF];
        hipLaunchKernelGGL(k_fuse, dim3(256), dim3(512), lds_bytes, 0, a);
    };
    CK(hipMemset(a.status, 0, 4)); CK(hipMemset(a.sums, 0, kF * 8));
    launch(0);
    CK(hipDeviceSynchronize());
    int status = 0; CK(hipMemcpy(&status, a.status, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> got(kF);
    CK(hipMemcpy(got.data(), a.sums, kF * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int c = 0; c < kF; c++) bad += got[c] != want[c];
    if (a.skip) printf("(skip %d: checksums are meaningless) ", a.skip);
    printf("stores %s, loads %s, %d products per wave and K block, matrix stores %d, look-ahead %d, ring %d: status %d, %d of %d channel checksums wrong\n",
           a.st_flavour ? "sc1" : "plain", a.ld_flavour == 0 ? "plain" : a.ld_flavour == 1 ? "sc1" : "sc0 sc1", a.mfma, a.out_stores, a.P, a.R, status, bad, kF);
    if (status) return 0;
    const int it = 20;
    for (int k = 0; k < 3; k++) launch(k);
    // (the three memsets per launch are inside the timed region: ~3 us; the real kernel would carry an epoch instead)
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < it; k++) launch(k);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&status, a.status, 4, hipMemcpyDeviceToHost));
    printf("  %.1f us per integration (256 x 512 x 1024; today's two kernels: 197 us), status %d\n", ms * 1e3f / it, status);
    return 0;
}
