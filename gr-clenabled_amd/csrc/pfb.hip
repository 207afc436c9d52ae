// clPolyphaseChannelizer as gfx950 HIP kernels.
// Reference behaviour (lib/clPolyphaseChannelizer_impl.cc): general_work :83-109 =
// H2D, kernel filterpfb2 (:156-167), batched clFFT BACKWARD of size M (:208-225),
// kernel channel_map (:169-177), blocking D2H.  Semantics, K taps,
// M channels, R inputs per output step, buf[] history-prefixed:
//   a_j(i)   = sum_{k = j, j+M, .. < K} buf[i*R - k + K-1] * taps[k]
//   v_i[(j + i*(M-R)) % M] = a_j(i)
//   u_i[c]   = sum_m v_i[m] exp(+2 pi i m c / M)
//   out[i*nmap + q] = u_i[ch_map[q]]
//
// Fast path (M a power of two <= 256, critically sampled R == M, <= 64 taps per arm):
// ONE fused kernel.  A workgroup owns 4096/M consecutive output steps.  Phase 1: the
// input span is staged once in LDS; lane = arm, each lane slides a register window over
// its arm's decimated sequence and produces 16 consecutive steps (arm taps live in
// registers across the persistent loop).  Phase 2: the branch outputs go through LDS
// into the 16-points-per-thread layout of fft_core and the M-point backward DFT runs
// there; the channel map is applied on the store (identity map: straight from
// registers).  HBM traffic: 8 B read + 8 B*nmap/M written per input sample.
// Everything else (oversampled R != M, M not a power of two such as the reference
// flowgraph's M=3, very long arms) takes the generic two-kernel path.
#include <cmath>
#include <mutex>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "fft_core.hpp"
#include "fft_mr.h"

using namespace fftc;

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(c32 *p, c32 v) { f2v o; o.x = v.x; o.y = v.y; __builtin_nontemporal_store(o, (f2v *)p); }

// ------------------------------------------------------------------------------------
// fast path
// ------------------------------------------------------------------------------------
template <int M, int PMAX> struct PfbGeo {
    static constexpr int T = 4096 / M;
    static constexpr int SPAN = (T + PMAX) * M;  // staged input samples per group (+1 row: the window prefetch overshoots)
    static constexpr int LDS_SLOTS = SPAN > 4096 ? SPAN : 4096;
    static constexpr int PER_CU = (160 * 1024) / (LDS_SLOTS * 8);          // workgroups the LDS admits
    static constexpr int WPE = PER_CU >= 3 ? 3 : (PER_CU >= 2 ? 2 : 1);     // register target follows the LDS limit
};

template <int M, int PMAX, bool IDENT>
__global__ __launch_bounds__(256, (PfbGeo<M, PMAX>::WPE)) void k_pfb(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                const float *__restrict__ taps_pad,  // PMAX*M floats, zero padded
                                                const c32 *__restrict__ tw_inv, const int *__restrict__ ch_map, int nmap,
                                                int K, long long n_in, int nsteps, int ngroups)
{
    using PL = Plan<M, false>;
    constexpr int TH = 256, T = 4096 / M, NP = PL::NP;
    constexpr int SEG = TH / M;            // time segments handled side by side (lane = arm)
    constexpr int U = 16;                  // consecutive steps per thread: SEG * U == T
    static_assert(SEG * U == T, "geometry");
    constexpr int SPAN = PfbGeo<M, PMAX>::SPAN, LDS_SLOTS = PfbGeo<M, PMAX>::LDS_SLOTS;
    __shared__ c32 lds[LDS_SLOTS];
    const int tid0 = threadIdx.x;

    // arm taps, reversed so the window slides upward: hrev[pp] = taps[jb + (PMAX-1-pp)*M]
    const int jb0 = tid0 % M;
    float hrev[PMAX];
#pragma unroll
    for (int pp = 0; pp < PMAX; pp++) hrev[pp] = taps_pad[jb0 + (PMAX - 1 - pp) * M];
    TwRegs<M> tw;
    load_twiddles<M, false>(tw, tid0, tw_inv);

    // Each workgroup owns a CONTIGUOUS range of groups: consecutive groups overlap by PMAX rows, so
    // after the first group only the T new rows are fetched (the tail rows survive phase 2, which
    // only uses the first 4096 LDS slots, and are moved to the front).
    const int per_wg = (ngroups + gridDim.x - 1) / gridDim.x;
    const int g_begin = blockIdx.x * per_wg, g_end = (g_begin + per_wg < ngroups) ? g_begin + per_wg : ngroups;
    constexpr int TAIL = PMAX * M, NEW = T * M;          // slots kept / fetched per group
    constexpr int TAIL_PT = (TAIL + TH - 1) / TH, NEW_PT = NEW / TH, FULL_PT = (SPAN + TH - 1) / TH;
    for (int grp = g_begin; grp < g_end; grp++) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int jb = tid % M, sg = tid / M;
        // ---- stage input samples n_lo .. n_lo+SPAN of the history-prefixed buffer --------------
        const long long n_lo = (long long)grp * T * M + K - (long long)PMAX * M;
        if (grp == g_begin) {
            __syncthreads();
            c32 st[FULL_PT];
#pragma unroll
            for (int q = 0; q < FULL_PT; q++) {  // all loads in flight before the first LDS write
                const long long n = n_lo + tid + q * TH;
                const bool ok = n >= 0 && n < n_in && tid + q * TH < SPAN;
                const c32 x = in[ok ? n : 0];
                st[q] = ok ? x : mk(0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < FULL_PT; q++)
                if (tid + q * TH < SPAN) lds[tid + q * TH] = st[q];
        } else {
            // new rows first (their latency hides behind the tail move), then tail -> front
            const c32 *__restrict__ src = in + (n_lo + TAIL);
            const long long left64 = n_in - (n_lo + TAIL);
            const unsigned left = left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)(left64 > 0 ? left64 : 0);
            c32 st[NEW_PT], tl[TAIL_PT];
#pragma unroll
            for (int q = 0; q < NEW_PT; q++) {
                const unsigned e = (unsigned)(tid + q * TH);
                const bool ok = e < left;
                const c32 x = src[ok ? e : 0u];
                st[q] = ok ? x : mk(0.f, 0.f);
            }
            __syncthreads();  // previous group's phase 2 is done with LDS
#pragma unroll
            for (int q = 0; q < TAIL_PT; q++)
                if (tid + q * TH < TAIL) tl[q] = lds[NEW + tid + q * TH];
            __syncthreads();  // tail fully read before it is overwritten (source and destination may overlap)
#pragma unroll
            for (int q = 0; q < TAIL_PT; q++)
                if (tid + q * TH < TAIL) lds[tid + q * TH] = tl[q];
#pragma unroll
            for (int q = 0; q < NEW_PT; q++) lds[TAIL + tid + q * TH] = st[q];
        }
        __syncthreads();
        // ---- phase 1: acc[u] = sum_pp hrev[pp] * X[u + pp], X[w] = stage[(sg*16 + w)*M + (M-1-jb)] ----
        // Packed math: on gfx950 v_pk_fma_f32 issues two FMAs per lane in about the time of one
        // v_fma_f32 (measured 4.5 vs 4 cycles per wave instruction), so (re, im) pairs go through
        // explicit 2-vectors; this loop is the VALU-bound part of the kernel.
        f2v acc[U], win[U + 8];
        const int col = (sg * U) * M + (M - 1 - jb);
        const f2v *stage = (const f2v *)lds;
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc[u] = (f2v){0.f, 0.f};
            win[u] = stage[col + u * M];
        }
#pragma unroll
        for (int p0 = 0; p0 < PMAX; p0 += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) win[U + i] = stage[col + (p0 + U + i) * M];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float h = hrev[p0 + i];
                asm volatile("" : "+v"(h));  // keep the taps as 1 register each; no hoisted products
                const f2v hh = {h, h};
#pragma unroll
                for (int u = 0; u < U; u++) acc[u] = __builtin_elementwise_fma(win[i + u], hh, acc[u]);  // fma like the reference (:163)
            }
#pragma unroll
            for (int u = 0; u < U; u++) win[u] = win[u + 8];
        }
        __syncthreads();  // every wave is done with the staged input: reuse LDS for the transform
        // ---- phase 2: branch outputs -> transform layout (R == M: no rotation) ---------------
#pragma unroll
        for (int u = 0; u < U; u++) lds[swz(sg * U * M + jb) ^ swz(u * M)] = mk(acc[u].x, acc[u].y);  // linear swizzle, disjoint bits
        __syncthreads();
        c32 v[16];
        constexpr int R0 = PL::radix(0), B0 = M / R0;
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int g = tid + TH * q, raw_swz = swz((g / B0) * M + (g % B0));
#pragma unroll
            for (int r = 0; r < R0; r++) v[q * R0 + r] = lds[raw_swz ^ swz(r * B0)];
        }
        if constexpr (NP > 1) __syncthreads();  // pass 0 writes LDS in place
        transform_regs<M, 1, false>(v, tw, lds, tid);
        // ---- store: out[(i0+fr)*nmap + q] = u[ch_map[q]] ---------------------------------------
        constexpr int RL = PL::radix(NP - 1), BL = M / RL;
        const int i0 = grp * T;
        if constexpr (IDENT) {
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, fr = g / BL, j = g % BL;
                if (i0 + fr < nsteps) {
                    c32 *__restrict__ o = out + (size_t)(i0 + fr) * M + j;
#pragma unroll
                    for (int s = 0; s < RL; s++) st_stream(o + orev<RL>(s) * BL, v[q * RL + s]);
                }
            }
        } else {
            __syncthreads();  // transform's LDS reads are done
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, fr = g / BL, j = g % BL;
#pragma unroll
                for (int s = 0; s < RL; s++) lds[fr * M + j + orev<RL>(s) * BL] = v[q * RL + s];
            }
            __syncthreads();
            const int steps = (nsteps - i0) < T ? (nsteps - i0) : T;
            for (int e = tid; e < steps * nmap; e += TH) {
                const int fr = e / nmap, qq = e - fr * nmap;
                out[(size_t)i0 * nmap + e] = lds[fr * M + ch_map[qq]];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// M == 64 / 128 / 256: one thread per arm (M/64 waves per workgroup), no LDS staging; for M == 64 no workgroup barrier.
// The wave owns a contiguous run of output steps and keeps its arm windows (PMAX + 16 input
// rows, one sample per lane and row) in a register ring across iterations: every input row is
// loaded from HBM exactly once, 512 B per wave instruction, straight into the ring slot whose
// row has just been retired -- so the loads for the next 16 steps are in flight while the
// current 16 are computed.  Buffer loads with hardware range checking supply the zeros before
// and after the stream.  Branch outputs go through 8 KiB of wave-private LDS into the
// 16-points-per-thread layout and the 64-point backward DFT runs there (GeoW: all exchanges
// stay inside the wave).
// ------------------------------------------------------------------------------------
// times (-i)^q.  On the bit patterns (swap, sign-bit flips): as floating-point negations the compiler folds them into the butterflies that
// produce v, which changes where it contracts multiply-adds -- and the ring kernel must agree bit for bit with k_pfbq (one long call = several
// short ones)
__device__ __forceinline__ c32 quarter_turns(c32 v, int q)
{
    unsigned x = __builtin_bit_cast(unsigned, v.x), y = __builtin_bit_cast(unsigned, v.y);
    if (q & 1) {  // times -i: (y, -x)
        const unsigned t = x;
        x = y;
        y = t ^ 0x80000000u;
    }
    if (q & 2) {  // times -1
        x ^= 0x80000000u;
        y ^= 0x80000000u;
    }
    return mk(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
}

// one thread per arm: M threads (M/64 waves) per workgroup, 16 steps per iteration
template <int M> struct GeoArm {
    static constexpr int TH = M, PTS = M * 16, F = 16, WPE = 2;
};

template <int M, int PMAX, bool IDENT, int OS = 1>
__global__ __launch_bounds__(M, (M >= 512 ? 1 : 2)) void k_pfbw(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ taps_pad,
                                                 const c32 *__restrict__ tw_inv, const int *__restrict__ ch_map, int nmap, int K,
                                                 long long n_in, int nsteps, int groups_per_wave, int par)
{
    // OS-fold oversampling (R = M / OS new samples per step; OS = 1: the critically sampled channelizer, par = 0): step n = OS m + par of the
    // oversampled channelizer is step m of a critically sampled one whose input starts par R samples later (the launcher passes that
    // pointer), its branch outputs rotated by par R slots (lib/clPolyphaseChannelizer_impl.cc:164) -- a factor exp(-2 pi i par c / os) on
    // channel c, a multiple of a quarter turn for OS = 2 and 4.  This launch handles the steps of one par; it writes rows OS m + par.
    // (OS is a template parameter so that the critically sampled instantiation stays the code k_pfbq agrees with bit for bit.)
    constexpr int U = 16, RS = PMAX + U;                        // ring slots = rows resident per lane
    constexpr int PERIOD = RS / (RS % 16 == 0 ? 16 : 8);  // = RS / gcd(RS, U): iterations until the ring mapping repeats
    static_assert((U * PERIOD) % RS == 0, "ring period");
    using G = GeoArm<M>;
    using PL = Plan<M, false>;
    __shared__ c32 lds[G::PTS];
    const int lane0 = threadIdx.x;
    // two taps per 64-bit register pair; the packed FMA picks its half and broadcasts it to both components (op_sel), so the taps take
    // PMAX registers (the compiler's own form of the broadcast kept every tap twice and rebuilt pairs inside the loop)
    f2v hp[PMAX / 2];
#pragma unroll
    for (int pp = 0; pp < PMAX; pp += 2) hp[pp / 2] = f2v{taps_pad[lane0 + (PMAX - 1 - pp) * M], taps_pad[lane0 + (PMAX - 2 - pp) * M]};
    TwRegs<M> tw;
    load_twiddles<M, false, G>(tw, lane0, tw_inv);

    const int ngroups = (nsteps + U - 1) / U;
    const int g_begin = blockIdx.x * groups_per_wave;
    int g_end = g_begin + groups_per_wave;
    if (g_end > ngroups) g_end = ngroups;
    if (g_begin >= g_end) return;

    // raw buffer over the readable input; offsets are 32-bit (the launcher guarantees n_in * 8 < 4 GiB - 64 KiB),
    // and a row before the start of the stream wraps to an out-of-range offset, i.e. reads as zero
    // (the buffer ends with the last row THIS wave's outputs use: the refill of the last iteration -- 16 rows past the wave's range, which the
    // next wave reads anyway -- is then out of range, i.e. no memory traffic: 6 % of the input bytes at 16 groups per wave)
    long long n_lim = (long long)g_begin * U * M + K - (long long)PMAX * M + ((long long)(g_end - g_begin) * U + PMAX - 1) * M;
    if (n_lim > n_in) n_lim = n_in;
    if (n_lim < 0) n_lim = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(unsigned)(n_lim * 8), 0x00020000);
    const unsigned lane_off = (unsigned)((M - 1 - lane0) * 8);
    // the output as a range-checked buffer (identity map, critically sampled: rows 0 .. nsteps-1 of M channels; the launcher guarantees < 4 GiB)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(unsigned)((size_t)nsteps * M * 8), 0x00020000);
    // sample index of ring row 0 of the first iteration: X[w] = in[n0 + w*M + (M-1-lane)]
    const long long n0 = (long long)g_begin * U * M + K - (long long)PMAX * M;
    auto load_row = [&](long long row) -> f2v {  // row = absolute row index relative to n0
        const unsigned off = (unsigned)((n0 + row * M) * 8) + lane_off;
        return __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    };
    f2v ring[RS];
#pragma unroll
    for (int w = 0; w < RS; w++) ring[w] = load_row(w);
    // (the compiler's wait-count pass merges this path -- RS loads in flight -- with the loop's back edge -- 16 loads + 16 stores -- into a
    // full drain, vmcnt(0), at the second product of every first-phase iteration; with nothing in flight on entry it keeps the exact counts)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    auto iteration = [&](auto phase_tag, int grp) {
        constexpr int PH = decltype(phase_tag)::value;
        const long long row0 = (long long)(grp - g_begin) * U;  // ring row 0 of this iteration
        int lane = lane0;
        asm volatile("" : "+v"(lane));  // LDS addresses are recomputed per iteration instead of living in ~50 registers
        const int lane_swz8 = swzn<M>(lane) * 8;  // byte offsets: one v_xor per access (see fftc::lds_slot)
        // ---- phase 1: two steps at a time; their rows are retired and refilled right after ----
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            f2v a0, a1;  // fma like the reference (:163), taps in ascending order per output, starting from +0
            asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(a0) : "v"(ring[(U * PH + u) % RS]), "v"(hp[0]));
            asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(a1) : "v"(ring[(U * PH + u + 1) % RS]), "v"(hp[0]));
#pragma unroll
            for (int pp = 1; pp < PMAX; pp++) {
                if (pp & 1) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a0) : "v"(ring[(U * PH + u + pp) % RS]), "v"(hp[pp / 2]));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a1) : "v"(ring[(U * PH + u + 1 + pp) % RS]), "v"(hp[pp / 2]));
                } else {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(ring[(U * PH + u + pp) % RS]), "v"(hp[pp / 2]));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1) : "v"(ring[(U * PH + u + 1 + pp) % RS]), "v"(hp[pp / 2]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // the refill below must not be hoisted above the last use of its slot
            *(c32 *)((char *)lds + (lane_swz8 ^ (swzn<M>(u * M) * 8))) = mk(a0.x, a0.y);  // swizzles are XOR-linear: swz(u*M + lane) = swz(lane) ^ swz(u*M)
            *(c32 *)((char *)lds + (lane_swz8 ^ (swzn<M>((u + 1) * M) * 8))) = mk(a1.x, a1.y);
            // unconditional: past the wave's range the rows are simply not used, past the stream they read as zero
            ring[(U * PH + u) % RS] = load_row(row0 + RS + u);
            ring[(U * PH + u + 1) % RS] = load_row(row0 + RS + u + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // orders the LDS writes before the reads below (a plain waitcnt when M = 64: single wave)
        // ---- phase 2: M-point backward DFT of the 16 steps ----
        c32 v[16];
        constexpr int R0 = PL::radix(0), B0 = M / R0;
        {
            const int raw_swz8 = swzn<M>((lane / B0) * M + (lane % B0)) * 8;
#pragma unroll
            for (int r = 0; r < R0; r++) v[r] = *(const c32 *)((const char *)lds + (raw_swz8 ^ (swzn<M>(r * B0) * 8)));
        }
        __syncthreads();
        transform_regs<M, 1, false, G>(v, tw, lds, lane);
        constexpr int NP = PL::NP, RL = PL::radix(NP - 1), BL = M / RL;
        const int i0 = grp * U;
        if constexpr (IDENT && OS == 1) {
            // Buffer stores with hardware range checking: the steps past the end of the call are dropped by the address unit, so the sixteen
            // stores are unconditional and straight-line.  With the stores under `if (step < nsteps)` the compiler could not count them (0 .. 16
            // were in flight at the next iteration's first use of a freshly loaded row) and waited for vmcnt(15): every row load of the previous
            // iteration AND, one by one, the previous iteration's stores -- a wave stood ~1.5 us per iteration waiting for store
            // acknowledgements.  Counted, the wait is vmcnt(30 + ...): only the load that is needed.
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = lane + M * q, fr = g / BL, j = g % BL;
                const unsigned o8 = (unsigned)(((size_t)(i0 + fr) * M + j) * 8);
#pragma unroll
                for (int t = 0; t < RL; t++) {
                    const c32 z = v[q * RL + t];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f2v, z), orsrc, o8 + (unsigned)(orev<RL>(t) * BL * 8), 0, 2);  // (2: nontemporal)
                }
            }
            __syncthreads();
        } else if constexpr (IDENT) {
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = lane + M * q, fr = g / BL, j = g % BL;
                if (i0 + fr < nsteps) {
                    c32 *__restrict__ o = out + (OS == 1 ? (size_t)(i0 + fr) : (size_t)(i0 + fr) * OS + par) * M + j;
#pragma unroll
                    for (int t = 0; t < RL; t++) {
                        const int c = j + orev<RL>(t) * BL;  // channel
                        if constexpr (OS == 1) st_stream(o + orev<RL>(t) * BL, v[q * RL + t]);
                        else st_stream(o + orev<RL>(t) * BL, quarter_turns(v[q * RL + t], par * c * (4 / OS)));
                    }
                }
            }
            __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = lane + M * q, fr = g / BL, j = g % BL;
#pragma unroll
                for (int t = 0; t < RL; t++) lds[fr * M + j + orev<RL>(t) * BL] = v[q * RL + t];
            }
            __syncthreads();
            const int steps = (nsteps - i0) < U ? (nsteps - i0) : U;
            for (int e = lane; e < steps * nmap; e += M) {
                const int fr = e / nmap, qq = e - fr * nmap, c = ch_map[qq];
                const c32 z = lds[fr * M + c];
                if constexpr (OS == 1) out[(size_t)i0 * nmap + e] = z;
                else out[((size_t)(i0 + fr) * OS + par) * nmap + qq] = quarter_turns(z, par * c * (4 / OS));
            }
            __syncthreads();
        }
    };
    for (int grp = g_begin; grp < g_end; grp += PERIOD) {
        iteration(std::integral_constant<int, 0>{}, grp);
        if constexpr (PERIOD > 1) { if (grp + 1 < g_end) iteration(std::integral_constant<int, 1 % PERIOD>{}, grp + 1); }
        if constexpr (PERIOD > 2) { if (grp + 2 < g_end) iteration(std::integral_constant<int, 2 % PERIOD>{}, grp + 2); }
    }
}

// ------------------------------------------------------------------------------------
// Small calls (the reference's own call size: buf_items = 65536 -> 1024 steps = 64 groups of 16): the ring kernel above would run
// 64 waves on a 1024-SIMD device, each a serial chain of row loads -> 512 packed FMAs -> DFT -> stores.  Here a workgroup owns ONE
// group of 16 steps and four sets of M threads each form 4 of its steps (PMAX + 3 rows per lane instead of PMAX + 16, re-read from
// L2 by the neighbours: the whole call is 1 MiB); the first set then runs the DFT of the 16 steps and stores.  Same FMA order per
// output and the same DFT code as k_pfbw, so one long call and several short ones agree bit for bit.
// ------------------------------------------------------------------------------------
template <int M, int PMAX, bool IDENT>
__global__ __launch_bounds__(4 * M) void k_pfbq(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ taps_pad,
                                                 const c32 *__restrict__ tw_inv, const int *__restrict__ ch_map, int nmap, int K,
                                                 long long n_in, int nsteps)
{
    constexpr int U = 16, UQ = 4, RQ = PMAX + UQ - 1;
    using G = GeoArm<M>;
    using PL = Plan<M, false>;
    __shared__ c32 lds[G::PTS];
    const int lane = threadIdx.x % M, quarter = threadIdx.x / M, grp = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(unsigned)(n_in * 8), 0x00020000);
    const unsigned lane_off = (unsigned)((M - 1 - lane) * 8);
    const long long n0 = ((long long)grp * U + quarter * UQ) * M + K - (long long)PMAX * M;  // row 0 of this quarter's window
    f2v row[RQ];
#pragma unroll
    for (int w = 0; w < RQ; w++)
        row[w] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (unsigned)((n0 + (long long)w * M) * 8) + lane_off, 0, 0));
    float hrev[PMAX];
#pragma unroll
    for (int pp = 0; pp < PMAX; pp++) hrev[pp] = taps_pad[lane + (PMAX - 1 - pp) * M];
    TwRegs<M> tw;
    if (quarter == 0) load_twiddles<M, false, G>(tw, lane, tw_inv);
    f2v acc[UQ];
#pragma unroll
    for (int u = 0; u < UQ; u++) acc[u] = (f2v){0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < PMAX; pp++) {
        const f2v hh = {hrev[pp], hrev[pp]};
#pragma unroll
        for (int u = 0; u < UQ; u++) acc[u] = __builtin_elementwise_fma(row[u + pp], hh, acc[u]);  // per output: pp ascending, like k_pfbw
    }
    const int lane_swz = swzn<M>(lane);
#pragma unroll
    for (int u = 0; u < UQ; u++) lds[swzn<M>(lane + (quarter * UQ + u) * M)] = mk(acc[u].x, acc[u].y);
    (void)lane_swz;
    __syncthreads();
    if (quarter != 0) return;  // (a finished wave no longer counts at the barriers below)
    c32 v[16];
    constexpr int R0 = PL::radix(0), B0 = M / R0;
    {
        const int raw_swz = swzn<M>((lane / B0) * M + (lane % B0));
#pragma unroll
        for (int r = 0; r < R0; r++) v[r] = lds[raw_swz ^ swzn<M>(r * B0)];
    }
    __syncthreads();
    transform_regs<M, 1, false, G>(v, tw, lds, lane);
    constexpr int NP = PL::NP, RL = PL::radix(NP - 1), BL = M / RL;
    const int i0 = grp * U;
    if constexpr (IDENT) {
#pragma unroll
        for (int q = 0; q < 16 / RL; q++) {
            const int g = lane + M * q, fr = g / BL, j = g % BL;
            if (i0 + fr < nsteps) {
                c32 *__restrict__ o = out + (size_t)(i0 + fr) * M + j;
#pragma unroll
                for (int t = 0; t < RL; t++) st_stream(o + orev<RL>(t) * BL, v[q * RL + t]);
            }
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16 / RL; q++) {
            const int g = lane + M * q, fr = g / BL, j = g % BL;
#pragma unroll
            for (int t = 0; t < RL; t++) lds[fr * M + j + orev<RL>(t) * BL] = v[q * RL + t];
        }
        __syncthreads();
        const int steps = (nsteps - i0) < U ? (nsteps - i0) : U;
        for (int e = lane; e < steps * nmap; e += M) {
            const int fr = e / nmap, qq = e - fr * nmap;
            out[(size_t)i0 * nmap + e] = lds[fr * M + ch_map[qq]];
        }
    }
}

// ------------------------------------------------------------------------------------
// M == 32 (written for 8 / 16 / 32): the same ring kernel with 64/M independent time streams per wave (lane = (stream, arm)); every
// stream owns a contiguous run of steps, so each lane still advances 16 rows per iteration and loads every row once.
// One iteration transforms 16 steps of every stream (1024/M frames of M points, single-wave geometry).
// ------------------------------------------------------------------------------------
template <int M, int PMAX, bool IDENT>
__global__ __launch_bounds__(64, 2) void k_pfbs(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ taps_pad,
                                                const c32 *__restrict__ tw_inv, const int *__restrict__ ch_map, int nmap, int K,
                                                long long n_in, int nsteps, int groups_per_stream)
{
    constexpr int U = 16, RS = PMAX + U, SEG = 64 / M;
    constexpr int PERIOD = RS / (RS % 16 == 0 ? 16 : 8);
    static_assert((U * PERIOD) % RS == 0 && PERIOD <= 3, "ring period");
    using G = GeoW<M>;
    using PL = Plan<M, false>;
    constexpr int NP = PL::NP;
    __shared__ c32 lds[G::PTS];
    const int lane0 = threadIdx.x, arm0 = lane0 % M, sg0 = lane0 / M;
    float hrev[PMAX];
#pragma unroll
    for (int pp = 0; pp < PMAX; pp++) hrev[pp] = taps_pad[arm0 + (PMAX - 1 - pp) * M];
    TwRegs<M> tw;
    load_twiddles<M, false, G>(tw, lane0, tw_inv);

    const int ngroups = (nsteps + U - 1) / U;
    const int stream0 = blockIdx.x * SEG;                       // first stream of this wave
    const int g_begin = (stream0 + sg0) * groups_per_stream;    // this lane's stream
    if (stream0 * groups_per_stream >= ngroups) return;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(unsigned)(n_in * 8), 0x00020000);
    const unsigned lane_off = (unsigned)((M - 1 - arm0) * 8);
    const long long n0 = (long long)g_begin * U * M + K - (long long)PMAX * M;
    auto load_row = [&](long long row) -> f2v {
        const long long n = n0 + row * M;
        // streams past the end of the data would wrap the 32-bit offset back into range: clamp them out explicitly
        const unsigned off = (n * 8 >= (4ll << 30)) ? 0xfffffff8u : (unsigned)(n * 8) + lane_off;
        return __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    };
    f2v ring[RS];
#pragma unroll
    for (int w = 0; w < RS; w++) ring[w] = load_row(w);

    auto iteration = [&](auto phase_tag, int it) {
        constexpr int PH = decltype(phase_tag)::value;
        const long long row0 = (long long)it * U;
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int arm = lane % M, sg = lane / M;
        const int arm_swz = swzn<M>(sg * U * M + arm);
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            f2v a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
            for (int pp = 0; pp < PMAX; pp++) {
                const f2v hh = {hrev[pp], hrev[pp]};
                a0 = __builtin_elementwise_fma(ring[(U * PH + u + pp) % RS], hh, a0);
                a1 = __builtin_elementwise_fma(ring[(U * PH + u + 1 + pp) % RS], hh, a1);
            }
            __builtin_amdgcn_sched_barrier(0);
            lds[arm_swz ^ swzn<M>(u * M)] = mk(a0.x, a0.y);  // swz(sg*U*M + u*M + arm) = swz(sg*U*M + arm) ^ swz(u*M): disjoint bits, linear swizzle
            lds[arm_swz ^ swzn<M>((u + 1) * M)] = mk(a1.x, a1.y);
            ring[(U * PH + u) % RS] = load_row(row0 + RS + u);
            ring[(U * PH + u + 1) % RS] = load_row(row0 + RS + u + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        c32 v[16];
        constexpr int R0 = PL::radix(0), B0 = M / R0;
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int g = lane + 64 * q, raw_swz = swzn<M>((g / B0) * M + (g % B0));
#pragma unroll
            for (int r = 0; r < R0; r++) v[q * R0 + r] = lds[raw_swz ^ swzn<M>(r * B0)];
        }
        __syncthreads();
        transform_regs<M, 1, false, G>(v, tw, lds, lane);
        constexpr int RL = PL::radix(NP - 1), BL = M / RL;
        // frame fr of the iteration = step fr%16 of stream fr/16
        auto first_step = [&](int fr, bool &ok) {
            const int gb = (stream0 + fr / U) * groups_per_stream, grp = gb + it;
            ok = it < groups_per_stream && grp < ngroups;
            return grp * U + (fr % U);
        };
        if constexpr (IDENT) {
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = lane + 64 * q, fr = g / BL, j = g % BL;
                bool ok;
                const int step = first_step(fr, ok);
                if (ok && step < nsteps) {
                    c32 *__restrict__ o = out + (size_t)step * M + j;
#pragma unroll
                    for (int t = 0; t < RL; t++) st_stream(o + orev<RL>(t) * BL, v[q * RL + t]);
                }
            }
            __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = lane + 64 * q, fr = g / BL, j = g % BL;
#pragma unroll
                for (int t = 0; t < RL; t++) lds[fr * M + j + orev<RL>(t) * BL] = v[q * RL + t];
            }
            __syncthreads();
            for (int e = lane; e < SEG * U * nmap; e += 64) {
                const int fr = e / nmap, qq = e - fr * nmap;
                bool ok;
                const int step = first_step(fr, ok);
                if (ok && step < nsteps) out[(size_t)step * nmap + qq] = lds[fr * M + ch_map[qq]];
            }
            __syncthreads();
        }
    };
    for (int it = 0; it < groups_per_stream; it += PERIOD) {
        iteration(std::integral_constant<int, 0>{}, it);
        if constexpr (PERIOD > 1) { if (it + 1 < groups_per_stream) iteration(std::integral_constant<int, 1 % PERIOD>{}, it + 1); }
        if constexpr (PERIOD > 2) { if (it + 2 < groups_per_stream) iteration(std::integral_constant<int, 2 % PERIOD>{}, it + 2); }
    }
}

// ------------------------------------------------------------------------------------
// generic path: one thread per branch output, then a direct M-point DFT per mapped channel
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pfb_branches(const c32 *__restrict__ in, c32 *__restrict__ filt,
                                                      const float *__restrict__ taps, int K, int M, int R, long long total)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / M;
        const int j = (int)(e - i * M);
        float sr = 0.f, si = 0.f;
        // (eight taps' loads requested before the first is used: the trip count is a run-time value and the compiler does not pipeline the
        // loop itself; the sums keep their order)
        int k = j;
        const c32 *__restrict__ xp = in + (i * R + K - 1);
        for (; k + 7 * M < K; k += 8 * M) {
            c32 x[8];
            float h[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { x[u] = xp[-(k + u * M)]; h[u] = taps[k + u * M]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { sr = fmaf(x[u].x, h[u], sr); si = fmaf(x[u].y, h[u], si); }
        }
        for (; k < K; k += M) {
            const c32 x = xp[-k];
            sr = fmaf(x.x, taps[k], sr);
            si = fmaf(x.y, taps[k], si);
        }
        const int slot = (int)((j + i * (long long)(M - R)) % M);
        filt[i * M + slot] = mk(sr, si);
    }
}

// The same branch filters for R = M (every step consumes one new sample per arm) and for 2- / 4-fold oversampling (R = M / 2, M / 4): y_j[i] = sum_p h[j + M p] x_j[i - p] with
// x_j[n] = in[n M - j + K - 1] is an ordinary FIR per arm.  A thread takes one arm and T consecutive steps and walks the taps in chunks
// of PC: PC taps and a window of T + PC - 1 samples in registers feed T x PC multiply-adds -- (2 PC + T - 1) / (T PC) = 0.3 loads per
// multiply-add instead of 2 (the per-output kernel above ran 32 taps per arm at 40 GS/s whatever the channel count).  Lanes run along
// the arms: every load is a run of consecutive samples.  Same operation order per output (fma, taps ascending).
template <int T, int PC, int S>
__global__ __launch_bounds__(256) void k_pfb_branches_t(const c32 *__restrict__ in, c32 *__restrict__ filt, const float *__restrict__ taps, int K,
                                                        int M, int nsteps, long long total /* blocks of T steps x M */, int xcd_runs)
{
    const int R = M / S;  // S-fold oversampling (R = M / S new samples per step): x_j[n] = in[n R - j + K - 1], y_j[i] = sum_p h[j + M p] x_j[i - S p]
    const int P = (K + M - 1) / M;
    // A thread's window reaches P - 1 rows back: the workgroups a few places earlier in time read the same rows.  Workgroups go to the XCDs round
    // robin, so with the plain order every XCD's L2 fetches most of the input for itself; xcd_runs gives each XCD one contiguous eighth of the time
    // axis instead (workgroup b -> place (b mod 8) * per + b / 8; the grid is a multiple of 8, the stride keeps a workgroup on its eighth).
    const long long nblk = (total + 255) / 256, per = (nblk + 7) / 8;
    for (long long b = blockIdx.x; b < (xcd_runs ? per * 8 : nblk); b += gridDim.x) {
        const long long lb = xcd_runs ? (b & 7) * per + (b >> 3) : b, e = lb * 256 + threadIdx.x;
        if (lb >= nblk || e >= total) continue;
        const long long tb = e / M;
        const int j = (int)(e - tb * M);
        const long long i0 = tb * T;
        float ar[T], ai[T];
#pragma unroll
        for (int t = 0; t < T; t++) ar[t] = ai[t] = 0.f;
        for (int c0 = 0; c0 < P; c0 += PC) {
            float h[PC];
#pragma unroll
            for (int q = 0; q < PC; q++) {
                const long long k = (long long)j + (long long)M * (c0 + q);
                h[q] = k < K ? taps[k] : 0.f;
            }
            constexpr int W = T + S * (PC - 1);
            const long long rb = i0 - (long long)S * (c0 + PC - 1);  // first row of the window
            c32 xw[W];
#pragma unroll
            for (int u = 0; u < W; u++) {
                const long long row = rb + u, idx = row * R - j + K - 1;
                xw[u] = (idx >= 0 && row < nsteps) ? in[idx] : mk(0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < PC; q++)  // taps ascending, as the reference kernel accumulates (lib/clPolyphaseChannelizer_impl.cc:156-167)
#pragma unroll
                for (int t = 0; t < T; t++) {
                    ar[t] = fmaf(xw[t + S * (PC - 1 - q)].x, h[q], ar[t]);
                    ai[t] = fmaf(xw[t + S * (PC - 1 - q)].y, h[q], ai[t]);
                }
        }
#pragma unroll
        for (int t = 0; t < T; t++)
            if (i0 + t < nsteps) {
                const long long i = i0 + t;
                const int slot = S == 1 ? j : (int)((j + i * (long long)(M - R)) % M);  // (:164: the branch outputs rotate with the step when R != M)
                filt[i * M + slot] = mk(ar[t], ai[t]);
            }
    }
}

// The branch filters of a critically sampled channelizer as a streaming kernel (the first of the two kernels wherever neither a ring kernel nor the
// one-kernel form k_pfb_mr takes the shape: more than 512 channels, more than 32 taps per arm, ...).  Thread (range q, arm j): the arm's last PMAX - 1
// samples stay in registers, ONE load per step (lanes along the arms: a row of consecutive samples), eight steps of packed fmas, taps ascending from +0
// (lib/clPolyphaseChannelizer_impl.cc:156-167; the operation order of k_pfb_branches_t, bit for bit), eight stores of consecutive lanes.  No barrier in
// the loop; the taps of the workgroup's arms sit in LDS.  k_pfb_branches_t spends six loads, their 64-bit addresses and 32 tap loads per output: 245 us
// per 2^25 outputs at 32 taps per arm against this kernel's time in DESIGN_EXPERIMENTS R6.12.
struct FirRing {
    const c32 *in;
    c32 *filt;
    const float *taps;
    int K, M, A, Q, nsteps, arm_blocks;  // A arms per workgroup (M, or 256 when M > 256), Q = 256 / A time ranges per workgroup
    unsigned m_A;
    long long range_len;
};

template <int PH, int PERIOD, class F>
__device__ __forceinline__ void fir_phases(F &&body, int it, int iters)
{
    if constexpr (PH < PERIOD) {
        if (it + PH < iters) {
            body(std::integral_constant<int, PH>{}, it + PH);
            fir_phases<PH + 1, PERIOD>(body, it, iters);
        }
    }
}

template <int PMAX>
__global__ __launch_bounds__(256) void k_pfb_fir(const FirRing f)
{
    constexpr int FS = 8;
    // the window (PMAX - 1 + FS rows) lives in a ring of RS = PMAX + FS registers; in phase PH row u of the window is slot (PH FS + u) mod RS, the next
    // iteration's rows take the slots of the FS rows this iteration retires (and the spare): no register of the window ever moves.  The loop is
    // unrolled over the RS / FS phases (moving the window instead cost twice its registers: 214 at 32 taps per arm)
    constexpr int RS = PMAX + FS, PERIOD = RS / FS;
    static_assert(RS % FS == 0, "ring period");
    extern __shared__ float fir_taps[];  // [PMAX][A]
    const int t = threadIdx.x;
    const int q = (int)__umulhi((unsigned)t, f.m_A), j0 = t - q * f.A;
    const int ab = blockIdx.x % f.arm_blocks;
    const long long rb = blockIdx.x / f.arm_blocks;
    const int j = ab * 256 + j0;
    const bool arm = q < f.Q && j < f.M;
    for (int i = t; i < PMAX * f.A; i += 256) {
        const int p = i / f.A, jj = ab * 256 + (i - p * f.A);
        const long long k = (long long)jj + (long long)f.M * p;
        fir_taps[i] = (jj < f.M && k < f.K) ? f.taps[k] : 0.f;
    }
    __syncthreads();
    const float *th = fir_taps + (arm ? j0 : 0);
    const long long s0 = (rb * f.Q + q) * f.range_len;
    const f2v *xp = (const f2v *)f.in + (f.K - 1 - j);  // x_j[r] = in[r M - j + K - 1]
    auto ld = [&](long long r) {
        const long long o = r * f.M;
        f2v x = {0.f, 0.f};
        if (arm && o + (f.K - 1 - j) >= 0 && r < f.nsteps) x = __builtin_nontemporal_load(xp + o);
        return x;
    };
    f2v ring[RS], nx[FS];
#pragma unroll
    for (int u = 0; u < PMAX - 1 + FS; u++) ring[u] = ld(s0 - (PMAX - 1) + u);
    ring[RS - 1] = f2v{0.f, 0.f};
#pragma unroll
    for (int s = 0; s < FS; s++) nx[s] = f2v{0.f, 0.f};
    const int iters = (int)(f.range_len / FS);
    auto iteration = [&](auto phase_tag, int it) {
        constexpr int PH = decltype(phase_tag)::value;
        const long long cur = s0 + (long long)it * FS;
        if (it + 1 < iters) {  // the next iteration's rows: requested in front of the arithmetic
#pragma unroll
            for (int s = 0; s < FS; s++) nx[s] = ld(cur + FS + s);
        }
        f2v acc[FS];
#pragma unroll
        for (int s = 0; s < FS; s++) acc[s] = f2v{0.f, 0.f};
        const float *thv = th;
        asm volatile("" : "+v"(thv));  // the taps are read from LDS every iteration, not kept in 2 x PMAX registers
#pragma unroll
        for (int p0 = 0; p0 < PMAX; p0 += 8) {  // per output: taps ascending from +0; eight taps' reads at a time
#pragma unroll
            for (int p = p0; p < p0 + 8; p++) {
                const float hp = thv[p * f.A];
                const f2v hh = {hp, hp};
#pragma unroll
                for (int s = 0; s < FS; s++) acc[s] = __builtin_elementwise_fma(ring[(PH * FS + PMAX - 1 + s - p) % RS], hh, acc[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (arm) {
#pragma unroll
            for (int s = 0; s < FS; s++)
                if (cur + s < f.nsteps) __builtin_nontemporal_store(acc[s], (f2v *)f.filt + (cur + s) * f.M + j);
        }
        // rows 0 .. FS-1 of the window retire; the new rows are rows PMAX-1 .. PMAX-2+FS of the next phase = the spare slot and the slots of rows 0 .. FS-2
#pragma unroll
        for (int s = 0; s < FS; s++) ring[(PH * FS + RS - 1 + s) % RS] = nx[s];
    };
    for (int it = 0; it < iters; it += PERIOD) fir_phases<0, PERIOD>(iteration, it, iters);
}

__global__ __launch_bounds__(256) void k_pfb_dft_map(const c32 *__restrict__ filt, c32 *__restrict__ out,
                                                     const c32 *__restrict__ twM,  // exp(+2 pi i t / M), t < M
                                                     const int *__restrict__ ch_map, int nmap, int M, long long total)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / nmap;
        const int c = ch_map[(int)(e - i * nmap)];
        const c32 *v = filt + i * M;
        float sr = 0.f, si = 0.f;
        int t = 0, m = 0;
        for (; m + 8 <= M; m += 8) {  // (loads of eight terms ahead of their use, as above)
            c32 w[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                w[u] = twM[t];
                x[u] = v[m + u];
                t += c;
                if (t >= M) t -= M;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                sr += x[u].x * w[u].x - x[u].y * w[u].y;
                si += x[u].x * w[u].y + x[u].y * w[u].x;
            }
        }
        for (; m < M; m++) {
            const c32 w = twM[t], x = v[m];
            sr += x.x * w.x - x.y * w.y;
            si += x.x * w.y + x.y * w.x;
            t += c;
            if (t >= M) t -= M;
        }
        out[e] = mk(sr, si);
    }
}

// channel_map after the transform (lib/clPolyphaseChannelizer_impl.cc:169-177): out[i nmap + q] = u_i[ch_map[q]]
__global__ __launch_bounds__(256) void k_pfb_map(const c32 *__restrict__ u, c32 *__restrict__ out, const int *__restrict__ ch_map, int nmap, int M,
                                                 long long total)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / nmap;
        out[e] = u[i * M + ch_map[(int)(e - i * nmap)]];
    }
}

}  // namespace

struct mi355_pfb {
    mi355_ctx *ctx;
    int K, M, R, buf_items, nmap, nsteps;
    bool fast, ident;
    bool fast_over = false;  // 64 / 128 / 256 channels, <= 32 taps per arm, 2- or 4-fold oversampled: the ring kernel once per residue of the step number
    int pmax;
    float *d_taps = nullptr;      // K floats (generic) or pmax*M zero padded (fast)
    void *d_tw = nullptr;         // M complex, exp(+2 pi i t / M)
    int *d_map = nullptr;
    void *d_filt = nullptr;       // generic path scratch: nsteps*M complex
    // generic path, 8 channels and more with 16 and more of them mapped: the M-point backward DFT of every step as a clFFT transform of
    // this library (the reference calls clFFT for it, lib/clPolyphaseChannelizer_impl.cc:100,208-225) instead of M products per output
    mi355_fft *dft = nullptr;
    void *d_filt2 = nullptr;      // the transforms' output when a channel map follows
    bool whole_map = false;       // ch_map = 0 .. M-1: the transform writes the output itself
    HostPipe pipe;
};

namespace {

template <int M, int PMAX>
int launch_wave(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps, long long buf_items)
{
    const int ngroups = (nsteps + 15) / 16;
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    // fewer 16-step groups than two per CU: one workgroup per group, its steps split over four sets of M threads (k_pfbq)
    const int small_on = getenv("MI355_PFB_SMALL") ? atoi(getenv("MI355_PFB_SMALL")) : 1;  // (read per call: a tuning / test switch)
    if constexpr (!(M == 256 && PMAX > 16) && M <= 256) {  // (1024 threads x 32 taps per arm would spill: that shape keeps the ring kernel; 512 channels too)
        if (small_on && ngroups <= 2 * cus) {
            const long long n_in = (long long)buf_items - h->R + h->K;
            if (h->ident)
                hipLaunchKernelGGL((k_pfbq<M, PMAX, true>), dim3(ngroups), dim3(4 * M), 0, st, (const c32 *)in, (c32 *)out, h->d_taps,
                                   (const c32 *)h->d_tw, h->d_map, h->nmap, h->K, n_in, nsteps);
            else
                hipLaunchKernelGGL((k_pfbq<M, PMAX, false>), dim3(ngroups), dim3(4 * M), 0, st, (const c32 *)in, (c32 *)out, h->d_taps,
                                   (const c32 *)h->d_tw, h->d_map, h->nmap, h->K, n_in, nsteps);
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
    // waves per CU of the ring kernel; interleaved A/B at 64 x 32 over 2^26 samples: 4 -> 227 us, 8 -> 218, 16 -> 214, 32 -> 218
    // (512 channels: one 8-wave workgroup per CU and a single round: 8 -> 133 us, 16 -> 142, 32 -> 157 per 2^25 items)
    const int wpc = getenv("MI355_PFB_WAVES_PER_CU") ? atoi(getenv("MI355_PFB_WAVES_PER_CU")) : (M >= 512 ? 8 : 16);
    long long wgs = (long long)cus * (wpc > 0 ? wpc : 16) / (M / 64);
    if (wgs > ngroups) wgs = ngroups;
    const int per = (int)((ngroups + wgs - 1) / wgs);
    const int grid = (ngroups + per - 1) / per;
    const long long n_in = (long long)buf_items - h->R + h->K;
    if (h->ident)
        hipLaunchKernelGGL((k_pfbw<M, PMAX, true>), dim3(grid), dim3(M), 0, st, (const c32 *)in, (c32 *)out, h->d_taps, (const c32 *)h->d_tw,
                           h->d_map, h->nmap, h->K, n_in, nsteps, per, 0);
    else
        hipLaunchKernelGGL((k_pfbw<M, PMAX, false>), dim3(grid), dim3(M), 0, st, (const c32 *)in, (c32 *)out, h->d_taps, (const c32 *)h->d_tw,
                           h->d_map, h->nmap, h->K, n_in, nsteps, per, 0);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// 2- / 4-fold oversampling on the ring kernel: one launch per residue of the step number (see k_pfbw)
template <int M, int PMAX>
int launch_wave_over(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps)
{
    const int S = h->M / h->R, cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    const long long n_in = (long long)nsteps * h->R - h->R + h->K;
    // (512 channels: one 8-wave workgroup per CU and a single round: 8 -> 133 us, 16 -> 142, 32 -> 157 per 2^25 items)
    const int wpc = getenv("MI355_PFB_WAVES_PER_CU") ? atoi(getenv("MI355_PFB_WAVES_PER_CU")) : (M >= 512 ? 8 : 16);
    for (int par = 0; par < S; par++) {
        const int nsub = (nsteps - par + S - 1) / S;
        if (nsub <= 0) continue;
        const int ngroups = (nsub + 15) / 16;
        long long wgs = (long long)cus * (wpc > 0 ? wpc : 16) / (M / 64) / S;
        if (wgs < 1) wgs = 1;
        if (wgs > ngroups) wgs = ngroups;
        const int per = (int)((ngroups + wgs - 1) / wgs);
        const int grid = (ngroups + per - 1) / per;
        const c32 *src = (const c32 *)in + (size_t)par * h->R;
        const long long n_p = n_in - (long long)par * h->R;
#define PFBW_OVER(ID, OSF)                                                                                                                  \
    hipLaunchKernelGGL((k_pfbw<M, PMAX, ID, OSF>), dim3(grid), dim3(M), 0, st, src, (c32 *)out, h->d_taps, (const c32 *)h->d_tw, h->d_map, \
                       h->nmap, h->K, n_p, nsub, per, par)
        if (S == 2) {
            if (h->ident) PFBW_OVER(true, 2); else PFBW_OVER(false, 2);
        } else {
            if (h->ident) PFBW_OVER(true, 4); else PFBW_OVER(false, 4);
        }
#undef PFBW_OVER
        MI355_HIP(hipGetLastError());
    }
    return MI355_OK;
}

template <int M, int PMAX>
int launch_streams(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps, long long buf_items)
{
    constexpr int SEG = 64 / M;
    const int ngroups = (nsteps + 15) / 16;
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    long long streams = (long long)cus * 8 * SEG;  // 8 waves per CU, SEG streams per wave
    if (streams > ngroups) streams = ngroups;
    const int gps = (int)((ngroups + streams - 1) / streams);
    const int nstreams = (ngroups + gps - 1) / gps;
    const int grid = (nstreams + SEG - 1) / SEG;
    const long long n_in = (long long)buf_items - h->R + h->K;
    if (h->ident)
        hipLaunchKernelGGL((k_pfbs<M, PMAX, true>), dim3(grid), dim3(64), 0, st, (const c32 *)in, (c32 *)out, h->d_taps, (const c32 *)h->d_tw,
                           h->d_map, h->nmap, h->K, n_in, nsteps, gps);
    else
        hipLaunchKernelGGL((k_pfbs<M, PMAX, false>), dim3(grid), dim3(64), 0, st, (const c32 *)in, (c32 *)out, h->d_taps, (const c32 *)h->d_tw,
                           h->d_map, h->nmap, h->K, n_in, nsteps, gps);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

template <int M, int PMAX>
int launch_fast(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps, long long buf_items)
{
    // measured: 32 channels gain 6-11 % over the staged kernel; 8 and 16 channels (64-128 byte rows, gathered stores) lose 15 %
    if constexpr (M == 32 && PMAX <= 32) {
        static const bool wave = getenv("MI355_PFB_WAVE") ? atoi(getenv("MI355_PFB_WAVE")) != 0 : true;
        const long long n_in = (long long)buf_items - h->R + h->K;
        if (wave && n_in * 8 < (4ll << 30) - (64 << 10)) return launch_streams<M, PMAX>(h, in, out, st, nsteps, buf_items);
    }
    if constexpr (M == 512 && PMAX > 32) return MI355_ERR_STATE;  // (create sends that shape down the two-kernel path)
    if constexpr ((M == 64 || M == 128 || M == 256 || M == 512) && PMAX <= 32) {
        static const bool wave = getenv("MI355_PFB_WAVE") ? atoi(getenv("MI355_PFB_WAVE")) != 0 : true;
        const long long n_in = (long long)buf_items - h->R + h->K;
        // 32-bit byte offsets: the input as it is read (n_in * 8) and the OUTPUT including the rows the last group's unconditional, range-checked stores
        // overshoot by (up to 16 rows + the reversed row order inside a group: < 32 rows of M channels) -- an offset that wrapped past 2^32 would be
        // back in range and overwrite the first output rows
        if (wave && n_in * 8 < (4ll << 30) - (64 << 10) && ((long long)nsteps + 32) * M * 8 < (4ll << 30))
            return launch_wave<M, PMAX>(h, in, out, st, nsteps, buf_items);
    }
    if constexpr (M == 512) return MI355_ERR_STATE;  // 512 channels exist on the ring kernel only (create and work_dev_n keep its calls inside the 32-bit offsets)
    else {
    constexpr int T = 4096 / M;
    int ngroups = (nsteps + T - 1) / T;
    // many short grid-stride workgroups (measured at 8 and 16 channels: 2 per CU 250 GS/s, 8 per CU 280, 32 per CU 302)
    int grid = mi355_balanced_grid(h->ctx, ngroups, 16, 32);
    long long n_in = (long long)buf_items - h->R + h->K;
    if (h->ident)
        hipLaunchKernelGGL((k_pfb<M, PMAX, true>), dim3(grid), dim3(256), 0, st, (const c32 *)in, (c32 *)out, h->d_taps,
                           (const c32 *)h->d_tw, h->d_map, h->nmap, h->K, n_in, nsteps, ngroups);
    else
        hipLaunchKernelGGL((k_pfb<M, PMAX, false>), dim3(grid), dim3(256), 0, st, (const c32 *)in, (c32 *)out, h->d_taps,
                           (const c32 *)h->d_tw, h->d_map, h->nmap, h->K, n_in, nsteps, ngroups);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
    }
}

template <int M>
int launch_fast_m(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps, long long buf_items)
{
    switch (h->pmax) {
    case 8: return launch_fast<M, 8>(h, in, out, st, nsteps, buf_items);
    case 16: return launch_fast<M, 16>(h, in, out, st, nsteps, buf_items);
    case 32: return launch_fast<M, 32>(h, in, out, st, nsteps, buf_items);
    case 64: return launch_fast<M, 64>(h, in, out, st, nsteps, buf_items);
    }
    return MI355_ERR_STATE;
}

int launch_pfb(mi355_pfb *h, const void *in, void *out, hipStream_t st, int nsteps, long long buf_items)
{
    if (h->fast) {
        switch (h->M) {
        case 2: return launch_fast_m<2>(h, in, out, st, nsteps, buf_items);
        case 4: return launch_fast_m<4>(h, in, out, st, nsteps, buf_items);
        case 8: return launch_fast_m<8>(h, in, out, st, nsteps, buf_items);
        case 16: return launch_fast_m<16>(h, in, out, st, nsteps, buf_items);
        case 32: return launch_fast_m<32>(h, in, out, st, nsteps, buf_items);
        case 64: return launch_fast_m<64>(h, in, out, st, nsteps, buf_items);
        case 128: return launch_fast_m<128>(h, in, out, st, nsteps, buf_items);
        case 256: return launch_fast_m<256>(h, in, out, st, nsteps, buf_items);
        case 512: return launch_fast_m<512>(h, in, out, st, nsteps, buf_items);
        }
        return MI355_ERR_STATE;
    }
    if (h->fast_over) {
#define OVER(MM)                                                                           \
    case MM:                                                                               \
        if (h->pmax == 8) return launch_wave_over<MM, 8>(h, in, out, st, nsteps);          \
        if (h->pmax == 16) return launch_wave_over<MM, 16>(h, in, out, st, nsteps);        \
        return launch_wave_over<MM, 32>(h, in, out, st, nsteps);
        switch (h->M) {
            OVER(64)
            OVER(128)
            OVER(256)
        }
#undef OVER
        return MI355_ERR_STATE;
    }
    // critically sampled, a transform length with a one-pass mixed-radix plan, at most 32 taps per arm: filters + transform in one kernel (fft_mr.hip)
    if (h->dft && h->R == h->M) {
        int sign = 0;
        const MrPlan *mp = mi355_fft_mr_plan_of(h->dft, &sign);
        if (mp && mi355_fft_mr_pfb_ok(*mp, sign, h->K, h->M, nsteps)) {
            const int rc = mi355_fft_mr_pfb_launch(*mp, h->ctx, in, h->whole_map ? out : h->d_filt2, h->d_taps, h->K, h->M, nsteps, st);
            if (rc) return rc;
            if (!h->whole_map) {
                const long long tot = (long long)nsteps * h->nmap, blocks = (tot + 255) / 256;
                const long long cu8 = (long long)(h->ctx->num_cus > 0 ? h->ctx->num_cus : 256) * 8;
                hipLaunchKernelGGL(k_pfb_map, dim3((unsigned)(blocks < cu8 ? (blocks < 1 ? 1 : blocks) : cu8)), dim3(256), 0, st, (const c32 *)h->d_filt2,
                                   (c32 *)out, h->d_map, h->nmap, h->M, tot);
                MI355_HIP(hipGetLastError());
            }
            return MI355_OK;
        }
    }
    int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    long long total = (long long)nsteps * h->M;
    long long blocks = (total + 255) / 256;
    long long grid = blocks < (long long)cus * 8 ? blocks : (long long)cus * 8;
    const int over = h->R > 0 && h->M % h->R == 0 ? h->M / h->R : 0;  // 1: critically sampled; 2, 4: oversampled by that factor
    const int P_arm = (h->K + h->M - 1) / h->M;
    if (over == 1 && (P_arm <= 8 || (P_arm > 16 && P_arm <= 32)) && !getenv("MI355_PFB_BRANCHES_PER_OUTPUT") && !getenv("MI355_PFB_NO_FIR_RING")) {  // (64 taps per arm: 309 + 53 registers, slower than k_pfb_branches_t; 9 ... 16: no faster)
        FirRing f;
        f.in = (const c32 *)in;
        f.filt = (c32 *)h->d_filt;
        f.taps = h->d_taps;
        f.K = h->K;
        f.M = h->M;
        f.A = h->M <= 256 ? h->M : 256;
        f.Q = 256 / f.A;
        f.nsteps = nsteps;
        f.arm_blocks = (h->M + 255) / 256;
        f.m_A = (unsigned)((0x100000000ull + (unsigned)f.A - 1) / (unsigned)f.A);
        if (f.A == 1) { f.m_A = 0; f.Q = 0; }  // (one channel never gets here: M >= 2 on this path; kept defined)
        const int pm = P_arm <= 8 ? 8 : 32;  // (a 16-tap instance takes 254 registers -- it spills at 168 -- and is no faster than the 32-tap one on zero taps)
        // time ranges: about 12 waves per CU, but at least 8 x the warm-up long
        const long long want = (long long)cus * 12 / 4;  // workgroups
        long long nr = want / f.arm_blocks * f.Q;
        if (nr < 1) nr = 1;
        long long len = (nsteps + nr - 1) / nr;
        if (len < 8LL * P_arm) len = 8LL * P_arm;
        len = (len + 7) / 8 * 8;
        f.range_len = len;
        const long long nrb = (nsteps + len * f.Q - 1) / (len * f.Q);  // range blocks
        const dim3 gd((unsigned)(nrb * f.arm_blocks));
        const size_t lds = (size_t)pm * f.A * 4;
        if (pm == 8) hipLaunchKernelGGL((k_pfb_fir<8>), gd, dim3(256), lds, st, f);
        else hipLaunchKernelGGL((k_pfb_fir<32>), gd, dim3(256), lds, st, f);
    } else
    if ((over == 1 || over == 2 || over == 4) && !getenv("MI355_PFB_BRANCHES_PER_OUTPUT")) {
        constexpr int T = 8;
        const long long tt = ((long long)nsteps + T - 1) / T * h->M, tb = ((tt + 255) / 256 + 7) / 8 * 8;
        const long long g2 = tb < (long long)cus * 16 ? tb : (long long)cus * 16;
        const dim3 gd((unsigned)(g2 < 8 ? 8 : g2));
        static const int xr = getenv("MI355_PFB_NO_XCD_RUNS") ? 0 : 1;
        if (over == 1)
            hipLaunchKernelGGL((k_pfb_branches_t<T, 16, 1>), gd, dim3(256), 0, st, (const c32 *)in, (c32 *)h->d_filt, h->d_taps, h->K, h->M, nsteps, tt, xr);
        else if (over == 2)
            hipLaunchKernelGGL((k_pfb_branches_t<T, 8, 2>), gd, dim3(256), 0, st, (const c32 *)in, (c32 *)h->d_filt, h->d_taps, h->K, h->M, nsteps, tt, xr);
        else
            hipLaunchKernelGGL((k_pfb_branches_t<T, 4, 4>), gd, dim3(256), 0, st, (const c32 *)in, (c32 *)h->d_filt, h->d_taps, h->K, h->M, nsteps, tt, xr);
    } else
    hipLaunchKernelGGL(k_pfb_branches, dim3((unsigned)grid), dim3(256), 0, st, (const c32 *)in, (c32 *)h->d_filt, h->d_taps, h->K,
                       h->M, h->R, total);
    total = (long long)nsteps * h->nmap;
    blocks = (total + 255) / 256;
    grid = blocks < (long long)cus * 8 ? blocks : (long long)cus * 8;
    if (grid < 1) grid = 1;
    if (h->dft) {
        MI355_HIP(hipGetLastError());
        const int rc = mi355_fft_work_dev(h->dft, nsteps, h->d_filt, h->whole_map ? out : h->d_filt2, st);
        if (rc) return rc;
        if (!h->whole_map) {
            hipLaunchKernelGGL(k_pfb_map, dim3((unsigned)grid), dim3(256), 0, st, (const c32 *)h->d_filt2, (c32 *)out, h->d_map, h->nmap, h->M, total);
            MI355_HIP(hipGetLastError());
        }
        return MI355_OK;
    }
    hipLaunchKernelGGL(k_pfb_dft_map, dim3((unsigned)grid), dim3(256), 0, st, (const c32 *)h->d_filt, (c32 *)out,
                       (const c32 *)h->d_tw, h->d_map, h->nmap, h->M, total);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

}  // namespace

extern "C" int mi355_pfb_destroy(mi355_pfb *h)
{
    if (!h) return MI355_OK;
    (void)hipSetDevice(h->ctx->device);
    h->pipe.release();
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_tw) (void)hipFree(h->d_tw);
    if (h->d_map) (void)hipFree(h->d_map);
    if (h->d_filt) (void)hipFree(h->d_filt);
    if (h->d_filt2) (void)hipFree(h->d_filt2);
    if (h->dft) (void)mi355_fft_destroy(h->dft);
    delete h;
    return MI355_OK;
}

extern "C" int mi355_pfb_create(mi355_ctx *ctx, const float *taps, int ntaps, int buf_items, int num_channels, int ninputs_per_iter,
                                const int *ch_map, int nmap, mi355_pfb **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(taps && ntaps >= 1, "taps must hold at least one tap");
    MI355_REQUIRE(num_channels >= 1 && ninputs_per_iter >= 1 && ninputs_per_iter <= num_channels,
                  "need 1 <= ninputs_per_iter <= num_channels");
    MI355_REQUIRE(buf_items > 0 && buf_items % num_channels == 0, "buf_items must be a multiple of num_channels");  // :59-62
    MI355_REQUIRE(buf_items % ninputs_per_iter == 0, "buf_items must be a multiple of ninputs_per_iter");
    MI355_REQUIRE(ch_map && nmap >= 1, "ch_map must hold at least one channel");
    for (int q = 0; q < nmap; q++) MI355_REQUIRE(ch_map[q] >= 0 && ch_map[q] < num_channels, "ch_map entry out of range");
    mi355_pfb *h = new (std::nothrow) mi355_pfb();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->K = ntaps; h->M = num_channels; h->R = ninputs_per_iter; h->buf_items = buf_items; h->nmap = nmap;
    h->nsteps = buf_items / ninputs_per_iter;
    const int M = h->M;
    const int per_arm = (ntaps + M - 1) / M;
    h->fast = (M >= 2 && M <= 256 && (M & (M - 1)) == 0 && h->R == M && per_arm <= 64);
    // 512 channels, at most 32 taps per arm: the ring kernel with one 512-thread workgroup per CU (256 registers per thread); it has no staged
    // fallback, so a buffer must fit the ring kernel's 32-bit offsets (launch_fast)
    if (M == 512 && h->R == M && per_arm <= 32 && !getenv("MI355_PFB_NO_RING_512") &&
        ((long long)buf_items - h->R + ntaps) * 8 < (4ll << 30) - (64 << 10) && ((long long)h->nsteps + 32) * M * 8 < (4ll << 30))
        h->fast = true;
    h->pmax = per_arm <= 8 ? 8 : per_arm <= 16 ? 16 : per_arm <= 32 ? 32 : 64;
    {
        const long long n_in1 = (long long)buf_items - h->R + ntaps;
        h->fast_over = !h->fast && (M == 64 || M == 128 || M == 256) && (h->R * 2 == M || h->R * 4 == M) && per_arm <= 32 &&
                       n_in1 * 8 < (4ll << 30) - (64 << 10) && !getenv("MI355_PFB_NO_FAST_OVERSAMPLED");
    }
    // the register-direct store writes short runs per lane for 8 <= M <= 16 (measured 10-27 % of HBM peak); those go through the
    // LDS gather of the mapped path, whose stores are lane-consecutive (46-49 %)
    h->ident = (nmap == M) && (M >= 32 || M <= 4);
    for (int q = 0; q < nmap && h->ident; q++) h->ident = (ch_map[q] == q);
    auto fail = [&](int rc) { mi355_pfb_destroy(h); return rc; };
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(MI355_ERR_HIP);
    std::vector<float> t;
    if (h->fast || h->fast_over) {
        t.assign((size_t)h->pmax * M, 0.0f);
        for (int k = 0; k < ntaps; k++) t[k] = taps[k];
    } else {
        t.assign(taps, taps + ntaps);
    }
    std::vector<float> tw(2 * (size_t)M);
    for (int k = 0; k < M; k++) {
        double a = 2.0 * M_PI * (double)k / (double)M;
        tw[2 * k] = (float)cos(a); tw[2 * k + 1] = (float)sin(a);
    }
    if (hipMalloc((void **)&h->d_taps, t.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_NOMEM);
    if (hipMalloc(&h->d_tw, tw.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_NOMEM);
    if (hipMalloc((void **)&h->d_map, (size_t)nmap * sizeof(int)) != hipSuccess) return fail(MI355_ERR_NOMEM);
    if (!h->fast && !h->fast_over && hipMalloc(&h->d_filt, (size_t)h->nsteps * M * 8) != hipSuccess) return fail(MI355_ERR_NOMEM);
    // the M-point transform as a clFFT handle when 16 or more channels are mapped (M products per output below that) -- and, whatever the map, when
    // the one-kernel form (k_pfb_mr: filters + transform, fft_mr.hip) takes this channelizer: 10 or 12 channels run on it as well
    const bool few = nmap < 16 || M < 8;
    const bool try_fused = h->R == M && M >= 6 && M <= 512 && (h->K + M - 1) / M <= 32;
    if (!h->fast && !h->fast_over && (!few || try_fused) && !getenv("MI355_PFB_DIRECT_DFT")) {
        h->whole_map = nmap == M;
        for (int q = 0; q < nmap && h->whole_map; q++) h->whole_map = ch_map[q] == q;
        const int rc = mi355_fft_create(ctx, M, MI355_FFT_BACKWARD, nullptr, 0, MI355_DTYPE_COMPLEX, 1, 0, &h->dft);
        if (rc != MI355_OK) return fail(rc);
        if (few) {
            int sign = 0;
            const MrPlan *mp = mi355_fft_mr_plan_of(h->dft, &sign);
            if (!mp || !mi355_fft_mr_pfb_ok(*mp, sign, h->K, M, h->nsteps)) {  // not the one-kernel form after all: the direct DFT as before
                (void)mi355_fft_destroy(h->dft);
                h->dft = nullptr;
            }
        }
        if (h->dft && !h->whole_map && hipMalloc(&h->d_filt2, (size_t)h->nsteps * M * 8) != hipSuccess) return fail(MI355_ERR_NOMEM);
    }
    if (mi355_upload(ctx, h->d_taps, t.data(), t.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_HIP);
    if (mi355_upload(ctx, h->d_tw, tw.data(), tw.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_HIP);
    if (mi355_upload(ctx, h->d_map, ch_map, (size_t)nmap * sizeof(int)) != hipSuccess) return fail(MI355_ERR_HIP);
    int rc = h->pipe.init(ctx);
    if (rc) return fail(rc);
    // (the table uploads ran on the context's upload stream and were waited for there: mi355_upload; no device-wide wait)
    mi355_log(ctx, MI355_LOG_INFO, "clPolyphaseChannelizer: %d channels, %d taps, %d inputs per step, %d of %d outputs mapped, %d items per call: %s kernel",
              num_channels, ntaps, ninputs_per_iter, nmap, num_channels, buf_items, h->fast ? "fused filter + transform" : "two-pass");
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_pfb_noutput(const mi355_pfb *h) { return h ? h->nmap * h->nsteps : MI355_ERR_INVALID_ARG; }

// input items one call reads, history included (buf_items - R + ntaps; equals the reference's
// buf_items + history() - num_channels for R == M, lib/clPolyphaseChannelizer_impl.cc:97)
extern "C" int mi355_pfb_ninput(const mi355_pfb *h) { return h ? h->buf_items - h->R + h->K : MI355_ERR_INVALID_ARG; }

// nbuf consecutive buffers of the stream in ONE launch: what general_work() does when the scheduler offers several output
// multiples at once (noutput_items = nbuf * noutput()).  in: nbuf * buf_items - ninputs_per_iter + ntaps samples, out: nbuf *
// noutput().  The reference handles one buffer per call (lib/clPolyphaseChannelizer_impl.cc:83-109); the results are the
// same samples (identical arithmetic per output), the launch cost is paid once.
extern "C" int mi355_pfb_work_dev_n(mi355_pfb *h, int nbuf, const void *in, void *out, void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    MI355_REQUIRE(in && out && nbuf >= 1, "NULL buffer or nbuf < 1");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
                  "device buffers must be 8-byte aligned");
    MI355_REQUIRE((long long)h->nsteps * nbuf <= 0x7fffffffLL, "too many buffers in one call");
    MI355_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = mi355_pick_stream(h->ctx, stream);
    // (2- / 4-fold oversampled on the ring kernel: buf_items is a whole number of M-item frames, so a buffer holds a whole number of
    // step residues and nbuf buffers are one longer stream)
    const bool over_one_stream = h->fast_over && ((long long)h->buf_items * nbuf + h->K) * 8 < (4ll << 30) - (64 << 10);  // (32-bit row offsets)
    // filters + transform in one kernel (k_pfb_mr) writing the output itself: no scratch, nbuf buffers are one longer stream
    bool mr_one_stream = false;
    if (!h->fast && !over_one_stream && h->dft && h->whole_map && h->R == h->M) {
        int sign = 0;
        const MrPlan *mp = mi355_fft_mr_plan_of(h->dft, &sign);
        mr_one_stream = mp && mi355_fft_mr_pfb_ok(*mp, sign, h->K, h->M, h->nsteps * nbuf);
    }
    if (!h->fast && !over_one_stream && !mr_one_stream) {  // the generic two-kernel path keeps a one-buffer scratch: one buffer at a time
        for (int b = 0; b < nbuf; b++) {
            const int rc = launch_pfb(h, (const char *)in + (size_t)b * h->buf_items * 8, (char *)out + (size_t)b * h->nmap * h->nsteps * 8, st,
                                      h->nsteps, h->buf_items);
            if (rc) return rc;
        }
        return MI355_OK;
    }
    if (h->M == 512 && h->fast && nbuf > 1 &&
        !(((long long)h->buf_items * nbuf - h->R + h->K) * 8 < (4ll << 30) - (64 << 10) && ((long long)h->nsteps * nbuf + 32) * h->M * 8 < (4ll << 30))) {
        for (int b = 0; b < nbuf; b++) {  // (the ring kernel's 32-bit offsets: one buffer at a time; same samples either way)
            const int rc = launch_pfb(h, (const char *)in + (size_t)b * h->buf_items * 8, (char *)out + (size_t)b * h->nmap * h->nsteps * 8, st,
                                      h->nsteps, h->buf_items);
            if (rc) return rc;
        }
        return MI355_OK;
    }
    return launch_pfb(h, in, out, st, h->nsteps * nbuf, (long long)h->buf_items * nbuf);
}

extern "C" int mi355_pfb_work_dev(mi355_pfb *h, const void *in, void *out, void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    MI355_REQUIRE(in && out, "NULL buffer");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
                  "device buffers must be 8-byte aligned");
    MI355_HIP(hipSetDevice(h->ctx->device));
    return launch_pfb(h, in, out, mi355_pick_stream(h->ctx, stream), h->nsteps, h->buf_items);
}

extern "C" int mi355_pfb_work(mi355_pfb *h, const void *in, void *out)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    MI355_REQUIRE(in && out, "NULL buffer");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    // one general_work() is one fixed-size transfer (buf_items is a block parameter): single slot
    size_t inb = (size_t)mi355_pfb_ninput(h) * 8, outb = (size_t)mi355_pfb_noutput(h) * 8;
    int rc = h->pipe.ensure(1, &inb, outb);
    if (rc) return rc;
    HostPipe &p = h->pipe;
    hipStream_t st = h->ctx->stream[0];
    mi355_copy(p.h_in[0][0], in, inb);
    if (mi355_direct_ok(inb > outb ? inb : outb)) {  // small call: the kernel works on the pinned staging itself (common.h)
        rc = launch_pfb(h, p.h_in[0][0], p.h_out[0], st, h->nsteps, h->buf_items);
        if (rc) return rc;
        MI355_HIP(mi355_direct_sync(st));
        mi355_copy(out, p.h_out[0], outb);
        return MI355_OK;
    }
    MI355_HIP(hipMemcpyAsync(p.d_in[0][0], p.h_in[0][0], inb, hipMemcpyHostToDevice, st));
    rc = launch_pfb(h, p.d_in[0][0], p.d_out[0], st, h->nsteps, h->buf_items);
    if (rc) return rc;
    MI355_HIP(hipMemcpyAsync(p.h_out[0], p.d_out[0], outb, hipMemcpyDeviceToHost, st));
    MI355_HIP(hipStreamSynchronize(st));
    mi355_copy(out, p.h_out[0], outb);
    return MI355_OK;
}
