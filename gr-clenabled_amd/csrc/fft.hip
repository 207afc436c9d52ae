// clFFT block as one hand-written gfx950 kernel.
// Reference behaviour: lib/clFFT_impl.cc:65-151 (plan), :526-634 (processOpenCL:
// per-frame H2D, MultiplyFloat window kernel, clfftEnqueueTransform batch 1,
// D2H, host fftshift); CPU twin :464-518.
//
// Design: Stockham autosort FFT with the frame resident in LDS.  A workgroup of
// 256 threads owns 4096 points (= 4096/N frames); every thread keeps 16 complex
// points in registers and does 16/R radix-R butterflies per pass (radix 16
// wherever possible: 4096 = 16*16*16 -> two LDS exchanges).  The first pass
// reads global memory and the last pass writes it, both with unit stride across
// lanes; the window multiply is fused into the load, fftshift into the store
// index (forward) or the load index (reverse).  Workgroups are persistent
// (grid-stride over frame groups) so the window values and the inter-pass
// twiddles -- which depend only on the thread's position -- are loaded once into
// registers from a double-precision-generated table and cost nothing per frame.
// HBM traffic per frame = N*8 B in + N*8 B out (16 B/sample), nothing else.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "fft_core.hpp"
#include "fft_mr.h"

using namespace fftc;
#ifndef MI355_FFT_WPE
#define MI355_FFT_WPE 3
#endif

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));

// Raw pass-0 inputs of one frame group: v[q*R0 + r] = x[fr][(j + r*B0) ^ in_xor].  Streaming
// (nontemporal) loads, branch free: a thread whose frame does not exist (ragged last group) reads
// frame 0 of the group instead and the value is discarded.  in_xor is 0 or N/2, a multiple of B0
// (R0 >= 2), so it only permutes the r*B0 term.
template <int N, bool REAL, class G>
__device__ __forceinline__ void load_group(c32 (&v)[16], const void *__restrict__ in, int grp, int tid, int nframes, int in_xor)
{
    using P = Plan<N>;
    constexpr int TH = G::TH, PTS = G::PTS, F = G::F, R0 = P::radix(0), B0 = N / R0;
    const int frames_left = nframes - grp * F;
    if constexpr (N <= 64) {  // consecutive elements per lane; redistributed through LDS by the caller
        (void)in_xor;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned e = (unsigned)(tid + TH * k);
            const bool ok = (int)(e / N) < frames_left;
            if constexpr (REAL) {
                const float x = __builtin_nontemporal_load((const float *)in + (size_t)grp * PTS + (ok ? e : 0u));
                v[k] = mk(ok ? x : 0.f, 0.f);
            } else {
                const f2v x = __builtin_nontemporal_load((const f2v *)in + (size_t)grp * PTS + (ok ? e : 0u));
                v[k] = ok ? mk(x.x, x.y) : mk(0.f, 0.f);
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 16 / R0; q++) {
        const int g = tid + TH * q, fr = g / B0;
        const bool ok = (F == 1) || fr < frames_left;
        const unsigned off = (unsigned)((ok ? fr * N : 0) + (g % B0));
#pragma unroll
        for (int r = 0; r < R0; r++) {
            const unsigned e = off + (unsigned)((r * B0) ^ in_xor);
            if constexpr (REAL) {
                const float x = __builtin_nontemporal_load((const float *)in + (size_t)grp * PTS + e);
                v[q * R0 + r] = mk(ok ? x : 0.f, 0.f);
            } else {
                const f2v x = __builtin_nontemporal_load((const f2v *)in + (size_t)grp * PTS + e);
                v[q * R0 + r] = ok ? mk(x.x, x.y) : mk(0.f, 0.f);
            }
        }
    }
}

template <int N, int SIGN, bool REAL, int PF, class G>
__global__ __launch_bounds__(G::TH, (N <= 4096 ? (PF == 2 ? 2 : MI355_FFT_WPE) : 1)) void k_fft(const void *__restrict__ in, c32 *__restrict__ out,
                                                                 const float *__restrict__ window,
                                                                 const c32 *__restrict__ twtab, int nframes, int ngroups,
                                                                 int shift)
{
    using P = Plan<N>;
    constexpr int TH = G::TH, PTS = G::PTS, F = G::F, NP = P::NP;
    // N <= 64: a thread's pass-0 operands are only N/16 (<= 4) consecutive elements, so loading them directly costs
    // one 128-B line per lane and instruction.  Instead every lane moves CONSECUTIVE elements and a padded LDS image
    // (PAD slots after every frame: conflict-free for both access patterns) redistributes them; same on the store side.
    constexpr bool SMALL = N <= 64;
    constexpr int PAD = (N / P::radix(0)) > 1 ? (N / P::radix(0)) : 1;
    constexpr int LDS_SLOTS = SMALL ? PTS + (PTS / N) * PAD : (NP > 1 ? PTS : 1);
    __shared__ c32 lds[LDS_SLOTS];
    const int tid0 = threadIdx.x;
    const int in_xor = (SIGN > 0 && shift) ? (N >> 1) : 0;   // reverse: halves swapped on load (:548-553)
    const int out_xor = (SIGN < 0 && shift) ? (N >> 1) : 0;  // forward: halves swapped on store (:594-607)

    // ---- per-thread constants: inter-pass twiddles and window values --------------
    TwRegs<N> tw;
    load_twiddles<N, false, G>(tw, tid0, twtab);
    constexpr int R0 = P::radix(0), B0 = N / R0;
    float win[16];  // the handle always carries a window (all ones when the block has none)
    if constexpr (SMALL) {
        win[0] = window[tid0 % N];  // the thread moves position tid % N of 16 different frames
    } else {
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int j = (tid0 + TH * q) % B0;
#pragma unroll
            for (int r = 0; r < R0; r++) win[q * R0 + r] = window[j + ((r * B0) ^ in_xor)];
        }
    }

    // One frame group: window, transform, store.  `tid` is an opaque per-iteration copy of the thread id: address arithmetic is
    // recomputed (a few dozen integer ops) rather than hoisted into ~60 loop-carried registers.
    auto body = [&](c32 (&cur)[16], int grp, int tid) {
        const int frames_left = nframes - grp * F;  // frames of this group that exist
        c32 v[16];
        if constexpr (SMALL) {
            // cur[k] = source element tid + TH*k of the group (position (tid % N) of its frame: one window value per thread)
            const float wv = win[0];
            const int pos = (tid % N) ^ in_xor, fr0 = tid / N;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int fr = fr0 + k * (TH / N);
                lds[fr * (N + PAD) + pos] = scale(cur[k], wv);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16 / R0; q++) {
                const int g = tid + TH * q, base = (g / B0) * (N + PAD) + (g % B0);
#pragma unroll
                for (int r = 0; r < R0; r++) v[q * R0 + r] = lds[base + r * B0];
            }
            __syncthreads();  // the transform reuses the LDS in its own layout
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = scale(cur[s], win[s]);
        }
        transform_regs<N, SIGN, false, G>(v, tw, lds, tid);
        // registers -> global (streaming stores), unit stride across lanes, fftshift fused.
        // out_xor is 0 or N/2, a multiple of BL, so it only permutes the s*BL term.
        if constexpr (SMALL) {
            constexpr int RL = P::radix(NP - 1), BL = N / RL;
            if constexpr (NP > 1) __syncthreads();  // the last pass' LDS reads are done
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, base = (g / BL) * (N + PAD) + (g % BL);
#pragma unroll
                for (int s = 0; s < RL; s++) lds[base + ((orev<RL>(s) * BL) ^ out_xor)] = v[q * RL + s];
            }
            __syncthreads();
            f2v *__restrict__ out_g = (f2v *)out + (size_t)grp * PTS;
            const int pos = tid % N, fr0 = tid / N;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int fr = fr0 + k * (TH / N);
                if (fr < frames_left) {
                    const c32 z = lds[fr * (N + PAD) + pos];
                    f2v o;
                    o.x = z.x;
                    o.y = z.y;
                    __builtin_nontemporal_store(o, out_g + tid + TH * k);
                }
            }
        } else {
            constexpr int RL = P::radix(NP - 1), BL = N / RL;
            f2v *__restrict__ out_g = (f2v *)out + (size_t)grp * PTS;
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, fr = g / BL;
                const unsigned off = (unsigned)(fr * N + (g % BL));
                if ((F == 1) || fr < frames_left) {
#pragma unroll
                    for (int s = 0; s < RL; s++) {
                        f2v o;
                        o.x = v[q * RL + s].x;
                        o.y = v[q * RL + s].y;
                        __builtin_nontemporal_store(o, out_g + off + (unsigned)((orev<RL>(s) * BL) ^ out_xor));
                    }
                }
            }
        }
        if constexpr (NP > 1 || SMALL) __syncthreads();  // last pass' LDS reads finish before the next group's writes
    };
    const int stride = gridDim.x;
    if constexpr (PF == 2) {
        // Two groups of lead (64 KiB in flight per workgroup, two persistent workgroups per CU): the loop is unrolled three times so
        // that the three register buffers rotate by name, not by moves.
        c32 b0[16], b1[16], b2[16];
        int grp = blockIdx.x;
        if (grp < ngroups) load_group<N, REAL, G>(b0, in, grp, tid0, nframes, in_xor);
        if (grp + stride < ngroups) load_group<N, REAL, G>(b1, in, grp + stride, tid0, nframes, in_xor);
#define MI355_FFT_STEP(CUR, FREE)                                                                   \
        {                                                                                           \
            if (grp >= ngroups) break;                                                              \
            int tid = tid0;                                                                         \
            asm volatile("" : "+v"(tid));                                                           \
            if (grp + 2 * stride < ngroups) load_group<N, REAL, G>(FREE, in, grp + 2 * stride, tid, nframes, in_xor); \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            body(CUR, grp, tid);                                                                    \
            grp += stride;                                                                          \
        }
        for (;;) {
            MI355_FFT_STEP(b0, b2)
            MI355_FFT_STEP(b1, b0)
            MI355_FFT_STEP(b2, b1)
        }
#undef MI355_FFT_STEP
    } else {
        // PF == 1: the loads of the NEXT frame group are issued before the current group is transformed
        c32 cur[16];
        if ((int)blockIdx.x < ngroups) load_group<N, REAL, G>(cur, in, blockIdx.x, tid0, nframes, in_xor);
        for (int grp = blockIdx.x; grp < ngroups; grp += stride) {
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            c32 nxt[16];
            if constexpr (PF == 1) {
                const int gnext = grp + stride;
                if (gnext < ngroups) load_group<N, REAL, G>(nxt, in, gnext, tid, nframes, in_xor);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the transform
            } else if (grp != (int)blockIdx.x) {
                load_group<N, REAL, G>(cur, in, grp, tid, nframes, in_xor);
            }
            body(cur, grp, tid);
            if constexpr (PF == 1) {
#pragma unroll
                for (int s = 0; s < 16; s++) cur[s] = nxt[s];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// N = 8192 / 16384: S = N/4096 interleaved 4096-point sub-transforms per thread (decimation in time), combined by a
// radix-S butterfly in registers.  256 threads, one frame per iteration.  Sub-frame s holds x[S*n + s], so one 16-B load
// per lane brings the same n of two sub-frames; the sub-transforms run one after the other through the same 32 KiB of
// LDS (3 workgroups per CU instead of the 1-2 a whole-frame LDS image allows, and two exchange passes per sub-frame
// instead of three per frame); X[k + m*4096] = sum_s W_N^(s k) W_S^(s m) E_s[k] needs all E_s[k] of one k in one thread,
// which the common sub-transform layout guarantees.  W_N^(s k) = W_N^(s tid) * (compile-time constant), k = tid + 256 c.
// ------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
// cos / sin of 2 pi m / circ (circ divides 128) as compile-time constants: the combine twiddles of k_fft_s.  A literal table:
// indexing a constexpr array folds away once the loops are unrolled (a constexpr series evaluation did not, and left the
// whole frame in scratch memory).
__host__ __device__ constexpr float cos128(int m)
{
    constexpr float t[128] = {1.0f, 0.99879545f, 0.99518472f, 0.989176512f, 0.980785251f, 0.970031261f, 0.956940353f, 0.941544056f, 0.923879504f, 0.903989315f, 0.881921291f, 0.857728601f, 0.831469595f, 0.803207517f, 0.773010433f, 0.740951121f, 0.707106769f, 0.671558976f, 0.634393275f, 0.59569931f, 0.555570245f, 0.514102757f, 0.471396744f, 0.427555084f, 0.382683426f, 0.336889863f, 0.290284663f, 0.242980182f, 0.195090324f, 0.146730468f, 0.0980171412f, 0.0490676761f, 0.0f, -0.0490676761f, -0.0980171412f, -0.146730468f, -0.195090324f, -0.242980182f, -0.290284663f, -0.336889863f, -0.382683426f, -0.427555084f, -0.471396744f, -0.514102757f, -0.555570245f, -0.59569931f, -0.634393275f, -0.671558976f, -0.707106769f, -0.740951121f, -0.773010433f, -0.803207517f, -0.831469595f, -0.857728601f, -0.881921291f, -0.903989315f, -0.923879504f, -0.941544056f, -0.956940353f, -0.970031261f, -0.980785251f, -0.989176512f, -0.99518472f, -0.99879545f, -1.0f, -0.99879545f, -0.99518472f, -0.989176512f, -0.980785251f, -0.970031261f, -0.956940353f, -0.941544056f, -0.923879504f, -0.903989315f, -0.881921291f, -0.857728601f, -0.831469595f, -0.803207517f, -0.773010433f, -0.740951121f, -0.707106769f, -0.671558976f, -0.634393275f, -0.59569931f, -0.555570245f, -0.514102757f, -0.471396744f, -0.427555084f, -0.382683426f, -0.336889863f, -0.290284663f, -0.242980182f, -0.195090324f, -0.146730468f, -0.0980171412f, -0.0490676761f, 0.0f, 0.0490676761f, 0.0980171412f, 0.146730468f, 0.195090324f, 0.242980182f, 0.290284663f, 0.336889863f, 0.382683426f, 0.427555084f, 0.471396744f, 0.514102757f, 0.555570245f, 0.59569931f, 0.634393275f, 0.671558976f, 0.707106769f, 0.740951121f, 0.773010433f, 0.803207517f, 0.831469595f, 0.857728601f, 0.881921291f, 0.903989315f, 0.923879504f, 0.941544056f, 0.956940353f, 0.970031261f, 0.980785251f, 0.989176512f, 0.99518472f, 0.99879545f};
    return t[m & 127];
}
__host__ __device__ constexpr float circ_cos(int m, int circ) { return cos128(m * (128 / circ)); }
__host__ __device__ constexpr float circ_sin(int m, int circ) { return cos128(m * (128 / circ) - 32); }
static_assert(circ_cos(0, 128) == 1.0f && circ_cos(32, 128) == 0.0f && circ_cos(16, 128) == 0.707106781f && circ_sin(32, 128) == 1.0f &&
                  circ_sin(8, 64) == 0.707106781f && circ_cos(16, 32) == -1.0f, "twiddle table");

// N = 32768: eight interleaved 4096-point sub-transforms per frame (x[8 n + s]), one frame per workgroup iteration, ONE pass
// over HBM.  The whole frame in the registers of 256 threads (k_fft_s with S = 8) leaves one wave per SIMD running eight
// transforms one after the other (400 us per 2^26 samples, no better than the two-kernel workspace scheme).  Here 512 threads
// form two sets of 256: set g transforms sub-frames 4g .. 4g+3 (64 complex values per thread, its own 32 KiB of LDS; the
// two sets run the same code between the same barriers), then each set hands the other the half of its outputs the other
// combines (128 KiB of LDS, reusing the transform areas) and runs the radix-8 combine
//   X[k + 4096 m] = sum_s W_N^(s k) W_8^(s m) E_s[k]
// for 8 of the 16 output rows a thread holds; rows are stored k-contiguous.
template <int SIGN, bool REAL>
__global__ __launch_bounds__(512, 2) void k_fft_32k(const void *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ window,
                                                    const c32 *__restrict__ twN, int nframes, int shift)
{
    constexpr int N = 32768, NS = 4096, S = 8, H = 4, BL = 256;
    using G = Geo<NS>;
    __shared__ c32 sm[4 * NS];  // transform areas of the two sets (2 x 32 KiB), then the exchange image (128 KiB)
    const int set = threadIdx.x >> 8, tid0 = threadIdx.x & 255;
    c32 *lds = sm + set * 2 * NS;  // two transform areas per set: sub-frames are transformed two at a time in lockstep
    const int in_xor = (SIGN > 0 && shift) ? 8 : 0;       // reverse: halves swapped on load == n ^ 2048 == r ^ 8
    const int m_xor = (SIGN < 0 && shift) ? (S / 2) : 0;  // forward: halves swapped on store

    // Input: a thread reads WHOLE 64-byte groups (element n of all eight sub-frames) -- set g the rows 8g .. 8g+7 -- and hands the
    // other set the half it transforms through LDS (128 KiB, the exchange image of the combine, free at this point).  When the two
    // sets each read their own 32 bytes of every group (round 2), the two requests for one 64-byte sector came from different
    // waves at almost the same time and a good part of them went to memory twice: FETCH_SIZE 1.67 x the input, now 1.11 x
    // (the rate is unchanged, 298 -> 299 us per 2^26 samples: this kernel is bound by its phases -- one workgroup per CU -- not by HBM).
    // The first rows of the NEXT frame are fetched before the combine of the current one (registers that are free at that
    // point: the peak is inside the transforms), so part of a frame's load latency runs under the combine and the stores.
    // (rows of the next frame fetched ahead, complex input, per 2^26 samples: 0 -> 288 us, 1 -> 299, 2 -> 312, 3 -> 322: the registers they
    // take are spilled -- 12 registers are spilled even with none -- so none are fetched ahead)
#ifndef PFR32K
#define PFR32K 0
#endif
    constexpr int PFW = REAL ? 2 : 4, PFR = REAL ? 2 * PFR32K : PFR32K;
    f4v pf[PFR > 0 ? PFR : 1][PFW];
    auto fetch = [&](int frame, int j, int tid, f4v (&t)[PFW]) {
        const unsigned n = (unsigned)(tid + (((8 * set + j) ^ in_xor) * BL));  // element n of every sub-frame
        const f4v *p = (const f4v *)in + (size_t)frame * (REAL ? N / 4 : N / 2) + (size_t)n * PFW;
#pragma unroll
        for (int k = 0; k < PFW; k++) t[k] = p[k];
    };
    if ((int)blockIdx.x < nframes) {
#pragma unroll
        for (int j = 0; j < PFR; j++) fetch(blockIdx.x, j, tid0, pf[j]);
    }
    for (int frame = blockIdx.x; frame < nframes; frame += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        c32 v[H][16];
        auto load_frame = [&](auto set_tag) {
            constexpr int SET = decltype(set_tag)::value, OTHER = 1 - SET;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 8 * SET + j;
                const unsigned n = (unsigned)(tid + ((r ^ in_xor) * BL));
                const f4v wa = *((const f4v *)window + (size_t)n * 2), wb = *((const f4v *)window + (size_t)n * 2 + 1);
                const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                f4v t[PFW];
                if (j < PFR) {
#pragma unroll
                    for (int k = 0; k < PFW; k++) t[k] = pf[j][k];
                } else {
                    fetch(frame, j, tid, t);
                }
                c32 x[S];
                if constexpr (REAL) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        x[4 * q] = mk(t[q].x * w[4 * q], 0.f); x[4 * q + 1] = mk(t[q].y * w[4 * q + 1], 0.f);
                        x[4 * q + 2] = mk(t[q].z * w[4 * q + 2], 0.f); x[4 * q + 3] = mk(t[q].w * w[4 * q + 3], 0.f);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        x[2 * q] = mk(t[q].x * w[2 * q], t[q].y * w[2 * q]);
                        x[2 * q + 1] = mk(t[q].z * w[2 * q + 1], t[q].w * w[2 * q + 1]);
                    }
                }
#pragma unroll
                for (int sp = 0; sp < H; sp++) {
                    v[sp][r] = x[H * SET + sp];
                    sm[((SET * H + sp) * 8 + j) * BL + tid] = x[H * OTHER + sp];  // xch[set of origin][sub-frame of the receiver][row][tid]
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int sp = 0; sp < H; sp++) v[sp][8 * OTHER + j] = sm[((OTHER * H + sp) * 8 + j) * BL + tid];
            __syncthreads();  // the image is read before the transforms reuse the area
        };
        if (set == 0) load_frame(std::integral_constant<int, 0>{});
        else load_frame(std::integral_constant<int, 1>{});
        TwRegs<NS> tw;  // re-read (L1/L2) per frame: their 24 registers are what the prefetch of the next frame lives in during the combine
        load_twiddles<NS, false, G>(tw, tid, twN + N);  // the 4096-point table follows the N-point one
        transform_regs2<NS, SIGN, false, G>(v[0], v[1], tw, lds, lds + NS, tid);
        __syncthreads();
        transform_regs2<NS, SIGN, false, G>(v[2], v[3], tw, lds, lds + NS, tid);
        __syncthreads();
        c32 wb[S - 1];
#pragma unroll
        for (int s = 1; s < S; s++) wb[s - 1] = twN[(s * tid) & (N - 1)];
        f2v *__restrict__ out_f = (f2v *)out + (size_t)frame * N + tid;
        // exchange image: xch[set of origin][sub-frame in set][row of the receiving set's half][tid]
        auto finish = [&](auto set_tag) {
            constexpr int SET = decltype(set_tag)::value, OTHER = 1 - SET;
#pragma unroll
            for (int s = 0; s < H; s++)
#pragma unroll
                for (int tl = 0; tl < 8; tl++) sm[((SET * H + s) * 8 + tl) * BL + tid] = v[s][8 * OTHER + tl];  // rows the other set combines
            // (half of v[] is dead from here on: room for the first rows of the next frame)
            if (frame + (int)gridDim.x < nframes) {
#pragma unroll
                for (int j = 0; j < PFR; j++) fetch(frame + gridDim.x, j, tid, pf[j]);
            }
            __syncthreads();
#pragma unroll
            for (int tl = 0; tl < 8; tl++) {
                constexpr int CIRC = 16 * S;  // W_N^(s * c * 256) = exp(sign 2 pi i s c / 128): a point of the 128-point circle
                const int t = 8 * SET + tl, c = orev<16>(t);
                c32 a[S];
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const c32 e = (s / H == SET) ? v[s % H][t] : sm[((OTHER * H + s % H) * 8 + tl) * BL + tid];
                    if (s == 0) { a[0] = e; continue; }
                    const int m = (s * c) % CIRC;
                    const c32 k = mk(circ_cos(m, CIRC), SIGN < 0 ? -circ_sin(m, CIRC) : circ_sin(m, CIRC));
                    a[s] = cmul(e, (m == 0) ? wb[s - 1] : cmul(wb[s - 1], k));
                }
                bfly<S, SIGN>(a);  // slot m holds y[orev<8>(m)]
#pragma unroll
                for (int m = 0; m < S; m++) {
                    f2v o;
                    o.x = a[m].x;
                    o.y = a[m].y;
                    __builtin_nontemporal_store(o, out_f + c * BL + ((orev<S>(m) ^ m_xor) * NS));
                }
            }
        };
        if (set == 0) finish(std::integral_constant<int, 0>{});
        else finish(std::integral_constant<int, 1>{});
        __syncthreads();  // the exchange image is read before the next frame's transforms overwrite it
    }
}

template <int N, int SIGN, bool REAL>
__global__ __launch_bounds__(256, (N / 4096 == 8 ? 1 : 2)) void k_fft_s(const void *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ window,
                                                  const c32 *__restrict__ twN, int nframes, int shift)
{
    constexpr int NS = 4096, S = N / NS, BL = 256;
    // S = 8 (32768 points): 128 complex values per thread -- one workgroup (one wave per SIMD, up to 512 registers) per CU, the
    // whole 256 KiB frame of loads in flight at once; still ONE pass over HBM instead of the two of the workspace scheme
    static_assert(S == 2 || S == 4 || S == 8, "two, four or eight sub-transforms");
    using G = Geo<NS>;
    __shared__ c32 lds[NS];
    const int tid0 = threadIdx.x;
    constexpr bool RELOAD = S == 4;  // 16384: 128 data registers -- the twiddles are re-read (L1/L2) per frame instead of spilling
    TwRegs<NS> tw;
    c32 wb[S - 1];
    if constexpr (!RELOAD) {
        load_twiddles<NS, false, G>(tw, tid0, twN + N);  // the 4096-point table follows the N-point one
#pragma unroll
        for (int s = 1; s < S; s++) wb[s - 1] = twN[(s * tid0) & (N - 1)];
    }
    const int in_xor = (SIGN > 0 && shift) ? 8 : 0;       // reverse: halves swapped on load == n ^ 2048 == r ^ 8
    const int m_xor = (SIGN < 0 && shift) ? (S / 2) : 0;  // forward: halves swapped on store

    for (int frame = blockIdx.x; frame < nframes; frame += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        c32 v[S][16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const unsigned n = (unsigned)(tid + ((r ^ in_xor) * BL));  // element n of every sub-frame
            float w[S];
            if constexpr (S == 2) {
                const f2v t = *((const f2v *)window + n);
                w[0] = t.x; w[1] = t.y;
            } else {
#pragma unroll
                for (int h = 0; h < S / 4; h++) {
                    const f4v t = *((const f4v *)window + (size_t)n * (S / 4) + h);
                    w[4 * h] = t.x; w[4 * h + 1] = t.y; w[4 * h + 2] = t.z; w[4 * h + 3] = t.w;
                }
            }
            if constexpr (REAL) {
                float x[S];
                if constexpr (S == 2) {
                    const f2v t = __builtin_nontemporal_load((const f2v *)in + (size_t)frame * (N / 2) + n);
                    x[0] = t.x; x[1] = t.y;
                } else {
#pragma unroll
                    for (int h = 0; h < S / 4; h++) {
                        const f4v t = __builtin_nontemporal_load((const f4v *)in + (size_t)frame * (N / 4) + (size_t)n * (S / 4) + h);
                        x[4 * h] = t.x; x[4 * h + 1] = t.y; x[4 * h + 2] = t.z; x[4 * h + 3] = t.w;
                    }
                }
#pragma unroll
                for (int s = 0; s < S; s++) v[s][r] = mk(x[s] * w[s], 0.f);
            } else {
#pragma unroll
                for (int h = 0; h < S / 2; h++) {
                    const f4v *src = (const f4v *)in + (size_t)frame * (N / 2) + (size_t)n * (S / 2) + h;
                    const f4v t = (S > 2) ? *src : __builtin_nontemporal_load(src);  // S > 2: a line is touched by S/2 load instructions
                    v[2 * h][r] = mk(t.x * w[2 * h], t.y * w[2 * h]);
                    v[2 * h + 1][r] = mk(t.z * w[2 * h + 1], t.w * w[2 * h + 1]);
                }
            }
        }
        if constexpr (RELOAD) load_twiddles<NS, false, G>(tw, tid, twN + N);
        // (spelled out: eight inlined transforms exceed the optimiser's pragma-unroll size limit, and a rolled loop would index
        // v[] dynamically, i.e. move the whole frame to scratch memory)
#define MI355_SUBT(K)                                                                                                  \
        if constexpr (S > K) {                                                                                         \
            transform_regs<NS, SIGN, false, G>(v[K], tw, lds, tid);                                                    \
            __syncthreads(); /* the last pass' LDS reads are done before the next sub-transform writes */              \
        }
        MI355_SUBT(0) MI355_SUBT(1) MI355_SUBT(2) MI355_SUBT(3) MI355_SUBT(4) MI355_SUBT(5) MI355_SUBT(6) MI355_SUBT(7)
#undef MI355_SUBT
        if constexpr (RELOAD) {
#pragma unroll
            for (int s = 1; s < S; s++) wb[s - 1] = twN[(s * tid) & (N - 1)];
        }
        f2v *__restrict__ out_f = (f2v *)out + (size_t)frame * N + tid;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            constexpr int CIRC = 16 * S;  // W_N^(s * c * 256) = exp(sign 2 pi i s c / (16 S)): a point of the (16 S)-point circle
            const int c = orev<16>(t);
            c32 a[S];
            a[0] = v[0][t];
#pragma unroll
            for (int s = 1; s < S; s++) {
                const int m = (s * c) % CIRC;
                const c32 k = mk(circ_cos(m, CIRC), SIGN < 0 ? -circ_sin(m, CIRC) : circ_sin(m, CIRC));
                a[s] = cmul(v[s][t], (m == 0) ? wb[s - 1] : cmul(wb[s - 1], k));
            }
            bfly<S, SIGN>(a);  // slot m holds y[orev<S>(m)] (identity for S = 2, 4)
#pragma unroll
            for (int m = 0; m < S; m++) {
                f2v o;
                o.x = a[m].x;
                o.y = a[m].y;
                __builtin_nontemporal_store(o, out_f + c * BL + ((orev<S>(m) ^ m_xor) * NS));
            }
        }
    }
}

template <int N>
int launch_s(mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, const void *tw, int nframes, int shift,
             int real_in, hipStream_t st)
{
    // interleaved A/B (tools/probe.py ab fft<N>, 2^26 samples): 16384 points 229 us with the two resident workgroups per CU as the grid,
    // 217 with 8 per CU, 206 with 12-16 (one or two frames per workgroup: nothing but the base twiddles is kept across frames, so short
    // workgroups cost nothing and even out the CUs); 8192 points 189.5 -> 185.4 us at 4 per CU; 32768 is flat (one workgroup fills a CU)
    const int grid = N == 32768   ? mi355_balanced_grid(ctx, nframes, 1, 1)
                     : N == 16384 ? mi355_balanced_grid(ctx, nframes, 16, 16)
                                  : mi355_balanced_grid(ctx, nframes, 4, 4);
    if constexpr (N == 32768) {
        if (!getenv("MI355_FFT_32768_IN_REGISTERS")) {
#define LAUNCH_32K(SG, RL) hipLaunchKernelGGL((k_fft_32k<SG, RL>), dim3(grid), dim3(512), 0, st, in, (c32 *)out, window, (const c32 *)tw, nframes, shift)
            if (sign < 0) { if (real_in) LAUNCH_32K(-1, true); else LAUNCH_32K(-1, false); }
            else          { if (real_in) LAUNCH_32K(1, true);  else LAUNCH_32K(1, false); }
#undef LAUNCH_32K
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
#define LAUNCH_S(SG, RL) hipLaunchKernelGGL((k_fft_s<N, SG, RL>), dim3(grid), dim3(256), 0, st, in, (c32 *)out, window, (const c32 *)tw, nframes, shift)
    if (sign < 0) { if (real_in) LAUNCH_S(-1, true); else LAUNCH_S(-1, false); }
    else          { if (real_in) LAUNCH_S(1, true);  else LAUNCH_S(1, false); }
#undef LAUNCH_S
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// ------------------------------------------------------------------------------------
// N = 32768 / 65536: two kernels through an N-point workspace (decimation in time, S = N/4096 = 8 / 16 sub-frames).
//   k_fft_sub:      E_s = FFT_4096( x[S n + s] * w[S n + s] ), one workgroup iteration per sub-frame, written k-contiguous
//   k_fft_combine:  X[k + 4096 m] = sum_s W_N^(s k) W_S^(s m) E_s[k]: lane = k, one radix-S butterfly per lane in registers
// Traffic is twice the single-kernel minimum (the price of frames that do not fit a workgroup's LDS and registers).
// ------------------------------------------------------------------------------------
template <int SIGN, bool REAL>
__global__ __launch_bounds__(256, 2) void k_fft_sub(const void *__restrict__ in, c32 *__restrict__ ws, const float *__restrict__ window,
                                                    const c32 *__restrict__ tw4096, int S, int nquad /* frames * S / 4 */, int in_xor)
{
    // one iteration = FOUR neighbouring sub-frames (s .. s+3): their elements x[S n + s .. s + 3] are adjacent, so every lane
    // moves 32 bytes of each line it touches (pairs, 16 bytes: 226 us per 2^25 samples -- the L2 request rate of partial-line
    // reads, the same wall as the X-engine's column slices; 128 data registers = the budget of k_fft_s<16384>)
    constexpr int NS = 4096, BL = 256, Q = 4;
    using G = Geo<NS>;
    __shared__ c32 lds[NS];
    const int tid0 = threadIdx.x;
    // The S/4 iterations of one frame each touch EVERY 128-byte line of the frame (32 bytes of every S*8), so they must
    // share an L2: workgroup id % 8 is the XCD, and frame f is given to XCD f % 8, whose workgroups walk its quads together
    // (the grid is a multiple of 8).  Unmapped, every XCD fetched every frame from HBM.
    const int per = S / Q, nframes = nquad / per;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    for (int q = blockIdx.x >> 3;; q += per_xcd) {
        const int frame = (q / per) * 8 + xcd, s = Q * (q % per);
        if (q / per >= (nframes + 7) / 8) break;
        if (frame >= nframes) continue;
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const size_t fb = (size_t)frame * NS * S;
        c32 v[Q][16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = tid + ((r ^ in_xor) * BL);  // reverse + shift: n ^ 2048
            const size_t e = (size_t)n * S + s;       // x[S n + s]; e is a multiple of 4
            const f4v w = *((const f4v *)window + (size_t)(s >> 2) * NS + n);  // quad-major window (see the plan)
            if constexpr (REAL) {
                const f4v x = *(const f4v *)((const float *)in + fb + e);  // plain loads: the line is shared with the other quads of the frame
                v[0][r] = mk(x.x * w.x, 0.f);
                v[1][r] = mk(x.y * w.y, 0.f);
                v[2][r] = mk(x.z * w.z, 0.f);
                v[3][r] = mk(x.w * w.w, 0.f);
            } else {
                const f4v x0 = *(const f4v *)((const c32 *)in + fb + e);  // plain loads: a nontemporal one drops the line the other quads share
                const f4v x1 = *((const f4v *)((const c32 *)in + fb + e) + 1);
                v[0][r] = mk(x0.x * w.x, x0.y * w.x);
                v[1][r] = mk(x0.z * w.y, x0.w * w.y);
                v[2][r] = mk(x1.x * w.z, x1.y * w.z);
                v[3][r] = mk(x1.z * w.w, x1.w * w.w);
            }
        }
        TwRegs<NS> tw;  // re-read (L1/L2) per iteration, after the frame's loads are issued: 128 data registers leave no room to keep them
        load_twiddles<NS, false, G>(tw, tid, tw4096);
#define MI355_SUBQ(K)                                                                                                   \
        {                                                                                                              \
            transform_regs<NS, SIGN, false, G>(v[K], tw, lds, tid);                                                    \
            f2v *__restrict__ o = (f2v *)ws + ((size_t)frame * S + s + K) * NS + tid;                                  \
            _Pragma("unroll") for (int t = 0; t < 16; t++) {                                                           \
                f2v z;                                                                                                 \
                z.x = v[K][t].x;                                                                                       \
                z.y = v[K][t].y;                                                                                       \
                o[orev<16>(t) * BL] = z; /* plain store: the combine kernel reads it back */                           \
            }                                                                                                          \
            __syncthreads();                                                                                           \
        }
        MI355_SUBQ(0) MI355_SUBQ(1) MI355_SUBQ(2) MI355_SUBQ(3)
#undef MI355_SUBQ
    }
}

template <int S, int SIGN>
__global__ __launch_bounds__(256) void k_fft_combine(const c32 *__restrict__ ws, c32 *__restrict__ out, const c32 *__restrict__ twN,
                                                     long long total /* frames * 4096 */, int m_xor)
{
    constexpr int NS = 4096, N = NS * S;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long frame = e / NS;
        const int k = (int)(e - frame * NS);
        const c32 *src = ws + (size_t)frame * N + k;
        c32 a[S];
#pragma unroll
        for (int s = 0; s < S; s++) a[s] = src[(size_t)s * NS];
        c32 wsl[tw_slots<S>()];  // W_N^(p k) for the stored powers; apply_twiddles builds the others with one product
#pragma unroll
        for (int i = 0; i < tw_slots<S>(); i++) wsl[i] = twN[(tw_power<S>(i) * k) & (N - 1)];
        apply_twiddles<S>(a, wsl);
        bfly<S, SIGN>(a);  // slot t holds y[orev<S>(t)]
        f2v *dst = (f2v *)out + (size_t)frame * N + k;
#pragma unroll
        for (int t = 0; t < S; t++) {
            f2v z;
            z.x = a[t].x;
            z.y = a[t].y;
            __builtin_nontemporal_store(z, dst + (size_t)((orev<S>(t) ^ m_xor) * NS));
        }
    }
}

// Powers of two above 65536 (131072 .. 1048576 = 4096 x 16 x S2, S2 = 2 .. 16): one more level of the same decimation in time.
//   sub-frame s = S2 a + b  =>  X[k] = sum_b W_N^(b k) G_b[k mod 65536],  G_b = 65536-point transform of x[S2 m + b]
//                                  = radix-16 combine of the sub-frames E_(S2 a + b), a = 0 .. 15
// k_fft_sub (all S sub-frames) -> workspace A;  k_fft_combine2<16> per b (A -> B: G_b);  k_fft_combine2<S2> (B -> out).
// Three passes over the frame instead of two: functional coverage of the sizes the reference's clFFT plans accept
// (lib/clFFT_impl.cc:91-128), one item of such a stream being 1 .. 8 MiB.
// One generalised combine: `nsub` points per input sub-transform, S of them `src_sub` elements apart (blockIdx.y selects one
// of gridDim.y interleaved sets, `src_set` / `dst_set` elements apart); W_(nsub S)^(s k) = twN[(s k tw_mul) & nmask].
template <int S, int SIGN>
__global__ __launch_bounds__(256) void k_fft_combine2(const c32 *__restrict__ src, c32 *__restrict__ dst, const c32 *__restrict__ twN,
                                                      long long total /* frames * nsub */, int nsub, long long frame_elems, long long src_sub,
                                                      long long src_set, long long dst_set, int tw_mul, int nmask, int m_xor)
{
    const int set = blockIdx.y;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long frame = e / nsub;
        const int k = (int)(e - frame * nsub);
        const c32 *in = src + (size_t)frame * frame_elems + (size_t)set * src_set + k;
        c32 a[S];
#pragma unroll
        for (int s = 0; s < S; s++) a[s] = in[(size_t)s * src_sub];
        c32 wsl[tw_slots<S>()];
#pragma unroll
        for (int i = 0; i < tw_slots<S>(); i++) wsl[i] = twN[(int)(((long long)tw_power<S>(i) * k * tw_mul) & nmask)];
        apply_twiddles<S>(a, wsl);
        bfly<S, SIGN>(a);  // slot t holds y[orev<S>(t)]
        f2v *out = (f2v *)dst + (size_t)frame * frame_elems + (size_t)set * dst_set + k;
#pragma unroll
        for (int t = 0; t < S; t++) {
            f2v z;
            z.x = a[t].x;
            z.y = a[t].y;
            out[(size_t)((orev<S>(t) ^ m_xor)) * nsub] = z;
        }
    }
}

// ------------------------------------------------------------------------------------
// 65536 .. 1048576 points in TWO passes over whole 128-byte lines (N = N1 x N2, both 256 .. 1024; n = n1 N2 + n2, k = k2 N1 + k1):
//   X[k2 N1 + k1] = sum_n2 W_N2^(n2 k2) { W_N^(n2 k1) sum_n1 x[n1 N2 + n2] W_N1^(n1 k1) }
// Both passes are the same kernel: a workgroup takes a TILE of sixteen neighbouring columns (16 x 8 B = one 128-byte line per
// row) of a matrix with NR rows, transforms the sixteen columns (NR-point transforms, the frame machinery of fft_core with sixteen
// frames per workgroup) and stores
//   pass A (x as N1 rows of N2):   column n2 as ROW n2 of the workspace, times W_N^(n2 k1)      -> ws[n2][k1], k1-contiguous stores
//   pass B (ws as N2 rows of N1):  back into the tile's own columns of the output                -> out[k2][k1]
// The decimation-in-time form before this (k_fft_sub / k_fft_combine: 4096-point sub-transforms of x[S n + s]) reads 32 bytes of every
// 128-byte line per workgroup -- the L2 request rate of partial lines -- and needs three passes above 65536 points; here every
// global access of both passes is a whole line and 2^20 points take two passes instead of three.
// The tile goes through LDS once on the way in (lanes run along the columns for the global access, along the rows for the
// transform; column stride NR + 1 slots keeps both sides conflict free) and, in pass B, once on the way out.
// ------------------------------------------------------------------------------------
template <int NR> struct GeoTile {
    static constexpr int TH = NR, PTS = 16 * NR, F = 16, WPE = 1;
};

template <int NR, int SIGN, bool PASS_B, bool REAL, bool WR>
__global__ __launch_bounds__(NR, 4) void k_fft_tile(const void *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ window,
                                                 const c32 *__restrict__ twN, const c32 *__restrict__ twR,  // W_N (N entries), W_NR (NR entries)
                                                 int ld /* columns of the matrix = elements per row */, int nmask /* N - 1 */,
                                                 long long nitems /* frames x tiles per frame */, int row_xor)
{
    using G = GeoTile<NR>;
    // pass A runs the reversed radix plan (remainder radix first): its LAST pass is then radix 16 for every NR, i.e. a thread ends up
    // with k1 = j + 16-step multiples of NR / 16 -- one twiddle base and one step per thread (see below) instead of one table gather per value
    constexpr bool REV = !PASS_B;
    using PL = Plan<NR, REV>;
    constexpr int TH = NR, CS = NR + 1;  // staging: column c at slots [c * CS, c * CS + NR)
    extern __shared__ __attribute__((aligned(16))) c32 tile_lds[];
    const int tid0 = threadIdx.x;
    TwRegs<NR> tw;
    load_twiddles<NR, REV, G>(tw, tid0, twR);
    const int tiles = ld / 16;
    const size_t frame_elems = (size_t)NR * ld;
    // (fetching the next item's tile into registers while the current one is transformed was measured: slower at every NR -- the
    // 1024-thread workgroups have 128 registers per thread and spill, the smaller ones lose more occupancy than they gain)
    // pass A's window values depend on the tile's place in the frame only: the grid is a multiple of the tiles per frame whenever it can be
    // (launch_tile), so a workgroup meets the same tile in every frame and these sixteen registers are loaded once per launch, not once per
    // item (a 4-byte load per value from L2 was 12 % of the 65536-point transform)
    // (NR = 1024: sixteen waves leave 128 registers per thread and the sixteen values spill -- that size keeps the load per value)
    constexpr bool WREG = !PASS_B && WR;
    float wreg[16];
    int c0_window = -1;
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long long frame = item / tiles;
        const int c0 = (int)(item - frame * tiles) * 16;
        if constexpr (WREG) {
            if (c0 != c0_window) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int e = tid + TH * i, col = e & 15, row = e >> 4;
                    wreg[i] = window[(size_t)(row ^ row_xor) * ld + c0 + col];  // indexed by the ORIGINAL position, like the reference (lib/clFFT_impl.cc:477-493)
                }
                c0_window = c0;
            }
        }
        // ---- tile in: lanes along the columns (16 lanes = one 128-byte line of a row) ----
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + TH * i, col = e & 15, row = e >> 4;
            const size_t idx = (size_t)(PASS_B ? row : (row ^ row_xor)) * ld + c0 + col;  // pass A, reverse + shift: the halves of the frame are swapped on load
            c32 x;
            if constexpr (REAL) x = mk(((const float *)in)[(size_t)frame * frame_elems + idx], 0.f);
            else {
                const f2v t = __builtin_nontemporal_load((const f2v *)in + (size_t)frame * frame_elems + idx);
                x = mk(t.x, t.y);
            }
            if constexpr (!PASS_B) {
                const float w = WREG ? wreg[i] : window[idx];
                x = mk(x.x * w, x.y * w);
            }
            tile_lds[col * CS + row] = x;
        }
        __syncthreads();
        // ---- registers in the layout transform_regs expects: v[q R0 + r] = column fr, row j + r B0 (g = tid + TH q, fr = g / B0, j = g % B0) ----
        c32 v[16];
        constexpr int R0 = PL::radix(0), B0 = NR / R0;
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int g = tid + TH * q, fr = g / B0, j = g % B0;
#pragma unroll
            for (int r = 0; r < R0; r++) v[q * R0 + r] = tile_lds[fr * CS + j + r * B0];
        }
        __syncthreads();  // the staged tile has been read: transform_regs works in the same memory
        transform_regs<NR, SIGN, REV, G>(v, tw, tile_lds, tid);
        constexpr int NP = PL::NP, RL = PL::radix(NP - 1), BL = NR / RL;
        if constexpr (!PASS_B) {
            // ---- pass A: column n2 = c0 + fr becomes row n2 of the workspace, times W_N^(n2 k1); lanes run along k1 ----
            static_assert(RL == 16, "reversed plan: the last pass is radix 16");
            const int fr = tid / BL, j = tid % BL, n2 = c0 + fr;
            // k1 = j + m BL, m = 0 .. 15:  W_N^(n2 k1) = W_N^(n2 j) (W_N^(n2 BL))^m -- two table reads per thread and tile; the sixteen
            // powers by squaring (at most four products deep).  A gather per value (sixteen per thread, 64 distinct lines per wave
            // instruction) made this pass run at 1.8 - 2.9 TB/s.
            const c32 base = twN[(int)(((long long)n2 * j) & nmask)], g1 = twN[(int)(((long long)n2 * BL) & nmask)];
            const c32 g2 = cmul(g1, g1), g4 = cmul(g2, g2), g8 = cmul(g4, g4);
            c32 gp[16];
            gp[0] = base;
            gp[1] = cmul(base, g1);
            gp[2] = cmul(base, g2);
            gp[3] = cmul(gp[2], g1);
            gp[4] = cmul(base, g4);
            gp[5] = cmul(gp[4], g1);
            gp[6] = cmul(gp[4], g2);
            gp[7] = cmul(gp[6], g1);
#pragma unroll
            for (int m = 0; m < 8; m++) gp[8 + m] = cmul(gp[m], g8);
            f2v *__restrict__ o = (f2v *)out + (size_t)frame * frame_elems + (size_t)n2 * NR + j;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const c32 z = cmul(v[t], gp[orev<16>(t)]);
                f2v zz;
                zz.x = z.x;
                zz.y = z.y;
                o[orev<16>(t) * BL] = zz;  // plain store: pass B reads it back
            }
            __syncthreads();  // the next item's tile overwrites the memory the last pass of the transform read
        } else {
            // ---- pass B: back into the tile's own columns; lanes along the columns again ----
            __syncthreads();  // every thread has finished the last pass' LDS reads
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, fr = g / BL, j = g % BL;
#pragma unroll
                for (int t = 0; t < RL; t++) tile_lds[fr * CS + j + orev<RL>(t) * BL] = v[q * RL + t];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = tid + TH * i, col = e & 15, row = e >> 4;
                const c32 z = tile_lds[col * CS + row];
                f2v zz;
                zz.x = z.x;
                zz.y = z.y;
                __builtin_nontemporal_store(zz, (f2v *)out + (size_t)frame * frame_elems + (size_t)(row ^ row_xor) * ld + c0 + col);  // forward + shift: halves swapped on store
            }
            __syncthreads();
        }
    }
}

// The same two passes with HALF the threads (NR / 2: thread t works on column t / (NR/16) of the tile's first eight columns and on the
// same column of the last eight, two transforms in lockstep) -- sixteen waves of 128 registers become eight of 256, which leaves room to
// fetch the NEXT tile into registers while this one is transformed: a 1024-row tile fills the CU's LDS, so no second workgroup can
// cover the loads, and without this the pass alternates between loading and computing.
template <int NR> struct GeoTileH {
    static constexpr int TH = NR / 2, PTS = 8 * NR, F = 8, WPE = 1;
};

template <int NR, int SIGN, bool PASS_B, bool REAL>
__global__ __launch_bounds__(NR / 2) void k_fft_tile_h(const void *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ window,
                                                       const c32 *__restrict__ twN, const c32 *__restrict__ twR, int ld, int nmask,
                                                       long long nitems, int row_xor)
{
    using G = GeoTileH<NR>;
    constexpr bool REV = !PASS_B;
    using PL = Plan<NR, REV>;
    constexpr int TH = NR / 2, CS = NR + 1;
    extern __shared__ __attribute__((aligned(16))) c32 tile_lds[];
    const int tid0 = threadIdx.x;
    TwRegs<NR> tw;
    load_twiddles<NR, REV, G>(tw, tid0, twR);
    const int tiles = ld / 16;
    const size_t frame_elems = (size_t)NR * ld;
    // element i of a thread: e = tid + TH i (i < 32), column e & 15, row e >> 4
    auto fetch = [&](long long item, c32 (&pf)[32], int tid) {
        const long long frame = item / tiles;
        const int c0 = (int)(item - frame * tiles) * 16;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int e = tid + TH * i, col = e & 15, row = e >> 4;
            const size_t idx = (size_t)(PASS_B ? row : (row ^ row_xor)) * ld + c0 + col;
            if constexpr (REAL) pf[i] = mk(((const float *)in)[(size_t)frame * frame_elems + idx], 0.f);
            else {
                const f2v t = __builtin_nontemporal_load((const f2v *)in + (size_t)frame * frame_elems + idx);
                pf[i] = mk(t.x, t.y);
            }
        }
    };
    c32 pf[32];
    if ((long long)blockIdx.x < nitems) fetch(blockIdx.x, pf, tid0);
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long long frame = item / tiles;
        const int c0 = (int)(item - frame * tiles) * 16;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int e = tid + TH * i, col = e & 15, row = e >> 4;
            c32 x = pf[i];
            if constexpr (!PASS_B) {
                const float w = window[(size_t)(row ^ row_xor) * ld + c0 + col];  // the ORIGINAL position (lib/clFFT_impl.cc:477-493)
                x = mk(x.x * w, x.y * w);
            }
            tile_lds[col * CS + row] = x;
        }
        __syncthreads();
        c32 va[16], vb[16];
        constexpr int R0 = PL::radix(0), B0 = NR / R0;
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int g = tid + TH * q, fr = g / B0, j = g % B0;
#pragma unroll
            for (int r = 0; r < R0; r++) {
                va[q * R0 + r] = tile_lds[fr * CS + j + r * B0];
                vb[q * R0 + r] = tile_lds[(fr + 8) * CS + j + r * B0];
            }
        }
        __syncthreads();
        if (item + gridDim.x < nitems) fetch(item + gridDim.x, pf, tid);  // in flight under the transform and the stores
        transform_regs2<NR, SIGN, REV, G>(va, vb, tw, tile_lds, tile_lds + 8 * NR, tid);
        constexpr int NP = PL::NP, RL = PL::radix(NP - 1), BL = NR / RL;
        if constexpr (!PASS_B) {
            static_assert(RL == 16, "reversed plan: the last pass is radix 16");
            const int fr = tid / BL, j = tid % BL;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int n2 = c0 + fr + 8 * h;
                const c32 base = twN[(int)(((long long)n2 * j) & nmask)], g1 = twN[(int)(((long long)n2 * BL) & nmask)];
                const c32 g2 = cmul(g1, g1), g4 = cmul(g2, g2), g8 = cmul(g4, g4);
                c32 gp[16];
                gp[0] = base;
                gp[1] = cmul(base, g1);
                gp[2] = cmul(base, g2);
                gp[3] = cmul(gp[2], g1);
                gp[4] = cmul(base, g4);
                gp[5] = cmul(gp[4], g1);
                gp[6] = cmul(gp[4], g2);
                gp[7] = cmul(gp[6], g1);
#pragma unroll
                for (int m = 0; m < 8; m++) gp[8 + m] = cmul(gp[m], g8);
                f2v *__restrict__ o = (f2v *)out + (size_t)frame * frame_elems + (size_t)n2 * NR + j;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const c32 z = cmul(h ? vb[t] : va[t], gp[orev<16>(t)]);
                    f2v zz;
                    zz.x = z.x;
                    zz.y = z.y;
                    o[orev<16>(t) * BL] = zz;
                }
            }
            __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int g = tid + TH * q, fr = g / BL, j = g % BL;
#pragma unroll
                for (int t = 0; t < RL; t++) {
                    tile_lds[fr * CS + j + orev<RL>(t) * BL] = va[q * RL + t];
                    tile_lds[(fr + 8) * CS + j + orev<RL>(t) * BL] = vb[q * RL + t];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const int e = tid + TH * i, col = e & 15, row = e >> 4;
                const c32 z = tile_lds[col * CS + row];
                f2v zz;
                zz.x = z.x;
                zz.y = z.y;
                __builtin_nontemporal_store(zz, (f2v *)out + (size_t)frame * frame_elems + (size_t)(row ^ row_xor) * ld + c0 + col);
            }
            __syncthreads();
        }
    }
}

template <int N, class G>
int launch_g(mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, const void *tw, int nframes, int shift,
             int real_in, hipStream_t st)
{
    constexpr int F = G::F, TH = G::TH, WAVES = TH / 64;
    int ngroups = (nframes + F - 1) / F;
    int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    // MI355_FFT_PREFETCH=1 selects the register-prefetch variant (2 workgroups/CU); measured equal to the
    // plain variant at 3 workgroups/CU on MI355X, so it is off by default.
    int pf = getenv("MI355_FFT_PREFETCH") ? atoi(getenv("MI355_FFT_PREFETCH")) : -1;  // (per call: tuning switch)
    // N <= 4096, 256-thread workgroups.  Measured on MI355X with interleaved A/B runs inside one process (tools/probe_fft_ab.py; the
    // drift between processes is +-4 %): EXACTLY two persistent workgroups per CU with the next group's loads issued before the
    // current group is transformed (PF = 1) beat every other residency by 3-4 % at N = 16 ... 4096 (4096: 178.8 -> 173.6 us per
    // GiB of traffic); 3 or 4 per CU with the same prefetch are 5-10 % SLOWER, a second group of lead (PF = 2, 190 registers) too.
    // Without the prefetch many short workgroups (8-16 per CU, each still looping over >= 4 groups with its twiddles and window in
    // registers) are best: that remains the schedule of calls too small to give every persistent workgroup >= 8 groups.
    const bool persistent2 = N <= 4096 && WAVES == 4 && ngroups >= cus * 2 * 8;
    if (pf < 0) pf = persistent2 ? 1 : 0;
    int grid = (N <= 4096) ? mi355_balanced_grid(ctx, ngroups, 32 / WAVES, 64 / WAVES, 0.015) : mi355_balanced_grid(ctx, ngroups, 1, 1);
    if (pf == 1 && persistent2) grid = cus * 2;
    if (const char *e = getenv("MI355_FFT_WG_PER_CU")) {
        if (atoi(e) > 0) grid = ngroups < cus * atoi(e) ? ngroups : cus * atoi(e);
    }
#define LAUNCH_FFT(SG, RL)                                                                                              \
    do {                                                                                                                     \
        if (pf == 2)                                                                                                         \
            hipLaunchKernelGGL((k_fft<N, SG, RL, 2, G>), dim3(grid), dim3(TH), 0, st, in, (c32 *)out, window, (const c32 *)tw, \
                               nframes, ngroups, shift);                                                                     \
        else if (pf == 1)                                                                                                    \
            hipLaunchKernelGGL((k_fft<N, SG, RL, 1, G>), dim3(grid), dim3(TH), 0, st, in, (c32 *)out, window, (const c32 *)tw, \
                               nframes, ngroups, shift);                                                                     \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_fft<N, SG, RL, 0, G>), dim3(grid), dim3(TH), 0, st, in, (c32 *)out, window,                \
                               (const c32 *)tw, nframes, ngroups, shift);                                                    \
    } while (0)
    if (sign < 0) { if (real_in) LAUNCH_FFT(-1, true); else LAUNCH_FFT(-1, false); }
    else          { if (real_in) LAUNCH_FFT(1, true);  else LAUNCH_FFT(1, false); }
#undef LAUNCH_FFT
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

template <int N>
int launch_n(mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, const void *tw, int nframes, int shift,
             int real_in, hipStream_t st)
{
    if constexpr (N >= 16 && N <= 1024) {
        // MI355_FFT_WAVE_GEO=1: one-wave workgroups (no workgroup barriers); measured slower for N >= 256, off by default
        static const bool wave = getenv("MI355_FFT_WAVE_GEO") ? atoi(getenv("MI355_FFT_WAVE_GEO")) != 0 : false;
        if (wave) return launch_g<N, GeoW<N>>(ctx, sign, in, out, window, tw, nframes, shift, real_in, st);
    }
    if constexpr (N == 32768) return launch_s<N>(ctx, sign, in, out, window, tw, nframes, shift, real_in, st);
    else if constexpr (N == 8192 || N == 16384) {
        // MI355_FFT_WHOLE_FRAME=1 selects the whole-frame kernel (512/1024 threads, frame image in LDS) for comparison
        static const bool whole = getenv("MI355_FFT_WHOLE_FRAME") ? atoi(getenv("MI355_FFT_WHOLE_FRAME")) != 0 : false;
        if (!whole) return launch_s<N>(ctx, sign, in, out, window, tw, nframes, shift, real_in, st);
    }
    if constexpr (N != 32768) return launch_g<N, Geo<N>>(ctx, sign, in, out, window, tw, nframes, shift, real_in, st);
}

int launch_fft(mi355_ctx *ctx, int n, int sign, const void *in, void *out, const float *window, const void *tw, int nframes,
               int shift, int real_in, hipStream_t st)
{
    switch (n) {
#define CASE_N(NN) case NN: return launch_n<NN>(ctx, sign, in, out, window, tw, nframes, shift, real_in, st)
        CASE_N(2); CASE_N(4); CASE_N(8); CASE_N(16); CASE_N(32); CASE_N(64); CASE_N(128); CASE_N(256); CASE_N(512);
        CASE_N(1024); CASE_N(2048); CASE_N(4096); CASE_N(8192); CASE_N(16384); CASE_N(32768);
#undef CASE_N
    }
    mi355_set_error("fft size %d unsupported (power of two, 2..32768)", n);
    return MI355_ERR_UNSUPPORTED;
}

}  // namespace

struct mi355_fft {
    mi355_ctx *ctx;
    int n, sign, dtype, nstreams, shift;
    float *d_window;  // n floats or NULL
    void *d_tw;       // n complex: exp(sign*2*pi*i*k/n), generated in double
    HostPipe pipe;
    // lengths 2^a 3^b 5^c 7^d that a workgroup holds: the mixed-radix kernel (fft_mr.hip); mr.n == 0: not used
    MrPlan mr;
    MrTilePlan mrt;  // the same lengths above 15360 points: two passes (mrt.n == 0: not used); the workspace is d_wa
    // every other size that is not a power of two: chirp-z (Bluestein) over power-of-two transforms of size m
    int m = 0;
    void *d_pre = nullptr, *d_post = nullptr, *d_bspec = nullptr;  // window*chirp (n), chirp (n), spectrum of the conjugate chirp / m (m)
    void *d_twm_f = nullptr, *d_twm_i = nullptr;                    // twiddle tables of the size-m forward / inverse transforms
    float *d_ones = nullptr;                                        // all-ones window of size m
    mi355_fft *sub_f = nullptr, *sub_i = nullptr;                   // m > 32768: the size-m transforms are handles of their own (multi-pass sizes)
    void *d_wa = nullptr, *d_wb = nullptr;                          // work buffers, cap_frames * m complex each
    size_t cap_frames = 0;
    // The two-kernel sizes (> 16384 points) and the unfused chirp-z path go through ONE workspace per handle.  Calls on
    // different streams / threads are serialised on it: the lock covers the enqueue, ws_done orders the streams (the next
    // user's stream waits for the previous user's kernels) and guards the re-allocation.
    bool two_kernel = false;  // workspace scheme (65536 points; 32768 only with MI355_FFT_32768_TWO_KERNELS)
    int tile_n1 = 0, tile_n2 = 0;  // 65536 .. 1048576 points: the two-pass tile scheme N = n1 x n2 (k_fft_tile); 0 = not used
    std::mutex ws_lock;
    hipEvent_t ws_done = nullptr;
    bool ws_used = false;
};

namespace {

// ---- chirp-z (Bluestein) for sizes that are not a power of two --------------------------------------------------
//   X[k] = a[k] * sum_n (x[n] a[n]) conj(a)[k-n],  a[n] = exp(sign * i*pi*n^2/N)
// evaluated as a circular convolution of size m = 2^ceil(log2(2N-1)) with the power-of-two kernels above:
//   pre (window, input shift and chirp folded into one table) -> FFT_m -> x spectrum of conj(a) (pre-scaled 1/m)
//   -> inverse FFT_m -> post (chirp, output shift).  Functional coverage, not a roofline path: ~7 passes over m-point frames.
__global__ __launch_bounds__(256) void k_blu_pre(const void *__restrict__ in, c32 *__restrict__ A, const c32 *__restrict__ pre,
                                                 const int *__restrict__ src, int N, int M, long long total, int real_in)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long f = e / M;
        const int m = (int)(e - f * M);
        c32 v = mk(0.f, 0.f);
        if (m < N) {
            const int i = src[m];  // original input position that lands on transform input m (reverse + shift swaps the halves)
            const c32 x = real_in ? mk(((const float *)in)[f * N + i], 0.f) : ((const c32 *)in)[f * N + i];
            v = cmul(x, pre[m]);
        }
        A[e] = v;
    }
}

__global__ __launch_bounds__(256) void k_blu_mul(c32 *__restrict__ B, const c32 *__restrict__ bspec, int M, long long total)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
        B[e] = cmul(B[e], bspec[e & (M - 1)]);
}

__global__ __launch_bounds__(256) void k_blu_post(const c32 *__restrict__ A, c32 *__restrict__ out, const c32 *__restrict__ post,
                                                  int N, int M, long long total, int lo)
{
    // lo = 0: out[p] = X[p];  forward + shift: out[p] = X[p + len] for p < N - len, X[p - (N - len)] after (len = ceil(N/2)), lo = len
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long f = e / N;
        const int p = (int)(e - f * N);
        const int k = lo ? (p < N - lo ? p + lo : p - (N - lo)) : p;
        out[e] = cmul(A[f * M + k], post[k]);
    }
}

// Fused chirp-z for m <= 16384 (N <= 8192; round 3: m = 8192 / 16384, 2 x the five-launch path -- 3001 points 26 -> 52 GS/s, 8191 36 -> 56): one kernel per call, HBM traffic = the frame read once and written once.
// Same structure as the overlap-save filter (filter.hip): forward FFT_m, spectrum multiply in registers, inverse FFT_m
// through the reversed radix plan.  The chirp tables are read from L2 at the positions a thread holds (coalesced).
template <int M, bool REAL>
__global__ __launch_bounds__(Geo<M>::TH, Geo<M>::WPE) void k_chirpz(const void *__restrict__ in, c32 *__restrict__ out,
                                                                   const c32 *__restrict__ pre, const int *__restrict__ src,
                                                                   const c32 *__restrict__ post, const c32 *__restrict__ bspec,
                                                                   const c32 *__restrict__ tw_fwd, const c32 *__restrict__ tw_inv,
                                                                   int N, int nframes, int ngroups, int lo)
{
    using G = Geo<M>;
    using PF = Plan<M, false>;
    using PI = Plan<M, true>;
    constexpr int TH = G::TH, F = G::F, NP = PF::NP;
    constexpr int RL = PF::radix(NP - 1), BL = M / RL;
    static_assert(PI::radix(0) == RL, "inverse plan must start with the forward plan's last radix");
    extern __shared__ __attribute__((aligned(16))) c32 chirp_lds[];  // PTS slots (64 / 128 KiB at m = 8192 / 16384)
    c32 *lds = chirp_lds;
    const int tid0 = threadIdx.x;
    constexpr bool SHARE = (PF::L % 4) == 0;  // all-radix-16 sizes: inverse twiddles = conjugates of the forward ones
    TwRegs<M> twf, twi;
    load_twiddles<M, false, G>(twf, tid0, tw_fwd);
    if constexpr (!SHARE) load_twiddles<M, true, G>(twi, tid0, tw_inv);
    c32 Breg[16];  // spectrum of the conjugate chirp (pre-scaled 1/m) at the bins this thread holds after the forward transform
#pragma unroll
    for (int q = 0; q < 16 / RL; q++) {
        const int j = (tid0 + TH * q) % BL;
#pragma unroll
        for (int t = 0; t < RL; t++) Breg[q * RL + t] = bspec[j + orev<RL>(t) * BL];
    }
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int frames_left = nframes - grp * F;
        const size_t fbase = (size_t)grp * F;
        c32 v[16];
        constexpr int R0 = PF::radix(0), B0 = M / R0;
#pragma unroll
        for (int q = 0; q < 16 / R0; q++) {
            const int g = tid + TH * q, fr = g / B0, j = g % B0;
#pragma unroll
            for (int r = 0; r < R0; r++) {
                const int m = j + r * B0;  // transform input position
                const bool ok = m < N && fr < frames_left;
                c32 x = mk(0.f, 0.f);
                if (ok) {
                    const int i = src[m];
                    if constexpr (REAL) x = mk(((const float *)in)[(fbase + fr) * N + i], 0.f);
                    else x = ((const c32 *)in)[(fbase + fr) * N + i];
                    x = cmul(x, pre[m]);
                }
                v[q * R0 + r] = x;
            }
        }
        transform_regs<M, -1, false, G>(v, twf, lds, tid);
        c32 w[16];
#pragma unroll
        for (int q = 0; q < 16 / RL; q++)
#pragma unroll
            for (int r = 0; r < RL; r++) w[q * RL + r] = cmul(v[q * RL + irev<RL>(r)], Breg[q * RL + irev<RL>(r)]);
        if constexpr (NP > 1) __syncthreads();
        if constexpr (SHARE) transform_regs<M, 1, true, G, 0, true>(w, twf, lds, tid);
        else transform_regs<M, 1, true, G>(w, twi, lds, tid);
        constexpr int RO = PI::radix(NP - 1), BO = M / RO;
#pragma unroll
        for (int q = 0; q < 16 / RO; q++) {
            const int g = tid + TH * q, fr = g / BO, j = g % BO;
#pragma unroll
            for (int t = 0; t < RO; t++) {
                const int k = j + orev<RO>(t) * BO;
                if (k < N && fr < frames_left) {
                    // forward + shift: X[k] goes to position k - lo (k >= lo) or k + N - lo  (lo = ceil(N/2); lo = 0: identity)
                    const int p = lo ? (k >= lo ? k - lo : k + (N - lo)) : k;
                    out[(fbase + fr) * N + p] = cmul(w[q * RO + t], post[k]);
                }
            }
        }
        if constexpr (NP > 1) __syncthreads();
    }
}

template <int M>
int launch_chirpz_m(mi355_fft *h, const void *in, void *out, int nframes, hipStream_t st)
{
    constexpr int F = Geo<M>::F, TH = Geo<M>::TH;
    const int ngroups = (nframes + F - 1) / F;
    const int grid = mi355_balanced_grid(h->ctx, ngroups, 2, 3);
    const int lo = (h->sign < 0 && h->shift) ? (h->n + 1) / 2 : 0;
    const int *src = (const int *)((const char *)h->d_post + (size_t)h->n * 8);
    constexpr int lds_bytes = Geo<M>::PTS * 8;
    if (h->dtype == MI355_DTYPE_FLOAT) {
        if (lds_bytes > 64 * 1024) MI355_HIP(hipFuncSetAttribute((const void *)k_chirpz<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        hipLaunchKernelGGL((k_chirpz<M, true>), dim3(grid), dim3(TH), lds_bytes, st, in, (c32 *)out, (const c32 *)h->d_pre, src, (const c32 *)h->d_post,
                           (const c32 *)h->d_bspec, (const c32 *)h->d_twm_f, (const c32 *)h->d_twm_i, h->n, nframes, ngroups, lo);
    } else {
        if (lds_bytes > 64 * 1024) MI355_HIP(hipFuncSetAttribute((const void *)k_chirpz<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        hipLaunchKernelGGL((k_chirpz<M, false>), dim3(grid), dim3(TH), lds_bytes, st, in, (c32 *)out, (const c32 *)h->d_pre, src, (const c32 *)h->d_post,
                           (const c32 *)h->d_bspec, (const c32 *)h->d_twm_f, (const c32 *)h->d_twm_i, h->n, nframes, ngroups, lo);
    }
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// workspace hand-over between streams: call with h->ws_lock held
int ws_acquire(mi355_fft *h, hipStream_t st, bool realloc)
{
    if (!h->ws_done) MI355_HIP(hipEventCreateWithFlags(&h->ws_done, hipEventDisableTiming));
    if (h->ws_used) {
        if (realloc) MI355_HIP(hipEventSynchronize(h->ws_done));  // nobody may still be reading the buffers that are freed
        else MI355_HIP(hipStreamWaitEvent(st, h->ws_done, 0));
    }
    return MI355_OK;
}
int ws_release(mi355_fft *h, hipStream_t st)
{
    MI355_HIP(hipEventRecord(h->ws_done, st));
    h->ws_used = true;
    return MI355_OK;
}

int launch_handle(mi355_fft *h, const void *in, void *out, int nvec, hipStream_t st);

int launch_bluestein(mi355_fft *h, const void *in, void *out, int nframes, hipStream_t st)
{
    const int N = h->n, M = h->m;
    static const bool fused = getenv("MI355_CHIRPZ_FUSED") ? atoi(getenv("MI355_CHIRPZ_FUSED")) != 0 : true;
    if (fused) {
        switch (M) {
        case 256: return launch_chirpz_m<256>(h, in, out, nframes, st);
        case 512: return launch_chirpz_m<512>(h, in, out, nframes, st);
        case 1024: return launch_chirpz_m<1024>(h, in, out, nframes, st);
        case 2048: return launch_chirpz_m<2048>(h, in, out, nframes, st);
        case 4096: return launch_chirpz_m<4096>(h, in, out, nframes, st);
        case 8192: return launch_chirpz_m<8192>(h, in, out, nframes, st);
        case 16384: return launch_chirpz_m<16384>(h, in, out, nframes, st);
        }
    }
    // bound the work buffers to 2 x 128 MiB
    size_t chunk = (128u << 20) / ((size_t)M * 8);
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)nframes) chunk = (size_t)nframes;
    std::lock_guard<std::mutex> ws_guard(h->ws_lock);
    {
        const int rc = ws_acquire(h, st, chunk > h->cap_frames);
        if (rc) return rc;
    }
    if (chunk > h->cap_frames) {
        MI355_HIP(hipStreamSynchronize(st));
        if (h->d_wa) (void)hipFree(h->d_wa);
        if (h->d_wb) (void)hipFree(h->d_wb);
        h->d_wa = h->d_wb = nullptr; h->cap_frames = 0;
        MI355_HIP(hipMalloc(&h->d_wa, chunk * (size_t)M * 8));
        MI355_HIP(hipMalloc(&h->d_wb, chunk * (size_t)M * 8));
        h->cap_frames = chunk;
    }
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    auto grid_for = [&](long long total) { long long b = (total + 255) / 256; return (unsigned)(b < (long long)cus * 16 ? b : (long long)cus * 16); };
    const size_t isz = h->dtype == MI355_DTYPE_FLOAT ? 4 : 8;
    const int len = (N + 1) / 2;
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += chunk) {
        const int nf = (int)((size_t)nframes - f0 < chunk ? (size_t)nframes - f0 : chunk);
        const long long tm = (long long)nf * M, tn = (long long)nf * N;
        hipLaunchKernelGGL(k_blu_pre, dim3(grid_for(tm)), dim3(256), 0, st, (const char *)in + f0 * N * isz, (c32 *)h->d_wa,
                           (const c32 *)h->d_pre, (const int *)((const char *)h->d_post + (size_t)N * 8), N, M, tm,
                           h->dtype == MI355_DTYPE_FLOAT ? 1 : 0);
        int rc = h->sub_f ? launch_handle(h->sub_f, h->d_wa, h->d_wb, nf, st) : launch_fft(h->ctx, M, -1, h->d_wa, h->d_wb, h->d_ones, h->d_twm_f, nf, 0, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_blu_mul, dim3(grid_for(tm)), dim3(256), 0, st, (c32 *)h->d_wb, (const c32 *)h->d_bspec, M, tm);
        rc = h->sub_i ? launch_handle(h->sub_i, h->d_wb, h->d_wa, nf, st) : launch_fft(h->ctx, M, 1, h->d_wb, h->d_wa, h->d_ones, h->d_twm_i, nf, 0, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_blu_post, dim3(grid_for(tn)), dim3(256), 0, st, (const c32 *)h->d_wa, (c32 *)out + f0 * N,
                           (const c32 *)h->d_post, N, M, tn, (h->sign < 0 && h->shift) ? len : 0);
        MI355_HIP(hipGetLastError());
    }
    return ws_release(h, st);
}

template <int NR, bool PASS_B>
int launch_tile(mi355_fft *h, const void *in, c32 *out, const c32 *twR, int ld, int nframes, int row_xor, bool real_in, hipStream_t st)
{
    constexpr int lds_bytes = 16 * (NR + 1) * 8;
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    const long long items = (long long)nframes * (ld / 16);
    const int per_cu = (160 * 1024) / lds_bytes;
    long long grid = (long long)cus * (per_cu < 1 ? 1 : per_cu);
    if (grid > items) grid = items;
    if (grid > ld / 16) grid -= grid % (ld / 16);  // a workgroup then meets the same tile of every frame (its window values stay in registers)
    const int nmask = h->n - 1;
    constexpr bool WRD = !PASS_B && NR == 256;  // window values of pass A in registers
    const bool wr = !PASS_B && NR < 1024 && (getenv("MI355_FFT_TILE_WREG") ? atoi(getenv("MI355_FFT_TILE_WREG")) != 0 : WRD);
#define TILE1(SG, RL, WR)                                                                                                       \
    do {                                                                                                                        \
        MI355_HIP(hipFuncSetAttribute((const void *)k_fft_tile<NR, SG, PASS_B, RL, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        hipLaunchKernelGGL((k_fft_tile<NR, SG, PASS_B, RL, WR>), dim3((unsigned)grid), dim3(NR), lds_bytes, st, in, out, h->d_window, (const c32 *)h->d_tw, \
                           twR, ld, nmask, items, row_xor);                                                                     \
    } while (0)
#define TILE(SG, RL)                                                                                                            \
    do {                                                                                                                        \
        if constexpr (!PASS_B && NR < 1024) { if (wr) TILE1(SG, RL, true); else TILE1(SG, RL, false); }                         \
        else TILE1(SG, RL, false);                                                                                              \
    } while (0)
    if constexpr (NR == 1024) {
        // 1024 rows: half the threads, two transforms each, the next tile fetched under the transform (2^20 points 540 -> 517 us, 2^19 500 -> 485
        // on one box; the same form at 512 rows, where two workgroups share the CU anyway, measured 3 % slower and is not instantiated)
        static const bool half = getenv("MI355_FFT_TILE_HALF") ? atoi(getenv("MI355_FFT_TILE_HALF")) != 0 : true;
        if (half) {
#define TILEH(SG, RL)                                                                                                           \
    do {                                                                                                                        \
        MI355_HIP(hipFuncSetAttribute((const void *)k_fft_tile_h<NR, SG, PASS_B, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        hipLaunchKernelGGL((k_fft_tile_h<NR, SG, PASS_B, RL>), dim3((unsigned)grid), dim3(NR / 2), lds_bytes, st, in, out, h->d_window, (const c32 *)h->d_tw, \
                           twR, ld, nmask, items, row_xor);                                                                     \
    } while (0)
            if constexpr (PASS_B) {
                if (h->sign < 0) TILEH(-1, false); else TILEH(1, false);
            } else {
                if (h->sign < 0) { if (real_in) TILEH(-1, true); else TILEH(-1, false); }
                else             { if (real_in) TILEH(1, true);  else TILEH(1, false); }
            }
#undef TILEH
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
    if constexpr (PASS_B) {
        if (h->sign < 0) TILE(-1, false); else TILE(1, false);
    } else {
        if (h->sign < 0) { if (real_in) TILE(-1, true); else TILE(-1, false); }
        else             { if (real_in) TILE(1, true);  else TILE(1, false); }
    }
#undef TILE
#undef TILE1
    (void)wr;
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

template <bool PASS_B>
int launch_tile_nr(mi355_fft *h, int nr, const void *in, c32 *out, const c32 *twR, int ld, int nframes, int row_xor, bool real_in, hipStream_t st)
{
    switch (nr) {
    case 256: return launch_tile<256, PASS_B>(h, in, out, twR, ld, nframes, row_xor, real_in, st);
    case 512: return launch_tile<512, PASS_B>(h, in, out, twR, ld, nframes, row_xor, real_in, st);
    default: return launch_tile<1024, PASS_B>(h, in, out, twR, ld, nframes, row_xor, real_in, st);
    }
}

int launch_big(mi355_fft *h, const void *in, void *out, int nframes, hipStream_t st)
{
    const int N = h->n, S = N / 4096;
    static const size_t ws_mb = getenv("MI355_FFT_WS_MB") ? (size_t)atoi(getenv("MI355_FFT_WS_MB")) : 256;
    size_t chunk = (ws_mb << 20) / ((size_t)N * 8);  // workspace bounded to 256 MiB
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)nframes) chunk = (size_t)nframes;
    std::lock_guard<std::mutex> ws_guard(h->ws_lock);
    {
        const int rc = ws_acquire(h, st, chunk > h->cap_frames);
        if (rc) return rc;
    }
    const bool three_pass = S > 16 && !h->tile_n1;  // 131072 points and more (decimation-in-time scheme): a second workspace for the intermediate transforms
    const bool four_pass = S > 256;  // 2^21 .. 2^24 points = 4096 x 16 x 16 x S3: one more combine level
    if (chunk > h->cap_frames) {
        MI355_HIP(hipStreamSynchronize(st));
        if (h->d_wa) (void)hipFree(h->d_wa);
        if (h->d_wb) (void)hipFree(h->d_wb);
        h->d_wa = h->d_wb = nullptr; h->cap_frames = 0;
        MI355_HIP(hipMalloc(&h->d_wa, chunk * (size_t)N * 8));
        if (three_pass) MI355_HIP(hipMalloc(&h->d_wb, chunk * (size_t)N * 8));
        h->cap_frames = chunk;
    }
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    const size_t isz = h->dtype == MI355_DTYPE_FLOAT ? 4 : 8;
    const int in_xor = (h->sign > 0 && h->shift) ? 8 : 0, m_xor = (h->sign < 0 && h->shift) ? S / 2 : 0;
    const c32 *tw4096 = (const c32 *)h->d_tw + N;
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += chunk) {
        const int nf = (int)((size_t)nframes - f0 < chunk ? (size_t)nframes - f0 : chunk);
        const int nsub = nf * S / 4;  // quads of sub-frames
        // whole octets: workgroup id % 8 = XCD (see k_fft_sub); one quad per workgroup (8 per CU) measures 465 us per 2^26 samples
        // against 494 with the two resident workgroups per CU walking four quads each
        const int grid = (mi355_balanced_grid(h->ctx, nsub, 8, 8) + 7) & ~7;
        const void *src = (const char *)in + f0 * N * isz;
        if (h->tile_n1) {  // two passes over whole lines: x (n1 rows of n2) -> ws[n2][k1] -> out[k2][k1]
            const int n1 = h->tile_n1, n2 = h->tile_n2;
            const c32 *tw1 = (const c32 *)h->d_tw + N + 4096, *tw2 = tw1 + n1;
            int rc = launch_tile_nr<false>(h, n1, src, (c32 *)h->d_wa, tw1, n2, nf, (h->sign > 0 && h->shift) ? n1 / 2 : 0, h->dtype == MI355_DTYPE_FLOAT, st);
            if (rc) return rc;
            rc = launch_tile_nr<true>(h, n2, h->d_wa, (c32 *)out + f0 * N, tw2, n1, nf, (h->sign < 0 && h->shift) ? n2 / 2 : 0, false, st);
            if (rc) return rc;
            continue;
        }
#define SUB(SG, RL) hipLaunchKernelGGL((k_fft_sub<SG, RL>), dim3(grid), dim3(256), 0, st, src, (c32 *)h->d_wa, h->d_window, tw4096, S, nsub, in_xor)
        if (h->sign < 0) { if (h->dtype == MI355_DTYPE_FLOAT) SUB(-1, true); else SUB(-1, false); }
        else             { if (h->dtype == MI355_DTYPE_FLOAT) SUB(1, true);  else SUB(1, false); }
#undef SUB
        c32 *dst = (c32 *)out + f0 * N;
#define COMB2(SS, SG, GY, ...) hipLaunchKernelGGL((k_fft_combine2<SS, SG>), dim3((unsigned)blocks, GY), dim3(256), 0, st, __VA_ARGS__)
        if (four_pass) {
            // sub-frame s = 16 S3 a + S3 b + c (a, b < 16, c < S3):
            //   H_(b,c) = radix-16 combine over a of E_s            (A -> B)   65536-point transforms of x[16 S3 m + S3 b + c]
            //   G_c     = radix-16 combine over b of H_(b,c)        (B -> A)   2^20-point transforms of x[S3 m + c]
            //   X       = radix-S3 combine over c of G_c            (A -> out)
            const int S3 = S / 256, T2 = 16 * S3;  // T2 = number of 65536-point intermediate transforms
            {
                const long long total = (long long)nf * 4096;
                long long blocks = (total + 255) / 256;
                if (blocks > (long long)cus * 16 / T2 + 1) blocks = (long long)cus * 16 / T2 + 1;
                if (h->sign < 0) COMB2(16, -1, T2, (const c32 *)h->d_wa, (c32 *)h->d_wb, (const c32 *)h->d_tw, total, 4096, (long long)N, (long long)T2 * 4096, 4096LL, 65536LL, T2, N - 1, 0);
                else             COMB2(16, 1, T2, (const c32 *)h->d_wa, (c32 *)h->d_wb, (const c32 *)h->d_tw, total, 4096, (long long)N, (long long)T2 * 4096, 4096LL, 65536LL, T2, N - 1, 0);
            }
            {
                const long long total = (long long)nf * 65536;
                long long blocks = (total + 255) / 256;
                if (blocks > (long long)cus * 16 / S3 + 1) blocks = (long long)cus * 16 / S3 + 1;
                if (h->sign < 0) COMB2(16, -1, S3, (const c32 *)h->d_wb, (c32 *)h->d_wa, (const c32 *)h->d_tw, total, 65536, (long long)N, (long long)S3 * 65536, 65536LL, 1048576LL, S3, N - 1, 0);
                else             COMB2(16, 1, S3, (const c32 *)h->d_wb, (c32 *)h->d_wa, (const c32 *)h->d_tw, total, 65536, (long long)N, (long long)S3 * 65536, 65536LL, 1048576LL, S3, N - 1, 0);
            }
            {
                const long long total = (long long)nf * 1048576;
                long long blocks = (total + 255) / 256;
                if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
                const int mx = (h->sign < 0 && h->shift) ? S3 / 2 : 0;
#define COMB2G(SS) do { if (h->sign < 0) COMB2(SS, -1, 1, (const c32 *)h->d_wa, dst, (const c32 *)h->d_tw, total, 1048576, (long long)N, 1048576LL, 0LL, 0LL, 1, N - 1, mx); \
                        else             COMB2(SS, 1, 1, (const c32 *)h->d_wa, dst, (const c32 *)h->d_tw, total, 1048576, (long long)N, 1048576LL, 0LL, 0LL, 1, N - 1, mx); } while (0)
                if (S3 == 2) COMB2G(2); else if (S3 == 4) COMB2G(4); else if (S3 == 8) COMB2G(8); else COMB2G(16);
#undef COMB2G
            }
            MI355_HIP(hipGetLastError());
            continue;
        }
        if (three_pass) {
            const int S2 = S / 16;
            {   // G_b = radix-16 combine of the sub-frames S2 a + b  (workspace A -> B), all b in one launch
                const long long total = (long long)nf * 4096;
                long long blocks = (total + 255) / 256;
                if (blocks > (long long)cus * 16 / S2 + 1) blocks = (long long)cus * 16 / S2 + 1;
                if (h->sign < 0) COMB2(16, -1, S2, (const c32 *)h->d_wa, (c32 *)h->d_wb, (const c32 *)h->d_tw, total, 4096, (long long)N, (long long)S2 * 4096, 4096LL, 65536LL, S2, N - 1, 0);
                else             COMB2(16, 1, S2, (const c32 *)h->d_wa, (c32 *)h->d_wb, (const c32 *)h->d_tw, total, 4096, (long long)N, (long long)S2 * 4096, 4096LL, 65536LL, S2, N - 1, 0);
            }
            {   // X = radix-S2 combine of the G_b  (workspace B -> out)
                const long long total = (long long)nf * 65536;
                long long blocks = (total + 255) / 256;
                if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
                const int mx = (h->sign < 0 && h->shift) ? S2 / 2 : 0;
#define COMB2F(SS) do { if (h->sign < 0) COMB2(SS, -1, 1, (const c32 *)h->d_wb, dst, (const c32 *)h->d_tw, total, 65536, (long long)N, 65536LL, 0LL, 0LL, 1, N - 1, mx); \
                        else             COMB2(SS, 1, 1, (const c32 *)h->d_wb, dst, (const c32 *)h->d_tw, total, 65536, (long long)N, 65536LL, 0LL, 0LL, 1, N - 1, mx); } while (0)
                if (S2 == 2) COMB2F(2); else if (S2 == 4) COMB2F(4); else if (S2 == 8) COMB2F(8); else COMB2F(16);
#undef COMB2F
            }
            MI355_HIP(hipGetLastError());
            continue;
        }
#undef COMB2
        const long long total = (long long)nf * 4096;
        long long blocks = (total + 255) / 256;
        if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
#define COMB(SS, SG) hipLaunchKernelGGL((k_fft_combine<SS, SG>), dim3((unsigned)blocks), dim3(256), 0, st, (const c32 *)h->d_wa, dst, (const c32 *)h->d_tw, total, m_xor)
        if (S == 8) { if (h->sign < 0) COMB(8, -1); else COMB(8, 1); }
        else        { if (h->sign < 0) COMB(16, -1); else COMB(16, 1); }
#undef COMB
        MI355_HIP(hipGetLastError());
    }
    return ws_release(h, st);
}

int launch_mr_tile(mi355_fft *h, const void *in, void *out, int nframes, hipStream_t st)
{
    const size_t N = (size_t)h->n;
    size_t chunk = ((size_t)256 << 20) / (N * 8);  // workspace bounded to 256 MiB
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)nframes) chunk = (size_t)nframes;
    std::lock_guard<std::mutex> ws_guard(h->ws_lock);
    {
        const int rc = ws_acquire(h, st, chunk > h->cap_frames);
        if (rc) return rc;
    }
    if (chunk > h->cap_frames) {
        MI355_HIP(hipStreamSynchronize(st));
        if (h->d_wa) (void)hipFree(h->d_wa);
        h->d_wa = nullptr; h->cap_frames = 0;
        MI355_HIP(hipMalloc(&h->d_wa, chunk * N * 8));
        h->cap_frames = chunk;
    }
    const size_t isz = h->dtype == MI355_DTYPE_FLOAT ? 4 : 8;
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += chunk) {
        const int nf = (int)((size_t)nframes - f0 < chunk ? (size_t)nframes - f0 : chunk);
        const int rc = mi355_fft_mr_tile_launch(h->mrt, h->ctx, h->sign, (const char *)in + f0 * N * isz, h->d_wa, (char *)out + f0 * N * 8, h->d_window, nf,
                                                h->shift, h->dtype == MI355_DTYPE_FLOAT, st);
        if (rc) return rc;
    }
    return ws_release(h, st);
}

int launch_handle(mi355_fft *h, const void *in, void *out, int nvec, hipStream_t st)
{
    if (h->mrt.n) return launch_mr_tile(h, in, out, nvec, st);
    if (h->mr.n) return mi355_fft_mr_launch(h->mr, h->ctx, h->sign, in, out, h->d_window, nvec, h->shift, h->dtype == MI355_DTYPE_FLOAT, st);
    if (h->m) return launch_bluestein(h, in, out, nvec, st);
    if (h->n > 32768 || (h->n == 32768 && h->two_kernel)) return launch_big(h, in, out, nvec, st);
    return launch_fft(h->ctx, h->n, h->sign, in, out, h->d_window, h->d_tw, nvec, h->shift, h->dtype == MI355_DTYPE_FLOAT, st);
}

// iterative radix-2 FFT in double (host side, used once per handle for the chirp spectrum)
void host_fft_pow2(std::vector<double> &re, std::vector<double> &im, int sign)
{
    const int n = (int)re.size();
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    // one table of n/2 roots (a sincos per butterfly took seconds at the sizes the large chirp-z lengths need)
    std::vector<double> cr(n / 2 > 0 ? n / 2 : 1), ci(n / 2 > 0 ? n / 2 : 1);
    for (int k = 0; k < n / 2; k++) {
        const double a = sign * 2.0 * M_PI * (double)k / (double)n;
        cr[k] = cos(a);
        ci[k] = sin(a);
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                const double wr = cr[(size_t)k * step], wi = ci[(size_t)k * step];
                const int p = i + k, q = p + len / 2;
                const double tr = re[q] * wr - im[q] * wi, ti = re[q] * wi + im[q] * wr;
                re[q] = re[p] - tr; im[q] = im[p] - ti;
                re[p] += tr; im[p] += ti;
            }
    }
}

int upload(mi355_ctx *ctx, void **d, const void *src, size_t bytes)
{
    if (hipMalloc(d, bytes) != hipSuccess) return MI355_ERR_NOMEM;
    if (mi355_upload(ctx, *d, src, bytes) != hipSuccess) return MI355_ERR_HIP;
    return MI355_OK;
}

std::vector<float> twiddle_table(int n, int sign)
{
    std::vector<float> tw;
    tw.reserve(2 * (size_t)n + 8192);
    for (int k = 0; k < n; k++) {
        const double a = sign * 2.0 * M_PI * (double)k / (double)n;
        tw.push_back((float)cos(a));
        tw.push_back((float)sin(a));
    }
    if (n > 4096)  // the interleaved sub-transform kernel also needs the 4096-point table
        for (int k = 0; k < 4096; k++) {
            const double a = sign * 2.0 * M_PI * (double)k / 4096.0;
            tw.push_back((float)cos(a));
            tw.push_back((float)sin(a));
        }
    return tw;
}

// tables of the chirp-z path; window == nullptr means all ones
int setup_bluestein(mi355_fft *h, const float *window)
{
    const int N = h->n;
    int M = 256;  // at least 256: the fused kernel's smallest instantiation (more zero padding is harmless)
    while (M < 2 * N - 1) M <<= 1;
    h->m = M;
    // a[n] = exp(sign * i*pi*n^2/N); n^2 reduced mod 2N in integers keeps the phase exact
    std::vector<double> ar(N), ai(N);
    for (int n = 0; n < N; n++) {
        const long long q = ((long long)n * n) % (2LL * N);
        const double ang = h->sign * M_PI * (double)q / (double)N;
        ar[n] = cos(ang); ai[n] = sin(ang);
    }
    const int half = N / 2;
    const bool rshift = h->sign > 0 && h->shift;
    std::vector<float> pre(2 * (size_t)N), post(2 * (size_t)N);
    std::vector<int> src(N);
    for (int m = 0; m < N; m++) {
        // reverse + shift: original position i lands on m = i < half ? i + (N - half) : i - half (lib/clFFT_impl.cc:477-493)
        const int i = rshift ? (m >= N - half ? m - (N - half) : m + half) : m;
        src[m] = i;
        const double w = window ? (double)window[i] : 1.0;
        pre[2 * m] = (float)(w * ar[m]); pre[2 * m + 1] = (float)(w * ai[m]);
        post[2 * m] = (float)ar[m]; post[2 * m + 1] = (float)ai[m];
    }
    std::vector<double> br(M, 0.0), bi(M, 0.0);
    for (int n = 0; n < N; n++) {
        br[n] = ar[n]; bi[n] = -ai[n];
        if (n) { br[M - n] = ar[n]; bi[M - n] = -ai[n]; }
    }
    host_fft_pow2(br, bi, -1);
    std::vector<float> bs(2 * (size_t)M);
    for (int k = 0; k < M; k++) { bs[2 * k] = (float)(br[k] / M); bs[2 * k + 1] = (float)(bi[k] / M); }
    // post table is followed by the int source map (one allocation)
    std::vector<char> post_blob((size_t)N * 8 + (size_t)N * 4);
    memcpy(post_blob.data(), post.data(), (size_t)N * 8);
    memcpy(post_blob.data() + (size_t)N * 8, src.data(), (size_t)N * 4);
    int rc;
    if ((rc = upload(h->ctx, &h->d_pre, pre.data(), pre.size() * 4))) return rc;
    if ((rc = upload(h->ctx, &h->d_post, post_blob.data(), post_blob.size()))) return rc;
    if ((rc = upload(h->ctx, &h->d_bspec, bs.data(), bs.size() * 4))) return rc;
    if (M > 32768) {  // the convolution's transforms are multi-pass sizes: handles of their own (tables, workspaces, launch plan)
        if ((rc = mi355_fft_create(h->ctx, M, MI355_FFT_FORWARD, nullptr, 0, MI355_DTYPE_COMPLEX, 1, 0, &h->sub_f))) return rc;
        return mi355_fft_create(h->ctx, M, MI355_FFT_BACKWARD, nullptr, 0, MI355_DTYPE_COMPLEX, 1, 0, &h->sub_i);
    }
    std::vector<float> tf = twiddle_table(M, -1), ti = twiddle_table(M, 1), ones(M, 1.0f);
    if ((rc = upload(h->ctx, &h->d_twm_f, tf.data(), tf.size() * 4))) return rc;
    if ((rc = upload(h->ctx, &h->d_twm_i, ti.data(), ti.size() * 4))) return rc;
    if ((rc = upload(h->ctx, (void **)&h->d_ones, ones.data(), ones.size() * 4))) return rc;
    return MI355_OK;
}

}  // namespace

extern "C" int mi355_fft_create(mi355_ctx *ctx, int fft_size, int direction, const float *window, int window_len, int dtype,
                                int num_streams, int shift, mi355_fft **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    const bool pow2 = fft_size >= 2 && (fft_size & (fft_size - 1)) == 0;
    if (fft_size < 2 || (pow2 && fft_size > 16777216) || (!pow2 && fft_size > 8388608)) {
        mi355_set_error("fft size %d unsupported (powers of two 2..16777216, any other size 3..8388608)", fft_size);
        return fft_size < 2 ? MI355_ERR_INVALID_ARG : MI355_ERR_UNSUPPORTED;
    }
    MI355_REQUIRE(window_len == 0 || window_len == fft_size, "window not the same length as fft_size");
    MI355_REQUIRE(window_len == 0 || window != nullptr, "window is NULL");
    MI355_REQUIRE(dtype == MI355_DTYPE_COMPLEX || dtype == MI355_DTYPE_FLOAT, "clFFT dtype must be complex or float");
    MI355_REQUIRE(num_streams >= 1, "num_streams must be >= 1");
    mi355_fft *h = new (std::nothrow) mi355_fft();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->n = fft_size;
    // lib/clFFT_impl.cc:84-89: anything that is not CLFFT_FORWARD (-1) is a backward transform
    h->sign = (direction == MI355_FFT_FORWARD) ? -1 : 1;
    h->dtype = dtype; h->nstreams = num_streams; h->shift = shift ? 1 : 0;
    h->d_window = nullptr; h->d_tw = nullptr;
    auto fail = [&](int rc) { mi355_fft_destroy(h); return rc; };
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(MI355_ERR_HIP);
    std::vector<float> tw(2 * (size_t)fft_size);
    for (int k = 0; k < fft_size; k++) {
        double a = h->sign * 2.0 * M_PI * (double)k / (double)fft_size;
        tw[2 * k] = (float)cos(a);
        tw[2 * k + 1] = (float)sin(a);
    }
    if (fft_size > 4096) {  // the interleaved sub-transform kernel also needs the 4096-point table
        for (int k = 0; k < 4096; k++) {
            double a = h->sign * 2.0 * M_PI * (double)k / 4096.0;
            tw.push_back((float)cos(a));
            tw.push_back((float)sin(a));
        }
    }
    if (pow2 && fft_size >= 65536 && fft_size <= 1048576 && !getenv("MI355_FFT_NO_TILE")) {
        // two-pass tile scheme: N = n1 x n2, both 256 .. 1024; their twiddle tables follow the 4096-point one
        int lg = 0;
        while ((1 << lg) < fft_size) lg++;
        // 131072 = 256 x 512: the 256-row pass first (its window values stay in registers); 2^19 = 1024 x 512 measured 3 % faster than 512 x 1024
        h->tile_n1 = lg == 17 ? 256 : 1 << ((lg + 1) / 2);
        if (const char *e = getenv("MI355_FFT_TILE_N1")) {  // (tuning switch)
            const int v = atoi(e);
            if ((v == 256 || v == 512 || v == 1024) && fft_size / v >= 256 && fft_size / v <= 1024) h->tile_n1 = v;
        }
        h->tile_n2 = fft_size / h->tile_n1;
        for (int nr : {h->tile_n1, h->tile_n2})
            for (int k = 0; k < nr; k++) {
                double a = h->sign * 2.0 * M_PI * (double)k / (double)nr;
                tw.push_back((float)cos(a));
                tw.push_back((float)sin(a));
            }
    }
    if (hipMalloc(&h->d_tw, tw.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_NOMEM);
    if (mi355_upload(ctx, h->d_tw, tw.data(), tw.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_HIP);
    {
        std::vector<float> w(fft_size, 1.0f);  // no window == all ones: the kernel has a single code path
        if (window_len) memcpy(w.data(), window, sizeof(float) * (size_t)fft_size);
        h->two_kernel = pow2 && (fft_size > 32768 || (fft_size == 32768 && getenv("MI355_FFT_32768_TWO_KERNELS") != nullptr));
        if (h->two_kernel && !h->tile_n1) {
            // two-kernel sizes: k_fft_sub reads the window values of the sub-frames (s .. s+3) for every n -- stored
            // quad-major, [s/4][n][4], they are one contiguous 16-byte load per lane instead of 16 bytes out of every S*4
            const int S = fft_size / 4096;
            std::vector<float> p((size_t)fft_size);
            for (int sq = 0; sq < S / 4; sq++)
                for (int n = 0; n < 4096; n++)
                    for (int j = 0; j < 4; j++) p[((size_t)sq * 4096 + n) * 4 + j] = w[(size_t)n * S + 4 * sq + j];
            w.swap(p);
        }
        if (hipMalloc((void **)&h->d_window, sizeof(float) * (size_t)fft_size) != hipSuccess) return fail(MI355_ERR_NOMEM);
        if (mi355_upload(ctx, h->d_window, w.data(), sizeof(float) * (size_t)fft_size) != hipSuccess)
            return fail(MI355_ERR_HIP);
    }
    int rc = h->pipe.init(ctx);
    if (rc) return fail(rc);
    if (!pow2) {
        std::vector<float> mtw;
        if (!getenv("MI355_FFT_NO_MR") && mi355_fft_mr_plan(fft_size, h->sign, 0, &h->mr, &mtw)) {
            // Two factorisations (fewest passes / at least 14 values per thread), each with its workgroup size and frames per iteration
            // measured once per length and process (fft_mr.hip); the faster one stays.  ~20 ms at the first handle of a length.
            auto put = [&](MrPlan &pl, const std::vector<float> &t) {
                if (hipMalloc(&pl.d_tw, t.size() * sizeof(float)) != hipSuccess) return (int)MI355_ERR_NOMEM;
                return mi355_upload(ctx, pl.d_tw, t.data(), t.size() * sizeof(float)) != hipSuccess ? (int)MI355_ERR_HIP : (int)MI355_OK;
            };
            // The factorisation decides the rounding of the result (workgroup size and frames per iteration do not), so it must not depend on
            // a wall-clock measurement: two processes, or a busy and an idle device, would return bitwise different transforms of the same
            // input.  The rule's factorisation (fewest passes) is used; MI355_FFT_MR_VARIANT=0 / 1 pins either one, and
            // MI355_FFT_MR_TIMED_VARIANT=1 brings back the measured choice (the second factorisation replaces the rule's when it measures at
            // least 10 % faster -- for tuning runs, not for production).  Workgroup size and frames per iteration, which do not change a single
            // bit of the result, are still measured once per length (MI355_FFT_MR_AUTOTUNE=0 skips that too).
            int known = mi355_fft_mr_cached_variant(fft_size);
            // checked-in exceptions to the rule: lengths whose alternative factorisation measured > 10 % faster (tools/r05_fft_mr_variants.py
            // over 28 lengths 96 ... 15360: 96 x 1.18, 100 x 1.23, 2400 x 1.14; every other length within 4 % or slower)
            if (known < 0 && !getenv("MI355_FFT_MR_TIMED_VARIANT")) known = (fft_size == 96 || fft_size == 100 || fft_size == 2400) ? 1 : 0;
            if (const char *e = getenv("MI355_FFT_MR_VARIANT")) known = atoi(e) != 0 ? 1 : 0;
            MrPlan alt;
            std::vector<float> atw;
            const bool have_alt = mi355_fft_mr_plan(fft_size, h->sign, 1, &alt, &atw) && (alt.npass != h->mr.npass || alt.per_thread != h->mr.per_thread);
            if (known == 1 && have_alt) {
                h->mr = alt;
                mtw.swap(atw);
            }
            if ((rc = put(h->mr, mtw))) return fail(rc);
            float ms0 = -1.f, ms1 = -1.f;
            if ((rc = mi355_fft_mr_tune(&h->mr, ctx, h->sign, h->d_window, &ms0))) return fail(rc);
            if (known < 0 && have_alt && ms0 > 0.f && getenv("MI355_FFT_MR_TIMED_VARIANT")) {
                if ((rc = put(alt, atw))) { if (alt.d_tw) (void)hipFree(alt.d_tw); return fail(rc); }
                rc = mi355_fft_mr_tune(&alt, ctx, h->sign, h->d_window, &ms1);
                if (rc) { (void)hipFree(alt.d_tw); return fail(rc); }
                if (ms1 > 0.f && ms1 < ms0 * 0.90f) {
                    (void)hipFree(h->mr.d_tw);
                    h->mr = alt;
                } else {
                    (void)hipFree(alt.d_tw);
                }
            }
            if (ms0 > 0.f) {
                mi355_fft_mr_remember(h->mr);
                mi355_log(ctx, MI355_LOG_INFO, "clFFT %d points (mixed radix): %d passes, %d threads x %d frame(s) per workgroup measured fastest", fft_size,
                          h->mr.npass, h->mr.threads, h->mr.frames);
            }
        } else {
            h->mr.n = 0;
            std::vector<float> twa, twb;
            if (!getenv("MI355_FFT_NO_MR") && mi355_fft_mr_tile_plan(fft_size, h->sign, &h->mrt, &twa, &twb)) {
                // two passes with the mixed-radix passes inside (fft_mr.hip); W_N^k is the table this handle already has
                h->mrt.a.d_tw = h->mrt.b.d_tw = nullptr;
                if (hipMalloc(&h->mrt.a.d_tw, twa.size() * sizeof(float) + 8) != hipSuccess) return fail(MI355_ERR_NOMEM);
                if (hipMalloc(&h->mrt.b.d_tw, twb.size() * sizeof(float) + 8) != hipSuccess) return fail(MI355_ERR_NOMEM);
                if (!twa.empty() && mi355_upload(ctx, h->mrt.a.d_tw, twa.data(), twa.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_HIP);
                if (!twb.empty() && mi355_upload(ctx, h->mrt.b.d_tw, twb.data(), twb.size() * sizeof(float)) != hipSuccess) return fail(MI355_ERR_HIP);
                h->mrt.d_twn = h->d_tw;
            } else {
                h->mrt.n = 0;
                rc = setup_bluestein(h, window_len ? window : nullptr);
                if (rc) return fail(rc);
            }
        }
    }
    // (the table uploads ran on the context's upload stream and were waited for there: mi355_upload; no device-wide wait)
    mi355_log(ctx, MI355_LOG_INFO, "clFFT: %d points, %s, %s input, %d stream(s), shift %d, window %s: %s", fft_size,
              h->sign < 0 ? "forward" : "reverse", dtype == MI355_DTYPE_COMPLEX ? "complex" : "float", num_streams, h->shift,
              window_len ? "given" : "none",
              h->mr.n ? "mixed radix, one pass" : h->mrt.n ? "mixed radix, two passes over sixteen-column tiles" : !pow2 ? "chirp-z over a power-of-two transform" : h->tile_n1 ? "two passes over 16-column tiles" : h->two_kernel ? "multi-pass" : "one pass");
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_fft_plan_text(int fft_size, char *buf, int buf_len)
{
    MI355_REQUIRE(buf && buf_len > 0, "NULL argument");
    const bool pow2 = fft_size >= 2 && (fft_size & (fft_size - 1)) == 0;
    if (fft_size < 2 || (pow2 && fft_size > 16777216) || (!pow2 && fft_size > 8388608)) {
        snprintf(buf, (size_t)buf_len, "unsupported (powers of two 2..16777216, any other size 3..8388608)");
        return MI355_ERR_UNSUPPORTED;
    }
    if (pow2) {
        if (fft_size <= 32768) snprintf(buf, (size_t)buf_len, "one pass");
        else if (fft_size <= 1048576) {
            int lg = 0;
            while ((1 << lg) < fft_size) lg++;
            const int n1 = lg == 17 ? 256 : 1 << ((lg + 1) / 2);
            snprintf(buf, (size_t)buf_len, "two tile passes %d x %d", n1, fft_size / n1);
        } else snprintf(buf, (size_t)buf_len, "four passes");
        return MI355_OK;
    }
    MrPlan mp;
    std::vector<float> tw;
    if (mi355_fft_mr_plan(fft_size, -1, 0, &mp, &tw)) {
        int at = snprintf(buf, (size_t)buf_len, "mixed radix ");
        for (int p = 0; p < mp.npass && at < buf_len; p++) at += snprintf(buf + at, (size_t)(buf_len - at), p ? " x %d" : "%d", mp.pass[p].radix);
        return MI355_OK;
    }
    MrTilePlan tp;
    std::vector<float> twb;
    if (mi355_fft_mr_tile_plan(fft_size, -1, &tp, &tw, &twb)) {
        snprintf(buf, (size_t)buf_len, "mixed radix, two passes %d x %d", tp.n1, tp.n2);
        return MI355_OK;
    }
    int m = 256;
    while (m < 2 * fft_size - 1) m <<= 1;
    snprintf(buf, (size_t)buf_len, "chirp-z, m = %d%s", m, m <= 16384 ? " (fused)" : "");
    return MI355_OK;
}

const MrPlan *mi355_fft_mr_plan_of(const mi355_fft *h, int *sign)
{
    if (!h || !h->mr.n) return nullptr;
    if (sign) *sign = h->sign;
    return &h->mr;
}

extern "C" int mi355_fft_destroy(mi355_fft *h)
{
    if (!h) return MI355_OK;
    (void)hipSetDevice(h->ctx->device);
    h->pipe.release();
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_tw) (void)hipFree(h->d_tw);
    if (h->mr.d_tw) (void)hipFree(h->mr.d_tw);
    if (h->mrt.a.d_tw) (void)hipFree(h->mrt.a.d_tw);
    if (h->mrt.b.d_tw) (void)hipFree(h->mrt.b.d_tw);
    for (void *p : {h->d_pre, h->d_post, h->d_bspec, h->d_twm_f, h->d_twm_i, (void *)h->d_ones, h->d_wa, h->d_wb})
        if (p) (void)hipFree(p);
    if (h->ws_done) (void)hipEventDestroy(h->ws_done);
    if (h->sub_f) (void)mi355_fft_destroy(h->sub_f);
    if (h->sub_i) (void)mi355_fft_destroy(h->sub_i);
    delete h;
    return MI355_OK;
}

extern "C" int mi355_fft_work_dev(mi355_fft *h, int nvec, const void *in, void *out, void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (nvec <= 0) return MI355_OK;
    MI355_REQUIRE(in && out, "NULL buffer");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
                  "device buffers must be 8-byte aligned");
    // the 8192/16384-point kernel reads 16 bytes per lane
    MI355_REQUIRE(h->m != 0 || h->n <= 4096 || (reinterpret_cast<uintptr_t>(in) & 15u) == 0,
                  "device input of an 8192/16384-point transform must be 16-byte aligned");
    MI355_HIP(hipSetDevice(h->ctx->device));
    return launch_handle(h, in, out, nvec, mi355_pick_stream(h->ctx, stream));
}

extern "C" int mi355_fft_work(mi355_fft *h, int nvec, const void *const *in_streams, void *const *out_streams)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (nvec <= 0) return MI355_OK;
    MI355_REQUIRE(in_streams && out_streams, "NULL stream arrays");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    const size_t isz = mi355_dtype_size(h->dtype), in_frame = isz * (size_t)h->n, out_frame = 8 * (size_t)h->n;
    size_t chunk_frames = mi355_chunk_bytes((size_t)nvec * out_frame) / out_frame;
    if (chunk_frames < 1) chunk_frames = 1;
    size_t first = (size_t)nvec < chunk_frames ? (size_t)nvec : chunk_frames;
    size_t inb = first * in_frame;
    const int nslots = first * out_frame >= ((size_t)32 << 20) ? 2 : HostPipe::kSlots;  // (a 2^22-point frame is 32 MiB: two staging slots)
    int rc = h->pipe.ensure(1, &inb, first * out_frame, nslots);
    if (rc) return rc;
    HostPipe &p = h->pipe;
    if ((size_t)nvec <= chunk_frames && mi355_direct_ok((size_t)nvec * out_frame)) {
        // small call: the kernels work on the pinned staging themselves (see common.h); streams one after the other
        hipStream_t st = h->ctx->stream[0];
        for (int s_i = 0; s_i < h->nstreams; s_i++) {
            MI355_REQUIRE(in_streams[s_i] && out_streams[s_i], "NULL stream buffer");
            mi355_copy(p.h_in[0][0], in_streams[s_i], (size_t)nvec * in_frame);
            rc = launch_handle(h, p.h_in[0][0], p.h_out[0], nvec, st);
            if (rc) return rc;
            MI355_HIP(mi355_direct_sync(st));
            mi355_copy(out_streams[s_i], p.h_out[0], (size_t)nvec * out_frame);
        }
        return MI355_OK;
    }
    // all (stream, chunk) pairs run through the staging slots back to back
    size_t nchunks = ((size_t)nvec + chunk_frames - 1) / chunk_frames;
    char *pend_dst[HostPipe::kSlots] = {};
    size_t pend_bytes[HostPipe::kSlots] = {};
    size_t seq = 0;
    const bool one_stream = h->m || h->mrt.n || h->n > 32768 || (h->n == 32768 && h->two_kernel);  // chirp-z / multi-pass sizes share work buffers: one stream
    for (int s_i = 0; s_i < h->nstreams; s_i++) {
        MI355_REQUIRE(in_streams[s_i] && out_streams[s_i], "NULL stream buffer");
        const char *pin = (const char *)in_streams[s_i];
        char *pout = (char *)out_streams[s_i];
        for (size_t ci = 0; ci < nchunks; ci++, seq++) {
            int s = (int)(seq % nslots);
            hipStream_t st = h->ctx->stream[one_stream ? 0 : (s & 1)];
            size_t f0 = ci * chunk_frames;
            size_t nf = (size_t)nvec - f0 < chunk_frames ? (size_t)nvec - f0 : chunk_frames;
            if (pend_bytes[s]) MI355_HIP(hipEventSynchronize(p.done[s]));
            // the slot's previous result out and its next input in, side by side
            mi355_copy2(pend_dst[s], p.h_out[s], pend_bytes[s], p.h_in[s][0], pin + f0 * in_frame, nf * in_frame);
            pend_bytes[s] = 0;
            MI355_HIP(hipMemcpyAsync(p.d_in[s][0], p.h_in[s][0], nf * in_frame, hipMemcpyHostToDevice, st));
            rc = launch_handle(h, p.d_in[s][0], p.d_out[s], (int)nf, st);
            if (rc) return rc;
            MI355_HIP(hipMemcpyAsync(p.h_out[s], p.d_out[s], nf * out_frame, hipMemcpyDeviceToHost, st));
            MI355_HIP(hipEventRecord(p.done[s], st));
            pend_dst[s] = pout + f0 * out_frame;
            pend_bytes[s] = nf * out_frame;
        }
    }
    for (int q = 0; q < nslots; q++) {
        int s = (int)((seq + q) % nslots);  // oldest slot first
        if (pend_bytes[s]) {
            MI355_HIP(hipEventSynchronize(p.done[s]));
            mi355_copy(pend_dst[s], p.h_out[s], pend_bytes[s]);
            pend_bytes[s] = 0;
        }
    }
    return MI355_OK;
}
