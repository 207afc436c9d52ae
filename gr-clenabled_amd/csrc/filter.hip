// clFilter / clComplexFilter as gfx950 HIP kernels:  y[m] = sum_k h[k] x[m*decim - k].
// Reference behaviour: lib/clFilter_impl.cc:50-83 (ctor), :504-589 (time domain),
// :591-681 (frequency domain: per 192-sample block clFFT fwd -> D2H -> host multiply
// -> H2D -> clFFT inv -> D2H -> host tail add, i.e. 5 PCIe transfers and 2 host
// syncs per block), lib/clComplexFilter_impl.cc:796-828,959-1030; CPU twins
// lib/fft_filter.cc:38-175, lib/fir_filter.cc:222-257,455-488.
//
// Frequency-domain mode = ONE fused overlap-save kernel.  With GNU Radio's history
// (ntaps-1 old samples in front of the buffer) every FFT block is independent:
//     block b reads in[b*L .. b*L+NF), L = NF-(ntaps-1); FFT_NF -> x H -> IFFT_NF;
//     y_blk[n], n >= ntaps-1, is y[b*L + n-(ntaps-1)]
// so there is no tail state (the reference's overlap-add tail, lib/fft_filter.cc:156-171,
// gives the same y).  A workgroup owns 4096 points = 4096/NF blocks; the forward
// transform's last pass leaves the spectrum in registers in exactly the order the
// inverse transform's first pass consumes (reversed radix plan), so the multiply by
// H happens in registers with no LDS trip between the two transforms.  H (pre-scaled
// by 1/NF like lib/fft_filter.cc:52-57) and all twiddles are per-thread constants
// held in registers across the persistent block loop.
// HBM traffic: 8 B read + 8 B written per input sample (the ntaps-1 overlap is
// re-read from L1/L2 by the neighbouring block of the same workgroup).
//
// Time-domain mode = register-tiled direct form: 8 outputs per thread, input tile in
// LDS, reversed taps fetched through the scalar cache.
#include <cmath>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "common.h"
#include "fft_core.hpp"

using namespace fftc;

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u2g __attribute__((vector_size(8)));  // operand type of the raw buffer builtins

__device__ __forceinline__ void st_stream(c32 *p, c32 v) { f2v o; o.x = v.x; o.y = v.y; __builtin_nontemporal_store(o, (f2v *)p); }

// ------------------------------------------------------------------------------------
// fused overlap-save fast convolution
// ------------------------------------------------------------------------------------
template <int NF, class G>
__global__ __launch_bounds__(G::TH, G::WPE) void k_ols(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                                   const c32 *__restrict__ Hspec,
                                                                   const c32 *__restrict__ tw_fwd,
                                                                   const c32 *__restrict__ tw_inv, int ntaps, int decim,
                                                                   int L,            // new samples per block (<= NF-ntaps+1)
                                                                   int s0,           // first stored output of a block (>= ntaps-1)
                                                                   long long n_in,   // readable input samples
                                                                   long long n_y,    // undecimated outputs wanted
                                                                   int nblocks, int ngroups,
                                                                   int accumulate,   // y += instead of y = (later segments of a partitioned long filter)
                                                                   int xcd_map)
{
    using PF = Plan<NF, false>;
    using PI = Plan<NF, true>;
    constexpr int TH = G::TH, PTS = G::PTS, F = G::F, NP = PF::NP;
    __shared__ c32 lds[NP > 1 ? PTS : 1];
    const int tid0 = threadIdx.x;

    // all-radix-16 sizes (256, 4096): the reversed plan IS the forward plan, so the inverse twiddles are the conjugates of the
    // forward ones at the same positions -- one register set serves both transforms
    constexpr bool SHARE = (PF::L % 4) == 0;
    TwRegs<NF> twf, twi;
    load_twiddles<NF, false, G>(twf, tid0, tw_fwd);
    if constexpr (!SHARE) load_twiddles<NF, true, G>(twi, tid0, tw_inv);
    // spectrum of the taps at the bins this thread holds after the forward transform
    constexpr int RL = PF::radix(NP - 1), BL = NF / RL;
    static_assert(PI::radix(0) == RL, "inverse plan must start with the forward plan's last radix");
    c32 Hreg[16];
#pragma unroll
    for (int q = 0; q < 16 / RL; q++) {
        const int j = (tid0 + TH * q) % BL;
#pragma unroll
        for (int s = 0; s < RL; s++) Hreg[q * RL + s] = Hspec[j + orev<RL>(s) * BL];
    }

    // bit k set: register k of the inverse transform's output lies outside the store window [s0, s0+L) of its block
    constexpr int RO = PI::radix(NP - 1), BO = NF / RO;
    int inval0 = 0;
#pragma unroll
    for (int q = 0; q < 16 / RO; q++) {
        const int j = (tid0 + TH * q) % BO;
#pragma unroll
        for (int s = 0; s < RO; s++) {
            const int n = j + orev<RO>(s) * BO;
            if (!(n >= s0 && n < s0 + L)) inval0 |= 1 << (q * RO + s);
        }
    }

    // Consecutive groups overlap by ntaps-1 input samples.  Workgroup id % 8 is the XCD: each XCD gets a contiguous run of
    // groups and its workgroups walk it side by side, so the overlap is re-read from that XCD's L2 instead of crossing the
    // fabric again (long filters read every sample twice).  Falls back to the plain stride when the grid is not whole octets.
    const bool xmap = (gridDim.x & 7) == 0 && xcd_map;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, chunk = (ngroups + 7) >> 3;
    for (int q = xmap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; q < (xmap ? chunk : ngroups); q += xmap ? per_xcd : (int)gridDim.x) {
        const int grp = xmap ? xcd * chunk + q : q;
        if (grp >= ngroups) break;
        c32 v[16];
        int tid = tid0;
        asm volatile("" : "+v"(tid));  // keep address arithmetic inside the loop (see fft.hip)
        // Group base sample (uniform); everything per thread is a 32-bit offset from it.
        const long long g0 = (long long)grp * F * L;
        const c32 *__restrict__ in_g = in + g0;
        const long long in_left64 = n_in - g0, y_left64 = n_y - g0;
        const unsigned in_left = in_left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)in_left64;
        const unsigned y_left = y_left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)(y_left64 > 0 ? y_left64 : 0);
        // ---- load: block fr of the group covers in_g[fr*L - pad + n], n < NF; zero outside the buffer.
        // The stored outputs are n in [s0, s0+L) with s0 = ntaps-1 rounded up to 16, so that every block's STORES start on
        // a 128-byte line; the pad = s0-(ntaps-1) samples the block reads earlier feed only outputs that are not stored
        // (for the first block they lie before the buffer and read as zero).
        constexpr int R0 = PF::radix(0), B0 = NF / R0;
        const int pad = s0 - (ntaps - 1);
        if (g0 > 0) {
            // every group but the first: one raw buffer over [in_g - pad, end of input) -- all offsets are non-negative, the
            // hardware range check supplies the zeros past the end, and the loads carry no compare / select / branch
            const long long left_b = (in_left64 + pad) * 8;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(in_g - pad), 0, left_b > 0x7ffffff8LL ? 0x7ffffff8 : (int)left_b, 0x00020000);
#pragma unroll
            for (int q = 0; q < 16 / R0; q++) {
                const int g = tid + TH * q, fr = g / B0, j = g % B0;
                const unsigned off = (unsigned)(fr * L + j) * 8u;
#pragma unroll
                for (int r = 0; r < R0; r++) {
                    const f2v x = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, off + (unsigned)(r * B0 * 8), 0, 0));
                    v[q * R0 + r] = mk(x.x, x.y);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16 / R0; q++) {
                const int g = tid + TH * q, fr = g / B0, j = g % B0;
                const int base = fr * L + j - pad;
#pragma unroll
                for (int r = 0; r < R0; r++) {
                    const int e = base + r * B0;
                    const bool ok = e >= 0 && (unsigned)e < in_left;  // the pad samples before the buffer read as zero
                    const c32 x = in_g[ok ? e : 0];
                    v[q * R0 + r] = ok ? x : mk(0.f, 0.f);
                }
            }
        }
        transform_regs<NF, -1, false, G>(v, twf, lds, tid);
        // ---- spectrum multiply in registers, permuted into the inverse plan's input order ----
        c32 w[16];
#pragma unroll
        for (int q = 0; q < 16 / RL; q++) {
#pragma unroll
            for (int r = 0; r < RL; r++) w[q * RL + r] = cmul(v[q * RL + irev<RL>(r)], Hreg[q * RL + irev<RL>(r)]);
        }
        if constexpr (NP > 1) __syncthreads();  // forward transform's LDS reads are done
        if constexpr (SHARE) transform_regs<NF, 1, true, G, 0, true>(w, twf, lds, tid);
        else transform_regs<NF, 1, true, G>(w, twi, lds, tid);
        // ---- store the valid part (n >= ntaps-1), decimated ---------------------------------
        if (decim == 1) {
            // raw buffer over this group's outputs: a lane whose sample lies outside the block's store window [s0, s0+L)
            // gets the offset 0xffffffff, so the range check drops its store like it drops the ones past n_y -- two integer
            // operations per store instead of three compares, a mask update and a branch
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + g0), 0, (int)(y_left > 0x0ffffffeu ? 0x7ffffff0u : y_left * 8u), 0x00020000);
            int inval = inval0;
            asm volatile("" : "+v"(inval));  // one register, bits extracted per store (not 16 hoisted selects)
#pragma unroll
            for (int q = 0; q < 16 / RO; q++) {
                const int g = tid + TH * q, fr = g / BO, j = g % BO;
                const unsigned off0 = (unsigned)(fr * L + j - s0) * 8u;
#pragma unroll
                for (int s = 0; s < RO; s++) {
                    const int k = q * RO + s;
                    const unsigned off = (off0 + (unsigned)(orev<RO>(s) * BO * 8)) | (unsigned)__builtin_amdgcn_sbfe(inval, k, 1);
                    f2v o;
                    o.x = w[k].x;
                    o.y = w[k].y;
                    if (accumulate) o += __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(ro, off, 0, 0));  // out of range reads 0, and the store is dropped
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2g, o), ro, off, 0, 2 /* nt */);
                }
            }
        } else {
            const unsigned g_phase = (unsigned)(g0 % decim);  // uniform: one 64-bit division per group, 32-bit ones per element
            const long long g_quot = g0 / decim;
#pragma unroll
            for (int q = 0; q < 16 / RO; q++) {
                const int g = tid + TH * q, fr = g / BO, j = g % BO;
                const int rel0 = fr * L + j - s0;
#pragma unroll
                for (int s = 0; s < RO; s++) {
                    const int n = j + orev<RO>(s) * BO;
                    const int rel = rel0 + orev<RO>(s) * BO;  // output index relative to g0
                    if (n >= s0 && n < s0 + L && (unsigned)rel < y_left) {
                        const unsigned t = g_phase + (unsigned)rel;  // (g0 + rel) mod decim == (g0 mod decim + rel) mod decim
                        if (t % (unsigned)decim == 0) {
                            c32 *o = out + g_quot + t / (unsigned)decim;
                            c32 z = w[q * RO + s];
                            if (accumulate) { z.x += o->x; z.y += o->y; }
                            *o = z;
                        }
                    }
                }
            }
        }
        if constexpr (NP > 1) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// partitioned fast convolution of a long filter (more than 2048 taps) in ONE pass: per output block the P segment spectra
// are applied to P forward transforms of the correspondingly delayed input blocks and SUMMED in the frequency domain, then
// one inverse transform: y_blk = IFFT( sum_p FFT(x delayed by the taps before segment p) * H_p ).  P + 1 transforms per
// block instead of the 2 P of P separate passes, the input is read from HBM once (the P spans of a block overlap with the
// spans of the neighbouring blocks: L2) and y is written once, never re-read.  NF = 4096; the segment spectra are read
// from L2 per block (16 values per thread and segment) instead of living in registers.
// ------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(G::TH, 2) void k_ols_part(const c32 *__restrict__ in, c32 *__restrict__ out, const c32 *__restrict__ Hspec,  // [nseg][NF]
                                                      const c32 *__restrict__ tw_fwd, int ktot, int nseg, int seg_len, int seg_first,
                                                      int decim, int L, int s0, long long n_y, int ngroups, int xcd_map)
{
    constexpr int NF = 4096;
    using PF = Plan<NF, false>;
    using PI = Plan<NF, true>;
    constexpr int TH = G::TH, PTS = G::PTS, F = G::F, NP = PF::NP;
    static_assert((PF::L % 4) == 0, "all-radix-16 size: one twiddle set serves both directions");
    __shared__ c32 lds[PTS];
    const int tid0 = threadIdx.x;
    TwRegs<NF> twf;
    load_twiddles<NF, false, G>(twf, tid0, tw_fwd);
    constexpr int RL = PF::radix(NP - 1), BL = NF / RL, R0 = PF::radix(0), B0 = NF / R0, RO = PI::radix(NP - 1), BO = NF / RO;

    // XCD-contiguous group mapping as in k_ols: here the P input spans of a block overlap the neighbouring blocks' almost entirely
    const bool xmap = (gridDim.x & 7) == 0 && xcd_map;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, chunk = (ngroups + 7) >> 3;
    for (int q = xmap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; q < (xmap ? chunk : ngroups); q += xmap ? per_xcd : (int)gridDim.x) {
        const int grp = xmap ? xcd * chunk + q : q;
        if (grp >= ngroups) break;
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long long g0 = (long long)grp * F * L;
        const long long y_left64 = n_y - g0;
        const unsigned y_left = y_left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)(y_left64 > 0 ? y_left64 : 0);
        c32 w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = mk(0.f, 0.f);
        for (int sg = 0; sg < nseg; sg++) {
            // segment sg: tp taps, applied to the input delayed by the taps before it = the buffer read `shift` samples later
            const int tp = sg == 0 ? seg_first : seg_len;
            const long long shift = sg == 0 ? (long long)ktot - seg_first : (long long)(nseg - 1 - sg) * seg_len;
            const int pad = s0 - (tp - 1);
            const c32 *__restrict__ in_g = in + shift + g0;
            const long long in_left64 = n_y + tp - 1 - g0;
            const unsigned in_left = in_left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)(in_left64 > 0 ? in_left64 : 0);
            c32 v[16];
            if (g0 > 0 || shift >= pad) {  // the pad samples in front of the block exist in the buffer
                const long long left_b = (in_left64 + pad) * 8;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(in_g - pad), 0, left_b <= 0 ? 0 : (left_b > 0x7ffffff8LL ? 0x7ffffff8 : (int)left_b), 0x00020000);
#pragma unroll
                for (int q = 0; q < 16 / R0; q++) {
                    const int g = tid + TH * q, fr = g / B0, j = g % B0;
                    const unsigned off = (unsigned)(fr * L + j) * 8u;
#pragma unroll
                    for (int r = 0; r < R0; r++) {
                        const f2v x = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, off + (unsigned)(r * B0 * 8), 0, 0));
                        v[q * R0 + r] = mk(x.x, x.y);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16 / R0; q++) {
                    const int g = tid + TH * q, fr = g / B0, j = g % B0;
                    const int base = fr * L + j - pad;
#pragma unroll
                    for (int r = 0; r < R0; r++) {
                        const int e = base + r * B0;
                        const bool ok = e >= 0 && (unsigned)e < in_left;  // in front of the buffer: feeds only outputs that are not stored
                        const c32 x = in_g[ok ? e : 0];
                        v[q * R0 + r] = ok ? x : mk(0.f, 0.f);
                    }
                }
            }
            transform_regs<NF, -1, false, G>(v, twf, lds, tid);
            const c32 *__restrict__ Hs = Hspec + (size_t)sg * NF;
#pragma unroll
            for (int q = 0; q < 16 / RL; q++) {
                const int j = (tid + TH * q) % BL;
#pragma unroll
                for (int r = 0; r < RL; r++) {
                    const int src = q * RL + irev<RL>(r);
                    const c32 t = cmul(v[src], Hs[j + orev<RL>(irev<RL>(r)) * BL]);
                    w[q * RL + r].x += t.x;
                    w[q * RL + r].y += t.y;
                }
            }
            __syncthreads();  // this transform's LDS reads are done before the next one (or the inverse) writes
        }
        transform_regs<NF, 1, true, G, 0, true>(w, twf, lds, tid);
        const unsigned g_phase = (unsigned)(g0 % decim);
        const long long g_quot = g0 / decim;
#pragma unroll
        for (int q = 0; q < 16 / RO; q++) {
            const int g = tid + TH * q, fr = g / BO, j = g % BO;
            const int rel0 = fr * L + j - s0;
#pragma unroll
            for (int s = 0; s < RO; s++) {
                const int n = j + orev<RO>(s) * BO;
                const int rel = rel0 + orev<RO>(s) * BO;
                if (n >= s0 && n < s0 + L && (unsigned)rel < y_left) {
                    if (decim == 1) st_stream(out + g0 + (unsigned)rel, w[q * RO + s]);
                    else {
                        const unsigned t = g_phase + (unsigned)rel;
                        if (t % (unsigned)decim == 0) out[g_quot + t / (unsigned)decim] = w[q * RO + s];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// uniformly partitioned overlap-save (2 .. 5 segments = 2049 .. 10240 taps): segments of exactly L = 2048 taps, the block
// length, so the input block that segment p needs for output block b IS the input block of output block b - p.  A workgroup
// walks a contiguous run of blocks and keeps the spectra of the last P - 1 input blocks in registers:
//     X_b = FFT(in block b);   y_blk(b) = IFFT( sum_p X_(b-p) * H_p )
// = TWO transforms per 2048 outputs whatever the filter length (k_ols_part: P + 1 per 2049 - 2600), plus P - 1 forward
// transforms at the start of a run.  Input samples in front of / behind the buffer are zeros: they meet only the zero
// padding of the last segment or outputs that are not stored.  H_p are read from L1/L2 per block (16 values per thread, segment).
// ------------------------------------------------------------------------------------
template <class G, int P>
__global__ __launch_bounds__(G::TH, 2) void k_ols_ups(const c32 *__restrict__ in, c32 *__restrict__ out, const c32 *__restrict__ Hspec,  // [P][NF]
                                                     const c32 *__restrict__ tw_fwd, int ktot, int decim, long long n_y, int nblocks, int run)
{
    constexpr int NF = 4096, L = 2048, S0 = 2048;
    using PF = Plan<NF, false>;
    using PI = Plan<NF, true>;
    constexpr int TH = G::TH, NP = PF::NP;
    static_assert(G::F == 1 && G::PTS == NF && TH * 16 == NF, "one block per workgroup iteration, 16 points per thread");
    static_assert((PF::L % 4) == 0, "all-radix-16 size: one twiddle set serves both directions");
    constexpr int RL = PF::radix(NP - 1), BL = NF / RL, R0 = PF::radix(0), B0 = NF / R0, RO = PI::radix(NP - 1), BO = NF / RO;
    static_assert(RL == 16 && R0 == 16 && RO == 16, "4096 = 16^3");
    __shared__ c32 lds[NF];
    const int tid0 = threadIdx.x;
    // P >= 4: the twiddles are re-read (L1) per transform; their 24 registers go to the spectrum ring (P = 3 with the reload: 767 us per 2^26 samples against 674)
    constexpr bool RELOAD = P >= 4;
    TwRegs<NF> twf0;
    if constexpr (!RELOAD) load_twiddles<NF, false, G>(twf0, tid0, tw_fwd);
    const int b_first = blockIdx.x * run, b_end = min(nblocks, b_first + run);
    const long long n_in = n_y + ktot - 1;  // samples in the buffer (history in front)

    // half h (0: n < 2048, 1: n >= 2048) of input block b = in[] index b * L - S0 + ktot - 1 + n, n < NF: 8 values per thread,
    // v[8 h + r] = element tid + (8 h + r) * 256; zeros outside the buffer
    auto load_half = [&](int b, int h, int tid, c32 (&d)[8]) {
        const long long start = (long long)b * L - S0 + ktot - 1 + (long long)h * (NF / 2);
        if (start >= 0) {
            const long long left_b = (n_in - start) * 8;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void *)(in + start), 0, left_b <= 0 ? 0 : (left_b > 0x7ffffff8LL ? 0x7ffffff8 : (int)left_b), 0x00020000);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const f2v t = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(tid + r * B0) * 8u, 0, 0));
                d[r] = mk(t.x, t.y);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const long long e = start + tid + r * B0;
                const bool ok = e >= 0 && e < n_in;
                const c32 t = in[ok ? e : 0];
                d[r] = ok ? t : mk(0.f, 0.f);
            }
        }
    };
    // spectrum of the block whose halves are lo / hi, in the register order the inverse transform consumes
    auto spectrum = [&](const c32 (&lo)[8], const c32 (&hi)[8], int tid, c32 (&x)[16]) {
        c32 v[16];
#pragma unroll
        for (int r = 0; r < 8; r++) { v[r] = lo[r]; v[8 + r] = hi[r]; }
        if constexpr (RELOAD) {
            TwRegs<NF> tw;
            load_twiddles<NF, false, G>(tw, tid, tw_fwd);
            transform_regs<NF, -1, false, G>(v, tw, lds, tid);
        } else {
            transform_regs<NF, -1, false, G>(v, twf0, lds, tid);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = v[irev<RL>(r)];
        __syncthreads();  // this transform's LDS reads are done before the next one (or the inverse) writes
    };

    // consecutive input blocks overlap by half: the upper half of block b - 1 is the lower half of block b.  `keep` carries it,
    // `nx` is the upper half of the block about to be transformed, fetched one iteration ahead.
    // (P >= 3: the ring takes those 32 registers; both halves are loaded when they are needed)
    constexpr bool PREF = P == 2;
    c32 ring[P - 1][16];  // ring[p - 1] = spectrum of input block b - p
    c32 keep[8], nx[8];
    {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if constexpr (PREF) load_half(b_first - (P - 1), 0, tid, keep);
#pragma unroll
        for (int p = P - 1; p >= 1; p--) {
            if constexpr (!PREF) load_half(b_first - p, 0, tid, keep);
            load_half(b_first - p, 1, tid, nx);
            spectrum(keep, nx, tid, ring[p - 1]);
            if constexpr (PREF) {
#pragma unroll
                for (int r = 0; r < 8; r++) keep[r] = nx[r];
            }
        }
        if constexpr (PREF) load_half(b_first, 1, tid, nx);
    }
    for (int b = b_first; b < b_end; b++) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        c32 x0[16], w[16];
        if constexpr (PREF) {
            c32 hi[8];
#pragma unroll
            for (int r = 0; r < 8; r++) hi[r] = nx[r];
            if (b + 1 < b_end) load_half(b + 1, 1, tid, nx);  // in flight under both transforms of this block
            spectrum(keep, hi, tid, x0);
#pragma unroll
            for (int r = 0; r < 8; r++) keep[r] = hi[r];
        } else {
            c32 lo[8], hi[8];
            load_half(b, 0, tid, lo);
            load_half(b, 1, tid, hi);
            spectrum(lo, hi, tid, x0);
        }
        const int j = tid % BL;
#pragma unroll
        for (int r = 0; r < 16; r++) w[r] = cmul(x0[r], Hspec[j + orev<RL>(irev<RL>(r)) * BL]);
#pragma unroll
        for (int p = 1; p < P; p++) {
            const c32 *__restrict__ Hs = Hspec + (size_t)p * NF;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const c32 t = cmul(ring[p - 1][r], Hs[j + orev<RL>(irev<RL>(r)) * BL]);
                w[r].x += t.x;
                w[r].y += t.y;
            }
        }
#pragma unroll
        for (int p = P - 1; p >= 2; p--)
#pragma unroll
            for (int r = 0; r < 16; r++) ring[p - 1][r] = ring[p - 2][r];
#pragma unroll
        for (int r = 0; r < 16; r++) ring[0][r] = x0[r];
        if constexpr (RELOAD) {
            TwRegs<NF> tw;
            load_twiddles<NF, false, G>(tw, tid, tw_fwd);
            transform_regs<NF, 1, true, G, 0, true>(w, tw, lds, tid);
        } else {
            transform_regs<NF, 1, true, G, 0, true>(w, twf0, lds, tid);
        }
        const long long g0 = (long long)b * L;
        const long long y_left64 = n_y - g0;
        const unsigned y_left = y_left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)(y_left64 > 0 ? y_left64 : 0);
        const unsigned g_phase = (unsigned)(g0 % decim);
        const long long g_quot = g0 / decim;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int n = tid + orev<RO>(s) * BO;
            const int rel = n - S0;
            if (n >= S0 && (unsigned)rel < y_left) {
                if (decim == 1) st_stream(out + g0 + (unsigned)rel, w[s]);
                else {
                    const unsigned t = g_phase + (unsigned)rel;
                    if (t % (unsigned)decim == 0) out[g_quot + t / (unsigned)decim] = w[s];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// direct-form FIR, decimation 1.  256 threads x 8 CONSECUTIVE outputs; the thread
// slides an 8+8 register window over its inputs, so U+K-1 LDS reads feed U*K FMAs.
// The tile is stored transposed in LDS -- sample n at slot (n%8)*S + n/8 with
// S = 2 (mod 16) -- which makes both the coalesced fill (ds_write_b64) and the
// per-lane window reads (ds_read_b64) bank-conflict free.
// ------------------------------------------------------------------------------------
constexpr int kTdThreads = 256, kTdU = 8, kTdTile = kTdThreads * kTdU;

__host__ __device__ inline int td_rows(int kpad)
{
    int rows = (kTdTile + kpad + kTdU) / kTdU + 1;
    if (rows < 260) rows = 260;  // the output transpose below needs 8 x 260 slots
    return rows + ((2 - rows) & 15);  // smallest S >= rows with S % 16 == 2
}

template <bool CTAPS>
__global__ __launch_bounds__(kTdThreads, 4) void k_fir_td(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                       const float *__restrict__ taps_rev,  // reversed, zero padded to kpad
                                                       int K, long long n_out /* undecimated outputs */, int kpad /* K rounded up to kTdU */,
                                                       int decim)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32 *tile = (c32 *)smem;
    const int tid = threadIdx.x;
    const int S = td_rows(kpad);
    const int span = kTdTile + kpad + kTdU;
    const long long ntiles = (n_out + kTdTile - 1) / kTdTile;
    const long long n_in = n_out + K - 1;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long long base = tl * kTdTile;
        const c32 *__restrict__ in_t = in + base;
        const long long left64 = n_in - base;
        const unsigned left = left64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)left64;
        __syncthreads();
        // fill in batches of 8 loads per thread so the loads are all in flight before the LDS writes.  (Fetching the next
        // tile into registers during the FMAs was measured too: the 18 extra registers cost the fourth wave per SIMD, -6 %.)
        for (int i0 = tid; i0 < span; i0 += 8 * kTdThreads) {
            c32 st[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q * kTdThreads;
                const bool ok = i < span && (unsigned)i < left;
                const c32 x = in_t[ok ? i : 0];
                st[q] = ok ? x : mk(0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q * kTdThreads;
                if (i < span) tile[(i & (kTdU - 1)) * S + (i >> 3)] = st[q];
            }
        }
        __syncthreads();
        // (re, im) pairs as explicit 2-vectors -> v_pk_fma_f32: two FMAs per lane per instruction.  The 16-sample window lives
        // in three rotating 8-sample register blocks: a step multiplies blocks (cur, next) while the LDS reads of the block
        // after them and the scalar loads of the next 8 taps are in flight, so no step waits on its own loads, and the
        // rotation (unrolled by three) costs no register moves.
        constexpr int TF = CTAPS ? 2 : 1;
        f2v acc[kTdU], b0[kTdU], b1[kTdU], b2[kTdU];
        float h0[TF * kTdU], h1[TF * kTdU];
        const f2v *tl2 = (const f2v *)tile;
        const int nblk = kpad / kTdU;
#pragma unroll
        for (int u = 0; u < kTdU; u++) {
            acc[u] = (f2v){0.f, 0.f};
            b0[u] = tl2[u * S + tid];      // x[tid*8 + u]
            b1[u] = tl2[u * S + tid + 1];  // x[tid*8 + 8 + u]
        }
#pragma unroll
        for (int i = 0; i < TF * kTdU; i++) h0[i] = taps_rev[i];  // uniform -> scalar loads
        auto step = [&](const f2v (&cur)[kTdU], const f2v (&nxt)[kTdU], f2v (&pre)[kTdU], const float (&hc)[TF * kTdU],
                        float (&hn)[TF * kTdU], int blk) {
            int row = tid + blk + 2;  // the block after `nxt` (one row past the last one used exists in the tile: td_rows)
            asm volatile("" : "+v"(row));  // one set of reads per step: keeps the compiler from pairing the reads of two steps (deeper, 180 VGPRs)
#pragma unroll
            for (int u = 0; u < kTdU; u++) pre[u] = tl2[u * S + row];
#pragma unroll
            for (int i = 0; i < TF * kTdU; i++) hn[i] = taps_rev[TF * kTdU * (blk + 1) + i];  // padded by one block (set_taps)
#pragma unroll
            for (int i = 0; i < kTdU; i++) {
#pragma unroll
                for (int u = 0; u < kTdU; u++) {
                    const f2v x = (i + u < kTdU) ? cur[(i + u) & (kTdU - 1)] : nxt[(i + u) & (kTdU - 1)];
                    if constexpr (CTAPS) {
                        const float hr = hc[2 * i], hi = hc[2 * i + 1];
                        acc[u] = __builtin_elementwise_fma(x, (f2v){hr, hr}, acc[u]);                   // (hr*x.x, hr*x.y)
                        acc[u] = __builtin_elementwise_fma((f2v){x.y, x.x}, (f2v){-hi, hi}, acc[u]);  // (-hi*x.y, hi*x.x)
                    } else {
                        acc[u] = __builtin_elementwise_fma(x, (f2v){hc[i], hc[i]}, acc[u]);
                    }
                }
            }
        };
        int blk = 0;
        for (; blk + 6 <= nblk; blk += 6) {
            step(b0, b1, b2, h0, h1, blk);
            step(b1, b2, b0, h1, h0, blk + 1);
            step(b2, b0, b1, h0, h1, blk + 2);
            step(b0, b1, b2, h1, h0, blk + 3);
            step(b1, b2, b0, h0, h1, blk + 4);
            step(b2, b0, b1, h1, h0, blk + 5);
        }
        // up to five blocks left, same rotation
        if (blk < nblk) { step(b0, b1, b2, h0, h1, blk); blk++;
            if (blk < nblk) { step(b1, b2, b0, h1, h0, blk); blk++;
                if (blk < nblk) { step(b2, b0, b1, h0, h1, blk); blk++;
                    if (blk < nblk) { step(b0, b1, b2, h1, h0, blk); blk++;
                        if (blk < nblk) { step(b1, b2, b0, h0, h1, blk); blk++; } } } } }
        // The thread holds 8 CONSECUTIVE outputs; storing them directly makes every wave instruction write 64 separate
        // 8-byte pieces 64 bytes apart (measured: the kernel was bound by that, not by the FMAs).  Transpose through the
        // tile (row stride 260 slots: conflict free both ways) so that lanes store consecutive samples.
        __syncthreads();  // every wave is done reading the input tile
        constexpr int OS = 260;
#pragma unroll
        for (int u = 0; u < kTdU; u++) tile[u * OS + tid] = mk(acc[u].x, acc[u].y);
        __syncthreads();
        c32 *__restrict__ out_t = out + base;
        const long long oleft = n_out - base;
        if (decim == 1) {
#pragma unroll
            for (int k = 0; k < kTdU; k++) {
                const int o = tid + kTdThreads * k;  // output o of the tile = thread o/8, slot o%8
                if (o < oleft) st_stream(out_t + o, tile[(o & (kTdU - 1)) * OS + (o >> 3)]);
            }
        } else {  // small decimations: every output is computed, every decim-th kept (cheaper than one output per thread)
            const unsigned ph = (unsigned)(base % decim);
            const long long q0 = base / decim;
#pragma unroll
            for (int k = 0; k < kTdU; k++) {
                const int o = tid + kTdThreads * k;
                const unsigned t = ph + (unsigned)o;
                if (o < oleft && t % (unsigned)decim == 0) out[q0 + t / (unsigned)decim] = tile[(o & (kTdU - 1)) * OS + (o >> 3)];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// direct-form FIR on the fp32 matrix cores (real taps, decimation 1).
//   y[base + 16 i + j] = sum_k' xin[base + 16 i + k'] * hrev[k' - j]      (xin = history-prefixed input, hrev = reversed taps)
// is a matrix product D[i][j] = sum_k' A[i][k'] B[k'][j] with A a window matrix of the input (row i starts 16 samples after
// row i-1) and B the Toeplitz matrix of the taps (zero outside [0, K)); k' runs over K + 15 values, four per
// v_mfma_f32_16x16x4_f32 (exact fp32 fused multiply-adds, the same peak as the packed vector FMAs but issued off the vector
// ALU, which is what limits k_fir_td).  Real and imaginary parts are two real products with the same B.
// A wave owns 4 blocks of 256 consecutive outputs (8 independent accumulators: no dependent MFMA back to back); the
// workgroup's 4096 + 4*KK input samples sit in LDS as separate re / im planes addressed n + 2*(n/16) (the pad makes the 16 rows
// of an A operand, 16 floats apart, fall into different banks); the tap table is read from LDS too (one word per lane and
// step).  The D layout (row = 4*(lane/16) + reg, col = lane%16) puts 16 consecutive outputs on 16 consecutive lanes: every
// store instruction writes four whole 128-byte lines straight from the accumulators -- no output transpose.
// The next tile's samples are fetched into registers while the current one is multiplied.  (One-wave workgroups without any
// workgroup barrier were measured too: 3-5 % slower -- three waves per SIMD and more halo re-reads.)
// ------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kMfTile = 4096, kMfThreads = 256, kMfPre = 18;  // prefetch registers: up to 18 * 256 = 4608 samples of span

__host__ __device__ inline int mf_pad(int n) { return n + ((n >> 4) << 1); }  // 16 samples -> 18 slots: rows 18 apart, the two row groups of a half-wave on even / odd banks (stride 17 left 48 % of the LDS cycles in conflicts)

template <bool CTAPS, bool DEC>
__global__ __launch_bounds__(kMfThreads, 2) void k_fir_mfma(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                            const float *__restrict__ hb,  // hb[m + 15] = hrev[m], zeros around: 4*KK + 16 floats (complex taps: a second table with the imaginary parts follows)
                                                            int K, int KK, long long n_out /* undecimated outputs */, int decim)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int span = kMfTile + 4 * KK;          // samples the tile's outputs read (K + 15 rounded up to the MFMA step)
    const int nq = (span + kMfThreads - 1) / kMfThreads;  // loads per thread and tile (<= kMfPre, the launcher checks)
    const int plane = mf_pad(nq * kMfThreads + 16) + 1;    // whole rounds of the fill, and one MFMA step of over-read
    float *xr = (float *)smem, *xi = xr + plane, *tb = xi + plane;
    const int tlen = 4 * KK + 24;  // table slots in LDS (one step of over-read included)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    for (int i = tid; i < tlen; i += kMfThreads) {
        tb[i] = i < 4 * KK + 16 ? hb[i] : 0.f;
        if constexpr (CTAPS) tb[tlen + i] = i < 4 * KK + 16 ? hb[4 * KK + 16 + i] : 0.f;
    }
    const long long ntiles = (n_out + kMfTile - 1) / kMfTile, n_in = n_out + K - 1;

    f2v pre[kMfPre];
    auto fetch = [&](long long tl) {
        const long long base = tl * kMfTile, left = n_in - base;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(in + base), 0, left * 8 > 0x7ffffff8LL ? 0x7ffffff8 : (int)(left * 8), 0x00020000);
#pragma unroll
        for (int q = 0; q < kMfPre; q++) {
            const int i = tid + q * kMfThreads;
            if (q < nq) pre[q] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)i * 8u, 0, 0));  // past the span: unused slots
        }
    };
    if ((long long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        __syncthreads();  // the previous tile's operand reads are done (first pass: the tap table is in place)
#pragma unroll
        for (int q = 0; q < kMfPre; q++) {
            const int i = tid + q * kMfThreads;
            if (q < nq) { xr[mf_pad(i)] = pre[q].x; xi[mf_pad(i)] = pre[q].y; }
        }
        if (tl + gridDim.x < ntiles) fetch(tl + gridDim.x);
        __syncthreads();
        v4f dr[4], di[4];
#pragma unroll
        for (int b = 0; b < 4; b++) dr[b] = di[b] = (v4f){0.f, 0.f, 0.f, 0.f};
        // operands of step kk: A[row c][k' = 4 kk + g] of block b = sample n + 256 b, n = wave*1024 + 16 c + g + 4 kk, at padded
        // slot n + 2*(n/16) + 288 b (256 b is a multiple of 16); B = tb[4 kk + g - c + 15].  The operands of the next step are read
        // before the current step's eight MFMAs are issued.
        int n = wave * 1024 + 16 * c + g;
        const float *tbl = tb + (g - c + 15);
        float ar[4], ai[4], bv, bw = 0.f;  // bw: imaginary part of the tap (complex taps)
        {
            const int a = mf_pad(n);
            bv = tbl[0];
            if constexpr (CTAPS) bw = tbl[tlen];
#pragma unroll
            for (int b = 0; b < 4; b++) { ar[b] = xr[a + 288 * b]; ai[b] = xi[a + 288 * b]; }
        }
        for (int kk = 0; kk < KK; kk++) {
            n += 4;
            const int a = mf_pad(n);  // one step past the last one stays inside the planes (sized for it)
            float nr[4], ni[4], nw = 0.f;
            const float nb = tbl[4 * kk + 4];
            if constexpr (CTAPS) nw = tbl[tlen + 4 * kk + 4];
#pragma unroll
            for (int b = 0; b < 4; b++) { nr[b] = xr[a + 288 * b]; ni[b] = xi[a + 288 * b]; }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                dr[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[b], bv, dr[b], 0, 0, 0);
                di[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[b], bv, di[b], 0, 0, 0);
            }
            if constexpr (CTAPS) {  // (xr + j xi)(hr + j hi): re -= xi hi, im += xr hi
                const float mw = -bw;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    dr[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[b], mw, dr[b], 0, 0, 0);
                    di[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[b], bw, di[b], 0, 0, 0);
                }
            }
            bv = nb;
            bw = nw;
#pragma unroll
            for (int b = 0; b < 4; b++) { ar[b] = nr[b]; ai[b] = ni[b]; }
        }
        // D[row = 4 g + reg][col = c] = y[base + 64 g + 16 reg + c]
        const long long obase = tl * kMfTile + wave * 1024, oleft = n_out - obase;
        if constexpr (!DEC) {
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + obase), 0, oleft <= 0 ? 0 : (oleft * 8 > 0x7ffffff8LL ? 0x7ffffff8 : (int)(oleft * 8)), 0x00020000);
#pragma unroll
            for (int b = 0; b < 4; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    f2v o;
                    o.x = dr[b][r];
                    o.y = di[b][r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2g, o), ro, (unsigned)(b * 256 + 64 * g + 16 * r + c) * 8u, 0, 2 /* nt */);
                }
            }
        } else {  // small decimations: every output is computed, every decim-th kept (the matrix cores have the headroom)
            const unsigned ph = (unsigned)(obase % decim);
            const long long q0 = obase / decim;
#pragma unroll
            for (int b = 0; b < 4; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = b * 256 + 64 * g + 16 * r + c;
                    const unsigned t = ph + (unsigned)o;
                    if (o < oleft && t % (unsigned)decim == 0) out[q0 + t / (unsigned)decim] = mk(dr[b][r], di[b][r]);
                }
            }
        }
    }
}

// Decimations above 8 in the time domain.  The per-output kernel below (k_fir_td_dec) lets lane m read x[m D + k]: lanes D samples apart,
// every load instruction 64 different lines, K times over -- 65 taps at decimation 16 ran at 16 GS/s of input.  Here the span of a tile of
// outputs is staged in LDS with coalesced loads (slot i + i/32: a wave's reads, D slots apart, meet different banks for every D) and
// each thread forms its outputs from there: the input is read from HBM once, whatever D.
constexpr int kDlThreads = 256, kDlSpan = 8192;  // samples of input per tile (64 KiB: two workgroups per CU)
__host__ __device__ inline int dl_slot(int i) { return i + (i >> 5); }

template <bool CTAPS>
__global__ __launch_bounds__(kDlThreads) void k_fir_dec_lds(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ taps_rev, int K,
                                                            int decim, long long n_out, int tile_out)
{
    extern __shared__ __attribute__((aligned(16))) c32 dl_x[];
    const int tid = threadIdx.x;
    const long long ntiles = (n_out + tile_out - 1) / tile_out;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long o0 = tile * tile_out, left = n_out - o0;
        const int no = left < tile_out ? (int)left : tile_out;
        const c32 *__restrict__ src = in + o0 * decim;
        const int span = (no - 1) * decim + K;
        __syncthreads();  // the previous tile's reads are done
        for (int i = tid; i < span; i += kDlThreads) {
            const f2v t = __builtin_nontemporal_load((const f2v *)src + i);
            dl_x[dl_slot(i)] = mk(t.x, t.y);
        }
        __syncthreads();
        for (int o = tid; o < no; o += kDlThreads) {
            const int base = o * decim;
            c32 acc = mk(0.f, 0.f);
            auto step = [&](const c32 v, int k) {
                if constexpr (CTAPS) {
                    const float hr = taps_rev[2 * k], hi = taps_rev[2 * k + 1];
                    acc.x += hr * v.x - hi * v.y;
                    acc.y += hr * v.y + hi * v.x;
                } else {
                    const float h = taps_rev[k];
                    acc.x += h * v.x;
                    acc.y += h * v.y;
                }
            };
            // eight samples requested from LDS before the first is used (one at a time, a tap cost an LDS round trip: the tap count is
            // a run-time value and the compiler does not pipeline the loop itself); the sums keep their order
            int k = 0;
            for (; k + 8 <= K; k += 8) {
                c32 v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = dl_x[dl_slot(base + k + j)];
#pragma unroll
                for (int j = 0; j < 8; j++) step(v[j], k + j);
            }
            for (; k < K; k++) step(dl_x[dl_slot(base + k)], k);
            out[o0 + o] = acc;
        }
    }
}


// Round 6: the same staging for EVEN decimations with everything 16 bytes wide.  k_fir_dec_lds spends its time in the staging loop (8-byte loads, one
// LDS slot computed and written per sample: ~ 3.7 ps per input sample, "reads every sample once" at a third of the read bandwidth) and in a tap loop that
// mixes scalar tap loads with the LDS reads on one counter.  Here a thread stages sample PAIRS (global_load_dwordx4, all of a thread's loads in
// flight before the first LDS write), pair p sits at 16-byte unit p (+ p / 32 where D is a multiple of 8: dec2_pad_shift; an output's window starts on a
// pair when D is even), the taps sit in LDS as well, padded with zeros to a multiple of eight (broadcast reads),
// and a tap step is eight samples: four 16-byte sample reads, two (real taps) or four (complex taps) 16-byte tap reads, 16 / 32 FMAs in the
// reference's order (lib/fir_filter.cc:222-241: one running sum over k).  Samples past the input's end are staged as zeros.
typedef float v4f_ __attribute__((ext_vector_type(4)));
constexpr int kD2Threads = 256;
// pair p sits at 16-byte unit p + (p >> sh): which padding spreads a wave's reads over the banks depends on the lanes' stride (D / 2 units for even D, D
// for odd D); sh = 31: none.  Chosen per decimation by the launcher (dec2_pad_shift).
__host__ __device__ inline int d2_unit(int p, int sh) { return p + (p >> sh); }

// ODD decimations: an output's window starts on the second sample of a pair for every other output; the outputs of a round of 256 are dealt so that
// a wave's windows all start alike (waves 0, 1: even outputs, waves 2, 3: odd ones), the odd waves read five units per tap step and use them shifted by
// one sample.  Tiles hold an even number of outputs, so a tile still starts on a pair.
template <bool CTAPS, bool ODD>
__global__ __launch_bounds__(kD2Threads) void k_fir_dec2(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ taps_rev, int K, int KP,
                                                         int decim, long long n_out, int tile_out, int tap_units /* 16-byte units of taps */, int sh)
{
    extern __shared__ __attribute__((aligned(16))) v4f_ d2_lds[];
    v4f_ *const tl = d2_lds;                 // taps: KP floats (real) / 2 KP floats (complex), zero padded
    v4f_ *const xl = d2_lds + tap_units;     // sample pairs
    const int tid = threadIdx.x;
    for (int i = tid; i < tap_units * 4; i += kD2Threads) {
        const int nt = CTAPS ? 2 * K : K;
        ((float *)tl)[i] = i < nt ? taps_rev[i] : 0.f;
    }
    const long long n_in = (n_out - 1) * decim + K;  // the samples the outputs need
    const long long ntiles = (n_out + tile_out - 1) / tile_out;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long o0 = tile * tile_out, left = n_out - o0;
        const int no = left < tile_out ? (int)left : tile_out;
        const long long s0 = o0 * decim;  // even
        const v4f_ *__restrict__ src = (const v4f_ *)(in + s0);
        const int pairs = ((no - 1) * decim + KP + 1) / 2 + (ODD ? 1 : 0);
        const long long avail = n_in - s0;  // valid samples from s0 on
        __syncthreads();  // the previous tile's reads are done (and, the first time, nothing)
        for (int p0 = 0; p0 < pairs; p0 += 8 * kD2Threads) {
            v4f_ v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p = p0 + j * kD2Threads + tid;
                v[j] = (v4f_){0.f, 0.f, 0.f, 0.f};
                if (p < pairs) {
                    if (2LL * p + 1 < avail) v[j] = __builtin_nontemporal_load(src + p);
                    else if (2LL * p < avail) {
                        const c32 a = in[s0 + 2LL * p];
                        v[j] = (v4f_){a.x, a.y, 0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p = p0 + j * kD2Threads + tid;
                if (p < pairs) xl[d2_unit(p, sh)] = v[j];
            }
        }
        __syncthreads();
        for (int ob = 0; ob < no; ob += kD2Threads) {
            const int o = ob + (ODD ? 2 * (tid & 127) + (tid >> 7) : tid);
            if (o >= no) continue;
            const int pb = (o * decim) >> 1;
            const bool shifted = ODD && (tid >> 7) != 0;  // (wave-uniform)
            float ax = 0.f, ay = 0.f;
            for (int k = 0; k < KP; k += 8) {
                v4f_ sm[4];
                if (shifted) {
                    v4f_ un[5];
#pragma unroll
                    for (int j = 0; j < 5; j++) un[j] = xl[d2_unit(pb + (k >> 1) + j, sh)];
#pragma unroll
                    for (int j = 0; j < 4; j++) sm[j] = (v4f_){un[j][2], un[j][3], un[j + 1][0], un[j + 1][1]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) sm[j] = xl[d2_unit(pb + (k >> 1) + j, sh)];
                }
                if constexpr (CTAPS) {
                    v4f_ t[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) t[j] = tl[(k >> 1) + j];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        ax += t[j][0] * sm[j][0] - t[j][1] * sm[j][1];
                        ay += t[j][0] * sm[j][1] + t[j][1] * sm[j][0];
                        ax += t[j][2] * sm[j][2] - t[j][3] * sm[j][3];
                        ay += t[j][2] * sm[j][3] + t[j][3] * sm[j][2];
                    }
                } else {
                    const v4f_ t0 = tl[k >> 2], t1 = tl[(k >> 2) + 1];
                    ax += t0[0] * sm[0][0]; ay += t0[0] * sm[0][1];
                    ax += t0[1] * sm[0][2]; ay += t0[1] * sm[0][3];
                    ax += t0[2] * sm[1][0]; ay += t0[2] * sm[1][1];
                    ax += t0[3] * sm[1][2]; ay += t0[3] * sm[1][3];
                    ax += t1[0] * sm[2][0]; ay += t1[0] * sm[2][1];
                    ax += t1[1] * sm[2][2]; ay += t1[1] * sm[2][3];
                    ax += t1[2] * sm[3][0]; ay += t1[2] * sm[3][1];
                    ax += t1[3] * sm[3][2]; ay += t1[3] * sm[3][3];
                }
            }
            out[o0 + o] = mk(ax, ay);
        }
    }
}

// measured over D = 6 ... 100 (tools/r06_dec2_pad_probe.py, 65 taps): without padding the reads are conflict free or nearly so for every D that is not a
// multiple of 8 (D = 10: 587 GS/s of input against 341 with one unit per 16, D = 15: 577 against 310); multiples of 8 need it (D = 16: 400 without,
// 620-636 with; D = 32: 403 / 669) and one unit per 32 is the shift that is never bad there
inline int dec2_pad_shift(int decim) { return decim % 8 == 0 ? 5 : 31; }

template <bool CTAPS>
__global__ __launch_bounds__(256) void k_fir_td_dec(const c32 *__restrict__ in, c32 *__restrict__ out,
                                                    const float *__restrict__ taps_rev, int K, int decim, long long n_out)
{
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < n_out; m += (long long)gridDim.x * 256) {
        const c32 *x = in + m * decim;
        c32 acc = mk(0.f, 0.f);
        auto step = [&](const c32 s, int k) {
            if constexpr (CTAPS) {
                const float hr = taps_rev[2 * k], hi = taps_rev[2 * k + 1];
                acc.x += hr * s.x - hi * s.y;
                acc.y += hr * s.y + hi * s.x;
            } else {
                const float h = taps_rev[k];
                acc.x += h * s.x;
                acc.y += h * s.y;
            }
        };
        int k = 0;
        for (; k + 8 <= K; k += 8) {  // (eight samples requested before the first is used, see k_fir_dec_lds)
            c32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = x[k + j];
#pragma unroll
            for (int j = 0; j < 8; j++) step(v[j], k + j);
        }
        for (; k < K; k++) step(x[k], k);
        out[m] = acc;
    }
}

}  // namespace

struct mi355_filter {
    mi355_ctx *ctx;
    int decim, ntaps, complex_taps, use_time;
    int nf;                        // FFT size of the fast-convolution kernel (0 in time-domain mode)
    int nseg = 1, seg_len = 0, seg_first = 0;  // partitioned fast convolution of a long filter: segments, taps per segment, taps of segment 0
    std::vector<float> taps_host;  // ntaps floats or 2*ntaps floats
    float *d_taps_rev = nullptr;
    float *d_hb = nullptr;  // matrix-core direct form: zero-padded reversed taps (real taps only)
    int mf_kk = 0;          // MFMA steps per output block, 0 = kernel not applicable
    void *d_H = nullptr, *d_twf = nullptr, *d_twi = nullptr;
    void *d_Hu = nullptr;  // uniformly partitioned long filter (k_ols_ups): spectra of the 2048-tap segments, [ups][4096]
    int ups = 0;           // number of those segments (0: not applicable)
    std::vector<void *> retired;   // tables of earlier taps, kept alive for kernels still in flight (see retire_dev)
    size_t retired_bytes = 0, table_bytes = 0;
    HostPipe pipe;
    std::mutex lock;
};

namespace {

constexpr int kOlsMaxTaps = 2048;  // longest filter one NF = 4096 block can overlap with at least half of it new samples
constexpr int kUpsMaxSeg = 5;      // k_ols_ups keeps the spectra of the previous segments' input blocks in registers: up to 10240 taps

int pick_fft_size(int ntaps)
{
    // reference rule (lib/fft_filter.cc:72-97): 2 * 2^ceil(log2(ntaps)); never below 256 so
    // the plan is two radix-16 passes, never above 4096 (one workgroup iteration).
    // MI355_FILTER_FFT overrides (tuning aid).
    int nf = 2;
    while (nf < 2 * ntaps) nf <<= 1;
    if (nf < 256) nf = 256;
    // The reference size leaves between 50 % and 100 % of every block as new samples.  The kernel's rate per transformed
    // point is nearly flat in the size (G points/s measured on MI355X below), so a larger transform whose blocks carry a
    // larger share of new samples is faster: pick the best of the reference size and the next three.
    auto rate = [](int n) { return n <= 256 ? 413.0 : n == 512 ? 314.0 : n == 1024 ? 295.0 : n == 2048 ? 292.0 : 290.0; };
    const int s0 = (ntaps - 1 + 15) & ~15;
    int best = nf;
    double best_score = 0.0;
    for (int c = nf; c <= 4096 && c <= 8 * nf; c <<= 1) {
        const double score = rate(c) * (double)((c - s0) & ~15) / (double)c;
        if (score > best_score * 1.02) { best_score = score; best = c; }
    }
    nf = best;
    if (const char *e = getenv("MI355_FILTER_FFT")) {
        int v = atoi(e);
        if (v >= 2 * ntaps && v >= 64 && v <= 4096 && (v & (v - 1)) == 0) nf = v;
    }
    return nf;
}

// tables of the previous taps: possibly still read by kernels in flight on the caller's streams
void retire_dev(mi355_filter *h)
{
    size_t held = 0;
    for (void *p : {(void *)h->d_taps_rev, (void *)h->d_hb, (void *)h->d_H, (void *)h->d_Hu, (void *)h->d_twf, (void *)h->d_twi})
        if (p) h->retired.push_back(p);
    h->retired_bytes += h->table_bytes;
    held = h->retired_bytes;
    h->table_bytes = 0;
    h->d_taps_rev = nullptr;
    h->d_hb = nullptr;
    h->d_H = h->d_Hu = h->d_twf = h->d_twi = nullptr;
    h->ups = 0;
    if (held > ((size_t)64 << 20)) {  // a long series of retunes: pay one device-wide wait and start over
        (void)hipDeviceSynchronize();
        for (void *p : h->retired) (void)hipFree(p);
        h->retired.clear();
        h->retired_bytes = 0;
    }
}

void free_dev(mi355_filter *h)
{
    for (void *p : h->retired) (void)hipFree(p);
    h->retired.clear();
    h->retired_bytes = 0;
    (void)hipSetDevice(h->ctx->device);
    if (h->d_taps_rev) (void)hipFree(h->d_taps_rev);
    if (h->d_hb) (void)hipFree(h->d_hb);
    h->d_hb = nullptr;
    if (h->d_H) (void)hipFree(h->d_H);
    if (h->d_Hu) (void)hipFree(h->d_Hu);
    h->d_Hu = nullptr;
    h->ups = 0;
    if (h->d_twf) (void)hipFree(h->d_twf);
    if (h->d_twi) (void)hipFree(h->d_twi);
    h->d_taps_rev = nullptr;
    h->d_H = h->d_twf = h->d_twi = nullptr;
}

int upload_taps_impl(mi355_filter *h, const void *taps, int ntaps)
{
    MI355_REQUIRE(taps && ntaps >= 1, "taps must hold at least one tap");
    const int per = h->complex_taps ? 2 : 1;
    int nf = 0, nseg = 1, seg_len = ntaps, seg_first = ntaps;
    if (!h->use_time) {
        if (ntaps <= kOlsMaxTaps) nf = pick_fft_size(ntaps);
        else {
            // More taps than the largest single-workgroup transform can overlap: partitioned fast convolution.  The filter is
            // cut into nseg segments of <= 2048 taps; y = sum_p (h_p * x delayed by the taps before segment p).  Every segment
            // is one pass of the NF = 4096 overlap-save kernel over the same buffers (the history in front of the input holds
            // the delayed samples), the first pass stores y and the others accumulate into it.  (lib/fft_filter.cc:72-97 has
            // no such limit -- it just takes a larger FFT; the result is the same y.)
            nseg = (ntaps + kOlsMaxTaps - 1) / kOlsMaxTaps;
            seg_len = (ntaps + nseg - 1) / nseg;
            seg_first = ntaps - (nseg - 1) * seg_len;
            nf = 4096;
        }
    }
    MI355_HIP(hipSetDevice(h->ctx->device));
    // kernels of earlier device-path calls may still be reading the current tables: they are retired (kept until the handle goes,
    // or until 64 MiB have piled up), not freed -- hipFree would wait for the whole device, as the hipDeviceSynchronize() that
    // stood here did, stalling every other block in the flowgraph while one filter is retuned
    retire_dev(h);
    h->ntaps = ntaps;
    h->nf = nf;
    h->nseg = nseg;
    h->seg_len = seg_len;
    h->seg_first = seg_first;
    h->taps_host.assign((const float *)taps, (const float *)taps + (size_t)per * ntaps);
    // reversed taps for the direct form (lib/fir_filter.cc:187-189 reverses them too)
    const int kpad = (ntaps + kTdU - 1) / kTdU * kTdU;
    std::vector<float> rev((size_t)per * (kpad + kTdU), 0.0f);  // zero padded: the kernel loops to kpad without a bound test and prefetches one block of taps past it
    for (int k = 0; k < ntaps; k++)
        for (int c = 0; c < per; c++) rev[(size_t)per * k + c] = h->taps_host[(size_t)per * (ntaps - 1 - k) + c];
    MI355_HIP(hipMalloc((void **)&h->d_taps_rev, rev.size() * sizeof(float)));
        h->table_bytes += rev.size() * sizeof(float);
    MI355_HIP(mi355_upload(h->ctx, h->d_taps_rev, rev.data(), rev.size() * sizeof(float)));
    h->mf_kk = 0;
    {
        const int kk = (ntaps + 15 + 3) / 4;
        // the tile's input span must fit the prefetch registers (<= 497 taps).  Longer filters were tried with an in-place fill:
        // no faster than the vector kernel (one workgroup per CU at that LDS size), and the extra path cost the short ones 15 %
        if (kMfTile + 4 * kk <= kMfPre * kMfThreads) {
            const size_t tl = (size_t)4 * kk + 16;
            std::vector<float> hb(tl * per, 0.0f);  // [real parts | imaginary parts]
            for (int m = 0; m < ntaps; m++)
                for (int cpt = 0; cpt < per; cpt++) hb[cpt * tl + m + 15] = h->taps_host[(size_t)per * (ntaps - 1 - m) + cpt];
            MI355_HIP(hipMalloc((void **)&h->d_hb, hb.size() * sizeof(float)));
        h->table_bytes += hb.size() * sizeof(float);
            MI355_HIP(mi355_upload(h->ctx, h->d_hb, hb.data(), hb.size() * sizeof(float)));
            h->mf_kk = kk;
        }
    }
    if (nf) {
        // H[k] = sum_n (h[n]/NF) exp(-2 pi i k n / NF), evaluated in double (lib/fft_filter.cc:52-66)
        std::vector<float> H(2 * (size_t)nf * nseg), twf(2 * (size_t)nf), twi(2 * (size_t)nf);
        std::vector<double> cs(nf), sn(nf);
        for (int k = 0; k < nf; k++) {
            double a = -2.0 * M_PI * (double)k / (double)nf;
            cs[k] = cos(a); sn[k] = sin(a);
            twf[2 * k] = (float)cs[k]; twf[2 * k + 1] = (float)sn[k];
            twi[2 * k] = (float)cs[k]; twi[2 * k + 1] = (float)(-sn[k]);
        }
        for (int sgm = 0; sgm < nseg; sgm++) {
            const int t0 = sgm == 0 ? 0 : seg_first + (sgm - 1) * seg_len, tn = sgm == 0 ? seg_first : seg_len;  // taps [t0, t0 + tn)
            for (int k = 0; k < nf; k++) {
                double re = 0, im = 0;
                for (int n = 0; n < tn; n++) {
                    double hr = h->taps_host[(size_t)per * (t0 + n)], hi = h->complex_taps ? h->taps_host[2 * (size_t)(t0 + n) + 1] : 0.0;
                    int idx = (int)(((long long)k * n) % nf);
                    re += hr * cs[idx] - hi * sn[idx];
                    im += hr * sn[idx] + hi * cs[idx];
                }
                H[2 * ((size_t)sgm * nf + k)] = (float)(re / nf); H[2 * ((size_t)sgm * nf + k) + 1] = (float)(im / nf);
            }
        }
        size_t bytes = 2 * (size_t)nf * sizeof(float);
        if (nseg > 1 && ntaps <= kUpsMaxSeg * kOlsMaxTaps) {
            // uniform partition for k_ols_ups: segment p = taps [2048 p, 2048 p + 2048), the last one zero padded
            const int ups = (ntaps + kOlsMaxTaps - 1) / kOlsMaxTaps;
            std::vector<float> Hu(2 * (size_t)nf * ups);
            for (int sgm = 0; sgm < ups; sgm++) {
                const int t0 = sgm * kOlsMaxTaps, tn = std::min(kOlsMaxTaps, ntaps - t0);
                for (int k = 0; k < nf; k++) {
                    double re = 0, im = 0;
                    for (int n = 0; n < tn; n++) {
                        double hr = h->taps_host[(size_t)per * (t0 + n)], hi = h->complex_taps ? h->taps_host[2 * (size_t)(t0 + n) + 1] : 0.0;
                        int idx = (int)(((long long)k * n) % nf);
                        re += hr * cs[idx] - hi * sn[idx];
                        im += hr * sn[idx] + hi * cs[idx];
                    }
                    Hu[2 * ((size_t)sgm * nf + k)] = (float)(re / nf); Hu[2 * ((size_t)sgm * nf + k) + 1] = (float)(im / nf);
                }
            }
            MI355_HIP(hipMalloc(&h->d_Hu, bytes * ups));
        h->table_bytes += bytes * ups;
            MI355_HIP(mi355_upload(h->ctx, h->d_Hu, Hu.data(), bytes * ups));
            h->ups = ups;
        }
        MI355_HIP(hipMalloc(&h->d_H, bytes * nseg));
        h->table_bytes += bytes * nseg;
        MI355_HIP(hipMalloc(&h->d_twf, bytes));
        h->table_bytes += bytes;
        MI355_HIP(hipMalloc(&h->d_twi, bytes));
        h->table_bytes += bytes;
        MI355_HIP(mi355_upload(h->ctx, h->d_H, H.data(), bytes * nseg));
        MI355_HIP(mi355_upload(h->ctx, h->d_twf, twf.data(), bytes));
        MI355_HIP(mi355_upload(h->ctx, h->d_twi, twi.data(), bytes));
    }
    // (every upload above ran on the context's upload stream and was waited for there: mi355_upload)
    return MI355_OK;
}

// A failure half way (out of memory while the tables are rebuilt) must not leave a handle whose sizes describe tables
// that do not exist: the handle is marked empty and work() refuses with MI355_ERR_STATE until set_taps succeeds again.
int upload_taps(mi355_filter *h, const void *taps, int ntaps)
{
    const int rc = upload_taps_impl(h, taps, ntaps);
    if (rc != MI355_OK && rc != MI355_ERR_INVALID_ARG) {
        free_dev(h);
        h->ntaps = 0;
        h->nf = 0;
        h->mf_kk = 0;
    }
    return rc;
}

template <int NF, class G>
int launch_ols_g(mi355_filter *h, size_t nout, const void *in, void *out, hipStream_t st)
{
    constexpr int F = G::F, TH = G::TH, WAVES = TH / 64;
    const long long n_y = (long long)nout * h->decim;
    static const bool align_stores = getenv("MI355_OLS_ALIGN") ? atoi(getenv("MI355_OLS_ALIGN")) != 0 : true;
    // XCD-contiguous groups: +5-7 % when the buffers fit the 256 MiB Infinity Cache, -6 % at 1 GiB buffers, equal for long filters: off
    const int xcd_map = getenv("MI355_OLS_XCD_MAP") ? atoi(getenv("MI355_OLS_XCD_MAP")) : 0;
    static const bool one_pass = !getenv("MI355_OLS_PART_ONE_PASS") || atoi(getenv("MI355_OLS_PART_ONE_PASS")) != 0;
    if constexpr (NF == 4096) {
        const bool ups_on = !getenv("MI355_OLS_UPS") || atoi(getenv("MI355_OLS_UPS")) != 0;
        if (h->ups > 1 && ups_on) {
            const long long nblocks = (n_y + 2047) / 2048;
            if (nblocks > 0x7fffffffLL) { mi355_set_error("work() call too large"); return MI355_ERR_INVALID_ARG; }
            // a run of consecutive blocks per workgroup (the P - 1 warm-up transforms of a run are the overhead): two
            // workgroups per CU when the call is large enough, one block per workgroup for scheduler-sized calls
            const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
            int wgs = cus * 2;
            if (const char *e = getenv("MI355_OLS_UPS_WGS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;
            const int run = (int)((nblocks + wgs - 1) / wgs);
            const unsigned grid = (unsigned)((nblocks + run - 1) / run);
#define LAUNCH_UPS(PP) hipLaunchKernelGGL((k_ols_ups<G, PP>), dim3(grid), dim3(TH), 0, st, (const c32 *)in, (c32 *)out, (const c32 *)h->d_Hu, \
                                          (const c32 *)h->d_twf, h->ntaps, h->decim, n_y, (int)nblocks, run)
            switch (h->ups) {
            case 2: LAUNCH_UPS(2); break;
            case 3: LAUNCH_UPS(3); break;
            case 4: LAUNCH_UPS(4); break;
            default: LAUNCH_UPS(5); break;
            }
#undef LAUNCH_UPS
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
        if (h->nseg > 1 && one_pass) {
            const int s0 = (h->seg_len - 1 + 15) & ~15;
            const int L = (NF - s0) & ~15;
            const long long nblocks = (n_y + L - 1) / L, ngroups = (nblocks + F - 1) / F;
            if (nblocks > 0x7fffffffLL) { mi355_set_error("work() call too large"); return MI355_ERR_INVALID_ARG; }
            long long grid = mi355_balanced_grid(h->ctx, ngroups, 2, 2);
            hipLaunchKernelGGL((k_ols_part<G>), dim3((unsigned)grid), dim3(TH), 0, st, (const c32 *)in, (c32 *)out, (const c32 *)h->d_H,
                               (const c32 *)h->d_twf, h->ntaps, h->nseg, h->seg_len, h->seg_first, h->decim, L, s0, n_y, (int)ngroups,
                               getenv("MI355_OLS_PART_XCD_MAP") ? atoi(getenv("MI355_OLS_PART_XCD_MAP")) : 1);
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
    for (int sgm = 0; sgm < h->nseg; sgm++) {
        // segment sgm of a partitioned filter (nseg == 1: the whole filter): its taps, and where its input starts --
        // in_p[i] = in[i + (segments after this one) * seg_len], see upload_taps
        const int tn = h->nseg == 1 ? h->ntaps : (sgm == 0 ? h->seg_first : h->seg_len);
        const long long shift = h->nseg == 1 ? 0 : (sgm == 0 ? (long long)h->ntaps - h->seg_first : (long long)(h->nseg - 1 - sgm) * h->seg_len);
        const long long n_in = n_y + tn - 1;
        // any block length <= NF-ntaps+1 is a valid overlap-save schedule; a multiple of 16 keeps every block's loads
        // and stores on 128-byte boundaries (ntaps = 3: 39 % -> 70 % of HBM peak)
        const int s0 = align_stores ? ((tn - 1 + 15) & ~15) : tn - 1;  // first stored output of a block (see the kernel)
        int L = NF - s0;
        if (L > 16 && !getenv("MI355_OLS_RAGGED_L")) L &= ~15;
        const long long nblocks = (n_y + L - 1) / L;
        const long long ngroups = (nblocks + F - 1) / F;
        if (nblocks > 0x7fffffffLL) { mi355_set_error("work() call too large"); return MI355_ERR_INVALID_ARG; }
        // one-wave workgroups (NF <= 256): 32-48 per CU, four rounds of the 8-12 resident ones, measured +5 %; the 256-thread
        // workgroups of the larger transforms carry 16 spectrum values and two twiddle sets each and are best at 2-3 per CU
        long long grid = WAVES == 1 ? mi355_balanced_grid(h->ctx, ngroups, 32, 48) : mi355_balanced_grid(h->ctx, ngroups, 8 / WAVES, 12 / WAVES);
        hipLaunchKernelGGL((k_ols<NF, G>), dim3((unsigned)grid), dim3(TH), 0, st, (const c32 *)in + shift, (c32 *)out,
                           (const c32 *)h->d_H + (size_t)sgm * NF, (const c32 *)h->d_twf, (const c32 *)h->d_twi, tn, h->decim, L, s0, n_in, n_y,
                           (int)nblocks, (int)ngroups, sgm > 0 ? 1 : 0, xcd_map);
        MI355_HIP(hipGetLastError());
    }
    return MI355_OK;
}

template <int NF>
int launch_ols(mi355_filter *h, size_t nout, const void *in, void *out, hipStream_t st)
{
    if constexpr (NF <= 1024) {
        // one-wave workgroups (every exchange inside the wave, no workgroup barrier): +5 % at NF = 256, slower at 512/1024
        // (measured on MI355X); MI355_FILTER_WAVE_GEO=0/1 forces either geometry
        const int wave = getenv("MI355_FILTER_WAVE_GEO") ? atoi(getenv("MI355_FILTER_WAVE_GEO")) : -1;
        if (wave == 1 || (wave < 0 && NF <= 256)) return launch_ols_g<NF, GeoW<NF>>(h, nout, in, out, st);
    }
    return launch_ols_g<NF, Geo<NF>>(h, nout, in, out, st);
}

int launch_filter(mi355_filter *h, size_t nout, const void *in, void *out, hipStream_t st)
{
    if (h->ntaps < 1) {
        mi355_set_error("the filter has no taps (the last set_taps failed)");
        return MI355_ERR_STATE;
    }
    if (nout == 0) return MI355_OK;
    if (!h->use_time && h->nf) {
        switch (h->nf) {
        case 64: return launch_ols<64>(h, nout, in, out, st);
        case 128: return launch_ols<128>(h, nout, in, out, st);
        case 256: return launch_ols<256>(h, nout, in, out, st);
        case 512: return launch_ols<512>(h, nout, in, out, st);
        case 1024: return launch_ols<1024>(h, nout, in, out, st);
        case 2048: return launch_ols<2048>(h, nout, in, out, st);
        case 4096: return launch_ols<4096>(h, nout, in, out, st);
        }
        mi355_set_error("internal: no fast-convolution kernel for FFT size %d", h->nf);
        return MI355_ERR_STATE;
    }
    int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    static const bool mf_on = !getenv("MI355_FIR_MFMA") || atoi(getenv("MI355_FIR_MFMA")) != 0;
    // decimations 2-8 keep every decim-th output of the same product; that variant needs more registers (2 workgroups per CU)
    // and only pays from ~100 taps (129 taps, decimation 2: 137 -> 148 GS/s; 65 taps: equal)
    static const int dl_min = getenv("MI355_FIR_DEC_LDS_MIN") ? atoi(getenv("MI355_FIR_DEC_LDS_MIN")) : 9;  // smallest decimation of the LDS-staged kernel
    const int dmax = dl_min - 1 < 8 ? dl_min - 1 : 8;  // largest decimation of the kernels that compute every undecimated output
    // decimations above 8: either every undecimated output on the matrix cores (rate ~ 17500 / (K + 15) GS/s of input whatever D: 33 taps
    // 280, 65 taps 218, 200 taps 100, 400 taps 58) or the LDS-staged kernel below (time per output ~ 0.17 K + 3.7 D ps since its tap loop
    // requests eight samples at a time: 65 taps 192 / 226 / 245 GS/s at D = 10 / 16 / 32, 200 taps 131 / 178 / 216, 400 taps 86 / 134 /
    // 180; tools/fir_dec_probe.py) -- whichever this model puts ahead
    // (complex taps: four real products per tap on the matrix cores instead of two -- 65 / 200 / 400 taps 148 / 64 / 35 GS/s -- and a
    // little more per tap in the LDS-staged kernel)
    double r_all = (h->complex_taps ? 12000.0 : 17500.0) / (h->ntaps + 15);
    double r_lds = h->decim / ((h->complex_taps ? 0.00022 : 0.00017) * h->ntaps + 0.0037 * h->decim);
    if (r_all > (h->complex_taps ? 210.0 : 300.0)) r_all = h->complex_taps ? 210.0 : 300.0;
    if (r_lds > 250.0) r_lds = 250.0;
    // even decimations of a 16-byte aligned input: the 16-byte-wide form k_fir_dec2 (round 6), time per output ~ 0.10 K + 1.5 D ps (65 taps 304 / 603 /
    // 637 / 653 GS/s of input at D = 10 / 16 / 32 / 64, 200 taps 404 / 467 / 505 at D = 16 / 32 / 64, 400 taps 242 / 331 / 363)
    const bool dec2_ok = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && h->ntaps <= 1024 && !getenv("MI355_FIR_DEC2_OFF") &&
                         (h->decim % 2 == 0 || !getenv("MI355_FIR_DEC2_EVEN_ONLY"));
    if (dec2_ok) {
        r_lds = h->decim / ((h->complex_taps ? 0.00012 : 0.00010) * h->ntaps + 0.0015 * h->decim);
        if (r_lds > 700.0) r_lds = 700.0;
    }
    // (a decimation above eight filter lengths skips most of the input: neither of the two, the per-output kernel reads only what it needs --
    // 33 taps at D = 600: 277 GS/s of input with every undecimated output on the matrix cores, several thousand per output)
    // ... or the per-output kernel (k_fir_td_dec), which reads only the K samples an output needs: since its tap loop requests eight samples
    // at a time it runs at ~ D / (K c(D)) GS/s of input, c = 0.5 + 0.03 D ps up to 1.9 (neighbouring outputs share lines while D is small):
    // 33 taps 370 / 490 / 1360 at D = 10 / 32 / 100, 65 taps 260 / 334 / 628, 200 taps 128 / 147 / 299, 400 taps 70 / 78 / 172, and
    // thousands once D passes a few filter lengths (before: 16 - 56 at D = 10 - 50, which is why it was only used above eight lengths)
    const double c_po = 0.0005 + 0.00003 * h->decim < 0.0019 ? 0.0005 + 0.00003 * h->decim : 0.0019;
    double r_po = h->decim / (h->ntaps * c_po);
    // test switches, read per call: MI355_FIR_DEC_KERNEL = per_output | lds | all forces one of the three where it applies
    const bool lds_off = getenv("MI355_FIR_DEC_LDS_OFF") != nullptr;
    if (const char *force = getenv("MI355_FIR_DEC_KERNEL")) {
        if (!strcmp(force, "per_output")) r_po = 1e30;
        else if (!strcmp(force, "lds")) r_po = r_all = 0.0;
        else if (!strcmp(force, "all")) r_po = r_lds = 0.0;
    }
    const bool mf_ok = mf_on && h->mf_kk && h->ntaps >= 16, lds_ok = h->ntaps <= kDlSpan / 2 && !lds_off;
    // decimations 6 ... 8 belong to the kernels that compute every undecimated output (the matrix-core one from 96 taps, the register-tiled
    // one, ~ 14600 / K GS/s, below) unless one of the other two is predicted clearly ahead: at D = 8, 200 / 400 taps 107 / 60 -> 133 / 89 GS/s
    // of input LDS-staged, 33 taps 297 -> 334 per output
    const double r_inc = h->ntaps >= 96 && mf_ok ? r_all : (14600.0 / h->ntaps < 420.0 ? 14600.0 / h->ntaps : 420.0);
    // ... and decimations 3 ... 5 to k_fir_dec2 where it applies (round 6: 65 taps D = 3 / 4 / 5: 215 / 217 / 218 -> 231 / 299 / 343 GS/s of input; from
    // 96 taps on only from D = 4: 200 taps 107 -> 140 / 160, 400 taps 60 -> 72 / 91; D = 2 stays: 207 against 207, 100 against 91)
    const bool early2 = dec2_ok && lds_ok && h->ntaps >= 16 && h->decim <= dmax && h->decim >= (h->ntaps < 96 ? 3 : 4) && h->decim <= 5 && r_po < r_lds &&
                        !getenv("MI355_FIR_DEC2_FROM_6");
    const bool early = (h->decim >= 6 && h->decim <= dmax && h->ntaps >= 16 && (r_po > 1.05 * r_inc || (lds_ok && r_lds > 1.05 * r_inc))) || early2;
    const bool above = h->decim > dmax || early;
    const bool per_output = above && r_po >= (mf_ok ? r_all : 0.0) && r_po >= (lds_ok ? r_lds : 0.0) && !(h->ntaps < 16 && h->decim <= 64);
    const bool all_outputs_above_8 = h->decim > dmax && h->decim >= dl_min && r_all >= r_lds && !per_output;
    if (mf_on && h->mf_kk && ((h->decim == 1 && h->ntaps >= 16) || (h->decim >= 2 && h->decim <= dmax && h->ntaps >= 96 && !early) || (all_outputs_above_8 && h->ntaps >= 16))) {  // fewer taps: the vector kernel's short loop wins (9 taps: 350 vs 330 GS/s)
        const int span = kMfTile + 4 * h->mf_kk;
        const int nq = (span + kMfThreads - 1) / kMfThreads;
        const size_t smem = ((size_t)2 * (mf_pad(nq * kMfThreads + 16) + 1) + (size_t)(h->complex_taps ? 2 : 1) * (4 * h->mf_kk + 24)) * sizeof(float);
        const long long n_y = (long long)nout * h->decim;  // undecimated outputs
        const long long ntiles = (n_y + kMfTile - 1) / kMfTile;
        // grid-stride workgroups per CU (interleaved A/B, 65 taps over 2^26 samples: 8 -> 241 us, 16 -> 239, 32 -> 234)
        const int per_cu = getenv("MI355_TD_WG_PER_CU") && atoi(getenv("MI355_TD_WG_PER_CU")) > 0 ? atoi(getenv("MI355_TD_WG_PER_CU")) : 32;
        const long long grid = ntiles < (long long)cus * per_cu ? ntiles : (long long)cus * per_cu;
#define LAUNCH_MF(CT, DC)                                                                                                     \
    do {                                                                                                                      \
        if (smem > 64 * 1024)                                                                                                 \
            MI355_HIP(hipFuncSetAttribute((const void *)k_fir_mfma<CT, DC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_fir_mfma<CT, DC>), dim3((unsigned)grid), dim3(kMfThreads), smem, st, (const c32 *)in, (c32 *)out, h->d_hb, \
                           h->ntaps, h->mf_kk, n_y, h->decim);                                                                \
    } while (0)
        if (h->complex_taps) { if (h->decim == 1) LAUNCH_MF(true, false); else LAUNCH_MF(true, true); }
        else                 { if (h->decim == 1) LAUNCH_MF(false, false); else LAUNCH_MF(false, true); }
#undef LAUNCH_MF
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    }
    const int kpad0 = (h->ntaps + kTdU - 1) / kTdU * kTdU;
    // the register-tiled kernel computes every undecimated output: worth it up to a decimation of 8
    if (h->decim == 1 || (((h->decim <= dmax && !early) || (h->ntaps < 16 && h->decim <= 64 && h->decim >= dl_min)) && (size_t)td_rows(kpad0) * kTdU * sizeof(c32) <= 160 * 1024)) {  // (fewer than 16 taps: this kernel runs at 400 GS/s of input whatever the decimation)
        const int kpad = kpad0;
        const size_t smem = (size_t)td_rows(kpad) * kTdU * sizeof(c32);
        if (smem > 160 * 1024) { mi355_set_error("time-domain mode supports up to ~18000 taps"); return MI355_ERR_UNSUPPORTED; }
        const long long n_y = (long long)nout * h->decim;  // undecimated outputs
        long long ntiles = (n_y + kTdTile - 1) / kTdTile;
        // grid-stride workgroups, several rounds of them: measured, 12-16 per CU beat exactly the resident number by 8 %
        // (workgroups that finish early are replaced at once, which evens out the barrier phases)
        int per_cu = 16;
        if (const char *e = getenv("MI355_TD_WG_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
        long long grid = ntiles < (long long)cus * per_cu ? ntiles : (long long)cus * per_cu;
#define LAUNCH_TD(CT)                                                                                                        \
    do {                                                                                                                     \
        if (smem > 64 * 1024)                                                                                                \
            MI355_HIP(hipFuncSetAttribute((const void *)k_fir_td<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_fir_td<CT>), dim3((unsigned)grid), dim3(kTdThreads), smem, st, (const c32 *)in, (c32 *)out,     \
                           h->d_taps_rev, h->ntaps, n_y, kpad, h->decim);                                                   \
    } while (0)
        if (h->complex_taps) LAUNCH_TD(true);
        else LAUNCH_TD(false);
#undef LAUNCH_TD
    } else if (lds_ok && !per_output && dec2_ok) {
        // even decimations, 16-byte aligned input: the 16-byte-wide form of the LDS-staged kernel
        const int KP = (h->ntaps + 7) / 8 * 8;
        const int tap_units = (h->complex_taps ? 2 : 1) * KP / 4;
        static const int span_env = getenv("MI355_FIR_DEC2_SPAN") ? atoi(getenv("MI355_FIR_DEC2_SPAN")) : 0;
        // samples staged per tile: 3072 (26 KiB: five or six workgroups per CU) up to 128 taps, 4096 above -- measured at 65 taps, D = 16: spans of
        // 2048 / 3072 / 4096 / 6144 / 8192 samples 489 / 603 / 550 / 464 / 377 GS/s of input; 400 taps 194 / 223 / 242 / 168 / 194 -- and whole rounds of
        // 256 outputs where a tile holds more than one (390 outputs per tile at D = 10 ran the second round of threads half empty)
        const int span_max = span_env > 0 ? span_env : (h->ntaps <= 128 ? 3072 : 4096);
        int tile_out = span_max > KP ? (span_max - KP) / h->decim + 1 : 1;
        if (tile_out > 2048) tile_out = 2048;
        if (tile_out > kD2Threads) tile_out = tile_out / kD2Threads * kD2Threads;
        if (h->decim % 2 && tile_out > 1) tile_out &= ~1;  // (odd decimations: tiles start on a sample pair)
        const int pairs = ((tile_out - 1) * h->decim + KP + 1) / 2 + 1;
        const int sh = getenv("MI355_FIR_DEC2_PAD") ? atoi(getenv("MI355_FIR_DEC2_PAD")) : dec2_pad_shift(h->decim);
        const size_t smem = (size_t)(tap_units + d2_unit(pairs, sh) + 2) * 16;
        const long long ntiles = ((long long)nout + tile_out - 1) / tile_out;
        const long long grid = ntiles < (long long)cus * 8 ? ntiles : (long long)cus * 8;
#define LAUNCH_D2P(CT, OD)                                                                                                   \
    do {                                                                                                                     \
        MI355_HIP(hipFuncSetAttribute((const void *)k_fir_dec2<CT, OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_fir_dec2<CT, OD>), dim3((unsigned)grid), dim3(kD2Threads), smem, st, (const c32 *)in, (c32 *)out,   \
                           h->d_taps_rev, h->ntaps, KP, h->decim, (long long)nout, tile_out, tap_units, sh);                 \
    } while (0)
#define LAUNCH_D2(CT) do { if (h->decim % 2) LAUNCH_D2P(CT, true); else LAUNCH_D2P(CT, false); } while (0)
        if (h->complex_taps) LAUNCH_D2(true);
        else LAUNCH_D2(false);
#undef LAUNCH_D2P
#undef LAUNCH_D2
    } else if (lds_ok && !per_output) {
        // (a decimation far above the filter length skips most of the input: the per-output kernel reads only what it needs)
        int tile_out = (kDlSpan - h->ntaps) / h->decim + 1;
        if (tile_out > 2048) tile_out = 2048;
        const size_t smem = (size_t)(dl_slot(kDlSpan) + 1) * sizeof(c32);
        const long long ntiles = ((long long)nout + tile_out - 1) / tile_out;
        const long long grid = ntiles < (long long)cus * 8 ? ntiles : (long long)cus * 8;
#define LAUNCH_DL(CT)                                                                                                          \
    do {                                                                                                                       \
        MI355_HIP(hipFuncSetAttribute((const void *)k_fir_dec_lds<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_fir_dec_lds<CT>), dim3((unsigned)grid), dim3(kDlThreads), smem, st, (const c32 *)in, (c32 *)out, \
                           h->d_taps_rev, h->ntaps, h->decim, (long long)nout, tile_out);                                      \
    } while (0)
        if (h->complex_taps) LAUNCH_DL(true);
        else LAUNCH_DL(false);
#undef LAUNCH_DL
    } else {
        long long blocks = ((long long)nout + 255) / 256;
        long long grid = blocks < (long long)cus * 8 ? blocks : (long long)cus * 8;
        if (h->complex_taps)
            hipLaunchKernelGGL((k_fir_td_dec<true>), dim3((unsigned)grid), dim3(256), 0, st, (const c32 *)in, (c32 *)out,
                               h->d_taps_rev, h->ntaps, h->decim, (long long)nout);
        else
            hipLaunchKernelGGL((k_fir_td_dec<false>), dim3((unsigned)grid), dim3(256), 0, st, (const c32 *)in, (c32 *)out,
                               h->d_taps_rev, h->ntaps, h->decim, (long long)nout);
    }
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

}  // namespace

extern "C" int mi355_filter_create(mi355_ctx *ctx, int decimation, const void *taps, int ntaps, int complex_taps, int use_time,
                                   mi355_filter **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(decimation >= 1, "decimation must be >= 1");
    mi355_filter *h = new (std::nothrow) mi355_filter();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->decim = decimation; h->complex_taps = complex_taps ? 1 : 0; h->use_time = use_time ? 1 : 0;
    h->ntaps = 0; h->nf = 0;
    int rc = upload_taps(h, taps, ntaps);
    if (rc == MI355_OK) rc = h->pipe.init(ctx);
    if (rc != MI355_OK) { free_dev(h); h->pipe.release(); delete h; return rc; }
    mi355_log(ctx, MI355_LOG_INFO, "%s: %d %s taps, decimation %d: %s (transform size %d)", complex_taps ? "clComplexFilter" : "clFilter", ntaps,
              complex_taps ? "complex" : "real", decimation, h->nf ? "overlap-save in the frequency domain" : "direct form", h->nf);
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_filter_destroy(mi355_filter *h)
{
    if (!h) return MI355_OK;
    free_dev(h);
    h->pipe.release();
    delete h;
    return MI355_OK;
}

extern "C" int mi355_filter_set_taps(mi355_filter *h, const void *taps, int ntaps)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    std::lock_guard<std::mutex> g(h->lock);  // lib/clFilter_impl.cc:443 takes d_mutex here
    MI355_HIP(hipSetDevice(h->ctx->device));
    return upload_taps(h, taps, ntaps);
}

extern "C" int mi355_filter_ntaps(const mi355_filter *h) { return h ? h->ntaps : MI355_ERR_INVALID_ARG; }

extern "C" int mi355_filter_fftsize(const mi355_filter *h) { return h ? h->nf : MI355_ERR_INVALID_ARG; }

extern "C" int mi355_filter_get_taps(const mi355_filter *h, void *taps_out, int cap)
{
    MI355_REQUIRE(h && taps_out, "NULL argument");
    MI355_REQUIRE(cap >= h->ntaps, "taps_out too small");
    memcpy(taps_out, h->taps_host.data(), h->taps_host.size() * sizeof(float));
    return h->ntaps;
}

extern "C" int mi355_filter_work_dev(mi355_filter *h, size_t noutput_items, const void *in, void *out, void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (noutput_items == 0) return MI355_OK;
    MI355_REQUIRE(in && out, "NULL buffer");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
                  "device buffers must be 8-byte aligned");
    std::lock_guard<std::mutex> g(h->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    return launch_filter(h, noutput_items, in, out, mi355_pick_stream(h->ctx, stream));
}

extern "C" int mi355_filter_work(mi355_filter *h, size_t noutput_items, const void *in, void *out)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (noutput_items == 0) return MI355_OK;
    MI355_REQUIRE(in && out, "NULL buffer");
    std::lock_guard<std::mutex> g(h->lock);
    std::lock_guard<std::mutex> gc(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    // chunks of outputs; each chunk re-sends its ntaps-1 samples of history
    const size_t hist = (size_t)h->ntaps - 1;
    // sized from the larger side of a chunk, its INPUT (decim x the outputs): a slot's staging stays within the 1 ... 8 MiB pieces the
    // pipeline overlaps in, whatever the decimation
    size_t chunk_out = mi355_chunk_bytes(noutput_items * (size_t)h->decim * 8) / (8 * (size_t)h->decim);
    if (chunk_out < 1) chunk_out = 1;
    size_t first = noutput_items < chunk_out ? noutput_items : chunk_out;
    size_t inb = (first * h->decim + hist) * 8;
    int rc = h->pipe.ensure(1, &inb, first * 8);
    if (rc) return rc;
    HostPipe &p = h->pipe;
    const char *pin = (const char *)in;
    char *pout = (char *)out;
    if (noutput_items <= chunk_out && mi355_direct_ok(inb)) {  // small call: the kernel works on the pinned staging itself
        hipStream_t st = h->ctx->stream[0];
        mi355_copy(p.h_in[0][0], pin, inb);
        rc = launch_filter(h, noutput_items, p.h_in[0][0], p.h_out[0], st);
        if (rc) return rc;
        MI355_HIP(mi355_direct_sync(st));
        mi355_copy(pout, p.h_out[0], noutput_items * 8);
        return MI355_OK;
    }
    size_t nchunks = (noutput_items + chunk_out - 1) / chunk_out;
    size_t pend_off[HostPipe::kSlots] = {}, pend_bytes[HostPipe::kSlots] = {};
    for (size_t ci = 0; ci < nchunks; ci++) {
        int s = (int)(ci % HostPipe::kSlots);
        hipStream_t st = h->ctx->stream[s & 1];
        if (pend_bytes[s]) MI355_HIP(hipEventSynchronize(p.done[s]));
        size_t o0 = ci * chunk_out;
        size_t no = noutput_items - o0 < chunk_out ? noutput_items - o0 : chunk_out;
        size_t in_bytes = (no * h->decim + hist) * 8;
        // the slot's previous result out and its next input in, side by side
        mi355_copy2(pout + pend_off[s], p.h_out[s], pend_bytes[s], p.h_in[s][0], pin + o0 * h->decim * 8, in_bytes);
        pend_bytes[s] = 0;
        MI355_HIP(hipMemcpyAsync(p.d_in[s][0], p.h_in[s][0], in_bytes, hipMemcpyHostToDevice, st));
        rc = launch_filter(h, no, p.d_in[s][0], p.d_out[s], st);
        if (rc) return rc;
        MI355_HIP(hipMemcpyAsync(p.h_out[s], p.d_out[s], no * 8, hipMemcpyDeviceToHost, st));
        MI355_HIP(hipEventRecord(p.done[s], st));
        pend_off[s] = o0 * 8; pend_bytes[s] = no * 8;
    }
    for (int q = 0; q < HostPipe::kSlots; q++) {
        int s = (int)((nchunks + q) % HostPipe::kSlots);  // oldest slot first
        if (pend_bytes[s]) {
            MI355_HIP(hipEventSynchronize(p.done[s]));
            mi355_copy(pout + pend_off[s], p.h_out[s], pend_bytes[s]);
            pend_bytes[s] = 0;
        }
    }
    return MI355_OK;
}
