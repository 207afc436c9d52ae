// Register-level building blocks of the LDS-resident Stockham FFT (gfx950).
// Shared by the clFFT kernel, the fused overlap-save filter and the polyphase
// channelizer.  Everything is compile-time unrolled: a thread always owns 16
// complex points (32 VGPRs) and performs 16/R radix-R butterflies per pass.
#pragma once
#include <hip/hip_runtime.h>

namespace fftc {

struct c32 { float x, y; };

__device__ __forceinline__ c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c32 cconj(c32 a) { return mk(a.x, -a.y); }
__device__ __forceinline__ c32 scale(c32 a, float s) { return mk(a.x * s, a.y * s); }
// multiply by -i (forward) or +i (inverse): the W4^1 twiddle
template <int SIGN> __device__ __forceinline__ c32 rot90(c32 a) { return SIGN < 0 ? mk(a.y, -a.x) : mk(-a.y, a.x); }
// multiply by exp(SIGN * i * pi/4) * sqrt(2)/... i.e. W8^1 (normalised)
template <int SIGN> __device__ __forceinline__ c32 rot45(c32 a)
{
    constexpr float h = 0.70710678118654752440f;
    return SIGN < 0 ? mk((a.x + a.y) * h, (a.y - a.x) * h) : mk((a.x - a.y) * h, (a.y + a.x) * h);
}
// W8^3 = rot90(rot45)
template <int SIGN> __device__ __forceinline__ c32 rot135(c32 a)
{
    constexpr float h = 0.70710678118654752440f;
    return SIGN < 0 ? mk((a.y - a.x) * h, -(a.x + a.y) * h) : mk(-(a.x + a.y) * h, (a.x - a.y) * h);
}
// multiply by W16^m, m = 1 or 3 (cos/sin of pi/8), SIGN<0 forward
template <int SIGN, int M> __device__ __forceinline__ c32 rot16(c32 a)
{
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    constexpr float c = (M == 1) ? c1 : s1, s = (M == 1) ? s1 : c1;  // W16^1 = c1 - i s1 ; W16^3 = s1 - i c1
    return SIGN < 0 ? mk(a.x * c + a.y * s, a.y * c - a.x * s) : mk(a.x * c - a.y * s, a.y * c + a.x * s);
}

// ---- butterflies: in-place DFT of R points held in v[0..R-1] --------------------
// After the call slot s holds X[orev<R>(s)].
template <int R> __host__ __device__ constexpr int orev(int s)
{
    if (R == 16) return (s >> 2) + 4 * (s & 3);
    if (R == 8) return (s >> 2) + 2 * (s & 3);  // slots 0-3: X0,X2,X4,X6 ; 4-7: X1,X3,X5,X7
    return s;
}

template <int SIGN> __device__ __forceinline__ void bfly2(c32 &a, c32 &b)
{
    c32 t = a - b;
    a = a + b;
    b = t;
}

template <int SIGN> __device__ __forceinline__ void bfly4(c32 &a, c32 &b, c32 &c, c32 &d)
{
    c32 t0 = a + c, t1 = a - c, t2 = b + d, t3 = rot90<SIGN>(b - d);
    a = t0 + t2;  // X0
    c = t0 - t2;  // X2
    b = t1 + t3;  // X1
    d = t1 - t3;  // X3
}

template <int SIGN> __device__ __forceinline__ void bfly8(c32 *v)
{
    // decimation in frequency: radix-2 split then two radix-4
    c32 d0 = v[0] - v[4], d1 = rot45<SIGN>(v[1] - v[5]), d2 = rot90<SIGN>(v[2] - v[6]), d3 = rot135<SIGN>(v[3] - v[7]);
    c32 s0 = v[0] + v[4], s1 = v[1] + v[5], s2 = v[2] + v[6], s3 = v[3] + v[7];
    bfly4<SIGN>(s0, s1, s2, s3);  // X0 X2 X4 X6
    bfly4<SIGN>(d0, d1, d2, d3);  // X1 X3 X5 X7
    v[0] = s0; v[1] = s1; v[2] = s2; v[3] = s3;
    v[4] = d0; v[5] = d1; v[6] = d2; v[7] = d3;
}

template <int SIGN> __device__ __forceinline__ void bfly16(c32 *v)
{
    // 4x4: inner radix-4 over m for each a (x[a+4m]) -> y_a[b] at v[a+4b]
#pragma unroll
    for (int a = 0; a < 4; a++) bfly4<SIGN>(v[a], v[a + 4], v[a + 8], v[a + 12]);
    // twiddle y_a[b] *= W16^(a*b)
    v[5] = rot16<SIGN, 1>(v[5]);               // a=1,b=1
    v[6] = rot45<SIGN>(v[6]);                  // a=2,b=1 -> W16^2
    v[7] = rot16<SIGN, 3>(v[7]);               // a=3,b=1 -> W16^3
    v[9] = rot45<SIGN>(v[9]);                  // a=1,b=2 -> W16^2
    v[10] = rot90<SIGN>(v[10]);                // a=2,b=2 -> W16^4
    v[11] = rot135<SIGN>(v[11]);               // a=3,b=2 -> W16^6
    v[13] = rot16<SIGN, 3>(v[13]);             // a=1,b=3 -> W16^3
    v[14] = rot135<SIGN>(v[14]);               // a=2,b=3 -> W16^6
    {                                          // a=3,b=3 -> W16^9 = -W16^1
        c32 t = rot16<SIGN, 1>(v[15]);
        v[15] = mk(-t.x, -t.y);
    }
    // outer radix-4 over a for each b -> X[b+4c] at v[4b+c]
#pragma unroll
    for (int b = 0; b < 4; b++) bfly4<SIGN>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
}

template <int R, int SIGN> __device__ __forceinline__ void bfly(c32 *v)
{
    if constexpr (R == 2) bfly2<SIGN>(v[0], v[1]);
    else if constexpr (R == 4) bfly4<SIGN>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) bfly8<SIGN>(v);
    else bfly16<SIGN>(v);
}

// ---- radix plan: as many 16s as possible; the remainder radix goes last, or first
// when REV (used by inverse transforms that start from a forward transform's registers)
template <int N, bool REV = false> struct Plan {
    static constexpr int log2n()
    {
        int l = 0;
        for (int n = N; n > 1; n >>= 1) l++;
        return l;
    }
    static constexpr int L = log2n();
    static constexpr int NP = (L + 3) / 4;
    __host__ __device__ static constexpr int radix(int p)
    {
        int full = L / 4, rem = L % 4;
        if (rem == 0) return 16;
        if (REV) return p == 0 ? (1 << rem) : 16;
        return p < full ? 16 : (1 << rem);
    }
    // product of the radices of passes before p
    __host__ __device__ static constexpr int ns(int p)
    {
        int s = 1;
        for (int q = 0; q < p; q++) s *= radix(q);
        return s;
    }
};

// workgroup geometry: every thread owns 16 points
template <int N> struct Geo {
    static constexpr int TH = (N <= 4096) ? 256 : N / 16;  // threads per workgroup
    static constexpr int PTS = TH * 16;                     // points per workgroup iteration
    static constexpr int F = PTS / N;                       // frames per iteration
    static constexpr int WPE = (N <= 4096) ? 2 : 1;         // min waves per SIMD asked of the register allocator
};
// single-wave geometry for N <= 1024: a frame's N/16 threads sit inside one wave, so a one-wave workgroup needs no
// workgroup barrier at all (the barriers below degenerate to waitcnts) and the waves of a CU run fully decoupled
template <int N> struct GeoW {
    static_assert(N <= 1024, "a frame must fit one wave");
    static constexpr int TH = 64;
    static constexpr int PTS = TH * 16;
    static constexpr int F = PTS / N;
    static constexpr int WPE = 2;
};

// LDS slot swizzle (8-byte slots): XOR the low four slot bits with the next four.
// Makes the three access patterns of the 16-point-per-thread passes conflict free
// for ds_write_b64 (16-lane groups) and ds_read_b64 (32-lane groups).
__host__ __device__ constexpr int swz(int s) { return s ^ ((s >> 4) & 15); }

// 256-point frames: a 32-lane read group spans two frames whose slots are 256 apart, i.e. the same banks; folding slot
// bit 8 (the frame parity) into bit 4 separates them.  Used only inside transform_regs (writes and reads agree).
template <int N> __host__ __device__ constexpr int swzn(int s)
{
    if (N == 256) return s ^ ((s >> 4) & 31);
    // 64-point frames: 32 lanes span 2 frames (4 butterflies per thread) or 8 frames (one 16-point butterfly per thread,
    // 4 lanes per frame); slot bits 6 and 8 folded into bit 4 keep both patterns conflict free
    if (N == 64) return s ^ ((s >> 4) & 15) ^ ((((s >> 6) ^ (s >> 8)) & 1) << 4);
    return swz(s);
}

// LDS slot of logical index (raw + c) where c is a compile-time multiple of STEP whose bits are disjoint from raw's (every
// caller: c selects a digit of the index that raw leaves zero).  All swizzles above are XOR-linear in the slot bits, so
// swz(raw + c) = swz(raw) ^ swz(c): one v_xor with a literal per access instead of re-deriving the swizzle (8-9 integer
// operations each, a third of the vector instructions of the transform kernels before this).
template <int STEP, int N = 0> __device__ __forceinline__ int lds_at(int raw, int raw_swz, int c)
{
    (void)raw;
    if constexpr (STEP % 256 == 0 && N != 256 && N != 64) return raw_swz + c;  // the swizzle only touches the low 8 slot bits: c folds into the DS offset field
    else return raw_swz ^ swzn<N>(c);
}

// The same slot as a BYTE offset from the LDS area's start, from the swizzled raw index already times 8: the XOR form then costs one
// v_xor with a literal per access (slot index XOR, then times 8, was two instructions), the additive form folds into the DS offset field
template <int STEP, int N = 0> __device__ __forceinline__ c32 &lds_slot(c32 *lds, int raw_swz8, int c)
{
    if constexpr (STEP % 256 == 0 && N != 256 && N != 64) return *(c32 *)((char *)lds + raw_swz8 + c * 8);
    else return *(c32 *)((char *)lds + (raw_swz8 ^ (swzn<N>(c) * 8)));
}

// inverse of orev: slot that holds output index r
template <int R> __host__ __device__ constexpr int irev(int r)
{
    if (R == 16) return (r >> 2) + 4 * (r & 3);
    if (R == 8) return (r & 1) * 4 + (r >> 1);
    return r;
}

// Inter-pass twiddles kept in registers.  For a radix-R butterfly only the powers
// r in {1,2,3} and {4,8,12} of the butterfly's base twiddle are stored (exactly
// rounded from the double-precision table); the rest are one product w[4a]*w[b].
// Slots per butterfly: R=16 -> 6, R=8 -> 4, R=4 -> 3, R=2 -> 1  (<= 12 per pass).
template <int R> __host__ __device__ constexpr int tw_slots() { return R == 16 ? 6 : R == 8 ? 4 : R == 4 ? 3 : 1; }
template <int R> __host__ __device__ constexpr int tw_power(int i) { return i < 3 ? i + 1 : (i - 2) * 4; }
constexpr int kTwPerPass = 12;
template <int N> struct TwRegs { c32 w[Plan<N>::NP > 1 ? Plan<N>::NP - 1 : 1][kTwPerPass]; };

// twtab[k] = exp(sign * 2*pi*i*k/N), k < N (generated in double on the host)
template <int N, bool REV, class G = Geo<N>, int P = 1>
__device__ __forceinline__ void load_twiddles(TwRegs<N> &tw, int tid, const c32 *__restrict__ twtab)
{
    using PL = Plan<N, REV>;
    if constexpr (P < PL::NP) {
        constexpr int TH = G::TH, R = PL::radix(P), NS = PL::ns(P), B = N / R, S = tw_slots<R>();
#pragma unroll
        for (int q = 0; q < 16 / R; q++) {
            const int j = (tid + TH * q) % B, k = j % NS;
#pragma unroll
            for (int i = 0; i < S; i++) tw.w[P - 1][q * S + i] = twtab[(tw_power<R>(i) * k * (N / (NS * R))) & (N - 1)];
        }
        load_twiddles<N, REV, G, P + 1>(tw, tid, twtab);
    }
}

// v[r] *= W^r for r = 1..R-1, W^r rebuilt from the stored powers
template <int R, bool CJ = false> __device__ __forceinline__ void apply_twiddles(c32 *v, const c32 *w_in)
{
    // Opaque copies: keeps the w[4a]*w[b] products inside the frame loop instead of
    // letting loop-invariant code motion turn them back into 15 live register pairs.
    c32 w[tw_slots<R>()];
#pragma unroll
    for (int i = 0; i < tw_slots<R>(); i++) {
        w[i] = w_in[i];
        if (CJ) w[i].y = -w[i].y;  // the stored table is the forward one; an all-radix-16 inverse plan needs its conjugate
        asm volatile("" : "+v"(w[i].x), "+v"(w[i].y));
    }
#pragma unroll
    for (int r = 1; r < R; r++) {
        const int lo = r & 3, hi = r >> 2;
        c32 t;
        if (hi == 0) t = w[lo - 1];
        else if (lo == 0) t = w[2 + hi];
        else t = cmul(w[2 + hi], w[lo - 1]);
        v[r] = cmul(v[r], t);
    }
}

// All passes of an N-point transform on the workgroup's PTS points.
// In : v[q*R0 + r]  = x[fr][j + r*B0]            (R0 = first radix, g = tid + TH*q, fr = g/B0, j = g%B0)
// Out: v[q*RL + s]  = X[fr][j + orev<RL>(s)*BL]  (RL = last radix,  fr = g/BL, j = g%BL)
// `lds` holds PTS slots and is used in place; the caller must __syncthreads() before
// reusing it for another transform.
template <int N, int SIGN, bool REV, class G = Geo<N>, int P = 0, bool CJ = false>
__device__ __forceinline__ void transform_regs(c32 (&v)[16], const TwRegs<N> &tw, c32 *lds, int tid)
{
    using PL = Plan<N, REV>;
    if constexpr (P < PL::NP) {
        constexpr int TH = G::TH, NP = PL::NP, R = PL::radix(P), NS = PL::ns(P), B = N / R;
        if constexpr (P > 0) {
            __syncthreads();  // previous pass' LDS writes are visible
#pragma unroll
            for (int q = 0; q < 16 / R; q++) {
                const int g = tid + TH * q, raw = (g / B) * N + (g % B), rs8 = swzn<N>(raw) * 8;
#pragma unroll
                for (int r = 0; r < R; r++) v[q * R + r] = lds_slot<B, N>(lds, rs8, r * B);
            }
#pragma unroll
            for (int q = 0; q < 16 / R; q++) apply_twiddles<R, CJ>(&v[q * R], &tw.w[P - 1][q * tw_slots<R>()]);
            if constexpr (P < NP - 1) __syncthreads();  // everyone has read before anyone overwrites in place
        }
#pragma unroll
        for (int q = 0; q < 16 / R; q++) bfly<R, SIGN>(&v[q * R]);
        if constexpr (P < NP - 1) {
            // registers -> LDS at the autosort position
#pragma unroll
            for (int q = 0; q < 16 / R; q++) {
                const int g = tid + TH * q, fr = g / B, j = g % B;
                const int raw = fr * N + (j / NS) * NS * R + (j % NS), rs8 = swzn<N>(raw) * 8;
#pragma unroll
                for (int s = 0; s < R; s++) lds_slot<NS, N>(lds, rs8, orev<R>(s) * NS) = v[q * R + s];
            }
        }
        transform_regs<N, SIGN, REV, G, P + 1, CJ>(v, tw, lds, tid);
    }
}

// Two independent N-point transforms in lockstep (each with its own PTS-slot LDS area): the same passes as transform_regs,
// but every barrier and every LDS round trip is shared by the two, and their butterflies interleave in the instruction
// stream -- for kernels that run one wave per SIMD and cannot hide those latencies with other waves.
template <int N, int SIGN, bool REV, class G = Geo<N>, int P = 0, bool CJ = false>
__device__ __forceinline__ void transform_regs2(c32 (&va)[16], c32 (&vb)[16], const TwRegs<N> &tw, c32 *lds_a, c32 *lds_b, int tid)
{
    using PL = Plan<N, REV>;
    if constexpr (P < PL::NP) {
        constexpr int TH = G::TH, NP = PL::NP, R = PL::radix(P), NS = PL::ns(P), B = N / R;
        if constexpr (P > 0) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16 / R; q++) {
                const int g = tid + TH * q, raw = (g / B) * N + (g % B), rs = swzn<N>(raw);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    va[q * R + r] = lds_a[lds_at<B, N>(raw, rs, r * B)];
                    vb[q * R + r] = lds_b[lds_at<B, N>(raw, rs, r * B)];
                }
            }
#pragma unroll
            for (int q = 0; q < 16 / R; q++) {
                apply_twiddles<R, CJ>(&va[q * R], &tw.w[P - 1][q * tw_slots<R>()]);
                apply_twiddles<R, CJ>(&vb[q * R], &tw.w[P - 1][q * tw_slots<R>()]);
            }
            if constexpr (P < NP - 1) __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 16 / R; q++) {
            bfly<R, SIGN>(&va[q * R]);
            bfly<R, SIGN>(&vb[q * R]);
        }
        if constexpr (P < NP - 1) {
#pragma unroll
            for (int q = 0; q < 16 / R; q++) {
                const int g = tid + TH * q, fr = g / B, j = g % B;
                const int raw = fr * N + (j / NS) * NS * R + (j % NS), rs = swzn<N>(raw);
#pragma unroll
                for (int s = 0; s < R; s++) {
                    lds_a[lds_at<NS, N>(raw, rs, orev<R>(s) * NS)] = va[q * R + s];
                    lds_b[lds_at<NS, N>(raw, rs, orev<R>(s) * NS)] = vb[q * R + s];
                }
            }
        }
        transform_regs2<N, SIGN, REV, G, P + 1, CJ>(va, vb, tw, lds_a, lds_b, tid);
    }
}

}  // namespace fftc
