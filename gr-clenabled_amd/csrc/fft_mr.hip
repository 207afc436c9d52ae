// clFFT for lengths 2^a 3^b 5^c 7^d 11^e 13^f that are not a power of two (the reference plans them natively through the clFFT library,
// lib/clFFT_impl.cc:91-128; block semantics -- window, shift, real input -- lib/clFFT_impl.cc:464-518, restated in oracle/o_fft.c).
//
// One kernel, the transform resident in a workgroup's LDS: a Stockham autosort pass per radix (odd radices first, then 16s, then the
// remaining power of two), every thread holding at most sixteen values = 16 / R butterflies of R points per pass.
//   pass with radix R, Ns = product of the radices before it, butterfly j < N / R, k = j mod Ns:
//     v[r] = x[j + r N/R] W^(r k), W = exp(sign 2 pi i / (Ns R));  X = DFT_R(v);  y[(j / Ns) Ns R + k + r Ns] = X[r]
// The first pass reads the frame straight from global memory (consecutive butterflies = consecutive addresses: coalesced), the last one
// writes the spectrum straight back (in its last pass Ns R = N, so the output index is j + r Ns: coalesced again); in between the data
// stays in LDS, in place (all reads of a pass, barrier, all writes).  Several short frames share a workgroup iteration (a butterfly
// number runs over frames x N / R).  HBM traffic: the frame once in, once out.
// j / (N/R) and j / Ns are one multiply-high each (ceil(2^32 / d), exact below 2^16).  One table read per butterfly (W^k, a
// contiguous run per pass); the R - 1 powers by repeated squaring / one product.  LDS index i sits at slot i + i / 32 (the passes of
// the power-of-two radices would otherwise put a wave's stores in a handful of banks).
#include "fft_mr.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "fft_core.hpp"

using namespace fftc;

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int kVals = 16;  // complex values per thread and pass
constexpr int kMaxPass = 12;

struct MrArgs {
    const void *in;
    c32 *out;
    const float *window;
    const c32 *tw;
    int n, nframes, frames, npass, in_rot, out_rot, real_in;
    int copy_in = 0;           // k_fft_mr: the workgroup's frames are loaded as one contiguous run into LDS (first-pass runs n / radix < 128 B)
    int copy_out = 0;          // k_fft_mr: the last pass leaves its frames in LDS and the workgroup stores them as one contiguous run (last-pass runs < 256 B)
    unsigned m_n = 0;          // ceil(2^32 / n)
    int tw_lds = 0, ntw = 0;   // k_pfb_mr: first LDS slot of the twiddle runs, their length
    int dbg = 0;
    int fs_shift = 0;          // k_pfb_mr: a workgroup's frames are `frames >> fs_shift` time ranges of 2^fs_shift steps each; 0 = frames in a row
    long long range_len = 0;   // k_pfb_mr: steps between the starts of two ranges
    long long ngroups;
    MrPass pass[kMaxPass];
};

__device__ __forceinline__ int slot(int i) { return i + (i >> 5); }

// composite radices R = A x B (both with natural-order butterflies): DFT_A down the columns (stride B), W_R^(b k1), DFT_B along the rows
template <int R> __host__ __device__ constexpr int fac_a() { return R == 6 ? 2 : R == 9 ? 3 : R == 10 ? 2 : R == 12 ? 4 : R == 14 ? 2 : R == 15 ? 3 : 1; }
template <int R> __host__ __device__ constexpr int out_index(int s)
{
    if (R == 8 || R == 16) return orev<R>(s);
    if (fac_a<R>() > 1) return s / (R / fac_a<R>()) + fac_a<R>() * (s % (R / fac_a<R>()));  // slot B k1 + k2 holds X[k1 + A k2]
    return s;
}

// cos / sin of 2 pi m / R for the composite radices (the twiddles between the two factors of a butterfly)
template <int R> struct Roots;
template <> struct Roots<6> {
    static constexpr float c[6] = {1.000000000e+00f, 5.000000000e-01f, -5.000000000e-01f, -1.000000000e+00f, -5.000000000e-01f, 5.000000000e-01f};
    static constexpr float s[6] = {0.000000000e+00f, 8.660254038e-01f, 8.660254038e-01f, 1.224646799e-16f, -8.660254038e-01f, -8.660254038e-01f};
};
template <> struct Roots<9> {
    static constexpr float c[9] = {1.000000000e+00f, 7.660444431e-01f, 1.736481777e-01f, -5.000000000e-01f, -9.396926208e-01f, -9.396926208e-01f, -5.000000000e-01f, 1.736481777e-01f, 7.660444431e-01f};
    static constexpr float s[9] = {0.000000000e+00f, 6.427876097e-01f, 9.848077530e-01f, 8.660254038e-01f, 3.420201433e-01f, -3.420201433e-01f, -8.660254038e-01f, -9.848077530e-01f, -6.427876097e-01f};
};
template <> struct Roots<10> {
    static constexpr float c[10] = {1.000000000e+00f, 8.090169944e-01f, 3.090169944e-01f, -3.090169944e-01f, -8.090169944e-01f, -1.000000000e+00f, -8.090169944e-01f, -3.090169944e-01f, 3.090169944e-01f, 8.090169944e-01f};
    static constexpr float s[10] = {0.000000000e+00f, 5.877852523e-01f, 9.510565163e-01f, 9.510565163e-01f, 5.877852523e-01f, 1.224646799e-16f, -5.877852523e-01f, -9.510565163e-01f, -9.510565163e-01f, -5.877852523e-01f};
};
template <> struct Roots<12> {
    static constexpr float c[12] = {1.000000000e+00f, 8.660254038e-01f, 5.000000000e-01f, 6.123233996e-17f, -5.000000000e-01f, -8.660254038e-01f, -1.000000000e+00f, -8.660254038e-01f, -5.000000000e-01f, -1.836970199e-16f, 5.000000000e-01f, 8.660254038e-01f};
    static constexpr float s[12] = {0.000000000e+00f, 5.000000000e-01f, 8.660254038e-01f, 1.000000000e+00f, 8.660254038e-01f, 5.000000000e-01f, 1.224646799e-16f, -5.000000000e-01f, -8.660254038e-01f, -1.000000000e+00f, -8.660254038e-01f, -5.000000000e-01f};
};
template <> struct Roots<14> {
    static constexpr float c[14] = {1.000000000e+00f, 9.009688679e-01f, 6.234898019e-01f, 2.225209340e-01f, -2.225209340e-01f, -6.234898019e-01f, -9.009688679e-01f, -1.000000000e+00f, -9.009688679e-01f, -6.234898019e-01f, -2.225209340e-01f, 2.225209340e-01f, 6.234898019e-01f, 9.009688679e-01f};
    static constexpr float s[14] = {0.000000000e+00f, 4.338837391e-01f, 7.818314825e-01f, 9.749279122e-01f, 9.749279122e-01f, 7.818314825e-01f, 4.338837391e-01f, 1.224646799e-16f, -4.338837391e-01f, -7.818314825e-01f, -9.749279122e-01f, -9.749279122e-01f, -7.818314825e-01f, -4.338837391e-01f};
};
template <> struct Roots<15> {
    static constexpr float c[15] = {1.000000000e+00f, 9.135454576e-01f, 6.691306064e-01f, 3.090169944e-01f, -1.045284633e-01f, -5.000000000e-01f, -8.090169944e-01f, -9.781476007e-01f, -9.781476007e-01f, -8.090169944e-01f, -5.000000000e-01f, -1.045284633e-01f, 3.090169944e-01f, 6.691306064e-01f, 9.135454576e-01f};
    static constexpr float s[15] = {0.000000000e+00f, 4.067366431e-01f, 7.431448255e-01f, 9.510565163e-01f, 9.945218954e-01f, 8.660254038e-01f, 5.877852523e-01f, 2.079116908e-01f, -2.079116908e-01f, -5.877852523e-01f, -8.660254038e-01f, -9.945218954e-01f, -9.510565163e-01f, -7.431448255e-01f, -4.067366431e-01f};
};

template <> struct Roots<11> {
    static constexpr float c[11] = {1.000000000e+00f, 8.412535328e-01f, 4.154150130e-01f, -1.423148383e-01f, -6.548607339e-01f, -9.594929736e-01f, -9.594929736e-01f, -6.548607339e-01f, -1.423148383e-01f, 4.154150130e-01f, 8.412535328e-01f};
    static constexpr float s[11] = {0.000000000e+00f, 5.406408175e-01f, 9.096319954e-01f, 9.898214419e-01f, 7.557495744e-01f, 2.817325568e-01f, -2.817325568e-01f, -7.557495744e-01f, -9.898214419e-01f, -9.096319954e-01f, -5.406408175e-01f};
};
template <> struct Roots<13> {
    static constexpr float c[13] = {1.000000000e+00f, 8.854560257e-01f, 5.680647467e-01f, 1.205366803e-01f, -3.546048870e-01f, -7.485107482e-01f, -9.709418174e-01f, -9.709418174e-01f, -7.485107482e-01f, -3.546048870e-01f, 1.205366803e-01f, 5.680647467e-01f, 8.854560257e-01f};
    static constexpr float s[13] = {0.000000000e+00f, 4.647231720e-01f, 8.229838659e-01f, 9.927088741e-01f, 9.350162427e-01f, 6.631226582e-01f, 2.393156643e-01f, -2.393156643e-01f, -6.631226582e-01f, -9.350162427e-01f, -9.927088741e-01f, -8.229838659e-01f, -4.647231720e-01f};
};

// DFT_R in place; slot s holds X[out_index<R>(s)] afterwards
template <int R, int SIGN> __device__ __forceinline__ void dft(c32 *v)
{
    if constexpr (R == 2) bfly2<SIGN>(v[0], v[1]);
    else if constexpr (R == 4) bfly4<SIGN>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) bfly8<SIGN>(v);
    else if constexpr (R == 16) bfly16<SIGN>(v);
    else if constexpr (R == 3) {
        constexpr float s3 = 0.86602540378443864676f;
        const c32 t = v[1] + v[2], d = rot90<SIGN>(scale(v[1] - v[2], s3));  // i sign sin(2 pi / 3) (x1 - x2)
        const c32 m = mk(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
        v[0] = v[0] + t;
        v[1] = m + d;
        v[2] = m - d;
    } else if constexpr (R == 5) {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const c32 t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        const c32 a1 = mk(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
        const c32 a2 = mk(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
        const c32 b1 = rot90<SIGN>(mk(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
        const c32 b2 = rot90<SIGN>(mk(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
        v[0] = v[0] + t1 + t2;
        v[1] = a1 + b1;
        v[4] = a1 - b1;
        v[2] = a2 + b2;
        v[3] = a2 - b2;
    } else if constexpr (fac_a<R>() > 1) {
        constexpr int A = fac_a<R>(), B = R / A;
#pragma unroll
        for (int b = 0; b < B; b++) {
            c32 t[A];
#pragma unroll
            for (int q = 0; q < A; q++) t[q] = v[B * q + b];
            dft<A, SIGN>(t);
#pragma unroll
            for (int k1 = 0; k1 < A; k1++) {
                c32 y = t[k1];
                if (b * k1 != 0) {
                    constexpr float sg = SIGN < 0 ? -1.f : 1.f;
                    y = cmul(y, mk(Roots<R>::c[(b * k1) % R], sg * Roots<R>::s[(b * k1) % R]));
                }
                v[B * k1 + b] = y;
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < A; k1++) dft<B, SIGN>(&v[B * k1]);
    } else if constexpr (R == 11 || R == 13) {
        // a prime radix as (R - 1) / 2 symmetric pairs: X[k], X[R - k] = a_k +- i sign b_k, a_k = x0 + sum_j cos(2 pi j k / R) (x_j + x_(R-j)),
        // b_k = sum_j sin(2 pi j k / R) (x_j - x_(R-j))   (clFFT's remaining radices, lib/clFFT_impl.cc:91-128: lengths with factors 11 and 13)
        constexpr int H = (R - 1) / 2;
        c32 t[H], u[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            t[j] = v[j + 1] + v[R - 1 - j];
            u[j] = v[j + 1] - v[R - 1 - j];
        }
        c32 x0 = v[0], sum = v[0];
#pragma unroll
        for (int j = 0; j < H; j++) sum = sum + t[j];
        v[0] = sum;
#pragma unroll
        for (int k = 1; k <= H; k++) {
            c32 ak = x0, bk = mk(0.f, 0.f);
#pragma unroll
            for (int j = 1; j <= H; j++) {
                const float c = Roots<R>::c[(j * k) % R], sn = Roots<R>::s[(j * k) % R];
                ak = mk(ak.x + c * t[j - 1].x, ak.y + c * t[j - 1].y);
                bk = mk(bk.x + sn * u[j - 1].x, bk.y + sn * u[j - 1].y);
            }
            const c32 jb = rot90<SIGN>(bk);
            v[k] = ak + jb;
            v[R - k] = ak - jb;
        }
    } else {
        static_assert(R == 7, "radix");
        constexpr float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
        constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
        const c32 t1 = v[1] + v[6], t2 = v[2] + v[5], t3 = v[3] + v[4], u1 = v[1] - v[6], u2 = v[2] - v[5], u3 = v[3] - v[4];
        // cos(2 pi j k / 7): k = 1: c1 c2 c3;  k = 2: c2 c3 c1;  k = 3: c3 c1 c2.   sin: k = 1: s1 s2 s3;  k = 2: s2 -s3 -s1;  k = 3: s3 -s1 s2
        const c32 a1 = mk(v[0].x + c1 * t1.x + c2 * t2.x + c3 * t3.x, v[0].y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
        const c32 a2 = mk(v[0].x + c2 * t1.x + c3 * t2.x + c1 * t3.x, v[0].y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
        const c32 a3 = mk(v[0].x + c3 * t1.x + c1 * t2.x + c2 * t3.x, v[0].y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
        const c32 b1 = rot90<SIGN>(mk(s1 * u1.x + s2 * u2.x + s3 * u3.x, s1 * u1.y + s2 * u2.y + s3 * u3.y));
        const c32 b2 = rot90<SIGN>(mk(s2 * u1.x - s3 * u2.x - s1 * u3.x, s2 * u1.y - s3 * u2.y - s1 * u3.y));
        const c32 b3 = rot90<SIGN>(mk(s3 * u1.x - s1 * u2.x + s2 * u3.x, s3 * u1.y - s1 * u2.y + s2 * u3.y));
        v[0] = v[0] + t1 + t2 + t3;
        v[1] = a1 + b1;
        v[6] = a1 - b1;
        v[2] = a2 + b2;
        v[5] = a2 - b2;
        v[3] = a3 + b3;
        v[4] = a3 - b3;
    }
}

// MODE 0: first pass (global -> LDS), 1: LDS -> LDS, 2: last pass (LDS -> global), 3: first pass on frames that are in LDS already (k_pfb_mr)
template <int R, int SIGN, int MODE, bool TWL = false>
__device__ __forceinline__ void mr_pass(const MrArgs &a, const MrPass &ps, c32 *lds, int tid, long long group)
{
    const int TH = blockDim.x;
    constexpr int B = kVals / R;
    c32 v[B][R], w1[B];
    const int n = a.n, nb = ps.nb, nbt = a.frames * nb;
    // (butterfly number -> frame, place in the frame, k: recomputed where they are needed -- four multiply-highs per butterfly cost less
    // than the registers that would carry them across the barrier)
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int b = tid + TH * i;
        const int fr = (int)__umulhi((unsigned)b, ps.m_nb), bb = b - fr * nb;
        if constexpr (MODE == 1 || MODE == 2) {
            const int k = bb - (int)__umulhi((unsigned)bb, ps.m_ns) * ps.ns;
            // the butterfly's twiddle comes from L1 / L2: asked for before the LDS reads and the barrier, not after them
            // (TWL, k_pfb_mr: the twiddle runs sit in LDS behind the frames -- a global load here would make the pass wait for every load before it,
            // the next rows' prefetch included: the memory counter is in order)
            if constexpr (TWL) w1[i] = b < nbt ? lds[a.tw_lds + ps.tw_off + k] : mk(1.f, 0.f);
            else
            w1[i] = b < nbt ? a.tw[ps.tw_off + k] : mk(1.f, 0.f);
        }
        if (b < nbt) {
            if constexpr (MODE == 0) {
                const long long frame = group * a.frames + fr;
                const bool live = frame < a.nframes;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int src = bb + r * nb + a.in_rot;  // reverse + shift: the halves of the frame are swapped on load
                    if (src >= n) src -= n;
                    c32 x = mk(0.f, 0.f);
                    if (live) {
                        if (a.real_in) x = mk(((const float *)a.in)[(size_t)frame * n + src], 0.f);
                        else {
                            const f2v t = __builtin_nontemporal_load((const f2v *)a.in + (size_t)frame * n + src);
                            x = mk(t.x, t.y);
                        }
                        const float w = a.window[src];  // indexed by the ORIGINAL position (lib/clFFT_impl.cc:477-493)
                        x = mk(x.x * w, x.y * w);
                    }
                    v[i][r] = x;
                }
            } else {
                const int base = fr * n + bb;
#pragma unroll
                for (int r = 0; r < R; r++) v[i][r] = lds[slot(base + r * nb)];
            }
        }
    }
    if constexpr (MODE != 0) __syncthreads();  // every value of this pass is in registers: the writes below go to the same memory
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int b = tid + TH * i;
        if (b < nbt) {
            const int fr = (int)__umulhi((unsigned)b, ps.m_nb), bb = b - fr * nb;
            int g = bb, k = 0;
            if constexpr (MODE == 1 || MODE == 2) {
                g = (int)__umulhi((unsigned)bb, ps.m_ns);
                k = bb - g * ps.ns;
                // W^r from W, W^2, W^4, W^8 (at most three products per value): four stored powers instead of R - 1 -- the register
                // count of the radix-14 ... 16 passes decides how many waves a SIMD holds, and this kernel lives on occupancy
                c32 sq[4];
                sq[0] = w1[i];
#pragma unroll
                for (int q = 1; q < 4; q++) sq[q] = (R > (1 << q)) ? cmul(sq[q - 1], sq[q - 1]) : sq[q - 1];
#pragma unroll
                for (int r = 1; r < R; r++) {
                    c32 t = mk(1.f, 0.f);
                    bool first = true;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if ((r >> q) & 1) {
                            t = first ? sq[q] : cmul(t, sq[q]);
                            first = false;
                        }
                    v[i][r] = cmul(v[i][r], t);
                }
            }
            dft<R, SIGN>(v[i]);
            if constexpr (MODE == 2) {
                // (k_pfb_mr: `group` is the step the workgroup's first range is at; frame fr = step fr mod 2^fs_shift of range fr >> fs_shift)
                const long long frame = a.fs_shift ? group + (fr >> a.fs_shift) * a.range_len + (fr & ((1 << a.fs_shift) - 1)) : group * a.frames + fr;
                if (frame < a.nframes) {
#pragma unroll
                    for (int s = 0; s < R; s++) {
                        int p = k + out_index<R>(s) * ps.ns - a.out_rot;  // (g = 0 in the last pass); forward + shift: out[p] = X[(p + ceil(n/2)) mod n]
                        if (p < 0) p += n;
                        f2v z;
                        z.x = v[i][s].x;
                        z.y = v[i][s].y;
                        __builtin_nontemporal_store(z, (f2v *)a.out + (size_t)frame * n + p);
                    }
                }
            } else {
                const int base = fr * n + g * ps.ns * R + k;
#pragma unroll
                for (int s = 0; s < R; s++) lds[slot(base + out_index<R>(s) * ps.ns)] = v[i][s];
            }
        }
    }
    if constexpr (MODE != 2) __syncthreads();
}

template <int SIGN>
__global__ __launch_bounds__(1024) void k_fft_mr(const MrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) c32 mr_lds[];
    const int tid0 = threadIdx.x;
#define MR_PASS(MODE, P)                                                                        \
    switch (a.pass[P].radix) {                                                                  \
    case 2: mr_pass<2, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    case 3: mr_pass<3, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    case 4: mr_pass<4, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    case 5: mr_pass<5, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    case 7: mr_pass<7, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    case 6: mr_pass<6, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 9: mr_pass<9, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 10: mr_pass<10, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 11: mr_pass<11, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 13: mr_pass<13, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 12: mr_pass<12, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 14: mr_pass<14, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 15: mr_pass<15, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 8: mr_pass<8, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;                \
    default: mr_pass<16, SIGN, MODE>(a, a.pass[P], mr_lds, tid, group); break;              \
    }
    for (long long group = blockIdx.x; group < a.ngroups; group += gridDim.x) {
        // opaque per iteration: everything below depends on the thread number and the pass only, and hoisted out of this loop -- for every
        // radix of the three switches at once -- it costs hundreds of registers (256 + 190 spilled before this line was here)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (a.copy_in) {
            // The first pass reads runs of n / radix values (20 = 5 x 4 points: four values, 32 bytes): under 128 bytes the workgroup's frames come in as
            // one contiguous run instead, lanes along it, windowed by the position in memory and rotated by the reverse shift on the way into LDS.
            const int tot = a.frames * a.n;
            for (int idx = tid0; idx < tot; idx += blockDim.x) {
                const int fr = (int)__umulhi((unsigned)idx, a.m_n), q = idx - fr * a.n;
                int i = q - a.in_rot;  // the first pass' value i is the sample at (i + in_rot) mod n
                if (i < 0) i += a.n;
                c32 x = mk(0.f, 0.f);
                if (group * a.frames + fr < a.nframes) {
                    const size_t at = (size_t)group * a.frames * a.n + idx;
                    if (a.real_in) x = mk(((const float *)a.in)[at], 0.f);
                    else {
                        const f2v t = __builtin_nontemporal_load((const f2v *)a.in + at);
                        x = mk(t.x, t.y);
                    }
                    const float w = a.window[q];
                    x = mk(x.x * w, x.y * w);
                }
                mr_lds[slot(fr * a.n + i)] = x;
            }
            __syncthreads();
            MR_PASS(3, 0)
        } else {
            MR_PASS(0, 0)
        }
        for (int p = 1; p < a.npass - 1; p++) { MR_PASS(1, p) }
        if (a.copy_out) {
            // The last pass stores runs of `ns` values (n = 48 = 3 x 16: three values, 24 bytes, some twenty lines per store instruction): below 256
            // bytes it leaves the frames in LDS instead (value p of frame fr at fr n + p) and the workgroup's frames -- neighbours in memory -- go out
            // as one contiguous run, lanes along it; the forward shift is a rotation of the read index.
            MR_PASS(1, a.npass - 1)
            const int tot = a.frames * a.n;
            for (int idx = tid0; idx < tot; idx += blockDim.x) {
                const int fr = (int)__umulhi((unsigned)idx, a.m_n);
                int p = idx - fr * a.n + a.out_rot;  // out[q] = X[(q + ceil(n/2)) mod n]
                if (p >= a.n) p -= a.n;
                if (group * a.frames + fr < a.nframes) {
                    const c32 v = mr_lds[slot(fr * a.n + p)];
                    f2v z;
                    z.x = v.x;
                    z.y = v.y;
                    __builtin_nontemporal_store(z, (f2v *)a.out + (size_t)group * a.frames * a.n + idx);
                }
            }
            __syncthreads();  // the frames have been read: the next group's first pass writes them
        } else {
            MR_PASS(2, a.npass - 1)
        }
    }
#undef MR_PASS
}


// ------------------------------------------------------------------------------------------------------------------------------
// clPolyphaseChannelizer with a channel count that is not one of pfb.hip's power-of-two kernels (10, 20, 100, 200 ... channels), critically
// sampled: branch filters AND the M-point transform in one pass over the input (lib/clPolyphaseChannelizer_impl.cc:150-175 filters into a
// scratch buffer and transforms that; pfb.hip's two-kernel form of it moves every sample four times).  A workgroup walks Q time ranges;
// thread (q, j) owns arm j of range q: the last PMAX - 1 samples of its arm stay in registers, every step loads ONE new sample (lanes along the arms:
// a row of M consecutive samples per range), y_j[i] = sum_p h[j + M p] x_j[i - p] with fmaf, taps ascending -- the operation order of
// k_pfb_branches_t, so the two forms agree bit for bit -- and goes to LDS as value j of frame (q, step); after FS steps the Q x FS frames are
// transformed in place by the passes of k_fft_mr and stored.  The next FS rows are requested before the transforms and arrive behind them.
// A range is nsteps / (workgroups x Q) steps long, so the PMAX - 1 rows of warm-up are a few per cent of what a range reads.
struct PfbMr {
    const c32 *in;
    const float *taps;
    int K, M, Q, nsteps, direct;
    unsigned m_M, m_run;  // ceil(2^32 / M), ceil(2^32 / (FS M))
};

template <int SIGN, int PMAX, int FS>
__global__ __launch_bounds__(512) void k_pfb_mr(const MrArgs a, const PfbMr f)
{
    extern __shared__ __attribute__((aligned(16))) c32 mr_lds[];
    const int tid0 = threadIdx.x;
    const int q = (int)__umulhi((unsigned)tid0, f.m_M), j = tid0 - q * f.M;
    const bool arm = q < f.Q;
    // the taps: PMAX x M floats in LDS behind the twiddle runs (tap p of arm j at p M + j: lanes along the arms, no bank conflict); a thread reads
    // each of its taps once per iteration -- held in registers they would cost a workgroup its third wave per SIMD
    float *tl = (float *)(mr_lds + a.tw_lds + a.ntw);
    for (int i = tid0; i < PMAX * f.M; i += blockDim.x) tl[i] = i < f.K ? f.taps[i] : 0.f;
    const float *th = tl + (arm ? j : 0);
    const long long r0 = (long long)blockIdx.x * f.Q * a.range_len, s0 = r0 + (long long)q * a.range_len;
    const f2v *xp = (const f2v *)f.in + (f.K - 1 - j);  // x_j[r] = in[r M - j + K - 1] (k_pfb_branches_t)
    auto ld = [&](long long r) {
        const long long o = r * f.M;
        f2v x = {0.f, 0.f};
        if (arm && o + (f.K - 1 - j) >= 0 && r < f.nsteps && !(a.dbg & 4)) x = __builtin_nontemporal_load(xp + o);
        return x;
    };
    for (int i = tid0; i < a.ntw; i += blockDim.x) mr_lds[a.tw_lds + i] = a.tw[i];  // (the first barrier of the loop is in front of their first use)
    f2v x[PMAX - 1 + FS], nx[FS];
#pragma unroll
    for (int u = 0; u < PMAX - 1 + FS; u++) x[u] = ld(s0 - (PMAX - 1) + u);  // the warm-up rows and the first FS rows
    const int iters = (int)(a.range_len / FS);
#pragma unroll
    for (int s = 0; s < FS; s++) nx[s] = f2v{0.f, 0.f};
    const int run = FS * a.n;  // values of a range per iteration: one contiguous run of the output
#define MR_PASS(MODE, P)                                                                        \
    switch (a.pass[P].radix) {                                                                  \
    case 2: mr_pass<2, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 3: mr_pass<3, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 4: mr_pass<4, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 5: mr_pass<5, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 7: mr_pass<7, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 6: mr_pass<6, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 9: mr_pass<9, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    case 10: mr_pass<10, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 11: mr_pass<11, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 13: mr_pass<13, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 12: mr_pass<12, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 14: mr_pass<14, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 15: mr_pass<15, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    case 8: mr_pass<8, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                    \
    default: mr_pass<16, SIGN, MODE, true>(a, a.pass[P], mr_lds, tid, group); break;                  \
    }
    for (int it = 0; it < iters; it++) {
        // this iteration's rows were requested at the top of the last one; the next one's go out here, in front of the arithmetic (requested at the
        // END of an iteration, behind its stores, and taken over behind the transform -- 226 -> 260 us at 100 channels, 219 -> 248 at 20)
        if (it > 0) {
#pragma unroll
            for (int s = 0; s < FS; s++) x[PMAX - 1 + s] = nx[s];
        }
        if (it + 1 < iters) {
            const long long nxt = s0 + (long long)(it + 1) * FS;
#pragma unroll
            for (int s = 0; s < FS; s++) nx[s] = ld(nxt + s);
        }
        if (arm) {
            f2v acc[FS];
#pragma unroll
            for (int s = 0; s < FS; s++) acc[s] = f2v{0.f, 0.f};
            if (a.dbg & 1) {
#pragma unroll
                for (int s = 0; s < FS; s++) acc[s] = x[PMAX - 1 + s];
            } else
#pragma unroll
            for (int p = 0; p < PMAX; p++) {  // per output: taps ascending (lib/clPolyphaseChannelizer_impl.cc:156-167); both components in one packed fma
                const float hp = th[p * f.M];
                const f2v hh = {hp, hp};
#pragma unroll
                for (int s = 0; s < FS; s++) acc[s] = __builtin_elementwise_fma(x[PMAX - 1 + s - p], hh, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < FS; s++) mr_lds[slot((q * FS + s) * a.n + j)] = mk(acc[s].x, acc[s].y);
        }
#pragma unroll
        for (int u = 0; u < PMAX - 1; u++) x[u] = x[u + FS];
        __syncthreads();
        int tid = tid0;
        asm volatile("" : "+v"(tid));  // (as in k_fft_mr: keeps the passes' index arithmetic out of the loop-invariant registers)
        const long long group = r0 + (long long)it * FS;
        if (!(a.dbg & 2)) {
            MR_PASS(3, 0)
            for (int p = 1; p < a.npass - 1; p++) { MR_PASS(1, p) }
            if (f.direct) {
                if (!(a.dbg & 8)) { MR_PASS(2, a.npass - 1) }  // runs of at least 64 bytes: the last pass stores them itself
            } else {
                MR_PASS(1, a.npass - 1)  // the last pass leaves its frames in LDS (g = 0: value p of frame fr at fr n + p)
            }
        }
        // store: a range's FS frames are one contiguous run of FS x M values -- lanes along the run (the last pass' own order would scatter
        // runs of `ns` values: 24 bytes at 48 channels)
        if (!(a.dbg & 8) && !f.direct)
            for (int idx = tid0; idx < f.Q * run; idx += blockDim.x) {
                const int qq = (int)__umulhi((unsigned)idx, f.m_run), e = idx - qq * run;
                const long long step0 = r0 + (long long)qq * a.range_len + (long long)it * FS;
                if (step0 * a.n + e < (long long)a.nframes * a.n) {
                    const c32 v = mr_lds[slot(idx)];
                    f2v z;
                    z.x = v.x;
                    z.y = v.y;
                    __builtin_nontemporal_store(z, (f2v *)a.out + step0 * a.n + e);
                }
            }
        __syncthreads();  // the frames have been read: the next iteration writes them
    }
#undef MR_PASS
}


// ------------------------------------------------------------------------------------------------------------------------------
// Longer lengths of the same kind (15361 ... 921600 points, N = N1 x N2 with both factors 6 ... 960): two passes over HBM, the scheme of
// k_fft_tile in fft.hip with the mixed-radix passes inside.  A workgroup takes sixteen neighbouring columns of a matrix with n rows
// (one 128-byte piece per row), transforms the sixteen columns in LDS and stores
//   pass A (x as N1 rows of N2):  column n2 as row n2 of the workspace, times W_N^(n2 k1)       -> ws[n2][k1]
//   pass B (ws as N2 rows of N1): column k1 back into column k1 of the output                    -> out[k2][k1]
// First pass of either: lanes run along the sixteen columns (coalesced gather), then the passes of mr_pass per column, last pass of
// B: lanes along the columns again (coalesced scatter).  The shifts of clFFT_impl are rotations of the linear index (by floor(N/2)
// on the way in, ceil(N/2) on the way out): a sixteen-column piece stays one contiguous run.  Columns beyond the matrix (N1 or N2
// not a multiple of 16) are idle lanes.
struct TileArgs {
    const void *in;
    c32 *out;
    const float *window;
    const c32 *tw;    // this pass' twiddle runs
    const c32 *twn;   // W_N^k, k < N (pass A)
    int n, ld, big, nframes, npass, in_rot, out_rot, real_in, fs /* LDS slots between the sixteen columns */;
    int nt;  // the sixteen-column pieces are whole 128-byte lines nobody else touches: nontemporal loads / stores (measured equal to plain ones on one box, A/B)
    long long nitems; // frames x tiles
    MrPass pass[kMaxPass];
};

// MODE 0: first pass (gather).  1: LDS -> LDS.  2: last pass of A (column -> workspace row, big twiddle).  3: last pass of B (scatter)
template <int R, int SIGN, int MODE>
__device__ __forceinline__ void tile_pass(const TileArgs &a, const MrPass &ps, c32 *lds, int tid, long long frame, int c0)
{
    constexpr int B = kVals / R;
    constexpr bool COLS = MODE == 0 || MODE == 3;  // lanes along the sixteen columns
    const int TH = blockDim.x, n = a.n, nb = ps.nb, nbt = 16 * nb, ld = a.ld;
    const size_t fbase = (size_t)frame * a.big;
    c32 v[B][R], w1[B];
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int b = tid + TH * i;
        const int fr = COLS ? (b & 15) : (int)__umulhi((unsigned)b, ps.m_nb), bb = COLS ? (b >> 4) : b - fr * nb;
        if constexpr (MODE != 0) {
            const int k = bb - (int)__umulhi((unsigned)bb, ps.m_ns) * ps.ns;
            w1[i] = b < nbt ? a.tw[ps.tw_off + k] : mk(1.f, 0.f);
        }
        if (b < nbt) {
            if constexpr (MODE == 0) {
                const bool live = c0 + fr < ld;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    c32 x = mk(0.f, 0.f);
                    if (live) {
                        long long src = (long long)(bb + r * nb) * ld + c0 + fr + a.in_rot;
                        if (src >= a.big) src -= a.big;
                        if (a.real_in) x = mk(((const float *)a.in)[fbase + src], 0.f);
                        else {
                            const f2v t = a.nt ? __builtin_nontemporal_load((const f2v *)a.in + fbase + src) : ((const f2v *)a.in)[fbase + src];
                            x = mk(t.x, t.y);
                        }
                        if (a.window) {
                            const float w = a.window[src];  // the ORIGINAL position (lib/clFFT_impl.cc:477-493)
                            x = mk(x.x * w, x.y * w);
                        }
                    }
                    v[i][r] = x;
                }
            } else {
                const int base = fr * a.fs + bb;
#pragma unroll
                for (int r = 0; r < R; r++) v[i][r] = lds[slot(base + r * nb)];
            }
        }
    }
    if constexpr (MODE != 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int b = tid + TH * i;
        if (b < nbt) {
            const int fr = COLS ? (b & 15) : (int)__umulhi((unsigned)b, ps.m_nb), bb = COLS ? (b >> 4) : b - fr * nb;
            int g = bb, k = 0;
            if constexpr (MODE != 0) {
                g = (int)__umulhi((unsigned)bb, ps.m_ns);
                k = bb - g * ps.ns;
                c32 sq[4];
                sq[0] = w1[i];
#pragma unroll
                for (int q = 1; q < 4; q++) sq[q] = (R > (1 << q)) ? cmul(sq[q - 1], sq[q - 1]) : sq[q - 1];
#pragma unroll
                for (int r = 1; r < R; r++) {
                    c32 t = mk(1.f, 0.f);
                    bool first = true;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if ((r >> q) & 1) {
                            t = first ? sq[q] : cmul(t, sq[q]);
                            first = false;
                        }
                    v[i][r] = cmul(v[i][r], t);
                }
            }
            dft<R, SIGN>(v[i]);
            if constexpr (MODE == 2) {
                const int n2 = c0 + fr;
                if (n2 < ld) {
                    // W_N^(n2 k1), k1 = k + r ns: one base and one step from the table, the step's powers from four squares
                    const c32 base = a.twn[(long long)n2 * k];
                    c32 sq[4];
                    sq[0] = a.twn[(long long)n2 * ps.ns];
#pragma unroll
                    for (int q = 1; q < 4; q++) sq[q] = (R > (1 << q)) ? cmul(sq[q - 1], sq[q - 1]) : sq[q - 1];
                    f2v *o = (f2v *)a.out + fbase + (size_t)n2 * n + k;
#pragma unroll
                    for (int s = 0; s < R; s++) {
                        const int r = out_index<R>(s);
                        c32 t = base;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if ((r >> q) & 1) t = cmul(t, sq[q]);
                        const c32 z = cmul(v[i][s], t);
                        f2v zz;
                        zz.x = z.x;
                        zz.y = z.y;
                        o[r * ps.ns] = zz;
                    }
                }
            } else if constexpr (MODE == 3) {
                if (c0 + fr < ld) {
#pragma unroll
                    for (int s = 0; s < R; s++) {
                        long long p = (long long)(k + out_index<R>(s) * ps.ns) * ld + c0 + fr - a.out_rot;  // forward + shift: X[k] -> position k - ceil(N/2)
                        if (p < 0) p += a.big;
                        f2v zz;
                        zz.x = v[i][s].x;
                        zz.y = v[i][s].y;
                        if (a.nt) __builtin_nontemporal_store(zz, (f2v *)a.out + fbase + p);
                        else ((f2v *)a.out)[fbase + p] = zz;
                    }
                }
            } else {
                const int base = fr * a.fs + g * ps.ns * R + k;
#pragma unroll
                for (int s = 0; s < R; s++) lds[slot(base + out_index<R>(s) * ps.ns)] = v[i][s];
            }
        }
    }
    if constexpr (MODE < 2) __syncthreads();
}

template <int SIGN, bool PASS_B>
__global__ __launch_bounds__(1024) void k_fft_mr_tile(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) c32 mr_lds[];
    const int tid0 = threadIdx.x;
    const int tiles = (a.ld + 15) / 16;
#define TL_PASS(MODE, P)                                                                       \
    switch (a.pass[P].radix) {                                                                 \
    case 2: tile_pass<2, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 3: tile_pass<3, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 4: tile_pass<4, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 5: tile_pass<5, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 6: tile_pass<6, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 7: tile_pass<7, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 8: tile_pass<8, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 9: tile_pass<9, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;             \
    case 10: tile_pass<10, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    case 11: tile_pass<11, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    case 12: tile_pass<12, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    case 13: tile_pass<13, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    case 14: tile_pass<14, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    case 15: tile_pass<15, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    default: tile_pass<16, SIGN, MODE>(a, a.pass[P], mr_lds, tid, frame, c0); break;           \
    }
    for (long long item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));  // (see k_fft_mr)
        const long long frame = item / tiles;
        const int c0 = (int)(item - frame * tiles) * 16;
        TL_PASS(0, 0)
        for (int p = 1; p < a.npass - 1; p++) { TL_PASS(1, p) }
        if constexpr (PASS_B) { TL_PASS(3, a.npass - 1) } else { TL_PASS(2, a.npass - 1) }
        __syncthreads();  // the next item's first pass writes the memory this one's last pass read
    }
#undef TL_PASS
}

unsigned magic(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }

}  // namespace

namespace {
// factorisation into the radices the kernel has: fewest passes first (every pass is a trip through LDS and two barriers), then the one
// that leaves a thread the most values (16 / R butterflies of R points: 15 at R = 15, 10 at R = 10), provided 1024 threads hold a frame
constexpr int kRadices[] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
struct Factorisation {
    int np = 99, per_thread = 0, sum = 1 << 30;  // (sum of the radices: on a tie the more even split -- 4000 = 10 x 10 x 8 x 5 rather than 16 x 10 x 5 x 5)
    int r[kMaxPass];
};
void search(int n, int m, int depth, int first, int per_thread, int floor_pt, int *cur, Factorisation *best)
{
    if (m == 1) {
        int sum = 0;
        for (int i = 0; i < depth; i++) sum += cur[i];
        if (depth >= 2 && per_thread >= floor_pt && 1024LL * per_thread >= n &&
            (depth < best->np || (depth == best->np && (per_thread > best->per_thread || (per_thread == best->per_thread && sum < best->sum))))) {
            best->np = depth;
            best->per_thread = per_thread;
            best->sum = sum;
            for (int i = 0; i < depth; i++) best->r[i] = cur[i];
        }
        return;
    }
    if (depth >= best->np || depth >= kMaxPass) return;
    for (int i = first; i < (int)(sizeof kRadices / sizeof kRadices[0]); i++) {
        const int r = kRadices[i];
        if (m % r) continue;
        cur[depth] = r;
        const int v = (kVals / r) * r;
        search(n, m / r, depth + 1, i, v < per_thread ? v : per_thread, floor_pt, cur, best);
    }
}
}  // namespace

namespace {
bool plan_len(int n, int sign, int variant, bool column_of_a_tile, MrPlan *plan, std::vector<float> *tw);
}

bool mi355_fft_mr_plan(int n, int sign, int variant, MrPlan *plan, std::vector<float> *tw)
{
    if ((n & (n - 1)) == 0) return false;  // powers of two have kernels of their own (fft.hip)
    return plan_len(n, sign, variant, false, plan, tw);
}

namespace {
// column_of_a_tile: the plan of one factor of a two-pass length -- sixteen columns per workgroup, any length incl. powers of two
bool plan_len(int n, int sign, int variant, bool column_of_a_tile, MrPlan *plan, std::vector<float> *tw)
{
    if (n < 4 || n > 16 * 1024) return false;
    Factorisation fz;
    {
        int cur[kMaxPass];
        search(n, n, 0, 0, kVals, variant ? 14 : 0, cur, &fz);
    }
    if (fz.np > kMaxPass) return false;  // a prime factor above 13, or longer than a workgroup holds: chirp-z
    // order: radices with an odd factor first, largest first (the first pass stores with a stride of R slots between lanes: an odd stride
    // is conflict free; a pass with a power-of-two radix wants a long run Ns of consecutive slots in front of it), then 16s, 8 / 4 / 2 last
    int radix[kMaxPass], np = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < fz.np; i++) {
            const bool pow2 = (fz.r[i] & (fz.r[i] - 1)) == 0;
            if ((pass == 0) != pow2) radix[np++] = fz.r[i];  // (the search lists each group in descending order already)
        }
    int per_thread = kVals;  // values a thread can hold in every pass of this plan
    for (int p = 0; p < np; p++) {
        const int v = (kVals / radix[p]) * radix[p];
        if (v < per_thread) per_thread = v;
    }
    // Workgroup size and frames per iteration (tools/fft_mr_sweep.py: every size x frame count, 12 ... 15360 points).  The fastest
    // arrangements fill the CU's LDS with frames and give them about 1536 threads, 10 - 12 values per thread rather than the 15 - 16 a
    // thread can hold: W workgroups per CU of 1536 / W threads, each with as many frames as its share of the LDS (and its threads) hold;
    // the W with the most values resident per CU wins, the smaller workgroup on a tie.
    int th = 0, frames = 0;
    long long best = 0;
    for (int w : {8, 6, 5, 4, 3, 2, 1}) {
        int t = w == 1 ? 1024 : 1536 / w / 64 * 64;
        long long f = (long long)t * per_thread / n;
        const long long lds_vals = (160 * 1024 / w / 8 - 1) * 32 / 33;  // slots of an LDS share, minus the padding
        if (f > lds_vals / n) f = lds_vals / n;
        if (f < 1) continue;
        if (w * f * n > best) {
            best = w * f * n;
            th = t;
            frames = (int)f;
        }
    }
    if (column_of_a_tile) {
        th = (int)((16LL * n + per_thread - 1) / per_thread + 63) / 64 * 64;
        frames = 16;
        if (th > 1024) return false;
    } else
    if (const char *e = getenv("MI355_FFT_MR_THREADS")) {  // (tuning switches, read at create: threads, frames per iteration)
        const int t = atoi(e), f = getenv("MI355_FFT_MR_FRAMES") ? atoi(getenv("MI355_FFT_MR_FRAMES")) : 1;
        if (t >= 64 && t <= 1024 && t % 64 == 0 && f >= 1 && (long long)f * n <= (long long)t * per_thread) { th = t; frames = f; }
    }
    if (!th) return false;  // longer than a workgroup holds: chirp-z
    plan->n = n;
    plan->npass = np;
    plan->threads = th;
    plan->frames = frames;
    plan->lds_bytes = (plan->frames * n + (plan->frames * n >> 5) + 1) * 8;
    plan->per_thread = per_thread;
    plan->variant = variant;
    tw->clear();
    int ns = 1;
    for (int p = 0; p < np; p++) {
        MrPass &ps = plan->pass[p];
        ps.radix = radix[p];
        ps.ns = ns;
        ps.nb = n / radix[p];
        ps.m_nb = magic(ps.nb);
        ps.m_ns = ns > 1 ? magic(ns) : 0;
        ps.tw_off = (int)(tw->size() / 2);
        if (p > 0)
            for (int k = 0; k < ns; k++) {
                const double a = sign * 2.0 * M_PI * (double)k / ((double)ns * radix[p]);
                tw->push_back((float)cos(a));
                tw->push_back((float)sin(a));
            }
        ns *= radix[p];
    }
    return true;
}
}  // namespace

namespace {
int lds_bytes_for(int n, int frames) { return (frames * n + (frames * n >> 5) + 1) * 8; }

int launch_with(const MrPlan &plan, int threads, int frames, mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, int nframes,
                int shift, int real_in, hipStream_t st)
{
    MrArgs a;
    a.in = in;
    a.out = (c32 *)out;
    a.window = window;
    a.tw = (const c32 *)plan.d_tw;
    a.n = plan.n;
    a.nframes = nframes;
    a.frames = frames;
    a.npass = plan.npass;
    a.in_rot = (sign > 0 && shift) ? plan.n / 2 : 0;         // floor(n/2), lib/clFFT_impl.cc:491
    a.out_rot = (sign < 0 && shift) ? (plan.n + 1) / 2 : 0;  // ceil(n/2), lib/clFFT_impl.cc:503-507
    a.real_in = real_in;
    a.ngroups = ((long long)nframes + frames - 1) / frames;
    for (int p = 0; p < plan.npass; p++) a.pass[p] = plan.pass[p];
    static const bool no_copy_out = getenv("MI355_FFT_MR_NO_COPY_OUT") != nullptr;
    static const int copy_ns = getenv("MI355_FFT_MR_COPY_OUT_NS") ? atoi(getenv("MI355_FFT_MR_COPY_OUT_NS")) : 32;  // (runs under 256 bytes; 120 points 455 -> 290 us per 2^26 samples, 1000 points and up: the last pass own stores are as fast or faster)
    a.copy_out = plan.pass[plan.npass - 1].ns < copy_ns && !no_copy_out;
    a.m_n = magic(plan.n);
    static const int copy_nb = getenv("MI355_FFT_MR_COPY_IN_NB") ? atoi(getenv("MI355_FFT_MR_COPY_IN_NB")) : 16;
    a.copy_in = plan.pass[0].nb < copy_nb && !no_copy_out;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int lds_bytes = lds_bytes_for(plan.n, frames);
    int per_cu = (160 * 1024) / lds_bytes;
    if (per_cu > 2048 / threads) per_cu = 2048 / threads;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)cus * per_cu;
    if (grid > a.ngroups) grid = a.ngroups;
    // (the kernels may use the whole LDS: said once per device and direction, not per launch)
    static std::map<int, bool> lds_ok;  // (device, direction)
    static std::mutex lds_ok_lock;
    {
        std::lock_guard<std::mutex> g(lds_ok_lock);
        const int which = ctx->device * 2 + (sign < 0 ? 0 : 1);
        if (!lds_ok[which]) {
            if (sign < 0) MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            else MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            lds_ok[which] = true;
        }
    }
#define MR_LAUNCH(SG) hipLaunchKernelGGL((k_fft_mr<SG>), dim3((unsigned)grid), dim3(threads), lds_bytes, st, a)
    if (sign < 0) MR_LAUNCH(-1); else MR_LAUNCH(1);
#undef MR_LAUNCH
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

std::mutex g_tuned_lock;
struct Tuned { int variant, threads, frames; };
std::map<int, Tuned> g_tuned;  // length -> what was measured fastest in this process
}  // namespace

int mi355_fft_mr_launch(const MrPlan &plan, mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, int nframes, int shift,
                        int real_in, hipStream_t st)
{
    if (nframes <= 0) return MI355_OK;
    return launch_with(plan, plan.threads, plan.frames, ctx, sign, in, out, window, nframes, shift, real_in, st);
}

// clPolyphaseChannelizer, branch filters + transform in one kernel (k_pfb_mr).  false: not this form (more than 32 taps per arm, more arms than a
// workgroup has threads, too few steps) -- the caller runs its two kernels.
bool mi355_fft_mr_pfb_ok(const MrPlan &plan, int sign, int K, int M, int nsteps)
{
    if (!plan.n || plan.n != M || sign <= 0 || plan.npass < 2 || getenv("MI355_PFB_NO_MR_FUSED")) return false;
    const int P = (K + M - 1) / M;
    return P <= 32 && M <= 512 && (long long)plan.per_thread * 512 >= 8LL * M && nsteps >= 1;
}

int mi355_fft_mr_pfb_launch(const MrPlan &plan, mi355_ctx *ctx, const void *in, void *out, const float *taps, int K, int M, int nsteps, hipStream_t st)
{
    constexpr int FS = 8, FS_SHIFT = 3;  // steps per iteration (16: 250 registers at 32 taps per arm, and no faster at 8 or 16 taps)
    const int P = (K + M - 1) / M;
    static const int th_env = getenv("MI355_PFB_MR_THREADS") ? atoi(getenv("MI355_PFB_MR_THREADS")) : 0;
    auto ranges_of = [&](int th) { const long long byv = (long long)th * plan.per_thread / (FS * M); const int byt = th / M; return (int)(byv < byt ? byv : byt); };
    int th = 512;  // (256 threads, three workgroups per CU: 268 against 226 us at 100 channels, 262 / 251 at 20, 274 / 262 at 200)
    if (th_env >= 64 && th_env <= 512 && th_env % 64 == 0 && ranges_of(th_env) >= 1) th = th_env;
    const int Q = ranges_of(th);
    if (Q < 1) return MI355_ERR_INVALID_ARG;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    int ntw = 0;
    for (int p = 1; p < plan.npass; p++) ntw += plan.pass[p].ns;
    const int tw_lds = lds_bytes_for(M, Q * FS) / 8;
    const int pmax = P <= 8 ? 8 : P <= 16 ? 16 : 32;
    const int lds_bytes = (tw_lds + ntw) * 8 + pmax * M * 4;
    static const int wg_env = getenv("MI355_PFB_MR_WG_PER_CU") ? atoi(getenv("MI355_PFB_MR_WG_PER_CU")) : 0;
    // workgroups a CU holds: 117 / 133 / 165 registers per thread at 8 / 16 / 32 taps per arm = 4 / 3 / 3 waves per SIMD
    int per_cu = wg_env > 0 ? wg_env : (pmax == 8 ? 16 : 12) / (th / 64);
    if (per_cu > (160 * 1024) / lds_bytes) per_cu = (160 * 1024) / lds_bytes;
    if (per_cu < 1) per_cu = 1;
    // ranges: one workgroup per CU first; then as many as the device runs at once, as long as a range stays 8 x the warm-up long
    long long len = (nsteps + (long long)cus * Q - 1) / ((long long)cus * Q);
    if (len >= 8LL * P) {
        const long long nr = (long long)cus * per_cu * Q;
        len = (nsteps + nr - 1) / nr;
        if (len < 8LL * P) len = 8LL * P;
    }
    len = (len + FS - 1) / FS * FS;
    const long long grid = (nsteps + len * Q - 1) / (len * Q);
    MrArgs a;
    a.in = nullptr;
    a.out = (c32 *)out;
    a.window = nullptr;
    a.tw = (const c32 *)plan.d_tw;
    a.n = M;
    a.nframes = nsteps;
    a.frames = Q * FS;
    a.npass = plan.npass;
    a.in_rot = a.out_rot = a.real_in = 0;
    a.fs_shift = FS_SHIFT;
    a.range_len = len;
    a.tw_lds = tw_lds;
    a.ntw = ntw;
    a.dbg = getenv("MI355_PFB_MR_DBG") ? atoi(getenv("MI355_PFB_MR_DBG")) : 0;
    a.ngroups = 0;
    for (int p = 0; p < plan.npass; p++) a.pass[p] = plan.pass[p];
    PfbMr f;
    f.in = (const c32 *)in;
    f.taps = taps;
    f.K = K;
    f.M = M;
    f.Q = Q;
    f.nsteps = nsteps;
    f.m_M = magic(M);
    f.m_run = magic(FS * M);
    f.direct = (a.dbg & 32) ? 1 : (a.dbg & 16) ? 0 : plan.pass[plan.npass - 1].ns >= 8;
    if (lds_bytes > 160 * 1024) return MI355_ERR_INVALID_ARG;
    if (lds_bytes > 48 * 1024) {  // (said once per device)
        static std::map<int, bool> lds_ok;
        static std::mutex lds_ok_lock;
        std::lock_guard<std::mutex> g(lds_ok_lock);
        if (!lds_ok[ctx->device]) {
            MI355_HIP(hipFuncSetAttribute((const void *)k_pfb_mr<1, 8, FS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MI355_HIP(hipFuncSetAttribute((const void *)k_pfb_mr<1, 16, FS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MI355_HIP(hipFuncSetAttribute((const void *)k_pfb_mr<1, 32, FS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            lds_ok[ctx->device] = true;
        }
    }
#define PFB_MR(PM) hipLaunchKernelGGL((k_pfb_mr<1, PM, FS>), dim3((unsigned)grid), dim3(th), lds_bytes, st, a, f)
    if (pmax == 8) PFB_MR(8); else if (pmax == 16) PFB_MR(16); else PFB_MR(32);
#undef PFB_MR
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// The rate over (threads, frames per iteration) is irregular -- 4000 points: 320 threads 163 GS/s, 384 threads 107, 512 threads 131; the
// rule of mi355_fft_mr_plan is within 5 % of the best for most lengths and 35 % off for some -- so the handle measures it once per
// length and process: every workgroup size x a few frame counts on 2^23 zero samples, on the context's upload stream (no other
// stream waits for it), about 10 ms.  MI355_FFT_MR_AUTOTUNE=0 keeps the rule.
int mi355_fft_mr_tune(MrPlan *plan, mi355_ctx *ctx, int sign, const float *window_dev, float *best_ms_out)
{
    const int per_thread = plan->per_thread;
    if (best_ms_out) *best_ms_out = -1.f;
    if (const char *e = getenv("MI355_FFT_MR_AUTOTUNE"))
        if (atoi(e) == 0) return MI355_OK;
    if (getenv("MI355_FFT_MR_THREADS")) return MI355_OK;
    {
        std::lock_guard<std::mutex> g(g_tuned_lock);
        auto it = g_tuned.find(plan->n);
        if (it != g_tuned.end() && it->second.variant == plan->variant) {
            plan->threads = it->second.threads;
            plan->frames = it->second.frames;
            plan->lds_bytes = lds_bytes_for(plan->n, plan->frames);
            return MI355_OK;
        }
    }
    const int n = plan->n;
    const size_t total = (size_t)1 << 23;
    const int nframes = (int)(total / n);
    void *din = nullptr, *dout = nullptr;
    // no room to measure: the rule stands (and the failed allocation must not stay behind as HIP's last error, where the next
    // launch's hipGetLastError() would report it)
    if (hipMalloc(&din, (size_t)nframes * n * 8) != hipSuccess) { (void)hipGetLastError(); return MI355_OK; }
    if (hipMalloc(&dout, (size_t)nframes * n * 8) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(din); return MI355_OK; }
    std::lock_guard<std::mutex> g(ctx->upload_lock);
    hipStream_t st = ctx->upload;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MI355_OK;
    auto done = [&](int r) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(din);
        (void)hipFree(dout);
        if (r == MI355_OK) (void)hipGetLastError();  // (a bail-out above leaves nothing sticky behind)
        return r;
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(MI355_OK);
    if (hipMemsetAsync(din, 0, (size_t)nframes * n * 8, st) != hipSuccess) return done(MI355_OK);
    float best_ms = 1e30f;
    int best_t = plan->threads, best_f = plan->frames;
    auto trial = [&](int t, int f) {
        if (f < 1 || (long long)f * n > (long long)t * per_thread || lds_bytes_for(n, f) > 160 * 1024) return;
        if (launch_with(*plan, t, f, ctx, sign, din, dout, window_dev, nframes, 0, 0, st) != MI355_OK) { rc = MI355_ERR_HIP; return; }
        float ms = 1e30f;
        for (int rep = 0; rep < 2; rep++) {  // the faster of two bursts of three launches
            (void)hipEventRecord(e0, st);
            for (int k = 0; k < 3; k++)
                if (launch_with(*plan, t, f, ctx, sign, din, dout, window_dev, nframes, 0, 0, st) != MI355_OK) { rc = MI355_ERR_HIP; return; }
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = MI355_ERR_HIP; return; }
            float m = 0.f;
            if (hipEventElapsedTime(&m, e0, e1) != hipSuccess) return;
            if (m < ms) ms = m;
        }
        if (ms < best_ms * 0.98f) {  // (a later candidate has to be clearly faster: ties keep the smaller workgroup)
            best_ms = ms;
            best_t = t;
            best_f = f;
        }
    };
    trial(plan->threads, plan->frames);  // the rule's choice first: it stays unless something is 2 % faster
    for (int t = 64; t <= 1024 && rc == MI355_OK; t += 64) {
        const int fmax = (int)((long long)t * per_thread / n);
        if (fmax < 1) continue;
        int tried[4] = {fmax, (fmax * 3 + 3) / 4, (fmax + 1) / 2, 1}, ntried = 0;
        for (int f : tried) {
            bool dup = false;
            for (int q = 0; q < ntried; q++) dup = dup || tried[q] == f;
            if (!dup && rc == MI355_OK) trial(t, f);
            tried[ntried++] = f;
        }
    }
    if (rc != MI355_OK) return done(rc);
    plan->threads = best_t;
    plan->frames = best_f;
    plan->lds_bytes = lds_bytes_for(n, best_f);
    if (best_ms_out) *best_ms_out = best_ms;
    return done(MI355_OK);
}

int mi355_fft_mr_cached_variant(int n)
{
    std::lock_guard<std::mutex> g(g_tuned_lock);
    auto it = g_tuned.find(n);
    return it == g_tuned.end() ? -1 : it->second.variant;
}

void mi355_fft_mr_remember(const MrPlan &plan)
{
    std::lock_guard<std::mutex> g(g_tuned_lock);
    g_tuned[plan.n] = Tuned{plan.variant, plan.threads, plan.frames};
}

// ---- two-pass lengths ------------------------------------------------------------------------------------------------------
namespace {
int tile_fs(int n) { return n | 1; }  // LDS slots between the sixteen columns: odd, so that lanes along the columns meet different banks
int tile_lds_bytes(int n) { const int v = 16 * tile_fs(n); return (v + (v >> 5) + 1) * 8; }
}  // namespace

bool mi355_fft_mr_tile_plan(int n, int sign, MrTilePlan *tp, std::vector<float> *twa, std::vector<float> *twb)
{
    if (n <= 15360 || n > 960 * 960) return false;
    double best = 1e30;
    MrPlan pa, pb;
    std::vector<float> ta, tb;
    for (int n1 = 4; n1 <= 960; n1++) {
        if (n % n1) continue;
        const int n2 = n / n1;
        if (n2 < 4 || n2 > 960) continue;
        bool ok = false;
        for (int va = 0; va < 2 && !ok; va++)
            for (int vb = 0; vb < 2 && !ok; vb++)
                ok = plan_len(n1, sign, va, true, &pa, &ta) && plan_len(n2, sign, vb, true, &pb, &tb) && tile_lds_bytes(n1) <= 160 * 1024 &&
                     tile_lds_bytes(n2) <= 160 * 1024;
        if (!ok) continue;
        // fewest passes; rows that start on a 128-byte line (the row length a multiple of 16 values) count as half a pass each; near-square
        double cost = pa.npass + pb.npass + (n2 % 16 ? 0.5 : 0.0) + (n1 % 16 ? 0.5 : 0.0) + 0.25 * fabs(log2((double)n1 / n2));
        if (cost < best) {
            best = cost;
            tp->n = n;
            tp->n1 = n1;
            tp->n2 = n2;
            tp->a = pa;
            tp->b = pb;
            *twa = ta;
            *twb = tb;
        }
    }
    return best < 1e29;
}

int mi355_fft_mr_tile_launch(const MrTilePlan &tp, mi355_ctx *ctx, int sign, const void *in, void *ws, void *out, const float *window, int nframes,
                             int shift, int real_in, hipStream_t st)
{
    if (nframes <= 0) return MI355_OK;
    static std::map<int, bool> lds_ok;  // (device, direction)
    static std::mutex lds_ok_lock;
    {
        std::lock_guard<std::mutex> g(lds_ok_lock);
        const int which = ctx->device * 2 + (sign < 0 ? 0 : 1);
        if (!lds_ok[which]) {
            if (sign < 0) {
                MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr_tile<-1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr_tile<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            } else {
                MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr_tile<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                MI355_HIP(hipFuncSetAttribute((const void *)k_fft_mr_tile<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            lds_ok[which] = true;
        }
    }
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    for (int pass = 0; pass < 2; pass++) {
        const MrPlan &pl = pass ? tp.b : tp.a;
        TileArgs a;
        a.in = pass ? ws : in;
        a.out = (c32 *)(pass ? out : ws);
        a.window = pass ? nullptr : window;
        a.tw = (const c32 *)pl.d_tw;
        a.twn = (const c32 *)tp.d_twn;
        a.n = pl.n;
        a.ld = pass ? tp.n1 : tp.n2;
        a.big = tp.n;
        a.nframes = nframes;
        a.npass = pl.npass;
        a.in_rot = (!pass && sign > 0 && shift) ? tp.n / 2 : 0;
        a.out_rot = (pass && sign < 0 && shift) ? (tp.n + 1) / 2 : 0;
        a.real_in = pass ? 0 : real_in;
        a.fs = tile_fs(pl.n);
        a.nt = a.ld % 16 == 0 && a.in_rot % 16 == 0 && a.out_rot % 16 == 0;
        a.nitems = (long long)nframes * ((a.ld + 15) / 16);
        for (int p = 0; p < pl.npass; p++) a.pass[p] = pl.pass[p];
        const int lds_bytes = tile_lds_bytes(pl.n);
        int per_cu = (160 * 1024) / lds_bytes;
        if (per_cu > 2048 / pl.threads) per_cu = 2048 / pl.threads;
        if (per_cu < 1) per_cu = 1;
        long long grid = (long long)cus * per_cu;
        if (grid > a.nitems) grid = a.nitems;
        if (sign < 0) {
            if (pass) hipLaunchKernelGGL((k_fft_mr_tile<-1, true>), dim3((unsigned)grid), dim3(pl.threads), lds_bytes, st, a);
            else hipLaunchKernelGGL((k_fft_mr_tile<-1, false>), dim3((unsigned)grid), dim3(pl.threads), lds_bytes, st, a);
        } else {
            if (pass) hipLaunchKernelGGL((k_fft_mr_tile<1, true>), dim3((unsigned)grid), dim3(pl.threads), lds_bytes, st, a);
            else hipLaunchKernelGGL((k_fft_mr_tile<1, false>), dim3((unsigned)grid), dim3(pl.threads), lds_bytes, st, a);
        }
        MI355_HIP(hipGetLastError());
    }
    return MI355_OK;
}
