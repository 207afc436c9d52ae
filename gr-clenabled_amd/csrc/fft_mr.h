// Mixed-radix clFFT path (fft_mr.hip): lengths 2^a 3^b 5^c 7^d 11^e 13^f that are not a power of two, one workgroup-resident Stockham pass
// per radix.  The reference's clFFT library plans these lengths natively (lib/clFFT_impl.cc:91-128: clfftCreateDefaultPlan on any
// length whose prime factors are 2, 3, 5, 7, 11, 13); everything else goes through the chirp-z path of fft.hip.
#pragma once
#include <vector>

#include "common.h"

struct MrPass {
    int radix, ns, nb;          // butterflies of this pass are `radix` points wide, `ns` = product of the radices before it, nb = n / radix
    unsigned m_nb, m_ns;        // ceil(2^32 / nb), ceil(2^32 / ns): exact quotients for operands below 2^16 (one multiply-high each)
    int tw_off;                 // first entry of this pass' twiddle run exp(sign 2 pi i k / (ns radix)), k < ns
};

struct MrPlan {
    int n = 0, npass = 0;
    int threads = 0;            // workgroup size (a multiple of 64, at most 1024)
    int frames = 0;             // frames a workgroup transforms per iteration
    int lds_bytes = 0;
    int variant = 0;            // 0: fewest passes; 1: at least 14 values per thread in every pass, then fewest passes
    int per_thread = 0;         // values a thread holds in the tightest pass (14 with a radix 7, 15 with 3 or 5)
    MrPass pass[12];
    void *d_tw = nullptr;       // the passes' twiddle runs, back to back
};

// false: n is not of this form (or does not fit a workgroup): use the chirp-z path.  tw receives the host copy of the twiddle runs.
bool mi355_fft_mr_plan(int n, int sign, int variant, MrPlan *plan, std::vector<float> *tw);

// Measures (threads, frames) for this plan and keeps the fastest; *best_ms = time of the fastest trial (-1: nothing was measured --
// MI355_FFT_MR_AUTOTUNE=0, a forced size, no memory, or this length's result is remembered).  plan->d_tw and window_dev must be in place.
int mi355_fft_mr_tune(MrPlan *plan, mi355_ctx *ctx, int sign, const float *window_dev, float *best_ms);
// what a previous handle of this process found for length n: the variant it kept (-1: nothing yet) / store the winner
int mi355_fft_mr_cached_variant(int n);
void mi355_fft_mr_remember(const MrPlan &plan);

// in: nframes frames of n values (complex, or float when real_in); out: nframes x n complex.  window: n floats (all ones = no window).
// shift as in oracle_fft_block: reverse = input halves swapped (window indexed by the original position), forward = output rotated by ceil(n/2).
int mi355_fft_mr_launch(const MrPlan &plan, mi355_ctx *ctx, int sign, const void *in, void *out, const float *window, int nframes, int shift,
                        int real_in, hipStream_t st);

// clPolyphaseChannelizer's branch filters + M-point transform in ONE kernel (k_pfb_mr, fft_mr.hip): in = the block's input (history first), taps =
// K floats on the device, out = nsteps x M complex.  _ok: this plan / tap count / call size has the fused form (sign: the plan's direction, +1 only).
bool mi355_fft_mr_pfb_ok(const MrPlan &plan, int sign, int K, int M, int nsteps);
int mi355_fft_mr_pfb_launch(const MrPlan &plan, mi355_ctx *ctx, const void *in, void *out, const float *taps, int K, int M, int nsteps, hipStream_t st);
struct mi355_fft;
const MrPlan *mi355_fft_mr_plan_of(const mi355_fft *h, int *sign);  // (fft.hip) the handle's one-pass mixed-radix plan, nullptr if it has none

// Two-pass form for longer lengths of the same kind (15361 ... 921600 points): n = n1 x n2, both 4 ... 960, every pass the mixed-radix
// passes over sixteen columns of a matrix (fft_mr.hip).  The caller uploads a.d_tw / b.d_tw (twa / twb) and d_twn = W_n^k, k < n.
struct MrTilePlan {
    int n = 0, n1 = 0, n2 = 0;
    MrPlan a, b;           // the n1-point transforms of pass A, the n2-point transforms of pass B
    void *d_twn = nullptr;
};
bool mi355_fft_mr_tile_plan(int n, int sign, MrTilePlan *tp, std::vector<float> *twa, std::vector<float> *twb);
// ws: nframes x n complex workspace
int mi355_fft_mr_tile_launch(const MrTilePlan &tp, mi355_ctx *ctx, int sign, const void *in, void *ws, void *out, const float *window, int nframes,
                             int shift, int real_in, hipStream_t st);
