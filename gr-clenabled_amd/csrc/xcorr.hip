// clxcorrelate_fft_vcf as ONE fused gfx950 kernel.
// Reference behaviour: lib/clxcorrelate_fft_vcf_impl.cc:689-711 (make / ctor: io = num_inputs vectors of fftSize
// complex in, num_inputs-1 vectors of fftSize float out; input_type 1 = spectra, 2 = time series), :886-910
// (MultConj: b <- a * conj(b), a = reference input 0), :912-935 (ComplexToMag: sqrt(fma(re,re,im*im))), :1058-1143
// (work: per frame and per signal s >= 1: [FFT both] -> ref * conj(sig) -> unscaled backward FFT (:731) -> magnitude
// -> D2H, then the host swaps the two halves of every output vector, vlen_2 = fftSize/2, :1133-1140).
// The reference enqueues 2-3 writes, up to 3 clFFT transforms, 2 kernels and a read PER FRAME PER SIGNAL.
//
// Here a workgroup owns 4096/N frames: the reference spectrum is produced once per frame group and kept in
// registers; for every other signal the forward transform's registers are multiplied with it in place (the
// reversed radix plan of the inverse consumes exactly the forward plan's output order -- same trick as the
// overlap-save filter), inverse-transformed, and |.| is stored straight into the half-swapped position.
// HBM traffic = the algorithmic minimum: every input sample read once (8 B), every output written once (4 B).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"
#include "fft_core.hpp"

using namespace fftc;

namespace {

constexpr int kMaxInputs = 32;

struct XcArgs {
    const c32 *in[kMaxInputs];
    float *out[kMaxInputs];  // out[s] belongs to input s (s >= 1)
};

typedef float f2v __attribute__((ext_vector_type(2)));

// load one group of frames in the input order of plan PL's first pass; frames >= nframes read as zero
template <int N, class PL>
__device__ __forceinline__ void load_frames(c32 (&v)[16], const c32 *__restrict__ src, int tid, int frames_left)
{
    constexpr int TH = Geo<N>::TH, R0 = PL::radix(0), B0 = N / R0;
#pragma unroll
    for (int q = 0; q < 16 / R0; q++) {
        const int g = tid + TH * q, fr = g / B0, j = g % B0;
        const bool ok = fr < frames_left;
        const c32 *p = src + (ok ? fr * N + j : 0);
#pragma unroll
        for (int r = 0; r < R0; r++) {
            const f2v x = __builtin_nontemporal_load((const f2v *)(p + r * B0));
            v[q * R0 + r] = ok ? mk(x.x, x.y) : mk(0.f, 0.f);
        }
    }
}

// N <= 128 with a radix-16 first pass: the thread's operands are runs of only N/16 elements, so the lanes move
// consecutive elements instead and a padded LDS image (stride 17*B0 slots: conflict free) redistributes them
template <int N, class PL>
__device__ __forceinline__ void load_frames_staged(c32 (&v)[16], const c32 *__restrict__ src, int tid, int frames_left, c32 *lds)
{
    constexpr int TH = Geo<N>::TH, R0 = PL::radix(0), B0 = N / R0, STR = N + B0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int e = tid + TH * k, fr = e / N;
        const bool ok = fr < frames_left;
        const f2v x = __builtin_nontemporal_load((const f2v *)(src + (ok ? e : 0)));
        lds[fr * STR + (e % N)] = ok ? mk(x.x, x.y) : mk(0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16 / R0; q++) {
        const int g = tid + TH * q, base = (g / B0) * STR + (g % B0);
#pragma unroll
        for (int r = 0; r < R0; r++) v[q * R0 + r] = lds[base + r * B0];
    }
    __syncthreads();
}

// runs shorter than 8 elements (64 B) go through the staged loader
template <int N, class PL>
__device__ __forceinline__ void load_any(c32 (&v)[16], const c32 *__restrict__ src, int tid, int frames_left, c32 *lds)
{
    if constexpr (N / PL::radix(0) < 8) load_frames_staged<N, PL>(v, src, tid, frames_left, lds);
    else load_frames<N, PL>(v, src, tid, frames_left);
}

template <int N, bool TIME>
__global__ __launch_bounds__(Geo<N>::TH, Geo<N>::WPE) void k_xcorr(XcArgs a, const c32 *__restrict__ tw_fwd,
                                                                  const c32 *__restrict__ tw_inv, int num_inputs, int nframes,
                                                                  int ngroups)
{
    using PF = Plan<N, false>;
    using PI = Plan<N, true>;
    constexpr int TH = Geo<N>::TH, PTS = Geo<N>::PTS, F = Geo<N>::F, NP = PF::NP;
    constexpr int RL = PF::radix(NP - 1);
    static_assert(PI::radix(0) == RL, "inverse plan must start with the forward plan's last radix");
    constexpr bool SMALL = N <= 128;                       // short runs on the load (forward plan) and the float store side
    __shared__ c32 lds[SMALL ? PTS + PTS / 16 : PTS];
    const int tid0 = threadIdx.x;
    // all-radix-16 sizes: the reversed plan is the forward plan, the inverse twiddles are the conjugates of the forward ones
    constexpr bool SHARE = TIME && (PF::L % 4) == 0;
    TwRegs<N> twf, twi;
    if constexpr (TIME) load_twiddles<N, false>(twf, tid0, tw_fwd);
    if constexpr (!SHARE) load_twiddles<N, true>(twi, tid0, tw_inv);

    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));  // keep address arithmetic inside the loop (see fft.hip)
        const size_t base = (size_t)grp * PTS;
        const int frames_left = nframes - grp * F;
        // ---- reference spectrum, in the inverse plan's input order: R[q*RL + r] = X0[fr][j + r*BL] ----
        c32 R[16];
        if constexpr (TIME) {
            c32 v[16];
            load_any<N, PF>(v, a.in[0] + base, tid, frames_left, lds);
            transform_regs<N, -1, false>(v, twf, lds, tid);
#pragma unroll
            for (int q = 0; q < 16 / RL; q++)
#pragma unroll
                for (int r = 0; r < RL; r++) R[q * RL + r] = v[q * RL + irev<RL>(r)];
            if constexpr (NP > 1) __syncthreads();
        } else {
            load_any<N, PI>(R, a.in[0] + base, tid, frames_left, lds);
        }
        for (int s = 1; s < num_inputs; s++) {
            c32 w[16];
            if constexpr (TIME) {
                c32 v[16];
                load_any<N, PF>(v, a.in[s] + base, tid, frames_left, lds);
                transform_regs<N, -1, false>(v, twf, lds, tid);
#pragma unroll
                for (int q = 0; q < 16 / RL; q++)
#pragma unroll
                    for (int r = 0; r < RL; r++) w[q * RL + r] = cmul(R[q * RL + r], cconj(v[q * RL + irev<RL>(r)]));
                if constexpr (NP > 1) __syncthreads();  // the forward transform's LDS reads are done
            } else {
                load_any<N, PI>(w, a.in[s] + base, tid, frames_left, lds);
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = cmul(R[i], cconj(w[i]));
            }
            if constexpr (SHARE) transform_regs<N, 1, true, Geo<N>, 0, true>(w, twf, lds, tid);
            else transform_regs<N, 1, true>(w, twi, lds, tid);
            // ---- |.|, stored with the two halves of the vector swapped ----
            constexpr int RO = PI::radix(NP - 1), BO = N / RO;
            float *__restrict__ dst = a.out[s] + base;
            if constexpr (SMALL) {
                // float image in LDS (frame stride 17*BO floats: conflict free), then lane-consecutive stores
                constexpr int STR = N + BO;
                float *ldsf = (float *)lds;
                if constexpr (NP > 1) __syncthreads();  // the inverse transform's LDS reads are done
#pragma unroll
                for (int q = 0; q < 16 / RO; q++) {
                    const int g = tid + TH * q, fb = (g / BO) * STR + (g % BO);
#pragma unroll
                    for (int t = 0; t < RO; t++) {
                        const c32 z = w[q * RO + t];
                        ldsf[fb + ((orev<RO>(t) * BO) ^ (N / 2))] = sqrtf(fmaf(z.x, z.x, z.y * z.y));
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int e = tid + TH * k, fr = e / N;
                    if (fr < frames_left) __builtin_nontemporal_store(ldsf[fr * STR + (e % N)], dst + e);
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int q = 0; q < 16 / RO; q++) {
                    const int g = tid + TH * q, fr = g / BO, j = g % BO;
                    if (fr < frames_left) {
#pragma unroll
                        for (int t = 0; t < RO; t++) {
                            const int n = j + orev<RO>(t) * BO;
                            const c32 z = w[q * RO + t];
                            __builtin_nontemporal_store(sqrtf(fmaf(z.x, z.x, z.y * z.y)), dst + fr * N + (n ^ (N / 2)));
                        }
                    }
                }
            }
            if constexpr (NP > 1) __syncthreads();
        }
    }
}

}  // namespace

struct mi355_xcorr_fft {
    mi355_ctx *ctx;
    int n, num_inputs, time_series;
    void *d_twf = nullptr, *d_twi = nullptr;
    // device staging for the host-pointer path (grow only)
    void *d_in = nullptr, *d_out = nullptr;
    size_t cap_frames = 0;
    // every other even size clFFT plans (8192, 16384, ... and lengths that are not a power of two): the steps of the reference's work()
    // one after the other over clFFT handles -- forward transforms, X0 conj(Xs), backward transform, |.| with the half swap
    bool composed = false;
    mi355_fft *fwd = nullptr, *inv = nullptr;
    void *d_ws[3] = {nullptr, nullptr, nullptr};  // reference spectrum, signal spectrum / magnitude source, product
    size_t ws_frames = 0;
    std::mutex ws_lock;
};

namespace {

template <int N> int launch_xcorr_n(mi355_xcorr_fft *h, const XcArgs &a, int nframes, hipStream_t st)
{
    constexpr int F = Geo<N>::F, TH = Geo<N>::TH;
    const int ngroups = (nframes + F - 1) / F;
    const int grid = mi355_balanced_grid(h->ctx, ngroups, 2, 3);
    if (h->time_series)
        hipLaunchKernelGGL((k_xcorr<N, true>), dim3(grid), dim3(TH), 0, st, a, (const c32 *)h->d_twf, (const c32 *)h->d_twi,
                           h->num_inputs, nframes, ngroups);
    else
        hipLaunchKernelGGL((k_xcorr<N, false>), dim3(grid), dim3(TH), 0, st, a, (const c32 *)h->d_twf, (const c32 *)h->d_twi,
                           h->num_inputs, nframes, ngroups);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// MultConj of the reference (lib/clxcorrelate_fft_vcf_impl.cc:886-910): p = a * conj(b), same operation order
__global__ __launch_bounds__(256) void k_xc_mulconj(const c32 *__restrict__ a, const c32 *__restrict__ b, c32 *__restrict__ p, long long total)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const c32 x = a[e], y = b[e];
        const float b_r = y.x, b_i = -y.y;
        p[e] = mk((x.x * b_r) - (x.y * b_i), (x.x * b_i) + (x.y * b_r));
    }
}
// ComplexToMag (:912-935) and the host's exchange of the two halves (:1133-1140) in one store
__global__ __launch_bounds__(256) void k_xc_mag_swap(const c32 *__restrict__ p, float *__restrict__ out, int n, long long total)
{
    const int half = n / 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long f = e / n;
        const int k = (int)(e - f * n);
        const c32 z = p[e];
        out[f * n + (k < half ? k + half : k - half)] = sqrtf(fmaf(z.x, z.x, z.y * z.y));
    }
}

int launch_composed(mi355_xcorr_fft *h, const XcArgs &a, int nframes, hipStream_t st)
{
    const size_t n = (size_t)h->n;
    size_t chunk = ((size_t)128 << 20) / (n * 8);  // three work buffers of at most 128 MiB
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)nframes) chunk = (size_t)nframes;
    std::lock_guard<std::mutex> g(h->ws_lock);
    if (chunk > h->ws_frames) {
        MI355_HIP(hipStreamSynchronize(st));
        for (void *&w : h->d_ws) {
            if (w) (void)hipFree(w);
            w = nullptr;
        }
        h->ws_frames = 0;
        for (void *&w : h->d_ws) MI355_HIP(hipMalloc(&w, chunk * n * 8));
        h->ws_frames = chunk;
    }
    const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += chunk) {
        const int nf = (int)((size_t)nframes - f0 < chunk ? (size_t)nframes - f0 : chunk);
        const long long total = (long long)nf * (long long)n;
        long long blocks = (total + 255) / 256;
        if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
        const c32 *ref = a.in[0] + f0 * n;
        int rc;
        if (h->time_series) {
            if ((rc = mi355_fft_work_dev(h->fwd, nf, ref, h->d_ws[0], st))) return rc;
            ref = (const c32 *)h->d_ws[0];
        }
        for (int s = 1; s < h->num_inputs; s++) {
            const c32 *sig = a.in[s] + f0 * n;
            if (h->time_series) {
                if ((rc = mi355_fft_work_dev(h->fwd, nf, sig, h->d_ws[1], st))) return rc;
                sig = (const c32 *)h->d_ws[1];
            }
            hipLaunchKernelGGL(k_xc_mulconj, dim3((unsigned)blocks), dim3(256), 0, st, ref, sig, (c32 *)h->d_ws[2], total);
            if ((rc = mi355_fft_work_dev(h->inv, nf, h->d_ws[2], h->d_ws[1], st))) return rc;
            hipLaunchKernelGGL(k_xc_mag_swap, dim3((unsigned)blocks), dim3(256), 0, st, (const c32 *)h->d_ws[1], a.out[s] + f0 * n, (int)n, total);
            MI355_HIP(hipGetLastError());
        }
    }
    // the buffers are shared by the calls on this handle: the next caller's stream has to see these kernels finished
    MI355_HIP(hipStreamSynchronize(st));
    return MI355_OK;
}

int launch_xcorr(mi355_xcorr_fft *h, const XcArgs &a, int nframes, hipStream_t st)
{
    if (h->composed) return launch_composed(h, a, nframes, st);
    switch (h->n) {
    case 16: return launch_xcorr_n<16>(h, a, nframes, st);
    case 32: return launch_xcorr_n<32>(h, a, nframes, st);
    case 64: return launch_xcorr_n<64>(h, a, nframes, st);
    case 128: return launch_xcorr_n<128>(h, a, nframes, st);
    case 256: return launch_xcorr_n<256>(h, a, nframes, st);
    case 512: return launch_xcorr_n<512>(h, a, nframes, st);
    case 1024: return launch_xcorr_n<1024>(h, a, nframes, st);
    case 2048: return launch_xcorr_n<2048>(h, a, nframes, st);
    case 4096: return launch_xcorr_n<4096>(h, a, nframes, st);
    }
    mi355_set_error("internal: no cross-correlator kernel for FFT size %d", h->n);
    return MI355_ERR_STATE;
}

}  // namespace

extern "C" int mi355_xcorr_fft_create(mi355_ctx *ctx, int fft_size, int num_inputs, int input_type, mi355_xcorr_fft **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(num_inputs >= 2, "the cross-correlator needs a reference input and at least one more (num_inputs >= 2)");
    if (num_inputs > kMaxInputs) {
        mi355_set_error("num_inputs %d exceeds the supported maximum of %d", num_inputs, kMaxInputs);
        return MI355_ERR_UNSUPPORTED;
    }
    MI355_REQUIRE(input_type == 1 || input_type == 2, "input_type must be 1 (spectra) or 2 (time series)");
    const bool fused = fft_size >= 16 && fft_size <= 4096 && (fft_size & (fft_size - 1)) == 0;
    if (!fused && (fft_size < 2 || (fft_size & 1) || fft_size > (1 << 22))) {
        // (an odd size would leave the last output of every vector unwritten in the reference: vlen_2 = fftSize / 2, :1133-1140)
        mi355_set_error("cross-correlator FFT size %d not supported (even sizes 2..4194304; powers of two 16..4096 run as one fused kernel)", fft_size);
        return MI355_ERR_UNSUPPORTED;
    }
    mi355_xcorr_fft *h = new (std::nothrow) mi355_xcorr_fft();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->n = fft_size; h->num_inputs = num_inputs; h->time_series = input_type == 2;
    if (hipSetDevice(ctx->device) != hipSuccess) { delete h; mi355_set_error("hipSetDevice failed"); return MI355_ERR_HIP; }
    if (!fused) {
        h->composed = true;
        int rc = h->time_series ? mi355_fft_create(ctx, fft_size, MI355_FFT_FORWARD, nullptr, 0, MI355_DTYPE_COMPLEX, 1, 0, &h->fwd) : MI355_OK;
        if (rc == MI355_OK) rc = mi355_fft_create(ctx, fft_size, MI355_FFT_BACKWARD, nullptr, 0, MI355_DTYPE_COMPLEX, 1, 0, &h->inv);
        if (rc != MI355_OK) {
            if (h->fwd) (void)mi355_fft_destroy(h->fwd);
            delete h;
            return rc;
        }
        mi355_log(ctx, MI355_LOG_INFO, "clxcorrelate_fft_vcf: %d points, %d inputs: clFFT transforms and two elementwise kernels per signal", fft_size, num_inputs);
        *out = h;
        return MI355_OK;
    }
    const int n = fft_size;
    std::vector<float> twf(2 * (size_t)n), twi(2 * (size_t)n);
    for (int k = 0; k < n; k++) {
        const double ang = -2.0 * M_PI * (double)k / (double)n;
        twf[2 * k] = (float)cos(ang); twf[2 * k + 1] = (float)sin(ang);
        twi[2 * k] = (float)cos(ang); twi[2 * k + 1] = (float)(-sin(ang));
    }
    const size_t bytes = 2 * (size_t)n * sizeof(float);
    if (hipMalloc(&h->d_twf, bytes) != hipSuccess || hipMalloc(&h->d_twi, bytes) != hipSuccess ||
        mi355_upload(ctx, h->d_twf, twf.data(), bytes) != hipSuccess ||
        mi355_upload(ctx, h->d_twi, twi.data(), bytes) != hipSuccess) {
        if (h->d_twf) (void)hipFree(h->d_twf);
        if (h->d_twi) (void)hipFree(h->d_twi);
        delete h;
        mi355_set_error("device allocation of the twiddle tables failed");
        return MI355_ERR_NOMEM;
    }
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_xcorr_fft_destroy(mi355_xcorr_fft *h)
{
    if (!h) return MI355_OK;
    (void)hipSetDevice(h->ctx->device);
    if (h->d_twf) (void)hipFree(h->d_twf);
    if (h->d_twi) (void)hipFree(h->d_twi);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->fwd) (void)mi355_fft_destroy(h->fwd);
    if (h->inv) (void)mi355_fft_destroy(h->inv);
    for (void *w : h->d_ws)
        if (w) (void)hipFree(w);
    delete h;
    return MI355_OK;
}

extern "C" int mi355_xcorr_fft_work_dev(mi355_xcorr_fft *h, int nframes, const void *const *d_inputs, void *const *d_outputs,
                                        void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (nframes <= 0) return MI355_OK;
    MI355_REQUIRE(d_inputs && d_outputs, "NULL pointer arrays");
    if ((long long)nframes * h->n > 0x7fffffffLL) { mi355_set_error("work() call too large"); return MI355_ERR_INVALID_ARG; }
    XcArgs a{};
    for (int s = 0; s < h->num_inputs; s++) {
        MI355_REQUIRE(d_inputs[s] != nullptr, "NULL input buffer");
        a.in[s] = (const c32 *)d_inputs[s];
        if (s >= 1) {
            MI355_REQUIRE(d_outputs[s - 1] != nullptr, "NULL output buffer");
            a.out[s] = (float *)d_outputs[s - 1];
        }
    }
    MI355_HIP(hipSetDevice(h->ctx->device));
    return launch_xcorr(h, a, nframes, mi355_pick_stream(h->ctx, stream));
}

extern "C" int mi355_xcorr_fft_work(mi355_xcorr_fft *h, int nframes, const void *const *inputs, void *const *outputs)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (nframes <= 0) return MI355_OK;
    MI355_REQUIRE(inputs && outputs, "NULL pointer arrays");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    const size_t in_frame = 8 * (size_t)h->n, out_frame = 4 * (size_t)h->n;
    // chunks of frames so the staging stays bounded (64 MiB of input per chunk at most)
    size_t chunk = (64u << 20) / (in_frame * (size_t)h->num_inputs);
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)nframes) chunk = (size_t)nframes;
    if (chunk > h->cap_frames) {
        if (h->d_in) (void)hipFree(h->d_in);
        if (h->d_out) (void)hipFree(h->d_out);
        h->d_in = h->d_out = nullptr; h->cap_frames = 0;
        MI355_HIP(hipMalloc(&h->d_in, chunk * in_frame * (size_t)h->num_inputs));
        MI355_HIP(hipMalloc(&h->d_out, chunk * out_frame * (size_t)(h->num_inputs - 1)));
        h->cap_frames = chunk;
    }
    hipStream_t st = h->ctx->stream[0];
    for (size_t f0 = 0; f0 < (size_t)nframes; f0 += chunk) {
        const size_t nf = (size_t)nframes - f0 < chunk ? (size_t)nframes - f0 : chunk;
        XcArgs a{};
        for (int s = 0; s < h->num_inputs; s++) {
            MI355_REQUIRE(inputs[s] != nullptr, "NULL input buffer");
            char *d = (char *)h->d_in + (size_t)s * chunk * in_frame;
            MI355_HIP(hipMemcpyAsync(d, (const char *)inputs[s] + f0 * in_frame, nf * in_frame, hipMemcpyHostToDevice, st));
            a.in[s] = (const c32 *)d;
            if (s >= 1) a.out[s] = (float *)((char *)h->d_out + (size_t)(s - 1) * chunk * out_frame);
        }
        int rc = launch_xcorr(h, a, (int)nf, st);
        if (rc) return rc;
        for (int s = 1; s < h->num_inputs; s++) {
            MI355_REQUIRE(outputs[s - 1] != nullptr, "NULL output buffer");
            MI355_HIP(hipMemcpyAsync((char *)outputs[s - 1] + f0 * out_frame, a.out[s], nf * out_frame, hipMemcpyDeviceToHost, st));
        }
        MI355_HIP(hipStreamSynchronize(st));
    }
    return MI355_OK;
}
