// The rest of gr-clenabled's elementwise family: clLog, clSNR, clComplexToMag,
// clComplexToArg, clComplexToMagPhase, clMagPhaseToComplex, clQuadratureDemod.
// Reference kernels: lib/clLog_impl.cc:113-147, lib/clSNR_impl.cc:98-116, lib/clComplexToMag_impl.cc:138-148,
// lib/clComplexToArg_impl.cc:136-151, lib/clComplexToMagPhase_impl.cc:150-164,
// lib/clMagPhaseToComplex_impl.cc:170-191, lib/clQuadratureDemod_impl.cc:118-146.
// The reference evaluates atan2 / sin / cos in DOUBLE on purpose (README.md:112-114,147-153: float trig
// broke downstream decoding); the same is done here.  All kernels stream with 8/16 B per lane.
#include <cmath>

#include "common.h"

namespace {

struct c32 { float x, y; };
constexpr int kT = 256;

template <int KIND>
__global__ __launch_bounds__(kT) void k_elem(const void *__restrict__ in0, const void *__restrict__ in1, void *__restrict__ out0,
                                             void *__restrict__ out1, size_t n, float p0, float p1)
{
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        if constexpr (KIND == MI355_ELEM_LOG10) {
            // log2To10Factor * log2(a) + k, lib/clLog_impl.cc:138-147 (n*log10(a)+k, :200-214)
            ((float *)out0)[i] = p0 * log10f(((const float *)in0)[i]) + p1;
        } else if constexpr (KIND == MI355_ELEM_SNR) {
            const float t = ((const float *)in0)[i] / ((const float *)in1)[i];
            ((float *)out0)[i] = fabsf(p0 * log10f(t) + p1);  // lib/clSNR_impl.cc:110-112
        } else if constexpr (KIND == MI355_ELEM_C2MAG) {
            const c32 a = ((const c32 *)in0)[i];
            ((float *)out0)[i] = sqrtf(a.y * a.y + a.x * a.x);  // :144-148
        } else if constexpr (KIND == MI355_ELEM_C2ARG) {
            const c32 a = ((const c32 *)in0)[i];
            ((float *)out0)[i] = (float)atan2((double)a.y, (double)a.x);  // :145-147
        } else if constexpr (KIND == MI355_ELEM_C2MAGPHASE) {
            const c32 a = ((const c32 *)in0)[i];
            ((float *)out0)[i] = sqrtf(a.y * a.y + a.x * a.x);            // :157
            ((float *)out1)[i] = (float)atan2((double)a.y, (double)a.x);  // :159
        } else if constexpr (KIND == MI355_ELEM_MAGPHASE2C) {
            const double mag = (double)((const float *)in0)[i], ph = (double)((const float *)in1)[i];  // :175-180
            double s, c;
            sincos(ph, &s, &c);
            c32 r;
            r.x = (float)(mag * c);
            r.y = (float)(mag * s);
            ((c32 *)out0)[i] = r;
        } else {  // QUADDEMOD: gain * atan2 of a[i+1] * conj(a[i]) in double, lib/clQuadratureDemod_impl.cc:125-141
            const c32 a1 = ((const c32 *)in0)[i + 1], a0 = ((const c32 *)in0)[i];
            const double ar = a1.x, ai = a1.y, br = a0.x, bi = -1.0 * (double)a0.y;
            const double re = ar * br - ai * bi, im = ar * bi + ai * br;
            ((float *)out0)[i] = (float)((double)p0 * atan2(im, re));
        }
    }
}

struct Shape { int nin, nout; size_t in_sz[2], out_sz[2]; int hist; };

Shape shape_of(int kind)
{
    switch (kind) {
    case MI355_ELEM_LOG10: return {1, 1, {4, 0}, {4, 0}, 0};
    case MI355_ELEM_SNR: return {2, 1, {4, 4}, {4, 0}, 0};
    case MI355_ELEM_C2MAG: return {1, 1, {8, 0}, {4, 0}, 0};
    case MI355_ELEM_C2ARG: return {1, 1, {8, 0}, {4, 0}, 0};
    case MI355_ELEM_C2MAGPHASE: return {1, 2, {8, 0}, {4, 4}, 0};
    case MI355_ELEM_MAGPHASE2C: return {2, 1, {4, 4}, {8, 0}, 0};
    case MI355_ELEM_QUADDEMOD: return {1, 1, {8, 0}, {4, 0}, 1};  // set_history(2), :81
    }
    return {0, 0, {0, 0}, {0, 0}, 0};
}

}  // namespace

struct mi355_elem {
    mi355_ctx *ctx;
    int kind;
    float p0, p1;
    Shape sh;
    void *h_in[2] = {nullptr, nullptr}, *d_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    size_t cap = 0;
};

namespace {

int launch_elem(mi355_elem *h, size_t n, const void *i0, const void *i1, void *o0, void *o1, hipStream_t st)
{
    int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
    size_t blocks = (n + kT - 1) / kT;
    if (blocks > (size_t)cus * 16) blocks = (size_t)cus * 16;
    if (blocks < 1) blocks = 1;
#define ELEM_CASE(K) case K: hipLaunchKernelGGL((k_elem<K>), dim3((unsigned)blocks), dim3(kT), 0, st, i0, i1, o0, o1, n, h->p0, h->p1); break
    switch (h->kind) {
        ELEM_CASE(MI355_ELEM_LOG10); ELEM_CASE(MI355_ELEM_SNR); ELEM_CASE(MI355_ELEM_C2MAG); ELEM_CASE(MI355_ELEM_C2ARG);
        ELEM_CASE(MI355_ELEM_C2MAGPHASE); ELEM_CASE(MI355_ELEM_MAGPHASE2C); ELEM_CASE(MI355_ELEM_QUADDEMOD);
    }
#undef ELEM_CASE
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

void elem_free(mi355_elem *h)
{
    for (int i = 0; i < 2; i++) {
        if (h->h_in[i]) (void)hipHostFree(h->h_in[i]);
        if (h->d_in[i]) (void)hipFree(h->d_in[i]);
        if (h->h_out[i]) (void)hipHostFree(h->h_out[i]);
        if (h->d_out[i]) (void)hipFree(h->d_out[i]);
        h->h_in[i] = h->d_in[i] = h->h_out[i] = h->d_out[i] = nullptr;
    }
    h->cap = 0;
}

}  // namespace

extern "C" int mi355_elem_create(mi355_ctx *ctx, int kind, float p0, float p1, mi355_elem **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(kind >= MI355_ELEM_LOG10 && kind <= MI355_ELEM_QUADDEMOD, "unknown elementwise kind");
    mi355_elem *h = new (std::nothrow) mi355_elem();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->kind = kind; h->p0 = p0; h->p1 = p1; h->sh = shape_of(kind);
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_elem_destroy(mi355_elem *h)
{
    if (!h) return MI355_OK;
    (void)hipSetDevice(h->ctx->device);
    elem_free(h);
    delete h;
    return MI355_OK;
}

extern "C" int mi355_elem_history(const mi355_elem *h) { return h ? h->sh.hist + 1 : MI355_ERR_INVALID_ARG; }

extern "C" int mi355_elem_work_dev(mi355_elem *h, size_t n, const void *in0, const void *in1, void *out0, void *out1, void *stream)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (n == 0) return MI355_OK;
    MI355_REQUIRE(in0 && out0 && (h->sh.nin < 2 || in1) && (h->sh.nout < 2 || out1), "NULL buffer");
    MI355_HIP(hipSetDevice(h->ctx->device));
    return launch_elem(h, n, in0, in1, out0, out1, mi355_pick_stream(h->ctx, stream));
}

// host path: one staged transfer per call (these blocks run at GNU Radio buffer sizes)
extern "C" int mi355_elem_work(mi355_elem *h, size_t n, const void *in0, const void *in1, void *out0, void *out1)
{
    MI355_REQUIRE(h != nullptr, "handle is NULL");
    if (n == 0) return MI355_OK;
    MI355_REQUIRE(in0 && out0 && (h->sh.nin < 2 || in1) && (h->sh.nout < 2 || out1), "NULL buffer");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    const size_t items = n + h->sh.hist;
    if (items > h->cap) {
        elem_free(h);
        for (int i = 0; i < h->sh.nin; i++) {
            MI355_HIP(hipHostMalloc(&h->h_in[i], items * h->sh.in_sz[i], hipHostMallocDefault));
            MI355_HIP(hipMalloc(&h->d_in[i], items * h->sh.in_sz[i]));
        }
        for (int i = 0; i < h->sh.nout; i++) {
            MI355_HIP(hipHostMalloc(&h->h_out[i], items * h->sh.out_sz[i], hipHostMallocDefault));
            MI355_HIP(hipMalloc(&h->d_out[i], items * h->sh.out_sz[i]));
        }
        h->cap = items;
    }
    hipStream_t st = h->ctx->stream[0];
    const void *ins[2] = {in0, in1};
    void *outs[2] = {out0, out1};
    for (int i = 0; i < h->sh.nin; i++) {
        memcpy(h->h_in[i], ins[i], items * h->sh.in_sz[i]);
        MI355_HIP(hipMemcpyAsync(h->d_in[i], h->h_in[i], items * h->sh.in_sz[i], hipMemcpyHostToDevice, st));
    }
    int rc = launch_elem(h, n, h->d_in[0], h->d_in[1], h->d_out[0], h->d_out[1], st);
    if (rc) return rc;
    for (int i = 0; i < h->sh.nout; i++) MI355_HIP(hipMemcpyAsync(h->h_out[i], h->d_out[i], n * h->sh.out_sz[i], hipMemcpyDeviceToHost, st));
    MI355_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < h->sh.nout; i++) memcpy(outs[i], h->h_out[i], n * h->sh.out_sz[i]);
    return MI355_OK;
}
