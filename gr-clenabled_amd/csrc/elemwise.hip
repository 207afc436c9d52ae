// The rest of gr-clenabled's elementwise family: clLog, clSNR, clComplexToMag,
// clComplexToArg, clComplexToMagPhase, clMagPhaseToComplex, clQuadratureDemod.
// Reference kernels: lib/clLog_impl.cc:113-147, lib/clSNR_impl.cc:98-116, lib/clComplexToMag_impl.cc:138-148,
// lib/clComplexToArg_impl.cc:136-151, lib/clComplexToMagPhase_impl.cc:150-164,
// lib/clMagPhaseToComplex_impl.cc:170-191, lib/clQuadratureDemod_impl.cc:118-146.
// The reference evaluates atan2 / sin / cos in DOUBLE on purpose (README.md:112-114,147-153: float trig
// broke downstream decoding); the same is done here.  Four items per thread through 16-byte streaming accesses.
#include <cmath>

#include "common.h"

namespace {

struct c32 { float x, y; };
constexpr int kT = 256;

// one item: a0/a1 = first input (complex: re, im; float: value in a0), b0 = second input or, for QUADDEMOD, b0/b1 = the
// NEXT complex item; r0/r1 = outputs (complex result in r0,r1; two float outputs in r0,r1)
template <int KIND>
__device__ __forceinline__ void elem_one(float a0, float a1, float b0, float b1, float p0, float p1, float &r0, float &r1)
{
    r1 = 0.f;
    if constexpr (KIND == MI355_ELEM_LOG10) {
        r0 = p0 * log10f(a0) + p1;  // log2To10Factor * log2(a) + k, lib/clLog_impl.cc:138-147 (n*log10(a)+k, :200-214)
    } else if constexpr (KIND == MI355_ELEM_SNR) {
        r0 = fabsf(p0 * log10f(a0 / b0) + p1);  // lib/clSNR_impl.cc:110-112
    } else if constexpr (KIND == MI355_ELEM_C2MAG) {
        r0 = sqrtf(a1 * a1 + a0 * a0);  // :144-148
    } else if constexpr (KIND == MI355_ELEM_C2ARG) {
        r0 = (float)atan2((double)a1, (double)a0);  // :145-147
    } else if constexpr (KIND == MI355_ELEM_C2MAGPHASE) {
        r0 = sqrtf(a1 * a1 + a0 * a0);               // :157
        r1 = (float)atan2((double)a1, (double)a0);   // :159
    } else if constexpr (KIND == MI355_ELEM_MAGPHASE2C) {
        double s, c;  // :175-180
        sincos((double)b0, &s, &c);
        r0 = (float)((double)a0 * c);
        r1 = (float)((double)a0 * s);
    } else {  // QUADDEMOD: gain * atan2 of a[i+1] * conj(a[i]) in double, lib/clQuadratureDemod_impl.cc:125-141
        const double ar = b0, ai = b1, br = a0, bi = -1.0 * (double)a1;
        const double re = ar * br - ai * bi, im = ar * bi + ai * br;
        r0 = (float)((double)p0 * atan2(im, re));
    }
}

typedef float f4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(kT) void k_elem(const void *__restrict__ in0, const void *__restrict__ in1, void *__restrict__ out0,
                                             void *__restrict__ out1, size_t n, float p0, float p1, int vec_ok)
{
    constexpr bool CIN = KIND == MI355_ELEM_C2MAG || KIND == MI355_ELEM_C2ARG || KIND == MI355_ELEM_C2MAGPHASE || KIND == MI355_ELEM_QUADDEMOD;
    constexpr bool TWO_IN = KIND == MI355_ELEM_SNR || KIND == MI355_ELEM_MAGPHASE2C;
    constexpr bool COUT = KIND == MI355_ELEM_MAGPHASE2C, TWO_OUT = KIND == MI355_ELEM_C2MAGPHASE;
    // 4 items per thread through 16-byte streaming accesses (the pointers are 16-byte aligned when vec_ok)
    const size_t n4 = vec_ok ? n / 4 : 0;
    for (size_t q = (size_t)blockIdx.x * kT + threadIdx.x; q < n4; q += (size_t)gridDim.x * kT) {
        float a0[4], a1[4] = {0.f, 0.f, 0.f, 0.f}, b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f}, r0[4], r1[4];
        if constexpr (CIN) {
            const f4v u = __builtin_nontemporal_load((const f4v *)in0 + 2 * q), v = __builtin_nontemporal_load((const f4v *)in0 + 2 * q + 1);
            a0[0] = u.x; a1[0] = u.y; a0[1] = u.z; a1[1] = u.w; a0[2] = v.x; a1[2] = v.y; a0[3] = v.z; a1[3] = v.w;
            if constexpr (KIND == MI355_ELEM_QUADDEMOD) {
                const float *nx = (const float *)in0 + 8 * q + 8;  // item 4q+4 (the buffer carries one item of history)
                b0[0] = a0[1]; b1[0] = a1[1]; b0[1] = a0[2]; b1[1] = a1[2]; b0[2] = a0[3]; b1[2] = a1[3]; b0[3] = nx[0]; b1[3] = nx[1];
            }
        } else {
            const f4v u = __builtin_nontemporal_load((const f4v *)in0 + q);
            a0[0] = u.x; a0[1] = u.y; a0[2] = u.z; a0[3] = u.w;
        }
        if constexpr (TWO_IN) {
            const f4v u = __builtin_nontemporal_load((const f4v *)in1 + q);
            b0[0] = u.x; b0[1] = u.y; b0[2] = u.z; b0[3] = u.w;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) elem_one<KIND>(a0[k], a1[k], b0[k], b1[k], p0, p1, r0[k], r1[k]);
        if constexpr (COUT) {
            __builtin_nontemporal_store((f4v){r0[0], r1[0], r0[1], r1[1]}, (f4v *)out0 + 2 * q);
            __builtin_nontemporal_store((f4v){r0[2], r1[2], r0[3], r1[3]}, (f4v *)out0 + 2 * q + 1);
        } else {
            __builtin_nontemporal_store((f4v){r0[0], r0[1], r0[2], r0[3]}, (f4v *)out0 + q);
            if constexpr (TWO_OUT) __builtin_nontemporal_store((f4v){r1[0], r1[1], r1[2], r1[3]}, (f4v *)out1 + q);
        }
    }
    // tail (and the whole call when a pointer is not 16-byte aligned): one item per thread
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        float a0, a1 = 0.f, b0 = 0.f, b1 = 0.f, r0, r1;
        if constexpr (CIN) {
            a0 = ((const float *)in0)[2 * i]; a1 = ((const float *)in0)[2 * i + 1];
            if constexpr (KIND == MI355_ELEM_QUADDEMOD) { b0 = ((const float *)in0)[2 * i + 2]; b1 = ((const float *)in0)[2 * i + 3]; }
        } else {
            a0 = ((const float *)in0)[i];
        }
        if constexpr (TWO_IN) b0 = ((const float *)in1)[i];
        elem_one<KIND>(a0, a1, b0, b1, p0, p1, r0, r1);
        if constexpr (COUT) { ((float *)out0)[2 * i] = r0; ((float *)out0)[2 * i + 1] = r1; }
        else {
            ((float *)out0)[i] = r0;
            if constexpr (TWO_OUT) ((float *)out1)[i] = r1;
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
    auto al16 = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec_ok = al16(i0) && al16(i1) && al16(o0) && al16(o1);
    size_t blocks = ((vec_ok ? (n + 3) / 4 : n) + kT - 1) / kT;
    if (blocks > (size_t)cus * 128) blocks = (size_t)cus * 128;  // many short grid-stride blocks: see mathop.hip
    if (blocks < 1) blocks = 1;
#define ELEM_CASE(K) case K: hipLaunchKernelGGL((k_elem<K>), dim3((unsigned)blocks), dim3(kT), 0, st, i0, i1, o0, o1, n, h->p0, h->p1, vec_ok); break
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
    if (mi355_direct_ok(items * 8)) {  // small call: the kernel works on the pinned staging itself (common.h)
        for (int i = 0; i < h->sh.nin; i++) mi355_copy(h->h_in[i], ins[i], items * h->sh.in_sz[i]);
        int rc = launch_elem(h, n, h->h_in[0], h->h_in[1], h->h_out[0], h->h_out[1], st);
        if (rc) return rc;
        MI355_HIP(mi355_direct_sync(st));
        for (int i = 0; i < h->sh.nout; i++) mi355_copy(outs[i], h->h_out[i], n * h->sh.out_sz[i]);
        return MI355_OK;
    }
    for (int i = 0; i < h->sh.nin; i++) {
        mi355_copy(h->h_in[i], ins[i], items * h->sh.in_sz[i]);
        MI355_HIP(hipMemcpyAsync(h->d_in[i], h->h_in[i], items * h->sh.in_sz[i], hipMemcpyHostToDevice, st));
    }
    int rc = launch_elem(h, n, h->d_in[0], h->d_in[1], h->d_out[0], h->d_out[1], st);
    if (rc) return rc;
    for (int i = 0; i < h->sh.nout; i++) MI355_HIP(hipMemcpyAsync(h->h_out[i], h->d_out[i], n * h->sh.out_sz[i], hipMemcpyDeviceToHost, st));
    MI355_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < h->sh.nout; i++) mi355_copy(outs[i], h->h_out[i], n * h->sh.out_sz[i]);
    return MI355_OK;
}
