// Internal helpers shared by the HIP translation units behind mi355_clenabled.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <new>

#include "mi355_clenabled.h"

struct mi355_ctx {
    int device = 0;
    int debug = 0;
    hipStream_t stream[2] = {nullptr, nullptr};  // [0] = compute stream handed out by mi355_ctx_stream
    int num_cus = 0;
    std::mutex lock;
    // table uploads (create / set_taps): their own stream, so that only this stream is waited for -- a hipDeviceSynchronize() here
    // would stall every other block working on the device while one filter is retuned
    hipStream_t upload = nullptr;
    std::mutex upload_lock;
};

// host -> device copy of a table on the context's upload stream, complete (on the device) when it returns; `src` may be freed at once
hipError_t mi355_upload(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
// device-side fill on the upload stream, complete when it returns
hipError_t mi355_fill(mi355_ctx *ctx, void *dst_dev, int value, size_t bytes);

void mi355_set_error(const char *fmt, ...);
// one diagnostics line to the registered sink (mi355_set_log_callback) or, by default, to stderr; DEBUG / INFO lines are
// dropped unless the context was created with debug != 0 (the reference's setDebug)
void mi355_log(const mi355_ctx *ctx, int level, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define MI355_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            mi355_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return MI355_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

#define MI355_REQUIRE(cond, msg)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            mi355_set_error("invalid argument: %s", msg); \
            return MI355_ERR_INVALID_ARG;                \
        }                                                \
    } while (0)

static inline size_t mi355_dtype_size(int dtype)
{
    switch (dtype) {
    case MI355_DTYPE_COMPLEX: return 8;
    case MI355_DTYPE_FLOAT: return 4;
    case MI355_DTYPE_INT: return 4;
    case MI355_DTYPE_SHORT: return 2;
    case MI355_DTYPE_BYTE: return 2;   // interleaved int8 I,Q (lib/clXEngine_impl.cc:66-68)
    case MI355_DTYPE_PACKEDXY: return 1;
    }
    return 0;
}

// `stream` is a hipStream_t passed through the C ABI as void*; NULL is HIP's
// default (null) stream, which is also torch's default stream.
static inline hipStream_t mi355_pick_stream(mi355_ctx *, void *stream) { return (hipStream_t)stream; }

// Persistent-kernel grid: k workgroups per CU, k in [kmin, kmax] (all resident).  More resident
// workgroups hide latency better, so kmax is the default; a smaller k is taken only when it divides
// the work units more evenly by more than `tol` (a ragged last round costs up to 1/rounds of the
// run time; latency-bound kernels pass a large tol, the bandwidth-bound FFT a small one).  MI355_WG_PER_CU overrides (tuning aid).
static inline int mi355_balanced_grid(const mi355_ctx *ctx, long long units, int kmin, int kmax, double tol = 0.08)
{
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    if (const char *e = getenv("MI355_WG_PER_CU")) {
        const int k = atoi(e);
        if (k > 0) return units < (long long)cus * k ? (units < 1 ? 1 : (int)units) : cus * k;
    }
    if (units <= (long long)cus * kmin) return units < 1 ? 1 : (int)units;
    auto eff = [&](int k) {
        const long long grid = (long long)cus * k, rounds = (units + grid - 1) / grid;
        return (double)units / (double)(rounds * grid);
    };
    int best = kmax;
    for (int k = kmax - 1; k >= kmin; k--)
        if (eff(k) > eff(best) + tol) best = k;
    return cus * best;
}

// ---------------------------------------------------------------------------
// Pinned double-buffered staging for the host-pointer work() path.
// Two slots; slot s runs H2D -> kernel -> D2H on ctx->stream[s], so the copy-in
// of chunk c+1 overlaps the kernel / copy-out of chunk c (north_star: "pinned
// double-buffered H2D/D2H overlapping compute on HIP streams").
// ---------------------------------------------------------------------------
// Host calls whose buffers are at most this large (a scheduler-sized work() call) skip the H2D / D2H copy submissions: the
// kernel reads and writes the pinned staging buffers across PCIe itself -- one launch and one synchronisation per call
// (8192-item clMathOp.work(): 58 us -> 23 us on MI355X).  MI355_NO_DIRECT=1 disables it.
constexpr size_t kDirectBytes = 512u << 10;
static inline bool mi355_direct_ok(size_t max_buffer_bytes)
{
    static const bool off = getenv("MI355_NO_DIRECT") != nullptr;
    return !off && max_buffer_bytes <= kDirectBytes;
}
// Completion wait of such a call (one kernel of a few microseconds): poll the stream for a bounded time, then sleep on it.
// The device's scheduling flags are left alone (a process-wide hipDeviceScheduleSpin would make every blocking wait of
// every block in a thread-per-block flowgraph burn a core); MI355_SPIN_US sets the polling window, 0 = always sleep.
hipError_t mi355_direct_sync(hipStream_t st);

// Staging copy of the host path (caller's pageable buffer <-> pinned staging).  One thread moves ~13 GB/s, which bounded the
// large host calls at 1.8 GS/s (PCIe would carry 3x that): copies of 2 MiB and more are split over a small persistent pool
// (MI355_COPY_THREADS, default 7 helpers; 0 = the calling thread alone), streaming stores (MI355_COPY_STREAM=0: plain memcpy).  If the pool is busy with another block's copy the caller just
// copies alone.
void mi355_copy(void *dst, const void *src, size_t bytes);
// Staging chunk of a large host call of `total` bytes per buffer: 8 MiB pieces keep both DMA directions busy (40 GB/s each way of
// the 48 this link carries with unbounded transfers); a call of only a few MiB is cut into >= 6 pieces of >= 1 MiB so that its
// copy-in, transfers and copy-out overlap at all (one 8 MiB piece ran them strictly one after the other).  MI355_CHUNK_MB overrides.
static inline size_t mi355_chunk_bytes(size_t total)
{
    static const size_t forced = getenv("MI355_CHUNK_MB") ? (size_t)atoi(getenv("MI355_CHUNK_MB")) << 20 : 0;
    if (forced) return forced;
    size_t c = total / 6;
    if (c > ((size_t)8 << 20)) c = (size_t)8 << 20;
    if (c < ((size_t)1 << 20)) c = (size_t)1 << 20;
    return c & ~(size_t)4095;
}
// two copies as one job of the pool (a staging slot's copy-out and copy-in side by side); either may be empty
void mi355_copy2(void *dst0, const void *src0, size_t bytes0, void *dst1, const void *src1, size_t bytes1);
// runs fn(arg, part, parts) for part = 0..parts-1 on the pool (part 0 on the caller); false = pool busy or disabled, nothing was run
bool mi355_parallel(void (*fn)(void *, int part, int parts), void *arg);

struct HostPipe {
    static constexpr int MAXIN = 2;
    // Four slots on the context's two streams (slot s runs on stream s & 1).  A slot's cycle is copy-in -> H2D -> kernel -> D2H ->
    // copy-out; with two slots the host copies of one slot and the DMAs of the other took turns (8 MiB per 0.49 ms = 34 GB/s each
    // way on a link that carries 48); with four the DMA queues never run dry.
    static constexpr int kSlots = 4;
    mi355_ctx *ctx = nullptr;
    size_t cap_in[MAXIN] = {0, 0};
    size_t cap_out = 0;
    void *h_in[kSlots][MAXIN] = {};
    void *d_in[kSlots][MAXIN] = {};
    void *h_out[kSlots] = {};
    void *d_out[kSlots] = {};
    hipEvent_t done[kSlots] = {};

    int init(mi355_ctx *c);
    // staging for `nslots` slots (the first nslots; a frame of 32 MiB and more is staged through two slots only)
    int ensure(int nin, const size_t *in_bytes, size_t out_bytes, int nslots = kSlots);
    int slots_ready = 0;
    void release();
};
