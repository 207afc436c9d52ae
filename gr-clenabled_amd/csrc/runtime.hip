// Device runtime behind mi355_clenabled.h: context, streams, errors, staging.
// Takes the place of the reference's GRCLBase OpenCL shim
// (include/clenabled/GRCLBase.h:77-141, lib/GRCLBase.cpp:17-474): one context =
// one HIP device + two streams; errors are status codes, never exit().
#include <cstdarg>

#include "common.h"

#include <atomic>

static thread_local char g_err[512] = "";

namespace {
struct LogSink {
    mi355_log_fn fn;
    void *user;
};
std::mutex g_log_lock;            // writers only; readers take a snapshot of the pair under it (calls are rare: not a hot path)
LogSink g_log_sink = {nullptr, nullptr};
LogSink log_sink()
{
    std::lock_guard<std::mutex> g(g_log_lock);
    return g_log_sink;
}
}  // namespace

extern "C" int mi355_set_log_callback(mi355_log_fn fn, void *user)
{
    std::lock_guard<std::mutex> g(g_log_lock);
    g_log_sink = {fn, fn ? user : nullptr};
    return MI355_OK;
}

void mi355_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    const LogSink s = log_sink();
    if (s.fn) s.fn(s.user, MI355_LOG_ERROR, g_err);
}

void mi355_log(const mi355_ctx *ctx, int level, const char *fmt, ...)
{
    if (level < MI355_LOG_WARN && !(ctx && ctx->debug)) return;
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof(line), fmt, ap);
    va_end(ap);
    const LogSink s = log_sink();
    if (s.fn)
        s.fn(s.user, level, line);
    else
        fprintf(stderr, "[mi355] %s\n", line);
}

extern "C" const char *mi355_last_error(void) { return g_err; }

extern "C" const char *mi355_version(void) { return "gr-clenabled_amd 0.1 (gfx950)"; }

extern "C" const char *mi355_strerror(int code)
{
    switch (code) {
    case MI355_OK: return "ok";
    case MI355_ERR_INVALID_ARG: return "invalid argument";
    case MI355_ERR_NO_DEVICE: return "no gfx950 device";
    case MI355_ERR_UNSUPPORTED: return "unsupported configuration";
    case MI355_ERR_HIP: return "HIP runtime error";
    case MI355_ERR_NOMEM: return "out of memory";
    case MI355_ERR_STATE: return "invalid handle state";
    }
    return "unknown error";
}

extern "C" int mi355_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        mi355_set_error("hipGetDeviceCount -> %s", hipGetErrorString(e));
        return MI355_ERR_NO_DEVICE;
    }
    return n;
}

hipError_t mi355_direct_sync(hipStream_t st)
{
    static const int window_us = getenv("MI355_SPIN_US") ? atoi(getenv("MI355_SPIN_US")) : 200;
    if (window_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady) return q;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > window_us) break;
        }
    }
    return hipStreamSynchronize(st);
}

extern "C" int mi355_ctx_create(int ocl_type, int dev_selector, int platform_id, int dev_id, int debug, mi355_ctx **out)
{
    MI355_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (ocl_type == MI355_OCLTYPE_CPU) {
        mi355_set_error("OCLTYPE_CPU requested: this library has no CPU path");
        return MI355_ERR_UNSUPPORTED;
    }
    MI355_REQUIRE(ocl_type == MI355_OCLTYPE_GPU || ocl_type == MI355_OCLTYPE_ACCELERATOR || ocl_type == MI355_OCLTYPE_ANY,
                  "openCLPlatformType must be 1, 2 or 4");
    MI355_REQUIRE(dev_selector == MI355_DEVSEL_FIRST || dev_selector == MI355_DEVSEL_SPECIFIC, "devSelector must be 1 or 2");
    int n = mi355_device_count();
    if (n <= 0) {
        if (n == 0) mi355_set_error("no HIP device visible");
        return MI355_ERR_NO_DEVICE;
    }
    int dev = 0;
    if (dev_selector == MI355_DEVSEL_SPECIFIC) {
        MI355_REQUIRE(platform_id == 0, "platformId must be 0 (single HIP platform)");
        MI355_REQUIRE(dev_id >= 0 && dev_id < n, "devId out of range");
        dev = dev_id;
    }
    hipDeviceProp_t prop;
    MI355_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mi355_set_error("device %d is %s, this library carries gfx950 code only", dev, prop.gcnArchName);
        return MI355_ERR_NO_DEVICE;
    }
    mi355_ctx *c = new (std::nothrow) mi355_ctx();
    if (!c) return MI355_ERR_NOMEM;
    c->device = dev;
    c->debug = debug;
    c->num_cus = prop.multiProcessorCount;
    hipError_t e = hipSetDevice(dev);
    // (The host does NOT switch the device to spin-wait scheduling: that flag is process wide.  Scheduler-sized calls poll
    // their own stream for a bounded time instead, mi355_direct_sync; MI355_SPIN=1 restores the device-wide flag.)
    if (e == hipSuccess && getenv("MI355_SPIN") && atoi(getenv("MI355_SPIN")) != 0) {
        (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
        (void)hipGetLastError();
    }
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&c->stream[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->upload, hipStreamNonBlocking);
    if (e != hipSuccess) {
        mi355_set_error("creating the context's streams on device %d -> %s", dev, hipGetErrorString(e));
        mi355_ctx_destroy(c);
        return MI355_ERR_HIP;
    }
    mi355_log(c, MI355_LOG_INFO, "context on device %d (%s, %d CUs, %.1f GB)", dev, prop.gcnArchName, prop.multiProcessorCount,
              prop.totalGlobalMem / 1e9);
    *out = c;
    return MI355_OK;
}

extern "C" int mi355_ctx_destroy(mi355_ctx *ctx)
{
    if (!ctx) return MI355_OK;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 2; i++)
        if (ctx->stream[i]) {
            (void)hipStreamSynchronize(ctx->stream[i]);
            (void)hipStreamDestroy(ctx->stream[i]);
        }
    if (ctx->upload) {
        (void)hipStreamSynchronize(ctx->upload);
        (void)hipStreamDestroy(ctx->upload);
    }
    delete ctx;
    return MI355_OK;
}

hipError_t mi355_upload(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    std::lock_guard<std::mutex> g(ctx->upload_lock);
    hipError_t e = hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->upload);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->upload);
    return e;
}

hipError_t mi355_fill(mi355_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    std::lock_guard<std::mutex> g(ctx->upload_lock);
    hipError_t e = hipMemsetAsync(dst_dev, value, bytes, ctx->upload);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->upload);
    return e;
}

extern "C" int mi355_ctx_device(const mi355_ctx *ctx) { return ctx ? ctx->device : MI355_ERR_INVALID_ARG; }

extern "C" void *mi355_ctx_stream(mi355_ctx *ctx) { return ctx ? (void *)ctx->stream[0] : nullptr; }

extern "C" int mi355_ctx_synchronize(mi355_ctx *ctx)
{
    MI355_REQUIRE(ctx != nullptr, "ctx is NULL");
    MI355_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < 2; i++) MI355_HIP(hipStreamSynchronize(ctx->stream[i]));
    return MI355_OK;
}

namespace {
// strided block copy: dst[b][r][0..width) = src[b][r][0..width) with independent row pitches and block strides
template <typename V>
__global__ __launch_bounds__(256) void k_pack3d(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src, size_t wv, size_t rows,
                                                size_t nblocks, size_t src_pitch, size_t src_block, size_t dst_pitch, size_t dst_block)
{
    const size_t total = nblocks * rows * wv;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t c = e % wv, rb = e / wv, r = rb % rows, b = rb / rows;
        const V v = __builtin_nontemporal_load((const V *)(src + b * src_block + r * src_pitch) + c);
        __builtin_nontemporal_store(v, (V *)(dst + b * dst_block + r * dst_pitch) + c);
    }
}
}  // namespace

// Device-side strided copy used around the X-engine's all-to-all corner turn (gr-clenabled_amd/shard.py): packs the channel
// slice every peer gets into one contiguous send block, i.e. [t][station][peer][chan slice] -> [peer][t][station][chan slice].
extern "C" int mi355_pack3d_dev(mi355_ctx *ctx, void *dst, const void *src, size_t width_bytes, size_t rows, size_t nblocks, size_t src_pitch,
                                size_t src_block_stride, size_t dst_pitch, size_t dst_block_stride, void *stream)
{
    MI355_REQUIRE(ctx && dst && src, "NULL argument");
    if (width_bytes == 0 || rows == 0 || nblocks == 0) return MI355_OK;
    MI355_HIP(hipSetDevice(ctx->device));
    hipStream_t st = mi355_pick_stream(ctx, stream);
    const uintptr_t all = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | width_bytes | src_pitch | src_block_stride | dst_pitch |
                          dst_block_stride;
    const size_t vec = (all & 15u) == 0 ? 16 : (all & 3u) == 0 ? 4 : 1;
    const size_t wv = width_bytes / vec, total = nblocks * rows * wv;
    size_t blocks = (total + 255) / 256;
    const size_t cap = (size_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * 32;
    if (blocks > cap) blocks = cap;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    if (vec == 16)
        hipLaunchKernelGGL((k_pack3d<v4i_t>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)dst, (const unsigned char *)src, wv, rows,
                           nblocks, src_pitch, src_block_stride, dst_pitch, dst_block_stride);
    else if (vec == 4)
        hipLaunchKernelGGL((k_pack3d<int>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)dst, (const unsigned char *)src, wv, rows,
                           nblocks, src_pitch, src_block_stride, dst_pitch, dst_block_stride);
    else
        hipLaunchKernelGGL((k_pack3d<unsigned char>), dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)dst, (const unsigned char *)src, wv,
                           rows, nblocks, src_pitch, src_block_stride, dst_pitch, dst_block_stride);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_malloc(mi355_ctx *ctx, size_t bytes, void **dptr)
{
    MI355_REQUIRE(ctx && dptr, "NULL argument");
    MI355_HIP(hipSetDevice(ctx->device));
    MI355_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return MI355_OK;
}

extern "C" int mi355_free(mi355_ctx *ctx, void *dptr)
{
    MI355_REQUIRE(ctx != nullptr, "ctx is NULL");
    MI355_HIP(hipSetDevice(ctx->device));
    MI355_HIP(hipFree(dptr));
    return MI355_OK;
}

extern "C" int mi355_memcpy_h2d(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    MI355_REQUIRE(ctx != nullptr, "ctx is NULL");
    MI355_HIP(hipSetDevice(ctx->device));
    MI355_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream[0]));
    MI355_HIP(hipStreamSynchronize(ctx->stream[0]));
    return MI355_OK;
}

extern "C" int mi355_memcpy_d2h(mi355_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    MI355_REQUIRE(ctx != nullptr, "ctx is NULL");
    MI355_HIP(hipSetDevice(ctx->device));
    MI355_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream[0]));
    MI355_HIP(hipStreamSynchronize(ctx->stream[0]));
    return MI355_OK;
}

// ---------------------------------------------------------------------------
int HostPipe::init(mi355_ctx *c)
{
    ctx = c;
    MI355_HIP(hipSetDevice(c->device));
    for (int s = 0; s < kSlots; s++) MI355_HIP(hipEventCreateWithFlags(&done[s], hipEventDisableTiming));
    return MI355_OK;
}

int HostPipe::ensure(int nin, const size_t *in_bytes, size_t out_bytes, int nslots)
{
    MI355_HIP(hipSetDevice(ctx->device));
    if (nslots < 1) nslots = 1;
    if (nslots > kSlots) nslots = kSlots;
    bool grow = nslots > slots_ready || out_bytes > cap_out;
    for (int i = 0; i < nin; i++) grow = grow || in_bytes[i] > cap_in[i];
    if (!grow) return MI355_OK;
    // (re)allocate every slot in use at the larger of the old and new sizes
    size_t want_in[MAXIN] = {cap_in[0], cap_in[1]}, want_out = cap_out > out_bytes ? cap_out : out_bytes;
    for (int i = 0; i < nin; i++) if (in_bytes[i] > want_in[i]) want_in[i] = in_bytes[i];
    const int upto = nslots > slots_ready ? nslots : slots_ready;
    const int rc = [&]() -> int {
        for (int s = 0; s < upto; s++) {
            for (int i = 0; i < MAXIN; i++) {
                if (want_in[i] == 0 || (want_in[i] == cap_in[i] && s < slots_ready)) continue;
                if (h_in[s][i]) MI355_HIP(hipHostFree(h_in[s][i]));
                h_in[s][i] = nullptr;
                if (d_in[s][i]) MI355_HIP(hipFree(d_in[s][i]));
                d_in[s][i] = nullptr;
                MI355_HIP(hipHostMalloc(&h_in[s][i], want_in[i], hipHostMallocDefault));
                MI355_HIP(hipMalloc(&d_in[s][i], want_in[i]));
            }
            if (want_out && !(want_out == cap_out && s < slots_ready)) {
                if (h_out[s]) MI355_HIP(hipHostFree(h_out[s]));
                h_out[s] = nullptr;
                if (d_out[s]) MI355_HIP(hipFree(d_out[s]));
                d_out[s] = nullptr;
                MI355_HIP(hipHostMalloc(&h_out[s], want_out, hipHostMallocDefault));
                MI355_HIP(hipMalloc(&d_out[s], want_out));
            }
        }
        return MI355_OK;
    }();
    if (rc != MI355_OK) {
        // a slot may be half replaced: drop all staging (the old capacities would vouch for buffers that no longer exist) and leave no
        // sticky HIP error behind; the next call allocates from scratch
        (void)hipGetLastError();
        for (int s = 0; s < kSlots; s++) {
            for (int i = 0; i < MAXIN; i++) {
                if (h_in[s][i]) (void)hipHostFree(h_in[s][i]);
                if (d_in[s][i]) (void)hipFree(d_in[s][i]);
                h_in[s][i] = d_in[s][i] = nullptr;
            }
            if (h_out[s]) (void)hipHostFree(h_out[s]);
            if (d_out[s]) (void)hipFree(d_out[s]);
            h_out[s] = d_out[s] = nullptr;
        }
        cap_in[0] = cap_in[1] = cap_out = 0;
        slots_ready = 0;
        (void)hipGetLastError();
        return rc;
    }
    for (int i = 0; i < MAXIN; i++) cap_in[i] = want_in[i];
    cap_out = want_out;
    slots_ready = upto;
    return MI355_OK;
}

void HostPipe::release()
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int s = 0; s < kSlots; s++) {
        for (int i = 0; i < MAXIN; i++) {
            if (h_in[s][i]) (void)hipHostFree(h_in[s][i]);
            if (d_in[s][i]) (void)hipFree(d_in[s][i]);
            h_in[s][i] = d_in[s][i] = nullptr;
        }
        if (h_out[s]) (void)hipHostFree(h_out[s]);
        if (d_out[s]) (void)hipFree(d_out[s]);
        h_out[s] = d_out[s] = nullptr;
        if (done[s]) (void)hipEventDestroy(done[s]);
        done[s] = nullptr;
    }
    cap_in[0] = cap_in[1] = cap_out = 0;
    slots_ready = 0;
}

// ---------------------------------------------------------------------------
// small persistent helper pool for host-side data movement (see common.h): staging copies and the X-engine frame gather
// ---------------------------------------------------------------------------
#include <immintrin.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <thread>

namespace {
struct JobPool {
    std::mutex use;                  // one job at a time
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::vector<std::thread> workers;
    void (*fn)(void *, int, int) = nullptr;
    void *arg = nullptr;
    unsigned long long job = 0;      // generation counter
    int pending = 0;
    bool stop = false;

    explicit JobPool(int n)
    {
        for (int i = 0; i < n; i++)
            workers.emplace_back([this, i] {
                unsigned long long seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return stop || job != seen; });
                    if (stop) return;
                    seen = job;
                    void (*f)(void *, int, int) = fn;
                    void *a = arg;
                    const int parts = (int)workers.size() + 1;
                    lk.unlock();
                    f(a, i + 1, parts);  // part 0 is the caller's
                    lk.lock();
                    if (--pending == 0) done_cv.notify_one();
                }
            });
    }
    ~JobPool()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
    void run(void (*f)(void *, int, int), void *a)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            fn = f; arg = a;
            pending = (int)workers.size();
            job++;
        }
        cv.notify_all();
        f(a, 0, (int)workers.size() + 1);
        std::unique_lock<std::mutex> lk(m);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
};
JobPool *job_pool()
{
    static JobPool *pool = [] {
        const char *e = getenv("MI355_COPY_THREADS");
        const int n = e ? atoi(e) : 7;
        return n > 0 ? new JobPool(n > 16 ? 16 : n) : (JobPool *)nullptr;  // lives until process exit
    }();
    return pool;
}
// Large copies between the caller's (pageable) buffers and the pinned staging buffers.  Streaming stores: a regular memcpy of
// a 1-2 MiB piece reads the destination lines before overwriting them (read-for-ownership), i.e. moves three bytes per byte
// copied; the staging buffer is read next by the DMA engine, not by this core, so there is nothing to keep in the cache.
__attribute__((target("avx2"))) void copy_stream_avx2(char *d, const char *s, size_t n)
{
    while ((reinterpret_cast<uintptr_t>(d) & 31u) && n) { *d++ = *s++; n--; }
    size_t blocks = n / 128;
    for (; blocks; blocks--, d += 128, s += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)s), b = _mm256_loadu_si256((const __m256i *)(s + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
        _mm256_stream_si256((__m256i *)d, a);
        _mm256_stream_si256((__m256i *)(d + 32), b);
        _mm256_stream_si256((__m256i *)(d + 64), c);
        _mm256_stream_si256((__m256i *)(d + 96), e);
    }
    _mm_sfence();
    if (n & 127) memcpy(d, s, n & 127);
}
void copy_piece(char *d, const char *s, size_t n)
{
    static const bool stream = __builtin_cpu_supports("avx2") && !(getenv("MI355_COPY_STREAM") && atoi(getenv("MI355_COPY_STREAM")) == 0);
    if (stream && n >= (1u << 20)) copy_stream_avx2(d, s, n);  // (scheduler-sized calls keep the cached memcpy: their data is re-read at once)
    else memcpy(d, s, n);
}
// up to two copies as ONE job: the bytes of both are cut into `parts` runs, so a staging slot's copy-out (previous result) and
// copy-in (next input) proceed side by side instead of one after the other on the calling thread's time line
struct CopyJob { char *dst[2]; const char *src[2]; size_t bytes[2]; };
void copy_part(void *a, int part, int parts)
{
    const CopyJob *j = (const CopyJob *)a;
    const size_t total = j->bytes[0] + j->bytes[1];
    const size_t per = ((total + parts - 1) / parts + 4095) & ~(size_t)4095;
    size_t lo = (size_t)part * per, hi = lo + per < total ? lo + per : total;
    if (lo >= total) return;
    if (lo < j->bytes[0]) {
        const size_t e = hi < j->bytes[0] ? hi : j->bytes[0];
        copy_piece(j->dst[0] + lo, j->src[0] + lo, e - lo);
        lo = e;
    }
    if (lo < hi) copy_piece(j->dst[1] + (lo - j->bytes[0]), j->src[1] + (lo - j->bytes[0]), hi - lo);
}
}  // namespace

bool mi355_parallel(void (*fn)(void *, int, int), void *arg)
{
    JobPool *p = job_pool();
    if (p && p->use.try_lock()) {
        p->run(fn, arg);
        p->use.unlock();
        return true;
    }
    return false;
}

void mi355_copy2(void *dst0, const void *src0, size_t bytes0, void *dst1, const void *src1, size_t bytes1)
{
    CopyJob j = {{(char *)dst0, (char *)dst1}, {(const char *)src0, (const char *)src1}, {bytes0, bytes1}};
    if (bytes0 + bytes1 < (2u << 20) || !mi355_parallel(copy_part, &j)) {
        if (bytes0) copy_piece((char *)dst0, (const char *)src0, bytes0);
        if (bytes1) copy_piece((char *)dst1, (const char *)src1, bytes1);
    }
}

void mi355_copy(void *dst, const void *src, size_t bytes) { mi355_copy2(dst, src, bytes, nullptr, nullptr, 0); }
