// clXEngine over several devices of ONE process (the reference runs one X-engine on one device and selects it per block with
// devId, lib/GRCLBase.cpp:115-134 -- a GNU Radio flowgraph is one process, so a sharded block must drive its devices from there).
//
// Rank r (= device_ids[r]) ingests the frames of its antenna group -- [window][t][station in group][chan][pol]{I,Q}, the reference's frame
// layout (lib/clXEngine_impl.cc:987-1061) with num_inputs / W stations -- and ends up with the matrices of its channel slab: channels
// [r F/W, (r+1) F/W) of [chan][baseline][pol^2] (:786-808), which is a contiguous piece of the reference's output.  In between, the FX
// correlator's corner turn: one strided device copy packs rank r's frames into W blocks, one per destination (mi355_pack3d_dev), W
// contiguous peer copies (hipMemcpyPeerAsync: xGMI between devices, a plain device copy when two ranks share a device) put block d into
// rank d's receive buffer [group][window][t][station in group][chan slab], and the fused kernel reads that buffer IN PLACE
// (mi355_xengine_xcorrelate_n_dev with stations_per_group), all windows of the exchange in one launch.  Two slots of send / receive buffers,
// an exchange stream and a compute stream per rank: exchange k+1 runs under correlation k.  Nothing here computes: the arithmetic is
// xengine_fused.hip's.  This is gr-clenabled_amd/shard.py (one process per device, RCCL all-to-all) moved below the C ABI for callers that
// cannot be one process per device.
#include <vector>

#include "common.h"
#include "xengine_fused.h"

struct mi355_xengine_shard {
    int world = 0, npol = 1, N = 0, F = 0, T = 0, windows = 1, Ng = 0, Fw = 0;
    size_t row = 0, wrow = 0;      // bytes of a (t, station) row of the full / the slab's channels
    size_t block = 0;              // bytes one rank sends to one rank per exchange = windows * T * Ng * wrow
    size_t frames_bytes = 0, slab_items = 0;  // per rank: input bytes per exchange, output items per window
    struct Rank {
        int dev = 0;
        mi355_ctx *ctx = nullptr;
        mi355_xengine *xe = nullptr;
        hipStream_t xs = nullptr, cs = nullptr;      // exchange / compute
        unsigned char *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};
        hipEvent_t mark = nullptr;                    // scratch: "everything enqueued on the compute stream so far"
        hipEvent_t sent[2] = {nullptr, nullptr};      // this rank's blocks of slot s have landed everywhere
        hipEvent_t corr_done[2] = {nullptr, nullptr}; // this rank's correlation of slot s has read its receive buffer
        bool corr_used[2] = {false, false};
        // host path: device copies of the rank's frames / matrices
        unsigned char *d_frames = nullptr;
        void *d_out = nullptr;
        // streaming host path (acquire / submit_acquired / wait): per host slot the rank's frames and matrices on the device, "slot's matrices are in
        // the pinned result buffer" on the compute stream
        unsigned char *hd_frames[2] = {nullptr, nullptr};
        void *hd_out[2] = {nullptr, nullptr};
        hipEvent_t h_done[2] = {nullptr, nullptr};
    };
    std::vector<Rank> rk;
    int next_slot = 0;
    std::mutex lock;
    std::mutex host_call;  // the synchronous host form, one call at a time per handle (its staging buffers are the handle's)
    // streaming host path: two slots of PINNED host memory (hipHostMallocPortable: every rank's device reads its antenna group out of the same
    // buffer over its own link, truly asynchronously) -- the reference pins its frame buffers too (lib/clXEngine_impl.cc:325-362)
    struct HostSlot {
        void *h_in = nullptr, *h_out = nullptr;
        bool busy = false;
    } hs[2];
    int host_next_submit = 0, host_next_wait = 0, host_pending = 0;
    bool host_acquired = false;
};

namespace {
void shard_free(mi355_xengine_shard *h)
{
    for (auto &r : h->rk) {
        if (r.ctx) (void)hipSetDevice(r.dev);
        for (int s = 0; s < 2; s++) {
            if (r.send[s]) (void)hipFree(r.send[s]);
            if (r.recv[s]) (void)hipFree(r.recv[s]);
            if (r.sent[s]) (void)hipEventDestroy(r.sent[s]);
            if (r.corr_done[s]) (void)hipEventDestroy(r.corr_done[s]);
        }
        if (r.mark) (void)hipEventDestroy(r.mark);
        if (r.d_frames) (void)hipFree(r.d_frames);
        if (r.d_out) (void)hipFree(r.d_out);
        for (int s = 0; s < 2; s++) {
            if (r.hd_frames[s]) (void)hipFree(r.hd_frames[s]);
            if (r.hd_out[s]) (void)hipFree(r.hd_out[s]);
            if (r.h_done[s]) (void)hipEventDestroy(r.h_done[s]);
        }
        if (r.xs) (void)hipStreamDestroy(r.xs);
        if (r.cs) (void)hipStreamDestroy(r.cs);
        if (r.xe) mi355_xengine_destroy(r.xe);
        if (r.ctx) mi355_ctx_destroy(r.ctx);
    }
    for (auto &sl : h->hs) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
    }
    delete h;
}
}  // namespace

extern "C" int mi355_xengine_shard_destroy(mi355_xengine_shard *h)
{
    if (h) shard_free(h);
    return MI355_OK;
}

extern "C" int mi355_xengine_shard_create(int world, const int *device_ids, int npol, int num_inputs, int num_channels, int integration, int windows,
                                          mi355_xengine_shard **out)
{
    MI355_REQUIRE(out && device_ids, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(world >= 1 && world <= 64, "world must be 1 .. 64");
    MI355_REQUIRE(npol == 1 || npol == 2, "polarization must be 1 or 2");
    MI355_REQUIRE(windows >= 1, "windows must be >= 1");
    MI355_REQUIRE(num_inputs % world == 0 && num_channels % world == 0, "the ranks must divide the inputs (antenna groups) and the channels (slabs)");
    // the exchanged blocks are read IN PLACE: the fused path (at most 64 rows) or, round 6, the whole-line kernel for 64 stations x two polarisations
    // (slabs of whole 128-byte lines = 32 channels; enough windows per exchange to fill the device, or the correlation returns MI355_ERR_UNSUPPORTED)
    MI355_REQUIRE(num_inputs * npol <= 64 || (num_inputs == 64 && npol == 2 && (num_channels / world) % 32 == 0),
                  "the sharded X-engine reads the exchanged blocks in place: IChar, at most 64 rows, or 64 stations x 2 polarisations with slabs of whole 32-channel lines");
    const size_t wrow = (size_t)(num_channels / world) * npol * 2;
    MI355_REQUIRE(wrow % 16 == 0, "a rank's channel slab must be whole 16-byte pieces per (t, station) row");
    mi355_xengine_shard *h = new (std::nothrow) mi355_xengine_shard();
    if (!h) return MI355_ERR_NOMEM;
    h->world = world; h->npol = npol; h->N = num_inputs; h->F = num_channels; h->T = integration; h->windows = windows;
    h->Ng = num_inputs / world; h->Fw = num_channels / world;
    h->row = (size_t)num_channels * npol * 2; h->wrow = wrow;
    h->block = (size_t)windows * integration * h->Ng * wrow;
    h->frames_bytes = (size_t)windows * integration * h->Ng * h->row;
    h->rk.resize((size_t)world);
    int rc = MI355_OK;
    for (int r = 0; r < world && rc == MI355_OK; r++) {
        auto &k = h->rk[(size_t)r];
        k.dev = device_ids[r];
        rc = mi355_ctx_create(MI355_OCLTYPE_GPU, MI355_DEVSEL_SPECIFIC, 0, k.dev, 0, &k.ctx);
        if (rc != MI355_OK) break;
        rc = mi355_xengine_create(k.ctx, MI355_DTYPE_BYTE, npol, num_inputs, h->Fw, integration, &k.xe);
        if (rc != MI355_OK) break;
        h->slab_items = mi355_xengine_output_items(k.xe);
        hipError_t e = hipSetDevice(k.dev);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&k.xs, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&k.cs, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&k.mark, hipEventDisableTiming);
        for (int s = 0; s < 2 && e == hipSuccess; s++) {
            e = hipMalloc((void **)&k.send[s], h->block * world);
            if (e == hipSuccess) e = hipMalloc((void **)&k.recv[s], h->block * world);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&k.sent[s], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&k.corr_done[s], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            mi355_set_error("sharded X-engine, rank %d on device %d: %s", r, k.dev, hipGetErrorString(e));
            rc = e == hipErrorOutOfMemory ? MI355_ERR_NOMEM : MI355_ERR_HIP;
        }
    }
    // peer access between every pair of distinct devices (a copy between two ranks of one device needs none)
    for (int a = 0; a < world && rc == MI355_OK; a++)
        for (int b = 0; b < world && rc == MI355_OK; b++) {
            const int da = h->rk[(size_t)a].dev, db = h->rk[(size_t)b].dev;
            if (da == db) continue;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, da, db);
            if (e == hipSuccess && can) {
                e = hipSetDevice(da);
                if (e == hipSuccess) e = hipDeviceEnablePeerAccess(db, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { e = hipSuccess; (void)hipGetLastError(); }
            }
            if (e != hipSuccess) {  // (without peer access hipMemcpyPeerAsync stages through the host: slower, still correct)
                (void)hipGetLastError();
            }
        }
    // 64 stations x two polarisations: only the whole-line kernel reads the receive buffer in place, and it needs enough (window, line, pair group) units
    // per rank to fill the device -- said here, at create, not by the first correlation
    if (rc == MI355_OK && num_inputs * npol > 64) {
        int need = windows;
        while (need <= 4096 && !mi355_xe_lines_ok(num_inputs, h->Fw, h->Fw, npol, integration, h->Ng, 0, need, h->rk[0].ctx->num_cus)) need++;
        if (need != windows) {
            if (need <= 4096)
                mi355_set_error("sharded X-engine, 64 inputs x 2 polarisations over %d ranks (%d channels per rank): needs at least %d windows per exchange, %d given",
                                world, h->Fw, need, windows);
            else
                mi355_set_error("sharded X-engine, 64 inputs x 2 polarisations over %d ranks: %d channels per rank are not whole 32-channel lines", world, h->Fw);
            rc = MI355_ERR_UNSUPPORTED;
        }
    }
    if (rc != MI355_OK) { shard_free(h); return rc; }
    *out = h;
    return MI355_OK;
}

extern "C" int mi355_xengine_shard_world(const mi355_xengine_shard *h) { return h ? h->world : MI355_ERR_INVALID_ARG; }
extern "C" int mi355_xengine_shard_device(const mi355_xengine_shard *h, int rank)
{
    return (h && rank >= 0 && rank < h->world) ? h->rk[(size_t)rank].dev : MI355_ERR_INVALID_ARG;
}
extern "C" size_t mi355_xengine_shard_frames_bytes(const mi355_xengine_shard *h) { return h ? h->frames_bytes : 0; }
extern "C" size_t mi355_xengine_shard_slab_items(const mi355_xengine_shard *h) { return h ? h->slab_items : 0; }
extern "C" void *mi355_xengine_shard_stream(mi355_xengine_shard *h, int rank)
{
    return (h && rank >= 0 && rank < h->world) ? (void *)h->rk[(size_t)rank].cs : nullptr;
}

// The rank's compute stream waits for everything enqueued so far on `stream` (a stream of the rank's device: the producer of the next
// frames_dev[rank], or the consumer that must finish before an out_dev[rank] is overwritten, runs there).
extern "C" int mi355_xengine_shard_wait_stream(mi355_xengine_shard *h, int rank, void *stream)
{
    MI355_REQUIRE(h && rank >= 0 && rank < h->world, "bad rank");
    std::lock_guard<std::mutex> g(h->lock);
    auto &k = h->rk[(size_t)rank];
    MI355_HIP(hipSetDevice(k.dev));
    MI355_HIP(hipEventRecord(k.mark, (hipStream_t)stream));
    MI355_HIP(hipStreamWaitEvent(k.cs, k.mark, 0));
    return MI355_OK;
}

// One exchange + correlation of `windows` integration windows, enqueued only.  frames_dev[r]: rank r's antenna-group frames on device r
// (frames_bytes() long, 16-byte aligned); they are read by the packing copy on the rank's exchange stream, which first waits for
// everything enqueued so far on the rank's compute stream (mi355_xengine_shard_stream: enqueue the producer of the frames there); they must
// stay untouched until the next submit / synchronize on this handle returns.  out_dev[r]: windows x slab_items() complex floats on device r.
static int shard_submit_locked(mi355_xengine_shard *h, const void *const *frames_dev, void *const *out_dev, int accumulate);
extern "C" int mi355_xengine_shard_submit_dev(mi355_xengine_shard *h, const void *const *frames_dev, void *const *out_dev, int accumulate)
{
    MI355_REQUIRE(h && frames_dev && out_dev, "NULL argument");
    std::lock_guard<std::mutex> g(h->lock);
    return shard_submit_locked(h, frames_dev, out_dev, accumulate);
}
static int shard_submit_locked(mi355_xengine_shard *h, const void *const *frames_dev, void *const *out_dev, int accumulate)
{
    const int W = h->world, s = h->next_slot;
    for (int r = 0; r < W; r++) MI355_REQUIRE(frames_dev[r] && out_dev[r] && (reinterpret_cast<uintptr_t>(frames_dev[r]) & 15u) == 0, "NULL or misaligned rank buffer");
    const size_t rows = (size_t)h->windows * h->T * h->Ng;
    // ---- exchange: pack, then one contiguous copy per destination
    for (int r = 0; r < W; r++) {
        auto &k = h->rk[(size_t)r];
        MI355_HIP(hipSetDevice(k.dev));
        // the frames' producer ran on the compute stream (or finished before the call); the send slot was last read by copies on this stream
        MI355_HIP(hipEventRecord(k.mark, k.cs));
        MI355_HIP(hipStreamWaitEvent(k.xs, k.mark, 0));
        int rc = mi355_pack3d_dev(k.ctx, k.send[s], frames_dev[r], h->wrow, rows, (size_t)W, h->row, h->wrow, h->wrow, h->block, (void *)k.xs);
        if (rc != MI355_OK) return rc;
        for (int d = 0; d < W; d++) {
            auto &kd = h->rk[(size_t)d];
            if (kd.corr_used[s]) MI355_HIP(hipStreamWaitEvent(k.xs, kd.corr_done[s], 0));  // rank d's last correlation on this slot has read its buffer
            MI355_HIP(hipMemcpyPeerAsync(kd.recv[s] + (size_t)r * h->block, kd.dev, k.send[s] + (size_t)d * h->block, k.dev, h->block, k.xs));
        }
        MI355_HIP(hipEventRecord(k.sent[s], k.xs));
    }
    // ---- correlation: every rank waits for all W senders, then ONE launch over the windows, the receive buffer read in place
    for (int d = 0; d < W; d++) {
        auto &kd = h->rk[(size_t)d];
        MI355_HIP(hipSetDevice(kd.dev));
        for (int r = 0; r < W; r++) MI355_HIP(hipStreamWaitEvent(kd.cs, h->rk[(size_t)r].sent[s], 0));
        int rc = mi355_xengine_xcorrelate_n_dev(kd.xe, h->windows, kd.recv[s], out_dev[d], accumulate, W > 1 ? h->Ng : 0, (void *)kd.cs);
        if (rc != MI355_OK) return rc;
        MI355_HIP(hipEventRecord(kd.corr_done[s], kd.cs));
        kd.corr_used[s] = true;
    }
    h->next_slot ^= 1;
    return MI355_OK;
}

extern "C" int mi355_xengine_shard_synchronize(mi355_xengine_shard *h)
{
    MI355_REQUIRE(h != nullptr, "NULL argument");
    std::lock_guard<std::mutex> g(h->lock);
    for (auto &k : h->rk) {
        MI355_HIP(hipSetDevice(k.dev));
        MI355_HIP(hipStreamSynchronize(k.xs));
        MI355_HIP(hipStreamSynchronize(k.cs));
    }
    return MI355_OK;
}

// Host form = the reference's xcorrelate(char *input_matrix, XComplex *cross_correlation) (lib/clXEngine_impl.h:179-201) over W devices:
// in_host = `windows` integration windows in the reference's frame layout [window][t][station][chan][pol]{I,Q}; every rank takes its antenna
// group over its own host link (one 2-D copy per rank: the group's stations of a time step are contiguous), the devices exchange and
// correlate, every rank's slab goes back into its place of out_host = [window][chan][baseline][pol^2].  accumulate: out += (pipeline
// integration; the previous matrices are uploaded first).  Returns when out_host is complete.
extern "C" int mi355_xengine_shard_xcorrelate(mi355_xengine_shard *h, const void *in_host, void *out_host, int accumulate)
{
    MI355_REQUIRE(h && in_host && out_host, "NULL argument");
    std::lock_guard<std::mutex> hg(h->host_call);  // (one host call at a time per handle; submit_dev below takes the handle's lock itself)
    const int W = h->world;
    const size_t grp = (size_t)h->Ng * h->row, step = (size_t)h->N * h->row, steps = (size_t)h->windows * h->T;
    const size_t slab_bytes = h->slab_items * 8, full_bytes = slab_bytes * W;
    std::vector<const void *> fr((size_t)W);
    std::vector<void *> ou((size_t)W);
    for (int r = 0; r < W; r++) {
        auto &k = h->rk[(size_t)r];
        MI355_HIP(hipSetDevice(k.dev));
        if (!k.d_frames) MI355_HIP(hipMalloc((void **)&k.d_frames, h->frames_bytes));
        if (!k.d_out) MI355_HIP(hipMalloc(&k.d_out, slab_bytes * h->windows));
        MI355_HIP(hipMemcpy2DAsync(k.d_frames, grp, (const char *)in_host + (size_t)r * grp, step, grp, steps, hipMemcpyHostToDevice, k.cs));
        if (accumulate)
            MI355_HIP(hipMemcpy2DAsync(k.d_out, slab_bytes, (const char *)out_host + (size_t)r * slab_bytes, full_bytes, slab_bytes, (size_t)h->windows,
                                       hipMemcpyHostToDevice, k.cs));
        fr[(size_t)r] = k.d_frames;
        ou[(size_t)r] = k.d_out;
    }
    int rc = mi355_xengine_shard_submit_dev(h, fr.data(), ou.data(), accumulate);
    if (rc != MI355_OK) return rc;
    for (int r = 0; r < W; r++) {
        auto &k = h->rk[(size_t)r];
        MI355_HIP(hipSetDevice(k.dev));
        MI355_HIP(hipMemcpy2DAsync((char *)out_host + (size_t)r * slab_bytes, full_bytes, k.d_out, slab_bytes, slab_bytes, (size_t)h->windows,
                                   hipMemcpyDeviceToHost, k.cs));
    }
    return mi355_xengine_shard_synchronize(h);
}


// ---- Streaming host path: the sharded counterpart of mi355_xengine_acquire / submit_acquired / wait (and of the reference's pinned double buffers +
// worker thread, lib/clXEngine_impl.cc:325-362, :1234-1299).  acquire() hands out a PINNED buffer for `windows` integration windows in the reference's
// frame layout [window][t][station][chan][pol]{I,Q}; the block gathers its frames straight into it; submit_acquired() enqueues, per rank and on the
// rank's own compute stream, the 2-D copy of its antenna group out of that buffer (asynchronous -- pinned memory -- so the W ranks' uploads run at
// once over W host links), the exchange, the correlation and the copy of the rank's slab into a pinned result buffer, and returns; wait() blocks for
// the OLDEST submitted exchange and writes its `windows` matrices [window][chan][baseline][pol^2].  At most two exchanges are in flight: the gather
// of exchange k+1 (the caller's) and its uploads run under the devices' work on exchange k.
extern "C" int mi355_xengine_shard_windows(const mi355_xengine_shard *h) { return h ? h->windows : MI355_ERR_INVALID_ARG; }
extern "C" size_t mi355_xengine_shard_input_bytes(const mi355_xengine_shard *h)
{
    return h ? (size_t)h->windows * h->T * h->N * h->row : 0;
}
extern "C" int mi355_xengine_shard_pending(const mi355_xengine_shard *h) { return h ? h->host_pending : MI355_ERR_INVALID_ARG; }

extern "C" int mi355_xengine_shard_acquire(mi355_xengine_shard *h, void **frame_buffer)
{
    MI355_REQUIRE(h && frame_buffer, "NULL argument");
    std::lock_guard<std::mutex> g(h->lock);
    if (h->host_pending >= 2) {
        mi355_set_error("two exchanges in flight: wait() for the oldest first");
        return MI355_ERR_STATE;
    }
    auto &sl = h->hs[h->host_next_submit];
    if (!sl.h_in) {
        const size_t in_bytes = (size_t)h->windows * h->T * h->N * h->row, out_bytes = (size_t)h->windows * h->slab_items * 8 * h->world;
        MI355_HIP(hipSetDevice(h->rk[0].dev));
        hipError_t e = hipHostMalloc(&sl.h_in, in_bytes, hipHostMallocPortable);
        if (e == hipSuccess) e = hipHostMalloc(&sl.h_out, out_bytes, hipHostMallocPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (sl.h_in) (void)hipHostFree(sl.h_in);
            sl.h_in = sl.h_out = nullptr;
            mi355_set_error("cannot pin %zu + %zu bytes of host memory for the sharded X-engine: %s", in_bytes, out_bytes, hipGetErrorString(e));
            return MI355_ERR_NOMEM;
        }
    }
    h->host_acquired = true;
    *frame_buffer = sl.h_in;
    return MI355_OK;
}

extern "C" int mi355_xengine_shard_submit_acquired(mi355_xengine_shard *h)
{
    MI355_REQUIRE(h != nullptr, "NULL argument");
    std::lock_guard<std::mutex> g(h->lock);
    if (!h->host_acquired) {
        mi355_set_error("submit_acquired without acquire");
        return MI355_ERR_STATE;
    }
    const int W = h->world, s = h->host_next_submit;
    auto &sl = h->hs[s];
    const size_t grp = (size_t)h->Ng * h->row, step = (size_t)h->N * h->row, steps = (size_t)h->windows * h->T;
    const size_t slab_bytes = h->slab_items * 8, full_bytes = slab_bytes * W;
    std::vector<const void *> fr((size_t)W);
    std::vector<void *> ou((size_t)W);
    for (int r = 0; r < W; r++) {
        auto &k = h->rk[(size_t)r];
        MI355_HIP(hipSetDevice(k.dev));
        if (!k.hd_frames[s]) MI355_HIP(hipMalloc((void **)&k.hd_frames[s], h->frames_bytes));
        if (!k.hd_out[s]) MI355_HIP(hipMalloc(&k.hd_out[s], slab_bytes * h->windows));
        if (!k.h_done[s]) MI355_HIP(hipEventCreateWithFlags(&k.h_done[s], hipEventDisableTiming));
        // (the device buffer's last reader, the packing copy two exchanges ago, is ordered in front of this copy through that exchange's correlation,
        // which waited for every rank's `sent` event and ran on this stream)
        MI355_HIP(hipMemcpy2DAsync(k.hd_frames[s], grp, (const char *)sl.h_in + (size_t)r * grp, step, grp, steps, hipMemcpyHostToDevice, k.cs));
        fr[(size_t)r] = k.hd_frames[s];
        ou[(size_t)r] = k.hd_out[s];
    }
    const int rc = shard_submit_locked(h, fr.data(), ou.data(), 0);
    if (rc != MI355_OK) return rc;
    for (int r = 0; r < W; r++) {
        auto &k = h->rk[(size_t)r];
        MI355_HIP(hipSetDevice(k.dev));
        MI355_HIP(hipMemcpy2DAsync((char *)sl.h_out + (size_t)r * slab_bytes, full_bytes, k.hd_out[s], slab_bytes, slab_bytes, (size_t)h->windows,
                                   hipMemcpyDeviceToHost, k.cs));
        MI355_HIP(hipEventRecord(k.h_done[s], k.cs));
    }
    sl.busy = true;
    h->host_acquired = false;
    h->host_next_submit ^= 1;
    h->host_pending++;
    return MI355_OK;
}

extern "C" int mi355_xengine_shard_wait(mi355_xengine_shard *h, void *out_host)
{
    MI355_REQUIRE(h && out_host, "NULL argument");
    int s;
    {
        std::lock_guard<std::mutex> g(h->lock);
        if (h->host_pending == 0) {
            mi355_set_error("wait without a submitted exchange");
            return MI355_ERR_STATE;
        }
        s = h->host_next_wait;
    }
    // (not under the lock: another thread may gather into and submit the other slot meanwhile; waits are single-consumer)
    for (auto &k : h->rk) {
        MI355_HIP(hipSetDevice(k.dev));
        MI355_HIP(hipEventSynchronize(k.h_done[s]));
    }
    mi355_copy(out_host, h->hs[s].h_out, (size_t)h->windows * h->slab_items * 8 * h->world);
    std::lock_guard<std::mutex> g(h->lock);
    h->hs[s].busy = false;
    h->host_next_wait ^= 1;
    h->host_pending--;
    return MI355_OK;
}
