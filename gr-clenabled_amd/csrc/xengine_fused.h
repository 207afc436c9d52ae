// Fused int8 (IChar) X-engine path: corner turn in LDS + MFMA correlation in one pass (xengine_fused.hip).
#pragma once
#include <cstddef>

#include "common.h"

struct XeFusedPlan {
    bool ok = false;     // geometry supported by the fused kernels
    int npol = 1, ntt = 0;
    int units = 0;       // 32-byte column slices of an input row (rows rounded up to whole 128-byte lines)
    int row_stride = 0;  // bytes of a row that exist = bytes between rows
    int tsplit = 1;      // time ranges (partial sums are combined by the reduce kernel, or by the fused kernel's own tail, when > 1)
    int cus = 256;       // compute units of the device (the in-launch reduction needs every workgroup resident at once)
    size_t part_bytes = 0;  // workspace needed: int32 partial sums + the reduction's counters (0 when tsplit == 1)
    size_t flag_offset = 0; // where the counters start
    size_t part_per_window = 0;  // bytes of partial sums of one integration window
};

// stations N, channels F (F * npol * 2 bytes per (t, station) row), integration T; nint integration windows per launch (the time
// split shrinks as nint grows: with nint * slices >= CUs every workgroup runs a whole window and there are no partial sums at all)
XeFusedPlan mi355_xe_fused_plan(int N, int F, int Fout, int npol, int T, int num_cus, int nint = 1);

// in: [t][station][chan][pol]{I,Q} int8 (16-byte aligned), out: [chan][baseline][pol^2] complex float.
// part: workspace of plan.part_bytes, zeroed once (unused when tsplit == 1; launches sharing a workspace must be stream-ordered).  kd: 1/127.
// stations_per_group (0 or N: the reference layout): the input is [group][t][station in group][chan][pol]{I,Q}, the blocks an
// all-to-all corner turn delivers (gr-clenabled_amd/shard.py) -- read in place, no re-layout pass.
// nint > 1: `in` holds nint windows, [window][t][station].. (reference layout) or [group][window][t][station in group].. (group-major: what
// ONE all-to-all of nint windows delivers); `out` holds nint matrices back to back; the plan must have been made for the same nint.
// epoch: the workspace's launch counter for the in-launch reduction (default where it applies; MI355_XE_INKERNEL_REDUCE=0 disables it); it advances only
// with launches that use the counters, so a process that mixes both forms of the reduction on one workspace stays consistent.
// NULL (or nint > 1): the reduce kernel is used.
int mi355_xe_fused_launch(const XeFusedPlan &p, const void *in, void *out, void *part, int N, int F, int Fout, int T, double kd,
                          int accumulate, hipStream_t st, int stations_per_group = 0, unsigned *epoch = nullptr, int nint = 1);
