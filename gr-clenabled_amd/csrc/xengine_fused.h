// Fused int8 (IChar) X-engine path: corner turn in LDS + MFMA correlation in one pass (xengine_fused.hip).
#pragma once
#include <cstddef>

#include "common.h"

// what the launch functions of this thread did last (mi355_xengine copies it into the handle under its lock: mi355_xengine_last_route)
extern thread_local mi355_xe_route mi355_xe_route_tls;
void mi355_xe_route_set(const char *kernel, int windows, int workgroups, int units_per_workgroup, int tsplit, int in_launch_reduce, int touches, int pace);

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
// epoch: the workspace's launch counter for the in-launch reduction (the default where it applies -- 64 rows, four time ranges of at most 256 frames,
// every workgroup of the launch resident, one or several windows; MI355_XE_INKERNEL_REDUCE=0 disables it).  The arrival words behind the partial sums
// (plan.flag_offset) are two banks of one 8-byte word per (window, slice): launch e counts in bank e % 2 and clears the other bank, so nothing an
// unfinished launch leaves behind is seen by a later one.  The counter advances only when a kernel that uses the words has been enqueued; a call that
// returns an error leaves it as it was (the caller should still treat the words as unknown: zero them and restart the counter from 0).
// Launches that share a workspace must be stream-ordered (the caller's job: mi355_xengine orders them with an event when the stream changes).
// NULL: the reduce kernel is used.
int mi355_xe_fused_launch(const XeFusedPlan &p, const void *in, void *out, void *part, int N, int F, int Fout, int T, double kd,
                          int accumulate, hipStream_t st, int stations_per_group = 0, unsigned *epoch = nullptr, int nint = 1);

// Whole-line form (xengine_lines.hip): 64 stations, one polarisation (or two -- 128 rows, the reference CLI's default, lib/test-clxengine.cc:66: lines
// of 32 channels, eight pair groups per line; launch with npol = 2), rows of whole 128-byte lines, enough (window, line, pair group) units to fill
// the device without time ranges.  mi355_xe_fused_launch routes to it where mi355_xe_lines_ok says so (MI355_XE_NO_LINES=1: never).
bool mi355_xe_lines_ok(int N, int F, int Fout, int npol, int T, int stations_per_group, int accumulate, int nint, int cus);
// Fewer units than compute units (one window of BASELINE config 5 -- the reference's xcorrelate(char*, XComplex*) shape, lib/clXEngine_impl.h:184-201): the
// integration is cut into 2 or 4 time ranges (mi355_xe_lines_split, 0: not this geometry) whose partial sums the kernel's tail combines inside the launch
// through `part` (inboxes of teams x 4 x tsplit x 64 KiB, then at flag_offset two banks of one 8-byte arrival word per team; both fit the workspace
// mi355_xe_fused_plan sizes when its tsplit equals this one).  *epoch: the workspace's launch counter, advanced once the kernel is enqueued.
int mi355_xe_lines_split(int N, int F, int Fout, int npol, int T, int stations_per_group, int accumulate, int nint, int cus);
int mi355_xe_lines_launch(const void *in, void *out, int N, int F, int Fout, int T, double kd, hipStream_t st, int stations_per_group, int nint, int cus,
                          int tsplit = 1, void *part = nullptr, size_t flag_offset = 0, unsigned *epoch = nullptr, int npol = 1);
