// clXEngine, IChar (int8 I/Q) input: corner turn AND correlation in one pass over the input.
// Reference behaviour: lib/clXEngine_impl.cc:708-817 (CharToComplex + XCorrelate kernels), :859-867 (IChar scale);
// input layout [t][station][chan][pol]{I,Q} (:766-767,987-1061), output [chan][baseline][pol^2] (:786-808).
//
// Why this shape (measured on MI355X, tools/ubench/slice_read.hip): a workgroup can hold the int32 accumulators of at most
// ~16 channels (84 registers per channel and lane), i.e. a 32-byte column slice of every (t, station) row; the four
// workgroups that share a 128-byte line sit on one XCD (blockIdx % 8) and run the same time range, so HBM sees every line
// once (24.8 us for the 134 MB of BASELINE config 5; 16-byte slices 40 us, 8-byte slices 75 us: the L2 request rate, not the
// byte count, is the limit).  16 channels per workgroup x 256 CUs = 4 x the channels -> the integration is split into 4
// time ranges whose exact int32 partial sums are combined afterwards (k_xe_i8_reduce, or the tail of k_xe_i8_fused).
//
// Workgroup = 8 waves.  Raw input goes global -> LDS by DMA (global_load_lds_dwordx4, no staging registers) into a ring of
// four 16-time-step stages; two stages (32 steps = one K block of v_mfma_i32_16x16x32_i8) are consumed while two are in
// flight.  Wave w owns the w-th 4-byte unit of the slice (one polarisation: channels 2w, 2w+1; two: channel w, X and Y):
// it reads its unit of 8 consecutive time steps per lane straight from the raw image (ds_read_b32; 4-way bank conflicts are
// inherent in reading one dword of 16-byte pieces, the stage halves are skewed by 16 B to keep it at 4), byte-transposes
// in registers (v_perm_b32) into MFMA operands and accumulates all row-tile pairs of its channels.
//   re  += I_a I_b^T + Q_a Q_b^T
//   im' += Q_a I_b^T + I_a (~Q_b)^T      (~q = -q - 1 is exact for every int8, -128 included, unlike -q)
//   im   = im' + sum_t I_a(t)             (row sums via v_sad_u8 on the operand bytes)
// so a channel needs 2 accumulators per tile pair (80 registers for 64 rows) + 4 row-sum registers.
#include "xengine_fused.h"
#include <atomic>

#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
struct c32 { float x, y; };

constexpr int kStageT = 16, kRing = 4, kChunk = 1024, kWaves = 8, kThreads = kWaves * 64;

// an integration that is not a multiple of 32 frames: the time steps past its end are fetched from here (zero samples add nothing to any
// sum: I = Q = 0 makes every product 0, ~Q = -1 meets I = 0, and the row sums' bias of 128 per byte is removed per step as for real data)
__device__ __attribute__((aligned(64))) unsigned char xe_zero_row[64];

struct FuArgs {
    const unsigned char *in;
    v4i *part;
    c32 *out;
    int N, F, Fout, T;  // F: channels of the workspace indexing (rows rounded up to whole 128-byte lines), Fout: the caller's
    int row_stride;     // bytes between (t, station) rows of the input = the bytes of a row that exist (a multiple of 16)
    int ng;     // stations per antenna group: input is [group][t][station in group][...] (ng == N: the reference layout)
    int nlines, tsplit, steps;  // 128-byte lines per input row; time ranges; K blocks (32 time steps) per time range
    int pinned, accumulate;
    int *flags;        // the in-launch reduction's arrival words: two banks of flag_bank 8-byte words, one word per (window, slice):
                       // {arrival count (high 32 bits) | launch tag << 4 | give-up bits (low 32 bits)}; launch e uses bank e % 2 and clears the other
    unsigned epoch;    // launch number on this workspace (>= 1)
    unsigned flag_bank;  // words per bank = windows of the launch x slices
    unsigned wait_ticks; // bound of a unit's wait for its slice's other units, 100 MHz ticks
    int k127;            // kd == 1 / 127 exactly: the single-precision form of the scale applies (xe_scale127_small)
    int rs;            // 1: the four time ranges of a slice are combined by the kernel's own tail (reduce-scatter), 0: by k_xe_i8_reduce
    int compact;       // partial matrices of the diagonal tile pairs as ONE record (re on and below the diagonal, im above it)
    // tuning aid (MI355_XE_DBG), bits: 1 no compute, 2 no partial-sum / matrix stores (and no scaling), 8192 the scaling without the matrix stores, 4 no DMA, 32 no priority for the second wave group, 64 no products, 128 no LDS
    // reads, 512 every bounded wait of the in-launch reduction runs out at once (the tests' way into its fallback), 1024 stamp the arrival of the pieces,
    // 4096 * m (m = 1..3) other line -> XCD maps, 65536 / 131072 only / all but lines 3 and 11 of a row, 1048576 * k lines rotated over the XCDs
    int dbg;
    unsigned pf_mask;  // the rows' slow 128-byte lines (bit l = line l of a row), which the workgroups of the other lines touch pf_dist K blocks ahead
    int pf_dist;       // (see prefetch_slow; 0: off)
    unsigned long long pf_lines;  // the slow lines' numbers, one byte each in rising order (at most eight)
    int slow_first;    // >= 0: ((input address >> 7) & 7); the units of the rows' slow lines get the lowest workgroup numbers (several windows, no time ranges)
    unsigned long long *ts;  // tuning aid (MI355_XE_TS): per-workgroup phase stamps (100 MHz wall clock), NULL in normal use
    double kd;
    // batched form: nint integration windows per launch, wgs workgroups each
    int wgs;                       // workgroups per window = units * tsplit
    int nint_launch;               // windows of this launch
    size_t in_window, in_group;    // bytes between windows / between antenna groups of the input
    size_t out_window;             // output elements per window
    size_t part_window;            // v4i elements of partial sums per window
    // persistent form: a workgroup runs `items` units one after the other, unit k of workgroup b is the unit the one-unit-per-workgroup form
    // gives workgroup b + k * gridDim.x (same XCD, same 32-byte sector); the first K blocks of unit k + 1 are requested before unit k's matrix is stored
    // (only where unit k + 1 of a workgroup is the SAME slice and time range of a later window: the request addresses move by unit_in_step bytes)
    int items;
    size_t unit_in_step;
};

__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ void stamp(const FuArgs &a, int k)
{
    if (a.ts && threadIdx.x == 0) a.ts[(size_t)blockIdx.x * 16 + k] = wall_clock64();
}

// 4 dwords (bytes b0..b3 of four consecutive time steps) -> out[j] = byte j of each input dword
__device__ __forceinline__ void transpose4x4(unsigned i0, unsigned i1, unsigned i2, unsigned i3, unsigned (&out)[4])
{
    const unsigned t0 = perm(i1, i0, 0x05010400u), t1 = perm(i1, i0, 0x07030602u);
    const unsigned t2 = perm(i3, i2, 0x05010400u), t3 = perm(i3, i2, 0x07030602u);
    out[0] = perm(t2, t0, 0x05040100u);
    out[1] = perm(t2, t0, 0x07060302u);
    out[2] = perm(t3, t1, 0x05040100u);
    out[3] = perm(t3, t1, 0x07060302u);
}

__device__ __forceinline__ long pack64(unsigned lo, unsigned hi) { return (long)(((unsigned long)hi << 32) | (unsigned long)lo); }

// ---- The IChar scale in single precision, bit for bit the oracle's (float)((double)S * kd * kd) at kd = 1 / 127 (lib/clXEngine_impl.cc:859-867).
// That expression is the correctly rounded float of the rational S / 16129: a double evaluation is off by < 2^-50, and S / 16129 is never
// nearer than 2^-39 (relative) to a float rounding boundary -- a boundary below 2^17 has an odd numerator over 2^(24-e), and
// S * 2^(24-e) - 16129 * odd is a non-zero integer.  So ANY evaluation good to 2^-40 rounds to the same float.  For |S| < 2^24 (S exact in a
// float): q = fl(S * c), r = S - 16129 q exactly (one fma: |r| < 2 and a multiple of ulp(q) with a 14-bit factor), result = fl(q + r * c) --
// off by ulp(q) * 2^-25 before the final rounding.  Four full-rate instructions instead of four half-rate double ones; the matrix stores of a unit
// are bound by this arithmetic, not by the store path (measured: the same values stored to linear addresses take the same time).
// Checked against the double expression for every |S| <= 2^24 (tests/test_xengine_gpu.py::test_ichar_scale_single_precision_is_exact).
__device__ __forceinline__ float xe_scale127_small(int S)
{
    const float c = 6.2000123e-05f;  // fl(1 / 16129)
    const float sf = (float)S;
    const float q = sf * c;
    const float r = __builtin_fmaf(-q, 16129.0f, sf);
    return __builtin_fmaf(r, c, q);
}
// all values of every lane inside (-2^24, 2^24)?  (wave-uniform)
__device__ __forceinline__ bool xe_all_small(unsigned m) { return __builtin_amdgcn_ballot_w64((m >> 25) != 0u) == 0ull; }
__device__ __forceinline__ unsigned xe_mag_bits(int v) { return (unsigned)(v + 0x1000000); }  // bits 25.. set unless -2^24 <= v < 2^24

// one LDS-DMA piece: every lane fetches 16 bytes from its own global address; they land at lds_dst + lane * 16
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// partial-sum traffic between the workgroups of one slice, the hand-off recipe of cdna_hip_programming.md (Guideline 16, R1): write-through
// (sc1) 16-byte stores, every storing wave drains them, one lane raises the count; the consumer polls that one word relaxed and, after the
// match, reads the pieces with sc1 loads (which may stand in for the agent-scope acquire when the producer stored sc1)
__device__ __forceinline__ void st_sys(v4i *p, v4i v)
{
    // (the s_nop: a store of more than 8 bytes reads its data registers a cycle or two after it issues, and the compiler's hazard pass does
    // not look inside an asm statement -- without it the next vector instruction may overwrite the data of the last lanes)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}

// ---- 24-bit partial sums (compact == 2): a time range of <= 256 steps keeps |re - 1| and |im - 1| below 2^23 (re <= 2^23 only for
// -128 * -128 throughout, im <= 255 * 128 * 256), so a value travels as its low 16 bits in one plane and bits 16..23 in another:
// 24 instead of 32 bytes per eight values, both planes whole dwords per lane (the 12-byte records of the first attempt were slower
// than the bytes they saved).  Per (range, channel): OD off-diagonal tile pairs x (64 x 16 B + 64 x 8 B), then NTT diagonal ones x
// (64 x 8 B + 64 x 4 B).
typedef int v2i __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr int pk_block_bytes(int NTT) { return NTT * (NTT - 1) / 2 * 1536 + NTT * 768; }
__device__ __forceinline__ unsigned pk_lo(int a, int b) { return perm((unsigned)b, (unsigned)a, 0x05040100u); }            // a.lo16 | b.lo16 << 16
__device__ __forceinline__ unsigned pk_hi(v4i w)                                                                             // bits 16..23 of the four
{
    return perm((unsigned)w[1], (unsigned)w[0], 0x0c0c0602u) | (perm((unsigned)w[3], (unsigned)w[2], 0x0c0c0602u) << 16);
}
__device__ __forceinline__ v4i pk_bias(v4i v) { return (v4i){v[0] - 1, v[1] - 1, v[2] - 1, v[3] - 1}; }
// value k of a packed quartet: lo = the dword holding its low half (k even: bits 0..15, odd: 16..31), hi = the dword of the four top bytes
__device__ __forceinline__ int pk_get(unsigned lo, unsigned hi, int k)
{
    const unsigned l16 = (k & 1) ? (lo >> 16) : (lo & 0xffffu), h8 = (hi >> (8 * k)) & 0xffu;
    return ((int)((h8 << 24) | (l16 << 8)) >> 8) + 1;
}

template <int NPOL, int NTT, bool SPLIT, bool PP, bool RS>
__global__ __launch_bounds__(kThreads, 2) void k_xe_i8_fused(FuArgs a)
{
    constexpr int NSH = (NTT * 16 / NPOL > 32) ? 2 : 1;     // 32-station halves of a time step
    constexpr int STAGE = kStageT * NSH * kChunk + 16;       // + the 16-byte skew of the second half of the stage
    constexpr int CPW = 2 / NPOL;                            // channels per wave
    constexpr int NP = NTT * (NTT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stamp(a, 0);
    if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, 4 bits
    // ---- which slice / time range: the 4 workgroups of a 128-byte line on one XCD, same time range
    // (the batched form adds the integration window to the combination: consecutive workgroups go to the eight XCDs round robin, so the
    // four workgroups of a 128-byte line must be 8 apart to meet in one L2)
    // b: the unit's number = the workgroup number of the one-unit-per-workgroup form
    auto map_unit = [&](int b, int &slice, int &q, int &win) {
        int combo, sector;
        if (a.pinned) {
            const int xcd = (b + ((a.dbg >> 20) & 7)) & 7, within = b >> 3;  // (dbg bits 20..22: which lines an XCD gets, a tuning aid)
            sector = within & 3;
            combo = xcd + 8 * (within >> 2);
        } else {
            sector = b & 3;
            combo = b >> 2;
        }
        slice = (combo % a.nlines) * 4 + sector;
        const int rest = combo / a.nlines;
        q = rest % a.tsplit;
        win = rest / a.tsplit;
        if (a.slow_first >= 0) {
            // Several windows per launch, every unit a whole integration, more units than CUs: the dispatcher hands out workgroups in
            // blockIdx order, and a unit of a slow line (address bits 7..9 == 3: 6.9 us per K block from HBM against 4.2) that starts in the
            // last round ends the launch 90 us after everything else.  Longest first: the slow lines of ALL windows, then the rest.
            const int n8 = a.nlines >> 3, nslow = n8 * a.nint_launch, l3 = (3 - a.slow_first) & 7;
            int line;
            if (combo < nslow) { line = l3 + 8 * (combo % n8); win = combo / n8; }
            else {
                const int c2 = combo - nslow, per = a.nlines - n8, k = c2 % per;  // k-th line of the window that is not slow
                win = c2 / per;
                line = k + (k + 7 - l3) / 7;  // skips l3, l3 + 8, ...: 7 lines between two slow ones
                if (k < l3) line = k;
            }
            slice = line * 4 + sector;
        }
        if (a.pinned && a.nlines == 16 && a.tsplit == 4 && a.nint_launch == 1 && ((a.dbg >> 12) & 3)) {  // tuning aid: other line -> XCD maps
            const int xcd = b & 7, within = b >> 3, mode = (a.dbg >> 12) & 3;
            int line;
            if (mode == 1) { line = ((within >> 2) & 1) ? 8 + ((xcd + 4) & 7) : xcd; q = within >> 3; }
            else if (mode == 2) { q = xcd >> 1; line = (xcd & 1) * 8 + (within >> 2); }
            else { q = xcd & 3; line = (xcd >> 2) * 8 + (within >> 2); }
            slice = line * 4 + (within & 3);
        }
    };
    if constexpr (RS) {
        // The arrival words come in two banks, launch e counts in bank e % 2 from zero to four and every launch first clears the OTHER bank's word
        // of its slice: whatever an earlier launch left there (a launch that did not run to its end, a debug switch that made workgroups leave
        // before they arrived) is gone before launch e + 1 -- stream-ordered behind this one -- looks at it.  No state survives two launches.
        if (tid == 0) {
            int slice, q, win;
            map_unit(blockIdx.x, slice, q, win);
            if (q == 0)
                __hip_atomic_store((unsigned long long *)a.flags + (size_t)((a.epoch + 1u) & 1u) * a.flag_bank + (size_t)win * (a.nlines * 4) + slice, 0ull,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if ((a.dbg >> 16) & 3) {  // tuning aid: only (1) / all but (2) the lines 3 and 11 of a row
        int slice, q, win;
        map_unit(blockIdx.x, slice, q, win);
        const bool l3 = ((slice >> 2) & 7) == 3;
        if ((((a.dbg >> 16) & 3) == 1) != l3) return;
    }
    const size_t row_bytes = (size_t)a.row_stride;  // (a row may end inside its last 128-byte line: the pieces past its end read the zero row)
    const unsigned lds0 = (unsigned)(size_t)lds;

    // ---- DMA of one 16-step stage: chunk = (time step, station half) = 32 stations x 32 B, 2 * NSH chunks per wave
    // station s of time step t sits at (s / ng) * in_group + (t * ng + s % ng) * row_bytes: the frames of antenna group s / ng are one
    // contiguous block (what the all-to-all corner turn of shard.py delivers; in_group = windows * T * ng rows); ng == N is the
    // reference's layout
    const size_t t_stride = (size_t)a.ng * row_bytes;
    // the unit whose K blocks are being REQUESTED (in the persistent form one unit ahead of the unit being multiplied at a unit's end)
    const unsigned char *src_lane[NSH];
    const unsigned char *pf_src = nullptr;
    bool in_row = false;
    int t_base = 0;
    // ---- The slow lines.  HBM serves the 128-byte lines whose address has bits 7..9 == 3 (two of a 2 KiB row's sixteen) with ~1.75 x the
    // latency of the others, and a CU's request stream is latency bound: from HBM the 32 workgroups of those lines end their loop at 69 us
    // when the other 224 are done at 37 (profiles/r04_xengine_counters.txt section 0).  Once a line sits in the Infinity Cache the gap is small
    // (35 against 29 us on a cache-resident input).  So the workgroups of the OTHER lines -- which run ahead -- touch the slow lines' rows of
    // their own time range two K blocks early: one 16-byte request per line, the 52 workgroups of the other lines share a block's 6144 lines
    // (lines 3 and 11, and line 15 whose latency is 1.15 x), 32 lanes of each wave of the first group, ~6 % more requests for them; the data land in a 1 KiB scratch behind the ring and are never read.  (First in the wave's queue of a
    // step, so the step's own loads are not held up by the slow class's latency; only the ping-pong schedule, whose waits are vmcnt(0).)
    // (the lane's source address for K block 0 is worked out once per unit; a step adds the block's offset: one address add and one request)
    auto setup_requests = [&](int b) {
        int slice, q, win;
        map_unit(b, slice, q, win);
        const unsigned char *in_w = a.in + (size_t)win * a.in_window;
        t_base = q * a.steps * 32;
#pragma unroll
        for (int sh = 0; sh < NSH; sh++) {
            const int s = sh * 32 + (lane >> 1);
            src_lane[sh] = in_w + (size_t)(s / a.ng) * a.in_group + (size_t)(s % a.ng) * row_bytes + (size_t)slice * 32 + (lane & 1) * 16;
        }
        in_row = slice * 32 + (lane & 1) * 16 < a.row_stride;  // this lane's 16 bytes of the slice exist
        pf_src = nullptr;
        if (a.pf_dist > 0 && wave < 4) {
            const int pf_line = slice >> 2, pf_n = __builtin_popcount(a.pf_mask);
            const int pf_wgs = (a.nlines - pf_n) * 4, pf_rank = __builtin_popcount(~a.pf_mask & ((1u << pf_line) - 1u)) * 4 + (slice & 3);  // this workgroup among those of the other lines
            const int items = 32 * a.N * pf_n, per = (items + pf_wgs - 1) / pf_wgs;  // (row of a K block, slow line): at most 128 per workgroup, 32 per wave of the first group
            const int k = wave * 32 + lane, i = pf_rank * per + k;
            if (!((a.pf_mask >> pf_line) & 1) && lane < 32 && k < per && i < items) {
                const int which = i % pf_n, row = i / pf_n, t = row / a.N, st = row - t * a.N;
                const int ln = (int)((a.pf_lines >> (8 * which)) & 0xffu);  // the which-th slow line
                pf_src = in_w + (size_t)(st / a.ng) * a.in_group + ((size_t)(t_base + t) * a.ng + st % a.ng) * row_bytes + (size_t)ln * 128;
            }
        }
    };
    setup_requests(blockIdx.x);
    // stage sigma (16 time steps) of the unit being requested, into ring slot ring_stage % 4 (ring_stage counts the stages of ALL units of
    // this workgroup: the ring does not care which unit a stage belongs to)
    auto issue_stage = [&](int sigma, int ring_stage) {
        const int slot = ring_stage & (kRing - 1), t0 = t_base + sigma * kStageT;
#pragma unroll
        for (int k = 0; k < 2 * NSH; k++) {
            const int idx = wave + kWaves * k, t16 = idx / NSH, sh = idx % NSH;
            const int s = sh * 32 + (lane >> 1);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + idx * kChunk + (t16 >> 3) * 16);
            const unsigned char *from = (t0 + t16 < a.T && in_row) ? src_lane[sh] + (size_t)(t0 + t16) * t_stride : xe_zero_row + (lane & 1) * 16;
            if (s < a.N && !(a.dbg & 4)) dma16(from, dst);
        }
    };

    v4i re[CPW][NP], im[CPW][NP];
    unsigned rs[CPW][NTT];  // biased row sums of I (v_sad_u8 of the bytes xor 0x80)
    if constexpr (!PP) {
#pragma unroll
        for (int c = 0; c < CPW; c++) {
#pragma unroll
            for (int p = 0; p < NP; p++) re[c][p] = im[c][p] = (v4i){0, 0, 0, 0};
#pragma unroll
            for (int rt = 0; rt < NTT; rt++) rs[c][rt] = 0u;
        }
    }

    const int r = lane & 15, g = lane >> 4;
    const int hpiece = wave >> 2, dq = wave & 3;
    const int lane_base = (g >> 1) * STAGE + (g & 1) * (8 * NSH * kChunk + 16) + hpiece * 16 + dq * 4 + ((NPOL == 1) ? r : (r >> 1)) * 32;
    const bool odd_pol = (r & 1) != 0;

    // Ring schedule: step j consumes stages 2j, 2j+1 (slots (2j) % 4, +1).  The raw bytes are pulled into registers (as MFMA
    // operands) first, so the slots are free again after a second barrier and the DMA of step j+2 is issued BEFORE the matrix
    // products of step j: two steps (128 KB per CU) are in flight while the wave multiplies.
    constexpr int PER_STEP = 4 * NSH;  // DMA instructions per wave and step (every wave issues all of them: N > 32 when NSH == 2)
    auto prefetch_slow = [&](int blk) {
        if (pf_src) dma16(pf_src + (size_t)blk * 32 * t_stride, __builtin_amdgcn_readfirstlane(lds0 + kRing * STAGE));
    };
    issue_stage(0, 0);
    issue_stage(1, 1);
    if (a.steps > 1) {
        issue_stage(2, 2);
        issue_stage(3, 3);
    }
    // the raw bytes of block j (stages 2j, 2j + 1) -> MFMA operands in registers
    auto read_block = [&](int j, long (&I)[CPW][NTT], long (&Q)[CPW][NTT]) {
        {
            const unsigned char *base = lds + ((2 * j) & (kRing - 1)) * STAGE + lane_base;
#pragma unroll
            for (int rt = 0; rt < NTT; rt++) {
                const int tile_imm = (NPOL == 1) ? (rt >> 1) * kChunk + (rt & 1) * 512 : rt * 256;
                unsigned raw[8];
#pragma unroll
                for (int i = 0; i < 8; i++) raw[i] = *(const unsigned *)(base + i * NSH * kChunk + tile_imm);
                unsigned o0[4], o1[4];
                transpose4x4(raw[0], raw[1], raw[2], raw[3], o0);
                transpose4x4(raw[4], raw[5], raw[6], raw[7], o1);
                if constexpr (NPOL == 1) {
                    // unit bytes: I(2w) Q(2w) I(2w+1) Q(2w+1); row = station
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        I[c][rt] = pack64(o0[2 * c], o1[2 * c]);
                        Q[c][rt] = pack64(o0[2 * c + 1], o1[2 * c + 1]);
                    }
                } else {
                    // unit bytes: XI XQ YI YQ of channel w; rows 2s (X), 2s+1 (Y): even / odd lanes of a station pair
                    I[0][rt] = odd_pol ? pack64(o0[2], o1[2]) : pack64(o0[0], o1[0]);
                    Q[0][rt] = odd_pol ? pack64(o0[3], o1[3]) : pack64(o0[1], o1[1]);
                }
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    const unsigned lo = (unsigned)(unsigned long)I[c][rt], hi = (unsigned)((unsigned long)I[c][rt] >> 32);
                    rs[c][rt] = __builtin_amdgcn_sad_u8(lo ^ 0x80808080u, 0u, rs[c][rt]);
                    rs[c][rt] = __builtin_amdgcn_sad_u8(hi ^ 0x80808080u, 0u, rs[c][rt]);
                }
                // (two row tiles' raw dwords in flight at most: all four at once cost sixteen more registers than the kernel has)
                if (rt & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // lockstep form: wait for the block's two stages, pull the raw bytes into registers (the slots are free again after the second barrier),
    // request the stages two blocks ahead
    auto load_block = [&](int j, long (&I)[CPW][NTT], long (&Q)[CPW][NTT]) {
        if (j + 1 < a.steps) {
            if constexpr (PER_STEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // stages 2j, 2j+1 have landed for every wave
        if (j == 0) stamp(a, 1);
        if (!(a.dbg & 1)) read_block(j, I, Q);
        __syncthreads();  // every wave holds its operands in registers: the two slots are free
        if (a.pf_dist > 0) {  // (older than the stages issued next: the next step's vmcnt wait covers it, nothing is miscounted)
            if (j == 0) for (int b2 = 2; b2 <= a.pf_dist && b2 < a.steps; b2++) prefetch_slow(b2);
            if (j + 1 + a.pf_dist < a.steps) prefetch_slow(j + 1 + a.pf_dist);
        }
        if (j + 2 < a.steps) {
            issue_stage(2 * j + 4, 2 * j + 4);
            issue_stage(2 * j + 5, 2 * j + 5);
        }
    };
    // two sweeps over the pairs so that consecutive MFMAs never touch the same accumulator
    auto mfma32 = [&](const long (&I)[CPW][NTT], const long (&Q)[CPW][NTT]) {
#pragma unroll
        for (int c = 0; c < CPW; c++) {
#pragma unroll
            for (int bi = 0; bi < NTT; bi++)
#pragma unroll
                for (int bj = 0; bj <= bi; bj++) {
                    const int p = bi * (bi + 1) / 2 + bj;
                    re[c][p] = __builtin_amdgcn_mfma_i32_16x16x32_i8(I[c][bi], I[c][bj], re[c][p], 0, 0, 0);
                    im[c][p] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Q[c][bi], I[c][bj], im[c][p], 0, 0, 0);
                }
#pragma unroll
            for (int bi = 0; bi < NTT; bi++)
#pragma unroll
                for (int bj = 0; bj <= bi; bj++) {
                    const int p = bi * (bi + 1) / 2 + bj;
                    re[c][p] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Q[c][bi], Q[c][bj], re[c][p], 0, 0, 0);
                    im[c][p] = __builtin_amdgcn_mfma_i32_16x16x32_i8(I[c][bi], ~Q[c][bj], im[c][p], 0, 0, 0);
                }
        }
    };
    __shared__ int s_mode, s_mask;
    // ---- what a unit does with its finished accumulators (b: the unit's number, see map_unit)
    auto finish_unit = [&](int b, int unit_idx, bool last_unit) {
    int slice, q, win;
    map_unit(b, slice, q, win);
    c32 *const out_w = a.out + (size_t)win * a.out_window;
    v4i *const part_w = a.part + (size_t)win * a.part_window;
    stamp(a, 2);
    if (a.ts && tid == 0 && unit_idx >= 1 && unit_idx < 4) a.ts[(size_t)blockIdx.x * 16 + 8 + 2 * unit_idx] = wall_clock64();  // (later units: loop end)
    // ---- Reduce-scatter of the four time ranges of a slice INSIDE the launch (64-row geometry, four ranges of at most 256 steps).
    // A lane holds 64 values per channel: 8 per off-diagonal tile pair (re, im) and 4 per diagonal one (re on and below the diagonal, im
    // above it), i.e. 16 quads; quads 4u .. 4u+3 belong to unit (time range) u of the slice: u = 0, 1, 2 two off-diagonal pairs each,
    // u = 3 the four diagonal ones.  Every unit keeps its own quads in registers, sends the other twelve as 24-bit planes (three 16-byte
    // pieces per lane and unit: write-through stores, the only store form that is cheap per byte at system scope) to the owners' inboxes,
    // raises the slice's arrival count, and once all four have arrived adds the three pieces it received to its registers, scales and
    // scatters its quarter of the matrix.  37.5 MB written and read back per BASELINE integration instead of 50 + 50 + a second kernel.
    // Placement independent: write-through (sc1) stores, sc1 loads, the count raised after the stores have drained.  Bounded wait:
    // a unit that gives up stores its own quads too, sets its bit in the slice's word and exits; the LAST unit to arrive sees the bits
    // with its own arrival and finishes those quarters from the inboxes -- complete for any dispatch order, nobody waits for a workgroup
    // that has not started.
    if constexpr (RS) {
        static_assert(SPLIT && NTT == 4, "the reduce-scatter tail is the 64-row, split form's");
        {
            const int QC = q;
            // (lane-derived values re-defined here: otherwise every index expression of this tail is hoisted to the kernel's entry and
            // lives -- in registers the main loop does not have -- across the whole integration)
            int r = lane & 15, g = lane >> 4, ll = lane;
            asm volatile("" : "+v"(r), "+v"(g), "+v"(ll));
            const int nbl = a.N * (a.N + 1) / 2, np2l = NPOL * NPOL, Al = a.N * NPOL;
            unsigned char *inbox = (unsigned char *)part_w + (size_t)slice * (4 * 4 * kWaves * CPW * 3072);
            auto slot = [&](int dst, int src, int c) { return inbox + ((size_t)((dst * 4 + src) * kWaves + wave) * CPW + c) * 3072 + ll * 16; };
            int own[CPW][16];
#pragma unroll
            for (int c = 0; c < CPW; c++) {
#pragma unroll
                for (int i = 0; i < 16; i++) own[c][i] = 0;
                int rsum[NTT];
#pragma unroll
                for (int rt = 0; rt < NTT; rt++) {
                    int v = (int)rs[c][rt] - 128 * 8 * a.steps;
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    rsum[rt] = v;
                }
                int corr[NTT][4];  // row sums of the rows this lane holds in the C layout: row = 4 * (lane / 16) + reg
#pragma unroll
                for (int rt = 0; rt < NTT; rt++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) corr[rt][reg] = __shfl(rsum[rt], 4 * g + reg);
                auto pair_vals = [&](int bi, int bj, v4i &vre, v4i &vim) {
                    const int p = bi * (bi + 1) / 2 + bj;
                    vre = re[c][p];
                    vim = im[c][p];
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) vim[reg] += corr[bi][reg];
                };
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int v[16];
                    if (u < 3) {
#pragma unroll
                        for (int it = 0; it < 2; it++) {
                            const int k = 2 * u + it, bi = k < 1 ? 1 : k < 3 ? 2 : 3, bj = k - bi * (bi - 1) / 2;
                            v4i vre, vim;
                            pair_vals(bi, bj, vre, vim);
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) { v[8 * it + reg] = vre[reg]; v[8 * it + 4 + reg] = vim[reg]; }
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; d++) {
                            v4i vre, vim;
                            pair_vals(d, d, vre, vim);
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) v[4 * d + reg] = (4 * g + reg >= r) ? vre[reg] : vim[reg];
                        }
                    }
                    const bool mine_u = u == QC;
#pragma unroll
                    for (int i = 0; i < 16; i++) own[c][i] = mine_u ? v[i] : own[c][i];  // (selects, not a branch: the quads stay in registers)
                    if (!mine_u && !(a.dbg & 2)) {
                        unsigned char *d = slot(u, QC, c);
#pragma unroll
                        for (int i = 0; i < 16; i++) v[i] -= 1;
                        const v4i c0 = (v4i){(int)pk_lo(v[0], v[1]), (int)pk_lo(v[2], v[3]), (int)pk_lo(v[4], v[5]), (int)pk_lo(v[6], v[7])};
                        const v4i c1 = (v4i){(int)pk_lo(v[8], v[9]), (int)pk_lo(v[10], v[11]), (int)pk_lo(v[12], v[13]), (int)pk_lo(v[14], v[15])};
                        const v4i c2 = (v4i){(int)pk_hi((v4i){v[0], v[1], v[2], v[3]}), (int)pk_hi((v4i){v[4], v[5], v[6], v[7]}),
                                             (int)pk_hi((v4i){v[8], v[9], v[10], v[11]}), (int)pk_hi((v4i){v[12], v[13], v[14], v[15]})};
                        st_sys((v4i *)d, c0);
                        st_sys((v4i *)(d + 1024), c1);
                        st_sys((v4i *)(d + 2048), c2);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one unit's quads at a time (the accumulators die as they are sent)
                }
            }
            stamp(a, 3);
            // arrival count in the high word, {launch tag, give-up bits} in the low; this launch's bank of words (see the kernel's entry)
            unsigned long long *state = (unsigned long long *)a.flags + (size_t)(a.epoch & 1u) * a.flag_bank + (size_t)win * (a.nlines * 4) + slice;
            const unsigned full = 4u, tag = a.epoch & 0x0fffffffu;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have completed
            __syncthreads();
            stamp(a, 4);
            if (tid == 0) {
                const unsigned long long old = __hip_atomic_fetch_add(state, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int mode = 0, mask = 0;
                if ((unsigned)(old >> 32) + 1u == full) {  // the last to arrive: the give-up bits are final (they can only be set while the count is short)
                    mode = 2;
                    if (((unsigned)old >> 4) == tag) mask = (int)((unsigned)old & 15u);
                    if ((unsigned)old) __hip_atomic_fetch_and(state, 0xffffffff00000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    // Bounded by TIME (100 MHz wall clock): the units of a slice finish within a microsecond or two of each other when all are
                    // resident.  A partner that is not -- the device shared with another stream's kernels, e.g. the exchange of the sharded
                    // pipeline: a workgroup of this kernel needs a whole CU's registers -- starts when the first workgroups leave and arrives
                    // one loop + send later; waiting that long costs the same as the second kernel would, giving up before it comes makes
                    // it finish the whole slice alone.  So the bound is about one loop of THIS geometry (FuArgs::wait_ticks: 10 us + 6 us per
                    // K block of a range, between 20 and 100 us); beyond that nobody holds a CU.
                    // (dbg 512: give up at once -- exercises the fallback in the tests)
                    const unsigned long long t_wait = wall_clock64(), limit = (a.dbg & 512) ? 0 : (unsigned long long)a.wait_ticks;
                    do {
                        const unsigned long long cur = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(cur >> 32) == full) mode = 1;
                        else __builtin_amdgcn_s_sleep(8);
                    } while (!mode && wall_clock64() - t_wait < limit);
                    if (!mode) mode = 3;
                }
                s_mode = mode;
                s_mask = mask;
            }
            __syncthreads();
            stamp(a, 5);
            if (s_mode == 3) {  // waited long enough: hand this unit's own quads over as well, then say so -- unless everybody has arrived meanwhile
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    unsigned char *d = slot(QC, QC, c);
                    int v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = own[c][i] - 1;
                    st_sys((v4i *)d, (v4i){(int)pk_lo(v[0], v[1]), (int)pk_lo(v[2], v[3]), (int)pk_lo(v[4], v[5]), (int)pk_lo(v[6], v[7])});
                    st_sys((v4i *)(d + 1024), (v4i){(int)pk_lo(v[8], v[9]), (int)pk_lo(v[10], v[11]), (int)pk_lo(v[12], v[13]), (int)pk_lo(v[14], v[15])});
                    st_sys((v4i *)(d + 2048), (v4i){(int)pk_hi((v4i){v[0], v[1], v[2], v[3]}), (int)pk_hi((v4i){v[4], v[5], v[6], v[7]}),
                                                    (int)pk_hi((v4i){v[8], v[9], v[10], v[11]}), (int)pk_hi((v4i){v[12], v[13], v[14], v[15]})});
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    int mode = -1;
                    while (mode < 0) {
                        unsigned long long cur = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(cur >> 32) == full) { mode = 1; break; }
                        const unsigned bits = (((unsigned)cur >> 4) == tag ? ((unsigned)cur & 15u) : 0u) | (1u << QC);
                        const unsigned long long want = (cur & 0xffffffff00000000ull) | (unsigned long long)((tag << 4) | bits);
                        if (__hip_atomic_compare_exchange_strong(state, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) mode = 0;
                    }
                    s_mode = mode;
                }
                __syncthreads();
                if (s_mode == 0) return;
            }
            // scale and scatter the sixteen quads' worth of unit u, channel c (values in the order they were sent)
            auto emit = [&](int u, int c, const int (&v)[16]) {
                const int f = slice * (16 / NPOL) + ((NPOL == 1) ? 2 * wave + c : wave);
                if (f >= a.Fout) return;
                unsigned mag = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) mag |= xe_mag_bits(v[i]);
                const bool small = a.k127 && xe_all_small(mag);  // (wave-uniform: the single-precision form of the scale, see xe_scale127_small)
                if (u < 3 && NPOL == 1 && Al == 64 && !a.accumulate) {
                    // whole tiles, one polarisation: neighbouring lanes hold neighbouring baselines of a row, so a lane pair swaps one value
                    // each and every lane stores 16 bytes (two baselines) of ONE row -- half the store instructions of the 8-byte form, and
                    // the tail of this kernel is bound by store issue, not by bytes
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int k = 2 * u + it, bi = k < 1 ? 1 : k < 3 ? 2 : 3, bj = k - bi * (bi - 1) / 2;
#pragma unroll
                        for (int rp = 0; rp < 4; rp += 2) {
                            c32 w[2];
#pragma unroll
                            for (int e = 0; e < 2; e++) {
                                if (small) {
                                    w[e].x = xe_scale127_small(v[8 * it + rp + e]);
                                    w[e].y = xe_scale127_small(v[8 * it + 4 + rp + e]);
                                } else {
                                    w[e].x = (float)((double)v[8 * it + rp + e] * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
                                    w[e].y = (float)((double)v[8 * it + 4 + rp + e] * a.kd * a.kd);
                                }
                            }
                            const bool odd = (r & 1) != 0;
                            const float sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;  // what the neighbour stores of this lane's values
                            const float gx = __shfl_xor(sx, 1), gy = __shfl_xor(sy, 1);
                            const int s1 = bi * 16 + 4 * g + rp + (odd ? 1 : 0), s2 = bj * 16 + (r & ~1);
                            const size_t o = (size_t)f * nbl + (s1 * (s1 + 1) / 2 + s2);
                            typedef float v4f __attribute__((ext_vector_type(4)));
                            const v4f q4 = odd ? (v4f){gx, gy, w[1].x, w[1].y} : (v4f){w[0].x, w[0].y, gx, gy};
                            __builtin_memcpy((void *)(out_w + o), &q4, 16);
                        }
                    }
                } else if (u < 3) {
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int k = 2 * u + it, bi = k < 1 ? 1 : k < 3 ? 2 : 3, bj = k - bi * (bi - 1) / 2;
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int r1 = bi * 16 + 4 * g + reg, r2 = bj * 16 + r;
                            if (r1 >= Al || r2 >= Al) continue;
                            const int s1 = r1 / NPOL, p1 = r1 % NPOL, s2 = r2 / NPOL, p2 = r2 % NPOL;
                            if (s1 < s2) continue;
                            const size_t o = ((size_t)f * nbl + (s1 * (s1 + 1) / 2 + s2)) * np2l + p1 * NPOL + p2;
                            c32 w;
                            w.x = (float)((double)v[8 * it + reg] * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
                            w.y = (float)((double)v[8 * it + 4 + reg] * a.kd * a.kd);
                            if (a.accumulate) { w.x += out_w[o].x; w.y += out_w[o].y; }
                            out_w[o] = w;
                        }
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        // v[4d + k] = C[i][j] at i = 4 g + k, j = r: re[i][j] for i >= j, im[i][j] for i < j.  C[j][i] sits in lane 16 (j / 4) + i,
                        // register j % 4
                        // the transposed tile through this wave's corner of the (now idle) ring: element (i, j) is written to row i, read
                        // back by the lane that holds (j, i) as the four consecutive words of ITS row -- four 4-byte writes and one 16-byte
                        // read per tile instead of sixteen lane permutes
                        int *tile = (int *)(lds + wave * (16 * 20 * 4));
#pragma unroll
                        for (int k = 0; k < 4; k++) tile[(4 * g + k) * 20 + r] = v[4 * d + k];
                        const v4i trow = *(const v4i *)(tile + r * 20 + 4 * g);  // C[r][4 g + reg], reg = 0 .. 3 (same wave wrote it: no barrier)
                        if (NPOL == 1 && Al == 64 && !a.accumulate && !(a.dbg & 2048)) {
                            // whole tiles, one polarisation: as in the off-diagonal quarters a lane pair swaps a value and every lane stores two
                            // neighbouring baselines of ONE row (16 bytes) -- or one (8 bytes) where the row's triangle ends inside the pair
#pragma unroll
                            for (int rp = 0; rp < 4; rp += 2) {
                                c32 w[2];
#pragma unroll
                                for (int e = 0; e < 2; e++) {
                                    const int i = 4 * g + rp + e;
                                    const int tr = trow[rp + e], sre = v[4 * d + rp + e];
                                    const int sim = i > r ? -tr : 0;  // (below the diagonal im[i][j] = -im[j][i]; on it 0; above it: not stored)
                                    if (small) {
                                        w[e].x = xe_scale127_small(sre);
                                        w[e].y = xe_scale127_small(sim);
                                    } else {
                                        w[e].x = (float)((double)sre * a.kd * a.kd);
                                        w[e].y = (float)((double)sim * a.kd * a.kd);
                                    }
                                }
                                const bool odd = (r & 1) != 0;
                                const float sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;
                                const float gx = __shfl_xor(sx, 1), gy = __shfl_xor(sy, 1);
                                const int i = 4 * g + rp + (odd ? 1 : 0), j0 = r & ~1;  // this lane stores columns j0, j0 + 1 of row i
                                const int s1 = d * 16 + i;
                                c32 *dst = out_w + (size_t)f * nbl + (s1 * (s1 + 1) / 2 + d * 16 + j0);
                                const c32 first = odd ? c32{gx, gy} : w[0], second = odd ? w[1] : c32{gx, gy};
                                if (j0 + 1 <= i) {
                                    typedef float v4f __attribute__((ext_vector_type(4)));
                                    const v4f q4 = (v4f){first.x, first.y, second.x, second.y};
                                    __builtin_memcpy((void *)dst, &q4, 16);
                                } else if (j0 <= i) {
                                    *dst = first;
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int i = 4 * g + reg;
                            const int tr = trow[reg];
                            int sre = v[4 * d + reg], sim;
                            if (i > r) sim = -tr;                 // im[i][j] = -im[j][i]
                            else if (i == r) sim = 0;
                            else { sim = sre; sre = tr; }         // above the diagonal (same-station polarisation products): re[i][j] = re[j][i]
                            const int r1 = d * 16 + i, r2 = d * 16 + r;
                            if (r1 >= Al || r2 >= Al) continue;
                            const int s1 = r1 / NPOL, p1 = r1 % NPOL, s2 = r2 / NPOL, p2 = r2 % NPOL;
                            if (s1 < s2) continue;
                            const size_t o = ((size_t)f * nbl + (s1 * (s1 + 1) / 2 + s2)) * np2l + p1 * NPOL + p2;
                            c32 w;
                            w.x = (float)((double)sre * a.kd * a.kd);
                            w.y = (float)((double)sim * a.kd * a.kd);
                            if (a.accumulate) { w.x += out_w[o].x; w.y += out_w[o].y; }
                            out_w[o] = w;
                        }
                    }
                }
            };
            auto unpack_add = [&](int (&v)[16], const v4i &c0, const v4i &c1, const v4i &c2) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int j = i >> 2, k = i & 3;
                    v[i] += pk_get((unsigned)((j >> 1) ? c1 : c0)[(j & 1) * 2 + (k >> 1)], (unsigned)c2[j], k);
                }
            };
            // this unit's quarter: the three other ranges' pieces for both channels in ONE batch of loads.  (The loads AND their wait are one
            // asm statement: the compiler may move or spill an asm's outputs right behind the statement, which for a bare load instruction
            // means before the data has arrived.  sc1 loads: served from the memory side like the sc1 stores that wrote the data, no acquire.)
            {
                v4i x[CPW][3][3];
                const unsigned char *d[CPW][3];
#pragma unroll
                for (int c = 0; c < CPW; c++)
#pragma unroll
                    for (int k = 0; k < 3; k++) d[c][k] = slot(QC, (QC + 1 + k) & 3, c);
                if constexpr (CPW == 2) {
                    asm volatile(
                        "global_load_dwordx4 %0, %18, off sc1\n\tglobal_load_dwordx4 %1, %18, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %18, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %3, %19, off sc1\n\tglobal_load_dwordx4 %4, %19, off offset:1024 sc1\n\tglobal_load_dwordx4 %5, %19, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %6, %20, off sc1\n\tglobal_load_dwordx4 %7, %20, off offset:1024 sc1\n\tglobal_load_dwordx4 %8, %20, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %9, %21, off sc1\n\tglobal_load_dwordx4 %10, %21, off offset:1024 sc1\n\tglobal_load_dwordx4 %11, %21, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %12, %22, off sc1\n\tglobal_load_dwordx4 %13, %22, off offset:1024 sc1\n\tglobal_load_dwordx4 %14, %22, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %15, %23, off sc1\n\tglobal_load_dwordx4 %16, %23, off offset:1024 sc1\n\tglobal_load_dwordx4 %17, %23, off offset:2048 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(x[0][0][0]), "=&v"(x[0][0][1]), "=&v"(x[0][0][2]), "=&v"(x[0][1][0]), "=&v"(x[0][1][1]), "=&v"(x[0][1][2]), "=&v"(x[0][2][0]),
                          "=&v"(x[0][2][1]), "=&v"(x[0][2][2]), "=&v"(x[CPW - 1][0][0]), "=&v"(x[CPW - 1][0][1]), "=&v"(x[CPW - 1][0][2]), "=&v"(x[CPW - 1][1][0]),
                          "=&v"(x[CPW - 1][1][1]), "=&v"(x[CPW - 1][1][2]), "=&v"(x[CPW - 1][2][0]), "=&v"(x[CPW - 1][2][1]), "=&v"(x[CPW - 1][2][2])
                        : "v"(d[0][0]), "v"(d[0][1]), "v"(d[0][2]), "v"(d[CPW - 1][0]), "v"(d[CPW - 1][1]), "v"(d[CPW - 1][2])
                        : "memory");
                } else {
                    asm volatile(
                        "global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %9, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %9, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %3, %10, off sc1\n\tglobal_load_dwordx4 %4, %10, off offset:1024 sc1\n\tglobal_load_dwordx4 %5, %10, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %6, %11, off sc1\n\tglobal_load_dwordx4 %7, %11, off offset:1024 sc1\n\tglobal_load_dwordx4 %8, %11, off offset:2048 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(x[0][0][0]), "=&v"(x[0][0][1]), "=&v"(x[0][0][2]), "=&v"(x[0][1][0]), "=&v"(x[0][1][1]), "=&v"(x[0][1][2]), "=&v"(x[0][2][0]),
                          "=&v"(x[0][2][1]), "=&v"(x[0][2][2])
                        : "v"(d[0][0]), "v"(d[0][1]), "v"(d[0][2])
                        : "memory");
                }
                if (a.dbg & 1024) stamp(a, 1);  // (tuning aid: "first data" then holds the time the pieces have arrived)
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    int v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = own[c][i];
#pragma unroll
                    for (int k = 0; k < 3; k++) unpack_add(v, x[c][k][0], x[c][k][1], x[c][k][2]);
                    emit(QC, c, v);
                }
            }
            // (last arriver only, and only after a bounded wait ran out somewhere) the quarters of the units that gave up: all four pieces
            // come from the inboxes, the unit's own one included
            const int todo = s_mode == 2 ? (s_mask & ~(1 << QC)) : 0;
            for (int u = 0; u < 4; u++) {
                if (!((todo >> u) & 1)) continue;
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    v4i x[4][3];
                    const unsigned char *d0 = slot(u, 0, c), *d1 = slot(u, 1, c), *d2 = slot(u, 2, c), *d3 = slot(u, 3, c);
                    asm volatile(
                        "global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %12, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %3, %13, off sc1\n\tglobal_load_dwordx4 %4, %13, off offset:1024 sc1\n\tglobal_load_dwordx4 %5, %13, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %14, off offset:1024 sc1\n\tglobal_load_dwordx4 %8, %14, off offset:2048 sc1\n\t"
                        "global_load_dwordx4 %9, %15, off sc1\n\tglobal_load_dwordx4 %10, %15, off offset:1024 sc1\n\tglobal_load_dwordx4 %11, %15, off offset:2048 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[1][0]), "=&v"(x[1][1]), "=&v"(x[1][2]), "=&v"(x[2][0]), "=&v"(x[2][1]),
                          "=&v"(x[2][2]), "=&v"(x[3][0]), "=&v"(x[3][1]), "=&v"(x[3][2])
                        : "v"(d0), "v"(d1), "v"(d2), "v"(d3)
                        : "memory");
                    int v[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = 0;
#pragma unroll
                    for (int src = 0; src < 4; src++) unpack_add(v, x[src][0], x[src][1], x[src][2]);
                    emit(u, c, v);
                }
            }
            if (a.ts) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(a, 6);
            }
            return;
        }
    }
    // ---- epilogue.  Row sums: lane (r, g) summed its 8 bytes of every step; total over the four g; remove the bias.
    if constexpr (!RS) {
    const int nb = a.N * (a.N + 1) / 2, np2 = NPOL * NPOL, A = a.N * NPOL;
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        const int f = slice * (16 / NPOL) + ((NPOL == 1) ? 2 * wave + c : wave);
        int rsum[NTT];
#pragma unroll
        for (int rt = 0; rt < NTT; rt++) {
            int v = (int)rs[c][rt] - 128 * 8 * a.steps;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            rsum[rt] = v;  // every lane: sum_t I of row (rt, lane % 16)
        }
#pragma unroll
        for (int bi = 0; bi < NTT; bi++) {
            int corr[4];  // row sums of the rows this lane holds in the C layout: row = 4 * (lane / 16) + reg
#pragma unroll
            for (int reg = 0; reg < 4; reg++) corr[reg] = __shfl(rsum[bi], 4 * g + reg);
#pragma unroll
            for (int bj = 0; bj <= bi; bj++) {
                const int p = bi * (bi + 1) / 2 + bj;
                v4i vre = re[c][p], vim = im[c][p];
#pragma unroll
                for (int reg = 0; reg < 4; reg++) vim[reg] += corr[reg];
                if (a.dbg & 2) { if (vre[0] == 0x12345678 && vim[1] == 0x7654321) part_w[lane] = vre; continue; }
                if constexpr (!SPLIT && NPOL == 1 && NTT == 4) {
                    // whole 64-row matrices, one polarisation: neighbouring lanes hold neighbouring baselines of a row, so a lane pair swaps one
                    // value each and every lane stores 16 bytes (two baselines) of ONE row -- half the store instructions of the 8-byte form
                    // (the matrix stores of a unit are bound by store issue, not by bytes); on a diagonal tile pair the row's triangle may end
                    // inside the pair (one baseline, 8 bytes) or before it (nothing)
                    if (A == 64 && !a.accumulate && !(a.dbg & 2048)) {
                        if (f >= a.Fout) continue;
                        const bool odd = (r & 1) != 0;
                        unsigned mag = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) mag |= xe_mag_bits(vre[k]) | xe_mag_bits(vim[k]);
                        const bool small = a.k127 && xe_all_small(mag);
#pragma unroll
                        for (int rp = 0; rp < 4; rp += 2) {
                            c32 w[2];
                            if (small) {
#pragma unroll
                                for (int e = 0; e < 2; e++) {
                                    w[e].x = xe_scale127_small(vre[rp + e]);
                                    w[e].y = xe_scale127_small(vim[rp + e]);
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 2; e++) {
                                    // int32 wrap-around can only have happened for re == +2^31 (every sample -128 over 65536 frames)
                                    const double dre = (vre[rp + e] == (int)0x80000000) ? 2147483648.0 : (double)vre[rp + e];
                                    w[e].x = (float)(dre * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
                                    w[e].y = (float)((double)vim[rp + e] * a.kd * a.kd);
                                }
                            }
                            const float sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;  // what the neighbour stores of this lane's values
                            const float gx = __shfl_xor(sx, 1), gy = __shfl_xor(sy, 1);
                            const int i = 4 * g + rp + (odd ? 1 : 0), j0 = r & ~1;  // this lane stores columns j0, j0 + 1 of row i of the tile pair
                            const int s1 = bi * 16 + i;
                            c32 *dst = out_w + (size_t)f * nb + (s1 * (s1 + 1) / 2 + bj * 16 + j0);
                            const c32 first = odd ? c32{gx, gy} : w[0], second = odd ? w[1] : c32{gx, gy};
                            if ((a.dbg & 8192) && !(first.x == 1.25e-30f && second.y == 3.5e-31f)) continue;  // (tuning aid: the arithmetic without the stores)
                            if (bi != bj || j0 + 1 <= i) {
                                typedef float v4f __attribute__((ext_vector_type(4)));
                                const v4f q4 = (v4f){first.x, first.y, second.x, second.y};
                                __builtin_memcpy((void *)dst, &q4, 16);  // (plain stores: nontemporal ones measured 6 % slower per launch)
                            } else if (j0 <= i) {
                                *dst = first;
                            }
                        }
                        continue;
                    }
                }
                if constexpr (SPLIT) {
                    if (a.compact == 2) {
                        constexpr int OD = NTT * (NTT - 1) / 2;
                        unsigned char *blk = (unsigned char *)part_w + ((size_t)q * a.F + f) * pk_block_bytes(NTT);
                        if (bi == bj) {
                            v4i comb;
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) comb[reg] = (4 * g + reg >= r) ? vre[reg] : vim[reg];
                            comb = pk_bias(comb);
                            unsigned char *d = blk + OD * 1536 + bi * 768;
                            __builtin_nontemporal_store((v2i){(int)pk_lo(comb[0], comb[1]), (int)pk_lo(comb[2], comb[3])}, (v2i *)d + lane);
                            __builtin_nontemporal_store((int)pk_hi(comb), (int *)(d + 512) + lane);
                        } else {
                            const v4i wr = pk_bias(vre), wi = pk_bias(vim);
                            unsigned char *d = blk + (bi * (bi - 1) / 2 + bj) * 1536;
                            __builtin_nontemporal_store((v4i){(int)pk_lo(wr[0], wr[1]), (int)pk_lo(wr[2], wr[3]), (int)pk_lo(wi[0], wi[1]), (int)pk_lo(wi[2], wi[3])},
                                                        (v4i *)d + lane);
                            __builtin_nontemporal_store((v2i){(int)pk_hi(wr), (int)pk_hi(wi)}, (v2i *)(d + 1024) + lane);
                        }
                    } else if (a.compact) {
                        // A diagonal tile pair needs re (symmetric) and im (antisymmetric, zero diagonal) of ONE triangle: both go into one
                        // 16 x 16 record, re on and below the diagonal, im above it (im[i][j] = -im[j][i] is rebuilt by the reduction).
                        // 2 NP - NTT records of 1 KiB per channel and time range instead of 2 NP: a fifth less partial-sum traffic at 64 rows.
                        v4i *dst = part_w + (((size_t)q * a.F + f) * (2 * NP - NTT) + (2 * p - bi)) * 64 + lane;
                        if (bi == bj) {
                            v4i comb;
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) comb[reg] = (4 * g + reg >= r) ? vre[reg] : vim[reg];
                            __builtin_nontemporal_store(comb, dst);
                        } else {
                            __builtin_nontemporal_store(vre, dst);
                            __builtin_nontemporal_store(vim, dst + 64);
                        }
                    } else {
                        v4i *dst = part_w + ((((size_t)q * a.F + f) * NP + p) * 2) * 64 + lane;
                        __builtin_nontemporal_store(vre, dst);
                        __builtin_nontemporal_store(vim, dst + 64);
                    }
                } else {
                    if (f >= a.Fout) continue;
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int r1 = bi * 16 + 4 * g + reg, r2 = bj * 16 + r;
                        if (r1 >= A || r2 >= A) continue;
                        const int s1 = r1 / NPOL, p1 = r1 % NPOL, s2 = r2 / NPOL, p2 = r2 % NPOL;
                        if (s1 < s2) continue;
                        const size_t o = ((size_t)f * nb + (s1 * (s1 + 1) / 2 + s2)) * np2 + p1 * NPOL + p2;
                        // int32 wrap-around can only have happened for re == +2^31 (every sample -128 over 65536 frames)
                        const double dre = (vre[reg] == (int)0x80000000) ? 2147483648.0 : (double)vre[reg];
                        c32 v;
                        v.x = (float)(dre * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
                        v.y = (float)((double)vim[reg] * a.kd * a.kd);
                        if (a.accumulate) { v.x += out_w[o].x; v.y += out_w[o].y; }
                        out_w[o] = v;
                    }
                }
            }
        }
    }
    if (a.ts) {
        stamp(a, 3);
        if (tid == 0 && unit_idx >= 1 && unit_idx < 4) a.ts[(size_t)blockIdx.x * 16 + 9 + 2 * unit_idx] = wall_clock64();  // (later units: stores issued)
        if (last_unit) {  // (a barrier after an earlier unit would pair with the other ping-pong group's step barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stamp(a, 4);
        }
    }
    }
    };
    if constexpr (PP) {
        // Ping-pong: the two waves of a SIMD (w and w + 4) run half a step apart.  Waves 0-3 read block j and then multiply it; waves 4-7
        // first multiply block j - 1 (operands kept across the barrier) and then read block j: one wave's matrix products run beside the
        // other's LDS reads and byte transposes instead of both doing the same thing at the same time.  ONE barrier per step: behind it
        // block j has landed and nobody reads block j - 1 any more, whose slots take block j + 1.
        // Both groups run the same body -- read block j, multiply block j -- and differ only in where the step's barrier sits: before the
        // read (group 0) or between read and multiply (group 1, which therefore multiplies block j while group 0 already reads j + 1).
        // (waves w and w + 4 share a SIMD: a workgroup's waves are placed round robin over the four SIMDs -- checked with HW_REG_HW_ID.)
        // The second group gets static priority: the first group's products (older waves win the arbitration) would otherwise hold back
        // the last few products of the second group's phase for a whole phase, and everybody waits for them at the barrier.
        const int grp = wave >> 2;
        if (grp == 1 && !(a.dbg & 32)) __builtin_amdgcn_s_setprio(1);
        // Persistent form (a.items > 1): the blocks of all units of this workgroup form ONE stream through the ring.  The request for the
        // first block of unit k + 1 goes out behind the barrier of unit k's last block, the second one behind the next barrier, so the
        // first-load latency of a unit (8-14 us from HBM) runs under the previous unit's last products and its matrix stores.
        const int items = SPLIT ? 1 : a.items;  // (time ranges: one unit per workgroup, their workgroups must all be resident)
        const int total = items * a.steps;
        auto top = [&](int gb) {  // block gb (counted over all units) has landed for every wave, nobody reads block gb - 1 any more: its slots take block gb + 1
            if (gb == 0 && total > 1) {
                if constexpr (PER_STEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (gb == 0) stamp(a, 1);
            if (gb + 1 < total) {
                const int jb = (gb + 1) % a.steps;  // the block to request, counted inside its unit
                if (jb == 0) {  // the next unit of this workgroup: the same slice of a later window (see FuArgs::items)
#pragma unroll
                    for (int sh = 0; sh < NSH; sh++) src_lane[sh] += a.unit_in_step;
                    if (pf_src) pf_src += a.unit_in_step;
                }
                if (a.pf_dist > 0) {  // (the slow lines' own requests for blocks 0 and 1 of the first unit go out at launch: nothing to gain there)
                    if (gb == 0) for (int b2 = 2; b2 < a.pf_dist && b2 < a.steps; b2++) prefetch_slow(b2);
                    if (jb + a.pf_dist - 1 < a.steps) prefetch_slow(jb + a.pf_dist - 1);
                }
                if (gb >= 1) {
                    issue_stage(2 * jb, 2 * gb + 2);
                    issue_stage(2 * jb + 1, 2 * gb + 3);
                }
            }
        };
        int gb = 0;  // blocks counted over all units of this workgroup
        for (int unit = 0; unit < items; unit++) {
#pragma unroll
            for (int c = 0; c < CPW; c++) {
#pragma unroll
                for (int p = 0; p < NP; p++) re[c][p] = im[c][p] = (v4i){0, 0, 0, 0};
#pragma unroll
                for (int rt = 0; rt < NTT; rt++) rs[c][rt] = 0u;
            }
            for (int j = 0; j < a.steps; j++, gb++) {
                long I[CPW][NTT], Q[CPW][NTT];
                if (grp == 0 || gb == 0) top(gb);
                if (!(a.dbg & (1 | 128))) read_block(gb, I, Q);
                else if (a.dbg & 128) {
#pragma unroll
                    for (int c = 0; c < CPW; c++)
#pragma unroll
                        for (int rt = 0; rt < NTT; rt++) { I[c][rt] = rs[c][rt]; Q[c][rt] = gb; }
                }
                if (grp == 1 && gb + 1 < total) top(gb + 1);
                if (!(a.dbg & (1 | 64))) mfma32(I, Q);
                else if (a.dbg & 64) {
#pragma unroll
                    for (int c = 0; c < CPW; c++)
#pragma unroll
                        for (int rt = 0; rt < NTT; rt++) rs[c][rt] += (unsigned)I[c][rt] ^ (unsigned)Q[c][rt];
                }
            }
            finish_unit(blockIdx.x + unit * gridDim.x, unit, unit + 1 == items);
        }
    } else {
        for (int j = 0; j < a.steps; j++) {
            long I[CPW][NTT], Q[CPW][NTT];
            load_block(j, I, Q);
            if (!(a.dbg & 1)) mfma32(I, Q);
        }
        finish_unit(blockIdx.x, 0, true);
    }
}

// sum of the time ranges' partial matrices (exact, int64), scale, scatter into the reference's output order
// (the default form of the reduction; MI355_XE_INKERNEL_REDUCE=1 selects the tail of k_xe_i8_fused instead)
template <int NPOL, int TS>  // TS > 0: the number of time ranges at compile time (all loads of an item issued back to back)
__global__ __launch_bounds__(256) void k_xe_i8_reduce(const v4i *__restrict__ part, c32 *__restrict__ out, int N, int F, int Fout, int NP, int NTT,
                                                      int tsplit_rt, double kd, int accumulate, int compact, int ipw, size_t part_window, size_t out_window)
{
    const int tsplit = TS > 0 ? TS : tsplit_rt;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    part += (size_t)blockIdx.y * part_window;  // integration window of the batched form
    out += (size_t)blockIdx.y * out_window;
    // ipw consecutive items per wave, one after the other: the grid is sized so that every wave is resident from the start
    for (int it = 0; it < ipw; it++) {
    size_t item = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ipw + it;  // (f, p)
    const bool live = item < (size_t)Fout * NP;                 // (whole waves; a dead wave still takes part in the shuffles below)
    if (!live) item = 0;
    const int f = (int)(item / NP), p = (int)(item % NP);
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= p) bi++;
    const int bj = p - bi * (bi + 1) / 2;
    const bool diag = compact && bi == bj;
    const int rpc = compact ? 2 * NP - NTT : 2 * NP, rec = compact ? 2 * p - bi : 2 * p;
    long sre[4] = {0, 0, 0, 0}, sim[4] = {0, 0, 0, 0};
    if (compact == 2) {
        const int OD = NTT * (NTT - 1) / 2;
        const size_t bb = (size_t)pk_block_bytes(NTT);
        const unsigned char *base = (const unsigned char *)part + (size_t)f * bb + (diag ? OD * 1536 + bi * 768 : (bi * (bi - 1) / 2 + bj) * 1536);
        constexpr int QB = TS > 0 ? TS : 4;
        for (int q0 = 0; q0 < tsplit; q0 += QB) {
            v4i lo[QB];
            v2i hi[QB];
#pragma unroll
            for (int u = 0; u < QB; u++) {
                const unsigned char *d = base + (size_t)(q0 + u) * F * bb;
                if (q0 + u >= tsplit) { lo[u] = (v4i){0, 0, 0, 0}; hi[u] = (v2i){0, 0}; continue; }
                if (diag) {
                    const v2i l = __builtin_nontemporal_load((const v2i *)d + lane);
                    lo[u] = (v4i){l[0], l[1], 0, 0};
                    hi[u] = (v2i){__builtin_nontemporal_load((const int *)(d + 512) + lane), 0};
                } else {
                    lo[u] = __builtin_nontemporal_load((const v4i *)d + lane);
                    hi[u] = __builtin_nontemporal_load((const v2i *)(d + 1024) + lane);
                }
            }
#pragma unroll
            for (int u = 0; u < QB; u++) {
                if (q0 + u >= tsplit) continue;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    sre[k] += pk_get((unsigned)lo[u][k >> 1], (unsigned)hi[u][0], k);
                    if (!diag) sim[k] += pk_get((unsigned)lo[u][2 + (k >> 1)], (unsigned)hi[u][1], k);
                }
            }
        }
    } else if constexpr (TS > 0) {
        v4i a[TS], b[TS];
#pragma unroll
        for (int q = 0; q < TS; q++) {
            const v4i *src = part + (((size_t)q * F + f) * rpc + rec) * 64 + lane;
            a[q] = __builtin_nontemporal_load(src);  // (plain loads / plain partial stores measure the same: 61.0 us either way)
            b[q] = diag ? (v4i){0, 0, 0, 0} : __builtin_nontemporal_load(src + 64);
        }
#pragma unroll
        for (int q = 0; q < TS; q++)
#pragma unroll
            for (int k = 0; k < 4; k++) { sre[k] += a[q][k]; sim[k] += b[q][k]; }
    } else {
        for (int q = 0; q < tsplit; q++) {
            const v4i *src = part + (((size_t)q * F + f) * rpc + rec) * 64 + lane;
            const v4i a = __builtin_nontemporal_load(src);
#pragma unroll
            for (int k = 0; k < 4; k++) sre[k] += a[k];
            if (!diag) {
                const v4i b = __builtin_nontemporal_load(src + 64);
#pragma unroll
                for (int k = 0; k < 4; k++) sim[k] += b[k];
            }
        }
    }
    if (diag) {
        // sre holds the combined tile C: C[i][j] = re[i][j] (i >= j), im[i][j] (i < j), at i = 4 g + reg, j = r.  The transposed entry
        // C[j][i] sits in lane 16 (j / 4) + i, register j % 4.
        const long cc[4] = {sre[0], sre[1], sre[2], sre[3]};  // (the loop below overwrites sre while other lanes still read C)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = 4 * g + reg, src_lane = 16 * (r >> 2) + i;
            long t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int lo = __shfl((int)(unsigned)(unsigned long)cc[k], src_lane), hi = __shfl((int)((unsigned long)cc[k] >> 32), src_lane);
                t[k] = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
            }
            const long tr = (r & 3) == 0 ? t[0] : (r & 3) == 1 ? t[1] : (r & 3) == 2 ? t[2] : t[3];
            if (i > r) sim[reg] = -tr;                         // im[i][j] = -im[j][i]
            else if (i == r) sim[reg] = 0;
            else { sim[reg] = cc[reg]; sre[reg] = tr; }        // above the diagonal (same-station polarisation products): re[i][j] = re[j][i]
        }
    }
    if (!live) continue;
    const int nb = N * (N + 1) / 2, np2 = NPOL * NPOL, A = N * NPOL;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int r1 = bi * 16 + 4 * g + reg, r2 = bj * 16 + r;
        if (r1 >= A || r2 >= A) continue;
        const int s1 = r1 / NPOL, p1 = r1 % NPOL, s2 = r2 / NPOL, p2 = r2 % NPOL;
        if (s1 < s2) continue;
        const size_t o = ((size_t)f * nb + (s1 * (s1 + 1) / 2 + s2)) * np2 + p1 * NPOL + p2;
        c32 v;
        v.x = (float)((double)sre[reg] * kd * kd);
        v.y = (float)((double)sim[reg] * kd * kd);
        if (accumulate) { v.x += out[o].x; v.y += out[o].y; }
        out[o] = v;
    }
    }
}

template <int NPOL, int NTT, bool SPLIT, bool PP, bool RS> int launch_fused_s(const XeFusedPlan &p, const FuArgs &a, hipStream_t st)
{
    constexpr int NSH = (NTT * 16 / NPOL > 32) ? 2 : 1;
    constexpr int lds_bytes = kRing * (kStageT * NSH * kChunk + 16) + 1024;  // (+ the scratch the slow lines' early touches land in)
    // (per instantiation AND per device: a function's attributes belong to the device that is current when they are set -- the single-process
    // multi-GPU form launches this kernel on every device of the handle; a second thread setting it again is harmless)
    static std::atomic<unsigned long long> attr_devs{0};
    {
        int dev = 0;
        MI355_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_i8_fused<NPOL, NTT, SPLIT, PP, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            attr_devs.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const int nint = a.nint_launch;
    hipLaunchKernelGGL((k_xe_i8_fused<NPOL, NTT, SPLIT, PP, RS>), dim3((unsigned)(p.units * p.tsplit * nint / a.items)), dim3(kThreads), lds_bytes, st, a);
    MI355_HIP(hipGetLastError());
    if (SPLIT && !RS) {
        const int NP = NTT * (NTT + 1) / 2;
        const size_t items = (size_t)a.Fout * NP;
        int ipw = 1;  // items per wave: as many as it takes for all waves to be resident together (32 per CU)
        if (const char *e = getenv("MI355_XE_REDUCE_IPW")) ipw = atoi(e) > 0 ? atoi(e) : 1;
        else while (items * nint > (size_t)ipw * 8192 && ipw < 8) ipw++;
        const unsigned grid = (unsigned)((items + (size_t)4 * ipw - 1) / ((size_t)4 * ipw));
        if (p.tsplit == 4)
            hipLaunchKernelGGL((k_xe_i8_reduce<NPOL, 4>), dim3(grid, nint), dim3(256), 0, st, (const v4i *)a.part, a.out, a.N, a.F,
                               a.Fout, NP, NTT, p.tsplit, a.kd, a.accumulate, a.compact, ipw, a.part_window, a.out_window);
        else
            hipLaunchKernelGGL((k_xe_i8_reduce<NPOL, 0>), dim3(grid, nint), dim3(256), 0, st, (const v4i *)a.part, a.out, a.N, a.F,
                               a.Fout, NP, NTT, p.tsplit, a.kd, a.accumulate, a.compact, ipw, a.part_window, a.out_window);
        MI355_HIP(hipGetLastError());
    }
    return MI355_OK;
}
template <int NPOL, int NTT> int launch_fused(const XeFusedPlan &p, const FuArgs &a, hipStream_t st)
{
    // ping-pong schedule (the two waves of a SIMD half a step apart) whenever a time range has at least four blocks
    const bool pp = a.steps >= 4 && !getenv("MI355_XE_NO_PINGPONG");
    if constexpr (NTT == 4) {
        if (a.rs) return pp ? launch_fused_s<NPOL, NTT, true, true, true>(p, a, st) : launch_fused_s<NPOL, NTT, true, false, true>(p, a, st);
    }
    if (p.tsplit > 1) return pp ? launch_fused_s<NPOL, NTT, true, true, false>(p, a, st) : launch_fused_s<NPOL, NTT, true, false, false>(p, a, st);
    return pp ? launch_fused_s<NPOL, NTT, false, true, false>(p, a, st) : launch_fused_s<NPOL, NTT, false, false, false>(p, a, st);
}

template <int NPOL> int launch_by_tiles(const XeFusedPlan &p, const FuArgs &a, hipStream_t st)
{
    if (p.ntt == 1) return launch_fused<NPOL, 1>(p, a, st);
    if (p.ntt == 2) return launch_fused<NPOL, 2>(p, a, st);
    return launch_fused<NPOL, 4>(p, a, st);
}

// tuning aid (MI355_XE_TS=1): one synchronous launch with per-workgroup phase stamps, summary on stderr
int launch_with_stamps(const XeFusedPlan &p, FuArgs a, hipStream_t st)
{
    const int wgs = p.units * p.tsplit * a.nint_launch / a.items;
    static unsigned long long *d_ts = nullptr;
    static int cap = 0;
    if (cap < wgs) {
        if (d_ts) (void)hipFree(d_ts);
        MI355_HIP(hipMalloc(&d_ts, (size_t)wgs * 128));
        cap = wgs;
    }
    MI355_HIP(hipMemsetAsync(d_ts, 0, (size_t)wgs * 128, st));
    a.ts = d_ts;
    const int rc = p.npol == 1 ? launch_by_tiles<1>(p, a, st) : launch_by_tiles<2>(p, a, st);
    if (rc != MI355_OK) return rc;
    MI355_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)wgs * 16);
    MI355_HIP(hipMemcpy(h.data(), d_ts, (size_t)wgs * 128, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < wgs; b++) t0 = std::min(t0, h[(size_t)b * 16]);
    static const char *names[7] = {"start", "first data", "loop end", "stores issued", "stores done", "all ranges in", "end"};
    fprintf(stderr, "[xe stamps] %d workgroups, us after the first start (min / median / max):\n", wgs);
    for (int k = 0; k < 7; k++) {
        std::vector<double> v;
        for (int b = 0; b < wgs; b++) if (h[(size_t)b * 16 + k]) v.push_back((double)(h[(size_t)b * 16 + k] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        fprintf(stderr, "  %-14s %7.2f %7.2f %7.2f\n", names[k], v.front(), v[v.size() / 2], v.back());
    }
    if (const char *path = getenv("MI355_XE_TS_FILE")) {  // one line per workgroup: block, XCD, then the five stamps in us
        if (FILE *f = fopen(path, "w")) {
            for (int b = 0; b < wgs; b++) {
                fprintf(f, "%d %d", b, (int)h[(size_t)b * 16 + 7]);
                for (int k = 0; k < 7; k++) fprintf(f, " %.2f", h[(size_t)b * 16 + k] ? (double)(h[(size_t)b * 16 + k] - t0) * 0.01 : -1.0);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    for (int u = 1; u < a.items && u < 4; u++)
        for (int k = 0; k < 2; k++) {
            std::vector<double> v;
            for (int b = 0; b < wgs; b++) if (h[(size_t)b * 16 + 8 + 2 * u + k]) v.push_back((double)(h[(size_t)b * 16 + 8 + 2 * u + k] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "  unit %d %-13s %7.2f %7.2f %7.2f\n", u + 1, k ? "stores issued" : "loop end", v.front(), v[v.size() / 2], v.back());
        }
    int bad = 0;
    for (int b = 0; b < wgs; b++) bad += ((int)h[(size_t)b * 16 + 7] != (b & 7));
    fprintf(stderr, "  workgroups not on XCD blockIdx %% 8: %d\n", bad);
    return MI355_OK;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_xe_scale_selftest(unsigned long long *bad)
{
    const double kd = 0.007874015748031496063;
    unsigned long long n = 0;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x - (1L << 24); v <= (1L << 24); v += (long)gridDim.x * 256) {
        const int S = (int)v;
        const float want = (float)((double)S * kd * kd), got = xe_scale127_small(S);
        n += __float_as_uint(want) != __float_as_uint(got) ? 1u : 0u;
    }
    if (n) atomicAdd(bad, n);
}
}  // namespace

extern "C" int mi355_xengine_selftest_scale(mi355_ctx *ctx, long long *mismatches)
{
    MI355_REQUIRE(ctx && mismatches, "NULL argument");
    MI355_HIP(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr, h = 0;
    MI355_HIP(hipMalloc(&d, 8));
    hipError_t e = hipMemset(d, 0, 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xe_scale_selftest, dim3(2048), dim3(256), 0, 0, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    MI355_HIP(e);
    *mismatches = (long long)h;
    return MI355_OK;
}

XeFusedPlan mi355_xe_fused_plan(int N, int F, int Fout, int npol, int T, int num_cus, int nint)
{
    XeFusedPlan p;
    (void)Fout;
    const int A = N * npol;
    // rows of whole 16-byte pieces; a row that ends inside a 128-byte line is treated as the whole line (the missing pieces read as zeros, the
    // channels they would hold have no output): 1000 channels run as 1024
    const size_t row_real = (size_t)F * npol * 2;
    if (A > 64 || A < 1 || row_real % 16 != 0 || getenv("MI355_XE_NO_FUSED")) return p;
    if (row_real % 128 != 0 && getenv("MI355_XE_FUSED_WHOLE_LINES")) return p;  // (tuning / test switch: such rows through the two-kernel path)
    const size_t row_bytes = (row_real + 127) / 128 * 128;
    p.row_stride = (int)row_real;
    F = (int)(row_bytes / (npol * 2));  // channels of the workspace indexing
    if (T % 32 != 0 && getenv("MI355_XE_FUSED_WHOLE_KBLOCKS")) return p;  // (tuning / test switch: ragged integrations through the two-kernel path)
    T = (T + 31) / 32 * 32;  // the frames past the end of the integration read as zeros (xe_zero_row)
    p.npol = npol;
    const int nt = (A + 15) / 16;
    p.ntt = nt <= 1 ? 1 : nt <= 2 ? 2 : 4;
    p.units = (int)(row_bytes / 32);
    // time split: fill the device (one workgroup per CU), whole K blocks per range
    const int cus = num_cus > 0 ? num_cus : 256;
    p.cus = cus;
    int s = 1;
    if (const char *e = getenv("MI355_XE_TSPLIT")) s = atoi(e) > 0 ? atoi(e) : 1;
    else
        // (not below four K blocks per range: 128 / 256 channels x 1024 frames measure 27.4 / 31.3 us with 8 ranges, 28.6 / 33.9 with 16 --
        // the per-rank problem of the 8-GPU sharded form, where two dispatches and the first-load latency dominate)
        while ((long)p.units * s * (nint > 0 ? nint : 1) < cus && s < 16 && T / (32 * s * 2) >= 4) s *= 2;
    while (s > 1 && (T % (32 * s) != 0)) s /= 2;
    p.tsplit = s;
    const int NP = p.ntt * (p.ntt + 1) / 2;
    p.part_per_window = s > 1 ? (size_t)s * F * NP * 2 * 1024 : 0;
    p.flag_offset = p.part_per_window * (nint > 0 ? nint : 1);
    // (behind the partial sums: two banks of one 8-byte arrival word per slice and window for the in-launch reduction)
    p.part_bytes = s > 1 ? p.flag_offset + (((size_t)2 * p.units * (nint > 0 ? nint : 1) * 8 + 255) & ~(size_t)255) : 0;
    p.ok = true;
    return p;
}

int mi355_xe_fused_launch(const XeFusedPlan &p, const void *in, void *out, void *part, int N, int F, int Fout, int T, double kd, int accumulate,
                          hipStream_t st, int stations_per_group, unsigned *epoch, int nint)
{
    // whole 128-byte lines per request, the row-tile pairs split over four workgroups (xengine_lines.hip): where there are enough units without
    // time ranges; needs neither the partial-sum workspace nor the arrival words
    if (p.tsplit == 1 && p.npol == 1 && p.row_stride == F * 2 && ((size_t)in & 15) == 0 &&
        mi355_xe_lines_ok(N, F, Fout, p.npol, T, stations_per_group, accumulate, nint, p.cus))
        return mi355_xe_lines_launch(in, out, N, F, Fout, T, kd, st, stations_per_group, nint, p.cus);
    // the same kernel with time ranges where the units alone do not fill the device (one or two windows of config 5): the tail combines the ranges
    // inside the launch through the plan's workspace (same size rule: the plan's tsplit is this one's)
    if (p.tsplit > 1 && p.npol == 1 && p.row_stride == F * 2 && ((size_t)in & 15) == 0 && epoch && part &&
        mi355_xe_lines_split(N, F, Fout, p.npol, T, stations_per_group, accumulate, nint, p.cus) == p.tsplit &&
        (size_t)(nint > 0 ? nint : 1) * (F / 64) * 4 * 4 * p.tsplit * 65536 <= p.flag_offset)
        return mi355_xe_lines_launch(in, out, N, F, Fout, T, kd, st, stations_per_group, nint, p.cus, p.tsplit, part, p.flag_offset, epoch);
    FuArgs a;
    a.in = (const unsigned char *)in;
    a.part = (v4i *)part;
    a.flags = p.tsplit > 1 ? (int *)((char *)part + p.flag_offset) : nullptr;
    a.epoch = 1;
    // The four time ranges of a 64-row slice are combined inside the launch (reduce-scatter tail of k_xe_i8_fused) when every workgroup of
    // the launch is resident at once (one per CU) and a range fits the 24-bit planes; MI355_XE_INKERNEL_REDUCE=0 keeps the second kernel.
    // (Read per launch: a tuning / test switch.  The workspace's epoch advances only with launches that use the arrival words, so
    // switching between the two forms mid-process leaves them consistent.)
    const int Tp = (T + 31) / 32 * 32;  // whole K blocks
    const char *rs_env = getenv("MI355_XE_INKERNEL_REDUCE");
    a.rs = (epoch && p.tsplit == 4 && p.ntt == 4 && Tp / 4 <= 256 && (long)p.units * 4 * (nint > 0 ? nint : 1) <= p.cus && !(rs_env && atoi(rs_env) == 0)) ? 1 : 0;
    // (the workspace's epoch is committed only once the kernel is enqueued: a launch that fails before that leaves the words and the count as they were)
    if (a.rs) a.epoch = *epoch + 1u;
    a.flag_bank = (unsigned)(p.units * (nint > 0 ? nint : 1));
    a.k127 = (kd == 0.007874015748031496063 && !getenv("MI355_XE_SCALE_F64")) ? 1 : 0;
    {
        const int us = 10 + 6 * (Tp / (32 * p.tsplit));
        a.wait_ticks = (unsigned)(us < 20 ? 20 : us > 100 ? 100 : us) * 100u;
        if (const char *e = getenv("MI355_XE_WAIT_US")) a.wait_ticks = (unsigned)atoi(e) * 100u;
    }
    a.compact = !getenv("MI355_XE_NO_COMPACT") ? 1 : 0;
    // 24-bit planes: time ranges of at most 256 steps, at least two row tiles (a lone diagonal record has nothing to pair with)
    if (a.compact && p.tsplit > 1 && Tp / p.tsplit <= 256 && p.ntt >= 2 && !getenv("MI355_XE_NO_PACK24")) a.compact = 2;
    a.out = (c32 *)out;
    (void)F;
    a.N = N; a.F = p.units * 32 / (p.npol * 2); a.Fout = Fout; a.T = T;
    a.ng = (stations_per_group > 0 && stations_per_group < N) ? stations_per_group : N;
    {
        const size_t row_bytes = (size_t)p.row_stride;
        a.row_stride = p.row_stride;
        a.nint_launch = nint > 0 ? nint : 1;
        a.wgs = p.units * p.tsplit;
        // reference layout: [window][t][station]; group-major: [group][window][t][station in group]
        a.in_window = (size_t)T * a.ng * row_bytes;
        a.in_group = (size_t)a.nint_launch * T * a.ng * row_bytes;
        a.out_window = (size_t)Fout * ((size_t)N * (N + 1) / 2) * p.npol * p.npol;
        a.part_window = p.part_per_window / 16;
    }
    a.nlines = p.units / 4;
    a.tsplit = p.tsplit;
    a.steps = Tp / (32 * p.tsplit);
    a.pinned = (((long)a.nlines * p.tsplit * (nint > 0 ? nint : 1)) % 8 == 0) ? 1 : 0;
    a.accumulate = accumulate;
    a.kd = kd;
    const int dbg = getenv("MI355_XE_DBG") ? atoi(getenv("MI355_XE_DBG")) : 0;
    a.dbg = dbg;
    a.ts = nullptr;
    // Early touches of the slow lines (prefetch_slow): rows of 8, 16 or 32 whole lines, whole K blocks, the ping-pong schedule.  Which lines:
    // address bits 7..9 == 3 (1.75 x the latency from HBM), and bits 7..10 == 15 (1.15 x) where a line keeps bit 10 from row to row.
    // MI355_XE_PF: distance in K blocks (default 2), + 16: without the 1.15 x line; MI355_XE_NO_PREFETCH: off.
    a.pf_mask = 0;
    a.pf_dist = 0;
    a.pf_lines = 0;
    if ((a.nlines == 8 || a.nlines == 16 || a.nlines == 32) && p.row_stride == a.nlines * 128 && ((size_t)in & 127) == 0 && T % 32 == 0 && a.steps >= 4 &&
        !getenv("MI355_XE_NO_PREFETCH")) {
        const int key = (int)(((size_t)in >> 7) & 15), tune = getenv("MI355_XE_PF") ? atoi(getenv("MI355_XE_PF")) : 2;
        for (int l = 0; l < a.nlines; l++) {
            const int c = (l + key) & 15;
            if ((c & 7) == 3 || (c == 15 && a.nlines >= 16 && !(tune & 16))) a.pf_mask |= 1u << l;
        }
        a.pf_dist = (tune & 15) >= 2 ? (tune & 15) : 2;
        int n = 0;
        for (int l = 0; l < a.nlines; l++)
            if ((a.pf_mask >> l) & 1) {
                if (n < 8) a.pf_lines |= (unsigned long long)l << (8 * n);
                n++;
            }
        if (n > 8) { a.pf_mask = 0; a.pf_dist = 0; }
    }
    a.slow_first = -1;
    if (p.tsplit == 1 && a.nint_launch > 1 && a.nlines % 8 == 0 && p.row_stride % 1024 == 0 && ((size_t)in & 127) == 0 &&
        (long)p.units * a.nint_launch > p.cus && (a.pf_dist == 0 || getenv("MI355_XE_SLOW_FIRST")) && !getenv("MI355_XE_NO_SLOW_FIRST"))
        a.slow_first = (int)(((size_t)in >> 7) & 7);
    // Persistent form: no time ranges, more units than CUs, the ping-pong schedule -- every workgroup runs total / grid units, one after the
    // other, the next unit's first K blocks requested under the current unit's matrix stores (MI355_XE_NO_PERSIST: one unit per workgroup)
    a.items = 1;
    a.unit_in_step = 0;
    {
        const long total = (long)p.units * p.tsplit * a.nint_launch;
        const int per = (int)((total + p.cus - 1) / p.cus);
        // workgroup b's unit k is unit b + k * grid of the plain form: with the pinned map (xcd = b % 8, sector = (b / 8) % 4, line and window from
        // b / 32) that is the same slice of window + k * grid / (4 * nlines) when grid / 4 is a multiple of the lines per row
        if (p.tsplit == 1 && per > 1 && total % per == 0 && a.steps >= 4 && a.pinned && a.slow_first < 0 && a.ng == N && !(dbg & ((7 << 20) | (3 << 12))) &&
            ((total / per) % 32) == 0 && ((total / per / 4) % a.nlines) == 0 && !getenv("MI355_XE_NO_PINGPONG") && !getenv("MI355_XE_NO_PERSIST")) {
            a.items = per;
            a.unit_in_step = (size_t)(total / per / 4 / a.nlines) * a.in_window;
        }
    }
    // MI355_XE_FAIL_LAUNCH (test switch): fail where a bad stream handle or an exhausted device would -- nothing enqueued, an error returned
    if (getenv("MI355_XE_FAIL_LAUNCH")) {
        mi355_set_error("fused X-engine launch failed (MI355_XE_FAIL_LAUNCH)");
        return MI355_ERR_HIP;
    }
    mi355_xe_route_set(a.rs || p.tsplit == 1 ? "k_xe_i8_fused" : "k_xe_i8_fused+k_xe_i8_reduce", a.nint_launch, p.units * p.tsplit * a.nint_launch / a.items, a.items,
                       p.tsplit, a.rs, a.pf_dist, 0);
    int rc;
    if (getenv("MI355_XE_TS")) rc = launch_with_stamps(p, a, st);
    else rc = p.npol == 1 ? launch_by_tiles<1>(p, a, st) : launch_by_tiles<2>(p, a, st);
    if (rc == MI355_OK && a.rs) *epoch = a.epoch;
    return rc;
}
