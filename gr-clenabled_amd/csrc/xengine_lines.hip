// clXEngine, IChar (int8 I/Q) input, 64 stations, one polarisation: corner turn AND correlation in one pass over the input, with WHOLE 128-byte
// lines per request (round 5).  Reference behaviour: lib/clXEngine_impl.cc:708-817 (CharToComplex + XCorrelate kernels), :859-867 (IChar scale);
// input layout [t][station][chan]{I,Q} (:766-767,987-1061), output [chan][baseline] (:786-808).
//
// Why this shape.  k_xe_i8_fused (xengine_fused.hip) gives a workgroup a 32-byte column slice (16 channels) of every (t, station) row and all ten
// row-tile pairs; its loop is bound by the number of 128-byte-line REQUESTS a CU can have in flight, not by bytes: one request per row and workgroup,
// 4.19 M per BASELINE integration, 40 us from HBM.  tools/ubench/xe_colgroup_read.hip measures the alternative: a workgroup takes the WHOLE line
// (64 channels) of a row but only a GROUP of the ten row-tile pairs,
//     A = {00, 10, 11} stations 0-31      B = {22, 32, 33} stations 32-63      C = {20, 21} stations 0-47      D = {30, 31} stations 0-31, 48-63
// 2.5 x the rows, a quarter of the requests per row: 2.62 M requests, and the four workgroups of a line sit on one XCD (blockIdx % 8) and walk the
// same frames, so the re-reads can hit that XCD's L2.  Request stream alone: 33.6 against 55.8 us per window (eight windows per launch, from HBM).
// Registers: 64 channels x 16 accumulator registers per lane for every group --
//   off-diagonal pair: re += I_a I_b^T + Q_a Q_b^T,  im += Q_a I_b^T + I_a (~Q_b)^T + I_a 1^T   (-q = ~q + 1 exactly, -128 included)     8 registers
//   diagonal pair: everything into ONE accumulator C = re + im (4 registers): re is symmetric and im antisymmetric, so
//                  re[i][j] = (C[i][j] + C[j][i]) / 2,  im[i][j] = (C[i][j] - C[j][i]) / 2 exactly (|C| <= T * 2^16: T <= 16384)
// Workgroup = EIGHT waves, two per SIMD: wave w owns the w-th 16-byte piece of every line = 8 channels = 128 accumulation registers, addressed BY
// NAME from inline assembly (see ln_mm), + at most 128 vector registers.  (One wave per SIMD with 16 channels was built first: a wave's own
// VALU instructions do not overlap its own matrix products -- tools/ubench/mfma_i8_rate.hip: 17.5 + 8 k clocks per product with k v_perm behind it
// -- so the byte transposes and the products of a K block add up; two waves per SIMD overlap each other's.)
//
// Data path.  A K block is 32 frames.  The lines of one row tile (16 stations) and 16 frames = 256 lines = one sub-stage, 32 LDS-DMA instructions
// (global_load_lds_dwordx4), each the line of ONE station for 8 consecutive frames (lane = frame * 8 + 16-byte piece), landing as a 1 KiB chunk; chunk
// starts are 1040 bytes apart, so the sixteen stations' copies of a piece fall into sixteen different 16-byte bank groups (conflict-free ds_read_b128).
// Ring of four sub-stages (130 KiB): one being read, three in flight.  Per sub-stage a lane (station r, group g) pulls 4 frames x 16 bytes (its
// wave's 8 channels) into registers and byte-transposes them (v_perm) into the operands of v_mfma_i32_16x16x64_i8: operand = (I | Q) of one channel,
// 8 frames each, K = 64 -- ONE product is the whole real part of a tile pair and K block.  The order of the frames inside a K block differs
// from k_xe_i8_fused's (any order is fine as long as both operands use it).  Products are issued BETWEEN the transposes (the pair yy of the previous
// K block under the sub-stages of x, xx under the first of y, yx channel by channel as the second of y completes its operands), the four requests
// of the sub-stage three ahead one at a time between them as well.
// Pacing.  The four workgroups of a line re-read each other's lines from L2 only while they walk the same frames; groups C / D take six
// sub-stages per K block, A / B four, so A / B run ahead until their partners' lines have left the L2 again (read traffic 1.8 x the input).  Every
// workgroup publishes the half K blocks it has started (ln_progress) and one that is more than `pace` ahead of its slowest partner waits.  The words
// stay in the XCD's L2 (plain stores, L1-bypassing loads): published at agent scope they cross the fabric to the array's home, the launch time then
// follows where that home is (the first paced form: 360 us in one process, 400 in the next) -- now 347-362 us per eight windows in every process, reads
// 1.44 x the input.  A workgroup's k-th unit is k lines further on than its first (the slow address class, lines 3 and 11 of a row, 1.25 x the time
// from HBM, spread over twice the workgroups); any equal share of up to 64 units per workgroup (128 windows per launch: 34.4 us per window).
// Early touches.  The diagonal groups of the other lines -- which wait for their partners 13 % of the time anyway -- request 16 bytes of every row of
// the slow lines four K blocks ahead (one instruction per wave 0 ... 3 and K block): the slow lines' own workgroups then find them in the Infinity
// Cache.  bench.py: 4 / 8 windows per launch 42.1 / 38.7 -> 36.9 / 37.1 us per window, 16 and 32 unchanged (35.4 / 35.6).  Round 5 ran
// unpaced with the touches on (35.9 / 35.5 / 35.2 / 35.1 us per window on its boxes); round 6 found processes in which the unpaced form runs every
// launch at 37-41 us per window with 1.6 x the input read, and none for the form paced by 2: pacing by 2 half K blocks is the default again
// (35.0 / 34.7 / 34.5 / 34.1 us per window at 4 / 8 / 16 / 32 windows per launch; see mi355_xe_lines_launch).
// Time ranges (round 6, k_xe_i8_lines<true>): fewer units than compute units -- one window of BASELINE config 5, the reference's one-integration-per-call
// shape -- cut into 2 or 4 ranges per (window, line, pair group) team, the ranges' exact partial sums combined by the kernel's own tail (see ln_body).
#include "xengine_fused.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <type_traits>
#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
struct c32 { float x, y; };

constexpr int kLnWaves = 8, kLnCh = 8, kLnU = 4;  // eight waves (two per SIMD), eight channels = one 16-byte piece = four dword units per wave
constexpr int kLnThreads = kLnWaves * 64, kLnChunk = 1040, kLnSlot = 32 * kLnChunk, kLnRing = 4;
constexpr int kLnTile = 16 * 20 * 4;  // a wave's scratch for the transposed diagonal tile
constexpr int kLnPoll = kLnRing * kLnSlot + kLnWaves * kLnTile;  // 256 bytes: where the partners' progress words land (pacing)
constexpr int kLnTouch = kLnPoll + 256;  // 1 KiB: where the early touches of the slow lines land (never read)
constexpr int kLnRs = kLnTouch + 1024;   // off-diagonal groups: a wave's row sums, [channel][lane] (2 KiB per wave)
constexpr int kLnLds = kLnRs + kLnWaves * kLnCh * 64 * 4;

struct LnArgs {
    const unsigned char *in;
    c32 *out;
    int Fout, T, ncols, row_stride, ng;  // ncols: 128-byte lines per row; ng: stations per antenna group (64: the reference layout)
    int units, items, pinned;            // units of the launch = windows x lines x 4 groups; units per workgroup (persistent form)
    int steps;                           // K blocks (32 frames) per unit
    size_t in_window, in_group, out_window;
    double kd;
    int k127;
    int dbg;  // tuning aid (MI355_XE_DBG): 2 no matrix stores, 4 no DMA, 8 no pacing
    unsigned tag;   // launch number (pacing of the four workgroups of a line: ln_progress)
    int pace;       // half K blocks a workgroup may run ahead of the slowest of its three partners (0: no pacing)
    int rot, grid;  // line rotation per unit of a workgroup (0: none); workgroups of the launch
    // early touches of the slow lines (address bits 7..9 == 3: ~1.7 x the latency from HBM): the diagonal groups of the OTHER lines, which wait for their
    // partners anyway, request 16 bytes of every slow line's rows pf_dist K blocks ahead (the lines are in the Infinity Cache when their own workgroups ask)
    int pf_dist, pf_lg, pf_per, pf_items;  // 0: off; log2(slow lines per row); (row, slow line) items per K block and toucher; items per K block
    unsigned pf_mask, pf_lines;            // bit l: line l of a row is slow; the slow lines' numbers, one byte each
    int pub_local;  // progress words by plain stores, kept in the XCD's L2 (0: agent-scope stores, MI355_XE_LINES_PUB=0)
    // time ranges (k_xe_i8_lines<true>): a (window, line, pair group) TEAM of tsplit workgroups, one per time range, whose exact int32 partial matrices are
    // combined inside the launch (ln_tail).  part: the teams' inboxes; flags: two banks of flag_bank 8-byte arrival words, one per team:
    // {arrival count (high 32 bits) | launch tag << 4 | give-up bits (low)}; launch e counts in bank e % 2 and clears the other bank's word
    int npol;  // 2: k_xe_i8_lines<false, 2> -- 64 stations x two polarisations, a line = 32 channels x {X, Y}, eight pair groups per line
    int tsplit;
    unsigned char *part;
    unsigned long long *flags;
    unsigned epoch, flag_bank, wait_ticks;
    unsigned long long *ts;
};

// (asm volatile: the transposes stay where they are written, between the products they are interleaved with -- left to the scheduler they are
// hoisted and sunk across the products, the raw rows of two sub-stages live at once, and the allocator starts moving accumulators)
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel)
{
    unsigned d;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "v"(lo), "s"(sel));
    return d;
}

// 4 dwords (bytes b0..b3 of four frames) -> out[j] = byte j of each input dword
__device__ __forceinline__ void transpose4x4(unsigned i0, unsigned i1, unsigned i2, unsigned i3, unsigned (&out)[4])
{
    const unsigned t0 = perm(i1, i0, 0x05010400u), t1 = perm(i1, i0, 0x07030602u);
    const unsigned t2 = perm(i3, i2, 0x05010400u), t3 = perm(i3, i2, 0x07030602u);
    out[0] = perm(t2, t0, 0x05040100u);
    out[1] = perm(t2, t0, 0x07060302u);
    out[2] = perm(t3, t1, 0x05040100u);
    out[3] = perm(t3, t1, 0x07060302u);
}

// the IChar scale in single precision, bit for bit (float)((double)S * kd * kd) at kd = 1 / 127 for |S| < 2^24 (see xengine_fused.hip)
__device__ __forceinline__ float ln_scale127_small(int S)
{
    const float c = 6.2000123e-05f;  // fl(1 / 16129)
    const float sf = (float)S;
    const float q = sf * c;
    const float r = __builtin_fmaf(-q, 16129.0f, sf);
    return __builtin_fmaf(r, c, q);
}
__device__ __forceinline__ unsigned ln_mag_bits(int v) { return (unsigned)(v + 0x1000000); }
__device__ __forceinline__ bool ln_all_small(unsigned m) { return __builtin_amdgcn_ballot_w64((m >> 25) != 0u) == 0ull; }

// LEAN (the two-polarisation kernel only): m0 is written and NOT restored -- two scalar moves less per request, 32 per K block and wave; the scalar
// unit is shared by the CU's eight waves and the loop's scalar instructions are most of its issue slots.  m0 is a reserved register to the compiler,
// which neither tracks a write to it nor reads it in the code it generates for this file on gfx950 (LDS instructions need no m0 there; every other
// user of m0 here is an asm statement that sets it itself): tests/test_isa_lines.py checks the built object for that.
// The DIAGONAL groups of the one-polarisation kernels keep the save and restore, the per-front timers and the guarded requests on purpose (see
// STRAIGHT in ln_body): what matters there is that the four workgroups of a line walk together.  The two-polarisation kernel, whose eight groups
// are alike, gains 6 % (106.2 -> 100.0 us per integration).
template <bool LEAN> __device__ __forceinline__ void ln_dma16(const void *gsrc, unsigned lds_dst)
{
    if constexpr (LEAN) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
    } else {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    }
}


// partial-sum traffic between the workgroups of a team (time ranges): the hand-off recipe of cdna_hip_programming.md (Guideline 16): write-through (sc1)
// 16-byte stores, every storing wave drains them, one lane raises the team's count; the consumer polls that word relaxed and reads the pieces with
// sc1 loads.  (s_nop: a store of more than 8 bytes reads its data a cycle or two after it issues and the hazard pass does not look inside an asm.)
__device__ __forceinline__ void ln_st_sys(unsigned char *p, v4i v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
// loads AND their wait in one statement (an asm's outputs may be moved or spilled right behind it -- for a bare load, before the data is there)
__device__ __forceinline__ void ln_ld4(const unsigned char *p, v4i (&x)[4])
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
                 : "v"(p)
                 : "memory");
}
__device__ __forceinline__ void ln_ld12(const unsigned char *p0, const unsigned char *p1, const unsigned char *p2, v4i (&x)[3][4])
{
    asm volatile("global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %12, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %3, %12, off offset:3072 sc1\n\t"
                 "global_load_dwordx4 %4, %13, off sc1\n\tglobal_load_dwordx4 %5, %13, off offset:1024 sc1\n\tglobal_load_dwordx4 %6, %13, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %7, %13, off offset:3072 sc1\n\t"
                 "global_load_dwordx4 %8, %14, off sc1\n\tglobal_load_dwordx4 %9, %14, off offset:1024 sc1\n\tglobal_load_dwordx4 %10, %14, off offset:2048 sc1\n\t"
                 "global_load_dwordx4 %11, %14, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[0][3]), "=&v"(x[1][0]), "=&v"(x[1][1]), "=&v"(x[1][2]), "=&v"(x[1][3]), "=&v"(x[2][0]),
                   "=&v"(x[2][1]), "=&v"(x[2][2]), "=&v"(x[2][3])
                 : "v"(p0), "v"(p1), "v"(p2)
                 : "memory");
}

// ---- The accumulators are 128 accumulation registers BY NAME: accumulator (k, ch) = a[(8 k + ch) * 4 .. + 3], k = 0 .. 3, ch = 0 .. 7.  The compiler never sees
// them as values -- given tied "+a" operands that fill the AGPR file it time-shares AGPRs between accumulators (v_accvgpr_read of a register a product issued one
// instruction earlier has not written yet: the hazards of an inline-assembly v_mfma are invisible to it) and given its own v_mfma it needs spare
// AGPRs and spills.  Every statement that touches them clobbers all 128, so nothing of the compiler's lives in an AGPR across any of them, and the
// file is built with -amdgpu-spill-vgpr-to-agpr=0.  Ordering and hazards by construction (see the K loop).
#define LN_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define LN_ALL_AGPRS                                                                                                                              \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", LN_A8(1), LN_A8(2), LN_A8(3), LN_A8(4), LN_A8(5), LN_A8(6), LN_A8(7), LN_A8(8),   \
        LN_A8(9), LN_A8(10), LN_A8(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

template <int I, int N, class F> __device__ __forceinline__ void ln_sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ln_sfor<I + 1, N>(f);
    }
}
// accumulator at AGPR R += A x B^T (16 x 16 x 64, int8): two idle cycles in front (a freshly written operand)
template <int R> __device__ __forceinline__ void ln_mm(const v4i &A, const v4i &B)
{
    asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(A), "v"(B), "n"(R), "n"(R + 3) : LN_ALL_AGPRS);
}
template <int R> __device__ __forceinline__ void ln_acc_zero4()
{
    asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\tv_accvgpr_write_b32 a[%3], 0" ::"n"(R), "n"(R + 1),
                 "n"(R + 2), "n"(R + 3)
                 : LN_ALL_AGPRS);
}
template <int R> __device__ __forceinline__ v4i ln_acc_read4()
{
    v4i v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
    return v;
}
constexpr int ln_areg(int k, int ch) { return (8 * k + ch) * 4; }

struct LnUnit { int col, grp, win, q; };

// The four workgroups of a (line, window) re-read each other's lines from the XCD's L2 only while they walk the same frames: the K block a
// workgroup has reached, {launch tag << 12 | K blocks done}, for its three partners to see (pinned map, every workgroup resident);
// word of workgroup b: (b % 8) * 1024 + b / 8 -- an XCD's words are its own 4 KiB, a team's four are neighbours
__device__ unsigned ln_progress[8192];

__device__ __forceinline__ LnUnit ln_map_unit(const LnArgs &a, int n)
{
    LnUnit u;
    int combo;
    const int gb = a.npol == 2 ? 3 : 2;  // log2(pair groups per line)
    if (a.pinned) {  // the four (two polarisations: eight) groups of a (line, window) on one XCD, next to each other in dispatch order
        const int xcd = n & 7, within = n >> 3;
        u.grp = within & ((1 << gb) - 1);
        combo = xcd + 8 * (within >> gb);
    } else {
        u.grp = n & ((1 << gb) - 1);
        combo = n >> gb;
    }
    u.col = combo % a.ncols;
    u.win = combo / a.ncols;
    u.q = 0;
    if (a.tsplit > 1) {  // (line, time range, window): the four pair groups of a line AND time range are neighbours on one XCD (they read the same rows);
        u.q = u.win % a.tsplit;  // with 8 | ncols the ranges of a team land on one XCD as well (speed only: the exchange is placement independent)
        u.win /= a.tsplit;
    }
    // a workgroup's k-th unit is rot * k lines further on than its first: the units of the slow lines (address bits 7..9 == 3: 1.25 x the time from HBM)
    // go to twice as many workgroups, one each, instead of two each to the same ones (the host sets rot only where every k covers whole windows)
    if (a.rot) u.col = (u.col + a.rot * (n / a.grid)) % a.ncols;
    return u;
}

// DIAG: groups A / B (row tiles x < y; pairs xx, yx, yy); otherwise C / D (row tile ra against row tiles 0 and 1)
// NP == 2 (two polarisations, 64 stations): a line is 32 channels x {X, Y} x {I, Q}, so a wave's 16-byte piece is FOUR channels and the operand
// X[u][c] is channel u, polarisation c.  The 128 rows (station, polarisation) make eight row tiles but only four STATION tiles have to be loaded:
//   DIAG groups 0 / 1: station tiles (0, 1) / (2, 3); per station tile s the pairs (sX, sX), (sY, sY) -- diagonal, one combined accumulator each --
//                      and (sY, sX): the single-polarisation body with "channel ch" = (u, c) for the diagonal sets and the pair "yx" re-aimed at
//                      (sY, sX) of station tile x (even ch) and y (odd ch)
//   other groups 2 .. 7: station tile pairs (1,0) (2,0) (2,1) (3,0) (3,1) (3,2): rows a, columns b, all four polarisation products;
//                      accumulator sets 2 p2 / 2 p2 + 1 = re / im against column polarisation p2, "channel ch" = (u, row polarisation)
// Every group loads 32 stations x 32 frames per K block (four sub-stages): 256 station rows per frame and line, eight groups per line.
template <bool DIAG, bool SPLIT, int NP> __device__ __forceinline__ void ln_body(const LnArgs &a, unsigned char *lds, const int grp)
{
    static_assert(NP == 1 || !SPLIT, "time ranges: one polarisation only");
    // who issues the early touches of the slow lines: the groups with time to spare -- one polarisation: the two diagonal groups of a line (four
    // sub-stages per K block against six); two polarisations: the six station-tile pairs (32 products per wave and K block against 72)
    constexpr bool TOUCHER = NP == 2 ? !DIAG : DIAG;
    constexpr int TPL = NP == 2 ? 6 : 2;  // touchers per line
    // Two polarisations: a straight-line loop.  Two younger sub-stages are ALWAYS in flight -- past the end of the stream the last sub-stage's rows are
    // requested again, into the slot that has just been read for the last time -- so the wait is vmcnt(8) everywhere and the requests are
    // unconditional (the run-time choice of the wait and the branch around every request were scalar instructions in the loop of eight waves that
    // share one scalar unit); the drain behind another unit's matrix stores is done once, in front of the unit's K loop; no per-front timers.
    // One polarisation: ONLY the off-diagonal groups (six sub-stages per K block: the ones that set the pace) run the straight-line loop; the diagonal
    // groups (four) keep the guarded one on purpose.  With both lean the diagonal groups run further ahead and their partners' re-reads miss the L2
    // (8 windows per launch 34.7 -> 37.0 us per window); with only the slower groups lean the four workgroups of a line walk closer together:
    // 36.4 / 35.6 / 35.4 / 35.2 -> 35.1 / 34.2 / 33.6 / 34.1 us per window at 4 / 8 / 16 / 32 windows per launch (paced by 2; unpaced 40 / 38 / 37.5 / 38
    // -> 35.9 / 34.3 / 34.3 / 34.6), one window per call 56.7 -> 55.0 (tools/ab_lib.sh, one box).
    constexpr bool STRAIGHT = NP == 2 || !DIAG;
    constexpr int NRT = (DIAG || NP == 2) ? 2 : 3, NS = 2 * NRT;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int grid = (int)gridDim.x;
    const size_t row_bytes = (size_t)a.row_stride, t_stride = (size_t)a.ng * row_bytes;
    // row tiles in load order
    // (DIAG: grp 0 -> 0, 1; grp 1 -> 2, 3.  else: grp 2 / 3 -> 2 / 3, then 0, 1; two polarisations: grp 2 .. 7 -> (1,0) (2,0) (2,1) (3,0) (3,1) (3,2))
    const int pa = grp - 2, pr = pa < 1 ? 1 : pa < 3 ? 2 : 3;
    const int rt0 = DIAG ? 2 * grp : (NP == 2 ? pr : grp), rt1 = DIAG ? 2 * grp + 1 : (NP == 2 ? pa - pr * (pr - 1) / 2 : 0), rt2 = 1;

    // ---- requests.  Wave w issues the chunks of frame octet w / 2 and stations 8 (w % 2) .. + 7 of the row tile: one instruction = one station's
    // line for 8 frames.  Address = (uniform) unit, row tile, station, K block, half + (per lane) frame and piece.
    const size_t lane_off = (size_t)((wave >> 2) * 8 + (lane >> 3)) * t_stride + (size_t)(lane & 7) * 16;
    const unsigned char *hb0 = nullptr, *hb1 = nullptr, *hb2 = nullptr;
    bool pf_on = false;
    int pf_first = 0;
    const unsigned char *pf_win = nullptr;
    auto setup_issue = [&](int n) {
        const LnUnit u = ln_map_unit(a, n);
        const unsigned char *in_w = a.in + (size_t)u.win * a.in_window + (size_t)u.col * 128;
        if constexpr (SPLIT) in_w += (size_t)u.q * (size_t)(32 * a.steps) * t_stride;  // this workgroup's time range
        auto half_base = [&](int rt) {
            const int st = 16 * rt + 4 * (wave & 3);
            return in_w + (size_t)(st / a.ng) * a.in_group + (size_t)(st % a.ng) * row_bytes;
        };
        hb0 = half_base(rt0);
        hb1 = half_base(rt1);
        hb2 = half_base(rt2);
        if constexpr (TOUCHER) {
            pf_on = a.pf_dist > 0 && !((a.pf_mask >> u.col) & 1u);
            pf_first = (__builtin_popcount(~a.pf_mask & ((1u << u.col) - 1u)) * TPL + (NP == 2 ? grp - 2 : grp)) * a.pf_per;  // this workgroup among the touchers of its window
            pf_win = a.in + (size_t)u.win * a.in_window;
            if constexpr (SPLIT) pf_win += (size_t)u.q * (size_t)(32 * a.steps) * t_stride;  // (touches stay inside the own time range; ng == 64 there)
        }
    };
    const int total_sub = a.items * a.steps * NS;  // sub-stages of this workgroup's stream
    int issue_unit = 0, issue_kb = -1;             // the unit / K block whose sub-stages are being requested
    setup_issue(blockIdx.x);
    // sub-stage number m of the stream (all units of this workgroup), j = m % NS = 2 * (row tile index) + half, known at compile time at every
    // call site; sub-stages are requested in stream order, so the K block and unit just count up
    const unsigned char *iss_p0 = nullptr;  // the sub-stage being requested: this lane's source of the wave's first chunk, the chunk's LDS address
    unsigned iss_dst0 = 0;
    bool iss_on = false;
    auto issue_prep = [&](int m, int j) {
        if (j == 0 && ++issue_kb == a.steps) {
            issue_kb = 0;
            issue_unit++;
            setup_issue(blockIdx.x + issue_unit * grid);
        }
        const int kb = issue_kb;
        const int i = j >> 1, h = j & 1;
        const unsigned char *hb = i == 0 ? hb0 : i == 1 ? hb1 : hb2;
        iss_p0 = hb + (size_t)(32 * kb + 16 * h) * t_stride + lane_off;
        iss_dst0 = lds0 + (m & (kLnRing - 1)) * kLnSlot + ((wave >> 2) * 16 + (wave & 3) * 4) * kLnChunk;
        if constexpr (STRAIGHT) {
            if (a.dbg & 4) iss_p0 = a.in + (size_t)(lane & 7) * 16;  // (tuning aid "no input stream": every request the first line of four rows -- cache hits)
        }
        // early touches: wave j, at the K block's j-th sub-stage, one instruction = up to 64 (row, slow line) items of the K block pf_dist further on.
        // (In front of the sub-stage's own requests: older than they are, so the waits that count them have seen it land -- like wave 0's poll; a
        // touch that takes long holds up a diagonal group, which has the time.)
        if constexpr (TOUCHER) {
            if (pf_on && wave == j) {
                const int k = wave * 64 + lane, i = pf_first + k;
                const bool mine = k < a.pf_per && i < a.pf_items;
                const int which = i & ((1 << a.pf_lg) - 1), row = i >> a.pf_lg, t = row >> 6, st = row & 63;
                const unsigned ln = (a.pf_lines >> (8 * which)) & 0xffu;
                const unsigned char *p0 = pf_win + ((size_t)t * 64 + (size_t)st) * row_bytes + (size_t)ln * 128;
                if (mine && kb + a.pf_dist < a.steps) ln_dma16<STRAIGHT>(p0 + (size_t)(32 * (kb + a.pf_dist)) * 64 * row_bytes, lds0 + kLnTouch);
                // (time ranges: touching the K blocks nearer than the distance as well, at the range's start, measured 57.5 against 57.4 us: not done)
            }
        }
    };
    // the wave's ii-th request of that sub-stage (one station's line of 8 frames).  The four requests of a sub-stage are issued one at a time
    // between the transposes: back to back, the eight waves' 32 instructions queue at the CU's address unit and every wave stands ~330 clocks
    auto issue_one = [&](int ii) {
        if constexpr (STRAIGHT) ln_dma16<true>(iss_p0 + (size_t)ii * row_bytes, __builtin_amdgcn_readfirstlane(iss_dst0 + ii * kLnChunk));  // (unconditional: see front)
        else if (iss_on && !(a.dbg & 4)) ln_dma16<false>(iss_p0 + (size_t)ii * row_bytes, __builtin_amdgcn_readfirstlane(iss_dst0 + ii * kLnChunk));
    };
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (j < total_sub) {
            iss_on = true;
            issue_prep(j, j % NS);
#pragma unroll
            for (int ii = 0; ii < 4; ii++) issue_one(ii);
        }

    // operands: X[u][c] = (I, Q) of channel 2u + c of the wave's eight as ONE 16-byte operand of v_mfma_i32_16x16x64_i8: components 0, 1 = I of
    // the two halves of the K block (4 frames each), 2, 3 = Q; u = the dword unit the two channels share in the raw rows
    v4i X0[kLnU][2], X1[kLnU][2];
    v4i raw[4];
    const int lane_lds = ((g & 1) * 16 + r) * kLnChunk + (g >> 1) * 512 + wave * 16;

    // ---- A sub-stage is taken in two parts: front(m), then the byte transposes of its four dword units with products and requests between them --
    int kb_done = 0;                 // half K blocks of this workgroup's stream that have been started
    long long pace_budget = 10000LL * a.items;  // 100 MHz ticks this workgroup may spend waiting for partners in all (100 us per unit): a partner that is not resident is not waited for for ever
    unsigned long long t_pace = 0;
    // The four workgroups of a line are neighbours in dispatch order on ONE XCD (blockIdx % 8 plus whatever offset the dispatcher's round robin starts
    // this launch with): their progress words are four adjacent words of one 4 KiB page of ln_progress per (blockIdx % 8).  pub_local: written by PLAIN
    // stores (the line stays in the XCD's L2) and read by L1-bypassing loads (L2 hits); agent-scope (sc1) stores drop the line from the L2, so that every
    // publication and every poll crosses the fabric to the word's home.  Nothing checks the placement: a partner on ANOTHER XCD never sees this launch's
    // tag in the words it reads (a word with another tag is nobody's progress), so it is not waited for -- pacing is performance only, and every wait
    // is bounded.  (HW_REG_XCC_ID == blockIdx % 8 does NOT hold in general: the round robin goes on from where the previous dispatch stopped.)
    const int my_slot = (int)((blockIdx.x >> 3) & 3), team0 = (int)((blockIdx.x & 7) * 1024 + 4 * (blockIdx.x >> 5));
    const bool xcd_local = a.pub_local != 0, pace_wait = true;
    unsigned long long t_wait = 0, t_bar1 = 0;  // (tuning aid, MI355_XE_TS: shader clocks wave 0 spends waiting for the DMA / at the barrier)
    //   front(m): wait until sub-stage m has landed, ONE barrier -- behind it every wave has also finished reading sub-stage m - 1, whose slot takes
    //             sub-stage m + 3 (requested piecemeal by the transposes that follow: issue_one) -- and the four 16-byte LDS reads of this lane
    auto front = [&](int m, int j, bool drain) {
        if constexpr (STRAIGHT) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
            if (m + 3 < total_sub) issue_prep(m + 3, (j + 3) % NS);
            else iss_dst0 = lds0 + ((m + 3) & (kLnRing - 1)) * kLnSlot + ((wave >> 2) * 16 + (wave & 3) * 4) * kLnChunk;  // (the same rows once more)
        } else {
        const int younger = total_sub - 1 - m;  // sub-stages requested after this one that may still be in flight (at most two)
        const unsigned long long c0 = a.ts ? __builtin_readcyclecounter() : 0;
        if (drain || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        const unsigned long long c1 = a.ts ? __builtin_readcyclecounter() : 0;
        __syncthreads();
        if (a.ts) {
            t_wait += c1 - c0;
            t_bar1 += __builtin_readcyclecounter() - c1;
        }
        iss_on = m + 3 < total_sub;
        if (iss_on) issue_prep(m + 3, (j + 3) % NS);
        }
        // ---- pacing.  Twice per K block (its first and its middle sub-stage) wave 0 publishes the half K blocks this workgroup has started and asks
        // for its partners' counts (three loads: older than the requests of sub-stage m + 3, so the vmcnt(8) of the front two sub-stages later has
        // seen them land); there, a workgroup more than `pace` half K blocks ahead of its slowest partner waits (bounded) -- the laggard's lines would
        // otherwise have left the L2.  (The check of one half and the publication of the next may fall on the same sub-stage: check first.)
        if (a.pace > 0 && wave == 0) {
            const bool pub = j == 0 || j == NS / 2, chk = j == 2 || j == (NS / 2 + 2) % NS;
            if (chk && pace_wait && kb_done > 0) {
                const unsigned seen = lane < 3 ? *(const unsigned *)(lds + kLnPoll + lane * 4) : 0u;  // (the words have landed: see above)
                unsigned mine = (unsigned)kb_done;
                auto behind = [&](unsigned v) { return lane < 3 && (v >> 12) == a.tag && (v & 0xfffu) + (unsigned)a.pace < mine; };
                if (__builtin_amdgcn_ballot_w64(behind(seen)) != 0ull && pace_budget > 0) {
                    const unsigned long long t0 = wall_clock64();
                    unsigned long long waited = 0;
                    bool lag = true;
                    while (lag && waited < (unsigned long long)pace_budget) {
                        __builtin_amdgcn_s_sleep(16);
                        unsigned v = 0;
                        if (lane < 3) v = __hip_atomic_load(&ln_progress[team0 + ((my_slot + 1 + lane) & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lag = __builtin_amdgcn_ballot_w64(behind(v)) != 0ull;
                        waited = wall_clock64() - t0;
                    }
                    pace_budget -= (long long)waited;
                    t_pace += waited;
                }
            }
            if (pub) {
                kb_done++;
                if (lane == 0) {
                    const unsigned word = (a.tag << 12) | (unsigned)kb_done;
                    // (a plain store: the line stays in the XCD's L2, where the partners' loads find it; "volatile" in C++ would be a system-scope store and a wait)
                    if (xcd_local) asm volatile("global_store_dword %0, %1, off" ::"v"(&ln_progress[team0 + my_slot]), "v"(word) : "memory");
                    else __hip_atomic_store(&ln_progress[team0 + my_slot], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane < 3) {
                    // (by LDS-DMA: a load into a register would be the compiler's to copy or spill before the data is there)
                    const unsigned *q = &ln_progress[team0 + ((my_slot + 1 + lane) & 3)];
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(q), "s"(lds0 + kLnPoll)
                                 : "memory");
                }
            }
        }
        const unsigned char *cb = lds + (m & (kLnRing - 1)) * kLnSlot + lane_lds;
#pragma unroll
        for (int ti = 0; ti < 4; ti++) raw[ti] = *(const v4i *)(cb + ti * 128);
    };
    auto perm_unit = [&](v4i (&X)[kLnU][2], int u, int H) {
        unsigned o[4];
        transpose4x4((unsigned)raw[0][u], (unsigned)raw[1][u], (unsigned)raw[2][u], (unsigned)raw[3][u], o);
        X[u][0][H] = (int)o[0];
        X[u][0][2 + H] = (int)o[1];
        X[u][1][H] = (int)o[2];
        X[u][1][2 + H] = (int)o[3];
        issue_one(u);
    };
    // The products (accumulators by name, see ln_mm).  What the compiler's hazard pass would do for its own v_mfma is done by construction: the
    // products of a list are issued type by type over the wave's eight channels (two products on one accumulator are seven products apart; in the
    // channel-by-channel order of perms_with_off at least four); every
    // product is preceded by two idle cycles; the accumulators are read after the pipeline has drained.
    // 16 x 16 x 64: K = (half, frame, I | Q), so ONE product is sum_t (I_a I_b + Q_a Q_b) -- the whole real part -- and with the second operand
    // (~Q_b, I_b) the whole of im'; the third, against (1, 0), adds sum_t I_a(t) to every column of row a: the "+ 1" of -q = ~q + 1
    auto swapped = [&](const v4i &x) { return (v4i){~x[2], ~x[3], x[0], x[1]}; };
    const v4i ones = (v4i){0x01010101, 0x01010101, 0, 0};
    // item I (0 .. 23) of a diagonal pair's list, accumulator set KC: C += (I, Q) (I, Q)^T | (I, Q) (~Q, I)^T | (I, Q) (1, 0)^T, eight channels each
    auto diag_item = [&](auto kc, const v4i (&X)[kLnU][2], auto ic) {
        constexpr int KC = decltype(kc)::value, I = decltype(ic)::value, ty = I / kLnCh, ch = I % kLnCh;
        const v4i &x = X[ch >> 1][ch & 1];
        if constexpr (ty == 0) ln_mm<ln_areg(KC, ch)>(x, x);
        else if constexpr (ty == 1) ln_mm<ln_areg(KC, ch)>(x, swapped(x));
        else ln_mm<ln_areg(KC, ch)>(x, ones);
    };
    // item I of an off-diagonal pair's list (rows XA, columns XB), accumulator sets KR (re) and KR + 1 (im)
    auto off_item = [&](auto kr, const v4i (&XA)[kLnU][2], const v4i (&XB)[kLnU][2], auto ic) {
        constexpr int KR = decltype(kr)::value, I = decltype(ic)::value, ty = I / kLnCh, ch = I % kLnCh;
        // (two polarisations, diagonal groups -- called with XA = station tile y, XB = station tile x: rows = polarisation Y, columns = polarisation X of
        // ONE station tile and channel ch / 2, x for even ch, y for odd ch)
        constexpr bool D2 = DIAG && NP == 2;
        const v4i &xa = D2 ? ((ch & 1) ? XA[ch >> 1][1] : XB[ch >> 1][1]) : XA[ch >> 1][ch & 1];
        const v4i &xb = D2 ? ((ch & 1) ? XA[ch >> 1][0] : XB[ch >> 1][0]) : XB[ch >> 1][ch & 1];
        if constexpr (ty == 0) ln_mm<ln_areg(KR, ch)>(xa, xb);
        else if constexpr (ty == 1) ln_mm<ln_areg(KR + 1, ch)>(xa, swapped(xb));
        else ln_mm<ln_areg(KR + 1, ch)>(xa, ones);
    };
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    typedef std::integral_constant<int, 2> K2;
    auto diag_range = [&](auto kc, const v4i (&X)[kLnU][2], auto lo, auto hi) {
        ln_sfor<decltype(lo)::value, decltype(hi)::value>([&](auto ic) { diag_item(kc, X, ic); });
    };
    // a sub-stage's eight transposes into half H of X with a stretch of a diagonal pair's list (all of whose operands are complete) spread over
    // them: the items of unit u FIRST (the first ones run under the latency of the LDS reads), CUM = the cumulative counts per unit
    auto perms_with_diag = [&](v4i (&X)[kLnU][2], auto hc, auto kc, const v4i (&XD)[kLnU][2], auto basec, auto cumc) {
        ln_sfor<0, kLnU>([&](auto uc) {
            constexpr int u = decltype(uc)::value, H = decltype(hc)::value, B = decltype(basec)::value;
            typedef decltype(cumc) CUM;
            diag_range(kc, XD, std::integral_constant<int, B + CUM::at(u)>{}, std::integral_constant<int, B + CUM::at(u + 1)>{});
            perm_unit(X, u, H);
        });
    };
    // the LAST sub-stage of an operand set: unit u completes channels 2u, 2u + 1 of X, whose products against the other, complete set follow at
    // once -- real part and im' of both channels, and the row-sum product of the PREVIOUS unit's channels (same accumulator as im': kept four
    // products apart).  ROWS: X holds the rows (XO the columns), else the columns.
    auto perms_with_off = [&](v4i (&X)[kLnU][2], auto kr, const v4i (&XO)[kLnU][2], auto rowsc) {
        constexpr bool ROWS = decltype(rowsc)::value != 0;
        auto item = [&](auto tyc, auto chc) {
            constexpr int I = decltype(tyc)::value * kLnCh + decltype(chc)::value;
            if constexpr (ROWS) off_item(kr, X, XO, std::integral_constant<int, I>{});
            else off_item(kr, XO, X, std::integral_constant<int, I>{});
        };
        typedef std::integral_constant<int, 0> T0;
        typedef std::integral_constant<int, 1> T1;
        typedef std::integral_constant<int, 2> T2;
        ln_sfor<0, kLnU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            perm_unit(X, u, 1);
            item(T0{}, std::integral_constant<int, 2 * u>{});
            item(T0{}, std::integral_constant<int, 2 * u + 1>{});
            item(T1{}, std::integral_constant<int, 2 * u>{});
            item(T1{}, std::integral_constant<int, 2 * u + 1>{});
            if constexpr (DIAG && u > 0) {
                item(T2{}, std::integral_constant<int, 2 * u - 2>{});
                item(T2{}, std::integral_constant<int, 2 * u - 1>{});
            }
        });
        if constexpr (DIAG) {
            item(T2{}, std::integral_constant<int, kLnCh - 2>{});
            item(T2{}, std::integral_constant<int, kLnCh - 1>{});
        }
    };
    // Off-diagonal groups: both pairs have their rows in ONE row tile (X0), so the "+ 1" of -q = ~q + 1 -- sum_t I_a(t) on every column of row a -- is
    // the same vector for both.  It is not a third product there (a third of the matrix pipe's time in the groups that set the pace) but a byte sum
    // of the lane's own I planes (station r, frame group g), added into the wave's 2 KiB of LDS per K block; the epilogue sums the four frame groups
    // and hands row i its total (as k_xe_i8_fused does with its row sums).
    int *const rs_lds = (int *)(lds + kLnRs) + wave * (kLnCh * 64) + lane;
    auto row_sums = [&]() {
#pragma unroll
        for (int ch = 0; ch < kLnCh; ch++) {
            const v4i &x = X0[ch >> 1][ch & 1];
            int t = __builtin_amdgcn_sdot4(x[0], 0x01010101, 0, false);
            t = __builtin_amdgcn_sdot4(x[1], 0x01010101, t, false);
            __hip_atomic_fetch_add(rs_lds + ch * 64, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // two polarisations, station tile pair (a, b): the last sub-stage of b completes channel u (both polarisations) of X1, whose eight products
    // against X0 follow at once: rows (a, p1), columns (b, p2), re into set 2 p2 and im' into set 2 p2 + 1 of "channel" 2 u + p1
    auto perms_with_off2 = [&]() {
        ln_sfor<0, kLnU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            perm_unit(X1, u, 1);
            ln_sfor<0, 4>([&](auto qc) {
                constexpr int p2 = decltype(qc)::value >> 1, p1 = decltype(qc)::value & 1;
                ln_mm<ln_areg(2 * p2, 2 * u + p1)>(X0[u][p1], X1[u][p2]);
            });
            ln_sfor<0, 4>([&](auto qc) {
                constexpr int p2 = decltype(qc)::value >> 1, p1 = decltype(qc)::value & 1;
                ln_mm<ln_areg(2 * p2 + 1, 2 * u + p1)>(X0[u][p1], swapped(X1[u][p2]));
            });
        });
    };
    // (a list = 3 x 8 = 24 items; half a list per sub-stage of x, 20 of the pair xx under the first sub-stage of y and its last 4 in front of the second)
    struct Cum24 { static constexpr int at(int u) { constexpr int c[5] = {0, 4, 7, 10, 12}; return c[u]; } };
    struct Cum40 { static constexpr int at(int u) { constexpr int c[5] = {0, 6, 11, 16, 20}; return c[u]; } };
    typedef std::integral_constant<int, 12> C24;
    typedef std::integral_constant<int, 20> C40;
    typedef std::integral_constant<int, 24> C48;
    typedef K0 H0;
    typedef K1 H1;

    const int nb = 64 * 65 / 2;
    int m = 0;  // sub-stage counter of the stream
    for (int unit = 0; unit < a.items; unit++) {
        ln_sfor<0, 4 * kLnCh>([&](auto qc) { ln_acc_zero4<4 * decltype(qc)::value>(); });
        // (the loop carries the pair yy of a K block into the next one; before the first K block it runs on zeros)
#pragma unroll
        for (int u = 0; u < kLnU; u++) X1[u][0] = X1[u][1] = (v4i){0, 0, 0, 0};
        if constexpr (!DIAG) {
#pragma unroll
            for (int ch = 0; ch < kLnCh; ch++) rs_lds[ch * 64] = 0;
        }
        if constexpr (STRAIGHT) {
            if (unit > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int kb = 0; kb < a.steps; kb++) {
            // (the first wait of a unit that follows another one drains everything: that unit's matrix stores share the counter with the DMA and
            // complete out of order with it)
            const bool drain = unit > 0 && kb == 0;
            if constexpr (DIAG) {
                // X0 = row tile x, X1 = row tile y; accumulator sets 0 = xx, 1 = yy, 2 / 3 = re / im of yx.  Every product runs between the
                // transposes of some sub-stage: the pair yy of the PREVIOUS K block (reads X1 only) under the two sub-stages of x, the pair xx under
                // the first of y, the pair yx channel by channel as the second of y completes X1.
                front(m, 0, drain);
                perms_with_diag(X0, H0{}, K1{}, X1, K0{}, Cum24{});
                front(m + 1, 1, false);
                perms_with_diag(X0, H1{}, K1{}, X1, C24{}, Cum24{});
                front(m + 2, 2, false);
                perms_with_diag(X1, H0{}, K0{}, X0, K0{}, Cum40{});
                front(m + 3, 3, false);
                diag_range(K0{}, X0, C40{}, C48{});
                perms_with_off(X1, K2{}, X0, K1{});
                m += 4;
            } else if constexpr (NP == 2) {
                // X0 = station tile a (rows), X1 = station tile b (columns)
                front(m, 0, drain);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X0, u, 0);
                front(m + 1, 1, false);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X0, u, 1);
                row_sums();
                front(m + 2, 2, false);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X1, u, 0);
                front(m + 3, 3, false);
                perms_with_off2();
                m += 4;
            } else {
                // X0 = row tile a, X1 = row tile 0, then 1; accumulator sets 0 / 1 = re / im of (a, 0), 2 / 3 of (a, 1): each pair channel by channel
                // as the second sub-stage of its column tile completes X1
                front(m, 0, drain);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X0, u, 0);
                front(m + 1, 1, false);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X0, u, 1);
                row_sums();
                front(m + 2, 2, false);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X1, u, 0);
                front(m + 3, 3, false);
                perms_with_off(X1, K0{}, X0, K0{});
                front(m + 4, 4, false);
#pragma unroll
                for (int u = 0; u < kLnU; u++) perm_unit(X1, u, 0);
                front(m + 5, 5, false);
                perms_with_off(X1, K2{}, X0, K0{});
                m += 6;
            }
        }
        // the products the loop carries: the pair yy of the last K block
        if constexpr (DIAG) diag_range(K1{}, X1, K0{}, C48{});
        if constexpr (STRAIGHT) {
            if (unit == a.items - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stream's trailing requests have landed: the ring may be reused)
        }
        if constexpr (!SPLIT) {
            if (a.ts && tid == 0 && unit == a.items - 1) a.ts[(size_t)blockIdx.x * 8 + 5] = wall_clock64();  // (tuning aid: the last unit's loop end)
        }
        // ---- the unit's matrices: scale, scatter into [chan][baseline]
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // (the last products have left the pipeline before an accumulator is read)
        const LnUnit un = ln_map_unit(a, blockIdx.x + unit * grid);
        c32 *const out_w = a.out + (size_t)un.win * a.out_window;
        int rr = lane & 15, gg = lane >> 4;
        asm volatile("" : "+v"(rr), "+v"(gg));  // (keeps the tail's index arithmetic out of the loop's registers)
        const bool odd = (rr & 1) != 0;
        int *tile = (int *)(lds + kLnRing * kLnSlot + wave * kLnTile);
        // two values per lane and row pair -> 16-byte stores of two neighbouring baselines of ONE row (a lane pair swaps a value each)
        auto store_rows = [&](const int (&vre)[4], const int (&vim)[4], int f, int bi, int bj, bool small) {
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                c32 w[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (small) {
                        w[e].x = ln_scale127_small(vre[rp + e]);
                        w[e].y = ln_scale127_small(vim[rp + e]);
                    } else {
                        w[e].x = (float)((double)vre[rp + e] * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
                        w[e].y = (float)((double)vim[rp + e] * a.kd * a.kd);
                    }
                }
                const float sx = odd ? w[0].x : w[1].x, sy = odd ? w[0].y : w[1].y;  // what the neighbour stores of this lane's values
                const float gx = __shfl_xor(sx, 1), gy = __shfl_xor(sy, 1);
                const int i = 4 * gg + rp + (odd ? 1 : 0), j0 = rr & ~1;  // this lane stores columns j0, j0 + 1 of row i of the tile pair
                const int s1 = bi * 16 + i;
                c32 *dst = out_w + (size_t)f * nb + (s1 * (s1 + 1) / 2 + bj * 16 + j0);
                const c32 first = odd ? c32{gx, gy} : w[0], second = odd ? w[1] : c32{gx, gy};
                if (bi != bj || j0 + 1 <= i) {
                    const v4f q4 = (v4f){first.x, first.y, second.x, second.y};
                    __builtin_memcpy((void *)dst, &q4, 16);
                } else if (j0 <= i) {
                    *dst = first;
                }
            }
        };
        auto emit_off = [&](const v4i &RE, const v4i &IM, int f, int bi, int bj) {
            int vre[4], vim[4];
            unsigned mag = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                vre[k] = RE[k];
                vim[k] = IM[k];
                mag |= ln_mag_bits(vre[k]) | ln_mag_bits(vim[k]);
            }
            if (a.dbg & 2) { if (vre[0] == 0x12345678 && vim[1] == 0x7654321) out_w[lane].x = 1.0f; return; }
            store_rows(vre, vim, f, bi, bj, a.k127 && ln_all_small(mag));
        };
        auto emit_diag = [&](const v4i &C, int f, int bi) {
            int ct[4], vre[4], vim[4];
            // C[i][j], i = 4 g + reg, j = r; its transpose through this wave's scratch (same wave: no barrier)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ct[k] = C[k];
                tile[(4 * gg + k) * 20 + rr] = ct[k];
            }
            const v4i trow = *(const v4i *)(tile + rr * 20 + 4 * gg);  // C_true[r][4 g + reg]
            unsigned mag = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                vre[k] = (ct[k] + trow[k]) >> 1;  // (even: exact)
                vim[k] = (ct[k] - trow[k]) >> 1;  // 0 on the diagonal
                mag |= ln_mag_bits(vre[k]) | ln_mag_bits(vim[k]);
            }
            if (a.dbg & 2) { if (vre[0] == 0x12345678 && vim[1] == 0x7654321) out_w[lane].x = 1.0f; return; }
            store_rows(vre, vim, f, bi, bi, a.k127 && ln_all_small(mag));
        };
        if constexpr (SPLIT) {
            // ---- Time ranges: the tsplit workgroups of a team hold exact partial sums of the same 32 tiles per wave.  A wave's tiles form four PIECES
            // of eight (diagonal groups: xx | yy | re, im of yx for channels 0-3 | 4-7; off-diagonal groups: re, im of (a, 0) for channels 0-3 | 4-7, of
            // (a, 1) for 0-3 | 4-7; the row sums are added to im' first, so every piece is a plain sum over the ranges); piece p belongs to range
            // p * tsplit / 4.  Every workgroup sends the pieces of the other ranges lane for lane (one 1 KiB store per tile and wave) to the owners'
            // inboxes, drains, raises the team's arrival count and, once all have arrived, adds what it received to its own pieces, scales and
            // scatters them: every workgroup emits 1 / tsplit of the team's matrix.  Placement independent (sc1 stores, sc1 loads, count raised after
            // the stores have completed); bounded wait: a workgroup that gives up stores its own pieces too, sets its bit and leaves, the LAST
            // to arrive sees the bits with its own arrival and finishes those pieces -- complete for any dispatch order (as in k_xe_i8_fused).
            const int S = a.tsplit, QC = un.q;
            const int team = (un.win * a.ncols + un.col) * 4 + grp;
            unsigned char *const box = a.part + (size_t)team * ((size_t)4 * S * 65536) + (size_t)wave * 8192 + (size_t)lane * 16;
            auto slot = [&](int p, int src) { return box + (size_t)(p * S + src) * 65536; };
            const int f0 = un.col * 64 + wave * kLnCh;
            // half H (tiles 4 H .. 4 H + 3) of piece P out of the accumulators
            auto get_half = [&](auto pc, auto hc, v4i (&v)[4]) {
                constexpr int P = decltype(pc)::value, H = decltype(hc)::value;
                if constexpr (DIAG && P < 2) {
                    ln_sfor<0, 4>([&](auto jc) { v[decltype(jc)::value] = ln_acc_read4<ln_areg(P, 4 * H + decltype(jc)::value)>(); });
                } else {
                    ln_sfor<0, 2>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        constexpr int ch = DIAG ? 4 * (P - 2) + 2 * H + c : 4 * (P & 1) + 2 * H + c, k0 = DIAG ? 2 : 2 * (P >> 1);
                        v[2 * c] = ln_acc_read4<ln_areg(k0, ch)>();
                        v[2 * c + 1] = ln_acc_read4<ln_areg(k0 + 1, ch)>();
                        if constexpr (!DIAG) {  // im = im' + sum_t I_a(t): this range's share of the row sums
                            int rsv = rs_lds[ch * 64];
                            rsv += __shfl_xor(rsv, 16);
                            rsv += __shfl_xor(rsv, 32);
#pragma unroll
                            for (int k = 0; k < 4; k++) v[2 * c + 1][k] += __shfl(rsv, 4 * gg + k);
                        }
                    });
                }
            };
            auto emit_half = [&](int p, int H, const v4i (&v)[4]) {
                if (DIAG && p < 2) {
#pragma unroll
                    for (int j = 0; j < 4; j++) emit_diag(v[j], f0 + 4 * H + j, p == 0 ? rt0 : rt1);
                } else {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        if constexpr (DIAG) emit_off(v[2 * c], v[2 * c + 1], f0 + 4 * (p - 2) + 2 * H + c, rt1, rt0);
                        else emit_off(v[2 * c], v[2 * c + 1], f0 + 4 * (p & 1) + 2 * H + c, rt0, p >> 1);
                    }
                }
            };
            auto send_piece = [&](auto pc) {  // this range's share of piece P -> slot (P, QC)
                unsigned char *d = slot(decltype(pc)::value, QC);
                ln_sfor<0, 2>([&](auto hc) {
                    v4i v[4];
                    get_half(pc, hc, v);
#pragma unroll
                    for (int j = 0; j < 4; j++) ln_st_sys(d + (4 * decltype(hc)::value + j) * 1024, v[j]);
                });
            };
            if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 8 + 5] = wall_clock64();  // loop end
            ln_sfor<0, 4>([&](auto pc) {
                if (decltype(pc)::value * S / 4 != QC) send_piece(pc);
                __builtin_amdgcn_sched_barrier(0);
            });
            unsigned long long *state = a.flags + (size_t)(a.epoch & 1u) * a.flag_bank + (size_t)team;
            const unsigned full = (unsigned)S, tag = a.epoch & 0x0fffffffu;
            int *const s_mode = (int *)(lds + kLnPoll), *const s_mask = s_mode + 1;  // (the pacing words' place: there is no pacing in this form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have completed
            __syncthreads();
            if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 8 + 6] = wall_clock64();  // pieces sent
            if (tid == 0) {
                const unsigned long long old = __hip_atomic_fetch_add(state, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int mode = 0, mask = 0;
                if ((unsigned)(old >> 32) + 1u == full) {  // the last to arrive: the give-up bits are final (they can only be set while the count is short)
                    mode = 2;
                    if (((unsigned)old >> 4) == tag) mask = (int)((unsigned)old & 15u);
                } else {
                    // bounded by time (100 MHz wall clock; about one loop of this geometry -- LnArgs::wait_ticks); dbg 512: give up at once (tests)
                    const unsigned long long t_w = wall_clock64(), limit = (a.dbg & 512) ? 0 : (unsigned long long)a.wait_ticks;
                    do {
                        const unsigned long long cur = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(cur >> 32) == full) mode = 1;
                        else __builtin_amdgcn_s_sleep(8);
                    } while (!mode && wall_clock64() - t_w < limit);
                    if (!mode) mode = 3;
                }
                *s_mode = mode;
                *s_mask = mask;
            }
            __syncthreads();
            if (*s_mode == 3) {  // waited long enough: hand the own pieces over as well, then say so -- unless everybody has arrived meanwhile
                ln_sfor<0, 4>([&](auto pc) {
                    if (decltype(pc)::value * S / 4 == QC) send_piece(pc);
                    __builtin_amdgcn_sched_barrier(0);
                });
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
                    *s_mode = mode;
                }
                __syncthreads();
                if (*s_mode == 0) {
                    if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
                    continue;  // (items == 1 in this form: the workgroup is done)
                }
            }
            if (a.ts && tid == 0) a.ts[(size_t)blockIdx.x * 8 + 7] = wall_clock64();  // all ranges in
            // this range's pieces: own share out of the accumulators + the other ranges' from the inbox, half a piece (four tiles) at a time
            ln_sfor<0, 4>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                if (P * S / 4 == QC) {
                    ln_sfor<0, 2>([&](auto hc) {
                        constexpr int H = decltype(hc)::value;
                        v4i v[4];
                        if (S == 4) {
                            v4i x[3][4];
                            ln_ld12(slot(P, (QC + 1) & 3) + H * 4096, slot(P, (QC + 2) & 3) + H * 4096, slot(P, (QC + 3) & 3) + H * 4096, x);
                            get_half(pc, hc, v);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += x[0][j] + x[1][j] + x[2][j];
                        } else {
                            v4i x[4];
                            ln_ld4(slot(P, QC ^ 1) + H * 4096, x);
                            get_half(pc, hc, v);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += x[j];
                        }
                        emit_half(P, H, v);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // (last arriver only, and only after a bounded wait ran out somewhere) the pieces of the ranges that gave up: every share from the inbox
            const int todo = *s_mode == 2 ? (*s_mask & ~(1 << QC)) : 0;
            if (todo) {
                for (int p = 0; p < 4; p++) {
                    if (!((todo >> (p * S / 4)) & 1)) continue;
                    for (int H = 0; H < 2; H++) {
                        v4i v[4] = {(v4i){0, 0, 0, 0}, (v4i){0, 0, 0, 0}, (v4i){0, 0, 0, 0}, (v4i){0, 0, 0, 0}};
                        for (int src = 0; src < S; src++) {
                            v4i x[4];
                            ln_ld4(slot(p, src) + H * 4096, x);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += x[j];
                        }
                        emit_half(p, H, v);
                    }
                }
            }
        } else if constexpr (NP == 2) {
            // ---- two polarisations: [chan][baseline][XX, XY, YX, YY] (first letter: the polarisation of station s1 >= s2, lib/clXEngine_impl.cc:786-808).
            // A lane holds element (i = 4 g + reg, j = r) of all four products of a station pair: 32 contiguous bytes per baseline, two 16-byte stores,
            // sixteen lanes a 512-byte run.
            auto store4 = [&](c32 *dst, const int (&re)[4], const int (&im)[4], bool small, bool on) {
                float w[8];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    w[2 * q] = small ? ln_scale127_small(re[q]) : (float)((double)re[q] * a.kd * a.kd);  // (the oracle's expression, rounded once)
                    w[2 * q + 1] = small ? ln_scale127_small(im[q]) : (float)((double)im[q] * a.kd * a.kd);
                }
                if (a.dbg & 2) { if (re[0] == 0x12345678 && im[1] == 0x7654321) out_w[lane].x = w[0]; return; }
                if (on) {
                    const v4f q0 = (v4f){w[0], w[1], w[2], w[3]}, q1 = (v4f){w[4], w[5], w[6], w[7]};
                    __builtin_memcpy((void *)dst, &q0, 16);
                    __builtin_memcpy((void *)(dst + 2), &q1, 16);
                }
            };
            if constexpr (DIAG) __syncthreads();  // every wave has read its last sub-stage: the ring is scratch now (the transposes below)
            ln_sfor<0, kLnU>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int f = un.col * 32 + wave * kLnU + u;
                if constexpr (DIAG) {
                    ln_sfor<0, 2>([&](auto sc) {
                        constexpr int S2 = decltype(sc)::value;  // 0: station tile x (rt0), 1: y (rt1)
                        const int bt = S2 ? rt1 : rt0;
                        const v4i cxx = ln_acc_read4<ln_areg(S2, 2 * u)>(), cyy = ln_acc_read4<ln_areg(S2, 2 * u + 1)>();
                        const v4i yre = ln_acc_read4<ln_areg(2, 2 * u + S2)>(), yim = ln_acc_read4<ln_areg(3, 2 * u + S2)>();
                        // (the four transposes of a station tile and channel through four scratch tiles at once: one LDS round trip instead of four)
                        int *const t4 = (int *)lds + wave * (4 * 320);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            t4[(4 * gg + k) * 20 + rr] = cxx[k];
                            t4[320 + (4 * gg + k) * 20 + rr] = cyy[k];
                            t4[640 + (4 * gg + k) * 20 + rr] = yre[k];
                            t4[960 + (4 * gg + k) * 20 + rr] = yim[k];
                        }
                        const v4i txx = *(const v4i *)(t4 + rr * 20 + 4 * gg), tyy = *(const v4i *)(t4 + 320 + rr * 20 + 4 * gg),
                                  tre = *(const v4i *)(t4 + 640 + rr * 20 + 4 * gg), tim = *(const v4i *)(t4 + 960 + rr * 20 + 4 * gg);
                        int re[4][4], im[4][4];
                        unsigned mag = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            re[k][0] = (cxx[k] + txx[k]) >> 1; im[k][0] = (cxx[k] - txx[k]) >> 1;  // XX: re symmetric, im antisymmetric (exact halves)
                            re[k][1] = tre[k];                 im[k][1] = -tim[k];                  // XY[i][j] = conj(YX[j][i])
                            re[k][2] = yre[k];                 im[k][2] = yim[k];                   // YX
                            re[k][3] = (cyy[k] + tyy[k]) >> 1; im[k][3] = (cyy[k] - tyy[k]) >> 1;  // YY
#pragma unroll
                            for (int q = 0; q < 4; q++) mag |= ln_mag_bits(re[k][q]) | ln_mag_bits(im[k][q]);
                        }
                        const bool small = a.k127 && ln_all_small(mag);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int i = 4 * gg + k, s1 = bt * 16 + i;
                            store4(out_w + ((size_t)f * nb + (s1 * (s1 + 1) / 2 + bt * 16 + rr)) * 4, re[k], im[k], small, rr <= i);
                        }
                    });
                } else {
                    // rows (a, p1) of the station tile pair: the sum over the four frame groups of station i's I bytes, per row polarisation
                    int corr[2][4];
#pragma unroll
                    for (int p1 = 0; p1 < 2; p1++) {
                        int v = rs_lds[(2 * u + p1) * 64];
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
#pragma unroll
                        for (int k = 0; k < 4; k++) corr[p1][k] = __shfl(v, 4 * gg + k);
                    }
                    int re[4][4], im[4][4];
                    unsigned mag = 0;
                    ln_sfor<0, 4>([&](auto qc) {
                        constexpr int q = decltype(qc)::value, p1 = q >> 1, p2 = q & 1;
                        const v4i vr = ln_acc_read4<ln_areg(2 * p2, 2 * u + p1)>(), vi = ln_acc_read4<ln_areg(2 * p2 + 1, 2 * u + p1)>();
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            re[k][q] = vr[k];
                            im[k][q] = vi[k] + corr[p1][k];
                            mag |= ln_mag_bits(re[k][q]) | ln_mag_bits(im[k][q]);
                        }
                    });
                    const bool small = a.k127 && ln_all_small(mag);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int s1 = rt0 * 16 + 4 * gg + k;
                        store4(out_w + ((size_t)f * nb + (s1 * (s1 + 1) / 2 + rt1 * 16 + rr)) * 4, re[k], im[k], small, true);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (one channel's values at a time)
            });
        } else {
            ln_sfor<0, kLnCh>([&](auto chc) {
                constexpr int ch = decltype(chc)::value;
                const int f = un.col * 64 + wave * kLnCh + ch;
                if (f < a.Fout) {
                    if constexpr (DIAG) {
                        emit_diag(ln_acc_read4<ln_areg(0, ch)>(), f, rt0);
                        emit_off(ln_acc_read4<ln_areg(2, ch)>(), ln_acc_read4<ln_areg(3, ch)>(), f, rt1, rt0);
                        emit_diag(ln_acc_read4<ln_areg(1, ch)>(), f, rt1);
                    } else {
                        // row i = 4 g + reg of the tile pair: the sum over the four frame groups of station i's I bytes
                        int v = rs_lds[ch * 64];
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        v4i im0 = ln_acc_read4<ln_areg(1, ch)>(), im1 = ln_acc_read4<ln_areg(3, ch)>();
    #pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int c = __shfl(v, 4 * gg + k);
                            im0[k] += c;
                            im1[k] += c;
                        }
                        emit_off(ln_acc_read4<ln_areg(0, ch)>(), im0, f, rt0, 0);
                        emit_off(ln_acc_read4<ln_areg(2, ch)>(), im1, f, rt0, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (one channel's values at a time)
            });
        }
    }
    if (a.ts && tid == 0) {
        a.ts[(size_t)blockIdx.x * 8 + 2] = t_wait;
        a.ts[(size_t)blockIdx.x * 8 + 3] = t_bar1;
        a.ts[(size_t)blockIdx.x * 8 + 4] = t_pace;
        if constexpr (!SPLIT) a.ts[(size_t)blockIdx.x * 8 + 6] = __builtin_readcyclecounter();
    }
}

// SPLIT: time ranges, combined inside the launch (its own kernel: with both endings in one kernel the loop's register allocation suffers)
template <bool SPLIT, int NP = 1> __global__ __launch_bounds__(kLnThreads) void k_xe_i8_lines(LnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (a.ts && threadIdx.x == 0) a.ts[(size_t)blockIdx.x * 8] = wall_clock64();
    // (a workgroup's units are all of one group: the host makes the grid a multiple of 32 -- pinned map -- or of 4)
    const LnUnit u0 = ln_map_unit(a, blockIdx.x);
    const int grp = u0.grp;
    if constexpr (SPLIT) {
        // the arrival words come in two banks: launch e counts in bank e % 2 from zero and first clears the OTHER bank's word of its team, so whatever
        // an unfinished launch left there is gone before launch e + 1 -- stream-ordered behind this one -- looks at it
        if (threadIdx.x == 0 && u0.q == 0)
            __hip_atomic_store(a.flags + (size_t)((a.epoch + 1u) & 1u) * a.flag_bank + (size_t)((u0.win * a.ncols + u0.col) * 4 + grp), 0ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (SPLIT) {
        // test switch (MI355_XE_DBG bits 16 / 17): only / all but the first time range of every team runs -- the others leave before they arrive, so
        // every team's count stays short of full (the units that do run wait, give up, and nobody finishes): what an aborted launch leaves behind
        const int m = (a.dbg >> 16) & 3;
        if (m && ((m == 1) != (u0.q == 0))) return;
    }
    if (grp < 2) ln_body<true, SPLIT, NP>(a, lds, grp);
    else ln_body<false, SPLIT, NP>(a, lds, grp);
    if (a.ts && threadIdx.x == 0) a.ts[(size_t)blockIdx.x * 8 + 1] = wall_clock64();
}

}  // namespace

// Geometry the whole-line kernel covers: one polarisation, exactly 64 stations, rows of whole 128-byte lines (a multiple of 64 channels, all of them
// output), whole K blocks, at most 16384 frames (the combined accumulator of the diagonal pairs), antenna groups of a multiple of 8 stations, enough
// (window, line, group) units to fill the device WITHOUT time ranges.  MI355_XE_NO_LINES=1: never.
bool mi355_xe_lines_ok(int N, int F, int Fout, int npol, int T, int stations_per_group, int accumulate, int nint, int cus)
{
    if (getenv("MI355_XE_NO_LINES")) return false;
    const int ng = (stations_per_group > 0 && stations_per_group < N) ? stations_per_group : N;
    // (two polarisations: lines of 32 channels, eight pair groups per line -- k_xe_i8_lines<false, 2>; MI355_XE_NO_LINES2=1: never)
    if ((npol != 1 && npol != 2) || N != 64 || F % (64 / npol) != 0 || Fout != F || T % 32 != 0 || T < 32 || T > 16384 || accumulate || ng % 8 != 0) return false;
    if (npol == 2 && getenv("MI355_XE_NO_LINES2")) return false;
    const int G = npol == 2 ? 8 : 4;  // pair groups per line
    const long units = (long)(nint > 0 ? nint : 1) * (F / (64 / npol)) * G;
    // MI355_XE_LINES_MIN_UNITS (test switch): any unit count.  Otherwise: enough units to fill the device and a multiple of 32 (the pinned map, which
    // the pacing of a line's four workgroups needs) that splits into equal shares of at most MI355_XE_LINES_MAX_ITEMS (default 64) units per workgroup
    // over a grid that is a multiple of 32 and at least 7/8 of the CUs (6 or 10 windows of config 5 would run on 192 / 160 workgroups) (measured at config 5, windows per launch 4 / 8 / 16: 180 / 350 / 637 us against 204 / 397 / 780 for the
    // 32-byte-slice kernel: the more units per workgroup, the smaller the share of the last units' matrix stores, which nothing overlaps)
    if (getenv("MI355_XE_LINES_MIN_UNITS")) return units >= atoi(getenv("MI355_XE_LINES_MIN_UNITS")) && units % G == 0;
    if (units < cus || units % (8 * G) != 0) return false;
    const long max_items = getenv("MI355_XE_LINES_MAX_ITEMS") ? atol(getenv("MI355_XE_LINES_MAX_ITEMS")) : 64;
    for (long items = (units + cus - 1) / cus; items <= max_items; items++)
        if (units % items == 0 && (units / items) % (8 * G) == 0) return (units / items) * 8 >= (long)cus * 7;  // (the share mi355_xe_lines_launch will find:
    return false;                                                                                     //  on at least 7/8 of the CUs, or the other kernel is faster)
}

// Time ranges (k_xe_i8_lines<true>): the geometry of mi355_xe_lines_ok with FEWER units than compute units -- one window of BASELINE config 5 is 64
// (line, pair group) teams on 256 CUs, the shape of the reference's own one-integration-per-call operator (lib/clXEngine_impl.h:184-201).  Returns the
// number of time ranges (2 or 4: the tail's pieces are quarters of a wave's tiles) or 0.  teams x ranges must fill at least 7/8 of the device and
// never exceed it (every workgroup of a team has to be resident for the in-launch exchange to be the fast path; it is correct either way).
int mi355_xe_lines_split(int N, int F, int Fout, int npol, int T, int stations_per_group, int accumulate, int nint, int cus)
{
    if (getenv("MI355_XE_NO_LINES") || getenv("MI355_XE_NO_LINES_SPLIT")) return 0;
    const int ng = (stations_per_group > 0 && stations_per_group < N) ? stations_per_group : N;
    if (npol != 1 || N != 64 || F % 64 != 0 || Fout != F || T % 32 != 0 || T > 16384 || accumulate || ng % 8 != 0) return 0;
    const long teams = (long)(nint > 0 ? nint : 1) * (F / 64) * 4;
    if (getenv("MI355_XE_LINES_SPLIT_ANY")) {  // test switch: the range count MI355_XE_TSPLIT forces on the plan, whatever the unit count
        const int S = getenv("MI355_XE_TSPLIT") ? atoi(getenv("MI355_XE_TSPLIT")) : 0;
        return ((S == 2 || S == 4) && T % (32 * S) == 0) ? S : 0;
    }
    for (int S = 2; S <= 4; S *= 2)
        if (teams * S <= cus && teams * S * 8 >= (long)cus * 7 && (teams * S) % 32 == 0 && T % (32 * S) == 0 && T / (32 * S) >= 2) return S;
    return 0;
}

int mi355_xe_lines_launch(const void *in, void *out, int N, int F, int Fout, int T, double kd, hipStream_t st, int stations_per_group, int nint, int cus,
                          int tsplit, void *part, size_t flag_offset, unsigned *epoch, int npol)
{
    LnArgs a;
    a.npol = npol == 2 ? 2 : 1;
    const int G = a.npol == 2 ? 8 : 4;
    a.tsplit = (tsplit > 1 && a.npol == 1) ? tsplit : 1;
    a.part = (unsigned char *)part;
    a.flags = (unsigned long long *)((unsigned char *)part + flag_offset);
    a.epoch = (tsplit > 1 && epoch) ? *epoch + 1u : 1u;  // (committed only once the kernel is enqueued)
    a.flag_bank = 0;
    a.wait_ticks = 0;
    a.in = (const unsigned char *)in;
    a.out = (c32 *)out;
    a.Fout = Fout;
    a.T = T;
    a.ncols = F / (64 / a.npol);
    a.row_stride = F * 2 * a.npol;
    a.ng = (stations_per_group > 0 && stations_per_group < N) ? stations_per_group : N;
    const int nw = nint > 0 ? nint : 1;
    a.units = nw * a.ncols * G * a.tsplit;
    a.steps = T / (32 * a.tsplit);
    a.flag_bank = (unsigned)(nw * a.ncols * 4);
    {
        const int us = 10 + 6 * a.steps;  // about one loop of this geometry (see k_xe_i8_fused's tail)
        a.wait_ticks = (unsigned)(us < 20 ? 20 : us > 100 ? 100 : us) * 100u;
        if (const char *e = getenv("MI355_XE_WAIT_US")) a.wait_ticks = (unsigned)atoi(e) * 100u;
    }
    // reference layout: [window][t][station]; group-major: [group][window][t][station in group]
    a.in_window = (size_t)T * a.ng * a.row_stride;
    a.in_group = (size_t)nw * T * a.ng * a.row_stride;
    a.out_window = (size_t)Fout * ((size_t)N * (N + 1) / 2) * a.npol * a.npol;
    a.kd = kd;
    a.k127 = (kd == 0.007874015748031496063 && !getenv("MI355_XE_SCALE_F64")) ? 1 : 0;
    a.dbg = getenv("MI355_XE_DBG") ? atoi(getenv("MI355_XE_DBG")) : 0;
    a.ts = nullptr;
    static std::atomic<unsigned> launch_seq{0};
    a.tag = (launch_seq.fetch_add(1u) + 1u) & 0xfffffu;
    // persistent form: units / grid units per workgroup; the grid a multiple of 32 (pinned map: a workgroup keeps its XCD and group) or of 4
    a.pinned = (a.units % (8 * G) == 0) ? 1 : 0;
    const int quantum = a.pinned ? 8 * G : G;
    int items = (a.units + cus - 1) / cus;
    while (items < a.units && (a.units % items != 0 || (a.units / items) % quantum != 0)) items++;
    if (a.units % items != 0 || (a.units / items) % quantum != 0) items = a.units / quantum;  // (one workgroup quantum: always divides)
    if (a.tsplit > 1) items = 1;  // (time ranges: one unit per workgroup, every workgroup of a team resident where the device is free)
    a.items = items;
    a.pub_local = (getenv("MI355_XE_LINES_PUB") && atoi(getenv("MI355_XE_LINES_PUB")) == 0) ? 0 : 1;
    // early touches (MI355_XE_LINES_PF: distance in K blocks, default 4 -- 2 ... 8 within 1 %, 12 slower --, 0: off): rows of 8, 16 or 32 whole lines in the reference layout
    a.pf_dist = a.pf_lg = a.pf_per = a.pf_items = 0;
    a.pf_mask = a.pf_lines = 0;
    {
        const int pf = getenv("MI355_XE_LINES_PF") ? atoi(getenv("MI355_XE_LINES_PF")) : 4;
        if (pf > 0 && (a.ncols == 8 || a.ncols == 16 || a.ncols == 32) && a.ng == N && ((size_t)in & 127) == 0 && a.steps > pf) {
            const int key = (int)(((size_t)in >> 7) & 15);
            int n = 0;
            for (int l = 0; l < a.ncols; l++)
                if ((((l + key) & 15) & 7) == 3) {
                    a.pf_mask |= 1u << l;
                    a.pf_lines |= (unsigned)l << (8 * n);
                    n++;
                }
            const int touchers = (a.ncols - n) * (a.npol == 2 ? 6 : 2);  // the diagonal groups (two polarisations: the station-tile pairs) of the other lines
            a.pf_items = 32 * 64 * n;
            a.pf_per = (a.pf_items + touchers - 1) / touchers;
            a.pf_lg = n == 1 ? 0 : n == 2 ? 1 : 2;
            if ((n == 1 || n == 2 || n == 4) && a.pf_per <= 256) a.pf_dist = pf;
        }
    }
    // pacing needs the pinned map, every workgroup resident (one per CU) and partners that can be told apart.  Default: 2 half K blocks, with or
    // without the early touches.  Round 5 left it OFF where the touches are on (its boxes: 36.0 / 35.5 / 35.2 / 35.1 us per window unpaced at 4 / 8 /
    // 16 / 32 windows per launch against 37.4-38.5 / 36.7-37.1 / 36.2-36.8 / 35.9-36.5 paced by 2); round 6's boxes say the opposite, and the unpaced
    // form has a bad mode there that the paced one has not: some processes run every launch at 37-41 us per window with 1.6 x the input read
    // (FETCH_SIZE 810-857 K KiB per 8 windows against 620-672 K) -- the diagonal groups drift ahead of their partners and out of the L2; paced by
    // 0 / 1 / 2 / 3 / 4: 35.8 / 35.4 / 35.0 / 35.5 / 35.5 (4 windows), 35.3 / 35.1 / 34.7 / 35.1 / 35.2 (8), 35.0 / 34.8 / 34.5 / 34.7 / 34.8 (16),
    // 37.1 / 34.1 / 34.1 / 34.8 / 35.3 (32), two processes alike (tools/r06_lines_pace_probe.py).  Bounded downside either way: pacing is the choice.
    // (Without touches AND without pacing: 45 / 40 / 43-47 / 46-52.)
    {
        const int pace = getenv("MI355_XE_LINES_PACE") ? atoi(getenv("MI355_XE_LINES_PACE")) : 2;
        a.pace = (a.pinned && a.units / a.items <= cus && a.units / a.items <= 8192 && (long)a.items * a.steps < 2048 && !(a.dbg & 8)) ? pace : 0;
        if (a.tsplit > 1) a.pace = 0;  // (the tail keeps its two words where the partners' progress words land)
        if (a.npol == 2) a.pace = 0;   // (teams of eight: not paced)
    }
    static std::atomic<unsigned long long> attr_devs{0};  // (per device: a function's attributes belong to the device that is current when they are set)
    {
        int dev = 0;
        MI355_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_i8_lines<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLnLds));
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_i8_lines<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLnLds));
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_i8_lines<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLnLds));
            attr_devs.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const unsigned grid = (unsigned)(a.units / a.items);
    a.grid = (int)grid;
    a.rot = (a.pinned && a.items > 1 && (grid / G) % (unsigned)a.ncols == 0) ? (getenv("MI355_XE_LINES_ROT") ? atoi(getenv("MI355_XE_LINES_ROT")) : 1) : 0;
    // MI355_XE_FAIL_LAUNCH (test switch): fail where a bad stream handle or an exhausted device would -- nothing enqueued, an error returned
    if (a.tsplit > 1 && getenv("MI355_XE_FAIL_LAUNCH")) {
        mi355_set_error("whole-line X-engine launch failed (MI355_XE_FAIL_LAUNCH)");
        return MI355_ERR_HIP;
    }
    mi355_xe_route_set(a.npol == 2 ? "k_xe_i8_lines<2 pol>" : a.tsplit > 1 ? "k_xe_i8_lines<split>" : "k_xe_i8_lines", nw, (int)grid, a.items, a.tsplit, a.tsplit > 1 ? 1 : 0, a.pf_dist, a.pace);
    if (getenv("MI355_XE_TS")) {  // tuning aid: one synchronous launch with start / end stamps per workgroup
        unsigned long long *d_ts = nullptr;
        MI355_HIP(hipMalloc(&d_ts, (size_t)grid * 64));
        MI355_HIP(hipMemsetAsync(d_ts, 0, (size_t)grid * 64, st));
        a.ts = d_ts;
        if (a.npol == 2) hipLaunchKernelGGL((k_xe_i8_lines<false, 2>), dim3(grid), dim3(kLnThreads), kLnLds, st, a);
        else if (a.tsplit > 1) hipLaunchKernelGGL(k_xe_i8_lines<true>, dim3(grid), dim3(kLnThreads), kLnLds, st, a);
        else hipLaunchKernelGGL(k_xe_i8_lines<false>, dim3(grid), dim3(kLnThreads), kLnLds, st, a);
        MI355_HIP(hipGetLastError());
        if (a.tsplit > 1 && epoch) *epoch = a.epoch;
        MI355_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)grid * 8);
        MI355_HIP(hipMemcpy(h.data(), d_ts, (size_t)grid * 64, hipMemcpyDeviceToHost));
        (void)hipFree(d_ts);
        unsigned long long t0 = ~0ull;
        for (unsigned b = 0; b < grid; b++) t0 = std::min(t0, h[8 * b]);
        double end_by_grp[4] = {0, 0, 0, 0}, end_by_col[64] = {0}, last = 0;
        int n_grp[4] = {0, 0, 0, 0}, n_col[64] = {0};
        for (unsigned b = 0; b < grid; b++) {
            const int gbits = a.npol == 2 ? 3 : 2;
            const int within = a.pinned ? (int)(b >> 3) : (int)b, grp = (within & ((1 << gbits) - 1)) & 3,
                      combo = a.pinned ? (int)((b & 7) + 8 * (within >> gbits)) : (int)(b >> gbits);
            const double e = (double)(h[8 * b + 1] - t0) * 0.01;
            end_by_grp[grp] += e; n_grp[grp]++;
            end_by_col[(combo % a.ncols) & 63] += e; n_col[(combo % a.ncols) & 63]++;
            last = std::max(last, e);
        }
        if (a.tsplit > 1) {  // time ranges: the phases of a workgroup, us after the first start (min / median / max)
            static const char *names[5] = {"start", "loop end", "pieces sent", "all ranges in", "end"};
            const int idx[5] = {0, 5, 6, 7, 1};
            for (int k = 0; k < 5; k++) {
                std::vector<double> v;
                for (unsigned b = 0; b < grid; b++) if (h[8 * b + idx[k]]) v.push_back((double)(h[8 * b + idx[k]] - t0) * 0.01);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                fprintf(stderr, "  %-14s %7.2f %7.2f %7.2f\n", names[k], v.front(), v[v.size() / 2], v.back());
            }
            for (int ty = 0; ty < 2; ty++) {
                std::vector<double> v;
                for (unsigned b = 0; b < grid; b++) if ((((a.pinned ? (b >> 3) : b) & 3) >> 1) == (unsigned)ty && h[8 * b + 5]) v.push_back((double)(h[8 * b + 5] - t0) * 0.01);
                std::sort(v.begin(), v.end());
                if (!v.empty()) fprintf(stderr, "  loop end, %s groups: %7.2f %7.2f %7.2f\n", ty ? "off-diagonal" : "diagonal", v.front(), v[v.size() / 2], v.back());
            }
        }
        if (a.tsplit == 1) {  // the last unit's loop end, by group type (us after the first start: min / median / max)
            for (int ty = 0; ty < 2; ty++) {
                std::vector<double> v;
                for (unsigned b = 0; b < grid; b++) {
                    const int within = a.pinned ? (int)(b >> 3) : (int)b, grp = within & (a.npol == 2 ? 7 : 3);
                    if ((grp < 2 ? 0 : 1) == ty && h[8 * b + 5]) v.push_back((double)(h[8 * b + 5] - t0) * 0.01);
                }
                std::sort(v.begin(), v.end());
                if (!v.empty()) fprintf(stderr, "  loop end, %s groups: %7.2f %7.2f %7.2f\n", ty ? "off-diagonal" : "diagonal", v.front(), v[v.size() / 2], v.back());
            }
        }
        fprintf(stderr, "[xe lines stamps] %u workgroups x %d units, last end %.1f us; mean end by group:", grid, a.items, last);
        for (int k = 0; k < 4; k++) fprintf(stderr, " %.1f", n_grp[k] ? end_by_grp[k] / n_grp[k] : 0.0);
        fprintf(stderr, "; by line:");
        for (int k = 0; k < a.ncols && k < 64; k++) fprintf(stderr, " %.0f", n_col[k] ? end_by_col[k] / n_col[k] : 0.0);
        fprintf(stderr, "\n");
        // shader clocks of wave 0 in the parts of front(), by group type (diagonal groups 0 / 1, off-diagonal 2 / 3), as a share of the kernel's clocks
        for (int ty = 0; ty < 2; ty++) {
            double w = 0, b1 = 0, pc = 0, dur = 0; int n = 0;
            for (unsigned b = 0; b < grid; b++) {
                const int within = a.pinned ? (int)(b >> 3) : (int)b, grp = within & (a.npol == 2 ? 7 : 3);
                if ((grp >> 1) != ty) continue;
                w += (double)h[8 * b + 2]; b1 += (double)h[8 * b + 3]; pc += (double)h[8 * b + 4];
                dur += (double)(h[8 * b + 1] - h[8 * b]) * 0.01; n++;
            }
            if (n) fprintf(stderr, "  %s groups: %.0f us per workgroup; clocks of wave 0 (k): DMA wait %.0f, barrier %.0f; paced %.1f us\n",
                           ty ? "off-diagonal" : "diagonal", dur / n, w / n / 1e3, b1 / n / 1e3, pc / n * 0.01);
        }
        return MI355_OK;
    }
    if (a.npol == 2) hipLaunchKernelGGL((k_xe_i8_lines<false, 2>), dim3(grid), dim3(kLnThreads), kLnLds, st, a);
    else if (a.tsplit > 1) hipLaunchKernelGGL(k_xe_i8_lines<true>, dim3(grid), dim3(kLnThreads), kLnLds, st, a);
    else hipLaunchKernelGGL(k_xe_i8_lines<false>, dim3(grid), dim3(kLnThreads), kLnLds, st, a);
    MI355_HIP(hipGetLastError());
    if (a.tsplit > 1 && epoch) *epoch = a.epoch;
    return MI355_OK;
}
