// clXEngine (FX-correlator X-engine) as gfx950 HIP kernels.
// Reference behaviour: lib/clXEngine_impl.h:150-201 (xcorrelate overloads: H2D, CharToComplex
// kernel into a 4x larger float copy, then one work-item per (channel, baseline) walking t with
// stride-frame_size loads), kernels lib/clXEngine_impl.cc:605-916, host gather :987-1061.
//   V[f][k][p1 p2] = sum_t x_{s1,p1}(t,f) * conj(x_{s2,p2}(t,f)),  k = s1(s1+1)/2 + s2, s1 >= s2
//
// int8 (IChar) and packed-4-bit inputs run on the matrix cores with EXACT integer arithmetic:
//   rows r = (station, pol); planes I_r(t), Q_r(t) (de-interleaved, so no byte negation is needed
//   and -128 is handled exactly);  per 16x16 row-tile pair (bi >= bj), K = time:
//       re  += I_bi I_bj^T + Q_bi Q_bj^T      (one accumulator)
//       u   += Q_bi I_bj^T ,  w += I_bi Q_bj^T   ->  im = u - w
//   with v_mfma_i32_16x16x64_i8 (K = 64 time steps per instruction), int32 accumulators, and one
//   double-precision scale by (1/127)^2 (or (1/7)^2) at the end -- bit-identical to the oracle's
//   exact path.  4 real MACs per complex MAC: no redundancy beyond the diagonal tiles.
// Two kernels: (1) k_xe_turn: corner turn of the reference's [t][station][chan][pol]{I,Q} buffer
//   into MFMA-operand order  [chan][kblock][plane][rowtile][lane*16 B]  (a tile = 16 rows x 64 t
//   = 1 KiB, exactly one dwordx4 per lane, so the correlator streams operands with unit stride);
//   reads are whole 128-B lines, the byte transposition is done in registers with v_perm_b32;
// (2) k_xe_corr: one workgroup per (channel, chunk of 12 tile pairs), 4 waves, 3 pairs per wave.
// HBM traffic today: input read + tile write + tile read + output write (2.8x the algorithmic
// bytes); fusing (1) into (2) or producing tile order in the host gather is the next step.
//
// Complex-float input keeps fp32 arithmetic (register-tiled VALU kernel, 8x8 station blocks).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"
#include "xengine_fused.h"

namespace {

struct c32 { float x, y; };
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kRowTile = 16, kKBlock = 64, kTileBytes = kRowTile * kKBlock;  // 1 KiB

// ------------------------------------------------------------------------------------
// (1) corner turn.  Work item = (16 consecutive t) x (one 4-byte unit of the input row).
//   IChar npol=1 : unit = channels (2u, 2u+1) of station s  -> samples a=(f0,row s), b=(f1,row s)
//   IChar npol=2 : unit = channel u, pols X,Y of station s   -> a=(f,row 2s), b=(f,row 2s+1)
//   packed 4-bit : unit = channels (2u, 2u+1), bytes X,Y      -> four samples, rows 2s, 2s+1
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 4 dwords (bytes b0..b3 each) -> 4 dwords where out[j] = byte j of each input dword (4x4 byte transpose)
__device__ __forceinline__ void transpose4x4(const unsigned (&in)[4], unsigned (&out)[4])
{
    // v_perm_b32 selector: bytes 0-3 = lo operand, 4-7 = hi operand
    const unsigned t0 = perm(in[1], in[0], 0x05010400u);  // in0.b0 in1.b0 in0.b1 in1.b1
    const unsigned t1 = perm(in[1], in[0], 0x07030602u);  // in0.b2 in1.b2 in0.b3 in1.b3
    const unsigned t2 = perm(in[3], in[2], 0x05010400u);
    const unsigned t3 = perm(in[3], in[2], 0x07030602u);
    out[0] = perm(t2, t0, 0x05040100u);  // b0 of in0..in3
    out[1] = perm(t2, t0, 0x07060302u);  // b1
    out[2] = perm(t3, t1, 0x05040100u);  // b2
    out[3] = perm(t3, t1, 0x07060302u);  // b3
}

struct XeGeo {
    int N, F, npol, T;     // stations, channels, pols, integration frames
    int A, NT, KB;         // rows = N*npol, row tiles, K blocks of 64 time steps
    int mode;              // 0 = IChar, 1 = packed 4-bit
    int f0, Fs;            // channel slab handled by this launch: [f0, f0 + Fs); the tile workspace holds one slab
    int Fout;              // channels of the caller's matrices (F - 1 when one zero channel pads the rows to 4-byte units)
};

__device__ __forceinline__ size_t tile_off(const XeGeo &g, int f, int kb, int plane, int rt)
{
    return ((((size_t)(f - g.f0) * g.KB + kb) * 2 + plane) * g.NT + rt) * kTileBytes;
}

// sign-extended 4-bit code with the reference's LUT quirk (code 8 -> 0), for 4 packed bytes
__device__ __forceinline__ unsigned nib_to_i8x4(unsigned n /* one nibble per byte, 0..15 */)
{
    // v = n < 8 ? n : (n == 8 ? 0 : n - 16)
    const unsigned ge8 = (n >> 3) & 0x01010101u;            // 1 where n >= 8
    const unsigned low = n & 0x07070707u;                   // n - 8 where n >= 8
    const unsigned nz = ((low + 0x07070707u) >> 3) & 0x01010101u;  // 1 where low != 0
    const unsigned neg = ge8 & nz;                          // 1 where 9..15
    // negative: low - 8 = low | 0xF8 ; n == 8: 0 ; else n
    const unsigned pos = n & ~(ge8 * 0xFFu);
    return pos | (neg * 0xF8u) | (low & (neg * 0xFFu));
}

__global__ __launch_bounds__(256) void k_xe_turn(const unsigned *__restrict__ in, unsigned char *__restrict__ tiles, XeGeo g)
{
    // grid.x = ceil(units_per_row/32), grid.y = ceil(stations/2), grid.z = KB
    // block = 32 units (one 128-byte line of the input row) x 4 time chunks x 2 stations
    const int lane_u = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;  // 0..7
    const int kc = sub & 3;
    const int units_per_row = (g.mode == 0) ? (g.F * g.npol * 2) / 4 : (g.F * 2) / 4;  // 4-byte units per (t, station)
    const int upc = (g.mode == 0 && g.npol == 2) ? 1 : 2;  // channels per 4-byte unit
    const int u = g.f0 / upc + blockIdx.x * 32 + lane_u;
    const int s = blockIdx.y * 2 + (sub >> 2), kb = blockIdx.z;
    if (u >= (g.f0 + g.Fs) / upc || u >= units_per_row || s >= g.N) return;
    const size_t row_units = (size_t)units_per_row;
    {
        unsigned w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int t = kb * kKBlock + kc * 16 + i;
            w[i] = (t < g.T) ? in[((size_t)t * g.N + s) * row_units + u] : 0u;
        }
        // byte-transpose 16 dwords -> 4 vectors of 16 bytes (byte j of every dword)
        unsigned col[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned a[4] = {w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            unsigned o[4];
            transpose4x4(a, o);
#pragma unroll
            for (int j = 0; j < 4; j++) col[j][q] = o[j];
        }
        if (g.mode == 0) {
            // bytes: a.I a.Q b.I b.Q
            int fa, fb, ra, rb;
            if (g.npol == 1) { fa = 2 * u; fb = 2 * u + 1; ra = rb = s; }
            else { fa = fb = u; ra = 2 * s; rb = 2 * s + 1; }
            const int f_[2] = {fa, fb}, r_[2] = {ra, rb};
#pragma unroll
            for (int smp = 0; smp < 2; smp++) {
#pragma unroll
                for (int plane = 0; plane < 2; plane++) {
                    unsigned char *dst = tiles + tile_off(g, f_[smp], kb, plane, r_[smp] / kRowTile) +
                                         (size_t)(kc * 16 + (r_[smp] % kRowTile)) * 16;
                    *(uint4 *)dst = make_uint4(col[smp * 2 + plane][0], col[smp * 2 + plane][1], col[smp * 2 + plane][2],
                                               col[smp * 2 + plane][3]);
                }
            }
        } else {
            // bytes: (f0,X) (f0,Y) (f1,X) (f1,Y), each byte = re nibble (high), im nibble (low)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int f = 2 * u + (b >> 1), r = 2 * s + (b & 1);
                unsigned re[4], im[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    re[q] = nib_to_i8x4((col[b][q] >> 4) & 0x0F0F0F0Fu);
                    im[q] = nib_to_i8x4(col[b][q] & 0x0F0F0F0Fu);
                }
                unsigned char *d0 = tiles + tile_off(g, f, kb, 0, r / kRowTile) + (size_t)(kc * 16 + (r % kRowTile)) * 16;
                unsigned char *d1 = tiles + tile_off(g, f, kb, 1, r / kRowTile) + (size_t)(kc * 16 + (r % kRowTile)) * 16;
                *(uint4 *)d0 = make_uint4(re[0], re[1], re[2], re[3]);
                *(uint4 *)d1 = make_uint4(im[0], im[1], im[2], im[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// (1b) fast corner turn for IChar when an input row (t, station) is a whole number of 128-byte
// lines.  One workgroup = 16 rows (one row tile) x 16 time steps (one kc chunk) x one 128-byte
// line of channels = 32 KiB.  Load phase: whole lines, 16 B per lane, into LDS (station stride
// padded so the transposed reads are conflict free).  Store phase: lanes run over the tile rows
// first, so every store instruction writes 256 contiguous bytes (16 rows x 16 B) of each tile.
// ------------------------------------------------------------------------------------
template <int NPOL, bool PACKED>  // PACKED: 4-bit X,Y bytes (rows 2s, 2s+1), two channels per 4-byte unit
__global__ __launch_bounds__(256) void k_xe_turn_lds(const uint4 *__restrict__ in, unsigned char *__restrict__ tiles, XeGeo g)
{
    static_assert(!PACKED || NPOL == 2, "packed input is dual polarisation");
    constexpr int SIN = kRowTile / NPOL;                 // stations per row tile
    constexpr int SSTRIDE = 2048 + (NPOL == 1 ? 8 : 16);  // bytes between stations in LDS (bank spread)
    __shared__ __attribute__((aligned(16))) unsigned char lds[SIN * SSTRIDE];
    const int cpl = PACKED ? 64 : 64 / NPOL;  // channels per 128-byte line
    const int line = g.f0 / cpl + blockIdx.x, rt = blockIdx.y, kb = blockIdx.z >> 2, kc = blockIdx.z & 3;
    const int tid = threadIdx.x;
    const int lines_per_row = PACKED ? (g.F * 2) / 128 : (g.F * NPOL * 2) / 128;
    const int t0 = kb * kKBlock + kc * 16;
    // ---- load: SIN stations x 16 t rows of 128 B; 8 lanes per row ----
#pragma unroll
    for (int it = 0; it < (SIN * 16 * 8) / 256; it++) {
        const int idx = tid + 256 * it, seg = idx & 7, row = idx >> 3, t = row & 15, sl = row >> 4;
        const int s = rt * SIN + sl;
        v4i v = (v4i){0, 0, 0, 0};
        if (s < g.N && t0 + t < g.T)
            v = __builtin_nontemporal_load((const v4i *)in + (((size_t)(t0 + t) * g.N + s) * lines_per_row + line) * 8 + seg);
        *(v4i *)(lds + sl * SSTRIDE + t * 128 + seg * 16) = v;
    }
    __syncthreads();
    // ---- transpose + store: item = (station sl, 4-byte unit u of the line) ----
#pragma unroll
    for (int it = 0; it < (SIN * 32) / 256; it++) {
        const int item = tid + 256 * it, sl = item % SIN, u = item / SIN;
        unsigned w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) w[t] = *(const unsigned *)(lds + sl * SSTRIDE + t * 128 + u * 4);
        unsigned col[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned a[4] = {w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            unsigned o[4];
            transpose4x4(a, o);
#pragma unroll
            for (int j = 0; j < 4; j++) col[j][q] = o[j];
        }
        const int U = line * 32 + u;
        if constexpr (PACKED) {
            // bytes of a unit: (f0,X) (f0,Y) (f1,X) (f1,Y); byte = re nibble (high) | im nibble (low)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int f = 2 * U + (b >> 1), rr = 2 * sl + (b & 1);
                unsigned re[4], im[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    re[q] = nib_to_i8x4((col[b][q] >> 4) & 0x0F0F0F0Fu);
                    im[q] = nib_to_i8x4(col[b][q] & 0x0F0F0F0Fu);
                }
                *(uint4 *)(tiles + tile_off(g, f, kb, 0, rt) + (size_t)(kc * 16 + rr) * 16) = make_uint4(re[0], re[1], re[2], re[3]);
                *(uint4 *)(tiles + tile_off(g, f, kb, 1, rt) + (size_t)(kc * 16 + rr) * 16) = make_uint4(im[0], im[1], im[2], im[3]);
            }
        } else {
            // bytes of a unit: a.I a.Q b.I b.Q ;  NPOL 1: a,b = channels 2U,2U+1 of row sl ; NPOL 2: a,b = rows 2sl,2sl+1 of channel U
#pragma unroll
            for (int smp = 0; smp < 2; smp++) {
                const int f = (NPOL == 1) ? 2 * U + smp : U;
                const int rr = (NPOL == 1) ? sl : 2 * sl + smp;
#pragma unroll
                for (int plane = 0; plane < 2; plane++) {
                    uint4 o = make_uint4(col[smp * 2 + plane][0], col[smp * 2 + plane][1], col[smp * 2 + plane][2], col[smp * 2 + plane][3]);
                    *(uint4 *)(tiles + tile_off(g, f, kb, plane, rt) + (size_t)(kc * 16 + rr) * 16) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// (2) MFMA correlator
// ------------------------------------------------------------------------------------
constexpr int kPairsPerWave = 3, kWaves = 4, kPairsPerWG = kPairsPerWave * kWaves;

__device__ __forceinline__ void pair_to_tiles(int p, int &bi, int &bj)
{
    // p = bi(bi+1)/2 + bj, bi >= bj
    int a = (int)((-1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= p) a++;
    while (a * (a + 1) / 2 > p) a--;
    bi = a;
    bj = p - a * (a + 1) / 2;
}

__global__ __launch_bounds__(256) void k_xe_corr(const unsigned char *__restrict__ tiles, c32 *__restrict__ out, XeGeo g,
                                                 int npairs, double scale2, int accumulate)
{
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bi[kPairsPerWave], bj[kPairsPerWave];
    bool live[kPairsPerWave];
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) {
        const int p = chunk * kPairsPerWG + q * kWaves + wave;
        live[q] = p < npairs;
        pair_to_tiles(live[q] ? p : 0, bi[q], bj[q]);
    }
    v4i re[kPairsPerWave], uu[kPairsPerWave], ww[kPairsPerWave];
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) re[q] = uu[q] = ww[q] = (v4i){0, 0, 0, 0};

    const unsigned char *base = tiles + (size_t)f * g.KB * 2 * g.NT * kTileBytes + (size_t)lane * 16;  // f = channel within the slab
    const size_t plane_stride = (size_t)g.NT * kTileBytes;
    for (int kb = 0; kb < g.KB; kb++) {
        const unsigned char *pI = base + (size_t)kb * 2 * plane_stride;
        const unsigned char *pQ = pI + plane_stride;
#pragma unroll
        for (int q = 0; q < kPairsPerWave; q++) {
            if (live[q]) {  // wave-uniform
                const v4i Ia = *(const v4i *)(pI + (size_t)bi[q] * kTileBytes);
                const v4i Qa = *(const v4i *)(pQ + (size_t)bi[q] * kTileBytes);
                const v4i Ib = *(const v4i *)(pI + (size_t)bj[q] * kTileBytes);
                const v4i Qb = *(const v4i *)(pQ + (size_t)bj[q] * kTileBytes);
                re[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ia, Ib, re[q], 0, 0, 0);
                re[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qa, Qb, re[q], 0, 0, 0);
                uu[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qa, Ib, uu[q], 0, 0, 0);
                ww[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ia, Qb, ww[q], 0, 0, 0);
            }
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int nb = g.N * (g.N + 1) / 2, np2 = g.npol * g.npol;
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) {
        if (!live[q]) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r1 = bi[q] * kRowTile + (lane >> 4) * 4 + reg, r2 = bj[q] * kRowTile + (lane & 15);
            if (r1 >= g.A || r2 >= g.A) continue;
            const int s1 = r1 / g.npol, p1 = r1 - s1 * g.npol, s2 = r2 / g.npol, p2 = r2 - s2 * g.npol;
            if (s1 < s2) continue;
            const int k = s1 * (s1 + 1) / 2 + s2;
            if (f + g.f0 >= g.Fout) continue;  // the padding channel has no output
            const size_t o = ((size_t)(f + g.f0) * nb + k) * np2 + p1 * g.npol + p2;
            // same expression as the oracle's exact path: (double)S * kd * kd, rounded once
            const double kd = scale2;  // 1/127 or 1/7
            c32 v;
            // the only int32 wrap-around possible up to 65536 frames: re == +2^31 (every sample -128), which reads back as INT_MIN
            v.x = (float)((re[q][reg] == (int)0x80000000 ? 2147483648.0 : (double)re[q][reg]) * kd * kd);
            v.y = (float)(((double)uu[q][reg] - (double)ww[q][reg]) * kd * kd);
            if (accumulate) { v.x += out[o].x; v.y += out[o].y; }
            out[o] = v;
        }
    }
}

// (2b) LDS-staged correlator: the workgroup stages each K block's tiles (2 planes x NT tiles) in LDS
// once (coalesced 16 B per lane, register prefetch of the next K block) and the four waves read
// their MFMA operands from LDS instead of each re-reading them through L1.
template <int NTT, int WAVES, int PPW>  // row tiles; waves per workgroup; tile pairs per wave (WAVES*PPW >= pairs => one WG per channel)
__global__ __launch_bounds__(WAVES * 64) void k_xe_corr_lds(const unsigned char *__restrict__ tiles, c32 *__restrict__ out, XeGeo g,
                                                            int npairs, double scale2, int accumulate)
{
    constexpr int kPairsPerWave = PPW, kWaves = WAVES, kPairsPerWG = PPW * WAVES, NTHR = WAVES * 64;
    constexpr int KBYTES = 2 * NTT * kTileBytes;        // bytes per K block
    constexpr int PER_THREAD = KBYTES / (NTHR * 16);     // dwordx4 loads per thread per K block
    static_assert(KBYTES % (NTHR * 16) == 0 && PER_THREAD >= 1, "tile bytes per K block must split over the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][KBYTES];
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int bi[kPairsPerWave], bj[kPairsPerWave];
    bool live[kPairsPerWave];
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) {
        const int p = chunk * kPairsPerWG + q * kWaves + wave;
        live[q] = p < npairs;
        pair_to_tiles(live[q] ? p : 0, bi[q], bj[q]);
    }
    v4i re[kPairsPerWave], uu[kPairsPerWave], ww[kPairsPerWave];
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) re[q] = uu[q] = ww[q] = (v4i){0, 0, 0, 0};

    const unsigned char *src = tiles + (size_t)f * g.KB * KBYTES + (size_t)tid * 16;
    v4i stage[PER_THREAD];
#pragma unroll
    for (int i = 0; i < PER_THREAD; i++) stage[i] = __builtin_nontemporal_load((const v4i *)(src + (size_t)i * NTHR * 16));
    for (int kb = 0; kb < g.KB; kb++) {
        unsigned char *buf = lds[kb & 1];
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++) *(v4i *)(buf + tid * 16 + i * NTHR * 16) = stage[i];
        if (kb + 1 < g.KB) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; i++)
                stage[i] = __builtin_nontemporal_load((const v4i *)(src + (size_t)(kb + 1) * KBYTES + (size_t)i * NTHR * 16));
        }
        __syncthreads();  // one barrier per K block: the other buffer was last read before the previous barrier
        const unsigned char *pI = buf + lane * 16, *pQ = pI + NTT * kTileBytes;
#pragma unroll
        for (int q = 0; q < kPairsPerWave; q++) {
            if (live[q]) {  // wave-uniform
                const v4i Ia = *(const v4i *)(pI + bi[q] * kTileBytes);
                const v4i Qa = *(const v4i *)(pQ + bi[q] * kTileBytes);
                const v4i Ib = *(const v4i *)(pI + bj[q] * kTileBytes);
                const v4i Qb = *(const v4i *)(pQ + bj[q] * kTileBytes);
                re[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ia, Ib, re[q], 0, 0, 0);
                re[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qa, Qb, re[q], 0, 0, 0);
                uu[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qa, Ib, uu[q], 0, 0, 0);
                ww[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ia, Qb, ww[q], 0, 0, 0);
            }
        }
    }
    const int nb = g.N * (g.N + 1) / 2, np2 = g.npol * g.npol;
#pragma unroll
    for (int q = 0; q < kPairsPerWave; q++) {
        if (!live[q]) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r1 = bi[q] * kRowTile + (lane >> 4) * 4 + reg, r2 = bj[q] * kRowTile + (lane & 15);
            if (r1 >= g.A || r2 >= g.A) continue;
            const int s1 = r1 / g.npol, p1 = r1 - s1 * g.npol, s2 = r2 / g.npol, p2 = r2 - s2 * g.npol;
            if (s1 < s2) continue;
            const int k = s1 * (s1 + 1) / 2 + s2;
            if (f + g.f0 >= g.Fout) continue;  // the padding channel has no output
            const size_t o = ((size_t)(f + g.f0) * nb + k) * np2 + p1 * g.npol + p2;
            const double kd = scale2;
            c32 v;
            // the only int32 wrap-around possible up to 65536 frames: re == +2^31 (every sample -128), which reads back as INT_MIN
            v.x = (float)((re[q][reg] == (int)0x80000000 ? 2147483648.0 : (double)re[q][reg]) * kd * kd);
            v.y = (float)(((double)uu[q][reg] - (double)ww[q][reg]) * kd * kd);
            if (accumulate) { v.x += out[o].x; v.y += out[o].y; }
            out[o] = v;
        }
    }
}

// (2c) Large arrays (65 .. 256 rows: 5 .. 16 row tiles, padded to an even count NT).  A PERSISTENT workgroup (one per CU at 16 tiles) walks a contiguous
// run of channels; a channel's whole lower triangle is accumulated in the workgroup's registers in ONE pass over its K blocks (time
// is not split, nothing is re-read):
//   * the K-block stream of a channel run is one contiguous byte range of the tile workspace ([chan][kblock][plane][rowtile][1 KiB]):
//     it is pulled global -> LDS by DMA (global_load_lds_dwordx4, one instruction = one 1 KiB tile, no staging registers) into a
//     ring of four K blocks, three in flight while one is multiplied; the ring runs straight across channel boundaries, so the
//     next channel's operands arrive while the finished channel's matrix is scaled and stored;
//   * NT / 2 waves; wave w owns the tile rows w and NT-1-w of the triangle: (w + 1) + (NT - w) = NT + 1 tile pairs for EVERY wave
//     (17 at 256 rows), i.e. the matrix-core work is balanced exactly.  Its two row operands stay in registers for the K block,
//     the column tiles are streamed from LDS two tiles ahead of their use (hand-issued ds_read_b128 with immediate offsets and
//     explicit lgkmcnt waits: left to the compiler every column tile went through one register quad and each group of MFMAs
//     waited out an LDS round trip), each feeding 4 or 8 v_mfma_i32_16x16x64_i8;
//   * two accumulators per pair: re += I_a I_b^T + Q_a Q_b^T, im' += Q_a I_b^T + (~I_a) Q_b^T with ~i = -i - 1 (exact for every int8,
//     one v_not per row operand and K block), im = im' + sum_t Q_b(t): every wave sums Q over its own two tile rows (v_sad_u8 on
//     the operand bytes) and the sums are exchanged through LDS when a channel is complete;
//   * output: 16 lanes x 8 B = one contiguous 128-byte run per accumulator register (tools/ubench/tri_store.hip: this shape writes
//     the 135 MB of 512 channels x 256 rows in 23 us = 5.9 TB/s, the column-operand-first shape -- 32 contiguous bytes per lane,
//     rows across lanes -- in 30 us).  The write burst at the end of a channel is what the kernel cannot hide: a CU holds one
//     channel's accumulators (1088 of its 2048 registers), so every CU stores at the same time.
// Per channel at 256 rows: 16 K blocks x 32 KiB from HBM against 16 x 136 x 4 MFMAs (2176 cycles per K block and SIMD): the kernel
// is bound by the tile stream (268 MB at 512 channels) and the matrix store (135 MB), not by the matrix cores.
__device__ __forceinline__ void sb_dma_tile(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// one 16-byte LDS read whose completion the CALLER waits for (s_waitcnt lgkmcnt): the result must not be touched before that
template <int OFF> __device__ __forceinline__ void sb_lds_read(v4i &dst, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int... Js, class F> __device__ __forceinline__ void sb_unroll(std::integer_sequence<int, Js...>, F &&f)
{
    (f(std::integral_constant<int, Js>{}), ...);
}

struct SbArgs {
    const unsigned char *tiles;
    c32 *out;
    XeGeo g;      // g.NT = even (padded) row-tile count
    double kd;
    int accumulate;
    int chan_per_wg;  // channels of the slab per workgroup (the last workgroups may get fewer / none)
    int dbg;          // tuning aid (MI355_XE_DBG): 1 = no matrix products, 2 = no output, 4 = no DMA
};

// Output side.  Lane (r = lane % 16, q = lane / 16) holds, for pair (bi, bj), matrix rows r1 = 16 bi + 4 q + reg and column r2 = 16 bj + r.
// Element offsets within a channel's block are (row part) + (column part):
//   one polarisation:  r1 (r1 + 1) / 2            +  r2
//   two:               4 (s1 (s1 + 1) / 2) + 2 p1  +  4 (r2 / 2) + (r2 & 1)      (row r1 = 2 s1 + p1, column r2 = 2 s2 + p2)
template <int NPOL> struct SbRow {
    c32 *ptr[4];   // out + channel block + row part + the lane's column part within a tile, per accumulator register
    bool live[4];  // r1 < A
    int s1[4];
};

template <int NPOL> __device__ __forceinline__ void sb_row_setup(SbRow<NPOL> &row, const SbArgs &a, c32 *chan, int bi, int lane)
{
    const int r = lane & 15, q = lane >> 4;
    const int lanecol = (NPOL == 1) ? r : 4 * (r >> 1) + (r & 1);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int r1 = bi * kRowTile + 4 * q + reg, s1 = r1 / NPOL, p1 = r1 % NPOL;
        row.s1[reg] = s1;
        row.live[reg] = r1 < a.g.A;
        row.ptr[reg] = chan + ((NPOL == 1) ? r1 * (r1 + 1) / 2 : 4 * (s1 * (s1 + 1) / 2) + 2 * p1) + lanecol;
    }
}

template <int NPOL, bool DIAG>
__device__ __forceinline__ void sb_store_pair(const SbArgs &a, const SbRow<NPOL> &row, int bj, v4i vre, v4i vim, const int *colsum, int lane)
{
    const int r2 = bj * kRowTile + (lane & 15);
    const int cs = colsum[r2];  // sum_t Q of the lane's column (LDS)
    const bool col_live = r2 < a.g.A;
    c32 v[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        // int32 wrap-around can only have happened for re == +2^31 (every sample -128 over 65536 frames)
        const double dre = (vre[reg] == (int)0x80000000) ? 2147483648.0 : (double)vre[reg];
        v[reg].x = (float)(dre * a.kd * a.kd);  // the oracle's expression: (double)S * kd * kd, rounded once
        v[reg].y = (float)((double)(vim[reg] + cs) * a.kd * a.kd);
    }
    if (a.dbg & 16) {  // tuning aid: the arithmetic without the stores
        if (v[0].x == 1.2345e-30f && v[3].y == 5.4321e-30f) *row.ptr[0] = v[1];
        return;
    }
    if constexpr (NPOL == 2) {
        // Registers (0, 1) and (2, 3) are the two polarisations p1 of one station s1, lanes (r, r ^ 1) the two p2 of one station s2: the
        // four products of a baseline are 32 contiguous bytes [p1 p2] = 00 01 10 11.  A lane pair swaps one value, then the even lane
        // holds 00 01 and the odd lane 10 11: one 16-byte store each, and 16 lanes cover 256 contiguous bytes.
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        const bool odd = (lane & 1) != 0;
        if (!a.accumulate) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const c32 mine = odd ? v[2 * h + 1] : v[2 * h], give = odd ? v[2 * h] : v[2 * h + 1];
                c32 got;
                got.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give.x), 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
                got.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give.y), 0xB1, 0xF, 0xF, false));
                bool ok = row.live[2 * h] && col_live;
                if (DIAG) ok = ok && row.s1[2 * h] >= r2 / 2;
                // even lane: [00 (mine), 01 (from the odd lane)] at the baseline's start; odd lane: [10 (from the even lane), 11 (mine)] 16 bytes on
                c32 *dst = row.ptr[2 * h] + bj * (kRowTile * 2) - (odd ? 1 : 0) + (odd ? 2 : 0);
                if (ok) *(v4f_ *)dst = odd ? (v4f_){got.x, got.y, mine.x, mine.y} : (v4f_){mine.x, mine.y, got.x, got.y};
            }
            return;
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        bool ok = row.live[reg] && col_live;
        if (DIAG) ok = ok && row.s1[reg] >= r2 / NPOL;  // station s1 >= s2 (off the diagonal tiles that holds by construction)
        c32 *dst = row.ptr[reg] + bj * (kRowTile * NPOL);
        if (ok) {
            c32 w = v[reg];
            if (a.accumulate) { w.x += dst->x; w.y += dst->y; }
            *dst = w;
        }
    }
}

// the whole K loop of wave W (tile rows W and NT-1-W): the accumulators never leave the registers
template <int NT, int W, int NPOL>
__device__ __forceinline__ void sb_wave(const SbArgs &a, unsigned char *lds, int c0, int nblk, int lane)
{
    constexpr int WAVES = NT / 2, RING = 4, KBYTES = 2 * NT * kTileBytes, PER = 4;  // DMA instructions per wave and K block: 2 NT / WAVES
    constexpr int RA = W, RB = NT - 1 - W;  // RB > RA
    constexpr int QOFF = NT * kTileBytes;   // plane Q of a K block
    const XeGeo &g = a.g;
    const unsigned char *stream = a.tiles + (size_t)c0 * g.KB * KBYTES + (size_t)lane * 16;
    const unsigned lds0 = (unsigned)(size_t)lds;
    int *colsum = (int *)(lds + RING * KBYTES);  // [NT * 16]: sum_t Q of every matrix row of the channel just completed
    auto issue = [&](int b) {
        const unsigned char *src = stream + (size_t)b * KBYTES;
        const unsigned dst = lds0 + (unsigned)(b & (RING - 1)) * KBYTES;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int tile = W + WAVES * k;
            if (!(a.dbg & 4)) sb_dma_tile(src + (size_t)tile * kTileBytes, __builtin_amdgcn_readfirstlane(dst + tile * kTileBytes));
        }
    };
    v4i reA[RA + 1], imA[RA + 1], reB[RB + 1], imB[RB + 1];
    unsigned qsA = 0u, qsB = 0u;
    auto clear = [&]() {
        qsA = qsB = 0u;
#pragma unroll
        for (int j = 0; j <= RA; j++) reA[j] = imA[j] = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j <= RB; j++) reB[j] = imB[j] = (v4i){0, 0, 0, 0};
    };
    clear();
    for (int b = 0; b < RING - 1 && b < nblk; b++) issue(b);
    // Stores and loads share vmcnt and may complete out of order with each other, so with stores in flight only vmcnt(0) identifies a
    // landed K block.  The channel epilogue therefore drains the (old) DMAs FIRST, then issues its stores, and the next RING - 1
    // iterations need no wait at all (their K blocks are known to be in LDS): the matrix store overlaps three K blocks of products
    // instead of stalling the workgroup until the device-wide write burst has drained.
    int landed_ahead = 0;       // K blocks after the current one known to have landed
    bool stores_pending = false;
    const unsigned lbase_u = lds0 + (unsigned)lane * 16;
    for (int b = 0; b < nblk; b++) {
        if (landed_ahead > 0) {
            landed_ahead--;
        } else {
            if (stores_pending || b + 1 >= nblk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (b + 2 >= nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            stores_pending = false;
        }
        __syncthreads();  // K block b has landed for every wave, and every wave is done with block b - 1 (whose slot is refilled now)
        const unsigned slot = lbase_u + (unsigned)(b & (RING - 1)) * KBYTES;
        v4i IA, QA, IB, QB, X[3][2];
        sb_lds_read<RB * kTileBytes>(IB, slot);
        sb_lds_read<QOFF + RB * kTileBytes>(QB, slot);
        sb_lds_read<0>(X[0][0], slot);
        sb_lds_read<QOFF>(X[0][1], slot);
        sb_lds_read<RA * kTileBytes>(IA, slot);
        sb_lds_read<QOFF + RA * kTileBytes>(QA, slot);
        sb_lds_read<kTileBytes>(X[1][0], slot);
        sb_lds_read<QOFF + kTileBytes>(X[1][1], slot);
        if (b + RING - 1 < nblk) issue(b + RING - 1);
        if (a.dbg & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); continue; }
        v4i nIA, nIB;
        sb_unroll(std::make_integer_sequence<int, RB + 1>{}, [&](auto J) {
            constexpr int j = decltype(J)::value;
            v4i &Ij = X[j % 3][0], &Qj = X[j % 3][1];
            // (LDS returns in order: with the reads for tiles j + 1 and j + 2 behind it, tile j has landed at lgkmcnt <= 4)
            if constexpr (j + 2 <= RB) {
                sb_lds_read<(j + 2) * kTileBytes>(X[(j + 2) % 3][0], slot);
                sb_lds_read<QOFF + (j + 2) * kTileBytes>(X[(j + 2) % 3][1], slot);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Ij), "+v"(Qj), "+v"(IA), "+v"(QA), "+v"(IB), "+v"(QB)::"memory");
            } else if constexpr (j + 1 <= RB) {
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Ij), "+v"(Qj), "+v"(IA), "+v"(QA), "+v"(IB), "+v"(QB)::"memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ij), "+v"(Qj), "+v"(IA), "+v"(QA), "+v"(IB), "+v"(QB)::"memory");
            }
            if constexpr (j == 0) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    qsA = __builtin_amdgcn_sad_u8((unsigned)QA[e] ^ 0x80808080u, 0u, qsA);
                    qsB = __builtin_amdgcn_sad_u8((unsigned)QB[e] ^ 0x80808080u, 0u, qsB);
                }
                nIA = (v4i){~IA[0], ~IA[1], ~IA[2], ~IA[3]};
                nIB = (v4i){~IB[0], ~IB[1], ~IB[2], ~IB[3]};
            }
            reB[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(IB, Ij, reB[j], 0, 0, 0);
            imB[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(QB, Ij, imB[j], 0, 0, 0);
            if constexpr (j <= RA) {
                reA[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(IA, Ij, reA[j], 0, 0, 0);
                imA[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(QA, Ij, imA[j], 0, 0, 0);
            }
            reB[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(QB, Qj, reB[j], 0, 0, 0);
            imB[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(nIB, Qj, imB[j], 0, 0, 0);
            if constexpr (j <= RA) {
                reA[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(QA, Qj, reA[j], 0, 0, 0);
                imA[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(nIA, Qj, imA[j], 0, 0, 0);
            }
        });
        if ((b + 1) % g.KB == 0) {  // the channel is complete
            const int f = g.f0 + c0 + b / g.KB;
            // every wave publishes sum_t Q of its two tile rows; lane (r, q) summed 16 of a K block's 64 bytes of row r, each biased by 128
            {
                int vA = (int)qsA - 128 * 16 * g.KB, vB = (int)qsB - 128 * 16 * g.KB;
                vA += __shfl_xor(vA, 16); vA += __shfl_xor(vA, 32);
                vB += __shfl_xor(vB, 16); vB += __shfl_xor(vB, 32);
                if (lane < 16) { colsum[RA * 16 + lane] = vA; colsum[RB * 16 + lane] = vB; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs for the next RING - 1 K blocks (issued long ago) have landed
            landed_ahead = RING - 1;
            __syncthreads();  // (the next reader-side hazard is a channel away: one buffer is enough)
            if (f < g.Fout && !(a.dbg & 2)) {
                c32 *chan = a.out + (size_t)((a.dbg & 8) ? (f & 7) : f) * ((size_t)g.N * (g.N + 1) / 2) * (NPOL * NPOL);
                SbRow<NPOL> row;
                sb_row_setup<NPOL>(row, a, chan, RA, lane);
#pragma unroll
                for (int j = 0; j < RA; j++) sb_store_pair<NPOL, false>(a, row, j, reA[j], imA[j], colsum, lane);
                sb_store_pair<NPOL, true>(a, row, RA, reA[RA], imA[RA], colsum, lane);
                sb_row_setup<NPOL>(row, a, chan, RB, lane);
#pragma unroll
                for (int j = 0; j < RB; j++) sb_store_pair<NPOL, false>(a, row, j, reB[j], imB[j], colsum, lane);
                sb_store_pair<NPOL, true>(a, row, RB, reB[RB], imB[RB], colsum, lane);
                stores_pending = true;
            }
            clear();
        }
    }
}

template <int NT, int NPOL>
__global__ __launch_bounds__(NT * 32, 2) void k_xe_corr_sb(SbArgs a)
{
    static_assert(NT % 2 == 0 && NT >= 6 && NT <= 16, "even row-tile counts 6 .. 16");
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * a.chan_per_wg;
    int c1 = c0 + a.chan_per_wg;
    if (c1 > a.g.Fs) c1 = a.g.Fs;
    if (c0 >= c1) return;
    const int nblk = (c1 - c0) * a.g.KB;
    switch (wave) {  // one instantiation per wave: its tile rows, hence its accumulator set, are compile-time constants
    case 0: sb_wave<NT, 0, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 1: sb_wave<NT, 1, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 2: sb_wave<NT, 2, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 3: if constexpr (NT >= 8) sb_wave<NT, 3, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 4: if constexpr (NT >= 10) sb_wave<NT, 4, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 5: if constexpr (NT >= 12) sb_wave<NT, 5, NPOL>(a, sb_lds, c0, nblk, lane); break;
    case 6: if constexpr (NT >= 14) sb_wave<NT, 6, NPOL>(a, sb_lds, c0, nblk, lane); break;
    default: if constexpr (NT >= 16) sb_wave<NT, 7, NPOL>(a, sb_lds, c0, nblk, lane); break;
    }
}

// ------------------------------------------------------------------------------------
// (3) complex-float input on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chains at
// the fp32 vector peak).  Same two-kernel structure as the int8 path with 16-time-step K blocks:
//   tile = 16 rows x 16 t floats = 1 KiB; lane l holds row l%16, time steps 4*(l/16) .. +3 as one
//   float4, i.e. the operands of the four consecutive MFMA steps of the K block.
// ------------------------------------------------------------------------------------
constexpr int kKB32 = 16;
typedef float v4f __attribute__((ext_vector_type(4)));
// row tiles are padded (with zero rows) up to a count the staged correlator is instantiated for; 0 = use the VALU kernel
static inline int xe_f32_row_tiles(int nt) { return nt <= 2 ? nt : nt <= 4 ? 4 : nt <= 6 ? 6 : nt <= 8 ? 8 : nt <= 10 ? 10 : nt <= 12 ? 12 : nt <= 16 ? 16 : 0; }

template <int NPOL>
__global__ __launch_bounds__(256) void k_xe_turn_f32(const v4i *__restrict__ in, unsigned char *__restrict__ tiles, XeGeo g)
{
    constexpr int SIN = kRowTile / NPOL;                  // stations per row tile
    constexpr int SSTRIDE = 2048 + (NPOL == 1 ? 16 : 32);  // LDS bytes between stations: conflict-free b64 reads
    __shared__ __attribute__((aligned(16))) unsigned char lds[SIN * SSTRIDE];
    const int line = blockIdx.x, rt = blockIdx.y, kb = blockIdx.z;  // one 128-byte line = 16 complex floats
    const int tid = threadIdx.x;
    const int lines_per_row = (g.F * NPOL * 8) / 128;
    const int t0 = kb * kKB32;
#pragma unroll
    for (int it = 0; it < (SIN * 16 * 8) / 256; it++) {
        const int idx = tid + 256 * it, seg = idx & 7, row = idx >> 3, t = row & 15, sl = row >> 4;
        const int s = rt * SIN + sl;
        v4i v = (v4i){0, 0, 0, 0};
        if (s < g.N && t0 + t < g.T) v = __builtin_nontemporal_load(in + (((size_t)(t0 + t) * g.N + s) * lines_per_row + line) * 8 + seg);
        *(v4i *)(lds + sl * SSTRIDE + t * 128 + seg * 16) = v;
    }
    __syncthreads();
    const size_t tile_stride = kTileBytes;  // [chan][kb][plane][rt]
#pragma unroll
    for (int it = 0; it < (SIN * 16 + 255) / 256; it++) {
        const int item = tid + 256 * it;
        if (item >= SIN * 16) break;
        const int sl = item % SIN, c = item / SIN;  // c = complex value within the line
        float re[16], im[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const float2 z = *(const float2 *)(lds + sl * SSTRIDE + t * 128 + c * 8);
            re[t] = z.x;
            im[t] = z.y;
        }
        const int f = (NPOL == 1) ? line * 16 + c : line * 8 + (c >> 1);
        const int rr = (NPOL == 1) ? sl : 2 * sl + (c & 1);
        unsigned char *tre = tiles + ((((size_t)f * g.KB + kb) * 2 + 0) * g.NT + rt) * tile_stride;
        unsigned char *tim = tre + (size_t)g.NT * tile_stride;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            *(float4 *)(tre + (ks * 16 + rr) * 16) = make_float4(re[4 * ks], re[4 * ks + 1], re[4 * ks + 2], re[4 * ks + 3]);
            *(float4 *)(tim + (ks * 16 + rr) * 16) = make_float4(im[4 * ks], im[4 * ks + 1], im[4 * ks + 2], im[4 * ks + 3]);
        }
    }
}

// One workgroup per channel; the NTT(NTT+1)/2 tile pairs are split EXACTLY over the waves (PPW each), so
// there is no per-pair predicate in the K loop and the accumulators stay in AGPRs.
template <int NTT, int WAVES, int PPW>
__global__ __launch_bounds__(WAVES * 64) void k_xe_corr_f32(const unsigned char *__restrict__ tiles, c32 *__restrict__ out, XeGeo g,
                                                            int npairs, int accumulate)
{
    constexpr int NTHR = WAVES * 64, KBYTES = 2 * NTT * kTileBytes, PER_THREAD = KBYTES / (NTHR * 16);
    static_assert(KBYTES % (NTHR * 16) == 0 && PER_THREAD >= 1, "tile bytes per K block must split over the workgroup");
    // (more than 8 row tiles -- 129 ... 256 rows: the triangle's pairs are split over gridDim.y workgroups of WAVES x PPW pairs; every one
    // of them stages all NTT row tiles of a K block)
    static_assert(NTT * (NTT + 1) / 2 <= WAVES * PPW || NTT > 8, "tile pairs must fit the waves");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][KBYTES];
    const int f = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int bi[PPW], bj[PPW];
#pragma unroll
    for (int q = 0; q < PPW; q++) {
        const int p = (int)blockIdx.y * WAVES * PPW + q * WAVES + wave;
        pair_to_tiles(p < npairs ? p : 0, bi[q], bj[q]);  // (a pair past the triangle repeats pair 0 and stores nothing)
    }
    v4f re[PPW], uu[PPW], ww[PPW];
#pragma unroll
    for (int q = 0; q < PPW; q++) re[q] = uu[q] = ww[q] = (v4f){0.f, 0.f, 0.f, 0.f};
    const unsigned char *src = tiles + (size_t)f * g.KB * KBYTES + (size_t)tid * 16;
    // two register stages: the load for K block kb+2 is issued while block kb is multiplied (the HBM round
    // trip under load is longer than one K block of MFMAs)
    constexpr int DEPTH = (NTT <= 4) ? 2 : 1;  // larger triangles need the registers for their accumulators
    v4i stage[DEPTH][PER_THREAD];
#pragma unroll
    for (int u = 0; u < DEPTH; u++)
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++)
            stage[u][i] = (u < g.KB) ? __builtin_nontemporal_load((const v4i *)(src + (size_t)u * KBYTES + (size_t)i * NTHR * 16)) : (v4i){0, 0, 0, 0};
    auto step = [&](int kb, v4i(&st)[PER_THREAD]) {
        unsigned char *buf = lds[kb & 1];
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++) *(v4i *)(buf + tid * 16 + i * NTHR * 16) = st[i];
        if (kb + DEPTH < g.KB) {
#pragma unroll
            for (int i = 0; i < PER_THREAD; i++)
                st[i] = __builtin_nontemporal_load((const v4i *)(src + (size_t)(kb + DEPTH) * KBYTES + (size_t)i * NTHR * 16));
        }
        __syncthreads();
        const unsigned char *pI = buf + lane * 16, *pQ = pI + NTT * kTileBytes;
        // operands of pair q+1 are read from LDS before the MFMAs of pair q are issued
        // three real products per element instead of four (see k_xe_f32_fused): k1 = (I1+Q1) I2, k2 = I1 (I2+Q2), k3 = Q1 (I2-Q2)
        v4f op[2][4];  // I1, Q1, I2, Q2 of the pair
        auto fetch = [&](int q, v4f(&o)[4]) {
            o[0] = *(const v4f *)(pI + bi[q] * kTileBytes);
            o[1] = *(const v4f *)(pQ + bi[q] * kTileBytes);
            o[2] = *(const v4f *)(pI + bj[q] * kTileBytes);
            o[3] = *(const v4f *)(pQ + bj[q] * kTileBytes);
        };
        fetch(0, op[0]);
#pragma unroll
        for (int q = 0; q < PPW; q++) {
            if (q + 1 < PPW) fetch(q + 1, op[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (the scheduler sinks it otherwise)
            const v4f(&o)[4] = op[q & 1];
            const v4f s1 = o[0] + o[1], s2 = o[2] + o[3], d2 = o[2] - o[3];
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                re[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(s1[kc], o[2][kc], re[q], 0, 0, 0);   // k1
                uu[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[0][kc], s2[kc], uu[q], 0, 0, 0);   // k2
                ww[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[1][kc], d2[kc], ww[q], 0, 0, 0);   // k3
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int kb = 0; kb < g.KB; kb += DEPTH) {
        step(kb, stage[0]);
        if (DEPTH == 2 && kb + 1 < g.KB) step(kb + 1, stage[DEPTH - 1]);
    }
    const int nb = g.N * (g.N + 1) / 2, np2 = g.npol * g.npol;
#pragma unroll
    for (int q = 0; q < PPW; q++) {
        if ((int)blockIdx.y * WAVES * PPW + q * WAVES + wave >= npairs) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int r1 = bi[q] * kRowTile + (lane >> 4) * 4 + reg, r2 = bj[q] * kRowTile + (lane & 15);
            if (r1 >= g.A || r2 >= g.A) continue;
            const int s1 = r1 / g.npol, p1 = r1 - s1 * g.npol, s2 = r2 / g.npol, p2 = r2 - s2 * g.npol;
            if (s1 < s2) continue;
            const size_t o = ((size_t)f * nb + (s1 * (s1 + 1) / 2 + s2)) * np2 + p1 * g.npol + p2;
            c32 v;
            v.x = re[q][reg] - ww[q][reg];  // k1 - k3
            v.y = re[q][reg] - uu[q][reg];  // k1 - k2
            if (accumulate) { v.x += out[o].x; v.y += out[o].y; }
            out[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// (3b) complex-float input, FUSED (rows = stations x pols <= 64): no tile round trip through HBM.
// A workgroup owns 8 channels (64 bytes of every (t, station) row for one polarisation, 128 for two: whole or half
// 128-byte lines; the workgroup of the other half is the neighbouring blockIdx, i.e. the same XCD's L2 in practice) and
// one of the time ranges.  Per 16-time-step K block the 8 waves stage the 64 KiB they need through registers into LDS in
// the MFMA operand order of section (3) (per channel: [plane][row tile][lane][4 floats]), double buffered, the global
// loads of block k+1 in flight while block k is multiplied.  Wave w correlates channel w: all row-tile pairs, 3 x 4
// accumulator registers per pair (120 for 64 rows) stay in AGPRs for the whole time range -- that register capacity
// (30 KiB per channel) is what limits a workgroup to 8 channels and rules the same design out for the 2-byte samples of
// the int8 path (8 channels = 16 bytes per row).  The partial matrices of the time ranges are summed by k_xe_reduce.
// HBM traffic = input once + 2-3 x the (small) output.
// ------------------------------------------------------------------------------------
template <int NTT, int NPOL, int CH>
__global__ __launch_bounds__(CH * 64) void k_xe_f32_fused(const v4i *__restrict__ in, c32 *__restrict__ part, XeGeo g, int tsplit,
                                                           int row_pieces /* 16-byte pieces per (t, station) row that exist = the row stride */)
{
    constexpr int NTHR = CH * 64, NP = NTT * (NTT + 1) / 2, CBYTES = 2 * NTT * kTileBytes;  // per channel and K block
    constexpr int SEGQ = CH * NPOL / 2;                 // 16-byte pieces per (t, station) segment
    constexpr int NS = kRowTile * NTT / NPOL;           // stations covered by the row tiles
    constexpr int ITEMS = 16 * NS * SEGQ, PER = ITEMS / NTHR;
    static_assert(ITEMS % NTHR == 0, "staging items must split over the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][CH * CBYTES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // For one polarisation a workgroup reads 64-byte halves of 128-byte lines; the workgroup reading the other half must sit on
    // the same XCD (same L2) or every line crosses the fabric twice.  Workgroups are dispatched round-robin over the 8 XCDs, so
    // the two members of a pair take linear ids with the same id % 8 (tools/ubench/sector_read.hip: 39 -> 22 us per 134 MB).
    const int ngrp = g.F / CH;
    int cgrp, ts;
    {
        const int b = blockIdx.x, total = ngrp * tsplit;
        constexpr int SH = 128 / (CH * NPOL * 8) > 0 ? 128 / (CH * NPOL * 8) : 1;  // workgroups sharing one 128-byte line
        if (SH > 1 && ngrp % SH == 0 && total % (8 * SH) == 0) {
            const int xcd = b & 7, within = b >> 3, member = within % SH, combo = xcd + 8 * (within / SH);
            cgrp = SH * (combo % (ngrp / SH)) + member;
            ts = combo / (ngrp / SH);
        } else {
            cgrp = b % ngrp;
            ts = b / ngrp;
        }
    }
    const int kb_total = (g.T + kKB32 - 1) / kKB32, kb_per = (kb_total + tsplit - 1) / tsplit;
    const int kb0 = ts * kb_per, kb1 = (kb0 + kb_per < kb_total) ? kb0 + kb_per : kb_total;
    // (g.F counts whole 128-byte lines per row; a row that ends inside its last line has fewer pieces: the missing ones read as zeros and
    // their channels have no output -- no padded copy of the input is needed)
    const size_t row_v4 = (size_t)row_pieces;
    const size_t seg0 = (size_t)cgrp * SEGQ;            // first piece of this workgroup's channels inside a row

    v4f re[NP], uu[NP], ww[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) re[q] = uu[q] = ww[q] = (v4f){0.f, 0.f, 0.f, 0.f};

    // One register stage: the loads of K block kb+1 are in flight while block kb is multiplied.
    v4i stA[PER];
    const int t_end = (kb1 * kKB32 < g.T) ? kb1 * kKB32 : g.T;  // a block past this workgroup's time range loads nothing (zeros)
    auto load_block = [&](int kb, v4i(&stage)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int idx = tid + NTHR * k, q16 = idx % SEGQ, sidx = (idx / SEGQ) % NS, t = idx / (SEGQ * NS);
            const int tt = kb * kKB32 + t;
            const bool ok = sidx < g.N && tt < t_end && seg0 + q16 < row_v4;
            v4i piece = (v4i){0, 0, 0, 0};
            if (ok) {
                const v4i *src = in + ((size_t)tt * g.N + sidx) * row_v4 + seg0 + q16;
                // one polarisation: the other half of the line belongs to the sibling workgroup on this XCD -> keep it in L2
                piece = (CH * NPOL * 8 < 128) ? *src : __builtin_nontemporal_load(src);
            }
            stage[k] = piece;
        }
    };
    auto store_block = [&](unsigned char *buf, const v4i(&stage)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int idx = tid + NTHR * k, q16 = idx % SEGQ, sidx = (idx / SEGQ) % NS, t = idx / (SEGQ * NS);
            // (whole-vector bit cast: __builtin_bit_cast of a single vector ELEMENT reads element 0 with this compiler)
            const v4f fv = __builtin_bit_cast(v4f, stage[k]);
            const float v[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
            for (int e = 0; e < 2; e++) {  // the two complex values of the piece
                const int ce = 2 * q16 + e, c = ce / NPOL, pol = ce % NPOL, row = sidx * NPOL + pol;
                // time step t of the K block sits in float (t%4 + rot)%4 of lane group t/4; rot depends on the channel only, so
                // both operands of a product see the same order, and it spreads the writes of a wave over all banks
                const int rot = (NPOL == 1) ? (c >> 1) & 3 : c & 3;
                const int off = c * CBYTES + (row / kRowTile) * kTileBytes + (((t >> 2) * 16 + (row % kRowTile)) * 16) + (((t + rot) & 3) * 4);
                *(float *)(buf + off) = v[2 * e];                           // I plane
                *(float *)(buf + off + NTT * kTileBytes) = v[2 * e + 1];    // Q plane
            }
        }
    };
    // x1 conj(x2) = (I1 I2 + Q1 Q2) + i (Q1 I2 - I1 Q2) with THREE real products per element instead of four:
    //   k1 = (I1 + Q1) I2,  k2 = I1 (I2 + Q2),  k3 = Q1 (I2 - Q2)   =>   re = k1 - k3,  im = k1 - k2   (epilogue)
    // S = I + Q and D = I - Q are formed in registers from the operand vectors (8 packed adds per K block).
    v4f I[NTT], Q[NTT], S[NTT], D[NTT];
    auto operands = [&](const unsigned char *buf) {
        const unsigned char *base = buf + wave * CBYTES + lane * 16;
#pragma unroll
        for (int rt = 0; rt < NTT; rt++) {
            I[rt] = *(const v4f *)(base + rt * kTileBytes);
            Q[rt] = *(const v4f *)(base + (NTT + rt) * kTileBytes);
        }
#pragma unroll
        for (int rt = 0; rt < NTT; rt++) {
            S[rt] = I[rt] + Q[rt];
            D[rt] = I[rt] - Q[rt];
        }
    };
    auto products = [&](auto lo_tag, auto hi_tag) {  // tile pairs [lo, hi)
        constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
        for (int bi = 0; bi < NTT; bi++) {
#pragma unroll
            for (int bj = 0; bj <= bi; bj++) {
                const int q = bi * (bi + 1) / 2 + bj;
                if (q < LO || q >= HI) continue;
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    re[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(S[bi][kc], I[bj][kc], re[q], 0, 0, 0);  // k1 = (I1+Q1) I2
                    uu[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(I[bi][kc], S[bj][kc], uu[q], 0, 0, 0);  // k2 = I1 (I2+Q2)
                    ww[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(Q[bi][kc], D[bj][kc], ww[q], 0, 0, 0);  // k3 = Q1 (I2-Q2)
                }
            }
        }
    };
    // The order of a K block is pinned (scheduling barriers): the global loads of the next block, the products of the first eight
    // tile pairs, THEN the staging stores of the next block, the workgroup barrier, and the last two pairs' products, which run
    // while the other waves arrive.  Left to itself the scheduler hoists the stores -- and with them the wait for global loads
    // issued a few hundred cycles before -- to a third of the way into the products: the wave then sits on that wait with the
    // matrix pipe idle.  That, not "loads making no progress under saturated MFMA", was the gap between the multiply loop alone
    // (178 us) and the whole kernel in round 1.  Measured at 64 antennas (interleaved A/B): 221 us unpinned, 199 with the stores
    // behind all products, 198 with 8 + 2 (6 + 4: 203; loads issued after the stores, two blocks ahead: 205-215).
    constexpr int P1 = NP >= 10 ? 8 : NP;  // tile pairs before the staging stores
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, P1>;
    using T2 = std::integral_constant<int, NP>;
    load_block(kb0, stA);
    store_block(lds[0], stA);
    __syncthreads();
    for (int kb = kb0; kb < kb1; kb++) {
        load_block(kb + 1, stA);  // unconditional (a block past the range reads as zeros): no control flow in the loop body
        operands(lds[(kb - kb0) & 1]);
        products(T0{}, T1{});
        __builtin_amdgcn_sched_barrier(0);
        store_block(lds[(kb + 1 - kb0) & 1], stA);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        products(T1{}, T2{});
        __builtin_amdgcn_sched_barrier(0);
    }
    // partial matrix of this time range, channel = cgrp*8 + wave
    const int f = cgrp * CH + wave;
    const int nb = g.N * (g.N + 1) / 2, np2 = NPOL * NPOL;
    c32 *__restrict__ dst = part + ((size_t)ts * g.F + f) * nb * np2;
#pragma unroll
    for (int bi = 0; bi < NTT; bi++) {
#pragma unroll
        for (int bj = 0; bj <= bi; bj++) {
            const int q = bi * (bi + 1) / 2 + bj;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r1 = bi * kRowTile + (lane >> 4) * 4 + reg, r2 = bj * kRowTile + (lane & 15);
                if (r1 >= g.A || r2 >= g.A) continue;
                const int s1 = r1 / NPOL, p1 = r1 % NPOL, s2 = r2 / NPOL, p2 = r2 % NPOL;
                if (s1 < s2) continue;
                c32 v;
                // k1 - k3 = I1 I2 + Q1 Q2 ;  k1 - k2 = Q1 I2 - I1 Q2
                v.x = re[q][reg] - ww[q][reg];
                v.y = re[q][reg] - uu[q][reg];
                dst[(size_t)(s1 * (s1 + 1) / 2 + s2) * np2 + p1 * NPOL + p2] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_xe_reduce(const c32 *__restrict__ part, c32 *__restrict__ out, size_t n, size_t stride, int tsplit, int accumulate)
{
    // n outputs (the caller's channels), partial matrices `stride` items apart (all channels incl. the padding ones, which come last)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        c32 a = part[i];
        for (int t = 1; t < tsplit; t++) { a.x += part[(size_t)t * stride + i].x; a.y += part[(size_t)t * stride + i].y; }
        if (accumulate) { a.x += out[i].x; a.y += out[i].y; }
        out[i] = a;
    }
}

// ------------------------------------------------------------------------------------
// complex-float input: fp32 arithmetic, 8x8 station blocks per thread, lanes along channels
// ------------------------------------------------------------------------------------
constexpr int kCfBlk = 4;  // rows per side of a thread's block

__global__ __launch_bounds__(256) void k_xe_cf32(const c32 *__restrict__ in, c32 *__restrict__ out, XeGeo g, int nblk,
                                                 int accumulate)
{
    // grid.x over channels (256 per block), grid.y over block pairs (bi >= bj) of kCfBlk rows
    const int f = blockIdx.x * 256 + threadIdx.x;
    int bi, bj;
    pair_to_tiles(blockIdx.y, bi, bj);
    if (f >= g.Fout) return;  // g.F (>= g.Fout: rows padded to whole lines) is only the input's row stride; `out` holds Fout channels
    (void)nblk;
    c32 acc[kCfBlk][kCfBlk];
#pragma unroll
    for (int i = 0; i < kCfBlk; i++)
#pragma unroll
        for (int j = 0; j < kCfBlk; j++) acc[i][j].x = acc[i][j].y = 0.f;
    const size_t frame = (size_t)g.F * g.A;  // elements per time step; element (t, s, f, p) at ((t*N+s)*F+f)*npol+p
    for (int t = 0; t < g.T; t++) {
        c32 a[kCfBlk], b[kCfBlk];
#pragma unroll
        for (int i = 0; i < kCfBlk; i++) {
            const int r1 = bi * kCfBlk + i, r2 = bj * kCfBlk + i;
            const int s1 = r1 / g.npol, p1 = r1 - s1 * g.npol, s2 = r2 / g.npol, p2 = r2 - s2 * g.npol;
            a[i].x = a[i].y = b[i].x = b[i].y = 0.f;
            if (r1 < g.A) a[i] = in[(size_t)t * frame + ((size_t)s1 * g.F + f) * g.npol + p1];
            if (r2 < g.A) b[i] = in[(size_t)t * frame + ((size_t)s2 * g.F + f) * g.npol + p2];
        }
#pragma unroll
        for (int i = 0; i < kCfBlk; i++)
#pragma unroll
            for (int j = 0; j < kCfBlk; j++) {
                // cxmac, lib/clXEngine_impl.cc:728-737: acc += z0 * conj(z1)
                acc[i][j].x += a[i].x * b[j].x + a[i].y * b[j].y;
                acc[i][j].y += a[i].y * b[j].x - a[i].x * b[j].y;
            }
    }
    const int nb = g.N * (g.N + 1) / 2, np2 = g.npol * g.npol;
#pragma unroll
    for (int i = 0; i < kCfBlk; i++)
#pragma unroll
        for (int j = 0; j < kCfBlk; j++) {
            const int r1 = bi * kCfBlk + i, r2 = bj * kCfBlk + j;
            if (r1 >= g.A || r2 >= g.A) continue;
            const int s1 = r1 / g.npol, p1 = r1 - s1 * g.npol, s2 = r2 / g.npol, p2 = r2 - s2 * g.npol;
            if (s1 < s2) continue;
            const size_t o = ((size_t)f * nb + (s1 * (s1 + 1) / 2 + s2)) * np2 + p1 * g.npol + p2;
            c32 v = acc[i][j];
            if (accumulate) { v.x += out[o].x; v.y += out[o].y; }
            out[o] = v;
        }
}

}  // namespace

struct mi355_xengine {
    mi355_ctx *ctx;
    int data_type;
    XeGeo g;
    size_t in_bytes, out_items, tile_bytes;
    int nslab = 1, slab_channels = 0;  // channel slabs per integration; channels the tile workspace holds
    unsigned char *d_tiles = nullptr;        // tile workspace of the device-pointer path and of slot 0
    // Host path: two slots (pinned host + device buffers + tile workspace), slot s runs on ctx->stream[s].
    // They replace the reference's pinned double buffers and worker thread (lib/clXEngine_impl.cc:304-382,
    // 1234-1299): submit() returns once the integration is enqueued, wait() hands back the oldest result.
    struct Slot {
        void *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
        unsigned char *d_tiles = nullptr, *d_pad = nullptr;
        hipEvent_t done = nullptr;
        bool busy = false;
    } slot[2];
    int next_submit = 0, next_wait = 0, pending = 0;
    bool acquired = false;  // the next slot's pinned frame buffer is handed out (zero-copy gather)
    unsigned fused_epoch[2] = {0, 0};  // launches of the fused IChar path per tile workspace (slot 0 / the handle's, slot 1)
    bool flags_stale[2] = {false, false};  // a non-fused launch wrote corner-turn tiles over the workspace: the in-launch reduction's counters are gone
    // batched form (mi355_xengine_xcorrelate_n_dev): partial sums of nint windows, grown on demand
    unsigned char *d_batch = nullptr;
    size_t batch_bytes = 0;
    unsigned batch_epoch = 0;  // launches of the in-launch reduction on d_batch with the current window count
    int batch_nint = 0;        // ... whose arrival words sit behind that count's partial sums (another count: another place, zeroed first)
    std::mutex dev_lock;    // device-pointer entry points: workspace growth, the reduction's counters and the launch itself, one caller at a time
    // Launches that share a workspace (0: d_tiles = the device-pointer path and slot 0, 1: slot 1's tiles, 2: d_batch) must not overlap: the
    // partial sums, the inboxes and the arrival words of the in-launch reduction belong to one launch at a time.  On one stream they are ordered
    // anyway; when a call arrives on ANOTHER stream than the workspace's last one, an event recorded on the old stream is waited for on the new one.
    hipStream_t ws_stream[3] = {nullptr, nullptr, nullptr};
    bool ws_used[3] = {false, false, false};
    hipEvent_t ws_done[3] = {nullptr, nullptr, nullptr};
    mi355_xe_route route = {};      // kernels of the last device-side call (mi355_xengine_last_route)
    char routes_seen[8][64] = {};   // routes already logged once
    int pad = 0;            // one zero channel appended on the device (odd channel count of 2-byte samples)
    size_t pad_bytes = 0;
    unsigned char *d_pad = nullptr;
};

thread_local mi355_xe_route mi355_xe_route_tls = {};
void mi355_xe_route_set(const char *kernel, int windows, int workgroups, int units_per_workgroup, int tsplit, int in_launch_reduce, int touches, int pace)
{
    mi355_xe_route &r = mi355_xe_route_tls;
    snprintf(r.kernel, sizeof(r.kernel), "%s", kernel);
    r.launches += 1;
    r.windows = windows; r.workgroups = workgroups; r.units_per_workgroup = units_per_workgroup; r.tsplit = tsplit;
    r.in_launch_reduce = in_launch_reduce; r.touches = touches; r.pace = pace;
}

namespace {

// the calling thread's route record starts a call at zero launches; at the end of the call it becomes the handle's (under dev_lock); a route the
// handle has not taken before is logged once
void xe_route_begin() { mi355_xe_route_tls = mi355_xe_route{}; }
void xe_route_commit(mi355_xengine *h);
struct XeRouteScope {  // (declared after the lock guard of an entry point: committed before the lock is released)
    mi355_xengine *h;
    explicit XeRouteScope(mi355_xengine *hh) : h(hh) { xe_route_begin(); }
    ~XeRouteScope() { xe_route_commit(h); }
};

// rows of (t, station) with an odd number of 2-byte channels: copy into rows padded by one zero channel (4-byte units)
__global__ __launch_bounds__(256) void k_xe_pad_rows(const unsigned short *__restrict__ in, unsigned short *__restrict__ out, size_t rows,
                                                     int src_units, int dst_units)
{
    const size_t total = rows * (size_t)dst_units;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const size_t row = u / dst_units;
        const int c = (int)(u - row * dst_units);
        out[u] = c < src_units ? in[row * src_units + c] : (unsigned short)0;
    }
}

// the same for rows of complex floats (8-byte units)
__global__ __launch_bounds__(256) void k_xe_pad_rows8(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, size_t rows,
                                                      int src_units, int dst_units)
{
    const size_t total = rows * (size_t)dst_units;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const size_t row = u / dst_units;
        const int c = (int)(u - row * dst_units);
        out[u] = c < src_units ? __builtin_nontemporal_load(in + row * src_units + c) : 0ull;
    }
}

// Stream-order this call on workspace ws behind the workspace's previous launch (see mi355_xengine::ws_stream); called under dev_lock.
// The previous stream is only ever touched when it is one of the context's own (they live as long as the context): an event recorded on it now is
// waited for on the new stream.  A CALLER's stream may have been destroyed since its launch -- handing the runtime a dead handle crashes it
// (tests/test_xengine_gpu.py::test_handle_survives_a_destroyed_stream) -- so behind a caller's stream the call waits for the device instead.  (An event
// behind EVERY launch on the launch's own stream would avoid that wait, and was measured: 2-4 us per launch at BASELINE config 5, 60.9 against 57.4 us
// for one window per call -- every caller would pay for a stream change that almost none makes.)
int xe_order_workspace(mi355_xengine *h, int ws, hipStream_t st)
{
    if (h->ws_used[ws] && h->ws_stream[ws] != st) {
        const hipStream_t old = h->ws_stream[ws];
        if (old == h->ctx->stream[0] || old == h->ctx->stream[1]) {
            if (!h->ws_done[ws]) MI355_HIP(hipEventCreateWithFlags(&h->ws_done[ws], hipEventDisableTiming));
            MI355_HIP(hipEventRecord(h->ws_done[ws], old));
            MI355_HIP(hipStreamWaitEvent(st, h->ws_done[ws], 0));
        } else {
            MI355_HIP(hipDeviceSynchronize());
        }
    }
    h->ws_stream[ws] = st;
    h->ws_used[ws] = true;
    return MI355_OK;
}

int launch_xe_body(mi355_xengine *h, const void *in, void *out, int accumulate, hipStream_t st, unsigned char *tiles, unsigned char *padbuf,
                   int stations_per_group);
int launch_xe(mi355_xengine *h, const void *in, void *out, int accumulate, hipStream_t st, unsigned char *tiles, unsigned char *padbuf,
              int stations_per_group = 0)
{
    if (tiles) {
        const int rc = xe_order_workspace(h, tiles == h->d_tiles ? 0 : 1, st);
        if (rc != MI355_OK) return rc;
    }
    return launch_xe_body(h, in, out, accumulate, st, tiles, padbuf, stations_per_group);
}
int launch_xe_body(mi355_xengine *h, const void *in, void *out, int accumulate, hipStream_t st, unsigned char *tiles, unsigned char *padbuf,
                   int stations_per_group)
{
    const XeGeo &g = h->g;
    // complex float: is this launch the fused kernel's?  (decided first: that kernel reads rows which end inside a 128-byte line as they
    // are, every other complex-float kernel needs them padded to whole lines)
    int tsplit = 1;
    size_t out_items = 0;
    bool cf32_fused = false;
    if (h->data_type == MI355_DTYPE_COMPLEX) {
        const int ts_env = getenv("MI355_XE_CF32_TSPLIT") ? atoi(getenv("MI355_XE_CF32_TSPLIT")) : 0;
        out_items = (size_t)g.F * (g.N * (g.N + 1) / 2) * g.npol * g.npol;
        tsplit = (ts_env > 0 && g.T % (16 * ts_env) == 0) ? ts_env : (g.T >= 64 ? 2 : 1);
        if (ts_env <= 0 && g.T >= 64) {
            // few channels: more time ranges, so that (channel groups) x (ranges) still covers the CUs -- 64 antennas x 16 channels x 16384
            // frames ran as 4 workgroups (2.2 ms); the ranges keep at least two K blocks each and their partial matrices fit the workspace
            const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256, groups = (g.F + 7) / 8, kb_total = (g.T + kKB32 - 1) / kKB32;
            int want = cus / groups;  // the most ranges that still run as one round of workgroups (125 groups x 3 ranges: a second round at 46 % -- slower than 2)
            if (want > kb_total / 2) want = kb_total / 2;
            while (want > 2 && (size_t)want * out_items * 8 > h->tile_bytes) want--;
            if (want > tsplit) tsplit = want;
        }
        cf32_fused = ((size_t)g.F * g.npol * 8) % 128 == 0 && tiles && xe_f32_row_tiles(g.NT) != 0 && !getenv("MI355_XE_CF32_VALU") && g.NT <= 4 &&
                     g.F % 8 == 0 && h->tile_bytes >= (size_t)tsplit * out_items * 8 && !getenv("MI355_XE_CF32_TWO_KERNELS");
    }
    const bool cf32_rows_as_given = h->data_type == MI355_DTYPE_COMPLEX && h->pad && cf32_fused && (g.Fout * g.npol) % 2 == 0 &&
                                    (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && !getenv("MI355_XE_CF32_PAD_COPY");
    if (h->pad && !cf32_rows_as_given) {
        const size_t rows = (size_t)g.T * g.N;
        const int dst_units = g.F, src_units = g.Fout;  // 2-byte units per row (IChar one polarisation / packed: 2 bytes per channel)
        size_t blocks = (rows * dst_units + 255) / 256;
        const size_t cap = (size_t)(h->ctx->num_cus > 0 ? h->ctx->num_cus : 256) * 16;
        if (blocks > cap) blocks = cap;
        if (h->data_type == MI355_DTYPE_COMPLEX) {
            const int du = g.F * g.npol, su = g.Fout * g.npol;  // complex values per row
            size_t b8 = (rows * du + 255) / 256;
            if (b8 > cap) b8 = cap;
            hipLaunchKernelGGL(k_xe_pad_rows8, dim3((unsigned)b8), dim3(256), 0, st, (const unsigned long long *)in, (unsigned long long *)padbuf, rows, su, du);
        } else
        hipLaunchKernelGGL(k_xe_pad_rows, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short *)in, (unsigned short *)padbuf, rows,
                           src_units, dst_units);
        MI355_HIP(hipGetLastError());
        in = padbuf;
    }
    if (h->data_type == MI355_DTYPE_COMPLEX) {
        const bool mfma = ((size_t)g.F * g.npol * 8) % 128 == 0 && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && tiles &&
                          xe_f32_row_tiles(g.NT) != 0 && !getenv("MI355_XE_CF32_VALU");
        // fused kernel: rows <= 64, whole groups of 8 channels; the partial matrices live in the tile workspace
        if (mfma && cf32_fused) {
            const int ntt = g.NT == 3 ? 4 : g.NT;
            // 8 channels per workgroup (one workgroup per CU, 64-byte pieces of a row) or 4 (two workgroups per CU, 32-byte pieces).
            // Interleaved A/B at 1024 channels x 1024 frames with the pinned K-block schedule: 64 antennas 226 us with 4 / 198 with 8,
            // 48 antennas 179 / 172, 32 antennas 95 / 72, 16 antennas x 2048 channels 64 / 50; two polarisations: 32 antennas
            // 176 / 178, 16 antennas 64 / 60.
            const int chw = getenv("MI355_XE_CF32_CH") ? atoi(getenv("MI355_XE_CF32_CH")) : 8;
            const int ch = (chw == 8) ? 8 : 4;
            dim3 grid((g.F / ch) * tsplit);
            const int row_pieces = ((cf32_rows_as_given || !h->pad) ? g.Fout : g.F) * g.npol / 2;
#define FUSED(NTT, NPOL, CHN) hipLaunchKernelGGL((k_xe_f32_fused<NTT, NPOL, CHN>), grid, dim3(CHN * 64), 0, st, (const v4i *)in, (c32 *)tiles, g, tsplit, row_pieces)
#define FUSED_CH(NTT, NPOL) do { if (ch == 8) FUSED(NTT, NPOL, 8); else FUSED(NTT, NPOL, 4); } while (0)
            if (g.npol == 1) { if (ntt == 1) FUSED_CH(1, 1); else if (ntt == 2) FUSED_CH(2, 1); else FUSED_CH(4, 1); }
            else             { if (ntt == 1) FUSED_CH(1, 2); else if (ntt == 2) FUSED_CH(2, 2); else FUSED_CH(4, 2); }
#undef FUSED_CH
#undef FUSED
            MI355_HIP(hipGetLastError());
            const size_t real_items = (size_t)g.Fout * (g.N * (g.N + 1) / 2) * g.npol * g.npol;
            size_t blocks = (real_items + 255) / 256;
            const size_t cap = (size_t)(h->ctx->num_cus > 0 ? h->ctx->num_cus : 256) * 16;
            if (blocks > cap) blocks = cap;
            hipLaunchKernelGGL(k_xe_reduce, dim3((unsigned)blocks), dim3(256), 0, st, (const c32 *)tiles, (c32 *)out, real_items, out_items, tsplit, accumulate);
            MI355_HIP(hipGetLastError());
            mi355_xe_route_set("k_xe_f32_fused+k_xe_reduce", 1, (int)grid.x, 1, tsplit, 0, 0, 0);
            return MI355_OK;
        }
        if (mfma) {
            XeGeo gf = g;
            gf.KB = (g.T + kKB32 - 1) / kKB32;
            gf.NT = xe_f32_row_tiles(g.NT);
            dim3 tgrid((unsigned)(((size_t)g.F * g.npol * 8) / 128), gf.NT, gf.KB);
            if (g.npol == 1) hipLaunchKernelGGL((k_xe_turn_f32<1>), tgrid, dim3(256), 0, st, (const v4i *)in, tiles, gf);
            else hipLaunchKernelGGL((k_xe_turn_f32<2>), tgrid, dim3(256), 0, st, (const v4i *)in, tiles, gf);
            MI355_HIP(hipGetLastError());
            const int npairs = gf.NT * (gf.NT + 1) / 2;
            const int chunks = gf.NT > 8 ? (npairs + 35) / 36 : 1;  // 129 ... 256 rows: 36 tile pairs per workgroup
#define CORR_F32(NTT, WV, PPW)                                                                                             \
    hipLaunchKernelGGL((k_xe_corr_f32<NTT, WV, PPW>), dim3(g.Fout, chunks), dim3(WV * 64), 0, st, \
                       (const unsigned char *)tiles, (c32 *)out, gf, npairs, accumulate)
            if (gf.NT == 1) CORR_F32(1, 1, 1);
            else if (gf.NT == 2) CORR_F32(2, 1, 3);
            else if (gf.NT == 4) CORR_F32(4, 2, 5);
            else if (gf.NT == 6) CORR_F32(6, 3, 7);
            else if (gf.NT == 8) CORR_F32(8, 4, 9);
            else if (gf.NT == 10) CORR_F32(10, 4, 9);
            else if (gf.NT == 12) CORR_F32(12, 4, 9);
            else CORR_F32(16, 4, 9);
#undef CORR_F32
            MI355_HIP(hipGetLastError());
            mi355_xe_route_set("k_xe_turn_f32+k_xe_corr_f32", 1, 0, 1, 1, 0, 0, 0);
            return MI355_OK;
        }
        const int nblk = (g.A + kCfBlk - 1) / kCfBlk;
        dim3 grid((g.Fout + 255) / 256, nblk * (nblk + 1) / 2);
        hipLaunchKernelGGL(k_xe_cf32, grid, dim3(256), 0, st, (const c32 *)in, (c32 *)out, g, nblk, accumulate);
        MI355_HIP(hipGetLastError());
        mi355_xe_route_set("k_xe_cf32", 1, 0, 1, 1, 0, 0, 0);
        return MI355_OK;
    }
    // The integration can be processed in channel slabs that reuse one tile workspace.  Measured on
    // MI355X: slabs small enough for the Infinity Cache (4-8 per integration at config 5) are SLOWER
    // than one pass (108-190 us vs 81 us: twice the launches at a fraction of the parallelism and no
    // visible cache benefit), so slabs are only used to bound the workspace (4 GiB) for huge problems.
    const size_t row_bytes = (g.mode == 0) ? (size_t)g.F * g.npol * 2 : (size_t)g.F * 2;
    // IChar with at most 64 rows and whole 128-byte lines per row: corner turn and correlation fused in one pass
    // (xengine_fused.hip); the tile workspace then only holds the int32 partial sums of the time ranges.
    // 64 stations x two polarisations (128 rows -- the reference CLI's default geometry, lib/test-clxengine.cc:66): the whole-line kernel reads the
    // reference layout directly, no corner-turn kernel and no tile workspace (k_xe_i8_lines<false, 2>)
    if (g.mode == 0 && g.npol == 2 && !h->pad && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 &&
        mi355_xe_lines_ok(g.N, g.F, g.Fout, 2, g.T, stations_per_group, accumulate, 1, h->ctx->num_cus))
        return mi355_xe_lines_launch(in, out, g.N, g.F, g.Fout, g.T, 0.007874015748031496063, st, stations_per_group, 1, h->ctx->num_cus, 1, nullptr, 0, nullptr, 2);
    if (g.mode == 0 && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
        const XeFusedPlan fp = mi355_xe_fused_plan(g.N, g.F, g.Fout, g.npol, g.T, h->ctx->num_cus);
        if (fp.ok && (fp.part_bytes == 0 || (tiles && fp.part_bytes <= h->tile_bytes))) {
            const int ws = tiles == h->d_tiles ? 0 : 1;
            if (h->flags_stale[ws] && fp.part_bytes > fp.flag_offset) {  // counters of the in-launch reduction restart from zero
                MI355_HIP(hipMemsetAsync(tiles + fp.flag_offset, 0, fp.part_bytes - fp.flag_offset, st));
                h->fused_epoch[ws] = 0;
                h->flags_stale[ws] = false;
            }
            const int rc = mi355_xe_fused_launch(fp, in, out, tiles, g.N, g.F, g.Fout, g.T, 0.007874015748031496063, accumulate, st, stations_per_group,
                                                 &h->fused_epoch[ws]);
            if (rc != MI355_OK) h->flags_stale[ws] = true;  // whatever failed: the arrival words are zeroed and the count restarts before the next launch
            return rc;
        }
    }
    if (tiles) h->flags_stale[tiles == h->d_tiles ? 0 : 1] = true;  // the corner turn below writes over the whole workspace
    if (stations_per_group > 0 && stations_per_group < g.N) {
        mi355_set_error("antenna-group-major input needs the fused IChar path (<= 64 rows, rows of whole 16-byte pieces)");
        return MI355_ERR_UNSUPPORTED;
    }
    const bool fast_turn = row_bytes % 128 == 0 && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && !getenv("MI355_XE_SLOW_TURN");
    const int cpl = (g.mode == 1) ? 64 : 64 / g.npol;              // channels per 128-byte input line
    const int align = fast_turn ? cpl : 2;                          // slab boundaries: whole lines / whole 4-byte units
    int nslab = h->nslab;
    if (const char *e = getenv("MI355_XE_SLABS")) nslab = atoi(e) > 0 ? atoi(e) : nslab;
    int per = (g.F + nslab - 1) / nslab;
    per = (per + align - 1) / align * align;
    if (per > h->slab_channels) per = h->slab_channels / align * align;  // workspace capacity
    const int npairs = g.NT * (g.NT + 1) / 2;
    const double kd = (g.mode == 0) ? 0.007874015748031496063 : 0.142857142857142857143;  // :861, :835
    const bool lds_corr = !getenv("MI355_XE_NO_LDS");
    for (int f0 = 0; f0 < g.F; f0 += per) {
        XeGeo gs = g;
        gs.f0 = f0;
        gs.Fs = (g.F - f0 < per) ? g.F - f0 : per;
        if (fast_turn) {
            dim3 tgrid((unsigned)(gs.Fs / cpl), g.NT, g.KB * 4);
            if (g.mode == 1) hipLaunchKernelGGL((k_xe_turn_lds<2, true>), tgrid, dim3(256), 0, st, (const uint4 *)in, tiles, gs);
            else if (g.npol == 1) hipLaunchKernelGGL((k_xe_turn_lds<1, false>), tgrid, dim3(256), 0, st, (const uint4 *)in, tiles, gs);
            else hipLaunchKernelGGL((k_xe_turn_lds<2, false>), tgrid, dim3(256), 0, st, (const uint4 *)in, tiles, gs);
        } else {
            const int upc = (g.mode == 0 && g.npol == 2) ? 1 : 2;
            dim3 tgrid((gs.Fs / upc + 31) / 32, (g.N + 1) / 2, g.KB);
            hipLaunchKernelGGL(k_xe_turn, tgrid, dim3(256), 0, st, (const unsigned *)in, tiles, gs);
        }
        MI355_HIP(hipGetLastError());
#define CORR_LDS(NTT, WV, PPW)                                                                                               \
    hipLaunchKernelGGL((k_xe_corr_lds<NTT, WV, PPW>), dim3(gs.Fs, (npairs + WV * PPW - 1) / (WV * PPW)), dim3(WV * 64), 0, st,  \
                       (const unsigned char *)tiles, (c32 *)out, gs, npairs, kd, accumulate)
        if (g.NT >= 6 && g.NT <= 16 && g.NT % 2 == 0 && !getenv("MI355_XE_NO_SB") && !(g.NT <= 8 && getenv("MI355_XE_NO_SB8"))) {  // 65 .. 256 rows (the row-tile count was padded to an even number at create)
            SbArgs sa;
            sa.tiles = tiles; sa.out = (c32 *)out; sa.g = gs; sa.kd = kd; sa.accumulate = accumulate;
            sa.dbg = getenv("MI355_XE_DBG") ? atoi(getenv("MI355_XE_DBG")) : 0;
            const int cus = h->ctx->num_cus > 0 ? h->ctx->num_cus : 256;
            // 8 row tiles: the ring is 64 KiB and a workgroup four waves, so two workgroups share a CU -- one stores its matrix while the
            // other multiplies (what the 16-tile form cannot do: there one channel's accumulators fill half the CU's registers)
            const int wgs = cus * (g.NT == 8 ? 2 : g.NT == 6 ? 3 : 1);  // (6 row tiles: 48 KiB rings, three workgroups of three waves)
            sa.chan_per_wg = (gs.Fs + wgs - 1) / wgs;
            const unsigned grid = (unsigned)((gs.Fs + sa.chan_per_wg - 1) / sa.chan_per_wg);
#define CORR_SB(NTT)                                                                                                          \
    do {                                                                                                                      \
        constexpr int lds_bytes = 4 * 2 * NTT * kTileBytes + NTT * 64;                                                                  \
        if (g.npol == 1) {                                                                                                    \
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_corr_sb<NTT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
            hipLaunchKernelGGL((k_xe_corr_sb<NTT, 1>), dim3(grid), dim3(NTT * 32), lds_bytes, st, sa);                        \
        } else {                                                                                                              \
            MI355_HIP(hipFuncSetAttribute((const void *)k_xe_corr_sb<NTT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
            hipLaunchKernelGGL((k_xe_corr_sb<NTT, 2>), dim3(grid), dim3(NTT * 32), lds_bytes, st, sa);                        \
        }                                                                                                                     \
    } while (0)
            if (g.NT == 16) CORR_SB(16);
            else if (g.NT == 14) CORR_SB(14);
            else if (g.NT == 12) CORR_SB(12);
            else if (g.NT == 10) CORR_SB(10);
            else if (g.NT == 8) CORR_SB(8);
            else CORR_SB(6);
#undef CORR_SB
            mi355_xe_route_set(fast_turn ? "k_xe_turn_lds+k_xe_corr_sb" : "k_xe_turn+k_xe_corr_sb", 1, (int)grid, sa.chan_per_wg, 1, 0, 0, 0);
        }
        else if (lds_corr && (g.NT == 2 || g.NT == 4 || g.NT == 6 || g.NT == 8)) {
            if (g.NT == 2) CORR_LDS(2, 4, 1);        //  3 pairs
            else if (g.NT == 4) CORR_LDS(4, 4, 3);   // 10 pairs
            else if (g.NT == 6) CORR_LDS(6, 4, 6);   // 21 pairs
            else CORR_LDS(8, 8, 5);                  // 36 pairs: one workgroup per channel
            mi355_xe_route_set("k_xe_turn+k_xe_corr_lds", 1, 0, 1, 1, 0, 0, 0);
        } else {
            hipLaunchKernelGGL(k_xe_corr, dim3(gs.Fs, (npairs + kPairsPerWG - 1) / kPairsPerWG), dim3(256), 0, st,
                               (const unsigned char *)tiles, (c32 *)out, gs, npairs, kd, accumulate);
            mi355_xe_route_set("k_xe_turn+k_xe_corr", 1, 0, 1, 1, 0, 0, 0);
        }
#undef CORR_LDS
        MI355_HIP(hipGetLastError());
    }
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

void xe_route_commit(mi355_xengine *h)
{
    const mi355_xe_route &r = mi355_xe_route_tls;
    if (r.launches == 0) return;  // (nothing was enqueued: the handle keeps what its last launch recorded)
    h->route = r;
    char key[64];
    snprintf(key, sizeof(key), "%.40s/%d/%d/%d", r.kernel, r.windows, r.tsplit, r.units_per_workgroup);
    for (auto &seen : h->routes_seen) {
        if (!strcmp(seen, key)) return;
        if (!seen[0]) {
            snprintf(seen, sizeof(seen), "%s", key);
            mi355_log(h->ctx, MI355_LOG_DEBUG, "clXEngine %d x %d x %d frames: %s, %d window(s) per launch, %d workgroups x %d unit(s), %d time range(s)%s, touches %d, pace %d",
                      h->g.N, h->g.Fout, h->g.T, r.kernel, r.windows, r.workgroups, r.units_per_workgroup, r.tsplit,
                      r.in_launch_reduce ? " combined in the launch" : "", r.touches, r.pace);
            return;
        }
    }
}

}  // namespace

extern "C" int mi355_xengine_last_route(const mi355_xengine *h, mi355_xe_route *out)
{
    MI355_REQUIRE(h && out, "NULL argument");
    std::lock_guard<std::mutex> dl(const_cast<mi355_xengine *>(h)->dev_lock);
    *out = h->route;
    return MI355_OK;
}

extern "C" int mi355_xengine_destroy(mi355_xengine *h)
{
    if (!h) return MI355_OK;
    (void)hipSetDevice(h->ctx->device);
    if (h->d_tiles) (void)hipFree(h->d_tiles);
    for (auto &sl : h->slot) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_tiles && sl.d_tiles != h->d_tiles) (void)hipFree(sl.d_tiles);
        if (sl.d_pad && sl.d_pad != h->d_pad) (void)hipFree(sl.d_pad);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (h->d_pad) (void)hipFree(h->d_pad);
    if (h->d_batch) (void)hipFree(h->d_batch);
    for (auto &e : h->ws_done)
        if (e) (void)hipEventDestroy(e);
    delete h;
    return MI355_OK;
}

extern "C" int mi355_xengine_create(mi355_ctx *ctx, int data_type, int npol, int num_inputs, int num_channels, int integration,
                                    mi355_xengine **out)
{
    MI355_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    MI355_REQUIRE(data_type == MI355_DTYPE_COMPLEX || data_type == MI355_DTYPE_BYTE || data_type == MI355_DTYPE_PACKEDXY,
                  "X-engine data type must be complex, byte (IChar) or packed-4-bit");
    if (data_type == MI355_DTYPE_PACKEDXY) npol = 2;  // lib/clXEngine_impl.cc:176-178
    MI355_REQUIRE(npol == 1 || npol == 2, "polarization must be 1 or 2");
    MI355_REQUIRE(num_inputs >= 2, "Please specify at least 2 inputs to correlate.");  // :106-109
    MI355_REQUIRE(num_channels >= 1 && integration >= 1, "num_channels and integration must be positive");
    MI355_REQUIRE(integration <= 65536, "integration above 65536 frames would overflow the int32 accumulators");
    // the corner turn consumes 4-byte units of the input rows: an odd count of 2-byte channels gets one zero channel on the device
    int pad = (data_type != MI355_DTYPE_COMPLEX && ((size_t)num_channels * (data_type == MI355_DTYPE_BYTE ? npol * 2 : 2)) % 4 != 0) ? 1 : 0;
    // complex float: the matrix-core kernels read whole 128-byte lines (16 values) of every (t, station) row; other channel counts get
    // zero channels on the device up to the next whole line (1000 channels ran 4 x slower per sample than 1024 on the vector-ALU kernel,
    // 100 channels 30 x) -- no output is produced for them
    if (data_type == MI355_DTYPE_COMPLEX && !getenv("MI355_XE_CF32_NO_PAD")) {
        const int vals = num_channels * npol, rest = vals % 16;
        if (rest) pad = (16 - rest) / npol;  // (16 - rest is even when npol = 2: vals is)
    }
    mi355_xengine *h = new (std::nothrow) mi355_xengine();
    if (!h) return MI355_ERR_NOMEM;
    h->ctx = ctx; h->data_type = data_type;
    XeGeo &g = h->g;
    g.N = num_inputs; g.F = num_channels + pad; g.Fout = num_channels; g.npol = npol; g.T = integration;
    h->pad = pad;
    g.A = g.N * npol; g.NT = (g.A + kRowTile - 1) / kRowTile; g.KB = (g.T + kKBlock - 1) / kKBlock;
    // 65 .. 256 rows of int8 / 4-bit samples go to k_xe_corr_sb, whose waves own two tile rows each: an odd row-tile count gets one
    // zero tile row (written by the corner turn's grid, or left at the workspace's initial zero by the slow turn)
    if (data_type != MI355_DTYPE_COMPLEX && g.NT > 4 && g.NT <= 16) g.NT = (g.NT + 1) / 2 * 2;
    g.mode = (data_type == MI355_DTYPE_PACKEDXY) ? 1 : 0;
    const size_t items = (size_t)g.N * g.Fout * npol * g.T;
    h->in_bytes = items * mi355_dtype_size(data_type);  // frame_size_times_integration_bytes, :198
    h->out_items = (size_t)g.Fout * ((size_t)g.N * (g.N + 1) / 2) * npol * npol;
    g.f0 = 0; g.Fs = g.F;
    // slab size: bound the tile workspace to 4 GiB
    {
        const size_t per_chan = (size_t)g.KB * 2 * g.NT * kTileBytes;  // tile bytes per channel (>= input bytes per channel)
        int nslab = (int)((per_chan * g.F + ((size_t)4 << 30) - 1) / ((size_t)4 << 30));
        if (nslab < 1) nslab = 1;
        int per = (g.F + nslab - 1) / nslab;
        per = (per + 63) / 64 * 64;  // whole 128-byte input lines for every mode
        if (per > g.F) per = (g.F + 63) / 64 * 64;
        h->nslab = (g.F + per - 1) / per;
        h->slab_channels = per;
        h->tile_bytes = per_chan * per;
        if (data_type == MI355_DTYPE_COMPLEX)  // fp32 tiles: 16-time-step K blocks, whole problem in one pass
            h->tile_bytes = (size_t)g.F * ((g.T + kKB32 - 1) / kKB32) * 2 * xe_f32_row_tiles(g.NT) * kTileBytes;
        if (data_type == MI355_DTYPE_BYTE) {  // the fused path keeps the partial sums of its time ranges here
            const XeFusedPlan fp = mi355_xe_fused_plan(g.N, g.F, g.Fout, npol, g.T, ctx->num_cus);
            if (fp.ok && fp.part_bytes > h->tile_bytes) h->tile_bytes = fp.part_bytes;
        }
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { delete h; return MI355_ERR_HIP; }
    if (h->tile_bytes && hipMalloc((void **)&h->d_tiles, h->tile_bytes) != hipSuccess) {
        mi355_set_error("cannot allocate %zu bytes of tile workspace", h->tile_bytes);
        delete h;
        return MI355_ERR_NOMEM;
    }
    // the fill runs on the context's upload stream and is waited for there (a null-stream hipMemset may return before the fill has run
    // and is not ordered with the context's non-blocking streams: the first integration's corner turn could be overwritten by it)
    if (h->tile_bytes && mi355_fill(ctx, h->d_tiles, 0, h->tile_bytes) != hipSuccess) {
        mi355_xengine_destroy(h);
        return MI355_ERR_HIP;
    }
    if (pad) {
        h->pad_bytes = data_type == MI355_DTYPE_COMPLEX ? (size_t)g.T * g.N * g.F * npol * 8 : (size_t)g.T * g.N * g.F * 2;
        if (hipMalloc((void **)&h->d_pad, h->pad_bytes) != hipSuccess) { mi355_xengine_destroy(h); return MI355_ERR_NOMEM; }
    }
    mi355_log(ctx, MI355_LOG_INFO, "clXEngine: %d inputs x %d pol, %d channels, %d frames per integration, %s input: %zu input bytes, %zu output items, %zu workspace bytes",
              g.N, npol, g.Fout, g.T, data_type == MI355_DTYPE_COMPLEX ? "complex" : data_type == MI355_DTYPE_BYTE ? "IChar" : "packed 4-bit",
              h->in_bytes, h->out_items, h->tile_bytes);
    *out = h;
    return MI355_OK;
}

extern "C" size_t mi355_xengine_input_bytes(const mi355_xengine *h) { return h ? h->in_bytes : 0; }
extern "C" size_t mi355_xengine_output_items(const mi355_xengine *h) { return h ? h->out_items : 0; }

// The same with the input as the blocks of an all-to-all corner turn: [group][t][station in group][chan][pol] (every group
// = stations_per_group consecutive stations, one contiguous block per sending rank).  Only the fused IChar path reads this
// in place; other geometries return MI355_ERR_UNSUPPORTED (re-lay the blocks out with mi355_pack3d_dev first).
extern "C" int mi355_xengine_xcorrelate_grouped_dev(mi355_xengine *h, const void *in_dev, void *out_dev, int accumulate, int stations_per_group,
                                                    void *stream)
{
    MI355_REQUIRE(h && in_dev && out_dev, "NULL argument");
    MI355_REQUIRE(stations_per_group >= 1 && h->g.N % stations_per_group == 0, "stations_per_group must divide the number of inputs");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in_dev) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out_dev) & 7u) == 0,
                  "device buffers must be 16-byte (input) / 8-byte (output) aligned");
    MI355_REQUIRE(h->data_type == MI355_DTYPE_BYTE && !h->pad, "group-major input: IChar with an even channel count only");
    MI355_HIP(hipSetDevice(h->ctx->device));
    std::lock_guard<std::mutex> dl(h->dev_lock);
    xe_route_begin();
    const int rc = launch_xe(h, in_dev, out_dev, accumulate, mi355_pick_stream(h->ctx, stream), h->d_tiles, h->d_pad, stations_per_group);
    xe_route_commit(h);
    return rc;
}

// nint windows in ONE launch where the geometry has such a kernel (the caller holds dev_lock)
static int xe_n_dev_launch(mi355_xengine *h, int nint, const void *in_dev, void *out_dev, int accumulate, int stations_per_group, hipStream_t st)
{
    const XeGeo &g = h->g;
    const bool grouped = stations_per_group > 0 && stations_per_group < g.N;
    // 64 stations x two polarisations: all windows in ONE launch of the whole-line kernel, a workgroup's units back to back (a unit's matrix stores run
    // under the next unit's first requests); no workspace
    // (from four windows on -- 95.7 / 94.3 us per window at 4 / 8 windows per launch against 98.9 one launch per window, 102.8 at two per launch -- and for
    // group-major input, whose windows a loop of launches cannot address)
    if (h->data_type == MI355_DTYPE_BYTE && g.npol == 2 && (nint >= 4 || (grouped && nint > 1)) && !h->pad && (reinterpret_cast<uintptr_t>(in_dev) & 15u) == 0 &&
        mi355_xe_lines_ok(g.N, g.F, g.Fout, 2, g.T, stations_per_group, accumulate, nint, h->ctx->num_cus))
        return mi355_xe_lines_launch(in_dev, out_dev, g.N, g.F, g.Fout, g.T, 0.007874015748031496063, st, stations_per_group, nint, h->ctx->num_cus, 1, nullptr, 0,
                                     nullptr, 2);
    // one launch for all windows: the fused IChar path (<= 64 rows, rows of whole 16-byte pieces, 16-byte aligned input)
    if (h->data_type == MI355_DTYPE_BYTE && !h->pad && (reinterpret_cast<uintptr_t>(in_dev) & 15u) == 0) {
        const XeFusedPlan fp = mi355_xe_fused_plan(g.N, g.F, g.Fout, g.npol, g.T, h->ctx->num_cus, nint);
        if (fp.ok) {
            if (fp.part_bytes > h->batch_bytes) {  // (first call at this batch size: the only allocation of the device path)
                std::lock_guard<std::mutex> lk(h->ctx->lock);
                // a launch at a smaller batch size may still be running on ANOTHER stream: wait for the device, not for this stream
                MI355_HIP(hipDeviceSynchronize());
                if (h->d_batch) MI355_HIP(hipFree(h->d_batch));
                h->d_batch = nullptr;
                h->batch_bytes = 0;
                if (hipMalloc((void **)&h->d_batch, fp.part_bytes) != hipSuccess) {
                    mi355_set_error("cannot allocate %zu bytes of partial-sum workspace for %d integration windows", fp.part_bytes, nint);
                    return MI355_ERR_NOMEM;
                }
                h->batch_bytes = fp.part_bytes;
                h->batch_nint = 0;
            }
            if (fp.part_bytes > 0) {  // (no time ranges -- the whole-line kernel, the persistent form -- no workspace: nothing to order, streams stay independent)
                const int rc = xe_order_workspace(h, 2, st);  // (before the fill below: it must not run under a launch still in flight on another stream)
                if (rc != MI355_OK) return rc;
            }
            if (h->batch_nint != nint && fp.part_bytes > fp.flag_offset) {  // the arrival words of this window count start from zero
                MI355_HIP(hipMemsetAsync(h->d_batch + fp.flag_offset, 0, fp.part_bytes - fp.flag_offset, st));
                h->batch_epoch = 0;
                h->batch_nint = nint;
            }
            const int rc = mi355_xe_fused_launch(fp, in_dev, out_dev, h->d_batch, g.N, g.F, g.Fout, g.T, 0.007874015748031496063, accumulate, st,
                                                 stations_per_group, &h->batch_epoch, nint);
            if (rc != MI355_OK) h->batch_nint = 0;  // whatever failed: the arrival words are zeroed and the count restarts before the next launch
            return rc;
        }
    }
    if (grouped && nint > 1) {
        mi355_set_error("group-major input of several windows needs the fused IChar path (<= 64 rows, rows of whole 16-byte pieces)");
        return MI355_ERR_UNSUPPORTED;
    }
    // every other geometry / sample format: one window after the other through the handle's workspace (stream ordered)
    const size_t out_bytes = h->out_items * 8;
    for (int i = 0; i < nint; i++) {
        const int rc = launch_xe(h, (const char *)in_dev + (size_t)i * h->in_bytes, (char *)out_dev + (size_t)i * out_bytes, accumulate, st, h->d_tiles,
                                 h->d_pad, grouped ? stations_per_group : 0);
        if (rc != MI355_OK) return rc;
    }
    return MI355_OK;
}

extern "C" int mi355_xengine_xcorrelate_n_dev(mi355_xengine *h, int nint, const void *in_dev, void *out_dev, int accumulate, int stations_per_group,
                                              void *stream)
{
    MI355_REQUIRE(h && in_dev && out_dev, "NULL argument");
    MI355_REQUIRE(nint >= 1, "nint must be >= 1");
    MI355_REQUIRE(stations_per_group == 0 || (stations_per_group >= 1 && h->g.N % stations_per_group == 0),
                  "stations_per_group must be 0 (reference layout) or divide the number of inputs");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in_dev) & (h->data_type == MI355_DTYPE_COMPLEX ? 7u : 3u)) == 0 &&
                      (reinterpret_cast<uintptr_t>(out_dev) & 7u) == 0,
                  "device buffers must be 4-byte (int8 / packed input) or 8-byte (complex input, output) aligned");
    MI355_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = mi355_pick_stream(h->ctx, stream);
    std::lock_guard<std::mutex> dl(h->dev_lock);
    XeRouteScope route_scope(h);
    const XeGeo &g = h->g;
    const bool grouped = stations_per_group > 0 && stations_per_group < g.N;
    // Window counts between the good ones.  The whole-line kernel takes the counts whose units split into equal shares over (nearly) all CUs -- at
    // BASELINE config 5: 4, 7, 8, 12, 14, 16, ... -- and runs them at 35-37 us per window; in between the 32-byte-slice kernel needs a second round of
    // workgroups for a few units (5 / 6 / 10 windows per launch: 71 / 62 / 55 us per window).  Such a call is cut into stream-ordered launches of good
    // counts, the rest in twos and ones (5 = 4 + 1, 6 = 4 + 2, 10 = 8 + 2, 11 = 8 + 2 + 1); windows are independent, so nothing else changes.
    // (Reference layout only: in the group-major layout a window's address depends on the call's window count.  MI355_XE_NO_SPLIT=1: one launch.)
    if (nint > 2 && !grouped && h->data_type == MI355_DTYPE_BYTE && !h->pad && !getenv("MI355_XE_NO_SPLIT")) {
        auto good = [&](int c) { return mi355_xe_lines_ok(g.N, g.F, g.Fout, g.npol, g.T, 0, accumulate, c, h->ctx->num_cus); };
        int top = 0;
        for (int c = nint; c >= 1 && !top; c--)
            if (good(c)) top = c;
        for (int c = nint + 1; c <= 64 && !top; c++)  // (fewer windows than the smallest good count: the rest rule below still applies -- 3 = 2 + 1)
            if (good(c)) top = -1;
        if (top != 0 && !good(nint)) {
            const size_t out_bytes = h->out_items * 8;
            int done = 0;
            while (done < nint) {
                const int rem = nint - done;
                int c = 0;
                for (int k = rem; k >= 1 && !c; k--)
                    if (good(k)) c = k;
                // (the rest: at most two windows per launch where one window is a quarter of the device or more -- config 5: one / two / three windows
                // in one launch 62 / 117 / 208 us -- otherwise all of it in one)
                if (!c) c = ((long)(g.F / 64) * 4 * 4 >= h->ctx->num_cus && rem > 2) ? 2 : rem;
                const int rc = xe_n_dev_launch(h, c, (const char *)in_dev + (size_t)done * h->in_bytes, (char *)out_dev + (size_t)done * out_bytes, accumulate, 0, st);
                if (rc != MI355_OK) return rc;
                done += c;
            }
            return MI355_OK;
        }
    }
    return xe_n_dev_launch(h, nint, in_dev, out_dev, accumulate, stations_per_group, st);
}

extern "C" int mi355_xengine_xcorrelate_dev(mi355_xengine *h, const void *in_dev, void *out_dev, int accumulate, void *stream)
{
    MI355_REQUIRE(h && in_dev && out_dev, "NULL argument");
    MI355_REQUIRE((reinterpret_cast<uintptr_t>(in_dev) & (h->data_type == MI355_DTYPE_COMPLEX ? 7u : 3u)) == 0 &&
                      (reinterpret_cast<uintptr_t>(out_dev) & 7u) == 0,
                  "device buffers must be 4-byte (int8 / packed input) or 8-byte (complex input, output) aligned");
    MI355_HIP(hipSetDevice(h->ctx->device));
    std::lock_guard<std::mutex> dl(h->dev_lock);
    xe_route_begin();
    const int rc = launch_xe(h, in_dev, out_dev, accumulate, mi355_pick_stream(h->ctx, stream), h->d_tiles, h->d_pad);
    xe_route_commit(h);
    return rc;
}

namespace {
int slot_prepare(mi355_xengine *h, int s)
{
    mi355_xengine::Slot &sl = h->slot[s];
    if (sl.d_in) return MI355_OK;
    const size_t outb = h->out_items * 8;
    MI355_HIP(hipMalloc(&sl.d_in, h->in_bytes));
    MI355_HIP(hipMalloc(&sl.d_out, outb));
    MI355_HIP(hipHostMalloc(&sl.h_in, h->in_bytes, hipHostMallocDefault));
    MI355_HIP(hipHostMalloc(&sl.h_out, outb, hipHostMallocDefault));
    MI355_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (h->pad) {
        if (s == 0) sl.d_pad = h->d_pad;
        else MI355_HIP(hipMalloc((void **)&sl.d_pad, h->pad_bytes));
    }
    if (s == 0 || h->tile_bytes == 0) sl.d_tiles = h->d_tiles;
    else {
        MI355_HIP(hipMalloc((void **)&sl.d_tiles, h->tile_bytes));
        // padding rows stay zero.  On the slot's own stream: a null-stream fill is not ordered with the kernels that follow on a
        // non-blocking stream and, when the device is busy, ran AFTER the slot's first corner turn (one wrong integration in ~2000)
        MI355_HIP(hipMemsetAsync(sl.d_tiles, 0, h->tile_bytes, h->ctx->stream[s]));
    }
    return MI355_OK;
}
}  // namespace

// Enqueue one integration: copy into the slot's pinned buffer, H2D, kernels, D2H on the slot's stream.
// At most two integrations are in flight; a third submit() is refused until wait() frees a slot.
// accumulate != 0 adds into the slot's previous result only when used with one slot in flight
// (pipeline integration keeps the accumulator on the host side of the block, like the reference).
static int xe_submit_slot(mi355_xengine *h, const void *in_host /* nullptr: the slot's pinned buffer is already filled */,
                          const void *acc_host)
{
    const int s = h->next_submit;
    int rc = slot_prepare(h, s);
    if (rc) return rc;
    mi355_xengine::Slot &sl = h->slot[s];
    hipStream_t st = h->ctx->stream[s];
    const size_t outb = h->out_items * 8;
    if (in_host) {
        // piece by piece: the transfer of a piece runs under the staging copy of the next one (one copy of the whole window followed
        // by one transfer took 2.2 + 2.4 ms at config 5; the transfer alone is the floor)
        const size_t piece = (size_t)8 << 20;
        for (size_t off = 0; off < h->in_bytes; off += piece) {
            const size_t nb = h->in_bytes - off < piece ? h->in_bytes - off : piece;
            mi355_copy((char *)sl.h_in + off, (const char *)in_host + off, nb);
            MI355_HIP(hipMemcpyAsync((char *)sl.d_in + off, (const char *)sl.h_in + off, nb, hipMemcpyHostToDevice, st));
        }
    } else {
        MI355_HIP(hipMemcpyAsync(sl.d_in, sl.h_in, h->in_bytes, hipMemcpyHostToDevice, st));
    }
    if (acc_host) {
        mi355_copy(sl.h_out, acc_host, outb);
        MI355_HIP(hipMemcpyAsync(sl.d_out, sl.h_out, outb, hipMemcpyHostToDevice, st));
    }
    {
        XeRouteScope route_scope(h);  // (callers hold dev_lock)
        rc = launch_xe(h, sl.d_in, sl.d_out, acc_host != nullptr, st, sl.d_tiles, sl.d_pad);
    }
    if (rc) return rc;
    MI355_HIP(hipMemcpyAsync(sl.h_out, sl.d_out, outb, hipMemcpyDeviceToHost, st));
    MI355_HIP(hipEventRecord(sl.done, st));
    sl.busy = true;
    h->acquired = false;
    h->next_submit ^= 1;
    h->pending++;
    return MI355_OK;
}

extern "C" int mi355_xengine_submit(mi355_xengine *h, const void *in_host, const void *acc_host)
{
    MI355_REQUIRE(h && in_host, "NULL argument");
    std::lock_guard<std::mutex> dl(h->dev_lock);  // (slot 0 shares its workspace with the device-pointer path; lock order: dev_lock, then the context's)
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    if (h->pending == 2) {
        mi355_set_error("two integrations already in flight: call mi355_xengine_wait first");
        return MI355_ERR_STATE;
    }
    MI355_REQUIRE(!h->acquired, "a frame buffer is acquired: finish it with mi355_xengine_submit_acquired");
    return xe_submit_slot(h, in_host, acc_host);
}

// Zero-copy form: hand out the pinned frame buffer of the next free slot so the block gathers its frames straight into
// it (the reference gathers into its pinned char_input / complex_input, lib/clXEngine_impl.cc:325-362,987-1061), then
// submit_acquired() enqueues H2D + kernels + D2H without another host copy.
extern "C" int mi355_xengine_acquire(mi355_xengine *h, void **frame_buffer)
{
    MI355_REQUIRE(h && frame_buffer, "NULL argument");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    if (h->pending == 2) {
        mi355_set_error("two integrations already in flight: call mi355_xengine_wait first");
        return MI355_ERR_STATE;
    }
    int rc = slot_prepare(h, h->next_submit);
    if (rc) return rc;
    h->acquired = true;
    *frame_buffer = h->slot[h->next_submit].h_in;
    return MI355_OK;
}

extern "C" int mi355_xengine_submit_acquired(mi355_xengine *h, const void *acc_host)
{
    MI355_REQUIRE(h != nullptr, "NULL argument");
    std::lock_guard<std::mutex> dl(h->dev_lock);
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    MI355_REQUIRE(h->acquired, "no frame buffer acquired");
    return xe_submit_slot(h, nullptr, acc_host);
}

// Block until the OLDEST submitted integration is complete and copy its matrix to out_host.
extern "C" int mi355_xengine_wait(mi355_xengine *h, void *out_host)
{
    MI355_REQUIRE(h && out_host, "NULL argument");
    std::lock_guard<std::mutex> g(h->ctx->lock);
    MI355_HIP(hipSetDevice(h->ctx->device));
    if (h->pending == 0) {
        mi355_set_error("nothing in flight");
        return MI355_ERR_STATE;
    }
    mi355_xengine::Slot &sl = h->slot[h->next_wait];
    MI355_HIP(hipEventSynchronize(sl.done));
    mi355_copy(out_host, sl.h_out, h->out_items * 8);
    sl.busy = false;
    h->next_wait ^= 1;
    h->pending--;
    return MI355_OK;
}

extern "C" int mi355_xengine_pending(const mi355_xengine *h) { return h ? h->pending : MI355_ERR_INVALID_ARG; }

// Synchronous form = the reference's xcorrelate() + blocking read-back (lib/clXEngine_impl.h:179-201, .cc:1257)
extern "C" int mi355_xengine_xcorrelate(mi355_xengine *h, const void *in_host, void *out_host, int accumulate)
{
    MI355_REQUIRE(h && in_host && out_host, "NULL argument");
    MI355_REQUIRE(h->pending == 0, "asynchronous integrations are still in flight");
    int rc = mi355_xengine_submit(h, in_host, accumulate ? out_host : nullptr);
    if (rc) return rc;
    return mi355_xengine_wait(h, out_host);
}

// Host frame gather of work_processor, lib/clXEngine_impl.cc:987-1061 (plain memcpy loops on the
// caller's thread, like the reference; the frame buffer keeps the reference's layout).
extern "C" int mi355_xengine_gather(const mi355_xengine *h, int nframes, int frame0, const void *const *inputs, void *frame_buffer)
{
    MI355_REQUIRE(h && inputs && frame_buffer, "NULL argument");
    const XeGeo &g = h->g;
    MI355_REQUIRE(nframes >= 0 && frame0 >= 0 && frame0 + nframes <= g.T, "frames outside the integration window");
    const size_t esz = mi355_dtype_size(h->data_type);
    const size_t frame_bytes = (size_t)g.Fout * g.N * g.npol * esz;
    struct Job { const mi355_xengine *h; int nframes, frame0; const void *const *inputs; char *dst; size_t esz, frame_bytes; } job =
        {h, nframes, frame0, inputs, (char *)frame_buffer, esz, frame_bytes};
    auto part = [](void *a, int p, int parts) {
        const Job &j = *(const Job *)a;
        const XeGeo &g = j.h->g;
        const int per = (j.nframes + parts - 1) / parts, b0 = p * per, b1 = b0 + per < j.nframes ? b0 + per : j.nframes;
        for (int b = b0; b < b1; b++) {
            char *fb = j.dst + j.frame_bytes * (size_t)(j.frame0 + b);
            for (int i = 0; i < g.N; i++) {
                if (g.npol == 1 || j.h->data_type == MI355_DTYPE_PACKEDXY) {
                    const size_t row = (size_t)g.Fout * g.npol * j.esz;
                    memcpy(fb + (size_t)i * row, (const char *)j.inputs[i] + (size_t)b * row, row);
                } else {
                    const char *x = (const char *)j.inputs[i] + (size_t)b * g.Fout * j.esz;
                    const char *y = (const char *)j.inputs[i + g.N] + (size_t)b * g.Fout * j.esz;
                    char *row = fb + (size_t)i * g.Fout * 2 * j.esz;
                    // X and Y of a station arrive as two streams and are interleaved per channel (lib/clXEngine_impl.cc:1020-1045).  Fixed-size
                    // copies the compiler turns into vector shuffles: the general memcpy(esz) loop moved 2 bytes per call
                    if (j.esz == 2) {
                        for (int c = 0; c < g.Fout; c++) {
                            unsigned short a, b;
                            memcpy(&a, x + (size_t)c * 2, 2);
                            memcpy(&b, y + (size_t)c * 2, 2);
                            const unsigned int v = (unsigned int)a | ((unsigned int)b << 16);
                            memcpy(row + (size_t)c * 4, &v, 4);
                        }
                    } else if (j.esz == 8) {
                        for (int c = 0; c < g.Fout; c++) {
                            memcpy(row + (size_t)c * 16, x + (size_t)c * 8, 8);
                            memcpy(row + (size_t)c * 16 + 8, y + (size_t)c * 8, 8);
                        }
                    } else {
                        for (int c = 0; c < g.Fout; c++) {
                            memcpy(row + (size_t)c * 2 * j.esz, x + (size_t)c * j.esz, j.esz);
                            memcpy(row + (size_t)c * 2 * j.esz + j.esz, y + (size_t)c * j.esz, j.esz);
                        }
                    }
                }
            }
        }
    };
    // many frames per call (a GNU Radio work() call carries hundreds): split them over the helper pool; a single frame stays here
    if (!(nframes >= 8 && frame_bytes * (size_t)nframes >= (2u << 20) && mi355_parallel(part, &job))) part(&job, 0, 1);
    return MI355_OK;
}
