// FUSION NET: eval-mode FusionNet_v3 / FusionNet_v2 (modules/model.py:164-283) as a chain of fused
// convolution kernels on the gfx950 matrix cores.
//
// Arithmetic (ojf_net_set_arithmetic, captured per net at creation), both with fp32 accumulation and
// fp32 activations in memory; tsdf_est agrees with the reference's fp32 CPU net to a few 1e-6
// (tolerance stated in the tests: 1e-5).  The +-0.1 truncation band of the TSDF rules out plain
// bf16/fp16 operands.
//   OJF_ARITH_F32   v_mfma_f32_16x16x4_f32: bitwise a k-ordered fmaf chain; 1024 MAC per 32 cycles.
//   OJF_ARITH_F16X3 (default) split-fp16: x = xh + xl, w = wh + wl with fp16 halves (round to nearest);
//                   x*w ~= wl*xh + wh*xl + wh*xh on v_mfma_f32_16x16x32_f16 (8192 MAC per 16 cycles,
//                   three of them per product block: 5.3x the fp32-input MFMA rate).  The dropped wl*xl
//                   term is <= 2^-22 |x w|; measured layer error is below the fp32 chain's
//                   (tests/microbench/conv_f16x3_bench.hip).  Weight rows are equilibrated by powers of two
//                   (row_scale) so that BN-folded rows of any magnitude keep their mantissa.  Operands must
//                   stay below 65504 in magnitude (fp16 range) - true for TSDF values, fp16 volume weights,
//                   metric depth and the BN-folded activations of a trained net; every kernel guards it
//                   (guard_max, ojf_net_check) and OJF_ARITH_F32 has no such limit.
//
// Data layout ("C4 planes"): an activation tensor with C channels (padded to a multiple of 4:
// 19 -> 20, 114 -> 116; pad channels carry zeros) is stored as C/4 planes of float4, element
// (channel group cg, pixel p) at plane[cg * npix + p].  A wave's 16 consecutive pixels of one
// channel group are 256 contiguous bytes, so every MFMA operand fetch and every result store is a
// fully coalesced 16-byte-per-lane access WITHOUT an LDS transpose (the first version used NHWC rows:
// its fragment-shaped loads touched 64 different cache lines per wave instruction and the kernel
// sat at the texture-addresser limit, ~4x off the MFMA bound).
//
// A convolution is an implicit GEMM  D[oc, pixel] = sum_K W[oc, K] * X[K, pixel]:
//   MFMA rows = 16 output channels (A operand = packed weights), MFMA cols = 16 consecutive pixels
//   (B operand = activations), K = flattened list of (tap, 4-channel group) pairs, G = tap*c4 + cg.
//   fp32: in superstep S lane group g (= lane >> 4) owns group G = 4S + g: it fetches that group's float4 of
//   ITS tap's source pixel and feeds element j to MFMA j of the superstep; split-fp16: a superstep is one
//   32-wide K block, lane group g owns groups 8S + 2g and 8S + 2g + 1, splits its 8 values into fp16 halves
//   and issues 3 MFMAs per output tile.  The packed weights use the same (g, j) permutation.  Any channel count
//   that is a multiple of 4 works without tails, out-of-image taps contribute zeros, supersteps whose every
//   source pixel is outside the image are skipped (dilation 9 / 27 near the borders), and the loop is software
//   pipelined (three operand stages in flight).
//   The accumulator of lane (pixel i, g) holds output channels 4g..4g+3 of its pixel: the epilogue
//   adds bias, applies the activation and writes ONE float4 into the output plane.
//
// Algebraic restructuring of VortexPooling (model.py:100-161), exact up to fp32 rounding:
//   * the four branch-entry 1x1 convolutions commute with the (linear) 3x3 average pools, so they
//     run as ONE 1x1 GEMM on x with 4*mid output channels, and the pools run on mid (19) channels
//     instead of in_chs (114/228);
//   * the global-average branch is constant over the image: it is reduced to a per-frame bias of
//     the final 1x1 convolution (two tiny kernels on a side stream), removing its 114 input columns from that GEMM.
// Launches of one forward pass (geometry-only v3): 10 dense-block convolutions, per VortexPooling {branch-entry
// GEMM, pool pyramid, two grouped launches of the four branches' dilated 3x3, fused tail}; the last tail also runs
// the 11-layer prediction head: 22 launches on the main stream.  Environment switches (tuning / ablation only):
// OJF_CONV_MT, OJF_NO_TAIL, OJF_NO_CHAIN, OJF_NO_HEAD_FUSION, OJF_NET_GRAPH=1 (opt-in hipGraph replay).
#ifndef OJF_CHAIN_DMA_HALF
#define OJF_CHAIN_DMA_HALF 1536
#endif
#ifndef OJF_CHAIN_BLOCKS
#define OJF_CHAIN_BLOCKS 3
#endif
#ifndef OJF_CHAIN_WAVES
#define OJF_CHAIN_WAVES 4
#endif
#ifndef OJF_CHAIN_AHEAD
#define OJF_CHAIN_AHEAD 2
#endif
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ojf_common.h"

namespace ojf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// x (8 fp32 values) -> fp16 halves xh + xl, both rounded to nearest; x - xh is exact in fp32
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_f16(const f32x4 &a, const f32x4 &b, f16x8 &hi, f16x8 &lo)
{
    const f32x8 x = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    hi = __builtin_convertvector(x, f16x8);  // 4 x v_cvt_pk_f16_f32
    const u32x4 h = __builtin_bit_cast(u32x4, hi);
    u32x4 l;
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = split_lo_pair(h[i], x[2 * i], x[2 * i + 1]);
    lo = __builtin_bit_cast(f16x8, l);
}

// One float4 -> its split-fp16 form in the same 16 bytes: halfs {h0 h1 h2 h3 | l0 l1 l2 l3}, the very halves split_f16
// makes of these values (same pairs, same instructions).  A producer that stores its planes this way ("split planes")
// spares every consumer tap the split: the grouped 3x3 convolutions re-split each input value nine times, 12 of the
// ~19 VALU instructions a 16-pixel tile costs per superstep.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 split_pack4(const f32x4 &x)
{
    const f16x4 hi = __builtin_convertvector(x, f16x4);  // 2 x v_cvt_pk_f16_f32
    const u32x2 h = __builtin_bit_cast(u32x2, hi);
    const u32x4 r{h[0], h[1], split_lo_pair(h[0], x[0], x[1]), split_lo_pair(h[1], x[2], x[3])};
    return __builtin_bit_cast(f32x4, r);
}
// split planes -> fp32: hi + lo (exact: the halves have 11 + 11 significant bits)
__device__ __forceinline__ f32x4 unsplit4(const f32x4 &s)
{
    typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
    const f16x8_ h = __builtin_bit_cast(f16x8_, s);
    return f32x4{(float)h[0] + (float)h[4], (float)h[1] + (float)h[5], (float)h[2] + (float)h[6], (float)h[3] + (float)h[7]};
}
// two split-plane float4 (channel groups A, B) -> the MFMA operand halves of their eight channels:
// (A0 A1 A2 A3)(B0 B1 B2 B3) -> (A0 A1 B0 B1)(A2 A3 B2 B3).
// Round 3 did this in place with two hand-placed v_swap_b32 (inline assembly: as plain vector shuffles the compiler spends
// moves).  Round 4 found that form UNSAFE next to MFMAs: hipcc's hazard recogniser does not look inside inline assembly, so
// nothing keeps a swap from rewriting a register an MFMA in flight still reads (or an MFMA from reading a register the
// swap has not finished writing) - dense_chain_kernel's packed-row loop computed wrong sums for exactly the operand dwords
// the swaps touch, on some waves, run after run, and a padding s_nop 7 did not cure all of it.  The convolution kernels
// never failed a test with it, but "never failed" is not a proof: plain shuffles (OJF_UNSAFE_SWAP=1 restores the old form
// for measurements: 2539 / 2542 frames/s with the shuffles against 2533 / 2538 with the swaps on one box, no difference).
#ifndef OJF_UNSAFE_SWAP
#define OJF_UNSAFE_SWAP 0
#endif
__device__ __forceinline__ void unpack_split(const f32x4 &a, const f32x4 &b, f16x8 &hi, f16x8 &lo)
{
#if OJF_UNSAFE_SWAP
    u32x4 A = __builtin_bit_cast(u32x4, a), B = __builtin_bit_cast(u32x4, b);
    unsigned a2 = A[2], a3 = A[3], b0 = B[0], b1 = B[1];
    asm("v_swap_b32 %0, %1" : "+v"(a2), "+v"(b0));
    asm("v_swap_b32 %0, %1" : "+v"(a3), "+v"(b1));
    A[2] = a2; A[3] = a3; B[0] = b0; B[1] = b1;
    hi = __builtin_bit_cast(f16x8, A);
    lo = __builtin_bit_cast(f16x8, B);
#else
    hi = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 0, 1, 4, 5));
    lo = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 2, 3, 6, 7));
#endif
}

// acc += W * X for one 16x16 tile over a 32-wide K block, W and X given as split halves
__device__ __forceinline__ f32x4 mfma_f16x3(const f32x4 &wh, const f32x4 &wl, const f16x8 &xh, const f16x8 &xl, f32x4 acc)
{
    const f16x8 h = __builtin_bit_cast(f16x8, wh), l = __builtin_bit_cast(f16x8, wl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xh, acc, 0, 0, 0);
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const f32x4 *in;   // input planes (in[-1] is a zero float4); group cg of the window = plane in_g0 + cg
    f32x4 *out;        // output planes (NULL when out_rows is used)
    float *out_rows;   // last layer only: row-major [npix, rows_stride] scalars, channels < rows_n
    const f32x4 *wp;   // packed weights [oc tile][superstep][lane] float4
    const float *bias; // [n_ot*16]
    const float *rinv; // split-fp16: [n_ot*16] inverse of the per-output-channel power-of-two weight scale
    int in_g0, out_g0, rows_stride, rows_n;
    int h, w, npix;
    int taps, dil;
    int c4;          // input channel groups per tap
    int nsteps;      // supersteps = ceil(taps * c4 / 4)
    int og_store;    // output channel groups written (planar mode)
    int act, act_n;  // activation applied to output channels < act_n
    float scale;
    int *ovf;        // split-fp16 only: set to 1 when an activation leaves the fp16 range (host-mapped flag)
    int accum;       // planar stores only: out += result (fan-out gradients of the training path); 0 = plain store
    const float *dscale;  // training backward-data: the input carries a power-of-two factor *dscale - the result is divided by it; NULL = off
    unsigned w_magic, c4_magic;  // ceil(2^32 / w), ceil(2^32 / c4): x / d == umulhi(x, magic) while x * d < 2^32 (0: divide)
    int in_split = 0, out_split = 0;  // split-fp16 launches: the input / output planes are split planes (split_pack4)
    // Banded split-fp16 3x3 launches with dilation r > 1 (round 6): the rows of the image are handed to the blocks in the order
    // (y mod r, y div r) instead of y.  An XCD's band of 30 rows of a 240-row frame then holds rows that are r apart - the
    // very rows its taps read - instead of 30 neighbours whose taps reach 27 rows into both neighbouring bands: with plain
    // bands every private L2 pulled (30 + 2 r) / 30 of its share through the fabric (2.8x for r = 27, 1.6x for r = 9: the
    // grouped launch fetched 40 MB for a 24.6-MB input, profiles/r05_final_traffic_pmc.txt); in class order the halo is one
    // row on either side of the band.  Which block computes which pixels changes, nothing else: the same bits.
    int row_perm = 0, perm_q = 0, perm_rem = 0;  // r (0: off), h / r, h % r
};

// position `pr` of the row order (y mod r, y div r) -> row y; classes c < h % r have h / r + 1 rows, the others h / r
__device__ __forceinline__ int perm_row(int pr, int r, int q, int rem)
{
    const int big = rem * (q + 1);
    if (pr < big) {
        const int c = pr / (q + 1);
        return c + (pr - c * (q + 1)) * r;
    }
    const int p2 = pr - big, c = p2 / q;
    return rem + c + (p2 - c * q) * r;
}

__device__ __forceinline__ int fast_div(int x, int d, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x / d; }

// Split-fp16 range guard.  Kernels keep a running maximum of the magnitudes of every value a later layer will
// split (2 VALU per float4: v_max3_f32 with |.| modifiers) and raise the flag if it exceeds the fp16 range.
// v_max ignores NaN operands, deliberately: a NaN input (the reference writes NaN TSDF into voxels it touches with
// zero total weight, and later frames gather them) propagates to NaN outputs in both arithmetics exactly as in the
// fp32 reference, so it is not an error here either.  With finite weights (checked at creation) an out-of-range
// event is always a finite value beyond 65504 or an infinity at a guarded point first.
__device__ __forceinline__ float guard_max(float mx, const f32x4 &v)
{
    return fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
}
__device__ __forceinline__ bool beyond_f16(const f32x4 &v)  // raw inputs; NaN is not a range violation (see above)
{
    return guard_max(0.0f, v) > 65504.0f;
}
__device__ __forceinline__ f32x4 fma4(const f32x4 &a, const f32x4 &b, const f32x4 &c)
{
    return f32x4{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1]), __builtin_fmaf(a[2], b[2], c[2]),
                 __builtin_fmaf(a[3], b[3], c[3])};
}

// ReLU / LeakyReLU / identity are one select with a slope (0, 0.01, 1); Tanh is a separate,
// wave-uniform path (last layer only)
__device__ __forceinline__ float act_slope(int act)
{
    return act == OJF_ACT_RELU ? 0.0f : (act == OJF_ACT_LEAKY ? 0.01f : 1.0f);
}
// x > 0 ? x : slope * x for 0 <= slope <= 1 as max(x, slope * x): one multiply (packed with its neighbour) + one max
// instead of compare + select + multiply - a third of the VALU instructions of the fused tails were activations.  The
// same bits for every finite x, +-0 and NaN (max(NaN, NaN) = NaN); -inf under ReLU gives -inf where the select form gave
// NaN (-inf * 0) - both are outside the range the split-fp16 guard admits.
// (v_max_f32 by hand: fmaxf() makes the compiler canonicalise both operands with an extra max each and unpack the
// neighbouring packed multiplies)
__device__ __forceinline__ float leaky_max(float x, float slope)
{
    const float sx = slope * x;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(sx));
    return r;
}
__device__ __forceinline__ f32x4 leaky_max4(const f32x4 &x, float slope)
{
    const f32x4 sx = x * slope;  // two v_pk_mul_f32
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) asm("v_max_f32 %0, %1, %2" : "=v"(r[j]) : "v"(x[j]), "v"(sx[j]));
    return r;
}

constexpr int kMaxSteps = 128;  // supersteps per conv (K <= 2048)
constexpr int kPadSteps = 4;    // dead supersteps appended for the three-stage prefetch (fetches reach S+4)

// C/D layout of the 16x16 MFMAs: lane (i16, g) holds column i16 (pixel) and rows 4g..4g+3 (output channels)
template <int MT, int NT, bool GUARD = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, const f32x4 (&acc)[MT][NT], int strip, int i16, int g,
                                              const f32x4 (&bvec)[NT], const f32x4 (&rvec)[NT])
{
    const float slope = act_slope(a.act);
    const bool use_tanh = a.act == OJF_ACT_TANH;
    const float oscale = a.dscale ? a.scale / *a.dscale : a.scale;
    float gmax = 0.0f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int og = n * 4 + g;  // output channel group
        const f32x4 b = bvec[n];
        const f32x4 ri = rvec[n];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = strip + m * 16 + i16;
            const f32x4 lin4 = GUARD ? fma4(acc[m][n], ri, b) : acc[m][n] + b;
            f32x4 v = lin4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lin = v[j];
                float r = lin > 0.0f ? lin : lin * slope;
                if (use_tanh) r = tanhf(lin);
                v[j] = (og * 4 + j < a.act_n ? r : lin) * oscale;
            }
            if (p >= a.npix) continue;
            if (a.out_rows) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (og * 4 + j < a.rows_n) a.out_rows[(size_t)p * a.rows_stride + og * 4 + j] = v[j];
            } else if (og < a.og_store) {
                f32x4 *dst = a.out + (size_t)(a.out_g0 + og) * a.npix + p;
                if (a.accum) v += *dst;
                *dst = a.out_split ? split_pack4(v) : v;
                if constexpr (GUARD) gmax = guard_max(gmax, lin4);  // pre-activation magnitude
            }
        }
    }
    if constexpr (GUARD)
        if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
}

// The common case of the split-fp16 launches - planar output, ReLU / LeakyReLU / no activation on every stored channel,
// unit scale, plain store - without the generic epilogue's per-value branches (row output, tanh, partial activation,
// scale pointers): the same operations on the same values in the same order, i.e. the same bits, in ~1/3 of the
// instructions (the grouped 19 -> 19 convolutions issue 9 VALU per MFMA and half of a wave's instructions were its
// prologue and epilogue: r03_final_inst_counters.txt).  The host picks it per launch (conv_lean_ok).
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue_lean(const ConvArgs &a, const f32x4 (&acc)[MT][NT], int strip, int i16, int g,
                                                   const f32x4 (&bvec)[NT], const f32x4 (&rvec)[NT])
{
    const float slope = act_slope(a.act);
    float gmax = 0.0f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int og = n * 4 + g;
        if (og >= a.og_store) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = strip + m * 16 + i16;
            const f32x4 lin4 = fma4(acc[m][n], rvec[n], bvec[n]);
            const f32x4 v = leaky_max4(lin4, slope);
            if (p < a.npix) {
                a.out[(size_t)(a.out_g0 + og) * a.npix + p] = a.out_split ? split_pack4(v) : v;
                gmax = guard_max(gmax, lin4);
            }
        }
    }
    if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
}

static inline bool conv_lean_ok(const ConvArgs &a)
{
    return !a.out_rows && a.act != OJF_ACT_TANH && (a.act == OJF_ACT_NONE || a.act_n >= 4 * a.og_store) && a.scale == 1.0f && !a.dscale && !a.accum;
}

// Per-(superstep, slot) source table: float4 index of the tap's plane origin relative to a.in (-1 = the zero
// float4 every activation buffer keeps in front of its planes) and the tap's (dy, dx) packed in one int.
__device__ __forceinline__ void build_tap_table(int2 *tab, const ConvArgs &a, int entries)
{
    for (int G = threadIdx.x; G < entries; G += 256) {
        const int t = G / a.c4, cg = G - t * a.c4;
        int off = -1, dy = -30000, dx = 0;
        if (t < a.taps) {
            dy = dx = 0;
            if (a.taps == 9) {
                const int ky = t / 3;
                dy = (ky - 1) * a.dil;
                dx = (t - 3 * ky - 1) * a.dil;
            }
            off = (a.in_g0 + cg) * a.npix + dy * a.w + dx;
        }
        tab[G] = int2{off, (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xffffu))};
    }
}

// split-fp16 kernel: byte offset of the tap's plane origin relative to a.in, and the tap's bit (0 = dead entry)
__device__ __forceinline__ void build_tap_table16(int2 *tab, const ConvArgs &a, int entries)
{
    for (int G = threadIdx.x; G < entries; G += 256) {
        const int t = fast_div(G, a.c4, a.c4_magic), cg = G - t * a.c4;
        int off = 0, bit = 0;
        if (t < a.taps) {
            int dy = 0, dx = 0;
            if (a.taps == 9) {
                const int ky = t / 3;
                dy = (ky - 1) * a.dil;
                dx = (t - 3 * ky - 1) * a.dil;
            }
            off = ((a.in_g0 + cg) * a.npix + dy * a.w + dx) * 16;
            bit = 1 << t;
        }
        tab[G] = int2{off, bit};
    }
}

// ABL is a profiling-only ablation mask (tests/microbench): 1 = no activation loads, 2 = no weight
// loads, 4 = no MFMA, 8 = no per-superstep index math.  Product launches always use ABL = 0.
struct ConvGroup {
    ConvArgs g[4];  // independent convolutions of identical tile shape run as one launch (blockIdx.y)
    int nblocks;    // XCD-banded launches: pixel blocks of the image (gridDim.x is that rounded up to 8); 0 = plain order
    // PERSISTENT split-fp16 launches (round 6; layers whose whole packed weights are ONE LDS chunk: the grouped 19 -> 19 3x3): a
    // block walks `band` = round_up(nblocks, 8) / 8 logical pixel blocks of ITS XCD's band in steps of gridDim.x / 8 and
    // copies the tap table and the 24 KB of weights ONCE - with one pixel block per launch block every 128 pixels (10 KB of
    // input) paid a 24-KB weight copy, a table build and three barriers, 2400 times per branch and launch.  0 = off.
    int band = 0;
};

// (xcd_band_block / banded_block_x: ojf_common.h)

// SKIP: skip supersteps whose every source pixel lies outside the image (worth it for dilation 9 / 27);
// without it the loop body is one basic block and the scheduler interleaves the next fetch's address
// math with the MFMAs.
template <int MT, int NT, int ABL = 0, bool SKIP = true>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvGroup grp)
{
    const ConvArgs &a = grp.g[blockIdx.y];
    // tap table built once per block; kPadSteps dead supersteps pad the end so that the three-stage
    // prefetch needs no tail handling
    __shared__ int2 tab[(kMaxSteps + kPadSteps) * 4];
    build_tap_table(tab, a, (a.nsteps + kPadSteps) * 4);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (blockIdx.x * 4 + wave) * (MT * 16);
    if (strip >= a.npix) return;
    constexpr int ot0 = 0;  // a wave covers every output-channel tile

    int py[MT], px[MT], plin[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = strip + m * 16 + i16;
        plin[m] = p;
        py[m] = p < a.npix ? p / a.w : -0x40000000;  // rows far outside the image for padding lanes
        px[m] = p - (p / a.w) * a.w;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f32x4 *wb[NT];  // packed weights carry kPadSteps zero supersteps at the end as well
#pragma unroll
    for (int n = 0; n < NT; ++n) wb[n] = a.wp + (size_t)(ot0 + n) * (a.nsteps + kPadSteps) * 64 + lane;
    // activations through a buffer resource: an invalid tap selects an out-of-range offset, which loads zeros - the
    // input needs no zero float4 in front of its planes (the training path feeds plain torch allocations)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.in), 0, (a.in_g0 + a.c4) * a.npix * 16, 0x00020000);

    // Operand fetch of superstep S.  Nothing consumes the loaded values before the MFMAs (invalid
    // taps select the INDEX of the zero float4 instead of masking data), so three stages stay in
    // flight behind counted waits.
    auto fetch = [&](f32x4(&xv)[MT], f32x4(&wv)[NT], int S, bool &live) {
        if constexpr (ABL & 8) {
            live = true;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (ABL & 1) xv[m] = f32x4{1.f, 2.f, 3.f, 4.f};
                else xv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)plin[m] * 16u, 0, 0));
            }
        } else {
            const int2 e = tab[S * 4 + g];
            const int dy = e.y >> 16, dx = (int)(short)(e.y & 0xffff);
            bool any_ok = false;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool ok = (unsigned)(py[m] + dy) < (unsigned)a.h && (unsigned)(px[m] + dx) < (unsigned)a.w;
                const unsigned boff = ok ? (unsigned)(e.x + plin[m]) * 16u : 0xffffffffu;
                if constexpr (ABL & 1) {
                    xv[m] = f32x4{1.f, 2.f, 3.f, 4.f};
                    asm volatile("" ::"v"(boff));
                } else {
                    xv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 0));
                }
                any_ok |= ok;
            }
            live = SKIP ? __any(any_ok) : true;  // dead superstep: every source pixel of the wave is outside the image
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if constexpr (ABL & 2) wv[n] = f32x4{1.f, 1.f, 1.f, 1.f};
            else wv[n] = wb[n][(size_t)S * 64];
        }
    };
    auto mac = [&](const f32x4(&xv)[MT], const f32x4(&wv)[NT]) {
        if constexpr (ABL & 4) {
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(xv[m][0]), "v"(xv[m][1]), "v"(xv[m][2]), "v"(xv[m][3]));
#pragma unroll
            for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(wv[n][0]), "v"(wv[n][1]), "v"(wv[n][2]), "v"(wv[n][3]));
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n][j], xv[m][j], acc[m][n], 0, 0, 0);
    };

    // three operand stages in flight: superstep S+2 is fetched while S computes
    f32x4 x0[MT], w0[NT], x1[MT], w1[NT], x2[MT], w2[NT];
    bool l0, l1, l2;
    fetch(x0, w0, 0, l0);
    fetch(x1, w1, 1, l1);
    for (int S = 0; S < a.nsteps; S += 3) {
        fetch(x2, w2, S + 2, l2);
        if (!SKIP || l0) mac(x0, w0);
        fetch(x0, w0, S + 3, l0);
        if (!SKIP || l1) mac(x1, w1);
        fetch(x1, w1, S + 4, l1);
        if (!SKIP || l2) mac(x2, w2);
    }

    f32x4 bvec[NT], rvec[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        bvec[n] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
        rvec[n] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    conv_epilogue<MT, NT>(a, acc, strip, i16, g, bvec, rvec);
}

// Wide fp32 launches (NT >= 6 output tiles: the training path's 1x1 layers on 114 .. 580 channels and their
// backward-data forms): the weight fragments of a superstep are 1 KB per output tile and every wave of the plain kernel
// fetches all of them for its 16 pixels - 4800 waves x 64 KB through the L1s per launch of the transposed final
// convolution, which ran at 16 % of the fp32-MFMA peak for it.  Here the four waves of a block share one copy of a
// three-superstep chunk in LDS (the split-fp16 kernel's scheme): same arithmetic, same order, same bits.
template <int NT>
__global__ __launch_bounds__(256) void conv_mfma_lds_kernel(const ConvGroup grp)
{
    constexpr int CS = 3;
    const ConvArgs &a = grp.g[blockIdx.y];
    __shared__ int2 tab[(kMaxSteps + kPadSteps) * 4];
    __shared__ f32x4 wl[CS * NT * 64];
    build_tap_table(tab, a, (a.nsteps + kPadSteps) * 4);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (blockIdx.x * 4 + wave) * 16;  // waves past the image still take part in the barriers
    const int p = strip + i16;
    const int py = p < a.npix ? p / a.w : -0x40000000;
    const int px = p - (p / a.w) * a.w;
    f32x4 acc[1][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.in), 0, (a.in_g0 + a.c4) * a.npix * 16, 0x00020000);
    auto fetch = [&](f32x4 &xv, int S) {
        const int2 e = tab[S * 4 + g];
        const int dy = e.y >> 16, dx = (int)(short)(e.y & 0xffff);
        const bool ok = (unsigned)(py + dy) < (unsigned)a.h && (unsigned)(px + dx) < (unsigned)a.w;
        xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(e.x + p) * 16u : 0xffffffffu, 0, 0));
    };
    auto mac = [&](const f32x4 &xv, int sl) {
        f32x4 wv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) wv[n] = wl[(sl * NT + n) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n][j], xv[j], acc[0][n], 0, 0, 0);
    };
    f32x4 x0, x1, x2;
    fetch(x0, 0);
    fetch(x1, 1);
    const int nsp = a.nsteps + kPadSteps;
    for (int S = 0; S < a.nsteps; S += 3) {
        if (S) __syncthreads();
        for (int i = threadIdx.x; i < CS * NT * 64; i += 256) {  // chunk [S, S + 3) of every tile ([tile][superstep][lane] in memory)
            const int n = i / (CS * 64), r = i - n * (CS * 64);
            wl[((r >> 6) * NT + n) * 64 + (r & 63)] = a.wp[((size_t)n * nsp + S) * 64 + r];
        }
        __syncthreads();
        fetch(x2, S + 2);
        mac(x0, 0);
        fetch(x0, S + 3);
        mac(x1, 1);
        fetch(x1, S + 4);
        mac(x2, 2);
    }
    f32x4 bvec[NT], rvec[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        bvec[n] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
        rvec[n] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    if (strip >= a.npix) return;
    conv_epilogue<1, NT>(a, acc, strip, i16, g, bvec, rvec);
}

// ------------------------------------------------------------------------------------------------
// Split-fp16 variant of the generic convolution (OJF_ARITH_F16X3).  Same C4-planar fp32 activations in
// and out; a superstep is one 32-wide K block = 8 (tap, 4-channel group) entries, lane group g owns
// entries 8S+2g and 8S+2g+1 (K slot j < 4: channel j of the first, j >= 4: channel j-4 of the second),
// converts its 8 fp32 values to fp16 halves in registers and issues 3 MFMAs per output tile.
// The packed weights ([S][n][hi|lo][lane] x 8 halfs) are staged through LDS in chunks of CS supersteps
// shared by the four waves of the block: with the 5x faster MFMA the per-wave L1 weight stream of the
// fp32 kernel would be the bound (measured: 35 -> 26 us on the grouped 19->19 3x3 launches).
// ------------------------------------------------------------------------------------------------
constexpr int kPad16 = 6;  // dead supersteps behind the weights and the tap table (chunk copies + prefetch)
constexpr int conv16_chunk(int nt) { return nt <= 2 ? 6 : 3; }  // supersteps per LDS chunk (<= 24 KB)

// ABL: profiling-only ablation mask (tests/microbench): 1 = no activation loads, 2 = no LDS weight reads,
// 4 = no MFMA, 8 = no fp16 split.  Product launches always use ABL = 0.
// INS: the input planes are split planes (every convolution of the launch: the host checks)
template <int MT, int NT, bool SKIP = true, int ABL = 0, bool LEAN = false, bool INS = false>
#ifndef OJF_CONV_LB
#define OJF_CONV_LB 1
#endif
// (the lean grouped 3x3 form keeps four waves per SIMD - 128 registers - now that its blocks are persistent: the pixel-block loop costs
// 12 registers, which without the bound dropped it to three; OJF_CONV_LB=0 builds the unbounded form for A/B runs)
__global__ __launch_bounds__(256, (OJF_CONV_LB && MT == 2 && NT == 2 && LEAN) ? 4 : 1) void conv_f16x3_kernel(const ConvGroup grp)
{
    constexpr int CS = conv16_chunk(NT);
    const ConvArgs &a = grp.g[blockIdx.y];
    const bool persist = grp.band != 0;  // (uniform)
    int sb = grp.nblocks ? xcd_band_block(blockIdx.x, gridDim.x) : blockIdx.x;
    int sb_end = sb + 1, sb_step = 1;
    if (persist) {  // block j of XCD x walks positions j, j + P, j + 2 P .. of the XCD's band [x band, (x + 1) band)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        sb = x * grp.band + j;
        sb_end = (x + 1) * grp.band;
        sb_step = (int)(gridDim.x >> 3);
    }
    if (grp.nblocks && sb >= grp.nblocks) return;  // padding block of a banded launch
    extern __shared__ int2 tab[];  // (nsteps + kPad16) * 8 entries, sized by the launch: LDS per block bounds the waves in flight
    __shared__ f32x4 wl[CS * NT * 128];
    build_tap_table16(tab, a, (a.nsteps + kPad16) * 8);
    if (persist) {  // the layer's weights are one chunk (the host checks): copied once for all of the block's pixel blocks
#pragma unroll
        for (int i = 0; i < CS * NT * 128 / 256; ++i) wl[i * 256 + threadIdx.x] = a.wp[i * 256 + threadIdx.x];
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    __syncthreads();  // tap table (and the persistent form's weights)
    for (; sb < sb_end && (!grp.nblocks || sb < grp.nblocks); sb += sb_step) {
    int strip = (sb * 4 + wave) * (MT * 16);  // waves past the image still take part in the barriers
    if (a.row_perm) {  // (the host sets it only when w is a multiple of MT * 16: a wave's pixel tiles lie in one row)
        const int pr = fast_div(strip, a.w, a.w_magic);
        if (pr < a.h) strip = perm_row(pr, a.row_perm, a.perm_q, a.perm_rem) * a.w + (strip - pr * a.w);
    }

    // vm: bit t set <=> tap t of this lane's pixel lies inside the image (bit 0 only for 1x1); p16: byte offset
    int p16[MT];
    unsigned vm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = strip + m * 16 + i16;
        p16[m] = p * 16;
        const int py = fast_div(p, a.w, a.w_magic), px = p - py * a.w;
        unsigned rm = 0, cm = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rm |= ((unsigned)(py + (k - 1) * a.dil) < (unsigned)a.h ? 1u : 0u) << k;
            cm |= ((unsigned)(px + (k - 1) * a.dil) < (unsigned)a.w ? 1u : 0u) << k;
        }
        const unsigned all9 = ((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 3 : 0u) | ((rm & 4u) ? cm << 6 : 0u);
        vm[m] = p < a.npix ? (a.taps == 9 ? all9 : 1u) : 0u;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            // MT = 1: the zeros are made opaque, so that every MFMA accumulates in place.  Left to itself clang folds them into
            // the first MFMA of the peeled first superstep (SrcC = 0) and, at NT >= 6, sends tile 0 through a scratch tuple
            // (a[0:3] = W_lo X_hi; a[0:3] += W_hi X_lo; a[8:11] = W_hi X_hi + a[0:3], next tile's first MFMA into a[0:3]
            // right behind): on MI355X that tile came out without its low-half products (2e-4 instead of 2e-6) or as NaN
            // (tests/test_net_gpu.py::test_conv2d_layer at 114 -> 95 and 19 -> 114 caught it when the wide launches moved to
            // MT = 1); the same back-to-back shapes occur in kernels that are right, so this is pinned by test, not by rule.
            if constexpr (MT == 1) asm volatile("" : "+v"(acc[m][n]));
        }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.in), 0, (a.in_g0 + a.c4) * a.npix * 16, 0x00020000);
    // bias and inverse row scale of the epilogue: for the narrow layers they are fetched here, so that their latency
    // hides behind the main loop (the wide ones cannot afford the registers: 268 VGPRs at NT = 8)
    constexpr bool kHoist = NT <= 4;
    f32x4 bvec[NT], rvec[NT];
    if constexpr (kHoist) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bvec[n] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
            rvec[n] = *reinterpret_cast<const f32x4 *>(a.rinv + (size_t)n * 16 + 4 * g);
        }
    }

    // Operand fetch of superstep S: two (tap, channel group) entries per lane group.  The 9 taps' in-image tests
    // are one bit each in vm[] (computed once per wave), so an entry costs and + compare + add + select; buffer
    // loads return zeros for the out-of-range offset of an invalid tap.
    const int4 *tab4 = reinterpret_cast<const int4 *>(tab) + g;  // lane group g's two entries of a superstep
    auto fetch = [&](f32x4(&xa)[MT], f32x4(&xb)[MT], const int4 e, bool &live) {
        unsigned any = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned o0 = (unsigned)(e.x + p16[m]) | ((vm[m] & (unsigned)e.y) ? 0u : 0xffffffffu);
            const unsigned o1 = (unsigned)(e.z + p16[m]) | ((vm[m] & (unsigned)e.w) ? 0u : 0xffffffffu);
            if constexpr (ABL & 1) {
                xa[m] = f32x4{1.f, 2.f, 3.f, 4.f};
                xb[m] = f32x4{1.f, 2.f, 3.f, 4.f};
                asm volatile("" ::"v"(o0), "v"(o1));
            } else {
                xa[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o0, 0, 0));
                xb[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 0));
            }
            any |= vm[m] & (unsigned)(e.y | e.w);
        }
        live = SKIP ? __any(any != 0) : true;  // dead superstep: every source pixel of the wave is outside the image
    };
    auto mac = [&](const f32x4(&xa)[MT], const f32x4(&xb)[MT], int sl) {
        f32x4 wh[NT], wlo[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if constexpr (ABL & 2) {
                wh[n] = wlo[n] = f32x4{1.f, 1.f, 1.f, 1.f};
                asm volatile("" ::"s"(sl));
            } else {
                wh[n] = wl[(sl * NT + n) * 128 + lane];
                wlo[n] = wl[(sl * NT + n) * 128 + 64 + lane];
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f16x8 xh, xl;
            if constexpr (ABL & 8) {
                xh = __builtin_bit_cast(f16x8, xa[m]);
                xl = __builtin_bit_cast(f16x8, xb[m]);
            } else if constexpr (INS) {
                unpack_split(xa[m], xb[m], xh, xl);
            } else {
                split_f16(xa[m], xb[m], xh, xl);
            }
            if constexpr (ABL & 4) {
                asm volatile("" ::"v"(xh), "v"(xl));
#pragma unroll
                for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(wh[n]), "v"(wlo[n]));
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mfma_f16x3(wh[n], wlo[n], xh, xl, acc[m][n]);
            }
        }
    };

    f32x4 xa0[MT], xb0[MT], xa1[MT], xb1[MT], xa2[MT], xb2[MT];
    bool v0, v1, v2;
    fetch(xa0, xb0, tab4[0], v0);
    fetch(xa1, xb1, tab4[4], v1);
    int sl = persist ? 0 : CS;
    for (int S = 0; S < a.nsteps; S += 3) {
        // the three table reads of this iteration are issued together (one LDS latency instead of three)
        const int4 t2 = tab4[(S + 2) * 4], t3 = tab4[(S + 3) * 4], t4 = tab4[(S + 4) * 4];
        if (sl == CS) {  // next weight chunk (CS is a multiple of the 3 supersteps of one iteration)
            // plain copy: prefetching the chunk through registers one chunk ahead measured slower on the wide and
            // the grouped launches (17.9 -> 20.7 us, 23.1 -> 25.1 us: VGPRs) and equal elsewhere
            if (S) __syncthreads();
            const f32x4 *src = a.wp + (size_t)S * NT * 128;
#pragma unroll
            for (int i = 0; i < CS * NT * 128 / 256; ++i) wl[i * 256 + threadIdx.x] = src[i * 256 + threadIdx.x];
            __syncthreads();
            sl = 0;
        }
        fetch(xa2, xb2, t2, v2);
        if (!SKIP || v0) mac(xa0, xb0, sl);
        fetch(xa0, xb0, t3, v0);
        if (!SKIP || v1) mac(xa1, xb1, sl + 1);
        fetch(xa1, xb1, t4, v1);
        if (!SKIP || v2) mac(xa2, xb2, sl + 2);
        sl += 3;
    }
    if constexpr (!kHoist) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bvec[n] = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
            rvec[n] = *reinterpret_cast<const f32x4 *>(a.rinv + (size_t)n * 16 + 4 * g);
        }
    }
    if constexpr (LEAN) conv_epilogue_lean<MT, NT>(a, acc, strip, i16, g, bvec, rvec);
    else conv_epilogue<MT, NT, true>(a, acc, strip, i16, g, bvec, rvec);
    }  // (pixel blocks of a persistent block)
}

// ------------------------------------------------------------------------------------------------
// Fused 1x1 chain (the prediction head, model.py:24-52,252-256): 11 pointwise layers in ONE launch.
// With weights as the MFMA A operand, the accumulator of lane (pixel i, g) for output tile n holds
// channels 16n+4g..+3 of its pixel - exactly the B-operand fragment (channel group 4S+g, S = n) the
// next layer needs.  Activations therefore never leave registers between layers: each layer is
//   nxt[n2] = sum_S sum_j mfma(W_l[n2][S][j], act(cur[S])[j]),
// only the weights stream, and they are staged per layer (in <= 24 KB parts) through LDS so that the
// four waves of a block share one fetch.  Tile counts are compile-time (two supported topologies:
// 19- and 20-channel growth), so every register array is statically indexed.
// ------------------------------------------------------------------------------------------------
constexpr int kChainLdsFloat4 = 48 * 1024 / 16;
constexpr int kChainDmaHalf = OJF_CHAIN_DMA_HALF;  // float4 per half of the split-fp16 weight double buffer

struct ChainArgs {
    const f32x4 *in;  // input planes, c4_in groups starting at in_g0
    const f32x4 *w;   // per layer: [n2 < NTOUT][K block][lane] fragments (see chain_unit), layers back to back
    const float *bias;  // per layer NTOUT*16 floats, back to back
    float *out_rows;  // [npix, rows_stride], channels < rows_n
    int in_g0, c4_in, npix, rows_stride, rows_n;
    float scale;
    int *ovf;  // split-fp16 range guard flag (see ConvArgs)
    // kChainEntry (branch-entry GEMM of a VortexPooling run on register-resident input): result planes
    f32x4 *out_planes;
    int out_g0, og_store, act_n;  // ReLU on channels < act_n (branch 0), the pooled branches stay linear
    struct ColSums *colsum;       // channel sums of the entry layer's INPUT (global-average branch), see block_colsum
    int in_split = 0;             // entry1x1_kernel: the input planes are split planes (the dense-growth buffer of dense_chain_kernel)
    int split_groups = 0;         // kChainEntry: channel groups < split_groups (branch 0, read by its 3x3 convolution only) are stored as split planes
};

// Weight fragments of one (output tile, K block) pair.  fp32: K block = one 16-channel input tile, 64 float4
// (lane (oc i16, g) holds W[oc][16S+4g .. +3]).  split-fp16: K block = two input tiles (32 channels), 128 float4 =
// hi | lo fragments of 8 halfs; K slot j < 4 is channel 4g+j of tile 2S', j >= 4 channel 4g+j-4 of tile 2S'+1 - i.e.
// exactly the two accumulator quads lane (pixel, g) holds, so activations still never leave registers.
constexpr int chain_kblocks(int arith, int ntin) { return arith == OJF_ARITH_F16X3 ? (ntin + 1) / 2 : ntin; }
constexpr int chain_unit(int arith) { return arith == OJF_ARITH_F16X3 ? 128 : 64; }
constexpr int chain_layer_size(int arith, int ntin, int ntout) { return ntout * chain_kblocks(arith, ntin) * chain_unit(arith); }
// How the weight parts reach LDS:
//   fp32:       global -> registers (`pre`, issued one part ahead) -> ds_write, one 48 KB buffer, two barriers per part
//   split-fp16: LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction) straight into the other half of a
//               2 x 24 KB double buffer while the current part computes: no staging registers (48 VGPRs less, which
//               is what lets more waves hide the LDS latency of the 5x faster MFMA phase), no ds_write pass, one
//               barrier per part.
constexpr bool chain_dma(int arith) { return arith == OJF_ARITH_F16X3; }
constexpr int chain_cap(int arith) { return chain_dma(arith) ? kChainDmaHalf : kChainLdsFloat4; }  // float4 per part
// output tiles per LDS part: as many whole tiles as fit; parts = the number of such groups
constexpr int chain_tiles_fit(int arith, int ntin)
{
    return chain_cap(arith) / (chain_kblocks(arith, ntin) * chain_unit(arith));
}
constexpr int chain_parts(int arith, int ntin, int ntout)
{
    return (ntout + chain_tiles_fit(arith, ntin) - 1) / chain_tiles_fit(arith, ntin);
}
constexpr int chain_per(int arith, int ntin, int ntout)
{   // balanced over the parts (a part never exceeds chain_tiles_fit)
    return (ntout + chain_parts(arith, ntin, ntout) - 1) / chain_parts(arith, ntin, ntout);
}
constexpr int chain_first_size(int arith, int ntin, int ntout)
{   // float4 count of a layer's first part
    return (chain_per(arith, ntin, ntout) < ntout ? chain_per(arith, ntin, ntout) : ntout) * chain_kblocks(arith, ntin) *
           chain_unit(arith);
}
// per-layer floats in the bias stream: bias[NTOUT*16], and for split-fp16 the inverse row scales behind it
constexpr int chain_bias_stride(int arith, int ntout) { return (arith == OJF_ARITH_F16X3 ? 2 : 1) * ntout * 16; }
constexpr int kChainAhead = OJF_CHAIN_AHEAD;  // weight fragments requested ahead of their MFMAs (split-fp16 chain layers)
constexpr int kChainWaves = OJF_CHAIN_WAVES, kChainThreads = 64 * kChainWaves;  // waves of a chain / tail / entry block
constexpr int kChainPre = kChainLdsFloat4 / kChainThreads;  // float4 registers per thread holding the prefetched next part

// LDS-DMA of `size` float4 (a multiple of 64) from src to the LDS address dst: wave w moves the 1 KB chunks
// w, w+4, ...; completion = this wave's vmcnt reaching 0, visibility to the block = the barrier after that
__device__ __forceinline__ void dma_part(const f32x4 *src, f32x4 *dst, int size, int wave, int lane)
{
#pragma unroll
    for (int c0 = 0; c0 < kChainDmaHalf / 64; c0 += kChainWaves) {
        const int c = c0 + wave;
        if (c * 64 < size)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + c * 64 + lane),
                                             (void __attribute__((address_space(3))) *)(dst + c * 64), 16, 0, 0);
    }
}

enum { kChainLeaky = 0, kChainRelu = 1, kChainAccumulate = 2, kChainLastRows = 3, kChainEntry = 4 };

// Epilogue vectors (bias | inverse row scale, chain_bias_stride floats per layer) travel one layer ahead of their use:
// threads < n fetch one float4 each of the NEXT epilogue-bearing layer while the current layer computes (vec_issue),
// write it into one half of a 2 x 1 KB LDS buffer at that layer's entry (vec_commit; the layer's first barrier publishes
// it), and the epilogue reads its four float4 per output tile from LDS.  Fetching them from global memory inside the
// epilogue put an L2 round trip at the end of every layer of every wave (measured with the loads ablated: tail + head
// 84 -> 73 us, tail + entry 64 -> 59 us).
constexpr int kVecF4 = 64;  // float4 per half: 8 output tiles x (4 bias + 4 scale)
struct VecStage {
    f32x4 *lds;  // [2][kVecF4]
    f32x4 reg;   // this thread's float4 of the staged layer (threads < its float4 count)
    int half;    // half the next commit writes
};
struct VecNext {  // where the next layer's vectors come from: na float4 at a, then nb float4 at b (either may be empty)
    const float *a = nullptr;
    int na = 0;
    const float *b = nullptr;
    int nb = 0;
};
__device__ __forceinline__ void vec_issue(VecStage &vs, const VecNext &nx)
{
    const int t = threadIdx.x;
    if (t < nx.na) vs.reg = reinterpret_cast<const f32x4 *>(nx.a)[t];
    else if (t < nx.na + nx.nb) vs.reg = reinterpret_cast<const f32x4 *>(nx.b)[t - nx.na];
}
// returns the base of the half now holding the vectors (valid for every thread after the next barrier)
__device__ __forceinline__ const f32x4 *vec_commit(VecStage &vs, int n)
{
    f32x4 *dst = vs.lds + vs.half * kVecF4;
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vs.reg;
    vs.half ^= 1;
    return dst;
}

// One pointwise layer on register-resident activations: out[n2] (+)= sum_K W[n2][K] * in[K], then (unless
// accumulating) bias + activation in place.  `in` and `out` are distinct, statically indexed register arrays
// (callers ping-pong two of them), so no staging copy exists.
// `pre` carries the weights of this layer's first part on entry (already fetched from HBM/L2 while the
// previous part computed) and the next layer's first part on exit: global latency never sits between two
// compute phases, only the two barriers around the LDS refill do.
template <int ARITH, int MT, int NTIN, int NTOUT, int MODE, int NEXT_FIRST, int NA, int NB>
__device__ __forceinline__ void chain_layer(const f32x4 (&in)[MT][NA], f32x4 (&out)[MT][NB], f32x4 *wlds, const f32x4 *wg,
                                            const ChainArgs &a, const int (&p)[MT], int lane,
                                            f32x4 (&pre)[kChainPre], int &buf, VecStage &vs, const VecNext nx = VecNext(),
                                            const f32x4 *next_src = nullptr, int commit_extra = 0)
{
    static_assert(NTIN <= NA && NTOUT <= NB, "register arrays too small for this layer");
    // this layer's epilogue vectors (issued one layer ago) go to LDS now; an accumulating layer may carry the vectors of
    // the code that follows it (commit_extra).  The first barrier below publishes them.
    const f32x4 *vec = vec_commit(vs, MODE != kChainAccumulate ? chain_bias_stride(ARITH, NTOUT) / 4 : commit_extra);
    constexpr int parts = chain_parts(ARITH, NTIN, NTOUT);
    constexpr int per = chain_per(ARITH, NTIN, NTOUT);
    constexpr int KB = chain_kblocks(ARITH, NTIN);
    constexpr int unit = chain_unit(ARITH);
    static_assert(per * KB * unit <= chain_cap(ARITH), "LDS part too large");
    // split-fp16: the fp16 halves of the layer input, once per layer
    f16x8 xh[MT][ARITH == OJF_ARITH_F16X3 ? KB : 1], xl[MT][ARITH == OJF_ARITH_F16X3 ? KB : 1];
    if constexpr (ARITH == OJF_ARITH_F16X3) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int S = 0; S < KB; ++S) {
                const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                split_f16(in[m][2 * S], 2 * S + 1 < NTIN ? in[m][2 * S + 1 < NTIN ? 2 * S + 1 : 0] : zero, xh[m][S], xl[m][S]);
            }
    }
#pragma unroll
    for (int part = 0; part < parts; ++part) {
        const int nb = part * per;
        const int ne = nb + per < NTOUT ? nb + per : NTOUT;
        const int size = (ne - nb) * KB * unit;
        // the following part (next part of this layer, or the next layer's first)
        const int nnb = ne;
        const int nne = nnb + per < NTOUT ? nnb + per : NTOUT;
        const int nsize = part + 1 < parts ? (nne - nnb) * KB * unit : NEXT_FIRST;
        const f32x4 *nsrc = part + 1 < parts ? wg + (size_t)nnb * KB * unit
                                             : (next_src ? next_src : wg + (size_t)chain_layer_size(ARITH, NTIN, NTOUT));
        const f32x4 *wpart = wlds;
        if constexpr (chain_dma(ARITH)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's chunks of the current part have landed
            __syncthreads();  // ... everybody's have, and the readers of the other buffer are done with it
            wpart = wlds + buf * chain_cap(ARITH);
            buf ^= 1;
            dma_part(nsrc, wlds + buf * chain_cap(ARITH), nsize, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
            if (part == 0) vec_issue(vs, nx);
        } else {
            __syncthreads();  // readers of the previous part are done
#pragma unroll
            for (int k = 0; k < kChainPre; ++k)
                if ((int)threadIdx.x + kChainThreads * k < size) wlds[threadIdx.x + kChainThreads * k] = pre[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kChainPre; ++k)
                if ((int)threadIdx.x + kChainThreads * k < nsize) pre[k] = nsrc[threadIdx.x + kChainThreads * k];
            if (part == 0) vec_issue(vs, nx);
        }
        if constexpr (MODE != kChainAccumulate) {
#pragma unroll
            for (int n2 = 0; n2 < NTOUT; ++n2)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (n2 >= nb && n2 < ne) out[m][n2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // K block outer, output tiles inner: (ne - nb) * MT independent accumulators per MFMA round
        if constexpr (ARITH == OJF_ARITH_F16X3) {
            // Software pipeline over the part's (K block, output tile) steps: the weight fragments of step t + kChainAhead
            // are requested from LDS before the MFMAs of step t are issued.  Left alone, the compiler sinks every
            // ds_read next to its use (read -> s_waitcnt lgkmcnt(0) -> 3 MFMAs): ~100 cycles of LDS latency per 48
            // cycles of matrix work, the "parked" half of this kernel's wave-cycles.
            const int cnt = ne - nb, steps = KB * cnt;
            f32x4 rh[kChainAhead], rl[kChainAhead];
            auto frag = [&](int t) { return ((t % cnt) * KB + t / cnt) * 128; };  // step t = (S = t / cnt, tile nb + t % cnt)
#pragma unroll
            for (int t = 0; t < kChainAhead; ++t)
                if (t < steps) {
                    rh[t] = wpart[frag(t) + lane];
                    rl[t] = wpart[frag(t) + 64 + lane];
                }
#pragma unroll
            for (int t = 0; t < KB * NTOUT; ++t) {
                if (t >= steps) break;
                const f32x4 wh = rh[t % kChainAhead], wl = rl[t % kChainAhead];
                if (t + kChainAhead < steps) {
                    rh[t % kChainAhead] = wpart[frag(t + kChainAhead) + lane];
                    rl[t % kChainAhead] = wpart[frag(t + kChainAhead) + 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int S = t / cnt, n2 = nb + t % cnt;
#pragma unroll
                for (int m = 0; m < MT; ++m) out[m][n2] = mfma_f16x3(wh, wl, xh[m][S], xl[m][S], out[m][n2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int S = 0; S < KB; ++S)
#pragma unroll
            for (int n2 = 0; n2 < NTOUT; ++n2) {
                if (n2 < nb || n2 >= ne) continue;
                {
                    const f32x4 wv = wpart[((n2 - nb) * KB + S) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            out[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], in[m][S][j], out[m][n2], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (MODE != kChainAccumulate) {
        const int g = lane >> 4;
        float gmax = 0.0f;
#pragma unroll
        for (int n2 = 0; n2 < NTOUT; ++n2) {
            const f32x4 b = vec[n2 * 4 + g];
            f32x4 ri{1.f, 1.f, 1.f, 1.f};
            if constexpr (ARITH == OJF_ARITH_F16X3) ri = vec[(NTOUT + n2) * 4 + g];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 lin4 = ARITH == OJF_ARITH_F16X3 ? fma4(out[m][n2], ri, b) : out[m][n2] + b;
                f32x4 v = lin4;
                if constexpr (MODE == kChainLastRows) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oc = n2 * 16 + 4 * g + j;
                        if (oc < a.rows_n && p[m] < a.npix)
                            a.out_rows[(size_t)p[m] * a.rows_stride + oc] = tanhf(v[j]) * a.scale;
                    }
                } else if constexpr (MODE == kChainEntry) {
                    const int og = n2 * 4 + g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (og * 4 + j < a.act_n && !(v[j] > 0.0f)) ? (v[j] != v[j] ? v[j] : 0.0f) : v[j];
                    if (p[m] < a.npix && og < a.og_store) {
                        a.out_planes[(size_t)(a.out_g0 + og) * a.npix + p[m]] = og < a.split_groups ? split_pack4(v) : v;
                        if constexpr (ARITH == OJF_ARITH_F16X3) gmax = guard_max(gmax, lin4);
                    }
                } else {
                    constexpr float slope = MODE == kChainRelu ? 0.0f : 0.01f;
                    v = leaky_max4(v, slope);
                    if constexpr (ARITH == OJF_ARITH_F16X3) gmax = guard_max(gmax, lin4);  // pre-activation magnitude
                    out[m][n2] = v;
                }
            }
        }
        if constexpr (ARITH == OJF_ARITH_F16X3)
            if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
    }
}

// layer l reads `x` and writes `y`; the recursion swaps the two arrays for layer l+1
template <int ARITH, int MT, int NTIN, int NTOUT>
__device__ __forceinline__ void chain_run(f32x4 (&x)[MT][8], f32x4 (&y)[MT][8], f32x4 *wlds, const f32x4 *wg,
                                          const float *bias, const ChainArgs &a, const int (&p)[MT], int lane,
                                          f32x4 (&pre)[kChainPre], int &buf, VecStage &vs)
{
    chain_layer<ARITH, MT, NTIN, NTOUT, kChainLastRows, 0>(x, y, wlds, wg, a, p, lane, pre, buf, vs);
}

template <int ARITH, int MT, int NTIN, int NTOUT, int NTNEXT, int... REST>
__device__ __forceinline__ void chain_run(f32x4 (&x)[MT][8], f32x4 (&y)[MT][8], f32x4 *wlds, const f32x4 *wg,
                                          const float *bias, const ChainArgs &a, const int (&p)[MT], int lane,
                                          f32x4 (&pre)[kChainPre], int &buf, VecStage &vs)
{
    // `bias` = this layer's vectors (already issued by the caller / the previous layer); the next layer's follow them
    VecNext nx;
    nx.a = bias + chain_bias_stride(ARITH, NTOUT);
    nx.na = chain_bias_stride(ARITH, NTNEXT) / 4;
    chain_layer<ARITH, MT, NTIN, NTOUT, kChainLeaky, chain_first_size(ARITH, NTOUT, NTNEXT)>(x, y, wlds, wg, a, p, lane, pre,
                                                                                             buf, vs, nx);
    chain_run<ARITH, MT, NTOUT, NTNEXT, REST...>(y, x, wlds, wg + (size_t)chain_layer_size(ARITH, NTIN, NTOUT),
                                                 bias + chain_bias_stride(ARITH, NTOUT), a, p, lane, pre, buf, vs);
}

template <int ARITH, int MT, int NT0, int... NTS>
__global__ __launch_bounds__(kChainThreads, OJF_CHAIN_BLOCKS) void chain1x1_kernel(const ChainArgs a)
{
    __shared__ f32x4 wlds[chain_dma(ARITH) ? 2 * kChainDmaHalf : kChainLdsFloat4];
    __shared__ f32x4 vec_lds[2 * kVecF4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (banded_block_x() * kChainWaves + wave) * (MT * 16);
    int p[MT];
    f32x4 x[MT][8], y[MT][8];
    VecStage vs;
    vs.lds = vec_lds; vs.half = 0;
    {
        constexpr int first[] = {NTS...};
        VecNext nx;
        nx.a = a.bias; nx.na = chain_bias_stride(ARITH, first[0]) / 4;
        vec_issue(vs, nx);  // the first layer's epilogue vectors
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        p[m] = strip + m * 16 + i16;  // waves past the image still take part in the barriers
#pragma unroll
        for (int S = 0; S < NT0; ++S) {
            const int G = 4 * S + g;
            const bool ok = p[m] < a.npix && G < a.c4_in;
            x[m][S] = a.in[ok ? (a.in_g0 + G) * a.npix + p[m] : -1];
        }
    }
    f32x4 pre[kChainPre];
    int buf = 0;
    {   // first part of the first layer
        constexpr int first[] = {NTS...};
        constexpr int size0 = chain_first_size(ARITH, NT0, first[0]);
        if constexpr (chain_dma(ARITH)) {
            dma_part(a.w, wlds, size0, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
        } else {
#pragma unroll
            for (int k = 0; k < kChainPre; ++k)
                if ((int)threadIdx.x + kChainThreads * k < size0) pre[k] = a.w[threadIdx.x + kChainThreads * k];
        }
    }
    chain_run<ARITH, MT, NT0, NTS...>(x, y, wlds, a.w, a.bias, a, p, lane, pre, buf, vs);
}

// Channel sums of a register-resident activation tile set (the global-average branch of the NEXT VortexPooling,
// model.py:107-112): x[m][S] of lane (pixel i16, g) holds channels 16S+4g..+3 of pixel p[m].  Within a block the sum
// is formed in fp32 in a fixed order (butterfly over the 16 pixels of a tile, tiles, then the four waves through LDS);
// across blocks it is accumulated as 2^-20 fixed point in int64 (associative: the result does not depend on the
// arrival order, like the integrate kernels' sums) with atomics spread over kColShards rows, so that the block which
// folds the branch into the final convolution's bias (gave_bias_block) reads 16 x 128 words instead of one row per
// block (1200 rows took one block 25 us).  A non-finite block sum sets the channel's flag instead: its mean is NaN,
// as the reference's would be.
constexpr int kColShards = 16;
constexpr float kColScale = 1048576.0f;  // 2^20; |block sum| <= 64 * 65504 -> 4.4e12, 1200 blocks: far inside int64

constexpr int kColMax = 256;  // channels (round 6: the 228-channel entry of the two-head net's last VortexPooling)
struct ColSums {
    long long fix[kColShards][kColMax];
    unsigned bad[kColMax];
};
constexpr int colsum_row(int nt) { return nt > 8 ? 256 : 128; }  // floats per wave row of block_colsum's LDS scratch

template <int MT, int NT, int NA>
__device__ __forceinline__ void block_colsum(const f32x4 (&x)[MT][NA], const int (&p)[MT], int npix, float *red /* LDS [waves][colsum_row(NT)] */,
                                             ColSums *out, int lane, int wave, int off = 0)
{
    constexpr int RS = colsum_row(NT);
    const int g = lane >> 4;
#pragma unroll
    for (int S = 0; S < NT; ++S) {
        f32x4 s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (p[m] < npix) s += x[m][S];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = s[j];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 1, 64);
            s[j] = v;
        }
        if ((lane & 15) == 0) *reinterpret_cast<f32x4 *>(red + wave * RS + S * 16 + 4 * g) = s;
    }
    __syncthreads();
    static_assert(NT * 16 <= kChainThreads && NT * 16 <= kColMax, "one thread per channel");
    if ((int)threadIdx.x < NT * 16) {
        const int t = threadIdx.x;
        float v = (red[t] + red[RS + t]) + (red[2 * RS + t] + red[3 * RS + t]);
#pragma unroll
        for (int w4 = 4; w4 < kChainWaves; w4 += 4)  // further waves of the block, four at a time in the same fixed order
            v += (red[w4 * RS + t] + red[(w4 + 1) * RS + t]) + (red[(w4 + 2) * RS + t] + red[(w4 + 3) * RS + t]);
        if (v - v == 0.0f) {  // finite
            const long long q = (long long)__builtin_rintf(v * kColScale);  // |v| < 2^23: the product is exact up to the rounding of rintf
            if (q) atomicAdd(reinterpret_cast<unsigned long long *>(&out->fix[blockIdx.x % kColShards][t + off]), (unsigned long long)q);
        } else {
            atomicOr(&out->bad[t + off], 1u);
        }
    }
}

// Branch-entry GEMM of a VortexPooling (the four branches' first 1x1 convolutions stacked: c_in -> 4 slots of cs
// channels; branch 0 gets bias + ReLU, the pooled branches stay linear) as ONE chain layer on register-resident
// input, plus the per-block channel sums of that input.  Stand-alone form: input = activation planes.  The generic
// convolution kernel needed 25 us for this launch (tap table, masks, 6 output tiles per wave) and a separate
// column-sum launch re-read the same 37 MB.
template <int ARITH, int MT, int NTIN, int NTOUT>
__global__ __launch_bounds__(kChainThreads, OJF_CHAIN_BLOCKS) void entry1x1_kernel(const ChainArgs a)
{
    __shared__ f32x4 wlds[chain_dma(ARITH) ? 2 * kChainDmaHalf : kChainLdsFloat4];
    __shared__ f32x4 vec_lds[2 * kVecF4];
    __shared__ float red[kChainWaves * colsum_row(NTIN)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (banded_block_x() * kChainWaves + wave) * (MT * 16);
    f32x4 pre[kChainPre];
    int buf = 0;
    VecStage vs;
    vs.lds = vec_lds; vs.half = 0;
    {
        VecNext nx;
        nx.a = a.bias; nx.na = chain_bias_stride(ARITH, NTOUT) / 4;
        vec_issue(vs, nx);
    }
    {
        constexpr int size0 = chain_first_size(ARITH, NTIN, NTOUT);
        if constexpr (chain_dma(ARITH)) {
            dma_part(a.w, wlds, size0, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
        } else {
#pragma unroll
            for (int k = 0; k < kChainPre; ++k)
                if ((int)threadIdx.x + kChainThreads * k < size0) pre[k] = a.w[threadIdx.x + kChainThreads * k];
        }
    }
    int p[MT];
    f32x4 x[MT][NTIN], y[MT][NTOUT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        p[m] = strip + m * 16 + i16;  // waves past the image still take part in the barriers
#pragma unroll
        for (int S = 0; S < NTIN; ++S) {
            const int G = 4 * S + g;
            const bool ok = p[m] < a.npix && G < a.c4_in;
            x[m][S] = a.in[ok ? (a.in_g0 + G) * a.npix + p[m] : -1];
            if (a.in_split) x[m][S] = unsplit4(x[m][S]);  // hi + lo: exact in fp32, the value every consumer of the halves means
        }
    }
    if (a.colsum) block_colsum<MT, NTIN>(x, p, a.npix, red, a.colsum, lane, wave);
    chain_layer<ARITH, MT, NTIN, NTOUT, kChainEntry, 0>(x, y, wlds, a.w, a, p, lane, pre, buf, vs);
}

// ------------------------------------------------------------------------------------------------
// Fused VortexPooling tail (model.py:131-141,157-159): for each of the four branches
//   t_b = ReLU(W1_b v_b + b1_b)   (mid -> out channels, the branch's closing 1x1 + BN + ReLU)
//   y  += Wf_b t_b                (that branch's column block of the final 1x1 over the concat)
// followed by y + bias' (final conv bias + BN + the folded global-average branch).  The 4 x 116-channel
// concat tensor of the reference never exists: t_b lives in registers between the two GEMMs (same
// accumulator -> operand identity as the prediction-head chain) and y accumulates across branches.
// ------------------------------------------------------------------------------------------------
struct TailArgs {
    const f32x4 *v[4];  // branch inputs (planes, c4 groups each)
    const f32x4 *w;     // stream: W1_0 [NO][NV], Wf_0 [NO][NO], W1_1, Wf_1, ... (chain_layer fragments)
    const float *b1;    // 4 x chain_bias_stride: closing-1x1 biases (+ inverse row scales)
    const float *bias_final;  // NO*16 (per-frame: includes the global-average branch)
    const float *rinv_final;  // split-fp16: NO*16 inverse row scales of the final conv (common to the four blocks)
    f32x4 *out;
    int c4, out_g0, og_store, npix;
    int *ovf;  // split-fp16 range guard flag (see ConvArgs)
    // CHAIN variants (the last VortexPooling of the net): the prediction head runs on the register-resident result
    const f32x4 *chain_w;
    const float *chain_b;
    float *out_rows;
    int rows_stride, rows_n;
    float scale;
    // CHAIN = kTailEntry (a VortexPooling that feeds the next one): the next branch-entry GEMM runs on the register-
    // resident result, which is never written: planes of 4*cs channels + per-block channel sums come out instead
    const f32x4 *entry_w;
    const float *entry_b;
    f32x4 *entry_out;
    int entry_og, entry_act_n;
    struct ColSums *colsum;
    int entry_split;  // ChainArgs::split_groups of the fused entry layer
    int colsum_off = 0;  // two-head nets: this head's channels start here in the next VortexPooling's input (hd * os)
};

constexpr int kTailEntry = 1;

// the prediction-head topologies chain_run is instantiated for (growth channels 19 / 20)
template <int ARITH, int MT, int KIND>
__device__ __forceinline__ void head_run(f32x4 (&x)[MT][8], f32x4 (&y)[MT][8], f32x4 *wlds, const f32x4 *wg, const float *bias,
                                         const ChainArgs &a, const int (&p)[MT], int lane, f32x4 (&pre)[kChainPre], int &buf,
                                         VecStage &vs)
{
    if constexpr (KIND == 19) chain_run<ARITH, MT, 8, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1>(x, y, wlds, wg, bias, a, p, lane, pre, buf, vs);
    else chain_run<ARITH, MT, 8, 7, 7, 5, 5, 4, 4, 3, 3, 2, 2, 1>(x, y, wlds, wg, bias, a, p, lane, pre, buf, vs);
}
constexpr int head_first_ntout(int kind) { return kind == 19 ? 6 : 7; }
constexpr int head_first_size(int arith, int kind) { return chain_first_size(arith, 8, kind == 19 ? 6 : 7); }

// CHAIN = 0: write the VortexPooling result planes.  CHAIN = 19 / 20: feed it straight into the prediction head
// (same accumulator -> operand identity) and write est rows; the 114-channel tensor between them never exists.
template <int ARITH, int MT, int NV, int NO, int CHAIN = 0>
__global__ __launch_bounds__(kChainThreads, OJF_CHAIN_BLOCKS) void vortex_tail_kernel(const TailArgs a)
{
    static_assert(CHAIN == 0 || NO == 8, "the fused prediction head / entry layer expects 8 input tiles");
    __shared__ f32x4 wlds[chain_dma(ARITH) ? 2 * kChainDmaHalf : kChainLdsFloat4];
    __shared__ f32x4 vec_lds[2 * kVecF4];
    __shared__ float red[CHAIN == kTailEntry ? kChainWaves * 128 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (banded_block_x() * kChainWaves + wave) * (MT * 16);
    int p[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) p[m] = strip + m * 16 + i16;
    // epilogue vectors one layer ahead (VecStage): branch b's closing-1x1 vectors travel during branch b - 1, the final
    // bias / scale during the last branch, the fused entry layer's / head's first vectors during the last accumulation
    constexpr int kVec1 = chain_bias_stride(ARITH, NO) / 4, kVecFin = NO * 4 * (ARITH == OJF_ARITH_F16X3 ? 2 : 1);
    VecStage vs;
    vs.lds = vec_lds; vs.half = 0;
    {
        VecNext nx;
        nx.a = a.b1; nx.na = kVec1;
        vec_issue(vs, nx);
    }
    ChainArgs ca;  // only npix is read by the layer code in these modes
    ca.npix = a.npix; ca.out_rows = nullptr; ca.rows_n = 0; ca.rows_stride = 0; ca.scale = 1.0f; ca.ovf = a.ovf;
    f32x4 y[MT][NO];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NO; ++n) y[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pre[kChainPre];
    int buf = 0;
    {
        constexpr int size0 = chain_first_size(ARITH, NV, NO);
        if constexpr (chain_dma(ARITH)) {
            dma_part(a.w, wlds, size0, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
        } else {
#pragma unroll
            for (int k = 0; k < kChainPre; ++k)
                if ((int)threadIdx.x + kChainThreads * k < size0) pre[k] = a.w[threadIdx.x + kChainThreads * k];
        }
    }
    constexpr size_t per_branch = (size_t)chain_layer_size(ARITH, NV, NO) + chain_layer_size(ARITH, NO, NO);
    f32x4 t[MT][NO];
    const f32x4 *vfin = vec_lds;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x4 vin[MT][NV];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int S = 0; S < NV; ++S) {
                const int G = 4 * S + g;
                const bool ok = p[m] < a.npix && G < a.c4;
                vin[m][S] = a.v[b][ok ? G * a.npix + p[m] : -1];
            }
        const f32x4 *w1 = a.w + b * per_branch, *wf = w1 + (size_t)chain_layer_size(ARITH, NV, NO);
        VecNext nx;  // what travels during this branch's closing 1x1
        if (b < 3) {
            nx.a = a.b1 + (b + 1) * chain_bias_stride(ARITH, NO); nx.na = kVec1;
        } else {
            nx.a = a.bias_final; nx.na = NO * 4;
            if constexpr (ARITH == OJF_ARITH_F16X3) { nx.b = a.rinv_final; nx.nb = NO * 4; }
        }
        chain_layer<ARITH, MT, NV, NO, kChainRelu, chain_first_size(ARITH, NO, NO)>(vin, t, wlds, w1, ca, p, lane, pre, buf, vs, nx);
        if (b < 3) {
            chain_layer<ARITH, MT, NO, NO, kChainAccumulate, chain_first_size(ARITH, NV, NO)>(t, y, wlds, wf, ca, p, lane, pre, buf, vs);
        } else {
            vfin = vs.lds + vs.half * kVecF4;  // the half the last accumulation commits the final vectors to
            VecNext nf;
            if constexpr (CHAIN == kTailEntry) { nf.a = a.entry_b; nf.na = chain_bias_stride(ARITH, 5) / 4; }
            else if constexpr (CHAIN != 0) { nf.a = a.chain_b; nf.na = chain_bias_stride(ARITH, head_first_ntout(CHAIN)) / 4; }
            chain_layer<ARITH, MT, NO, NO, kChainAccumulate,
                        CHAIN == kTailEntry ? chain_first_size(ARITH, 8, 5) : (CHAIN ? head_first_size(ARITH, CHAIN) : 0)>(
                t, y, wlds, wf, ca, p, lane, pre, buf, vs, nf, CHAIN == kTailEntry ? a.entry_w : a.chain_w, kVecFin);
        }
    }
    float gmax = 0.0f;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
        const int og = n * 4 + g;
        const f32x4 bf = vfin[n * 4 + g];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 v = y[m][n] + bf;
            if constexpr (ARITH == OJF_ARITH_F16X3) v = fma4(y[m][n], vfin[NO * 4 + n * 4 + g], bf);
            if constexpr (ARITH == OJF_ARITH_F16X3) gmax = guard_max(gmax, v);
            if constexpr (CHAIN) y[m][n] = v;
            else if (p[m] < a.npix && og < a.og_store) a.out[(size_t)(a.out_g0 + og) * a.npix + p[m]] = v;
        }
    }
    if constexpr (ARITH == OJF_ARITH_F16X3)
        if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
    if constexpr (CHAIN == kTailEntry) {
        block_colsum<MT, NO>(y, p, a.npix, red, a.colsum, lane, wave, a.colsum_off);
        ca.out_planes = a.entry_out; ca.out_g0 = 0; ca.og_store = a.entry_og; ca.act_n = a.entry_act_n; ca.split_groups = a.entry_split;
        chain_layer<ARITH, MT, 8, 5, kChainEntry, 0>(y, t, wlds, a.entry_w, ca, p, lane, pre, buf, vs);
    } else if constexpr (CHAIN) {
        ca.out_rows = a.out_rows; ca.rows_stride = a.rows_stride; ca.rows_n = a.rows_n; ca.scale = a.scale;
        head_run<ARITH, MT, CHAIN>(y, t, wlds, a.chain_w, a.chain_b, ca, p, lane, pre, buf, vs);
    }
}

// Pool pyramid of a VortexPooling (model.py:143-155): branch b = 1..3 sees its entry-conv pre-activation pooled b
// times by nn.AvgPool2d(3, 1, 1) (zero padding, count_include_pad: always / 9), then bias + ReLU.  One launch:
// a block takes a 32x8-pixel tile of ONE channel group, stages it with a halo of b pixels in LDS and applies the b
// pooling levels there (every level is zero outside the image, like the padded intermediate tensors of the
// reference); the intermediate levels never reach HBM.  Summation order per level = row-major taps, as before.
// (used by gave_bias_block below; declared here because the pyramid launch carries one)
struct GaveArgs {
    const float *partial;   // legacy flow: nparts rows of pstride floats, added in row order
    struct ColSums *fixed;  // chain flow: fixed-point sums (read, then zeroed for the next frame)
    int nparts, pstride, cphys, npix;
    const float *WgT, *bg, *WfgT, *bf;  // WgT [cphys][c_out], WfgT [c_out(j)][c_out(o)]: lane o reads consecutive addresses
    int c_out;
    float *bias_out;
    int bias_len;
};

struct PyramidArgs {
    const f32x4 *z;        // entry-conv output planes; branch b's pre-activation = groups [b*c4, (b+1)*c4)
    f32x4 *q[3];           // branch inputs out: planes of c4 groups each
    const float *bias[3];  // per branch: c4*4 floats
    int h, w, c4;
    int tiles;      // pixel tiles; with fold_gave, block column `tiles` folds the global-average branch instead
    int fold_gave;
    // two-head nets: the entry GEMM arrives as two partial sums (one per head, no bias, no activation): z + z2 on load; block rows
    // [3 c4, 4 c4) form branch 0 = ReLU(z + z2 + bias0) (no pooling) into q0
    const f32x4 *z2 = nullptr;
    f32x4 *q0 = nullptr;
    const float *bias0 = nullptr;
    int split_out;  // q[] are split planes (split_pack4): their only reader is the branch's first 3x3 convolution
    GaveArgs gave;
};

constexpr int kPoolTW = 32, kPoolTH = 8, kPoolStride = kPoolTW + 6;

// LV = pooling levels of this block's channel group: region sizes are compile-time (the per-element index
// divisions become multiplies)
template <int LV>
__device__ __forceinline__ void pool_pyramid_body(const PyramidArgs &a, f32x4 (&buf)[2][kPoolStride * (kPoolTH + 6)], int cg)
{
    const int tiles_x = (a.w + kPoolTW - 1) / kPoolTW;
    const int tile = banded_block_x();
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * kPoolTW, y0 = ty * kPoolTH, npix = a.h * a.w;
    const f32x4 *plane = a.z + (size_t)(LV * a.c4 + cg) * npix;
    const f32x4 *plane2 = a.z2 ? a.z2 + (size_t)(LV * a.c4 + cg) * npix : nullptr;  // (uniform)
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    {
        constexpr int W0 = kPoolTW + 2 * LV, H0 = kPoolTH + 2 * LV;
#pragma unroll
        for (int i0 = 0; i0 < W0 * H0; i0 += 256) {
            const int i = i0 + threadIdx.x;
            if (i >= W0 * H0) break;
            const int ly = i / W0, lx = i - ly * W0;
            const int gy = y0 - LV + ly, gx = x0 - LV + lx;
            const bool in = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            f32x4 v = in ? plane[gy * a.w + gx] : zero;
            if (plane2 && in) v += plane2[gy * a.w + gx];
            buf[0][ly * kPoolStride + lx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int l = 1; l <= LV; ++l) {
        const int halo = LV - l;  // halo of this level's region; the source region has halo + 1
        const int Wl = kPoolTW + 2 * halo, Hl = kPoolTH + 2 * halo;
        const f32x4 *src = buf[(l - 1) & 1];
#pragma unroll
        for (int i0 = 0; i0 < Wl * Hl; i0 += 256) {
            const int i = i0 + threadIdx.x;
            if (i >= Wl * Hl) break;
            const int ly = i / Wl, lx = i - ly * Wl;
            const int gy = y0 - halo + ly, gx = x0 - halo + lx;
            const bool in = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            f32x4 s = zero;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) s += src[(ly + dy) * kPoolStride + lx + dx];
            s = s / 9.0f;
            if (l < LV) {
                buf[l & 1][ly * kPoolStride + lx] = in ? s : zero;
            } else if (in) {
                s += *reinterpret_cast<const f32x4 *>(a.bias[LV - 1] + 4 * cg);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] = s[j] < 0.0f ? 0.0f : s[j];  // keeps NaN, like torch.relu
                a.q[LV - 1][(size_t)cg * npix + gy * a.w + gx] = a.split_out ? split_pack4(s) : s;
            }
        }
        if (l < LV) __syncthreads();
    }
}

__device__ __forceinline__ void gave_bias_block(const GaveArgs &a, float *mean, float *gv);

__global__ __launch_bounds__(256) void pool_pyramid_kernel(const PyramidArgs a)
{
    __shared__ f32x4 buf[2][kPoolStride * (kPoolTH + 6)];
    if (banded_block_x() >= a.tiles) {  // one of the padding blocks: the global-average branch -> bias of the final conv (hidden behind the pools)
        if (blockIdx.y == 0 && banded_block_x() == a.tiles && a.fold_gave)
            gave_bias_block(a.gave, reinterpret_cast<float *>(buf), reinterpret_cast<float *>(buf) + 256);
        return;
    }
    // levels of this block's group.  The groups are dispatched in blockIdx.y order: the three-level blocks (three cascaded pools from one
    // window) first, the one-level blocks and the two-head nets' pool-free row last, so that the launch ends on its cheap blocks
    const int yg = blockIdx.y / a.c4, cg = blockIdx.y - yg * a.c4;
    const int lv = yg < 3 ? 3 - yg : 4;
    if (lv == 4) {  // two-head nets: branch 0 of the VortexPooling = ReLU(sum of the two heads' partial entry sums + bias), no pooling
        const int tiles_x = (a.w + kPoolTW - 1) / kPoolTW;
        const int tile = banded_block_x();
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int gy = ty * kPoolTH + (int)threadIdx.x / kPoolTW, gx = tx * kPoolTW + (int)threadIdx.x % kPoolTW;
        if (gy < a.h && gx < a.w) {
            const size_t at = (size_t)cg * a.h * a.w + gy * a.w + gx;
            f32x4 v = a.z[at] + a.z2[at] + *reinterpret_cast<const f32x4 *>(a.bias0 + 4 * cg);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.0f ? 0.0f : v[j];  // keeps NaN, like torch.relu
            a.q0[at] = a.split_out ? split_pack4(v) : v;
        }
        return;
    }
    if (lv == 1) pool_pyramid_body<1>(a, buf, cg);
    else if (lv == 2) pool_pyramid_body<2>(a, buf, cg);
    else pool_pyramid_body<3>(a, buf, cg);
}

constexpr int kSumBlocks = 32;

// per-channel partial sums (deterministic two-stage global average): block (b, cg) sums the float4s
// of plane cg over pixel strip b; fixed-order shuffle + LDS reduction
__global__ __launch_bounds__(256) void colsum_kernel(const f32x4 *in, int g0, int npix, float *partial)
{
    __shared__ f32x4 red[4];
    const int cg = blockIdx.y;
    const int per = (npix + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
    const f32x4 *plane = in + (size_t)(g0 + cg) * npix;
    f32x4 s{0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + threadIdx.x; p < p1; p += 256) s += plane[p];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        *reinterpret_cast<f32x4 *>(partial + (size_t)blockIdx.x * 256 + 4 * cg) = s;
    }
}

// gave_pool branch (model.py:107-112) folded into the bias of the final 1x1 conv:
//   mean -> 1x1 conv (+BN folded) -> g[c_out];  bias' = bias_final + W_final[:, gave columns] @ g
// `partial`: nparts rows of pstride floats (per-block channel sums), added in row order.

__device__ __forceinline__ void gave_bias_block(const GaveArgs &a, float *mean /* LDS [256] */, float *gv /* LDS [256] */)
{
    const int t = threadIdx.x;
    if (a.fixed) {
        if (t < kColMax) {
            long long q = 0;
#pragma unroll
            for (int sh = 0; sh < kColShards; ++sh) {
                q += a.fixed->fix[sh][t];
                a.fixed->fix[sh][t] = 0;  // this block is the only reader; the next writers come later in stream order
            }
            const bool bad = a.fixed->bad[t] != 0;
            a.fixed->bad[t] = 0;
            const float sum = (float)((double)q * (1.0 / (double)kColScale));
            mean[t] = t < a.cphys ? (bad ? __builtin_nanf("") : sum / (float)a.npix) : 0.0f;
        }
    } else if (a.cphys <= 128) {  // two threads per channel (rows split in halves) keep 256 loads in flight; fixed order
        const int ch = t & 127, half = t >> 7;
        float s = 0.0f;
        if (ch < a.cphys) {
            const int r0 = half ? (a.nparts + 1) / 2 : 0, r1 = half ? a.nparts : (a.nparts + 1) / 2;
#pragma unroll 8
            for (int b = r0; b < r1; ++b) s += a.partial[(size_t)b * a.pstride + ch];
        }
        gv[t] = s;
        __syncthreads();
        mean[t] = t < a.cphys ? (gv[t] + gv[128 + t]) / (float)a.npix : 0.0f;
    } else {
        float s = 0.0f;
        if (t < a.cphys) {
#pragma unroll 8
            for (int b = 0; b < a.nparts; ++b) s += a.partial[(size_t)b * a.pstride + t];
        }
        mean[t] = s / (float)a.npix;
    }
    __syncthreads();
    float s1 = 0.0f;
    if (t < a.c_out) {
        s1 = a.bg[t];
#pragma unroll 8
        for (int c = 0; c < a.cphys; ++c) s1 = __builtin_fmaf(a.WgT[(size_t)c * a.c_out + t], mean[c], s1);
    }
    __syncthreads();
    gv[t] = s1;
    __syncthreads();
    if (t < a.bias_len) {
        float s = 0.0f;
        if (t < a.c_out) {
            s = a.bf[t];
#pragma unroll 8
            for (int j = 0; j < a.c_out; ++j) s = __builtin_fmaf(a.WfgT[(size_t)j * a.c_out + t], gv[j], s);
        }
        a.bias_out[t] = s;
    }
}

__global__ __launch_bounds__(256) void gave_bias_kernel(const GaveArgs a)
{
    __shared__ float mean[256];
    __shared__ float gv[256];
    gave_bias_block(a, mean, gv);
}

struct PrepArgs {
    const float *values;   // [npix, rows_stride] fusion_values
    const float *weights;  // [npix, rows_stride] fusion_weights
    const float *depth;
    const uint8_t *sem;
    f32x4 *x0;
    f32x4 *x1;
    int rows_stride, in_layout, npix, P, cs4, n_classes, v2_sem;
    int *ovf;  // split-fp16 range guard flag (NULL for fp32 arithmetic)
    int split; // the slots are kept as split planes (dense_chain_kernel)
};

// modules/pipeline.py:74-102 _prepare_fusion_input: channels [values(P) | weights(P) | depth | (sem)]
// packed into the first dense-growth slot of each head; semantic_frame = (1 + id) / n_classes
__global__ __launch_bounds__(256) void prepare_input_kernel(const PrepArgs a)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npix) return;
    // rows: element (p, c) at [p*stride + c]; sample planes: [c*stride + p]
    const size_t pix = a.in_layout ? (size_t)p : (size_t)p * a.rows_stride;
    const size_t chn = a.in_layout ? (size_t)a.rows_stride : 1;
    const float *v = a.values + pix;
    const float *wt = a.weights + pix;
    const float d = a.depth[p];
    const float sf = a.sem ? (1.0f + (float)a.sem[p]) / (float)a.n_classes : 0.0f;
    bool bad = false;
    for (int cg = 0; cg < a.cs4; ++cg) {
        f32x4 r0, r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * cg + j;
            float base = 0.0f;
            if (c < a.P) base = v[c * chn];
            else if (c < 2 * a.P) base = wt[(c - a.P) * chn];
            r0[j] = base;
            r1[j] = base;
            if (c == 2 * a.P) { r0[j] = d; r1[j] = sf; }
            if (c == 2 * a.P + 1 && a.v2_sem) r0[j] = sf;
        }
        a.x0[(size_t)cg * a.npix + p] = a.split ? split_pack4(r0) : r0;
        if (a.x1) a.x1[(size_t)cg * a.npix + p] = a.split ? split_pack4(r1) : r1;
        bad = bad || beyond_f16(r0) || beyond_f16(r1);
    }
    if (bad && a.ovf) guard_raise(a.ovf, 1);
}

// rows <-> planes (ojf_conv2d test entry point only)
__global__ __launch_bounds__(256) void rows_to_planes_kernel(const float *rows, int stride, int off, int c, f32x4 *planes,
                                                              int c4, int npix)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= npix * c4) return;
    const int cg = item / npix, p = item - cg * npix;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (4 * cg + j < c) ? rows[(size_t)p * stride + off + 4 * cg + j] : 0.0f;
    planes[(size_t)cg * npix + p] = v;
}

__global__ __launch_bounds__(256) void planes_to_rows_kernel(const f32x4 *planes, int c4, int npix, float *rows, int stride,
                                                              int off)
{
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= npix * c4) return;
    const int cg = item / npix, p = item - cg * npix;
    const f32x4 v = planes[(size_t)cg * npix + p];
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[(size_t)p * stride + off + 4 * cg + j] = v[j];
}

}  // namespace ojf
#include "ojf_net_pair.h"
#include "ojf_net_chain.h"
namespace ojf {

// ------------------------------------------------------------------------------------------------
// host side: packing and the layer schedule
// ------------------------------------------------------------------------------------------------
struct PackedConv {
    float *wp = nullptr;    // device; layout depends on arith
    float *bias = nullptr;  // device, n_ot*16 floats
    float *rinv = nullptr;  // device, n_ot*16 floats (split-fp16 only)
    int c_in_phys = 0, c_out_phys = 0, taps = 1, dil = 1, n_ot = 0;
    int arith = OJF_ARITH_F32, nsteps = 0;  // supersteps: 4 (fp32) or 8 (split-fp16) K entries each
};

static int g_default_arith = OJF_ARITH_F16X3;

// Range guard of the split-fp16 arithmetic: one flag per process, in two places.  Kernels raise it (guard_raise,
// ojf_common.h) when a value that a later layer would split leaves the fp16 range (or is NaN): in a device-resident
// block {flag, skipped integrate calls, pointer to the mirror} - what ojf_integrate* test before they touch a volume, so
// that a frame whose net tripped the guard (and every frame after it, until the host has dealt with it) is NOT fused -
// and in a host-mapped mirror, which the host polls without synchronising (ojf_net_forward) or after a stream
// synchronise (ojf_net_check).  No traffic unless it fires.
static volatile int *g_ovf_host = nullptr;
static int *g_ovf_dev = nullptr;  // device block: [0] flag, [1] integrate calls skipped, [2..3] device address of the mirror

static int *overflow_flag()
{
    if (!g_ovf_dev) {
        void *h = nullptr, *hd = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, sizeof(int), hipHostMallocMapped) != hipSuccess) return nullptr;
        *static_cast<int *>(h) = 0;
        if (hipHostGetDevicePointer(&hd, h, 0) != hipSuccess) return nullptr;
        if (hipMalloc(&d, 16) != hipSuccess) return nullptr;
        struct { int flag, skipped; void *mirror; } init{0, 0, hd};
        static_assert(sizeof(init) == 16, "guard block layout");
        if (hipMemcpy(d, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        g_ovf_host = static_cast<volatile int *>(h);
        g_ovf_dev = static_cast<int *>(d);
    }
    return g_ovf_dev;
}

int *range_flag_device() { return overflow_flag(); }  // shared with ojf_seg.hip
const int *range_guard_if_any() { return g_ovf_dev; }  // ojf_integrate*.hip: NULL while no split-fp16 kernel ever got the flag

// Host side of a fired guard, after the stream was synchronised: the value (0 none, 1 range, 2 dense chain stuck), how many
// integrate calls were skipped because of it; `clear` re-arms both copies.
static int guard_take(int *skipped, bool clear)
{
    if (!g_ovf_host || !*g_ovf_host) {
        if (skipped) *skipped = 0;
        return 0;
    }
    const int what = *g_ovf_host;
    int blk[2] = {what, 0};
    (void)hipMemcpy(blk, g_ovf_dev, sizeof(blk), hipMemcpyDeviceToHost);
    if (skipped) *skipped = blk[1];
    if (clear) {
        (void)hipMemset(g_ovf_dev, 0, 2 * sizeof(int));
        (void)hipStreamSynchronize(nullptr);  // (non-blocking streams do not order against the null stream's fill: see alloc_planes)
        *g_ovf_host = 0;
    }
    return what;
}

// Launch log of a forward pass: every launch site calls mark_launch(name, stream) right after its launch.  It counts the
// launches (ojf_net_launch_count) and, in profile mode (ojf_net_profile), records a fence-free timing event behind the
// launch, so that the time between consecutive marks of one stream is that kernel's duration as the stream sees it.
struct LaunchMark { const char *name; hipStream_t st; hipEvent_t ev; };
static thread_local int g_launches = 0;
static thread_local bool g_profile = false;
static thread_local std::vector<LaunchMark> g_marks;

static inline void mark_launch(const char *name, hipStream_t st)
{
    ++g_launches;
    if (!g_profile) return;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_marks.push_back(LaunchMark{name, st, e});
}

// Fork / join events between streams of the same device: no timing, and (OJF_EVENT_FENCE=1 restores it) no system-scope
// fence - nothing here is read by the host or another device through these events.
static unsigned event_flags()
{
    static const bool fence = getenv("OJF_EVENT_FENCE") && atoi(getenv("OJF_EVENT_FENCE"));
    return fence ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}

static const char *kChainStuckMsg =
    "fusion net: dense_chain_kernel gave up waiting for a neighbouring tile (internal error: the results of this forward pass are invalid)";
static const char *kOverflowMsg =
    "split-fp16 arithmetic: an activation or input left the fp16 range (|x| > 65504 or NaN); results since the last "
    "ojf_net_check are invalid - use ojf_net_set_arithmetic(OJF_ARITH_F32) for this network";

// Split-fp16 weights are equilibrated per output channel: row oc is multiplied by the power of two that brings its
// largest magnitude to [2^13, 2^14) before it is split, and the accumulator is multiplied by the inverse in the
// epilogue (both exact).  BN folding scales whole rows by 1/sqrt(var) - without this, rows of ~1e-5 (large input
// channels such as the fp16 weight counts) would sit in the fp16 subnormals and lose their mantissa.
static inline float row_scale(float row_max)
{
    if (!(row_max > 0.0f) || !std::isfinite(row_max)) return 1.0f;
    int e = 0;
    (void)std::frexp(row_max, &e);  // row_max = m * 2^e, m in [0.5, 1)
    int k = 14 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return std::ldexp(1.0f, k);
}

// fp16 halves of a weight, both rounded to nearest (host side of split_f16)
static inline void split_weight(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

constexpr int kNT = 2;  // packed weights are padded to a multiple of this many output-channel tiles

struct ConvBuilder {
    int c_in_phys, c_out_phys, taps, dil;
    std::vector<float> W;  // [c_out_phys][taps][c_in_phys]
    std::vector<float> B;
    ConvBuilder(int cin_phys, int cout_phys, int ksize, int dilation)
        : c_in_phys(cin_phys), c_out_phys(cout_phys), taps(ksize * ksize), dil(dilation),
          W((size_t)cout_phys * ksize * ksize * cin_phys, 0.0f), B(cout_phys, 0.0f) {}
    // copy input columns [ci0, ci1) of `L` (all of its output rows) to output rows oc_off..,
    // input column ci -> physical channel in_map[ci - ci0]
    void add(const ojf_conv_layer &L, int ci0, int ci1, const std::vector<int> &in_map, int oc_off, bool with_bias)
    {
        const int kk = L.ksize * L.ksize;
        for (int o = 0; o < L.c_out; ++o) {
            for (int ci = ci0; ci < ci1; ++ci)
                for (int t = 0; t < kk; ++t)
                    W[((size_t)(oc_off + o) * taps + t) * c_in_phys + in_map[ci - ci0]] =
                        L.weight_host[((size_t)o * L.c_in + ci) * kk + t];
            if (with_bias) B[oc_off + o] = L.bias_host[o];
        }
    }
};

static int upload(const std::vector<float> &h, float **d)
{
    OJF_HIP(hipMalloc(reinterpret_cast<void **>(d), h.size() * sizeof(float)));
    OJF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static int finish(const ConvBuilder &b, PackedConv &pc, int arith = OJF_ARITH_F32)
{
    if (b.c_in_phys % 4) return fail("conv packing: c_in_phys must be a multiple of 4");
    pc.c_in_phys = b.c_in_phys;
    pc.c_out_phys = b.c_out_phys;
    pc.taps = b.taps;
    pc.dil = b.dil;
    pc.arith = arith;
    pc.n_ot = round_up(round_up(b.c_out_phys, 16) / 16, kNT);
    const int c4 = b.c_in_phys / 4, groups = b.taps * c4;
    const int per = arith == OJF_ARITH_F16X3 ? 8 : 4;  // K entries per superstep
    const int nsteps = (groups + per - 1) / per;
    if (nsteps > kMaxSteps) return fail("conv packing: K too large for the tap table");
    pc.nsteps = nsteps;
    std::vector<float> wp, bias((size_t)pc.n_ot * 16, 0.0f);
    if (arith == OJF_ARITH_F16X3) {
        std::vector<float> rs((size_t)pc.n_ot * 16, 1.0f), rinv((size_t)pc.n_ot * 16, 1.0f);
        for (int oc = 0; oc < b.c_out_phys; ++oc) {
            float mx = 0.0f;
            const float *row = b.W.data() + (size_t)oc * b.taps * b.c_in_phys;
            for (int i = 0; i < b.taps * b.c_in_phys; ++i) mx = std::fmax(mx, std::fabs(row[i]));
            rs[oc] = row_scale(mx);
            rinv[oc] = 1.0f / rs[oc];
        }
        if (upload(rinv, &pc.rinv)) return -2;
        // [S][ot][hi|lo][lane] x 8 halfs; dead (all-zero) supersteps behind for the chunk copies
        const int nsp = nsteps + kPad16;
        wp.assign((size_t)nsp * pc.n_ot * 128 * 4, 0.0f);
        _Float16 *hp = reinterpret_cast<_Float16 *>(wp.data());
        for (int S = 0; S < nsteps; ++S)
            for (int ot = 0; ot < pc.n_ot; ++ot)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = ot * 16 + (lane & 15), G = 8 * S + 2 * (lane >> 4) + (j >> 2);
                        if (oc >= b.c_out_phys || G >= groups) continue;
                        const int t = G / c4, cg = G % c4;
                        const float v = rs[oc] * b.W[((size_t)oc * b.taps + t) * b.c_in_phys + 4 * cg + (j & 3)];
                        const size_t base = ((size_t)S * pc.n_ot + ot) * 2 * 64 * 8;
                        split_weight(v, hp[base + (size_t)lane * 8 + j], hp[base + 64 * 8 + (size_t)lane * 8 + j]);
                    }
    } else {
        const int nsp = nsteps + kPadSteps;  // dead (all-zero) supersteps for the prefetch tail
        wp.assign((size_t)pc.n_ot * nsp * 256, 0.0f);
        for (int ot = 0; ot < pc.n_ot; ++ot)
            for (int S = 0; S < nsteps; ++S)
                for (int lane = 0; lane < 64; ++lane) {
                    const int oc = ot * 16 + (lane & 15), G = 4 * S + (lane >> 4);
                    if (oc >= b.c_out_phys || G >= groups) continue;
                    const int t = G / c4, cg = G % c4;
                    const float *row = b.W.data() + ((size_t)oc * b.taps + t) * b.c_in_phys + 4 * cg;
                    float *dst = wp.data() + (((size_t)ot * nsp + S) * 64 + lane) * 4;
                    for (int j = 0; j < 4; ++j) dst[j] = row[j];
                }
    }
    for (int o = 0; o < b.c_out_phys; ++o) bias[o] = b.B[o];
    if (upload(wp, &pc.wp)) return -2;
    if (upload(bias, &pc.bias)) return -2;
    return 0;
}

static void release(PackedConv &pc)
{
    if (pc.wp) (void)hipFree(pc.wp);
    if (pc.bias) (void)hipFree(pc.bias);
    if (pc.rinv) (void)hipFree(pc.rinv);
    pc.wp = pc.bias = pc.rinv = nullptr;
}

static inline f32x4 *planes(float *p) { return reinterpret_cast<f32x4 *>(p); }
static inline const f32x4 *planes(const float *p) { return reinterpret_cast<const f32x4 *>(p); }

// ---- fused 3x3 -> 3x3 pair (dense_pair_kernel, ojf_net_pair.h) ----------------------------------------------------
#ifdef OJF_PAIR_TIMING
static long long *g_pair_dbg = nullptr;
#endif
struct PackedPair {
    float *wa = nullptr, *wb = nullptr, *vec = nullptr;  // vec: bias_a | rinv_a | bias_b | rinv_b (32 floats each)
    int c4_in = 0, n_chunks = 0, np_last = 0, np_b = 0, og_store = 0;
    int cfg = 0;  // PairCfg the weights were packed for (chunk size)
};

// Launch shapes of dense_pair_kernel (PairGeom): picked per frame size, the chunk size is part of the weight packing
enum PairCfg { PAIR_20x16_W16 = 0, PAIR_12x8_W16 = 1, PAIR_20x8_W8 = 2, PAIR_16x8_W8 = 3, PAIR_20x8_W8_CP4 = 4,
               PAIR_20x16_W16_P5 = 5, PAIR_12x8_W16_P5 = 6, PAIR_CFG_COUNT = 7 };
static inline int pair_cfg_chunk_pairs(int cfg) { return cfg == PAIR_20x8_W8 || cfg == PAIR_16x8_W8 ? 3 : 4; }
static inline bool pair_cfg_pack(int cfg) { return cfg == PAIR_20x16_W16_P5 || cfg == PAIR_12x8_W16_P5; }
// c_out_phys: physical output channels of both convolutions (the packed-row shapes are written for 20 = 19 + padding)
static int pair_cfg_for(int h, int w, int c_out_phys)
{
    static const int forced = getenv("OJF_PAIR_CFG") ? atoi(getenv("OJF_PAIR_CFG")) : -1;  // tuning switch
    // one block per tile, 16 waves: the large tile when it still gives every CU a block (320x240: 240 blocks)
    const bool big = ((w + 19) / 20) * ((h + 15) / 16) >= 200;
    // (round 4 measured the other shapes - two 8-wave blocks per CU, packed rows - against this one: DESIGN.md §5.0, no gain)
    int cfg = big ? PAIR_20x16_W16 : PAIR_12x8_W16;
    if (forced >= 0 && forced < PAIR_CFG_COUNT && (!pair_cfg_pack(forced) || c_out_phys == 20)) cfg = forced;
    return cfg;
}

static void release(PackedPair &pp)
{
    float *ptrs[] = {pp.wa, pp.wb, pp.vec};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
    pp.wa = pp.wb = pp.vec = nullptr;
}

// K blocks of `np_chunk`-pair chunks: unit u = 4S + g -> (tap = u / np, pair = u % np), K slot j = channel 8*pair + j
// packed-row form (PairGeom PACK): [K block][3 row tiles][lane] x 8 halfs; packed row R = hi half of output row R for
// R < c_out_phys, lo half of output row R - c_out_phys for R < 2 c_out_phys, zero beyond
static void pack_pair_conv5(const ConvBuilder &b, const std::vector<float> &rs, std::vector<float> &dst, int cp)
{
    const int npairs = (b.c_in_phys + 7) / 8, n_chunks = (npairs + cp - 1) / cp, ocp = b.c_out_phys;
    for (int c = 0; c < n_chunks; ++c) {
        const int np = c == n_chunks - 1 ? npairs - cp * c : cp, nkb = (9 * np + 3) / 4;
        const size_t base = dst.size();
        dst.resize(base + (size_t)nkb * 192 * 4, 0.0f);
        _Float16 *hp = reinterpret_cast<_Float16 *>(dst.data() + base);
        for (int S = 0; S < nkb; ++S)
            for (int t = 0; t < 3; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int R = t * 16 + (lane & 15), u = 4 * S + (lane >> 4);
                        if (R >= 2 * ocp || u >= 9 * np) continue;
                        const int oc = R < ocp ? R : R - ocp;
                        const int tap = u / np, ch = 8 * (cp * c + u % np) + j;
                        if (ch >= b.c_in_phys) continue;
                        const float v = rs[oc] * b.W[((size_t)oc * b.taps + tap) * b.c_in_phys + ch];
                        _Float16 hi, lo;
                        split_weight(v, hi, lo);
                        hp[(((size_t)S * 3 + t) * 64 + lane) * 8 + j] = R < ocp ? hi : lo;
                    }
    }
}

static void pack_pair_conv(const ConvBuilder &b, const std::vector<float> &rs, std::vector<float> &dst, int cp)
{
    const int npairs = (b.c_in_phys + 7) / 8, n_chunks = (npairs + cp - 1) / cp;
    for (int c = 0; c < n_chunks; ++c) {
        const int np = c == n_chunks - 1 ? npairs - cp * c : cp, nkb = (9 * np + 3) / 4;
        const size_t base = dst.size();
        dst.resize(base + (size_t)nkb * 256 * 4, 0.0f);
        _Float16 *hp = reinterpret_cast<_Float16 *>(dst.data() + base);
        for (int S = 0; S < nkb; ++S)
            for (int nt = 0; nt < 2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = nt * 16 + (lane & 15), u = 4 * S + (lane >> 4);
                        if (oc >= b.c_out_phys || u >= 9 * np) continue;
                        const int tap = u / np, ch = 8 * (cp * c + u % np) + j;
                        if (ch >= b.c_in_phys) continue;
                        const float v = rs[oc] * b.W[((size_t)oc * b.taps + tap) * b.c_in_phys + ch];
                        const size_t ub = ((size_t)S * 2 + nt) * 2 * 64 * 8;
                        split_weight(v, hp[ub + (size_t)lane * 8 + j], hp[ub + 64 * 8 + (size_t)lane * 8 + j]);
                    }
    }
}

static int finish_pair(const ConvBuilder &ba, const ConvBuilder &bb, PackedPair &pp, int cfg = PAIR_20x16_W16)
{
    const int cp = pair_cfg_chunk_pairs(cfg);
    pp.cfg = cfg;
    if (ba.taps != 9 || bb.taps != 9 || ba.dil != 1 || bb.dil != 1 || ba.c_out_phys > 24 || bb.c_in_phys != ba.c_out_phys ||
        bb.c_out_phys > 32 || ba.c_in_phys % 4)
        return fail("pair packing: unsupported layer shapes");
    std::vector<float> vec(128, 0.0f), wa, wb;
    auto scales = [&](const ConvBuilder &b, int off) {
        std::vector<float> rs(32, 1.0f);
        for (int oc = 0; oc < b.c_out_phys; ++oc) {
            float mx = 0.0f;
            const float *row = b.W.data() + (size_t)oc * b.taps * b.c_in_phys;
            for (int i = 0; i < b.taps * b.c_in_phys; ++i) mx = std::fmax(mx, std::fabs(row[i]));
            rs[oc] = row_scale(mx);
            vec[off + oc] = b.B[oc];
        }
        for (int oc = 0; oc < 32; ++oc) vec[off + 32 + oc] = 1.0f / rs[oc];
        return rs;
    };
    const std::vector<float> ra = scales(ba, 0), rb = scales(bb, 64);
    if (pair_cfg_pack(cfg)) {
        if (ba.c_out_phys != 20 || bb.c_out_phys != 20) return fail("pair packing: the packed-row shapes need 20 physical output channels");
        pack_pair_conv5(ba, ra, wa, cp);
        pack_pair_conv5(bb, rb, wb, 4);
    } else {
        pack_pair_conv(ba, ra, wa, cp);
        pack_pair_conv(bb, rb, wb, 4);  // (3 pairs: one chunk whatever the chunk size)
    }
    pp.np_b = (bb.c_in_phys + 7) / 8;
    const int npairs = (ba.c_in_phys + 7) / 8;
    pp.c4_in = ba.c_in_phys / 4;
    pp.n_chunks = (npairs + cp - 1) / cp;
    pp.np_last = npairs - cp * (pp.n_chunks - 1);
    pp.og_store = round_up(bb.c_out_phys, 4) / 4;
    if (upload(wa, &pp.wa) || upload(wb, &pp.wb) || upload(vec, &pp.vec)) return -2;
    return 0;
}

template <int TW, int TH, int WAVES = 16, int CP = 4, bool ALIAS = false, bool PACK = false>
static int launch_pair_t(PairArgs &a, hipStream_t st)
{
    using G = PairGeom<TW, TH, WAVES, CP, ALIAS, PACK>;
    static bool configured = false;
    if (!configured) {
        OJF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_pair_kernel<TW, TH, WAVES, CP, ALIAS, PACK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        configured = true;
    }
    a.tiles_x = (a.w + TW - 1) / TW;
    const int tiles = a.tiles_x * ((a.h + TH - 1) / TH);
    hipLaunchKernelGGL((dense_pair_kernel<TW, TH, WAVES, CP, ALIAS, PACK>), dim3(tiles), dim3(G::THREADS), G::LDS_BYTES, st, a);
    mark_launch("dense_pair_kernel", st);
    return check_hip(hipGetLastError(), "dense_pair_kernel launch");
}

// in / out: plane buffers (may be the same allocation: the output groups are not part of the input window)
static int launch_pair(const PackedPair &pp, const float *in, int in_g0, float *out, int out_g0, int h, int w, hipStream_t st)
{
    PairArgs a;
    a.in = planes(in); a.out = planes(out);
    a.wa = planes(pp.wa); a.wb = planes(pp.wb);
    a.bias_a = pp.vec; a.rinv_a = pp.vec + 32; a.bias_b = pp.vec + 64; a.rinv_b = pp.vec + 96;
    a.in_g0 = in_g0; a.c4_in = pp.c4_in; a.out_g0 = out_g0; a.og_store = pp.og_store;
    a.h = h; a.w = w; a.npix = h * w; a.tiles_x = 0;
    a.n_chunks = pp.n_chunks; a.np_last = pp.np_last; a.np_b = pp.np_b;
    static const bool no_band = getenv("OJF_NO_XCD_BAND") != nullptr;  // tuning switch only
    a.xcd_bands = no_band ? 0 : 1;
    a.ovf = overflow_flag();
#ifdef OJF_PAIR_TIMING
    a.dbg = g_pair_dbg;
#endif
    switch (pp.cfg) {
    case PAIR_20x16_W16: return launch_pair_t<20, 16>(a, st);
    case PAIR_12x8_W16: return launch_pair_t<12, 8>(a, st);
    case PAIR_20x8_W8: return launch_pair_t<20, 8, 8, 3, true>(a, st);
    case PAIR_16x8_W8: return launch_pair_t<16, 8, 8, 3, true>(a, st);
    case PAIR_20x8_W8_CP4: return launch_pair_t<20, 8, 8, 4, true>(a, st);
    case PAIR_20x16_W16_P5: return launch_pair_t<20, 16, 16, 4, false, true>(a, st);
    case PAIR_12x8_W16_P5: return launch_pair_t<12, 8, 16, 4, false, true>(a, st);
    }
    return fail("dense pair: unknown launch shape");
}

// ---- all dense Blocks of a head in one persistent launch (dense_chain_kernel, ojf_net_chain.h) ---------------------
#ifdef OJF_CHAIN_TIMING
static long long *g_chain_dbg = nullptr;
#endif
struct PackedChain {
    float *w = nullptr, *vec = nullptr;
    int *sync = nullptr;  // ChainDenseArgs::sync
    int layers = 0;
};
constexpr int kChainSyncInts = kChainFlags0 + 8192;

static void release(PackedChain &pc)
{
    if (pc.w) (void)hipFree(pc.w);
    if (pc.vec) (void)hipFree(pc.vec);
    if (pc.sync) (void)hipFree(pc.sync);
    pc.w = pc.vec = nullptr; pc.sync = nullptr; pc.layers = 0;
}

// one step of dense_chain_kernel: input channels [ch0, ch0 + 20) of `b`, [K block][packed row tile][lane] x 8 halfs;
// packed row tiles (ojf_net_chain.h): 0 = hi halves of channels 0..15, 1 = their lo halves, 2 = hi | lo halves of channels
// 16..19 in rows 0..3 | 4..7; lane (row r16, lane group g), half j: position (R = 2 S + j / 4, g) -> (tap, channel group),
// input channel 4 q + j % 4
static void pack_chain_step(const ConvBuilder &b, const std::vector<float> &rs, int ch0, std::vector<float> &dst)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)kChainStepF4 * 4, 0.0f);
    _Float16 *hp = reinterpret_cast<_Float16 *>(dst.data() + base);
    for (int S = 0; S < kChainKB; ++S)
        for (int t = 0; t < 3; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r16 = lane & 15, g = lane >> 4, R = 2 * S + (j >> 2);
                    const int oc = t < 2 ? r16 : (r16 < 8 ? 16 + (r16 & 3) : -1);
                    const bool want_lo = t == 1 || (t == 2 && r16 >= 4);
                    const int tap = chain_hu_tap(R, g), ch = ch0 + 4 * chain_hu_q(R, g) + (j & 3);
                    if (oc < 0 || oc >= b.c_out_phys || tap < 0 || ch >= b.c_in_phys) continue;
                    const float v = rs[oc] * b.W[((size_t)oc * b.taps + tap) * b.c_in_phys + ch];
                    _Float16 hi, lo;
                    split_weight(v, hi, lo);
                    hp[(((size_t)S * 3 + t) * 64 + lane) * 8 + j] = want_lo ? lo : hi;
                }
}

// ba[l] / bb[l]: the two convolutions of Block l (ba[l].c_in_phys = 20 (l + 1), everything else 20 channels)
static int finish_chain(const std::vector<ConvBuilder> &ba, const std::vector<ConvBuilder> &bb, PackedChain &pc)
{
    const int L = (int)ba.size();
    if (L < 1 || L > kChainMaxLayers || (int)bb.size() != L) return fail("chain packing: unsupported Block count");
    std::vector<float> vec((size_t)L * 128, 0.0f), w;
    for (int l = 0; l < L; ++l) {
        const ConvBuilder &a = ba[l], &b = bb[l];
        if (a.taps != 9 || b.taps != 9 || a.dil != 1 || b.dil != 1 || a.c_out_phys != 4 * kChainNG || b.c_in_phys != 4 * kChainNG ||
            b.c_out_phys != 4 * kChainNG || a.c_in_phys != 4 * kChainNG * (l + 1))
            return fail("chain packing: unsupported layer shapes");
        auto scales = [&](const ConvBuilder &cb, int off) {
            std::vector<float> rs(32, 1.0f);
            for (int oc = 0; oc < cb.c_out_phys; ++oc) {
                float mx = 0.0f;
                const float *row = cb.W.data() + (size_t)oc * cb.taps * cb.c_in_phys;
                for (int i = 0; i < cb.taps * cb.c_in_phys; ++i) mx = std::fmax(mx, std::fabs(row[i]));
                rs[oc] = row_scale(mx);
                vec[(size_t)l * 128 + off + oc] = cb.B[oc];
            }
            for (int oc = 0; oc < 32; ++oc) vec[(size_t)l * 128 + off + 32 + oc] = 1.0f / rs[oc];
            return rs;
        };
        const std::vector<float> ra = scales(a, 0), rb = scales(b, 64);
        for (int c = 0; c <= l; ++c) pack_chain_step(a, ra, 4 * kChainNG * c, w);
        pack_chain_step(b, rb, 0, w);
    }
    pc.layers = L;
    if (upload(w, &pc.w) || upload(vec, &pc.vec)) return -2;
    OJF_HIP(hipMalloc(reinterpret_cast<void **>(&pc.sync), kChainSyncInts * sizeof(int)));
    OJF_HIP(hipMemset(pc.sync, 0, kChainSyncInts * sizeof(int)));
    OJF_HIP(hipStreamSynchronize(nullptr));  // (see alloc_planes)
    const int epoch0 = kChainEpoch;
    OJF_HIP(hipMemcpy(pc.sync, &epoch0, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

static int device_cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    }
    return cus;
}

template <int TW, int TH, int WAVES>
static int launch_chain_t(ChainDenseArgs &a, hipStream_t st)
{
    using G = ChainGeom<TW, TH, WAVES>;
    static bool configured = false;
    if (!configured) {
        OJF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_chain_kernel<TW, TH, WAVES>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        configured = true;
    }
    a.tiles_x = (a.w_img + TW - 1) / TW;
    a.tiles_y = (a.h + TH - 1) / TH;
    const int tiles = a.tiles_x * a.tiles_y, cus = device_cu_count();
    if (tiles > kChainSyncInts - kChainFlags0) return fail("dense chain: too many tiles for the flag array");
    // one block per CU (160 KB of LDS); the blocks draw (Block, tile) items until none is left
    const int grid = cus > 0 && cus < tiles ? cus : tiles;
    hipLaunchKernelGGL((dense_chain_kernel<TW, TH, WAVES>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, st, a);
    mark_launch("dense_chain_kernel", st);
    return check_hip(hipGetLastError(), "dense_chain_kernel launch");
}

// x: the dense-growth buffer as split planes (slot 0 filled; slots 1..layers are written)
static int launch_chain(const PackedChain &pc, float *x, int h, int w, hipStream_t st)
{
    ChainDenseArgs a;
    a.x = planes(x); a.xo = planes(x);
    a.w = planes(pc.w); a.vec = pc.vec;
    a.sync = pc.sync; a.layers = pc.layers;
    a.h = h; a.w_img = w; a.npix = h * w; a.tiles_x = a.tiles_y = 0;
    static const bool no_band = getenv("OJF_NO_XCD_BAND") != nullptr;  // tuning switch only
    a.xcd_bands = no_band ? 0 : 1;
    a.ovf = overflow_flag();
#ifdef OJF_CHAIN_TIMING
    a.dbg = g_chain_dbg;
#endif
    static const int waves = getenv("OJF_CHAIN_WAVES") ? atoi(getenv("OJF_CHAIN_WAVES")) : 16;  // tuning switch
    if (((w + 19) / 20) * ((h + 15) / 16) >= 200)  // (the large tile when it still gives every CU an item per Block)
        return waves == 16 ? launch_chain_t<20, 16, 16>(a, st) : (waves == 12 ? launch_chain_t<20, 16, 12>(a, st) : launch_chain_t<20, 16, 8>(a, st));
    return launch_chain_t<12, 8, 8>(a, st);
}

// ---- the four branches of a VortexPooling in one persistent launch (vortex_branch_kernel, ojf_net_branch.h) -----------
}  // namespace ojf
#include "ojf_net_branch.h"
namespace ojf {

struct PackedBranches {
    float *w = nullptr, *vec = nullptr;
    int4 *items = nullptr;
    int n_items = 0;
    int4 *sub_items = nullptr;  // subconv_kernel's table
    int n_sub = 0;
    int dil[4] = {0, 0, 0, 0};
};

static void release(PackedBranches &pb)
{
    if (pb.w) (void)hipFree(pb.w);
    if (pb.vec) (void)hipFree(pb.vec);
    if (pb.items) (void)hipFree(pb.items);
    if (pb.sub_items) (void)hipFree(pb.sub_items);
    pb = PackedBranches();
}

// The item table of vortex_branch_kernel for an h x w frame (ojf_net_branch.h): per branch the kind with the fewest MFMA
// pixel tiles, the table sorted by kind; inside a branch the phases of a tile 8 items apart (same XCD: measured 82 against
// 95 us per frame for the kernel with the sub-images of a tile dealt to neighbouring blocks, i.e. to eight L2s).
static std::vector<int4> branch_items(const int (&dil)[4], int h, int w)
{
    struct Plan { int br, kind, ntx, nty, cost; };
    std::vector<Plan> plans;
    for (int br = 0; br < 4; ++br) {
        const int d = dil[br], smw = (w + d - 1) / d, smh = (h + d - 1) / d;
        Plan p{br, 0, 0, 0, 0};
        if (smw <= BranchK2::TW && smh <= BranchK2::TH) {
            p.kind = 2; p.ntx = p.nty = 1;
        } else {
            const int ntx = (smw + 19) / 20, n0 = (smh + BranchK0::TH - 1) / BranchK0::TH, n1 = (smh + BranchK1::TH - 1) / BranchK1::TH;
            const int c0 = n0 * (BranchK0::TILES_A + BranchK0::TILES_B), c1 = n1 * (BranchK1::TILES_A + BranchK1::TILES_B);
            p.kind = c1 < c0 ? 1 : 0; p.ntx = ntx; p.nty = c1 < c0 ? n1 : n0;
        }
        plans.push_back(p);
    }
    std::vector<int4> items;
    for (int kind = 0; kind < 3; ++kind)
        for (const Plan &p : plans) {
            if (p.kind != kind) continue;
            const int d = dil[p.br];
            std::vector<int> phases;  // px | py << 16 of the non-empty sub-images
            for (int py = 0; py < d && py < h; ++py)
                for (int px = 0; px < d && px < w; ++px) phases.push_back(px | (py << 16));
            const int head = p.br | (kind << 4);
            if (kind == 2) {
                for (size_t i = 0; i < phases.size(); i += 2)
                    items.push_back(int4{head, phases[i], i + 1 < phases.size() ? phases[i + 1] : -1, 0});
                continue;
            }
            const int tw = 20, th = kind == 1 ? BranchK1::TH : BranchK0::TH, nt = p.ntx * p.nty;
            for (int t0 = 0; t0 < nt; t0 += 8) {
                const int n = nt - t0 < 8 ? nt - t0 : 8;
                for (int ph : phases)
                    for (int c = 0; c < n; ++c) {
                        // full groups: position c of an eight holds the tile whose number is congruent to the item's index mod 8
                        const int t = n == 8 ? t0 + (int)(items.size() & 7u) : t0 + c;
                        items.push_back(int4{head, ph, -1, ((t % p.ntx) * tw) | (((t / p.ntx) * th) << 16)});
                    }
            }
        }
    return items;
}

// The item table of subconv_kernel (one block per item): per branch the kind with the fewest MFMA pixel tiles; inside a branch
// the phases of a tile 8 items apart (blocks land on XCD blockIdx % 8: the r x r sub-images of one image region share an L2).
static std::vector<int4> subconv_items(const int (&dil)[4], int h, int w)
{
    struct Kind { int tw, th, tiles; };
    const Kind kinds[4] = {{SubK0::TW, SubK0::TH, SubK0::TILES}, {SubK1::TW, SubK1::TH, SubK1::TILES}, {SubK2::TW, SubK2::TH, SubK2::TILES},
                           {SubK3::TW, SubK3::TH, SubK3::TILES}};
    std::vector<int4> items;
    for (int br = 0; br < 4; ++br) {
        const int d = dil[br], smw = (w + d - 1) / d, smh = (h + d - 1) / d;
        int best = 0;
        long best_cost = -1;
        for (int k = 0; k < 4; ++k) {
            const long cost = (long)((smw + kinds[k].tw - 1) / kinds[k].tw) * ((smh + kinds[k].th - 1) / kinds[k].th) * kinds[k].tiles;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = k; }
        }
        const Kind &K = kinds[best];
        const int ntx = (smw + K.tw - 1) / K.tw, nty = (smh + K.th - 1) / K.th, nt = ntx * nty, head = br | (best << 4);
        std::vector<int> phases;
        for (int py = 0; py < d && py < h; ++py)
            for (int px = 0; px < d && px < w; ++px) phases.push_back(px | (py << 16));
        for (int t0 = 0; t0 < nt; t0 += 8) {
            const int n = nt - t0 < 8 ? nt - t0 : 8;
            for (int ph : phases)
                for (int c = 0; c < n; ++c) {
                    const int t = n == 8 ? t0 + (int)(items.size() & 7u) : t0 + c;
                    items.push_back(int4{head, ph, -1, ((t % ntx) * K.tw) | (((t / ntx) * K.th) << 16)});
                }
        }
    }
    return items;
}

// ba[br] / bb[br]: the two dilated 3x3 of branch br (20 -> 20 channels)
static int finish_branches(const std::vector<ConvBuilder> &ba, const std::vector<ConvBuilder> &bb, int h, int w, PackedBranches &pb)
{
    if (ba.size() != 4 || bb.size() != 4) return fail("branch packing: four branches expected");
    std::vector<float> vec(4 * 128, 0.0f), wts;
    for (int br = 0; br < 4; ++br) {
        const ConvBuilder &a = ba[br], &b = bb[br];
        if (a.taps != 9 || b.taps != 9 || a.dil != b.dil || a.dil < 1 || a.c_in_phys != 4 * kChainNG || a.c_out_phys != 4 * kChainNG ||
            b.c_in_phys != 4 * kChainNG || b.c_out_phys != 4 * kChainNG)
            return fail("branch packing: unsupported layer shapes");
        pb.dil[br] = a.dil;
        auto scales = [&](const ConvBuilder &cb, int off) {
            std::vector<float> rs(32, 1.0f);
            for (int oc = 0; oc < cb.c_out_phys; ++oc) {
                float mx = 0.0f;
                const float *row = cb.W.data() + (size_t)oc * cb.taps * cb.c_in_phys;
                for (int i = 0; i < cb.taps * cb.c_in_phys; ++i) mx = std::fmax(mx, std::fabs(row[i]));
                rs[oc] = row_scale(mx);
                vec[(size_t)br * 128 + off + oc] = cb.B[oc];
            }
            for (int oc = 0; oc < 32; ++oc) vec[(size_t)br * 128 + off + 32 + oc] = 1.0f / rs[oc];
            return rs;
        };
        const std::vector<float> ra = scales(a, 0), rb = scales(b, 64);
        pack_chain_step(a, ra, 0, wts);
        pack_chain_step(b, rb, 0, wts);
    }
    const std::vector<int4> items = branch_items(pb.dil, h, w);
    if (items.empty() || items.size() > (1u << 20)) return fail("branch packing: bad item count");
    pb.n_items = (int)items.size();
    if (upload(wts, &pb.w) || upload(vec, &pb.vec)) return -2;
    OJF_HIP(hipMalloc(reinterpret_cast<void **>(&pb.items), items.size() * sizeof(int4)));
    OJF_HIP(hipMemcpy(pb.items, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice));
    const std::vector<int4> sub = subconv_items(pb.dil, h, w);
    if (sub.empty() || sub.size() > (1u << 22)) return fail("branch packing: bad subconv item count");
    pb.n_sub = (int)sub.size();
    OJF_HIP(hipMalloc(reinterpret_cast<void **>(&pb.sub_items), sub.size() * sizeof(int4)));
    OJF_HIP(hipMemcpy(pb.sub_items, sub.data(), sub.size() * sizeof(int4), hipMemcpyHostToDevice));
    return 0;
}

// one dilated 3x3 of the four branches (second = 0 / 1): in[br] split planes, out[br] split planes or fp32 (5 groups each)
static int launch_subconv(const PackedBranches &pb, const float *const *in, float *const *out, int second, bool out_split, int h, int w,
                          hipStream_t st)
{
    static bool configured = false;
    if (!configured) {
        OJF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&subconv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSubLdsBytes));
        configured = true;
    }
    SubArgs a;
    for (int br = 0; br < 4; ++br) { a.in[br] = planes(in[br]); a.out[br] = planes(out[br]); a.dil[br] = pb.dil[br]; }
    a.w = planes(pb.w); a.vec = pb.vec; a.items = pb.sub_items;
    a.second = second; a.out_split = out_split ? 1 : 0;
    a.h = h; a.w_img = w; a.npix = h * w;
    a.ovf = overflow_flag();
    hipLaunchKernelGGL(subconv_kernel, dim3(pb.n_sub), dim3(256), kSubLdsBytes, st, a);
    mark_launch("subconv_kernel", st);
    return check_hip(hipGetLastError(), "subconv_kernel launch");
}

// in[br]: split planes of the branch inputs; out: fp32 planes, branch br at groups [5 br, 5 br + 5)
static int launch_branches(const PackedBranches &pb, const float *const *in, float *out, int h, int w, hipStream_t st)
{
    static bool configured = false;
    if (!configured) {
        OJF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&vortex_branch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kBranchLdsBytes));
        configured = true;
    }
    BranchArgs a;
    for (int br = 0; br < 4; ++br) { a.in[br] = planes(in[br]); a.dil[br] = pb.dil[br]; }
    a.out = planes(out);
    a.w = planes(pb.w); a.vec = pb.vec; a.items = pb.items; a.n_items = pb.n_items;
    a.h = h; a.w_img = w; a.npix = h * w;
    a.ovf = overflow_flag();
    const int cus = device_cu_count();
    const int grid = cus > 0 && cus < pb.n_items ? cus : pb.n_items;
    hipLaunchKernelGGL(vortex_branch_kernel, dim3(grid), dim3(64 * kBranchWaves), kBranchLdsBytes, st, a);
    mark_launch("vortex_branch_kernel", st);
    return check_hip(hipGetLastError(), "vortex_branch_kernel launch");
}

// Launches n (<= 4) independent convolutions with the same number of output tiles as ONE grid
// (blockIdx.y = problem): the four branches of a VortexPooling run together instead of as four
// under-filled launches with their own ramp-up and tail.
static int launch_conv_args(const ConvArgs *args, int n, int nt, hipStream_t st, int arith)
{
    ConvGroup grp;
    for (int i = 0; i < n; ++i) grp.g[i] = args[i];
    for (int i = n; i < 4; ++i) grp.g[i] = args[0];
    grp.nblocks = 0;
    // one wave computes ALL output-channel tiles of its pixel strip (activations are fetched once);
    // MT (16-pixel tiles per wave) trades operand reuse against the number of waves in flight.
    // fp32: MT = 1 everywhere: 4800 waves over 1024 SIMDs quantise to 5 rounds where MT = 2 (2400 waves) needs 3
    // rounds of twice the length (78 % vs 94 % balance); measured 14.9 vs 17.3 us on the 19->19 3x3 layers.
    // split-fp16: two pixel tiles per wave share the LDS weight reads and the per-wave prologue; measured in one run
    // (net stage): MT = 2 everywhere 0.501 ms, MT = 2 only for grouped / wide launches 0.508 ms, MT = 1 0.514 ms.
    static const int mt_env = getenv("OJF_CONV_MT") ? atoi(getenv("OJF_CONV_MT")) : 0;  // tuning switch only
    // The wide launches of the training executor (NT >= 6: 96 / 128 output channels) need 204-212 + 64 registers at MT = 2 -
    // one wave per SIMD, nothing to run while a block waits for its weight chunk; MT = 1 (140 + 32) keeps two: training
    // frame step 216 -> 220 and 204 -> 217 frames/s on two boxes (NT <= 4: no difference).
    const int mt = arith == OJF_ARITH_F16X3 ? (mt_env ? mt_env : (nt >= 6 ? 1 : 2)) : 1;
    const int strips = (args[0].npix + mt * 16 - 1) / (mt * 16);
    const dim3 grid((strips + 3) / 4, n), block(256);
#define OJF_LAUNCH32(NT_) hipLaunchKernelGGL((conv_mfma_kernel<1, NT_>), grid, block, 0, st, grp)
    int max_steps = 0;
    for (int i = 0; i < n; ++i) max_steps = args[i].nsteps > max_steps ? args[i].nsteps : max_steps;
    const size_t tab_bytes = (size_t)(max_steps + kPad16) * 8 * sizeof(int2);  // (no measurable effect vs the full 8.6 KB)
    // split-fp16 launches hand every XCD one band of the image (xcd_band_block)
    static const bool no_band = getenv("OJF_NO_XCD_BAND") != nullptr;  // tuning switch only
    dim3 grid16 = grid;
    if (arith == OJF_ARITH_F16X3 && !no_band && grid.x >= 64) {
        grp.nblocks = (int)grid.x;
        grid16.x = round_up((int)grid.x, 8);
    }
    static const bool no_perm = getenv("OJF_CONV_ROW_PERM") && atoi(getenv("OJF_CONV_ROW_PERM")) == 0;  // A/B switch
    if (grp.nblocks && !no_perm)
        for (int i = 0; i < n; ++i) {
            ConvArgs &g = grp.g[i];
            if (g.taps == 9 && g.dil > 1 && g.dil < g.h && g.w % (mt * 16) == 0) {
                g.row_perm = g.dil; g.perm_q = g.h / g.dil; g.perm_rem = g.h % g.dil;
            }
        }
    // persistent blocks for single-chunk layers (ConvGroup::band): `bpc` blocks per CU over all members of the group
    static const int persist_bpc = getenv("OJF_CONV_PERSIST") ? atoi(getenv("OJF_CONV_PERSIST")) : 3;  // A/B switch (0: one pixel block per launch block; measured 2 / 3 / 4 / 6: 3 is best, profiles/r06_fusion_net_experiments.txt)
    if (grp.nblocks && persist_bpc > 0 && arith == OJF_ARITH_F16X3) {
        bool single = true;
        for (int i = 0; i < n; ++i) single = single && args[i].nsteps <= conv16_chunk(nt);
        const int cus = device_cu_count();
        const int n8 = round_up(grp.nblocks, 8);
        int G = cus > 0 ? (cus * persist_bpc / n) / 8 * 8 : 0;
        if (single && G >= 8 && G < n8) {
            grp.band = n8 / 8;
            grid16.x = (unsigned)G;
        }
    }
    static const bool no_lean = getenv("OJF_CONV_LEAN") && atoi(getenv("OJF_CONV_LEAN")) == 0;  // A/B switch
    bool lean = arith == OJF_ARITH_F16X3 && !no_lean;
    for (int i = 0; i < n && lean; ++i) lean = conv_lean_ok(args[i]);
    int n_ins = 0, n_outs = 0;
    for (int i = 0; i < n; ++i) { n_ins += args[i].in_split ? 1 : 0; n_outs += args[i].out_split ? 1 : 0; }
    if ((n_ins || n_outs) && (arith != OJF_ARITH_F16X3 || (n_ins && (n_ins != n || mt != 2 || nt != 2))))
        return fail("conv: split planes are a format of the split-fp16 grouped 3x3 launches only");
#define OJF_LAUNCH16(MT_, NT_)                                                                                      \
    do {                                                                                                             \
        if (lean) hipLaunchKernelGGL((conv_f16x3_kernel<MT_, NT_, true, 0, true>), grid16, block, tab_bytes, st, grp); \
        else hipLaunchKernelGGL((conv_f16x3_kernel<MT_, NT_>), grid16, block, tab_bytes, st, grp);                   \
    } while (0)
    if (arith == OJF_ARITH_F16X3 && mt == 2) {
        switch (nt) {
            case 2:
                if (n_ins && lean) hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, true, 0, true, true>), grid16, block, tab_bytes, st, grp);
                else if (n_ins) hipLaunchKernelGGL((conv_f16x3_kernel<2, 2, true, 0, false, true>), grid16, block, tab_bytes, st, grp);
                else OJF_LAUNCH16(2, 2);
                break;
            case 4: OJF_LAUNCH16(2, 4); break;
            case 6: OJF_LAUNCH16(2, 6); break;
            case 8: OJF_LAUNCH16(2, 8); break;
            default: return fail("conv: unsupported number of output tiles");
        }
    } else if (arith == OJF_ARITH_F16X3) {
        switch (nt) {
            case 2: OJF_LAUNCH16(1, 2); break;
            case 4: OJF_LAUNCH16(1, 4); break;
            case 6: OJF_LAUNCH16(1, 6); break;
            case 8: OJF_LAUNCH16(1, 8); break;
            default: return fail("conv: unsupported number of output tiles");
        }
    } else {
        static const bool no_lds32 = getenv("OJF_CONV32_LDS") && atoi(getenv("OJF_CONV32_LDS")) == 0;  // A/B switch
        switch (nt) {
            case 2: OJF_LAUNCH32(2); break;
            case 4: OJF_LAUNCH32(4); break;
            case 6:
                if (no_lds32) OJF_LAUNCH32(6);
                else hipLaunchKernelGGL((conv_mfma_lds_kernel<6>), grid, block, 0, st, grp);
                break;
            case 8:
                if (no_lds32) OJF_LAUNCH32(8);
                else hipLaunchKernelGGL((conv_mfma_lds_kernel<8>), grid, block, 0, st, grp);
                break;
            default: return fail("conv: unsupported number of output tiles");
        }
    }
#undef OJF_LAUNCH32
#undef OJF_LAUNCH16
    mark_launch(arith == OJF_ARITH_F16X3 ? (n > 1 ? "conv_f16x3_kernel (grouped)" : "conv_f16x3_kernel") : "conv_mfma_kernel", st);
    return check_hip(hipGetLastError(), "conv kernel launch");
}

// magic of fast_div for divisor d and dividends below `limit` (0 = use a real division)
static unsigned div_magic(int d, uint64_t limit)
{
    if (d <= 1 || limit * (uint64_t)d >= (1ull << 32)) return 0;
    return (unsigned)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d);
}

static void fill_conv_args(ConvArgs &a, const PackedConv &pc, const float *in, int in_g0, float *out, int out_g0,
                           const float *bias, int act, int act_n, float scale, int h, int w)
{
    a.ovf = pc.arith == OJF_ARITH_F16X3 ? overflow_flag() : nullptr;
    a.accum = 0; a.dscale = nullptr;
    a.in = planes(in); a.out = planes(out); a.out_rows = nullptr;
    a.wp = planes(pc.wp); a.bias = bias ? bias : pc.bias; a.rinv = pc.rinv;
    a.in_g0 = in_g0; a.out_g0 = out_g0; a.rows_stride = 0; a.rows_n = 0;
    a.h = h; a.w = w; a.npix = h * w;
    a.taps = pc.taps; a.dil = pc.dil;
    a.c4 = pc.c_in_phys / 4; a.nsteps = pc.nsteps;
    a.w_magic = div_magic(w, (uint64_t)h * w);
    a.c4_magic = div_magic(a.c4, (uint64_t)(pc.nsteps + kPad16) * 8);
    a.og_store = round_up(pc.c_out_phys, 4) / 4;
    a.act = act; a.act_n = act_n; a.scale = scale;
}

// in / out: plane buffers; in_g0 / out_g0: first channel group of the window.
// rows != NULL: the result goes to row-major scalars instead (last layer -> caller's est rows).
static int launch_conv(const PackedConv &pc, const float *in, int in_g0, float *out, int out_g0, const float *bias,
                       int act, int act_n, float scale, int h, int w, hipStream_t st, float *rows = nullptr,
                       int rows_stride = 0, int rows_n = 0)
{
    ConvArgs a;
    fill_conv_args(a, pc, in, in_g0, out, out_g0, bias, act, act_n, scale, h, w);
    a.out_rows = rows; a.rows_stride = rows_stride; a.rows_n = rows_n;
    return launch_conv_args(&a, 1, pc.n_ot, st, pc.arith);
}

static std::vector<int> slot_map(int n_logical, int group, int slot)
{   // logical channel j of a concatenation of `group`-wide tensors stored in `slot`-wide slots
    std::vector<int> m(n_logical);
    for (int j = 0; j < n_logical; ++j) m[j] = (j / group) * slot + (j % group);
    return m;
}

// per-output-row power-of-two scales of a pointwise layer (ones for fp32 arithmetic); n_k input channels
template <class F>
static std::vector<float> chain_row_scales(int arith, int nt_out, int n_k, F weight)
{
    std::vector<float> r((size_t)nt_out * 16, 1.0f);
    if (arith != OJF_ARITH_F16X3) return r;
    for (int oc = 0; oc < nt_out * 16; ++oc) {
        float mx = 0.0f;
        for (int k = 0; k < n_k; ++k) mx = std::fmax(mx, std::fabs(weight(oc, k)));
        r[oc] = row_scale(mx);
    }
    return r;
}

// Fragments of one pointwise layer for chain_layer, appended to dst; weight(oc, k) returns 0 outside the layer,
// rs = row scales (chain_row_scales).
template <class F>
static void pack_chain_layer(std::vector<float> &dst, int arith, int nt_out, int nt_in, const std::vector<float> &rs, F weight)
{
    const size_t base = dst.size();
    dst.resize(base + (size_t)chain_layer_size(arith, nt_in, nt_out) * 4, 0.0f);
    if (arith == OJF_ARITH_F16X3) {
        const int KB = chain_kblocks(arith, nt_in);
        _Float16 *hp = reinterpret_cast<_Float16 *>(dst.data() + base);
        for (int n2 = 0; n2 < nt_out; ++n2)
            for (int S = 0; S < KB; ++S)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = n2 * 16 + (lane & 15), g = lane >> 4;
                        const int k = 32 * S + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                        const size_t ub = ((size_t)n2 * KB + S) * 2 * 64 * 8;
                        split_weight(rs[oc] * weight(oc, k), hp[ub + (size_t)lane * 8 + j], hp[ub + 64 * 8 + (size_t)lane * 8 + j]);
                    }
    } else {
        for (int n2 = 0; n2 < nt_out; ++n2)
            for (int S = 0; S < nt_in; ++S)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j)
                        dst[base + (((size_t)n2 * nt_in + S) * 64 + lane) * 4 + j] =
                            weight(n2 * 16 + (lane & 15), 16 * S + 4 * (lane >> 4) + j);
    }
}

// bias stream entry of a chain layer: bias[nt_out*16] and, for split-fp16, the inverse row scales behind it
static void append_chain_bias(std::vector<float> &dst, int arith, int nt_out, int c_out, const float *bias_host,
                              const std::vector<float> &rs)
{
    const size_t bb = dst.size();
    dst.resize(bb + (size_t)chain_bias_stride(arith, nt_out), 0.0f);
    for (int o = 0; o < c_out; ++o) dst[bb + o] = bias_host[o];
    if (arith == OJF_ARITH_F16X3)
        for (int o = 0; o < nt_out * 16; ++o) dst[bb + (size_t)nt_out * 16 + o] = 1.0f / rs[o];
}

struct Vortex {
    int c_in = 0, c_in_phys = 0;
    PackedConv stacked, b3a[4], b3b[4], b1[4], fin;
    float *pool_bias[4] = {nullptr, nullptr, nullptr, nullptr};
    float *Wg = nullptr, *bg = nullptr, *Wfg = nullptr, *bf = nullptr, *bias_final = nullptr;
    float *tail_w = nullptr, *tail_b1 = nullptr, *tail_rinv = nullptr;  // fused tail (closing 1x1s + final conv), when supported
    float *entry_w = nullptr, *entry_b = nullptr;  // the stacked entry GEMM as one chain layer (8 or 16 input tiles -> 5), when supported
    int entry_ntin = 8;
    // two-head nets (round 6): the last VortexPooling's entry GEMM over the concatenation of the two heads' outputs splits into
    // one 8-tile -> 5 layer per head (columns [hd * os, (hd + 1) * os) of the stacked weights, no bias, no activation): each head's
    // tail runs ITS half on its register-resident result; the pool pyramid adds the halves (and forms branch 0)
    float *entry_wh[2] = {nullptr, nullptr}, *entry_bh[2] = {nullptr, nullptr}, *bias0 = nullptr;
    PackedBranches branches;  // both 3x3 of the four branches for vortex_branch_kernel (split-fp16, 20-channel slots)
};

}  // namespace ojf

struct ojf_net {
    int version, P, c, cs, gf, sem, heads, h, w, npix;
    int arith;  // OJF_ARITH_* captured at creation
    int pool_in, os;  // (gf+1)*c and its padded slot
    float scale;
    int64_t macs_per_pixel;
    int last_launches = 0;  // kernel launches of the most recent ojf_net_forward (all streams)
    std::vector<ojf::PackedConv> dense[2];  // block0 / block2 (or v2's block): 2*gf convs each
    std::vector<ojf::PackedPair> pairs[2];  // split-fp16: the same layers packed for dense_pair_kernel (one launch per Block)
    ojf::PackedChain chains[2];             // ... and for dense_chain_kernel (all Blocks of a head in one launch)
    bool chain_dense = false;               // X[] hold split planes, run_dense = one dense_chain_kernel launch per head
    ojf::Vortex vortex[3];                  // v3: vortex0, vortex2, vortex3 ; v2: vortex, -, vortex_final
    std::vector<ojf::PackedConv> pred;
    float *chain_w = nullptr, *chain_b = nullptr;  // fused prediction head (when the topology is supported)
    int chain_kind = 0;                             // 0 = unfused, 19 / 20 = growth channels of the fused kernel
    // activation planes (C4 layout), sizes in channels
    float *X[2] = {nullptr, nullptr};  // dense-growth buffers, (gf+1)*cs
    // Scratch of one head's dense block + VortexPooling.  Two sets: with two heads (v3 + semantics) the second head runs
    // concurrently on its own stream (the kernels are latency-bound and leave most of the GPU idle), so it needs its
    // own intermediates, side stream and events; everything else (and the last VortexPooling) uses set 0.
    struct Scratch {
        float *T = nullptr, *Z = nullptr, *Q1 = nullptr, *Q2 = nullptr, *Q3 = nullptr, *U = nullptr, *V = nullptr;
        float *Q0 = nullptr;  // two-head nets: branch 0 of the last VortexPooling (the pyramid's level-0 pass)
        float *partial = nullptr;
        ojf::ColSums *colsum = nullptr;  // chain flow: fixed-point channel sums of the entry layer's input (zero between frames)
        float *CAT = nullptr;  // 4*os, unfused tail only (allocated at first use)
        hipStream_t side = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_entry = nullptr;
    } sc[2];
    hipStream_t head1 = nullptr;  // stream of the second head
    hipStream_t head_paired_with = nullptr;  // launch stream head1 was tested against (pair_head_stream)
    bool head_paired = false;
    hipEvent_t ev_head_fork = nullptr, ev_head_join = nullptr;
    // (per set: T cs | Z 4*cs entry-conv output | Q1..Q3 cs pool-pyramid outputs | U, V 4*cs outputs of the branches'
    //  first / closing 3x3 | partial kSumBlocks*256 | side stream + events of the global-average branch)
    // two-head nets: the channel sums of the LAST VortexPooling's input come from the two heads' tails, which run on two streams beside
    // the first head's own VortexPooling - whose pyramid reads and zeroes sc[0].colsum: they need an accumulator of their own
    ojf::ColSums *colsum3 = nullptr;
    float *YY = nullptr;               // heads*os (vortex0 | vortex2 outputs)
    float *Y3 = nullptr;               // os
    float *PA = nullptr, *PB = nullptr;  // pred ping-pong, os each
    // Optional (OJF_NET_GRAPH=1): the launch sequence of a forward pass captured once per output buffer and
    // replayed with hipGraphLaunch.  Measured on ROCm 7.2 / MI355X: host enqueue time of a forward 160 -> 87 us,
    // but the GPU runs the graph 3-8 % SLOWER than the plain launches (0.555 vs 0.538 ms at 320x240,
    // 0.293 vs 0.270 ms at 160x120), and the host (0.21 ms/frame in Pipeline.fuse) is not the bound: off by default.
    hipStream_t cap = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    float *g_est = nullptr;
    int g_stride = 0;
    bool use_graph = false;
};

namespace ojf {

// every activation buffer keeps kPrefix floats of zeros in front of its planes: index -1 (in float4
// units) is what out-of-image taps read
constexpr size_t kPrefix = 64;  // keeps the planes 256-byte aligned

static int alloc_planes(float **p, size_t npix, int ch)
{
    float *raw = nullptr;
    OJF_HIP(hipMalloc(reinterpret_cast<void **>(&raw), (npix * ch + kPrefix) * sizeof(float)));
    OJF_HIP(hipMemset(raw, 0, (npix * ch + kPrefix) * sizeof(float)));
    // The fill runs on the null stream; the second head's stream and the side streams are hipStreamNonBlocking and do NOT
    // order against it.  A buffer allocated at its first use inside a forward pass (ensure_planes) could be written by a
    // convolution on such a stream and zeroed afterwards (round 5: one wrong output in ~8 suite runs of the generic two-head
    // topology, never in isolation).  The fill is therefore complete before anybody learns the pointer.
    OJF_HIP(hipStreamSynchronize(nullptr));
    *p = raw + kPrefix;
    return 0;
}

static void free_planes(float *p)
{
    if (p) (void)hipFree(p - kPrefix);
}

// buffers only the unfused fall-back paths need are allocated at their first use (synchronous, outside any capture)
static int ensure_planes(float **p, size_t npix, int ch)
{
    return *p ? 0 : alloc_planes(p, npix, ch);
}

static int build_vortex(ojf_net *net, Vortex &v, const ojf_conv_layer *L, int c_in, const std::vector<int> &in_map,
                        int c_in_phys)
{
    // L[0] gave 1x1 | L[1+4b .. 4+4b] branch b: 1x1, 3x3, 3x3, 1x1 | L[17] final 1x1 (5*out -> out)
    const int c = net->c, cs = net->cs, out = net->pool_in, os = net->os;
    v.c_in = c_in;
    v.c_in_phys = c_in_phys;
    for (int i = 0; i < 18; ++i) {
        const bool k3 = (i >= 1 && i <= 16) && ((i - 1) % 4 == 1 || (i - 1) % 4 == 2);
        if (L[i].ksize != (k3 ? 3 : 1)) return fail("ojf_net_create: unexpected kernel size in a VortexPooling layer");
    }
    if (L[0].c_in != c_in || L[0].c_out != out || L[17].c_in != 5 * out || L[17].c_out != out)
        return fail("ojf_net_create: VortexPooling layer shapes do not match the topology");
    {   // stacked branch-entry 1x1: c_in -> 4 slots of cs; only branch 0 gets its bias (+ReLU) here
        ConvBuilder b(c_in_phys, 4 * cs, 1, 1);
        for (int br = 0; br < 4; ++br) {
            const ojf_conv_layer &l = L[1 + 4 * br];
            if (l.c_in != c_in || l.c_out != c) return fail("ojf_net_create: branch entry conv shape mismatch");
            b.add(l, 0, c_in, in_map, br * cs, br == 0);
            if (br > 0) {
                std::vector<float> pb(cs, 0.0f);
                for (int o = 0; o < c; ++o) pb[o] = l.bias_host[o];
                if (upload(pb, &v.pool_bias[br])) return -2;
            }
        }
        if (finish(b, v.stacked, net->arith)) return -2;
        static const bool no_wide_entry = getenv("OJF_NO_WIDE_ENTRY") != nullptr;  // A/B switch (round 6)
        if ((c_in_phys <= 128 || (c_in_phys <= 256 && !no_wide_entry)) && 4 * cs <= 80) {
            // chain-layer form (entry1x1_kernel / the previous tail): 8 input tiles -> 5; the two-head net's last VortexPooling
            // (228 = 2 x 114 input channels): 16 -> 5 (round 6: it ran the generic convolution + two side-stream kernels + three events)
            v.entry_ntin = c_in_phys <= 128 ? 8 : 16;
            auto we = [&](int oc, int k) { return oc < 4 * cs && k < c_in_phys ? b.W[(size_t)oc * c_in_phys + k] : 0.0f; };
            const std::vector<float> re = chain_row_scales(net->arith, 5, c_in_phys, we);
            std::vector<float> ew, eb;
            pack_chain_layer(ew, net->arith, 5, v.entry_ntin, re, we);
            append_chain_bias(eb, net->arith, 5, 4 * cs, b.B.data(), re);
            if (upload(ew, &v.entry_w) || upload(eb, &v.entry_b)) return -2;
            static const bool no_half = getenv("OJF_NO_HALF_ENTRY") != nullptr;  // A/B switch (round 6)
            if (v.entry_ntin == 16 && c_in_phys == 2 * net->os && !no_half) {
                const std::vector<float> zero_bias(4 * cs, 0.0f);
                for (int hd = 0; hd < 2; ++hd) {
                    auto wh = [&](int oc, int k) { return oc < 4 * cs && k < net->os ? b.W[(size_t)oc * c_in_phys + hd * net->os + k] : 0.0f; };
                    const std::vector<float> rh = chain_row_scales(net->arith, 5, net->os, wh);
                    std::vector<float> hw, hb;
                    pack_chain_layer(hw, net->arith, 5, 8, rh, wh);
                    append_chain_bias(hb, net->arith, 5, 4 * cs, zero_bias.data(), rh);
                    if (upload(hw, &v.entry_wh[hd]) || upload(hb, &v.entry_bh[hd])) return -2;
                }
                std::vector<float> b0(cs, 0.0f);
                for (int o = 0; o < cs; ++o) b0[o] = b.B[o];  // branch 0's bias (+ ReLU): applied by the pyramid's level-0 pass
                if (upload(b0, &v.bias0)) return -2;
            }
        }
    }
    const std::vector<int> id_c = slot_map(c, c, cs);
    std::vector<ConvBuilder> bas, bbs;
    for (int br = 0; br < 4; ++br) {
        const ojf_conv_layer &la = L[2 + 4 * br], &lb = L[3 + 4 * br], &l1 = L[4 + 4 * br];
        if (la.c_in != c || la.c_out != c || lb.c_in != c || lb.c_out != c || l1.c_in != c || l1.c_out != out)
            return fail("ojf_net_create: branch conv shape mismatch");
        ConvBuilder ba(cs, cs, 3, la.dilation), bb(cs, cs, 3, lb.dilation), b1(cs, os, 1, 1);
        ba.add(la, 0, c, id_c, 0, true);
        bb.add(lb, 0, c, id_c, 0, true);
        b1.add(l1, 0, c, id_c, 0, true);
        if (finish(ba, v.b3a[br], net->arith) || finish(bb, v.b3b[br], net->arith) || finish(b1, v.b1[br], net->arith)) return -2;
        bas.push_back(ba);
        bbs.push_back(bb);
    }
    if (net->arith == OJF_ARITH_F16X3 && cs == 4 * kChainNG && v.entry_w && bas[0].dil == bbs[0].dil && bas[1].dil == bbs[1].dil &&
        bas[2].dil == bbs[2].dil && bas[3].dil == bbs[3].dil && finish_branches(bas, bbs, net->h, net->w, v.branches))
        return -2;
    {   // final 1x1 over [gave | b0 | b1 | b2 | b3]; the gave columns become a per-frame bias
        const ojf_conv_layer &lf = L[17];
        ConvBuilder b(4 * os, os, 1, 1);
        b.add(lf, out, 5 * out, slot_map(4 * out, out, os), 0, false);
        if (finish(b, v.fin, net->arith)) return -2;
        std::vector<float> Wg((size_t)out * c_in_phys, 0.0f), bg(out), Wfg((size_t)out * out), bf(out);
        for (int o = 0; o < out; ++o) {  // stored transposed: [input][output]
            for (int ci = 0; ci < c_in; ++ci) Wg[(size_t)in_map[ci] * out + o] = L[0].weight_host[(size_t)o * c_in + ci];
            bg[o] = L[0].bias_host[o];
            for (int j = 0; j < out; ++j) Wfg[(size_t)j * out + o] = lf.weight_host[(size_t)o * 5 * out + j];
            bf[o] = lf.bias_host[o];
        }
        if (upload(Wg, &v.Wg) || upload(bg, &v.bg) || upload(Wfg, &v.Wfg) || upload(bf, &v.bf)) return -2;
        std::vector<float> zero((size_t)v.fin.n_ot * 16, 0.0f);
        if (upload(zero, &v.bias_final)) return -2;
    }
    if ((cs + 15) / 16 == 2 && (os + 15) / 16 == 8) {  // fused tail: [W1_b | Wf_b] fragments per branch
        const int NV = 2, NO = 8;
        std::vector<float> tw, tb;
        const ojf_conv_layer &lf = L[17];
        // the four column blocks of the final conv accumulate into one set of rows: one common row scale
        const std::vector<float> rf = chain_row_scales(net->arith, NO, 4 * out, [&](int oc, int k) {
            return oc < out ? lf.weight_host[(size_t)oc * 5 * out + out + k] : 0.0f;
        });
        for (int br = 0; br < 4; ++br) {
            const ojf_conv_layer &l1 = L[4 + 4 * br];
            auto w1 = [&](int oc, int k) { return oc < out && k < c ? l1.weight_host[(size_t)oc * c + k] : 0.0f; };
            const std::vector<float> r1 = chain_row_scales(net->arith, NO, c, w1);
            pack_chain_layer(tw, net->arith, NO, NV, r1, w1);
            pack_chain_layer(tw, net->arith, NO, NO, rf, [&](int oc, int k) {
                return oc < out && k < out ? lf.weight_host[(size_t)oc * 5 * out + out * (br + 1) + k] : 0.0f;
            });
            append_chain_bias(tb, net->arith, NO, out, l1.bias_host, r1);
        }
        std::vector<float> rfi(rf.size());
        for (size_t i = 0; i < rf.size(); ++i) rfi[i] = 1.0f / rf[i];
        if (upload(rfi, &v.tail_rinv)) return -2;
        if (upload(tw, &v.tail_w) || upload(tb, &v.tail_b1)) return -2;
    }
    return 0;
}

static void free_vortex(Vortex &v)
{
    release(v.stacked);
    release(v.fin);
    for (int b = 0; b < 4; ++b) {
        release(v.b3a[b]);
        release(v.b3b[b]);
        release(v.b1[b]);
        if (v.pool_bias[b]) (void)hipFree(v.pool_bias[b]);
    }
    release(v.branches);
    float *ptrs[] = {v.Wg, v.bg, v.Wfg, v.bf, v.bias_final, v.tail_w, v.tail_b1, v.tail_rinv, v.entry_w, v.entry_b,
                     v.entry_wh[0], v.entry_wh[1], v.entry_bh[0], v.entry_bh[1], v.bias0};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
}

// in: planes, window starting at group in_g0 (c_in_phys/4 groups); out: planes at group out_g0
static GaveArgs gave_args(const ojf_net *net, const Vortex &v, const float *partial, int nparts, int pstride, ColSums *fixed = nullptr)
{
    GaveArgs ga;
    ga.partial = partial; ga.fixed = fixed; ga.nparts = nparts; ga.pstride = pstride; ga.cphys = v.c_in_phys; ga.npix = net->npix;
    ga.WgT = v.Wg; ga.bg = v.bg; ga.WfgT = v.Wfg; ga.bf = v.bf; ga.c_out = net->pool_in;
    ga.bias_out = v.bias_final; ga.bias_len = v.fin.n_ot * 16;
    return ga;
}

static int chain_blocks(const ojf_net *net) { return ((net->npix + 15) / 16 + ojf::kChainWaves - 1) / ojf::kChainWaves; }  // blocks of the MT = 1 chain kernels

// in: planes, window starting at group in_g0 (c_in_phys/4 groups); out: planes at group out_g0.
// entry_done: the previous VortexPooling's tail already left this one's entry planes (sc.Z) and column sums (sc.colsum).
// next: when given (and supported), this tail runs the NEXT VortexPooling's entry GEMM on its register-resident result
//       instead of writing `out`; *next_done reports it.
// half / next_colsum (two-head nets): this head's tail runs half `half` of `next`'s entry GEMM (partial sums into sc.Z, the channel
//       sums of its result into next_colsum at channel half * os).  z2: the OTHER head's partial entry planes - this
//       VortexPooling's pyramid adds them to sc.Z and forms branch 0 (entry_done must be set).
static int run_vortex(ojf_net *net, Vortex &v, const float *in, int in_g0, float *out, int out_g0, hipStream_t st,
                      ojf_net::Scratch &sc, const ChainArgs *head = nullptr, bool *head_done = nullptr,
                      bool entry_done = false, const Vortex *next = nullptr, bool *next_done = nullptr, bool in_split = false,
                      int half = -1, ColSums *next_colsum = nullptr, const float *z2 = nullptr)
{
    const int h = net->h, w = net->w, c4 = net->cs / 4, o4 = net->os / 4;
    const bool h16 = net->arith == OJF_ARITH_F16X3;
    static const bool legacy_env = getenv("OJF_LEGACY_VORTEX") != nullptr;  // ablation switch only
    // Chain flow (no side stream, no events): entry GEMM + column sums in one chain-layer launch (or inside the previous
    // tail), the global-average fold as one extra block of the pyramid launch, all four branches in the grouped launches.
    const bool chain_flow = v.entry_w && v.tail_w && !legacy_env;
    // split planes between the branch-entry producers (entry GEMM: branch 0; pool pyramid: branches 1-3), the branches' first
    // 3x3 and their second 3x3 (whose result the tail reads as plain fp32 planes)
    static const bool no_split = getenv("OJF_CONV_SPLIT") && atoi(getenv("OJF_CONV_SPLIT")) == 0;  // A/B switch
    const bool split = h16 && chain_flow && !no_split;
    if (in_split && (!chain_flow || entry_done)) return fail("run_vortex: internal error (split-plane input without the entry kernel)");
    if (!chain_flow) {
        if (entry_done) return fail("run_vortex: internal error (entry planes without the chain flow)");
        // global-average branch -> bias of the final conv, on the side stream (fork here, join before the tail)
        OJF_HIP(hipEventRecord(sc.ev_fork, st));
        OJF_HIP(hipStreamWaitEvent(sc.side, sc.ev_fork, 0));
        hipLaunchKernelGGL(colsum_kernel, dim3(kSumBlocks, v.c_in_phys / 4), dim3(256), 0, sc.side, planes(in), in_g0,
                           net->npix, sc.partial);
        mark_launch("colsum_kernel", sc.side);
        OJF_HIP(hipGetLastError());
        hipLaunchKernelGGL(gave_bias_kernel, dim3(1), dim3(256), 0, sc.side, gave_args(net, v, sc.partial, kSumBlocks, 256));
        mark_launch("gave_bias_kernel", sc.side);
        OJF_HIP(hipGetLastError());
        // branch entries: one GEMM, branch 0 gets bias + ReLU in the epilogue
        if (launch_conv(v.stacked, in, in_g0, sc.Z, 0, nullptr, OJF_ACT_RELU, net->cs, 1.0f, h, w, st)) return -2;
        OJF_HIP(hipEventRecord(sc.ev_entry, st));
    } else if (!entry_done) {
        ChainArgs ea;
        ea.in = planes(in); ea.w = planes(v.entry_w); ea.bias = v.entry_b; ea.out_rows = nullptr;
        ea.in_g0 = in_g0; ea.c4_in = v.c_in_phys / 4; ea.npix = net->npix; ea.rows_stride = 0; ea.rows_n = 0; ea.scale = 1.0f;
        ea.ovf = h16 ? overflow_flag() : nullptr;
        ea.out_planes = planes(sc.Z); ea.out_g0 = 0; ea.og_store = 4 * c4; ea.act_n = net->cs; ea.colsum = sc.colsum;
        ea.split_groups = split ? c4 : 0;
        ea.in_split = in_split ? 1 : 0;
        const dim3 grid(chain_blocks(net)), block(ojf::kChainThreads);
        // (two pixel tiles per wave - half the blocks, half the weight stream per pixel - measured 22.2 against 23.7 us and changes the
        // rounding of the channel sums: not kept, profiles/r06_fusion_net_experiments.txt)
        if (v.entry_ntin == 16 && h16) hipLaunchKernelGGL((entry1x1_kernel<OJF_ARITH_F16X3, 1, 16, 5>), grid, block, 0, st, ea);
        else if (v.entry_ntin == 16) hipLaunchKernelGGL((entry1x1_kernel<OJF_ARITH_F32, 1, 16, 5>), grid, block, 0, st, ea);
        else if (h16) hipLaunchKernelGGL((entry1x1_kernel<OJF_ARITH_F16X3, 1, 8, 5>), grid, block, 0, st, ea);
        else hipLaunchKernelGGL((entry1x1_kernel<OJF_ARITH_F32, 1, 8, 5>), grid, block, 0, st, ea);
        mark_launch("entry1x1_kernel", st);
        OJF_HIP(hipGetLastError());
    }
    {   // pool pyramid on the pre-activations of branches 1..3: Q_b = ReLU(pool^b(Z[slot b]) + bias_b), one launch
        PyramidArgs pa;
        pa.z = planes(sc.Z);
        pa.q[0] = planes(sc.Q1); pa.q[1] = planes(sc.Q2); pa.q[2] = planes(sc.Q3);
        for (int b = 0; b < 3; ++b) pa.bias[b] = v.pool_bias[b + 1];
        pa.h = h; pa.w = w; pa.c4 = c4;
        pa.tiles = ((w + kPoolTW - 1) / kPoolTW) * ((h + kPoolTH - 1) / kPoolTH);
        pa.gave = gave_args(net, v, nullptr, 0, 0, (z2 && net->colsum3) ? net->colsum3 : sc.colsum);
        pa.fold_gave = chain_flow ? 1 : 0;
        pa.split_out = split ? 1 : 0;
        if (z2) {
            if (!entry_done || !chain_flow || !sc.Q0 || !v.bias0) return fail("run_vortex: internal error (partial entry sums without the chain flow)");
            pa.z2 = planes(z2); pa.q0 = planes(sc.Q0); pa.bias0 = v.bias0;
        }
        // grid.x: the tiles (+ the global-average block of the chain flow) rounded up to a multiple of 8 (XCD bands)
        hipLaunchKernelGGL(pool_pyramid_kernel, dim3(round_up(pa.tiles + (chain_flow ? 1 : 0), 8), (z2 ? 4 : 3) * c4), dim3(256), 0, st, pa);
        mark_launch("pool_pyramid_kernel", st);
        OJF_HIP(hipGetLastError());
    }
    const float *bin[4] = {z2 ? sc.Q0 : sc.Z, sc.Q1, sc.Q2, sc.Q3};
    static const bool unfused = getenv("OJF_NO_TAIL") != nullptr;  // ablation switch only
    const bool fused = v.tail_w && !unfused;
    // (measured equal to the two grouped launches, not faster - 82 against 79-80 us per frame, profiles/r04_pair_experiments.txt -
    // so it is opt-in: OJF_BRANCH_KERNEL=1)
    static const bool branch_kernel = getenv("OJF_BRANCH_KERNEL") != nullptr && atoi(getenv("OJF_BRANCH_KERNEL")) != 0;
    static const bool subconv = getenv("OJF_SUBCONV") != nullptr && atoi(getenv("OJF_SUBCONV")) != 0;
    if (split && v.branches.n_items && branch_kernel) {  // both 3x3 of all four branches: one launch (ojf_net_branch.h)
        if (launch_branches(v.branches, bin, sc.V, h, w, st)) return -2;
    } else if (split && v.branches.n_sub && subconv) {  // LDS-staged sub-image blocks: one launch per 3x3 (ojf_net_branch.h)
        float *u[4], *vv[4];
        const float *uc[4];
        for (int br = 0; br < 4; ++br) {
            u[br] = sc.U + (size_t)br * c4 * 4 * net->npix;
            uc[br] = u[br];
            vv[br] = sc.V + (size_t)br * c4 * 4 * net->npix;
        }
        if (launch_subconv(v.branches, bin, u, 0, true, h, w, st)) return -2;
        if (launch_subconv(v.branches, uc, vv, 1, false, h, w, st)) return -2;
    } else {   // the four branches' dilated 3x3 pairs: two grouped launches (all first convs, then all second convs)
        ConvArgs ga[4], gb[4];
        for (int br = 0; br < 4; ++br) {
            fill_conv_args(ga[br], v.b3a[br], bin[br], 0, sc.U, br * c4, nullptr, OJF_ACT_RELU, net->cs, 1.0f, h, w);
            fill_conv_args(gb[br], v.b3b[br], sc.U, br * c4, sc.V, br * c4, nullptr, OJF_ACT_RELU, net->cs, 1.0f, h, w);
            ga[br].in_split = ga[br].out_split = gb[br].in_split = split ? 1 : 0;
        }
        // Legacy flow: branch 0 needs no pooling, its 3x3 pair follows the global-average kernels on the side stream
        static const bool b0_main = getenv("OJF_BRANCH0_MAIN") != nullptr;  // ablation switch only
        if (chain_flow || b0_main || net->npix < 32768) {  // (small frames: the stream hand-over costs more than it hides)
            if (launch_conv_args(ga, 4, v.b3a[0].n_ot, st, net->arith)) return -2;
            if (launch_conv_args(gb, 4, v.b3b[0].n_ot, st, net->arith)) return -2;
        } else {
            OJF_HIP(hipStreamWaitEvent(sc.side, sc.ev_entry, 0));
            if (launch_conv_args(ga, 1, v.b3a[0].n_ot, sc.side, net->arith)) return -2;
            if (launch_conv_args(gb, 1, v.b3b[0].n_ot, sc.side, net->arith)) return -2;
            if (launch_conv_args(ga + 1, 3, v.b3a[1].n_ot, st, net->arith)) return -2;
            if (launch_conv_args(gb + 1, 3, v.b3b[1].n_ot, st, net->arith)) return -2;
        }
    }
    if (!chain_flow) {
        OJF_HIP(hipEventRecord(sc.ev_join, sc.side));
        OJF_HIP(hipStreamWaitEvent(st, sc.ev_join, 0));  // bias of the final conv is ready
    }
    if (!fused) {
        if (ensure_planes(&sc.CAT, (size_t)net->npix, 4 * net->os)) return -2;
        for (int br = 0; br < 4; ++br)
            if (launch_conv(v.b1[br], sc.V, br * c4, sc.CAT, br * o4, nullptr, OJF_ACT_RELU, net->os, 1.0f, h, w, st))
                return -2;
    }
    if (!fused) return launch_conv(v.fin, sc.CAT, 0, out, out_g0, v.bias_final, OJF_ACT_NONE, 0, 1.0f, h, w, st);
    TailArgs ta;
    for (int br = 0; br < 4; ++br) ta.v[br] = planes(sc.V) + (size_t)br * c4 * net->npix;
    ta.w = planes(v.tail_w); ta.b1 = v.tail_b1; ta.bias_final = v.bias_final; ta.rinv_final = v.tail_rinv;
    ta.out = planes(out); ta.c4 = c4; ta.out_g0 = out_g0; ta.og_store = o4; ta.npix = net->npix;
    ta.ovf = h16 ? overflow_flag() : nullptr;
    ta.chain_w = nullptr; ta.chain_b = nullptr; ta.out_rows = nullptr; ta.rows_stride = 0; ta.rows_n = 0; ta.scale = 1.0f;
    ta.entry_w = nullptr; ta.entry_b = nullptr; ta.entry_out = nullptr; ta.entry_og = 0; ta.entry_act_n = 0; ta.colsum = nullptr; ta.entry_split = 0;
    // MT = 1: two pixel tiles per wave (324 VGPRs, one wave per SIMD) measured slower (0.624 vs 0.609 ms net)
    const dim3 grid(chain_blocks(net)), block(ojf::kChainThreads);
    static const bool no_head_fusion = getenv("OJF_NO_HEAD_FUSION") != nullptr;  // ablation switch only
    static const bool no_entry_fusion = getenv("OJF_NO_ENTRY_FUSION") != nullptr;  // ablation switch only
    if (head && net->chain_kind && !no_head_fusion) {  // last VortexPooling: the prediction head rides along
        ta.chain_w = head->w; ta.chain_b = head->bias; ta.out_rows = head->out_rows;
        ta.rows_stride = head->rows_stride; ta.rows_n = head->rows_n; ta.scale = head->scale;
        if (net->chain_kind == 19 && h16) hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F16X3, 1, 2, 8, 19>), grid, block, 0, st, ta);
        else if (net->chain_kind == 19) hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F32, 1, 2, 8, 19>), grid, block, 0, st, ta);
        else if (h16) hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F16X3, 1, 2, 8, 20>), grid, block, 0, st, ta);
        else hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F32, 1, 2, 8, 20>), grid, block, 0, st, ta);
        mark_launch("vortex_tail_kernel (+ prediction head)", st);
        if (head_done) *head_done = true;
    } else if (chain_flow && next && half >= 0 && half < 2 && next->entry_wh[half] && next->tail_w && next_colsum && !legacy_env && !no_entry_fusion) {
        // two-head net: this head's HALF of the next VortexPooling's entry GEMM rides along (partial sums: no bias, no activation,
        // plain planes); `out` is never written
        ta.entry_w = planes(next->entry_wh[half]); ta.entry_b = next->entry_bh[half]; ta.entry_out = planes(sc.Z);
        ta.entry_og = 4 * c4; ta.entry_act_n = 0; ta.colsum = next_colsum; ta.colsum_off = half * net->os;
        ta.entry_split = 0;
        if (h16) hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F16X3, 1, 2, 8, kTailEntry>), grid, block, 0, st, ta);
        else hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F32, 1, 2, 8, kTailEntry>), grid, block, 0, st, ta);
        mark_launch("vortex_tail_kernel (+ half of the next entry GEMM)", st);
        if (next_done) *next_done = true;
    } else if (chain_flow && next && next->entry_w && next->tail_w && next->c_in_phys == net->os && !legacy_env && !no_entry_fusion) {
        // this VortexPooling feeds the next one: its entry GEMM and column sums ride along, `out` is never written
        ta.entry_w = planes(next->entry_w); ta.entry_b = next->entry_b; ta.entry_out = planes(sc.Z);
        ta.entry_og = 4 * c4; ta.entry_act_n = net->cs; ta.colsum = sc.colsum;
        ta.entry_split = (h16 && !no_split) ? c4 : 0;  // (the next VortexPooling runs the chain flow: same rule as its `split`)
        if (h16) hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F16X3, 1, 2, 8, kTailEntry>), grid, block, 0, st, ta);
        else hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F32, 1, 2, 8, kTailEntry>), grid, block, 0, st, ta);
        mark_launch("vortex_tail_kernel (+ next entry GEMM)", st);
        if (next_done) *next_done = true;
    } else if (h16) {
        hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F16X3, 1, 2, 8>), grid, block, 0, st, ta);
        mark_launch("vortex_tail_kernel", st);
    } else {
        hipLaunchKernelGGL((vortex_tail_kernel<OJF_ARITH_F32, 1, 2, 8>), grid, block, 0, st, ta);
        mark_launch("vortex_tail_kernel", st);
    }
    return check_hip(hipGetLastError(), "vortex_tail_kernel launch");
}

static int run_dense(ojf_net *net, int head, hipStream_t st)
{
    const int c4 = net->cs / 4;
    float *T = net->sc[head].T;
    static const bool no_pair = getenv("OJF_NO_PAIR") != nullptr;  // ablation switch only
    if (net->chain_dense) return launch_chain(net->chains[head], net->X[head], net->h, net->w, st);  // one launch (ojf_net_chain.h)
    if (!no_pair && (int)net->pairs[head].size() == net->gf) {  // one fused launch per Block (ojf_net_pair.h)
        for (int i = 0; i < net->gf; ++i)
            if (launch_pair(net->pairs[head][i], net->X[head], 0, net->X[head], (i + 1) * c4, net->h, net->w, st)) return -2;
        return 0;
    }
    for (int i = 0; i < net->gf; ++i) {
        if (launch_conv(net->dense[head][2 * i], net->X[head], 0, T, 0, nullptr, OJF_ACT_LEAKY, net->cs, 1.0f,
                        net->h, net->w, st)) return -2;
        if (launch_conv(net->dense[head][2 * i + 1], T, 0, net->X[head], (i + 1) * c4, nullptr, OJF_ACT_LEAKY,
                        net->cs, 1.0f, net->h, net->w, st)) return -2;
    }
    return 0;
}

static int layer_count(int version, int gf, int sem)
{
    const int pred = 2 * (gf - 1) + 3;
    if (version == 3) return (sem ? 2 : 1) * (2 * gf + 18) + 18 + pred;
    if (version == 2) return 2 * gf + 18 + 18 + pred;
    return -1;
}

}  // namespace ojf

OJF_API int ojf_net_layer_count(int version, int n_points, int growth, int use_semantics)
{
    if (n_points < 1 || growth < 1) return -1;
    return ojf::layer_count(version, growth, use_semantics ? 1 : 0);
}

namespace ojf { static void drop_graph(ojf_net *net); }

OJF_API void ojf_net_destroy(ojf_net *net)
{
    using namespace ojf;
    if (!net) return;
    for (int hd = 0; hd < 2; ++hd) {
        for (auto &pc : net->dense[hd]) release(pc);
        for (auto &pp : net->pairs[hd]) release(pp);
        release(net->chains[hd]);
    }
    for (auto &pc : net->pred) release(pc);
    for (auto &v : net->vortex) free_vortex(v);
    float *bufs[] = {net->X[0], net->X[1], net->YY, net->Y3, net->PA, net->PB};
    for (float *p : bufs) free_planes(p);
    for (auto &sc : net->sc) {
        float *sb[] = {sc.T, sc.Z, sc.Q0, sc.Q1, sc.Q2, sc.Q3, sc.U, sc.V, sc.partial, sc.CAT};
        if (sc.colsum) (void)hipFree(sc.colsum);
        for (float *p : sb) free_planes(p);
        if (sc.ev_fork) (void)hipEventDestroy(sc.ev_fork);
        if (sc.ev_join) (void)hipEventDestroy(sc.ev_join);
        if (sc.ev_entry) (void)hipEventDestroy(sc.ev_entry);
        if (sc.side) (void)hipStreamDestroy(sc.side);
    }
    if (net->colsum3) (void)hipFree(net->colsum3);
    if (net->ev_head_fork) (void)hipEventDestroy(net->ev_head_fork);
    if (net->ev_head_join) (void)hipEventDestroy(net->ev_head_join);
    if (net->head1) (void)hipStreamDestroy(net->head1);
    drop_graph(net);
    if (net->cap) (void)hipStreamDestroy(net->cap);
    if (net->chain_w) (void)hipFree(net->chain_w);
    if (net->chain_b) (void)hipFree(net->chain_b);
    delete net;
}

OJF_API int ojf_net_create(ojf_net **out, int version, int n_points, int growth, int use_semantics, float output_scale,
                           const ojf_conv_layer *L, int n_layers, int h, int w)
{
    using namespace ojf;
    if (!out || !L) return fail("ojf_net_create: null pointer argument");
    *out = nullptr;
    if (version != 2 && version != 3) return fail("ojf_net_create: version must be 2 or 3");
    if (n_points < 1 || growth < 1 || h <= 0 || w <= 0) return fail("ojf_net_create: bad sizes");
    const int sem = use_semantics ? 1 : 0;
    if (n_layers != layer_count(version, growth, sem)) return fail("ojf_net_create: wrong number of folded layers");
    for (int i = 0; i < n_layers; ++i)
        if (!L[i].weight_host || !L[i].bias_host || (L[i].ksize != 1 && L[i].ksize != 3) || L[i].dilation < 1)
            return fail("ojf_net_create: malformed layer descriptor");
    for (int i = 0; i < n_layers; ++i) {  // the split-fp16 range guard relies on finite parameters
        const size_t nw = (size_t)L[i].c_out * L[i].c_in * L[i].ksize * L[i].ksize;
        for (size_t k = 0; k < nw; ++k)
            if (!std::isfinite(L[i].weight_host[k])) return fail("ojf_net_create: non-finite weight");
        for (int k = 0; k < L[i].c_out; ++k)
            if (!std::isfinite(L[i].bias_host[k])) return fail("ojf_net_create: non-finite bias");
    }

    ojf_net *net = new ojf_net();
    net->version = version; net->P = n_points; net->gf = growth; net->sem = sem;
    net->arith = g_default_arith;
    net->c = 2 * n_points + 1 + (version == 2 ? sem : 0);
    net->cs = round_up(net->c, 4);
    net->heads = (version == 3 && sem) ? 2 : 1;
    net->h = h; net->w = w; net->npix = h * w;
    net->pool_in = net->c * (growth + 1);
    net->os = round_up(net->pool_in, 4);
    net->scale = output_scale;
    const int c = net->c, cs = net->cs, gf = growth, os = net->os;
    if (net->heads * os > 256 || (gf + 1) * cs > 256) { delete net; return fail("ojf_net_create: topology too wide"); }
    net->macs_per_pixel = 0;
    for (int i = 0; i < n_layers; ++i) net->macs_per_pixel += (int64_t)L[i].c_in * L[i].c_out * L[i].ksize * L[i].ksize;

    int rc = 0;
    int li = 0;
    auto build_dense = [&](int head) -> int {
        std::vector<ConvBuilder> bas, bbs;
        for (int i = 0; i < gf; ++i) {
            const ojf_conv_layer &la = L[li++], &lb = L[li++];
            if (la.ksize != 3 || lb.ksize != 3 || la.c_in != (i + 1) * c || la.c_out != c || lb.c_in != c || lb.c_out != c)
                return fail("ojf_net_create: dense block layer shape mismatch");
            ConvBuilder ba((i + 1) * cs, cs, 3, la.dilation), bb(cs, cs, 3, lb.dilation);
            ba.add(la, 0, (i + 1) * c, slot_map((i + 1) * c, c, cs), 0, true);
            bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
            PackedConv pa, pb;
            if (finish(ba, pa, net->arith) || finish(bb, pb, net->arith)) return -2;
            net->dense[head].push_back(pa);
            net->dense[head].push_back(pb);
            if (net->arith == OJF_ARITH_F16X3 && cs <= 24) {
                PackedPair pp;
                if (finish_pair(ba, bb, pp, pair_cfg_for(net->h, net->w, cs))) return -2;
                net->pairs[head].push_back(pp);
                bas.push_back(ba);
                bbs.push_back(bb);
            }
        }
        if ((int)bas.size() == gf && cs == 4 * kChainNG && gf <= kChainMaxLayers && finish_chain(bas, bbs, net->chains[head])) return -2;
        return 0;
    };
    const std::vector<int> dense_map = slot_map(net->pool_in, c, cs);
    const std::vector<int> flat_map = slot_map(net->pool_in, net->pool_in, os);
    if (version == 3) {
        rc = build_dense(0);
        if (!rc) { rc = build_vortex(net, net->vortex[0], L + li, net->pool_in, dense_map, (gf + 1) * cs); li += 18; }
        if (!rc && sem) {
            rc = build_dense(1);
            if (!rc) { rc = build_vortex(net, net->vortex[1], L + li, net->pool_in, dense_map, (gf + 1) * cs); li += 18; }
        }
        if (!rc) {
            rc = build_vortex(net, net->vortex[2], L + li, net->heads * net->pool_in,
                              slot_map(net->heads * net->pool_in, net->pool_in, os), net->heads * os);
            li += 18;
        }
    } else {
        rc = build_dense(0);
        if (!rc) { rc = build_vortex(net, net->vortex[0], L + li, net->pool_in, dense_map, (gf + 1) * cs); li += 18; }
        if (!rc) { rc = build_vortex(net, net->vortex[2], L + li, net->pool_in, flat_map, os); li += 18; }
    }
    if (!rc) {
        int prev_phys = os;
        for (; li < n_layers; ++li) {
            const ojf_conv_layer &l = L[li];
            if (l.ksize != 1) { rc = fail("ojf_net_create: prediction head must be 1x1 convolutions"); break; }
            if (round_up(l.c_in, 4) != prev_phys) { rc = fail("ojf_net_create: prediction head shapes do not chain"); break; }
            ConvBuilder b(prev_phys, round_up(l.c_out, 4), 1, 1);
            b.add(l, 0, l.c_in, slot_map(l.c_in, l.c_in, prev_phys), 0, true);
            PackedConv pc;
            if (finish(b, pc, net->arith)) { rc = -2; break; }
            net->pred.push_back(pc);
            prev_phys = round_up(l.c_out, 4);
        }
    }
    if (!rc && n_points == 9 && growth == 5 && (c == 19 || c == 20)) {
        // fused prediction head: repack the 11 pointwise layers as chain_layer fragments
        const int first = n_layers - (2 * (gf - 1) + 3);
        std::vector<float> cw, cb;
        int prev_phys = os;
        for (int l = first; l < n_layers; ++l) {
            const ojf_conv_layer &ly = L[l];
            const int nt_in = (prev_phys + 15) / 16, out_phys = round_up(ly.c_out, 4), nt_out = (out_phys + 15) / 16;
            auto wl = [&](int oc, int k) { return oc < ly.c_out && k < ly.c_in ? ly.weight_host[(size_t)oc * ly.c_in + k] : 0.0f; };
            const std::vector<float> rs = chain_row_scales(net->arith, nt_out, ly.c_in, wl);
            pack_chain_layer(cw, net->arith, nt_out, nt_in, rs, wl);
            append_chain_bias(cb, net->arith, nt_out, ly.c_out, ly.bias_host, rs);
            prev_phys = out_phys;
        }
        if (upload(cw, &net->chain_w) || upload(cb, &net->chain_b)) rc = -2;
        else net->chain_kind = c;
    }
    if (!rc) {  // the gave convs act on a 1x1 map: not a per-pixel cost
        int idx = 0;
        auto skip_dense = [&]() { idx += 2 * gf; };
        auto sub_gave = [&]() { net->macs_per_pixel -= (int64_t)L[idx].c_in * L[idx].c_out; idx += 18; };
        skip_dense(); sub_gave();
        if (version == 3 && sem) { skip_dense(); sub_gave(); }
        sub_gave();
    }
    if (!rc) {
        // One dense_chain_kernel launch per head instead of gf dense_pair_kernel launches: the dense-growth buffers then hold
        // split planes, which only the chain-flow entry kernel reads (OJF_NO_DENSE_CHAIN=1 / the legacy flow: pair kernels)
        static const bool no_chain = getenv("OJF_NO_DENSE_CHAIN") != nullptr || getenv("OJF_LEGACY_VORTEX") != nullptr ||
                                     getenv("OJF_NO_PAIR") != nullptr;
        bool ok = net->arith == OJF_ARITH_F16X3 && !no_chain;
        {
            //  frames the chain kernel is not the right tool for fall back to one dense_pair_kernel launch per Block:
            //  * more tiles than the flag array holds (frames beyond ~2.6 Mpixel used to fail in ojf_net_forward);
            //  * w % 8 != 0: a 128-byte line of a plane is 8 pixels, so a line then straddles two image rows - pixels of tiles
            //    whose flags a reader never checked - and only the slot of the Block just finished is read with sc1 loads: an
            //    early reader could pull a not-yet-written pixel of an OLDER slot into its XCD's L2 and a later item read it stale.
            const int big = ((w + 19) / 20) * ((h + 15) / 16);
            const int tiles = big >= 200 ? big : ((w + 11) / 12) * ((h + 7) / 8);
            if (tiles > kChainSyncInts - kChainFlags0 || (w % 8) != 0) ok = false;
        }
        for (int hd = 0; hd < net->heads && ok; ++hd)
            ok = net->chains[hd].layers == gf && net->vortex[hd].entry_w && net->vortex[hd].tail_w;
        net->chain_dense = ok;
    }
    const size_t np = (size_t)net->npix;
    if (!rc) rc = alloc_planes(&net->X[0], np, (gf + 1) * cs);
    if (!rc && net->heads == 2) rc = alloc_planes(&net->X[1], np, (gf + 1) * cs);
    for (int hd = 0; hd < net->heads && !rc; ++hd) {
        ojf_net::Scratch &sc = net->sc[hd];
        rc = alloc_planes(&sc.T, np, cs);
        if (!rc) rc = alloc_planes(&sc.Z, np, 4 * cs);
        if (!rc && hd == 0 && net->heads == 2) rc = alloc_planes(&sc.Q0, np, cs);
        if (!rc) rc = alloc_planes(&sc.Q1, np, cs);
        if (!rc) rc = alloc_planes(&sc.Q2, np, cs);
        if (!rc) rc = alloc_planes(&sc.Q3, np, cs);
        if (!rc) rc = alloc_planes(&sc.U, np, 4 * cs);
        if (!rc) rc = alloc_planes(&sc.V, np, 4 * cs);
        if (!rc) rc = alloc_planes(&sc.partial, kSumBlocks, 256);
        if (!rc) rc = check_hip(hipMalloc(reinterpret_cast<void **>(&sc.colsum), sizeof(ColSums)), "hipMalloc");
        if (!rc) rc = check_hip(hipMemset(sc.colsum, 0, sizeof(ColSums)), "hipMemset");
        if (!rc && hd == 0 && net->heads == 2) {
            rc = check_hip(hipMalloc(reinterpret_cast<void **>(&net->colsum3), sizeof(ColSums)), "hipMalloc");
            if (!rc) rc = check_hip(hipMemset(net->colsum3, 0, sizeof(ColSums)), "hipMemset");
            if (!rc) rc = check_hip(hipStreamSynchronize(nullptr), "hipStreamSynchronize");  // (the heads' streams are non-blocking: see alloc_planes)
        }
        if (!rc) rc = check_hip(hipStreamSynchronize(nullptr), "hipStreamSynchronize");  // (see alloc_planes)
        if (!rc) rc = check_hip(hipStreamCreateWithFlags(&sc.side, hipStreamNonBlocking), "hipStreamCreate");
        if (!rc) rc = check_hip(hipEventCreateWithFlags(&sc.ev_fork, event_flags()), "hipEventCreate");
        if (!rc) rc = check_hip(hipEventCreateWithFlags(&sc.ev_join, event_flags()), "hipEventCreate");
        if (!rc) rc = check_hip(hipEventCreateWithFlags(&sc.ev_entry, event_flags()), "hipEventCreate");
    }
    if (!rc && net->heads == 2) {
        rc = check_hip(hipStreamCreateWithFlags(&net->head1, hipStreamNonBlocking), "hipStreamCreate");
        if (!rc) rc = check_hip(hipEventCreateWithFlags(&net->ev_head_fork, event_flags()), "hipEventCreate");
        if (!rc) rc = check_hip(hipEventCreateWithFlags(&net->ev_head_join, event_flags()), "hipEventCreate");
    }
    if (!rc) rc = alloc_planes(&net->YY, np, net->heads * os);
    if (!rc) rc = alloc_planes(&net->Y3, np, os);
    if (!rc) rc = check_hip(hipStreamCreateWithFlags(&net->cap, hipStreamNonBlocking), "hipStreamCreate");
    {
        const char *g = getenv("OJF_NET_GRAPH");  // opt-in: see the note at ojf_net::cap
        net->use_graph = g && g[0] == '1';
    }
    if (rc) {
        const std::string keep = ojf_last_error();
        ojf_net_destroy(net);
        set_error(keep);
        return rc;
    }
    *out = net;
    return 0;
}

namespace ojf {
int net_input_slot(::ojf_net *net, NetInputSlot *slot)
{
    if (!net || !slot || net->sem || net->heads != 1) return 1;
    slot->x0 = net->X[0]; slot->cs4 = net->cs / 4; slot->P = net->P; slot->h = net->h; slot->w = net->w;
    slot->ovf = net->arith == OJF_ARITH_F16X3 ? overflow_flag() : nullptr;
    slot->split = net->chain_dense ? 1 : 0;
    return 0;
}
}  // namespace ojf

OJF_API int ojf_net_prepare_input(ojf_net *net, const float *values, const float *weights, int rows_stride,
                                  int in_layout, const float *depth, const uint8_t *sem_ids, int n_classes, ojf_stream_t stream)
{
    using namespace ojf;
    if (!net || !values || !weights || !depth) return fail("ojf_net_prepare_input: null pointer argument");
    if (in_layout != 0 && in_layout != 1) return fail("ojf_net_prepare_input: in_layout must be 0 or 1");
    if (in_layout == 0 && rows_stride < net->P) return fail("ojf_net_prepare_input: rows_stride < n_points");
    if (in_layout == 1 && rows_stride < net->npix) return fail("ojf_net_prepare_input: plane stride < h*w");
    if (net->sem && (!sem_ids || n_classes <= 0))
        return fail("ojf_net_prepare_input: this net uses semantics: sem_ids and n_classes are required");
    PrepArgs a;
    a.values = values; a.weights = weights; a.depth = depth;
    a.sem = net->sem ? sem_ids : nullptr;
    a.x0 = planes(net->X[0]);
    a.x1 = net->heads == 2 ? planes(net->X[1]) : nullptr;
    a.rows_stride = rows_stride; a.in_layout = in_layout; a.npix = net->npix; a.P = net->P; a.cs4 = net->cs / 4;
    a.n_classes = n_classes;
    a.v2_sem = (net->version == 2 && net->sem) ? 1 : 0;
    a.ovf = net->arith == OJF_ARITH_F16X3 ? overflow_flag() : nullptr;
    a.split = net->chain_dense ? 1 : 0;
    hipLaunchKernelGGL(prepare_input_kernel, dim3((a.npix + 255) / 256), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "prepare_input_kernel launch");
}

namespace ojf {

// Two HIP streams only run concurrently when the runtime maps them to different hardware queues (a small round-robin
// pool): a second-head stream that aliases the launch stream's queue silently serialises the two heads (measured:
// net stage 0.60 -> 0.70 ms for the two-head net when the library is used by the third engine of a process).  The
// pairing is therefore TESTED once per (net, launch stream): two 150 us spin kernels, one per stream, take ~150 us
// when the streams overlap and ~300 us when they do not; up to 8 fresh streams are tried per role.
__global__ void spin_kernel(long long cycles)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

static double spin_pair_us(hipStream_t a, hipStream_t b, long long cycles)
{
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    // a's spin is followed by a kernel that DEPENDS on it: two streams that share a hardware queue run independent kernels
    // side by side, but the barrier packet in front of a dependent one holds back every later packet of that queue, the
    // other stream's included (round 5: a look-ahead stream that passed the two-spins test shared the queue of the
    // second head's stream and ran 20 % slower than one that did not)
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, cycles);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 1LL);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, cycles);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// makes `s` a stream that overlaps with every stream of `others` (replaces it by fresh ones until the test passes)
static void pair_stream(hipStream_t &s, const hipStream_t *others, int n, double alone, long long cycles, const char *what)
{
    static const bool dbg = getenv("OJF_DEBUG_STREAMS") != nullptr;
    std::vector<hipStream_t> rejected;
    for (int attempt = 0; attempt < 8; ++attempt) {
        double worst = 0.0;
        for (int k = 0; k < n; ++k) {
            const double both = spin_pair_us(others[k], s, cycles);
            worst = both > worst ? both : worst;
        }
        if (dbg) fprintf(stderr, "ojf: %s stream, attempt %d: one spin %.0f us, one per stream %.0f us\n", what, attempt, alone, worst);
        if (worst < 1.5 * alone) break;  // overlaps with all of them
        // the replacement is created while the rejected streams still exist: a stream created right after a destroy
        // gets the freed queue slot again
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        rejected.push_back(s);
        s = fresh;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
}

}  // namespace ojf

OJF_API int ojf_streams_overlap(ojf_stream_t a, ojf_stream_t b)
{
    using namespace ojf;
    hipStream_t sa = as_stream(a), sb = as_stream(b);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(sa, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return fail("ojf_streams_overlap: stream a is capturing");
    if (hipStreamIsCapturing(sb, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return fail("ojf_streams_overlap: stream b is capturing");
    const long long cycles = 300000;  // ~130 us
    (void)spin_pair_us(sa, sb, 1000);  // (the first launch of the kernel in a process loads its code object)
    double alone = 1e30, both = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        const double t1 = spin_pair_us(sa, sa, cycles) * 0.5;  // two spins back to back on one stream
        const double t2 = spin_pair_us(sa, sb, cycles);
        alone = t1 < alone ? t1 : alone;
        both = t2 < both ? t2 : both;
    }
    if (hipGetLastError() != hipSuccess) return fail("ojf_streams_overlap: spin kernel launch failed");
    return both < 1.5 * alone ? 1 : 0;
}

namespace ojf {

static void pair_head_stream(ojf_net *net, hipStream_t st)
{
    if (net->heads != 2 || (net->head_paired && net->head_paired_with == st) || !net->head1) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;  // not inside a capture
    net->head_paired_with = st;
    net->head_paired = true;
    const long long cycles = 300000;  // ~130 us
    (void)spin_pair_us(st, st, 1000);  // (the first launch of the kernel in a process loads its code object: not in the yardstick)
    const double alone = spin_pair_us(st, st, cycles) * 0.5;  // two spins back to back on one stream
    pair_stream(net->head1, &st, 1, alone, cycles, "second head");
    // the side streams of the legacy VortexPooling flow (global-average branch, branch 0): each must overlap with the
    // stream of ITS head (only three queues ran concurrently in every test, so four-way distinctness is not asked for)
    if (net->sc[1].side) pair_stream(net->sc[1].side, &net->head1, 1, alone, cycles, "head 1 side");
    if (net->sc[0].side) pair_stream(net->sc[0].side, &st, 1, alone, cycles, "head 0 side");
}

// every launch of one forward pass, in order, on `st` (+ the side stream, forked and joined with events)
static int forward_launches(ojf_net *net, float *est, int est_stride, hipStream_t st)
{
    const int o4 = net->os / 4;
    // prediction head: 1x1 chain; BN-folded LeakyReLU stages, the last layer is Tanh * output_scale
    static const bool unfused = getenv("OJF_NO_CHAIN") != nullptr;  // ablation switch only
    ChainArgs ca;
    ca.in = planes(net->Y3); ca.w = planes(net->chain_w); ca.bias = net->chain_b;
    ca.out_rows = est; ca.in_g0 = 0; ca.c4_in = o4; ca.npix = net->npix;
    ca.rows_stride = est_stride; ca.rows_n = net->P; ca.scale = net->scale;
    ca.ovf = net->arith == OJF_ARITH_F16X3 ? overflow_flag() : nullptr;
    const ChainArgs *head = (net->chain_kind && !unfused) ? &ca : nullptr;
    bool head_done = false;  // set when the last VortexPooling's tail kernel ran the head as well
    static const bool serial_heads = getenv("OJF_SERIAL_HEADS") != nullptr;  // ablation switch only
    if (net->version == 3) {
        const bool two = net->heads == 2;
        if (two && !serial_heads) pair_head_stream(net, st);
        hipStream_t s1 = (two && !serial_heads) ? net->head1 : st;
        if (two && s1 != st) {  // the semantic head runs beside the geometric one
            OJF_HIP(hipEventRecord(net->ev_head_fork, st));
            OJF_HIP(hipStreamWaitEvent(s1, net->ev_head_fork, 0));
        }
        // two heads: each head's tail runs ITS half of vortex3's entry GEMM (the concatenation YY is then never written)
        const bool halves = two && net->vortex[2].entry_wh[0] && net->vortex[2].entry_wh[1] && net->sc[0].Q0 && net->colsum3;
        bool done1 = false;
        if (two) {
            if (run_dense(net, 1, s1)) return -2;
            if (run_vortex(net, net->vortex[1], net->X[1], 0, net->YY, o4, s1, net->sc[1], nullptr, nullptr, false,
                           halves ? &net->vortex[2] : nullptr, &done1, net->chain_dense, halves ? 1 : -1, net->colsum3)) return -2;
        }
        if (run_dense(net, 0, st)) return -2;
        bool entry_done = false;  // single head: vortex0's tail runs vortex3's entry GEMM (its own output is never written)
        if (run_vortex(net, net->vortex[0], net->X[0], 0, net->YY, 0, st, net->sc[0], nullptr, nullptr, false,
                       (two && !halves) ? nullptr : &net->vortex[2], &entry_done, net->chain_dense, halves ? 0 : -1, halves ? net->colsum3 : net->sc[0].colsum)) return -2;
        if (two && s1 != st) {
            OJF_HIP(hipEventRecord(net->ev_head_join, s1));
            OJF_HIP(hipStreamWaitEvent(st, net->ev_head_join, 0));
        }
        if (halves && entry_done != done1) return fail("ojf_net_forward: internal error (one head ran its half of the entry GEMM, the other did not)");
        if (run_vortex(net, net->vortex[2], net->YY, 0, net->Y3, 0, st, net->sc[0], head, &head_done, entry_done, nullptr, nullptr, false, -1,
                       nullptr, (halves && entry_done) ? net->sc[1].Z : nullptr)) return -2;
    } else {
        if (run_dense(net, 0, st)) return -2;
        bool entry_done = false;
        if (run_vortex(net, net->vortex[0], net->X[0], 0, net->YY, 0, st, net->sc[0], nullptr, nullptr, false, &net->vortex[2],
                       &entry_done, net->chain_dense)) return -2;
        if (run_vortex(net, net->vortex[2], net->YY, 0, net->Y3, 0, st, net->sc[0], head, &head_done, entry_done)) return -2;
    }
    if (head_done) return 0;
    if (net->chain_kind && !unfused) {
        // stand-alone head (the tail fusion is off or unavailable); one pixel tile per wave: two measured slower
        const bool h16 = net->arith == OJF_ARITH_F16X3;
        const dim3 grid(chain_blocks(net)), block(ojf::kChainThreads);
        if (net->chain_kind == 19 && h16)
            hipLaunchKernelGGL((chain1x1_kernel<OJF_ARITH_F16X3, 1, 8, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1>), grid, block, 0, st, ca);
        else if (net->chain_kind == 19)
            hipLaunchKernelGGL((chain1x1_kernel<OJF_ARITH_F32, 1, 8, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1>), grid, block, 0, st, ca);
        else if (h16)
            hipLaunchKernelGGL((chain1x1_kernel<OJF_ARITH_F16X3, 1, 8, 7, 7, 5, 5, 4, 4, 3, 3, 2, 2, 1>), grid, block, 0, st, ca);
        else
            hipLaunchKernelGGL((chain1x1_kernel<OJF_ARITH_F32, 1, 8, 7, 7, 5, 5, 4, 4, 3, 3, 2, 2, 1>), grid, block, 0, st, ca);
        mark_launch("chain1x1_kernel", st);
        return check_hip(hipGetLastError(), "chain1x1_kernel launch");
    }
    const float *pin = net->Y3;
    const int np = (int)net->pred.size();
    if (ensure_planes(&net->PA, (size_t)net->npix, net->os) || ensure_planes(&net->PB, (size_t)net->npix, net->os)) return -2;
    float *pp[2] = {net->PA, net->PB};
    for (int i = 0; i < np; ++i) {
        const PackedConv &pc = net->pred[i];
        const bool last = (i == np - 1);
        if (last) {  // exactly n_points channels into the caller's row-major est
            if (launch_conv(pc, pin, 0, nullptr, 0, nullptr, OJF_ACT_TANH, pc.n_ot * 16, net->scale, net->h, net->w, st,
                            est, est_stride, net->P)) return -2;
        } else {
            if (launch_conv(pc, pin, 0, pp[i & 1], 0, nullptr, OJF_ACT_LEAKY, pc.n_ot * 16, 1.0f, net->h, net->w, st))
                return -2;
            pin = pp[i & 1];
        }
    }
    return 0;
}

static void drop_graph(ojf_net *net)
{
    if (net->gexec) (void)hipGraphExecDestroy(net->gexec);
    if (net->graph) (void)hipGraphDestroy(net->graph);
    net->gexec = nullptr;
    net->graph = nullptr;
}

// Records the ~30 launches of a forward pass (both streams) once per (est, est_stride) into a hipGraph.
// The capture runs on the net's own stream: the caller's stream may be the legacy default stream, which
// cannot be captured but can launch a graph.  A failed capture disables graphs for this net (plain launches).
static void capture_graph(ojf_net *net, float *est, int est_stride)
{
    drop_graph(net);
    net->g_est = est;
    net->g_stride = est_stride;
    // no allocation may happen while capturing: have the fall-back paths' buffers in place
    if (ensure_planes(&net->sc[0].CAT, (size_t)net->npix, 4 * net->os) ||
        (net->heads == 2 && ensure_planes(&net->sc[1].CAT, (size_t)net->npix, 4 * net->os)) ||
        ensure_planes(&net->PA, (size_t)net->npix, net->os) ||
        ensure_planes(&net->PB, (size_t)net->npix, net->os)) { net->use_graph = false; return; }
    if (hipStreamBeginCapture(net->cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { net->use_graph = false; return; }
    const int rc = forward_launches(net, est, est_stride, net->cap);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(net->cap, &g);
    if (rc || e != hipSuccess || !g || hipGraphInstantiate(&net->gexec, g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        net->gexec = nullptr;
        net->use_graph = false;
        (void)hipGetLastError();
        return;
    }
    net->graph = g;
}

}  // namespace ojf

OJF_API int ojf_net_forward(ojf_net *net, float *est, int est_stride, ojf_stream_t stream)
{
    using namespace ojf;
    if (!net || !est) return fail("ojf_net_forward: null pointer argument");
    if (est_stride < net->P) return fail("ojf_net_forward: est_stride < n_points");
    hipStream_t st = as_stream(stream);
    if (net->arith == OJF_ARITH_F16X3 && g_ovf_host && *g_ovf_host) return fail(*g_ovf_host == 2 ? kChainStuckMsg : kOverflowMsg);
    if (net->use_graph) {
        if (!net->gexec || net->g_est != est || net->g_stride != est_stride) capture_graph(net, est, est_stride);
        if (net->gexec) return check_hip(hipGraphLaunch(net->gexec, st), "hipGraphLaunch (net forward)");
    }
    const int before = g_launches;
    const int rc = forward_launches(net, est, est_stride, st);
    net->last_launches = g_launches - before;
    return rc;
}

OJF_API int64_t ojf_net_macs_per_pixel(const ojf_net *net) { return net ? net->macs_per_pixel : -1; }

OJF_API int ojf_net_launch_count(const ojf_net *net) { return net ? net->last_launches : -1; }

OJF_API int ojf_net_side_streams(ojf_net *net, ojf_stream_t stream, ojf_stream_t out[3])
{
    using namespace ojf;
    if (!net || !out) return fail("ojf_net_side_streams: null pointer argument");
    pair_head_stream(net, as_stream(stream));  // (what the first forward pass on `stream` would do; a no-op for one head / inside a capture)
    out[0] = net->heads == 2 ? reinterpret_cast<ojf_stream_t>(net->head1) : nullptr;
    out[1] = reinterpret_cast<ojf_stream_t>(net->sc[0].side);
    out[2] = net->heads == 2 ? reinterpret_cast<ojf_stream_t>(net->sc[1].side) : nullptr;
    return (out[0] != nullptr) + (out[1] != nullptr) + (out[2] != nullptr);
}

OJF_API int ojf_net_profile(ojf_net *net, float *est, int est_stride, ojf_stream_t stream, char *names, int names_cap,
                            float *micros, int max_entries)
{
    using namespace ojf;
    if (!net || !est || !names || !micros || names_cap < 1 || max_entries < 1) return fail("ojf_net_profile: bad argument");
    hipStream_t st = as_stream(stream);
    hipEvent_t e0 = nullptr;
    OJF_HIP(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
    g_marks.clear();
    g_profile = true;
    (void)hipEventRecord(e0, st);
    const int rc = forward_launches(net, est, est_stride, st);
    g_profile = false;
    int n = 0;
    size_t used = 0;
    names[0] = 0;
    if (!rc && hipStreamSynchronize(st) == hipSuccess) {
        for (size_t i = 0; i < g_marks.size() && n < max_entries; ++i) {
            // previous mark on the same stream (the start mark for the first launch of the caller's stream)
            hipEvent_t prev = nullptr;
            for (size_t j = i; j-- > 0;)
                if (g_marks[j].st == g_marks[i].st) { prev = g_marks[j].ev; break; }
            if (!prev && g_marks[i].st == st) prev = e0;
            float ms = 0.0f;
            (void)hipEventSynchronize(g_marks[i].ev);
            if (!prev || hipEventElapsedTime(&ms, prev, g_marks[i].ev) != hipSuccess) ms = -1e-3f;  // first launch of a side stream
            const size_t len = strlen(g_marks[i].name);
            if (used + len + 2 > (size_t)names_cap) break;
            memcpy(names + used, g_marks[i].name, len);
            names[used + len] = '\n';
            used += len + 1;
            names[used] = 0;
            micros[n++] = ms * 1e3f;
        }
    }
    for (auto &m : g_marks) (void)hipEventDestroy(m.ev);
    g_marks.clear();
    (void)hipEventDestroy(e0);
    return rc ? rc : n;
}

OJF_API int ojf_net_set_arithmetic(int arithmetic)
{
    if (arithmetic != OJF_ARITH_F32 && arithmetic != OJF_ARITH_F16X3)
        return ojf::fail("ojf_net_set_arithmetic: unknown arithmetic (OJF_ARITH_F32 or OJF_ARITH_F16X3)");
    ojf::g_default_arith = arithmetic;
    return 0;
}

OJF_API int ojf_net_get_arithmetic(const ojf_net *net) { return net ? net->arith : ojf::g_default_arith; }

OJF_API int ojf_net_check(ojf_stream_t stream)
{
    using namespace ojf;
    OJF_HIP(hipStreamSynchronize(as_stream(stream)));
    if (const int what = guard_take(nullptr, true)) return fail(what == 2 ? kChainStuckMsg : kOverflowMsg);
    return 0;
}

OJF_API int ojf_guard_poll(void) { return ojf::g_ovf_host ? *ojf::g_ovf_host : 0; }

OJF_API int ojf_guard_status(ojf_stream_t stream, int *flag, int *skipped)
{
    using namespace ojf;
    OJF_HIP(hipStreamSynchronize(as_stream(stream)));
    const int what = guard_take(skipped, false);
    if (flag) *flag = what;
    return 0;
}

OJF_API int ojf_conv2d(const float *in, int in_stride, int in_off, float *out, int out_stride, int out_off,
                       const ojf_conv_layer *layer, int act, int h, int w, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || !layer || !layer->weight_host || !layer->bias_host) return fail("ojf_conv2d: null pointer argument");
    if ((layer->ksize != 1 && layer->ksize != 3) || layer->dilation < 1) return fail("ojf_conv2d: 1x1 or 3x3 kernels only");
    const int cin_phys = round_up(layer->c_in, 4), cout_phys = round_up(layer->c_out, 4);
    if (in_off + layer->c_in > in_stride || out_off + cout_phys > out_stride)
        return fail("ojf_conv2d: channel window exceeds the row stride");
    hipStream_t st = as_stream(stream);
    const int npix = h * w;
    ConvBuilder b(cin_phys, cout_phys, layer->ksize, layer->dilation);
    b.add(*layer, 0, layer->c_in, slot_map(layer->c_in, layer->c_in, cin_phys), 0, true);
    PackedConv pc;
    if (finish(b, pc, g_default_arith)) return -2;
    float *pin = nullptr, *pout = nullptr;
    int rc = alloc_planes(&pin, npix, cin_phys);
    if (!rc) rc = alloc_planes(&pout, npix, cout_phys);
    if (!rc) {
        hipLaunchKernelGGL(rows_to_planes_kernel, dim3((npix * (cin_phys / 4) + 255) / 256), dim3(256), 0, st, in, in_stride,
                           in_off, layer->c_in, planes(pin), cin_phys / 4, npix);
        rc = launch_conv(pc, pin, 0, pout, 0, nullptr, act, cout_phys, 1.0f, h, w, st);
    }
    if (!rc) {
        hipLaunchKernelGGL(planes_to_rows_kernel, dim3((npix * (cout_phys / 4) + 255) / 256), dim3(256), 0, st,
                           planes(pout), cout_phys / 4, npix, out, out_stride, out_off);
        rc = check_hip(hipStreamSynchronize(st), "ojf_conv2d sync");  // test-only API: packs per call
        if (!rc && guard_take(nullptr, true)) rc = fail(kOverflowMsg);
    }
    free_planes(pin);
    free_planes(pout);
    release(pc);
    return rc;
}

#include "ojf_net_train.h"
#include "ojf_train_net.h"
