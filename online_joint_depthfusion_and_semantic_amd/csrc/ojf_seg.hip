// SEGCONV: the convolutions of the 2-D semantic front-end (AdapNet++, modules/adapnet.py: ResNet-50 bottlenecks :101-130,
// multi-scale units :12-84, eASPP :152-216, SSMA :320-354, decoder 3x3 stacks :219-317) as ONE implicit-GEMM kernel
// family on the matrix cores, with what follows each of them in the reference folded into the launch: BatchNorm
// (eval mode, folded into the weights and a bias), the residual add, ReLU / sigmoid.
//
// Layout: activations are NHWC fp32 (torch channels_last, batch 1): a pixel's channels are contiguous, so a lane's
// B fragment of v_mfma_f32_16x16x32_f16 - 8 consecutive K values = 8 consecutive input channels of one tap - is two
// 16-byte loads.  GEMM view: D[c_out][pixel] = sum_k W[c_out][k] X[k][pixel], k = (tap, channel); weights are the A
// operand, packed on the host in fragment order as split fp16 halves (same arithmetic as the fusion net, see
// ojf_net.hip: x*w = wl*xh + wh*xl + wh*xh, fp32 accumulate, rows equilibrated by a power of two).  A lane's
// accumulator holds 4 consecutive output channels of one pixel: one 16-byte NHWC store.
//
// Three shapes of the same loop (ojf_segconv_forward picks by the size of the layer):
//   * many independent waves: a wave owns 64 output channels x 32 pixels, 4 waves per block over (channels x pixels);
//   * few pixels (15x20 .. 60x80 maps): the 4 waves of a block share one (channels, pixels) pair and split K four
//     ways (LDS reduction in fixed order); 16 / 32 / 64 channels per block so that at least ~150 blocks exist;
//   * many pixels (>= 256 blocks of 64 channels x 128 pixels): segconv_wide_kernel, weights through LDS (LDS-DMA).
// Operands come straight from global memory through L1 (buffer loads; out-of-image taps are out-of-range offsets that
// return zeros), three K blocks in flight per wave in a branch-free loop.  Channel slices of a wider tensor
// (concatenations) are addressed by pointer + row stride.  Transposed convolutions: see ojf_segdeconv_create.
// Tuning-only environment switches: OJF_SEG_MW, OJF_SEG_NO_WIDE, OJF_SEG_WIDE_MIN.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ojf_common.h"

#ifndef OJF_GEMM_ABL
#define OJF_GEMM_ABL 0
#endif

namespace ojf {

int *range_flag_device();  // ojf_net.hip: the split-fp16 range guard flag (host-mapped), shared by all kernels

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, f16x8 &hi, f16x8 &lo)
{
    const f32x8 x = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    hi = __builtin_convertvector(x, f16x8);
    const u32x4 h = __builtin_bit_cast(u32x4, hi);
    u32x4 l;
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = ojf::split_lo_pair(h[i], x[2 * i], x[2 * i + 1]);  // exact remainder, rounded once
    lo = __builtin_bit_cast(f16x8, l);
}

__device__ __forceinline__ f32x4 mfma3(const f32x4 &wh, const f32x4 &wl, const f16x8 &xh, const f16x8 &xl, f32x4 acc)
{
    const f16x8 h = __builtin_bit_cast(f16x8, wh), l = __builtin_bit_cast(f16x8, wl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xh, acc, 0, 0, 0);
}

constexpr int kMW = 4;  // output-channel tiles per wave

struct SegArgs {
    const float *in;
    float *out;
    const float *res;      // residual rows added before the activation, or NULL
    const float *mul;      // rows multiplied in after the activation (SSMA gate), or NULL
    const f32x4 *wp;       // [c_out tile][K block][hi, lo][lane]
    const float *rinv;     // [tiles*16] inverse row scales
    const float *bias;     // [tiles*16]
    int in_stride, out_stride, res_stride, mul_stride;  // floats per pixel row
    int H, W, Ho, Wo;
    int B;                 // images per tensor ([B, H, W, C] rows, batch-major): a pixel index runs over B * Ho * Wo
    int stride, pad, dil, ksize;
    int c8;                // groups of 8 input channels per tap
    int n_kb;              // K blocks of 4 (tap, group) entries
    int n_ct;              // packed output-channel tiles (multiple of kMW)
    int c_out;
    int act;               // 0 none, 1 ReLU, 2 sigmoid
    int vec_store;         // rows are 16-byte aligned: bit 0 float4 stores, bit 1 float4 residual loads, bit 2 float4 gate loads
    unsigned in_bytes;
    int *ovf;
    int up, up_c, up_cp;   // transposed convolution: upscale factor (1: none), real / padded channels per phase
    int pad_to;            // channels c_out .. pad_to - 1 of every output row are written as zeros (OJF_SEG_ACT_ZERO_PAD), else = c_out
    const unsigned long long *rng;  // always-on dropout of a multi-scale unit (adapnet.py:80-82): {seed, frame counter}, or NULL
    unsigned long long *rng_bump;   // this launch advances the frame counter (the last convolution of a forward pass), or NULL
    unsigned drop_id;               // stream id of this layer inside a frame
};

// Philox-4x32-10 (Salmon et al., the generator torch's own dropout draws from): counter-based, so the mask of element e of
// layer `id` in frame f is a pure function of (seed, f, id, e) - no state to carry between launches, replayable in a graph.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = uint4{hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0};
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
// Up to kSegGroup convolutions of the same shape (channels, kernel size, stride, frame: the launch geometry) run as ONE
// launch, blockIdx.z = member: the two modality encoders of AdapNet++ in lock-step, the two dilations of a multi-scale
// unit, the three cascades of an eASPP - half the graph nodes of the front end and no cross-stream fork / join.
constexpr int kSegGroup = 8;
// Block -> (pixel block x, channel block y, member z) of a launch with the logical grid X x Y x Z, launched as a 1-D grid.
// Block b runs on XCD b % 8 (ojf_common.h) and every XCD has its own L2: with the plain (x, y, z) numbering the 19 pixel
// blocks of a channel block on a 15x20 map land on all eight XCDs and every XCD pulls every weight through the fabric -
// measured (round 5, rocprofv3 FETCH_SIZE per launch): 190 MB fetched by a layer4 shortcut whose weights are 17 MB, and
// its 37 us are those bytes at ~5 TB/s.  Here a (channel block, member) pair - the unit that shares weights - lives on
// ONE XCD: pair q = y + Y z goes to XCD q % 8 with all its pixel blocks.  Few pairs (Q = Y Z not a multiple of 8): the
// pixel blocks of a pair are cut into S contiguous chunks ("virtual pairs" v = q S + s, V = Q S a multiple of 8 where S
// <= 8 allows) so that all XCDs work.  Padding blocks (v >= V or x >= X) exit.  Speed only: any placement computes the same.
struct SegMap { int X, Y, Z, S, chunk, V; };
// (heterogeneous launches - ojf_segconv_forward_multi - : every member has its own map; member z owns the blocks
// [off[z], off[z + 1]) of the 1-D grid, each range a multiple of 8 so that a block's XCD is the same in both numberings)
struct SegHetero { int n; int off[kSegGroup + 1]; SegMap map[kSegGroup]; };
struct SegGroupArgs { SegArgs a[kSegGroup]; SegMap map; SegHetero het; int abl; };  // abl: ablation bits of segconv_gemm_kernel (tools/ only, 0 in production)

__device__ __forceinline__ bool seg_block_map(const SegMap &m, int b, int &bx, int &by, int &bz);

__device__ __forceinline__ bool seg_block(const SegGroupArgs &g, int &bx, int &by, int &bz)
{
    if (g.het.n == 0) return seg_block_map(g.map, (int)blockIdx.x, bx, by, bz);
    int z = 0;
#pragma unroll
    for (int i = 1; i < kSegGroup; ++i)
        if (i < g.het.n && (int)blockIdx.x >= g.het.off[i]) z = i;
    int zz;
    const bool ok = seg_block_map(g.het.map[z], (int)blockIdx.x - g.het.off[z], bx, by, zz);
    bz = z;
    return ok;
}

__device__ __forceinline__ bool seg_block_map(const SegMap &m, int b, int &bx, int &by, int &bz)
{
    if (m.S == 0) {  // plain numbering (OJF_SEG_XCD=0: A/B switch)
                bx = b % m.X;
        const int q = b / m.X;
        by = q % m.Y;
        bz = q / m.Y;
        return true;
    }
    const int xcd = b & 7, slot = b >> 3;
    const int vl = slot / m.chunk, xl = slot - vl * m.chunk;
    const int v = vl * 8 + xcd;
    if (v >= m.V) return false;
    const int q = v / m.S, sp = v - q * m.S;
    bx = sp * m.chunk + xl;
    by = q % m.Y;
    bz = q / m.Y;
    return bx < m.X;
}


// KS = 1: the 4 waves of a block take different (channel group, pixel tiles) pairs: WM along the channels.
// KS = 4 | 8 (layers with few pixels: too few waves to hide the weight stream otherwise): the KS waves of a block share
// ONE pair and each walks 1/KS of K; partial sums meet in LDS and wave 0 runs the epilogue.  Fixed order: deterministic.
// Measured on the 15x20 maps of layer3/4: parallelism beats operand reuse (NW = 1: 2.33 ms per frame, 2: 2.69, 4: 3.69).
// the epilogue's per-channel vectors: requested before the K loop (seg_vectors), so that their L2 round trip does not
// sit at the end of a kernel that is all latency (the small layers run 10 us, a dependent chain of ~95 per frame):
// 1.58 -> 1.49 ms per replayed forward pass.  (Requesting the residual tile up front as well changed nothing.)
template <int MW>
__device__ __forceinline__ void seg_vectors(const SegArgs &a, int ct0, int kg, f32x4 (&rv)[MW], f32x4 (&bv)[MW])
{
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int c = (ct0 + m) * 16 + kg * 4;
        const bool ok = c < a.c_out;
        rv[m] = *reinterpret_cast<const f32x4 *>(a.rinv + (ok ? c : 0));
        bv[m] = *reinterpret_cast<const f32x4 *>(a.bias + (ok ? c : 0));
    }
}

template <int MW, int NW, bool DROP = false>
__device__ __forceinline__ void seg_epilogue_px(const SegArgs &a, const f32x4 (&acc)[MW][NW], int ct0, const int (&pidx)[NW], int kg,
                                                const f32x4 (&rvs)[MW], const f32x4 (&bvs)[MW]);

template <int MW, int NW, bool DROP = false>
__device__ __forceinline__ void seg_epilogue(const SegArgs &a, const f32x4 (&acc)[MW][NW], int ct0, int pt0, int n_pix, int col, int kg,
                                             const f32x4 (&rvs)[MW], const f32x4 (&bvs)[MW])
{
    int pidx[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int p = (pt0 + n) * 16 + col;
        pidx[n] = p < n_pix ? p : -1;
    }
    seg_epilogue_px<MW, NW, DROP>(a, acc, ct0, pidx, kg, rvs, bvs);
}

// pidx[n]: the (batch-major, row-major) output pixel this lane holds in accumulator column n, or -1
template <int MW, int NW, bool DROP>
__device__ __forceinline__ void seg_epilogue_px(const SegArgs &a, const f32x4 (&acc)[MW][NW], int ct0, const int (&pidx)[NW], int kg,
                                                const f32x4 (&rvs)[MW], const f32x4 (&bvs)[MW])
{
    // epilogue: lane holds output channels c .. c+3 of pixel pidx[n]
    float gmax = 0.0f;
    unsigned long long seed = 0, frame = 0;
    if constexpr (DROP) { seed = a.rng[0]; frame = a.rng[1]; }
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int c = (ct0 + m) * 16 + kg * 4;
        if (c >= a.pad_to) continue;
        const f32x4 rv = rvs[m], bv = bvs[m];
        // transposed convolution: GEMM row c = (phase, channel); phase (ay, ax) of input pixel (y, x) is output pixel
        // (y*up + ay, x*up + ax).  up_cp is a multiple of 4, so a lane's four rows share the phase.
        int co = c, n_co = a.c_out, ay = 0, ax = 0;
        if (a.up > 1) {
            const int phase = c / a.up_cp;
            co = c - phase * a.up_cp;
            n_co = a.up_c;
            ay = phase / a.up;
            ax = phase - ay * a.up;
            if (co >= n_co) continue;
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int p = pidx[n];
            if (p < 0) continue;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(acc[m][n][i], rv[i], bv[i]);
            const bool full = co + 3 < n_co;
            if (a.res) {
                const float *r = a.res + (size_t)p * a.res_stride + c;
                if (full && (a.vec_store & 2)) {  // 16-byte aligned residual rows: one load instead of four
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += q[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (full || co + i < n_co) v[i] += r[i];
                }
            }
            gmax = fmaxf(fmaxf(fmaxf(gmax, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
            if (a.act == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];  // keeps NaN, like torch.relu
            } else if (a.act == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = 1.0f / (1.0f + expf(-v[i]));
            }
            if (a.mul) {
                const float *g = a.mul + (size_t)p * a.mul_stride + c;
                if (full && (a.vec_store & 4)) {
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= q[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (full || co + i < n_co) v[i] *= g[i];
                }
            }
            if constexpr (DROP) {  // nn.Dropout(p = 0.5) in training mode, always (adapnet.py:80-82): keep with probability 1/2, scale by 2
                const unsigned e = (unsigned)p * (unsigned)(a.c_out >> 2) + (unsigned)(c >> 2);
                const uint4 r = philox4x32_10(uint4{e, a.drop_id, (unsigned)frame, (unsigned)(frame >> 32)},
                                              uint2{(unsigned)seed, (unsigned)(seed >> 32)});
                v[0] = (r.x & 1u) ? v[0] + v[0] : 0.0f;
                v[1] = (r.y & 1u) ? v[1] + v[1] : 0.0f;
                v[2] = (r.z & 1u) ? v[2] + v[2] : 0.0f;
                v[3] = (r.w & 1u) ? v[3] + v[3] : 0.0f;
            }
            if (a.pad_to > a.c_out) {  // pad channels of an engine-owned row: zeros (the next layer reads groups of 8)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i >= n_co) v[i] = 0.0f;
            }
            size_t row = (size_t)p;
            if (a.up > 1) {
                const int b = p / (a.Ho * a.Wo), q = p - b * (a.Ho * a.Wo);
                const int oy = q / a.Wo, ox = q - oy * a.Wo;
                row = (size_t)b * a.Ho * a.Wo * a.up * a.up + ((size_t)oy * a.up + ay) * ((size_t)a.Wo * a.up) + (size_t)ox * a.up + ax;
            }
            float *o = a.out + row * a.out_stride + co;
            if ((full || co + 3 < a.pad_to) && (a.vec_store & 1)) {
                *reinterpret_cast<f32x4 *>(o) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i < n_co || co + i < a.pad_to) o[i] = v[i];
            }
        }
    }
    if (a.ovf && gmax > 65504.0f) guard_raise(a.ovf, 1);  // a later layer would split this value: outside the fp16 range
    // (at the END of the kernel: tested first thing, the scalar load of this field sat in front of every other argument load
    // of every launch - 0.4 us each)
    if (a.rng_bump && blockIdx.x == 0 && threadIdx.x == 0) a.rng_bump[1] += 1;  // next frame: new dropout masks
}

// kDepth = K blocks in flight per wave (a cold weight fetch costs ~1 us, the MFMAs of a block ~0.1 us).
template <int MW, int NW, int WM, int KS, int kDepth, bool DROP = false>
__global__ __launch_bounds__(KS > 4 ? 64 * KS : 256) void segconv_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    constexpr bool SPLITK = KS > 1;  // KS waves of a block split K
    constexpr int WN = SPLITK ? 1 : 4 / WM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ct0 = SPLITK ? by * MW : (by * WM + wave % WM) * MW;
    const int pt0 = SPLITK ? bx * NW : (bx * WN + wave / WM) * NW;
    const int n_pix = a.B * a.Ho * a.Wo;
    // wave-uniform (n_ct is a multiple of 4 >= MW); block-uniform for SPLITK: a member of a heterogeneous launch
    // (ojf_segconv_forward_multi) may have fewer channel / pixel blocks than the grid
    if (ct0 >= a.n_ct || pt0 * 16 >= n_pix) return;
    const int col = lane & 15, kg = lane >> 4;
    int kb0 = 0, kb1 = a.n_kb;
    if constexpr (SPLITK) {
        const int per = (a.n_kb + KS - 1) / KS;
        kb0 = wave * per;
        kb0 = kb0 < a.n_kb ? kb0 : a.n_kb;
        kb1 = kb0 + per < a.n_kb ? kb0 + per : a.n_kb;
    }

    int iy0[NW], ix0[NW], img0[NW];
    bool live[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int p = (pt0 + n) * 16 + col;
        live[n] = p < n_pix;
        const int b = p / (a.Ho * a.Wo), q = p - b * (a.Ho * a.Wo);  // image, pixel inside it
        const int oy = q / a.Wo, ox = q - oy * a.Wo;
        iy0[n] = oy * a.stride - a.pad;
        ix0[n] = ox * a.stride - a.pad;
        img0[n] = b * a.H * a.W;
    }
    // this lane group's walk over the (tap, channel group) entries: entry kb*4 + kg of K block kb
    const int e0 = kb0 * 4 + kg;
    int tap = e0 / a.c8, cg = e0 - tap * a.c8;
    int ty = tap / a.ksize, tx = tap - ty * a.ksize;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    const f32x4 *wlane = a.wp + (size_t)ct0 * a.n_kb * 128 + lane;

    f32x4 rvs[MW], bvs[MW];
    seg_vectors<MW>(a, ct0, kg, rvs, bvs);
    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 xa[kDepth][NW], xb[kDepth][NW], wh[kDepth][MW], wl[kDepth][MW];
    // Branch-free on purpose: a conditional around the loads makes the compiler drain every outstanding load at the
    // join (s_waitcnt vmcnt(0)), which serialises the pipeline.  K blocks past the end fetch zeros (activations) and
    // the last valid weights (finite), so they add exact zeros.
    auto fetch = [&](int kb, f32x4 (&fa)[NW], f32x4 (&fb)[NW], f32x4 (&fh)[MW], f32x4 (&fl)[MW]) {
        const int dy = ty * a.dil, dx = tx * a.dil;
        const bool in_range = kb < kb1;
        const int kbc = kb < a.n_kb ? kb : a.n_kb - 1;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const f32x4 *w = wlane + ((size_t)m * a.n_kb + kbc) * 128;
            fh[m] = w[0];
            fl[m] = w[64];
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int iy = iy0[n] + dy, ix = ix0[n] + dx;
            const bool ok = in_range && live[n] && ty < a.ksize && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img0[n] + iy * a.W + ix) * a.in_stride + cg * 8) * 4) : 0xfffffff0u;
            fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
        }
        cg += 4;  // next K block: four entries further
        while (cg >= a.c8) {
            cg -= a.c8;
            if (++tx == a.ksize) {
                tx = 0;
                ++ty;
            }
        }
    };
    auto multiply = [&](const f32x4 (&fa)[NW], const f32x4 (&fb)[NW], const f32x4 (&fh)[MW], const f32x4 (&fl)[MW]) {
        f16x8 xh[NW], xl[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) split8(fa[n], fb[n], xh[n], xl[n]);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = mfma3(fh[m], fl[m], xh[n], xl[n], acc[m][n]);
    };

    const int rounds = (kb1 - kb0 + kDepth - 1) / kDepth;  // groups of kDepth K blocks (may be 0 for an idle split-K wave)
#pragma unroll
    for (int s = 0; s < kDepth; ++s) fetch(kb0 + s, xa[s], xb[s], wh[s], wl[s]);
    for (int r = 1; r < rounds; ++r) {
#pragma unroll
        for (int s = 0; s < kDepth; ++s) {
            multiply(xa[s], xb[s], wh[s], wl[s]);
            fetch(kb0 + r * kDepth + s, xa[s], xb[s], wh[s], wl[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < kDepth; ++s) multiply(xa[s], xb[s], wh[s], wl[s]);

    if constexpr (SPLITK) {
        __shared__ f32x4 part[KS - 1][MW * NW][64];
        if (wave) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) part[wave - 1][m * NW + n][lane] = acc[m][n];
        }
        __syncthreads();
        if (wave) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const f32x4 q = part[w][m * NW + n][lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][n][i] += q[i];
                }
    }

    seg_epilogue<MW, NW, DROP>(a, acc, ct0, pt0, n_pix, col, kg, rvs, bvs);
}

// Many-pixel layers (decoder 3x3 stacks, layer1): the four waves of a block work on the SAME 64 output channels and
// 32 pixels each, and the weight fragments of a K block reach them through LDS (LDS-DMA, 8 KB per K block, three
// stages): one global fetch per block instead of one per wave.  With per-wave fetches these layers were bound by
// L1 (12 KB per wave per K block against 24 MFMAs); B operands stay per-wave buffer loads.
template <int NW, int D>
__global__ __launch_bounds__(256) void segconv_wide_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    constexpr int MW = 4;  // D = stages of the weight ring / K blocks of pixel operands in flight per wave
    __shared__ f32x4 wtile[D][MW * 2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ct0 = by * MW;
    const int pt0 = (bx * 4 + wave) * NW;
    const int n_pix = a.B * a.Ho * a.Wo;
    const int col = lane & 15, kg = lane >> 4;
    const int n_kb = a.n_kb;

    int iy0[NW], ix0[NW], img0[NW];
    bool live[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int p = (pt0 + n) * 16 + col;
        live[n] = p < n_pix;
        const int b = p / (a.Ho * a.Wo), q = p - b * (a.Ho * a.Wo);  // image, pixel inside it
        const int oy = q / a.Wo, ox = q - oy * a.Wo;
        iy0[n] = oy * a.stride - a.pad;
        ix0[n] = ox * a.stride - a.pad;
        img0[n] = b * a.H * a.W;
    }
    int tap = kg / a.c8, cg = kg - tap * a.c8;
    int ty = tap / a.ksize, tx = tap - ty * a.ksize;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);

    f32x4 rvs[MW], bvs[MW];
    seg_vectors<MW>(a, ct0, kg, rvs, bvs);
    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave w moves chunks w and w + 4 of the 8 one-KB chunks (channel tile m, half h) of a K block
    auto stream_weights = [&](int kb, int stage) {
        const int kbc = kb < n_kb ? kb : n_kb - 1;  // past the end: refetch the last block (multiplied by zeros)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = wave + 4 * r, m = c >> 1, h = c & 1;
            const f32x4 *src = a.wp + ((size_t)(ct0 + m) * n_kb + kbc) * 128 + h * 64 + lane;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                             (void __attribute__((address_space(3))) *)(&wtile[stage][c * 64]), 16, 0, 0);
        }
    };
    f32x4 xa[D][NW], xb[D][NW];
    auto fetch_pixels = [&](int kb, f32x4 (&fa)[NW], f32x4 (&fb)[NW]) {
        const int dy = ty * a.dil, dx = tx * a.dil;
        const bool in_range = kb < n_kb;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int iy = iy0[n] + dy, ix = ix0[n] + dx;
            const bool ok = in_range && live[n] && ty < a.ksize && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img0[n] + iy * a.W + ix) * a.in_stride + cg * 8) * 4) : 0xfffffff0u;
            fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
        }
        cg += 4;
        while (cg >= a.c8) {
            cg -= a.c8;
            if (++tx == a.ksize) {
                tx = 0;
                ++ty;
            }
        }
    };
    auto multiply = [&](int stage, const f32x4 (&fa)[NW], const f32x4 (&fb)[NW]) {
        f16x8 xh[NW], xl[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) split8(fa[n], fb[n], xh[n], xl[n]);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const f32x4 wh = wtile[stage][(m * 2) * 64 + lane], wl = wtile[stage][(m * 2 + 1) * 64 + lane];
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = mfma3(wh, wl, xh[n], xl[n], acc[m][n]);
        }
    };

    // every wave issues exactly 2 + 2*NW memory operations per K block, weights first: when the pixel operands of
    // block kb have arrived, so have this wave's weight chunks of block kb (in-order return); the barrier then makes
    // the other waves' chunks visible and guarantees that nobody still reads the stage refilled next
    const int rounds = (n_kb + D - 1) / D;
#pragma unroll
    for (int s = 0; s < D - 1; ++s) {
        stream_weights(s, s);
        fetch_pixels(s, xa[s], xb[s]);
    }
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int kb = r * D + s;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (2 + 2 * NW)) : "memory");  // all but the D - 2 newest K blocks' operations
            __syncthreads();
            stream_weights(kb + D - 1, (s + D - 1) % D);
            fetch_pixels(kb + D - 1, xa[(s + D - 1) % D], xb[(s + D - 1) % D]);
            multiply(s, xa[s], xb[s]);
        }
    }
    if (pt0 * 16 >= n_pix) return;
    seg_epilogue<MW, NW>(a, acc, ct0, pt0, n_pix, col, kg, rvs, bvs);
}

// The same tile as segconv_wide_kernel (64 output channels x 4 waves x NW pixel tiles, weights shared through LDS), with the
// weight fragments staged through REGISTERS instead of LDS-DMA (round 5).  A wave's `global_load_lds` instructions do not
// overlap: the second one issues when the first has returned (tools/microbench/dma_issue_bench.hip, round 4), so the wide
// kernel paid two memory latencies per K block whatever its ring depth - 1 us per K block on a 60x80 decoder layer, 71 us
// for 72 K blocks where the MFMAs need 12.  Plain loads do not block: every wave keeps P K blocks of its two weight chunks
// and of its pixel operands in flight in registers, writes the chunks of block i + 1 into the other half of a two-slot LDS
// tile while block i is multiplied, one barrier per K block.
template <int NW, int U>
__global__ __launch_bounds__(256) void segconv_tile_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    // U K blocks per step (= per barrier): with one wave per SIMD - what these launches have - the phases of a step (wait,
    // LDS write, split, loads, LDS reads, MFMAs, barrier) run one after the other, ~1300 cycles per K block at U = 1 for 192
    // cycles of MFMAs (SQ counters, round 5); a longer step amortises the waits and lets the LDS reads of one K block sit
    // under the MFMAs of the one before.  P steps in flight per wave.
    constexpr int MW = 4, P = 2;
    __shared__ f32x4 wtile[2][U][MW * 2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ct0 = by * MW;
    const int pt0 = (bx * 4 + wave) * NW;
    const int n_pix = a.B * a.Ho * a.Wo;
    const int col = lane & 15, kg = lane >> 4;
    const int n_kb = a.n_kb;

    int iy0[NW], ix0[NW], img0[NW];
    bool live[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int p = (pt0 + n) * 16 + col;
        live[n] = p < n_pix;
        const int b = p / (a.Ho * a.Wo), q = p - b * (a.Ho * a.Wo);  // image, pixel inside it
        const int oy = q / a.Wo, ox = q - oy * a.Wo;
        iy0[n] = oy * a.stride - a.pad;
        ix0[n] = ox * a.stride - a.pad;
        img0[n] = b * a.H * a.W;
    }
    int tap = kg / a.c8, cg = kg - tap * a.c8;
    int ty = tap / a.ksize, tx = tap - ty * a.ksize;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);

    f32x4 rvs[MW], bvs[MW];
    seg_vectors<MW>(a, ct0, kg, rvs, bvs);
    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave w stages chunks w and w + 4 of the 8 one-KB chunks (channel tile m, half h) of a K block
    const f32x4 *wsrc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c = wave + 4 * r, m = c >> 1, h = c & 1;
        wsrc[r] = a.wp + (size_t)(ct0 + m) * n_kb * 128 + h * 64 + lane;
    }
    f32x4 wr[P][U][2], xa[P][U][NW], xb[P][U][NW];
    auto issue = [&](int kb0, f32x4 (&fw)[U][2], f32x4 (&fa)[U][NW], f32x4 (&fb)[U][NW]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {  // all weight chunks of the step first: they are wanted one step earlier than the pixels
            const int kb = kb0 + u, kbc = kb < n_kb ? kb : n_kb - 1;  // past the end: the last block again (multiplied by zeros)
            fw[u][0] = wsrc[0][(size_t)kbc * 128];
            fw[u][1] = wsrc[1][(size_t)kbc * 128];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kb = kb0 + u;
            const int dy = ty * a.dil, dx = tx * a.dil;
            const bool in_range = kb < n_kb;
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int iy = iy0[n] + dy, ix = ix0[n] + dx;
                const bool ok = in_range && live[n] && ty < a.ksize && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(((img0[n] + iy * a.W + ix) * a.in_stride + cg * 8) * 4) : 0xfffffff0u;
                fa[u][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                fb[u][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
            }
            cg += 4;
            while (cg >= a.c8) {
                cg -= a.c8;
                if (++tx == a.ksize) {
                    tx = 0;
                    ++ty;
                }
            }
        }
    };
    auto stage = [&](int buf, const f32x4 (&fw)[U][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wtile[buf][u][wave * 64 + lane] = fw[u][0];
            wtile[buf][u][(wave + 4) * 64 + lane] = fw[u][1];
        }
    };
    constexpr int kW = 2 * U, kX = 2 * NW * U;  // memory operations of one step per wave, in issue order: weight chunks, then pixels
#pragma unroll
    for (int s = 0; s < P; ++s) issue(s * U, wr[s], xa[s], xb[s]);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * (kW + kX) + kX) : "memory");  // the weight chunks of step 0
    stage(0, wr[0]);
    __syncthreads();
    const int steps = (n_kb + U - 1) / U, rounds = (steps + P - 1) / P;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            // here: steps i .. i + P - 1 are in flight (i = r P + s); wanted: the pixels of i, the weight chunks of i + 1
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 2) * (kW + kX) + kX) : "memory");
            stage((s + 1) & 1, wr[(s + 1) % P]);
            f16x8 xh[U][NW], xl[U][NW];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int n = 0; n < NW; ++n) split8(xa[s][u][n], xb[s][u][n], xh[u][n], xl[u][n]);
            issue((r * P + s + P) * U, wr[s], xa[s], xb[s]);  // (the registers of step i are free: its chunks sit in LDS, its pixels are split)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const f32x4 wh = wtile[s & 1][u][(m * 2) * 64 + lane], wl = wtile[s & 1][u][(m * 2 + 1) * 64 + lane];
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[m][n] = mfma3(wh, wl, xh[u][n], xl[u][n], acc[m][n]);
                }
            __syncthreads();  // step i + 1's chunks are visible; nobody reads slot i & 1 any more
        }
    }
    if (pt0 * 16 >= n_pix) return;
    seg_epilogue<MW, NW>(a, acc, ct0, pt0, n_pix, col, kg, rvs, bvs);
}

// GEMM-shaped form for layers with enough pixels to tile both ways (round 5: the frames of several scenes / a look-ahead
// chunk as one [B, H, W, C] pass, the 60x80 decoder layers): block = WM x WN waves (WN = 4 / WM), wave = MT channel tiles x
// NT pixel tiles (accumulators MT x NT x 4 registers), block tile 16 WM MT channels x 16 WN NT pixels.  BOTH operands go
// through LDS, once per block: the weight fragments as they are packed, the pixel operands ALREADY SPLIT into fp16 halves
// by the wave that fetched them (one split per element instead of one per consuming wave; the consumers' K loop is
// ds_read_b128 + MFMA only).  Two K blocks in flight per wave in registers, two LDS stages, one barrier per K block.
// What the measurements of round 5 say about this form (profiles/r05_seg_experiments.txt):
// * with ONE wave per SIMD the K loop is bound by what the wave must ISSUE: the arguments the loop needs live in locals;
//   the weights are fetched through a buffer resource with the K block as the SCALAR offset (no per-load address VALU);
//   and when the layer's channel groups come in fours (c_in a multiple of 32: every heavy layer) the four lane groups
//   of a K block share the tap, so the tap walk is scalar, the bounds test runs once per tap and a K block's pixel
//   address is one add (ALIGNED).
// * a launch's time is a STEP function of its block count: 120 .. 256 blocks of the 64 x 64 tile on a 256 -> 256 3x3 layer
//   take 29 .. 31 us, 260 .. 512 blocks 43 .. 54 us, 520 blocks 69 us (one block per CU runs no faster when its
//   neighbour idles).  Neither more waves per CU (K groups inside a block), nor a second set of operand registers
//   (next block's LDS reads under this block's MFMAs), nor 3 / 4 K blocks in flight, nor another K order moved the
//   time of a given grid by more than 2 %.  Hence the MENU of tile shapes in seg_launch: the shape is chosen so that
//   the block count lands just under a multiple of the CU count.
template <int MT, int NT, bool DROP, bool ALIGNED, int WM = 2, int P = 2>  // (P = 3 / 4: measured no faster)
__global__ __launch_bounds__(256, 2) void segconv_gemm_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    static_assert(WM == 1 || WM == 2 || WM == 4, "WM");
    constexpr int WN = 4 / WM;
    constexpr int MB = WM * MT, NB = WN * NT;  // channel / pixel tiles of the block
    constexpr int PT = (NB + 3) / 4;           // pixel tiles a wave fetches and splits per K block
    constexpr int CW = (2 * MB + 3) / 4;       // one-KB weight chunks (channel tile, half) a wave fetches per K block
    __shared__ f32x4 As[2][MB][2][64];
    __shared__ f32x4 Bs[2][NB][2][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (a scalar: the guards below are branches, not lane masks)
    const int wm = wave % WM, wn = wave / WM;
    constexpr bool kShortPix = 4 * PT > NB, kShortW = 4 * CW > 2 * MB;  // the last wave(s) fetch less than a full share
    const int ct0 = by * MB, pt0 = bx * NB;
    // (locals: the compiler re-fetches fields of a kernarg-resident struct behind a dynamic index whenever it runs short of SGPRs)
    const int H = a.H, W = a.W, HoWo = a.Ho * a.Wo, Wo = a.Wo, c8 = a.c8, ksize = a.ksize, dil = a.dil, in_stride = a.in_stride;
    const int n_pix = a.B * HoWo, n_kb = a.n_kb, n_ct = a.n_ct;
    const int col = lane & 15, kg = lane >> 4;
    constexpr int abl = OJF_GEMM_ABL;  // build-time ablation (tools/r5_run24.sh, r5_run36.sh): 1 no MFMA, 2 no LDS operand reads, 4 no global loads, 8 no staging, 16 no barrier, 32 weight walk rotated per pixel block, 64 taps inner, 128 no fp16 split of the pixel operands

    // producer side: this wave's PT pixel tiles (tiles wave * PT .. of the block) and CW weight chunks
    int iy0[PT], ix0[PT], img0[PT];
    bool live[PT];
#pragma unroll
    for (int n = 0; n < PT; ++n) {
        const int p = (pt0 + wave * PT + n) * 16 + col;
        live[n] = p < n_pix && (!kShortPix || wave * PT + n < NB);
        const int b = p / HoWo, q = p - b * HoWo;
        const int oy = q / Wo, ox = q - oy * Wo;
        iy0[n] = oy * a.stride - a.pad;
        ix0[n] = ox * a.stride - a.pad;
        img0[n] = b * H * W;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    // weights: one resource over the layer's packed fragments; per chunk a lane offset, per K block the scalar offset kb * 2 KB
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(a.wp), 0, (unsigned)((size_t)n_ct * n_kb * 2048), 0x00020000);
    unsigned woff[CW];
#pragma unroll
    for (int r = 0; r < CW; ++r) {
        const int c = wave * CW + r, m = c >> 1, h = c & 1;  // chunk c of the block's 2 MB one-KB chunks (channel tile m, half h)
        int ct = ct0 + m;
        ct = ct < n_ct ? ct : n_ct - 1;  // (a block past the layer's last channel tile: any valid tile, results unused)
        woff[r] = (unsigned)(((ct * n_kb) * 128 + h * 64 + lane) * 16);
    }
    // K walk of this lane group.  ALIGNED: (tap, first channel group of the K block) are wave-uniform scalars, the lane adds kg
    const int n_cgb = c8 >> 2;
    int tap_s = 0, cgb_s = 0;                        // ALIGNED: tap, K block inside the tap (c8 / 4 of them)
    unsigned toff[PT];                               // ALIGNED: byte offset of (tap, channel group kg) per pixel tile, or the out-of-range sentinel
    int tap = kg / c8, cg = kg - tap * c8;           // general: per-lane walk
    int ty = tap / ksize, tx = tap - ty * ksize;
    auto tap_offsets = [&](int t) {
        const int y = t / ksize, x = t - y * ksize;
#pragma unroll
        for (int n = 0; n < PT; ++n) {
            const int iy = iy0[n] + y * dil, ix = ix0[n] + x * dil;
            const bool ok = live[n] && y < ksize && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            toff[n] = ok ? (unsigned)(((img0[n] + iy * W + ix) * in_stride + kg * 8) * 4) : 0xfffffff0u;
        }
    };
    if constexpr (ALIGNED) tap_offsets(0);
    f32x4 wr[P][CW], xa[P][PT], xb[P][PT];  // P K blocks in flight per wave
    auto issue = [&](int kb, f32x4 (&fw)[CW], f32x4 (&fa)[PT], f32x4 (&fb)[PT]) {
        int kbc = kb < n_kb ? kb : n_kb - 1;  // past the end: the last block again (multiplied by zeros)
        if (abl & 32) kbc = (kbc + bx * 5) % n_kb;  // (timing only, wrong sums) every pixel block of a channel block on another weight line at any time
        if (abl & 4) return;
#pragma unroll
        for (int r = 0; r < CW; ++r)
            if (!kShortW || wave * CW + r < 2 * MB) fw[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[r], kbc * 2048, 0));
        if constexpr (ALIGNED) {
            const unsigned step = (unsigned)cgb_s * 128u;  // four channel groups of 8 floats per K block
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                if (kShortPix && wave * PT + n >= NB) continue;  // (wave-uniform: the block's NB tiles do not fill the last wave's share)
                const bool ok = toff[n] != 0xfffffff0u;
                const unsigned off = ok ? toff[n] + step : 0xfffffff0u;
                fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
            }
            if (abl & 64) {  // (timing only, wrong sums) taps inner, channel groups outer: a K block's pixel lines are the previous block's, shifted
                if (++tap_s == ksize * ksize) { tap_s = 0; ++cgb_s; }
                tap_offsets(tap_s);
            } else if (++cgb_s == n_cgb) {  // (uniform) next tap; past the last one every tile is out of range: zeros
                cgb_s = 0;
                tap_offsets(++tap_s);
            }
        } else {
            const int dy = ty * dil, dx = tx * dil;
            const bool in_range = kb < n_kb;
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                if (kShortPix && wave * PT + n >= NB) continue;
                const int iy = iy0[n] + dy, ix = ix0[n] + dx;
                const bool ok = in_range && live[n] && ty < ksize && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned off = ok ? (unsigned)(((img0[n] + iy * W + ix) * in_stride + cg * 8) * 4) : 0xfffffff0u;
                fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
            }
            cg += 4;
            while (cg >= c8) {
                cg -= c8;
                if (++tx == ksize) {
                    tx = 0;
                    ++ty;
                }
            }
        }
    };
    f32x4 sink{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int st, const f32x4 (&fw)[CW], const f32x4 (&fa)[PT], const f32x4 (&fb)[PT]) {
        if (abl & 8) {  // (the loads stay alive: their sum goes into an accumulator)
#pragma unroll
            for (int r = 0; r < CW; ++r) sink += fw[r];
#pragma unroll
            for (int n = 0; n < PT; ++n) sink += fa[n] + fb[n];
            return;
        }
#pragma unroll
        for (int r = 0; r < CW; ++r) {
            const int c = wave * CW + r;
            if (!kShortW || c < 2 * MB) As[st][c >> 1][c & 1][lane] = fw[r];
        }
#pragma unroll
        for (int n = 0; n < PT; ++n) {
            if (kShortPix && wave * PT + n >= NB) continue;
            if (abl & 128) {  // (timing only, wrong sums) as if the producer had stored the activations as fp16 halves: no split
                Bs[st][wave * PT + n][0][lane] = fa[n];
                Bs[st][wave * PT + n][1][lane] = fb[n];
                continue;
            }
            f16x8 xh, xl;
            split8(fa[n], fb[n], xh, xl);
            Bs[st][wave * PT + n][0][lane] = __builtin_bit_cast(f32x4, xh);
            Bs[st][wave * PT + n][1][lane] = __builtin_bit_cast(f32x4, xl);
        }
    };
    // memory operations of one K block of THIS wave (the count s_waitcnt works with); the waves of a block differ when NB or
    // 2 MB is not a multiple of four: a wave with a short share waits for all of its few loads
    constexpr int kOpsFull = CW + 2 * PT;
    int my_ops = kOpsFull;
    if constexpr (kShortPix || kShortW) {
        my_ops = 0;
        for (int r = 0; r < CW; ++r) my_ops += !kShortW || wave * CW + r < 2 * MB;
        for (int t = 0; t < PT; ++t) my_ops += 2 * (!kShortPix || wave * PT + t < NB);
    }
    auto wait_landed = [&]() {  // all but the newest P - 1 blocks of this wave have landed
        if constexpr (kShortPix || kShortW) {
            static_assert(kOpsFull <= 12 && P == 2, "wait_landed: cases");
            switch (my_ops) {  // (s_waitcnt takes an immediate)
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * kOpsFull) : "memory");
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ctw = ct0 + wm * MT, ptw = pt0 + wn * NT;  // this wave's accumulator tiles

#pragma unroll
    for (int q = 0; q < P; ++q) issue(q, wr[q], xa[q], xb[q]);
    wait_landed();
    stage(0, wr[0], xa[0], xb[0]);
    issue(P, wr[0], xa[0], xb[0]);
    __syncthreads();
    // K block i sits in LDS stage i & 1; blocks i + 1 .. i + P are in flight, block j in register slot j % P
    const int rounds = (n_kb + 2 * P - 1) / (2 * P);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 2 * P; ++u) {  // (2 P: a common period of the LDS stage and the register slot)
            const int i = r * 2 * P + u;
            if (i >= n_kb) break;  // (uniform; only in the last round)
            // this block's pixel operands and the first channel tile's weights out of LDS FIRST, back to back: their latency
            // (hundreds of cycles with the block's waves queueing reads) then passes under the staging of the next block instead
            // of in front of every group of MFMAs
            f32x4 bh[NT], bl[NT], ah[2], al[2];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bh[n] = (abl & 2) ? xa[0][0] : Bs[u & 1][wn * NT + n][0][lane];
                bl[n] = (abl & 2) ? xb[0][0] : Bs[u & 1][wn * NT + n][1][lane];
            }
            ah[0] = (abl & 2) ? wr[0][0] : As[u & 1][wm * MT][0][lane];
            al[0] = (abl & 2) ? wr[1 % P][0] : As[u & 1][wm * MT][1][lane];
            __builtin_amdgcn_sched_barrier(0);
            wait_landed();  // block i + 1 has landed
            stage((u + 1) & 1, wr[(u + 1) % P], xa[(u + 1) % P], xb[(u + 1) % P]);
            issue(i + 1 + P, wr[(u + 1) % P], xa[(u + 1) % P], xb[(u + 1) % P]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m + 1 < MT) {  // the next channel tile's weight fragments while this one's MFMAs run
                    ah[(m + 1) & 1] = (abl & 2) ? wr[0][0] : As[u & 1][wm * MT + m + 1][0][lane];
                    al[(m + 1) & 1] = (abl & 2) ? wr[1 % P][0] : As[u & 1][wm * MT + m + 1][1][lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(abl & 1)) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = mfma3(ah[m & 1], al[m & 1], __builtin_bit_cast(f16x8, bh[n]), __builtin_bit_cast(f16x8, bl[n]), acc[m][n]);
                } else {
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n][0] += ah[m & 1][0] * bh[n][0] + al[m & 1][1] * bl[n][1];  // (keeps the operands alive)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(abl & 16)) __syncthreads();  // the other stage is complete for the next K block; nobody reads this one any more
        }
    }
    if (abl & 8) acc[0][0] += sink;
    if (ctw >= n_ct || ptw * 16 >= n_pix) return;
    f32x4 rvs[MT], bvs[MT];
    seg_vectors<MT>(a, ctw, kg, rvs, bvs);
    seg_epilogue<MT, NT, DROP>(a, acc, ctw, ptw, n_pix, col, kg, rvs, bvs);
}

// Producer / consumer form of the GEMM-shaped kernel (round 6).  What round 5's ablations said about segconv_gemm_kernel: with
// the SAME four waves fetching, splitting, staging and multiplying, the pieces of a K block ADD - loads + split + ds_write + barrier
// cost 42 us on top of the 60 us of LDS reads + MFMAs of the 60x80 256 -> 256 layer - because a wave is in one phase at a time and a
// SIMD holds one such wave.  Here the phases belong to DIFFERENT waves of one block: waves 0 .. 3 (one per SIMD, 2 x 2 over the block
// tile) only read operands from LDS and issue MFMAs; waves 4 .. 7 (one per SIMD, beside a consumer) only move data - plain 16-byte
// loads of the packed weight chunks and of the fp32 pixel rows, P K blocks in flight in registers, the fp16 split of the pixel
// operands, ds_write_b128 into the other LDS stage.  A SIMD's MFMA pipe runs the consumer's 16-cycle instructions while its VALU /
// memory ports take the producer's; one block-wide barrier per K block hands a finished stage over.  Same K order and the same
// arithmetic as segconv_gemm_kernel (one accumulator walks the K blocks in order): the same bits.
template <int MT, int NT, bool DROP, bool ALIGNED, int P = 3>
__global__ __launch_bounds__(512) void segconv_ws_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    constexpr int WM = 2, WN = 2;
    constexpr int MB = WM * MT, NB = WN * NT;  // channel / pixel tiles of the block
    static_assert(NB % 4 == 0 && (2 * MB) % 4 == 0, "every producer wave moves the same share");
    constexpr int PT = NB / 4;                 // pixel tiles a producer wave fetches and splits per K block
    constexpr int CW = 2 * MB / 4;             // one-KB weight chunks (channel tile, half) a producer wave fetches per K block
    __shared__ f32x4 As[2][MB][2][64];
    __shared__ f32x4 Bs[2][NB][2][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct0 = by * MB, pt0 = bx * NB;
    const int n_kb = a.n_kb, n_ct = a.n_ct;
    const int col = lane & 15, kg = lane >> 4;

    if (wave >= 4) {
        // ---------------- producer ----------------
        const int pw = wave - 4;
        const int H = a.H, W = a.W, HoWo = a.Ho * a.Wo, Wo = a.Wo, c8 = a.c8, ksize = a.ksize, dil = a.dil, in_stride = a.in_stride;
        const int n_pix = a.B * HoWo;
        int iy0[PT], ix0[PT], img0[PT];
        bool live[PT];
#pragma unroll
        for (int n = 0; n < PT; ++n) {
            const int p = (pt0 + pw * PT + n) * 16 + col;
            live[n] = p < n_pix;
            const int b = p / HoWo, q = p - b * HoWo;
            const int oy = q / Wo, ox = q - oy * Wo;
            iy0[n] = oy * a.stride - a.pad;
            ix0[n] = ox * a.stride - a.pad;
            img0[n] = b * H * W;
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(a.wp), 0, (unsigned)((size_t)n_ct * n_kb * 2048), 0x00020000);
        unsigned woff[CW];
#pragma unroll
        for (int r = 0; r < CW; ++r) {
            const int c = pw * CW + r, m = c >> 1, h = c & 1;
            int ct = ct0 + m;
            ct = ct < n_ct ? ct : n_ct - 1;  // (a block past the layer's last channel tile: any valid tile, results unused)
            woff[r] = (unsigned)(((ct * n_kb) * 128 + h * 64 + lane) * 16);
        }
        const int n_cgb = c8 >> 2;
        int tap_s = 0, cgb_s = 0;
        unsigned toff[PT];
        int tap = kg / c8, cg = kg - tap * c8;
        int ty = tap / ksize, tx = tap - ty * ksize;
        auto tap_offsets = [&](int t) {
            const int y = t / ksize, x = t - y * ksize;
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                const int iy = iy0[n] + y * dil, ix = ix0[n] + x * dil;
                const bool ok = live[n] && y < ksize && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                toff[n] = ok ? (unsigned)(((img0[n] + iy * W + ix) * in_stride + kg * 8) * 4) : 0xfffffff0u;
            }
        };
        if constexpr (ALIGNED) tap_offsets(0);
        f32x4 wr[P][CW], xa[P][PT], xb[P][PT];
        auto issue = [&](int kb, f32x4 (&fw)[CW], f32x4 (&fa)[PT], f32x4 (&fb)[PT]) {
            const int kbc = kb < n_kb ? kb : n_kb - 1;  // past the end: the last block again (never staged)
#pragma unroll
            for (int r = 0; r < CW; ++r) fw[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[r], kbc * 2048, 0));
            if constexpr (ALIGNED) {
                const unsigned step = (unsigned)cgb_s * 128u;
#pragma unroll
                for (int n = 0; n < PT; ++n) {
                    const bool ok = toff[n] != 0xfffffff0u;
                    const unsigned off = ok ? toff[n] + step : 0xfffffff0u;
                    fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                    fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
                }
                if (++cgb_s == n_cgb) {
                    cgb_s = 0;
                    tap_offsets(++tap_s);
                }
            } else {
                const int dy = ty * dil, dx = tx * dil;
                const bool in_range = kb < n_kb;
#pragma unroll
                for (int n = 0; n < PT; ++n) {
                    const int iy = iy0[n] + dy, ix = ix0[n] + dx;
                    const bool ok = in_range && live[n] && ty < ksize && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    const unsigned off = ok ? (unsigned)(((img0[n] + iy * W + ix) * in_stride + cg * 8) * 4) : 0xfffffff0u;
                    fa[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                    fb[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
                }
                cg += 4;
                while (cg >= c8) {
                    cg -= c8;
                    if (++tx == ksize) {
                        tx = 0;
                        ++ty;
                    }
                }
            }
        };
        auto stage = [&](int st, const f32x4 (&fw)[CW], const f32x4 (&fa)[PT], const f32x4 (&fb)[PT]) {
#pragma unroll
            for (int r = 0; r < CW; ++r) {
                const int c = pw * CW + r;
                As[st][c >> 1][c & 1][lane] = fw[r];
            }
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                f16x8 xh, xl;
                split8(fa[n], fb[n], xh, xl);
                Bs[st][pw * PT + n][0][lane] = __builtin_bit_cast(f32x4, xh);
                Bs[st][pw * PT + n][1][lane] = __builtin_bit_cast(f32x4, xl);
            }
        };
        constexpr int kOps = CW + 2 * PT;  // memory operations of one K block of a producer wave
#pragma unroll
        for (int q = 0; q < P; ++q) issue(q, wr[q], xa[q], xb[q]);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * kOps) : "memory");
        stage(0, wr[0], xa[0], xb[0]);
        issue(P, wr[0], xa[0], xb[0]);
        __syncthreads();  // stage 0 is complete
        // at iteration i: K blocks i + 1 .. i + P are in flight, block j in register slot j % P; block i + 1 goes into stage (i + 1) & 1
        const int rounds = (n_kb + 2 * P - 1) / (2 * P);
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int u = 0; u < 2 * P; ++u) {  // (2 P: a common period of the LDS stage and the register slot)
                const int i = r * 2 * P + u;
                if (i >= n_kb) break;  // (uniform; only in the last round)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * kOps) : "memory");  // block i + 1 has landed
                if (i + 1 < n_kb) stage((u + 1) & 1, wr[(u + 1) % P], xa[(u + 1) % P], xb[(u + 1) % P]);
                issue(i + 1 + P, wr[(u + 1) % P], xa[(u + 1) % P], xb[(u + 1) % P]);
                __syncthreads();  // stage (i + 1) & 1 is complete; the consumers are done with stage i & 1
            }
        }
        return;
    }

    // ---------------- consumer ----------------
    const int wm = wave % WM, wn = wave / WM;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ctw = ct0 + wm * MT, ptw = pt0 + wn * NT;  // this wave's accumulator tiles
    __syncthreads();  // stage 0 is complete
    for (int i = 0; i < n_kb; ++i) {
        const int st = i & 1;
        f32x4 bh[NT], bl[NT], ah[2], al[2];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[n] = Bs[st][wn * NT + n][0][lane];
            bl[n] = Bs[st][wn * NT + n][1][lane];
        }
        ah[0] = As[st][wm * MT][0][lane];
        al[0] = As[st][wm * MT][1][lane];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m + 1 < MT) {  // the next channel tile's weight fragments while this one's MFMAs run
                ah[(m + 1) & 1] = As[st][wm * MT + m + 1][0][lane];
                al[(m + 1) & 1] = As[st][wm * MT + m + 1][1][lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = mfma3(ah[m & 1], al[m & 1], __builtin_bit_cast(f16x8, bh[n]), __builtin_bit_cast(f16x8, bl[n]), acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // the other stage is complete for the next K block; nobody reads this one any more
    }
    const int n_pix = a.B * a.Ho * a.Wo;
    if (ctw >= n_ct || ptw * 16 >= n_pix) return;
    f32x4 rvs[MT], bvs[MT];
    seg_vectors<MT>(a, ctw, kg, rvs, bvs);
    seg_epilogue<MT, NT, DROP>(a, acc, ctw, ptw, n_pix, col, kg, rvs, bvs);
}

// WINDOW form (round 6) for 3x3 / stride 1 / dilation 1 / padding 1 layers whose channel groups come in fours (c_in % 32 == 0).  What
// this round's gate measured (profiles/r06_seg_ws_gate.txt): the GEMM-shaped launches sit on an L2 -> CU plateau of ~8 TB/s because every
// (tap, channel block) of a K walk re-fetches its pixel rows - nine times per 3x3 layer - and every pixel block re-fetches the weights.
// Here the K walk is (32-channel chunk) outer, (tap) inner: the chunk's pixels of the block's TH x 16 output tile PLUS its one-pixel halo
// are fetched ONCE, split into fp16 halves once, and stay in LDS for all nine taps, which read them at shifted positions (pixel pitch 144
// bytes: the 16 lanes of an operand read hit 16 different 16-byte bank groups).  The weights of a (tap, chunk) K block stream through a
// two-stage LDS buffer, two K blocks in flight in registers, one barrier per K block, the next chunk's window loads ride along (issued at
// tap 0, written at tap 3).  Block = 2 x 2 waves over (2 MT channel tiles) x (TH rows of 16 pixels).  K blocks are added chunk-major
// instead of tap-major: the same products, another rounding order than the other forms (like every change of form: include/ojf.h).
template <int MT, int TH, bool DROP = false>
__global__ __launch_bounds__(256) void segconv_win_kernel(const SegGroupArgs grp)
{
    int bx, by, bz;
    if (!seg_block(grp, bx, by, bz)) return;  // block-uniform
    const SegArgs &a = grp.a[bz];
    constexpr int MB = 2 * MT, NT = TH / 2, WW = 18, WH = TH + 2, WPIX = WW * WH, PIXB = 144;
    constexpr int WI = (WPIX * 4 + 255) / 256;  // window items (pixel, 8-channel group) per thread
    constexpr int CW = 2 * MB / 4;              // one-KB weight chunks per wave and K block
    static_assert(TH % 2 == 0 && (2 * MB) % 4 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char win_lds[];
    unsigned char *Ws = win_lds;                                             // [2][WPIX * PIXB]
    f32x4 *As = reinterpret_cast<f32x4 *>(win_lds + 2 * WPIX * PIXB);        // [2][MB][2][64]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int col = lane & 15, kg = lane >> 4;
    const int H = a.H, W = a.W, in_stride = a.in_stride, n_kb = a.n_kb, n_ct = a.n_ct;
    const int n_chunks = a.c8 >> 2, total = 9 * n_chunks;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + TH - 1) / TH;
    const int img = bx / (tiles_x * tiles_y), rb = bx - img * (tiles_x * tiles_y);
    const int tyb = rb / tiles_x, txb = rb - tyb * tiles_x;
    const int y0 = tyb * TH, x0 = txb * 16;
    const int ct0 = by * MB;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(a.wp), 0, (unsigned)((size_t)n_ct * n_kb * 2048), 0x00020000);
    unsigned woff[CW];
#pragma unroll
    for (int r = 0; r < CW; ++r) {
        const int c = wave * CW + r, m = c >> 1, h = c & 1;
        int ct = ct0 + m;
        ct = ct < n_ct ? ct : n_ct - 1;  // (a block past the layer's last channel tile: any valid tile, results unused)
        woff[r] = (unsigned)(((ct * n_kb) * 128 + h * 64 + lane) * 16);
    }
    // this thread's window items: global byte offset of channel group 0 of its pixel (or the out-of-range sentinel), LDS byte offset
    unsigned goff[WI], loff[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int item = (int)threadIdx.x + 256 * j;
        const int wp = item >> 2, q = item & 3;
        const int wy = wp / WW, wx = wp - wy * WW;
        const int iy = y0 - 1 + wy, ix = x0 - 1 + wx;
        const bool ok = item < WPIX * 4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        goff[j] = ok ? (unsigned)((((img * H + iy) * W + ix) * in_stride + q * 8) * 4) : 0xfffffff0u;
        loff[j] = item < WPIX * 4 ? (unsigned)(wp * PIXB + q * 32) : 0xffffffffu;
    }
    f32x4 wa[WI], wb[WI];
    auto issue_window = [&](int chunk) {  // (a chunk past the last one: zeros, no traffic)
        const unsigned step = (unsigned)chunk * 128u;
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const bool ok = goff[j] != 0xfffffff0u && chunk < n_chunks;
            const unsigned off = ok ? goff[j] + step : 0xfffffff0u;
            wa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            wb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : 0xfffffff0u, 0, 0));
        }
    };
    auto stage_window = [&](int st) {
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            if (loff[j] == 0xffffffffu) continue;
            f16x8 xh, xl;
            split8(wa[j], wb[j], xh, xl);
            unsigned char *dst = Ws + st * (WPIX * PIXB) + loff[j];
            *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, xh);
            *reinterpret_cast<f32x4 *>(dst + 16) = __builtin_bit_cast(f32x4, xl);
        }
    };
    // the K block of step s = (chunk c, tap t) sits at index t * n_chunks + c of the layer's packed K blocks
    f32x4 wr[2][CW];
    int it = 0, ic = 0;  // (tap, chunk) of the next weight K block to request
    auto issue_weights = [&](f32x4 (&fw)[CW]) {
        const int kb = ic < n_chunks ? it * n_chunks + ic : n_kb - 1;  // past the end: the last block again (never used)
#pragma unroll
        for (int r = 0; r < CW; ++r) fw[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[r], kb * 2048, 0));
        if (++it == 9) { it = 0; ++ic; }
    };
    auto stage_weights = [&](int st, const f32x4 (&fw)[CW]) {
#pragma unroll
        for (int r = 0; r < CW; ++r) {
            const int c = wave * CW + r;
            As[((st * MB + (c >> 1)) * 2 + (c & 1)) * 64 + lane] = fw[r];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: window of chunk 0, weights of steps 0 (staged), 1 and 2 (in flight)
    issue_window(0);
    issue_weights(wr[0]);
    issue_weights(wr[1]);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW) : "memory");
    stage_window(0);
    stage_weights(0, wr[0]);
    issue_weights(wr[0]);
    __syncthreads();
    // this lane's operand position in the window: output pixel (row wn * NT + n, column col) reads window pixel (row + ky, col + kx)
    const unsigned bbase = (unsigned)(((wn * NT) * WW + col) * PIXB + kg * 32);
    int t = 0, c = 0;
    for (int s2 = 0; s2 < total; s2 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = s2 + u;
            if (s >= total) break;  // (uniform; only in the last round)
            // weights of step s sit in As stage u (= s & 1: s2 is even), of step s + 1 in register slot (u + 1) & 1
            const unsigned char *wsrc = Ws + (c & 1) * (WPIX * PIXB) + bbase + (unsigned)(((t / 3) * WW + (t % 3)) * PIXB);
            f32x4 bh[NT], bl[NT], ah[2], al[2];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bh[n] = *reinterpret_cast<const f32x4 *>(wsrc + n * (WW * PIXB));
                bl[n] = *reinterpret_cast<const f32x4 *>(wsrc + n * (WW * PIXB) + 16);
            }
            ah[0] = As[((u * MB + wm * MT) * 2 + 0) * 64 + lane];
            al[0] = As[((u * MB + wm * MT) * 2 + 1) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            // landed: the weights of step s + 1 (and, at tap 3, the next chunk's window, requested at tap 0 behind tap 0's weights)
            if (t == 1 || t == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW + 2 * WI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW) : "memory");
            stage_weights((u + 1) & 1, wr[(u + 1) & 1]);
            if (t == 3) stage_window((c + 1) & 1);
            issue_weights(wr[(u + 1) & 1]);
            if (t == 0) issue_window(c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m + 1 < MT) {
                    ah[(m + 1) & 1] = As[((u * MB + wm * MT + m + 1) * 2 + 0) * 64 + lane];
                    al[(m + 1) & 1] = As[((u * MB + wm * MT + m + 1) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = mfma3(ah[m & 1], al[m & 1], __builtin_bit_cast(f16x8, bh[n]), __builtin_bit_cast(f16x8, bl[n]), acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (++t == 9) { t = 0; ++c; }
        }
    }
    const int ctw = ct0 + wm * MT;
    if (ctw >= n_ct) return;
    int pidx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int y = y0 + wn * NT + n, x = x0 + col;
        pidx[n] = (y < H && x < W) ? (img * H + y) * W + x : -1;
    }
    f32x4 rvs[MT], bvs[MT];
    seg_vectors<MT>(a, ctw, kg, rvs, bvs);
    seg_epilogue_px<MT, NT, DROP>(a, acc, ctw, pidx, kg, rvs, bvs);
}

inline float pow2_row_scale(float row_max)
{
    if (!(row_max > 0.0f) || !std::isfinite(row_max)) return 1.0f;
    int e = 0;
    (void)std::frexp(row_max, &e);
    int k = 14 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return std::ldexp(1.0f, k);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace
}  // namespace ojf

struct ojf_segconv {
    int c_in, c_out, ksize, stride, dil, pad;
    int c8, n_kb, n_ct;
    ojf::f32x4 *wp;
    float *rinv, *bias;
    int up, up_c, up_cp;  // ojf_segdeconv_create: upscale factor, channels per phase (real / padded to 4)
    const unsigned long long *rng = nullptr;  // ojf_segconv_set_dropout
    unsigned long long *rng_bump = nullptr;
    unsigned drop_id = 0;
};

OJF_API int ojf_segconv_create(ojf_segconv **out, const float *weight, const float *scale, const float *bias, int c_in, int c_out,
                               int ksize, int stride, int dilation, int padding)
{
    using namespace ojf;
    if (!out || !weight) return fail("ojf_segconv_create: null pointer argument");
    if (c_in < 1 || c_out < 1 || ksize < 1 || ksize > 7 || stride < 1 || dilation < 1 || padding < 0)
        return fail("ojf_segconv_create: bad layer geometry");
    const int taps = ksize * ksize, c8 = round_up(c_in, 8) / 8;
    const int n_entries = taps * c8, n_kb = (n_entries + 3) / 4, n_ct = round_up((c_out + 15) / 16, kMW);
    const size_t n_w = (size_t)c_out * c_in * taps;
    for (size_t i = 0; i < n_w; ++i)
        if (!std::isfinite(weight[i])) return fail("ojf_segconv_create: non-finite weight");
    std::vector<_Float16> packed((size_t)n_ct * n_kb * 2 * 64 * 8, (_Float16)0.0f);
    std::vector<float> rinv((size_t)n_ct * 16, 1.0f), b((size_t)n_ct * 16, 0.0f);
    for (int oc = 0; oc < c_out; ++oc) {
        const float s = scale ? scale[oc] : 1.0f;
        if (!std::isfinite(s) || (bias && !std::isfinite(bias[oc]))) return fail("ojf_segconv_create: non-finite scale or bias");
        float mx = 0.0f;
        for (size_t i = 0; i < (size_t)c_in * taps; ++i) mx = std::fmax(mx, std::fabs(weight[(size_t)oc * c_in * taps + i] * s));
        const float rs = pow2_row_scale(mx);
        rinv[oc] = 1.0f / rs;
        b[oc] = bias ? bias[oc] : 0.0f;
        const int ct = oc / 16, row = oc % 16;
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < c_in; ++ci) {
                const float w = weight[((size_t)oc * c_in + ci) * taps + t] * s * rs;  // BN scale folded, row equilibrated
                const int e = t * c8 + ci / 8, j = ci % 8;                               // entry, element
                const int kb = e / 4, kg = e % 4, lane = kg * 16 + row;
                const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
                const size_t base = (((size_t)ct * n_kb + kb) * 2) * 64;
                packed[(base + lane) * 8 + j] = hi;
                packed[(base + 64 + lane) * 8 + j] = lo;
            }
    }
    ojf_segconv *c = new ojf_segconv{c_in, c_out, ksize, stride, dilation, padding, c8, n_kb, n_ct, nullptr, nullptr, nullptr, 1, 0, 0};
    int rc = check_hip(hipMalloc(&c->wp, packed.size() * sizeof(_Float16)), "hipMalloc(segconv weights)");
    if (!rc) rc = check_hip(hipMalloc(&c->rinv, rinv.size() * sizeof(float)), "hipMalloc(segconv rinv)");
    if (!rc) rc = check_hip(hipMalloc(&c->bias, b.size() * sizeof(float)), "hipMalloc(segconv bias)");
    if (!rc) rc = check_hip(hipMemcpy(c->wp, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice), "segconv H2D");
    if (!rc) rc = check_hip(hipMemcpy(c->rinv, rinv.data(), rinv.size() * sizeof(float), hipMemcpyHostToDevice), "segconv H2D");
    if (!rc) rc = check_hip(hipMemcpy(c->bias, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice), "segconv H2D");
    if (rc) {
        (void)hipFree(c->wp); (void)hipFree(c->rinv); (void)hipFree(c->bias);
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

// ConvTranspose2d(kernel 2s, stride s, padding s/2) = a 3x3 convolution (padding 1) of the input to s*s "phase" copies
// of the channels + a pixel shuffle (done by the kernel's store): output row oy = s*y + ay of phase ay takes input rows
// y + dy, dy in {-1, 0, 1}, through kernel row ky = ay + s/2 - s*dy when that exists (two of the three dy do).
OJF_API int ojf_segdeconv_create(ojf_segconv **out, const float *weight, const float *scale, const float *bias, int c_in, int c_out,
                                 int stride)
{
    using namespace ojf;
    if (!out || !weight) return fail("ojf_segdeconv_create: null pointer argument");
    if (c_in < 1 || c_out < 1 || stride < 2 || stride % 2 || stride > 8) return fail("ojf_segdeconv_create: stride must be 2, 4, 6 or 8");
    const int s = stride, k = 2 * s, pad = s / 2, cp = round_up(c_out, 4), rows = s * s * cp;
    std::vector<float> w((size_t)rows * c_in * 9, 0.0f), sc((size_t)rows, 1.0f), b((size_t)rows, 0.0f);
    for (int ay = 0; ay < s; ++ay)
        for (int ax = 0; ax < s; ++ax)
            for (int co = 0; co < c_out; ++co) {
                const int r = (ay * s + ax) * cp + co;
                sc[r] = scale ? scale[co] : 1.0f;
                b[r] = bias ? bias[co] : 0.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int ky = ay + pad - s * dy, kx = ax + pad - s * dx;
                        if (ky < 0 || ky >= k || kx < 0 || kx >= k) continue;
                        for (int ci = 0; ci < c_in; ++ci)  // torch layout [c_in][c_out][k][k]
                            w[((size_t)r * c_in + ci) * 9 + (dy + 1) * 3 + (dx + 1)] = weight[(((size_t)ci * c_out + co) * k + ky) * k + kx];
                    }
            }
    int rc = ojf_segconv_create(out, w.data(), sc.data(), b.data(), c_in, rows, 3, 1, 1, 1);
    if (rc) return rc;
    (*out)->up = s;
    (*out)->up_c = c_out;
    (*out)->up_cp = cp;
    return 0;
}

OJF_API int ojf_segconv_set_dropout(ojf_segconv *c, const unsigned long long *rng_state_dev, unsigned stream_id, int advance)
{
    if (!c) return ojf::fail("ojf_segconv_set_dropout: null layer");
    if (advance && !rng_state_dev) return ojf::fail("ojf_segconv_set_dropout: advance needs the state");
    c->rng = advance ? nullptr : rng_state_dev;
    c->rng_bump = advance ? const_cast<unsigned long long *>(rng_state_dev) : nullptr;
    c->drop_id = stream_id;
    return 0;
}

OJF_API void ojf_segconv_destroy(ojf_segconv *c)
{
    if (!c) return;
    (void)hipFree(c->wp); (void)hipFree(c->rinv); (void)hipFree(c->bias);
    delete c;
}

namespace ojf {
namespace {

// argument checks + kernel arguments of one convolution; the launch geometry comes back in Ho / Wo of `a`
int seg_fill(const ojf_segconv *c, int batch, const float *in, int in_stride, float *out, int out_stride, const float *res, int res_stride,
             const float *mul, int mul_stride, int act, int h, int w, SegArgs &a)
{
    if (batch < 1 || batch > 64) return fail("ojf_segconv_forward: batch must be 1..64");
    if (!c || !in || !out) return fail("ojf_segconv_forward: null pointer argument");
    const bool zero_pad = (act & OJF_SEG_ACT_ZERO_PAD) != 0;
    act &= ~OJF_SEG_ACT_ZERO_PAD;
    if (h < 1 || w < 1 || act < 0 || act > 2) return fail("ojf_segconv_forward: bad size or activation");
    if (in_stride < c->c8 * 8 || in_stride % 4 || (reinterpret_cast<uintptr_t>(in) & 15))
        return fail("ojf_segconv_forward: input rows must hold round_up(c_in, 8) channels, 16-byte aligned");
    if (c->up > 1 ? (out_stride < c->up_c || res || mul) : (out_stride < c->c_out || (res && res_stride < c->c_out) || (mul && mul_stride < c->c_out)))
        return fail("ojf_segconv_forward: row stride smaller than c_out (or residual / gate given to a transposed convolution)");
    const int span = c->dil * (c->ksize - 1) + 1;
    const int Ho = (h + 2 * c->pad - span) / c->stride + 1, Wo = (w + 2 * c->pad - span) / c->stride + 1;
    if (Ho < 1 || Wo < 1) return fail("ojf_segconv_forward: empty output");
    const size_t in_bytes = ((size_t)batch * h * w - 1) * in_stride * 4 + (size_t)c->c8 * 32;
    if (in_bytes >= 0xfffffff0ull) return fail("ojf_segconv_forward: input larger than 4 GB");
    a.in = in; a.out = out; a.res = res; a.mul = mul; a.wp = c->wp; a.rinv = c->rinv; a.bias = c->bias;
    a.in_stride = in_stride; a.out_stride = out_stride; a.res_stride = res_stride; a.mul_stride = mul_stride;
    a.H = h; a.W = w; a.Ho = Ho; a.Wo = Wo; a.B = batch; a.stride = c->stride; a.pad = c->pad; a.dil = c->dil; a.ksize = c->ksize;
    a.c8 = c->c8; a.n_kb = c->n_kb; a.n_ct = c->n_ct; a.c_out = c->c_out; a.act = act;
    // bit 0: float4 stores, bit 1: float4 residual loads, bit 2: float4 gate loads (rows 16-byte aligned)
    a.vec_store = ((out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0) |
                  ((res && res_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0) ? 2 : 0) |
                  ((mul && mul_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(mul) & 15) == 0) ? 4 : 0);
    a.in_bytes = (unsigned)in_bytes;
    a.ovf = range_flag_device();
    a.up = c->up; a.up_c = c->up_c; a.up_cp = c->up_cp;
    a.pad_to = a.c_out;
    if (zero_pad) {
        if (c->up > 1) return fail("ojf_segconv_forward: OJF_SEG_ACT_ZERO_PAD is not defined for transposed convolutions");
        a.pad_to = round_up(c->c_out, 8);
        if (out_stride < a.pad_to) return fail("ojf_segconv_forward: OJF_SEG_ACT_ZERO_PAD needs rows of round_up(c_out, 8) floats");
    }
    a.rng = c->rng; a.rng_bump = c->rng_bump; a.drop_id = c->drop_id;
    if (a.rng && c->up > 1) return fail("ojf_segconv_forward: dropout on a transposed convolution");
    return 0;
}

// 1-D launch geometry of the logical grid X x Y x Z (see SegMap)
unsigned seg_map(SegMap &m, int X, int Y, int Z)
{
    static const int xcd = getenv("OJF_SEG_XCD") ? atoi(getenv("OJF_SEG_XCD")) : 1;  // A/B switch (0: plain numbering)
    m.X = X; m.Y = Y; m.Z = Z;
    if (!xcd) {
        m.S = 0; m.chunk = X; m.V = Y * Z;
        return (unsigned)(X * Y * Z);
    }
    const int Q = Y * Z;
    int S = 1;
    while (S < 8 && (Q * S) % 8 != 0 && (X + 2 * S - 1) / (2 * S) >= 2) S *= 2;  // chunks of >= 2 pixel blocks
    m.S = S; m.chunk = (X + S - 1) / S; m.V = Q * S;
    return (unsigned)((m.V + 7) / 8 * 8 * m.chunk);
}

// n members of one shape (the first one's n_kb / n_ct / output size decide the launch)
int seg_launch(SegGroupArgs &g, int n, hipStream_t st)
{
    g.het.n = 0;
    static const int abl = getenv("OJF_SEG_ABL") ? atoi(getenv("OJF_SEG_ABL")) : 0;  // tuning only
    g.abl = abl;
    const SegArgs &a = g.a[0];
    const int n_pt = (a.B * a.Ho * a.Wo + 15) / 16, groups = a.n_ct / kMW;  // pixel tiles, 64-channel groups
    // Enough independent waves (>= 4 per CU) to hide the operand latency: waves own their (channels, pixels) pair.
    // Otherwise the four waves of a block split K (when K is long enough to be worth the LDS reduction).
    const long waves2 = (long)groups * ((n_pt + 1) / 2) * n;
    static const int force_mw = getenv("OJF_SEG_MW") ? atoi(getenv("OJF_SEG_MW")) : 0;  // tuning only
    static const int no_wide = getenv("OJF_SEG_NO_WIDE") ? atoi(getenv("OJF_SEG_NO_WIDE")) : 0;  // tuning only
    static const int wide_min = getenv("OJF_SEG_WIDE_MIN") ? atoi(getenv("OJF_SEG_WIDE_MIN")) : 256;  // tuning only
    static const int trace = getenv("OJF_SEG_TRACE") ? atoi(getenv("OJF_SEG_TRACE")) : 0;  // tuning only: one line per launch
    static const int wide1_min = getenv("OJF_SEG_WIDE1_MIN") ? atoi(getenv("OJF_SEG_WIDE1_MIN")) : (1 << 30);  // tuning only
    static const int splitk_nw2_min = getenv("OJF_SEG_SPLITK_NW2_MIN") ? atoi(getenv("OJF_SEG_SPLITK_NW2_MIN")) : (1 << 30);  // tuning only
    static const int nw2_min_kb = getenv("OJF_SEG_NW2_MIN_KB") ? atoi(getenv("OJF_SEG_NW2_MIN_KB")) : 32;  // tuning only
    static const int nw2_max_blocks = getenv("OJF_SEG_NW2_MAX_BLOCKS") ? atoi(getenv("OJF_SEG_NW2_MAX_BLOCKS")) : 400;  // tuning only (0: never)
    static const int plain_nw1_min = getenv("OJF_SEG_PLAIN_NW1_MIN") ? atoi(getenv("OJF_SEG_PLAIN_NW1_MIN")) : 50;  // one pixel tile per wave in the plain form from 50 pixel tiles on (twice the waves in flight for layers that are all latency: -1.4 % of the frame)
    static const int splitk_min_kb = getenv("OJF_SEG_SPLITK_MIN_KB") ? atoi(getenv("OJF_SEG_SPLITK_MIN_KB")) : 8;  // tuning only
    static const int tile_u = getenv("OJF_SEG_TILE_U") ? atoi(getenv("OJF_SEG_TILE_U")) : 1;  // tuning only: K blocks per barrier of the tile kernel
    static const int use_tile = getenv("OJF_SEG_TILE") ? atoi(getenv("OJF_SEG_TILE")) : 0;  // tuning only: register-staged tile kernel
    static const int wide_depth = getenv("OJF_SEG_WIDE_DEPTH") ? atoi(getenv("OJF_SEG_WIDE_DEPTH")) : 3;  // tuning only: 3 | 6 | 8
    const char *variant;
    // GEMM-shaped form (segconv_gemm_kernel) where both the channels and the pixels tile: the 128 x 128 tile when that alone
    // gives gemm_min blocks (measured 118 vs 176 us at 300 blocks, but 68 vs 51 at 75), else the 64 x 64 tile from gemm22_min
    // blocks on (30 vs 38 us at 152 blocks, 23 vs 28 at 640; 53 vs 33 at 76).  A launch's time is a step function of its
    // block count (120 .. 256 blocks of one layer 29 .. 31 us, 260 .. 512 blocks 43 .. 54 us, 520 blocks 69 us), so the
    // 128 x 160 tile takes over where it saves a round of 256 blocks (four frames of a 60x80 map: 240 blocks for 300, 94 vs
    // 103 us).  Measured and not in the menu: 64 x 80 / 64 x 96 / 64 x 160 tiles (44 / 46 / 65 us against 46 for 64 x 64 on the
    // 60x80 256 -> 256 3x3 layer of one frame, although 240 / 200 / 120 blocks instead of 300: a block with more work per K
    // block is slower by more than its share), the 64 x 128 tile (never won a layer).
    static const int gemm_min = getenv("OJF_SEG_GEMM_MIN") ? atoi(getenv("OJF_SEG_GEMM_MIN")) : 256;  // 128 x 128 tile alone gives this many blocks (1 << 30: off)
    static const int gemm22_min = getenv("OJF_SEG_GEMM22_MIN") ? atoi(getenv("OJF_SEG_GEMM22_MIN")) : 128;  // ... or the 64 x 64 tile this many
    static const int gemm_min_kb = getenv("OJF_SEG_GEMM_MIN_KB") ? atoi(getenv("OJF_SEG_GEMM_MIN_KB")) : 4;
    static const int gemm_shape = getenv("OJF_SEG_GEMM_SHAPE") ? atoi(getenv("OJF_SEG_GEMM_SHAPE")) : -1;  // tuning: force menu entry
    static const int gemm_menu = getenv("OJF_SEG_GEMM_MENU") ? atoi(getenv("OJF_SEG_GEMM_MENU")) : 0xff;  // tuning: bit 2 = the 128 x 160 tile allowed, bit 3 = the long-K model
    {
        bool drop_all = true, drop_any = false;
        for (int i = 0; i < n; ++i) { drop_any = drop_any || g.a[i].rng; drop_all = drop_all && g.a[i].rng; }
        const long b44 = (long)((a.n_ct + 7) / 8) * ((n_pt + 7) / 8) * n, b22 = (long)((a.n_ct + 3) / 4) * ((n_pt + 3) / 4) * n;
        bool aligned = true;  // every member's channel groups come in fours: the scalar tap walk
        for (int i = 0; i < n; ++i) aligned = aligned && (g.a[i].c8 % 4) == 0;
#define OJF_WS_LAUNCH(MT_, NT_, GRID_)                                                                                                \
    do {                                                                                                                             \
        const dim3 grid__ = GRID_;                                                                                                   \
        if (drop_any && aligned) hipLaunchKernelGGL((segconv_ws_kernel<MT_, NT_, true, true>), grid__, dim3(512), 0, st, g);          \
        else if (drop_any) hipLaunchKernelGGL((segconv_ws_kernel<MT_, NT_, true, false>), grid__, dim3(512), 0, st, g);               \
        else if (aligned) hipLaunchKernelGGL((segconv_ws_kernel<MT_, NT_, false, true>), grid__, dim3(512), 0, st, g);                \
        else hipLaunchKernelGGL((segconv_ws_kernel<MT_, NT_, false, false>), grid__, dim3(512), 0, st, g);                            \
    } while (0)
#define OJF_GEMM_LAUNCH(MT_, NT_, WM_, GRID_)                                                                                         \
    do {                                                                                                                             \
        const dim3 grid__ = GRID_;                                                                                                   \
        if (drop_any && aligned) hipLaunchKernelGGL((segconv_gemm_kernel<MT_, NT_, true, true, WM_>), grid__, dim3(256), 0, st, g);   \
        else if (drop_any) hipLaunchKernelGGL((segconv_gemm_kernel<MT_, NT_, true, false, WM_>), grid__, dim3(256), 0, st, g);        \
        else if (aligned) hipLaunchKernelGGL((segconv_gemm_kernel<MT_, NT_, false, true, WM_>), grid__, dim3(256), 0, st, g);         \
        else hipLaunchKernelGGL((segconv_gemm_kernel<MT_, NT_, false, false, WM_>), grid__, dim3(256), 0, st, g);                     \
    } while (0)
        // WINDOW form (segconv_win_kernel): 3x3 / stride 1 / dilation 1 layers with c_in % 32 == 0 and rows of >= 40 pixels (or a multiple of 16)
        static const int win = getenv("OJF_SEG_WIN") ? atoi(getenv("OJF_SEG_WIN")) : 0;  // 0 off | 1 by block count | 2 force 64 ch x 4 rows | 3 force 128 ch x 8 rows
        if (win && a.ksize == 3 && a.stride == 1 && a.dil == 1 && a.pad == 1 && aligned && !drop_any && a.c8 >= 8 && (a.W % 16 == 0 || a.W >= 40)) {
            const int tx = (a.W + 15) / 16;
            const long blocksB = (long)a.B * ((a.H + 7) / 8) * tx * ((a.n_ct + 7) / 8) * n, blocksA = (long)a.B * ((a.H + 3) / 4) * tx * ((a.n_ct + 3) / 4) * n;
            const int pick = win == 2 ? 0 : (win == 3 ? 1 : (blocksB >= 192 ? 1 : (blocksA >= 128 ? 0 : -1)));
            if (pick >= 0) {
                g.het.n = 0;
                static bool configured = false;
                constexpr int ldsA = 2 * (18 * 6) * 144 + 2 * 4 * 2 * 64 * 16, ldsB = 2 * (18 * 10) * 144 + 2 * 8 * 2 * 64 * 16;
                if (!configured) {
                    if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&segconv_win_kernel<4, 8, false>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, ldsB), "segconv_win_kernel LDS")) return rc;
                    if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&segconv_win_kernel<2, 4, false>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, ldsA), "segconv_win_kernel LDS")) return rc;
                    configured = true;
                }
                if (pick) {
                    variant = "win 128x(8x16)";
                    hipLaunchKernelGGL((segconv_win_kernel<4, 8, false>), dim3(seg_map(g.map, a.B * ((a.H + 7) / 8) * tx, (a.n_ct + 7) / 8, n)), dim3(256), ldsB, st, g);
                } else {
                    variant = "win 64x(4x16)";
                    hipLaunchKernelGGL((segconv_win_kernel<2, 4, false>), dim3(seg_map(g.map, a.B * ((a.H + 3) / 4) * tx, (a.n_ct + 3) / 4, n)), dim3(256), ldsA, st, g);
                }
                if (trace)
                    fprintf(stderr, "segconv %-14s n %d  c_in %4d c_out %4d k 3  in %3dx%3d B %d  n_kb %4d  grid %dx%dx%d%s\n", variant, n, a.c8 * 8, a.c_out, a.H, a.W, a.B,
                            a.n_kb, g.map.X, g.map.Y, g.map.Z, a.up > 1 ? " deconv" : "");
                return check_hip(hipGetLastError(), "segconv_win_kernel launch");
            }
        }
        struct Shape { int a, b; const char *name; };  // channel / pixel tiles of the block
        static const Shape menu[] = {{4, 4, "gemm 64x64"}, {8, 8, "gemm 128x128"}, {8, 10, "gemm 128x160"}, {8, 5, "gemm 128x80"}};
        const bool big = b44 >= gemm_min && a.n_ct >= 8;
        if (a.n_kb >= gemm_min_kb && (drop_all || !drop_any) && (big || b22 >= gemm22_min)) {
            int best = big ? 1 : 0;
            if (big && ((gemm_menu >> 2) & 1)) {
                const long b810 = (long)((a.n_ct + 7) / 8) * ((n_pt + 9) / 10) * n;
                if (b810 >= 200 && (b810 + 255) / 256 < (b44 + 255) / 256) best = 2;
            }
            // long K (3x3 layers from 256 input channels on): the tile by a two-factor model of the measurements - a block of
            // each shape alone on a CU (us per 72 K blocks: 28 / 59.5 / 94.5 / 51), times what the launch's block count costs
            // in rounds of 256 (two blocks share a CU at 1.65x the time of one: 1, 1.65, 2.65, 3.3, 4.3 ... for 1, 2, 3, 4, 5
            // rounds).  Checked against the layers it re-decides (profiles/r05_seg_experiments.txt): 60x80 256 -> 256 at 2 / 4
            // frames 128x80 (57 against 70 us, 87 against 94), 15x20 512 -> 512 of both encoders at 8 frames 128x80 (108 against 127).
            if (a.n_kb >= 64 && ((gemm_menu >> 3) & 1)) {
                static const double alone[4] = {28.0, 59.5, 94.5, 51.0};
                double best_t = 0.0;
                int pick = -1;
                for (int i = 0; i < 4; ++i) {
                    if (menu[i].a == 8 && a.n_ct < 8) continue;
                    const long blocks = (long)((a.n_ct + menu[i].a - 1) / menu[i].a) * ((n_pt + menu[i].b - 1) / menu[i].b) * n;
                    if (blocks < 100) continue;
                    const long k = (blocks + 255) / 256;
                    const double t = alone[i] * (1.65 * (double)(k / 2) + (double)(k & 1));
                    if (pick < 0 || t < best_t) { pick = i; best_t = t; }
                }
                if (pick >= 0) best = pick;
            }
            if (gemm_shape >= 0) best = gemm_shape > 3 ? 3 : gemm_shape;
            if (best >= 0) {
                g.het.n = 0;
                const Shape &sh = menu[best];
                variant = sh.name;
                const dim3 grid(seg_map(g.map, (n_pt + sh.b - 1) / sh.b, (a.n_ct + sh.a - 1) / sh.a, n));
                static const int ws = getenv("OJF_SEG_WS") ? atoi(getenv("OJF_SEG_WS")) : 0;  // producer / consumer form (round 6)
                if (ws && best <= 1) {
                    variant = best ? "ws 128x128" : "ws 64x64";
                    if (best) OJF_WS_LAUNCH(4, 4, grid); else OJF_WS_LAUNCH(2, 2, grid);
                    if (trace)
                        fprintf(stderr, "segconv %-12s n %d  c_in %4d c_out %4d k %d  in %3dx%3d  n_kb %4d  grid %dx%dx%d\n", variant, n, a.c8 * 8, a.c_out, a.ksize,
                                a.H, a.W, a.n_kb, g.map.X, g.map.Y, g.map.Z);
                    return check_hip(hipGetLastError(), "segconv_ws_kernel launch");
                }
                switch (best) {
                case 0: OJF_GEMM_LAUNCH(2, 2, 2, grid); break;
                case 1: OJF_GEMM_LAUNCH(4, 4, 2, grid); break;
                case 2: OJF_GEMM_LAUNCH(4, 5, 2, grid); break;
                default: OJF_GEMM_LAUNCH(2, 5, 4, grid); break;
                }
                if (trace)
                    fprintf(stderr, "segconv %-12s n %d  c_in %4d c_out %4d k %d s %d d %2d  in %3dx%3d out %3dx%3d  n_kb %4d  grid %dx%dx%d S %d%s%s%s\n", variant, n,
                            a.c8 * 8, a.c_out, a.ksize, a.stride, a.dil, a.H, a.W, a.Ho, a.Wo, a.n_kb, g.map.X, g.map.Y, g.map.Z, g.map.S,
                            a.res ? " +res" : "", a.mul ? " *mul" : "", a.up > 1 ? " deconv" : "");
                return check_hip(hipGetLastError(), "segconv_gemm_kernel launch");
            }
        }
    }
    // a layer with the always-on dropout in its epilogue (the last convolution of a multi-scale unit): the DROP instantiation of
    // the same kernel - the plain ones do not carry the generator's code (it cost every launch ~0.6 us)
    bool drop = false;
    for (int i = 0; i < n; ++i) drop = drop || g.a[i].rng != nullptr;
    if (drop) {
        for (int i = 0; i < n; ++i)
            if (!g.a[i].rng) return fail("ojf_segconv_forward_group: dropout on some members only");
        int mw = 4;
        while (mw > 1 && (long)n_pt * (a.n_ct / mw) * n < 150) mw /= 2;
        if (mw == 1) hipLaunchKernelGGL((segconv_kernel<1, 1, 1, 4, 3, true>), dim3(seg_map(g.map, n_pt, a.n_ct, n)), dim3(256), 0, st, g);
        else if (mw == 2) hipLaunchKernelGGL((segconv_kernel<2, 1, 1, 4, 3, true>), dim3(seg_map(g.map, n_pt, a.n_ct / 2, n)), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((segconv_kernel<4, 1, 1, 4, 3, true>), dim3(seg_map(g.map, n_pt, groups, n)), dim3(256), 0, st, g);
        variant = "<*,1,1,4> drop";
    } else if (!no_wide && a.n_kb >= 6 && (long)groups * ((n_pt + 7) / 8) * n >= wide_min) {
        variant = use_tile ? "tile<2>" : "wide<2>";
        if (use_tile && tile_u == 4) hipLaunchKernelGGL((segconv_tile_kernel<2, 4>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
        else if (use_tile && tile_u == 2) hipLaunchKernelGGL((segconv_tile_kernel<2, 2>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
        else if (use_tile) hipLaunchKernelGGL((segconv_tile_kernel<2, 1>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
        else if (wide_depth == 3) hipLaunchKernelGGL((segconv_wide_kernel<2, 3>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
        else if (wide_depth == 8) hipLaunchKernelGGL((segconv_wide_kernel<2, 8>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((segconv_wide_kernel<2, 6>), dim3(seg_map(g.map, (n_pt + 7) / 8, groups, n)), dim3(256), 0, st, g);
    } else if (!no_wide && a.n_kb >= 6 && (long)groups * ((n_pt + 3) / 4) * n >= wide1_min) {
        variant = use_tile ? "tile<1>" : "wide<1>";
        if (use_tile && tile_u == 4) hipLaunchKernelGGL((segconv_tile_kernel<1, 4>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
        else if (use_tile && tile_u == 2) hipLaunchKernelGGL((segconv_tile_kernel<1, 2>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
        else if (use_tile) hipLaunchKernelGGL((segconv_tile_kernel<1, 1>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
        else if (wide_depth == 3) hipLaunchKernelGGL((segconv_wide_kernel<1, 3>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
        else if (wide_depth == 8) hipLaunchKernelGGL((segconv_wide_kernel<1, 8>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((segconv_wide_kernel<1, 6>), dim3(seg_map(g.map, (n_pt + 3) / 4, groups, n)), dim3(256), 0, st, g);
    } else if (waves2 >= 1024 || a.n_kb < splitk_min_kb) {
        if (groups == 1) {
            variant = "<4,2,1,1>";
            hipLaunchKernelGGL((segconv_kernel<4, 2, 1, 1, 3>), dim3(seg_map(g.map, (n_pt + 7) / 8, 1, n)), dim3(256), 0, st, g);
        } else if (n_pt >= plain_nw1_min) {
            variant = "<4,1,2,1>";
            hipLaunchKernelGGL((segconv_kernel<4, 1, 2, 1, 3>), dim3(seg_map(g.map, (n_pt + 1) / 2, (groups + 1) / 2, n)), dim3(256), 0, st, g);
        } else {
            variant = "<4,2,2,1>";
            hipLaunchKernelGGL((segconv_kernel<4, 2, 2, 1, 3>), dim3(seg_map(g.map, (n_pt + 3) / 4, (groups + 1) / 2, n)), dim3(256), 0, st, g);
        }
    } else {
        // few pixels: the 4 waves of a block split K.  Channel tiles per block: as many as leave >= 150 blocks (measured
        // per layer shape on the 15x20 / 30x40 maps: one CU cannot pull a block's operands faster than ~150 GB/s, so
        // small layers want many small blocks; cross-block K splitting is not an option - the device-scope fence it
        // needs writes back the whole L2 and doubled the frame time)
        int mw = 4;
        while (mw > 1 && (long)n_pt * (a.n_ct / mw) * n < 150) mw /= 2;
        if (force_mw) mw = force_mw;
        if (mw == 1) {
            variant = "<1,1,1,4>";
            hipLaunchKernelGGL((segconv_kernel<1, 1, 1, 4, 3>), dim3(seg_map(g.map, n_pt, a.n_ct, n)), dim3(256), 0, st, g);
        } else if (mw == 2) {
            variant = "<2,1,1,4>";
            hipLaunchKernelGGL((segconv_kernel<2, 1, 1, 4, 3>), dim3(seg_map(g.map, n_pt, a.n_ct / 2, n)), dim3(256), 0, st, g);
        } else if (n_pt >= splitk_nw2_min || (a.n_kb >= nw2_min_kb && (long)((n_pt + 1) / 2) * groups * n >= 150 && (long)((n_pt + 1) / 2) * groups * n < nw2_max_blocks)) {
            // two pixel tiles per wave (each weight fragment feeds twice the MFMAs) where that still leaves 150 .. 400 blocks:
            // measured per layer (round 5): -1 .. -3.4 us on the 30x40 3x3 layers, layer4's 3x3 and 2048 -> 512, the first
            // transposed convolution; slower with fewer blocks (15x20 maps of 256 channels) and with many (60x80 maps)
            variant = "<4,2,1,4>";
            hipLaunchKernelGGL((segconv_kernel<4, 2, 1, 4, 3>), dim3(seg_map(g.map, (n_pt + 1) / 2, groups, n)), dim3(256), 0, st, g);
        } else {
            variant = "<4,1,1,4>";
            hipLaunchKernelGGL((segconv_kernel<4, 1, 1, 4, 3>), dim3(seg_map(g.map, n_pt, groups, n)), dim3(256), 0, st, g);
        }
    }
    if (trace)
        fprintf(stderr, "segconv %-10s n %d  c_in %4d c_out %4d k %d s %d d %2d  in %3dx%3d out %3dx%3d  n_kb %4d  grid %dx%dx%d S %d%s%s%s\n", variant, n,
                a.c8 * 8, a.c_out, a.ksize, a.stride, a.dil, a.H, a.W, a.Ho, a.Wo, a.n_kb, g.map.X, g.map.Y, g.map.Z, g.map.S,
                a.res ? " +res" : "", a.mul ? " *mul" : "", a.up > 1 ? " deconv" : "");
    return check_hip(hipGetLastError(), "segconv_kernel launch");
}

// Heterogeneous members (ojf_segconv_forward_multi): convolutions of DIFFERENT shapes that do not depend on each other - a
// unit's shortcut convolution next to its first 1x1, an encoder's skip projection next to the next unit's first layers, the three
// SSMA blocks of the decoder in lock-step - as ONE launch: every member keeps its own arguments (SegArgs is per member), the
// grid covers the largest member and the blocks beyond a member's channel / pixel blocks exit.  One kernel form for all: the
// in-block split-K form when every member would take it on its own, the plain form when none would; a mix runs as separate
// launches (the forms differ in what a block is).  Launches cost ~4.7 us each before any work: 17 fewer per frame.
int seg_launch_multi(SegGroupArgs &g, int n, hipStream_t st)
{
    static const int no_multi = getenv("OJF_SEG_NO_MULTI") ? atoi(getenv("OJF_SEG_NO_MULTI")) : 0;  // A/B switch
    static const int trace = getenv("OJF_SEG_TRACE") ? atoi(getenv("OJF_SEG_TRACE")) : 0;
    int n_split = 0, n_plain = 0, X1 = 0, Y1 = 0, X2 = 0, Y2 = 0;
    long sk_blocks4 = 0;
    bool special = false;
    // members of one shape (the two encoders' copies of a layer) form a natural group: the form each would take is judged with
    // the size of ITS group, as ojf_segconv_forward_group would
    auto same_shape = [&](int i, int j) {
        const SegArgs &a = g.a[i], &b = g.a[j];
        return a.n_kb == b.n_kb && a.n_ct == b.n_ct && a.c8 == b.c8 && a.c_out == b.c_out && a.ksize == b.ksize && a.stride == b.stride && a.Ho == b.Ho &&
               a.Wo == b.Wo && a.up == b.up && a.in_stride == b.in_stride && a.out_stride == b.out_stride && a.res_stride == b.res_stride &&
               a.mul_stride == b.mul_stride && a.act == b.act && a.pad_to == b.pad_to && (a.res != nullptr) == (b.res != nullptr) && (a.mul != nullptr) == (b.mul != nullptr);
    };
    for (int i = 0; i < n; ++i) {
        const SegArgs &a = g.a[i];
        const int n_pt = (a.B * a.Ho * a.Wo + 15) / 16, groups = a.n_ct / kMW;
        int same = 0;
        for (int j = 0; j < n; ++j) same += same_shape(i, j) ? 1 : 0;
        special = special || a.rng != nullptr || a.rng_bump != nullptr;
        const long waves2 = (long)groups * ((n_pt + 1) / 2) * same;
        if (a.n_kb >= 6 && (long)groups * ((n_pt + 7) / 8) * same >= 256) special = true;  // (a wide-kernel layer: with its own group)
        if (waves2 >= 1024 || a.n_kb < 8) ++n_plain; else ++n_split;
        X1 = n_pt > X1 ? n_pt : X1;
        Y1 = a.n_ct > Y1 ? a.n_ct : Y1;
        X2 = (n_pt + 3) / 4 > X2 ? (n_pt + 3) / 4 : X2;
        Y2 = (groups + 1) / 2 > Y2 ? (groups + 1) / 2 : Y2;
        sk_blocks4 += (long)n_pt * groups;
    }
    if (no_multi || special || (n_split && n_plain)) {  // the natural groups, one launch each
        bool done[kSegGroup] = {};
        for (int i = 0; i < n; ++i) {
            if (done[i]) continue;
            SegGroupArgs grp;
            int m = 0;
            for (int j = i; j < n; ++j)
                if (!done[j] && same_shape(i, j)) { grp.a[m++] = g.a[j]; done[j] = true; }
            if (int rc = seg_launch(grp, m, st)) return rc;
        }
        return 0;
    }
    // every member gets its own block map (its own pixel x channel blocks, XCD-aware) and a range of the 1-D grid
    const char *variant;
    int mw = 4;
    if (!n_plain)
        while (mw > 1 && sk_blocks4 * (4 / mw) < 150) mw /= 2;
    g.het.n = n;
    g.het.off[0] = 0;
    for (int i = 0; i < n; ++i) {
        const SegArgs &a = g.a[i];
        const int n_pt = (a.B * a.Ho * a.Wo + 15) / 16, groups = a.n_ct / kMW;
        const unsigned blocks = n_plain ? seg_map(g.het.map[i], (n_pt + 3) / 4, (groups + 1) / 2, 1) : seg_map(g.het.map[i], n_pt, a.n_ct / mw, 1);
        g.het.off[i + 1] = g.het.off[i] + (int)((blocks + 7) / 8 * 8);
    }
    for (int i = n + 1; i <= kSegGroup; ++i) g.het.off[i] = g.het.off[n];
    g.map = g.het.map[0];
    const dim3 grid((unsigned)g.het.off[n]);
    if (n_plain) { variant = "multi<4,2,2,1>"; hipLaunchKernelGGL((segconv_kernel<4, 2, 2, 1, 3>), grid, dim3(256), 0, st, g); }
    else if (mw == 1) { variant = "multi<1,1,1,4>"; hipLaunchKernelGGL((segconv_kernel<1, 1, 1, 4, 3>), grid, dim3(256), 0, st, g); }
    else if (mw == 2) { variant = "multi<2,1,1,4>"; hipLaunchKernelGGL((segconv_kernel<2, 1, 1, 4, 3>), grid, dim3(256), 0, st, g); }
    else { variant = "multi<4,1,1,4>"; hipLaunchKernelGGL((segconv_kernel<4, 1, 1, 4, 3>), grid, dim3(256), 0, st, g); }
    if (trace)
        for (int i = 0; i < n; ++i) {
            const SegArgs &a = g.a[i];
            fprintf(stderr, "segconv %-14s member %d/%d  c_in %4d c_out %4d k %d s %d d %2d  in %3dx%3d out %3dx%3d  n_kb %4d  grid %dx%dx%d S %d%s%s\n", variant, i, n,
                    a.c8 * 8, a.c_out, a.ksize, a.stride, a.dil, a.H, a.W, a.Ho, a.Wo, a.n_kb, g.het.map[i].X, g.het.map[i].Y, g.het.off[n], g.het.map[i].S, a.res ? " +res" : "", a.mul ? " *mul" : "");
        }
    return check_hip(hipGetLastError(), "segconv_kernel launch (multi)");
}

}  // namespace
}  // namespace ojf

OJF_API int ojf_segconv_forward_multi(int n, int batch, const ojf_segconv *const *convs, const float *const *ins, const int *in_strides,
                                      float *const *outs, const int *out_strides, const float *const *ress, const int *res_strides,
                                      const float *const *muls, const int *mul_strides, const int *acts, const int *hs, const int *ws,
                                      ojf_stream_t stream)
{
    using namespace ojf;
    if (n < 1 || n > kSegGroup || !convs || !ins || !in_strides || !outs || !out_strides || !acts || !hs || !ws)
        return fail("ojf_segconv_forward_multi: 1..8 members, non-null arrays");
    SegGroupArgs g;
    for (int i = 0; i < n; ++i)
        if (int rc = seg_fill(convs[i], batch, ins[i], in_strides[i], outs[i], out_strides[i], ress ? ress[i] : nullptr, ress && res_strides ? res_strides[i] : 0,
                              muls ? muls[i] : nullptr, muls && mul_strides ? mul_strides[i] : 0, acts[i], hs[i], ws[i], g.a[i])) return rc;
    return seg_launch_multi(g, n, as_stream(stream));
}

OJF_API int ojf_segconv_forward_batch(const ojf_segconv *c, int batch, const float *in, int in_stride, float *out, int out_stride,
                                      const float *res, int res_stride, const float *mul, int mul_stride, int act, int h, int w,
                                      ojf_stream_t stream)
{
    using namespace ojf;
    SegGroupArgs g;
    if (int rc = seg_fill(c, batch, in, in_stride, out, out_stride, res, res_stride, mul, mul_stride, act, h, w, g.a[0])) return rc;
    return seg_launch(g, 1, as_stream(stream));
}

OJF_API int ojf_segconv_forward(const ojf_segconv *c, const float *in, int in_stride, float *out, int out_stride, const float *res,
                                int res_stride, const float *mul, int mul_stride, int act, int h, int w, ojf_stream_t stream)
{
    return ojf_segconv_forward_batch(c, 1, in, in_stride, out, out_stride, res, res_stride, mul, mul_stride, act, h, w, stream);
}

OJF_API int ojf_segconv_forward_group_batch(int n, int batch, const ojf_segconv *const *convs, const float *const *ins, int in_stride,
                                            float *const *outs, int out_stride, const float *const *ress, int res_stride,
                                            const float *const *muls, int mul_stride, int act, int h, int w, ojf_stream_t stream)
{
    using namespace ojf;
    if (n < 1 || n > kSegGroup || !convs || !ins || !outs) return fail("ojf_segconv_forward_group: 1..8 members, non-null arrays");
    SegGroupArgs g;
    for (int i = 0; i < n; ++i) {
        if (int rc = seg_fill(convs[i], batch, ins[i], in_stride, outs[i], out_stride, ress ? ress[i] : nullptr, res_stride,
                              muls ? muls[i] : nullptr, mul_stride, act, h, w, g.a[i])) return rc;
        const SegArgs &a = g.a[i], &b = g.a[0];
        if (a.n_kb != b.n_kb || a.n_ct != b.n_ct || a.c8 != b.c8 || a.c_out != b.c_out || a.ksize != b.ksize || a.stride != b.stride ||
            a.Ho != b.Ho || a.Wo != b.Wo || a.up != b.up || (i > 0 && a.rng_bump))
            return fail("ojf_segconv_forward_group: the members must share channels, kernel size, stride and output size");
    }
    return seg_launch(g, n, as_stream(stream));
}

OJF_API int ojf_segconv_forward_group(int n, const ojf_segconv *const *convs, const float *const *ins, int in_stride, float *const *outs,
                                      int out_stride, const float *const *ress, int res_stride, const float *const *muls,
                                      int mul_stride, int act, int h, int w, ojf_stream_t stream)
{
    return ojf_segconv_forward_group_batch(n, 1, convs, ins, in_stride, outs, out_stride, ress, res_stride, muls, mul_stride, act, h, w, stream);
}
