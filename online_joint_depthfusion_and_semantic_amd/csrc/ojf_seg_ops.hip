// Non-convolution operators of the AdapNet++ front end (modules/adapnet.py, modules/pipeline.py:42-60,181-185) on
// NHWC fp32 rows of batch 1 (pointer + floats per pixel row, like ojf_segconv_forward), so that the inference path of
// `semantic_strategy: predict` is libojf launches end to end:
//   ojf_seg_pack_input   image / 255 (or depth replicated to three channels) -> the stem's 8-channel NHWC rows
//   ojf_seg_maxpool      nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem
//   ojf_seg_mean         global average over the pixels (eASPP branch 5, Decoder._skip), fixed summation order
//   ojf_seg_broadcast    a per-channel vector to every pixel (bilinear upsampling of a 1x1 map), optionally times a tensor
//   ojf_seg_softmax_max  softmax over the classes + max: (score, id) per pixel
// All are small HBM-streaming kernels: 16 bytes per lane where the row stride allows it.
#include "ojf_common.h"

namespace ojf {

__global__ __launch_bounds__(256) void seg_pack_input_kernel(const float *src, int chan_stride, float scale, int npix, float *out,
                                                              int out_stride)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float *o = out + (size_t)p * out_stride;
    // inputs['image'] = data['image'] / 255.0 (pipeline.py:44): a true division, like torch
    const float a = src[p], b = src[(size_t)chan_stride + p], c = src[2 * (size_t)chan_stride + p];
    o[0] = scale == 1.0f ? a : a / scale;
    o[1] = scale == 1.0f ? b : b / scale;
    o[2] = scale == 1.0f ? c : c / scale;
#pragma unroll
    for (int j = 3; j < 8; ++j) o[j] = 0.0f;
}

// one lane per (output pixel, 4-channel group); -inf padding like torch (NaN inputs propagate: v > m is false for NaN
// m only, so a NaN in the window wins through the explicit test)
__global__ __launch_bounds__(256) void seg_maxpool_kernel(const float *in, int in_stride, int C, int H, int W, float *out,
                                                           int out_stride, int Ho, int Wo, int B)
{
    const int c4 = (C + 3) / 4;
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)B * Ho * Wo * c4) return;
    const int cg = (int)(item % c4);
    const int po = (int)(item / c4);  // output pixel over the B images
    const int b = po / (Ho * Wo), q = po - b * (Ho * Wo);
    const int oy = q / Wo, ox = q - oy * Wo;
    in += (size_t)b * H * W * in_stride;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * oy - 1 + dy;
        if ((unsigned)y >= (unsigned)H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int x = 2 * ox - 1 + dx;
            if ((unsigned)x >= (unsigned)W) continue;
            const float *r = in + ((size_t)y * W + x) * in_stride + 4 * cg;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * cg + j < C) {
                    const float v = r[j];
                    m[j] = (v > m[j] || v != v) ? v : m[j];
                }
        }
    }
    float *o = out + (size_t)po * out_stride + 4 * cg;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * cg + j < C) o[j] = m[j];
}

// Two fixed-order stages: block (channel group of 64, pixel slice y) sums its slice (four interleaved phases, added in
// order) into partial[y][c]; the finishing launch adds the kMeanSlices partials in order and divides.  (One block per 64
// channels over ALL pixels was a 200 us serial walk on the 60x80 x 256-channel decoder state.)
constexpr int kMeanSlices = 32;

__global__ __launch_bounds__(256) void seg_mean_partial_kernel(const float *in, int in_stride, int C, int npix, float *partial)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const int per = (npix + kMeanSlices - 1) / kMeanSlices;
    const int p0 = blockIdx.y * per, p1 = min(npix, p0 + per);
    float s = 0.0f;
    if (c < C)
        for (int p = p0 + ph; p < p1; p += 4) s += in[(size_t)p * in_stride + c];
    part[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < C)
        partial[(size_t)blockIdx.y * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void seg_mean_finish_kernel(const float *partial, int C, int npix, float *out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
#pragma unroll 8
    for (int y = 0; y < kMeanSlices; ++y) s += partial[(size_t)y * C + c];
    out[c] = s / (float)npix;
}

__global__ __launch_bounds__(256) void seg_broadcast_kernel(const float *vec, const float *mul, int mul_stride, int C, int npix,
                                                             float *out, int out_stride)
{
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)npix * C) return;
    const int c = (int)(item % C);
    const int p = (int)(item / C);
    const float v = vec[c];
    out[(size_t)p * out_stride + c] = mul ? v * mul[(size_t)p * mul_stride + c] : v;
}

// torch.softmax(logits, dim=1).max(dim=1): score = exp(l_max - l_max) / sum_j exp(l_j - l_max), id = first arg max.
// VEC: rows are 16-byte aligned - a lane fetches its pixel's classes as float4 (the scalar form issued C 4-byte loads per
// lane, each spread over 64 cache lines: 20 us for a 320x240 frame with 30 classes; same operations in the same order).
template <bool VEC>
__global__ __launch_bounds__(256) void seg_softmax_max_kernel(const float *logits, int stride, int C, int npix, float *scores,
                                                               uint8_t *ids)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float *l = logits + (size_t)p * stride;
    constexpr int kMaxVec = 64;  // classes held in registers by the vector form (the launch falls back beyond that)
    float v[VEC ? kMaxVec : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int c4 = 0; c4 < kMaxVec / 4; ++c4)
            if (c4 * 4 < C) {
                const float4 q = *reinterpret_cast<const float4 *>(l + c4 * 4);  // (the row holds round_up(C, 4) readable floats)
                v[c4 * 4] = q.x; v[c4 * 4 + 1] = q.y; v[c4 * 4 + 2] = q.z; v[c4 * 4 + 3] = q.w;
            }
    }
    auto at = [&](int c) { if constexpr (VEC) return v[c]; else return l[c]; };
    float m = at(0);
    int am = 0;
    bool nan = m != m;
    if constexpr (VEC) {
#pragma unroll
        for (int c = 1; c < kMaxVec; ++c)
            if (c < C) {
                const float x = v[c];
                if (x != x && !nan) { nan = true; am = c; }  // torch.max returns the first NaN
                if (!nan && x > m) { m = x; am = c; }
            }
    } else {
        for (int c = 1; c < C; ++c) {
            const float x = l[c];
            if (x != x && !nan) { nan = true; am = c; }
            if (!nan && x > m) { m = x; am = c; }
        }
    }
    float s = 0.0f;
    if constexpr (VEC) {
#pragma unroll
        for (int c = 0; c < kMaxVec; ++c)
            if (c < C) s += expf(v[c] - m);
    } else {
        for (int c = 0; c < C; ++c) s += expf(l[c] - m);
    }
    scores[p] = nan ? __builtin_nanf("") : 1.0f / s;
    ids[p] = (uint8_t)am;
}

// ---- squeeze chains: global average -> 1x1 convolution on the 1x1 map -> broadcast (x gate) ------------------------------
// eASPP branch 5 (adapnet.py:204-210: pool, conv, upsample of a 1x1 map = broadcast) and Decoder._skip (:292-296: the pooled
// decoder state through a 1x1 convolution gates the encoder skip) were four launches each - partial means, finish, a
// segconv launch on ONE pixel, the broadcast - of ~5 us apiece whatever their size; two launches now, for all members.
constexpr int kPoolSlicesMax = 128;

struct PoolIns { const float *p[8]; };

__global__ __launch_bounds__(256) void seg_pool_partial_kernel(const PoolIns ins, int in_stride, int C, int npix, int slices,
                                                                float *partial)
{
    __shared__ float part[4][64];
    const float *in = ins.p[blockIdx.z];
    float *dst = partial + (size_t)blockIdx.z * kPoolSlicesMax * C;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const int per = (npix + slices - 1) / slices;
    const int p0 = blockIdx.y * per, p1 = min(npix, p0 + per);
    float s = 0.0f;
    if (c < C) {
        int p = p0 + ph;
        for (; p + 28 < p1; p += 32) {  // eight loads in flight, added in pixel order (a plain loop waits for every load in turn)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = in[(size_t)(p + 4 * j) * in_stride + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; p < p1; p += 4) s += in[(size_t)p * in_stride + c];
    }
    part[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < C)
        dst[(size_t)blockIdx.y * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

struct PoolFcArgs {  // (pointer tables by value: the call must be capturable into a device graph)
    const float *weights[8], *biases[8], *muls[8];
    float *outs[8];
    const float *partial;
    int c_in, c_out, npix_in, npix_out, slices, act, mul_stride, out_stride, px_chunks;
};

// block (pixel chunk x, 16 output channels y, member z): every block finishes the mean vector itself (slices x c_in partials:
// at most 1 MB, from L2), computes ITS 16 outputs (wave w: channels 4w .. 4w+3, lanes stride the c_in products, fixed
// butterfly order) and writes them to its chunk of the output pixels.  Redundant across x: the launch heuristic keeps the
// repeated work below the broadcast's own traffic.
__global__ __launch_bounds__(256) void seg_pool_fc_kernel(PoolFcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float mean[];  // [c_in], then [16] results
    float *res = mean + a.c_in;
    const int z = blockIdx.z;
    const float *part = a.partial + (size_t)z * kPoolSlicesMax * a.c_in;
    for (int k = threadIdx.x; k < a.c_in; k += 256) {
        float s = 0.0f;
        for (int y = 0; y < a.slices; y += 8) {  // (slices is a multiple of 8) eight loads in flight, added in slice order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(y + j) * a.c_in + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        mean[k] = s / (float)a.npix_in;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *W = a.weights[z];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = blockIdx.y * 16 + wave * 4 + j;
        float s = 0.0f;
        if (c < a.c_out) {
            int k = lane;
            for (; k + 448 < a.c_in; k += 512) {  // eight weight loads in flight
                float wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = W[(size_t)c * a.c_in + k + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) s = __builtin_fmaf(wv[j], mean[k + 64 * j], s);
            }
            for (; k < a.c_in; k += 64) s = __builtin_fmaf(W[(size_t)c * a.c_in + k], mean[k], s);
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            float v = 0.0f;
            if (c < a.c_out) {
                v = s + (a.biases[z] ? a.biases[z][c] : 0.0f);
                if (a.act == 1) v = v < 0.0f ? 0.0f : v;
            }
            res[wave * 4 + j] = v;
        }
    }
    __syncthreads();
    const int c0 = blockIdx.y * 16, nc = min(16, a.c_out - c0);
    const int per = (a.npix_out + a.px_chunks - 1) / a.px_chunks;
    const int p0 = blockIdx.x * per, p1 = min(a.npix_out, p0 + per);
    float *out = a.outs[z];
    const float *mul = a.muls[z];
    for (int i = threadIdx.x; i < (p1 - p0) * 16; i += 256) {
        const int p = p0 + (i >> 4), j = i & 15;
        if (j >= nc) continue;
        const float v = res[j];
        out[(size_t)p * a.out_stride + c0 + j] = mul ? v * mul[(size_t)p * a.mul_stride + c0 + j] : v;
    }
}

}  // namespace ojf

OJF_API int ojf_seg_pack_input(const float *src, int chan_stride, float divisor, int h, int w, float *out, int out_stride,
                               ojf_stream_t stream)
{
    using namespace ojf;
    if (!src || !out || h < 1 || w < 1 || out_stride < 8 || chan_stride < 0 || !(divisor > 0.0f))
        return fail("ojf_seg_pack_input: bad argument");
    const int npix = h * w;
    hipLaunchKernelGGL(seg_pack_input_kernel, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), src, chan_stride, divisor,
                       npix, out, out_stride);
    return check_hip(hipGetLastError(), "seg_pack_input_kernel launch");
}

OJF_API int ojf_seg_maxpool_batch(int batch, const float *in, int in_stride, int c, int h, int w, float *out, int out_stride, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || c < 1 || h < 1 || w < 1 || in_stride < c || out_stride < c || batch < 1 || batch > 64) return fail("ojf_seg_maxpool: bad argument");
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;  // floor((h + 2 - 3) / 2) + 1
    const long items = (long)batch * Ho * Wo * ((c + 3) / 4);
    hipLaunchKernelGGL(seg_maxpool_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), in, in_stride, c, h, w,
                       out, out_stride, Ho, Wo, batch);
    return check_hip(hipGetLastError(), "seg_maxpool_kernel launch");
}

OJF_API int ojf_seg_maxpool(const float *in, int in_stride, int c, int h, int w, float *out, int out_stride, ojf_stream_t stream)
{
    return ojf_seg_maxpool_batch(1, in, in_stride, c, h, w, out, out_stride, stream);
}

OJF_API int ojf_seg_mean(const float *in, int in_stride, int c, int npix, float *partial, float *out, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || !partial || c < 1 || npix < 1 || in_stride < c) return fail("ojf_seg_mean: bad argument");
    hipLaunchKernelGGL(seg_mean_partial_kernel, dim3((c + 63) / 64, kMeanSlices), dim3(256), 0, as_stream(stream), in, in_stride, c, npix,
                       partial);
    hipLaunchKernelGGL(seg_mean_finish_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), partial, c, npix, out);
    return check_hip(hipGetLastError(), "seg_mean kernels launch");
}

OJF_API int ojf_seg_broadcast(const float *vec, const float *mul, int mul_stride, int c, int npix, float *out, int out_stride,
                              ojf_stream_t stream)
{
    using namespace ojf;
    if (!vec || !out || c < 1 || npix < 1 || out_stride < c || (mul && mul_stride < c)) return fail("ojf_seg_broadcast: bad argument");
    const long items = (long)npix * c;
    hipLaunchKernelGGL(seg_broadcast_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), vec, mul, mul_stride, c,
                       npix, out, out_stride);
    return check_hip(hipGetLastError(), "seg_broadcast_kernel launch");
}

OJF_API int ojf_seg_softmax_max(const float *logits, int stride, int n_classes, int npix, float *scores, uint8_t *ids,
                                ojf_stream_t stream)
{
    using namespace ojf;
    if (!logits || !scores || !ids || n_classes < 1 || n_classes > 256 || npix < 1 || stride < n_classes)
        return fail("ojf_seg_softmax_max: bad argument");
    // (vector form: 16-byte aligned rows that hold round_up(n_classes, 4) readable floats)
    const bool vec = n_classes <= 64 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && stride >= (n_classes + 3) / 4 * 4;
    if (vec) hipLaunchKernelGGL(seg_softmax_max_kernel<true>, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), logits, stride, n_classes, npix, scores, ids);
    else hipLaunchKernelGGL(seg_softmax_max_kernel<false>, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), logits, stride, n_classes, npix, scores, ids);
    return check_hip(hipGetLastError(), "seg_softmax_max_kernel launch");
}

OJF_API int ojf_seg_pool_fc(int n, const float *const *ins, int in_stride, int c_in, int npix_in, const float *const *weights,
                            const float *const *biases, int c_out, int act, const float *const *muls, int mul_stride, float *const *outs,
                            int out_stride, int npix_out, float *partial, ojf_stream_t stream)
{
    using namespace ojf;
    if (n < 1 || n > 8 || !ins || !weights || !outs || !partial) return fail("ojf_seg_pool_fc: 1..8 members, non-null arrays");
    if (c_in < 1 || c_in > 8192 || c_out < 1 || npix_in < 1 || npix_out < 1 || in_stride < c_in || out_stride < c_out || (muls && mul_stride < c_out) ||
        act < 0 || act > 1)
        return fail("ojf_seg_pool_fc: bad argument");
    hipStream_t st = as_stream(stream);
    PoolIns pi{};
    PoolFcArgs a{};
    for (int i = 0; i < n; ++i) {
        if (!ins[i] || !weights[i] || !outs[i]) return fail("ojf_seg_pool_fc: null member pointer");
        pi.p[i] = ins[i]; a.weights[i] = weights[i]; a.biases[i] = biases ? biases[i] : nullptr; a.muls[i] = muls ? muls[i] : nullptr;
        a.outs[i] = outs[i];
    }
    const int slices = npix_in > 1200 ? 16 : 8;  // every block of the second launch re-adds slices * c_in partials
    hipLaunchKernelGGL(seg_pool_partial_kernel, dim3((c_in + 63) / 64, slices, n), dim3(256), 0, st, pi, in_stride, c_in, npix_in, slices, partial);
    OJF_HIP(hipGetLastError());
    a.partial = partial; a.c_in = c_in; a.c_out = c_out; a.npix_in = npix_in; a.npix_out = npix_out; a.slices = slices; a.act = act;
    a.mul_stride = mul_stride; a.out_stride = out_stride;
    // pixel chunks: at most 8 output elements per thread (their loads / stores are one latency each), but every extra chunk
    // repeats (slices + 16) * c_in * 4 bytes of reads: wide inputs get few chunks
    long chunks = ((long)npix_out * 16 + 2047) / 2048;
    const long cap = c_in >= 1024 ? 4 : 64;
    a.px_chunks = (int)(chunks < 1 ? 1 : (chunks > cap ? cap : chunks));
    hipLaunchKernelGGL(seg_pool_fc_kernel, dim3(a.px_chunks, (c_out + 15) / 16, n), dim3(256), (c_in + 16) * sizeof(float), st, a);
    return check_hip(hipGetLastError(), "seg_pool_fc kernels launch");
}
