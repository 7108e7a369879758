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
                                                           int out_stride, int Ho, int Wo)
{
    const int c4 = (C + 3) / 4;
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)Ho * Wo * c4) return;
    const int cg = (int)(item % c4);
    const int po = (int)(item / c4);
    const int oy = po / Wo, ox = po - oy * Wo;
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

// torch.softmax(logits, dim=1).max(dim=1): score = exp(l_max - l_max) / sum_j exp(l_j - l_max), id = first arg max
__global__ __launch_bounds__(256) void seg_softmax_max_kernel(const float *logits, int stride, int C, int npix, float *scores,
                                                               uint8_t *ids)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float *l = logits + (size_t)p * stride;
    float m = l[0];
    int am = 0;
    bool nan = m != m;
    for (int c = 1; c < C; ++c) {
        const float v = l[c];
        if (v != v && !nan) { nan = true; am = c; }  // torch.max returns the first NaN
        if (!nan && v > m) { m = v; am = c; }
    }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(l[c] - m);
    scores[p] = nan ? __builtin_nanf("") : 1.0f / s;
    ids[p] = (uint8_t)am;
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

OJF_API int ojf_seg_maxpool(const float *in, int in_stride, int c, int h, int w, float *out, int out_stride, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || c < 1 || h < 1 || w < 1 || in_stride < c || out_stride < c) return fail("ojf_seg_maxpool: bad argument");
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;  // floor((h + 2 - 3) / 2) + 1
    const long items = (long)Ho * Wo * ((c + 3) / 4);
    hipLaunchKernelGGL(seg_maxpool_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), in, in_stride, c, h, w,
                       out, out_stride, Ho, Wo);
    return check_hip(hipGetLastError(), "seg_maxpool_kernel launch");
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
    hipLaunchKernelGGL(seg_softmax_max_kernel, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), logits, stride, n_classes, npix,
                       scores, ids);
    return check_hip(hipGetLastError(), "seg_softmax_max_kernel launch");
}
