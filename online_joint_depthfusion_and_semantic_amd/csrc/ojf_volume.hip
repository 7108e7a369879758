// Full-grid passes of the Database (modules/database.py:108-112, 351-370; utils/metrics.py:111-127).
// Pure streaming kernels: 16-byte accesses per lane, grid-stride, bandwidth roofline.
#include "ojf_common.h"

namespace ojf {

static inline int stream_grid(size_t work_items)
{
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 resident blocks, grid-stride the rest
    if (blocks == 0) blocks = 1;
    return (int)blocks;
}

__global__ __launch_bounds__(256) void fill_u16_kernel(uint16_t *v, size_t n, uint16_t bits)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t head = ((16 - ((uintptr_t)v & 15)) & 15) / 2;  // elements before 16-B alignment
    const size_t nh = head < n ? head : n;
    if (tid < nh) v[tid] = bits;
    const size_t nvec = (n - nh) / 8;
    uint4 pat;
    pat.x = pat.y = pat.z = pat.w = (uint32_t)bits | ((uint32_t)bits << 16);
    uint4 *vv = reinterpret_cast<uint4 *>(v + nh);
    for (size_t i = tid; i < nvec; i += step) vv[i] = pat;
    const size_t tail0 = nh + nvec * 8;
    if (tid < n - tail0) v[tail0 + tid] = bits;
}

__global__ __launch_bounds__(256) void filter_kernel(uint16_t *tsdf, uint16_t *wgt, size_t n, float thr,
                                                      uint16_t init_bits)
{
    // Database.filter: volume[weights < value] = init ; weights[weights < value] = 0
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += step) {
        const float w = h2f(wgt[i]);
        if (w < thr) {
            tsdf[i] = init_bits;
            wgt[i] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void evaluate_kernel(const uint16_t *est, const uint16_t *gt,
                                                        const uint16_t *wgt, size_t n, double *sums)
{
    // utils/metrics.py:111-127: nan_to_num, clip to +-0.04, masked mse / mad / occupancy iou / sign acc
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = tid; i < n; i += step) {
        if (!(h2f(wgt[i]) > 0.0f)) continue;
        float e = h2f(est[i]), g = h2f(gt[i]);
        e = (e != e) ? 0.0f : e;  // nan_to_num (infinities are clipped below)
        g = (g != g) ? 0.0f : g;
        e = fminf(fmaxf(e, -0.04f), 0.04f);
        g = fminf(fmaxf(g, -0.04f), 0.04f);
        const double d = (double)e - (double)g;
        acc[0] += 1.0;
        acc[1] += d * d;
        acc[2] += fabs(d);
        const bool eo = e < 0.0f, go = g < 0.0f;
        acc[3] += (eo && go) ? 1.0 : 0.0;
        acc[4] += (eo || go) ? 1.0 : 0.0;
        acc[5] += (eo == go) ? 1.0 : 0.0;  // acc_fn: tp (both < 0) + tn (both >= 0)
    }
    __shared__ double red[6][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[j][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x], v);
    }
}


// Database.evaluate_semantics (modules/database.py:311-349) -> utils/metrics.py:69-108 semantic_evaluation: with
// mask = weights > 0, est' = est * mask, gt' = gt * mask (masked-out voxels count as the pair (0, 0)),
//   hist[gt' * C + est'] += 1  for gt' < C   (np.bincount of the flattened pair index; an est' >= C spills into the
//                                              next row exactly like the reference's flat index does, and is dropped
//                                              beyond C*C where the reference's reshape would raise),
//   present_est[c] / present_gt[c] = 1 if label c occurs in est' / gt' (np.unique, no class bound).
// One pass over 5 B/voxel.  Per-block LDS histogram for C <= 64 (the reference's 30 / 40 classes), the dominant
// (0, 0) pair is counted in a register; u64 global accumulation.
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t *est, const uint8_t *gt, const uint16_t *wgt,
                                                         size_t n, int C, unsigned long long *hist, uint32_t *present)
{
    __shared__ uint32_t lh[LDS_HIST ? 64 * 64 : 1];
    __shared__ uint32_t lp[512];
    const int cc = C * C;
    if (LDS_HIST)
        for (int i = threadIdx.x; i < cc; i += 256) lh[i] = 0;
    for (int i = threadIdx.x; i < 512; i += 256) lp[i] = 0;
    __syncthreads();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned long long zero_pairs = 0;
    auto one = [&](uint32_t e, uint32_t g, uint16_t wbits) {
        const bool m = h2f(wbits) > 0.0f;
        e = m ? e : 0u;
        g = m ? g : 0u;
        if ((e | g) == 0u) {
            ++zero_pairs;
            return;
        }
        lp[e] = 1u;        // benign race: every writer stores 1
        lp[256 + g] = 1u;
        if ((int)g < C) {
            const int idx = (int)g * C + (int)e;
            if (idx < cc) {
                if (LDS_HIST) atomicAdd(&lh[idx], 1u);
                else atomicAdd(&hist[idx], 1ull);
            }
        }
    };
    const bool aligned = (((uintptr_t)est | (uintptr_t)gt) & 7) == 0 && ((uintptr_t)wgt & 15) == 0;
    const size_t nvec = aligned ? n / 8 : 0;
    for (size_t i = tid; i < nvec; i += step) {
        const uint2 e8 = reinterpret_cast<const uint2 *>(est)[i];
        const uint2 g8 = reinterpret_cast<const uint2 *>(gt)[i];
        const uint4 w8 = reinterpret_cast<const uint4 *>(wgt)[i];
        const uint32_t ew[2] = {e8.x, e8.y}, gw[2] = {g8.x, g8.y}, ww[4] = {w8.x, w8.y, w8.z, w8.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            one((ew[k >> 2] >> (8 * (k & 3))) & 0xffu, (gw[k >> 2] >> (8 * (k & 3))) & 0xffu,
                (uint16_t)(ww[k >> 1] >> (16 * (k & 1))));
    }
    for (size_t i = nvec * 8 + tid; i < n; i += step) one(est[i], gt[i], wgt[i]);
    // (0, 0) pairs: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) zero_pairs += __shfl_down(zero_pairs, off, 64);
    if ((threadIdx.x & 63) == 0 && zero_pairs) {
        atomicAdd(&hist[0], zero_pairs);
        lp[0] = 1u;
        lp[256] = 1u;
    }
    __syncthreads();
    if (LDS_HIST)
        for (int i = threadIdx.x; i < cc; i += 256)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
    for (int i = threadIdx.x; i < 512; i += 256)
        if (lp[i]) present[i] = 1u;
}


// Database.filter_semantics (modules/database.py:114-116): scipy.ndimage.median_filter(ids, size=5) on the
// u8 label volume - 5x5x5 window, 'reflect' boundary (d c b a | a b c d | d c b a), median = element of
// rank 62 of the 125 sorted values.  One block filters an 8x8x8 brick: the 12^3 neighbourhood is staged in
// LDS once, each lane then radix-selects the rank-62 byte of its window (8 bit-steps of counting).
__device__ __forceinline__ int reflect_index(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

__global__ __launch_bounds__(512) void median5_u8_kernel(const uint8_t *in, uint8_t *out, int X, int Y, int Z)
{
    __shared__ uint8_t tile[12][12][12];
    const int bx = blockIdx.x * 8, by = blockIdx.y * 8, bz = blockIdx.z * 8;
    for (int i = threadIdx.x; i < 12 * 12 * 12; i += 512) {
        const int lx = i / 144, ly = (i / 12) % 12, lz = i % 12;
        const int gx = reflect_index(bx + lx - 2, X), gy = reflect_index(by + ly - 2, Y), gz = reflect_index(bz + lz - 2, Z);
        tile[lx][ly][lz] = in[((size_t)gx * Y + gy) * Z + gz];
    }
    __syncthreads();
    const int lx = threadIdx.x >> 6, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x & 7;
    const int x = bx + lx, y = by + ly, z = bz + lz;
    if (x >= X || y >= Y || z >= Z) return;
    // radix select: fix the bits of the rank-62 element from the most significant down
    unsigned int prefix = 0, rank = 62;
    for (int bit = 7; bit >= 0; --bit) {
        const unsigned int mask = (0xffu << (bit + 1)) & 0xffu;  // bits already decided
        unsigned int zeros = 0;  // candidates (matching the prefix) whose current bit is 0
        for (int dx = 0; dx < 5; ++dx)
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dz = 0; dz < 5; ++dz) {
                    const unsigned int v = tile[lx + dx][ly + dy][lz + dz];
                    zeros += ((v & mask) == prefix && !((v >> bit) & 1u)) ? 1u : 0u;
                }
        if (rank >= zeros) {
            rank -= zeros;
            prefix |= 1u << bit;
        }
    }
    out[((size_t)x * Y + y) * Z + z] = (uint8_t)prefix;
}

}  // namespace ojf

OJF_API int ojf_volume_fill_f16(uint16_t *vol, size_t n, float value, ojf_stream_t stream)
{
    using namespace ojf;
    if (!vol && n) return fail("ojf_volume_fill_f16: null volume");
    if (n == 0) return 0;
    const _Float16 hv = (_Float16)value;
    uint16_t bits;
    __builtin_memcpy(&bits, &hv, 2);
    hipLaunchKernelGGL(fill_u16_kernel, dim3(stream_grid(n / 8 + 16)), dim3(256), 0, as_stream(stream), vol, n, bits);
    return check_hip(hipGetLastError(), "ojf_volume_fill_f16 launch");
}

OJF_API int ojf_volume_fill_u8(uint8_t *vol, size_t n, uint8_t value, ojf_stream_t stream)
{
    using namespace ojf;
    if (!vol && n) return fail("ojf_volume_fill_u8: null volume");
    if (n == 0) return 0;
    return check_hip(hipMemsetAsync(vol, value, n, as_stream(stream)), "ojf_volume_fill_u8");
}

OJF_API int ojf_volume_filter(uint16_t *tsdf, uint16_t *wgt, size_t n, float threshold, float init_value,
                              ojf_stream_t stream)
{
    using namespace ojf;
    if ((!tsdf || !wgt) && n) return fail("ojf_volume_filter: null volume");
    if (n == 0) return 0;
    const _Float16 hv = (_Float16)init_value;
    uint16_t bits;
    __builtin_memcpy(&bits, &hv, 2);
    hipLaunchKernelGGL(filter_kernel, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), tsdf, wgt, n,
                       threshold, bits);
    return check_hip(hipGetLastError(), "ojf_volume_filter launch");
}

OJF_API int ojf_volume_evaluate(const uint16_t *est, const uint16_t *gt, const uint16_t *wgt, size_t n,
                                double *sums, ojf_stream_t stream)
{
    using namespace ojf;
    if (!est || !gt || !wgt || !sums) return fail("ojf_volume_evaluate: null pointer argument");
    OJF_HIP(hipMemsetAsync(sums, 0, 8 * sizeof(double), as_stream(stream)));
    if (n == 0) return 0;
    hipLaunchKernelGGL(evaluate_kernel, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), est, gt, wgt, n,
                       sums);
    return check_hip(hipGetLastError(), "ojf_volume_evaluate launch");
}

OJF_API int ojf_volume_confusion(const uint8_t *est, const uint8_t *gt, const uint16_t *wgt, size_t n, int n_classes,
                                 unsigned long long *hist, uint32_t *present, ojf_stream_t stream)
{
    using namespace ojf;
    if (!est || !gt || !wgt || !hist || !present) return fail("ojf_volume_confusion: null pointer argument");
    if (n_classes < 1 || n_classes > 256) return fail("ojf_volume_confusion: n_classes must be in [1, 256]");
    OJF_HIP(hipMemsetAsync(hist, 0, (size_t)n_classes * n_classes * sizeof(unsigned long long), as_stream(stream)));
    OJF_HIP(hipMemsetAsync(present, 0, 512 * sizeof(uint32_t), as_stream(stream)));
    if (n == 0) return 0;
    const dim3 grid(stream_grid(n / 8 + 1));
    if (n_classes <= 64)
        hipLaunchKernelGGL(confusion_kernel<true>, grid, dim3(256), 0, as_stream(stream), est, gt, wgt, n, n_classes, hist, present);
    else
        hipLaunchKernelGGL(confusion_kernel<false>, grid, dim3(256), 0, as_stream(stream), est, gt, wgt, n, n_classes, hist, present);
    return check_hip(hipGetLastError(), "ojf_volume_confusion launch");
}

OJF_API int ojf_volume_median5_u8(const uint8_t *in, uint8_t *out, int X, int Y, int Z, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out) return fail("ojf_volume_median5_u8: null volume");
    if (in == out) return fail("ojf_volume_median5_u8: in-place filtering is not supported");
    if (X <= 0 || Y <= 0 || Z <= 0) return fail("ojf_volume_median5_u8: non-positive size");
    const dim3 grid((X + 7) / 8, (Y + 7) / 8, (Z + 7) / 8);
    hipLaunchKernelGGL(median5_u8_kernel, grid, dim3(512), 0, as_stream(stream), in, out, X, Y, Z);
    return check_hip(hipGetLastError(), "ojf_volume_median5_u8 launch");
}
