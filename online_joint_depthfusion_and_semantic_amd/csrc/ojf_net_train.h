// TRAINING path of the fusion net (modules/pipeline.py:301-363 fuse_training with the net in train() mode; included by
// ojf_net.hip only).  One fused layer unit of the reference's Sequentials (modules/model.py:4-52,115-141)
//
//     conv (1x1 / dilated 3x3, bias) -> BatchNorm2d with BATCH statistics -> ReLU / LeakyReLU / Tanh -> Dropout2d
//
// as device kernels for forward and backward; Python (train.py) wraps a unit as ONE torch.autograd.Function and leaves the
// glue of the net (concatenations, average pools, the 1x1 global-average map, loss) to torch on the same C4-planar
// tensors, so that autograd accumulates the fan-out gradients and the reference's optimizer / checkpoint code runs
// unchanged.  All convolutions here use the fp32-input MFMA kernel (conv_mfma_kernel: bitwise a k-ordered fmaf chain):
// gradients are tiny (d loss / d est ~ 1e-6) and of unbounded dynamic range - the split-fp16 arithmetic of the
// inference path would need loss scaling; the weight gradient runs on the fp32 MFMA as well (K = pixels).
//
//   ojf_train_pack          torch weights [oc][ic][k][k] -> the conv kernel's fragment layout, on the device, every step:
//                           forward form, or transposed + tap-flipped form (backward-data IS a convolution with it)
//   ojf_train_conv          convolution on C4 planes (any c_out: chunks of 8 output tiles per launch)
//   ojf_train_bn_act        per-channel batch mean / 1 / sqrt(var + eps) (fp64 sums, fixed order) + running-stat update, then
//                           out = drop_c * act(gamma_c * (y - mean_c) * invstd_c + beta_c)
//   ojf_train_bn_act_bwd    dgamma, dbeta, dbias and dy = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat))
//   ojf_train_wgrad         dW[oc][ic][tap] = sum_p dy[oc][p] * x[ic][p + tap]  (pixel slabs -> partial sums -> fixed-order sum)
#pragma once

namespace ojf {

constexpr int kTrainSlabs = 64;  // pixel slabs of the per-channel reductions (partial sums added in slab order)

// ---- weight packing on the device ----------------------------------------------------------------------------------
// Packed element (row r, K group G = tap * c4 + cg, component j) <- weight(oc, ic, tap'):
//   forward:     oc = r,            ic = unslot(4 cg + j),  tap' = tap
//   transposed:  ic = unslot(r),    oc = 4 cg + j,          tap' = taps - 1 - tap   (backward-data)
// unslot: physical channel of a concatenation of `group`-wide tensors stored in `slot`-wide slots -> logical channel.
struct PackArgs {
    const float *w;     // [OC][IC][taps]
    const float *bias;  // [OC] or NULL
    float *wp;          // [n_ot][nsteps + kPadSteps][64][4]
    float *bp;          // [n_ot * 16] (forward form only; NULL otherwise)
    int OC, IC, taps, group, slot, c4, nsteps, n_ot, transposed;
    // stacked layers (the four branch-entry 1x1 of a VortexPooling as ONE convolution): this weight tensor owns output
    // channels [oc_base, oc_base + OC) of the stack; with `partial` only its elements are written (the buffer starts zeroed)
    int oc_base, partial;
    // a convolution that reads only input channels [ic_base, ic_base + IC) of a weight tensor with ic_total input channels
    int ic_base, ic_total;
};

__device__ __forceinline__ int train_unslot(int x, int group, int slot, int n_logical)
{
    const int s = x / slot, in = x - s * slot;
    const int l = s * group + in;
    return (in < group && l < n_logical) ? l : -1;
}

__global__ __launch_bounds__(256) void train_pack_kernel(const PackArgs a)
{
    const int nsp = a.nsteps + kPadSteps;
    const long total = (long)a.n_ot * nsp * 256;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)a.n_ot * 16 && a.bp) {
        const long ob = i - a.oc_base;
        const bool mine = ob >= 0 && ob < a.OC;
        if (mine || !a.partial) a.bp[i] = (a.bias && mine) ? a.bias[ob] : 0.0f;
    }
    if (i >= total) return;
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long rest = i >> 8;
    const int S = (int)(rest % nsp), ot = (int)(rest / nsp);
    const int r = ot * 16 + (lane & 15), G = 4 * S + (lane >> 4);
    float v = 0.0f;
    bool mine = false;
    if (S < a.nsteps && G < a.taps * a.c4) {
        const int tap = G / a.c4, ch = 4 * (G - tap * a.c4) + j;
        int oc, ic, t;
        if (!a.transposed) { oc = r - a.oc_base; ic = train_unslot(ch, a.group, a.slot, a.IC); t = tap; }
        else { ic = train_unslot(r, a.group, a.slot, a.IC); oc = ch - a.oc_base; t = a.taps - 1 - tap; }
        mine = oc >= 0 && oc < a.OC;
        if (mine && ic >= 0) v = a.w[((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + t];
    }
    if (mine || !a.partial) a.wp[i] = v;
}

// ---- BatchNorm statistics ------------------------------------------------------------------------------------------
// block (slab, channel group): fp64 sums of y and y^2 over the slab's pixels of the group's four channels
__device__ __forceinline__ void train_stats_partial_body(const f32x4 *y, int g0, int npix, double *partial /* [slabs][c4][8] */)
{
    __shared__ double red[4][8];
    const int cg = blockIdx.y, c4 = gridDim.y;
    const int per = (npix + kTrainSlabs - 1) / kTrainSlabs;
    const int p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
    const f32x4 *plane = y + (size_t)(g0 + cg) * npix;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four loads in flight per lane (the kernels of this file are latency-, not bandwidth-bound); the sums run in the
    // same order as a plain loop
    int p = p0 + threadIdx.x;
    for (; p + 768 < p1; p += 1024) {
        const f32x4 v4[4] = {plane[p], plane[p + 256], plane[p + 512], plane[p + 768]};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += (double)v4[u][j]; s[4 + j] += (double)v4[u][j] * (double)v4[u][j]; }
    }
    for (; p < p1; p += 256) {
        const f32x4 v = plane[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += (double)v[j]; s[4 + j] += (double)v[j] * (double)v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = s[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8)
        partial[((size_t)blockIdx.x * c4 + cg) * 8 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void train_stats_partial_kernel(const f32x4 *y, int g0, int npix, double *partial)
{
    train_stats_partial_body(y, g0, npix, partial);
}

// Totals of the kTrainSlabs partial rows of channel group cg (8 doubles each: what the partial kernels wrote), formed
// by every block that needs them instead of a one-block "finish" launch per layer (two of the six small kernels a layer
// used to cost): lane b of the calling wave takes row b, an xor butterfly adds them - a fixed tree, every lane and
// every block get bit-identical totals.
static_assert(kTrainSlabs == 64, "one partial row per lane");
__device__ __forceinline__ void train_group_totals(const double *partial, int c4, int cg, double (&tot)[8])
{
    const double *row = partial + ((size_t)(threadIdx.x & 63) * c4 + cg) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = row[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        tot[j] = v;
    }
}

// ---- normalise + activation + channel dropout ----------------------------------------------------------------------
struct BnActArgs {
    const f32x4 *y;      // conv output planes (window at y_g0)
    f32x4 *out;          // result planes (window at out_g0)
    const f32x4 *dout;   // backward: gradient of `out`
    f32x4 *dy;           // backward: gradient of y
    const float *mean, *invstd;       // [c4 * 4] (identity when has_bn == 0)
    const float *gamma, *beta;        // [C] or NULL
    const float *drop;                // [C] per-channel Dropout2d scale (0 or 1 / (1 - p)) or NULL
    double *partial;                  // backward reduce: [slabs][c4][8]
    int y_g0, out_g0, dout_g0, dy_g0, c4, C, npix, act, has_bn, training;
    float scale;                      // output scale of the last layer (tanh(.) * output_scale)
    // forward with BatchNorm: statistics are finished in the kernel's prologue (train_group_totals of `partial`, or the
    // running statistics in eval mode); block column 0 publishes mean / invstd and updates the running statistics
    float *mean_out, *invstd_out, *running_mean, *running_var;
    float momentum, eps;
    // backward: block column 0 publishes the parameter gradients (added to what is there when `accumulate`)
    float *dgamma, *dbeta, *dbias;
    float *sum_dy;                    // backward, optional: [C] per-channel sums of dy (0 under batch statistics), plain store
    // backward, optional (the whole-net executor): dy is STORED multiplied by a power of two, so that the split-fp16
    // backward-data convolution and weight gradient see operands inside the fp16 range with all their bits; the consumers
    // divide again (exact).  The factor is derived IN THIS PASS from a guaranteed bound on |dy|, not from the previous pass:
    //   dy = gamma invstd (dz - mean(dz) - xhat mean(dz xhat))  =>  |dy| <= |gamma invstd| max|dz| (2 + max|xhat|^2)
    // (without batch statistics: |gamma invstd| max|dz|).  train_bn_bwd_reduce_kernel, which reads every dz and xhat anyway,
    // leaves  A_c = |gamma_c invstd_c| max|dz_c|  and  X_c = max|xhat_c|  in bnd[c] / bnd[bnd_stride + c] (atomicMax on the
    // bits of non-negative floats: kTrainSlabs arrivals per word); train_bn_bwd_apply_kernel turns them into the power of
    // two that puts the bound into [2^13, 2^14) and publishes it in *dy_scale_out.  No lag, no headroom to outgrow, nothing
    // to guard: a gradient of any magnitude is in range by construction (round 3 used the previous pass's maximum with 12
    // binades of headroom - a larger jump turned dy into +-inf halves and NaN sums that no guard could see).
    unsigned *bnd;
    int bnd_stride;
    float *dy_scale_out;
    int accumulate;
};

__device__ __forceinline__ float train_act(float z, int act)
{
    if (act == OJF_ACT_RELU) return z > 0.0f ? z : (z != z ? z : 0.0f);
    if (act == OJF_ACT_LEAKY) return z > 0.0f ? z : 0.01f * z;
    if (act == OJF_ACT_TANH) return tanhf(z);
    return z;
}
__device__ __forceinline__ float train_act_grad(float z, int act)
{
    if (act == OJF_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (act == OJF_ACT_LEAKY) return z > 0.0f ? 1.0f : 0.01f;
    if (act == OJF_ACT_TANH) { const float t = tanhf(z); return 1.0f - t * t; }
    return 1.0f;
}

// per-lane channel constants of group cg: (scale, shift) with z = y * scale + shift, drop, gamma * invstd
__device__ __forceinline__ void train_channel_consts(const BnActArgs &a, int cg, f32x4 &mu, f32x4 &is, f32x4 &ga, f32x4 &be, f32x4 &dr)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * cg + j;
        const bool real = c < a.C;
        mu[j] = (a.has_bn && a.mean) ? a.mean[c] : 0.0f;  // (forward: overwritten from the prologue's statistics)
        is[j] = real ? ((a.has_bn && a.invstd) ? a.invstd[c] : 1.0f) : 0.0f;
        ga[j] = real ? (a.gamma ? a.gamma[c] : 1.0f) : 0.0f;
        be[j] = real ? (a.beta ? a.beta[c] : 0.0f) : 0.0f;
        dr[j] = real ? (a.drop ? a.drop[c] : 1.0f) : 0.0f;
    }
}

// Up to four units of identical shape run as one launch (blockIdx.z): the four branches of a VortexPooling
// share_scale: the units' dy tensors are consumed as ONE stacked tensor (the branch entries) - one common factor
struct BnGroup { BnActArgs g[4]; int share_scale; };

__global__ __launch_bounds__(256) void train_stats_group_kernel(const BnGroup grp)
{
    const BnActArgs &a = grp.g[blockIdx.z];
    if (!(a.has_bn && a.training)) return;
    train_stats_partial_body(a.y, a.y_g0, a.npix, a.partial);
}

__global__ __launch_bounds__(256) void train_bn_act_fwd_kernel(const BnGroup grp)
{
    const BnActArgs &a = grp.g[blockIdx.z];
    const int cg = blockIdx.y;
    __shared__ float stat[8];  // mean[4], invstd[4] of this group
    // the first three loads of the streaming loop are requested BEFORE the statistics are finished (a chain of dependent
    // loads, shuffles and a barrier in front of them cost every one of ~30 launches per pass 2-3 us)
    const f32x4 *yp = a.y + (size_t)(a.y_g0 + cg) * a.npix;
    const int stride = gridDim.x * blockDim.x;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool first3 = p + 2 * stride < a.npix;
    f32x4 py0{0.f, 0.f, 0.f, 0.f}, py1 = py0, py2 = py0;
    if (first3) { py0 = yp[p]; py1 = yp[p + stride]; py2 = yp[p + 2 * stride]; }
    if (a.has_bn) {
        if (threadIdx.x < 64) {
            double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a.training) train_group_totals(a.partial, a.c4, cg, tot);
            if (threadIdx.x < 4) {
                const int j = threadIdx.x, c = 4 * cg + j;
                float m = 0.0f, is = 0.0f;  // padding channels stay exactly zero
                if (c < a.C && !a.training) {
                    m = a.running_mean[c];
                    is = 1.0f / sqrtf(a.running_var[c] + a.eps);
                } else if (c < a.C) {
                    const double n = (double)a.npix, md = tot[j] / n;
                    double var = tot[4 + j] / n - md * md;
                    var = var < 0.0 ? 0.0 : var;
                    m = (float)md;
                    is = (float)(1.0 / sqrt(var + (double)a.eps));
                    if (blockIdx.x == 0) {
                        // nn.BatchNorm2d: running = (1 - momentum) * running + momentum * batch (variance: the unbiased estimate)
                        a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * m;
                        const double unbiased = a.npix > 1 ? var * n / (n - 1.0) : var;
                        a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
                    }
                }
                stat[j] = m; stat[4 + j] = is;
                if (blockIdx.x == 0) { a.mean_out[c] = m; a.invstd_out[c] = is; }
            }
        }
        __syncthreads();
    }
    f32x4 mu, is, ga, be, dr;
    train_channel_consts(a, cg, mu, is, ga, be, dr);
    if (a.has_bn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = stat[j]; is[j] = 4 * cg + j < a.C ? stat[4 + j] : 0.0f; }
    }
    f32x4 *op = a.out + (size_t)(a.out_g0 + cg) * a.npix;
    auto one = [&](const f32x4 &y) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y[j] - mu[j]) * is[j];
            const float z = xh * ga[j] + be[j];
            o[j] = 4 * cg + j < a.C ? train_act(z, a.act) * a.scale * dr[j] : 0.0f;
        }
        return o;
    };
    if (first3) {
        op[p] = one(py0); op[p + stride] = one(py1); op[p + 2 * stride] = one(py2);
        p += 3 * stride;
    }
    for (; p + 2 * stride < a.npix; p += 3 * stride) {  // three loads in flight per lane
        const f32x4 y0 = yp[p], y1 = yp[p + stride], y2 = yp[p + 2 * stride];
        op[p] = one(y0); op[p + stride] = one(y1); op[p + 2 * stride] = one(y2);
    }
    for (; p < a.npix; p += stride) op[p] = one(yp[p]);
}

// dz = dout * scale * drop * act'(z); partial sums of dz and dz * xhat per channel (fp64, fixed order)
__global__ __launch_bounds__(256) void train_bn_bwd_reduce_kernel(const BnGroup grp)
{
    const BnActArgs &a = grp.g[blockIdx.z];
    __shared__ double red[4][8];
    const int cg = blockIdx.y;
    f32x4 mu, is, ga, be, dr;
    train_channel_consts(a, cg, mu, is, ga, be, dr);
    const int per = (a.npix + kTrainSlabs - 1) / kTrainSlabs;
    const int p0 = blockIdx.x * per, p1 = min(a.npix, p0 + per);
    const f32x4 *yp = a.y + (size_t)(a.y_g0 + cg) * a.npix;
    const f32x4 *gp = a.dout + (size_t)(a.dout_g0 + cg) * a.npix;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // max |dz|, max |xhat| per channel (the bound behind dy's factor)
    auto add = [&](const f32x4 &y, const f32x4 &g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y[j] - mu[j]) * is[j];
            const float z = xh * ga[j] + be[j];
            const float dz = g[j] * a.scale * dr[j] * train_act_grad(z, a.act);
            s[j] += (double)dz;
            s[4 + j] += (double)dz * (double)xh;
            mx[j] = fmaxf(mx[j], fabsf(dz));
            mx[4 + j] = fmaxf(mx[4 + j], fabsf(xh));
        }
    };
    int p = p0 + threadIdx.x;
    for (; p + 768 < p1; p += 1024) {  // eight loads in flight per lane, sums in the order of a plain loop
        const f32x4 y4[4] = {yp[p], yp[p + 256], yp[p + 512], yp[p + 768]};
        const f32x4 g4[4] = {gp[p], gp[p + 256], gp[p + 512], gp[p + 768]};
#pragma unroll
        for (int u = 0; u < 4; ++u) add(y4[u], g4[u]);
    }
    for (; p < p1; p += 256) add(yp[p], gp[p]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = s[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8)
        a.partial[((size_t)blockIdx.x * a.c4 + cg) * 8 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (a.bnd) {
        __shared__ float redm[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = mx[j];
            for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
            if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6][j] = v;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int j = threadIdx.x & 3, c = 4 * cg + j;
            float v = fmaxf(fmaxf(redm[0][threadIdx.x], redm[1][threadIdx.x]), fmaxf(redm[2][threadIdx.x], redm[3][threadIdx.x]));
            if (threadIdx.x < 4) v *= fabsf(ga[j] * is[j]);  // (the per-lane constants are the same in every lane)
            // (an infinite maximum orders above every finite one; a NaN never wins an fmaxf: NaN gradients are the
            // reference's NaN gradients, not a range event)
            if (c < a.C && v > 0.0f) atomicMax(a.bnd + (threadIdx.x < 4 ? 0 : a.bnd_stride) + c, __float_as_uint(v));
        }
    }
}

// the power of two that brings `bound` into [2^13, 2^14) (1 for a zero or non-finite bound)
__device__ __forceinline__ float train_dy_factor(float bound)
{
    if (!(bound > 0.0f) || !(bound < 3.0e38f)) return 1.0f;
    int e = 0;
    (void)frexpf(bound, &e);  // bound = f 2^e, f in [0.5, 1)
    int k = 14 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return ldexpf(1.0f, k);
}

__global__ __launch_bounds__(256) void train_bn_bwd_apply_kernel(const BnGroup grp)
{
    const BnActArgs &a = grp.g[blockIdx.z];
    const int cg = blockIdx.y;
    f32x4 mu, is, ga, be, dr;
    train_channel_consts(a, cg, mu, is, ga, be, dr);
    // (the first six loads of the streaming loop go out before the reductions are finished and the bound is scanned: see
    // train_bn_act_fwd_kernel)
    const f32x4 *yp = a.y + (size_t)(a.y_g0 + cg) * a.npix;
    const f32x4 *gp = a.dout + (size_t)(a.dout_g0 + cg) * a.npix;
    const int stride = gridDim.x * blockDim.x;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool first3 = p + 2 * stride < a.npix;
    f32x4 py0{0.f, 0.f, 0.f, 0.f}, py1 = py0, py2 = py0, pg0 = py0, pg1 = py0, pg2 = py0;
    if (first3) {
        py0 = yp[p]; py1 = yp[p + stride]; py2 = yp[p + 2 * stride];
        pg0 = gp[p]; pg1 = gp[p + stride]; pg2 = gp[p + 2 * stride];
    }
    // the two reductions are finished here (see train_group_totals); block column 0 publishes the parameter gradients
    __shared__ float means[8];
    if (threadIdx.x < 64) {
        double tot[8];
        train_group_totals(a.partial, a.c4, cg, tot);
        if (threadIdx.x < 4) {
            const int j = threadIdx.x, c = 4 * cg + j;
            const bool batch = a.has_bn && a.training;
            means[j] = batch ? (float)(tot[j] / (double)a.npix) : 0.0f;          // mean(dz)
            means[4 + j] = batch ? (float)(tot[4 + j] / (double)a.npix) : 0.0f;  // mean(dz * xhat)
            if (blockIdx.x == 0 && c < a.C) {
                if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.0f) + (float)tot[4 + j];
                if (a.dbeta) a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.0f) + (float)tot[j];
                if (a.sum_dy) a.sum_dy[c] = batch ? 0.0f : (float)(tot[j] * (double)(a.has_bn ? (a.gamma ? a.gamma[c] : 1.0f) * a.invstd[c] : 1.0f));
                if (a.dbias) {
                    // sum_p dy: zero under batch statistics (the normalisation removes the mean), gamma * invstd * sum(dz) otherwise
                    const float gi = a.has_bn ? (a.gamma ? a.gamma[c] : 1.0f) * a.invstd[c] : 1.0f;
                    a.dbias[c] = (a.accumulate ? a.dbias[c] : 0.0f) + (batch ? 0.0f : (float)(tot[j] * (double)gi));
                }
            }
        }
    }
    __syncthreads();
    f32x4 m1, m2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { m1[j] = means[j]; m2[j] = means[4 + j]; }
    f32x4 *dp = a.dy + (size_t)(a.dy_g0 + cg) * a.npix;
    float dys = 1.0f;
    if (a.bnd) {  // this pass's factor from the bound the reduce launch left (the same value in every block of the scale group)
        __shared__ float bw[4];
        const int n_units = grp.share_scale ? (int)gridDim.z : 1, per = 4 * a.c4;
        float b = 0.0f;
        // (strided: a wide net - ojf_trainer_create takes widths up to 1024 - has more bound words than the block has threads)
        for (int i = threadIdx.x; i < n_units * per; i += blockDim.x) {
            const int u = i / per, c = i - u * per;
            const BnActArgs &o = grp.g[grp.share_scale ? u : (int)blockIdx.z];
            if (c < o.C) {
                const float A = __uint_as_float(o.bnd[c]), X = __uint_as_float(o.bnd[o.bnd_stride + c]);
                float bi = (o.has_bn && o.training) ? A * (2.0f + X * X) : A;
                if (bi != bi) bi = 3.4e38f;  // (inf * 0)
                b = fmaxf(b, bi);
            }
        }
        for (int off = 32; off > 0; off >>= 1) b = fmaxf(b, __shfl_xor(b, off, 64));
        if ((threadIdx.x & 63) == 0) bw[threadIdx.x >> 6] = b;
        __syncthreads();
        dys = train_dy_factor(fmaxf(fmaxf(bw[0], bw[1]), fmaxf(bw[2], bw[3])));
        if (a.dy_scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.dy_scale_out = dys;
    }
    auto one = [&](const f32x4 &y, const f32x4 &g) {
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y[j] - mu[j]) * is[j];
            const float z = xh * ga[j] + be[j];
            const float dz = g[j] * a.scale * dr[j] * train_act_grad(z, a.act);
            const float v = ga[j] * is[j] * (dz - m1[j] - xh * m2[j]);  // m1 = m2 = 0 without batch statistics
            d[j] = v * dys;
        }
        return d;
    };
    if (first3) {
        dp[p] = one(py0, pg0); dp[p + stride] = one(py1, pg1); dp[p + 2 * stride] = one(py2, pg2);
        p += 3 * stride;
    }
    for (; p + 2 * stride < a.npix; p += 3 * stride) {  // six loads in flight per lane
        const f32x4 y0 = yp[p], y1 = yp[p + stride], y2 = yp[p + 2 * stride];
        const f32x4 g0 = gp[p], g1 = gp[p + stride], g2 = gp[p + 2 * stride];
        dp[p] = one(y0, g0); dp[p + stride] = one(y1, g1); dp[p + 2 * stride] = one(y2, g2);
    }
    for (; p < a.npix; p += stride) dp[p] = one(yp[p], gp[p]);
}

// ---- weight gradient -----------------------------------------------------------------------------------------------
// dW[oc][ic][tap] = sum_p dy[oc][p] * x[ic][p + offset(tap)] (zero outside the image) on the fp32 MFMA, K = pixels.
// Partial sums per pixel slab go to `partial`; wgrad_reduce_kernel adds the slabs in order and writes torch's
// [OC][IC][k][k] layout (physical input channels un-slotted).  (Round 2 started with a register-tiled fp32 FMA kernel -
// 4 x 4 (oc, ic) tile per thread, 40-80 us per layer; the MFMA form takes 12-36 us on the same layers.)
struct WgradArgs {
    const f32x4 *x;   // input planes of the forward conv (window at x_g0, c4_in groups)
    const f32x4 *dy;  // gradient planes of its output (window at dy_g0, c4_out groups)
    float *partial;   // [slabs][taps][ocp][icp], ocp / icp = channel counts rounded up to 32
    int x_g0, c4_in, dy_g0, c4_out, h, w, npix, taps, dil, slabs, ocp, icp;
    int *ovf;  // split-fp16 form: range-guard flag for dy where no backward-data convolution checks it (else NULL)
};

// One wave (= one block) = one 32 x 32 (oc, ic) tile of ONE tap over one
// pixel slab, D += A * B with v_mfma_f32_32x32x2_f32, K = 2 pixels per instruction: lane (r = lane % 32, k = lane / 32)
// supplies A[r][k] = dy[oc0 + r][p + k] and B[k][r] = x[ic0 + r][p + k + tap offset].  The C4 planes hold 4 channels
// per 16 bytes, so those scalars are transposed through LDS: per 64-pixel chunk every lane fetches ITS pixel's eight
// float4 of dy and of x (1 KB contiguous per wave instruction; taps outside the image, channels outside the tensor and
// pixels outside the slab become zeros here) and writes them as a [pixel][32 channels] row (pitch 36 floats:
// conflict-free float4 writes), then 32 K steps read one float per lane and operand.  (Fetching the scalars straight
// from global memory - 16 sixteen-byte segments per wave load - ran at a quarter of the MFMA rate: the L1 address
// path, not the pipe, was the bound.)  Same `partial` layout as the VALU kernel ([slab][tap][ocp][icp], ocp / icp
// multiples of 32), same fixed-order reduction afterwards.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWgChunk = 64, kWgPitch = 36;

struct WgradGroup { WgradArgs g[4]; int tiles; };  // blockIdx.y = unit * tiles + (oc, ic) tile

// F16: split-fp16 form (v_mfma_f32_32x32x16_f16, three per 16 pixels: dy_hi x_hi + dy_hi x_lo + dy_lo x_hi, fp32 accumulate)
// for passes whose dy carries its power-of-two factor (the trainer's second backward pass on) - the matrix pipe's share of a
// 64-pixel chunk drops from 32 x 64 to 12 x 32 cycles.  Same staging; the transposed LDS reads fetch EIGHT consecutive
// pixels of the lane's channel (K = 8 k .. 8 k + 7 of a 16-pixel step, the same pixels for A and for B) and are split in
// registers.  x is an activation the forward convolution already split (inside the fp16 range by its guard); dy is
// guarded by the backward-data convolution that reads the same planes, or here (a.ovf) where there is none.
template <bool F16>
__global__ __launch_bounds__(64) void train_wgrad_mfma_kernel(const WgradGroup grp, unsigned w_magic)
{
    __shared__ __attribute__((aligned(16))) float tile[2][kWgChunk * kWgPitch];
    const int unit = blockIdx.y / grp.tiles, tile_i = blockIdx.y - unit * grp.tiles;
    const WgradArgs &a = grp.g[unit];
    const int lane = threadIdx.x, r = lane & 31, k = lane >> 5;
    const int n_it = a.icp / 32;
    const int ot = tile_i / n_it, it = tile_i - ot * n_it;
    const int tap = blockIdx.z;
    int dyo = 0, dxo = 0;
    if (a.taps == 9) { dyo = (tap / 3 - 1) * a.dil; dxo = (tap % 3 - 1) * a.dil; }
    const int per = ((a.npix + a.slabs - 1) / a.slabs + 1) & ~1;  // even: a K step never straddles two slabs
    const int p0 = blockIdx.x * per, p1 = min(a.npix, p0 + per);
    const int shift = dyo * a.w + dxo;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    f32x4 dv[8], xv[8];
    float gmax = 0.0f;
    // buffer loads: a lane outside the slab / a tap outside the image gets an offset beyond the descriptor's range and
    // reads zeros, and so does a channel group outside the tensor - no per-load branches (as plain predicated loads the
    // sixteen fetches of a chunk were sixteen exec-mask regions, ~130 VALU + ~240 scalar instructions)
    const int n_og = min(8, a.c4_out - ot * 8), n_ig = min(8, a.c4_in - it * 8);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.dy + (size_t)(a.dy_g0 + ot * 8) * a.npix), 0, n_og * a.npix * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.x + (size_t)(a.x_g0 + it * 8) * a.npix), 0, n_ig * a.npix * 16, 0x00020000);
    const int plane = a.npix * 16;
    auto fetch = [&](int pc) {  // this lane's pixel of the chunk at pc: its eight channel groups of dy and of x
        const int p = pc + lane;
        const int py = fast_div(p, a.w, w_magic), px = p - py * a.w;
        const bool in = p < p1;
        const bool tap_ok = in && (unsigned)(py + dyo) < (unsigned)a.h && (unsigned)(px + dxo) < (unsigned)a.w;
        const unsigned od = in ? (unsigned)p * 16u : 0x80000000u, ox = tap_ok ? (unsigned)(p + shift) * 16u : 0x80000000u;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // (the plane offset rides in the per-lane offset, which the range check sees: groups past the tensor read zeros too)
            dv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, od + (unsigned)(g * plane), 0, 0));
            xv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ox + (unsigned)(g * plane), 0, 0));
        }
    };
    fetch(p0);
    for (int pc = p0; pc < p1; pc += kWgChunk) {
        __syncthreads();  // (one wave per block: orders the previous chunk's reads before these writes)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            *reinterpret_cast<f32x4 *>(&tile[0][lane * kWgPitch + 4 * g]) = dv[g];
            *reinterpret_cast<f32x4 *>(&tile[1][lane * kWgPitch + 4 * g]) = xv[g];
        }
        if constexpr (F16)
            if (a.ovf) {
#pragma unroll
                for (int g = 0; g < 8; ++g) gmax = guard_max(gmax, dv[g]);
            }
        __syncthreads();
        fetch(pc + kWgChunk);  // the next chunk travels while this one is multiplied (past the slab: all zeros, no traffic)
        if constexpr (F16) {
            // a register block = one 16-pixel K step: 8 + 8 transposed reads, two splits, three MFMAs
            float av[2][8], bv[2][8];
            auto read_step = [&](int s, float (&ar)[8], float (&br)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ar[u] = tile[0][(16 * s + 8 * k + u) * kWgPitch + r];
                    br[u] = tile[1][(16 * s + 8 * k + u) * kWgPitch + r];
                }
            };
            read_step(0, av[0], bv[0]);
#pragma unroll
            for (int s = 0; s < kWgChunk / 16; ++s) {
                if (s + 1 < kWgChunk / 16) read_step(s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const float(&ar)[8] = av[s & 1];
                const float(&br)[8] = bv[s & 1];
                f16x8 dh, dl, xh, xl;
                split_f16(f32x4{ar[0], ar[1], ar[2], ar[3]}, f32x4{ar[4], ar[5], ar[6], ar[7]}, dh, dl);
                split_f16(f32x4{br[0], br[1], br[2], br[3]}, f32x4{br[4], br[5], br[6], br[7]}, xh, xl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, xh, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // operands of eight K steps per register block; the next block's LDS reads are issued before this block's
            // MFMAs (read -> wait -> 2 MFMAs, as the compiler schedules the plain loop, exposes the LDS latency 16 times)
            float av[2][8], bv[2][8];
            auto read_block = [&](int blk, float (&ar)[8], float (&br)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ar[u] = tile[0][(2 * (8 * blk + u) + k) * kWgPitch + r];
                    br[u] = tile[1][(2 * (8 * blk + u) + k) * kWgPitch + r];
                }
            };
            read_block(0, av[0], bv[0]);
#pragma unroll
            for (int blk = 0; blk < kWgChunk / 16; ++blk) {
                if (blk + 1 < kWgChunk / 16) read_block(blk + 1, av[(blk + 1) & 1], bv[(blk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks every read next to its MFMA again)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[blk & 1][u], bv[blk & 1][u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (F16)
        if (a.ovf && gmax > 65504.0f) guard_raise(a.ovf, 1);
    // D layout: acc[v] = D[i][j], j = lane % 32 (ic), i = 8 * (v / 4) + 4 * (lane / 32) + v % 4 (oc)
    float *dst = a.partial + (((size_t)blockIdx.x * a.taps + tap) * a.ocp + (size_t)ot * 32) * a.icp + (size_t)it * 32 + r;
#pragma unroll
    for (int v = 0; v < 16; ++v) dst[(size_t)(8 * (v / 4) + 4 * k + (v % 4)) * a.icp] = acc[v];
}

// (Round 3 tried a row-merged form for the 3x3 layers - one wave = the three taps of a kernel row, dy staged once per three
// taps, x once per 64 + 2 d pixel band with per-pixel tap masks at read time, a third of the traffic through L2: correct,
// but 174 against 191 frames/s for the training step - a third of the waves with three accumulators each hide the LDS
// transposes worse than they save fetches.)
struct WgradReduceArgs {
    const float *partial;
    float *dw;  // [OC][IC][taps]
    int slabs, taps, ocp, icp, OC, IC, group, slot, c_in_phys;
    int accumulate;  // dw += (gradient accumulation over frames happens here instead of in a torch add per parameter)
    int oc_base;     // first row of `partial` that belongs to this weight tensor (stacked layers; else 0)
    int ic_base, ic_total;  // the gradient covers input channels [ic_base, ic_base + IC) of a tensor with ic_total of them
    const float *dy_scale;  // dy carried this power-of-two factor (NULL: none): the sums are divided by it
};
struct WgradReduceGroup { WgradReduceArgs g[4]; };  // blockIdx.y = unit

// eight lanes per weight: lane `sub` adds slabs sub, sub + 8, ... in order, then a fixed xor tree joins the eight sums
__global__ __launch_bounds__(256) void train_wgrad_reduce_kernel(const WgradReduceGroup grp)
{
    const WgradReduceArgs &a = grp.g[blockIdx.y];
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = t >> 3;
    const int sub = (int)(t & 7);
    const long total = (long)a.taps * a.OC * a.c_in_phys;
    const bool live = i < total;
    const long ii = live ? i : 0;
    const int icp_i = (int)(ii % a.c_in_phys);
    const long r = ii / a.c_in_phys;
    const int oc = (int)(r % a.OC), tap = (int)(r / a.OC);
    float s = 0.0f;
    if (live)
        for (int b = sub; b < a.slabs; b += 8) s += a.partial[(((size_t)b * a.taps + tap) * a.ocp + a.oc_base + oc) * a.icp + icp_i];
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (!live || sub) return;
    if (a.dy_scale) s /= *a.dy_scale;  // exact: a power of two
    const int ic = train_unslot(icp_i, a.group, a.slot, a.IC);
    if (ic >= 0) {
        float *d = a.dw + ((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + tap;
        *d = a.accumulate ? *d + s : s;
    }
}

// nn.AvgPool2d(3, stride 1, padding 1), count_include_pad: out = (sum of the 3x3 neighbourhood inside the image) / 9.  The
// operator is symmetric, so its backward pass is the same kernel applied to the gradient.
__global__ __launch_bounds__(256) void train_avgpool3_kernel(const f32x4 *in, f32x4 *out, int h, int w)
{
    const int npix = h * w;
    const f32x4 *ip = in + (size_t)blockIdx.y * npix;
    f32x4 *op = out + (size_t)blockIdx.y * npix;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        f32x4 s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx)
                if ((unsigned)(y + dy) < (unsigned)h && (unsigned)(x + dx) < (unsigned)w) s += ip[(y + dy) * w + x + dx];
        op[p] = s / 9.0f;
    }
}

static float *g_train_zero_bias = nullptr;  // 4096 zero floats: the conv kernel always reads a bias vector

}  // namespace ojf

// ---- C ABI ----------------------------------------------------------------------------------------------------------
OJF_API size_t ojf_train_packed_floats(int c_out_phys, int c_in_phys, int ksize)
{
    using namespace ojf;
    if (c_out_phys < 1 || c_in_phys < 4 || c_in_phys % 4 || (ksize != 1 && ksize != 3)) return 0;
    const int n_ot = round_up(round_up(c_out_phys, 16) / 16, kNT), nsteps = (ksize * ksize * (c_in_phys / 4) + 3) / 4;
    if (nsteps > kMaxSteps) return 0;
    return (size_t)n_ot * (nsteps + kPadSteps) * 256;
}

OJF_API int ojf_train_pack(const float *w, const float *bias, int OC, int IC, int ksize, int group, int slot, int c_in_phys,
                           int c_out_phys, int transposed, float *packed, float *bias_packed, ojf_stream_t stream)
{
    using namespace ojf;
    if (!w || !packed || OC < 1 || IC < 1 || group < 1 || slot < group) return fail("ojf_train_pack: bad argument");
    // forward: rows = output channels (c_out_phys), K = input planes (c_in_phys); transposed: rows = input planes, K = output channels
    const int rows = transposed ? c_in_phys : c_out_phys, kch = transposed ? c_out_phys : c_in_phys;
    if (ojf_train_packed_floats(rows, kch, ksize) == 0) return fail("ojf_train_pack: unsupported layer shape (K too long or channels not padded to 4)");
    PackArgs a;
    a.w = w; a.bias = transposed ? nullptr : bias; a.wp = packed; a.bp = transposed ? nullptr : bias_packed;
    a.OC = OC; a.IC = IC; a.taps = ksize * ksize; a.group = group; a.slot = slot;
    a.c4 = kch / 4; a.nsteps = (a.taps * a.c4 + 3) / 4; a.n_ot = round_up(round_up(rows, 16) / 16, kNT); a.transposed = transposed ? 1 : 0;
    a.oc_base = 0; a.partial = 0; a.ic_base = 0; a.ic_total = IC;
    const long total = (long)a.n_ot * (a.nsteps + kPadSteps) * 256;
    hipLaunchKernelGGL(train_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "train_pack_kernel launch");
}

OJF_API int ojf_train_conv(const float *in, int in_g0, int c_in_phys, float *out, int out_g0, int c_out_phys, const float *packed,
                           const float *bias_packed, int ksize, int dil, int h, int w, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || !packed) return fail("ojf_train_conv: null pointer argument");
    if (ojf_train_packed_floats(c_out_phys, c_in_phys, ksize) == 0 || dil < 1 || h < 1 || w < 1) return fail("ojf_train_conv: unsupported shape");
    if (!g_train_zero_bias) {
        OJF_HIP(hipMalloc(reinterpret_cast<void **>(&g_train_zero_bias), 4096 * sizeof(float)));
        OJF_HIP(hipMemset(g_train_zero_bias, 0, 4096 * sizeof(float)));
        OJF_HIP(hipStreamSynchronize(nullptr));
    }
    const int taps = ksize * ksize, c4 = c_in_phys / 4, nsteps = (taps * c4 + 3) / 4;
    const int n_ot = round_up(round_up(c_out_phys, 16) / 16, kNT), og_total = round_up(c_out_phys, 4) / 4;
    if (n_ot * 16 > 4096) return fail("ojf_train_conv: too many output channels");
    for (int ot0 = 0; ot0 < n_ot; ot0 += 8) {  // a launch covers up to 8 output tiles per wave
        const int nt = n_ot - ot0 < 8 ? n_ot - ot0 : 8;
        ConvArgs a;
        a.ovf = nullptr; a.accum = 0; a.dscale = nullptr;
        a.in = planes(in); a.out = planes(out); a.out_rows = nullptr;
        a.wp = planes(packed) + (size_t)ot0 * (nsteps + kPadSteps) * 64;
        a.bias = (bias_packed ? bias_packed : g_train_zero_bias) + (size_t)ot0 * 16; a.rinv = nullptr;
        a.in_g0 = in_g0; a.out_g0 = out_g0 + ot0 * 4; a.rows_stride = 0; a.rows_n = 0;
        a.h = h; a.w = w; a.npix = h * w; a.taps = taps; a.dil = dil; a.c4 = c4; a.nsteps = nsteps;
        a.w_magic = 0; a.c4_magic = 0;
        const int left = og_total - ot0 * 4;
        a.og_store = left < nt * 4 ? left : nt * 4;
        a.act = OJF_ACT_NONE; a.act_n = 0; a.scale = 1.0f;
        if (launch_conv_args(&a, 1, nt, as_stream(stream), OJF_ARITH_F32)) return -2;
    }
    return 0;
}

OJF_API int ojf_train_avgpool3(const float *in, float *out, int c_phys, int h, int w, ojf_stream_t stream)
{
    using namespace ojf;
    if (!in || !out || in == out || c_phys < 4 || c_phys % 4 || h < 1 || w < 1) return fail("ojf_train_avgpool3: bad argument");
    const int bx = (h * w + 255) / 256 < 256 ? (h * w + 255) / 256 : 256;
    hipLaunchKernelGGL(train_avgpool3_kernel, dim3(bx, c_phys / 4), dim3(256), 0, as_stream(stream), planes(in), planes(out), h, w);
    return check_hip(hipGetLastError(), "train_avgpool3_kernel launch");
}

// Per-channel sums over the frame as kTrainSlabs partial rows ([slab][c_phys / 4][8] doubles: sums in [0..3], sums of
// squares in [4..7]; the caller adds the rows): the global-average pooling of a VortexPooling in the training path.
OJF_API int ojf_train_channel_sums(const float *y, int y_g0, int c_phys, int h, int w, double *partial, ojf_stream_t stream)
{
    using namespace ojf;
    if (!y || !partial || c_phys < 4 || c_phys % 4 || h < 1 || w < 1) return fail("ojf_train_channel_sums: bad argument");
    hipLaunchKernelGGL(train_stats_partial_kernel, dim3(kTrainSlabs, c_phys / 4), dim3(256), 0, as_stream(stream), planes(y), y_g0, h * w, partial);
    return check_hip(hipGetLastError(), "train_stats_partial_kernel launch");
}

OJF_API size_t ojf_train_partial_doubles(int c_phys) { return (size_t)ojf::kTrainSlabs * (c_phys / 4) * 8; }

static ojf::BnActArgs train_bn_args(const float *y, int y_g0, int c_phys, int C, int h, int w, const float *mean, const float *invstd,
                                    const float *gamma, const float *beta, const float *drop, int act, float scale, int has_bn, int training)
{
    ojf::BnActArgs a;
    a.y = ojf::planes(y); a.out = nullptr; a.dout = nullptr; a.dy = nullptr;
    a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.drop = drop; a.partial = nullptr;
    a.y_g0 = y_g0; a.out_g0 = 0; a.dout_g0 = 0; a.dy_g0 = 0; a.c4 = c_phys / 4; a.C = C; a.npix = h * w; a.act = act;
    a.has_bn = has_bn; a.training = training; a.scale = scale;
    a.mean_out = a.invstd_out = a.running_mean = a.running_var = nullptr; a.momentum = 0.0f; a.eps = 0.0f;
    a.dgamma = a.dbeta = a.dbias = nullptr; a.sum_dy = nullptr; a.accumulate = 0; a.bnd = nullptr; a.bnd_stride = 0; a.dy_scale_out = nullptr;
    return a;
}

OJF_API int ojf_train_bn_act(const float *y, int y_g0, float *out, int out_g0, int c_phys, int C, int h, int w, const float *gamma,
                             const float *beta, const float *drop, int act, float scale, int has_bn, int training, float momentum,
                             float eps, float *running_mean, float *running_var, double *partial, float *mean, float *invstd,
                             ojf_stream_t stream)
{
    using namespace ojf;
    if (!y || !out || c_phys % 4 || C > c_phys) return fail("ojf_train_bn_act: bad argument");
    if (has_bn && (!mean || !invstd || !partial || !running_mean || !running_var))
        return fail("ojf_train_bn_act: BatchNorm needs mean / invstd outputs, the partial-sum scratch and the running statistics");
    hipStream_t st = as_stream(stream);
    if (has_bn && training)
        hipLaunchKernelGGL(train_stats_partial_kernel, dim3(kTrainSlabs, c_phys / 4), dim3(256), 0, st, planes(y), y_g0, h * w, partial);
    BnActArgs a = train_bn_args(y, y_g0, c_phys, C, h, w, nullptr, nullptr, gamma, beta, drop, act, scale, has_bn, training ? 1 : 0);
    a.out = planes(out); a.out_g0 = out_g0; a.partial = partial;
    a.mean_out = mean; a.invstd_out = invstd; a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps;
    const int bx = (h * w + 255) / 256 < 128 ? (h * w + 255) / 256 : 128;
    hipLaunchKernelGGL(train_bn_act_fwd_kernel, dim3(bx, c_phys / 4), dim3(256), 0, st, BnGroup{{a, a, a, a}});
    return check_hip(hipGetLastError(), "train_bn_act kernels launch");
}

OJF_API int ojf_train_bn_act_bwd(const float *y, int y_g0, const float *dout, int dout_g0, float *dy, int dy_g0, int c_phys, int C,
                                 int h, int w, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                 const float *drop, int act, float scale, int has_bn, int training, double *partial,
                                 float *dgamma, float *dbeta, float *dbias, int accumulate, ojf_stream_t stream)
{
    using namespace ojf;
    if (!y || !dout || !dy || !partial || c_phys % 4 || C > c_phys || (has_bn && (!mean || !invstd)))
        return fail("ojf_train_bn_act_bwd: bad argument");
    hipStream_t st = as_stream(stream);
    BnActArgs a = train_bn_args(y, y_g0, c_phys, C, h, w, mean, invstd, gamma, beta, drop, act, scale, has_bn, training);
    a.dout = planes(dout); a.dout_g0 = dout_g0; a.dy = planes(dy); a.dy_g0 = dy_g0; a.partial = partial;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.accumulate = accumulate ? 1 : 0;
    const BnGroup grp{{a, a, a, a}};
    hipLaunchKernelGGL(train_bn_bwd_reduce_kernel, dim3(kTrainSlabs, c_phys / 4), dim3(256), 0, st, grp);
    const int bx = (h * w + 255) / 256 < 128 ? (h * w + 255) / 256 : 128;
    hipLaunchKernelGGL(train_bn_bwd_apply_kernel, dim3(bx, c_phys / 4), dim3(256), 0, st, grp);
    return check_hip(hipGetLastError(), "train_bn_bwd kernels launch");
}

namespace ojf {
struct WgradPlan { int ocp, icp, slabs; };
static WgradPlan wgrad_plan(int c_out_phys, int c_in_phys, int taps, int npix, int units = 1)
{
    WgradPlan p;
    p.ocp = round_up(c_out_phys, 32);
    p.icp = round_up(c_in_phys, 32);
    // two waves per SIMD (2048 one-wave blocks) hide the load latency; at most 256 slabs for the reduction
    const int waves = units * taps * (p.ocp / 32) * (p.icp / 32);  // (units: layers of identical shape sharing one grouped launch)
    int slabs = (2048 + waves - 1) / waves;
    slabs = slabs > 256 ? 256 : slabs;
    const int max_slabs = (npix + 63) / 64;
    slabs = slabs > max_slabs ? max_slabs : slabs;
    p.slabs = slabs < 1 ? 1 : slabs;
    return p;
}
}  // namespace ojf

OJF_API size_t ojf_train_wgrad_partial_floats(int c_out_phys, int c_in_phys, int ksize, int h, int w)
{
    const ojf::WgradPlan p = ojf::wgrad_plan(c_out_phys, c_in_phys, ksize * ksize, h * w);
    return (size_t)p.slabs * ksize * ksize * p.ocp * p.icp;
}

OJF_API int ojf_train_wgrad(const float *x, int x_g0, int c_in_phys, const float *dy, int dy_g0, int c_out_phys, int OC, int IC,
                            int ksize, int dil, int group, int slot, int h, int w, float *partial, float *dw, int accumulate, ojf_stream_t stream)
{
    using namespace ojf;
    if (!x || !dy || !partial || !dw || c_in_phys % 4 || c_out_phys % 4 || (ksize != 1 && ksize != 3) || OC > c_out_phys)
        return fail("ojf_train_wgrad: bad argument");
    hipStream_t st = as_stream(stream);
    const int taps = ksize * ksize;
    const WgradPlan p = wgrad_plan(c_out_phys, c_in_phys, taps, h * w);
    WgradArgs a;
    a.x = planes(x); a.dy = planes(dy); a.partial = partial; a.x_g0 = x_g0; a.c4_in = c_in_phys / 4; a.dy_g0 = dy_g0; a.c4_out = c_out_phys / 4;
    a.h = h; a.w = w; a.npix = h * w; a.taps = taps; a.dil = dil; a.slabs = p.slabs; a.ocp = p.ocp; a.icp = p.icp;
    a.ovf = nullptr;
    hipLaunchKernelGGL(train_wgrad_mfma_kernel<false>, dim3(p.slabs, (p.ocp / 32) * (p.icp / 32), taps), dim3(64), 0, st,
                       WgradGroup{{a, a, a, a}, (p.ocp / 32) * (p.icp / 32)}, div_magic(w, (uint64_t)h * w + 2 * kWgChunk));
    WgradReduceArgs r;
    r.partial = partial; r.dw = dw; r.slabs = p.slabs; r.taps = taps; r.ocp = p.ocp; r.icp = p.icp; r.OC = OC; r.IC = IC;
    r.group = group; r.slot = slot; r.c_in_phys = c_in_phys; r.accumulate = accumulate ? 1 : 0; r.oc_base = 0; r.ic_base = 0; r.ic_total = IC; r.dy_scale = nullptr;
    const long total = (long)taps * OC * c_in_phys * 8;
    hipLaunchKernelGGL(train_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, WgradReduceGroup{{r, r, r, r}});
    return check_hip(hipGetLastError(), "train_wgrad kernels launch");
}
