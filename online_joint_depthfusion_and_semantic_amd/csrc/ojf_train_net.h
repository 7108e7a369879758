// Whole-net TRAINING executor (ojf_trainer_*): forward and backward pass of FusionNet_v3 / _v2 in train() or eval()
// mode as two C calls - modules/pipeline.py:322 (`self._fusion(...)` inside fuse_training) and the `loss.backward()` of
// train_fusion.py:171 - instead of one Python autograd node per layer (included by ojf_net.hip only).
//
// Why: round 2's training frame step issued 712 launches from ~60 Python autograd nodes (10 ms of host time, 9.6 ms of
// kernels per 320x240 frame).  Here the layer walk is C++, every activation / gradient buffer is owned by the trainer and
// allocated once, and the four branches of a VortexPooling run as GROUPED launches (one launch = the same unit of all
// four branches, blockIdx.y / z picks the branch) - convolution, statistics, normalise + activate, both BatchNorm-
// backward kernels, weight gradient and backward-data alike.  The branch-entry 1x1 convolutions see pooled inputs
// (modules/model.py:143-161: branch i reads pool^i(x)); a 3x3 average pool (zero padding, count_include_pad) is linear and
// acts on pixels while a 1x1 convolution acts on channels, so W_i (P^i x) + b_i = P^i (W_i x) + b_i: the four entries are
// ONE stacked 114 -> 4 x 19 convolution of the unpooled input followed by the pools on 19 instead of 114 channels (the
// same identity the inference path uses); backward pools the 19-channel gradients and runs ONE weight-gradient launch.
//
// Layer units, arithmetic and reductions are those of ojf_net_train.h (fp32-MFMA convolutions, fp64 fixed-order slab
// sums); parameters, BatchNorm buffers and gradient tensors stay the caller's (torch's) - the trainer only reads / writes
// them through the pointer table handed over with every call.
#pragma once

#include <cstring>
#include <map>

namespace ojf {

// ---- small kernels of the executor ---------------------------------------------------------------------------------
struct PackInArgs {
    const float *src[4];  // NCHW sources [n_i][npix]
    int nch[4];
    int n_src, c4, npix;
    f32x4 *dst;           // planes [c4][npix]
};

__global__ __launch_bounds__(256) void train_pack_input_kernel(const PackInArgs a)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npix) return;
    const int cg = blockIdx.y;
    f32x4 v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int l = 4 * cg + j;
        for (int s = 0; s < a.n_src; ++s) {
            if (l < a.nch[s]) { v[j] = a.src[s][(size_t)l * a.npix + p]; break; }
            l -= a.nch[s];
        }
    }
    a.dst[(size_t)cg * a.npix + p] = v;
}

// planes [c4][npix] <-> NCHW [C][npix] (est out / d est in); channels >= C are dropped / zero
__global__ __launch_bounds__(256) void train_planes_to_nchw_kernel(const f32x4 *pl, int c4, int C, int npix, float *dst)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const f32x4 v = pl[(size_t)blockIdx.y * npix + p];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * (int)blockIdx.y + j < C) dst[(size_t)(4 * blockIdx.y + j) * npix + p] = v[j];
}

__global__ __launch_bounds__(256) void train_nchw_to_planes_kernel(const float *src, int C, int npix, f32x4 *pl)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    f32x4 v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * (int)blockIdx.y + j < C) v[j] = src[(size_t)(4 * blockIdx.y + j) * npix + p];
    pl[(size_t)blockIdx.y * npix + p] = v;
}

// out[group g] = P^lv(in[group g]) (+ bias of the branch), lv = g / sl4: branch i of a VortexPooling sees i successive
// nn.AvgPool2d(3, 1, 1) (count_include_pad: zeros outside the image, divide by 9 - every level zero-pads anew).  One block =
// one 32 x 8 pixel tile of one channel group, the intermediate levels live in LDS with a shrinking halo.  The operator is
// symmetric, so the backward pass is the same kernel on the gradient (without bias).
struct TrainPyramidArgs {
    const f32x4 *in;
    f32x4 *out;
    const float *bias[4];  // per branch: [OC] logical, or NULL
    int h, w, sl4, OC;
};
constexpr int kTpW = 32, kTpH = 8, kTpStride = kTpW + 6;

__global__ __launch_bounds__(256) void train_pyramid_kernel(const TrainPyramidArgs a)
{
    __shared__ f32x4 buf[2][kTpStride * (kTpH + 6)];
    const int g = blockIdx.y, lv = g / a.sl4, cg = g - lv * a.sl4;
    const int tiles_x = (a.w + kTpW - 1) / kTpW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * kTpW, y0 = ty * kTpH, npix = a.h * a.w;
    const f32x4 *plane = a.in + (size_t)g * npix;
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
    f32x4 b = zero;
    if (a.bias[lv])
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * cg + j < a.OC) b[j] = a.bias[lv][4 * cg + j];
    if (lv == 0) {
        const int ly = threadIdx.x / kTpW, lx = threadIdx.x - ly * kTpW;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy < a.h && gx < a.w) a.out[(size_t)g * npix + gy * a.w + gx] = plane[gy * a.w + gx] + b;
        return;
    }
    {
        const int W0 = kTpW + 2 * lv, H0 = kTpH + 2 * lv;
        for (int i = threadIdx.x; i < W0 * H0; i += 256) {
            const int ly = i / W0, lx = i - ly * W0;
            const int gy = y0 - lv + ly, gx = x0 - lv + lx;
            const bool in = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            buf[0][ly * kTpStride + lx] = in ? plane[gy * a.w + gx] : zero;
        }
    }
    __syncthreads();
    for (int l = 1; l <= lv; ++l) {
        const int halo = lv - l;
        const int Wl = kTpW + 2 * halo, Hl = kTpH + 2 * halo;
        const f32x4 *src = buf[(l - 1) & 1];
        for (int i = threadIdx.x; i < Wl * Hl; i += 256) {
            const int ly = i / Wl, lx = i - ly * Wl;
            const int gy = y0 - halo + ly, gx = x0 - halo + lx;
            const bool in = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            f32x4 s = zero;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) s += src[(ly + dy) * kTpStride + lx + dx];
            s = s / 9.0f;
            if (l < lv) buf[l & 1][ly * kTpStride + lx] = in ? s : zero;
            else if (in) a.out[(size_t)g * npix + gy * a.w + gx] = s + b;
        }
        __syncthreads();
    }
}

// ---- global-average branch of a VortexPooling (modules/model.py:103-109,146): AdaptiveAvgPool2d(1) -> 1x1 conv ->
// up-sampling of the 1x1 map (a broadcast) -> BatchNorm2d.  The BatchNorm sees a constant map: with batch statistics its
// output is beta (variance 0), nothing upstream receives a gradient, running_mean moves towards the map's value and
// running_var towards 0; with running statistics it is an affine map of g = W mean(x) + b and gradients flow to W, b,
// gamma, beta and (as a per-channel constant over all pixels) to x.  One block does the vector work.
struct GaveTrainArgs {
    const double *partial;   // [slabs][c4_in][8]: channel sums of x (forward) / of d cat[gave slot] (backward)
    int c4_in, c4_out, IC, OC, group, slot, npix, training, accumulate;
    float momentum, eps;
    const float *W, *b, *gamma, *beta;
    float *running_mean, *running_var;
    float *pooled, *xhat, *gis;  // saved for backward: [4 c4_in] mean of x per physical channel, [OC] x-hat, [OC] gamma * invstd
    float *vec;                  // forward: [4 c4_out] value of the branch's constant map (padding channels 0)
    float *dW, *db, *dgamma, *dbeta;
    float *dpooled;              // backward (eval): [4 c4_in] per-channel constant added to d x
    // The branch's constant map is input channels [0, OC) of the VortexPooling's final 1x1 convolution (weights Wf
    // [OCf][ICf_total], bias bf): over a constant map that convolution is a per-frame BIAS, Wf[:, :OC] vec - so the map is
    // never materialised, the final convolution reads the four branch slots only, and backward needs just the channel
    // sums of the final convolution's dy: d vec = Wf[:, :OC]^T sum_dy,  d Wf[:, :OC] = sum_dy (x) vec.
    const float *Wf, *bf;
    int OCf, ICf_total, accumulate_f;
    float *bias_eff;             // forward: [OCf] bf + Wf[:, :OC] vec -> the final unit's packed bias
    const float *sum_dy;         // backward: [OCf]
    float *dWf;                  // backward: the final convolution's weight gradient (columns [0, OC) written here)
};

// sum of the kTrainSlabs partial rows of physical channel ch, by one wave (lane b takes row b; fixed xor tree)
__device__ __forceinline__ double gave_slab_sum(const double *partial, int c4, int ch)
{
    double v = partial[((size_t)(threadIdx.x & 63) * c4 + (ch >> 2)) * 8 + (ch & 3)];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int kGaveThreads = 1024, kGaveMaxC = 1024;

__global__ __launch_bounds__(kGaveThreads) void train_gave_fwd_kernel(const GaveTrainArgs a)
{
    __shared__ float pooled[kGaveMaxC], vecs[kGaveMaxC];
    const int wave = threadIdx.x >> 6, nw = kGaveThreads / 64;
    for (int ch = wave; ch < 4 * a.c4_in; ch += nw) {
        const float m = (float)(gave_slab_sum(a.partial, a.c4_in, ch) / (double)a.npix);
        if ((threadIdx.x & 63) == 0) { pooled[ch] = m; a.pooled[ch] = m; }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * a.c4_out; o += kGaveThreads) {
        float vec = 0.0f;
        if (o < a.OC) {
            float g = a.b ? a.b[o] : 0.0f;
            for (int l = 0; l < a.IC; ++l) g = fmaf(a.W[(size_t)o * a.IC + l], pooled[(l / a.group) * a.slot + l % a.group], g);
            const float ga = a.gamma ? a.gamma[o] : 1.0f, be = a.beta ? a.beta[o] : 0.0f;
            if (a.training) {
                a.running_mean[o] = (1.0f - a.momentum) * a.running_mean[o] + a.momentum * g;
                a.running_var[o] = (1.0f - a.momentum) * a.running_var[o];
                a.xhat[o] = 0.0f;
                a.gis[o] = 0.0f;
                vec = be;
            } else {
                const float is = 1.0f / sqrtf(a.running_var[o] + a.eps);
                const float xh = (g - a.running_mean[o]) * is;
                a.xhat[o] = xh;
                a.gis[o] = ga * is;
                vec = xh * ga + be;
            }
        }
        a.vec[o] = vec;
        vecs[o] = vec;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < a.OCf; o += kGaveThreads) {  // the constant map's share of the final convolution = a bias
        float b = a.bf ? a.bf[o] : 0.0f;
        for (int c = 0; c < a.OC; ++c) b = fmaf(a.Wf[(size_t)o * a.ICf_total + c], vecs[c], b);
        a.bias_eff[o] = b;
    }
}

__global__ __launch_bounds__(kGaveThreads) void train_gave_bwd_kernel(const GaveTrainArgs a)
{
    __shared__ float dg[kGaveMaxC], sdy[kGaveMaxC];
    for (int o = threadIdx.x; o < a.OCf; o += kGaveThreads) sdy[o] = a.sum_dy[o];
    __syncthreads();
    for (int c = threadIdx.x; c < a.OC; c += kGaveThreads) {
        float dv = 0.0f;  // d vec = Wf[:, :OC]^T sum_dy
        for (int o = 0; o < a.OCf; ++o) dv = fmaf(a.Wf[(size_t)o * a.ICf_total + c], sdy[o], dv);
        if (a.dbeta) a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.0f) + dv;
        if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.0f) + dv * a.xhat[c];
        const float d = dv * a.gis[c];  // 0 under batch statistics
        dg[c] = d;
        if (a.db) a.db[c] = (a.accumulate ? a.db[c] : 0.0f) + d;
    }
    __syncthreads();
    if (a.dW)
        for (int i = threadIdx.x; i < a.OC * a.IC; i += kGaveThreads) {
            const int o = i / a.IC, l = i - o * a.IC;
            a.dW[i] = (a.accumulate ? a.dW[i] : 0.0f) + dg[o] * a.pooled[(l / a.group) * a.slot + l % a.group];
        }
    if (a.dWf)
        for (int i = threadIdx.x; i < a.OCf * a.OC; i += kGaveThreads) {
            const int o = i / a.OC, c = i - o * a.OC;
            float *d = a.dWf + (size_t)o * a.ICf_total + c;
            *d = (a.accumulate_f ? *d : 0.0f) + sdy[o] * a.vec[c];
        }
    if (a.training) return;  // nothing flows back to x under batch statistics (dg == 0)
    for (int ch = threadIdx.x; ch < 4 * a.c4_in; ch += kGaveThreads) {
        const int s = ch / a.slot, in = ch - s * a.slot, l = s * a.group + in;
        float d = 0.0f;
        if (in < a.group && l < a.IC)
            for (int o = 0; o < a.OC; ++o) d = fmaf(a.W[(size_t)o * a.IC + l], dg[o], d);
        a.dpooled[ch] = d / (float)a.npix;
    }
}

// planes[g][p] = vec[4g .. 4g+3] (add != 0: +=)
__global__ __launch_bounds__(256) void train_bcast_planes_kernel(const float *vec, f32x4 *pl, int npix, int add)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(vec + 4 * blockIdx.y);
    f32x4 *d = pl + (size_t)blockIdx.y * npix + p;
    *d = add ? *d + v : v;
}

// ---- the frame step's glue around the net (modules/pipeline.py:104-135, utils/loss.py:65-103) -------------------------
// fused[r][k] = (max(w, 0) v + clamp(est, +-init)) / (max(w, 0) + 1) at pixel valid[r], sample k: the "weighted update"
// of pipeline.py:107-116 followed by the masking of :121-127, from the plane layout [P][n] into the API's rows [Nv][P].
struct FuseOutArgs {
    const float *est, *fv, *fw;  // [P][n]
    const long long *valid;      // [Nv] pixel indices
    float *rows;                 // forward: fused [Nv][P]; backward: d fused [Nv][P] (read)
    float *d_est;                // backward: [P][n], zeroed by the caller
    int n, P;
    long long n_valid;
    float init;
};

__global__ __launch_bounds__(256) void train_fuse_output_kernel(const FuseOutArgs a)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_valid) return;
    const long long p = a.valid[r];
    for (int k = 0; k < a.P; ++k) {
        const size_t i = (size_t)k * a.n + p;
        float w = a.fw[i];
        w = w < 0.0f ? 0.0f : w;                                        // pipeline.py:113-114
        float e = a.est[i];
        e = e < -a.init ? -a.init : (e > a.init ? a.init : e);           // :109-111 (NaN passes through like torch.clamp)
        a.rows[(size_t)r * a.P + k] = (w * a.fv[i] + e) / (w + 1.0f);   // :116
    }
}

__global__ __launch_bounds__(256) void train_fuse_output_bwd_kernel(const FuseOutArgs a)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_valid) return;
    const long long p = a.valid[r];
    for (int k = 0; k < a.P; ++k) {
        const size_t i = (size_t)k * a.n + p;
        float w = a.fw[i];
        w = w < 0.0f ? 0.0f : w;
        const float e = a.est[i];
        const bool pass = !(e < -a.init) && !(e > a.init);  // torch.clamp: the gradient passes inside the closed interval
        a.d_est[i] = pass ? a.rows[(size_t)r * a.P + k] / (w + 1.0f) : 0.0f;
    }
}

// FusionLoss (utils/loss.py:65-103): w1 mean|e - t| + w2 mean (e - t)^2 + w3 mean_j (1 - cos_j), where the cosine runs
// over the reference's RESHAPED sign tensors: column j of the [P, Nv] reinterpretation of the flat [Nv, P] array, i.e. the
// flat elements i Nv + j, i < P.  Thread j owns exactly those elements, so it also carries their share of the first two
// sums; per-block fp64 partial sums in a fixed tree, added in block order by the finishing block: bit-reproducible.
constexpr int kLossThreads = 256;
struct LossArgs {
    const float *est, *tgt;   // flat [Nv * P]
    long long nv;
    int P;
    float w1, w2, w3;
    double *partial;          // [blocks][4]
    float *loss;
    const float *grad_out;    // backward: d loss (device scalar)
    float *d_est;             // backward: [Nv * P]
    int blocks;
};

__device__ __forceinline__ float loss_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(kLossThreads) void train_loss_partial_kernel(const LossArgs a)
{
    __shared__ double red[kLossThreads / 64][3];
    const long long j = (long long)blockIdx.x * kLossThreads + threadIdx.x;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (j < a.nv) {
        float dot = 0.0f, n1 = 0.0f, n2 = 0.0f;
        for (int i = 0; i < a.P; ++i) {
            const size_t f = (size_t)i * a.nv + j;
            const float e = a.est[f], t = a.tgt[f], d = e - t;
            s1 += (double)fabsf(d);
            s2 += (double)(d * d);
            const float se = loss_sign(e), st = loss_sign(t);
            dot += se * st; n1 += se * se; n2 += st * st;
        }
        s3 = (double)(1.0f - dot / sqrtf((n1 + 1e-12f) * (n2 + 1e-12f)));
    }
    double v[3] = {s1, s2, s3};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) a.partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void train_loss_finish_kernel(const LossArgs a)
{
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < a.blocks; b += 64)  // lane l adds blocks l, l + 64, ... in order
        for (int q = 0; q < 3; ++q) s[q] += a.partial[(size_t)b * 4 + q];
    for (int q = 0; q < 3; ++q)
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_xor(s[q], off, 64);
    if (threadIdx.x == 0) {
        const double n = (double)a.nv * (double)a.P;
        *a.loss = (float)((double)a.w1 * s[0] / n + (double)a.w2 * s[1] / n + (double)a.w3 * s[2] / (double)a.nv);
    }
}

__global__ __launch_bounds__(256) void train_loss_bwd_kernel(const LossArgs a)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = a.nv * a.P;
    if (i >= total) return;
    const float g = *a.grad_out / (float)total;
    const float d = a.est[i] - a.tgt[i];
    a.d_est[i] = g * (a.w1 * loss_sign(d) + 2.0f * a.w2 * d);  // the sign-cosine term is piecewise constant
}

// ---- split-fp16 weights for the FORWARD convolutions, packed on the device -------------------------------------------
// The forward pass's convolutions see BatchNorm-scaled activations of order one: the inference path's split-fp16 arithmetic
// (three fp16 MFMAs per product block, fp32 accumulate, row-equilibrated weights: DESIGN.md 3.2) applies unchanged and runs
// the MFMA part 5.3x faster than the fp32-input instruction.  Gradients (tiny, of unbounded range) stay on fp32 MFMAs.
// Layout = finish()'s: [superstep][oc tile][hi | lo][lane] x 8 halfs, K entry G = 8 S + 2 (lane / 16) + j / 4.
struct Pack16Args {
    const float *w;        // [OC][ic_total][taps]
    float *rs;             // [n_ot * 16] power-of-two row scales (written by the row-max pass)
    float *rinv;           // [n_ot * 16] their inverses
    _Float16 *wp;          // packed halves
    int OC, IC, taps, group, slot, c4, nsteps, n_ot, oc_base, ic_base, ic_total;
};

__global__ __launch_bounds__(256) void train_pack16_rowscale_kernel(const Pack16Args a)
{
    __shared__ float red[4];
    const int oc = blockIdx.x;  // row of THIS weight tensor
    float mx = 0.0f;
    const int n = a.IC * a.taps;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ic = i / a.taps, t = i - ic * a.taps;
        mx = fmaxf(mx, fabsf(a.w[((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + t]));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float rs = 1.0f;
        if (mx > 0.0f && mx < 3.0e38f) {  // row_scale(): the largest magnitude of the row lands in [2^13, 2^14)
            int e = 0;
            (void)frexpf(mx, &e);
            int k = 14 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            rs = ldexpf(1.0f, k);
        }
        a.rs[a.oc_base + oc] = rs;
        a.rinv[a.oc_base + oc] = 1.0f / rs;
    }
}

__global__ __launch_bounds__(256) void train_pack16_kernel(const Pack16Args a)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.nsteps * a.n_ot * 512;  // (S, ot, lane, j)
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long rest = i >> 9;
    const int ot = (int)(rest % a.n_ot), S = (int)(rest / a.n_ot);
    const int row = ot * 16 + (lane & 15), G = 8 * S + 2 * (lane >> 4) + (j >> 2);
    const int oc = row - a.oc_base;
    if (oc < 0 || oc >= a.OC || G >= a.taps * a.c4) return;  // (the buffer starts zeroed; other tensors of a stack own the other rows)
    const int t = G / a.c4, ch = 4 * (G - t * a.c4) + (j & 3);
    const int ic = train_unslot(ch, a.group, a.slot, a.IC);
    float v = 0.0f;
    if (ic >= 0) v = a.rs[row] * a.w[((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + t];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t base = ((size_t)S * a.n_ot + ot) * 2 * 64 * 8;
    a.wp[base + (size_t)lane * 8 + j] = hi;
    a.wp[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
}

// Backward-data form of the same: rows = the convolution's INPUT channels (physical, chunk of <= 128 rows starting at
// row0), K entry (tap, 4 output channels) <- w[oc][ic][taps - 1 - tap]; a stack of up to four weight tensors shares the K
// axis (kper output channels each: the four branch entries of a VortexPooling).
struct Pack16TArgs {
    const float *w[4];     // [OC][ic_total][taps] each
    int n_src, kper;       // K channels [b * kper, b * kper + OC) belong to tensor b
    float *rs, *rinv;      // [rows of the whole layer, padded]
    _Float16 *wp;          // this chunk's packed halves
    int OC, IC, taps, group, slot, c4k, nsteps, n_ot, row0, rows, ic_base, ic_total;
};

__global__ __launch_bounds__(256) void train_pack16t_rowscale_kernel(const Pack16TArgs a)
{
    __shared__ float red[4];
    const int r = blockIdx.x;  // physical input channel
    const int ic = train_unslot(r, a.group, a.slot, a.IC);
    float mx = 0.0f;
    if (ic >= 0)
        for (int b = 0; b < a.n_src; ++b)
            for (int i = threadIdx.x; i < a.OC * a.taps; i += 256) {
                const int oc = i / a.taps, t = i - oc * a.taps;
                mx = fmaxf(mx, fabsf(a.w[b][((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + t]));
            }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float rs = 1.0f;
        if (mx > 0.0f && mx < 3.0e38f) {
            int e = 0;
            (void)frexpf(mx, &e);
            int k = 14 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            rs = ldexpf(1.0f, k);
        }
        a.rs[r] = rs;
        a.rinv[r] = 1.0f / rs;
    }
}

__global__ __launch_bounds__(256) void train_pack16t_kernel(const Pack16TArgs a)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.nsteps * a.n_ot * 512;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long rest = i >> 9;
    const int ot = (int)(rest % a.n_ot), S = (int)(rest / a.n_ot);
    const int row = a.row0 + ot * 16 + (lane & 15), G = 8 * S + 2 * (lane >> 4) + (j >> 2);
    float v = 0.0f;
    if (row < a.rows && G < a.taps * a.c4k) {
        const int tap = G / a.c4k, ch = 4 * (G - tap * a.c4k) + (j & 3);
        const int b = ch / a.kper, oc = ch - b * a.kper;
        const int ic = train_unslot(row, a.group, a.slot, a.IC);
        if (b < a.n_src && oc < a.OC && ic >= 0)
            v = a.rs[row] * a.w[b][((size_t)oc * a.ic_total + a.ic_base + ic) * a.taps + (a.taps - 1 - tap)];
    }
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t base = ((size_t)S * a.n_ot + ot) * 2 * 64 * 8;
    a.wp[base + (size_t)lane * 8 + j] = hi;
    a.wp[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
}

// ---- plan ----------------------------------------------------------------------------------------------------------
struct TUnit {           // conv -> [BatchNorm2d] -> activation -> [Dropout2d], one entry of the caller's layer table
    int li, OC, IC, k, dil, group, slot, act, has_bn;
    int ic_base, IC_total;   // the unit reads input channels [ic_base, ic_base + IC) of a weight tensor with IC_total of them
    float *sum_dy;           // backward, optional: per-channel sums of dy
    float scale;
    int c4_in, c4_out;
    float *in; int in_g0;    // input planes (window of a buffer)
    float *din;              // gradient buffer of the input buffer (same window), NULL: not needed
    float *y, *dy;           // convolution output / its gradient, [c4_out] groups (y_g0 inside a shared buffer for the stacked entry)
    int y_g0;
    float *out; int out_g0;  // unit output (window of a buffer)
    float *dout;             // gradient buffer of the output buffer (same window)
    float *wp, *bp, *wpT;    // packed weights: forward, bias, transposed + flipped (backward-data)
    float *wp16, *rs16, *rinv16;  // forward weights in the split-fp16 layout, row scales and their inverses
    int nsteps16;
    float *wpT16, *rsT16, *rinvT16;  // backward-data weights in the split-fp16 layout: chunks of <= 8 row tiles, one after the other
    int nsteps16T;
    int sg;                  // scale group of this unit's dy (index into the trainer's factors)
    unsigned *bnd;           // [2][4 c4_out]: this pass's bounds behind dy's factor (BnActArgs::bnd), zeroed behind every pass
    int n_ot, nsteps, n_otT, nstepsT;
    float *mean, *invstd;
    double *partial;
    float *wpart;
    WgradPlan wplan;
};

struct TVortex {
    int li0;                  // layer index of the global-average conv; branches follow (4 each), then the final conv
    float *x; int c4x, IC, group, slot; float *dx;  // input buffer (whole), its gradient
    int o4, out_c, sl4, mid;
    float *u, *du;            // stacked entry convolution output [4 sl4] (before the pools) / its gradient
    float *ycat, *dycat;      // pooled + bias: the four entries' y as slices of one buffer
    float *cat, *dcat;        // [5 o4]: gave | branch 0..3
    int entry[4], c1[4], c2[4], close[4], final_u;  // unit indices
    float *wp_stack, *wpT_stack, *wpart_stack;      // stacked entry weights [4 sl4 * 4 rows][K = 4 c4x]
    float *wp16_stack, *rs16_stack, *rinv16_stack;
    int nsteps16_s;
    float *wpT16_stack, *rsT16_stack, *rinvT16_stack;
    int nsteps16T_s;
    int n_ot_s, nsteps_s, n_otT_s, nstepsT_s;
    WgradPlan wplan_s;
    double *gpartial;         // channel sums (forward: of x; backward: of d cat[gave])
    float *pooled, *xhat, *gis, *vec, *dpooled, *sum_dy;
};

}  // namespace ojf

struct ojf_trainer {
    int version, P, gf, sem, h, w, npix, c, sl4, out_c, o4, n_layers;
    float scale;
    std::vector<void *> allocs;
    std::vector<ojf::TUnit> units;
    std::vector<ojf::TVortex> vortex;
    std::vector<std::vector<int>> dense;  // per head: unit indices a0, b0, a1, b1, ...
    std::vector<float *> dbuf, ddbuf;     // per head: dense concatenation buffer / its gradient
    std::vector<int> pred;                // unit indices
    float *zero_bias = nullptr;
    float *est_planes = nullptr, *dest_planes = nullptr;
    std::map<const float *, std::vector<char>> written;  // gradient buffers: which channel groups this backward pass has stored
    std::map<const float *, int> groups_of;
    unsigned long long epoch = ~0ull;
    bool have_forward = false;
    int launches = 0;
    // weight gradients run on a second stream beside the dy -> dx chain (nothing of the pass depends on them)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_ev;
    hipEvent_t join_ev = nullptr;
    size_t next_fork = 0;
    bool use_side = true;
    int fwd_arith = OJF_ARITH_F16X3;  // arithmetic of the forward AND backward-data convolutions (ojf_trainer_set_arithmetic)
    // power-of-two factors the dy tensors are stored with (one per unit; the four entries of a VortexPooling share one),
    // written by every backward pass's own BatchNorm-backward launches before their consumers read them
    float *scales = nullptr;
    int n_sg = 0;
    unsigned *bnd_pool = nullptr;  // all units' bound words in one allocation (one memset behind a pass)
    size_t bnd_words = 0;
    int bwd_arith = OJF_ARITH_F16X3;  // OJF_ARITH_F32: backward-data and weight gradients stay on the fp32-input MFMA path
    // Optional replay (ojf_trainer_set_graph): the launches of a pass captured once per (layer table, tensor addresses,
    // arithmetic) on a stream of the trainer's own and replayed with ONE hipGraphLaunch on the caller's stream.  A pass
    // whose key was not seen on the previous miss runs as plain launches (tensors that move every frame never capture).
    struct Replay { unsigned long long key; hipGraphExec_t exec; int launches; unsigned long long stamp; };
    bool graph_on = false;
    hipStream_t cap = nullptr;
    std::vector<Replay> replays[2];          // [0] forward, [1] backward
    unsigned long long missed[2] = {0, 0};   // key of the last pass that ran as plain launches
    unsigned long long graph_clock = 0;
    int graph_replays = 0, graph_captures = 0;
};

namespace ojf {

static int t_alloc(ojf_trainer *t, void **p, size_t bytes, bool zero = false)
{
    OJF_HIP(hipMalloc(p, bytes ? bytes : 16));
    t->allocs.push_back(*p);
    if (zero) {
        OJF_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
        OJF_HIP(hipStreamSynchronize(nullptr));  // (the trainer's streams may be non-blocking: ojf_net.hip alloc_planes)
    }
    return 0;
}

static int t_planes(ojf_trainer *t, float **p, int groups, bool grad = false)
{
    if (t_alloc(t, reinterpret_cast<void **>(p), (size_t)groups * t->npix * 16, true)) return -2;
    if (grad) t->groups_of[*p] = groups;
    return 0;
}

// physical channels of `n` logical channels laid out as `group`-wide tensors in `slot`-wide slots
static int slotted_phys(int n, int group, int slot) { return (n + group - 1) / group * slot; }

static int t_add_unit(ojf_trainer *t, int li, int OC, int IC, int k, int dil, int group, int slot, int act, int has_bn, float scale,
                      float *in, int in_g0, int c4_in, float *din, float *out, int out_g0, float *dout, bool own_y = true)
{
    TUnit u{};
    u.ic_base = 0; u.IC_total = IC; u.sum_dy = nullptr;
    u.li = li; u.OC = OC; u.IC = IC; u.k = k; u.dil = dil; u.group = group; u.slot = slot; u.act = act; u.has_bn = has_bn; u.scale = scale;
    u.c4_in = c4_in; u.c4_out = (OC + 3) / 4;
    u.in = in; u.in_g0 = in_g0; u.din = din; u.out = out; u.out_g0 = out_g0; u.dout = dout; u.y_g0 = 0;
    const int cop = u.c4_out * 4, cip = c4_in * 4, taps = k * k;
    if (ojf_train_packed_floats(cop, cip, k) == 0 || ojf_train_packed_floats(cip, cop, k) == 0) return fail("ojf_trainer: unsupported layer shape");
    u.n_ot = round_up(round_up(cop, 16) / 16, kNT); u.nsteps = (taps * c4_in + 3) / 4;
    u.n_otT = round_up(round_up(cip, 16) / 16, kNT); u.nstepsT = (taps * u.c4_out + 3) / 4;
    if (u.n_ot > 8) return fail("ojf_trainer: more than 128 output channels in one unit");
    if (own_y) {
        if (t_planes(t, &u.y, u.c4_out) || t_planes(t, &u.dy, u.c4_out)) return -2;
        if (t_alloc(t, reinterpret_cast<void **>(&u.wp), ojf_train_packed_floats(cop, cip, k) * 4, true)) return -2;
        if (t_alloc(t, reinterpret_cast<void **>(&u.wpT), ojf_train_packed_floats(cip, cop, k) * 4, true)) return -2;
        u.wplan = wgrad_plan(cop, cip, taps, t->npix);
        u.nsteps16T = (taps * u.c4_out + 7) / 8;
        if (din && u.nsteps16T <= kMaxSteps) {
            if (t_alloc(t, reinterpret_cast<void **>(&u.wpT16), (size_t)(u.nsteps16T + kPad16) * u.n_otT * 128 * 16, true)) return -2;
            if (t_alloc(t, reinterpret_cast<void **>(&u.rsT16), (size_t)u.n_otT * 16 * 4, true) || t_alloc(t, reinterpret_cast<void **>(&u.rinvT16), (size_t)u.n_otT * 16 * 4, true)) return -2;
        }
        u.nsteps16 = (taps * c4_in + 7) / 8;
        if (u.nsteps16 <= kMaxSteps) {
            if (t_alloc(t, reinterpret_cast<void **>(&u.wp16), (size_t)(u.nsteps16 + kPad16) * u.n_ot * 128 * 16, true)) return -2;
            if (t_alloc(t, reinterpret_cast<void **>(&u.rs16), (size_t)u.n_ot * 16 * 4, true) || t_alloc(t, reinterpret_cast<void **>(&u.rinv16), (size_t)u.n_ot * 16 * 4, true)) return -2;
        }
        if (t_alloc(t, reinterpret_cast<void **>(&u.wpart), (size_t)u.wplan.slabs * taps * u.wplan.ocp * u.wplan.icp * 4)) return -2;
    }
    if (t_alloc(t, reinterpret_cast<void **>(&u.bp), (size_t)u.n_ot * 16 * 4, true)) return -2;
    if (t_alloc(t, reinterpret_cast<void **>(&u.mean), (size_t)cop * 4, true) || t_alloc(t, reinterpret_cast<void **>(&u.invstd), (size_t)cop * 4, true)) return -2;
    if (t_alloc(t, reinterpret_cast<void **>(&u.partial), ojf_train_partial_doubles(cop) * 8, true)) return -2;
    u.sg = t->n_sg++;
    t->units.push_back(u);
    return (int)t->units.size() - 1;
}

// dense head: the concatenation buffer D holds gf + 1 slots; block i reads slots [0, i], writes slot i + 1
static int t_build_dense(ojf_trainer *t, int li0)
{
    const int sl4 = t->sl4, c = t->c, slot = 4 * sl4;
    float *D, *dD;
    if (t_planes(t, &D, (t->gf + 1) * sl4) || t_planes(t, &dD, (t->gf + 1) * sl4, true)) return -2;
    t->dbuf.push_back(D); t->ddbuf.push_back(dD);
    std::vector<int> ids;
    for (int i = 0; i < t->gf; ++i) {
        float *T, *dT;
        if (t_planes(t, &T, sl4) || t_planes(t, &dT, sl4, true)) return -2;
        // the first block's input is the net input: no gradient needed there
        const int a = t_add_unit(t, li0 + 2 * i, c, (i + 1) * c, 3, 1, c, slot, OJF_ACT_LEAKY, 1, 1.0f, D, 0, (i + 1) * sl4, i ? dD : nullptr, T, 0, dT);
        if (a < 0) return -2;
        const int b = t_add_unit(t, li0 + 2 * i + 1, c, c, 3, 1, c, slot, OJF_ACT_LEAKY, 1, 1.0f, T, 0, sl4, dT, D, (i + 1) * sl4, dD);
        if (b < 0) return -2;
        ids.push_back(a); ids.push_back(b);
    }
    t->dense.push_back(ids);
    return 0;
}

static const int kRates[4] = {1, 3, 9, 27};

static int t_build_vortex(ojf_trainer *t, int li0, float *x, int c4x, int IC, int group, int slot, float *dx, float *out, int out_g0, float *dout)
{
    TVortex v{};
    v.li0 = li0; v.x = x; v.c4x = c4x; v.IC = IC; v.group = group; v.slot = slot; v.dx = dx;
    v.out_c = t->out_c; v.o4 = t->o4; v.sl4 = t->sl4; v.mid = t->c;
    const int sl4 = v.sl4, mid = v.mid, o4 = v.o4;
    if (t_planes(t, &v.u, 4 * sl4) || t_planes(t, &v.du, 4 * sl4) || t_planes(t, &v.ycat, 4 * sl4) || t_planes(t, &v.dycat, 4 * sl4)) return -2;
    if (t_planes(t, &v.cat, 5 * o4) || t_planes(t, &v.dcat, 5 * o4, true)) return -2;
    float *E, *dE, *F, *dF, *G, *dG;
    if (t_planes(t, &E, 4 * sl4) || t_planes(t, &dE, 4 * sl4, true) || t_planes(t, &F, 4 * sl4) || t_planes(t, &dF, 4 * sl4, true) ||
        t_planes(t, &G, 4 * sl4) || t_planes(t, &dG, 4 * sl4, true))
        return -2;
    for (int r = 0; r < 4; ++r) {
        const int l = li0 + 1 + 4 * r;
        // entry: y lives in the shared pooled buffer (slice r), no packed weights / wgrad scratch of its own (stacked)
        v.entry[r] = t_add_unit(t, l, mid, IC, 1, 1, group, slot, OJF_ACT_RELU, 1, 1.0f, x, 0, c4x, dx, E, r * sl4, dE, false);
        if (v.entry[r] < 0) return -2;
        TUnit &e = t->units[v.entry[r]];
        e.y = v.ycat; e.dy = v.dycat; e.y_g0 = r * sl4;
        e.sg = t->units[v.entry[0]].sg;  // one factor for the four slices: the pooled gradients feed ONE stacked convolution
        v.c1[r] = t_add_unit(t, l + 1, mid, mid, 3, kRates[r], mid, 4 * sl4, OJF_ACT_RELU, 1, 1.0f, E, r * sl4, sl4, dE, F, r * sl4, dF);
        v.c2[r] = t_add_unit(t, l + 2, mid, mid, 3, kRates[r], mid, 4 * sl4, OJF_ACT_RELU, 1, 1.0f, F, r * sl4, sl4, dF, G, r * sl4, dG);
        v.close[r] = t_add_unit(t, l + 3, v.out_c, mid, 1, 1, mid, 4 * sl4, OJF_ACT_RELU, 1, 1.0f, G, r * sl4, sl4, dG, v.cat, (1 + r) * o4, v.dcat);
        if (v.c1[r] < 0 || v.c2[r] < 0 || v.close[r] < 0) return -2;
        for (int id : {v.c1[r], v.c2[r], v.close[r]}) {  // launched four at a time: a quarter of the pixel slabs fills the chip
            TUnit &u = t->units[id];
            u.wplan = wgrad_plan(4 * u.c4_out, 4 * u.c4_in, u.k * u.k, t->npix, 4);
        }
    }
    // final 1x1 over the concatenation [global-average map | branch 0..3]: the constant map's share is a per-frame bias
    // (train_gave_fwd_kernel), the convolution proper reads the four branch slots
    v.final_u = t_add_unit(t, li0 + 17, v.out_c, 4 * v.out_c, 1, 1, v.out_c, 4 * o4, OJF_ACT_NONE, 1, 1.0f, v.cat, o4, 4 * o4, v.dcat, out, out_g0, dout);
    if (v.final_u < 0) return -2;
    if (t_alloc(t, reinterpret_cast<void **>(&v.sum_dy), (size_t)o4 * 16, true)) return -2;
    {
        TUnit &f = t->units[v.final_u];
        f.ic_base = v.out_c; f.IC_total = 5 * v.out_c; f.sum_dy = v.sum_dy;
    }
    // stacked entry: rows = 4 slots of 4 sl4 channels, K = the input's physical channels
    const int rows = 16 * sl4, kch = 4 * c4x;
    if (ojf_train_packed_floats(rows, kch, 1) == 0 || ojf_train_packed_floats(kch, rows, 1) == 0) return fail("ojf_trainer: unsupported VortexPooling width");
    v.n_ot_s = round_up(round_up(rows, 16) / 16, kNT); v.nsteps_s = (c4x + 3) / 4;
    v.n_otT_s = round_up(round_up(kch, 16) / 16, kNT); v.nstepsT_s = (4 * sl4 + 3) / 4;
    if (v.n_ot_s > 8) return fail("ojf_trainer: stacked entry wider than 128 channels");
    if (t_alloc(t, reinterpret_cast<void **>(&v.wp_stack), ojf_train_packed_floats(rows, kch, 1) * 4, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.wpT_stack), ojf_train_packed_floats(kch, rows, 1) * 4, true))
        return -2;
    v.wplan_s = wgrad_plan(rows, kch, 1, t->npix);
    v.nsteps16T_s = (4 * sl4 + 7) / 8;
    if (t_alloc(t, reinterpret_cast<void **>(&v.wpT16_stack), (size_t)(v.nsteps16T_s + kPad16) * v.n_otT_s * 128 * 16, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.rsT16_stack), (size_t)v.n_otT_s * 16 * 4, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.rinvT16_stack), (size_t)v.n_otT_s * 16 * 4, true))
        return -2;
    v.nsteps16_s = (c4x + 7) / 8;
    if (t_alloc(t, reinterpret_cast<void **>(&v.wp16_stack), (size_t)(v.nsteps16_s + kPad16) * v.n_ot_s * 128 * 16, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.rs16_stack), (size_t)v.n_ot_s * 16 * 4, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.rinv16_stack), (size_t)v.n_ot_s * 16 * 4, true))
        return -2;
    if (t_alloc(t, reinterpret_cast<void **>(&v.wpart_stack), (size_t)v.wplan_s.slabs * v.wplan_s.ocp * v.wplan_s.icp * 4)) return -2;
    const int cmax = c4x > o4 ? c4x : o4;
    if (t_alloc(t, reinterpret_cast<void **>(&v.gpartial), ojf_train_partial_doubles(4 * cmax) * 8, true)) return -2;
    if (t_alloc(t, reinterpret_cast<void **>(&v.pooled), (size_t)c4x * 16, true) || t_alloc(t, reinterpret_cast<void **>(&v.dpooled), (size_t)c4x * 16, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.xhat), (size_t)o4 * 16, true) || t_alloc(t, reinterpret_cast<void **>(&v.gis), (size_t)o4 * 16, true) ||
        t_alloc(t, reinterpret_cast<void **>(&v.vec), (size_t)o4 * 16, true))
        return -2;
    t->vortex.push_back(v);
    return 0;
}

// ---- launches --------------------------------------------------------------------------------------------------------
struct TCtx {
    ojf_trainer *t;
    const ojf_train_layer *L;
    hipStream_t st;
};

// stream of the weight-gradient launches of one unit group: the side stream (after a fork from the main stream, whose
// dy is what they read) or the main stream itself
static int t_wgrad_stream(TCtx &c, hipStream_t *ws)
{
    ojf_trainer *t = c.t;
    *ws = c.st;
    if (!t->use_side || !t->side) return 0;
    hipEvent_t e = t->fork_ev[t->next_fork++ % t->fork_ev.size()];
    OJF_HIP(hipEventRecord(e, c.st));
    OJF_HIP(hipStreamWaitEvent(t->side, e, 0));
    *ws = t->side;
    return 0;
}

static inline dim3 px_grid(int npix, int gy) { return dim3((unsigned)((npix + 255) / 256), (unsigned)gy); }

static int t_check_unit(const TUnit &u, const ojf_train_layer &l)
{
    if (!l.weight || l.out_channels != u.OC || l.in_channels != u.IC_total || l.ksize != u.k || l.dilation != u.dil)
        return fail("ojf_trainer: layer table does not match the net topology (layer " + std::to_string(u.li) + ")");
    if (u.has_bn && (!l.running_mean || !l.running_var)) return fail("ojf_trainer: BatchNorm layer without running statistics");
    return 0;
}

static int t_pack_weights(TCtx &c)
{
    ojf_trainer *t = c.t;
    for (TUnit &u : t->units) {
        const ojf_train_layer &l = c.L[u.li];
        if (t_check_unit(u, l)) return -2;
        if (!u.wp) continue;  // stacked entry units: packed with their VortexPooling below
        for (int tr = 0; tr < (u.din ? 2 : 1); ++tr) {
            PackArgs a;
            const int rows = tr ? 4 * u.c4_in : 4 * u.c4_out, kch = tr ? 4 * u.c4_out : 4 * u.c4_in;
            a.w = l.weight; a.bias = tr ? nullptr : l.bias; a.wp = tr ? u.wpT : u.wp; a.bp = tr ? nullptr : u.bp;
            a.OC = u.OC; a.IC = u.IC; a.taps = u.k * u.k; a.group = u.group; a.slot = u.slot;
            a.c4 = kch / 4; a.nsteps = (a.taps * a.c4 + 3) / 4; a.n_ot = round_up(round_up(rows, 16) / 16, kNT); a.transposed = tr;
            a.oc_base = 0; a.partial = 0; a.ic_base = u.ic_base; a.ic_total = u.IC_total;
            const long total = (long)a.n_ot * (a.nsteps + kPadSteps) * 256;
            hipLaunchKernelGGL(train_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.st, a);
            ++t->launches;
        }
        if (t->fwd_arith == OJF_ARITH_F16X3 && u.wp16) {
            Pack16Args q;
            q.w = l.weight; q.rs = u.rs16; q.rinv = u.rinv16; q.wp = reinterpret_cast<_Float16 *>(u.wp16);
            q.OC = u.OC; q.IC = u.IC; q.taps = u.k * u.k; q.group = u.group; q.slot = u.slot; q.c4 = u.c4_in; q.nsteps = u.nsteps16; q.n_ot = u.n_ot;
            q.oc_base = 0; q.ic_base = u.ic_base; q.ic_total = u.IC_total;
            hipLaunchKernelGGL(train_pack16_rowscale_kernel, dim3(u.OC), dim3(256), 0, c.st, q);
            const long tot16 = (long)q.nsteps * q.n_ot * 512;
            hipLaunchKernelGGL(train_pack16_kernel, dim3((unsigned)((tot16 + 255) / 256)), dim3(256), 0, c.st, q);
            t->launches += 2;
        }
        if (t->fwd_arith == OJF_ARITH_F16X3 && u.wpT16) {
            Pack16TArgs q{};
            q.w[0] = l.weight; q.n_src = 1; q.kper = 4 * u.c4_out; q.rs = u.rsT16; q.rinv = u.rinvT16;
            q.OC = u.OC; q.IC = u.IC; q.taps = u.k * u.k; q.group = u.group; q.slot = u.slot; q.c4k = u.c4_out; q.nsteps = u.nsteps16T;
            q.rows = 4 * u.c4_in; q.ic_base = u.ic_base; q.ic_total = u.IC_total;
            hipLaunchKernelGGL(train_pack16t_rowscale_kernel, dim3(4 * u.c4_in), dim3(256), 0, c.st, q);
            ++t->launches;
            for (int ot0 = 0; ot0 < u.n_otT; ot0 += 8) {
                q.n_ot = u.n_otT - ot0 < 8 ? u.n_otT - ot0 : 8;
                q.row0 = ot0 * 16;
                q.wp = reinterpret_cast<_Float16 *>(u.wpT16) + (size_t)(u.nsteps16T + kPad16) * ot0 * 128 * 8;
                const long tot16 = (long)q.nsteps * q.n_ot * 512;
                hipLaunchKernelGGL(train_pack16t_kernel, dim3((unsigned)((tot16 + 255) / 256)), dim3(256), 0, c.st, q);
                ++t->launches;
            }
        }
    }
    for (TVortex &v : t->vortex)
        for (int r = 0; r < 4; ++r) {
            const TUnit &u = t->units[v.entry[r]];
            const ojf_train_layer &l = c.L[u.li];
            for (int tr = 0; tr < 2; ++tr) {
                PackArgs a;
                const int rows = tr ? 4 * v.c4x : 16 * v.sl4, kch = tr ? 16 * v.sl4 : 4 * v.c4x;
                a.w = l.weight; a.bias = nullptr; a.wp = tr ? v.wpT_stack : v.wp_stack; a.bp = nullptr;
                a.OC = u.OC; a.IC = u.IC; a.taps = 1; a.group = u.group; a.slot = u.slot;
                a.c4 = kch / 4; a.nsteps = (a.c4 + 3) / 4; a.n_ot = round_up(round_up(rows, 16) / 16, kNT); a.transposed = tr;
                a.oc_base = r * 4 * v.sl4; a.partial = 1; a.ic_base = 0; a.ic_total = u.IC;
                const long total = (long)a.n_ot * (a.nsteps + kPadSteps) * 256;
                hipLaunchKernelGGL(train_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.st, a);
                ++t->launches;
            }
            if (t->fwd_arith == OJF_ARITH_F16X3) {
                Pack16Args q;
                q.w = l.weight; q.rs = v.rs16_stack; q.rinv = v.rinv16_stack; q.wp = reinterpret_cast<_Float16 *>(v.wp16_stack);
                q.OC = u.OC; q.IC = u.IC; q.taps = 1; q.group = u.group; q.slot = u.slot; q.c4 = v.c4x; q.nsteps = v.nsteps16_s; q.n_ot = v.n_ot_s;
                q.oc_base = r * 4 * v.sl4; q.ic_base = 0; q.ic_total = u.IC;
                hipLaunchKernelGGL(train_pack16_rowscale_kernel, dim3(u.OC), dim3(256), 0, c.st, q);
                const long tot16 = (long)q.nsteps * q.n_ot * 512;
                hipLaunchKernelGGL(train_pack16_kernel, dim3((unsigned)((tot16 + 255) / 256)), dim3(256), 0, c.st, q);
                t->launches += 2;
            }
        }
    if (t->fwd_arith == OJF_ARITH_F16X3)
        for (TVortex &v : t->vortex) {
            const TUnit &u0 = t->units[v.entry[0]];
            Pack16TArgs q{};
            for (int r = 0; r < 4; ++r) q.w[r] = c.L[t->units[v.entry[r]].li].weight;
            q.n_src = 4; q.kper = 4 * v.sl4; q.rs = v.rsT16_stack; q.rinv = v.rinvT16_stack;
            q.OC = u0.OC; q.IC = u0.IC; q.taps = 1; q.group = u0.group; q.slot = u0.slot; q.c4k = 4 * v.sl4; q.nsteps = v.nsteps16T_s;
            q.rows = 4 * v.c4x; q.ic_base = 0; q.ic_total = u0.IC;
            hipLaunchKernelGGL(train_pack16t_rowscale_kernel, dim3(4 * v.c4x), dim3(256), 0, c.st, q);
            ++t->launches;
            for (int ot0 = 0; ot0 < v.n_otT_s; ot0 += 8) {
                q.n_ot = v.n_otT_s - ot0 < 8 ? v.n_otT_s - ot0 : 8;
                q.row0 = ot0 * 16;
                q.wp = reinterpret_cast<_Float16 *>(v.wpT16_stack) + (size_t)(v.nsteps16T_s + kPad16) * ot0 * 128 * 8;
                const long tot16 = (long)q.nsteps * q.n_ot * 512;
                hipLaunchKernelGGL(train_pack16t_kernel, dim3((unsigned)((tot16 + 255) / 256)), dim3(256), 0, c.st, q);
                ++t->launches;
            }
        }
    return check_hip(hipGetLastError(), "ojf_trainer weight packing");
}

// convolution launches of up to four units (forward), chunked over output tiles when there are more than eight
static int t_conv(TCtx &c, const ConvArgs *args, int n, int n_ot)
{
    for (int ot0 = 0; ot0 < n_ot; ot0 += 8) {
        const int nt = n_ot - ot0 < 8 ? n_ot - ot0 : 8;
        ConvArgs a[4];
        for (int i = 0; i < n; ++i) {
            a[i] = args[i];
            a[i].wp = args[i].wp + (size_t)ot0 * (args[i].nsteps + kPadSteps) * 64;
            a[i].bias = args[i].bias + (size_t)ot0 * 16;
            a[i].out_g0 = args[i].out_g0 + ot0 * 4;
            const int left = args[i].og_store - ot0 * 4;
            a[i].og_store = left < nt * 4 ? left : nt * 4;
        }
        if (launch_conv_args(a, n, nt, c.st, OJF_ARITH_F32)) return -2;
        ++c.t->launches;
    }
    return 0;
}

static ConvArgs t_conv_args(const ojf_trainer *t, const float *in, int in_g0, int c4_in, float *out, int out_g0, int c4_out, const float *wp,
                            const float *bias, int nsteps, int k, int dil, int accum)
{
    ConvArgs a;
    a.ovf = nullptr; a.accum = accum; a.dscale = nullptr;
    a.in = planes(in); a.out = planes(out); a.out_rows = nullptr;
    a.wp = planes(wp); a.bias = bias; a.rinv = nullptr;
    a.in_g0 = in_g0; a.out_g0 = out_g0; a.rows_stride = 0; a.rows_n = 0;
    a.h = t->h; a.w = t->w; a.npix = t->npix; a.taps = k * k; a.dil = dil; a.c4 = c4_in; a.nsteps = nsteps;
    a.w_magic = 0; a.c4_magic = 0;
    a.og_store = c4_out;
    a.act = OJF_ACT_NONE; a.act_n = 0; a.scale = 1.0f;
    return a;
}

// the same launch in the split-fp16 arithmetic (forward convolutions): packed halves, inverse row scales, range guard
static ConvArgs t_conv_args16(const ojf_trainer *t, const float *in, int in_g0, int c4_in, float *out, int out_g0, int c4_out, const float *wp16,
                              const float *rinv, const float *bias, int nsteps16, int k, int dil)
{
    ConvArgs a = t_conv_args(t, in, in_g0, c4_in, out, out_g0, c4_out, wp16, bias, nsteps16, k, dil, 0);
    a.rinv = rinv;
    a.ovf = overflow_flag();
    a.w_magic = div_magic(t->w, (uint64_t)t->npix);
    a.c4_magic = div_magic(c4_in, (uint64_t)(nsteps16 + kPad16) * 8);
    return a;
}

// store / accumulate decision of a gradient write into groups [g0, g0 + n) of buffer `buf`
static int t_grad_mode(ojf_trainer *t, const float *buf, int g0, int n, int *accum)
{
    auto it = t->written.find(buf);
    if (it == t->written.end()) return fail("ojf_trainer: untracked gradient buffer");
    std::vector<char> &w = it->second;
    int set = 0;
    for (int g = g0; g < g0 + n; ++g) set += w[g];
    if (set != 0 && set != n) return fail("ojf_trainer: mixed store / accumulate gradient window");
    *accum = set ? 1 : 0;
    for (int g = g0; g < g0 + n; ++g) w[g] = 1;
    return 0;
}

// backward-data of up to four units (or of ONE wide one, in chunks of eight row tiles): d in (+)= conv(dy, W^T flipped) / s,
// where s is the power of two dy was stored with.  From the second pass on (the first one measures the gradients' magnitude)
// in the split-fp16 arithmetic, else on the fp32-input MFMA.
// split-fp16 backward (backward-data AND weight gradients): with the forward arithmetic, unless the caller asked for fp32
// gradients (ojf_trainer_set_backward_arithmetic)
static inline bool t_bwd_f16(const ojf_trainer *t) { return t->fwd_arith == OJF_ARITH_F16X3 && t->bwd_arith == OJF_ARITH_F16X3; }

struct TBwdData {
    const float *dy; int dy_g0, c4k;       // input: gradient planes of the unit's convolution output
    float *din; int in_g0, c4_in;          // output window
    const float *wpT; int nstepsT;         // fp32 form
    const float *wpT16, *rinvT16; int nsteps16T;
    int n_otT, k, dil, sg;
};

static int t_backward_data(TCtx &c, const TBwdData *b, int n)
{
    ojf_trainer *t = c.t;
    const bool f16 = t_bwd_f16(t) && b[0].wpT16;
    int accum[4];
    for (int i = 0; i < n; ++i)
        if (t_grad_mode(t, b[i].din, b[i].in_g0, b[i].c4_in, &accum[i])) return -2;
    if (!f16) {
        ConvArgs ca[4];
        for (int i = 0; i < n; ++i) {
            ca[i] = t_conv_args(t, b[i].dy, b[i].dy_g0, b[i].c4k, b[i].din, b[i].in_g0, b[i].c4_in, b[i].wpT, t->zero_bias, b[i].nstepsT, b[i].k, b[i].dil, accum[i]);
            ca[i].dscale = t->scales + b[i].sg;
        }
        return t_conv(c, ca, n, b[0].n_otT);
    }
    if (b[0].n_otT > 8 && n != 1) return fail("ojf_trainer: grouped backward-data wider than 128 channels");
    for (int ot0 = 0; ot0 < b[0].n_otT; ot0 += 8) {
        const int nt = b[0].n_otT - ot0 < 8 ? b[0].n_otT - ot0 : 8;
        ConvArgs ca[4];
        for (int i = 0; i < n; ++i) {
            const float *wp = b[i].wpT16 + (size_t)(b[i].nsteps16T + kPad16) * ot0 * 128 * 4;
            ca[i] = t_conv_args16(t, b[i].dy, b[i].dy_g0, b[i].c4k, b[i].din, b[i].in_g0 + ot0 * 4, 0, wp, b[i].rinvT16 + (size_t)ot0 * 16,
                                  t->zero_bias, b[i].nsteps16T, b[i].k, b[i].dil);
            const int left = b[i].c4_in - ot0 * 4;
            ca[i].og_store = left < nt * 4 ? left : nt * 4;
            ca[i].accum = accum[i];
            ca[i].dscale = t->scales + b[i].sg;
        }
        if (launch_conv_args(ca, n, nt, c.st, OJF_ARITH_F16X3)) return -2;
        ++t->launches;
    }
    return 0;
}

static BnActArgs t_bn_args(const ojf_trainer *t, const TUnit &u, const ojf_train_layer &l)
{
    BnActArgs a = train_bn_args(u.y, u.y_g0, 4 * u.c4_out, u.OC, t->h, t->w, u.mean, u.invstd, u.has_bn ? l.gamma : nullptr, u.has_bn ? l.beta : nullptr,
                                l.drop_scale, u.act, u.scale, u.has_bn, (u.has_bn && l.bn_training) ? 1 : 0);
    a.partial = u.partial;
    return a;
}

// forward of up to four units of identical shape: convolution (unless the caller already produced y), statistics, normalise + activate
static int t_units_forward(TCtx &c, const int *ids, int n, bool conv = true)
{
    ojf_trainer *t = c.t;
    if (conv) {
        ConvArgs ca[4];
        const bool f16 = t->fwd_arith == OJF_ARITH_F16X3 && t->units[ids[0]].wp16;
        for (int i = 0; i < n; ++i) {
            const TUnit &u = t->units[ids[i]];
            ca[i] = f16 ? t_conv_args16(t, u.in, u.in_g0, u.c4_in, u.y, u.y_g0, u.c4_out, u.wp16, u.rinv16, u.bp, u.nsteps16, u.k, u.dil)
                        : t_conv_args(t, u.in, u.in_g0, u.c4_in, u.y, u.y_g0, u.c4_out, u.wp, u.bp, u.nsteps, u.k, u.dil, 0);
        }
        if (f16) {
            if (launch_conv_args(ca, n, t->units[ids[0]].n_ot, c.st, OJF_ARITH_F16X3)) return -2;
            ++t->launches;
        } else if (t_conv(c, ca, n, t->units[ids[0]].n_ot)) return -2;
    }
    BnGroup grp;
    grp.share_scale = 0;
    bool any_stats = false;
    for (int i = 0; i < 4; ++i) {
        const TUnit &u = t->units[ids[i < n ? i : 0]];
        const ojf_train_layer &l = c.L[u.li];
        BnActArgs a = t_bn_args(t, u, l);
        a.mean = nullptr; a.invstd = nullptr;
        a.out = planes(u.out); a.out_g0 = u.out_g0;
        a.mean_out = u.mean; a.invstd_out = u.invstd; a.running_mean = l.running_mean; a.running_var = l.running_var;
        a.momentum = l.momentum; a.eps = l.eps;
        any_stats |= a.has_bn && a.training;
        grp.g[i] = a;
    }
    const int c4 = t->units[ids[0]].c4_out;
    if (any_stats) {
        hipLaunchKernelGGL(train_stats_group_kernel, dim3(kTrainSlabs, c4, n), dim3(256), 0, c.st, grp);
        ++t->launches;
    }
    const int bx = (t->npix + 255) / 256 < 128 ? (t->npix + 255) / 256 : 128;
    hipLaunchKernelGGL(train_bn_act_fwd_kernel, dim3(bx, c4, n), dim3(256), 0, c.st, grp);
    ++t->launches;
    return check_hip(hipGetLastError(), "ojf_trainer unit forward");
}

// backward of up to four units: BatchNorm / activation backward (dy, d gamma, d beta, d bias); then - unless `tail` is
// false (stacked entry: the caller pools dy first) - the weight gradient and backward-data
// weight gradients in split-fp16: with the forward arithmetic, from the pass on whose dy carries measured factors
static inline bool t_wgrad_f16(const ojf_trainer *t)
{
    static const bool off = getenv("OJF_TRAIN_WGRAD16") && atoi(getenv("OJF_TRAIN_WGRAD16")) == 0;  // A/B switch
    return !off && t_bwd_f16(t);
}

static int t_units_backward(TCtx &c, const int *ids, int n, bool tail = true)
{
    ojf_trainer *t = c.t;
    BnGroup grp;
    for (int i = 0; i < 4; ++i) {
        const TUnit &u = t->units[ids[i < n ? i : 0]];
        const ojf_train_layer &l = c.L[u.li];
        BnActArgs a = t_bn_args(t, u, l);
        a.dout = planes(u.dout); a.dout_g0 = u.out_g0; a.dy = planes(u.dy); a.dy_g0 = u.y_g0;
        a.dgamma = u.has_bn ? l.grad_gamma : nullptr; a.dbeta = u.has_bn ? l.grad_beta : nullptr; a.dbias = l.bias ? l.grad_bias : nullptr;
        a.accumulate = l.accumulate ? 1 : 0;
        a.sum_dy = u.sum_dy;
        a.bnd = u.bnd; a.bnd_stride = 4 * u.c4_out; a.dy_scale_out = t->scales + u.sg;
        grp.g[i] = a;
    }
    grp.share_scale = tail ? 0 : 1;  // the stacked branch entries: one factor for the four dy tensors
    const int c4 = t->units[ids[0]].c4_out;
    hipLaunchKernelGGL(train_bn_bwd_reduce_kernel, dim3(kTrainSlabs, c4, n), dim3(256), 0, c.st, grp);
    const int bx = (t->npix + 255) / 256 < 128 ? (t->npix + 255) / 256 : 128;
    hipLaunchKernelGGL(train_bn_bwd_apply_kernel, dim3(bx, c4, n), dim3(256), 0, c.st, grp);
    t->launches += 2;
    OJF_HIP(hipGetLastError());
    if (!tail) return 0;
    // weight gradients
    {
        const TUnit &u0 = t->units[ids[0]];
        const int taps = u0.k * u0.k, tiles = (u0.wplan.ocp / 32) * (u0.wplan.icp / 32);
        WgradGroup wg;
        WgradReduceGroup rg;
        long total = 0;
        for (int i = 0; i < 4; ++i) {
            const TUnit &u = t->units[ids[i < n ? i : 0]];
            const ojf_train_layer &l = c.L[u.li];
            WgradArgs a;
            a.x = planes(u.in); a.dy = planes(u.dy); a.partial = u.wpart; a.x_g0 = u.in_g0; a.c4_in = u.c4_in; a.dy_g0 = u.y_g0; a.c4_out = u.c4_out;
            a.h = t->h; a.w = t->w; a.npix = t->npix; a.taps = taps; a.dil = u.dil; a.slabs = u.wplan.slabs; a.ocp = u.wplan.ocp; a.icp = u.wplan.icp;
            a.ovf = nullptr;  // (dy is inside the fp16 range by construction: BnActArgs::bnd; x was guarded by the forward pass)
            wg.g[i] = a;
            WgradReduceArgs r;
            r.partial = u.wpart; r.dw = l.grad_weight; r.slabs = u.wplan.slabs; r.taps = taps; r.ocp = u.wplan.ocp; r.icp = u.wplan.icp;
            r.OC = u.OC; r.IC = u.IC; r.group = u.group; r.slot = u.slot; r.c_in_phys = 4 * u.c4_in; r.accumulate = l.accumulate ? 1 : 0; r.oc_base = 0;
            r.ic_base = u.ic_base; r.ic_total = u.IC_total; r.dy_scale = t->scales + u.sg;
            rg.g[i] = r;
            total = (long)taps * u.OC * 4 * u.c4_in * 8;
            if (!l.grad_weight) return fail("ojf_trainer_backward: layer without a weight-gradient tensor");
        }
        wg.tiles = tiles;
        hipStream_t ws;
        if (t_wgrad_stream(c, &ws)) return -2;
        if (t_wgrad_f16(t))
            hipLaunchKernelGGL(train_wgrad_mfma_kernel<true>, dim3(u0.wplan.slabs, tiles * n, taps), dim3(64), 0, ws, wg,
                               div_magic(t->w, (uint64_t)t->npix + 2 * kWgChunk));
        else
            hipLaunchKernelGGL(train_wgrad_mfma_kernel<false>, dim3(u0.wplan.slabs, tiles * n, taps), dim3(64), 0, ws, wg,
                               div_magic(t->w, (uint64_t)t->npix + 2 * kWgChunk));
        hipLaunchKernelGGL(train_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256), n), dim3(256), 0, ws, rg);
        t->launches += 2;
        OJF_HIP(hipGetLastError());
    }
    // backward-data: d in (+)= conv(dy, transposed + flipped weights)
    if (t->units[ids[0]].din) {
        TBwdData bd[4];
        for (int i = 0; i < n; ++i) {
            const TUnit &u = t->units[ids[i]];
            bd[i] = TBwdData{u.dy, u.y_g0, u.c4_out, u.din, u.in_g0, u.c4_in, u.wpT, u.nstepsT, u.wpT16, u.rinvT16, u.nsteps16T, u.n_otT, u.k, u.dil, u.sg};
        }
        if (t_backward_data(c, bd, n)) return -2;
    }
    return 0;
}

static int t_vortex_forward(TCtx &c, TVortex &v)
{
    ojf_trainer *t = c.t;
    const ojf_train_layer &lg = c.L[v.li0];
    if (!lg.weight || lg.out_channels != v.out_c || lg.in_channels != v.IC || lg.ksize != 1 || !lg.running_mean || !lg.running_var)
        return fail("ojf_trainer: global-average layer does not match the topology");
    // global-average branch -> a bias of the final convolution (on the side stream it gained nothing: 191 vs 192 frames/s)
    hipLaunchKernelGGL(train_stats_partial_kernel, dim3(kTrainSlabs, v.c4x), dim3(256), 0, c.st, planes(v.x), 0, t->npix, v.gpartial);
    GaveTrainArgs g{};
    g.partial = v.gpartial; g.c4_in = v.c4x; g.c4_out = v.o4; g.IC = v.IC; g.OC = v.out_c; g.group = v.group; g.slot = v.slot; g.npix = t->npix;
    g.training = lg.bn_training ? 1 : 0; g.accumulate = 0; g.momentum = lg.momentum; g.eps = lg.eps;
    g.W = lg.weight; g.b = lg.bias; g.gamma = lg.gamma; g.beta = lg.beta; g.running_mean = lg.running_mean; g.running_var = lg.running_var;
    g.pooled = v.pooled; g.xhat = v.xhat; g.gis = v.gis; g.vec = v.vec;
    {
        const TUnit &f = t->units[v.final_u];
        const ojf_train_layer &lf = c.L[f.li];
        g.Wf = lf.weight; g.bf = lf.bias; g.OCf = f.OC; g.ICf_total = f.IC_total; g.bias_eff = f.bp;
    }
    if (v.out_c > kGaveMaxC || 4 * v.c4x > kGaveMaxC) return fail("ojf_trainer: VortexPooling wider than 1024 channels");
    hipLaunchKernelGGL(train_gave_fwd_kernel, dim3(1), dim3(kGaveThreads), 0, c.st, g);
    t->launches += 2;
    // the four branch entries: ONE stacked 1x1 convolution of the unpooled input, then the pools (+ bias) on the narrow result
    if (t->fwd_arith == OJF_ARITH_F16X3) {
        ConvArgs ca = t_conv_args16(t, v.x, 0, v.c4x, v.u, 0, 4 * v.sl4, v.wp16_stack, v.rinv16_stack, t->zero_bias, v.nsteps16_s, 1, 1);
        if (launch_conv_args(&ca, 1, v.n_ot_s, c.st, OJF_ARITH_F16X3)) return -2;
        ++t->launches;
    } else {
        ConvArgs ca = t_conv_args(t, v.x, 0, v.c4x, v.u, 0, 4 * v.sl4, v.wp_stack, t->zero_bias, v.nsteps_s, 1, 1, 0);
        if (t_conv(c, &ca, 1, v.n_ot_s)) return -2;
    }
    TrainPyramidArgs pa;
    pa.in = planes(v.u); pa.out = planes(v.ycat); pa.h = t->h; pa.w = t->w; pa.sl4 = v.sl4; pa.OC = v.mid;
    for (int r = 0; r < 4; ++r) pa.bias[r] = c.L[t->units[v.entry[r]].li].bias;
    const int tiles = ((t->w + kTpW - 1) / kTpW) * ((t->h + kTpH - 1) / kTpH);
    hipLaunchKernelGGL(train_pyramid_kernel, dim3(tiles, 4 * v.sl4), dim3(256), 0, c.st, pa);
    ++t->launches;
    OJF_HIP(hipGetLastError());
    if (t_units_forward(c, v.entry, 4, false)) return -2;
    if (t_units_forward(c, v.c1, 4) || t_units_forward(c, v.c2, 4) || t_units_forward(c, v.close, 4)) return -2;
    return t_units_forward(c, &v.final_u, 1);
}

static int t_vortex_backward(TCtx &c, TVortex &v)
{
    ojf_trainer *t = c.t;
    if (t_units_backward(c, &v.final_u, 1)) return -2;
    if (t_units_backward(c, v.close, 4) || t_units_backward(c, v.c2, 4) || t_units_backward(c, v.c1, 4)) return -2;
    // entries: BatchNorm backward per branch, pool the gradients, one stacked weight gradient + one backward-data
    if (t_units_backward(c, v.entry, 4, false)) return -2;
    TrainPyramidArgs pa;
    pa.in = planes(v.dycat); pa.out = planes(v.du); pa.h = t->h; pa.w = t->w; pa.sl4 = v.sl4; pa.OC = v.mid;
    for (int r = 0; r < 4; ++r) pa.bias[r] = nullptr;
    const int tiles = ((t->w + kTpW - 1) / kTpW) * ((t->h + kTpH - 1) / kTpH);
    {
        WgradArgs a;
        a.x = planes(v.x); a.dy = planes(v.du); a.partial = v.wpart_stack; a.x_g0 = 0; a.c4_in = v.c4x; a.dy_g0 = 0; a.c4_out = 4 * v.sl4;
        a.h = t->h; a.w = t->w; a.npix = t->npix; a.taps = 1; a.dil = 1; a.slabs = v.wplan_s.slabs; a.ocp = v.wplan_s.ocp; a.icp = v.wplan_s.icp;
        a.ovf = nullptr;
        const int wt = (a.ocp / 32) * (a.icp / 32);
        hipStream_t ws;
        hipLaunchKernelGGL(train_pyramid_kernel, dim3(tiles, 4 * v.sl4), dim3(256), 0, c.st, pa);
        if (t_wgrad_stream(c, &ws)) return -2;
        if (t_wgrad_f16(t))
            hipLaunchKernelGGL(train_wgrad_mfma_kernel<true>, dim3(a.slabs, wt, 1), dim3(64), 0, ws, WgradGroup{{a, a, a, a}, wt},
                               div_magic(t->w, (uint64_t)t->npix + 2 * kWgChunk));
        else
            hipLaunchKernelGGL(train_wgrad_mfma_kernel<false>, dim3(a.slabs, wt, 1), dim3(64), 0, ws, WgradGroup{{a, a, a, a}, wt},
                               div_magic(t->w, (uint64_t)t->npix + 2 * kWgChunk));
        WgradReduceGroup rg;
        long total = 0;
        for (int r = 0; r < 4; ++r) {
            const TUnit &u = t->units[v.entry[r]];
            const ojf_train_layer &l = c.L[u.li];
            if (!l.grad_weight) return fail("ojf_trainer_backward: layer without a weight-gradient tensor");
            WgradReduceArgs ra;
            ra.partial = v.wpart_stack; ra.dw = l.grad_weight; ra.slabs = a.slabs; ra.taps = 1; ra.ocp = a.ocp; ra.icp = a.icp;
            ra.OC = u.OC; ra.IC = u.IC; ra.group = u.group; ra.slot = u.slot; ra.c_in_phys = 4 * v.c4x; ra.accumulate = l.accumulate ? 1 : 0;
            ra.oc_base = r * 4 * v.sl4; ra.ic_base = 0; ra.ic_total = u.IC; ra.dy_scale = t->scales + u.sg;
            rg.g[r] = ra;
            total = (long)u.OC * 4 * v.c4x * 8;
        }
        hipLaunchKernelGGL(train_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256), 4), dim3(256), 0, ws, rg);
    }
    t->launches += 3;
    OJF_HIP(hipGetLastError());
    if (v.dx) {
        const TBwdData bd{v.du, 0, 4 * v.sl4, v.dx, 0, v.c4x, v.wpT_stack, v.nstepsT_s, v.wpT16_stack, v.rinvT16_stack, v.nsteps16T_s, v.n_otT_s, 1, 1,
                          t->units[v.entry[0]].sg};
        if (t_backward_data(c, &bd, 1)) return -2;
    }
    // global-average branch
    const ojf_train_layer &lg = c.L[v.li0];
    GaveTrainArgs g{};
    g.partial = v.gpartial; g.c4_in = v.c4x; g.c4_out = v.o4; g.IC = v.IC; g.OC = v.out_c; g.group = v.group; g.slot = v.slot; g.npix = t->npix;
    g.training = lg.bn_training ? 1 : 0; g.accumulate = lg.accumulate ? 1 : 0;
    g.W = lg.weight; g.b = lg.bias; g.gamma = lg.gamma; g.beta = lg.beta;
    g.pooled = v.pooled; g.xhat = v.xhat; g.gis = v.gis;
    g.dW = lg.grad_weight; g.db = lg.bias ? lg.grad_bias : nullptr; g.dgamma = lg.grad_gamma; g.dbeta = lg.grad_beta; g.dpooled = v.dpooled;
    {
        const TUnit &f = t->units[v.final_u];
        const ojf_train_layer &lf = c.L[f.li];
        g.Wf = lf.weight; g.OCf = f.OC; g.ICf_total = f.IC_total; g.sum_dy = v.sum_dy; g.vec = v.vec; g.dWf = lf.grad_weight;
        g.accumulate_f = lf.accumulate ? 1 : 0;
    }
    hipLaunchKernelGGL(train_gave_bwd_kernel, dim3(1), dim3(kGaveThreads), 0, c.st, g);
    ++t->launches;
    if (!lg.bn_training && v.dx) {  // running statistics: the branch is an affine map of mean(x) - a per-channel constant flows back to every pixel
        hipLaunchKernelGGL(train_bcast_planes_kernel, px_grid(t->npix, v.c4x), dim3(256), 0, c.st, v.dpooled, planes(v.dx), t->npix, 1);
        ++t->launches;
    }
    return check_hip(hipGetLastError(), "ojf_trainer VortexPooling backward");
}

// A pass = the launches that touch the caller's per-frame tensors (net input, est, d_est: plain launches, whatever the addresses)
// around the launches between the trainer's own buffers and the layer table's tensors (parameters, gradients, BatchNorm
// statistics, dropout factors: addresses that stay for the life of a training run - what ojf_trainer_set_graph replays).

// net input -> slot 0 of every dense head (modules/model.py:266-270 / :203-207)
static int t_forward_in(TCtx &c, const float *values, const float *weights, const float *frame, const float *semantic_frame)
{
    ojf_trainer *t = c.t;
    for (size_t hd = 0; hd < t->dbuf.size(); ++hd) {
        PackInArgs a{};
        a.src[0] = values; a.nch[0] = t->P; a.src[1] = weights; a.nch[1] = t->P;
        a.n_src = 3;
        if (t->version == 3) { a.src[2] = hd == 0 ? frame : semantic_frame; a.nch[2] = 1; }
        else {
            a.src[2] = frame; a.nch[2] = 1;
            if (t->sem) { a.src[3] = semantic_frame; a.nch[3] = 1; a.n_src = 4; }
        }
        a.c4 = t->sl4; a.npix = t->npix; a.dst = planes(t->dbuf[hd]);
        hipLaunchKernelGGL(train_pack_input_kernel, px_grid(t->npix, t->sl4), dim3(256), 0, c.st, a);
    }
    return check_hip(hipGetLastError(), "ojf_trainer_forward (net input)");
}

static int t_forward_net(TCtx &c)
{
    ojf_trainer *t = c.t;
    const size_t heads = t->dbuf.size();
    for (size_t hd = 0; hd < heads; ++hd) {
        for (int id : t->dense[hd])
            if (t_units_forward(c, &id, 1)) return -2;
        if (t_vortex_forward(c, t->vortex[hd])) return -2;
    }
    for (size_t vi = heads; vi < t->vortex.size(); ++vi)
        if (t_vortex_forward(c, t->vortex[vi])) return -2;
    for (int id : t->pred)
        if (t_units_forward(c, &id, 1)) return -2;
    return check_hip(hipGetLastError(), "ojf_trainer_forward");
}

static int t_forward_out(TCtx &c, float *est)
{
    ojf_trainer *t = c.t;
    hipLaunchKernelGGL(train_planes_to_nchw_kernel, px_grid(t->npix, (t->P + 3) / 4), dim3(256), 0, c.st, planes(t->est_planes), (t->P + 3) / 4, t->P,
                       t->npix, est);
    return check_hip(hipGetLastError(), "ojf_trainer_forward (est)");
}

static int t_backward_in(TCtx &c, const float *d_est)
{
    ojf_trainer *t = c.t;
    hipLaunchKernelGGL(train_nchw_to_planes_kernel, px_grid(t->npix, (t->P + 3) / 4), dim3(256), 0, c.st, d_est, t->P, t->npix, planes(t->dest_planes));
    return check_hip(hipGetLastError(), "ojf_trainer_backward (d_est)");
}

static int t_backward_net(TCtx &c)
{
    ojf_trainer *t = c.t;
    for (auto &kv : t->groups_of) t->written[kv.first].assign(kv.second, 0);
    for (auto it = t->pred.rbegin(); it != t->pred.rend(); ++it) {
        const int id = *it;
        if (t_units_backward(c, &id, 1)) return -2;
    }
    const size_t heads = t->dbuf.size();
    for (size_t vi = t->vortex.size(); vi-- > heads;)
        if (t_vortex_backward(c, t->vortex[vi])) return -2;
    for (size_t hd = heads; hd-- > 0;) {
        if (t_vortex_backward(c, t->vortex[hd])) return -2;
        for (auto it = t->dense[hd].rbegin(); it != t->dense[hd].rend(); ++it) {
            const int id = *it;
            if (t_units_backward(c, &id, 1)) return -2;
        }
    }
    if (t->use_side && t->side) {  // the gradient tensors are complete for whatever the caller enqueues next
        OJF_HIP(hipEventRecord(t->join_ev, t->side));
        OJF_HIP(hipStreamWaitEvent(c.st, t->join_ev, 0));
    }
    OJF_HIP(hipMemsetAsync(t->bnd_pool, 0, t->bnd_words * 4, c.st));  // (behind every launch that read this pass's bounds)
    ++t->launches;
    return check_hip(hipGetLastError(), "ojf_trainer_backward");
}

// ---- optional hipGraph replay of a pass ---------------------------------------------------------------------------------
static inline void t_mix(unsigned long long &h, unsigned long long v)
{
    for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 1099511628211ull; }  // FNV-1a
}
static inline unsigned long long t_bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline unsigned long long t_addr(const void *p) { return (unsigned long long)(uintptr_t)p; }

// everything the replayed launches of a pass take from the caller: the layer table field by field, the arithmetic
static unsigned long long t_pass_key(const ojf_trainer *t, const ojf_train_layer *L, int backward)
{
    unsigned long long h = 1469598103934665603ull;
    t_mix(h, (unsigned long long)backward * 4 + (unsigned long long)t->fwd_arith * 2 + (unsigned long long)t->bwd_arith * 16 + (t->use_side ? 1 : 0));
    for (int i = 0; i < t->n_layers; ++i) {
        const ojf_train_layer &l = L[i];
        const void *ptrs[] = {l.weight, l.bias, l.gamma, l.beta, l.running_mean, l.running_var, l.drop_scale};
        for (const void *q : ptrs) t_mix(h, t_addr(q));
        if (backward) {  // (a forward pass reads neither the gradient tensors nor the accumulate flag)
            const void *grads[] = {l.grad_weight, l.grad_bias, l.grad_gamma, l.grad_beta};
            for (const void *q : grads) t_mix(h, t_addr(q));
            t_mix(h, (unsigned long long)(unsigned)l.accumulate);
        }
        t_mix(h, ((unsigned long long)(unsigned)l.out_channels << 32) | (unsigned)l.in_channels);
        t_mix(h, ((unsigned long long)(unsigned)l.ksize << 32) | (unsigned)l.dilation);
        t_mix(h, (unsigned long long)(unsigned)l.bn_training);
        t_mix(h, (t_bits(l.momentum) << 32) | t_bits(l.eps));
    }
    return h ? h : 1;
}

static void t_drop_replays(ojf_trainer *t)
{
    for (auto &v : t->replays) {
        for (auto &r : v) (void)hipGraphExecDestroy(r.exec);
        v.clear();
    }
    t->missed[0] = t->missed[1] = 0;
}

// Replays the pass `key` on `st` if a graph of it exists; captures one (and replays it) if the previous miss had the same
// key.  Returns 1 = the pass is enqueued, 0 = run the plain launches, < 0 = error.  `body(ctx)` enqueues the pass on ctx.st.
template <class Body>
static int t_replay(ojf_trainer *t, const ojf_train_layer *L, int which, unsigned long long key, hipStream_t st, Body body)
{
    auto &v = t->replays[which];
    for (auto &r : v)
        if (r.key == key) {
            OJF_HIP(hipGraphLaunch(r.exec, st));
            r.stamp = ++t->graph_clock;
            t->launches = r.launches;
            ++t->graph_replays;
            return 1;
        }
    if (t->missed[which] != key) { t->missed[which] = key; return 0; }
    if (!t->cap && hipStreamCreateWithFlags(&t->cap, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); t->graph_on = false; return 0; }
    if (hipStreamBeginCapture(t->cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); t->graph_on = false; return 0; }
    TCtx c{t, L, t->cap};
    t->launches = 0;
    const int rc = body(c);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(t->cap, &g);
    hipGraphExec_t exec = nullptr;
    if (rc || e != hipSuccess || !g || hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        t->graph_on = false;  // (capture is an optimisation only: the plain launches remain)
        return rc ? rc : 0;
    }
    (void)hipGraphDestroy(g);
    if (v.size() >= 8) {  // tensors that alternate between a few addresses: keep the most recent graphs
        size_t old = 0;
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i].stamp < v[old].stamp) old = i;
        (void)hipGraphExecDestroy(v[old].exec);
        v.erase(v.begin() + old);
    }
    v.push_back({key, exec, t->launches, ++t->graph_clock});
    ++t->graph_captures;
    OJF_HIP(hipGraphLaunch(exec, st));
    ++t->graph_replays;
    return 1;
}

}  // namespace ojf

// ---- C ABI ----------------------------------------------------------------------------------------------------------
OJF_API void ojf_trainer_destroy(ojf_trainer *t)
{
    if (!t) return;
    for (void *p : t->allocs) (void)hipFree(p);
    for (hipEvent_t e : t->fork_ev) (void)hipEventDestroy(e);
    if (t->join_ev) (void)hipEventDestroy(t->join_ev);
    if (t->side) (void)hipStreamDestroy(t->side);
    ojf::t_drop_replays(t);
    if (t->cap) (void)hipStreamDestroy(t->cap);
    delete t;
}

OJF_API int ojf_trainer_create(ojf_trainer **out, int version, int n_points, int growth, int use_semantics, float output_scale, int h, int w)
{
    using namespace ojf;
    if (!out) return fail("ojf_trainer_create: null output pointer");
    *out = nullptr;
    if ((version != 2 && version != 3) || n_points < 1 || growth < 1 || h < 1 || w < 1) return fail("ojf_trainer_create: bad argument");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count <= 0) return fail("ojf_trainer_create: no HIP device");
    ojf_trainer *t = new ojf_trainer();
    t->version = version; t->P = n_points; t->gf = growth; t->sem = use_semantics ? 1 : 0; t->h = h; t->w = w; t->npix = h * w; t->scale = output_scale;
    t->c = 2 * n_points + 1 + ((version == 2 && t->sem) ? 1 : 0);
    t->sl4 = (t->c + 3) / 4;
    t->out_c = t->c * (growth + 1);
    t->o4 = (t->out_c + 3) / 4;
    t->n_layers = layer_count(version, growth, t->sem);
    int rc = 0;
    auto build = [&]() -> int {
        if (t_alloc(t, reinterpret_cast<void **>(&t->zero_bias), 4096 * 4, true)) return -2;
        const int slot = 4 * t->sl4, oslot = 4 * t->o4;
        int li = 0;
        float *vin = nullptr, *dvin = nullptr;  // input of the last VortexPooling
        int vin_c4 = 0, vin_ic = 0;
        if (version == 3) {
            const int heads = t->sem ? 2 : 1;
            if (t_planes(t, &vin, heads * t->o4) || t_planes(t, &dvin, heads * t->o4, true)) return -2;
            for (int hd = 0; hd < heads; ++hd) {
                if (t_build_dense(t, li)) return -2;
                li += 2 * growth;
                if (t_build_vortex(t, li, t->dbuf[hd], (growth + 1) * t->sl4, t->out_c, t->c, slot, t->ddbuf[hd], vin, hd * t->o4, dvin)) return -2;
                li += 18;
            }
            vin_c4 = heads * t->o4; vin_ic = heads * t->out_c;
        } else {
            if (t_build_dense(t, li)) return -2;
            li += 2 * growth;
            if (t_planes(t, &vin, t->o4) || t_planes(t, &dvin, t->o4, true)) return -2;
            if (t_build_vortex(t, li, t->dbuf[0], (growth + 1) * t->sl4, t->out_c, t->c, slot, t->ddbuf[0], vin, 0, dvin)) return -2;
            li += 18;
            vin_c4 = t->o4; vin_ic = t->out_c;
        }
        float *pin, *dpin;
        if (t_planes(t, &pin, t->o4) || t_planes(t, &dpin, t->o4, true)) return -2;
        if (t_build_vortex(t, li, vin, vin_c4, vin_ic, t->out_c, oslot, dvin, pin, 0, dpin)) return -2;
        li += 18;
        // prediction head (modules/model.py:24-52): gf stacks; the last one ends conv -> leaky, conv -> tanh
        float *cur = pin, *dcur = dpin;
        int cin = t->out_c;
        for (int i = 0; i < growth; ++i) {
            const int cout = (growth - i) * t->c;
            const bool last = i == growth - 1;
            const int widths[3] = {cout, cout, n_points};
            for (int k = 0; k < (last ? 3 : 2); ++k) {
                const int oc = widths[k];
                float *o, *dobuf;
                if (t_planes(t, &o, (oc + 3) / 4) || t_planes(t, &dobuf, (oc + 3) / 4, true)) return -2;
                const bool tanh_layer = last && k == 2;
                const int has_bn = (last && k >= 1) ? 0 : 1;
                const int u = t_add_unit(t, li++, oc, cin, 1, 1, cin, round_up(cin, 4), tanh_layer ? OJF_ACT_TANH : OJF_ACT_LEAKY, has_bn,
                                         tanh_layer ? output_scale : 1.0f, cur, 0, (cin + 3) / 4, dcur, o, 0, dobuf);
                if (u < 0) return -2;
                t->pred.push_back(u);
                cur = o; dcur = dobuf; cin = oc;
            }
        }
        t->est_planes = cur; t->dest_planes = dcur;
        if (li != t->n_layers) return fail("ojf_trainer_create: internal layer count mismatch");
        if (t_alloc(t, reinterpret_cast<void **>(&t->scales), (size_t)t->n_sg * 4)) return -2;
        std::vector<float> ones((size_t)t->n_sg, 1.0f);
        OJF_HIP(hipMemcpy(t->scales, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
        t->bnd_words = 0;
        for (const TUnit &u : t->units) t->bnd_words += (size_t)8 * u.c4_out;
        if (t_alloc(t, reinterpret_cast<void **>(&t->bnd_pool), t->bnd_words * 4, true)) return -2;
        size_t at = 0;
        for (TUnit &u : t->units) { u.bnd = t->bnd_pool + at; at += (size_t)8 * u.c4_out; }
        return 0;
    };
    rc = build();
    if (!rc) {
        static const bool no_side = getenv("OJF_TRAIN_SIDE") && atoi(getenv("OJF_TRAIN_SIDE")) == 0;  // A/B switch
        t->use_side = !no_side;
        if (t->use_side) {
            rc = check_hip(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking), "trainer side stream");
            for (int i = 0; i < 48 && !rc; ++i) {
                hipEvent_t e = nullptr;
                rc = check_hip(hipEventCreateWithFlags(&e, event_flags()), "trainer fork event");
                if (!rc) t->fork_ev.push_back(e);
            }
            if (!rc) rc = check_hip(hipEventCreateWithFlags(&t->join_ev, event_flags()), "trainer join event");
        }
    }
    if (rc) { ojf_trainer_destroy(t); return rc; }
    static const bool graph_env = getenv("OJF_TRAIN_GRAPH") && atoi(getenv("OJF_TRAIN_GRAPH")) != 0;  // A/B switch
    t->graph_on = graph_env;
    *out = t;
    return 0;
}

OJF_API int ojf_trainer_set_graph(ojf_trainer *t, int enable)
{
    using namespace ojf;
    if (!t) return fail("ojf_trainer_set_graph: null trainer");
    t->graph_on = enable != 0;
    if (!t->graph_on) t_drop_replays(t);
    return 0;
}

OJF_API int ojf_trainer_graph_replays(const ojf_trainer *t) { return t ? t->graph_replays : -1; }

OJF_API int ojf_trainer_set_arithmetic(ojf_trainer *t, int arithmetic)
{
    using namespace ojf;
    if (!t || (arithmetic != OJF_ARITH_F32 && arithmetic != OJF_ARITH_F16X3)) return fail("ojf_trainer_set_arithmetic: bad argument");
    if (t->fwd_arith != arithmetic) t->epoch = ~0ull;  // the packed copies of the other arithmetic are stale
    t->fwd_arith = arithmetic;
    return 0;
}

OJF_API int ojf_trainer_set_backward_arithmetic(ojf_trainer *t, int arithmetic)
{
    using namespace ojf;
    if (!t || (arithmetic != OJF_ARITH_F32 && arithmetic != OJF_ARITH_F16X3)) return fail("ojf_trainer_set_backward_arithmetic: bad argument");
    t->bwd_arith = arithmetic;
    return 0;
}

OJF_API int ojf_trainer_layer_count(const ojf_trainer *t) { return t ? t->n_layers : -1; }
OJF_API int ojf_trainer_launch_count(const ojf_trainer *t) { return t ? t->launches : -1; }

OJF_API int ojf_trainer_forward(ojf_trainer *t, const ojf_train_layer *layers, int n_layers, unsigned long long weights_epoch,
                                const float *values, const float *weights, const float *frame, const float *semantic_frame,
                                float *est, ojf_stream_t stream)
{
    using namespace ojf;
    if (!t || !layers || !values || !weights || !frame || !est) return fail("ojf_trainer_forward: null pointer argument");
    if (n_layers != t->n_layers) return fail("ojf_trainer_forward: wrong number of layers");
    if (t->sem && !semantic_frame) return fail("ojf_trainer_forward: the net has a semantic channel but semantic_frame is NULL");
    // (polled without synchronising and NOT cleared here: the flag is shared with the inference nets and the 2-D engine,
    // whose frames since the event must stay unfused until ojf_net_check has reported it - the caller clears it there
    // and may then switch the arithmetic and go on)
    if (t->fwd_arith == OJF_ARITH_F16X3 && g_ovf_host && *g_ovf_host) return fail(*g_ovf_host == 2 ? kChainStuckMsg : kOverflowMsg);
    TCtx c{t, layers, as_stream(stream)};
    t->launches = 0;
    if (t->epoch != weights_epoch) {
        if (t_pack_weights(c)) return -2;
        t->epoch = weights_epoch;
    }
    if (t_forward_in(c, values, weights, frame, semantic_frame)) return -2;
    const int n_in = (int)t->dbuf.size();
    auto body = [&](TCtx &cc) { return t_forward_net(cc); };
    int done = 0;
    if (t->graph_on) {
        done = t_replay(t, layers, 0, t_pass_key(t, layers, 0), c.st, body);
        if (done < 0) return done;
    }
    if (!done && body(c)) return -2;
    if (t_forward_out(c, est)) return -2;
    t->launches += n_in + 1;
    t->have_forward = true;
    return 0;
}

OJF_API int ojf_trainer_backward(ojf_trainer *t, const ojf_train_layer *layers, int n_layers, const float *d_est, ojf_stream_t stream)
{
    using namespace ojf;
    if (!t || !layers || !d_est) return fail("ojf_trainer_backward: null pointer argument");
    if (n_layers != t->n_layers) return fail("ojf_trainer_backward: wrong number of layers");
    if (!t->have_forward) return fail("ojf_trainer_backward: no forward pass to differentiate (one backward per forward)");
    TCtx c{t, layers, as_stream(stream)};
    t->launches = 0;
    if (t_backward_in(c, d_est)) return -2;
    auto body = [&](TCtx &cc) { return t_backward_net(cc); };
    int done = 0;
    if (t->graph_on) {
        done = t_replay(t, layers, 1, t_pass_key(t, layers, 1), c.st, body);
        if (done < 0) return done;
    }
    if (!done && body(c)) return -2;
    t->launches += 1;
    t->have_forward = false;
    return 0;
}

OJF_API int ojf_train_fuse_output(const float *est_pn, const float *fv_pn, const float *fw_pn, const long long *valid, int n, int n_points,
                                  long long n_valid, float init_value, float *fused_rows, ojf_stream_t stream)
{
    using namespace ojf;
    if (!est_pn || !fv_pn || !fw_pn || !fused_rows || (n_valid > 0 && !valid)) return fail("ojf_train_fuse_output: null pointer argument");
    if (n < 1 || n_points < 1 || n_valid < 0 || n_valid > n) return fail("ojf_train_fuse_output: bad sizes");
    if (n_valid == 0) return 0;
    FuseOutArgs a{est_pn, fv_pn, fw_pn, valid, fused_rows, nullptr, n, n_points, n_valid, init_value};
    hipLaunchKernelGGL(train_fuse_output_kernel, dim3((unsigned)((n_valid + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "train_fuse_output_kernel launch");
}

OJF_API int ojf_train_fuse_output_bwd(const float *d_fused_rows, const float *est_pn, const float *fw_pn, const long long *valid, int n,
                                      int n_points, long long n_valid, float init_value, float *d_est_pn, ojf_stream_t stream)
{
    using namespace ojf;
    if (!d_fused_rows || !est_pn || !fw_pn || !d_est_pn || (n_valid > 0 && !valid)) return fail("ojf_train_fuse_output_bwd: null pointer argument");
    if (n < 1 || n_points < 1 || n_valid < 0 || n_valid > n) return fail("ojf_train_fuse_output_bwd: bad sizes");
    OJF_HIP(hipMemsetAsync(d_est_pn, 0, (size_t)n * n_points * sizeof(float), as_stream(stream)));
    if (n_valid == 0) return 0;
    FuseOutArgs a{est_pn, nullptr, fw_pn, valid, const_cast<float *>(d_fused_rows), d_est_pn, n, n_points, n_valid, init_value};
    hipLaunchKernelGGL(train_fuse_output_bwd_kernel, dim3((unsigned)((n_valid + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "train_fuse_output_bwd_kernel launch");
}

OJF_API size_t ojf_train_loss_partial_doubles(long long n_valid) { return n_valid > 0 ? (size_t)((n_valid + ojf::kLossThreads - 1) / ojf::kLossThreads) * 4 : 4; }

OJF_API int ojf_train_fusion_loss(const float *est_rows, const float *target_rows, long long n_valid, int n_points, float w_l1, float w_l2,
                                  float w_cos, double *partial, float *loss_out, ojf_stream_t stream)
{
    using namespace ojf;
    if (!est_rows || !target_rows || !partial || !loss_out) return fail("ojf_train_fusion_loss: null pointer argument");
    if (n_valid < 1 || n_points < 1) return fail("ojf_train_fusion_loss: empty batch (utils/loss.py:81-82 returns the constant 1: handle it on the host)");
    LossArgs a{};
    a.est = est_rows; a.tgt = target_rows; a.nv = n_valid; a.P = n_points; a.w1 = w_l1; a.w2 = w_l2; a.w3 = w_cos; a.partial = partial; a.loss = loss_out;
    a.blocks = (int)((n_valid + kLossThreads - 1) / kLossThreads);
    hipLaunchKernelGGL(train_loss_partial_kernel, dim3(a.blocks), dim3(kLossThreads), 0, as_stream(stream), a);
    hipLaunchKernelGGL(train_loss_finish_kernel, dim3(1), dim3(64), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "train_loss kernels launch");
}

OJF_API int ojf_train_fusion_loss_bwd(const float *est_rows, const float *target_rows, long long n_valid, int n_points, float w_l1, float w_l2,
                                      const float *grad_out, float *d_est_rows, ojf_stream_t stream)
{
    using namespace ojf;
    if (!est_rows || !target_rows || !grad_out || !d_est_rows) return fail("ojf_train_fusion_loss_bwd: null pointer argument");
    if (n_valid < 1 || n_points < 1) return fail("ojf_train_fusion_loss_bwd: empty batch");
    LossArgs a{};
    a.est = est_rows; a.tgt = target_rows; a.nv = n_valid; a.P = n_points; a.w1 = w_l1; a.w2 = w_l2; a.grad_out = grad_out; a.d_est = d_est_rows;
    const long long total = n_valid * n_points;
    hipLaunchKernelGGL(train_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "train_loss_bwd_kernel launch");
}
