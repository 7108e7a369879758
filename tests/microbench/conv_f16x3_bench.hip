// Profiling-only microbenchmark (not product, not a test): split-fp16 ("f16x3") variant of the generic
// convolution kernel against the fp32-MFMA one, on the same C4-planar fp32 activations.
//   x = xh + xl, w = wh + wl (fp16 each, RNE);  x*w ~= wl*xh + wh*xl + wh*xh  (drops wl*xl ~ 2^-22 |x w|)
// on v_mfma_f32_16x16x32_f16 (8192 MAC / 16 cycles vs 1024 MAC / 32 cycles for v_mfma_f32_16x16x4_f32).
// Build: see tests/microbench/run_f16x3.sh
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// K slot (g, j) of superstep S: j < 4 -> channel j of entry 8S + 2g, j >= 4 -> channel j-4 of entry 8S + 2g + 1
template <int MT, int NT, bool SKIP = true>
__global__ __launch_bounds__(256) void conv_f16x3_kernel(const ConvGroup grp)
{
    const ConvArgs &a = grp.g[blockIdx.y];  // nsteps counts 8-entry supersteps here
    __shared__ int2 tab[(kMaxSteps + kPadSteps) * 8];
    for (int G = threadIdx.x; G < (a.nsteps + kPadSteps) * 8; G += 256) {
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
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (blockIdx.x * 4 + wave) * (MT * 16);
    if (strip >= a.npix) return;

    int py[MT], px[MT], plin[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = strip + m * 16 + i16;
        plin[m] = p;
        py[m] = p < a.npix ? p / a.w : -0x40000000;
        px[m] = p - (p / a.w) * a.w;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f32x4 *wb[NT];  // [ot][S][hi|lo][lane] 16-byte fragments
#pragma unroll
    for (int n = 0; n < NT; ++n) wb[n] = a.wp + (size_t)n * (a.nsteps + kPadSteps) * 128 + lane;

    auto fetch = [&](f32x4(&xa)[MT], f32x4(&xb)[MT], f32x4(&wh)[NT], f32x4(&wl)[NT], int S, bool &live) {
        const int2 e0 = tab[S * 8 + 2 * g], e1 = tab[S * 8 + 2 * g + 1];
        const int dy0 = e0.y >> 16, dx0 = (int)(short)(e0.y & 0xffff);
        const int dy1 = e1.y >> 16, dx1 = (int)(short)(e1.y & 0xffff);
        bool any_ok = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool ok0 = (unsigned)(py[m] + dy0) < (unsigned)a.h && (unsigned)(px[m] + dx0) < (unsigned)a.w;
            const bool ok1 = (unsigned)(py[m] + dy1) < (unsigned)a.h && (unsigned)(px[m] + dx1) < (unsigned)a.w;
            xa[m] = a.in[ok0 ? e0.x + plin[m] : -1];
            xb[m] = a.in[ok1 ? e1.x + plin[m] : -1];
            any_ok |= ok0 | ok1;
        }
        live = SKIP ? __any(any_ok) : true;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            wh[n] = wb[n][(size_t)S * 128];
            wl[n] = wb[n][(size_t)S * 128 + 64];
        }
    };
    auto mac = [&](const f32x4(&xa)[MT], const f32x4(&xb)[MT], const f32x4(&wh)[NT], const f32x4(&wl)[NT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x8 x = __builtin_shufflevector(xa[m], xb[m], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 xh = __builtin_convertvector(x, f16x8);
            const f16x8 xl = __builtin_convertvector(x - __builtin_convertvector(xh, f32x8), f16x8);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f16x8 h = __builtin_bit_cast(f16x8, wh[n]), l = __builtin_bit_cast(f16x8, wl[n]);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l, xh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xh, acc[m][n], 0, 0, 0);
            }
        }
    };

    f32x4 xa0[MT], xb0[MT], h0[NT], l0[NT], xa1[MT], xb1[MT], h1[NT], l1[NT], xa2[MT], xb2[MT], h2[NT], l2[NT];
    bool v0, v1, v2;
    fetch(xa0, xb0, h0, l0, 0, v0);
    fetch(xa1, xb1, h1, l1, 1, v1);
    for (int S = 0; S < a.nsteps; S += 3) {
        fetch(xa2, xb2, h2, l2, S + 2, v2);
        if (!SKIP || v0) mac(xa0, xb0, h0, l0);
        fetch(xa0, xb0, h0, l0, S + 3, v0);
        if (!SKIP || v1) mac(xa1, xb1, h1, l1);
        fetch(xa1, xb1, h1, l1, S + 4, v1);
        if (!SKIP || v2) mac(xa2, xb2, h2, l2);
    }

    const float slope = act_slope(a.act);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int og = n * 4 + g;
        const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = strip + m * 16 + i16;
            f32x4 v = acc[m][n] + b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lin = v[j];
                const float r = lin > 0.0f ? lin : lin * slope;
                v[j] = (og * 4 + j < a.act_n ? r : lin) * a.scale;
            }
            if (p < a.npix && og < a.og_store) a.out[(size_t)(a.out_g0 + og) * a.npix + p] = v;
        }
    }
}


// v2: weights staged through LDS in chunks of CS supersteps shared by the 4 waves of a block
// global layout [S][n][hi|lo][lane] so that a chunk is contiguous
constexpr int kPad2 = 6;
template <int NT> struct ChunkSteps { static constexpr int value = NT <= 2 ? 6 : 3; };

template <int MT, int NT, bool SKIP = true>
__global__ __launch_bounds__(256) void conv_f16x3_lds_kernel(const ConvGroup grp)
{
    constexpr int CS = ChunkSteps<NT>::value;
    const ConvArgs &a = grp.g[blockIdx.y];
    __shared__ int2 tab[(kMaxSteps + kPad2) * 8];
    __shared__ f32x4 wl[CS * NT * 128];
    for (int G = threadIdx.x; G < (a.nsteps + kPad2) * 8; G += 256) {
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

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int strip = (blockIdx.x * 4 + wave) * (MT * 16);

    int py[MT], px[MT], plin[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = strip + m * 16 + i16;
        plin[m] = p;
        py[m] = p < a.npix ? p / a.w : -0x40000000;
        px[m] = p - (p / a.w) * a.w;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](f32x4(&xa)[MT], f32x4(&xb)[MT], int S, bool &live) {
        const int2 e0 = tab[S * 8 + 2 * g], e1 = tab[S * 8 + 2 * g + 1];
        const int dy0 = e0.y >> 16, dx0 = (int)(short)(e0.y & 0xffff);
        const int dy1 = e1.y >> 16, dx1 = (int)(short)(e1.y & 0xffff);
        bool any_ok = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool ok0 = (unsigned)(py[m] + dy0) < (unsigned)a.h && (unsigned)(px[m] + dx0) < (unsigned)a.w;
            const bool ok1 = (unsigned)(py[m] + dy1) < (unsigned)a.h && (unsigned)(px[m] + dx1) < (unsigned)a.w;
            xa[m] = a.in[ok0 ? e0.x + plin[m] : -1];
            xb[m] = a.in[ok1 ? e1.x + plin[m] : -1];
            any_ok |= ok0 | ok1;
        }
        live = SKIP ? __any(any_ok) : true;
    };
    auto mac = [&](const f32x4(&xa)[MT], const f32x4(&xb)[MT], int sl) {
        f32x4 wh[NT], wlo[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            wh[n] = wl[(sl * NT + n) * 128 + lane];
            wlo[n] = wl[(sl * NT + n) * 128 + 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x8 x = __builtin_shufflevector(xa[m], xb[m], 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 xh = __builtin_convertvector(x, f16x8);
            const f16x8 xl = __builtin_convertvector(x - __builtin_convertvector(xh, f32x8), f16x8);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f16x8 h = __builtin_bit_cast(f16x8, wh[n]), l = __builtin_bit_cast(f16x8, wlo[n]);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l, xh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, xh, acc[m][n], 0, 0, 0);
            }
        }
    };

    __syncthreads();  // tab
    f32x4 xa0[MT], xb0[MT], xa1[MT], xb1[MT], xa2[MT], xb2[MT];
    bool v0, v1, v2;
    fetch(xa0, xb0, 0, v0);
    fetch(xa1, xb1, 1, v1);
    int sl = 0;
    for (int S = 0; S < a.nsteps; S += 3) {
        if (sl == CS || S == 0) {
            if (S) __syncthreads();
            const f32x4 *src = a.wp + (size_t)S * NT * 128;
#pragma unroll
            for (int i = 0; i < CS * NT * 128 / 256; ++i) wl[i * 256 + threadIdx.x] = src[i * 256 + threadIdx.x];
            __syncthreads();
            sl = 0;
        }
        fetch(xa2, xb2, S + 2, v2);
        if (!SKIP || v0) mac(xa0, xb0, sl);
        fetch(xa0, xb0, S + 3, v0);
        if (!SKIP || v1) mac(xa1, xb1, sl + 1);
        fetch(xa1, xb1, S + 4, v1);
        if (!SKIP || v2) mac(xa2, xb2, sl + 2);
        sl += 3;
    }

    const float slope = act_slope(a.act);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int og = n * 4 + g;
        const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + (size_t)n * 16 + 4 * g);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = strip + m * 16 + i16;
            f32x4 v = acc[m][n] + b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lin = v[j];
                const float r = lin > 0.0f ? lin : lin * slope;
                v[j] = (og * 4 + j < a.act_n ? r : lin) * a.scale;
            }
            if (p < a.npix && og < a.og_store) a.out[(size_t)(a.out_g0 + og) * a.npix + p] = v;
        }
    }
}

static int pack16s(const ConvBuilder &b, int n_ot, _Float16 **dev, int *nsteps_out)
{
    const int c4 = b.c_in_phys / 4, groups = b.taps * c4, nsteps = (groups + 7) / 8, nsp = nsteps + kPad2;
    std::vector<_Float16> wp((size_t)n_ot * nsp * 2 * 64 * 8, (_Float16)0.0f);
    for (int ot = 0; ot < n_ot; ++ot)
        for (int S = 0; S < nsteps; ++S)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int oc = ot * 16 + (lane & 15), G = 8 * S + 2 * (lane >> 4) + (j >> 2);
                    if (oc >= b.c_out_phys || G >= groups) continue;
                    const int t = G / c4, cg = G % c4;
                    const float v = b.W[((size_t)oc * b.taps + t) * b.c_in_phys + 4 * cg + (j & 3)];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    wp[((((size_t)S * n_ot + ot) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    wp[((((size_t)S * n_ot + ot) * 2 + 1) * 64 + lane) * 8 + j] = lo;
                }
    hipMalloc(reinterpret_cast<void **>(dev), wp.size() * 2);
    hipMemcpy(*dev, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    *nsteps_out = nsteps;
    return 0;
}

// packed split weights: [ot][S][hi|lo][lane] x 8 halfs, same (g, j) -> (entry, channel) map as the kernel
static int pack16(const ConvBuilder &b, int n_ot, _Float16 **dev, int *nsteps_out)
{
    const int c4 = b.c_in_phys / 4, groups = b.taps * c4, nsteps = (groups + 7) / 8, nsp = nsteps + kPadSteps;
    std::vector<_Float16> wp((size_t)n_ot * nsp * 2 * 64 * 8, (_Float16)0.0f);
    for (int ot = 0; ot < n_ot; ++ot)
        for (int S = 0; S < nsteps; ++S)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int oc = ot * 16 + (lane & 15), G = 8 * S + 2 * (lane >> 4) + (j >> 2);
                    if (oc >= b.c_out_phys || G >= groups) continue;
                    const int t = G / c4, cg = G % c4;
                    const float v = b.W[((size_t)oc * b.taps + t) * b.c_in_phys + 4 * cg + (j & 3)];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    wp[((((size_t)ot * nsp + S) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    wp[((((size_t)ot * nsp + S) * 2 + 1) * 64 + lane) * 8 + j] = lo;
                }
    hipMalloc(reinterpret_cast<void **>(dev), wp.size() * 2);
    hipMemcpy(*dev, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    *nsteps_out = nsteps;
    return 0;
}

template <typename F>
static float time_it(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int NT>
static void run_shape(const char *name, int cin, int cout, int k, int dil, int h, int w, int ngroup, float xs = 1.0f)
{
    const int cin_p = round_up(cin, 4), cout_p = round_up(cout, 4), npix = h * w;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> wt((size_t)cout * cin * k * k), bs(cout);
    const float ws = std::sqrt(2.0f / (cin * k * k));
    for (auto &v : wt) v = nd(rng) * ws;
    for (auto &v : bs) v = 0.0f;
    ojf_conv_layer L{cin, cout, k, dil, wt.data(), bs.data()};
    ConvBuilder b(cin_p, cout_p, k, dil);
    b.add(L, 0, cin, slot_map(cin, cin, cin_p), 0, true);
    PackedConv pc;
    if (finish(b, pc)) { printf("pack failed\n"); return; }
    if (pc.n_ot != NT) { printf("%s: n_ot=%d != NT=%d\n", name, pc.n_ot, NT); return; }
    _Float16 *w16; int ns16;
    pack16(b, pc.n_ot, &w16, &ns16);

    _Float16 *w16s; int ns16s;
    pack16s(b, pc.n_ot, &w16s, &ns16s);
    float *in, *out32, *out16;
    alloc_planes(&in, npix, cin_p);
    alloc_planes(&out32, npix, cout_p);
    alloc_planes(&out16, npix, cout_p);
    std::vector<float> hx((size_t)npix * cin_p, 0.f);
    for (int cg = 0; cg < cin_p / 4; ++cg)
        for (int p = 0; p < npix; ++p)
            for (int j = 0; j < 4; ++j)
                if (cg * 4 + j < cin) hx[((size_t)cg * npix + p) * 4 + j] = xs * nd(rng) * (1.0f + 3.0f * ((cg * 4 + j) % 3 == 0));
    hipMemcpy(in, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);

    ConvArgs a;
    fill_conv_args(a, pc, in, 0, out32, 0, nullptr, OJF_ACT_LEAKY, cout_p, 1.0f, h, w);
    ConvArgs a16 = a;
    a16.out = planes(out16); a16.wp = reinterpret_cast<const f32x4 *>(w16); a16.nsteps = ns16;
    ConvArgs a16s = a16; a16s.wp = reinterpret_cast<const f32x4 *>(w16s);
    ConvGroup g32, g16, g16s;
    for (int i = 0; i < 4; ++i) { g32.g[i] = a; g16.g[i] = a16; g16s.g[i] = a16s; }
    const int strips = (npix + 15) / 16;
    const dim3 grid((strips + 3) / 4, ngroup), block(256);
    const float t32 = time_it([&] { hipLaunchKernelGGL((conv_mfma_kernel<1, NT>), grid, block, 0, 0, g32); }, 50);
    const float t16 = time_it([&] { hipLaunchKernelGGL((conv_f16x3_kernel<1, NT>), grid, block, 0, 0, g16); }, 50);
    const float t16n = time_it([&] { hipLaunchKernelGGL((conv_f16x3_kernel<1, NT, false>), grid, block, 0, 0, g16); }, 50);
    const dim3 grid2((strips / 2 + 3) / 4, ngroup);
    const float t16m2 = time_it([&] { hipLaunchKernelGGL((conv_f16x3_kernel<2, NT>), grid2, block, 0, 0, g16); }, 50);
    const float tl = time_it([&] { hipLaunchKernelGGL((conv_f16x3_lds_kernel<1, NT>), grid, block, 0, 0, g16s); }, 50);
    const float tln = time_it([&] { hipLaunchKernelGGL((conv_f16x3_lds_kernel<1, NT, false>), grid, block, 0, 0, g16s); }, 50);
    const float tl2 = time_it([&] { hipLaunchKernelGGL((conv_f16x3_lds_kernel<2, NT>), grid2, block, 0, 0, g16s); }, 50);
    hipMemset(out16 , 0, (size_t)npix * cout_p * 4);
    hipLaunchKernelGGL((conv_f16x3_lds_kernel<1, NT>), grid, block, 0, 0, g16s);
    hipDeviceSynchronize();

    std::vector<float> o32((size_t)npix * cout_p), o16((size_t)npix * cout_p);
    hipMemcpy(o32.data(), out32, o32.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o16.data(), out16, o16.size() * 4, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, mx = 0;
    for (int s = 0; s < 3000; ++s) {
        const int p = (int)(rng() % npix), y = p / w, x = p % w;
        for (int oc = 0; oc < cout; ++oc) {
            double acc = bs[oc];
            for (int t = 0; t < k * k; ++t) {
                const int yy = y + (k == 3 ? (t / 3 - 1) * dil : 0), xx = x + (k == 3 ? (t % 3 - 1) * dil : 0);
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                for (int ci = 0; ci < cin; ++ci)
                    acc += (double)wt[((size_t)oc * cin + ci) * k * k + t] *
                           (double)hx[((size_t)(ci / 4) * npix + yy * w + xx) * 4 + (ci & 3)];
            }
            const double ref = acc > 0 ? acc : acc * (double)0.01f;
            const size_t o = ((size_t)(oc / 4) * npix + p) * 4 + (oc & 3);
            e32 = std::fmax(e32, std::fabs(o32[o] - ref));
            e16 = std::fmax(e16, std::fabs(o16[o] - ref));
            mx = std::fmax(mx, std::fabs(ref));
        }
    }
    printf("%-22s x%d NT=%d | f32 %.1f us | f16x3 %.1f us (noskip %.1f, MT2 %.1f) | lds %.1f (noskip %.1f, MT2 %.1f) | max|out| %.2e  err f32 %.2e  f16x3 %.2e\n", name,
           ngroup, NT, t32, t16, t16n, t16m2, tl, tln, tl2, mx, e32, e16);
    free_planes(in); free_planes(out32); free_planes(out16); release(pc); hipFree(w16);
}

int main()
{
    const int h = 240, w = 320;
    run_shape<2>("3x3 19->19 d1", 19, 19, 3, 1, h, w, 1);
    run_shape<2>("3x3 19->19 d1", 19, 19, 3, 1, h, w, 4);
    run_shape<2>("3x3 19->19 d9", 19, 19, 3, 9, h, w, 4);
    run_shape<2>("3x3 19->19 d27", 19, 19, 3, 27, h, w, 4);
    run_shape<2>("3x3 19->19 x*1e-2", 19, 19, 3, 1, h, w, 1, 1e-2f);
    run_shape<2>("3x3 19->19 x*1e-3", 19, 19, 3, 1, h, w, 1, 1e-3f);
    run_shape<2>("3x3 19->19 x*1e-4", 19, 19, 3, 1, h, w, 1, 1e-4f);
    run_shape<2>("3x3 57->19 d1", 57, 19, 3, 1, h, w, 1);
    run_shape<2>("3x3 95->19 d1", 95, 19, 3, 1, h, w, 1);
    run_shape<6>("1x1 114->76", 114, 76, 1, 1, h, w, 1);
    run_shape<8>("1x1 114->114", 114, 114, 1, 1, h, w, 1);
    return 0;
}
