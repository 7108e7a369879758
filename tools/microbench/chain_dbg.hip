// Debug-only: which (tap, input channel) weights of the second convolution reach output channel `oc` wrongly in
// dense_chain_kernel (delta weights, one Block), compared with dense_pair_kernel.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"
using namespace ojf;
static void to_split(const float *v4, uint16_t *dst8)
{
    for (int j = 0; j < 4; ++j) {
        const _Float16 hi = (_Float16)v4[j];
        const _Float16 lo = (_Float16)(v4[j] - (float)hi);
        dst8[j] = __builtin_bit_cast(uint16_t, hi);
        dst8[4 + j] = __builtin_bit_cast(uint16_t, lo);
    }
}
static float from_split(const uint16_t *s8, int j) { return (float)__builtin_bit_cast(_Float16, s8[j]) + (float)__builtin_bit_cast(_Float16, s8[4 + j]); }
int main(int argc, char **argv)
{
    const int h = 240, w = 320, npix = h * w, c = 19, cs = 20;
    const int oc_t = argc > 1 ? atoi(argv[1]) : 16;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *X, *XS;
    alloc_planes(&X, npix, 2 * cs);
    alloc_planes(&XS, npix, 2 * cs);
    std::vector<float> hx((size_t)npix * cs, 0.0f);
    for (int q = 0; q < 5; ++q)
        for (int p = 0; p < npix; ++p)
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < c) hx[((size_t)q * npix + p) * 4 + j] = nd(rng);
    std::vector<float> hs(hx.size());
    for (size_t i = 0; i < hx.size(); i += 4) to_split(&hx[i], reinterpret_cast<uint16_t *>(&hs[i]));
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(XS, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> wa((size_t)c * c * 9, 0.0f), bs(c, 0.0f);
    for (int o = 0; o < c; ++o) wa[((size_t)o * c + o) * 9 + 4] = 1.0f;  // conv a = identity
    for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < c; ++ci) {
            std::vector<float> wb((size_t)c * c * 9, 0.0f);
            wb[((size_t)oc_t * c + ci) * 9 + tap] = 1.0f;
            ojf_conv_layer la{c, c, 3, 1, wa.data(), bs.data()}, lb{c, c, 3, 1, wb.data(), bs.data()};
            ConvBuilder ba(cs, cs, 3, 1), bb(cs, cs, 3, 1);
            ba.add(la, 0, c, slot_map(c, c, cs), 0, true);
            bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
            PackedPair pp;
            PackedChain pc;
            if (finish_pair(ba, bb, pp, 0) || finish_chain({ba}, {bb}, pc)) { printf("pack failed\n"); return 1; }
            launch_pair(pp, X, 0, X, 5, h, w, 0);
            if (launch_chain(pc, XS, h, w, 0)) { printf("launch failed: %s\n", ojf_last_error()); return 1; }
            (void)hipDeviceSynchronize();
            std::vector<float> ya((size_t)npix * cs), yb(ya.size());
            (void)hipMemcpy(ya.data(), X + (size_t)npix * cs, ya.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(yb.data(), XS + (size_t)npix * cs, yb.size() * 4, hipMemcpyDeviceToHost);
            double md = 0;
            size_t bad = 0;
            for (int q = 0; q < 5; ++q)
                for (int p = 0; p < npix; ++p)
                    for (int j = 0; j < 4; ++j) {
                        const size_t idx = ((size_t)q * npix + p) * 4;
                        const double d = std::fabs((double)ya[idx + j] - from_split(reinterpret_cast<const uint16_t *>(&yb[idx]), j));
                        md = std::fmax(md, d);
                        bad += d > 1e-4;
                    }
            printf("tap %d ci %2d: max diff %.3e, %zu bad%s", tap, ci, md, bad, ci % 4 == 3 || ci == c - 1 ? "\n" : " | ");
            release(pp);
            release(pc);
        }
    return 0;
}
