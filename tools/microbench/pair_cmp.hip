// Debug-only: dense_pair_kernel shape A vs shape B on the same inputs, max |difference| per output channel group and block.
// Build like pair_bench.hip (without -DOJF_PAIR_TIMING).  Usage: pair_cmp.exe cfgA cfgB [h w]
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

int main(int argc, char **argv)
{
    const int cfgA = argc > 1 ? atoi(argv[1]) : 0, cfgB = argc > 2 ? atoi(argv[2]) : 5;
    const int h = argc > 4 ? atoi(argv[3]) : 240, w = argc > 4 ? atoi(argv[4]) : 320, npix = h * w, c = 19, cs = 20, gf = 5;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *X[2];
    std::vector<float> hx((size_t)npix * (gf + 1) * cs);
    for (auto &v : hx) v = nd(rng);
    for (int k = 0; k < 2; ++k) {
        alloc_planes(&X[k], npix, (gf + 1) * cs);
        (void)hipMemcpy(X[k], hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < gf; ++i) {
        std::vector<float> wa((size_t)c * (i + 1) * c * 9), wb((size_t)c * c * 9), bs(c, 0.01f);
        for (auto &v : wa) v = nd(rng) * std::sqrt(2.0f / ((i + 1) * c * 9));
        for (auto &v : wb) v = nd(rng) * std::sqrt(2.0f / (c * 9));
        ojf_conv_layer la{(i + 1) * c, c, 3, 1, wa.data(), bs.data()}, lb{c, c, 3, 1, wb.data(), bs.data()};
        ConvBuilder ba((i + 1) * cs, cs, 3, 1), bb(cs, cs, 3, 1);
        ba.add(la, 0, (i + 1) * c, slot_map((i + 1) * c, c, cs), 0, true);
        bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
        std::vector<float> out[2];
        for (int k = 0; k < 2; ++k) {
            PackedPair pp;
            if (finish_pair(ba, bb, pp, k ? cfgB : cfgA)) { printf("pack failed: %s\n", ojf_last_error()); return 1; }
            if (launch_pair(pp, X[k], 0, X[k], (i + 1) * (cs / 4), h, w, 0)) { printf("launch failed: %s\n", ojf_last_error()); return 1; }
            (void)hipDeviceSynchronize();
            out[k].resize((size_t)npix * cs);
            (void)hipMemcpy(out[k].data(), X[k] + (size_t)(i + 1) * cs * npix, out[k].size() * 4, hipMemcpyDeviceToHost);
            release(pp);
        }
        printf("pair %d:", i);
        for (int og = 0; og < 5; ++og) {
            double md = 0, mx = 0;
            for (int p = 0; p < npix; ++p)
                for (int j = 0; j < 4; ++j) {
                    const size_t idx = ((size_t)og * npix + p) * 4 + j;
                    md = std::fmax(md, std::fabs((double)out[0][idx] - out[1][idx]));
                    mx = std::fmax(mx, std::fabs((double)out[0][idx]));
                }
            printf("  og%d max|d| %.3e (max %.2f)", og, md, mx);
        }
        printf("\n");
        // keep the two buffers identical for the next block
        (void)hipMemcpy(X[1] + (size_t)(i + 1) * cs * npix, X[0] + (size_t)(i + 1) * cs * npix, (size_t)npix * cs * 4, hipMemcpyDeviceToDevice);
    }
    return 0;
}
