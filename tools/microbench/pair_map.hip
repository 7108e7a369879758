// Debug-only: which output channel does each packed weight row feed?  conv a: zero weights, bias 1 -> T = 1; conv b: a single
// weight (oc*, centre tap, input channel ci) = 1 + 2^-12 (non-zero lo half), bias 0 -> out[oc*] = 1 + 2^-12, others 0.
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"
using namespace ojf;
int main(int argc, char **argv)
{
    const int cfg = argc > 1 ? atoi(argv[1]) : 5, h = 48, w = 80, npix = h * w, c = 19, cs = 20;
    float *X;
    alloc_planes(&X, npix, 2 * cs);
    std::vector<float> hx((size_t)npix * 2 * cs, 0.5f);
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int ocs = 0; ocs < c; ++ocs) {
        std::vector<float> wa((size_t)c * c * 9, 0.0f), wb((size_t)c * c * 9, 0.0f), ba_(c, 1.0f), bb_(c, 0.0f);
        const int ci = (ocs * 7) % c;
        wb[((size_t)ocs * c + ci) * 9 + 4] = 1.0f + 1.0f / 4096.0f;
        ojf_conv_layer la{c, c, 3, 1, wa.data(), ba_.data()}, lb{c, c, 3, 1, wb.data(), bb_.data()};
        ConvBuilder ba(cs, cs, 3, 1), bb(cs, cs, 3, 1);
        ba.add(la, 0, c, slot_map(c, c, cs), 0, true);
        bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
        PackedPair pp;
        if (finish_pair(ba, bb, pp, cfg)) { printf("pack failed: %s\n", ojf_last_error()); return 1; }
        if (launch_pair(pp, X, 0, X, cs / 4, h, w, 0)) { printf("launch failed: %s\n", ojf_last_error()); return 1; }
        (void)hipDeviceSynchronize();
        std::vector<float> out((size_t)npix * cs);
        (void)hipMemcpy(out.data(), X + (size_t)cs * npix, out.size() * 4, hipMemcpyDeviceToHost);
        const int p = 20 * w + 33;
        printf("oc* %2d:", ocs);
        for (int oc = 0; oc < cs; ++oc) {
            const float v = out[((size_t)(oc / 4) * npix + p) * 4 + oc % 4];
            if (v != 0.0f) printf("  [%d] %.9g", oc, v);
        }
        printf("   (want [%d] %.9g)\n", ocs, 1.0 + 1.0 / 4096.0);
        release(pp);
    }
    return 0;
}
