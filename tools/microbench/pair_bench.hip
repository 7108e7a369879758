// Profiling-only microbenchmark (not product, not a test): dense_pair_kernel per dense Block i = 0..4 at 320x240,
// wall time per launch and s_memtime phase stamps of one block (compile with -DOJF_PAIR_TIMING).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOJF_PAIR_TIMING -c tools/microbench/pair_bench.hip -o /tmp/pb.o
//        hipcc --offload-arch=gfx950 /tmp/pb.o online_joint_depthfusion_and_semantic_amd/csrc/ojf_api.o -o tools/microbench/pair_bench.bin
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

int main(int argc, char **argv)
{
    const int h = argc > 2 ? atoi(argv[1]) : 240, w = argc > 2 ? atoi(argv[2]) : 320, npix = h * w, c = 19, cs = 20, gf = 5;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *X;
    alloc_planes(&X, npix, (gf + 1) * cs);
    std::vector<float> hx((size_t)npix * (gf + 1) * cs);
    for (auto &v : hx) v = nd(rng);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_pair_dbg, 64 * sizeof(long long));
    for (int i = 0; i < gf; ++i) {
        std::vector<float> wa((size_t)c * (i + 1) * c * 9), wb((size_t)c * c * 9), bs(c, 0.01f);
        for (auto &v : wa) v = nd(rng) * std::sqrt(2.0f / ((i + 1) * c * 9));
        for (auto &v : wb) v = nd(rng) * std::sqrt(2.0f / (c * 9));
        ojf_conv_layer la{(i + 1) * c, c, 3, 1, wa.data(), bs.data()}, lb{c, c, 3, 1, wb.data(), bs.data()};
        ConvBuilder ba((i + 1) * cs, cs, 3, 1), bb(cs, cs, 3, 1);
        ba.add(la, 0, (i + 1) * c, slot_map((i + 1) * c, c, cs), 0, true);
        bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
        PackedPair pp;
        if (finish_pair(ba, bb, pp, pair_cfg_for(h, w, cs))) { printf("pack failed: %s\n", ojf_last_error()); return 1; }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 5; ++r) launch_pair(pp, X, 0, X, (i + 1) * (cs / 4), h, w, 0);
        hipEventRecord(e0, 0);
        const int reps = 50;
        for (int r = 0; r < reps; ++r) launch_pair(pp, X, 0, X, (i + 1) * (cs / 4), h, w, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long st[64];
        hipMemcpy(st, g_pair_dbg, sizeof(st), hipMemcpyDeviceToHost);
        const int n = 2 + 3 * pp.n_chunks + 5;
        printf("pair %d (%d chunks): %.2f us per launch; stamps (cycles from start):", i, pp.n_chunks, ms * 1e3f / reps);
        for (int k = 1; k < n; ++k) printf(" %lld", st[k] - st[0]);
        printf("\n");
        release(pp);
    }
    return 0;
}
