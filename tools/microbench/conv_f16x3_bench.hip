// Profiling-only microbenchmark (not product, not a test): ablations of the split-fp16 convolution kernel
// (conv_f16x3_kernel<MT, NT, SKIP, ABL>) next to the fp32-MFMA kernel on one layer shape.
//   ABL bits: 1 no activation loads, 2 no LDS weight reads, 4 no MFMA, 8 no fp16 split
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -c tests/microbench/conv_f16x3_bench.hip -o /tmp/b.o
//        hipcc --offload-arch=gfx950 /tmp/b.o online_joint_depthfusion_and_semantic_amd/csrc/ojf_api.o -o tests/microbench/conv_f16x3_bench.bin
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

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

template <int MT, int NT>
static void run_shape(const char *name, int cin, int cout, int k, int dil, int h, int w, int ngroup)
{
    const int cin_p = round_up(cin, 4), cout_p = round_up(cout, 4), npix = h * w;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> wt((size_t)cout * cin * k * k), bs(cout, 0.0f);
    const float ws = std::sqrt(2.0f / (cin * k * k));
    for (auto &v : wt) v = nd(rng) * ws;
    ojf_conv_layer L{cin, cout, k, dil, wt.data(), bs.data()};
    ConvBuilder b(cin_p, cout_p, k, dil);
    b.add(L, 0, cin, slot_map(cin, cin, cin_p), 0, true);
    PackedConv p32, p16;
    if (finish(b, p32, OJF_ARITH_F32) || finish(b, p16, OJF_ARITH_F16X3)) { printf("pack failed\n"); return; }
    if (p16.n_ot != NT) { printf("%s: n_ot=%d != NT=%d\n", name, p16.n_ot, NT); return; }
    float *in, *out;
    alloc_planes(&in, npix, cin_p);
    alloc_planes(&out, npix, cout_p);
    std::vector<float> hx((size_t)npix * cin_p, 0.f);
    for (auto &v : hx) v = nd(rng);
    hipMemcpy(in, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    ConvArgs a32, a16;
    fill_conv_args(a32, p32, in, 0, out, 0, nullptr, OJF_ACT_LEAKY, cout_p, 1.0f, h, w);
    fill_conv_args(a16, p16, in, 0, out, 0, nullptr, OJF_ACT_LEAKY, cout_p, 1.0f, h, w);
    ConvGroup g32, g16;
    g32.nblocks = g16.nblocks = 0;
    for (int i = 0; i < 4; ++i) { g32.g[i] = a32; g16.g[i] = a16; }
    const int strips1 = (npix + 15) / 16, strips = (npix + MT * 16 - 1) / (MT * 16);
    const dim3 grid1((strips1 + 3) / 4, ngroup), grid((strips + 3) / 4, ngroup), block(256);
    const float t32 = time_it([&] { hipLaunchKernelGGL((conv_mfma_kernel<1, NT>), grid1, block, 0, 0, g32); }, 50);
    const size_t tab_bytes = (size_t)(a16.nsteps + kPad16) * 8 * sizeof(int2);
#define T16(ABL_) time_it([&] { hipLaunchKernelGGL((conv_f16x3_kernel<MT, NT, true, ABL_>), grid, block, tab_bytes, 0, g16); }, 50)
    const float t0 = T16(0), t1 = T16(1), t2 = T16(2), t4 = T16(4), t8 = T16(8), t12 = T16(12), t3 = T16(3), t15 = T16(15), t14 = T16(14), t13 = T16(13);
    printf("%-16s x%d MT=%d NT=%d | f32 %.1f | f16x3 %.1f | noX %.1f noW %.1f noMFMA %.1f noSplit %.1f noMFMA+noSplit %.1f noXW %.1f "
           "onlyX %.1f onlyW %.1f nothing %.1f us\n",
           name, ngroup, MT, NT, t32, t0, t1, t2, t4, t8, t12, t3, t14, t13, t15);
    free_planes(in); free_planes(out); release(p32); release(p16);
}

int main()
{
    const int h = 240, w = 320;
    // dense block today: K grows
    run_shape<2, 2>("3x3 19->19", 19, 19, 3, 1, h, w, 1);
    run_shape<2, 2>("3x3 19->19", 19, 19, 3, 1, h, w, 4);
    run_shape<2, 2>("3x3 19->19 d9", 19, 19, 3, 9, h, w, 4);
    run_shape<2, 2>("3x3 38->19", 38, 19, 3, 1, h, w, 1);
    run_shape<2, 2>("3x3 57->19", 57, 19, 3, 1, h, w, 1);
    run_shape<2, 2>("3x3 76->19", 76, 19, 3, 1, h, w, 1);
    run_shape<2, 2>("3x3 95->19", 95, 19, 3, 1, h, w, 1);
    // alternative: every new 19-channel slot feeds all later layers at once (N grows instead of K)
    run_shape<2, 8>("3x3 19->100", 19, 100, 3, 1, h, w, 1);
    run_shape<1, 8>("3x3 19->100", 19, 100, 3, 1, h, w, 1);
    run_shape<2, 6>("3x3 19->80", 19, 80, 3, 1, h, w, 1);
    run_shape<2, 4>("3x3 19->60", 19, 60, 3, 1, h, w, 1);
    run_shape<2, 4>("3x3 19->40", 19, 40, 3, 1, h, w, 1);
    run_shape<2, 6>("1x1 114->76", 114, 76, 1, 1, h, w, 1);
    return 0;
}
