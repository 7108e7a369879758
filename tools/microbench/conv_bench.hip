// Profiling-only microbenchmark (not product, not a test): times conv_mfma_kernel variants on one layer shape.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. tests/microbench/conv_bench.hip \
//        online_joint_depthfusion_and_semantic_amd/csrc/ojf_api.o -o gpurun_out/conv_bench   (see run_conv_bench.sh)
#include <cstdio>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

template <int MT, int NT, int ABL, bool SKIP = true>
static float time_variant(const PackedConv &pc, float *in, float *out, int h, int w, int reps)
{
    ConvArgs a;
    a.in = planes(in); a.out = planes(out); a.out_rows = nullptr;
    a.wp = planes(pc.wp); a.bias = pc.bias;
    a.in_g0 = 0; a.out_g0 = 0; a.rows_stride = 0; a.rows_n = 0;
    a.h = h; a.w = w; a.npix = h * w; a.taps = pc.taps; a.dil = pc.dil;
    a.c4 = pc.c_in_phys / 4; a.nsteps = (pc.taps * a.c4 + 3) / 4;
    a.og_store = pc.c_out_phys / 4; a.act = OJF_ACT_RELU; a.act_n = pc.c_out_phys; a.scale = 1.0f;
    ConvGroup grp;
    grp.nblocks = 0;
    for (int i = 0; i < 4; ++i) grp.g[i] = a;
    const int strips = (a.npix + MT * 16 - 1) / (MT * 16);
    dim3 grid((strips + 3) / 4, 1), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, ABL, SKIP>), grid, block, 0, 0, grp);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, ABL, SKIP>), grid, block, 0, 0, grp);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int MT, int NT>
static void run_shape(const char *name, int cin, int cout, int k, int dil, int h, int w)
{
    const int cin_p = round_up(cin, 4), cout_p = round_up(cout, 4);
    std::vector<float> wt((size_t)cout * cin * k * k, 0.01f), bs(cout, 0.1f);
    ojf_conv_layer L{cin, cout, k, dil, wt.data(), bs.data()};
    ConvBuilder b(cin_p, cout_p, k, dil);
    b.add(L, 0, cin, slot_map(cin, cin, cin_p), 0, true);
    PackedConv pc;
    if (finish(b, pc)) { printf("pack failed: %s\n", ojf_last_error()); return; }
    if (pc.n_ot != NT) { printf("%s: n_ot=%d != NT=%d\n", name, pc.n_ot, NT); return; }
    float *in, *out;
    alloc_planes(&in, (size_t)h * w, cin_p);
    alloc_planes(&out, (size_t)h * w, cout_p);
    const double gmac = (double)h * w * cin_p * k * k * (NT * 16) * 1e-9;
    const int reps = 50;
    float t0 = time_variant<MT, NT, 0>(pc, in, out, h, w, reps);
    float t1 = time_variant<MT, NT, 1>(pc, in, out, h, w, reps);
    float t2 = time_variant<MT, NT, 2>(pc, in, out, h, w, reps);
    float t3 = time_variant<MT, NT, 3>(pc, in, out, h, w, reps);
    float t4 = time_variant<MT, NT, 4>(pc, in, out, h, w, reps);
    float t7 = time_variant<MT, NT, 7>(pc, in, out, h, w, reps);
    float t11 = time_variant<MT, NT, 11>(pc, in, out, h, w, reps);
    float t15 = time_variant<MT, NT, 15>(pc, in, out, h, w, reps);
    float tn = time_variant<MT, NT, 0, false>(pc, in, out, h, w, reps);
    float tn3 = time_variant<MT, NT, 3, false>(pc, in, out, h, w, reps);
    printf("%-26s MT=%d NT=%d padded %.2f GMAC (MFMA-bound %.1f us) | full %.1f  noX %.1f  noW %.1f  noXW %.1f  noMFMA %.1f  "
           "noXW+noMFMA %.1f  noXW+noIdx %.1f  nothing %.1f | NOSKIP full %.1f noXW %.1f us\n",
           name, MT, NT, gmac, gmac / 78.6e-3, t0, t1, t2, t3, t4, t7, t11, t15, tn, tn3);
    free_planes(in); free_planes(out); release(pc);
}

int main()
{
    const int h = 240, w = 320;
    run_shape<2, 2>("3x3 19->19 d1", 19, 19, 3, 1, h, w);
    run_shape<1, 2>("3x3 19->19 d1", 19, 19, 3, 1, h, w);
    run_shape<2, 2>("3x3 95->19 d1", 95, 19, 3, 1, h, w);
    run_shape<1, 2>("3x3 95->19 d1", 95, 19, 3, 1, h, w);
    run_shape<1, 8>("1x1 464->116", 464, 116, 1, 1, h, w);
    run_shape<2, 8>("1x1 464->116", 464, 116, 1, 1, h, w);
    run_shape<1, 8>("1x1 20->116", 20, 116, 1, 1, h, w);
    run_shape<1, 6>("1x1 116->96", 116, 96, 1, 1, h, w);
    return 0;
}
