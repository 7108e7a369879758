// Profiling-only microbenchmark (not product, not a test): the five dense Blocks of a head as five dense_pair_kernel
// launches (fp32 planes) against ONE dense_chain_kernel launch (split planes): max |difference| per Block and us per chain.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DOJF_CHAIN_TIMING] -c tools/microbench/chain_bench.hip -o /tmp/cb.o
//        hipcc --offload-arch=gfx950 /tmp/cb.o online_joint_depthfusion_and_semantic_amd/csrc/ojf_api.o -o tools/microbench/chain_bench.bin
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../online_joint_depthfusion_and_semantic_amd/csrc/ojf_net.hip"

using namespace ojf;

static void to_split(const float *v4, uint16_t *dst8)  // host split_pack4
{
    for (int j = 0; j < 4; ++j) {
        const _Float16 hi = (_Float16)v4[j];
        const _Float16 lo = (_Float16)(v4[j] - (float)hi);
        dst8[j] = __builtin_bit_cast(uint16_t, hi);
        dst8[4 + j] = __builtin_bit_cast(uint16_t, lo);
    }
}
static float from_split(const uint16_t *s8, int j)
{
    return (float)__builtin_bit_cast(_Float16, s8[j]) + (float)__builtin_bit_cast(_Float16, s8[4 + j]);
}

int main(int argc, char **argv)
{
    const int h = argc > 2 ? atoi(argv[1]) : 240, w = argc > 2 ? atoi(argv[2]) : 320, npix = h * w, c = 19, cs = 20, gf = 5;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *X, *XS;
    alloc_planes(&X, npix, (gf + 1) * cs);
    alloc_planes(&XS, npix, (gf + 1) * cs);
    std::vector<float> hx((size_t)npix * cs, 0.0f);
    for (int q = 0; q < 5; ++q)
        for (int p = 0; p < npix; ++p)
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < c) hx[((size_t)q * npix + p) * 4 + j] = nd(rng);
    std::vector<float> hs(hx.size());
    for (size_t i = 0; i < hx.size(); i += 4) to_split(&hx[i], reinterpret_cast<uint16_t *>(&hs[i]));
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(XS, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
#ifdef OJF_CHAIN_TIMING
    (void)hipMalloc(&g_chain_dbg, 128 * sizeof(long long) + kChainStepF4 * 16);
    (void)hipMemset(g_chain_dbg, 0, 96 * sizeof(long long));
#endif
    std::vector<PackedPair> pairs(gf);
    std::vector<ConvBuilder> bas, bbs;
    for (int i = 0; i < gf; ++i) {
        std::vector<float> wa((size_t)c * (i + 1) * c * 9), wb((size_t)c * c * 9), bs(c, 0.01f);
        for (auto &v : wa) v = nd(rng) * std::sqrt(2.0f / ((i + 1) * c * 9));
        for (auto &v : wb) v = nd(rng) * std::sqrt(2.0f / (c * 9));
        if (getenv("CHAIN_DBG_ZERO16"))  // debugging aid: the second convolution ignores its input channels 16..18
            for (int o = 0; o < c; ++o)
                for (int ci = 16; ci < c; ++ci)
                    for (int t = 0; t < 9; ++t) wb[((size_t)o * c + ci) * 9 + t] = 0.0f;
        ojf_conv_layer la{(i + 1) * c, c, 3, 1, wa.data(), bs.data()}, lb{c, c, 3, 1, wb.data(), bs.data()};
        ConvBuilder ba((i + 1) * cs, cs, 3, 1), bb(cs, cs, 3, 1);
        ba.add(la, 0, (i + 1) * c, slot_map((i + 1) * c, c, cs), 0, true);
        bb.add(lb, 0, c, slot_map(c, c, cs), 0, true);
        if (finish_pair(ba, bb, pairs[i], pair_cfg_for(h, w, cs))) { printf("pack failed: %s\n", ojf_last_error()); return 1; }
        bas.push_back(ba);
        bbs.push_back(bb);
    }
    PackedChain pc;
    if (finish_chain(bas, bbs, pc)) { printf("chain pack failed: %s\n", ojf_last_error()); return 1; }

    auto run_pairs = [&]() {
        for (int i = 0; i < gf; ++i) launch_pair(pairs[i], X, 0, X, (i + 1) * (cs / 4), h, w, 0);
    };
    auto run_chain = [&]() { return launch_chain(pc, XS, h, w, 0); };
    run_pairs();
    if (run_chain()) { printf("chain launch failed: %s\n", ojf_last_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
    {
        int sync[4];
        (void)hipMemcpy(sync, pc.sync, sizeof(sync), hipMemcpyDeviceToHost);
        printf("sync: epoch %d done %d err %d\n", sync[0], sync[1], sync[2]);
    }
#ifdef OJF_CHAIN_TIMING
    {
        std::vector<float> got((size_t)kChainStepF4 * 4), ref(got.size());
        (void)hipMemcpy(got.data(), g_chain_dbg + 128, got.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ref.data(), pc.w + (size_t)kChainStepF4 * 4, ref.size() * 4, hipMemcpyDeviceToHost);
        for (int pcs = 0; pcs < kChainStepF4 / 64; ++pcs) {
            int diff = 0;
            for (int i = 0; i < 256; ++i) diff += memcmp(&got[(size_t)pcs * 256 + i], &ref[(size_t)pcs * 256 + i], 4) != 0;
            printf("conv b weights of Block 0, piece %d: %d of 256 words differ\n", pcs, diff);
        }
    }
#endif
    std::vector<float> ya((size_t)npix * cs * gf), yb(ya.size());
    (void)hipMemcpy(ya.data(), X + (size_t)npix * cs, ya.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(yb.data(), XS + (size_t)npix * cs, yb.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < gf; ++i) {
        double md = 0, mx = 0;
        size_t bad = 0;
        for (int q = 0; q < 5; ++q)
            for (int p = 0; p < npix; ++p) {
                const size_t idx = (((size_t)i * 5 + q) * npix + p) * 4;
                for (int j = 0; j < 4; ++j) {
                    const float ref = ya[idx + j], got = from_split(reinterpret_cast<const uint16_t *>(&yb[idx]), j);
                    const double d = std::fabs((double)ref - got);
                    if (!(d <= 1e-4)) ++bad;
                    md = std::fmax(md, d);
                    mx = std::fmax(mx, std::fabs((double)ref));
                }
            }
        printf("Block %d: max |pairs - chain| %.3e (max |value| %.2f), %zu values beyond 1e-4\n", i, md, mx, bad);
        if (bad && getenv("CHAIN_DBG_VAL")) {
            for (int x = 14; x < 26; ++x) {
                printf("x %d:", x);
                for (int q = 0; q < 5; ++q) {
                    const size_t idx = (((size_t)i * 5 + q) * npix + (size_t)1 * w + x) * 4;
                    for (int j = 0; j < 4; ++j) printf(" %.4f/%.4f", ya[idx + j], from_split(reinterpret_cast<const uint16_t *>(&yb[idx]), j));
                }
                printf("\n");
            }
        }
        if (bad && getenv("CHAIN_DBG_MAP")) {  // debugging aid: which pixels of channel 16 are wrong (one character per pixel)
            for (int y = 0; y < 40; ++y) {
                for (int x = 0; x < 80; ++x) {
                    const size_t idx = (((size_t)i * 5 + 4) * npix + (size_t)y * w + x) * 4;
                    const double d = std::fabs((double)ya[idx] - from_split(reinterpret_cast<const uint16_t *>(&yb[idx]), 0));
                    putchar(d <= 1e-4 ? '.' : '#');
                }
                putchar('\n');
            }
        }
        if (bad)
            for (int q = 0; q < 5; ++q)
                for (int j = 0; j < 4; ++j) {
                    double mq = 0;
                    for (int p = 0; p < npix; ++p) {
                        const size_t idx = (((size_t)i * 5 + q) * npix + p) * 4;
                        mq = std::fmax(mq, std::fabs((double)ya[idx + j] - from_split(reinterpret_cast<const uint16_t *>(&yb[idx]), j)));
                    }
                    printf("  channel %d: %.3e%s", 4 * q + j, mq, (4 * q + j) % 5 == 4 ? "\n" : "");
                }
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int k = 0; k < 2; ++k) {
        for (int r = 0; r < 5; ++r) k ? (void)run_chain() : run_pairs();
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) k ? (void)run_chain() : run_pairs();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per chain of %d Blocks\n", k ? "dense_chain_kernel (1 launch)" : "dense_pair_kernel x 5", ms * 1e3f / reps, gf);
    }
#ifdef OJF_CHAIN_TIMING
    long long st[96];
    (void)hipMemcpy(st, g_chain_dbg, sizeof(st), hipMemcpyDeviceToHost);
    printf("chain stamps of tile 37 (cycles from start):");
    for (int k = 1; k < 96 && st[k]; ++k) printf(" %lld", st[k] - st[0]);
    printf("\n");
#endif
    {
        int sync[4];
        (void)hipMemcpy(sync, pc.sync, sizeof(sync), hipMemcpyDeviceToHost);
        printf("sync: epoch %d done %d err %d\n", sync[0], sync[1], sync[2]);
    }
    return 0;
}
