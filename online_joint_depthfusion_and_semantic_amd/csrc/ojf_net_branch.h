// The four branches of a VortexPooling (modules/model.py:119-134: ... 3x3 dilation r -> BN -> ReLU -> 3x3 dilation r -> BN ->
// ReLU ..., r = 1, 3, 9, 27) - both dilated 3x3 of every branch in ONE persistent launch on split-fp16 MFMA (included by
// ojf_net.hip only).  Round 4; replaces the two grouped conv_f16x3_kernel launches of a VortexPooling.
//
// A convolution with dilation r never mixes pixels of different residues (x mod r, y mod r): on each of the r x r
// SUB-IMAGES (pixels (px + r sx, py + r sy)) it is a plain 3x3.  So every branch runs through the LDS-resident pair of
// ojf_net_chain.h - window by LDS-DMA from the split planes its producer left (entry GEMM / pool pyramid), first 3x3 into
// the LDS planes T, second 3x3 from T, packed weight rows (five MFMAs per product block) - with a 2-PIXEL halo in
// sub-image coordinates whatever the dilation, and the intermediate of a branch never leaves the CU.  (The grouped
// launches fetched every operand from L1 once per tap, 230 MB per launch, and wrote / re-read the intermediate planes.)
//
// Work item = (branch, sub-image(s), tile), from a table the host builds once per net.  Kinds (slot regions as in
// ojf_net_chain.h, pitch PW = TW + 4):
//   kind 0: 20 x 16 outputs of one sub-image (r = 1, 3 and whatever does not fit the other kinds);
//   kind 1: 20 x 14 (r = 9 at 320 x 240: sub-images of 36 x 27 are 2 x 2 such tiles);
//   kind 2: two WHOLE sub-images of at most 12 x 9 stacked in one region (r = 27 at 320 x 240) - outside a sub-image
//           everything is zero, so the rows between the two serve as the lower halo of one and the upper halo of the other.
// The table is sorted by kind and dealt to the blocks round robin (block b: items b, b + grid, ...); the phases of one
// tile sit 8 items apart so that - blocks land on XCD b % 8 - the r x r items that share the lines of one image region
// (16 bytes of every 16 r) gather from and scatter into ONE L2.
#pragma once

namespace ojf {

constexpr int kBranchLoaders = 2;  // waves of a block that only fetch (see branch_run)
template <int KIND_, int TW_, int TH_, int NST_, int WAVES_>
struct BranchGeom {
    static constexpr int KIND = KIND_, TW = TW_, TH = TH_, NST = NST_, WAVES = WAVES_, THREADS = 64 * WAVES_;
    static constexpr int CW = WAVES_ - kBranchLoaders;  // waves that compute
    static constexpr int PW = TW + 4;
    static constexpr int WR = NST * (TH + 2) + 2;  // window rows (stacked sub-images share their zero halo rows)
    static constexpr int XS = WR * PW, TS = (WR - 2) * PW, OS = (WR - 4) * PW;
    static constexpr int TILES_A = (TS + 15) / 16, TILES_B = (OS + 15) / 16;
    static constexpr int MT_A = (TILES_A + CW - 1) / CW, MT_B = (TILES_B + CW - 1) / CW;
    static constexpr int XP = pair_round16(pair_max(XS, TILES_A * 16 + 2 * PW + 3));  // plane pitch (window and T)
    static constexpr int NPIECE = (kChainNG * XP + 63) / 64;
    static constexpr int X_F4 = NPIECE * 64;
    static constexpr int NPX = (NPIECE + WAVES - 1) / WAVES;
};
constexpr int kBranchWaves = 16;
using BranchK0 = BranchGeom<0, 20, 16, 1, kBranchWaves>;
using BranchK1 = BranchGeom<1, 20, 14, 1, kBranchWaves>;
using BranchK2 = BranchGeom<2, 12, 9, 2, kBranchWaves>;
constexpr int kBranchXF4 = pair_max(BranchK0::X_F4, pair_max(BranchK1::X_F4, BranchK2::X_F4));
constexpr int kBranchNPW = (kChainStepF4 / 64 + kBranchWaves - 1) / kBranchWaves;
constexpr size_t kBranchLdsBytes = (size_t)(3 * kBranchXF4 + 2 * kChainStepF4) * 16 + 4 * 128 * sizeof(float);

// item: x = branch | kind << 4, y = px | py << 16 of the first sub-image, z = the same of the second one (kind 2; -1: none),
//       w = sx0 | sy0 << 16 (tile origin in sub-image coordinates)
struct BranchArgs {
    const f32x4 *in[4];  // split planes of the branch inputs (5 channel groups each; float4 -1 is zero)
    f32x4 *out;          // fp32 planes: branch br = groups [5 br, 5 br + 5)
    const f32x4 *w;      // steps of ojf_net_chain.h: branch br = [2 br] first, [2 br + 1] second 3x3
    const float *vec;    // per branch: bias_a | rinv_a | bias_b | rinv_b, 32 floats each
    const int4 *items;
    int n_items;
    int dil[4];
    int h, w_img, npix;
    int *ovf;            // split-fp16 range guard flag
};

struct BranchItem {  // (scalars)
    int br, d, px0, py0, px1, py1, sx0, sy0;
};
__device__ __forceinline__ BranchItem branch_item(const BranchArgs &a, const int4 &it)
{
    BranchItem r;
    r.br = __builtin_amdgcn_readfirstlane(it.x) & 3;
    r.d = a.dil[r.br];
    const int y = __builtin_amdgcn_readfirstlane(it.y), z = __builtin_amdgcn_readfirstlane(it.z), w = __builtin_amdgcn_readfirstlane(it.w);
    r.px0 = y & 0xffff; r.py0 = y >> 16;
    r.px1 = z < 0 ? -1 : (z & 0xffff); r.py1 = z < 0 ? (1 << 20) : (z >> 16);  // (absent: below every image)
    r.sx0 = w & 0xffff; r.sy0 = w >> 16;
    return r;
}

// window row `row`, column `col` of an item's region -> pixel index, or -1 (outside the sub-image / the image)
template <class G>
__device__ __forceinline__ int branch_pixel(const BranchArgs &a, const BranchItem &it, int row, int col)
{
    int k = 0;
    if (G::NST > 1) k = row >= G::TH + 2 ? 1 : 0;  // (NST <= 2)
    const int lr = row - k * (G::TH + 2) - 2, lc = col - 2;
    const int px = k ? it.px1 : it.px0, py = k ? it.py1 : it.py0;
    const int sx = it.sx0 + lc, sy = it.sy0 + lr;
    const int gx = px + it.d * sx, gy = py + it.d * sy;
    bool ok = sx >= 0 && sy >= 0 && gx < a.w_img && gy < a.h && px >= 0;
    if (G::NST > 1) ok = ok && (unsigned)lr < (unsigned)G::TH;
    return ok ? gy * a.w_img + gx : -1;
}

template <class G>
__device__ __forceinline__ void branch_load_x(const BranchArgs &a, const BranchItem &it, f32x4 *xbuf, int wave, int lane)
{
    const f32x4 *base = a.in[it.br];
#pragma unroll
    for (int j = 0; j < G::NPX; ++j) {
        const int pc = wave + G::WAVES * j;
        if (pc < G::NPIECE) {
            const int e = pc * 64 + lane;
            const int q = e / G::XP, sl = e - q * G::XP;
            const int ry = sl / G::PW, rx = sl - ry * G::PW;
            int p = branch_pixel<G>(a, it, ry, rx);
            if (q >= kChainNG || sl >= G::XS) p = -1;
            const f32x4 *src = p >= 0 ? base + (q * a.npix + p) : base - 1;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                             (void __attribute__((address_space(3))) *)(xbuf + pc * 64), 16, 0, 0);
        }
    }
}
// A loader wave's share of item `it`'s window, through registers (plain loads, then ds_write) - not by LDS-DMA: a window
// was written by the previous kernel on other XCDs, it comes from beyond the L2 (~2.5 us under this load), and a wave is
// held at its next global_load_lds while a few earlier ones are outstanding (measured per piece: 130 cycles for weights,
// which hit the L2, 600-1500 for windows); plain loads are asked for and forgotten until they are used.
// A loader shares its SIMD with waves that issue MFMAs and gets an instruction through every ~10 cycles, so a load must
// cost a handful of them: one load = RPP whole rows of one plane (lane = row * PW + column, lanes beyond RPP * PW idle),
// what depends on the lane alone is computed once per item; per load there remain an add, a compare and the address.
// Load (q, j) = rows [j RPP, (j + 1) RPP) of plane q; loader lw takes those with (q NRG + j) % kBranchLoaders == lw,
// batch 0 / 1 = the first / second half of its list.
template <class G>
struct BranchRows {
    static constexpr int RPP = 64 / G::PW, NRG = (G::WR + RPP - 1) / RPP, NLOAD = kChainNG * NRG;
    static constexpr int PER = (NLOAD + kBranchLoaders - 1) / kBranchLoaders, HALF = (PER + 1) / 2;  // loads per loader / per batch
};
template <class G>
__device__ __forceinline__ void branch_rows_load(const BranchArgs &a, const BranchItem &it, int lw, int lane, int batch,
                                                 f32x4 (&stg)[BranchRows<G>::HALF])
{
    using R = BranchRows<G>;
    const f32x4 *base = a.in[it.br];
    const int lrow = lane / G::PW, col = lane - lrow * G::PW;
#pragma unroll
    for (int n = 0; n < R::HALF; ++n) {
        // (n-th load of this batch: index lw + kBranchLoaders (n + batch HALF) of the (q, j) list - q and j are wave-uniform)
        const int idx = lw + kBranchLoaders * (n + batch * R::HALF);
        const int q = idx / R::NRG, j = idx - q * R::NRG;
        const int row = j * R::RPP + lrow;
        const int kk = G::NST > 1 ? (row >= G::TH + 2 ? 1 : 0) : 0;
        const int lr = row - kk * (G::TH + 2) - 2;
        const int px = kk ? it.px1 : it.px0, py = kk ? it.py1 : it.py0;
        const int gx = px + it.d * (it.sx0 + col - 2), gy = py + it.d * (it.sy0 + lr);
        // (a column / row in front of the sub-image gives a negative coordinate: px < d)
        const bool ok = idx < R::NLOAD && lrow < R::RPP && row < G::WR && (G::NST == 1 || (unsigned)lr < (unsigned)G::TH) &&
                        (unsigned)gx < (unsigned)a.w_img && (unsigned)gy < (unsigned)a.h;
        // (an address select, not a value select: nothing may depend on the loaded value before it is stored)
        const int off = ok ? q * a.npix + gy * a.w_img + gx : -1;
        stg[n] = base[off];
    }
}
template <class G>
__device__ __forceinline__ void branch_rows_store(f32x4 *xbuf, int lw, int lane, int batch, const f32x4 (&stg)[BranchRows<G>::HALF])
{
    using R = BranchRows<G>;
#pragma unroll
    for (int n = 0; n < R::HALF; ++n) {
        const int idx = lw + kBranchLoaders * (n + batch * R::HALF);
        const int q = idx / R::NRG, j = idx - q * R::NRG;
        if (idx < R::NLOAD && lane < R::RPP * G::PW && j * R::RPP * G::PW + lane < G::XS) xbuf[q * G::XP + j * R::RPP * G::PW + lane] = stg[n];
    }
}
// pieces p0, p0 + stride, ... < p1 of a window by LDS-DMA with the full arithmetic (the first item of the next kind)
template <class G>
__device__ __forceinline__ void branch_load_part(const BranchArgs &a, const BranchItem &it, f32x4 *xbuf, int p0, int p1, int stride, int lane)
{
    const f32x4 *base = a.in[it.br];
    for (int pc = p0; pc < p1; pc += stride) {
        const int e = pc * 64 + lane;
        const int q = e / G::XP, sl = e - q * G::XP;
        const int ry = sl / G::PW, rx = sl - ry * G::PW;
        int p = branch_pixel<G>(a, it, ry, rx);
        if (q >= kChainNG || sl >= G::XS) p = -1;
        const f32x4 *src = p >= 0 ? base + (q * a.npix + p) : base - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                         (void __attribute__((address_space(3))) *)(xbuf + pc * 64), 16, 0, 0);
    }
}
__device__ __forceinline__ void branch_issue_w_part(const BranchArgs &a, int step, f32x4 *wbuf, int lw, int lane)
{
    const f32x4 *src = a.w + (size_t)step * kChainStepF4;
    for (int pc = lw; pc < kChainStepF4 / 64; pc += kBranchLoaders)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pc * 64 + lane),
                                         (void __attribute__((address_space(3))) *)(wbuf + pc * 64), 16, 0, 0);
}
__device__ __forceinline__ void branch_issue_w(const BranchArgs &a, int step, f32x4 *wbuf, int wave, int lane)
{
    const f32x4 *src = a.w + (size_t)step * kChainStepF4;
#pragma unroll
    for (int j = 0; j < kBranchNPW; ++j) {
        const int pc = wave + kBranchWaves * j;
        if (pc < kChainStepF4 / 64)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pc * 64 + lane),
                                             (void __attribute__((address_space(3))) *)(wbuf + pc * 64), 16, 0, 0);
    }
}
// the window and the first convolution's weights of item `it4` (whatever its kind)
__device__ __forceinline__ void branch_fetch(const BranchArgs &a, const int4 &it4, f32x4 *xbuf, f32x4 *wbuf, int wave, int lane)
{
    const BranchItem it = branch_item(a, it4);
    const int kind = __builtin_amdgcn_readfirstlane(it4.x) >> 4;
    if (kind == 0) branch_load_x<BranchK0>(a, it, xbuf, wave, lane);
    else if (kind == 1) branch_load_x<BranchK1>(a, it, xbuf, wave, lane);
    else branch_load_x<BranchK2>(a, it, xbuf, wave, lane);
    branch_issue_w(a, 2 * it.br, wbuf, wave, lane);
}

// the items of kind G::KIND from index i on (the table is sorted by kind); returns the first index it did not take.
// cur4 / nxt4: the table entries of items i and i + gridDim.x; wbr: the branch whose second 3x3's weights sit in wl[1]
// (weights are fetched when the branch changes).
//
// Who fetches.  Whatever a wave that computes asks for costs matrix-pipe time: its third global_load_lds holds it until
// earlier ones have returned (a window comes from beyond the L2, ~2.5 us under this load: 1600-3300 cycles per item
// measured with every wave fetching its share), and plain loads in flight across a K loop slow it as much.  So the last
// kBranchLoaders waves of a block do nothing else: behind the first barrier of an item they ask for the second 3x3's
// weights (LDS-DMA: in L2, 130 cycles apiece) and for ALL their pieces of the next item's window as plain loads into
// registers (a loader has the register file of a wave to itself: 20 x 16 bytes per lane in flight), behind the second
// barrier for the next item's first weights; the window goes to LDS at the end of the item - two phases after it was
// asked for.  The other CW waves run the MFMAs (27 / 24 pixel tiles over 14 waves: still at most two per wave).
template <class G>
__device__ __forceinline__ int branch_run(const BranchArgs &a, int i, int iend, int4 &cur4, int4 &nxt4, int &wbr, f32x4 *wl, f32x4 *xl, f32x4 *tl,
                                          const float *vl, int &xb, float &gmax, const __amdgpu_buffer_rsrc_t &ors)
{
    constexpr int PW = G::PW, XP = G::XP, TSTRIDE = G::CW * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const bool loader = wave >= G::CW;
    const int lw = wave - G::CW;
    const int mt_a = (G::TILES_A - wave + G::CW - 1) / G::CW, mt_b = (G::TILES_B - wave + G::CW - 1) / G::CW;
    int ls = wave * 16 + i16;
    int t9[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int tap = 0;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            if (g == gg) tap = chain_read_tap(9 + r, gg);
        t9[r] = (tap / 3) * PW + tap % 3;
    }
#ifdef OJF_BRANCH_TIMING
    long long st[9];
    int printed = 0;
#define OJF_BSTAMP(k) st[k] = (long long)__builtin_amdgcn_s_memtime()
#else
#define OJF_BSTAMP(k) do {} while (0)
#endif
    if (loader) {
        // ---- a loader wave's item loop ---------------------------------------------------------------------------------
        // (first in line at its SIMD's issue port: beside three waves that issue MFMAs it got an instruction through every ~50
        // cycles, 6000 cycles for the dozen loads and nine LDS-DMA of a phase)
        __builtin_amdgcn_s_setprio(3);
        while (i < iend) {
            OJF_BSTAMP(0);
            const int4 it4 = cur4;
            if ((__builtin_amdgcn_readfirstlane(it4.x) >> 4) != G::KIND) break;
            const int br = __builtin_amdgcn_readfirstlane(it4.x) & 3;
            const int nxt = i + (int)gridDim.x;
            const bool more = nxt < iend;
            const bool same = more && (__builtin_amdgcn_readfirstlane(nxt4.x) >> 4) == G::KIND;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            OJF_BSTAMP(1);
            int4 nn4 = nxt4;
            if (nxt + (int)gridDim.x < iend) nn4 = a.items[nxt + (int)gridDim.x];
            using R = BranchRows<G>;
            f32x4 stg[R::HALF];
            const BranchItem ni = branch_item(a, nxt4);
            const bool need_b = wbr != br;  // (the second 3x3's weights of another branch are in wl[1])
            wbr = br;
            const int nbr = __builtin_amdgcn_readfirstlane(nxt4.x) & 3;
            if (same) {
                branch_rows_load<G>(a, ni, lw, lane, 0, stg);
                if (need_b) branch_issue_w_part(a, 2 * br + 1, wl + kChainStepF4, lw, lane);
                OJF_BSTAMP(2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                branch_rows_store<G>(xl + (xb ^ 1) * kBranchXF4, lw, lane, 0, stg);
            } else {
                if (need_b) branch_issue_w_part(a, 2 * br + 1, wl + kChainStepF4, lw, lane);
                if (more) {
                    const int kind = __builtin_amdgcn_readfirstlane(nxt4.x) >> 4;  // (the next kind's first item: by LDS-DMA)
                    if (kind == 0) branch_load_part<BranchK0>(a, ni, xl + (xb ^ 1) * kBranchXF4, lw, BranchK0::NPIECE, kBranchLoaders, lane);
                    else if (kind == 1) branch_load_part<BranchK1>(a, ni, xl + (xb ^ 1) * kBranchXF4, lw, BranchK1::NPIECE, kBranchLoaders, lane);
                    else branch_load_part<BranchK2>(a, ni, xl + (xb ^ 1) * kBranchXF4, lw, BranchK2::NPIECE, kBranchLoaders, lane);
                }
                OJF_BSTAMP(2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            OJF_BSTAMP(3);
            __syncthreads();
            OJF_BSTAMP(4);
            if (same) branch_rows_load<G>(a, ni, lw, lane, 1, stg);
            if (more && nbr != br) branch_issue_w_part(a, 2 * nbr, wl, lw, lane);
            OJF_BSTAMP(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            OJF_BSTAMP(6);
            if (same) branch_rows_store<G>(xl + (xb ^ 1) * kBranchXF4, lw, lane, 1, stg);
            OJF_BSTAMP(7);
#ifdef OJF_BRANCH_TIMING
            if ((blockIdx.x == 37 || blockIdx.x == 200) && tid == 960 && printed < 2) {
                ++printed;
                printf("branch item blk %d LOADER kind %d br %d: top %lld ask %lld weights %lld barrier %lld w-next %lld window %lld store %lld\n", (int)blockIdx.x, G::KIND, br,
                       st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
            }
#endif
            xb ^= 1;
            i = nxt;
            cur4 = nxt4;
            nxt4 = nn4;
        }
        return i;
    }
    // ---- a computing wave's item loop -------------------------------------------------------------------------------------
    while (i < iend) {
        OJF_BSTAMP(0);
        const int4 it4 = cur4;
        if ((__builtin_amdgcn_readfirstlane(it4.x) >> 4) != G::KIND) break;
        const BranchItem it = branch_item(a, it4);
        const int nxt = i + (int)gridDim.x;
        // ---- first 3x3: window (this buffer) and weights (wl[0]) were fetched during the previous item -----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        OJF_BSTAMP(1);
        // (the table entry after the next: asked for two items ahead, a load in front of its use would wait a memory round trip)
        int4 nn4 = nxt4;
        if (nxt + (int)gridDim.x < iend) nn4 = a.items[nxt + (int)gridDim.x];
        f32x4 acc[G::MT_A][3];
#pragma unroll
        for (int m = 0; m < G::MT_A; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            asm volatile("" : "+v"(ls));
            const f32x4 *xbuf = xl + xb * kBranchXF4 + ls, *wl_lane = wl + lane;
            chain_mac<G::MT_A, PW, TSTRIDE>(acc, xbuf + g * XP, xbuf + 4 * XP, wl_lane, mt_a, t9, []() {});
        }
        OJF_BSTAMP(2);
        {   // bias, ReLU, zero outside the sub-image, split, into the T planes
            const float *v = vl + it.br * 128;
            const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
#pragma unroll
            for (int m = 0; m < G::MT_A; ++m) {
                if (m >= mt_a) continue;
                const int s = ls + TSTRIDE * m;
                const int ry = s / PW, rx = s - ry * PW;
                const bool ok = s < G::TS && rx < G::TW + 2 && branch_pixel<G>(a, it, ry + 1, rx + 1) >= 0;
                f32x4 main, extra;
                chain_unpack(acc[m], main, extra);
                auto put_t = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                    const f32x4 lin = fma4(raw, r4, b4);
                    if (ok) gmax = guard_max(gmax, lin);
                    f32x4 val;
#pragma unroll
                    for (int j = 0; j < 4; ++j) val[j] = ok ? leaky_max(lin[j], 0.0f) : 0.0f;
                    tl[og * XP + s] = split_pack4(val);
                };
                put_t(g, main, rm, bm);
                if (g == 0) put_t(4, extra, re, be);
            }
        }
        OJF_BSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        OJF_BSTAMP(4);
        // ---- second 3x3 from T ----------------------------------------------------------------------------------------
        OJF_BSTAMP(5);
        f32x4 accb[G::MT_B][3];
#pragma unroll
        for (int m = 0; m < G::MT_B; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) accb[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            asm volatile("" : "+v"(ls));
            const f32x4 *wl_lane = wl + kChainStepF4 + lane, *tbuf = tl + ls;
            chain_mac<G::MT_B, PW, TSTRIDE>(accb, tbuf + g * XP, tbuf + 4 * XP, wl_lane, mt_b, t9, []() {});
        }
        OJF_BSTAMP(6);
        {
            const float *v = vl + it.br * 128 + 64;
            const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
#pragma unroll
            for (int m = 0; m < G::MT_B; ++m) {
                if (m >= mt_b) continue;
                const int s = ls + TSTRIDE * m;
                const int oy = s / PW, ox = s - oy * PW;
                const int p = (s < G::OS && ox < G::TW) ? branch_pixel<G>(a, it, oy + 2, ox + 2) : -1;
                f32x4 main, extra;
                chain_unpack(accb[m], main, extra);
                auto put_o = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                    if (p < 0) return;
                    const f32x4 lin = fma4(raw, r4, b4);
                    gmax = guard_max(gmax, lin);
                    const f32x4 val = leaky_max4(lin, 0.0f);
                    const unsigned off = (unsigned)((((it.br * kChainNG) + og) * a.npix + p) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ors, off, 0, 0);
                };
                put_o(g, main, rm, bm);
                if (g == 0) put_o(4, extra, re, be);
            }
        }
        OJF_BSTAMP(7);
#ifdef OJF_BRANCH_TIMING
        if ((blockIdx.x == 37 || blockIdx.x == 200) && tid == 0 && printed < 2) {
            ++printed;
            printf("branch item blk %d tid %d kind %d br %d: wait %lld mac_a %lld epi_a %lld wait %lld fetch %lld mac_b %lld epi_b %lld\n", (int)blockIdx.x, tid, G::KIND, it.br,
                   st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
        }
#endif
        xb ^= 1;
        i = nxt;
        cur4 = nxt4;
        nxt4 = nn4;
    }
#undef OJF_BSTAMP
    return i;
}

__global__ __launch_bounds__(64 * kBranchWaves, kBranchWaves / 4) void vortex_branch_kernel(const BranchArgs a)
{
    extern __shared__ f32x4 branch_lds[];
    // (LDS-DMA destinations first; T and the vectors, written by ds_write, behind them)
    f32x4 *wl = branch_lds;                                   // [2][kChainStepF4]: first | second 3x3 of the current item
    f32x4 *xl = branch_lds + 2 * kChainStepF4;                // [2][kBranchXF4]: windows (5 planes of XP slots)
    f32x4 *tl = xl + 2 * kBranchXF4;                          // [kBranchXF4]: the intermediate T
    float *vl = reinterpret_cast<float *>(tl + kBranchXF4);   // [4][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the block's share of the table: items blockIdx.x, blockIdx.x + gridDim.x, ... (see the header: blocks b, b + 8, ... share an L2)
    int i = blockIdx.x;
    const int iend = a.n_items;
    int4 cur4 = int4{0, 0, 0, 0}, nxt4 = int4{0, 0, 0, 0};  // items[i], items[i + gridDim.x]
    if (i < iend) {
        cur4 = a.items[i];
        if (i + (int)gridDim.x < iend) nxt4 = a.items[i + (int)gridDim.x];
        branch_fetch(a, cur4, xl, wl, wave, lane);
    }
    for (int k = tid; k < 4 * 32; k += 64 * kBranchWaves) reinterpret_cast<f32x4 *>(vl)[k] = reinterpret_cast<const f32x4 *>(a.vec)[k];
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 4 * kChainNG * a.npix * 16, 0x00020000);
    int xb = 0, wbr = -1;
    float gmax = 0.0f;
    i = branch_run<BranchK0>(a, i, iend, cur4, nxt4, wbr, wl, xl, tl, vl, xb, gmax, ors);
    i = branch_run<BranchK1>(a, i, iend, cur4, nxt4, wbr, wl, xl, tl, vl, xb, gmax, ors);
    i = branch_run<BranchK2>(a, i, iend, cur4, nxt4, wbr, wl, xl, tl, vl, xb, gmax, ors);
    if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
}


// ---- one dilated 3x3 of the four branches as small LDS-staged blocks (subconv_kernel) ---------------------------------------
// The same decomposition - dilation r = a plain 3x3 on each of the r x r sub-images - for ONE convolution per launch and
// four-wave blocks, several per CU: a block stages the window of its tile (halo 1 in sub-image coordinates) and the
// convolution's 18 KB of packed weights in LDS by LDS-DMA, then runs the K loop of ojf_net_chain.h on it.  Against
// conv_f16x3_kernel, which fetches every operand from L1 once per tap (24 KB per 32 pixels): 15 KB of window per ~112
// pixels, five MFMAs per product block instead of six, no tap table; what one block waits for, the other blocks of its CU
// cover (what the one-block-per-CU kernel above could not).  MEASURED SLOWER, opt-in (OJF_SUBCONV=1): 27 against 20 us per
// launch - the sub-images of r >= 3 are 16-byte granules 16 r bytes apart, so windows (8.9 us of a launch by ablation) and
// stores (6 us) move at a fraction of the rate of conv_f16x3_kernel's 256-byte runs; the K loop is 2 us.  Kinds = (pitch PW, rows TH) of the output region, usable
// columns TW = PW - 2: (32, 4) for r = 1, (16, 8) for r = 3, (40, 3) for 36-wide sub-images (r = 9 at 320 x 240),
// (16, 9) for 12 x 9 sub-images (r = 27) - every pitch a multiple of 8 (conflict-free reads of the fifth group's taps).
template <int KIND_, int PW_, int TH_>
struct SubGeom {
    static constexpr int KIND = KIND_, PW = PW_, TH = TH_, TW = PW_ - 2, WAVES = 4;
    static constexpr int XS = (TH + 2) * PW, OS = TH * PW;
    static constexpr int TILES = (OS + 15) / 16, MT = (TILES + WAVES - 1) / WAVES;
    static constexpr int XP = pair_round16(pair_max(XS, TILES * 16 + 2 * PW + 3));
    static constexpr int NPIECE = (kChainNG * XP + 63) / 64, X_F4 = NPIECE * 64;
    static constexpr int NPX = (NPIECE + WAVES - 1) / WAVES;
};
using SubK0 = SubGeom<0, 32, 4>;
using SubK1 = SubGeom<1, 16, 8>;
using SubK2 = SubGeom<2, 40, 3>;
using SubK3 = SubGeom<3, 16, 9>;
constexpr int kSubXF4 = pair_max(pair_max(SubK0::X_F4, SubK1::X_F4), pair_max(SubK2::X_F4, SubK3::X_F4));
constexpr size_t kSubLdsBytes = (size_t)(kSubXF4 + kChainStepF4) * 16;

struct SubArgs {
    const f32x4 *in[4];  // split planes of the four inputs (5 channel groups each; float4 -1 is zero)
    f32x4 *out[4];       // output planes of the four branches (5 groups each): split planes or fp32
    const f32x4 *w;      // steps of ojf_net_chain.h: branch br = step 2 br + second
    const float *vec;    // per branch: bias_a | rinv_a | bias_b | rinv_b, 32 floats each
    const int4 *items;   // x = branch | kind << 4, y = px | py << 16, w = sx0 | sy0 << 16
    int second;          // 0 / 1: the branches' first / second 3x3
    int out_split;
    int dil[4];
    int h, w_img, npix;
    int *ovf;
};

template <class G>
__device__ __forceinline__ void subconv_body(const SubArgs &a, const int4 &it4, f32x4 *wl, f32x4 *xl)
{
    constexpr int PW = G::PW, XP = G::XP, TSTRIDE = G::WAVES * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int br = __builtin_amdgcn_readfirstlane(it4.x) & 3, d = a.dil[br];
    const int yy = __builtin_amdgcn_readfirstlane(it4.y), ww = __builtin_amdgcn_readfirstlane(it4.w);
    const int px = yy & 0xffff, py = yy >> 16, sx0 = ww & 0xffff, sy0 = ww >> 16;
    // window slot (row, col) <-> sub-image (sx0 - 1 + col, sy0 - 1 + row); pixel index or -1
    auto pixel = [&](int row, int col) {
        const int gx = px + d * (sx0 - 1 + col), gy = py + d * (sy0 - 1 + row);
        // (a column / row in front of the sub-image gives a negative coordinate: px < d)
        return (unsigned)gx < (unsigned)a.w_img && (unsigned)gy < (unsigned)a.h ? gy * a.w_img + gx : -1;
    };
    {   // weights and window by LDS-DMA
        const f32x4 *src = a.w + (size_t)(2 * br + a.second) * kChainStepF4;
#pragma unroll
        for (int j = 0; j < (kChainStepF4 / 64 + G::WAVES - 1) / G::WAVES; ++j) {
            const int pc = wave + G::WAVES * j;
            if (pc < kChainStepF4 / 64)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pc * 64 + lane),
                                                 (void __attribute__((address_space(3))) *)(wl + pc * 64), 16, 0, 0);
        }
        const f32x4 *base = a.in[br];
#pragma unroll
        for (int j = 0; j < G::NPX; ++j) {
            const int pc = wave + G::WAVES * j;
            if (pc < G::NPIECE) {
                const int e = pc * 64 + lane;
                const int q = e / XP, sl = e - q * XP;
                const int ry = sl / PW, rx = sl - ry * PW;
                int p = pixel(ry, rx);
                if (q >= kChainNG || sl >= G::XS) p = -1;
                const f32x4 *s4 = p >= 0 ? base + (q * a.npix + p) : a.in[0] - 1;  // (in[0] is the start of a buffer: the zero float4 in front of it)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)s4,
                                                 (void __attribute__((address_space(3))) *)(xl + pc * 64), 16, 0, 0);
            }
        }
    }
    const int mt = (G::TILES - wave + G::WAVES - 1) / G::WAVES;
    const int ls = wave * 16 + i16;
    int t9[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int tap = 0;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            if (g == gg) tap = chain_read_tap(9 + r, gg);
        t9[r] = (tap / 3) * PW + tap % 3;
    }
    // bias / 1 / row scale of this lane's channels (in L2: every block reads the same 512 bytes)
    const float *v = a.vec + br * 128 + a.second * 64;
    const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
    f32x4 acc[G::MT][3];
#pragma unroll
    for (int m = 0; m < G::MT; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chain_mac<G::MT, PW, TSTRIDE>(acc, xl + ls + g * XP, xl + ls + 4 * XP, wl + lane, mt, t9, []() {});
    float gmax = 0.0f;
    f32x4 *out = a.out[br];
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
        if (m >= mt) continue;
        const int s = ls + TSTRIDE * m;
        const int oy = s / PW, ox = s - oy * PW;
        const int p = (s < G::OS && ox < G::TW) ? pixel(oy + 1, ox + 1) : -1;
        f32x4 main, extra;
        chain_unpack(acc[m], main, extra);
        auto put_o = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
            if (p < 0) return;
            const f32x4 lin = fma4(raw, r4, b4);
            gmax = guard_max(gmax, lin);
            const f32x4 val = leaky_max4(lin, 0.0f);
            out[og * a.npix + p] = a.out_split ? split_pack4(val) : val;
        };
        put_o(g, main, rm, bm);
        if (g == 0) put_o(4, extra, re, be);
    }
    if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
}

__global__ __launch_bounds__(256, 4) void subconv_kernel(const SubArgs a)
{
    extern __shared__ f32x4 sub_lds[];
    f32x4 *wl = sub_lds, *xl = sub_lds + kChainStepF4;
    const int4 it4 = a.items[blockIdx.x];
    const int kind = __builtin_amdgcn_readfirstlane(it4.x) >> 4;
    if (kind == 0) subconv_body<SubK0>(a, it4, wl, xl);
    else if (kind == 1) subconv_body<SubK1>(a, it4, wl, xl);
    else if (kind == 2) subconv_body<SubK2>(a, it4, wl, xl);
    else subconv_body<SubK3>(a, it4, wl, xl);
}

}  // namespace ojf
