// Fused 3x3 -> 3x3 convolution pair on split-fp16 MFMA (included by ojf_net.hip only).
//
// A dense Block of the fusion net (modules/model.py:4-21) is  slot[i+1] = act(conv3x3_b(act(conv3x3_a(slots 0..i)))):
// the intermediate T (19 channels) is consumed by nobody else.  The generic kernel ran the two convolutions as two
// launches with T round-tripping through HBM/L2 and fetched every input once per tap (the wide first convolutions were
// L1-bound: 265 MB through L1 for 95 -> 19).  Here ONE launch does both, LDS-resident:
//
//   block = 16 waves, one TW x TH output tile (20 x 16 at 320x240: 240 blocks = one round over the 256 CUs)
//   window   X: (TH+4) x (TW+4) input pixels, 32 channels (4 channel PAIRS of 8) at a time, fetched ONCE from global
//               memory, split into fp16 halves ONCE (not once per tap) and kept as hi / lo planes in LDS
//   conv a   over the (TH+2) x (TW+2) region the second convolution needs (1.3x recompute instead of a round trip),
//            MFMA B operands = one ds_read_b128 (hi) + one (lo) per (tap, channel pair); result -> bias, LeakyReLU,
//            zero outside the image (the second convolution's zero padding), split, into the T planes in LDS
//   conv b   from the T planes; epilogue writes the fp32 C4 planes of slot i+1.
//
// "Pitch-linear" slots: every region is addressed with the window's row pitch PW = TW + 4, slot s = row * PW + col.
// A 16-pixel MFMA tile is 16 consecutive slots (it may wrap over a row end; the (PW - TW - 2) junk columns cost 9 %
// more MFMAs at TW = 20), and a tap is a CONSTANT slot offset (dy * PW + dx): no per-lane index math, no bounds
// tests in the loop, 16 consecutive 16-byte LDS reads per lane group = conflict-free (plane lengths are multiples
// of 16 slots, so the four lane groups of a K block, which read four different channel pairs, stay on distinct banks).
//
// K blocks (32 wide) are built from units = (tap, channel pair): lane group g of K block S owns unit 4S + g.  A chunk
// with 4 pairs has 36 units = 9 K blocks (K block = tap, lane group = pair); the last chunk of a layer may hold
// 1..3 pairs (ceil(9 np / 4) K blocks, unit -> (tap, pair) through a 36-entry LDS table); conv b: 3 pairs, 7 K blocks.
// Weights ([K block][oc tile][hi|lo][lane] x 8 halfs, row-equilibrated like every split-fp16 layer) reach LDS per
// chunk; the next chunk's window and weights travel through registers while the current chunk computes.
#pragma once

namespace ojf {

constexpr int pair_round16(int x) { return (x + 15) / 16 * 16; }
constexpr int pair_max(int a, int b) { return a > b ? a : b; }

// Two shapes of the same kernel:
//   <TW, TH, 16, 4, false>  one 16-wave block per CU (143 KB of LDS at 20 x 16): four waves per SIMD take turns through the
//                           load / split / epilogue phases of the ONE block a CU holds (8 waves 101 us, 12 waves 95 us, 16
//                           waves 90 us per frame) - but all blocks of a launch fetch, compute and store in the same phases
//   <TW, TH, 8, CP, true>   round 4: 8-wave blocks small enough for TWO per CU (<= 80 KB: chunks of CP = 3 channel pairs,
//                           the T planes alias the window planes - the window is dead once conv a has finished), so that one
//                           block's fetch / split / barrier phases run beside the other's MFMA phase
//   <..., PACK = true>      round 4: FIVE MFMAs per (K block, pixel tile) instead of six.  The 19 (20 padded) output channels
//                           occupy 2 x 16 MFMA rows per weight half = 4 row tiles for the hi and lo halves, 41 % of them
//                           padding.  Packed, the 20 hi rows and the 20 lo rows are 40 consecutive rows = 3 row tiles:
//                           x_hi meets all three (w_hi x_hi and w_lo x_hi), x_lo the two that hold hi rows (w_hi x_lo;
//                           the lo rows riding along add w_lo x_lo, the term the split otherwise drops).  The epilogue
//                           adds the hi-row and lo-row accumulators of a channel group (they sit 5 lane groups apart).
//                           The K loops of this kernel run at ~89 % of the matrix pipe (s_memtime stamps,
//                           profiles/r04_pair_stamps.txt), so the sixth MFMA was the bound.
template <int TW, int TH, int WAVES_ = 16, int CP_ = 4, bool ALIAS_ = false, bool PACK_ = false>
struct PairGeom {
    static constexpr int WAVES = WAVES_, THREADS = 64 * WAVES, CP = CP_;
    static constexpr bool ALIAS = ALIAS_, PACK = PACK_;
    static constexpr int WT = PACK ? 3 : 4;               // weight float4 per lane and K block: 3 packed row tiles, or 2 x (hi, lo)
    static constexpr int NACC = PACK ? 3 : 2;             // accumulator tiles per pixel tile
    static constexpr int NKB = (9 * CP + 3) / 4;           // K blocks of a full chunk
    static constexpr int PW = TW + 4;                      // slot pitch = window width
    static constexpr int XS = (TH + 4) * PW;               // window slots
    static constexpr int TS = (TH + 2) * PW;               // slots of the intermediate
    static constexpr int OS = TH * PW;                     // output slots
    static constexpr int TILES_A = (TS + 15) / 16, TILES_B = (OS + 15) / 16;
    static constexpr int MT_A = (TILES_A + WAVES - 1) / WAVES, MT_B = (TILES_B + WAVES - 1) / WAVES;
    // plane lengths in slots: reads of junk columns / junk tiles must stay inside the plane
    static constexpr int XP = pair_round16(pair_max(XS, TILES_A * 16 + 2 * PW + 2));
    static constexpr int TP = pair_round16(pair_max(TILES_A * 16, TILES_B * 16 + 2 * PW + 2));
    // float4 counts of the three LDS areas (conv b always has 3 pairs = 7 K blocks of weights)
    static constexpr int X_F4 = CP * 2 * XP, T_F4 = 3 * 2 * TP, W_F4 = pair_max(NKB, 7) * 64 * WT;
    static constexpr int XT_F4 = ALIAS ? pair_max(X_F4, T_F4) : X_F4 + T_F4;
    static constexpr int NXI = (CP * XS + THREADS - 1) / THREADS;  // window items (pair, slot) per thread and chunk
    static constexpr int NWI = (W_F4 + THREADS - 1) / THREADS;     // weight float4 per thread and chunk
    static constexpr int NPRE = NXI > NWI ? NXI : NWI;
    static constexpr size_t LDS_BYTES = (size_t)(XT_F4 + W_F4) * 16 + 128 * sizeof(int) + 128 * sizeof(float);
    static constexpr int BLOCKS_PER_CU = LDS_BYTES <= 80 * 1024 && WAVES <= 8 ? 2 : 1;
};

struct PairArgs {
    const f32x4 *in;  // input planes (window = groups [in_g0, in_g0 + c4_in))
    f32x4 *out;       // output planes, groups [out_g0, out_g0 + og_store)
    const f32x4 *wa;  // conv a: chunks back to back, each [K block][2][hi|lo][lane]
    const f32x4 *wb;  // conv b: [ceil(9 np_b / 4)][2][hi|lo][lane]
    const float *bias_a, *rinv_a, *bias_b, *rinv_b;  // 32 floats each
    int in_g0, c4_in, out_g0, og_store;
    int h, w, npix, tiles_x;
    int n_chunks, np_last;  // chunks of CP channel pairs; pairs in the last chunk (1..CP)
    int np_b;               // channel pairs of the intermediate (1..3)
    int xcd_bands;          // 1: tile = xcd_band_block(blockIdx.x) (tuning switch)
    int *ovf;               // split-fp16 range guard flag
#ifdef OJF_PAIR_TIMING
    long long *dbg;         // profiling builds only (tools/microbench/pair_bench.hip): s_memtime stamps of block 0
#endif
};

// 8 values of one channel pair -> fp16 halves (same rounding as split_f16)
__device__ __forceinline__ void pair_split(const f32x4 &a, const f32x4 &b, f32x4 &hi, f32x4 &lo)
{
    f16x8 h, l;
    split_f16(a, b, h, l);
    hi = __builtin_bit_cast(f32x4, h);
    lo = __builtin_bit_cast(f32x4, l);
}

// acc[m][n] += W[K block][n] * act[unit(K block, g)][slot[m] + tap] over `nkb` K blocks
// `hook(S)` runs once per K block before its MFMAs: the caller trickles the next chunk's global loads through it (a
// burst of 13 loads per lane at the top of the loop stalled the in-order waves at issue until the memory queue drained)
template <int MT, int PLANE, int NACC, class Hook>
__device__ __forceinline__ void pair_mac(f32x4 (&acc)[MT][NACC], const f32x4 *act, const int *uo, int nkb, const f32x4 *wl,
                                         const int (&slot)[MT], int mt_wave, int lane, int g, Hook hook)
{
    for (int S = 0; S < nkb; ++S) {
        hook(S);
        const int off = uo[4 * S + g];
        if constexpr (NACC == 2) {
            f32x4 wh[2], wlo[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                wh[n] = wl[(S * 2 + n) * 128 + lane];
                wlo[n] = wl[(S * 2 + n) * 128 + 64 + lane];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < mt_wave) {  // wave-uniform
                    const f16x8 xh = __builtin_bit_cast(f16x8, act[off + slot[m]]);
                    const f16x8 xl = __builtin_bit_cast(f16x8, act[off + PLANE + slot[m]]);
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_f16x3(wh[n], wlo[n], xh, xl, acc[m][n]);
                }
            }
        } else {  // packed rows: tile 0 = hi rows 0..15, tile 1 = hi rows 16..19 | lo rows 0..11, tile 2 = lo rows 12..19
            f16x8 w[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) w[t] = __builtin_bit_cast(f16x8, wl[(S * 3 + t) * 64 + lane]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < mt_wave) {  // wave-uniform
                    const f16x8 xh = __builtin_bit_cast(f16x8, act[off + slot[m]]);
                    const f16x8 xl = __builtin_bit_cast(f16x8, act[off + PLANE + slot[m]]);
                    // (small terms first, like mfma_f16x3)
                    acc[m][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2], xh, acc[m][2], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], xl, acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xl, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], xh, acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xh, acc[m][0], 0, 0, 0);
                }
            }
        }
    }
}

// Packed rows (PairGeom PACK), 20 physical output channels: channel group og's hi rows are packed group og, its lo rows packed
// group og + 5; lane (i16, g) of accumulator tile t holds packed group 4 t + g.  Returns hi + lo of group g in `main` (every
// lane) and of group 4 in `extra` (meaningful in lanes g == 0).
__device__ __forceinline__ void pair_unpack5(const f32x4 (&acc)[3], int lane, int i16, int g, f32x4 &main, f32x4 &extra)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (through float temporaries: __builtin_bit_cast applied to a vector ELEMENT lvalue reads element 0 whatever the index
        // - hipcc 7.2 - and all four exchanges collapse into one)
        const float f1 = acc[1][j], f2 = acc[2][j];
        const int a1 = __float_as_int(f1), a2 = __float_as_int(f2);
        const int up1 = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, a1);  // tile 1, lane group g + 1: lo rows of groups 0..2
        const int lo3 = __builtin_amdgcn_ds_bpermute(i16 * 4, a2);                 // tile 2, lane group 0: lo rows of group 3
        const int up2 = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, a2);  // tile 2, lane group 1: lo rows of group 4
        main[j] = acc[0][j] + __int_as_float(g < 3 ? up1 : lo3);
        extra[j] = f1 + __int_as_float(up2);
    }
}

template <int TW, int TH, int WAVES, int CP, bool ALIAS, bool PACK>
__global__ __launch_bounds__(64 * WAVES, (WAVES / 4) * (WAVES <= 8 ? 2 : 1))  // (8-wave blocks: two per CU, <= 128 VGPRs)
void dense_pair_kernel(const PairArgs a)
{
    using G = PairGeom<TW, TH, WAVES, CP, ALIAS, PACK>;
    static_assert(WAVES > 8 || G::BLOCKS_PER_CU == 2, "8-wave shapes must fit two blocks into a CU's LDS");
    constexpr int PW = G::PW, XP = G::XP, TP = G::TP;
    extern __shared__ f32x4 pair_lds[];
    f32x4 *xl = pair_lds;                          // [CP pairs][hi | lo][XP]
    f32x4 *tl = ALIAS ? xl : xl + G::X_F4;         // [3 pairs][hi | lo][TP] (ALIAS: over the window, once conv a is done)
    f32x4 *wl = pair_lds + G::XT_F4;               // one chunk of weights
    int *uo = reinterpret_cast<int *>(wl + G::W_F4);  // unit tables: [0] full chunk, [1] last chunk, [2] conv b
    float *vl = reinterpret_cast<float *>(uo + 128);  // bias_a | rinv_a | bias_b | rinv_b (their global latency hides behind conv a)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    // (tile rows banded over the XCDs when the grid allows it: neighbouring tiles share their halo rows in one L2)
    const int tile = (gridDim.x & 7) == 0 && a.xcd_bands ? xcd_band_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int last = a.n_chunks - 1, nkb_b = (9 * a.np_b + 3) >> 2;
#ifdef OJF_PAIR_TIMING
    int stamp_i = 0;
#define OJF_STAMP() do { if (blockIdx.x == 37 && tid == 0) a.dbg[stamp_i] = (long long)__builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define OJF_STAMP() do {} while (0)
#endif
    OJF_STAMP();  // 0: start

    if (tid < 3 * 36) {  // unit -> slot offset (float4 units) of its (tap, pair) inside the window / T planes
        const int type = tid / 36, u = tid - type * 36;
        const int np = type == 0 ? CP : (type == 1 ? a.np_last : a.np_b);
        int off = 0;
        if (u < 9 * np) {
            const int tap = u / np, pr = u - tap * np;
            off = pr * 2 * (type == 2 ? TP : XP) + (tap / 3) * PW + (tap % 3);
        }
        uo[tid] = off;
    }
    if (tid >= 128 && tid < 160)  // the four epilogue vectors are contiguous (PackedPair::vec)
        reinterpret_cast<f32x4 *>(vl)[tid - 128] = reinterpret_cast<const f32x4 *>(a.bias_a)[tid - 128];
    // the T planes' tails (slots a junk output column may read; conv a writes every slot below) must hold finite values
    auto zero_t_tails = [&]() {
        constexpr int TAIL = TP - G::TILES_A * 16;
        if constexpr (TAIL > 0) {
            if (tid >= 192 && tid < 192 + 6 * TAIL) {
                const int i = tid - 192, pl = i / TAIL, sl = i - pl * TAIL;
                tl[pl * TP + G::TILES_A * 16 + sl] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    static_assert(192 + 6 * (TP - G::TILES_A * 16) <= G::THREADS, "T tail fill needs more threads");
    if constexpr (!ALIAS) zero_t_tails();

    // ---- window items of this thread: (pair, slot) -> byte offset of the pixel (or out of range) -----------------
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f32x4 *>(a.in), 0, (a.in_g0 + a.c4_in) * a.npix * 16, 0x00020000);
    unsigned poff[G::NXI];
    int xdst[G::NXI];  // float4 index of the hi element in xl (-1: no item)
#pragma unroll
    for (int k = 0; k < G::NXI; ++k) {
        const int item = k * G::THREADS + tid;
        const int pr = item / G::XS, s = item - pr * G::XS;
        const int sy = s / PW, sx = s - sy * PW;
        const int gy = y0 - 2 + sy, gx = x0 - 2 + sx;
        const bool ok = item < CP * G::XS && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        poff[k] = ok ? (unsigned)(((a.in_g0 + 2 * pr) * a.npix + gy * a.w + gx) * 16) : 0xffffffffu;
        xdst[k] = item < CP * G::XS ? pr * 2 * XP + s : -1;
    }
    const unsigned chunk_bytes = (unsigned)(2 * CP * a.npix * 16), group_bytes = (unsigned)(a.npix * 16);
    f32x4 xpa[G::NXI], xpb[G::NXI], wpre[G::NWI];
    // piece k of the next chunk: window item k (two float4) and weight float4 k of this thread
    auto prefetch_piece = [&](int k, int c, bool with_x, const f32x4 *wsrc_, int n_f4) {
#pragma unroll
        for (int kk = 0; kk < G::NPRE; ++kk) {
            if (kk != k) continue;
            if (kk < G::NXI && with_x) {
                // groups beyond the window lie beyond num_records -> zeros (odd group counts, partial last chunk)
                const unsigned o = poff[kk] == 0xffffffffu ? 0xffffffffu : poff[kk] + (unsigned)c * chunk_bytes;
                const unsigned o2 = poff[kk] == 0xffffffffu ? 0xffffffffu : o + group_bytes;
                xpa[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
                xpb[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o2, 0, 0));
            }
            if (kk < G::NWI && kk * G::THREADS + tid < n_f4) wpre[kk] = wsrc_[kk * G::THREADS + tid];
        }
    };
    auto nkb_of = [&](int c) { return c == last ? (9 * a.np_last + 3) >> 2 : G::NKB; };

    // ---- conv a ----------------------------------------------------------------------------------------------
    const int mt_a = (G::TILES_A - wave + G::WAVES - 1) / G::WAVES;  // tiles wave, wave + WAVES, ... < TILES_A
    int slot_a[G::MT_A];
#pragma unroll
    for (int m = 0; m < G::MT_A; ++m) slot_a[m] = (wave + G::WAVES * (m < mt_a ? m : 0)) * 16 + i16;
    f32x4 acc[G::MT_A][G::NACC];
#pragma unroll
    for (int m = 0; m < G::MT_A; ++m)
#pragma unroll
        for (int n = 0; n < G::NACC; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    OJF_STAMP();  // 1: setup done
#pragma unroll
    for (int k = 0; k < G::NPRE; ++k) prefetch_piece(k, 0, true, a.wa, nkb_of(0) * 64 * G::WT);
    const f32x4 *wsrc = a.wa;
    for (int c = 0; c <= last; ++c) {
        const int nkb = nkb_of(c), wf4 = 64 * G::WT;  // weight float4 per K block
        if (c) __syncthreads();  // readers of the previous chunk are done
        OJF_STAMP();  // 2 + 3c: chunk c: previous compute done
#pragma unroll
        for (int k = 0; k < G::NXI; ++k) {
            if (xdst[k] >= 0) {
                f32x4 hi, lo;
                pair_split(xpa[k], xpb[k], hi, lo);
                xl[xdst[k]] = hi;
                xl[xdst[k] + XP] = lo;
            }
        }
#pragma unroll
        for (int k = 0; k < G::NWI; ++k)
            if (k * G::THREADS + tid < nkb * wf4) wl[k * G::THREADS + tid] = wpre[k];
        OJF_STAMP();  // 3 + 3c: operands arrived and written
        __syncthreads();
        OJF_STAMP();  // 4 + 3c: barrier passed
        wsrc += nkb * wf4;
        // the next chunk's window and weights (or conv b's weights), one piece per K block
        const bool more = c < last;
        const f32x4 *nsrc = more ? wsrc : a.wb;
        const int n_f4 = more ? nkb_of(c + 1) * wf4 : nkb_b * wf4;
        pair_mac<G::MT_A, XP, G::NACC>(acc, xl, uo + (c == last ? 36 : 0), nkb, wl, slot_a, mt_a, lane, g,
                              [&](int S) { if (S < G::NPRE) prefetch_piece(S, c + 1, more, nsrc, n_f4); });
#pragma unroll
        for (int k = 0; k < G::NPRE; ++k)
            if (k >= nkb) prefetch_piece(k, c + 1, more, nsrc, n_f4);  // short chunks: the rest
    }

    OJF_STAMP();  // conv a done
    if constexpr (ALIAS) {  // the window is dead: T takes its place, conv b's weights take conv a's
        __syncthreads();
        zero_t_tails();
#pragma unroll
        for (int k = 0; k < G::NWI; ++k)
            if (k * G::THREADS + tid < nkb_b * 64 * G::WT) wl[k * G::THREADS + tid] = wpre[k];
    }
    // epilogue a: bias, LeakyReLU, zero outside the image / the needed region, split, into the T planes
    float gmax = 0.0f;
    {
        f32x4 bv[2], rv[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bv[n] = *reinterpret_cast<const f32x4 *>(vl + n * 16 + 4 * g);
            rv[n] = *reinterpret_cast<const f32x4 *>(vl + 32 + n * 16 + 4 * g);
        }
#pragma unroll
        for (int m = 0; m < G::MT_A; ++m) {
            if (m >= mt_a) continue;
            const int s = slot_a[m];
            const int ry = s / PW, rx = s - ry * PW;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            const bool ok = s < G::TS && rx < TW + 2 && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            // channel group og of this pixel -> the T planes (pair og / 2, half og & 1)
            auto put_t = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                const f32x4 lin = fma4(raw, r4, b4);
                if (ok) gmax = guard_max(gmax, lin);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ok ? leaky_max(lin[j], 0.01f) : 0.0f;
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                const f16x4 h4 = __builtin_convertvector(v, f16x4);
                const uint2 hp = __builtin_bit_cast(uint2, h4);
                const uint2 l4 = uint2{split_lo_pair(hp.x, v[0], v[1]), split_lo_pair(hp.y, v[2], v[3])};
                const int pr = og >> 1;
                uint2 *dh = reinterpret_cast<uint2 *>(tl + pr * 2 * TP + s) + (og & 1);
                uint2 *dl = reinterpret_cast<uint2 *>(tl + pr * 2 * TP + TP + s) + (og & 1);
                *dh = __builtin_bit_cast(uint2, h4);
                *dl = l4;
            };
            if constexpr (G::PACK) {
                f32x4 main, extra;
                pair_unpack5(acc[m], lane, i16, g, main, extra);
                put_t(g, main, rv[0], bv[0]);
                if (g == 0) put_t(4, extra, rv[1], bv[1]);
                if (g == 1) {  // group 5 = padding channels 20..23 of the third pair: zeros
                    reinterpret_cast<uint2 *>(tl + 2 * 2 * TP + s)[1] = uint2{0u, 0u};
                    reinterpret_cast<uint2 *>(tl + 2 * 2 * TP + TP + s)[1] = uint2{0u, 0u};
                }
            } else {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (2 * n + (g >> 1) >= 3) continue;
                    put_t(4 * n + g, acc[m][n], rv[n], bv[n]);
                }
            }
        }
    }
    OJF_STAMP();  // epilogue a done
    __syncthreads();  // T complete, conv a's weights no longer read
    if constexpr (!ALIAS) {
#pragma unroll
        for (int k = 0; k < G::NWI; ++k)
            if (k * G::THREADS + tid < nkb_b * 64 * G::WT) wl[k * G::THREADS + tid] = wpre[k];
        __syncthreads();
    }

    OJF_STAMP();  // conv b weights in place
    // ---- conv b ----------------------------------------------------------------------------------------------
    const int mt_b = (G::TILES_B - wave + G::WAVES - 1) / G::WAVES;
    int slot_b[G::MT_B];
#pragma unroll
    for (int m = 0; m < G::MT_B; ++m) slot_b[m] = (wave + G::WAVES * (m < mt_b ? m : 0)) * 16 + i16;
    f32x4 accb[G::MT_B][G::NACC];
#pragma unroll
    for (int m = 0; m < G::MT_B; ++m)
#pragma unroll
        for (int n = 0; n < G::NACC; ++n) accb[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    pair_mac<G::MT_B, TP, G::NACC>(accb, tl, uo + 72, nkb_b, wl, slot_b, mt_b, lane, g, [](int) {});
    OJF_STAMP();  // conv b done
    {
        f32x4 bv[2], rv[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bv[n] = *reinterpret_cast<const f32x4 *>(vl + 64 + n * 16 + 4 * g);
            rv[n] = *reinterpret_cast<const f32x4 *>(vl + 96 + n * 16 + 4 * g);
        }
#pragma unroll
        for (int m = 0; m < G::MT_B; ++m) {
            if (m >= mt_b) continue;
            const int s = slot_b[m];
            const int oy = s / PW, ox = s - oy * PW;
            const int gy = y0 + oy, gx = x0 + ox;
            const bool ok = s < G::OS && ox < TW && gy < a.h && gx < a.w;
            const int p = gy * a.w + gx;
            auto put_o = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                if (!ok || og >= a.og_store) return;
                const f32x4 lin = fma4(raw, r4, b4);
                gmax = guard_max(gmax, lin);
                a.out[(size_t)(a.out_g0 + og) * a.npix + p] = leaky_max4(lin, 0.01f);
            };
            if constexpr (G::PACK) {
                f32x4 main, extra;
                pair_unpack5(accb[m], lane, i16, g, main, extra);  // (every lane takes part in the exchanges)
                put_o(g, main, rv[0], bv[0]);
                if (g == 0) put_o(4, extra, rv[1], bv[1]);
            } else {
#pragma unroll
                for (int n = 0; n < 2; ++n) put_o(4 * n + g, accb[m][n], rv[n], bv[n]);
            }
        }
    }
    if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
    OJF_STAMP();  // stores issued
#undef OJF_STAMP
}

}  // namespace ojf
