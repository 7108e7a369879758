// All dense Blocks of one head (modules/model.py:4-21,219-283: x <- cat(x, Block_i(x)), i = 0..gf-1) as ONE persistent
// launch on split-fp16 MFMA (included by ojf_net.hip only).  Round 4; replaces five dense_pair_kernel launches.
//
// What the five-launch form left on the table (s_memtime stamps, profiles/r04_pair_experiments.txt): its K loops ran at
// 84-95 % of the matrix pipe, but 40 % of every launch was not a K loop - the first window fill of 240 blocks at once
// (15 %), register-staged split + ds_write + two barriers between chunks (7 %), the second convolution's weights (7 %),
// the store tail (4 %), and a launch boundary per Block on top.  Here:
//
//   * the dense-growth buffer holds SPLIT PLANES (split_pack4: {4 hi halfs | 4 lo halfs} per pixel and channel group -
//     the halves every consumer would make of the fp32 value anyway), so a window reaches LDS by LDS-DMA
//     (global_load_lds_dwordx4): no staging registers, no split, no ds_write pass, and the MFMA operand of a lane is two
//     ds_read_b128 + two v_swap (unpack_split);
//   * a STEP = one slot (5 channel groups x 9 taps = 45 half-units in 6 K blocks) of one convolution; every step of
//     every Block has the same shape (24 KB of weights, one 5-plane window), so the kernel is a uniform pipeline:
//     double-buffered windows and weights, the next step's DMA issued at the top of the current one, ONE barrier per
//     step; the intermediate T of a Block takes the place of the last window it was computed from;
//   * the Blocks follow each other inside the launch: a tile's next Block needs its 8 neighbours' previous outputs
//     (2-pixel halo), handed over as the guide's R1 form - write-through (sc1) stores, drained, one sc1 flag per tile
//     and Block; the consumer polls the 9 flags with relaxed sc1 loads and reads the fresh slot with sc1 loads.  The
//     slots a Block already had are prefetched BEFORE that dependency: the fresh slot is the last step of its Block.
//
// Slot s of a region (pitch PW = TW + 4) as in ojf_net_pair.h: an MFMA pixel tile is 16 consecutive slots, a tap is a
// constant slot offset.  Half-unit positions: read R = 2 S + ab (S = K block, ab = first / second ds_read of the lane),
// lane group g.  R < 9: (tap R, group g) - the four lane groups of a read differ by whole planes (multiples of 16
// slots), so a ds_read_b128 is conflict-free; R = 9..11 hold the fifth group's nine taps, paired so that the lane
// groups serviced together (0 with 1, 2 with 3) read slots that are congruent mod 16 (rows 0 and 2 at PW = 24 or 16).
#pragma once

namespace ojf {

constexpr int kChainNG = 5;            // channel groups per slot (19 channels + 1 of padding)
constexpr int kChainKB = 6;            // K blocks per step
constexpr int kChainStepF4 = kChainKB * 3 * 64;  // weight float4 per step: [K block][packed row tile][lane]
constexpr int kChainMaxLayers = 7;
constexpr int kChainEpoch = 8;         // flag = epoch + Blocks completed (epoch advances by this per launch)
// ints of the synchronisation array: the eight band counters sit 128 bytes apart (returning atomics on one line serialise
// whatever word they address: ~12 ns each, 3 us for 240 blocks), the per-tile flags behind them
constexpr int kChainBand0 = 32, kChainBandStride = 32, kChainFlags0 = kChainBand0 + 8 * kChainBandStride;

__host__ __device__ constexpr int chain_hu_q(int R, int g) { return R < 9 ? g : 4; }
__host__ __device__ constexpr int chain_hu_tap(int R, int g)  // -1: empty position (zero weights)
{
    return R < 9 ? R
                 : (R == 9 ? (g == 0 ? 0 : g == 1 ? 6 : g == 2 ? 1 : 7)
                           : (R == 10 ? (g == 0 ? 2 : g == 1 ? 8 : g == 2 ? 3 : -1) : (g == 0 ? 4 : g == 2 ? 5 : -1)));
}
// the slot a position reads (an empty position reads what its partner reads: same banks, no conflict)
__host__ __device__ constexpr int chain_read_tap(int R, int g) { return chain_hu_tap(R, g) >= 0 ? chain_hu_tap(R, g) : chain_hu_tap(R, g - 1); }

template <int TW, int TH, int WAVES_>
struct ChainGeom {
    static constexpr int WAVES = WAVES_, THREADS = 64 * WAVES;
    static constexpr int PW = TW + 4;
    static constexpr int XS = (TH + 4) * PW, TS = (TH + 2) * PW, OS = TH * PW;
    static constexpr int TILES_A = (TS + 15) / 16, TILES_B = (OS + 15) / 16;
    static constexpr int MT_A = (TILES_A + WAVES - 1) / WAVES, MT_B = (TILES_B + WAVES - 1) / WAVES;
    static constexpr int XP = pair_round16(pair_max(XS, TILES_A * 16 + 2 * PW + 3));  // plane pitch (window and T)
    static constexpr int NPIECE = (kChainNG * XP + 63) / 64;     // 1 KB LDS-DMA pieces of a window
    static constexpr int X_F4 = NPIECE * 64;                     // float4 per window buffer
    static constexpr int NPX = (NPIECE + WAVES - 1) / WAVES;     // pieces per wave
    static constexpr int NPW = (kChainStepF4 / 64 + WAVES - 1) / WAVES;
    static constexpr size_t LDS_BYTES = (size_t)(3 * X_F4 + 2 * kChainStepF4) * 16 + kChainMaxLayers * 128 * sizeof(float) + 16;
};

struct ChainDenseArgs {
    const f32x4 *x;   // dense-growth buffer, split planes: slot s = groups [5 s, 5 s + 5); float4 -1 is zero
    f32x4 *xo;        // the same buffer (Block l writes slot l + 1)
    const f32x4 *w;   // steps back to back (Block 0: slot 0, conv b; Block 1: slots 0, 1, conv b; ...)
    const float *vec; // per Block: bias_a | rinv_a | bias_b | rinv_b, 32 floats each
    int *sync;        // [0] epoch (advanced by the block that finishes last: the launch is graph-replayable), [1] blocks finished,
                      // [2] error flag (a neighbour's flag never arrived: bounded spin), [3] tickets of Blocks 1.. handed out,
                      // [kChainBand0 + kChainBandStride q] items of Block 0 handed out in band q,
                      // [kChainFlags0 + tile] epoch + Blocks completed on that tile
    int layers;
    int h, w_img, npix, tiles_x, tiles_y;
    int xcd_bands;
    int *ovf;         // split-fp16 range guard flag
#ifdef OJF_CHAIN_TIMING
    long long *dbg;
#endif
};

// acc[m][t] += Wpacked[S][t] * window[position (2S, g), (2S+1, g)][slot[m]] for the six K blocks of a step and the
// wave's MTW pixel tiles.  PACKED ROWS: the 20 output channels' hi and lo weight halves occupy three 16-row MFMA tiles
// instead of 2 x (hi, lo) = four -
//     tile 0 = hi halves of channels 0..15,  tile 1 = lo halves of channels 0..15,
//     tile 2 = hi halves of channels 16..19 (rows 0..3) | lo halves of channels 16..19 (rows 4..7) | zeros
// - so a product block costs FIVE MFMAs (x_hi meets all three tiles, x_lo the two that hold hi rows; the lo rows riding
// along in tile 2 add w_lo x_lo, the term the split otherwise drops) and THREE weight reads instead of six and four: the
// K loop of this kernel is bound by the matrix pipe AND the LDS pipe together (measured alone: 4700 and 3900 cycles of a
// 5700-cycle step), both shrink by a sixth.  hi and lo rows of channels 0..15 sit in the SAME lane group of tiles 0 and
// 1: the epilogue adds two accumulators; channels 16..19 need one cross-lane-group exchange (pair of lane groups 0, 1).
// xg: window buffer + g * XP (the lane group's plane for R < 9); x4: window buffer + 4 * XP; t9: the lane's slot offsets
// for R = 9, 10, 11.  Straight-line code per tile count (the caller branches once per step).
template <int MTW, int MT, int PW, int TSTRIDE, class Hook>
__device__ __forceinline__ void chain_mac_t(f32x4 (&acc)[MT][3], const f32x4 *xg, const f32x4 *x4, const f32x4 *wl_lane,
                                            const int (&t9)[3], Hook hook)
{
    // (xg / x4 point at the lane's slot of pixel tile 0; tile m lies TSTRIDE slots further: an immediate offset)
#pragma unroll
    for (int S = 0; S < kChainKB; ++S) {
        f16x8 w[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) w[t] = __builtin_bit_cast(f16x8, wl_lane[(S * 3 + t) * 64]);
        if (S == kChainKB / 2) hook();  // (mid-step: the fresh slot's flags and loads)
        const int RA = 2 * S, RB = 2 * S + 1;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4 A, B;
            if (RA < 9) A = xg[m * TSTRIDE + (RA / 3) * PW + RA % 3];
            else A = x4[m * TSTRIDE + t9[RA >= 9 ? RA - 9 : 0]];
            if (RB < 9) B = xg[m * TSTRIDE + (RB / 3) * PW + RB % 3];
            else B = x4[m * TSTRIDE + t9[RB >= 9 ? RB - 9 : 0]];
            // (plain shuffles, not unpack_split's hand-placed v_swap_b32: hipcc cannot see what inline assembly writes, and with
            // the x_lo product first in line an MFMA read the swapped registers too early - wrong tile-2 sums on MI355X)
            const f16x8 xh = __builtin_bit_cast(f16x8, __builtin_shufflevector(A, B, 0, 1, 4, 5));
            const f16x8 xl = __builtin_bit_cast(f16x8, __builtin_shufflevector(A, B, 2, 3, 6, 7));
            // (small terms first, like mfma_f16x3)
            acc[m][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2], xl, acc[m][2], 0, 0, 0);
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xl, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], xh, acc[m][1], 0, 0, 0);
            acc[m][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2], xh, acc[m][2], 0, 0, 0);
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xh, acc[m][0], 0, 0, 0);
        }
        // (keeps hipcc from lifting later K blocks' reads above this one's MFMAs: it spends every register of the
        // 128-VGPR budget on that and then spills the per-lane state of the item loop - each reload waits for the
        // window pieces and write-through stores in flight)
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int MT, int PW, int TSTRIDE, class Hook>
__device__ __forceinline__ void chain_mac(f32x4 (&acc)[MT][3], const f32x4 *xg, const f32x4 *x4, const f32x4 *wl_lane,
                                          int mt_wave, const int (&t9)[3], Hook hook)
{
    // (pixel tiles are dealt round robin: a wave has MT or MT - 1 of them)
    if (MT == 1 || mt_wave == MT) chain_mac_t<MT, MT, PW, TSTRIDE>(acc, xg, x4, wl_lane, t9, hook);
    else chain_mac_t<(MT > 1 ? MT - 1 : 1), MT, PW, TSTRIDE>(acc, xg, x4, wl_lane, t9, hook);
}
// Packed accumulators of one pixel tile -> lane (pixel i16, g): `main` = channels 4 g .. 4 g + 3, `extra` = channels
// 16..19 (meaningful in lane group 0): the lo rows of tile 2 sit one lane group up; v_permlane16_swap_b32 (gfx950) of a
// register with a copy of itself leaves row 1 in row 0 of its second result
__device__ __forceinline__ void chain_unpack(const f32x4 (&acc)[3], f32x4 &main, f32x4 &extra)
{
    main = acc[0] + acc[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f2 = acc[2][j];
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(f2), __float_as_uint(f2), false, false);
        extra[j] = f2 + __uint_as_float(r[1]);
    }
}

template <int TW, int TH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void dense_chain_kernel(const ChainDenseArgs a)
{
    using G = ChainGeom<TW, TH, WAVES>;
    constexpr int PW = G::PW, XP = G::XP;
    extern __shared__ f32x4 chain_lds[];
    // (LDS-DMA destinations first: every piece lands below 128 KB; T and the vectors, written by ds_write, behind them)
    f32x4 *wl = chain_lds;                                  // [2][kChainStepF4]
    f32x4 *xl = chain_lds + 2 * kChainStepF4;               // [2][X_F4]: window (5 planes of XP slots)
    f32x4 *tl = xl + 2 * G::X_F4;                           // [X_F4]: the intermediate T, same plane layout
    float *vl = reinterpret_cast<float *>(tl + G::X_F4);    // [layers][128]
    int *tk = reinterpret_cast<int *>(vl + kChainMaxLayers * 128);  // [0..1]: the block's tickets (written by thread 0), [2]: look-ahead

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int L = a.layers, total = L * ntiles;
    const int epoch = *a.sync;  // (constant during the launch: only the last block to finish advances it)
    int *flags = a.sync + kChainFlags0;
#ifdef OJF_CHAIN_TIMING
    int stamp_i = 0;
#define OJF_CSTAMP() do { if (blockIdx.x == 37 && tid == 0 && stamp_i < 96) a.dbg[stamp_i] = (long long)__builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define OJF_CSTAMP() do {} while (0)
#endif
    OJF_CSTAMP();

    // ---- work items: (Block l, tile t) = ticket l * ntiles + t, handed out in that order --------------------------
    // A block draws its NEXT ticket while it works on the current one.  Every item waits only for items of the previous
    // Block, i.e. for smaller tickets, and the smallest unfinished ticket is always some running block's CURRENT item
    // (whoever holds it as its next one is still busy with a smaller one): the launch makes progress with any number of
    // resident blocks - two chains on two streams, two processes on one device, a grid larger than the device.
    // Drawing (one wave, lane 0 keeps the result).  Block 0's items have no predecessors, so their order is free: they come
    // from eight counters, one per band of tiles - block b starts with band b % 8 (blocks are observed to land on XCD
    // b % 8: a band's tiles share their halos in one L2; and 30 blocks per word instead of 240 at the start of a launch,
    // ~0.3 instead of ~3 us).  Everything else comes from ONE counter, and only a wave that has SEEN all eight bands
    // exhausted may use it: a ticket of a later Block exists => every item of Block 0 is in the hands of a running block.
    // Later draws are issued by the wave with the fewest pixel tiles behind the first step of an item and handed over
    // before the item's second convolution: nobody waits for them.
    bool bands_open = true;
    auto band_first = [&](int q) { return q * ntiles / 8; };
    auto draw = [&]() -> int {  // (call with the whole wave)
        if (bands_open) {
            for (int tries = 0; tries < 64; ++tries) {
                int left = 0;
                if (lane < 8) left = band_first(lane + 1) - band_first(lane) - __hip_atomic_load(a.sync + kChainBand0 + kChainBandStride * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long open = __builtin_amdgcn_ballot_w64(left > 0);
                if (!open) break;
                const int home = blockIdx.x & 7;
                const unsigned rot = (unsigned)((open >> home) | (open << (8 - home))) & 0xffu;  // bands home, home + 1, ... as bits 0, 1, ...
                const int q = (home + __builtin_ctz(rot)) & 7;
                int k = 0;
                if (lane == 0) k = __hip_atomic_fetch_add(a.sync + kChainBand0 + kChainBandStride * q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k = __builtin_amdgcn_readfirstlane(k);
                if (k < band_first(q + 1) - band_first(q)) return band_first(q) + k;
            }
            bands_open = false;
        }
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(a.sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ntiles + t;  // (lane 0)
    };
    if (wave == 0) {
        // (the first draw skips the look at the counters: the block's own band has items left unless the grid is tiny)
        const int home = blockIdx.x & 7;
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(a.sync + kChainBand0 + kChainBandStride * home, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k = __builtin_amdgcn_readfirstlane(k);
        int t = k < band_first(home + 1) - band_first(home) ? band_first(home) + k : draw();
        if (lane == 0) {
            tk[0] = t;
            tk[2] = 0;
        }
    }
    for (int i = tid; i < L * 32; i += G::THREADS)
        reinterpret_cast<f32x4 *>(vl)[i] = reinterpret_cast<const f32x4 *>(a.vec)[i];
    __syncthreads();
    int ticket = __builtin_amdgcn_readfirstlane(tk[0]);

    // ---- per item: tile origin, this wave's window pieces, the neighbours' flags -------------------------------------
    // window piece: item = piece * 64 + lane -> (group, slot) -> pixel (float4 offset inside an input slot, -1: none)
    // An item is four scalars (the kernel runs at 128 VGPRs: whatever is per lane - the window piece of a lane, its
    // neighbour flag - is recomputed where it is used, a few dozen VALU instructions per 5000-cycle step)
    struct Item { int l, tile, x0, y0; };
    auto make_item = [&](int t_, Item &it) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        it.l = t / ntiles;
        it.tile = t - it.l * ntiles;
        const int ty = it.tile / a.tiles_x, tx = it.tile - ty * a.tiles_x;
        it.x0 = tx * TW; it.y0 = ty * TH;
    };
    // window piece: element = piece * 64 + lane -> (group, slot) -> pixel, or the zero float4 in front of the planes
    // fresh: the slot is read for the first time after another CU of this launch wrote it (sc1: L1 bypassed, coherent with
    // the write-through stores); any later read of the same slot is a plain load and stays in the XCD's L2
    auto load_x = [&](const Item &it, int slot, int buf, bool fresh) {
        const f32x4 *base = a.x + (size_t)slot * kChainNG * a.npix;
#pragma unroll
        for (int j = 0; j < G::NPX; ++j) {
            const int pc = wave + G::WAVES * j;
            if (pc < G::NPIECE) {
                const int e = pc * 64 + lane;
                const int q = e / XP, sl = e - q * XP;
                const int sy = sl / PW, sx = sl - sy * PW;
                const int gy = it.y0 - 2 + sy, gx = it.x0 - 2 + sx;
                const bool ok = q < kChainNG && sl < G::XS && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w_img;
                const f32x4 *src = ok ? base + (q * a.npix + gy * a.w_img + gx) : a.x - 1;
                if (fresh)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                                     (void __attribute__((address_space(3))) *)(xl + buf * G::X_F4 + pc * 64), 16, 0, 16);
                else
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                                     (void __attribute__((address_space(3))) *)(xl + buf * G::X_F4 + pc * 64), 16, 0, 0);
            }
        }
    };
    auto issue_w = [&](int step, int buf) {
        const f32x4 *src = a.w + (size_t)step * kChainStepF4;
#pragma unroll
        for (int j = 0; j < G::NPW; ++j) {
            const int pc = wave + G::WAVES * j;
            if (pc < kChainStepF4 / 64)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pc * 64 + lane),
                                                 (void __attribute__((address_space(3))) *)(wl + buf * kChainStepF4 + pc * 64), 16, 0, 0);
        }
    };
    // Blocks completed by every neighbour of `it` >= done?  (one relaxed sc1 load per lane 0..8)
    auto neighbours_done = [&](const Item &it, int done) {
        const int need = epoch + done;
        int v = need;  // lanes 0..8 look at one neighbour each (the tile itself included)
        if (lane < 9) {
            const int ty = it.tile / a.tiles_x, tx = it.tile - ty * a.tiles_x;
            const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
            if ((unsigned)ny < (unsigned)a.tiles_y && (unsigned)nx < (unsigned)a.tiles_x)
                v = __hip_atomic_load(flags + ny * a.tiles_x + nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return __builtin_amdgcn_ballot_w64(v - need < 0) == 0;
    };
    auto wait_neighbours = [&](const Item &it, int done) {
        for (int spins = 0; !neighbours_done(it, done); ++spins) {
            if (spins > (1 << 20)) {  // (cannot happen by construction; a launch must end all the same: the host reports the flag)
                if (lane == 0) {
                    a.sync[2] = 1;
                    if (a.ovf) guard_raise(a.ovf, 2);  // (reported by ojf_net_forward / ojf_net_check)
                }
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };
    auto first_step = [](int l) { return l * (l + 3) / 2; };  // Block l: steps for slots 0..l, then conv b

    // ---- pixel tiles of this wave ---------------------------------------------------------------------------------
    const int mt_a = (G::TILES_A - wave + G::WAVES - 1) / G::WAVES, mt_b = (G::TILES_B - wave + G::WAVES - 1) / G::WAVES;
    // pixel tile m of the wave = tile wave + WAVES m: slot = ls + TSTRIDE m (a wave has MT or MT - 1 tiles: chain_mac; the
    // epilogues skip the rest).  `ls` is refreshed through an opaque no-op at the top of every step: left alone, hipcc
    // hoists every (buffer, plane, tile) address combination out of the item loop - some thirty VGPRs - and spills.
    constexpr int TSTRIDE = G::WAVES * 16;
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
    float gmax = 0.0f;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.xo, 0, (L + 1) * kChainNG * a.npix * 16, 0x00020000);

    Item cur;
    int xb = 0, wb = 0;              // window / weight buffer of the coming step
    int pub_tile = -1, pub_val = 0;  // the finished item whose flag is published behind the next barrier
    if (ticket < total) {
        make_item(ticket, cur);
        load_x(cur, 0, 0, false);
        issue_w(first_step(cur.l), 0);
    }
    OJF_CSTAMP();
    while (ticket < total) {
        const int l = cur.l;
        int step = first_step(l);
        int nxt_l = 0;
        int known = 0;  // Blocks every neighbour is known to have completed (slots 0..known may be read)
        int next_ticket = total, drawn = total;
        bool draw_pending = false;  // (wave WAVES - 1: the returning atomic is this wave's youngest memory operation)
        f32x4 acc[G::MT_A][3];
#pragma unroll
        for (int m = 0; m < G::MT_A; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- conv a: one step per slot, the slot Block l - 1 produced comes last --------------------------------
        for (int c = 0; c <= l; ++c) {
            // this wave's pieces have landed, its stores are drained (the ticket may stay in flight across one barrier)
            if (draw_pending) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            draw_pending = false;
            __syncthreads();
            OJF_CSTAMP();
            if (c == 0) {
                if (pub_tile >= 0 && tid == 0)  // every wave's stores of the previous item were drained before the barrier
                    __hip_atomic_store(flags + pub_tile, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pub_tile = -1;
                known = __builtin_amdgcn_readfirstlane(tk[2]);  // (what the previous item's look-ahead found)
            }
            const bool last = c == l;
            issue_w(step + 1, wb ^ 1);  // (the next slot's, or conv b's)
            // the window of the coming step: this item's next slot, or - behind its last slot - the next item's first one
            const bool fresh = !last && c + 1 == l;  // the slot Block l - 1 wrote: wanted last, asked for in mid-step
            if (!last && !fresh) {
                if (known < c + 1) {  // (only when the neighbours lag by more than a Block: this tile's earlier Blocks ran elsewhere)
                    wait_neighbours(cur, l - 1);
                    known = l - 1;
                }
                load_x(cur, c + 1, xb ^ 1, false);
            }
            asm volatile("" : "+v"(ls));
            const f32x4 *xbuf = xl + xb * G::X_F4 + ls, *wl_lane = wl + wb * kChainStepF4 + lane;
            chain_mac<G::MT_A, PW, TSTRIDE>(acc, xbuf + g * XP, xbuf + 4 * XP, wl_lane, mt_a, t9, [&]() {
                if (fresh) {
                    wait_neighbours(cur, l);
                    load_x(cur, c + 1, xb ^ 1, true);
                }
            });
            if (c == 0 && wave == G::WAVES - 1) {  // (behind everything else this step asked for)
                drawn = draw();
                draw_pending = !last;
            }
            ++step;
            wb ^= 1;
            if (!last) xb ^= 1;
        }
        OJF_CSTAMP();
        // ---- epilogue a: bias, LeakyReLU, zero outside the image / the needed region, split, into the T planes ----
        {
            const float *v = vl + l * 128;
            const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
#pragma unroll
            for (int m = 0; m < G::MT_A; ++m) {
                if (m >= mt_a) continue;
                const int s = ls + TSTRIDE * m;
                const int ry = s / PW, rx = s - ry * PW;
                const int gy = cur.y0 - 1 + ry, gx = cur.x0 - 1 + rx;
                const bool ok = s < G::TS && rx < TW + 2 && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w_img;
                f32x4 main, extra;
                chain_unpack(acc[m], main, extra);
                auto put_t = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                    const f32x4 lin = fma4(raw, r4, b4);
                    if (ok) gmax = guard_max(gmax, lin);
                    f32x4 val;
#pragma unroll
                    for (int j = 0; j < 4; ++j) val[j] = ok ? leaky_max(lin[j], 0.01f) : 0.0f;
                    tl[og * XP + s] = split_pack4(val);
                };
                put_t(g, main, rm, bm);
                if (g == 0) put_t(4, extra, re, be);
            }
        }
        OJF_CSTAMP();
        if (wave == G::WAVES - 1 && lane == 0) tk[0] = drawn;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // conv b's weights
        __syncthreads();
        next_ticket = __builtin_amdgcn_readfirstlane(tk[0]);
        if (next_ticket < total) {  // the next item's first window and weights travel while this item's second convolution runs
            Item nxt;
            make_item(next_ticket, nxt);
            load_x(nxt, 0, xb ^ 1, false);
            nxt_l = nxt.l;
            issue_w(first_step(nxt_l), wb ^ 1);
        }
        // ---- conv b -------------------------------------------------------------------------------------------
        f32x4 accb[G::MT_B][3];
#pragma unroll
        for (int m = 0; m < G::MT_B; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) accb[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            asm volatile("" : "+v"(ls));
            const f32x4 *wl_lane = wl + wb * kChainStepF4 + lane, *tbuf = tl + ls;
            chain_mac<G::MT_B, PW, TSTRIDE>(accb, tbuf + g * XP, tbuf + 4 * XP, wl_lane, mt_b, t9, [&]() {
                // A look at the next item's neighbours by the wave with the fewest pixel tiles, while the matrix pipe is busy:
                // in step with the rest of the grid they have completed the Blocks before the one this block is finishing,
                // which covers every slot the next item reads before its own mid-step poll
                if (wave == G::WAVES - 1) {
                    bool ok = false;
                    if (next_ticket < total && nxt_l >= 2) {
                        Item nxt;
                        make_item(next_ticket, nxt);
                        ok = neighbours_done(nxt, nxt_l - 1);
                    }
                    if (lane == 0) tk[2] = ok ? nxt_l - 1 : 0;
                }
            });
        }
        OJF_CSTAMP();
        {
            const float *v = vl + l * 128 + 64;
            const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
#pragma unroll
            for (int m = 0; m < G::MT_B; ++m) {
                if (m >= mt_b) continue;
                const int s = ls + TSTRIDE * m;
                const int oy = s / PW, ox = s - oy * PW;
                const int gy = cur.y0 + oy, gx = cur.x0 + ox;
                const bool ok = s < G::OS && ox < TW && gy < a.h && gx < a.w_img;
                const int p = gy * a.w_img + gx;
                f32x4 main, extra;
                chain_unpack(accb[m], main, extra);
                auto put_o = [&](int og, const f32x4 &raw, const f32x4 &r4, const f32x4 &b4) {
                    if (!ok) return;
                    const f32x4 lin = fma4(raw, r4, b4);
                    gmax = guard_max(gmax, lin);
                    const f32x4 val = split_pack4(leaky_max4(lin, 0.01f));
                    const unsigned off = (unsigned)((((l + 1) * kChainNG + og) * a.npix + p) * 16);
                    // sc1: write-through, for the neighbours that read this slot inside the launch; the last Block's slot is
                    // read by the next kernel only: plain stores leave it in the XCD's L2 (an sc1 store drops the line)
                    if (l + 1 < L) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ors, off, 0, 16);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ors, off, 0, 0);
                };
                put_o(g, main, rm, bm);
                if (g == 0) put_o(4, extra, re, be);
            }
        }
        OJF_CSTAMP();
        pub_tile = cur.tile;
        pub_val = epoch + l + 1;
        wb ^= 1;
        xb ^= 1;  // the next item's first window went there
        ticket = next_ticket;
        if (ticket < total) make_item(ticket, cur);
    }
    // the last item's flag: somebody may be waiting for it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pub_tile >= 0 && tid == 0) __hip_atomic_store(flags + pub_tile, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gmax > 65504.0f && a.ovf) guard_raise(a.ovf, 1);
    if (tid == 0) {  // the block that finishes last re-arms tickets and flags for the next launch (every block has drawn its last ticket)
        const int done = __hip_atomic_fetch_add(a.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            a.sync[1] = 0;
            a.sync[3] = 0;
            for (int q = 0; q < 8; ++q) a.sync[kChainBand0 + kChainBandStride * q] = 0;
            a.sync[0] = epoch + kChainEpoch;
        }
    }
#undef OJF_CSTAMP
}

}  // namespace ojf
