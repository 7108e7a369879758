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
    static constexpr size_t LDS_BYTES = (size_t)(3 * X_F4 + 2 * kChainStepF4) * 16 + kChainMaxLayers * 128 * sizeof(float);
};

struct ChainDenseArgs {
    const f32x4 *x;   // dense-growth buffer, split planes: slot s = groups [5 s, 5 s + 5); float4 -1 is zero
    f32x4 *xo;        // the same buffer (Block l writes slot l + 1)
    const f32x4 *w;   // steps back to back (Block 0: slot 0, conv b; Block 1: slots 0, 1, conv b; ...)
    const float *vec; // per Block: bias_a | rinv_a | bias_b | rinv_b, 32 floats each
    int *sync;        // [0] epoch (advanced by the block that finishes last: the launch is graph-replayable), [1] blocks finished,
                      // [2] error flag (a neighbour's flag never arrived: bounded spin), [16 + tile] epoch + Blocks completed
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
template <int MTW, int MT, int PW, class Hook>
__device__ __forceinline__ void chain_mac_t(f32x4 (&acc)[MT][3], const f32x4 *xg, const f32x4 *x4, const f32x4 *wl_lane,
                                            const int (&slot)[MT], const int (&t9)[3], Hook hook)
{
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
            if (RA < 9) A = xg[slot[m] + (RA / 3) * PW + RA % 3];
            else A = x4[slot[m] + t9[RA >= 9 ? RA - 9 : 0]];
            if (RB < 9) B = xg[slot[m] + (RB / 3) * PW + RB % 3];
            else B = x4[slot[m] + t9[RB >= 9 ? RB - 9 : 0]];
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
    }
}
template <int MT, int PW, class Hook>
__device__ __forceinline__ void chain_mac(f32x4 (&acc)[MT][3], const f32x4 *xg, const f32x4 *x4, const f32x4 *wl_lane,
                                          const int (&slot)[MT], int mt_wave, const int (&t9)[3], Hook hook)
{
    // (pixel tiles are dealt round robin: a wave has MT or MT - 1 of them)
    if (MT == 1 || mt_wave == MT) chain_mac_t<MT, MT, PW>(acc, xg, x4, wl_lane, slot, t9, hook);
    else chain_mac_t<(MT > 1 ? MT - 1 : 1), MT, PW>(acc, xg, x4, wl_lane, slot, t9, hook);
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

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = (gridDim.x & 7) == 0 && a.xcd_bands ? xcd_band_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    if (tile >= ntiles) return;  // (grid padded to a multiple of 8; nobody waits for a padding block)
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int L = a.layers;
    const int epoch = *a.sync;  // (constant during the launch: only the last block to finish advances it)
    int *flags = a.sync + 16;
#ifdef OJF_CHAIN_TIMING
    int stamp_i = 0;
#define OJF_CSTAMP() do { if (tile == 37 && tid == 0 && stamp_i < 96) a.dbg[stamp_i] = (long long)__builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define OJF_CSTAMP() do {} while (0)
#endif
    OJF_CSTAMP();

    for (int i = tid; i < L * 32; i += G::THREADS)
        reinterpret_cast<f32x4 *>(vl)[i] = reinterpret_cast<const f32x4 *>(a.vec)[i];

    // ---- this wave's window pieces: item = piece * 64 + lane -> (group, slot) -> pixel (float4 offset in a slot) ----
    // fresh: the slot was written by this launch and is read for the first time (sc1: L1 bypassed, coherent with the
    // neighbours' write-through stores); later reads of the same slot are plain loads and stay in the XCD's L2
    int xoff[G::NPX];
#pragma unroll
    for (int j = 0; j < G::NPX; ++j) {
        const int item = (wave + G::WAVES * j) * 64 + lane;
        const int q = item / XP, sl = item - q * XP;
        const int sy = sl / PW, sx = sl - sy * PW;
        const int gy = y0 - 2 + sy, gx = x0 - 2 + sx;
        const bool ok = q < kChainNG && sl < G::XS && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w_img;
        xoff[j] = ok ? q * a.npix + gy * a.w_img + gx : -1;
    }
    auto load_x = [&](int slot, int buf, bool fresh) {
        const f32x4 *base = a.x + (size_t)slot * kChainNG * a.npix;
#pragma unroll
        for (int j = 0; j < G::NPX; ++j) {
            const int pc = wave + G::WAVES * j;
            if (pc < G::NPIECE) {
                const f32x4 *src = xoff[j] >= 0 ? base + xoff[j] : a.x - 1;
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
    load_x(0, 0, false);
    issue_w(0, 0);

    // ---- neighbours (lanes 0..8 of every wave poll one flag each) -------------------------------------------------
    const int *nbflag = nullptr;
    if (lane < 9) {
        const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
        if ((unsigned)ny < (unsigned)a.tiles_y && (unsigned)nx < (unsigned)a.tiles_x) nbflag = flags + ny * a.tiles_x + nx;
    }
    auto wait_neighbours = [&](int done) {  // until every neighbour has completed `done` Blocks
        const int need = epoch + done;
        for (int spins = 0;; ++spins) {
            int v = need;
            if (nbflag) v = __hip_atomic_load(nbflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(v - need < 0) == 0) break;
            if (spins > (1 << 20)) {  // (a launch that cannot make progress must end: the host reports the flag)
                if (lane == 0) a.sync[2] = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };

    // ---- pixel tiles of this wave ---------------------------------------------------------------------------------
    const int mt_a = (G::TILES_A - wave + G::WAVES - 1) / G::WAVES, mt_b = (G::TILES_B - wave + G::WAVES - 1) / G::WAVES;
    int slot_a[G::MT_A], slot_b[G::MT_B];
#pragma unroll
    for (int m = 0; m < G::MT_A; ++m) slot_a[m] = (wave + G::WAVES * (m < mt_a ? m : 0)) * 16 + i16;
#pragma unroll
    for (int m = 0; m < G::MT_B; ++m) slot_b[m] = (wave + G::WAVES * (m < mt_b ? m : 0)) * 16 + i16;
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

    int xb = 0, wb = 0, step = 0;
    OJF_CSTAMP();
    for (int l = 0; l < L; ++l) {
        f32x4 acc[G::MT_A][3];
#pragma unroll
        for (int m = 0; m < G::MT_A; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- conv a: one step per slot, the slot Block l - 1 produced comes last --------------------------------
        for (int c = 0; c <= l; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed, its stores are drained
            __syncthreads();
            OJF_CSTAMP();
            if (c == 0 && l > 0 && tid == 0)  // every wave's stores of Block l - 1 were drained before the barrier
                __hip_atomic_store(flags + tile, epoch + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = c == l;
            issue_w(step + 1, wb ^ 1);  // (the next slot's, or conv b's)
            const int nslot = last ? 0 : c + 1;  // behind the last slot: the next Block's first window
            const bool have_next = !last || l + 1 < L;
            const bool fresh = !last && c + 1 == l;  // written by this launch: this tile and its neighbours
            if (have_next && !fresh) load_x(nslot, xb ^ 1, false);
            const f32x4 *xbuf = xl + xb * G::X_F4, *wl_lane = wl + wb * kChainStepF4 + lane;
            chain_mac<G::MT_A, PW>(acc, xbuf + g * XP, xbuf + 4 * XP, wl_lane, slot_a, mt_a, t9, [&]() {
                if (fresh) {
                    wait_neighbours(l);
                    load_x(nslot, xb ^ 1, true);
                }
            });
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
                const int s = slot_a[m];
                const int ry = s / PW, rx = s - ry * PW;
                const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // conv b's weights
        __syncthreads();
        if (l + 1 < L) issue_w(step + 1, wb ^ 1);
#ifdef OJF_CHAIN_TIMING
        if (l == 0 && tile == 37)  // debugging aid: what conv b of Block 0 finds in its weight buffer
            for (int i = tid; i < kChainStepF4; i += G::THREADS) reinterpret_cast<f32x4 *>(a.dbg + 128)[i] = wl[wb * kChainStepF4 + i];
#endif
        // ---- conv b -------------------------------------------------------------------------------------------
        f32x4 accb[G::MT_B][3];
#pragma unroll
        for (int m = 0; m < G::MT_B; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) accb[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const f32x4 *wl_lane = wl + wb * kChainStepF4 + lane;
            chain_mac<G::MT_B, PW>(accb, tl + g * XP, tl + 4 * XP, wl_lane, slot_b, mt_b, t9, []() {});
        }
        OJF_CSTAMP();
        {
            const float *v = vl + l * 128 + 64;
            const f32x4 bm = *reinterpret_cast<const f32x4 *>(v + 4 * g), rm = *reinterpret_cast<const f32x4 *>(v + 32 + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(v + 16), re = *reinterpret_cast<const f32x4 *>(v + 48);
#pragma unroll
            for (int m = 0; m < G::MT_B; ++m) {
                if (m >= mt_b) continue;
                const int s = slot_b[m];
                const int oy = s / PW, ox = s - oy * PW;
                const int gy = y0 + oy, gx = x0 + ox;
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ors, off, 0, 16 /* sc1: write-through */);
                };
                put_o(g, main, rm, bm);
                if (g == 0) put_o(4, extra, re, be);
            }
        }
        OJF_CSTAMP();
        ++step;
        wb ^= 1;
        xb ^= 1;  // the next Block's first window went there
    }
    if (gmax > 65504.0f && a.ovf) *a.ovf = 1;
    if (tid == 0) {  // the block that finishes last re-arms the flags for the next launch
        const int done = __hip_atomic_fetch_add(a.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == ntiles - 1) {
            a.sync[1] = 0;
            a.sync[0] = epoch + kChainEpoch;
        }
    }
#undef OJF_CSTAMP
}

}  // namespace ojf
