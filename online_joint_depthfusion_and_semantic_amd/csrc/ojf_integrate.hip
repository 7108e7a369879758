// INTEGRATE: scatter of the per-ray TSDF updates back into the fp16 volumes
// (modules/pipeline.py:137-171 + modules/integrator.py:15-126), without materialising the
// reference's int64 index / fp64 weight tensors or its two dense fp32 caches.
//
// FAST mode (this file).  Colliding voxel writes (~16 entries per touched voxel) are resolved with
// order-free integer arithmetic so that the result does not depend on scheduling:
//   * sum(w) and sum(w*v) are accumulated as 2^-44 fixed point in 64-bit integers (integer adds
//     are associative, so any combining order - in-LDS per tile, then per voxel - gives the same
//     bits); the total is rounded once to fp32 where the reference rounds after every add, which
//     moves at most a handful of voxels per frame by one fp16 ulp;
//   * the semantic "last writer wins" rule is an integer max over entry ids, i.e. exact.
// Each pixel tile emits ONE record per voxel it touches and links it into that voxel's list with a
// single atomic exchange on a dense 4-byte head table (measured: every additional global atomic per
// record costs ~18 us per frame, a per-record append to one shared counter ~350 us).  Records and
// first-touch lists live in per-tile slices of the workspace (tile t owns [t*2048, (t+1)*2048) of both), so
// the hot path has NO same-address global atomics: three per-tile counter atomics on one cache line cost
// 14 us per frame.  Only the rare hash-full entries and the entry-list path allocate from counters, behind
// the slices.  The finalize pass walks the touched voxels, sums their ~2 records, applies the running-mean
// update and zeroes the head entries, so every call leaves the workspace clean.
//
// PARITY mode (ojf_integrate_parity.hip) reproduces the reference's sequential fp32 sums bit for bit.
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "ojf_integrate.h"

namespace ojf {

constexpr int kSlots = 2048;
constexpr unsigned int kEmpty = 0xffffffffu;
// Pixel tile of one accumulate block: 16 x 8.  Round 3 (was 8 x 8): a tile's rays hit 1150-1390 distinct voxels at most
// (host count over the bench streams at 320x240 -> 256^3 and 640x480 -> 512^3; 8 x 8: 730-810), still inside the 2048-slot
// hash; a voxel now receives 2.1 instead of 2.6-2.9 records per frame, and 600 blocks at 320x240 are ONE round over the
// 768 block slots of the chip where 1200 were 1.56 (= two) rounds of a latency-bound kernel.
// Pixel tile of an accumulate block.  OJF_ACC_TILE = 1 (round 6): 8 columns x 16 rows with the tile's pixels numbered DOWN the columns, so
// that the lanes of a wave (64 consecutive pixel numbers at one ray offset) walk along the image axis that, for an upright camera, is
// the volume's contiguous z axis (what pays in extract_tile_kernel, profiles/r06_extract_tile_shape.txt): integrate 57.2 -> 56.0 us in a
// same-box A/B, the semantic form unchanged; 0: 16 x 8, row-major numbering (rounds 3-5).  Any tiling gives the same volumes (integer sums,
// entry-id maxima).  Measured alongside and NOT kept: the finalize kernel reading the tile's records in order instead of a first-touch list
// (one dependent hop less per voxel, but every record is read by an otherwise idle lane): +2 us.
#ifndef OJF_ACC_TILE
#define OJF_ACC_TILE 1
#endif
constexpr int kTileW = OJF_ACC_TILE ? 8 : 16, kTileH = OJF_ACC_TILE ? 16 : 8, kTilePix = kTileW * kTileH;
__device__ __forceinline__ void tile_pixel(int p, int &rl, int &cl)  // pixel number inside the tile -> (row, column) inside the tile
{
    if (OJF_ACC_TILE) { cl = p / kTileH; rl = p - cl * kTileH; }
    else { rl = p / kTileW; cl = p - rl * kTileW; }
}

static size_t tile_count(int h, int w) { return (size_t)((h + kTileH - 1) / kTileH) * ((w + kTileW - 1) / kTileW); }
// records / first touches: a 2048-element slice per tile, then room for every entry that found its tile's hash full
static size_t list_capacity(int h, int w, int n_tail)
{
    const size_t per_tile = (size_t)kTilePix * n_tail * 8;
    return tile_count(h, w) * (kSlots + per_tile);
}

size_t fast_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail)
{
    const size_t nvox = (size_t)X * Y * Z, cap = list_capacity(h, w, n_tail);
    return kHeaderBytes + nvox * sizeof(unsigned int) + cap * sizeof(VoxelRec) + cap * sizeof(unsigned int) +
           tile_count(h, w) * sizeof(unsigned int);
}

// publish one record; returns true when this is the voxel's first record of the frame
__device__ __forceinline__ bool link_record(const IntegrateArgs &a, unsigned int idx, unsigned int lin,
                                            unsigned long long xw, unsigned long long xu, unsigned int e_last,
                                            unsigned int e_diff)
{
    const unsigned int prev = atomicExch(&a.head[lin], idx + 1u);
    VoxelRec r;
    r.lin = lin; r.next = prev; r.w = xw; r.u = xu; r.e_last = e_last; r.e_diff = e_diff;
    a.recs[idx] = r;
    return prev == 0;
}

// LDS-aggregated accumulate: one block owns a 16x8 pixel tile and all n_tail samples of its rays.
// A wave is 64 pixels of the tile at one ray offset, so its lanes hit a few dozen distinct voxels;
// colliding writes are first combined in a 2048-slot LDS hash (integer adds / maxima), and only one
// record per (tile, voxel) goes to HBM (~22 entries/voxel -> 2.1 tiles/voxel).  Entries that find the
// hash full become single-entry records.
// SEM = false (geometry only) drops the two entry-id tables: 46 KB instead of 62 KB of LDS per block, i.e. three
// blocks per CU instead of two for a kernel that is bound by the latency of its atomics.
// threads of an accumulate block (one 16x8 tile): eight waves - the tile's 896 items in two passes, four slots per thread
// to publish; LDS allows three blocks per CU either way (measured: 256 threads 62 us, 512 56 us, 1024 68 us per frame)
constexpr int kAccThreads = 512;
// WCOMB (round 4, OJF_INTEGRATE_WAVE_COMBINE=1; off by default): colliding writes INSIDE a wave are combined in registers
// before they reach the LDS hash - measured 60.2 against 56.8 us per frame for the integrate stage at 320x240 -> 256^3 (the
// exchanges cost more VALU time than the same-address LDS atomics they spare; results identical bit for bit)
#ifdef OJF_ACC_STAMPS  // profiling build only (tools/acc_stamps.py): phase stamps of every accumulate block, thread 0
__device__ unsigned long long g_acc_stamps[4096][8];
#define ACC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_acc_stamps[blockIdx.x][i] = (i) >= 6 ? wall_clock64() : clock64(); } while (0)
__device__ unsigned long long g_acc_items[4096][8];
#define ACC_ITEM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && (i) < 8) g_acc_items[blockIdx.x][i] = clock64(); } while (0)
#else
#define ACC_STAMP(i) do { } while (0)
#define ACC_ITEM_STAMP(i) do { } while (0)
#endif
template <bool SEM, bool WCOMB>
__device__ __forceinline__ void accumulate_tiled_body(const IntegrateArgs &a, const Camera &cam)
{
    __shared__ unsigned int keys[kSlots];
    __shared__ unsigned long long accw[kSlots];
    __shared__ unsigned long long accu[kSlots];
    __shared__ unsigned int elast[SEM ? kSlots : 1];
    __shared__ unsigned int ediff[SEM ? kSlots : 1];
    __shared__ unsigned int n_entries, n_new, n_rec;
    __shared__ unsigned short recidx[kSlots];  // record of the slot inside the tile's slice (claim order)
    __shared__ double frame[6][kTilePix];  // ray frame (voxel-space point, unit direction) of the tile's pixels
    ACC_STAMP(6);
    ACC_STAMP(0);
    for (int s = threadIdx.x; s < kSlots; s += kAccThreads) {
        keys[s] = kEmpty; accw[s] = 0; accu[s] = 0;
        if constexpr (SEM) { elast[s] = 0; ediff[s] = 0; }
    }
    if (threadIdx.x == 0) { n_entries = 0; n_new = 0; n_rec = 0; }
    // the counter set of this call (header word kPhaseAcc: flipped by the previous call's finalize kernel, nobody writes it now)
    const unsigned int phase = a.phased ? a.counters[kPhaseAcc] & 1u : 0u;
    unsigned int *const counters = a.counters + 32 * phase;
    if (a.phased && blockIdx.x == 0 && threadIdx.x == 0) a.counters[kPhaseFin] = phase;  // (read by this call's finalize kernel only)
    const int tiles_x = (a.w + kTileW - 1) / kTileW;
    const int tile = banded_block_x();  // one band of the image per XCD: neighbouring tiles hit the same voxels
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    // Round 6: no numbering pass and no publish round of exchanges.  The thread that CLAIMS a hash slot numbers its record (the tile's own
    // slice of the record array) and fires the voxel's list-head exchange at once and files the returned predecessor in the record when the
    // item's eight corners are through: the ~830 k returning atomics of a frame (10.6 us of a block's 25.9 when they all went out in a
    // publish phase of their own, profiles/r06_accumulate_stamps.txt) now travel under the items' LDS work.
    const unsigned int base_rec = (unsigned int)tile * kSlots;
    // (records numbered densely in claim order - recidx[slot], 4 KB of LDS: 51 KB per geometry-only block, three of which fit a CU's 160 KB,
    // tools/microbench/lds_occupancy.hip - so that the finalize kernel finds two 32-byte records per line as before; records AT their slot's
    // position, 68 % of the slice occupied, cost that kernel 24 MB more fetch per frame)
    auto claim = [&](unsigned int sidx, unsigned int lin) {
        const unsigned int ri = atomicAdd(&n_rec, 1u);
        recidx[sidx] = (unsigned short)ri;
        return atomicExch(&a.head[lin], base_rec + ri + 1u);
    };
    auto file_prev = [&](int sidx, unsigned int prev) {
        a.recs[base_rec + recidx[sidx]].next = prev;
        // first touch of the voxel in this call: into the tile's slice of the first-touch list, in the order the exchanges come back -
        // the order of the items, i.e. of neighbouring rays: the finalize kernel's lanes then walk neighbouring voxels (a list in hash
        // order, which rounds 2-5 wrote, costs that kernel 2 us at 320x240 and 12 us at 640x480: profiles/r06_accumulate_stamps.txt)
        if (prev == 0) a.touched[base_rec + atomicAdd(&n_new, 1u)] = keys[sidx];
    };
    // the call's range-guard decision (kGuardLatch): this kernel fills the workspace only and never skips; the finalize kernel
    // reads the latch and leaves the volumes alone when the net's range guard had fired
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[kGuardLatch] = guard_set(a) ? 1u : 0u;
    if (threadIdx.x < kTilePix) {  // once per pixel instead of once per (pixel, sample): three fp64 divisions and a sqrt each
        const int p = threadIdx.x;
        int rl, cl;
            tile_pixel(p, rl, cl);
            const int r = ty * kTileH + rl, c = tx * kTileW + cl;
        if (r < a.h && c < a.w) {
            float pw[3];
            double cv[3], dir[3];
            unproject(r, c, frame_depth(a, r * a.w + c), cam, pw);
            ray_frame(pw, cam, cv, dir);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                frame[i][p] = cv[i];
                frame[3 + i][p] = dir[i];
            }
        }
    }
    __syncthreads();
    ACC_STAMP(1);

    constexpr bool sem = SEM;
    const int half = (a.n_points - 1) / 2;
    unsigned int n_in = 0;
    if constexpr (!WCOMB) {
        for (int item = threadIdx.x; item < kTilePix * a.n_tail; item += kAccThreads) {
            ACC_ITEM_STAMP(4 * (item / kAccThreads));
            const int k = item / kTilePix, p = item % kTilePix;
            int rl, cl;
            tile_pixel(p, rl, cl);
            const int r = ty * kTileH + rl, c = tx * kTileW + cl;
            if (r >= a.h || c >= a.w) continue;
            const int n = r * a.w + c;
            const float z = frame_depth(a, n);
            if (!(z != 0.0f)) continue;  // modules/pipeline.py:145-146
            const double cv[3] = {frame[0][p], frame[1][p], frame[2][p]};
            const double dir[3] = {frame[3][p], frame[4][p], frame[5][p]};
            RaySample s;
            ray_sample(cv, dir, k, half, s);
            float v = a.est[(size_t)n * a.est_stride + k];  // pipeline.py:153-156
            v = v < -a.trunc ? -a.trunc : (v > a.trunc ? a.trunc : v);
            const uint8_t id_e = sem ? a.sem_ids[n] : 0;
            const unsigned int e0 = ((unsigned int)n * a.n_tail + k) * 8u + 1u;
            ACC_ITEM_STAMP(4 * (item / kAccThreads) + 1);
            int won[8];
            unsigned int pv[8];
    #pragma unroll
            for (int q = 0; q < 8; ++q) won[q] = -1;
    #pragma unroll
            for (int q = 0; q < 8; ++q) {
                int64_t idx[3];
                double wq;
                corner(s, q, idx, wq);
                if (!in_volume(idx, a.X, a.Y, a.Z)) continue;  // integrator.py:48-53
                const unsigned int lin = (unsigned int)(((size_t)idx[0] * a.Y + (size_t)idx[1]) * a.Z + (size_t)idx[2]);
                const float we = (float)wq;  // integrator.py:45
                const float ue = we * v;     // integrator.py:55
                const unsigned long long xw = (unsigned long long)__double2ll_rn((double)we * kFixScale);
                const unsigned long long xu = (unsigned long long)__double2ll_rn((double)ue * kFixScale);
                const unsigned int e = e0 + q;
                const unsigned int ed = (sem && a.id_vol[lin] != id_e) ? e : 0u;  // integrator.py:105
                ++n_in;
                const unsigned int h0 = (lin * 2654435761u) >> 21;
                int slot = -1;
                for (int probe = 0; probe < 16; ++probe) {
                    const unsigned int sidx = (h0 + probe) & (kSlots - 1);
                    const unsigned int prev = atomicCAS(&keys[sidx], kEmpty, lin);
                    if (prev == kEmpty) { slot = (int)sidx; won[q] = slot; pv[q] = claim(sidx, lin); break; }  // (the exchange is on its way)
                    if (prev == lin) { slot = (int)sidx; break; }
                }
                if (slot >= 0) {
                    atomicAdd(&accw[slot], xw);
                    atomicAdd(&accu[slot], xu);
                    if constexpr (SEM) {
                        atomicMax(&elast[slot], e);
                        if (ed) atomicMax(&ediff[slot], ed);
                    }
                } else {  // hash full: a record of its own behind the tile slices (rare)
                    const unsigned int ridx = a.list_base + atomicAdd(&counters[2], 1u);
                    if (link_record(a, ridx, lin, xw, xu, e, ed)) a.touched[a.list_base + atomicAdd(&counters[0], 1u)] = lin;
                    if (a.stats) atomicAdd(&a.stats[2], 1u);
                }
            }
            ACC_ITEM_STAMP(4 * (item / kAccThreads) + 2);
    #pragma unroll
            for (int q = 0; q < 8; ++q)
                if (won[q] >= 0) file_prev(won[q], pv[q]);  // (first use of the exchanges' results: all eight were issued above)
            ACC_ITEM_STAMP(4 * (item / kAccThreads) + 3);
        }
    } else {
        const int lane = threadIdx.x & 63;
        // (wave-uniform trip count: the lanes of a wave are 64 consecutive pixels of the tile at ONE ray offset k, and every
        // lane of a running wave walks the eight corners - dead ones with `ok` false - because the lanes talk to each other)
        for (int item0 = threadIdx.x - lane; item0 < kTilePix * a.n_tail; item0 += kAccThreads) {
            const int item = item0 + lane;
            const int k = item / kTilePix, p = item % kTilePix;
            int rl, cl;
            tile_pixel(p, rl, cl);
            const int r = ty * kTileH + rl, c = tx * kTileW + cl;
            bool live = item < kTilePix * a.n_tail && r < a.h && c < a.w;
            const int n = live ? r * a.w + c : 0;
            const float z = live ? frame_depth(a, n) : 0.0f;
            live = live && z != 0.0f;  // modules/pipeline.py:145-146
            const double cv[3] = {frame[0][p], frame[1][p], frame[2][p]};
            const double dir[3] = {frame[3][p], frame[4][p], frame[5][p]};
            RaySample s;
            ray_sample(cv, dir, k, half, s);
            float v = live ? a.est[(size_t)n * a.est_stride + k] : 0.0f;  // pipeline.py:153-156
            v = v < -a.trunc ? -a.trunc : (v > a.trunc ? a.trunc : v);
            const uint8_t id_e = (sem && live) ? a.sem_ids[n] : 0;
            const unsigned int e0 = ((unsigned int)n * a.n_tail + k) * 8u + 1u;
            int won[8];
            unsigned int pv[8];
    #pragma unroll
            for (int q = 0; q < 8; ++q) won[q] = -1;
    #pragma unroll
            for (int q = 0; q < 8; ++q) {
                int64_t idx[3];
                double wq;
                corner(s, q, idx, wq);
                bool ok = live && in_volume(idx, a.X, a.Y, a.Z);  // integrator.py:48-53
                unsigned int lin = ok ? (unsigned int)(((size_t)idx[0] * a.Y + (size_t)idx[1]) * a.Z + (size_t)idx[2]) : kEmpty;
                const float we = (float)wq;  // integrator.py:45
                const float ue = we * v;     // integrator.py:55
                unsigned long long xw = (unsigned long long)__double2ll_rn((double)we * kFixScale);
                unsigned long long xu = (unsigned long long)__double2ll_rn((double)ue * kFixScale);
                unsigned int e = e0 + q;
                unsigned int ed = (sem && ok && a.id_vol[lin] != id_e) ? e : 0u;  // integrator.py:105
                if (ok) ++n_in;
                // Colliding writes INSIDE the wave are combined before they reach the LDS hash: neighbouring pixels' rays fall
                // into the same voxel more often than not (a voxel is ~3 pixels wide on the bench stream), and same-address LDS
                // atomics serialise (round 3's counters: half of the kernel's LDS-active cycles were such conflicts).  Two
                // butterfly steps over the lane's quad - lane ^ 1, then lane ^ 2, DPP quad permutes, no LDS traffic - : where
                // the partner holds the same voxel the lower lane takes both contributions (integer sums and entry-id maxima:
                // any combining order gives the same bits) and the upper lane drops out.
    #pragma unroll
                for (int stp = 0; stp < 2; ++stp) {
                    const int ctrl = stp == 0 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
                    auto dpp = [&](unsigned int x) {
                        return stp == 0 ? (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true)
                                        : (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);
                    };
                    (void)ctrl;
                    const unsigned int plin = dpp(ok ? lin : kEmpty);
                    const unsigned long long pxw = ((unsigned long long)dpp((unsigned int)(xw >> 32)) << 32) | dpp((unsigned int)xw);
                    const unsigned long long pxu = ((unsigned long long)dpp((unsigned int)(xu >> 32)) << 32) | dpp((unsigned int)xu);
                    unsigned int pe = 0, ped = 0;
                    if constexpr (SEM) { pe = dpp(e); ped = dpp(ed); }
                    const bool same = ok && plin == lin;  // (a dead partner sends kEmpty)
                    const bool upper = (lane >> stp) & 1;
                    if (same && !upper) {
                        xw += pxw; xu += pxu;
                        if constexpr (SEM) { e = pe > e ? pe : e; ed = ped > ed ? ped : ed; }
                    }
                    if (same && upper) ok = false;
                }
                if (!ok) continue;
                const unsigned int h0 = (lin * 2654435761u) >> 21;
                int slot = -1;
                for (int probe = 0; probe < 16; ++probe) {
                    const unsigned int sidx = (h0 + probe) & (kSlots - 1);
                    const unsigned int prev = atomicCAS(&keys[sidx], kEmpty, lin);
                    if (prev == kEmpty) { slot = (int)sidx; won[q] = slot; pv[q] = claim(sidx, lin); break; }  // (the exchange is on its way)
                    if (prev == lin) { slot = (int)sidx; break; }
                }
                if (slot >= 0) {
                    atomicAdd(&accw[slot], xw);
                    atomicAdd(&accu[slot], xu);
                    if constexpr (SEM) {
                        atomicMax(&elast[slot], e);
                        if (ed) atomicMax(&ediff[slot], ed);
                    }
                } else {  // hash full: a record of its own behind the tile slices (rare)
                    const unsigned int ridx = a.list_base + atomicAdd(&counters[2], 1u);
                    if (link_record(a, ridx, lin, xw, xu, e, ed)) a.touched[a.list_base + atomicAdd(&counters[0], 1u)] = lin;
                    if (a.stats) atomicAdd(&a.stats[2], 1u);
                }
            }
    #pragma unroll
            for (int q = 0; q < 8; ++q)
                if (won[q] >= 0) file_prev(won[q], pv[q]);  // (first use of the exchanges' results: all eight were issued above)
        }
    }
    if (n_in) atomicAdd(&n_entries, n_in);
    __syncthreads();
    ACC_STAMP(2);
    // publish the slots' sums (the records' `next` fields and the first-touch list were written while the items ran)
    unsigned int n_mine = 0;
#pragma unroll
    for (int j = 0; j < kSlots / kAccThreads; ++j) {
        const int s = threadIdx.x + kAccThreads * j;
        if (keys[s] == kEmpty) continue;
        VoxelRec *r = a.recs + base_rec + recidx[s];
        r->lin = keys[s]; r->w = accw[s]; r->u = accu[s];
        r->e_last = SEM ? elast[s] : 0u; r->e_diff = SEM ? ediff[s] : 0u;
        ++n_mine;
    }
    if (a.stats) {  // test / profiling only: same-line atomics cost 14 us per frame
        if (n_mine) atomicAdd(&a.stats[2], n_mine);
        if (threadIdx.x == 0) atomicAdd(&a.stats[1], n_entries);
    }
    ACC_STAMP(3);
    __syncthreads();
    ACC_STAMP(4);
    if (threadIdx.x == 0) a.tile_new[tile] = n_new;
    ACC_STAMP(5);
    ACC_STAMP(7);
}

// (blocks per CU by LDS: three geometry-only, two with the entry-id tables = six / four waves per SIMD; the second launch-bounds argument holds the register
// allocation to that - 80 VGPRs for six waves per SIMD - now that a lane keeps eight exchange results in flight)
template <bool SEM, bool WCOMB>
__global__ __launch_bounds__(kAccThreads, SEM ? 4 : 6) void integrate_accumulate_tiled_kernel(IntegrateArgs a, Camera cam)
{
    accumulate_tiled_body<SEM, WCOMB>(a, cam);
}

// One frame of each of up to OJF_MAX_SCENES scenes in one launch (ojf_integrate_many): blockIdx.y = scene, every scene with its
// own volumes, camera, est rows and workspace.  The same blocks run the same code per scene: the same bits as separate calls.
struct IntegrateMany { IntegrateArgs a[OJF_MAX_SCENES]; Camera cam[OJF_MAX_SCENES]; };
template <bool SEM>
__global__ __launch_bounds__(kAccThreads, SEM ? 4 : 6) void integrate_accumulate_many_kernel(IntegrateMany m)
{
    accumulate_tiled_body<SEM, false>(m.a[blockIdx.y], m.cam[blockIdx.y]);
}

// Entry-list variant for the reference's own Integrator.forward signature (modules/integrator.py:15-126):
// the caller hands over materialised updates - per row r (valid pixel x sample) a clamped value, 8 int64
// corner indices and 8 fp64 corner weights - exactly the tensors Pipeline._prepare_volume_update builds
// (pipeline.py:137-171).  Every in-volume entry becomes its own record (index r*8 + q: no counter);
// entry id = r*8 + q reproduces the reference's entry order for the semantic "last writer wins" rule.
struct EntryArgs {
    const float *values;     // [R]
    const int64_t *indices;  // [R, 8, 3]
    const double *weights;   // [R, 8]
    const uint8_t *row_ids;  // [R] or NULL
    int R;
};

__global__ __launch_bounds__(256) void integrate_entries_kernel(IntegrateArgs a, EntryArgs e)
{
    __shared__ unsigned int newlist[256 * 8];
    __shared__ unsigned int n_new, base_new, n_entries;
    if (threadIdx.x == 0) { n_new = 0; n_entries = 0; }
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int n_in = 0;
    if (r == 0) a.counters[kGuardLatch] = guard_set(a) ? 1u : 0u;  // (the call's range-guard decision: see kGuardLatch)
    if (r < e.R) {
        const float v = e.values[r];
        const bool sem = a.id_vol != nullptr;
        const uint8_t id_e = sem ? e.row_ids[r] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t *ix = e.indices + ((size_t)r * 8 + q) * 3;
            const int64_t idx[3] = {ix[0], ix[1], ix[2]};
            if (!in_volume(idx, a.X, a.Y, a.Z)) continue;  // integrator.py:48-53
            const unsigned int lin = (unsigned int)(((size_t)idx[0] * a.Y + (size_t)idx[1]) * a.Z + (size_t)idx[2]);
            const float we = (float)e.weights[(size_t)r * 8 + q];  // integrator.py:45
            const float ue = we * v;                               // integrator.py:55
            const unsigned int eid = (unsigned int)r * 8u + q + 1u;
            const unsigned int ed = (sem && a.id_vol[lin] != id_e) ? eid : 0u;
            if (link_record(a, (unsigned int)r * 8u + q, lin, (unsigned long long)__double2ll_rn((double)we * kFixScale),
                            (unsigned long long)__double2ll_rn((double)ue * kFixScale), eid, ed))
                newlist[atomicAdd(&n_new, 1u)] = lin;
            ++n_in;
        }
    }
    if (n_in) atomicAdd(&n_entries, n_in);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n_new) base_new = atomicAdd(&a.counters[0], n_new);
        if (a.stats && n_entries) atomicAdd(&a.stats[1], n_entries);
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n_new; i += 256) a.touched[a.list_base + base_new + i] = newlist[i];
}

__device__ __forceinline__ void finalize_voxel(const IntegrateArgs &a, size_t lin, unsigned int per_pixel, bool sem, bool skip)
{
    unsigned int ri = a.head[lin];
    a.head[lin] = 0;  // leave the workspace clean
    if (skip) return;  // (block-uniform: the call's range-guard latch) the records are dropped, the volumes stay as they are
    const unsigned int first = ri;
    long long sw = 0, su = 0;
    bool wrapped = false;
    unsigned int e_last = 0, e_diff = 0;
    while (ri) {  // 2-3 records per voxel; integer sums: any order gives the same bits
        const VoxelRec r = a.recs[ri - 1];
        wrapped |= __builtin_add_overflow(sw, (long long)r.w, &sw);
        wrapped |= __builtin_add_overflow(su, (long long)r.u, &su);
        e_last = r.e_last > e_last ? r.e_last : e_last;
        e_diff = r.e_diff > e_diff ? r.e_diff : e_diff;
        ri = r.next;
    }
    double Wd = (double)sw * kFixInv, Ud = (double)su * kFixInv;
    if (wrapped) {
        // > 5e5 of per-frame weight on one voxel (a degenerate frame: 19 integer bits of the 2^-44 fixed point): the
        // 64-bit sum wrapped.  Redo it in fp64 over the records - each of them is exact, a tile cannot overflow on its
        // own - so that the weight saturates to fp16 infinity like the reference's instead of coming out garbage.
        Wd = 0.0; Ud = 0.0;
        for (unsigned int rj = first; rj;) {
            const VoxelRec r = a.recs[rj - 1];
            Wd += (double)(long long)r.w * kFixInv;
            Ud += (double)(long long)r.u * kFixInv;
            rj = r.next;
        }
    }
    const float W = (float)Wd;
    const float U = (float)Ud;
    const float w_old = h2f(a.wgt[lin]), v_old = h2f(a.tsdf[lin]);  // integrator.py:72-75
    const float w_new = w_old + W;                                   // :77
    const float num = w_old * v_old + U;                             // :82
    a.wgt[lin] = f2h(w_new);                                         // :78,87
    a.tsdf[lin] = f2h(num / w_new);                                  // :83,88
    if (sem) {  // integrator.py:93-124 with "highest entry wins" for duplicates
        const float s_old = h2f(a.score_vol[lin]);
        const float s_last = a.sem_scores[(e_last - 1u) / per_pixel];
        a.score_vol[lin] = f2h(s_last > s_old ? s_last : s_old);     // :113-114,124
        if (e_diff) {                                                 // :105,116-117,123
            const unsigned int n_d = (e_diff - 1u) / per_pixel;
            if (a.sem_scores[n_d] > s_old) a.id_vol[lin] = a.sem_ids[n_d];
        }
    }
}

// K voxels per lane, their dependent loads interleaved: a voxel is a chain of ~6 dependent memory round trips (touched list ->
// head -> 2-3 records -> volumes) and a 16x8 tile first-touches ~320 voxels - with one voxel per lane a 256-thread block
// walked them in two rounds of one chain each and the kernel sat 74 % parked (round 4's counters); two chains per lane
// are one round with twice the loads in flight.  Same operations per voxel as finalize_voxel: same bits.
#ifdef OJF_ACC_STAMPS
__device__ unsigned long long g_fin_stamps[4096][8];
#define FIN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_fin_stamps[blockIdx.x][i] = (i) >= 6 ? wall_clock64() : clock64(); } while (0)
#else
#define FIN_STAMP(i) do { } while (0)
#endif
template <int K>
__device__ __forceinline__ void finalize_voxels(const IntegrateArgs &a, const unsigned int *list, unsigned int i0, unsigned int stride,
                                                unsigned int n, unsigned int per_pixel, bool sem, bool skip)
{
    size_t lin[K];
    unsigned int ri[K], first[K];
    bool on[K];
    float w_old[K], v_old[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        on[k] = i0 + k * stride < n;
        lin[k] = on[k] ? list[i0 + k * stride] : 0;
    }
    if (skip) {  // (block-uniform: the call's range-guard latch) drop the records, leave the workspace clean, touch no volume
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (on[k]) a.head[lin[k]] = 0;
        return;
    }
    FIN_STAMP(1);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ri[k] = on[k] ? a.head[lin[k]] : 0u;
        // the pre-frame values do not depend on the records: requested now, used after the walk
        w_old[k] = on[k] ? h2f(a.wgt[lin[k]]) : 0.0f;
        v_old[k] = on[k] ? h2f(a.tsdf[lin[k]]) : 0.0f;
        first[k] = ri[k];
    }
    long long sw[K], su[K];
    bool wrapped[K];
    unsigned int e_last[K], e_diff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { sw[k] = 0; su[k] = 0; wrapped[k] = false; e_last[k] = 0; e_diff[k] = 0; }
    bool any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) any = any || ri[k] != 0;
    FIN_STAMP(2);
    while (any) {  // 2-3 records per voxel; integer sums: any order gives the same bits
        VoxelRec r[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (ri[k]) r[k] = a.recs[ri[k] - 1];
        any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!ri[k]) continue;
            wrapped[k] |= __builtin_add_overflow(sw[k], (long long)r[k].w, &sw[k]);
            wrapped[k] |= __builtin_add_overflow(su[k], (long long)r[k].u, &su[k]);
            e_last[k] = r[k].e_last > e_last[k] ? r[k].e_last : e_last[k];
            e_diff[k] = r[k].e_diff > e_diff[k] ? r[k].e_diff : e_diff[k];
            ri[k] = r[k].next;
            any = any || ri[k] != 0;
        }
    }
    FIN_STAMP(3);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!on[k]) continue;
        a.head[lin[k]] = 0;  // leave the workspace clean
        double Wd = (double)sw[k] * kFixInv, Ud = (double)su[k] * kFixInv;
        if (wrapped[k]) {  // see finalize_voxel
            Wd = 0.0; Ud = 0.0;
            for (unsigned int rj = first[k]; rj;) {
                const VoxelRec r = a.recs[rj - 1];
                Wd += (double)(long long)r.w * kFixInv;
                Ud += (double)(long long)r.u * kFixInv;
                rj = r.next;
            }
        }
        const float W = (float)Wd;
        const float U = (float)Ud;
        const float w_new = w_old[k] + W;                                   // integrator.py:77
        const float num = w_old[k] * v_old[k] + U;                          // :82
        a.wgt[lin[k]] = f2h(w_new);                                         // :78,87
        a.tsdf[lin[k]] = f2h(num / w_new);                                  // :83,88
        if (sem) {  // integrator.py:93-124 with "highest entry wins" for duplicates
            const float s_old = h2f(a.score_vol[lin[k]]);
            const float s_last = a.sem_scores[(e_last[k] - 1u) / per_pixel];
            a.score_vol[lin[k]] = f2h(s_last > s_old ? s_last : s_old);     // :113-114,124
            if (e_diff[k]) {                                                 // :105,116-117,123
                const unsigned int n_d = (e_diff[k] - 1u) / per_pixel;
                if (a.sem_scores[n_d] > s_old) a.id_vol[lin[k]] = a.sem_ids[n_d];
            }
        }
    }
}

// first the per-tile slices (the voxels each tile touched first), then the counter-allocated list
__device__ __forceinline__ void finalize_body(const IntegrateArgs &a)
{
    const unsigned int per_pixel = (unsigned int)a.n_tail * 8u;  // entries per sem_ids / sem_scores element
    const bool sem = a.id_vol != nullptr;
    // the call's range-guard decision, latched by ONE thread of the kernel that filled the workspace (kGuardLatch): stable while
    // this kernel runs, the same for every block - all of the call's voxels are updated or none
    const bool skip = a.guard != nullptr && a.counters[kGuardLatch] != 0u;
    constexpr int K = 2;
    FIN_STAMP(6);
    FIN_STAMP(0);
    for (int tile = banded_block_x(); tile < a.n_tiles; tile += gridDim.x) {  // (the XCD that accumulated the tile)
        const unsigned int n = a.tile_new[tile];
        const unsigned int *list = a.touched + (size_t)tile * kSlots;
        for (unsigned int i = threadIdx.x; i < n; i += K * blockDim.x) finalize_voxels<K>(a, list, i, blockDim.x, n, per_pixel, sem, skip);
        if (a.stats && threadIdx.x == 0 && n) atomicAdd(&a.stats[0], n);
    }
    FIN_STAMP(4);
    FIN_STAMP(7);
    // (header word kPhaseFin: written by this call's accumulate kernel, stable while finalize runs)
    const unsigned int phase = a.phased ? a.counters[kPhaseFin] & 1u : 0u;
    const unsigned int count = a.counters[32 * phase];
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x)
        finalize_voxel(a, a.touched[a.list_base + t], per_pixel, sem, skip);
    if (a.stats && blockIdx.x == 0 && threadIdx.x == 0 && count) atomicAdd(&a.stats[0], count);
    if (skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.guard + 1, 1);  // one more call whose frame did not reach the volumes
    // the counter set the NEXT call will use (nobody touches it during this one): clean it here instead of a memset
    // launch in front of every frame's accumulate kernel
    if (a.phased && blockIdx.x == 0 && threadIdx.x < 32) a.counters[32 * (1 - phase) + threadIdx.x] = 0;
    if (a.phased && blockIdx.x == 0 && threadIdx.x == 0) a.counters[kPhaseAcc] = phase ^ 1u;  // (the next call's accumulate kernel reads it)
}

__global__ __launch_bounds__(256) void integrate_finalize_kernel(IntegrateArgs a) { finalize_body(a); }

struct FinalizeMany { IntegrateArgs a[OJF_MAX_SCENES]; };
__global__ __launch_bounds__(256) void integrate_finalize_many_kernel(FinalizeMany m) { finalize_body(m.a[blockIdx.y]); }

}  // namespace ojf

namespace ojf {
// FAST frame path: the header holds two counter sets used alternately - a call counts in one and its finalize kernel
// zeroes the other, so no memset launch sits between the net and the accumulate kernel (~8 us per frame).  Which set
// comes next is DEVICE state since round 4 (two phase words in the header, each written by one kernel and read by the
// other: rounds 1-3 kept the phase on the host per workspace address, which is why the call refused stream capture): the
// call sequence is the same every time and can be captured into a HIP graph.  Any phase is valid on a zeroed header
// (fresh workspace, ojf_integrate_workspace_init, the entry-list and PARITY paths, which zero the header themselves).
}  // namespace ojf

OJF_API size_t ojf_integrate_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail, int mode)
{
    using namespace ojf;
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0 || n_tail <= 0) return 0;
    if (mode == OJF_MODE_FAST) return fast_workspace_bytes(X, Y, Z, h, w, n_tail);
    if (mode == OJF_MODE_PARITY) return parity_workspace_bytes(X, Y, Z, h, w, n_tail);
    return 0;
}

OJF_API int ojf_integrate_workspace_init(void *ws, size_t ws_bytes, int X, int Y, int Z, int h, int w,
                                         int n_tail, int mode, ojf_stream_t stream)
{
    using namespace ojf;
    if (!ws) return fail("ojf_integrate_workspace_init: null workspace");
    const size_t need = ojf_integrate_workspace_bytes(X, Y, Z, h, w, n_tail, mode);
    if (need == 0) return fail("ojf_integrate_workspace_init: bad sizes or mode");
    if (ws_bytes < need) return fail("ojf_integrate_workspace_init: workspace too small");
    // FAST: header + dense head table must start zeroed (every call restores that invariant)
    const size_t zero_bytes = mode == OJF_MODE_FAST ? kHeaderBytes + (size_t)X * Y * Z * sizeof(unsigned int)
                                                    : kHeaderBytes;
    return check_hip(hipMemsetAsync(ws, 0, zero_bytes, as_stream(stream)), "workspace memset");
}

OJF_API int ojf_integrate(const float *depth_filtered, const float *Ki, const float *E, const double *origin,
                          double res, const float *est, int est_stride, int n_points, int n_tail, float trunc,
                          uint16_t *tsdf, uint16_t *wgt, const uint8_t *sem_ids, const float *sem_scores,
                          uint8_t *id_vol, uint16_t *score_vol, int X, int Y, int Z, int h, int w, int mode,
                          void *ws, size_t ws_bytes, uint32_t *stats, ojf_stream_t stream)
{
    return ojf_integrate_masked(depth_filtered, nullptr, Ki, E, origin, res, est, est_stride, n_points, n_tail, trunc, tsdf, wgt,
                                sem_ids, sem_scores, id_vol, score_vol, X, Y, Z, h, w, mode, ws, ws_bytes, stats, stream);
}

OJF_API int ojf_integrate_masked(const float *depth_filtered, const uint8_t *mask, const float *Ki, const float *E,
                                 const double *origin, double res, const float *est, int est_stride, int n_points, int n_tail,
                                 float trunc, uint16_t *tsdf, uint16_t *wgt, const uint8_t *sem_ids, const float *sem_scores,
                                 uint8_t *id_vol, uint16_t *score_vol, int X, int Y, int Z, int h, int w, int mode, void *ws,
                                 size_t ws_bytes, uint32_t *stats, ojf_stream_t stream)
{
    using namespace ojf;
    if (!depth_filtered || !Ki || !E || !origin || !est || !tsdf || !wgt || !ws)
        return fail("ojf_integrate: null pointer argument");
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0) return fail("ojf_integrate: non-positive size");
    if (n_points < 1 || (n_points & 1) == 0) return fail("ojf_integrate: n_points must be odd and >= 1");
    if (n_tail < 1 || n_tail > n_points) return fail("ojf_integrate: need 1 <= n_tail <= n_points");
    if (est_stride < n_tail) return fail("ojf_integrate: est_stride < n_tail");
    if (!(res > 0.0)) return fail("ojf_integrate: resolution must be > 0");
    if ((uint64_t)X * Y * Z >= 0xffffffffull) return fail("ojf_integrate: volume has >= 2^32 voxels");
    if ((uint64_t)h * w * n_tail * 8 >= 0xffffffffull) return fail("ojf_integrate: frame too large");
    const int n_sem = (sem_ids != nullptr) + (sem_scores != nullptr) + (id_vol != nullptr) + (score_vol != nullptr);
    if (n_sem != 0 && n_sem != 4)
        return fail("ojf_integrate: sem_ids, sem_scores, id_vol, score_vol must be all set or all NULL");
    const size_t need = ojf_integrate_workspace_bytes(X, Y, Z, h, w, n_tail, mode);
    if (need == 0) return fail("ojf_integrate: unknown mode");
    if (ws_bytes < need) return fail("ojf_integrate: workspace too small");

    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(ws);
    IntegrateArgs a;
    a.depth = depth_filtered; a.mask = mask; a.est = est; a.tsdf = tsdf; a.wgt = wgt;
    a.sem_ids = sem_ids; a.sem_scores = sem_scores; a.id_vol = id_vol; a.score_vol = score_vol;
    a.counters = reinterpret_cast<unsigned int *>(base);
    a.counters_next = nullptr; a.phased = 0;
    a.guard = const_cast<int *>(range_guard_if_any());
    a.head = nullptr; a.recs = nullptr; a.touched = nullptr; a.tile_new = nullptr; a.list_base = 0; a.n_tiles = 0; a.stats = stats;
    a.X = X; a.Y = Y; a.Z = Z; a.h = h; a.w = w; a.n_points = n_points; a.n_tail = n_tail;
    a.est_stride = est_stride; a.trunc = trunc;
    const Camera cam = make_camera(Ki, E, origin, res);

    if (mode == OJF_MODE_PARITY) {
        OJF_HIP(hipMemsetAsync(a.counters, 0, kHeaderBytes, st));
        return integrate_parity(a, cam, base + kHeaderBytes, ws_bytes - kHeaderBytes, st);
    }
    a.phased = 1;  // (a.counters = the header)

    const int tiles = (int)tile_count(h, w);
    {
        const size_t nvox = (size_t)X * Y * Z, cap = list_capacity(h, w, n_tail);
        char *q = base + kHeaderBytes;
        a.head = reinterpret_cast<unsigned int *>(q); q += nvox * sizeof(unsigned int);
        a.recs = reinterpret_cast<VoxelRec *>(q); q += cap * sizeof(VoxelRec);
        a.touched = reinterpret_cast<unsigned int *>(q); q += cap * sizeof(unsigned int);
        a.tile_new = reinterpret_cast<unsigned int *>(q);
        a.n_tiles = tiles;
        a.list_base = (unsigned int)tiles * kSlots;
    }
    if (stats) OJF_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), st));
    static const bool wcomb = getenv("OJF_INTEGRATE_WAVE_COMBINE") != nullptr;  // measured slower: see the kernel
    if (id_vol && wcomb) hipLaunchKernelGGL((integrate_accumulate_tiled_kernel<true, true>), dim3(tiles), dim3(kAccThreads), 0, st, a, cam);
    else if (id_vol) hipLaunchKernelGGL((integrate_accumulate_tiled_kernel<true, false>), dim3(tiles), dim3(kAccThreads), 0, st, a, cam);
    else if (wcomb) hipLaunchKernelGGL((integrate_accumulate_tiled_kernel<false, true>), dim3(tiles), dim3(kAccThreads), 0, st, a, cam);
    else hipLaunchKernelGGL((integrate_accumulate_tiled_kernel<false, false>), dim3(tiles), dim3(kAccThreads), 0, st, a, cam);
    OJF_HIP(hipGetLastError());
    hipLaunchKernelGGL(integrate_finalize_kernel, dim3(tiles < 1024 ? 1024 : tiles), dim3(256), 0, st, a);
    return check_hip(hipGetLastError(), "ojf_integrate launch");
}

// One frame of each of n scenes (1 <= n <= OJF_MAX_SCENES; one frame / grid size): ojf_integrate_masked's two FAST kernels with
// the scene as blockIdx.y (include/ojf.h).
OJF_API int ojf_integrate_many(int n, const ojf_integrate_job *jobs, int n_points, int n_tail, float trunc, int X, int Y, int Z, int h,
                               int w, ojf_stream_t stream)
{
    using namespace ojf;
    if (n < 1 || n > OJF_MAX_SCENES || !jobs) return fail("ojf_integrate_many: 1..OJF_MAX_SCENES jobs");
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0) return fail("ojf_integrate_many: non-positive size");
    if (n_points < 1 || (n_points & 1) == 0) return fail("ojf_integrate_many: n_points must be odd and >= 1");
    if (n_tail < 1 || n_tail > n_points) return fail("ojf_integrate_many: need 1 <= n_tail <= n_points");
    if ((uint64_t)X * Y * Z >= 0xffffffffull) return fail("ojf_integrate_many: volume has >= 2^32 voxels");
    if ((uint64_t)h * w * n_tail * 8 >= 0xffffffffull) return fail("ojf_integrate_many: frame too large");
    const size_t need = fast_workspace_bytes(X, Y, Z, h, w, n_tail);
    const int tiles = (int)tile_count(h, w);
    const size_t nvox = (size_t)X * Y * Z, cap = list_capacity(h, w, n_tail);
    IntegrateMany m;
    FinalizeMany f;
    int n_with_sem = 0;
    for (int i = 0; i < n; ++i) {
        const ojf_integrate_job &j = jobs[i];
        if (!j.depth_dev || !j.Kinv_host || !j.E_host || !j.origin_host || !j.est_dev || !j.tsdf_dev || !j.weights_dev || !j.workspace_dev)
            return fail("ojf_integrate_many: null pointer in a job");
        if (j.est_stride < n_tail) return fail("ojf_integrate_many: est_stride < n_tail");
        if (!(j.resolution > 0.0)) return fail("ojf_integrate_many: resolution must be > 0");
        const int n_sem = (j.sem_ids_dev != nullptr) + (j.sem_scores_dev != nullptr) + (j.id_vol_dev != nullptr) + (j.score_vol_dev != nullptr);
        if (n_sem != 0 && n_sem != 4) return fail("ojf_integrate_many: sem_ids, sem_scores, id_vol, score_vol must be all set or all NULL");
        n_with_sem += n_sem == 4;
        if (j.workspace_bytes < need) return fail("ojf_integrate_many: workspace too small");
        for (int k = 0; k < i; ++k)
            if (jobs[k].workspace_dev == j.workspace_dev || jobs[k].tsdf_dev == j.tsdf_dev || jobs[k].weights_dev == j.weights_dev)
                return fail("ojf_integrate_many: two jobs share a workspace or a volume (one frame per SCENE, one workspace per job)");
        IntegrateArgs &a = m.a[i];
        char *base = static_cast<char *>(j.workspace_dev);
        a.depth = j.depth_dev; a.mask = j.mask_dev; a.est = j.est_dev; a.tsdf = j.tsdf_dev; a.wgt = j.weights_dev;
        a.sem_ids = j.sem_ids_dev; a.sem_scores = j.sem_scores_dev; a.id_vol = j.id_vol_dev; a.score_vol = j.score_vol_dev;
        a.counters = reinterpret_cast<unsigned int *>(base);
        a.counters_next = nullptr; a.phased = 1;
        a.guard = const_cast<int *>(range_guard_if_any());
        char *q = base + kHeaderBytes;
        a.head = reinterpret_cast<unsigned int *>(q); q += nvox * sizeof(unsigned int);
        a.recs = reinterpret_cast<VoxelRec *>(q); q += cap * sizeof(VoxelRec);
        a.touched = reinterpret_cast<unsigned int *>(q); q += cap * sizeof(unsigned int);
        a.tile_new = reinterpret_cast<unsigned int *>(q);
        a.n_tiles = tiles;
        a.list_base = (unsigned int)tiles * kSlots;
        a.stats = nullptr;
        a.X = X; a.Y = Y; a.Z = Z; a.h = h; a.w = w; a.n_points = n_points; a.n_tail = n_tail;
        a.est_stride = j.est_stride; a.trunc = trunc;
        m.cam[i] = make_camera(j.Kinv_host, j.E_host, j.origin_host, j.resolution);
        f.a[i] = a;
    }
    if (n_with_sem != 0 && n_with_sem != n) return fail("ojf_integrate_many: semantics for all jobs or for none");
    hipStream_t st = as_stream(stream);
    if (n_with_sem) hipLaunchKernelGGL((integrate_accumulate_many_kernel<true>), dim3(tiles, n), dim3(kAccThreads), 0, st, m);
    else hipLaunchKernelGGL((integrate_accumulate_many_kernel<false>), dim3(tiles, n), dim3(kAccThreads), 0, st, m);
    OJF_HIP(hipGetLastError());
    hipLaunchKernelGGL(integrate_finalize_many_kernel, dim3(tiles < 1024 ? 1024 : tiles, n), dim3(256), 0, st, f);
    return check_hip(hipGetLastError(), "ojf_integrate_many launch");
}

OJF_API int ojf_integrate_entries(const float *values, const int64_t *indices, const double *weights,
                                  const uint8_t *row_ids, const float *row_scores, int64_t n_rows, uint16_t *tsdf,
                                  uint16_t *wgt, uint8_t *id_vol, uint16_t *score_vol, int X, int Y, int Z, void *ws,
                                  size_t ws_bytes, uint32_t *stats, ojf_stream_t stream)
{
    using namespace ojf;
    if (!values || !indices || !weights || !tsdf || !wgt || !ws) return fail("ojf_integrate_entries: null pointer argument");
    if (X <= 0 || Y <= 0 || Z <= 0 || n_rows < 0) return fail("ojf_integrate_entries: bad sizes");
    if ((uint64_t)X * Y * Z >= 0xffffffffull) return fail("ojf_integrate_entries: volume has >= 2^32 voxels");
    if ((uint64_t)n_rows * 8 >= 0xffffffffull) return fail("ojf_integrate_entries: too many rows");
    const int n_sem = (row_ids != nullptr) + (row_scores != nullptr) + (id_vol != nullptr) + (score_vol != nullptr);
    if (n_sem != 0 && n_sem != 4)
        return fail("ojf_integrate_entries: row_ids, row_scores, id_vol, score_vol must be all set or all NULL");
    const size_t nvox = (size_t)X * Y * Z;
    const size_t entries = (size_t)n_rows * 8;
    const size_t need = kHeaderBytes + nvox * sizeof(unsigned int) + entries * sizeof(VoxelRec) +
                        (entries < nvox ? entries : nvox) * sizeof(unsigned int);
    if (ws_bytes < need) return fail("ojf_integrate_entries: workspace too small");
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(ws);
    IntegrateArgs a;
    a.depth = nullptr; a.mask = nullptr; a.est = nullptr; a.tsdf = tsdf; a.wgt = wgt;
    a.sem_ids = row_ids; a.sem_scores = row_scores; a.id_vol = id_vol; a.score_vol = score_vol;
    a.counters = reinterpret_cast<unsigned int *>(base);
    a.head = reinterpret_cast<unsigned int *>(base + kHeaderBytes);
    a.recs = reinterpret_cast<VoxelRec *>(base + kHeaderBytes + nvox * sizeof(unsigned int));
    a.touched = reinterpret_cast<unsigned int *>(base + kHeaderBytes + nvox * sizeof(unsigned int) + entries * sizeof(VoxelRec));
    a.stats = stats; a.tile_new = nullptr; a.list_base = 0; a.n_tiles = 0; a.counters_next = nullptr; a.phased = 0;
    a.guard = const_cast<int *>(range_guard_if_any());
    a.X = X; a.Y = Y; a.Z = Z; a.h = 1; a.w = 1; a.n_points = 1; a.est_stride = 0; a.trunc = 0.0f;
    a.n_tail = 1;  // finalize maps entry id -> row as (id - 1) / (n_tail * 8)
    OJF_HIP(hipMemsetAsync(base, 0, kHeaderBytes, st));
    // (the header was zeroed above: both counter sets clean, phase 0 - the frame path may share this workspace)
    if (stats) OJF_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), st));
    if (n_rows > 0) {
        EntryArgs e{values, indices, weights, row_ids, (int)n_rows};
        hipLaunchKernelGGL(integrate_entries_kernel, dim3((unsigned int)((n_rows + 255) / 256)), dim3(256), 0, st, a, e);
        OJF_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(integrate_finalize_kernel, dim3(1024), dim3(256), 0, st, a);
    return check_hip(hipGetLastError(), "ojf_integrate_entries launch");
}

#ifdef OJF_ACC_STAMPS
extern "C" __attribute__((visibility("default"))) int ojf_debug_acc_stamps(void *host_dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ojf::g_acc_stamps), bytes < sizeof(ojf::g_acc_stamps) ? bytes : sizeof(ojf::g_acc_stamps));
}
extern "C" __attribute__((visibility("default"))) int ojf_debug_acc_item_stamps(void *host_dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ojf::g_acc_items), bytes < sizeof(ojf::g_acc_items) ? bytes : sizeof(ojf::g_acc_items));
}
extern "C" __attribute__((visibility("default"))) int ojf_debug_fin_stamps(void *host_dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ojf::g_fin_stamps), bytes < sizeof(ojf::g_fin_stamps) ? bytes : sizeof(ojf::g_fin_stamps));
}
#endif
