// INTEGRATE, PARITY mode: per-voxel fp32 sums accumulated strictly in the reference's entry order
// (valid pixel ascending, sample k, corner q - modules/integrator.py:38-67 on one CPU thread), so
// the post-frame fp16 TSDF / weight volumes are bit-identical to the reference.
//
// Pipeline: emit (voxel key, {entry id, weight}) for every scatter slot in entry order ->
// stable LSD radix sort on the voxel key (rocPRIM device primitive; stability keeps the entry
// order inside each voxel's run) -> one lane per run walks it sequentially.
// This is the validation mode: it proves that FAST mode differs from the reference by summation
// order only.  It trades ~3 extra passes over the entry list for exactness.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ojf_integrate.h"

namespace ojf {

struct EntryVal {
    unsigned int e;   // entry id = (n*n_tail + k)*8 + q
    float we;         // fp32 corner weight (integrator.py:45)
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static unsigned int key_bits(size_t nvox)
{
    unsigned int b = 1;
    while (((size_t)1 << b) <= nvox) ++b;  // keys span [0, nvox] (nvox = sentinel)
    return b;
}

struct ParityLayout {
    size_t keys_in, keys_out, vals_in, vals_out, temp, temp_bytes, total;
};

static int parity_layout(int X, int Y, int Z, int h, int w, int n_tail, ParityLayout &L, hipStream_t stream)
{
    const size_t nvox = (size_t)X * Y * Z;
    const size_t M = (size_t)h * w * n_tail * 8;
    size_t off = 0;
    L.keys_in = off; off = align_up(off + M * sizeof(unsigned int), 256);
    L.keys_out = off; off = align_up(off + M * sizeof(unsigned int), 256);
    L.vals_in = off; off = align_up(off + M * sizeof(EntryVal), 256);
    L.vals_out = off; off = align_up(off + M * sizeof(EntryVal), 256);
    size_t temp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned int *)nullptr, (unsigned int *)nullptr,
                                             (EntryVal *)nullptr, (EntryVal *)nullptr, M, 0u, key_bits(nvox), stream);
    if (e != hipSuccess) return check_hip(e, "rocprim::radix_sort_pairs (size query)");
    L.temp = off;
    L.temp_bytes = temp_bytes;
    L.total = align_up(off + temp_bytes, 256);
    return 0;
}

size_t parity_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail)
{
    ParityLayout L;
    if (parity_layout(X, Y, Z, h, w, n_tail, L, nullptr)) return 0;
    return kHeaderBytes + L.total;
}

__global__ __launch_bounds__(256) void parity_emit_kernel(IntegrateArgs a, Camera cam, unsigned int *keys, EntryVal *vals,
                                                           unsigned int sentinel)
{
    const int N = a.h * a.w;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= N * a.n_tail) return;
    const int n = item / a.n_tail;   // pixel-major so that slot order == entry order
    const int k = item - n * a.n_tail;
    const size_t slot0 = (size_t)item * 8;
    const float z = frame_depth(a, n);
    if (item == 0) a.counters[kGuardLatch] = guard_set(a) ? 1u : 0u;  // the call's range-guard decision (ojf_integrate.h kGuardLatch); this kernel never skips
    const bool valid = z != 0.0f;
    RaySample s;
    if (valid) {
        const int r = n / a.w, c = n - r * a.w;
        float pw[3];
        double cv[3], dir[3];
        unproject(r, c, z, cam, pw);
        ray_frame(pw, cam, cv, dir);
        ray_sample(cv, dir, k, (a.n_points - 1) / 2, s);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned int key = sentinel;
        EntryVal ev{(unsigned int)(slot0 + q), 0.0f};
        if (valid) {
            int64_t idx[3];
            double wq;
            corner(s, q, idx, wq);
            if (in_volume(idx, a.X, a.Y, a.Z)) {
                key = (unsigned int)(((size_t)idx[0] * a.Y + (size_t)idx[1]) * a.Z + (size_t)idx[2]);
                ev.we = (float)wq;
            }
        }
        keys[slot0 + q] = key;
        vals[slot0 + q] = ev;
    }
}

// One WAVE per 64 consecutive slots of the sorted entry list (round 6; rounds 1-5: one lane per slot, the run heads walking their runs
// alone - 3 busy lanes of 64 on a chain of dependent loads, 1.5 ms per frame).  The wave loads its 64 entries with one coalesced request,
// every lane forms ITS entry's  ue = we * clamp(est)  (integrator.py:55) in parallel, and each run whose head lies in the chunk is then
// summed in entry order with wave-uniform fp32 adds over the lanes' values (v_readlane): the same operations in the same order as the
// reference's sequential index_add_, without a dependent load per entry.  A run that leaves the chunk pulls the next 64 slots.
__device__ __forceinline__ float lane_f(float x, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j)); }

__global__ __launch_bounds__(256) void parity_walk_kernel(IntegrateArgs a, const unsigned int *keys, const EntryVal *vals,
                                                           size_t M, unsigned int sentinel)
{
    if (a.guard && a.counters[kGuardLatch]) {  // (uniform) the net's range guard had fired when this call began: no volume is touched
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.guard + 1, 1);
        return;
    }
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    if (base >= M) return;  // (wave-uniform)
    const bool sem = a.id_vol != nullptr;
    const unsigned int per_pixel = (unsigned int)a.n_tail * 8u;
    // this lane's entry of a chunk: key, weight, update, semantic label / score of its pixel
    auto load = [&](size_t at, unsigned int &key, float &we, float &ue, unsigned int &id_e, float &s_e) {
        key = sentinel; we = 0.0f; ue = 0.0f; id_e = 0; s_e = 0.0f;
        if (at < M) key = keys[at];
        if (key != sentinel) {
            const EntryVal ev = vals[at];
            const unsigned int n = ev.e / per_pixel;
            const unsigned int k = (ev.e / 8u) % (unsigned int)a.n_tail;
            float v = a.est[(size_t)n * a.est_stride + k];  // pipeline.py:153-156
            v = v < -a.trunc ? -a.trunc : (v > a.trunc ? a.trunc : v);
            we = ev.we;
            ue = ev.we * v;  // integrator.py:55
            if (sem) { id_e = a.sem_ids[n]; s_e = a.sem_scores[n]; }
        }
    };
    unsigned int key, id_e;
    float we, ue, s_e;
    load(base + lane, key, we, ue, id_e, s_e);
    unsigned int prevkey = (unsigned int)__shfl_up((int)key, 1, 64);
    if (lane == 0) prevkey = base > 0 ? keys[base - 1] : sentinel;
    unsigned long long heads = __ballot(key != sentinel && key != prevkey);
    unsigned int runs = 0, entries = 0;
    while (heads) {  // (wave-uniform) the runs that begin in this chunk, in order
        const int h = __builtin_amdgcn_readfirstlane(__builtin_ctzll(heads));
        heads &= heads - 1;
        const unsigned int key_h = (unsigned int)__builtin_amdgcn_readlane((int)key, h);
        const size_t lin = key_h;
        float Wsum = 0.0f, Usum = 0.0f;  // integrator.py:59-67, sequential fp32 in entry order
        unsigned int id_old = 0, id_new = 0;
        float s_old = 0.0f;
        uint16_t sc_new = 0;
        if (sem) {
            id_old = a.id_vol[lin];
            id_new = id_old;
            sc_new = a.score_vol[lin];
            s_old = h2f(sc_new);
        }
        unsigned int n_run = 0;
        // the run's slots inside this chunk are the contiguous lanes h .. end - 1 (sorted keys)
        int end = h + __builtin_popcountll(__ballot(key == key_h && lane >= h));
        auto add = [&](float we_c, float ue_c, unsigned int id_c, float s_c, int from, int to) {
            for (int j = from; j < to; ++j) {  // (wave-uniform: every lane forms the same sums)
                Wsum += lane_f(we_c, j);
                Usum += lane_f(ue_c, j);
                if (sem) {  // integrator.py:93-124, later entries overwrite earlier ones
                    const unsigned int id_j = (unsigned int)__builtin_amdgcn_readlane((int)id_c, j);
                    const float s_j = lane_f(s_c, j);
                    sc_new = f2h(s_j > s_old ? s_j : s_old);
                    if (id_old != id_j) id_new = (s_j > s_old) ? id_j : id_old;
                }
                ++n_run;
            }
        };
        add(we, ue, id_e, s_e, h, end);
        size_t next = base + 64;
        while (end == 64 && next < M) {  // the run goes on behind the chunk: pull the next 64 slots (rare: a run is ~22 entries)
            unsigned int key2, id2;
            float we2, ue2, s2;
            load(next + lane, key2, we2, ue2, id2, s2);
            const int len = __builtin_popcountll(__ballot(key2 == key_h));  // (a prefix of the chunk: sorted keys)
            add(we2, ue2, id2, s2, 0, len);
            end = len == 64 ? 64 : 0;
            next += 64;
        }
        if (lane == 0) {
            const float w_old = h2f(a.wgt[lin]), v_old = h2f(a.tsdf[lin]);
            const float w_new = w_old + Wsum;
            const float num = w_old * v_old + Usum;
            a.wgt[lin] = f2h(w_new);
            a.tsdf[lin] = f2h(num / w_new);
            if (sem) {
                a.score_vol[lin] = sc_new;
                a.id_vol[lin] = (uint8_t)id_new;
            }
        }
        ++runs;
        entries += n_run;
    }
    // test / profiling only (ojf_integrate's stats_dev): two same-address atomics per wave - 134 k per frame - were 1.3 ms of this kernel's 1.5
    if (a.stats && lane == 0 && runs) {
        atomicAdd(&a.counters[0], runs);
        atomicAdd(&a.counters[1], entries);
    }
}

__global__ void parity_stats_kernel(IntegrateArgs a)
{
    a.stats[0] = a.counters[0];
    a.stats[1] = a.counters[1];
    a.stats[2] = 0;
    a.stats[3] = 0;
}

int integrate_parity(const IntegrateArgs &a, const Camera &cam, void *ws, size_t ws_bytes, hipStream_t stream)
{
    ParityLayout L;
    int rc = parity_layout(a.X, a.Y, a.Z, a.h, a.w, a.n_tail, L, stream);
    if (rc) return rc;
    if (ws_bytes < L.total) return fail("ojf_integrate(parity): workspace too small");
    char *base = static_cast<char *>(ws);
    unsigned int *keys_in = reinterpret_cast<unsigned int *>(base + L.keys_in);
    unsigned int *keys_out = reinterpret_cast<unsigned int *>(base + L.keys_out);
    EntryVal *vals_in = reinterpret_cast<EntryVal *>(base + L.vals_in);
    EntryVal *vals_out = reinterpret_cast<EntryVal *>(base + L.vals_out);
    const size_t nvox = (size_t)a.X * a.Y * a.Z;
    const size_t M = (size_t)a.h * a.w * a.n_tail * 8;
    const unsigned int sentinel = (unsigned int)nvox;
    const int items = a.h * a.w * a.n_tail;
    hipLaunchKernelGGL(parity_emit_kernel, dim3((items + 255) / 256), dim3(256), 0, stream, a, cam, keys_in, vals_in,
                       sentinel);
    OJF_HIP(hipGetLastError());
    size_t temp_bytes = L.temp_bytes;
    OJF_HIP(rocprim::radix_sort_pairs(base + L.temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, M, 0u,
                                      key_bits(nvox), stream));
    hipLaunchKernelGGL(parity_walk_kernel, dim3((unsigned int)((M + 255) / 256)), dim3(256), 0, stream, a, keys_out,
                       vals_out, M, sentinel);
    OJF_HIP(hipGetLastError());
    if (a.stats) {
        hipLaunchKernelGGL(parity_stats_kernel, dim3(1), dim3(1), 0, stream, a);
        OJF_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace ojf
