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

__global__ __launch_bounds__(256) void parity_walk_kernel(IntegrateArgs a, const unsigned int *keys, const EntryVal *vals,
                                                           size_t M, unsigned int sentinel)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.guard && a.counters[kGuardLatch]) {  // (uniform) the net's range guard had fired when this call began: no volume is touched
        if (i == 0) atomicAdd(a.guard + 1, 1);
        return;
    }
    if (i >= M) return;
    const unsigned int key = keys[i];
    if (key == sentinel) return;
    if (i > 0 && keys[i - 1] == key) return;  // not the head of this voxel's run
    const size_t lin = key;
    const bool sem = a.id_vol != nullptr;
    const unsigned int per_pixel = (unsigned int)a.n_tail * 8u;
    float Wsum = 0.0f, Usum = 0.0f;  // integrator.py:59-67, sequential fp32 in entry order
    uint8_t id_old = 0, id_new = 0;
    float s_old = 0.0f;
    uint16_t sc_new = 0;
    if (sem) {
        id_old = a.id_vol[lin];
        id_new = id_old;
        sc_new = a.score_vol[lin];
        s_old = h2f(sc_new);
    }
    unsigned int n_run = 0;
    for (size_t j = i; j < M && keys[j] == key; ++j) {
        const EntryVal ev = vals[j];
        const unsigned int n = ev.e / per_pixel;
        const unsigned int k = (ev.e / 8u) % (unsigned int)a.n_tail;
        float v = a.est[(size_t)n * a.est_stride + k];  // pipeline.py:153-156
        v = v < -a.trunc ? -a.trunc : (v > a.trunc ? a.trunc : v);
        const float ue = ev.we * v;  // integrator.py:55
        Wsum += ev.we;
        Usum += ue;
        if (sem) {  // integrator.py:93-124, later entries overwrite earlier ones
            const uint8_t id_e = a.sem_ids[n];
            const float s_e = a.sem_scores[n];
            sc_new = f2h(s_e > s_old ? s_e : s_old);
            if (id_old != id_e) id_new = (s_e > s_old) ? id_e : id_old;
        }
        ++n_run;
    }
    const float w_old = h2f(a.wgt[lin]), v_old = h2f(a.tsdf[lin]);
    const float w_new = w_old + Wsum;
    const float num = w_old * v_old + Usum;
    a.wgt[lin] = f2h(w_new);
    a.tsdf[lin] = f2h(num / w_new);
    if (sem) {
        a.score_vol[lin] = sc_new;
        a.id_vol[lin] = id_new;
    }
    atomicAdd(&a.counters[0], 1u);
    atomicAdd(&a.counters[1], n_run);
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
