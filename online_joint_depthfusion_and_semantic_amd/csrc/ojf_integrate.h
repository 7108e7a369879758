// Types shared by the two integrate translation units (fast: ojf_integrate.hip, parity:
// ojf_integrate_parity.hip).
#pragma once
#include "ojf_common.h"

namespace ojf {

// One partial sum of one voxel, produced by one pixel tile (or one entry on the slow paths).  Records of
// a voxel form a singly linked list hanging off the dense `head` table (index + 1, 0 = end).
struct alignas(16) VoxelRec {
    unsigned int lin;       // linear voxel index
    unsigned int next;      // next record of the same voxel (index + 1), 0 = end of list
    unsigned long long w;   // sum of corner weights, 2^-44 fixed point (two's complement)
    unsigned long long u;   // sum of weight * clamped update
    unsigned int e_last;    // 1 + highest entry id in this partial
    unsigned int e_diff;    // 1 + highest entry id whose class differs from the voxel's old class (0 = none)
};
static_assert(sizeof(VoxelRec) == 32, "VoxelRec layout");

// 2^-44: a voxel whose weight barely survives fp16 (W ~ 2^-25, a few dozen entries) still gets sum(w v) / sum(w) to
// half an fp16 ulp (2^-36 left up to 6 ulp there: tests/test_extract_integrate_gpu.py::test_full_size_properties_config_C);
// 19 integer bits remain: per-frame weight sums below 5e5 per voxel.
constexpr double kFixScale = 17592186044416.0;        // 2^44
constexpr double kFixInv = 1.0 / 17592186044416.0;
constexpr size_t kHeaderBytes = 512;                  // two sets of 32 counters ([0] counter-allocated touched voxels, [2] counter-allocated records), then the two phase words (kPhaseAcc, kPhaseFin)
constexpr int kPhaseAcc = 64, kPhaseFin = 65;         // uint index in the header: counter set the next accumulate / the coming finalize uses
// The range-guard decision of ONE integrate call (round 6): the flag has concurrent producers (the look-ahead 2-D pass on its side
// stream, the nets of fuse_many's other slots), so blocks - even threads - that each read it could disagree and a frame would be
// integrated in part.  The kernels that only fill the workspace (accumulate, the entry-list kernel, PARITY emit) therefore never
// skip; ONE thread of them copies the flag into this header word, and the kernel that writes the volumes (finalize, PARITY walk),
// a later launch on the same stream, reads nothing else: all of a call's voxels are updated or none is.
constexpr int kGuardLatch = 66;

struct IntegrateArgs {
    const float *depth;  // filtered frame, or the raw frame when `mask` is set
    const unsigned char *mask;  // optional validity mask (pipeline.py:196): depth counts as 0 where it is 0
    const float *est;
    uint16_t *tsdf;
    uint16_t *wgt;
    const uint8_t *sem_ids;
    const float *sem_scores;
    uint8_t *id_vol;
    uint16_t *score_vol;
    unsigned int *counters;
    unsigned int *counters_next;  // unused (kept for layout)
    int phased;                   // FAST frame path: `counters` is the header; the two counter sets alternate by the phase words IN the header
    unsigned int *head;   // dense [X*Y*Z]: first record of the voxel (index + 1), 0 = untouched; left zeroed
    VoxelRec *recs;
    unsigned int *touched;
    unsigned int *tile_new;   // tiled FAST path: voxels first touched by each tile (its slice of `touched`)
    unsigned int list_base;   // first element of the counter-allocated parts of `touched` / `recs`
    int n_tiles;              // 0 on the entry-list path
    uint32_t *stats;
    // range guard of the split-fp16 net (ojf_common.h guard_raise): device block {flag, skipped calls, ...} or NULL.  While
    // the flag is set no integrate call touches a volume - the est rows of a tripped frame are invalid, and so is every
    // frame until the host has reported the event (ojf_net_check) - and [1] counts the calls skipped.
    int *guard;
    int X, Y, Z, h, w, n_points, n_tail, est_stride;
    float trunc;
};

__device__ __forceinline__ float frame_depth(const IntegrateArgs &a, int n)
{
    const float z = a.depth[n];
    return (a.mask && !a.mask[n]) ? 0.0f : z;  // torch.where(mask == 0, 0, frame)
}

const int *range_guard_if_any();  // ojf_net.hip
__device__ __forceinline__ bool guard_set(const IntegrateArgs &a) { return a.guard && *static_cast<volatile const int *>(a.guard) != 0; }

size_t fast_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail);
size_t parity_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail);
int integrate_parity(const IntegrateArgs &a, const Camera &cam, void *ws, size_t ws_bytes, hipStream_t stream);

}  // namespace ojf
