// Types shared by the two integrate translation units (fast: ojf_integrate.hip, parity:
// ojf_integrate_parity.hip).
#pragma once
#include "ojf_common.h"

namespace ojf {

struct alignas(8) VoxelAcc {
    unsigned long long w;  // sum of corner weights, 2^-36 fixed point (two's complement)
    unsigned long long u;  // sum of weight * clamped update
    unsigned int e_last;   // 1 + highest entry id that hit this voxel (0 = untouched)
    unsigned int e_diff;   // 1 + highest entry id whose class differs from the voxel's old class
};
static_assert(sizeof(VoxelAcc) == 24, "VoxelAcc layout");

constexpr double kFixScale = 68719476736.0;          // 2^36
constexpr double kFixInv = 1.0 / 68719476736.0;
constexpr size_t kHeaderBytes = 256;                  // counters: [0] touched voxels, [1] entries

struct IntegrateArgs {
    const float *depth;  // filtered frame
    const float *est;
    uint16_t *tsdf;
    uint16_t *wgt;
    const uint8_t *sem_ids;
    const float *sem_scores;
    uint8_t *id_vol;
    uint16_t *score_vol;
    unsigned int *counters;
    VoxelAcc *acc;
    unsigned int *touched;
    uint32_t *stats;
    int X, Y, Z, h, w, n_points, n_tail, est_stride;
    float trunc;
};

size_t fast_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail);
size_t parity_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail);
int integrate_parity(const IntegrateArgs &a, const Camera &cam, void *ws, size_t ws_bytes, hipStream_t stream);

}  // namespace ojf
