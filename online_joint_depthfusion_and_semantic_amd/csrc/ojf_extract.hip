// EXTRACT: ray-cast gather of fp16 TSDF / weight voxels along unprojected depth rays with the
// reference's 8-corner interpolation (modules/extractor.py:24-79, :640-681).
//
// Mapping: a block is 64 consecutive pixels x n_points samples, wave k = ray offset k, so the 64 lanes of a wave
// are 64 consecutive pixels of one image row at the same ray offset: their stencils walk neighbouring voxels (the
// z axis is contiguous in HBM) and the 16 fp16 loads per lane are all independent and in flight together.
// Wave 0 computes the 64 ray frames (fp32 unprojection, fp64 normalisation: three fp64 divisions and a square
// root - a third of the per-item instruction count) ONCE and hands them to the other waves through 3 KB of LDS; the
// kernel is bound by instruction issue of the fp64 index math and the gather latency, not by bytes: the per-frame
// gather footprint (<= a few MB of 128-B lines) lives in L2/MALL.
#include <cmath>
#include <cstdlib>

#include "ojf_common.h"

namespace ojf {

struct ExtractArgs {
    const float *depth;
    const uint16_t *tsdf;
    const uint16_t *wgt;
    float *out_values;
    float *out_weights;
    int64_t *dbg_idx;
    double *dbg_w;
    double *dbg_pts;
    float *dbg_pcl;
    int X, Y, Z, h, w, n_points, out_stride, out_layout;
    float pad_value;
    // out_layout == 2 (ojf_extract_to_net): the results go straight into the fusion net's input planes - channels
    // [0, P) values, [P, 2P) weights, 2P the raw depth, zeros up to 4 * net_cs4 (modules/pipeline.py:74-102)
    float4 *net_x0;
    int net_cs4;
    int *ovf;  // split-fp16 range flag of the net (or NULL)
    int net_split;  // the net keeps this slot as split planes (split_planes4)
    int n_tiles = 0;  // extract_tile_kernel: pixel tiles of the frame (the grid is that rounded up to a multiple of 8: XCD bands)
};

// Pixel tile of a block of extract_tile_kernel: kTileW columns x kTileH rows, the 64 lanes of a wave walk DOWN the columns (lane = column *
// kTileH + row).  Round 6: with 64 consecutive pixels of an image ROW per wave (rounds 1-5) the lanes of a gather hit ~40-64 different 128-byte
// lines - an image row runs along the camera's right vector, which for an upright camera lies in the volume's x-y plane, where neighbouring
// voxels are 512 bytes (one z row) or more apart - and the texture-address path takes one cycle per line.  Image COLUMNS run along the
// camera's down vector, for an upright camera the volume's contiguous z axis: 16 rows of a column are ~10 voxels of ONE z row = one line.
// Both orientations are built (4 x 16: lanes walk down columns; 16 x 4: along rows) and the host picks per frame the image axis whose
// world direction has the larger component along the volume's z axis (extract_tile_shape); the same bits either way.

constexpr int kNetPitch = 36;  // floats per pixel of the LDS transpose tile (<= 8 channel groups + padding against bank conflicts)

constexpr int kMaxTilePoints = 16;  // at most 64 * n_points threads per block
// Waves of an extract_tile_kernel block: every wave takes the samples k = wave, wave + waves, ... of the block's 64 pixels.  Rounds 1-6
// ran one wave per sample: nine-wave blocks, of which a CU held TWO (nine waves do not spread 7 / 7 / 7 / 6 over the SIMDs that 72 VGPRs
// allow), so the 1200 blocks of a 320x240 frame ran in three waves of 4.5 us each (profiles/r06_accumulate_stamps.txt).  Three samples per
// wave: three-wave blocks, eight or nine per CU, the whole frame resident at once.  OJF_EXTRACT_WAVES=n forces n (A/B).
static int extract_block_waves(int n_points)
{
    static const int forced = getenv("OJF_EXTRACT_WAVES") ? atoi(getenv("OJF_EXTRACT_WAVES")) : 0;
    int waves = forced > 0 ? forced : (n_points + 2) / 3;
    if (waves > n_points) waves = n_points;
    return waves < 1 ? 1 : waves;
}

// body shared by the two mappings: sample k of pixel n on the ray frame (cv, dir)
__device__ __forceinline__ void extract_item(const ExtractArgs &a, int n, int k, const double cv[3], const double dir[3],
                                             const float pw[3], float *net_tile = nullptr)
{
    const int half = (a.n_points - 1) / 2;
    RaySample s;
    ray_sample(cv, dir, k, half, s);

    // Gathers.  Corners 2m and 2m+1 share (x, y) and differ by nb.z in {-1, 0, +1} along the contiguous z axis: when
    // both lie inside the volume ONE (2-byte aligned) 4-byte load per volume fetches the pair, which halves the
    // number of scattered load instructions (the gathers are 60 % of this kernel).  All loads are issued before
    // the first use.
    float val[8], wt[8];
    double wq[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int64_t i0[3], i1[3];
        corner(s, 2 * m, i0, wq[2 * m]);
        corner(s, 2 * m + 1, i1, wq[2 * m + 1]);
        const bool in0 = in_volume(i0, a.X, a.Y, a.Z), in1 = in_volume(i1, a.X, a.Y, a.Z);
        val[2 * m] = val[2 * m + 1] = a.pad_value;  // modules/extractor.py:663-664
        wt[2 * m] = wt[2 * m + 1] = 0.0f;
        const size_t lin0 = ((size_t)i0[0] * a.Y + (size_t)i0[1]) * a.Z + (size_t)i0[2];
        if (in0 && in1 && i1[2] != i0[2]) {
            const bool up = i1[2] > i0[2];                   // corner 2m+1 is the upper z neighbour
            const size_t lo = up ? lin0 : lin0 - 1;
            unsigned int tv, wv;                              // two fp16 each: [z_lo | z_lo + 1]
            __builtin_memcpy(&tv, a.tsdf + lo, 4);
            __builtin_memcpy(&wv, a.wgt + lo, 4);
            const uint16_t t_lo = (uint16_t)(tv & 0xffffu), t_hi = (uint16_t)(tv >> 16);
            const uint16_t w_lo = (uint16_t)(wv & 0xffffu), w_hi = (uint16_t)(wv >> 16);
            val[2 * m] = h2f(up ? t_lo : t_hi);
            val[2 * m + 1] = h2f(up ? t_hi : t_lo);
            wt[2 * m] = h2f(up ? w_lo : w_hi);
            wt[2 * m + 1] = h2f(up ? w_hi : w_lo);
        } else {
            if (in0) {
                val[2 * m] = h2f(a.tsdf[lin0]);
                wt[2 * m] = h2f(a.wgt[lin0]);
            }
            if (in1) {
                const size_t lin1 = ((size_t)i1[0] * a.Y + (size_t)i1[1]) * a.Z + (size_t)i1[2];
                val[2 * m + 1] = h2f(a.tsdf[lin1]);
                wt[2 * m + 1] = h2f(a.wgt[lin1]);
            }
        }
        if (a.dbg_idx) {
            int64_t *o = a.dbg_idx + ((size_t)n * a.n_points + k) * 24 + 6 * m;
            o[0] = i0[0]; o[1] = i0[1]; o[2] = i0[2];
            o[3] = i1[0]; o[4] = i1[1]; o[5] = i1[2];
        }
    }
    // fp64 products; the 8-term sums of torch.sum(.., dim=1) (extractor.py:673-674) in ATen's row_sum order - four
    // interleaved partial sums s_k = p_k + p_{k+4}, then ((s_0 + s_1) + s_2) + s_3 - rounded once to fp32
    double pv[8], pt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        pv[q] = (double)val[q] * wq[q];
        pt[q] = (double)wt[q] * wq[q];
    }
    const double sv = (((pv[0] + pv[4]) + (pv[1] + pv[5])) + (pv[2] + pv[6])) + (pv[3] + pv[7]);
    const double sw = (((pt[0] + pt[4]) + (pt[1] + pt[5])) + (pt[2] + pt[6])) + (pt[3] + pt[7]);
    if (a.out_layout == 2) {
        net_tile[0] = (float)sv;            // caller-provided LDS slots of (pixel, k) and (pixel, P + k)
        net_tile[a.n_points] = (float)sw;
    } else {
        const size_t o = a.out_layout ? (size_t)k * a.out_stride + n : (size_t)n * a.out_stride + k;
        a.out_values[o] = (float)sv;
        a.out_weights[o] = (float)sw;
    }

    if (a.dbg_w) {
        double *o = a.dbg_w + ((size_t)n * a.n_points + k) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = wq[q];
    }
    if (a.dbg_pts) {
        double *o = a.dbg_pts + ((size_t)n * a.n_points + k) * 3;
        o[0] = s.p[0]; o[1] = s.p[1]; o[2] = s.p[2];
    }
    if (a.dbg_pcl && k == 0) {
        float *o = a.dbg_pcl + (size_t)n * 3;
        o[0] = pw[0]; o[1] = pw[1]; o[2] = pw[2];
    }
}

#ifdef OJF_EXT_STAMPS  // profiling build only (tools/acc_stamps.py): phase stamps of every extract block, thread 0 and the last wave's lane 0
__device__ unsigned long long g_ext_stamps[4096][8];
#define EXT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ext_stamps[blockIdx.x][i] = (i) >= 6 ? wall_clock64() : clock64(); } while (0)
#else
#define EXT_STAMP(i) do { } while (0)
#endif
// n_points <= kMaxTilePoints: blockDim.x = 64 * waves (extract_block_waves), wave v = samples v, v + waves, ... of the block's 64 pixels
template <int kTileW, int kTileH>
__device__ __forceinline__ void extract_tile_body(const ExtractArgs &a, const Camera &cam)
{
    static_assert(kTileW * kTileH == 64, "one wave = one pixel tile");
    __shared__ double frame[6][64];
    __shared__ float pcl[3][64];
    const int N = a.h * a.w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int ptile = banded_block_x();  // one band of tile rows per XCD: neighbouring rays gather the same lines
    if (ptile >= a.n_tiles) return;      // (block-uniform: a padding block of the banded grid)
    EXT_STAMP(6);
    EXT_STAMP(0);
    const int tiles_x = (a.w + kTileW - 1) / kTileW;
    const int ty = ptile / tiles_x, tx = ptile - ty * tiles_x;
    const int r = ty * kTileH + lane % kTileH, c = tx * kTileW + lane / kTileH;
    const bool valid = r < a.h && c < a.w;
    const int n = valid ? r * a.w + c : N;  // (n >= N: no pixel)
    if (wave == 0 && n < N) {
        float pw[3];
        double cv[3], dir[3];
        unproject(r, c, a.depth[n], cam, pw);
        ray_frame(pw, cam, cv, dir);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            frame[i][lane] = cv[i];
            frame[3 + i][lane] = dir[i];
            pcl[i][lane] = pw[i];
        }
    }
    __syncthreads();
    EXT_STAMP(1);
    if (a.out_layout != 2) {
        if (n >= N) return;
        const double cv[3] = {frame[0][lane], frame[1][lane], frame[2][lane]};
        const double dir[3] = {frame[3][lane], frame[4][lane], frame[5][lane]};
        const float pw[3] = {pcl[0][lane], pcl[1][lane], pcl[2][lane]};
        for (int k = wave; k < a.n_points; k += waves) extract_item(a, n, k, cv, dir, pw);
        return;
    }
    // net-input form: the block's 64 x (2P + 1) results are transposed through LDS into the net's float4 channel planes
    // (64 consecutive pixels of a plane = 1 KB per wave store) - no sample planes, no prepare_input launch
    __shared__ __attribute__((aligned(16))) float tile[64 * kNetPitch];
    if (n < N) {
        const double cv[3] = {frame[0][lane], frame[1][lane], frame[2][lane]};
        const double dir[3] = {frame[3][lane], frame[4][lane], frame[5][lane]};
        const float pw[3] = {pcl[0][lane], pcl[1][lane], pcl[2][lane]};
        for (int k = wave; k < a.n_points; k += waves) extract_item(a, n, k, cv, dir, pw, tile + lane * kNetPitch + k);
        if (wave == 0) {
            tile[lane * kNetPitch + 2 * a.n_points] = a.depth[n];
            for (int c = 2 * a.n_points + 1; c < 4 * a.net_cs4; ++c) tile[lane * kNetPitch + c] = 0.0f;
        }
    }
    EXT_STAMP(2);
    __syncthreads();
    EXT_STAMP(3);
    bool bad = false;
    for (int t = threadIdx.x; t < 64 * a.net_cs4; t += blockDim.x) {
        // stores walk ALONG the rows (the planes are row-major): px -> (row px / kTileW, column px % kTileW) = 64 contiguous bytes per row
        const int cg = t >> 6, px = t & 63;
        const int rl = px / kTileW, cl = px - rl * kTileW;
        const int rr = ty * kTileH + rl, cc = tx * kTileW + cl;
        if (rr >= a.h || cc >= a.w) continue;
        const float4 v = *reinterpret_cast<const float4 *>(tile + (cl * kTileH + rl) * kNetPitch + 4 * cg);
        a.net_x0[(size_t)cg * N + rr * a.w + cc] = a.net_split ? split_planes4(v) : v;
        bad = bad || fabsf(v.x) > 65504.0f || fabsf(v.y) > 65504.0f || fabsf(v.z) > 65504.0f || fabsf(v.w) > 65504.0f;
    }
    if (bad && a.ovf) guard_raise(a.ovf, 1);  // split-fp16 range guard of the net input (NaN passes, like everywhere else)
    EXT_STAMP(4);
    EXT_STAMP(7);
}

template <int TW, int TH>
__global__ __launch_bounds__(64 * kMaxTilePoints) void extract_tile_kernel(ExtractArgs a, Camera cam) { extract_tile_body<TW, TH>(a, cam); }

// The frames of up to kMaxScenes SCENES in one launch (ojf_extract_many, round 6): blockIdx.y = scene, every scene with its
// own volumes, camera and outputs.  One frame's launch is one round of 1200 blocks that each walk a chain of dependent steps
// (ray frame -> gathers -> sums -> transposed store): S frames side by side in ONE launch share the ramp and
// fill the CUs while other blocks wait - which S launches on S streams do not (they take turns on the queue).  The same
// blocks, the same code per block: the same bits as S separate calls.
struct ExtractMany { ExtractArgs a[OJF_MAX_SCENES]; Camera cam[OJF_MAX_SCENES]; };
template <int TW, int TH>
__global__ __launch_bounds__(64 * kMaxTilePoints) void extract_tile_many_kernel(ExtractMany m)
{
    extract_tile_body<TW, TH>(m.a[blockIdx.y], m.cam[blockIdx.y]);
}

// true: 4 x 16 tiles (lanes walk down image columns), false: 16 x 4 (along rows).  E = rows of the 3 x 4 camera-to-world matrix: E[8] / E[9] are the
// volume-z components of the camera's x (image row direction) and y (image column direction) axes.
static inline bool extract_columns(const float *E)
{
    static const int force = getenv("OJF_EXTRACT_TILE") ? atoi(getenv("OJF_EXTRACT_TILE")) : 0;  // A/B: 1 columns, 2 rows
    if (force) return force == 1;
    return std::fabs(E[9]) >= std::fabs(E[8]);
}
static inline int extract_tiles(int h, int w, bool columns)
{
    const int tw = columns ? 4 : 16, th = columns ? 16 : 4;
    return ((w + tw - 1) / tw) * ((h + th - 1) / th);
}

// any n_points: one lane per (sample k, pixel n), k-major; every item computes its own ray frame
__global__ __launch_bounds__(256) void extract_kernel(ExtractArgs a, Camera cam)
{
    const int N = a.h * a.w;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= N * a.n_points) return;
    const int k = item / N;
    const int n = item - k * N;
    const int r = n / a.w, c = n - r * a.w;
    float pw[3];
    double cv[3], dir[3];
    unproject(r, c, a.depth[n], cam, pw);
    ray_frame(pw, cam, cv, dir);
    extract_item(a, n, k, cv, dir, pw);
}

}  // namespace ojf

OJF_API int ojf_extract(const float *depth, const float *Ki, const float *E, const double *origin,
                        double res, const uint16_t *tsdf, const uint16_t *wgt, int X, int Y, int Z,
                        int h, int w, int n_points, float pad_value, float *out_values,
                        float *out_weights, int out_stride, int out_layout, int64_t *dbg_idx, double *dbg_w,
                        double *dbg_pts, float *dbg_pcl, ojf_stream_t stream)
{
    using namespace ojf;
    if (!depth || !Ki || !E || !origin || !tsdf || !wgt || !out_values || !out_weights)
        return fail("ojf_extract: null pointer argument");
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0)
        return fail("ojf_extract: non-positive volume or frame size");
    if (n_points < 1 || (n_points & 1) == 0) return fail("ojf_extract: n_points must be odd and >= 1");
    if (out_layout != 0 && out_layout != 1) return fail("ojf_extract: out_layout must be 0 (rows) or 1 (sample planes)");
    if (out_layout == 0 && out_stride < n_points) return fail("ojf_extract: out_stride < n_points");
    if (out_layout == 1 && out_stride < h * w) return fail("ojf_extract: plane stride < h*w");
    if ((int64_t)h * w * n_points > 0x7fffffffLL) return fail("ojf_extract: frame too large");
    if (!(res > 0.0)) return fail("ojf_extract: resolution must be > 0");
    ExtractArgs a{depth, tsdf, wgt, out_values, out_weights, dbg_idx, dbg_w, dbg_pts, dbg_pcl,
                  X, Y, Z, h, w, n_points, out_stride, out_layout, pad_value, nullptr, 0, nullptr, 0};
    const bool cols = extract_columns(E);
    a.n_tiles = extract_tiles(h, w, cols);
    const Camera cam = make_camera(Ki, E, origin, res);
    if (n_points <= kMaxTilePoints) {
        if (cols) hipLaunchKernelGGL((extract_tile_kernel<4, 16>), dim3((a.n_tiles + 7) / 8 * 8), dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), a, cam);
        else hipLaunchKernelGGL((extract_tile_kernel<16, 4>), dim3((a.n_tiles + 7) / 8 * 8), dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), a, cam);
    } else {
        const int items = h * w * n_points;
        hipLaunchKernelGGL(extract_kernel, dim3((items + 255) / 256), dim3(256), 0, as_stream(stream), a, cam);
    }
    return check_hip(hipGetLastError(), "ojf_extract launch");
}

OJF_API int ojf_extract_to_net(const float *depth, const float *Ki, const float *E, const double *origin, double res,
                               const uint16_t *tsdf, const uint16_t *wgt, int X, int Y, int Z, int h, int w, int n_points,
                               float pad_value, ojf_net *net, ojf_stream_t stream)
{
    using namespace ojf;
    if (!depth || !Ki || !E || !origin || !tsdf || !wgt || !net) return fail("ojf_extract_to_net: null pointer argument");
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0) return fail("ojf_extract_to_net: non-positive volume or frame size");
    if (!(res > 0.0)) return fail("ojf_extract_to_net: resolution must be > 0");
    NetInputSlot slot;
    if (net_input_slot(net, &slot))
        return fail("ojf_extract_to_net: this net takes a semantic channel or has two heads: use ojf_extract + ojf_net_prepare_input");
    if (slot.h != h || slot.w != w || slot.P != n_points) return fail("ojf_extract_to_net: frame size / n_points differ from the net's");
    if (n_points < 1 || (n_points & 1) == 0 || n_points > kMaxTilePoints || 4 * slot.cs4 > kNetPitch || 2 * n_points + 1 > 4 * slot.cs4)
        return fail("ojf_extract_to_net: unsupported n_points / slot width");
    ExtractArgs a{depth, tsdf, wgt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                  X, Y, Z, h, w, n_points, h * w, 2, pad_value, reinterpret_cast<float4 *>(slot.x0), slot.cs4, slot.ovf, slot.split};
    const bool cols = extract_columns(E);
    a.n_tiles = extract_tiles(h, w, cols);
    const Camera cam = make_camera(Ki, E, origin, res);
    if (cols) hipLaunchKernelGGL((extract_tile_kernel<4, 16>), dim3((a.n_tiles + 7) / 8 * 8), dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), a, cam);
    else hipLaunchKernelGGL((extract_tile_kernel<16, 4>), dim3((a.n_tiles + 7) / 8 * 8), dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), a, cam);
    return check_hip(hipGetLastError(), "ojf_extract_to_net launch");
}

// One frame of each of n scenes (1 <= n <= OJF_MAX_SCENES) of ONE frame size and sample count as a single launch.
OJF_API int ojf_extract_many(int n, const ojf_extract_job *jobs, int X, int Y, int Z, int h, int w, int n_points, float pad_value,
                             ojf_stream_t stream)
{
    using namespace ojf;
    if (n < 1 || n > OJF_MAX_SCENES || !jobs) return fail("ojf_extract_many: 1..OJF_MAX_SCENES jobs");
    if (X <= 0 || Y <= 0 || Z <= 0 || h <= 0 || w <= 0) return fail("ojf_extract_many: non-positive volume or frame size");
    if (n_points < 1 || (n_points & 1) == 0 || n_points > kMaxTilePoints) return fail("ojf_extract_many: n_points must be odd, 1..16");
    if ((int64_t)h * w * n_points > 0x7fffffffLL) return fail("ojf_extract_many: frame too large");
    ExtractMany m;
    for (int i = 0; i < n; ++i) {
        const ojf_extract_job &j = jobs[i];
        if (!j.depth_dev || !j.Kinv_host || !j.E_host || !j.origin_host || !j.tsdf_dev || !j.weights_dev) return fail("ojf_extract_many: null pointer in a job");
        if (!(j.resolution > 0.0)) return fail("ojf_extract_many: resolution must be > 0");
        for (int k = 0; k < i; ++k)
            if ((j.net && j.net == jobs[k].net) || (j.out_values_dev && j.out_values_dev == jobs[k].out_values_dev))
                return fail("ojf_extract_many: two jobs write the same output");
        if (j.net) {
            NetInputSlot slot;
            if (net_input_slot(j.net, &slot))
                return fail("ojf_extract_many: this net takes a semantic channel or has two heads: hand over sample planes instead");
            if (slot.h != h || slot.w != w || slot.P != n_points) return fail("ojf_extract_many: frame size / n_points differ from the net's");
            if (4 * slot.cs4 > kNetPitch || 2 * n_points + 1 > 4 * slot.cs4) return fail("ojf_extract_many: unsupported slot width");
            m.a[i] = ExtractArgs{j.depth_dev, j.tsdf_dev, j.weights_dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 X, Y, Z, h, w, n_points, h * w, 2, pad_value, reinterpret_cast<float4 *>(slot.x0), slot.cs4, slot.ovf, slot.split};
        } else {
            if (!j.out_values_dev || !j.out_weights_dev) return fail("ojf_extract_many: a job needs a net or output buffers");
            if (j.out_layout != 0 && j.out_layout != 1) return fail("ojf_extract_many: out_layout must be 0 (rows) or 1 (sample planes)");
            if (j.out_layout == 0 ? j.out_stride < n_points : j.out_stride < h * w) return fail("ojf_extract_many: output stride too small");
            m.a[i] = ExtractArgs{j.depth_dev, j.tsdf_dev, j.weights_dev, j.out_values_dev, j.out_weights_dev, nullptr, nullptr, nullptr, nullptr,
                                 X, Y, Z, h, w, n_points, j.out_stride, j.out_layout, pad_value, nullptr, 0, nullptr, 0};
        }
        m.cam[i] = make_camera(j.Kinv_host, j.E_host, j.origin_host, j.resolution);
    }
    // one tile shape per launch: the first scene's camera decides (any shape computes the same; the scenes of a call are usually filmed alike)
    const bool cols = extract_columns(jobs[0].E_host);
    for (int i = 0; i < n; ++i) m.a[i].n_tiles = extract_tiles(h, w, cols);
    const dim3 grid((extract_tiles(h, w, cols) + 7) / 8 * 8, n);
    if (cols) hipLaunchKernelGGL((extract_tile_many_kernel<4, 16>), grid, dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), m);
    else hipLaunchKernelGGL((extract_tile_many_kernel<16, 4>), grid, dim3(64 * extract_block_waves(n_points)), 0, as_stream(stream), m);
    return check_hip(hipGetLastError(), "ojf_extract_many launch");
}

#ifdef OJF_EXT_STAMPS
extern "C" __attribute__((visibility("default"))) int ojf_debug_ext_stamps(void *host_dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ojf::g_ext_stamps), bytes < sizeof(ojf::g_ext_stamps) ? bytes : sizeof(ojf::g_ext_stamps));
}
#endif
