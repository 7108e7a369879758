// Shared host/device helpers of libojf (gfx950 only; no CUDA/portability layer by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/ojf.h"

#define OJF_API extern "C" __attribute__((visibility("default")))

namespace ojf {

void set_error(const std::string &msg);
int fail(const std::string &msg);
int check_hip(hipError_t e, const char *what);

#define OJF_HIP(call)                                  \
    do {                                               \
        int rc__ = ::ojf::check_hip((call), #call);    \
        if (rc__) return rc__;                         \
    } while (0)

static inline hipStream_t as_stream(ojf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- camera + ray geometry ------------------------------------------------------------------
// Passed by value as a kernel argument (lives in SGPRs / the kernarg segment).
struct Camera {
    float Ki[9];    // inverse intrinsics, row-major
    float E[12];    // rows of the 3x4 camera-to-world matrix
    double origin[3];
    double res;
    double eye_v[3];  // (E[:,3] - origin) / res, hoisted to the host: same IEEE f64 ops
};

Camera make_camera(const float *Ki, const float *E, const double *origin, double res);

// fp16 bit pattern <-> fp32 (round-to-nearest-even on the way down, like torch .half())
// Block b of a launch is observed to run on XCD b % 8 (8 XCDs with a private 4 MB L2 each).  For a 1-D grid that is a
// multiple of 8 this renumbers the blocks so that every XCD works on ONE contiguous eighth of the range - for pixel
// tiles: one band of the image, the same band in every kernel of the frame, so that what neighbouring tiles share and
// what the previous kernel wrote is found in the XCD's own L2.  A pure speed choice: any placement computes the same.
__device__ __forceinline__ int xcd_band_block(int b, int n8) { return (b & 7) * (n8 >> 3) + (b >> 3); }
__device__ __forceinline__ int banded_block_x() { return (gridDim.x & 7) == 0 ? xcd_band_block(blockIdx.x, gridDim.x) : (int)blockIdx.x; }

// Range guard of the split-fp16 arithmetic (ojf_net.hip: overflow_flag): a DEVICE-resident block {flag, integrate calls
// skipped, pointer to the host-mapped mirror of the flag}.  A kernel that meets a value a later split could not
// represent raises the flag here - the device word is what the integrate kernels test before they touch a volume (a
// frame whose net tripped the guard must not be fused), the host-mapped mirror is what the host polls without
// synchronising.  Runs only when the guard fires: no traffic otherwise.
__device__ __forceinline__ void guard_raise(int *g, int v)
{
    g[0] = v;
    int *host = *reinterpret_cast<int *const *>(g + 2);
    *host = v;
}

// Low halves of the split-fp16 operands: fp16(x0 - hi.lo), fp16(x1 - hi.hi) packed like `hpair` (the two fp16 high
// halves of x0, x1).  v_fma_mix{lo,hi}_f16 forms x - hi exactly in fp32 and rounds once to fp16: the same bits as
// converting, subtracting and converting again, in 2 instructions instead of 5 (clang folds the source-level
// fma(hi, -1, x) into a subtraction before it can select the mix form, hence the assembly).
__device__ __forceinline__ unsigned split_lo_pair(unsigned hpair, float x0, float x1)
{
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hpair), "v"(x1));
    return r;
}

// One float4 -> its split-fp16 form in the same 16 bytes: halfs {h0 h1 h2 h3 | l0 l1 l2 l3} ("split planes", the form the
// fusion net's dense-growth buffer holds for dense_chain_kernel; bit for bit ojf_net.hip's split_pack4)
__device__ __forceinline__ float4 split_planes4(const float4 &v)
{
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    const f4_t x{v.x, v.y, v.z, v.w};
    const h4_t hi = __builtin_convertvector(x, h4_t);  // 2 x v_cvt_pk_f16_f32
    const u2_t h = __builtin_bit_cast(u2_t, hi);
    return float4{__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(split_lo_pair(h[0], v.x, v.y)),
                  __uint_as_float(split_lo_pair(h[1], v.z, v.w))};
}

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// One sample of one ray: everything the gather and the scatter need, bit-identical in both.
// The rounding points are normative (oracle/ojf_oracle.c): compiled with
// -ffp-contract=off, the only fused operations are the explicit fma calls below.
struct RaySample {
    int64_t fl[3];   // floor(p)
    int nb[3];       // sign(centre - p) in {-1,0,1}
    double a[3];     // |p - centre|
    double p[3];
};

__device__ __forceinline__ void unproject(int r, int c, float z, const Camera &cam, float pw[3])
{
    // modules/extractor.py:112-117.  K^-1 @ p: every product and sum rounded separately;
    // E @ [pc;1]: rounded first product, then a single-rounding fma chain (see the oracle).
    const float u = (float)c * z;
    const float v = (float)r * z;
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t0 = cam.Ki[3 * i + 0] * u, t1 = cam.Ki[3 * i + 1] * v, t2 = cam.Ki[3 * i + 2] * z;
        pc[i] = (t0 + t1) + t2;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pw[i] = __builtin_fmaf(cam.E[4 * i + 3], 1.0f,
                               __builtin_fmaf(cam.E[4 * i + 2], pc[2],
                                              __builtin_fmaf(cam.E[4 * i + 1], pc[1], cam.E[4 * i + 0] * pc[0])));
}

__device__ __forceinline__ void ray_frame(const float pw[3], const Camera &cam, double cv[3], double dir[3])
{
    // modules/extractor.py:314-318 (fp64 because origin is an fp64 tensor)
    double d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        cv[i] = ((double)pw[i] - cam.origin[i]) / cam.res;
        d[i] = cv[i] - cam.eye_v[i];
    }
    const double ss = __builtin_fma(d[2], d[2], __builtin_fma(d[1], d[1], d[0] * d[0]));
    double nrm = __builtin_sqrt(ss);
    nrm = nrm < 1e-12 ? 1e-12 : nrm;
#pragma unroll
    for (int i = 0; i < 3; ++i) dir[i] = d[i] / nrm;
}

__device__ __forceinline__ void ray_sample(const double cv[3], const double dir[3], int k, int half, RaySample &s)
{
    // modules/extractor.py:327-331 (sample position), :535-555 (interpolation frame)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double p;
        if (k == half)
            p = cv[i];
        else if (k > half)
            p = cv[i] + (double)(k - half) * dir[i];
        else
            p = cv[i] - (double)(half - k) * dir[i];
        const double fl = __builtin_floor(p);
        const double ctr = fl + 0.5;
        const double sg = ctr - p;
        s.p[i] = p;
        // keeps the int conversion defined for absurd or NaN depths; such samples are outside
        // every volume in the reference as well (|idx| >> grid size, NaN -> INT64_MIN)
        const double flc = (fl >= -4.0e15 && fl <= 4.0e15) ? fl : -4.0e15;
        s.fl[i] = (int64_t)flc;
        s.nb[i] = sg > 0.0 ? 1 : (sg < 0.0 ? -1 : 0);
        s.a[i] = __builtin_fabs(p - ctr);
    }
}

// corner q of the 2x2x2 stencil in the reference's order (i,j,k) = 000,001,...,111
// (modules/extractor.py:560-586): index and fp64 weight, product evaluated left to right.
__device__ __forceinline__ void corner(const RaySample &s, int q, int64_t idx[3], double &wq)
{
    const int bi = (q >> 2) & 1, bj = (q >> 1) & 1, bk = q & 1;
    const double w1 = bi ? s.a[0] : 1.0 - s.a[0];
    const double w2 = bj ? s.a[1] : 1.0 - s.a[1];
    const double w3 = bk ? s.a[2] : 1.0 - s.a[2];
    wq = w1 * w2 * w3;
    idx[0] = s.fl[0] + (bi ? s.nb[0] : 0);
    idx[1] = s.fl[1] + (bj ? s.nb[1] : 0);
    idx[2] = s.fl[2] + (bk ? s.nb[2] : 0);
}

__device__ __forceinline__ bool in_volume(const int64_t idx[3], int X, int Y, int Z)
{
    return idx[0] >= 0 && idx[0] < X && idx[1] >= 0 && idx[1] < Y && idx[2] >= 0 && idx[2] < Z;
}

// (ojf_net.hip) where a single-head geometry net keeps the first slot of its dense block, for ojf_extract_to_net:
// plane buffer ([cs4][h*w] float4), its group count, n_points, frame size and the split-fp16 range flag (or NULL).
// split: the buffer holds split planes (split_planes4).  Returns nonzero when the net needs ojf_net_prepare_input (semantic
// channel, two heads).
struct NetInputSlot { float *x0; int cs4, P, h, w; int *ovf; int split; };
int net_input_slot(::ojf_net *net, NetInputSlot *slot);

}  // namespace ojf
