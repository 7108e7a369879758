/*
 * ojf_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's per-frame gather ("extract") and scatter
 * ("integrate") stages, used only as the parity checker for the HIP path (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under
 * online_joint_depthfusion_and_semantic_amd/ may link or call this file.
 *
 * Pinning: the reference has no tests or golden vectors for this path;
 * this restatement is pinned against outputs of the reference's own Python modules,
 * imported in the build container by tests/golden/make_golden.py, whose results are
 * committed under tests/golden/ (bit-exact: indices, corner weights, fusion_values,
 * fusion_weights, post-frame fp16 TSDF/weight volumes, u8 ids, fp16 scores).
 *
 * Each function cites the reference lines (relative to the reference repo root) it follows.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- IEEE binary16 <-> binary32, round-to-nearest-even (what torch .half()/.float() do) ---- */
static float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    float f;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: man * 2^-24, exact in f32 */
            f = (float)man * 5.9604644775390625e-8f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    memcpy(&f, &bits, 4);
    return f;
}

static uint16_t f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520 rounds to inf */
    if (ax < 0x38800000u) { /* below the smallest normal half: result = rint(|f| * 2^24) */
        float af;
        memcpy(&af, &ax, 4);
        return (uint16_t)(sign | (uint32_t)lrintf(af * 16777216.0f));
    }
    uint32_t h = (((ax >> 23) - 112u) << 10) | ((ax & 0x7fffffu) >> 13);
    uint32_t rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

/* ---- geometry of one ray sample ------------------------------------------------------------ */

/* modules/extractor.py:82-120 compute_coordinates.  fp32, in the accumulation order the
 * reference's two torch.matmul calls produce in the build container (probe: tests/golden/
 * make_golden.py --probe-matmul): K^-1 @ p (:114, [3,3] x transposed view) rounds every product
 * and every sum separately (k = 0,1,2); E[:3] @ [pc;1] (:117, [3,4] x contiguous) is a
 * single-rounding fma chain over k = 0..3 whose first term is a rounded product. */
static void unproject(int r, int c, float z, const float *Ki, const float *E, float pw[3])
{
    float u = (float)c * z; /* :112  points_p[:,:,0] = col * z */
    float v = (float)r * z; /* :113  points_p[:,:,1] = row * z */
    float pc[3];
    for (int i = 0; i < 3; ++i) { /* :114  K^-1 @ p */
        float t0 = Ki[3 * i + 0] * u, t1 = Ki[3 * i + 1] * v, t2 = Ki[3 * i + 2] * z;
        pc[i] = (t0 + t1) + t2;
    }
    for (int i = 0; i < 3; ++i) /* :115-117  E[:3] @ [pc;1] */
        pw[i] = fmaf(E[4 * i + 3], 1.0f,
                     fmaf(E[4 * i + 2], pc[2], fmaf(E[4 * i + 1], pc[1], E[4 * i + 0] * pc[0])));
}

/* modules/extractor.py:309-345 extract_values (method): voxel-space surface point, eye and unit
 * ray direction, all fp64 because origin is an fp64 tensor. */
static void ray_frame(const float pw[3], const float *E, const double *origin, double res,
                      double cv[3], double dir[3])
{
    double d[3];
    for (int i = 0; i < 3; ++i) {
        cv[i] = ((double)pw[i] - origin[i]) / res;            /* :314 */
        double ev = ((double)E[4 * i + 3] - origin[i]) / res; /* :315, eye = E[:, :3, 3] */
        d[i] = cv[i] - ev;                                    /* :317 */
    }
    /* :318 F.normalize -> linalg.vector_norm: the reference's CPU reduction accumulates x*x with
     * a contracted fma (probe: 0 of 19200 norms differ with this form, 1758 with separate
     * multiply and add), then one correctly rounded sqrt. */
    double ss = fma(d[2], d[2], fma(d[1], d[1], d[0] * d[0]));
    double nrm = sqrt(ss); /* x / max(||x||, eps=1e-12) */
    if (nrm < 1e-12) nrm = 1e-12;
    for (int i = 0; i < 3; ++i) dir[i] = d[i] / nrm;
}

/* sample k of 0..n_points-1 along the ray; the centre sample is k = half (:327-331) */
static void ray_sample(const double cv[3], const double dir[3], int k, int half, double p[3])
{
    for (int i = 0; i < 3; ++i) {
        if (k == half)
            p[i] = cv[i];
        else if (k > half)
            p[i] = cv[i] + (double)(k - half) * dir[i];
        else
            p[i] = cv[i] - (double)(half - k) * dir[i];
    }
}

/* modules/extractor.py:533-593 interpolation_weights: 8 corner indices and fp64 weights.
 * Quirk kept: the neighbour is idx + sign(centre - p), i.e. a point in the LOWER half of a voxel
 * pairs with the UPPER neighbour (the opposite of textbook trilinear). */
static void corners(const double p[3], int64_t idx[8][3], double wq[8])
{
    double fl[3], nb[3], a[3], ai[3];
    for (int i = 0; i < 3; ++i) {
        fl[i] = floor(p[i]);
        double ctr = fl[i] + 0.5;                           /* :537 */
        double s = ctr - p[i];                              /* :538 sign(centre - p) */
        nb[i] = (s > 0.0) ? 1.0 : ((s < 0.0) ? -1.0 : 0.0);
        a[i] = fabs(p[i] - ctr);                            /* :554 */
        ai[i] = 1.0 - a[i];                                 /* :555 */
    }
    for (int q = 0; q < 8; ++q) { /* :560-586, corner order (i,j,k) = 000,001,...,111 */
        int bi = (q >> 2) & 1, bj = (q >> 1) & 1, bk = q & 1;
        double w1 = bi ? a[0] : ai[0];
        double w2 = bj ? a[1] : ai[1];
        double w3 = bk ? a[2] : ai[2];
        wq[q] = w1 * w2 * w3; /* left-to-right product, :582 */
        idx[q][0] = (int64_t)(bi ? fl[0] + nb[0] : fl[0]);
        idx[q][1] = (int64_t)(bj ? fl[1] + nb[1] : fl[1]);
        idx[q][2] = (int64_t)(bk ? fl[2] + nb[2] : fl[2]);
    }
}

static int in_bounds(const int64_t i[3], int X, int Y, int Z)
{ /* modules/extractor.py:596-607, modules/integrator.py:129-145 */
    return i[0] >= 0 && i[0] < X && i[1] >= 0 && i[1] < Y && i[2] >= 0 && i[2] < Z;
}

/* ---- EXTRACT: modules/extractor.py:24-79 Extractor.forward + :640-681 trilinear_interpolation --
 * depth [h*w] f32 (unfiltered frame), Ki = inverse(K) f32[9] row-major, E f32[12] = rows of the
 * 3x4 camera-to-world matrix, volumes fp16 bit patterns [X,Y,Z] row-major.
 * Outputs: values/weights [h*w, n_points] f32.  Optional (may be NULL): idx [h*w,n_points,8,3] i64,
 * cw [h*w,n_points,8] f64, pts [h*w,n_points,3] f64, pcl [h*w,3] f32. */
int ojf_oracle_extract(const float *depth, const float *Ki, const float *E, const double *origin,
                       double res, const uint16_t *tsdf, const uint16_t *wgt, int X, int Y, int Z,
                       int h, int w, int n_points, float pad_value, float *out_values,
                       float *out_weights, int64_t *out_idx, double *out_cw, double *out_pts,
                       float *out_pcl)
{
    const int half = (n_points - 1) / 2; /* :59 */
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            const size_t n = (size_t)r * w + c;
            float pw[3];
            double cv[3], dir[3];
            unproject(r, c, depth[n], Ki, E, pw);
            if (out_pcl) memcpy(out_pcl + 3 * n, pw, sizeof pw);
            ray_frame(pw, E, origin, res, cv, dir);
            for (int k = 0; k < n_points; ++k) {
                double p[3], wq[8];
                int64_t idx[8][3];
                ray_sample(cv, dir, k, half, p);
                corners(p, idx, wq);
                /* :660-681: out-of-bounds corners read pad_value (-0.1, hard-coded in the
                 * reference) and weight 0; products are fp64; the 8-term fp64 sums of
                 * torch.sum(.., dim=1) (:673-674) run in ATen's row_sum order - four interleaved
                 * partial sums  s_k = p_k + p_{k+4},  then ((s_0 + s_1) + s_2) + s_3 - and are rounded
                 * once to fp32.  (A corner-order sum differs from it by one fp32 ulp on ~3 of 1e6
                 * samples of a volume with sign changes: found by the 320x240 -> 256^3 fuse_training
                 * fixture, whose target is interpolated from the ground-truth grid.) */
                double pv[8], pwt[8];
                for (int q = 0; q < 8; ++q) {
                    float val = pad_value, wt = 0.0f;
                    if (in_bounds(idx[q], X, Y, Z)) {
                        size_t lin = ((size_t)idx[q][0] * Y + (size_t)idx[q][1]) * Z + (size_t)idx[q][2];
                        val = h2f(tsdf[lin]);
                        wt = h2f(wgt[lin]);
                    }
                    pv[q] = (double)val * wq[q];
                    pwt[q] = (double)wt * wq[q];
                }
                const double sv = (((pv[0] + pv[4]) + (pv[1] + pv[5])) + (pv[2] + pv[6])) + (pv[3] + pv[7]);
                const double sw = (((pwt[0] + pwt[4]) + (pwt[1] + pwt[5])) + (pwt[2] + pwt[6])) + (pwt[3] + pwt[7]);
                const size_t s = n * n_points + k;
                out_values[s] = (float)sv;
                out_weights[s] = (float)sw;
                if (out_pts) memcpy(out_pts + 3 * s, p, sizeof p);
                if (out_cw) memcpy(out_cw + 8 * s, wq, sizeof wq);
                if (out_idx) memcpy(out_idx + 24 * s, idx, sizeof idx);
            }
        }
    return 0;
}

/* ---- INTEGRATE: modules/pipeline.py:137-171 _prepare_volume_update + modules/integrator.py:15-126
 * depth_filtered [h*w] = where(mask, depth, 0) (pipeline.py:196); a pixel integrates iff it is
 * non-zero (pipeline.py:145-146).  est [h*w, est_stride] f32 is the net output, the first n_tail
 * samples of each valid ray are clamped to +-trunc (pipeline.py:153-156) and scattered.
 * Entry order: valid pixel ascending, sample k, corner q (integrator.py:38-46).
 * Per-voxel sums are fp32 and strictly sequential in entry order (index_add_ on one thread,
 * integrator.py:59-67).  Semantics (ids/scores volumes may be NULL):
 * integrator.py:90-124 with "last writer wins" for duplicate indices.
 * n_touched (optional) receives the number of distinct voxels written. */
int ojf_oracle_integrate(const float *depth_filtered, const float *Ki, const float *E,
                         const double *origin, double res, const float *est, int est_stride,
                         int n_points, int n_tail, float trunc, uint16_t *tsdf, uint16_t *wgt,
                         const uint8_t *sem_ids, const float *sem_scores, uint8_t *id_vol,
                         uint16_t *score_vol, int X, int Y, int Z, int h, int w,
                         int64_t *n_touched)
{
    const size_t nvox = (size_t)X * Y * Z;
    const int half = (n_points - 1) / 2;
    const int sem = sem_ids && sem_scores && id_vol && score_vol;
    float *Wsum = (float *)calloc(nvox, sizeof(float));  /* integrator.py:59 */
    float *Usum = (float *)calloc(nvox, sizeof(float));  /* integrator.py:64 */
    uint8_t *hit = (uint8_t *)calloc(nvox, 1);
    uint32_t *list = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)h * w * n_tail * 8);
    uint8_t *id_new = NULL;
    uint16_t *sc_new = NULL;
    size_t n_list = 0;
    if (!Wsum || !Usum || !hit || !list) return -1;
    if (sem) {
        id_new = (uint8_t *)malloc(nvox);
        sc_new = (uint16_t *)malloc(nvox * 2);
        if (!id_new || !sc_new) return -1;
        memcpy(id_new, id_vol, nvox);
        memcpy(sc_new, score_vol, nvox * 2);
    }
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            const size_t n = (size_t)r * w + c;
            const float z = depth_filtered[n];
            if (!(z != 0.0f)) continue; /* pipeline.py:145 */
            float pw[3];
            double cv[3], dir[3];
            unproject(r, c, z, Ki, E, pw);
            ray_frame(pw, E, origin, res, cv, dir);
            for (int k = 0; k < n_tail; ++k) {
                double p[3], wq[8];
                int64_t idx[8][3];
                ray_sample(cv, dir, k, half, p);
                corners(p, idx, wq);
                float v = est[n * est_stride + k]; /* pipeline.py:153-156 */
                v = v < -trunc ? -trunc : (v > trunc ? trunc : v);
                for (int q = 0; q < 8; ++q) {
                    if (!in_bounds(idx[q], X, Y, Z)) continue; /* integrator.py:48-53 */
                    const size_t lin = ((size_t)idx[q][0] * Y + (size_t)idx[q][1]) * Z + (size_t)idx[q][2];
                    const float we = (float)wq[q]; /* integrator.py:45 .float() */
                    const float ue = we * v;       /* integrator.py:55 */
                    Wsum[lin] += we;               /* integrator.py:60 */
                    Usum[lin] += ue;               /* integrator.py:65 */
                    if (!hit[lin]) {
                        hit[lin] = 1;
                        list[n_list++] = (uint32_t)lin;
                    }
                    if (sem) { /* integrator.py:93-124; *_old are PRE-frame values */
                        const uint8_t id_e = sem_ids[n], id_old = id_vol[lin];
                        const float s_e = sem_scores[n], s_old = h2f(score_vol[lin]);
                        sc_new[lin] = f2h(s_e > s_old ? s_e : s_old);     /* :113-114,124 */
                        if (id_old != id_e)                                /* :105,117,123 */
                            id_new[lin] = (s_e > s_old) ? id_e : id_old;   /* :116 */
                    }
                }
            }
        }
    for (size_t t = 0; t < n_list; ++t) { /* integrator.py:72-88 */
        const size_t lin = list[t];
        const float w_old = h2f(wgt[lin]), v_old = h2f(tsdf[lin]);
        const float w_new = w_old + Wsum[lin];                     /* :77 */
        const float num = w_old * v_old + Usum[lin];               /* :82 */
        const float den = w_old + Wsum[lin];
        wgt[lin] = f2h(w_new);                                     /* :78,87 */
        tsdf[lin] = f2h(num / den);                                /* :83,88 */
    }
    if (sem) {
        memcpy(id_vol, id_new, nvox);
        memcpy(score_vol, sc_new, nvox * 2);
        free(id_new);
        free(sc_new);
    }
    if (n_touched) *n_touched = (int64_t)n_list;
    free(Wsum);
    free(Usum);
    free(hit);
    free(list);
    return 0;
}

/* Distinct in-bounds voxels among the gather (n_points samples, every pixel) and scatter
 * (n_tail samples, valid pixels) entries of one frame: U_g and U_s of SURVEY.md §8d, the
 * inputs of the algorithmic-bytes figure used by bench.py's roofline object. */
int ojf_oracle_unique_voxels(const float *depth, const float *depth_filtered, const float *Ki,
                             const float *E, const double *origin, double res, int X, int Y,
                             int Z, int h, int w, int n_points, int n_tail, int64_t *u_gather,
                             int64_t *u_scatter)
{
    const size_t nvox = (size_t)X * Y * Z;
    const int half = (n_points - 1) / 2;
    uint8_t *seen = (uint8_t *)calloc(nvox, 1);
    if (!seen) return -1;
    int64_t ug = 0, us = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const float *d = pass ? depth_filtered : depth;
        const int np_ = pass ? n_tail : n_points;
        const uint8_t bit = pass ? 2 : 1;
        for (int r = 0; r < h; ++r)
            for (int c = 0; c < w; ++c) {
                const size_t n = (size_t)r * w + c;
                if (pass && !(d[n] != 0.0f)) continue;
                float pw[3];
                double cv[3], dir[3];
                unproject(r, c, d[n], Ki, E, pw);
                ray_frame(pw, E, origin, res, cv, dir);
                for (int k = 0; k < np_; ++k) {
                    double p[3], wq[8];
                    int64_t idx[8][3];
                    ray_sample(cv, dir, k, half, p);
                    corners(p, idx, wq);
                    for (int q = 0; q < 8; ++q) {
                        if (!in_bounds(idx[q], X, Y, Z)) continue;
                        const size_t lin = ((size_t)idx[q][0] * Y + (size_t)idx[q][1]) * Z + (size_t)idx[q][2];
                        if (!(seen[lin] & bit)) {
                            seen[lin] |= bit;
                            if (pass) ++us; else ++ug;
                        }
                    }
                }
            }
    }
    free(seen);
    *u_gather = ug;
    *u_scatter = us;
    return 0;
}

/* fp16 helpers exported for tests */
float ojf_oracle_h2f(uint16_t h) { return h2f(h); }
uint16_t ojf_oracle_f2h(float f) { return f2h(f); }
