/*
 * ojf.h — C ABI of libojf.so, the MI355X (gfx950) implementation of the per-frame hot path of
 * online joint depth fusion + semantics:  extract (ray gather)  ->  fusion net  ->  integrate
 * (scatter).  This is the drop-in boundary: the reference itself has no native layer on this path
 * (SURVEY.md §2.3), so every entry point below names the reference *Python* interface it replaces.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; ojf_last_error() gives the message of the
 *     calling thread's last failure.  Nothing throws, nothing aborts.
 *   - "dev" pointers are device (HBM) addresses, "host" pointers are ordinary host memory read
 *     before the call returns.  No torch types cross this boundary.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     is asynchronous; the library never synchronises the device on the hot path.
 *   - volumes are [X,Y,Z] row-major (linear = (ix*Y + iy)*Z + iz, modules/integrator.py:57):
 *     TSDF fp16, weights fp16, semantic ids u8, semantic scores fp16 (modules/database.py:60-76).
 *   - camera: Kinv = inverse(float(K)) row-major f32[9] (modules/extractor.py:39,104),
 *     E = first three rows of the float32 camera-to-world matrix, row-major f32[12]
 *     (modules/extractor.py:40,57,115), origin f64[3] and resolution f64 (modules/database.py:57-58).
 */
#ifndef OJF_H
#define OJF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ojf_stream_t; /* hipStream_t */

/* integrate modes */
#define OJF_MODE_FAST 0   /* order-free fixed-point accumulation; deterministic; TSDF within 1 fp16 ulp */
#define OJF_MODE_PARITY 1 /* per-voxel fp32 sums in the reference's entry order; bit-exact TSDF/weights */

/* activation codes for ojf_conv2d */
#define OJF_ACT_NONE 0
#define OJF_ACT_RELU 1
#define OJF_ACT_LEAKY 2 /* negative slope 0.01 (nn.LeakyReLU default, modules/model.py:12) */
#define OJF_ACT_TANH 3

/* arithmetic of the MFMA kernels of the net (fp32 activations in memory, fp32 accumulation in both) */
#define OJF_ARITH_F32 0   /* v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain */
#define OJF_ARITH_F16X3 1 /* default: operands split into two fp16 halves, three v_mfma_f32_16x16x32_f16 per
                             product block; fp32-class accuracy (drops only lo*lo ~ 2^-22), operands < 65504 */

const char *ojf_version(void);
const char *ojf_last_error(void);
/* number of visible HIP devices, or <0 with ojf_last_error() set (no GPU / no driver) */
int ojf_device_count(void);

/* ---- EXTRACT -------------------------------------------------------------------------------
 * Replaces Extractor.forward (modules/extractor.py:24-79): compute_coordinates (:82-120),
 * extract_values (:309-345), interpolation_weights (:533-593), trilinear_interpolation (:640-681).
 * depth: dev f32[h*w], the UNFILTERED frame (modules/pipeline.py:202).
 * out_values / out_weights: dev f32.  out_layout 0 (rows): element (pixel n, sample k) at
 *   [n*out_stride + k] - the reference's fusion_values / fusion_weights are the out_stride == n_points
 *   case.  out_layout 1 (sample planes): element at [k*out_stride + n], out_stride >= h*w - every store of
 *   a wave is contiguous (the row layout scatters 4-byte stores at a 36-byte pitch: 6x write traffic).
 * pad_value: what out-of-volume corners read for TSDF (-0.1 in the reference, extractor.py:663).
 * Optional debug outputs (NULL to skip) reproduce the other entries of the reference's dict:
 *   dbg_indices i64[h*w,n_points,8,3], dbg_weights f64[h*w,n_points,8], dbg_points f64[h*w,n_points,3],
 *   dbg_pcl f32[h*w,3]. */
int ojf_extract(const float *depth_dev, const float *Kinv_host, const float *E_host,
                const double *origin_host, double resolution, const uint16_t *tsdf_dev,
                const uint16_t *weights_dev, int X, int Y, int Z, int h, int w, int n_points,
                float pad_value, float *out_values_dev, float *out_weights_dev, int out_stride,
                int out_layout, int64_t *dbg_indices_dev, double *dbg_weights_dev, double *dbg_points_dev,
                float *dbg_pcl_dev, ojf_stream_t stream);

/* ---- INTEGRATE -----------------------------------------------------------------------------
 * Replaces Pipeline._prepare_volume_update (modules/pipeline.py:137-171) + Integrator.forward
 * (modules/integrator.py:15-126).  Indices and corner weights are recomputed from depth and pose
 * (bit-identical to ojf_extract) instead of being materialised.
 * depth_filtered: dev f32[h*w] = where(mask, depth, 0) (pipeline.py:196); a pixel integrates iff != 0.
 * est: dev f32, net output for (pixel n, sample k) at [n*est_stride + k]; clamped to +-trunc here.
 * sem_ids u8[h*w] / sem_scores f32[h*w] / id_vol / score_vol: all four non-NULL to run the semantic
 *   update (integrator.py:90-124, "test" mode); all NULL to skip it.
 * workspace: dev memory of ojf_integrate_workspace_bytes(), prepared once by
 *   ojf_integrate_workspace_init(); it is left clean by every call and may be shared by all
 *   scenes of one grid size on one stream.
 * stats_dev (optional, NULL to skip - the counters cost ~14 us per frame of same-line atomics): dev u32[4]
 *   receiving {touched voxels, scatter entries, records (FAST), 0}.
 * Threading / streams: calls that share a workspace must be ordered on one stream (OJF_MODE_FAST alternates two counter
 *   sets per call; which one is next is kept IN the workspace header since round 4, so the call can be captured into a HIP
 *   graph and replayed); distinct workspaces are independent.
 * Range: per frame and voxel the summed corner weight must stay below 2^19 in the 2^-44 fixed point; beyond that the
 *   sums are redone in fp64 and the weight saturates to fp16 infinity like the reference's. */
size_t ojf_integrate_workspace_bytes(int X, int Y, int Z, int h, int w, int n_tail, int mode);
int ojf_integrate_workspace_init(void *workspace_dev, size_t workspace_bytes, int X, int Y, int Z,
                                 int h, int w, int n_tail, int mode, ojf_stream_t stream);
int ojf_integrate(const float *depth_filtered_dev, const float *Kinv_host, const float *E_host,
                  const double *origin_host, double resolution, const float *est_dev,
                  int est_stride, int n_points, int n_tail, float trunc, uint16_t *tsdf_dev,
                  uint16_t *weights_dev, const uint8_t *sem_ids_dev, const float *sem_scores_dev,
                  uint8_t *id_vol_dev, uint16_t *score_vol_dev, int X, int Y, int Z, int h, int w,
                  int mode, void *workspace_dev, size_t workspace_bytes, uint32_t *stats_dev,
                  ojf_stream_t stream);
/* Same, with the validity mask of modules/pipeline.py:196 applied inside the kernels: depth_dev is the RAW frame and
 * mask_dev u8[h*w] (torch bool; NULL = none) marks the valid pixels - torch.where(mask == 0, 0, frame) without its
 * launch.  Bit-identical to ojf_integrate on the filtered frame. */
int ojf_integrate_masked(const float *depth_dev, const uint8_t *mask_dev, const float *Kinv_host, const float *E_host,
                         const double *origin_host, double resolution, const float *est_dev, int est_stride,
                         int n_points, int n_tail, float trunc, uint16_t *tsdf_dev, uint16_t *weights_dev,
                         const uint8_t *sem_ids_dev, const float *sem_scores_dev, uint8_t *id_vol_dev,
                         uint16_t *score_vol_dev, int X, int Y, int Z, int h, int w, int mode, void *workspace_dev,
                         size_t workspace_bytes, uint32_t *stats_dev, ojf_stream_t stream);

/* Entry-list variant with the reference Integrator's own inputs (modules/integrator.py:15-126 fed by
 * Pipeline._prepare_volume_update, modules/pipeline.py:137-171): n_rows = valid pixels x n_tail rows, each
 * with a clamped value f32, 8 corner indices i64[8,3] and 8 corner weights f64[8] (all dev); row_ids u8 /
 * row_scores f32 per row with id_vol / score_vol for the semantic update (all four or none).  FAST
 * accumulation; the workspace is the OJF_MODE_FAST one of a (h*w*n_tail >= n_rows) configuration. */
int ojf_integrate_entries(const float *values_dev, const int64_t *indices_dev, const double *weights_dev,
                          const uint8_t *row_ids_dev, const float *row_scores_dev, int64_t n_rows,
                          uint16_t *tsdf_dev, uint16_t *weights_vol_dev, uint8_t *id_vol_dev,
                          uint16_t *score_vol_dev, int X, int Y, int Z, void *workspace_dev,
                          size_t workspace_bytes, uint32_t *stats_dev, ojf_stream_t stream);

/* ---- FUSION NET ------------------------------------------------------------------------------
 * Replaces FusionNet_v3.forward / FusionNet_v2.forward in eval mode (modules/model.py:164-283)
 * together with Pipeline._prepare_fusion_input / _fusion (modules/pipeline.py:62-102).
 * The host folds every BatchNorm into the preceding convolution and hands the folded layers over
 * in the canonical order documented in DESIGN.md §"net layer order"; the library repacks them for
 * the MFMA kernels and owns that copy plus the activation workspace (sized for h x w). */
typedef struct ojf_net ojf_net;
typedef struct ojf_conv_layer {
    int c_in, c_out, ksize, dilation; /* square kernels, stride 1, padding = dilation*(ksize/2) */
    const float *weight_host;         /* [c_out, c_in, ksize, ksize] fp32, BN folded */
    const float *bias_host;           /* [c_out] fp32, BN folded */
} ojf_conv_layer;

/* version: 2 or 3 (FusionNet_v2 / FusionNet_v3).  n_points = 9, growth = growth_factor - 1 = 5 in
 * every reference config (the YAML files under configs/fusion). */
int ojf_net_create(ojf_net **out, int version, int n_points, int growth, int use_semantics,
                   float output_scale, const ojf_conv_layer *layers_host, int n_layers, int h,
                   int w);
void ojf_net_destroy(ojf_net *net);
/* number of folded conv layers ojf_net_create expects for this topology, <0 if unsupported */
int ojf_net_layer_count(int version, int n_points, int growth, int use_semantics);
/* Packs the net's input from the extractor's row-major outputs (modules/pipeline.py:74-102):
 * channels [0,P) fusion_values, [P,2P) fusion_weights (both f32, laid out as ojf_extract's out_layout
 * `in_layout` with stride `rows_stride`), [2P] the depth
 * frame; with semantics (v3) a second head gets values | weights | (1+sem_id)/n_classes
 * (model.py:274), v2 appends the semantic channel to the single head (model.py:207).  The net's
 * internal activation layout (planes of 4 channels) is private to the library. */
int ojf_net_prepare_input(ojf_net *net, const float *values_dev, const float *weights_dev, int rows_stride,
                          int in_layout, const float *depth_dev, const uint8_t *sem_ids_dev, int n_classes,
                          ojf_stream_t stream);
/* Runs the net; est_dev receives output_scale*tanh(.) for (pixel n, sample k) at [n*est_stride+k]. */
int ojf_net_forward(ojf_net *net, float *est_dev, int est_stride, ojf_stream_t stream);
/* useful multiply-accumulates per pixel of this topology (padding excluded) */
int64_t ojf_net_macs_per_pixel(const ojf_net *net);
/* Kernel launches of the most recent ojf_net_forward on this net (all streams; 0 before the first call). */
int ojf_net_launch_count(const ojf_net *net);
/* The streams the net launches on beside `stream` (the second head of a two-head net, the side streams of the unfused
 * VortexPooling flow), after pairing them with `stream` as the first forward pass would: out[0..2], NULL where there is
 * none; returns how many are non-NULL.  For callers that run other work beside the net (Pipeline's look-ahead pass) and
 * want a stream that shares a hardware queue with none of them (ojf_streams_overlap). */
int ojf_net_side_streams(ojf_net *net, ojf_stream_t stream, ojf_stream_t out[3]);
/* Profiling run of one forward pass (same launches as ojf_net_forward, plus one fence-free HIP event behind every
 * launch): kernel names ('\n'-separated, in launch order) into `names` and each launch's duration in microseconds as
 * its stream saw it (time since the previous launch of that stream completed) into `micros`.  Synchronises the
 * stream.  Returns the number of entries (<= max_entries) or < 0.  For bench.py's per-kernel table; not a hot path. */
int ojf_net_profile(ojf_net *net, float *est_dev, int est_stride, ojf_stream_t stream, char *names, int names_cap,
                    float *micros, int max_entries);
/* Arithmetic used by nets created (and ojf_conv2d calls made) AFTER this call: OJF_ARITH_F32 | OJF_ARITH_F16X3.
 * A net keeps the arithmetic it was created with.  Not thread-safe against concurrent ojf_net_create. */
int ojf_net_set_arithmetic(int arithmetic);
/* arithmetic of `net`, or the current default when net == NULL */
int ojf_net_get_arithmetic(const ojf_net *net);
/* Range guard of OJF_ARITH_F16X3.  Every kernel that produces a value a later layer splits into fp16 halves
 * raises a process-wide flag if that value is beyond +-65504 (it cannot be split; a trained, BN-folded FusionNet
 * stays orders of magnitude below).  NaN inputs are not violations: they propagate to NaN outputs as in fp32.
 * ojf_net_forward polls the flag without synchronising and fails once it is set; ojf_net_check synchronises `stream`, returns non-zero (and clears the flag) if it was raised
 * since the last check.  Results produced in between are invalid: switch that network to OJF_ARITH_F32.
 * The volumes are protected: while the flag is set every ojf_integrate / ojf_integrate_masked / ojf_integrate_entries
 * call (both modes) returns 0 without touching a volume - the frame whose net tripped the guard and every frame
 * enqueued after it stay unfused until ojf_net_check has reported (and cleared) the event, so a scene is never
 * corrupted silently; the caller re-fuses those frames after switching the arithmetic (Pipeline does, with
 * FUSION_MODEL.guard_policy = 'f32').  (The reference has no counterpart: its fp32 torch ops have no range limit,
 * modules/pipeline.py:62-72.) */
int ojf_net_check(ojf_stream_t stream);
/* Synchronises `stream`; *flag = 0 none | 1 range violation | 2 internal (dense chain gave up waiting), *skipped = the
 * number of integrate calls skipped since the flag was raised.  Clears nothing (ojf_net_check does).  Either pointer may
 * be NULL. */
int ojf_guard_status(ojf_stream_t stream, int *flag, int *skipped);
/* The flag as the host sees it right now (no synchronisation: what ojf_net_forward tests). */
int ojf_guard_poll(void);
/* 1 when kernels on `a` and `b` run side by side, 0 when the runtime mapped the two streams to one hardware queue (they
 * then take turns: a small round-robin pool of queues backs all streams), < 0 on error.  Synchronises both streams and
 * runs two ~130-us one-wave spin kernels: for set-up code that creates side streams (Pipeline.fuse_many's slot streams,
 * the look-ahead stream of Pipeline.fuse_sequence), not for the frame path.  (No counterpart in the reference, which
 * runs on one stream.) */
int ojf_streams_overlap(ojf_stream_t a, ojf_stream_t b);

/* Stand-alone fused convolution on NHWC fp32 rows (the kernel the net is built from; exported so
 * the parity tests can pin it layer by layer against torch.nn.functional.conv2d).
 * in: [h*w, in_stride], channels [in_off, in_off+c_in); out likewise.  weight [c_out,c_in,k,k]. */
int ojf_conv2d(const float *in_dev, int in_stride, int in_off, float *out_dev, int out_stride,
               int out_off, const ojf_conv_layer *layer_host, int act, int h, int w,
               ojf_stream_t stream);

/* ---- SEGCONV: convolutions of the 2-D semantic front-end (AdapNet++) ---------------------------
 * One prepacked layer = nn.Conv2d (groups 1, square kernel <= 7, any stride / dilation / zero padding) with what
 * follows it in modules/adapnet.py folded in: eval-mode BatchNorm as scale_host[c_out] (multiplied into the weights)
 * and bias_host[c_out] (either may be NULL), then per call an optional residual add (Bottleneck :33-38 /
 * BottleneckSSMA :60-84), activation act = 0 none | 1 ReLU | 2 sigmoid (SSMA gate :331-337) and an optional
 * elementwise product with mul_dev after the activation (x * link(x), :353).
 * Tensors are NHWC fp32, batch 1 (torch channels_last): pointer to the first channel of pixel 0 + floats per pixel
 * row, so channel slices of a concatenation buffer are addressed in place.  The input rows must hold
 * round_up(c_in, 8) readable, finite channels (the extra ones meet zero weights) and be 16-byte aligned.
 * Output size: floor((h + 2*padding - dilation*(ksize-1) - 1) / stride) + 1 per axis, like torch.
 * Arithmetic: split-fp16 MFMA with fp32 accumulation (see ojf_net_set_arithmetic); the range guard of
 * ojf_net_check covers these launches too.  weight_host is [c_out][c_in][ksize][ksize] (torch layout).
 * act may carry OJF_SEG_ACT_ZERO_PAD: the launch also writes zeros into channels c_out .. round_up(c_out, 8) - 1 of every
 * output row (rows the caller owns with that many floats), which the next layer reads as part of its last group of 8 -
 * instead of a fill launch in front of every such layer. */
#define OJF_SEG_ACT_ZERO_PAD 0x100
typedef struct ojf_segconv ojf_segconv;
int ojf_segconv_create(ojf_segconv **out, const float *weight_host, const float *scale_host, const float *bias_host,
                       int c_in, int c_out, int ksize, int stride, int dilation, int padding);
/* nn.ConvTranspose2d(c_in, c_out, kernel 2*stride, stride, padding stride/2) - the three upsampling layers of the
 * decoder (adapnet.py:226,236,247) - as a 3x3 convolution to stride^2 phase copies of the channels whose store does
 * the pixel shuffle.  weight_host is [c_in][c_out][2*stride][2*stride] (torch layout); scale / bias as above.  Run it
 * with ojf_segconv_forward: h, w are the INPUT size, out_dev is the [h*stride, w*stride] NHWC tensor; no residual /
 * gate.  Deterministic (MIOpen's transposed convolution sums with atomics). */
int ojf_segdeconv_create(ojf_segconv **out, const float *weight_host, const float *scale_host, const float *bias_host,
                         int c_in, int c_out, int stride);
void ojf_segconv_destroy(ojf_segconv *conv);
/* The reference's multi-scale units end in `nn.Dropout(p=0.5)(out)` on a module constructed inside forward(), i.e.
 * ALWAYS in training mode (modules/adapnet.py:80-82): activations are dropped at inference too.  With a state set, the
 * launches of this layer apply that dropout in their epilogue (after residual + ReLU): element e of the layer is kept
 * (and doubled) iff bit 0 of Philox-4x32-10(counter = {e / 4, stream_id, frame}, key = seed) word e % 4 is set, with
 * rng_state_dev = {seed, frame} (two u64 in device memory, owned by the caller).  The masks are a pure function of
 * (seed, frame, stream_id, e): deterministic, graph-replayable, a fresh draw per frame.  advance != 0 instead marks
 * the layer whose launch increments `frame` (the LAST convolution of a forward pass; no dropout there).
 * rng_state_dev == NULL switches both off.  (torch's own generator is not consumed: seed it from torch.initial_seed().) */
int ojf_segconv_set_dropout(ojf_segconv *conv, const unsigned long long *rng_state_dev, unsigned stream_id, int advance);
int ojf_segconv_forward(const ojf_segconv *conv, const float *in_dev, int in_stride, float *out_dev, int out_stride,
                        const float *res_dev, int res_stride, const float *mul_dev, int mul_stride, int act, int h,
                        int w, ojf_stream_t stream);
/* n (1..8) convolutions of one shape - channels, kernel size, stride, frame; dilation / padding may differ as long as the
 * output size agrees - as ONE launch: the two modality encoders of modules/adapnet.py:364-380 in lock-step, the two
 * dilations of a multi-scale unit (:56-73), the three cascades of an eASPP (:175-202).  Arrays of n device pointers
 * (ress / muls: NULL, or per-member pointers); the row strides are common to the members.  Same arithmetic and the same
 * bits per member as n ojf_segconv_forward calls. */
int ojf_segconv_forward_group(int n, const ojf_segconv *const *convs, const float *const *ins_dev, int in_stride,
                              float *const *outs_dev, int out_stride, const float *const *ress_dev, int res_stride,
                              const float *const *muls_dev, int mul_stride, int act, int h, int w, ojf_stream_t stream);
/* The same on `batch` (1..64) images per tensor: every tensor is [batch, h, w, C] NHWC (batch-major, same row stride), the
 * frames of several scenes through the 2-D network at once (Pipeline.fuse_many).  A pixel's result does not depend on
 * the batch it travels in as long as the launch takes the same kernel form (the form is chosen by the number of pixel
 * tiles, and the forms differ in the order they add the K blocks): equal to the single-image call up to that rounding
 * (a few 1e-7 relative).  The weights of a layer are fetched once per launch whatever the batch: on the 15x20 / 30x40
 * maps, where a single frame leaves the matrix pipe idle behind the weight stream, this is what batching buys. */
/* n (1..8) convolutions of DIFFERENT shapes that do not depend on each other, as one launch where their kernel forms allow it
 * (separate launches otherwise; same results either way up to the K-block order of the form): a residual unit's shortcut
 * convolution next to its first 1x1 (modules/adapnet.py:33-38,75-76 run them one after the other), an encoder's skip projection
 * (:142-147) next to the next stage's first layers, the three SSMA blocks of the decoder (:367-369,404-408).  Everything is per
 * member: row strides, residual / gate (NULL entries allowed), act (with OJF_SEG_ACT_ZERO_PAD), input size. */
int ojf_segconv_forward_multi(int n, int batch, const ojf_segconv *const *convs, const float *const *ins_dev, const int *in_strides,
                              float *const *outs_dev, const int *out_strides, const float *const *ress_dev, const int *res_strides,
                              const float *const *muls_dev, const int *mul_strides, const int *acts, const int *hs, const int *ws,
                              ojf_stream_t stream);
int ojf_segconv_forward_batch(const ojf_segconv *conv, int batch, const float *in_dev, int in_stride, float *out_dev, int out_stride,
                              const float *res_dev, int res_stride, const float *mul_dev, int mul_stride, int act, int h,
                              int w, ojf_stream_t stream);
int ojf_segconv_forward_group_batch(int n, int batch, const ojf_segconv *const *convs, const float *const *ins_dev, int in_stride,
                                    float *const *outs_dev, int out_stride, const float *const *ress_dev, int res_stride,
                                    const float *const *muls_dev, int mul_stride, int act, int h, int w, ojf_stream_t stream);

/* ---- FUSION NET, TRAINING (modules/pipeline.py:301-363 with the net in train() mode; csrc/ojf_net_train.h) -------
 * One layer unit of the reference's Sequentials (modules/model.py:4-52,115-141) - conv -> BatchNorm2d (batch
 * statistics) -> activation -> Dropout2d - as device operations on "C4 planes": a tensor with C channels padded to
 * c_phys (multiple of 4) is c_phys/4 planes of float4, element (group cg, pixel p) at [(g0 + cg) * h*w + p]; g0 selects
 * a channel window inside a wider buffer.  The Python wrapper (train.py) turns a unit into one autograd node.
 * Concatenated inputs keep `group`-wide tensors in `slot`-wide slots (19 in 20, 114 in 116): logical input channel j
 * sits at physical channel (j / group) * slot + j % group.
 * ojf_train_packed_floats: floats of a packed weight buffer (0 = unsupported shape).
 * ojf_train_pack: torch weights [OC][IC][k][k] (+ bias [OC]) -> fragment layout of the fp32-MFMA conv kernel, on the
 *   device; transposed != 0 packs the transposed, tap-flipped form whose convolution is the backward-data pass.
 * ojf_train_conv: y = conv(x) + bias on planes (c_out_phys of any size; dilation for 3x3).
 * ojf_train_bn_act: out = drop[c] * scale * act(gamma[c] * (y - mean[c]) * invstd[c] + beta[c]) (has_bn == 0: act(y)).
 *   With has_bn: mean and 1/sqrt(var + eps) per channel are computed here - training != 0: batch statistics (biased
 *   variance, fp64 sums over pixel slabs added in a fixed order) plus the running-statistics update of nn.BatchNorm2d
 *   (momentum, unbiased variance); training == 0: from the running statistics - and returned in mean / invstd for the
 *   backward pass.  partial: ojf_train_partial_doubles(c_phys) doubles of scratch.
 * ojf_train_bn_act_bwd: gradient of that w.r.t. y (batch-statistics form when training != 0), gamma, beta and the
 *   convolution's bias.  accumulate != 0 (here and in ojf_train_wgrad): the parameter gradients are ADDED to the buffers
 *   (gradient accumulation over frames, train_fusion.py:174-189, without a torch add per parameter).
 * ojf_train_wgrad: dW[oc][ic][tap] = sum_p dy[oc][p] * x[ic][p + tap] into torch's layout; partial:
 *   ojf_train_wgrad_partial_floats(...) floats of scratch (pixel slabs, added in slab order: deterministic). */
size_t ojf_train_packed_floats(int c_out_phys, int c_in_phys, int ksize);
int ojf_train_pack(const float *w_dev, const float *bias_dev, int oc, int ic, int ksize, int group, int slot, int c_in_phys,
                   int c_out_phys, int transposed, float *packed_dev, float *bias_packed_dev, ojf_stream_t stream);
int ojf_train_conv(const float *in_dev, int in_g0, int c_in_phys, float *out_dev, int out_g0, int c_out_phys,
                   const float *packed_dev, const float *bias_packed_dev, int ksize, int dilation, int h, int w, ojf_stream_t stream);
/* nn.AvgPool2d(3, 1, 1) on planes (count_include_pad); symmetric: the same call is its backward pass. */
int ojf_train_avgpool3(const float *in_dev, float *out_dev, int c_phys, int h, int w, ojf_stream_t stream);
size_t ojf_train_partial_doubles(int c_phys);
/* per-channel sums of a plane tensor as 64 partial rows [slab][c_phys / 4][8] (sums, then sums of squares), fixed order */
int ojf_train_channel_sums(const float *y_dev, int y_g0, int c_phys, int h, int w, double *partial_dev, ojf_stream_t stream);
int ojf_train_bn_act(const float *y_dev, int y_g0, float *out_dev, int out_g0, int c_phys, int c, int h, int w, const float *gamma_dev,
                     const float *beta_dev, const float *drop_dev, int act, float scale, int has_bn, int training, float momentum,
                     float eps, float *running_mean_dev, float *running_var_dev, double *partial_dev, float *mean_dev,
                     float *invstd_dev, ojf_stream_t stream);
int ojf_train_bn_act_bwd(const float *y_dev, int y_g0, const float *dout_dev, int dout_g0, float *dy_dev, int dy_g0, int c_phys, int c,
                         int h, int w, const float *mean_dev, const float *invstd_dev, const float *gamma_dev, const float *beta_dev,
                         const float *drop_dev, int act, float scale, int has_bn, int training, double *partial_dev,
                         float *dgamma_dev, float *dbeta_dev, float *dbias_dev, int accumulate, ojf_stream_t stream);
size_t ojf_train_wgrad_partial_floats(int c_out_phys, int c_in_phys, int ksize, int h, int w);
int ojf_train_wgrad(const float *x_dev, int x_g0, int c_in_phys, const float *dy_dev, int dy_g0, int c_out_phys, int oc, int ic,
                    int ksize, int dilation, int group, int slot, int h, int w, float *partial_dev, float *dw_dev, int accumulate,
                    ojf_stream_t stream);

/* ---- whole-net TRAINING executor (csrc/ojf_train_net.h) ---------------------------------------------------------
 * The net of modules/pipeline.py:322 inside fuse_training and the loss.backward() of train_fusion.py:171 as two calls:
 * forward and backward pass of FusionNet_v3 / _v2 (modules/model.py:164-283) in train() or eval() mode on the layer-unit
 * kernels above, with every activation / gradient buffer owned by the trainer and the four branches of a VortexPooling
 * run as grouped launches.  Parameters, BatchNorm running statistics and gradient tensors stay the caller's: every call
 * receives a table of n_layers = ojf_trainer_layer_count() entries in the layer order of ojf_net_create
 * (v3: block0 (2 per Block) | vortex0 = {global-average conv, 4 x {1x1, 3x3, 3x3, 1x1}, final} | [block2 | vortex2] |
 * vortex3 | pred; v2: block | vortex | vortex_final | pred).
 *   weight [oc][ic][k][k], bias [oc] (or NULL); gamma / beta NULL = no BatchNorm after this convolution;
 *   bn_training: batch statistics + running-statistics update (nn.BatchNorm2d.training of THAT module);
 *   drop_scale: [oc] per-channel Dropout2d factors (0 or 1 / (1 - p)) of this pass, NULL = no dropout;
 *   grad_*: where ojf_trainer_backward writes (accumulate == 0) or adds (accumulate != 0) the parameter gradients.
 * ojf_trainer_forward: values / weights [n_points][h][w], frame [h][w] (+ semantic_frame [h][w] when the net has a
 *   semantic channel) -> est [n_points][h][w] = net(x) * output_scale, all device fp32 NCHW.  weights_epoch: any number
 *   that changes whenever a weight or bias changed (the packed copies are refreshed then).
 * ojf_trainer_backward: d_est [n_points][h][w] -> parameter gradients; exactly one backward per forward (the forward's
 *   activations live in the trainer).  No gradient w.r.t. the inputs is produced (fuse_training's come from the volumes).
 * A trainer serves one stream at a time (its buffers, its side stream and its dy factors are per-trainer state); create one
 * per concurrent training stream.  Frame sizes are arbitrary (no multiple-of-16 requirement). */
typedef struct ojf_trainer ojf_trainer;
typedef struct ojf_train_layer {
    const float *weight, *bias, *gamma, *beta;
    float *running_mean, *running_var;
    float *grad_weight, *grad_bias, *grad_gamma, *grad_beta;
    const float *drop_scale;
    int out_channels, in_channels, ksize, dilation;
    int bn_training, accumulate;
    float momentum, eps;
} ojf_train_layer;
int ojf_trainer_create(ojf_trainer **out, int version, int n_points, int growth, int use_semantics, float output_scale, int h, int w);
void ojf_trainer_destroy(ojf_trainer *t);
/* arithmetic of the convolutions: OJF_ARITH_F16X3 (default; activations must stay inside the fp16 range like in
 * ojf_net_forward - a violation makes the next ojf_trainer_forward fail; the flag STAYS set (it is shared with the inference
 * executors and the integrate calls, which skip while it is set): the caller reports and clears it with ojf_net_check
 * before it continues - a loop that catches the error and carries on without that call has every later ojf_integrate* skip.
 * Pipeline.fuse_training does this before it re-raises) or OJF_ARITH_F32.
 * Under OJF_ARITH_F16X3 backward-data AND the weight gradients run in split-fp16 as well: every dy tensor is stored under a
 * per-unit power-of-two factor that the SAME pass derives from a guaranteed bound on |dy| (max|dz| and max|xhat| collected by
 * the BatchNorm-backward reduction), so gradients of any magnitude stay inside the fp16 range; the factor is divided out
 * exactly by the consumers. */
int ojf_trainer_set_arithmetic(ojf_trainer *t, int arithmetic);
/* arithmetic of the BACKWARD convolutions (backward-data and weight gradients) under a split-fp16 forward:
 * OJF_ARITH_F16X3 (default: as described above) or OJF_ARITH_F32 (every backward pass on the fp32-input MFMA path).
 * Replaces nothing in the reference (loss.backward() of train_fusion.py:171 is fp32 throughout). */
int ojf_trainer_set_backward_arithmetic(ojf_trainer *t, int arithmetic);
int ojf_trainer_layer_count(const ojf_trainer *t);
int ojf_trainer_launch_count(const ojf_trainer *t);
/* Optional replay of the passes as device graphs (default off; OJF_TRAIN_GRAPH=1 in the environment turns it on at create):
 * the launches of a forward / backward pass between the trainer's own buffers and the layer table's tensors are captured once
 * per (layer table contents, arithmetic) on a stream of the trainer's own and replayed with one hipGraphLaunch on the caller's
 * stream - same kernels, same arguments, same bits; a pass is captured the second time its key misses in a row, so a table
 * that changes every frame stays on plain launches.  The launches that touch the per-frame tensors (net input, est, d_est) and
 * the weight packing (weights_epoch) always run as plain launches around the replay.  ojf_trainer_graph_replays: passes served
 * by a replay since create (a test hook).  Replaces nothing in the reference. */
int ojf_trainer_set_graph(ojf_trainer *t, int enable);
int ojf_trainer_graph_replays(const ojf_trainer *t);
int ojf_trainer_forward(ojf_trainer *t, const ojf_train_layer *layers, int n_layers, unsigned long long weights_epoch,
                        const float *values_dev, const float *weights_dev, const float *frame_dev, const float *semantic_frame_dev,
                        float *est_dev, ojf_stream_t stream);
int ojf_trainer_backward(ojf_trainer *t, const ojf_train_layer *layers, int n_layers, const float *d_est_dev, ojf_stream_t stream);

/* The frame step's glue around the net, one launch each instead of ~70 tensor operations:
 * ojf_train_fuse_output(_bwd): modules/pipeline.py:104-127 - fused = (max(w, 0) v + clamp(est, +-init)) / (max(w, 0) + 1) on
 *   the sample planes [n_points][n] (est: net output, v / w: ojf_extract with out_layout 1), gathered at the n_valid pixel
 *   indices valid[] into rows [n_valid][n_points]; backward: d est planes (zero at masked pixels and outside the clamp).
 * ojf_train_fusion_loss(_bwd): utils/loss.py:65-103 FusionLoss on rows [n_valid][n_points] -> *loss_out (device scalar),
 *   including the reference's reshape quirk of the sign-cosine term; fp64 partial sums in fixed order (partial:
 *   ojf_train_loss_partial_doubles(n_valid) doubles of scratch).  Backward: d est rows = *grad_out * d loss / d est. */
int ojf_train_fuse_output(const float *est_planes_dev, const float *values_planes_dev, const float *weights_planes_dev,
                          const long long *valid_dev, int n, int n_points, long long n_valid, float init_value, float *fused_rows_dev,
                          ojf_stream_t stream);
int ojf_train_fuse_output_bwd(const float *d_fused_rows_dev, const float *est_planes_dev, const float *weights_planes_dev,
                              const long long *valid_dev, int n, int n_points, long long n_valid, float init_value,
                              float *d_est_planes_dev, ojf_stream_t stream);
size_t ojf_train_loss_partial_doubles(long long n_valid);
int ojf_train_fusion_loss(const float *est_rows_dev, const float *target_rows_dev, long long n_valid, int n_points, float w_l1, float w_l2,
                          float w_cos, double *partial_dev, float *loss_out_dev, ojf_stream_t stream);
int ojf_train_fusion_loss_bwd(const float *est_rows_dev, const float *target_rows_dev, long long n_valid, int n_points, float w_l1,
                              float w_l2, const float *grad_out_dev, float *d_est_rows_dev, ojf_stream_t stream);

/* ojf_extract writing straight into the fusion net's input planes (values | weights | depth of
 * modules/pipeline.py:74-102), bit-identical to ojf_extract + ojf_net_prepare_input, for nets without a semantic
 * channel and with one head (others: error - use the two calls): no sample planes, no prepare launch. */
int ojf_extract_to_net(const float *depth_dev, const float *Kinv_host, const float *E_host, const double *origin_host,
                       double resolution, const uint16_t *tsdf_dev, const uint16_t *weights_dev, int X, int Y, int Z, int h,
                       int w, int n_points, float pad_value, ojf_net *net, ojf_stream_t stream);

/* ---- one frame of each of SEVERAL scenes as single launches (round 6; Pipeline.fuse_many) ------------------------------
 * modules/extractor.py:24-79 and modules/integrator.py:15-124 for n <= OJF_MAX_SCENES frames of DISTINCT scenes (one frame
 * size, one sample count, one grid size; every job with its own volumes, camera, outputs and integrate workspace): the
 * kernels of ojf_extract / ojf_extract_to_net and of ojf_integrate_masked with the scene as blockIdx.y, so that the
 * dependent-step chains of S frames share one launch ramp and fill the chip while other blocks wait.  Per scene the same
 * blocks run the same code on the same arguments: every output and every volume comes out bit for bit as from the n
 * separate calls.  The reference has no counterpart (its drivers fuse one frame at a time: test_fusion.py:68-80).
 * ojf_extract_job: `net` set = ojf_extract_to_net into that net's input planes (out_* ignored); else out_values /
 *   out_weights with out_stride / out_layout as in ojf_extract.
 * ojf_integrate_job: the arguments of ojf_integrate_masked (mask / semantic pointers NULL where unused, semantics for all
 *   jobs or none); OJF_MODE_FAST only - the PARITY mode's sort is per scene, call ojf_integrate_masked for it. */
#define OJF_MAX_SCENES 8
typedef struct ojf_extract_job {
    const float *depth_dev;
    const float *Kinv_host, *E_host;
    const double *origin_host;
    double resolution;
    const uint16_t *tsdf_dev, *weights_dev;
    ojf_net *net;
    float *out_values_dev, *out_weights_dev;
    int out_stride, out_layout;
} ojf_extract_job;
typedef struct ojf_integrate_job {
    const float *depth_dev;
    const uint8_t *mask_dev;
    const float *Kinv_host, *E_host;
    const double *origin_host;
    double resolution;
    const float *est_dev;
    int est_stride;
    uint16_t *tsdf_dev, *weights_dev;
    const uint8_t *sem_ids_dev;
    const float *sem_scores_dev;
    uint8_t *id_vol_dev;
    uint16_t *score_vol_dev;
    void *workspace_dev;
    size_t workspace_bytes;
} ojf_integrate_job;
int ojf_extract_many(int n, const ojf_extract_job *jobs, int X, int Y, int Z, int h, int w, int n_points, float pad_value,
                     ojf_stream_t stream);
int ojf_integrate_many(int n, const ojf_integrate_job *jobs, int n_points, int n_tail, float trunc, int X, int Y, int Z, int h,
                       int w, ojf_stream_t stream);

/* ---- AdapNet++ front end: the operators around the convolutions (csrc/ojf_seg_ops.hip) --------------------
 * NHWC fp32 rows of batch 1 like ojf_segconv_forward (pointer to channel 0 of pixel 0 + floats per pixel row).
 * ojf_seg_pack_input: modules/pipeline.py:44,50 - three source planes (src[c * chan_stride + p]; chan_stride 0 =
 *   the depth map replicated to three channels) divided by `divisor` (255 for the colour image, 1 for depth) into the
 *   8-channel rows the stem convolution reads (channels 3..7 zero).
 * ojf_seg_maxpool: nn.MaxPool2d(3, 2, 1) of the ResNet stem (modules/adapnet.py:101, torchvision layout).
 * ojf_seg_mean: mean over the pixels per channel (eASPP branch 5 :204-208, Decoder._skip :292-296) -> out[c].
 * ojf_seg_broadcast: out[p][c] = vec[c] (* mul[p][c]): the bilinear upsampling of a 1x1 map / the gated skip.
 * ojf_seg_softmax_max: pipeline.py:57,183 softmax over the classes then max: scores f32[npix], ids u8[npix].
 * ojf_seg_pool_fc: the squeeze chains of eASPP branch 5 (adapnet.py:204-210) and Decoder._skip (:292-296) as two launches
 *   for n (1..8) members: out_m[p][c] = act(bias[c] + sum_k W[c][k] * mean_p' in_m[p'][k]) (* mul_m[p][c]) for every pixel p
 *   of the OUTPUT map - global average (two fixed-order stages), a 1x1 convolution on the 1x1 map in fp32 (W_dev [c_out][c_in],
 *   bias_dev [c_out] or NULL, per member), ReLU (act 1) or none (0), and the broadcast that bilinear upsampling of a 1x1
 *   map is.  partial_dev: n * 128 * c_in floats of scratch. */
int ojf_seg_pack_input(const float *src_dev, int chan_stride, float divisor, int h, int w, float *out_dev, int out_stride,
                       ojf_stream_t stream);
int ojf_seg_maxpool(const float *in_dev, int in_stride, int c, int h, int w, float *out_dev, int out_stride, ojf_stream_t stream);
int ojf_seg_maxpool_batch(int batch, const float *in_dev, int in_stride, int c, int h, int w, float *out_dev, int out_stride,
                          ojf_stream_t stream);  /* [batch, h, w, C] tensors */
int ojf_seg_mean(const float *in_dev, int in_stride, int c, int npix, float *partial_dev /* 32 * c floats of scratch */,
                 float *out_dev, ojf_stream_t stream);
int ojf_seg_broadcast(const float *vec_dev, const float *mul_dev, int mul_stride, int c, int npix, float *out_dev, int out_stride,
                      ojf_stream_t stream);
int ojf_seg_pool_fc(int n, const float *const *ins_dev, int in_stride, int c_in, int npix_in, const float *const *weights_dev,
                    const float *const *biases_dev, int c_out, int act, const float *const *muls_dev, int mul_stride,
                    float *const *outs_dev, int out_stride, int npix_out, float *partial_dev, ojf_stream_t stream);
int ojf_seg_softmax_max(const float *logits_dev, int stride, int n_classes, int npix, float *scores_dev, uint8_t *ids_dev,
                        ojf_stream_t stream);

/* ---- VOLUME HELPERS (Database) -------------------------------------------------------------
 * ojf_volume_fill_*: Database.reset (modules/database.py:351-370).
 * ojf_volume_filter: Database.filter (:108-112): where weights < value: tsdf = init_value, weights = 0.
 * ojf_volume_evaluate: utils/metrics.py:111-127 evaluation() on device: with mask = weights > 0 and
 *   est/gt clipped to +-0.04, sums_dev f64[8] receives {n_mask, sum_sq_err, sum_abs_err,
 *   n_intersection(est<0 & gt<0), n_union(est<0 | gt<0), n_sign_equal, 0, 0}.
 * ojf_volume_confusion: Database.evaluate_semantics (:311-349) -> utils/metrics.py:69-108 semantic_evaluation on
 *   device: with mask = weights > 0 and est' = est * mask, gt' = gt * mask, hist_dev u64[n_classes^2] receives the
 *   confusion counts (row = gt', column = est'; flat index gt' * n_classes + est' as in the reference's bincount) and
 *   present_dev u32[512] the label presence flags of est' ([0, 256)) and gt' ([256, 512)).
 * ojf_volume_median5_u8: Database.filter_semantics (:114-116) = scipy.ndimage.median_filter(ids, size=5):
 *   5x5x5 window, 'reflect' boundary, rank-62 element; out must differ from in. */
int ojf_volume_median5_u8(const uint8_t *in_dev, uint8_t *out_dev, int X, int Y, int Z, ojf_stream_t stream);
int ojf_volume_fill_f16(uint16_t *vol_dev, size_t n, float value, ojf_stream_t stream);
int ojf_volume_fill_u8(uint8_t *vol_dev, size_t n, uint8_t value, ojf_stream_t stream);
int ojf_volume_filter(uint16_t *tsdf_dev, uint16_t *weights_dev, size_t n, float threshold,
                      float init_value, ojf_stream_t stream);
int ojf_volume_evaluate(const uint16_t *est_dev, const uint16_t *gt_dev, const uint16_t *weights_dev,
                        size_t n, double *sums_dev, ojf_stream_t stream);
int ojf_volume_confusion(const uint8_t *ids_est_dev, const uint8_t *ids_gt_dev, const uint16_t *weights_dev, size_t n,
                         int n_classes, unsigned long long *hist_dev, uint32_t *present_dev, ojf_stream_t stream);

/* ---- MESH (Database.get_mesh / save 'ply' and 'test' modes) ------------------------------------
 * ojf_mesh_extract: iso-surface of a fused volume as a triangle list, replacing the host-side
 *   skimage.measure.marching_cubes call in modules/database.py:118-139,203-261 and utils/saving.py:42-47.
 *   Marching tetrahedra on the 6-tetrahedra split of every cell (watertight, no ambiguous cases) - same surface up
 *   to the triangulation, NOT the same triangle list as skimage's Lewiner tables.  Cells with a NaN corner or
 *   (weights_dev != NULL) an unobserved corner are skipped.
 *   workspace_dev: ojf_mesh_workspace_bytes(X,Y,Z) bytes of device memory (per-block counts / offsets).
 *   Call once with capacity = 0 (vertices may be NULL) to get *count_dev = number of triangles, then again with
 *   buffers of capacity triangles: vertices_dev f32[capacity][3][3] = origin + voxel_index * resolution (origin 0
 *   gives the reference's mesh frame); labels_dev u8[capacity][3] (or NULL) = ids_dev at the voxel nearest to each
 *   vertex, ties to even as np.round does (database.py:124-127; 0 when ids_dev is NULL); keys_dev u64[capacity][3]
 *   (or NULL) = 8 * linear index of the lower voxel of the grid edge the vertex lies on + direction code (bit i set:
 *   the edge advances on axis i) - equal keys <=> same vertex, bit-identical position, so an indexed watertight
 *   mesh is one integer sort away.  *count_dev always receives the full number of triangles; those beyond capacity
 *   are dropped.  No atomics: the triangle order is a pure function of the volume. */
size_t ojf_mesh_workspace_bytes(int X, int Y, int Z);
int ojf_mesh_extract(const uint16_t *tsdf_dev, const uint16_t *weights_dev, const uint8_t *ids_dev, int X, int Y,
                     int Z, float iso, const double *origin_host, double resolution, void *workspace_dev,
                     size_t workspace_bytes, float *vertices_dev, uint8_t *labels_dev, uint64_t *keys_dev,
                     uint32_t capacity, uint32_t *count_dev, ojf_stream_t stream);

/* ojf_points_within: the inner loop of the reconstruction F-score (the reference quotes F-scores, README.md:6, but
 *   holds no code for them - SURVEY.md 0.10; definition in metrics.py / mesh.py).  hit_dev[i] (or NULL) = 1 when a
 *   point of the set lies within tau of query i, *n_hit_dev = number of such queries.  The set arrives binned:
 *   points_sorted_dev f64[M][3] ordered by cell, cell = floor((p - grid_origin) / cell_size) per axis on a
 *   GX x GY x GZ grid (z fastest), cell_start_dev u32[GX*GY*GZ + 1] the offsets; cell_size >= tau.  Queries outside
 *   the grid are legal.  Points and distances are f64 and the test is sqrt(dx^2+dy^2+dz^2) <= tau, the comparison
 *   scipy's cKDTree-based host metric makes, so the counts agree with it. */
int ojf_points_within(const double *query_dev, size_t n_query, const double *points_sorted_dev,
                      const uint32_t *cell_start_dev, const double *grid_origin_host, double cell_size, int GX, int GY,
                      int GZ, double tau, uint8_t *hit_dev, uint32_t *n_hit_dev, ojf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OJF_H */
