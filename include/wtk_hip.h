/*
 * wtk_hip.h — C ABI of libwtk_hip.so, the MI355X (gfx950) implementation of WTracker's
 * per-frame detection + prediction hot path.
 *
 * The reference (giladfrid009/WTracker) is pure Python and has no FFI; the interfaces this
 * library stands in for are the Python methods cited at each entry point (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes binding a WTracker maintainer
 * would add to call these from the reference's own controllers.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; wtk_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure on the calling thread;
 *   - plain pointers and sizes only; `stream` arguments are a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - a handle owns its weights and workspace (device memory); callers own all I/O buffers;
 *   - a handle is not re-entrant: use one handle per stream / thread;
 *   - "no detection" is not an error: the bbox row is 4 x NaN (yolo_controller.py:84-85).
 */
#ifndef WTK_HIP_H
#define WTK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WTK_ABI_VERSION 7 /* 7: + wtk_yolo_create_planned / wtk_yolo_plan, wtk_yolo_status (additive); 2: + wtk_yolo_predict_views / _nms, wtk_track_*, wtk_comm_*; 3: + WTK_F16X3, wtk_recheck_*; 4: + wtk_recheck_select_counted; 5: + wtk_recheck_enqueue / _scatter; 6: + wtk_hybrid_* (additive: every earlier entry point is unchanged) */

typedef enum wtk_dtype {
    WTK_F32 = 0, /* fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): parity mode   */
    WTK_F16 = 1, /* fp16 storage, fp16 MFMA (v_mfma_f32_16x16x32_f16) with fp32 accumulate */
    WTK_F16X3 = 2 /* split-fp16 storage (hi + lo * 2^-11 pairs), three fp16 MFMAs per product: fp32-grade results (the error of a plain
                     fp32 dot product) from the fp16 matrix pipe.  Channel widths in multiples of 64 (YOLOv8 s / l).                     */
} wtk_dtype;

const char *wtk_last_error(void);
int wtk_abi_version(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
int wtk_device_count(void);

/* ------------------------------------------------------------------------------------------
 * ResMLP trajectory predictor.
 * Replaces: WormPredictor.forward -> RMLP.forward   wtracker/neural/mlp.py:47-48,176-188
 *           (MLPLayer = Linear+BatchNorm1d(eval)+ReLU, mlp.py:67-71; MlpBlock, mlp.py:121-126)
 * ------------------------------------------------------------------------------------------ */
typedef struct wtk_mlp wtk_mlp;

/* One folded affine layer  y = act(W x + b).  BatchNorm1d running statistics are folded into
 * (W, b) by the caller (eval mode, mlp_controllers.py:29).  W is row-major [out_dim][in_dim]. */
typedef struct wtk_mlp_layer {
    int32_t in_dim;
    int32_t out_dim;
    int32_t relu;        /* 1: ReLU after the affine */
    int32_t reserved;
    const float *weight; /* host pointer, [out_dim][in_dim] */
    const float *bias;   /* host pointer, [out_dim]         */
} wtk_mlp_layer;

typedef struct wtk_mlp_desc {
    int32_t device;           /* HIP device ordinal */
    int32_t n_layers;         /* total: 1 (input) + n_blocks*layers_per_block + 1 (output) */
    int32_t n_blocks;         /* residual blocks: h <- h + block(h)   (mlp.py:185-187)      */
    int32_t layers_per_block; /* 4 in both shipped models                                    */
    const wtk_mlp_layer *layers; /* order: input, block0.l0..l3, block1..., output           */
} wtk_mlp_desc;

int wtk_mlp_create(wtk_mlp **out, const wtk_mlp_desc *desc);
void wtk_mlp_destroy(wtk_mlp *h);
/* y[B][out] = model(x[B][in]); x and y are DEVICE pointers, fp32 row-major. */
int wtk_mlp_forward(wtk_mlp *h, const float *x_dev, int32_t batch, float *y_dev, void *stream);
/* Same with HOST pointers (synchronous): what MLPController.provide_movement_vector needs for
 * its single [1,28] sample (mlp_controllers.py:59). */
int wtk_mlp_forward_host(wtk_mlp *h, const float *x_host, int32_t batch, float *y_host);

/* Batched open-loop form of MLPController.provide_movement_vector (mlp_controllers.py:36-68)
 * over a device-resident track: for every sample s with anchor frame t = anchor_frames[s]
 * gather the boxes track[t + input_frames[j]] (j < n_in; out-of-range or non-finite -> the
 * sample is flagged invalid, mlp_controllers.py:43-44), subtract box 0's x/y from every x/y
 * (mlp_controllers.py:50-56), run the model.  pred[s] = raw model output (dx, dy) before the
 * host-side clip/round; valid[s] = 1/0.  track is [n_frames][4] fp32 xywh (absolute coords). */
int wtk_mlp_predict_track(wtk_mlp *h, const float *track_dev, int32_t n_frames,
                          const int32_t *anchor_frames_dev, int32_t n_samples,
                          const int32_t *input_frames_host, int32_t n_in,
                          float *pred_dev /*[n_samples][2]*/, int32_t *valid_dev /*[n_samples]*/,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * YOLOv8 detector (single image size per handle).
 * Replaces: YoloController.predict          wtracker/sim/sim_controllers/yolo_controller.py:64-90
 *           i.e. ultralytics YOLO.predict(source=frames, max_det=1, imgsz=..., conf=...):
 *           letterbox + BGR->RGB + /255, DetectionModel forward, DFL decode,
 *           non_max_suppression, scale_boxes, xyxy -> xywh.
 * ------------------------------------------------------------------------------------------ */
typedef struct wtk_yolo wtk_yolo;

/* One fused conv (Conv2d + folded BatchNorm2d [+ SiLU]).  Weight layout is O-H-W-I:
 * [cout][kh][kw][cin] fp32 (cin fastest).  The conv list order is fixed; see
 * wtk_yolo_conv_count / wtk_yolo_conv_info. */
typedef struct wtk_conv_blob {
    int32_t cout, cin, k, stride;
    int32_t act; /* 1: SiLU, 0: linear (the Detect heads' last 1x1) */
    int32_t reserved;
    const float *weight; /* host pointer */
    const float *bias;   /* host pointer, [cout] */
} wtk_conv_blob;

typedef struct wtk_yolo_desc {
    int32_t device;
    int32_t dtype;      /* wtk_dtype */
    int32_t imgsz_h;    /* network input height (multiple of 32) */
    int32_t imgsz_w;    /* network input width  (multiple of 32) */
    int32_t max_batch;  /* frames per forward pass the workspace is sized for */
    int32_t nc;         /* number of classes, 1 .. 80 (1 for the worm detector, yolo_train_config.yaml:27; 80 = a stock YOLOv8 head) */
    float width_mult;   /* 0.50 for YOLOv8s */
    float depth_mult;   /* 0.33 for YOLOv8s */
    int32_t max_channels; /* 1024 for YOLOv8s */
    int32_t n_convs;    /* must equal wtk_yolo_conv_count(width, depth, maxch, nc) */
    const wtk_conv_blob *convs;
} wtk_yolo_desc;

/* Number / description of the fused convs a model of this scale has, in the order the blob
 * table must follow.  name_out receives e.g. "model.2.m.0.cv1" (ultralytics module path). */
int wtk_yolo_conv_count(float width_mult, float depth_mult, int32_t max_channels, int32_t nc);
int wtk_yolo_conv_info(float width_mult, float depth_mult, int32_t max_channels, int32_t nc,
                       int32_t index, int32_t *cout, int32_t *cin, int32_t *k, int32_t *stride,
                       int32_t *act, char *name_out, size_t name_cap);

int wtk_yolo_create(wtk_yolo **out, const wtk_yolo_desc *desc);
void wtk_yolo_destroy(wtk_yolo *h);
/* The same with the launch plan named by the caller.  A handle is planned ONCE, for one of two regimes:
 *   WTK_PLAN_THROUGHPUT  large batches (BASELINE configs 3-5): window / implicit-GEMM kernels with big tiles, fused Detect tails;
 *   WTK_PLAN_LATENCY     the reference's own operating point — one call of cycle_frame_num (9 / 15) frames and one single-frame
 *                        call per cycle at imgsz 384 (yolo_controller.py:96-98,108-109; initialize_experiment.ipynb: imgsz 384):
 *                        every conv is cut along K as well (split-K implicit GEMM + a slab-combining pass), so that a layer of a
 *                        few thousand pixels still runs on every CU, and the convs of one dependency level of the network (a Detect
 *                        tower's box and class convs, a PAN layer and the tower of the feature map before it) run as ONE launch on
 *                        the caller's stream: 48 dependent launches per YOLOv8s forward, no side streams.  WTK_F32 and WTK_F16X3
 *                        only (WTK_F16 handles stay on the throughput plan).
 *   WTK_PLAN_AUTO        (= wtk_yolo_create) latency when max_batch <= 4 and the dtype allows it, else throughput (measured on MI355X at imgsz 384:
 *                        B = 1 0.51 ms against 0.67-1.00 ms on the throughput plan, B = 15 1.5 ms against 1.1 ms — the cross-over lies near B = 6); the
 *                        environment variable WTK_LATENCY_PLAN=0 / 1 overrides AUTO only.
 * The first eager call of a latency-plan handle at a batch size also TIMES its launch choices (every grouped launch's tile / form candidates, inside the real
 * forward pass: ~0.1 s once; WTK_SK_AUTOTUNE=0 keeps the cost model's); the choice changes no result bit.
 * Launches are eager.  Replaying a captured hipGraph of the forward pass is opt-in (environment WTK_GRAPH=1, read when the handle is created): the
 * throughput plan's capture forks into the library's side streams, and the runtime's handling of such graphs is where round 5's two open problems
 * lived (profiles/r06_notes.md section 1).
 * The plan never changes per call: within a handle a frame's logits do not depend on the batch it arrives in.  Both plans meet the
 * same tolerances against the fp32 restatement; they are not bit-identical to each other (K is summed in a different order). */
typedef enum wtk_plan { WTK_PLAN_AUTO = 0, WTK_PLAN_THROUGHPUT = 1, WTK_PLAN_LATENCY = 2 } wtk_plan;
int wtk_yolo_create_planned(wtk_yolo **out, const wtk_yolo_desc *desc, int32_t plan);
/* The plan a handle runs on: WTK_PLAN_THROUGHPUT or WTK_PLAN_LATENCY (-1 on a null handle). */
int wtk_yolo_plan(wtk_yolo *h);

/* frames: DEVICE pointer, uint8, [B][H][W][C] with C = 1 (gray; replicated to 3 channels as
 * yolo_controller.py:68-69 does) or C = 3 (BGR).  H x W is letterboxed to imgsz_h x imgsz_w
 * (identity when equal).  Outputs (DEVICE pointers, may be NULL except out_xywh):
 *   out_xywh   [B][4] fp32: x_topleft, y_topleft, w, h in input-image pixels; 4 x NaN = none
 *   out_conf   [B]    fp32: score of the kept detection (0 when none)
 *   out_anchor [B]    int32: anchor index of the kept detection (-1 when none)
 * conf/iou/max_det follow non_max_suppression; max_det must be 1 (yolo_controller.py:76). */
int wtk_yolo_predict(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W,
                     int32_t C, float conf, float iou, int32_t max_det, float *out_xywh,
                     float *out_conf, int32_t *out_anchor, void *stream);
/* Same with HOST pointers for frames and outputs (synchronous; PCIe-inclusive). */
int wtk_yolo_predict_host(wtk_yolo *h, const uint8_t *frames_host, int32_t B, int32_t H,
                          int32_t W, int32_t C, float conf, float iou, int32_t max_det,
                          float *out_xywh, float *out_conf, int32_t *out_anchor);

/* Test / profiling hooks: raw head outputs of the last forward pass, copied to HOST as fp32.
 *   level 0..2 -> stride 8/16/32.  box: [B][h*w][64] DFL logits; cls: [B][h*w][nc] logits. */
int wtk_yolo_debug_head(wtk_yolo *h, int32_t level, int32_t B, float *box_host, float *cls_host);
/* Test hook: the output tensor of conv blob `conv_index` (wtk_yolo_conv_info order) after the last forward pass,
 * copied to HOST as fp32 [B][h][w][cout].  shape_hwc (optional) receives {h, w, cout}; pass out_host = NULL to
 * query the shape only.  Intermediates that a fused kernel keeps on chip (model.0 / model.1 when the fused
 * front is active) hold stale data. */
int wtk_yolo_debug_tensor(wtk_yolo *h, int32_t conv_index, int32_t B, float *out_host, size_t out_cap,
                          int32_t *shape_hwc);
/* Run ONLY decode + selection on caller-provided head logits (HOST fp32, same layout as
 * wtk_yolo_debug_head, levels concatenated in anchor order) — isolates the NMS/argmax logic
 * from conv rounding for bit-exact index tests. */
int wtk_yolo_decode_host(wtk_yolo *h, const float *box_host /*[B][A][64]*/,
                         const float *cls_host /*[B][A][nc]*/, int32_t B, int32_t H, int32_t W,
                         float conf, float *out_xywh, float *out_conf, int32_t *out_anchor);
/* Range guard of the fp16-storage modes (WTK_F16, WTK_F16X3: activations and weights live as fp16 values / pairs, |x| < 65 504 in the library's
 * log2(e)-scaled activation domain).  The reference runs a trained, BatchNorm-folded checkpoint in fp32 (yolo_controller.py:42-45,
 * yolo/yolo_train_config.yaml:51) — weights this library has never seen — so both ends are checked:
 *   at wtk_yolo_create   a packed weight beyond the fp16 range is REFUSED (the message names the conv).  Small weights need no guard: below the fp16
 *                        normal range the hi half is a subnormal (spacing 2^-24) and the lo half carries the rest times 2^11, so the pair still
 *                        holds the weight to an ABSOLUTE error of ~2^-36 — far below the rounding of the products it takes part in;
 *   at run time          every max_det = 1 / NMS call reads every class logit: an activation that overflowed anywhere in the network reaches the
 *                        head as inf / NaN in its receptive field, and the head kernels then raise WTK_STATUS_NONFINITE (sticky).
 * wtk_yolo_status: the flags as of the work the caller has synchronised with (no device call: the word lives in pinned host memory);
 * clear != 0 resets the sticky bits.  A WTK_F32 handle reports NONFINITE only for inf / NaN that fp32 itself produces. */
#define WTK_STATUS_NONFINITE 1
int wtk_yolo_status(wtk_yolo *h, int32_t *flags, int32_t clear);
/* Algorithmic work of one forward pass: conv MACs per frame and the number of anchors. */
int wtk_yolo_workload(wtk_yolo *h, double *macs_per_frame, int32_t *anchors);
/* Average device time (ms) per launch of each kernel class over the frames processed since
 * the last reset, measured with HIP events on the caller's stream when profiling is enabled.
 * kernel_class: 0 stem, 1 conv (implicit GEMM), 2 pool, 3 head (decode+select). */
int wtk_yolo_set_profiling(wtk_yolo *h, int32_t enabled);
int wtk_yolo_get_profile(wtk_yolo *h, int32_t kernel_class, double *total_ms, int64_t *launches);
/* The same measurement per kernel (source file) of the forward pass, with the algorithmic FLOPs (2 x MACs) its launches
 * executed: kernel_id 0 stem_mfma_kernel, 1 conv_igemm_kernel + conv1x1_wide_kernel, 2 sppf_pool_kernel, 3 head kernels, 4 conv3x3_halo_kernel
 * (including its fused 1x1 tails), 5 front_fused_kernel + c2f32_fused_kernel, 6 conv3x3_c32_kernel.  Class 1 of
 * wtk_yolo_get_profile is the sum of ids 1, 4, 5 and 6.  (Build-side instrumentation for bench.py's roofline object; the
 * reference has no counterpart.) */
int wtk_yolo_get_kernel_profile(wtk_yolo *h, int32_t kernel_id, double *total_ms, int64_t *launches, double *flops);

/* ------------------------------------------------------------------------------------------
 * Camera / microscope view extraction for a batch of (frame, platform position) pairs.
 * Replaces: ViewController.camera_view / micro_view -> read() (cv.copyMakeBorder REPLICATE by
 *           camera_size//2) + _custom_view slice   wtracker/sim/view_controller.py:45-61,143-181
 * frames [N][H][W][C] uint8, pos_xy [N][2] int32 (x, y), views [N][view_w][view_h][C] — rows = w,
 * cols = h exactly as the reference slices (view_controller.py:171); all DEVICE pointers.
 * ------------------------------------------------------------------------------------------ */
int wtk_crop_views(const uint8_t *frames_dev, int32_t N, int32_t H, int32_t W, int32_t C,
                   const int32_t *pos_xy_dev, int32_t view_w, int32_t view_h, uint8_t *views_dev,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * Decision margin of every frame of the handle's last max_det = 1 call, in class-logit units:
 *   min( best anchor's logit - best logit among all other anchors ,  | best logit - logit(conf) | )
 * i.e. how far the result is from selecting another survivor or from flipping between a detection and a NaN row.  The fp16
 * mode's head logits differ from the fp32 reference's by ~0.01: a frame whose margin is well above that has the fp32
 * survivor; a caller that needs the reference's survivor on EVERY frame re-runs the few frames below its threshold on an fp32
 * handle (wtracker_amd.controllers.YoloConfig.recheck_margin does).  No counterpart in the reference.
 * wtk_yolo_margin_buffer: device pointer of the handle's [max_batch] float buffer (valid once the call's stream work is done);
 * wtk_yolo_last_margins_host: synchronises the device and copies B values.
 * ------------------------------------------------------------------------------------------ */
int wtk_yolo_margin_buffer(wtk_yolo *h, const float **margins_dev);
int wtk_yolo_last_margins_host(wtk_yolo *h, int32_t B, float *margins_host);

/* ------------------------------------------------------------------------------------------
 * Detector with the GENERAL greedy NMS (max_det >= 1 boxes per frame): the part of ultralytics'
 * non_max_suppression(conf, iou, agnostic=False, max_det) that the reference's call site never reaches because it
 * hard-wires max_det = 1 (yolo_controller.py:76; yolo/yolo_train_config.yaml:49-50,61 give iou 0.7, max_det 300, class-aware).
 * Candidates = anchors whose best class score > conf, visited in descending score (lowest anchor index on ties); a
 * candidate is dropped when its IoU with an already kept box of the same class exceeds `iou`.
 * Outputs per frame, rows in descending score: out_xywh [B][max_det][4] (x, y, w, h in input-image pixels, NaN rows past the
 * last box), out_conf [B][max_det], out_cls [B][max_det] (-1 past the last), out_anchor [B][max_det], out_count [B]; all but
 * out_xywh may be NULL.  With max_det = 1 the first row equals wtk_yolo_predict's result.
 * wtk_yolo_decode_nms_host: the same selection on caller-supplied head logits (test hook, as wtk_yolo_decode_host).
 * ------------------------------------------------------------------------------------------ */
int wtk_yolo_predict_nms(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float iou,
                         int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_cls, int32_t *out_anchor,
                         int32_t *out_count, void *stream);
int wtk_yolo_decode_nms_host(wtk_yolo *h, const float *box_host, const float *cls_host, int32_t B, int32_t H, int32_t W,
                             float conf, float iou, int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_cls,
                             int32_t *out_anchor, int32_t *out_count);

/* ------------------------------------------------------------------------------------------
 * The other predictors behind provide_movement_vector and the training-pair builder, batched over a
 * DEVICE-resident track [n_frames][4] xywh (float32: the detector's track; float64: a loaded bboxes.csv), NaN row =
 * missed detection.  One sample per cycle; the result is the predicted ABSOLUTE head position — the caller subtracts the
 * camera centre and rounds in float64 on the host exactly as the reference does (the prediction itself does not depend
 * on the platform position, so all cycles of a track are computed in one launch).
 *
 * wtk_track_median_centers  replaces OptimalController.provide_movement_vector  optimal_controller.py:16-32
 *     pred[i] = per-axis median of the finite box centres of frames [(cycles[i]+1)*cycle_frame_num, +imaging_frame_num)
 * wtk_track_polyfit         replaces PolyfitController.provide_movement_vector  polyfit_controller.py:54-84
 *     weighted least-squares polynomial (numpy.polynomial.polynomial.polyfit semantics: column-scaled, minimum norm)
 *     through the finite centres at frames cycles[i]*cycle_frame_num + sample_times, evaluated at t_eval
 *     (= cycle_frame_num + imaging_frame_num // 2); sample_times / weights as PolyfitConfig.__post_init__ leaves them
 * wtk_track_training_pairs  replaces NumpyDataset.create_from_config            neural/dataset.py:42-96
 *     rows row0 .. row0+n_rows-1: X = boxes at row + input_frames, Y = centres at row + pred_frames, float64 -> float32,
 *     then relative to the float32 corner of the first input box; keep[i] = 0 where the row holds a NaN (caller compacts)
 * valid[i] = 0: no usable frame (the reference returns (0, 0) for that cycle).  pred is float64.
 * ------------------------------------------------------------------------------------------ */
int wtk_track_median_centers(const void *track_dev, int32_t track_is_f64, int32_t n_frames, const int32_t *cycles_dev,
                             int32_t n_samples, int32_t cycle_frame_num, int32_t imaging_frame_num, double *pred_dev,
                             int32_t *valid_dev, void *stream);
int wtk_track_polyfit(const void *track_dev, int32_t track_is_f64, int32_t n_frames, const int32_t *cycles_dev,
                      int32_t n_samples, int32_t cycle_frame_num, const int32_t *sample_times_host,
                      const double *weights_host, int32_t n_times, int32_t degree, double t_eval, double *pred_dev,
                      int32_t *valid_dev, void *stream);
int wtk_track_training_pairs(const void *track_dev, int32_t track_is_f64, int32_t n_frames, int32_t row0, int32_t n_rows,
                             const int32_t *input_frames_host, int32_t n_in, const int32_t *pred_frames_host,
                             int32_t n_out, float *x_dev, float *y_dev, int32_t *keep_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Second look at weak decisions, on the device (extension; host-side form: YoloConfig.recheck_margin).  The fp16 detector
 * reports a decision margin per frame (wtk_yolo_margin_buffer); the K frames of a batch with the smallest margins are
 * detected again by a full-precision handle (WTK_F16X3 / WTK_F32, e.g. through wtk_yolo_predict_views with
 * frame_index = slots) and replace the fast rows where margin < `margin`.  No host synchronisation: K is fixed.
 *   wtk_recheck_select  slots[k] = batch row of the k-th smallest margin (ties: lower row first; NaN = +inf), k < K;
 *                       *n_weak (nullable) = min(K, rows with margin < `margin`): the leading slots that will be merged
 *   wtk_recheck_select_counted  the same, and *n_overflow (nullable) += max(rows with margin < `margin` - K, 0): the weak rows
 *                       the ceiling K cut off.  They keep their fast-pass result WITHOUT a second look, so a caller that
 *                       promises full-precision decisions must either use K = B (no row can be cut off; with the dynamic
 *                       batch below the cost still follows the number of weak rows) or check this counter.
 *   wtk_yolo_set_dynamic_batch  the handle reads the number of batch rows that matter from DEVICE memory at run time
 *                       (nullptr: off): kernels skip the tiles of images beyond it, rows beyond it hold scratch.  With
 *                       n_dev = n_weak the second look costs what the weak frames cost, with no host round trip.
 *   wtk_recheck_merge   row slots[k] of dst_* = row k of src_* where margins[slots[k]] < margin; *n_replaced += count
 * Replaces nothing in the reference (yolo_controller.py:62-90 computes every frame in fp32).
 * ------------------------------------------------------------------------------------------ */
int wtk_recheck_select(const float *margins_dev, int32_t B, int32_t K, float margin, int32_t *slots_dev,
                       int32_t *n_weak_dev, void *stream);
int wtk_recheck_select_counted(const float *margins_dev, int32_t B, int32_t K, float margin, int32_t *slots_dev,
                               int32_t *n_weak_dev, int32_t *n_overflow_dev, void *stream);
int wtk_yolo_set_dynamic_batch(wtk_yolo *h, const int32_t *n_dev);
/* Concurrency inside one forward pass: n = 2 (default) runs the P3 and the P4 Detect tower on a side stream each next to the PAN path,
 * 1 puts both on one side stream, 0 keeps every launch on the caller's stream.  A process that keeps several handles busy at once
 * (two lanes, a second-look handle) can have too many streams in flight; the side streams themselves are ONE pair per process and
 * device, shared by all handles. */
int wtk_yolo_set_side_streams(wtk_yolo *h, int32_t n);
/* ABI v3 symbol of the (removed) block cache of destroyed handles: nothing is cached any more, the call returns 0. */
int wtk_release_cached_memory(void);
int wtk_recheck_merge(const float *margins_dev, const int32_t *slots_dev, int32_t B, int32_t K, float margin,
                      const float *src_xywh, const float *src_conf, const int32_t *src_anchor, float *dst_xywh,
                      float *dst_conf, int32_t *dst_anchor, int32_t *n_replaced_dev, void *stream);
/* Deferred form of the same second look: the weak rows of SEVERAL fast passes share ONE full-precision pass, whose fixed cost
 * (a forward pass of ~60 launches costs ~1.2 ms however few frames are live) is then paid once per D batches.
 *   wtk_recheck_enqueue  every row of the batch with margin < `margin` is appended to a device-side queue: a copy of its frame
 *                        (frames_dev [B][frame_bytes] -> q_frames_dev [q_cap][frame_bytes]) and the addresses of its three
 *                        outputs (dst_xywh + 4 b, dst_conf + b, dst_anchor + b; the last two nullable).  *q_len += rows queued;
 *                        rows that find the queue full keep their fast result and are COUNTED in *n_overflow (nullable) — a
 *                        caller that flushes every D batches with q_cap >= D * B can never overflow.
 *   (the caller then runs a full-precision handle over q_frames_dev with wtk_yolo_set_dynamic_batch(h, q_len_dev))
 *   wtk_recheck_scatter  rows 0 .. *q_len - 1 of src_* go to the queued addresses; *n_replaced += count; *q_len = 0.
 * The output rows named at enqueue time must stay valid until the scatter.  pos_scratch_dev: int32 [B].
 * Replaces nothing in the reference (yolo_controller.py:62-90 computes every frame in fp32). */
int wtk_recheck_enqueue(const float *margins_dev, int32_t B, float margin, const uint8_t *frames_dev, int64_t frame_bytes,
                        uint8_t *q_frames_dev, int32_t q_cap, int32_t *q_len_dev, void **q_xywh_ptrs_dev,
                        void **q_conf_ptrs_dev, void **q_anchor_ptrs_dev, float *dst_xywh, float *dst_conf,
                        int32_t *dst_anchor, int32_t *pos_scratch_dev, int32_t *n_overflow_dev, void *stream);
int wtk_recheck_scatter(int32_t *q_len_dev, int32_t q_cap, const float *src_xywh, const float *src_conf,
                        const int32_t *src_anchor, void **q_xywh_ptrs_dev, void **q_conf_ptrs_dev,
                        void **q_anchor_ptrs_dev, int32_t *n_replaced_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hybrid detector: the two-handle "look twice where fp16 is unsure" scheme above as ONE object behind this boundary, so that a
 * C / C++ host gets the fast mode with full-precision decisions from a single call per batch — what YoloController.predict
 * (yolo_controller.py:62-90, every frame in fp32: yolo/yolo_train_config.yaml:51 `half: False`) returns, at the fp16 handle's
 * speed on the frames whose decision is clear.  `fast` and `exact` are wtk_yolo handles of the same model on the same device
 * (typically WTK_F16 and WTK_F16X3); the object borrows them (the caller destroys them AFTER wtk_hybrid_destroy) and owns the
 * exact handle's dynamic batch while it lives: a full-precision handle serves at most one hybrid object at a time (create refuses
 * a handle whose dynamic batch is already set).
 *   margin  rows whose fast-pass decision margin is below it are looked at again (choose it from measurements on the model:
 *           see wtracker_amd/hybrid.py calibrate(); a fixed number is not a guarantee).
 *   k       ceiling of rows per second look; 0 = the largest value that can never cut a weak row off (the whole batch; the
 *           exact handle's max_batch in deferred mode).  A smaller k COUNTS what it cuts off (wtk_hybrid_counters).
 *   defer   1: every call looks again before it returns its rows to the stream.  D > 1: the weak rows of D consecutive calls
 *           share one full-precision pass (device-side queue of k frame copies); the rows a call named are final only after
 *           the flush that follows — automatically on every D-th call, or wtk_hybrid_flush.  The output buffers named by a
 *           call must stay valid until then; every call must bring frames of the same shape.
 * Everything is enqueued on `stream`; no entry point but _counters synchronises.  One thread per object.
 * ------------------------------------------------------------------------------------------ */
typedef struct wtk_hybrid wtk_hybrid;
int wtk_hybrid_create(wtk_hybrid **out, wtk_yolo *fast, wtk_yolo *exact, float margin, int32_t k, int32_t defer);
void wtk_hybrid_destroy(wtk_hybrid *h);
int wtk_hybrid_set_margin(wtk_hybrid *h, float margin);
/* same arguments and results as wtk_yolo_predict with max_det = 1 */
int wtk_hybrid_predict(wtk_hybrid *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf,
                       float *out_xywh, float *out_conf, int32_t *out_anchor, void *stream);
/* same arguments and results as wtk_yolo_predict_views with max_det = 1; defer must be 1 */
int wtk_hybrid_predict_views(wtk_hybrid *h, const uint8_t *frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t C,
                             const int32_t *frame_index_dev, const int32_t *pos_xy_dev, int32_t B, int32_t view_w,
                             int32_t view_h, float conf, float *out_xywh, float *out_conf, int32_t *out_anchor,
                             void *stream);
int wtk_hybrid_flush(wtk_hybrid *h, void *stream);
/* calls whose weak rows still wait for their full-precision pass (0: every row handed out so far is final); -1 on a null handle */
int wtk_hybrid_pending(wtk_hybrid *h);
/* rows replaced so far / weak rows the ceiling cut off so far (must read 0 for the full-precision claim).  Synchronises the device. */
int wtk_hybrid_counters(wtk_hybrid *h, int64_t *rows_replaced, int64_t *rows_overflowed);
/* hold = 1: the full-precision handle gets its static batch back so that the caller can run it directly on whole batches (calibration of
 * the margin); hold = 0: the object takes it again.  Needs wtk_hybrid_pending() == 0; predict / flush are refused while held. */
int wtk_hybrid_hold(wtk_hybrid *h, int32_t hold);
/* the configuration in effect: ceiling k, defer, margin (each nullable) */
int wtk_hybrid_config(wtk_hybrid *h, int32_t *k, int32_t *defer, float *margin);

/* ------------------------------------------------------------------------------------------
 * Detector on camera views of device-resident full frames: view cropping fused with the letterbox in front of
 * the detector, so the per-frame host crop + upload of the reference's loop disappears.
 * Replaces: YoloController.on_camera_frame -> sim.camera_view()   yolo_controller.py:58-59
 *           ViewController.read (copyMakeBorder REPLICATE) + _custom_view   view_controller.py:45-61,143-172
 *           + the LetterBox resize inside YOLO.predict                    yolo_controller.py:72-78
 * frames [n_frames][H][W][C] uint8 (DEVICE); batch row n is the w x h view of frame frame_index[n] (NULL: frame n)
 * centred on platform position pos_xy[n] = (x, y) — rows = w, cols = h as the reference slices.  Outputs as
 * wtk_yolo_predict, boxes in VIEW pixel coordinates (what predict() on the cropped frames returns).
 * ------------------------------------------------------------------------------------------ */
int wtk_yolo_predict_views(wtk_yolo *h, const uint8_t *frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t C,
                           const int32_t *frame_index_dev, const int32_t *pos_xy_dev, int32_t B,
                           int32_t view_w, int32_t view_h, float conf, float iou, int32_t max_det,
                           float *out_xywh, float *out_conf, int32_t *out_anchor, void *stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: the path's ONLY collective, for callers that do not go through PyTorch.
 * The reference has no distributed code; this replaces nothing — it is what lets frames be sharded over the GPUs of a node:
 * every rank detects its n_local frames of a super-batch, one all-gather (RCCL over xGMI, 16 B per frame: latency-bound)
 * gives every rank the whole [world * n_local][4] block in rank order = frame order (wtracker_amd/pipeline.py shows the
 * sharding; it uses torch.distributed's RCCL communicator for the same ncclAllGather).
 * Rendezvous: rank 0 calls wtk_comm_unique_id and hands the 128 bytes to the other ranks out of band (file, socket, MPI);
 * every rank then calls wtk_comm_create.  librccl is loaded on the first wtk_comm_* call, not with libwtk_hip.so.
 * ------------------------------------------------------------------------------------------ */
#define WTK_COMM_ID_BYTES 128
typedef struct wtk_comm wtk_comm;
int wtk_comm_unique_id(uint8_t *id_out, size_t cap);
int wtk_comm_create(wtk_comm **out, int32_t device, int32_t rank, int32_t world, const uint8_t *id);
void wtk_comm_destroy(wtk_comm *c);
int wtk_allgather_tracks(wtk_comm *c, const float *local_dev, int32_t n_local, float *all_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WTK_HIP_H */
