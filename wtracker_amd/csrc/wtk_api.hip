// C ABI of libwtk_hip.so (see include/wtk_hip.h), part 1: errors and versions, the ResMLP, the batched per-cycle predictors, the second-look
// helpers and the view crop.  The detector handle lives in wtk_plan.hip / wtk_run.hip, the hybrid object in wtk_hybrid.hip (wtk_internal.h).
#include "wtk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

using namespace wtk;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int wtk::fail(const std::string &msg) {
    g_err = msg;
    return 1;
}
int wtk::fail_hip(const char *what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int wtk_set_error(const std::string &msg) { return fail(msg); } // for the other translation units (comm.hip)
extern "C" const char *wtk_last_error(void) { return g_err.c_str(); }
extern "C" int wtk_abi_version(void) { return WTK_ABI_VERSION; }
extern "C" int wtk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}


// Kernel attributes (dynamic LDS above 64 KiB) are per device: initialise them once for every device a handle is created on.
static unsigned long long g_attr_done = 0; // bit d = device d initialised (guarded by g_attr_mu: handles may be created from several host threads)
static std::mutex g_attr_mu;
int wtk::ensure_attributes(int device) {
    if (device < 0 || device >= 64) return fail("device id out of range");
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (g_attr_done & (1ull << device)) return 0;
    HIP_TRY(conv_init_attributes());
    HIP_TRY(pool_init_attributes());
    g_attr_done |= 1ull << device;
    return 0;
}

// =============================================================================================
// ResMLP
// =============================================================================================
struct wtk_mlp {
    int device = 0;
    int n_layers = 0, n_blocks = 0, layers_per_block = 0;
    int in_dim = 0, out_dim = 0;
    float *params = nullptr;
    int n_params = 0;
    MlpLayerDev *layers = nullptr;
    // scratch for the host-pointer entry point
    float *x_dev = nullptr, *y_dev = nullptr;
    int scratch_cap = 0;
};

extern "C" int wtk_mlp_create(wtk_mlp **out, const wtk_mlp_desc *d) {
    if (!out || !d || !d->layers) return fail("wtk_mlp_create: null argument");
    if (d->n_layers != 2 + d->n_blocks * d->layers_per_block) return fail("wtk_mlp_create: n_layers != 2 + n_blocks*layers_per_block");
    if (d->n_layers > kMlpMaxLayers) return fail("wtk_mlp_create: too many layers");
    if (wtk_device_count() <= d->device) return fail("wtk_mlp_create: no such HIP device (is a GPU visible?)");
    DEVICE_GUARD(d);
    std::vector<MlpLayerDev> L(d->n_layers);
    std::vector<float> blob;
    for (int i = 0; i < d->n_layers; ++i) {
        const wtk_mlp_layer &s = d->layers[i];
        if (s.in_dim <= 0 || s.out_dim <= 0 || s.in_dim > kMlpMaxDim || s.out_dim > kMlpMaxDim) return fail("wtk_mlp_create: layer dim out of range");
        if (i > 0) {
            const bool block_first = (i - 1) % std::max(d->layers_per_block, 1) == 0 && i < d->n_layers - 1;
            const int prev_out = (block_first || i == d->n_layers - 1) ? L[0].out_dim : L[i - 1].out_dim;
            if (s.in_dim != prev_out) return fail("wtk_mlp_create: layer dims do not chain");
        }
        MlpLayerDev &l = L[i];
        l.in_dim = s.in_dim;
        l.out_dim = s.out_dim;
        l.in_pad = (s.in_dim + 3) / 4 * 4;
        l.out_pad = (s.out_dim + 15) / 16 * 16;
        l.relu = s.relu;
        l.w_off = (int)blob.size();
        blob.resize(blob.size() + (size_t)l.out_pad * l.in_pad, 0.f);
        for (int o = 0; o < s.out_dim; ++o)
            for (int k = 0; k < s.in_dim; ++k) blob[l.w_off + (size_t)o * l.in_pad + k] = s.weight[(size_t)o * s.in_dim + k];
        l.b_off = (int)blob.size();
        blob.resize(blob.size() + l.out_pad, 0.f);
        for (int o = 0; o < s.out_dim; ++o) blob[l.b_off + o] = s.bias[o];
        l.pad_ = 0;
    }
    // every block must return to the residual width
    for (int b = 0; b < d->n_blocks; ++b)
        if (L[d->layers_per_block * (b + 1)].out_dim != L[0].out_dim) return fail("wtk_mlp_create: block output dim != residual dim");
    blob.resize((blob.size() + 3) / 4 * 4, 0.f); // the kernel copies it to LDS in float4 units
    wtk_mlp *h = new wtk_mlp();
    h->device = d->device;
    h->n_params = (int)blob.size();
    h->n_layers = d->n_layers;
    h->n_blocks = d->n_blocks;
    h->layers_per_block = d->layers_per_block;
    h->in_dim = L[0].in_dim;
    h->out_dim = L.back().out_dim;
    hipError_t e;
    if ((e = hipMalloc(&h->params, blob.size() * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&h->layers, L.size() * sizeof(MlpLayerDev))) != hipSuccess ||
        (e = hipMemcpy(h->params, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->layers, L.data(), L.size() * sizeof(MlpLayerDev), hipMemcpyHostToDevice)) != hipSuccess) {
        wtk_mlp_destroy(h);
        return fail_hip("wtk_mlp_create", e);
    }
    *out = h;
    return 0;
}

extern "C" void wtk_mlp_destroy(wtk_mlp *h) {
    if (!h) return;
    (void)hipFree(h->params);
    (void)hipFree(h->layers);
    (void)hipFree(h->x_dev);
    (void)hipFree(h->y_dev);
    delete h;
}

static MlpArgs mlp_base_args(wtk_mlp *h) {
    MlpArgs a;
    std::memset(&a, 0, sizeof(a));
    a.params = h->params;
    a.n_params = h->n_params;
    a.layers = h->layers;
    a.n_layers = h->n_layers;
    a.n_blocks = h->n_blocks;
    a.layers_per_block = h->layers_per_block;
    a.in_dim = h->in_dim;
    a.out_dim = h->out_dim;
    return a;
}

extern "C" int wtk_mlp_forward(wtk_mlp *h, const float *x_dev, int32_t batch, float *y_dev, void *stream) {
    if (!h || !x_dev || !y_dev) return fail("wtk_mlp_forward: null argument");
    if (batch < 0) return fail("wtk_mlp_forward: negative batch");
    if (batch == 0) return 0;
    DEVICE_GUARD(h);
    MlpArgs a = mlp_base_args(h);
    a.x = x_dev;
    a.y = y_dev;
    a.B = batch;
    HIP_TRY(launch_mlp(a, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_mlp_forward_host(wtk_mlp *h, const float *x_host, int32_t batch, float *y_host) {
    if (!h || !x_host || !y_host) return fail("wtk_mlp_forward_host: null argument");
    if (batch <= 0) return batch == 0 ? 0 : fail("wtk_mlp_forward_host: negative batch");
    DEVICE_GUARD(h);
    if (batch > h->scratch_cap) {
        (void)hipFree(h->x_dev);
        (void)hipFree(h->y_dev);
        h->x_dev = h->y_dev = nullptr;
        h->scratch_cap = 0;
        const int cap = std::max(batch, 256);
        HIP_TRY(hipMalloc(&h->x_dev, (size_t)cap * h->in_dim * sizeof(float)));
        HIP_TRY(hipMalloc(&h->y_dev, (size_t)cap * h->out_dim * sizeof(float)));
        h->scratch_cap = cap;
    }
    HIP_TRY(hipMemcpy(h->x_dev, x_host, (size_t)batch * h->in_dim * sizeof(float), hipMemcpyHostToDevice));
    if (wtk_mlp_forward(h, h->x_dev, batch, h->y_dev, nullptr)) return 1;
    HIP_TRY(hipMemcpy(y_host, h->y_dev, (size_t)batch * h->out_dim * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int wtk_mlp_predict_track(wtk_mlp *h, const float *track_dev, int32_t n_frames, const int32_t *anchor_frames_dev,
                                     int32_t n_samples, const int32_t *input_frames_host, int32_t n_in, float *pred_dev,
                                     int32_t *valid_dev, void *stream) {
    if (!h || !track_dev || !anchor_frames_dev || !input_frames_host || !pred_dev) return fail("wtk_mlp_predict_track: null argument");
    if (n_in <= 0 || n_in > kMlpMaxInputFrames || n_in * 4 != h->in_dim) return fail("wtk_mlp_predict_track: n_in*4 must equal the model's input dim");
    if (n_samples < 0 || n_frames < 0) return fail("wtk_mlp_predict_track: negative size");
    if (n_samples == 0) return 0;
    DEVICE_GUARD(h);
    MlpArgs a = mlp_base_args(h);
    a.x = nullptr;
    a.track = track_dev;
    a.n_frames = n_frames;
    a.anchor_frames = anchor_frames_dev;
    for (int i = 0; i < n_in; ++i) a.input_frames[i] = input_frames_host[i];
    a.n_in = n_in;
    a.valid = valid_dev;
    a.y = pred_dev;
    a.B = n_samples;
    HIP_TRY(launch_mlp(a, (hipStream_t)stream));
    return 0;
}

// =============================================================================================
// Batched per-cycle predictors over a device track
// =============================================================================================
extern "C" int wtk_track_median_centers(const void *track_dev, int32_t track_is_f64, int32_t n_frames, const int32_t *cycles_dev, int32_t n_samples,
                                        int32_t cycle_frame_num, int32_t imaging_frame_num, double *pred_dev, int32_t *valid_dev, void *stream) {
    if (!track_dev || !cycles_dev || !pred_dev || !valid_dev) return fail("wtk_track_median_centers: null argument");
    if (n_samples < 0 || n_frames < 0) return fail("wtk_track_median_centers: negative size");
    if (imaging_frame_num <= 0 || imaging_frame_num > kTrackMaxWindow || cycle_frame_num <= 0)
        return fail("wtk_track_median_centers: imaging_frame_num must be in [1, 64] and cycle_frame_num positive");
    if (n_samples == 0) return 0;
    TrackMedianArgs a;
    a.track = track_dev, a.n_frames = n_frames, a.cycles = cycles_dev, a.n_samples = n_samples;
    a.cycle_frame_num = cycle_frame_num, a.imaging_frame_num = imaging_frame_num;
    a.pred = pred_dev, a.valid = valid_dev;
    HIP_TRY(launch_track_median(a, track_is_f64 != 0, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_track_polyfit(const void *track_dev, int32_t track_is_f64, int32_t n_frames, const int32_t *cycles_dev, int32_t n_samples,
                                 int32_t cycle_frame_num, const int32_t *sample_times_host, const double *weights_host, int32_t n_times, int32_t degree,
                                 double t_eval, double *pred_dev, int32_t *valid_dev, void *stream) {
    if (!track_dev || !cycles_dev || !sample_times_host || !weights_host || !pred_dev || !valid_dev) return fail("wtk_track_polyfit: null argument");
    if (n_samples < 0 || n_frames < 0) return fail("wtk_track_polyfit: negative size");
    if (n_times <= 0 || n_times > kTrackMaxTimes) return fail("wtk_track_polyfit: 1..16 sample times");
    if (degree < 0 || degree + 1 > kTrackMaxCoef) return fail("wtk_track_polyfit: degree must be in [0, 7]");
    if (cycle_frame_num <= 0) return fail("wtk_track_polyfit: cycle_frame_num must be positive");
    if (n_samples == 0) return 0;
    TrackPolyfitArgs a;
    a.track = track_dev, a.n_frames = n_frames, a.cycles = cycles_dev, a.n_samples = n_samples, a.cycle_frame_num = cycle_frame_num;
    for (int i = 0; i < kTrackMaxTimes; ++i) a.times[i] = i < n_times ? sample_times_host[i] : 0, a.weights[i] = i < n_times ? weights_host[i] : 0.0;
    a.n_times = n_times, a.degree = degree, a.t_eval = t_eval;
    a.pred = pred_dev, a.valid = valid_dev;
    HIP_TRY(launch_track_polyfit(a, track_is_f64 != 0, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_track_training_pairs(const void *track_dev, int32_t track_is_f64, int32_t n_frames, int32_t row0, int32_t n_rows,
                                        const int32_t *input_frames_host, int32_t n_in, const int32_t *pred_frames_host, int32_t n_out, float *x_dev,
                                        float *y_dev, int32_t *keep_dev, void *stream) {
    if (!track_dev || !input_frames_host || !pred_frames_host || !x_dev || !y_dev || !keep_dev) return fail("wtk_track_training_pairs: null argument");
    if (n_in <= 0 || n_in > kTrackMaxTimes || n_out <= 0 || n_out > kTrackMaxTimes) return fail("wtk_track_training_pairs: 1..16 input / target frames");
    if (n_rows < 0 || n_frames < 0) return fail("wtk_track_training_pairs: negative size");
    if (n_rows == 0) return 0;
    TrackPairsArgs a;
    a.track = track_dev, a.n_frames = n_frames, a.row0 = row0, a.n_rows = n_rows;
    for (int i = 0; i < kTrackMaxTimes; ++i) a.in_frames[i] = i < n_in ? input_frames_host[i] : 0, a.out_frames[i] = i < n_out ? pred_frames_host[i] : 0;
    a.n_in = n_in, a.n_out = n_out, a.X = x_dev, a.Y = y_dev, a.keep = keep_dev;
    HIP_TRY(launch_track_pairs(a, track_is_f64 != 0, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_recheck_select_counted(const float *margins_dev, int32_t B, int32_t K, float margin, int32_t *slots_dev, int32_t *n_weak_dev,
                                          int32_t *n_overflow_dev, void *stream) {
    if (!margins_dev || !slots_dev) return fail("wtk_recheck_select: null argument");
    if (B <= 0 || B > 1024 || K <= 0 || K > B) return fail("wtk_recheck_select: need 1 <= K <= B <= 1024");
    RecheckArgs a;
    std::memset(&a, 0, sizeof(a));
    a.margins = margins_dev, a.B = B, a.K = K, a.slots = slots_dev, a.thr = margin, a.n_weak = n_weak_dev, a.n_overflow = n_overflow_dev;
    HIP_TRY(launch_recheck_select(a, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_recheck_select(const float *margins_dev, int32_t B, int32_t K, float margin, int32_t *slots_dev, int32_t *n_weak_dev, void *stream) {
    return wtk_recheck_select_counted(margins_dev, B, K, margin, slots_dev, n_weak_dev, nullptr, stream);
}

extern "C" int wtk_recheck_merge(const float *margins_dev, const int32_t *slots_dev, int32_t B, int32_t K, float margin, const float *src_xywh,
                                 const float *src_conf, const int32_t *src_anchor, float *dst_xywh, float *dst_conf, int32_t *dst_anchor,
                                 int32_t *n_replaced_dev, void *stream) {
    if (!margins_dev || !slots_dev || !src_xywh || !dst_xywh) return fail("wtk_recheck_merge: null argument");
    if (B <= 0 || K <= 0 || K > B) return fail("wtk_recheck_merge: need 1 <= K <= B");
    if (reinterpret_cast<uintptr_t>(src_xywh) % 16 || reinterpret_cast<uintptr_t>(dst_xywh) % 16) return fail("wtk_recheck_merge: xywh rows must be 16-byte aligned");
    RecheckArgs a;
    std::memset(&a, 0, sizeof(a));
    a.margins = margins_dev, a.B = B, a.K = K, a.slots = const_cast<int32_t *>(slots_dev), a.thr = margin;
    a.src_xywh = src_xywh, a.src_conf = src_conf, a.src_anchor = src_anchor;
    a.dst_xywh = dst_xywh, a.dst_conf = dst_conf, a.dst_anchor = dst_anchor, a.n_replaced = n_replaced_dev;
    HIP_TRY(launch_recheck_merge(a, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_recheck_enqueue(const float *margins_dev, int32_t B, float margin, const uint8_t *frames_dev, int64_t frame_bytes, uint8_t *q_frames_dev,
                                   int32_t q_cap, int32_t *q_len_dev, void **q_xywh_ptrs_dev, void **q_conf_ptrs_dev, void **q_anchor_ptrs_dev, float *dst_xywh,
                                   float *dst_conf, int32_t *dst_anchor, int32_t *pos_scratch_dev, int32_t *n_overflow_dev, void *stream) {
    if (!margins_dev || !frames_dev || !q_frames_dev || !q_len_dev || !q_xywh_ptrs_dev || !q_conf_ptrs_dev || !q_anchor_ptrs_dev || !dst_xywh || !pos_scratch_dev)
        return fail("wtk_recheck_enqueue: null argument");
    if (B <= 0 || B > 1024 || q_cap <= 0) return fail("wtk_recheck_enqueue: need 1 <= B <= 1024 and a queue of at least one row");
    if (frame_bytes <= 0 || frame_bytes % 16 || reinterpret_cast<uintptr_t>(frames_dev) % 16 || reinterpret_cast<uintptr_t>(q_frames_dev) % 16)
        return fail("wtk_recheck_enqueue: frames must be 16-byte aligned and a multiple of 16 bytes each");
    if (reinterpret_cast<uintptr_t>(dst_xywh) % 16) return fail("wtk_recheck_enqueue: xywh rows must be 16-byte aligned");
    RecheckQueueArgs a;
    std::memset(&a, 0, sizeof(a));
    a.margins = margins_dev, a.B = B, a.thr = margin, a.frames = frames_dev, a.frame_bytes = frame_bytes, a.q_frames = q_frames_dev, a.q_cap = q_cap, a.q_len = q_len_dev;
    a.q_xywh = reinterpret_cast<float **>(q_xywh_ptrs_dev), a.q_conf = reinterpret_cast<float **>(q_conf_ptrs_dev), a.q_anchor = reinterpret_cast<int **>(q_anchor_ptrs_dev);
    a.dst_xywh = dst_xywh, a.dst_conf = dst_conf, a.dst_anchor = dst_anchor, a.pos = pos_scratch_dev, a.n_overflow = n_overflow_dev;
    HIP_TRY(launch_recheck_enqueue(a, (hipStream_t)stream));
    return 0;
}

extern "C" int wtk_recheck_scatter(int32_t *q_len_dev, int32_t q_cap, const float *src_xywh, const float *src_conf, const int32_t *src_anchor, void **q_xywh_ptrs_dev,
                                   void **q_conf_ptrs_dev, void **q_anchor_ptrs_dev, int32_t *n_replaced_dev, void *stream) {
    if (!q_len_dev || !src_xywh || !q_xywh_ptrs_dev || !q_conf_ptrs_dev || !q_anchor_ptrs_dev) return fail("wtk_recheck_scatter: null argument");
    if (q_cap <= 0) return fail("wtk_recheck_scatter: empty queue capacity");
    if (reinterpret_cast<uintptr_t>(src_xywh) % 16) return fail("wtk_recheck_scatter: xywh rows must be 16-byte aligned");
    RecheckQueueArgs a;
    std::memset(&a, 0, sizeof(a));
    a.q_len = q_len_dev, a.q_cap = q_cap, a.src_xywh = src_xywh, a.src_conf = src_conf, a.src_anchor = src_anchor;
    a.q_xywh = reinterpret_cast<float **>(q_xywh_ptrs_dev), a.q_conf = reinterpret_cast<float **>(q_conf_ptrs_dev), a.q_anchor = reinterpret_cast<int **>(q_anchor_ptrs_dev);
    a.n_replaced = n_replaced_dev;
    HIP_TRY(launch_recheck_scatter(a, (hipStream_t)stream));
    return 0;
}

// =============================================================================================
// View extraction
// =============================================================================================
extern "C" int wtk_crop_views(const uint8_t *frames_dev, int32_t N, int32_t H, int32_t W, int32_t C, const int32_t *pos_xy_dev,
                              int32_t view_w, int32_t view_h, uint8_t *views_dev, void *stream) {
    if (!frames_dev || !pos_xy_dev || !views_dev) return fail("wtk_crop_views: null argument");
    if (N <= 0 || H <= 0 || W <= 0 || view_w <= 0 || view_h <= 0 || (C != 1 && C != 3)) return fail("wtk_crop_views: bad shape");
    CropArgs a;
    a.frames = frames_dev;
    a.pos_xy = pos_xy_dev;
    a.views = views_dev;
    a.N = N, a.H = H, a.W = W, a.C = C;
    a.view_w = view_w, a.view_h = view_h;
    a.rows = view_w, a.cols = view_h; // frame[y : y + w, x : x + h], view_controller.py:171
    HIP_TRY(launch_crop_views(a, (hipStream_t)stream));
    return 0;
}

