// C ABI of libwtk_hip.so, the look-twice detector (see wtk_internal.h for the file map).
#include "wtk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace wtk;

// =============================================================================================
// Hybrid detector (include/wtk_hip.h, wtk_hybrid_*): a fast handle on every frame + a full-precision handle on the frames whose
// decision margin is below a threshold, composed from the entry points above.  Host code only: every device step is one of the
// library's own launches on the caller's stream.
// =============================================================================================
struct wtk_hybrid {
    wtk_yolo *fast = nullptr, *exact = nullptr;
    int device = 0;
    float margin = 0.f;
    int k = 0, defer = 1;
    // device scratch: rows of the second look, slot list, counters
    int32_t *slots = nullptr, *n_weak = nullptr, *overflow = nullptr, *replaced = nullptr, *anchor = nullptr;
    float *xywh = nullptr, *conf = nullptr;
    int32_t *pos_full = nullptr; // [k][2] view centre that makes a "view" the whole frame (immediate mode gathers weak frames through the views path)
    int pos_h = -1, pos_w = -1;
    int32_t *idx_tmp = nullptr, *pos_tmp = nullptr; // weak rows' (frame, view centre) of wtk_hybrid_predict_views
    // deferred mode: queue of frame copies + output-row addresses
    uint8_t *q_frames = nullptr;
    int q_H = 0, q_W = 0, q_C = 0;
    void **q_ptrs[3] = {nullptr, nullptr, nullptr};
    int32_t *pos_scratch = nullptr;
    int calls = 0;
    float conf_thr = 0.1f;
    bool held = false; // wtk_hybrid_hold: the full-precision handle is the caller's for a while
    std::vector<void *> allocs;
};

static int hybrid_alloc(wtk_hybrid *h, void **p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    h->allocs.push_back(*p);
    HIP_TRY(hipMemset(*p, 0, bytes));
    return 0;
}

extern "C" void wtk_hybrid_destroy(wtk_hybrid *h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    if (h->exact) (void)wtk_yolo_set_dynamic_batch(h->exact, nullptr);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->q_frames) (void)hipFree(h->q_frames);
    delete h;
}

extern "C" int wtk_hybrid_create(wtk_hybrid **out, wtk_yolo *fast, wtk_yolo *exact, float margin, int32_t k, int32_t defer) {
    if (!out || !fast || !exact) return fail("wtk_hybrid_create: null argument");
    if (fast == exact) return fail("wtk_hybrid_create: the fast and the full-precision handle must be two handles");
    if (fast->device != exact->device) return fail("wtk_hybrid_create: both handles must live on the same device");
    if (fast->S_h != exact->S_h || fast->S_w != exact->S_w || fast->anchors != exact->anchors)
        return fail("wtk_hybrid_create: both handles must be the same model at the same network size");
    if (exact->n_dyn) return fail("wtk_hybrid_create: the full-precision handle already takes its batch size from device memory (another hybrid object owns it, or wtk_yolo_set_dynamic_batch was called)");
    if (defer < 1) return fail("wtk_hybrid_create: defer >= 1");
    if (k == 0) k = defer > 1 ? exact->max_batch : std::min(fast->max_batch, exact->max_batch);
    if (k < 1 || k > exact->max_batch) return fail("wtk_hybrid_create: 1 <= k <= max_batch of the full-precision handle");
    if (!(margin >= 0.f)) return fail("wtk_hybrid_create: margin must be a non-negative number");
    DEVICE_GUARD(fast);
    wtk_hybrid *h = new wtk_hybrid();
    h->fast = fast, h->exact = exact, h->device = fast->device, h->margin = margin, h->k = k, h->defer = defer;
    const size_t K = (size_t)k;
    int rc = hybrid_alloc(h, (void **)&h->slots, K * 4) || hybrid_alloc(h, (void **)&h->n_weak, 4) || hybrid_alloc(h, (void **)&h->overflow, 4) ||
             hybrid_alloc(h, (void **)&h->replaced, 4) || hybrid_alloc(h, (void **)&h->xywh, K * 16) || hybrid_alloc(h, (void **)&h->conf, K * 4) ||
             hybrid_alloc(h, (void **)&h->anchor, K * 4) || hybrid_alloc(h, (void **)&h->pos_full, K * 8) || hybrid_alloc(h, (void **)&h->idx_tmp, K * 4) ||
             hybrid_alloc(h, (void **)&h->pos_tmp, K * 8);
    if (!rc && defer > 1) {
        for (int i = 0; i < 3 && !rc; ++i) rc = hybrid_alloc(h, (void **)&h->q_ptrs[i], K * sizeof(void *));
        if (!rc) rc = hybrid_alloc(h, (void **)&h->pos_scratch, (size_t)std::max(fast->max_batch, 1) * 4);
    }
    if (rc) {
        h->exact = nullptr; // nothing set on it yet
        wtk_hybrid_destroy(h);
        return 1;
    }
    if (wtk_yolo_set_dynamic_batch(exact, h->n_weak)) { // the second look costs what the weak rows cost
        h->exact = nullptr;
        wtk_hybrid_destroy(h);
        return 1;
    }
    *out = h;
    return 0;
}

extern "C" int wtk_hybrid_set_margin(wtk_hybrid *h, float margin) {
    if (!h || !(margin >= 0.f)) return fail("wtk_hybrid_set_margin: bad argument");
    h->margin = margin;
    return 0;
}

extern "C" int wtk_hybrid_config(wtk_hybrid *h, int32_t *k, int32_t *defer, float *margin) {
    if (!h) return fail("wtk_hybrid_config: null handle");
    if (k) *k = h->k;
    if (defer) *defer = h->defer;
    if (margin) *margin = h->margin;
    return 0;
}

extern "C" int wtk_hybrid_hold(wtk_hybrid *h, int32_t hold) {
    if (!h) return fail("wtk_hybrid_hold: null handle");
    if (h->defer > 1 && h->calls > 0) return fail("wtk_hybrid_hold: rows are pending (wtk_hybrid_flush first)");
    if (wtk_yolo_set_dynamic_batch(h->exact, hold ? nullptr : h->n_weak)) return 1;
    h->held = hold != 0;
    return 0;
}

extern "C" int wtk_hybrid_pending(wtk_hybrid *h) { return h ? (h->defer > 1 ? h->calls : 0) : -1; }

extern "C" int wtk_hybrid_flush(wtk_hybrid *h, void *stream) {
    if (!h) return fail("wtk_hybrid_flush: null handle");
    if (h->held) return fail("wtk_hybrid_flush: the full-precision handle is held by the caller (wtk_hybrid_hold)");
    DEVICE_GUARD(h); // the recheck launches below go to the CURRENT device
    if (h->defer <= 1 || !h->q_frames || h->calls == 0) return 0;
    // the full-precision handle runs over the queue with its device-side dynamic batch = the queue's length; rows go back to the addresses queued with them
    if (wtk_yolo_predict(h->exact, h->q_frames, h->k, h->q_H, h->q_W, h->q_C, h->conf_thr, 0.7f, 1, h->xywh, h->conf, h->anchor, stream)) return 1;
    if (wtk_recheck_scatter(h->n_weak, h->k, h->xywh, h->conf, h->anchor, h->q_ptrs[0], h->q_ptrs[1], h->q_ptrs[2], h->replaced, stream)) return 1;
    h->calls = 0;
    return 0;
}

// the slot list of the immediate forms: the (up to kk) weakest rows of the fast pass just enqueued
static int hybrid_select(wtk_hybrid *h, int B, int kk, void *stream) {
    return wtk_recheck_select_counted(h->fast->o_margin, B, kk, h->margin, h->slots, h->n_weak, h->overflow, stream);
}

extern "C" int wtk_hybrid_predict(wtk_hybrid *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float *out_xywh,
                                  float *out_conf, int32_t *out_anchor, void *stream) {
    if (!h) return fail("wtk_hybrid_predict: null handle");
    if (h->held) return fail("wtk_hybrid_predict: the full-precision handle is held by the caller (wtk_hybrid_hold)");
    DEVICE_GUARD(h); // the recheck launches below go to the CURRENT device
    const long long fb = (long long)H * W * C;
    if (h->defer > 1) {
        // every precondition of the deferred form is checked BEFORE the fast pass is enqueued: a call that fails must not have
        // written fp16 rows that then never get their second look and are not counted in `overflow` either
        if (!frames_dev || !out_xywh) return fail("wtk_hybrid_predict: null argument");
        if (B <= 0 || B > 1024 || B > h->fast->max_batch) return fail("wtk_hybrid_predict (defer > 1): need 1 <= B <= min(1024, max_batch of the fast handle)");
        if (fb <= 0 || fb % 16 || reinterpret_cast<uintptr_t>(frames_dev) % 16)
            return fail("wtk_hybrid_predict (defer > 1): frames must be 16-byte aligned and a multiple of 16 bytes each");
        if (reinterpret_cast<uintptr_t>(out_xywh) % 16) return fail("wtk_hybrid_predict (defer > 1): xywh rows must be 16-byte aligned");
        // ... including the fast pass's own argument checks: a call it would reject must not fix the queue's frame shape for the object's life
        if ((C != 1 && C != 3) || H <= 0 || W <= 0) return fail("wtk_hybrid_predict: frames must be H x W x 1 (gray) or H x W x 3 (BGR)");
        if (h->q_frames && (h->q_H != H || h->q_W != W || h->q_C != C))
            return fail("wtk_hybrid_predict (defer > 1): every call must bring frames of the same shape");
        if (!h->q_frames) { // the queue's frame copies: allocated at the first call, for its frame shape
            HIP_TRY(hipMalloc((void **)&h->q_frames, (size_t)h->k * (size_t)fb));
            h->q_H = H, h->q_W = W, h->q_C = C;
        }
    }
    if (wtk_yolo_predict(h->fast, frames_dev, B, H, W, C, conf, 0.7f, 1, out_xywh, out_conf, out_anchor, stream)) return 1;
    if (h->defer > 1) {
        if (wtk_recheck_enqueue(h->fast->o_margin, B, h->margin, frames_dev, fb, h->q_frames, h->k, h->n_weak, h->q_ptrs[0], h->q_ptrs[1], h->q_ptrs[2], out_xywh,
                                out_conf, out_anchor, h->pos_scratch, h->overflow, stream))
            return 1;
        ++h->calls;
        h->conf_thr = conf;
        if (h->calls % h->defer == 0) return wtk_hybrid_flush(h, stream);
        return 0;
    }
    const int kk = std::min(h->k, (int)B);
    if (hybrid_select(h, B, kk, stream)) return 1;
    if (h->pos_h != H || h->pos_w != W) {
        // wtk_yolo_predict_views cuts frame[y0 : y0 + view_w, x0 : x0 + view_h] with (x0, y0) = pos - (view_w / 2, view_h / 2)
        // (view_controller.py:158-172): the view (H, W) centred at (H / 2, W / 2) is the frame itself
        std::vector<int32_t> p((size_t)h->k * 2);
        for (int i = 0; i < h->k; ++i) p[2 * i] = H / 2, p[2 * i + 1] = W / 2;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); // a pass enqueued earlier may still read the old centres
        HIP_TRY(hipMemcpy(h->pos_full, p.data(), p.size() * 4, hipMemcpyHostToDevice));
        h->pos_h = H, h->pos_w = W;
    }
    if (wtk_yolo_predict_views(h->exact, frames_dev, B, H, W, C, h->slots, h->pos_full, kk, H, W, conf, 0.7f, 1, h->xywh, h->conf, h->anchor, stream)) return 1;
    return wtk_recheck_merge(h->fast->o_margin, h->slots, B, kk, h->margin, h->xywh, h->conf, h->anchor, out_xywh, out_conf, out_anchor, h->replaced, stream);
}

extern "C" int wtk_hybrid_predict_views(wtk_hybrid *h, const uint8_t *frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t C,
                                        const int32_t *frame_index_dev, const int32_t *pos_xy_dev, int32_t B, int32_t view_w, int32_t view_h, float conf,
                                        float *out_xywh, float *out_conf, int32_t *out_anchor, void *stream) {
    if (!h) return fail("wtk_hybrid_predict_views: null handle");
    if (h->defer > 1) return fail("wtk_hybrid_predict_views: the views entry point has no deferred form (create the object with defer = 1)");
    if (h->held) return fail("wtk_hybrid_predict_views: the full-precision handle is held by the caller (wtk_hybrid_hold)");
    DEVICE_GUARD(h); // the recheck launches below go to the CURRENT device
    if (wtk_yolo_predict_views(h->fast, frames_dev, n_frames, H, W, C, frame_index_dev, pos_xy_dev, B, view_w, view_h, conf, 0.7f, 1, out_xywh, out_conf,
                               out_anchor, stream))
        return 1;
    const int kk = std::min(h->k, (int)B);
    if (hybrid_select(h, B, kk, stream)) return 1;
    HIP_TRY(launch_recheck_gather_views(h->slots, kk, frame_index_dev, pos_xy_dev, h->idx_tmp, h->pos_tmp, (hipStream_t)stream));
    if (wtk_yolo_predict_views(h->exact, frames_dev, n_frames, H, W, C, h->idx_tmp, h->pos_tmp, kk, view_w, view_h, conf, 0.7f, 1, h->xywh, h->conf, h->anchor,
                               stream))
        return 1;
    return wtk_recheck_merge(h->fast->o_margin, h->slots, B, kk, h->margin, h->xywh, h->conf, h->anchor, out_xywh, out_conf, out_anchor, h->replaced, stream);
}

extern "C" int wtk_hybrid_counters(wtk_hybrid *h, int64_t *rows_replaced, int64_t *rows_overflowed) {
    if (!h) return fail("wtk_hybrid_counters: null handle");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    int32_t r = 0, o = 0;
    HIP_TRY(hipMemcpy(&r, h->replaced, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&o, h->overflow, 4, hipMemcpyDeviceToHost));
    if (rows_replaced) *rows_replaced = r;
    if (rows_overflowed) *rows_overflowed = o;
    return 0;
}

