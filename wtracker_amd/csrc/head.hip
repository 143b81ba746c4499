// Detect-head post-processing: class score arg-max over all anchors (ultralytics
// non_max_suppression with max_det = 1 degenerates to a thresholded arg-max),
// DFL decode of the survivor (a6), scale_boxes + clip (a8) and xyxy -> xywh
// (BoxConverter.to_xywh, wtracker/utils/bbox_utils.py:232-253).
//
// One 1024-thread block per image.  Every thread scans a strided share of the anchors keeping
// (best fp32 score, lowest anchor index among equal scores); a wavefront butterfly over 64 lanes and a 16-wave LDS step
// finish the reduction; only the surviving anchor's 64 DFL logits are decoded.
#include "wtk_kernels.h"

namespace wtk {

template <typename T> __device__ __forceinline__ float ldf(const T *p) { return (float)*p; }

// Range guard (wtk_yolo_status): the fp16 and f16x3 modes keep activations as fp16 (pairs), so a model whose activations leave the fp16 range
// (|x| >= 65 504 in the log2(e)-scaled domain) produces infinities, which every layer behind them turns into more infinities / NaNs — all the way into
// the head logits of the affected receptive fields.  The arg-max below would silently step over such anchors (a NaN never compares greater), so the
// scan that reads every class logit anyway raises a sticky flag instead: one v_cmp_class per logit here, nothing in any conv epilogue.
constexpr int kStatusNonFinite = 1;
__device__ __forceinline__ bool not_finite(float x) { return !(fabsf(x) <= 3.402823466e+38f); }

// order: larger score wins; equal scores -> lower anchor index wins (stable-sort tie break)
__device__ __forceinline__ void better(float &s, int &i, float s2, int i2) {
    if (s2 > s || (s2 == s && i2 < i)) {
        s = s2;
        i = i2;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void head_select_kernel(const HeadArgs a) {
    const int n = blockIdx.x;
    const int A0 = a.lh[0] * a.lw[0], A1 = a.lh[1] * a.lw[1], A2 = a.lh[2] * a.lw[2];
    const int A = A0 + A1 + A2;

    // The reference sorts fp32 SCORES (sigmoid of the best class logit) in descending order with a stable sort and keeps the first
    // (max_det = 1, yolo_controller.py:72-78): two anchors whose logits differ but whose fp32 sigmoid is equal (every logit above 16.64 gives
    // exactly 1.0f; neighbours a few ulp apart at ordinary logits) resolve to the LOWER anchor index.  So the order here is by score, as in
    // head_nms_kernel; the winner's logit and the best logit among all OTHER anchors are carried along for the decision margin, which stays
    // logit-based (ties with the winner count: margin <= 0).
    float best = -INFINITY, best_l = -INFINITY, second = -INFINITY;
    int best_i = 0x7fffffff;
    bool bad = false;
    auto merge = [](float &b, int &bi, float &bl, float &s2nd, float b2, int bi2, float bl2, float s2) __attribute__((always_inline)) {
        if (b2 > b || (b2 == b && bi2 < bi)) {
            s2nd = fmaxf(fmaxf(bl, s2nd), s2);
            b = b2;
            bi = bi2;
            bl = bl2;
        } else {
            s2nd = fmaxf(fmaxf(bl2, s2nd), s2);
        }
    };
    for (int i = threadIdx.x; i < A; i += 1024) {
        int lvl, j;
        if (i < A0) {
            lvl = 0, j = i;
        } else if (i < A0 + A1) {
            lvl = 1, j = i - A0;
        } else {
            lvl = 2, j = i - A0 - A1;
        }
        const int Al = lvl == 0 ? A0 : (lvl == 1 ? A1 : A2);
        const T *c = reinterpret_cast<const T *>(a.cls[lvl]) + ((long long)n * Al + j) * a.cls_ld;
        float m = ldf(c);
        bad |= not_finite(m);
        for (int k = 1; k < a.nc; ++k) {
            const float v = ldf(c + k);
            bad |= not_finite(v);
            m = fmaxf(m, v); // conf = max over classes
        }
        // score = sigmoid(logit) in fp32, the same expression as head_nms_kernel's
        merge(best, best_i, best_l, second, 1.0f / (1.0f + expf(-m)), i, m, -INFINITY);
    }
    if (bad && a.status) __hip_atomic_fetch_or(a.status, kStatusNonFinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // rare path: only a broken model gets here
    // wavefront butterfly (64 lanes)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float b2 = __shfl_xor(best, off, 64);
        const int i2 = __shfl_xor(best_i, off, 64);
        const float l2 = __shfl_xor(best_l, off, 64);
        const float s2 = __shfl_xor(second, off, 64);
        merge(best, best_i, best_l, second, b2, i2, l2, s2);
    }
    __shared__ float ws[16], wl[16], wsec[16];
    __shared__ int wi[16];
    if ((threadIdx.x & 63) == 0) {
        ws[threadIdx.x >> 6] = best;
        wi[threadIdx.x >> 6] = best_i;
        wl[threadIdx.x >> 6] = best_l;
        wsec[threadIdx.x >> 6] = second;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    best = ws[0];
    best_i = wi[0];
    best_l = wl[0];
    second = wsec[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) merge(best, best_i, best_l, second, ws[w], wi[w], wl[w], wsec[w]);
    if (a.out_margin && threadIdx.x == 0) a.out_margin[n] = fminf(best_l - second, fabsf(best_l - a.conf_logit));

    // candidate iff score > conf (non_max_suppression `xc`)
    const float score = best;
    const bool keep = (best_i != 0x7fffffff) && (score > a.conf);

    // ---- DFL decode of the survivor: lanes 0..3 = sides l, t, r, b
    float dist = 0.f;
    int lvl = 0, j = 0;
    if (keep) {
        if (best_i < A0) {
            lvl = 0, j = best_i;
        } else if (best_i < A0 + A1) {
            lvl = 1, j = best_i - A0;
        } else {
            lvl = 2, j = best_i - A0 - A1;
        }
    }
    const int lane = threadIdx.x;
    if (keep && lane < 4) {
        const int Al = lvl == 0 ? A0 : (lvl == 1 ? A1 : A2);
        const T *b = reinterpret_cast<const T *>(a.box[lvl]) + ((long long)n * Al + j) * 64 + lane * 16;
        float x[16];
        float mx = -INFINITY;
        bool badb = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = ldf(b + k);
            badb |= not_finite(x[k]);
            mx = fmaxf(mx, x[k]);
        }
        if (badb && a.status) __hip_atomic_fetch_or(a.status, kStatusNonFinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the survivor's box logits
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = expf(x[k] - mx);
            sum += x[k];
        }
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) d += (x[k] / sum) * (float)k; // DFL: softmax expectation over 16 bins
        dist = d;
    }
    const float dl = __shfl(dist, 0, 64), dt = __shfl(dist, 1, 64), dr = __shfl(dist, 2, 64), db = __shfl(dist, 3, 64);
    if (lane != 0) return;
    float *o = a.out_xywh + (long long)n * 4;
    if (!keep) {
        const float nanv = __builtin_nanf("");
        o[0] = o[1] = o[2] = o[3] = nanv;
        if (a.out_conf) a.out_conf[n] = 0.f;
        if (a.out_anchor) a.out_anchor[n] = -1;
        return;
    }
    const int w_l = a.lw[lvl];
    const float stride = lvl == 0 ? 8.f : (lvl == 1 ? 16.f : 32.f);
    const float ax = (float)(j % w_l) + 0.5f, ay = (float)(j / w_l) + 0.5f;
    // dist2bbox(xywh=True) * stride, then xywh2xyxy (as Detect / non_max_suppression do)
    const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
    const float cx = (x1 + x2) / 2.f * stride, cy = (y1 + y2) / 2.f * stride;
    const float w = (x2 - x1) * stride, h = (y2 - y1) * stride;
    float bx1 = cx - w / 2.f, by1 = cy - h / 2.f, bx2 = cx + w / 2.f, by2 = cy + h / 2.f;
    // scale_boxes: remove letterbox padding, undo gain, clip to the original image
    bx1 = (bx1 - a.pad_x) / a.gain;
    bx2 = (bx2 - a.pad_x) / a.gain;
    by1 = (by1 - a.pad_y) / a.gain;
    by2 = (by2 - a.pad_y) / a.gain;
    bx1 = fminf(fmaxf(bx1, 0.f), a.img_w);
    bx2 = fminf(fmaxf(bx2, 0.f), a.img_w);
    by1 = fminf(fmaxf(by1, 0.f), a.img_h);
    by2 = fminf(fmaxf(by2, 0.f), a.img_h);
    o[0] = bx1;
    o[1] = by1;
    o[2] = bx2 - bx1;
    o[3] = by2 - by1;
    if (a.out_conf) a.out_conf[n] = score;
    if (a.out_anchor) a.out_anchor[n] = best_i;
}

// ---------------------------------------------------------------------------------------------------------------
// General greedy NMS for max_det > 1 (the reference call site hard-wires max_det = 1, this is the rest of
// ultralytics' non_max_suppression): candidates = anchors whose best class score > conf; boxes = DFL decode in network pixels,
// shifted by class * 7680 (class-aware); repeatedly keep the highest-scoring live candidate (lowest anchor index on ties = the
// stable descending sort + torchvision nms order) and kill every live candidate with IoU > iou against it; stop at max_det.
// One 1024-thread block per image.  Selection is a wave butterfly + 16-wave LDS step per kept box, suppression a strided sweep:
// O(max_det * A / 1024) per thread, a few microseconds per kept box.  (The reference's cap of 30 000 candidates and its
// wall-clock time limit are not reproduced: the cap can only bind at A > 30 000 with nearly every anchor above conf.)
// ---------------------------------------------------------------------------------------------------------------
constexpr float kNmsMaxWh = 7680.f;

template <typename T> __device__ __forceinline__ void decode_anchor(const HeadArgs &a, int n, int lvl, int j, int Al, float (&xyxy)[4]) {
    const T *b = reinterpret_cast<const T *>(a.box[lvl]) + ((long long)n * Al + j) * 64;
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        float x[16];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = ldf(b + side * 16 + k);
            mx = fmaxf(mx, x[k]);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = expf(x[k] - mx);
            sum += x[k];
        }
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) d += (x[k] / sum) * (float)k;
        dist[side] = d;
    }
    const int w_l = a.lw[lvl];
    const float stride = lvl == 0 ? 8.f : (lvl == 1 ? 16.f : 32.f);
    const float ax = (float)(j % w_l) + 0.5f, ay = (float)(j / w_l) + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float cx = (x1 + x2) / 2.f * stride, cy = (y1 + y2) / 2.f * stride;
    const float w = (x2 - x1) * stride, h = (y2 - y1) * stride;
    xyxy[0] = cx - w / 2.f, xyxy[1] = cy - h / 2.f, xyxy[2] = cx + w / 2.f, xyxy[3] = cy + h / 2.f;
}

template <typename T>
__global__ __launch_bounds__(1024) void head_nms_kernel(const NmsArgs q) {
    const HeadArgs &a = q.h;
    const int n = blockIdx.x;
    const int A0 = a.lh[0] * a.lw[0], A1 = a.lh[1] * a.lw[1], A2 = a.lh[2] * a.lw[2];
    const int A = A0 + A1 + A2;
    float *sc = q.scratch_score + (long long)n * A;
    int *cl = q.scratch_cls + (long long)n * A;
    float *bx = q.scratch_box + (long long)n * A * 4;
    // ---- phase 1: scores, best class, boxes of the candidates
    for (int i = threadIdx.x; i < A; i += 1024) {
        int lvl, j;
        if (i < A0) {
            lvl = 0, j = i;
        } else if (i < A0 + A1) {
            lvl = 1, j = i - A0;
        } else {
            lvl = 2, j = i - A0 - A1;
        }
        const int Al = lvl == 0 ? A0 : (lvl == 1 ? A1 : A2);
        const T *c = reinterpret_cast<const T *>(a.cls[lvl]) + ((long long)n * Al + j) * a.cls_ld;
        float m = ldf(c);
        int mk = 0;
        bool bad = not_finite(m);
        for (int k = 1; k < a.nc; ++k) { // first maximum wins, as torch.max
            const float v = ldf(c + k);
            bad |= not_finite(v);
            if (v > m) m = v, mk = k;
        }
        if (bad && a.status) __hip_atomic_fetch_or(a.status, kStatusNonFinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const float score = 1.0f / (1.0f + expf(-m));
        const bool cand = score > a.conf;
        sc[i] = cand ? score : -INFINITY;
        cl[i] = mk;
        if (cand) {
            float b4[4];
            decode_anchor<T>(a, n, lvl, j, Al, b4);
            bx[4 * i + 0] = b4[0], bx[4 * i + 1] = b4[1], bx[4 * i + 2] = b4[2], bx[4 * i + 3] = b4[3];
        }
    }
    __syncthreads();
    __shared__ float ws[16];
    __shared__ int wi[16];
    __shared__ float kept[4];
    __shared__ int kept_i;
    int count = 0;
    for (int det = 0; det < q.max_det; ++det) {
        // ---- next survivor: highest live score, lowest index on ties
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int i = threadIdx.x; i < A; i += 1024) {
            const float s = sc[i];
            if (s > -INFINITY) better(best, best_i, s, i);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float s2 = __shfl_xor(best, off, 64);
            const int i2 = __shfl_xor(best_i, off, 64);
            better(best, best_i, s2, i2);
        }
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = best, wi[threadIdx.x >> 6] = best_i;
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = ws[0];
            int bi = wi[0];
            for (int w = 1; w < 16; ++w) better(b, bi, ws[w], wi[w]);
            kept_i = bi;
            if (bi != 0x7fffffff) {
                const int kc = cl[bi];
                const float x1 = bx[4 * bi], y1 = bx[4 * bi + 1], x2 = bx[4 * bi + 2], y2 = bx[4 * bi + 3];
                const float offc = (float)kc * kNmsMaxWh;
                kept[0] = x1 + offc, kept[1] = y1 + offc, kept[2] = x2 + offc, kept[3] = y2 + offc;
                // output row: scale_boxes (remove letterbox padding, undo gain, clip) + xyxy -> xywh
                float ox1 = (x1 - a.pad_x) / a.gain, ox2 = (x2 - a.pad_x) / a.gain, oy1 = (y1 - a.pad_y) / a.gain, oy2 = (y2 - a.pad_y) / a.gain;
                ox1 = fminf(fmaxf(ox1, 0.f), a.img_w), ox2 = fminf(fmaxf(ox2, 0.f), a.img_w);
                oy1 = fminf(fmaxf(oy1, 0.f), a.img_h), oy2 = fminf(fmaxf(oy2, 0.f), a.img_h);
                const long long r = (long long)n * q.max_det + det;
                q.out_xywh[4 * r] = ox1, q.out_xywh[4 * r + 1] = oy1, q.out_xywh[4 * r + 2] = ox2 - ox1, q.out_xywh[4 * r + 3] = oy2 - oy1;
                if (q.out_conf) q.out_conf[r] = b;
                if (q.out_cls) q.out_cls[r] = kc;
                if (q.out_anchor) q.out_anchor[r] = bi;
                sc[bi] = -INFINITY;
            }
        }
        __syncthreads();
        if (kept_i == 0x7fffffff) break; // uniform: shared
        ++count;
        if (det + 1 == q.max_det) break;
        // ---- suppression sweep: IoU of every live candidate against the kept box, both shifted by their class offset
        const float k0 = kept[0], k1 = kept[1], k2 = kept[2], k3 = kept[3];
        const float ka = (k2 - k0) * (k3 - k1);
        for (int i = threadIdx.x; i < A; i += 1024) {
            if (!(sc[i] > -INFINITY)) continue;
            const float offc = (float)cl[i] * kNmsMaxWh;
            const float b0 = bx[4 * i] + offc, b1 = bx[4 * i + 1] + offc, b2 = bx[4 * i + 2] + offc, b3 = bx[4 * i + 3] + offc;
            const float ix1 = fmaxf(k0, b0), iy1 = fmaxf(k1, b1), ix2 = fminf(k2, b2), iy2 = fminf(k3, b3);
            const float inter = fmaxf(ix2 - ix1, 0.f) * fmaxf(iy2 - iy1, 0.f);
            const float ba = (b2 - b0) * (b3 - b1);
            if (inter / (ka + ba - inter) > q.iou) sc[i] = -INFINITY;
        }
        __syncthreads();
    }
    // ---- rows past the last survivor
    for (int r = count + threadIdx.x; r < q.max_det; r += 1024) {
        const long long o = (long long)n * q.max_det + r;
        const float nanv = __builtin_nanf("");
        q.out_xywh[4 * o] = q.out_xywh[4 * o + 1] = q.out_xywh[4 * o + 2] = q.out_xywh[4 * o + 3] = nanv;
        if (q.out_conf) q.out_conf[o] = 0.f;
        if (q.out_cls) q.out_cls[o] = -1;
        if (q.out_anchor) q.out_anchor[o] = -1;
    }
    if (threadIdx.x == 0 && q.out_count) q.out_count[n] = count;
}

hipError_t launch_head_nms(const NmsArgs &a, int is_f16, hipStream_t stream) {
    if (a.h.N <= 0 || a.h.nc < 1 || a.h.cls_ld < a.h.nc || a.max_det < 1 || !a.scratch_score || !a.scratch_cls || !a.scratch_box || !a.out_xywh)
        return hipErrorInvalidValue;
    if (is_f16)
        hipLaunchKernelGGL((head_nms_kernel<_Float16>), dim3(a.h.N), dim3(1024), 0, stream, a);
    else
        hipLaunchKernelGGL((head_nms_kernel<float>), dim3(a.h.N), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_head(const HeadArgs &a, int is_f16, hipStream_t stream) {
    if (a.N <= 0 || a.nc < 1 || a.cls_ld < a.nc) return hipErrorInvalidValue;
    if (is_f16)
        hipLaunchKernelGGL((head_select_kernel<_Float16>), dim3(a.N), dim3(1024), 0, stream, a);
    else
        hipLaunchKernelGGL((head_select_kernel<float>), dim3(a.N), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
