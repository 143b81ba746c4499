// Detect-head post-processing: class score arg-max over all anchors (ultralytics
// non_max_suppression with max_det = 1 degenerates to a thresholded arg-max),
// DFL decode of the survivor (a6), scale_boxes + clip (a8) and xyxy -> xywh
// (BoxConverter.to_xywh, wtracker/utils/bbox_utils.py:232-253).
//
// One 1024-thread block per image.  Every thread scans a strided share of the anchors keeping
// (best logit, lowest anchor index); a wavefront butterfly over 64 lanes and a 16-wave LDS step
// finish the reduction; only the surviving anchor's 64 DFL logits are decoded.
#include "wtk_kernels.h"

namespace wtk {

template <typename T> __device__ __forceinline__ float ldf(const T *p) { return (float)*p; }

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

    float best = -INFINITY;
    int best_i = 0x7fffffff;
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
        for (int k = 1; k < a.nc; ++k) m = fmaxf(m, ldf(c + k)); // conf = max over classes
        better(best, best_i, m, i);
    }
    // wavefront butterfly (64 lanes)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float s2 = __shfl_xor(best, off, 64);
        const int i2 = __shfl_xor(best_i, off, 64);
        better(best, best_i, s2, i2);
    }
    __shared__ float ws[16];
    __shared__ int wi[16];
    if ((threadIdx.x & 63) == 0) {
        ws[threadIdx.x >> 6] = best;
        wi[threadIdx.x >> 6] = best_i;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    best = ws[0];
    best_i = wi[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) better(best, best_i, ws[w], wi[w]);

    // score = sigmoid(logit) in fp32, candidate iff score > conf (non_max_suppression `xc`)
    const float score = 1.0f / (1.0f + expf(-best));
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
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = ldf(b + k);
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

hipError_t launch_head(const HeadArgs &a, int is_f16, hipStream_t stream) {
    if (a.N <= 0 || a.nc < 1 || a.cls_ld < a.nc) return hipErrorInvalidValue;
    if (is_f16)
        hipLaunchKernelGGL((head_select_kernel<_Float16>), dim3(a.N), dim3(1024), 0, stream, a);
    else
        hipLaunchKernelGGL((head_select_kernel<float>), dim3(a.N), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
