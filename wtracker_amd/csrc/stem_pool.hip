// Stem (preprocess + model.0), letterbox and SPPF pooling kernels.  HBM-bound byte/elementwise
// work: coalesced 16-byte accesses, LDS-resident maps for the chained pools.
#include "wtk_kernels.h"

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// Stem: fuses ultralytics' predictor preprocess (BGR->RGB, HWC uint8 -> float /255)
// with model.0 = Conv(3, c0, k=3, s=2, p=1)+BN+SiLU (a5) on the matrix cores.
//
// One block = one 16x16 tile of output pixels x all c0 couts.  The 33x33 input patch is normalised
// once into LDS as [pixel][R,G,B,0]; K is laid out as k = tap*4 + channel (36 of 64 / 36 slots used),
// so an MFMA B-fragment is two 8-byte LDS reads (two taps) in fp16 mode and one 4-byte read per tap in
// fp32 mode — no im2col buffer.  Weights are pre-packed to the same K layout and live in registers.
// Output lanes own 8 consecutive couts of one pixel (16-byte NHWC stores).
// ---------------------------------------------------------------------------------------------
typedef float floatx4_s __attribute__((ext_vector_type(4)));
typedef _Float16 half4_s __attribute__((ext_vector_type(4)));

constexpr int kStemPatch = 33;

template <typename T> struct StemElem;
template <> struct StemElem<_Float16> {
    using px_t = half4_s; // [R,G,B,0]
};
template <> struct StemElem<float> {
    using px_t = float4;
};

// SPLIT (T = fp16, split-fp16 handles): pixels / 255 and the weights as split pairs (hi = fp16(x), lo = fp16((x - hi) * 2^11), wtk_kernels.h):
// per k-step acc += Wh.Xh; acc1 += Wl.Xh; acc1 += Wh.Xl, value = acc + 2^-11 acc1 — the arithmetic of every other conv of such a handle, a third of
// the matrix time of the fp32 path; split store.  a.w = [Cout][16][4] hi halves followed by [Cout][16][4] lo halves.
template <typename T, int TC, bool SPLIT = false>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemArgs a) {
    using px_t = typename StemElem<T>::px_t;
    static_assert(!SPLIT || (sizeof(T) == 2 && TC % 2 == 0), "split stem: fp16 operands, 32-channel blocks");
    __shared__ __attribute__((aligned(16))) px_t patch[kStemPatch * kStemPatch + 3];
    __shared__ __attribute__((aligned(16))) px_t patch_lo[SPLIT ? kStemPatch * kStemPatch + 3 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int tiles_x = (a.Wo + 15) >> 4, tiles_y = (a.Ho + 15) >> 4;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    if (a.n_dyn && n >= *a.n_dyn) return; // dynamic batch: images beyond the device-side count are not computed (block-uniform)
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;

    // ---- patch: uint8 (gray or BGR) -> RGB/255 in LDS, zero outside the image
    const uint8_t *img = a.frames + (long long)n * a.H * a.W * a.C;
    for (int i = tid; i < kStemPatch * kStemPatch; i += 256) {
        const int py = i / kStemPatch, px = i - py * kStemPatch;
        const int iy = iy0 + py, ix = ix0 + px;
        float r = 0.f, g = 0.f, b = 0.f;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const uint8_t *p = img + ((long long)iy * a.W + ix) * a.C;
            if (a.C == 1) {
                r = g = b = (float)p[0] / 255.0f; // gray -> 3 identical channels (yolo_controller.py:68-69)
            } else {
                b = (float)p[0] / 255.0f, g = (float)p[1] / 255.0f, r = (float)p[2] / 255.0f;
            }
        }
        px_t v;
        v.x = (T)r, v.y = (T)g, v.z = (T)b, v.w = (T)0.f;
        patch[i] = v;
        if constexpr (SPLIT) {
            px_t l;
            l.x = (T)((r - (float)v.x) * kSplitScale), l.y = (T)((g - (float)v.y) * kSplitScale), l.z = (T)((b - (float)v.z) * kSplitScale), l.w = (T)0.f;
            patch_lo[i] = l;
        }
    }

    // ---- weights -> registers.  cout of (tile tc, MFMA row r): (r>>2)*4*TC + tc*4 + (r&3)
    // accumulators start at the bias of the couts this lane owns (lg*4*TC .. +4*TC-1): no add in the epilogue
    floatx4_s acc[TC][4];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int c0 = lg * 4 * TC + i * 4;
        const floatx4_s b4 = (floatx4_s){a.bias[c0], a.bias[c0 + 1], a.bias[c0 + 2], a.bias[c0 + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = b4;
    }
    __syncthreads();

    floatx4_s acc1[SPLIT ? TC : 1][SPLIT ? 4 : 1]; // split mode: the 2^-11 cross terms
    if constexpr (SPLIT) {
        typedef _Float16 half8_s __attribute__((ext_vector_type(8)));
        const _Float16 *w = reinterpret_cast<const _Float16 *>(a.w);
        const _Float16 *wlo = w + (long long)a.Cout * 64;
        half8_s wh[TC][2], wl[TC][2];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co = (lr >> 2) * 4 * TC + i * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                wh[i][ks] = *reinterpret_cast<const half8_s *>(w + (co * 16 + ks * 8 + 2 * lg) * 4);
                wl[i][ks] = *reinterpret_cast<const half8_s *>(wlo + (co * 16 + ks * 8 + 2 * lg) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = (floatx4_s){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = (wave * 4 + j) * 2, px = lr * 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_s ph, pl;
                const int tap0 = ks * 8 + 2 * lg;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tap = tap0 + h;
                    half4_s vh = (half4_s){0, 0, 0, 0}, vl = vh;
                    if (tap < 9) {
                        vh = patch[(py + tap / 3) * kStemPatch + px + tap % 3];
                        vl = patch_lo[(py + tap / 3) * kStemPatch + px + tap % 3];
                    }
                    ph[4 * h + 0] = vh.x, ph[4 * h + 1] = vh.y, ph[4 * h + 2] = vh.z, ph[4 * h + 3] = vh.w;
                    pl[4 * h + 0] = vl.x, pl[4 * h + 1] = vl.y, pl[4 * h + 2] = vl.z, pl[4 * h + 3] = vl.w;
                }
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i][ks], ph, acc[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i][ks], ph, acc1[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i][ks], pl, acc1[i][j], 0, 0, 0);
            }
        }
    } else if constexpr (sizeof(T) == 2) {
        typedef _Float16 half8_s __attribute__((ext_vector_type(8)));
        // packed weights: [cout][16 taps][4] fp16, taps 9..15 zero; lane holds k = 8*lg .. 8*lg+7 = taps 2lg, 2lg+1
        const _Float16 *w = reinterpret_cast<const _Float16 *>(a.w);
        half8_s wf[TC][2];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co = (lr >> 2) * 4 * TC + i * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[i][ks] = *reinterpret_cast<const half8_s *>(w + (co * 16 + ks * 8 + 2 * lg) * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { // pixel tile j = output row 4*wave + j of the block tile, 16 px wide
            const int py = (wave * 4 + j) * 2, px = lr * 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_s pf;
                const int tap0 = ks * 8 + 2 * lg;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tap = tap0 + h;
                    half4_s v = (half4_s){0, 0, 0, 0};
                    if (tap < 9) v = patch[(py + tap / 3) * kStemPatch + px + tap % 3];
                    pf[4 * h + 0] = v.x, pf[4 * h + 1] = v.y, pf[4 * h + 2] = v.z, pf[4 * h + 3] = v.w;
                }
#pragma unroll
                for (int i = 0; i < TC; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i][ks], pf, acc[i][j], 0, 0, 0);
            }
        }
    } else {
        // packed weights: [cout][9 taps][4] fp32; MFMA 16x16x4: lane holds k = lg (channel lg of the tap)
        const float *w = reinterpret_cast<const float *>(a.w);
        float wf[TC][9];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int co = (lr >> 2) * 4 * TC + i * 4 + (lr & 3);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf[i][tap] = w[(co * 9 + tap) * 4 + lg];
        }
        const float *pl = reinterpret_cast<const float *>(patch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = (wave * 4 + j) * 2, px = lr * 2;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float pv = pl[((py + tap / 3) * kStemPatch + px + tap % 3) * 4 + lg];
#pragma unroll
                for (int i = 0; i < TC; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][tap], pv, acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane (pixel lr of row j, group lg) owns couts lg*4*TC .. +4*TC-1
    constexpr int NV = 4 * TC;
    const int cb = lg * NV;
    T *out = reinterpret_cast<T *>(a.out);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + wave * 4 + j, ox = ox0 + lr;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = acc[i][j][r]; // bias already inside
                if constexpr (SPLIT) s = wtk_split_value(s, acc1[i][j][r]);
                v[i * 4 + r] = wtk_silu_scaled(s);
            }
        if constexpr (SPLIT) { // one pixel = 2 * Cout halves
            wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out) + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.Cout * 2, cb, v);
            continue;
        }
        T *o = out + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.Cout + cb;
        if constexpr (sizeof(T) == 4 && NV % 8 == 0) {
            if (a.out_split) { // one pixel = 2 * Cout halves = Cout floats
                wtk_split_store<NV>(reinterpret_cast<_Float16 *>(out + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.Cout), cb, v);
                continue;
            }
        }
        if constexpr (sizeof(T) == 2) {
            // NV = 4*TC halves per lane: 8-byte units (TC odd: 16 / 48 couts) or 16-byte units
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                half4_s h;
                h.x = (_Float16)v[i], h.y = (_Float16)v[i + 1], h.z = (_Float16)v[i + 2], h.w = (_Float16)v[i + 3];
                *reinterpret_cast<half4_s *>(o + i) = h;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; i += 4) *reinterpret_cast<float4 *>(o + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
        }
    }
}

hipError_t launch_stem(const StemArgs &a, int is_f16, hipStream_t stream) {
    if ((a.Cout != 16 && a.Cout != 32 && a.Cout != 48 && a.Cout != 64) || (a.C != 1 && a.C != 3)) return hipErrorInvalidValue;
    if (a.Ho != (a.H + 1) / 2 || a.Wo != (a.W + 1) / 2) return hipErrorInvalidValue;
    if (a.out_split && (is_f16 || a.Cout % 32 != 0)) return hipErrorInvalidValue;
    if (a.in_split && (is_f16 || !a.out_split || (a.Cout != 32 && a.Cout != 64))) return hipErrorInvalidValue;
    const long long blocks = (long long)a.N * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks);
#define WTK_STEM(T, TC) hipLaunchKernelGGL((stem_mfma_kernel<T, TC>), grid, dim3(256), 0, stream, a)
    if (a.in_split) { // split-fp16 handles: split operands on the fp16 matrix instruction
        if (a.Cout == 32)
            hipLaunchKernelGGL((stem_mfma_kernel<_Float16, 2, true>), grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((stem_mfma_kernel<_Float16, 4, true>), grid, dim3(256), 0, stream, a);
    } else if (is_f16) {
        switch (a.Cout / 16) {
        case 1: WTK_STEM(_Float16, 1); break;
        case 2: WTK_STEM(_Float16, 2); break;
        case 3: WTK_STEM(_Float16, 3); break;
        default: WTK_STEM(_Float16, 4); break;
        }
    } else {
        switch (a.Cout / 16) {
        case 1: WTK_STEM(float, 1); break;
        case 2: WTK_STEM(float, 2); break;
        case 3: WTK_STEM(float, 3); break;
        default: WTK_STEM(float, 4); break;
        }
    }
#undef WTK_STEM
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Letterbox = ultralytics LetterBox(new_shape, auto=...) on uint8 HWC frames: cv2.resize
// INTER_LINEAR (fixed point: 11-bit coefficients, rounding shift by 22, half-pixel centres,
// edge clamp) to (new_h, new_w), then constant 114 border.  Identity sizes bypass this kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.N * a.Sh * a.Sw;
    if (idx >= total) return;
    const int n = (int)(idx / ((long long)a.Sh * a.Sw));
    const int rem = (int)(idx - (long long)n * a.Sh * a.Sw);
    const int y = rem / a.Sw, x = rem - y * a.Sw;
    uint8_t *d = a.dst + idx * a.C;
    const int ry = y - a.top, rx = x - a.left;
    if (ry < 0 || ry >= a.new_h || rx < 0 || rx >= a.new_w) {
        for (int c = 0; c < a.C; ++c) d[c] = 114;
        return;
    }
    const uint8_t *s = a.src + (long long)n * a.H * a.W * a.C;
    if (a.new_h == a.H && a.new_w == a.W) {
        for (int c = 0; c < a.C; ++c) d[c] = s[((long long)ry * a.W + rx) * a.C + c];
        return;
    }
    // cv2 INTER_LINEAR: src = (dst + 0.5) * scale - 0.5, scale = src_size / dst_size
    const float sx_scale = (float)a.W / (float)a.new_w, sy_scale = (float)a.H / (float)a.new_h;
    float fx = ((float)rx + 0.5f) * sx_scale - 0.5f;
    float fy = ((float)ry + 0.5f) * sy_scale - 0.5f;
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx;
    fy -= (float)sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= a.W - 1) { sx = a.W - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= a.H - 1) { sy = a.H - 1; fy = 0.f; }
    const int sx1 = min(sx + 1, a.W - 1), sy1 = min(sy + 1, a.H - 1);
    // 11-bit fixed-point coefficients, as cv2's 8-bit path (INTER_RESIZE_COEF_BITS = 11)
    const int ax1 = (int)rintf(fx * 2048.0f), ax0 = 2048 - ax1;
    const int ay1 = (int)rintf(fy * 2048.0f), ay0 = 2048 - ay1;
    for (int c = 0; c < a.C; ++c) {
        const int p00 = s[((long long)sy * a.W + sx) * a.C + c], p01 = s[((long long)sy * a.W + sx1) * a.C + c];
        const int p10 = s[((long long)sy1 * a.W + sx) * a.C + c], p11 = s[((long long)sy1 * a.W + sx1) * a.C + c];
        const int r0 = p00 * ax0 + p01 * ax1, r1 = p10 * ax0 + p11 * ax1;
        // cv2 VResizeLinear<uchar>: ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
        const int v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

hipError_t launch_letterbox(const LetterboxArgs &a, hipStream_t stream) {
    const long long total = (long long)a.N * a.Sh * a.Sw;
    hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Camera / microscope view extraction: ViewController._custom_view on a replicate-padded frame
// (wtracker/sim/view_controller.py:45-61,143-172) for a batch of (frame, platform position) pairs.
// The padding (camera_size//2, BORDER_REPLICATE) cancels against the window offset, so view pixel (r, c)
// is frame pixel (clamp(pos_y - cols//2... see below).  The reference slices `rows = w` and `cols = h`
// (view_controller.py:171); callers pass (rows, cols) explicitly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_views_kernel(const CropArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.N * a.rows * a.cols;
    if (idx >= total) return;
    const int n = (int)(idx / ((long long)a.rows * a.cols));
    const int rem = (int)(idx - (long long)n * a.rows * a.cols);
    const int r = rem / a.cols, c = rem - r * a.cols;
    const int px = a.pos_xy[2 * n], py = a.pos_xy[2 * n + 1];
    // window origin in padded coordinates: pos + pad - size//2; minus pad again -> unpadded, then clamp
    int y = py - a.view_h / 2 + r, x = px - a.view_w / 2 + c;
    y = min(max(y, 0), a.H - 1);
    x = min(max(x, 0), a.W - 1);
    const uint8_t *s = a.frames + (((long long)n * a.H + y) * a.W + x) * a.C;
    uint8_t *d = a.views + idx * a.C;
    for (int k = 0; k < a.C; ++k) d[k] = s[k];
}

hipError_t launch_crop_views(const CropArgs &a, hipStream_t stream) {
    if (a.N <= 0 || a.rows <= 0 || a.cols <= 0 || (a.C != 1 && a.C != 3)) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.rows * a.cols;
    hipLaunchKernelGGL(crop_views_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Camera view + letterbox fused: what ViewController.camera_view (view_controller.py:45-61,143-172)
// followed by ultralytics' LetterBox (cv2.resize INTER_LINEAR + 114 border) produces, straight from the full frame.
// View pixel (r, c) = frame pixel (clamp(pos_y - view_h/2 + r), clamp(pos_x - view_w/2 + c)) as in crop_views_kernel; the
// bilinear arithmetic is letterbox_kernel's (11-bit fixed point), with the view's shape (rows, cols) as the source image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void view_letterbox_kernel(const ViewLetterboxArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.N * a.Sh * a.Sw;
    if (idx >= total) return;
    const int n = (int)(idx / ((long long)a.Sh * a.Sw));
    const int rem = (int)(idx - (long long)n * a.Sh * a.Sw);
    const int y = rem / a.Sw, x = rem - y * a.Sw;
    uint8_t *d = a.dst + idx * a.C;
    const int ry = y - a.top, rx = x - a.left;
    if (ry < 0 || ry >= a.new_h || rx < 0 || rx >= a.new_w) {
        for (int c = 0; c < a.C; ++c) d[c] = 114;
        return;
    }
    const int f = min(max(a.frame_index ? a.frame_index[n] : n, 0), a.F - 1); // a bad index must not become a bad address
    const uint8_t *s = a.frames + (long long)f * a.H * a.W * a.C;
    const int oy = a.pos_xy[2 * n + 1] - a.view_h / 2, ox = a.pos_xy[2 * n] - a.view_w / 2; // view origin in frame coordinates
    auto px = [&](int vr, int vc, int c) -> int { // view pixel through the replicate border
        const int fy = min(max(oy + vr, 0), a.H - 1), fx = min(max(ox + vc, 0), a.W - 1);
        return s[((long long)fy * a.W + fx) * a.C + c];
    };
    if (a.new_h == a.rows && a.new_w == a.cols) {
        for (int c = 0; c < a.C; ++c) d[c] = (uint8_t)px(ry, rx, c);
        return;
    }
    const float sx_scale = (float)a.cols / (float)a.new_w, sy_scale = (float)a.rows / (float)a.new_h;
    float fx = ((float)rx + 0.5f) * sx_scale - 0.5f;
    float fy = ((float)ry + 0.5f) * sy_scale - 0.5f;
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx;
    fy -= (float)sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= a.cols - 1) { sx = a.cols - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= a.rows - 1) { sy = a.rows - 1; fy = 0.f; }
    const int sx1 = min(sx + 1, a.cols - 1), sy1 = min(sy + 1, a.rows - 1);
    const int ax1 = (int)rintf(fx * 2048.0f), ax0 = 2048 - ax1;
    const int ay1 = (int)rintf(fy * 2048.0f), ay0 = 2048 - ay1;
    for (int c = 0; c < a.C; ++c) {
        const int r0 = px(sy, sx, c) * ax0 + px(sy, sx1, c) * ax1, r1 = px(sy1, sx, c) * ax0 + px(sy1, sx1, c) * ax1;
        const int v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

hipError_t launch_view_letterbox(const ViewLetterboxArgs &a, hipStream_t stream) {
    if (a.N <= 0 || a.rows <= 0 || a.cols <= 0 || (a.C != 1 && a.C != 3) || a.Sh <= 0 || a.Sw <= 0) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.Sh * a.Sw;
    hipLaunchKernelGGL(view_letterbox_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SPPF chained 5x5 max pools (stride 1, pad 2, implicit -inf padding as nn.MaxPool2d).
// One block = one image x one 16-byte channel group; the whole map lives in LDS and each 5x5
// pool runs as a separable row-max / column-max pair.  y1,y2,y3 are written back to their slices.
// ---------------------------------------------------------------------------------------------
// 16-byte channel group as a native vector: element-wise max compiles to v_pk_max_f16 (4 per group) / v_max_f32
template <typename T, int CE> struct PoolVec;
template <> struct PoolVec<_Float16, 8> {
    typedef _Float16 type __attribute__((ext_vector_type(8)));
};
template <> struct PoolVec<float, 4> {
    typedef float type __attribute__((ext_vector_type(4)));
};

// SPLIT (T = float): the buffer holds split-fp16 pairs; hi + lo * 2^-11 is exact in fp32, so the maxima are those of the stored values
template <typename T, int CE, int GP, bool SPLIT = false>
__global__ __launch_bounds__(512) void sppf_pool_kernel(const PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_pool[];
    using V = typename PoolVec<T, CE>::type;
    const int HW = a.H * a.W;
    V *A = reinterpret_cast<V *>(smem_pool); // [pixel][GP channel groups]
    V *Bv = A + HW * GP;
    const int groups = a.c / (CE * GP);
    const int n = blockIdx.x / groups, g0 = (blockIdx.x - n * groups) * GP;
    const int ld = 4 * a.c;
    T *base = reinterpret_cast<T *>(a.buf) + (long long)n * HW * ld + g0 * CE;
    const int items = HW * GP; // item = pixel*GP + group: consecutive lanes read consecutive 16-B groups of a pixel
    typedef _Float16 half4_p __attribute__((ext_vector_type(4)));
    auto split_at = [&](int px, int g, int slice) __attribute__((always_inline)) -> _Float16 * { // 4 real channels of a group: hi at the result, lo 32 halves on
        const int c0 = (g0 + g) * CE;
        return reinterpret_cast<_Float16 *>(a.buf) + ((long long)n * HW + px) * (2 * ld) + 2 * slice * a.c + 64 * (c0 >> 5) + (c0 & 31);
    };
    for (int i = threadIdx.x; i < items; i += 512) {
        const int px = i / GP, g = i - px * GP;
        if constexpr (SPLIT) {
            const _Float16 *p = split_at(px, g, 0);
            const half4_p hv = *reinterpret_cast<const half4_p *>(p), lv = *reinterpret_cast<const half4_p *>(p + 32);
            V v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wtk_split_value((float)hv[e], (float)lv[e]);
            A[i] = v;
        } else {
            A[i] = *reinterpret_cast<const V *>(base + (long long)px * ld + g * CE);
        }
    }
    __syncthreads();
    for (int pass = 1; pass <= 3; ++pass) {
        // horizontal 5-max: A -> B   (neighbours in x are +-GP items away; the row ends clip the window)
        for (int i = threadIdx.x; i < items; i += 512) {
            const int px = i / GP;
            const int y = (int)fdiv((unsigned)px, a.d_w), x = px - y * a.W;
            V m = A[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                if (d == 0) continue;
                if ((unsigned)(x + d) < (unsigned)a.W) m = __builtin_elementwise_max(m, A[i + d * GP]);
            }
            Bv[i] = m;
        }
        __syncthreads();
        // vertical 5-max: B -> A, and out
        for (int i = threadIdx.x; i < items; i += 512) {
            const int px = i / GP, g = i - px * GP;
            const int y = (int)fdiv((unsigned)px, a.d_w);
            V m = Bv[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                if (d == 0) continue;
                if ((unsigned)(y + d) < (unsigned)a.H) m = __builtin_elementwise_max(m, Bv[i + d * a.W * GP]);
            }
            A[i] = m;
            if constexpr (SPLIT) {
                _Float16 *p = split_at(px, g, pass);
                half4_p hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 hh = (_Float16)m[e];
                    hv[e] = hh;
                    lv[e] = (_Float16)((m[e] - (float)hh) * kSplitScale);
                }
                *reinterpret_cast<half4_p *>(p) = hv;
                *reinterpret_cast<half4_p *>(p + 32) = lv;
            } else {
                *reinterpret_cast<V *>(base + (long long)px * ld + pass * a.c + g * CE) = m;
            }
        }
        __syncthreads();
    }
}

hipError_t pool_init_attributes() {
    hipError_t e;
#define WTK_POOL_ATTR(T, CE, GP)                                                                                        \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sppf_pool_kernel<T, CE, GP>),                          \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess)                \
        return e;
    WTK_POOL_ATTR(_Float16, 8, 1)
    WTK_POOL_ATTR(_Float16, 8, 4)
    WTK_POOL_ATTR(float, 4, 1)
    WTK_POOL_ATTR(float, 4, 4)
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sppf_pool_kernel<float, 4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sppf_pool_kernel<float, 4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#undef WTK_POOL_ATTR
    return hipSuccess;
}

hipError_t launch_sppf_pool(const PoolArgs &a_in, int is_f16, hipStream_t stream) {
    PoolArgs a = a_in;
    a.d_w = make_fastdiv((unsigned)a.W);
    const int ce = is_f16 ? 8 : 4;
    if (a.c % ce != 0) return hipErrorInvalidValue;
    const size_t per_group = (size_t)2 * a.H * a.W * 16;
    if (per_group > 160 * 1024) return hipErrorInvalidValue; // map too large for the LDS-resident form
    // 4 channel groups per block (64-byte contiguous global accesses) while two blocks still fit a CU
    const int gp = (a.c % (4 * ce) == 0 && 4 * per_group <= 72 * 1024) ? 4 : 1;
    const size_t lds = per_group * gp;
    const unsigned blocks = (unsigned)(a.N * (a.c / (ce * gp)));
    if (is_f16) {
        if (gp == 4)
            hipLaunchKernelGGL((sppf_pool_kernel<_Float16, 8, 4>), dim3(blocks), dim3(512), lds, stream, a);
        else
            hipLaunchKernelGGL((sppf_pool_kernel<_Float16, 8, 1>), dim3(blocks), dim3(512), lds, stream, a);
    } else if (a.split) {
        if (a.c % 32 != 0) return hipErrorInvalidValue;
        if (gp == 4)
            hipLaunchKernelGGL((sppf_pool_kernel<float, 4, 4, true>), dim3(blocks), dim3(512), lds, stream, a);
        else
            hipLaunchKernelGGL((sppf_pool_kernel<float, 4, 1, true>), dim3(blocks), dim3(512), lds, stream, a);
    } else {
        if (gp == 4)
            hipLaunchKernelGGL((sppf_pool_kernel<float, 4, 4>), dim3(blocks), dim3(512), lds, stream, a);
        else
            hipLaunchKernelGGL((sppf_pool_kernel<float, 4, 1>), dim3(blocks), dim3(512), lds, stream, a);
    }
    return hipGetLastError();
}

} // namespace wtk
