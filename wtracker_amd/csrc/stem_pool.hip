// Stem (preprocess + model.0), letterbox and SPPF pooling kernels.  HBM-bound byte/elementwise
// work: coalesced 16-byte accesses, LDS-resident maps for the chained pools.
#include "wtk_kernels.h"

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// Stem: fuses ultralytics' predictor preprocess (BGR->RGB, HWC uint8 -> float /255)
// with model.0 = Conv(3, c0, k=3, s=2, p=1)+BN+SiLU (a5).  One thread = one output pixel x CO couts.
// Weights [Cout][3][3][3] (O-H-W-I, I = RGB) live in LDS and are read as wave-uniform broadcasts.
// ---------------------------------------------------------------------------------------------
template <typename T, int CO>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
    __shared__ float w_s[CO * 27];
    __shared__ float b_s[CO];
    const int co0 = blockIdx.y * CO;
    for (int i = threadIdx.x; i < CO * 27; i += 256) w_s[i] = a.w[co0 * 27 + i];
    for (int i = threadIdx.x; i < CO; i += 256) b_s[i] = a.bias[co0 + i];
    __syncthreads();

    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    if (m >= M) return;
    const int HoWo = a.Ho * a.Wo;
    const int n = (int)(m / HoWo);
    const int rem = (int)(m - (long long)n * HoWo);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;

    float x[27]; // (kh, kw, rgb)
    const uint8_t *img = a.frames + (long long)n * a.H * a.W * a.C;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            float r = 0.f, g = 0.f, b = 0.f;
            if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
                const uint8_t *p = img + ((long long)hi * a.W + wi) * a.C;
                if (a.C == 1) {
                    r = g = b = (float)p[0] / 255.0f; // gray -> 3 identical channels (yolo_controller.py:68-69)
                } else {
                    b = (float)p[0] / 255.0f;
                    g = (float)p[1] / 255.0f;
                    r = (float)p[2] / 255.0f;
                }
            }
            x[(kh * 3 + kw) * 3 + 0] = r;
            x[(kh * 3 + kw) * 3 + 1] = g;
            x[(kh * 3 + kw) * 3 + 2] = b;
        }
    }
    T *out = reinterpret_cast<T *>(a.out) + m * a.Cout + co0;
#pragma unroll
    for (int c8 = 0; c8 < CO; c8 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
            const float *w = &w_s[(c8 + j) * 27];
#pragma unroll
            for (int k = 0; k < 27; ++k) s = fmaf(w[k], x[k], s);
            s += b_s[c8 + j];
            v[j] = s * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(s * -1.4426950408889634f));
        }
        if constexpr (sizeof(T) == 2) {
            half8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
            *reinterpret_cast<half8 *>(out + c8) = h;
        } else {
            *reinterpret_cast<float4 *>(out + c8) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(out + c8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

hipError_t launch_stem(const StemArgs &a, int is_f16, hipStream_t stream) {
    if (a.Cout % 16 != 0 || (a.C != 1 && a.C != 3)) return hipErrorInvalidValue;
    if (a.Ho != (a.H + 1) / 2 || a.Wo != (a.W + 1) / 2) return hipErrorInvalidValue;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((M + 255) / 256), a.Cout / 16);
    if (is_f16)
        hipLaunchKernelGGL((stem_kernel<_Float16, 16>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((stem_kernel<float, 16>), grid, dim3(256), 0, stream, a);
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
// SPPF chained 5x5 max pools (stride 1, pad 2, implicit -inf padding as nn.MaxPool2d).
// One block = one image x one 16-byte channel group; the whole map lives in LDS and each 5x5
// pool runs as a separable row-max / column-max pair.  y1,y2,y3 are written back to their slices.
// ---------------------------------------------------------------------------------------------
template <typename T, int CE> struct __attribute__((aligned(16))) Vec16 {
    T v[CE];
};

template <typename T, int CE>
__global__ __launch_bounds__(256) void sppf_pool_kernel(const PoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_pool[];
    using V = Vec16<T, CE>;
    V *A = reinterpret_cast<V *>(smem_pool);
    V *Bv = A + a.H * a.W;
    const int groups = a.c / CE;
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int HW = a.H * a.W;
    const int ld = 4 * a.c;
    T *base = reinterpret_cast<T *>(a.buf) + (long long)n * HW * ld + g * CE;
    for (int i = threadIdx.x; i < HW; i += 256) A[i] = *reinterpret_cast<const V *>(base + (long long)i * ld);
    __syncthreads();
    for (int pass = 1; pass <= 3; ++pass) {
        // horizontal 5-max: A -> B
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int y = i / a.W, x = i - y * a.W;
            V m = A[i];
            for (int d = -2; d <= 2; ++d) {
                const int xx = x + d;
                if (d == 0 || xx < 0 || xx >= a.W) continue;
                const V o = A[y * a.W + xx];
#pragma unroll
                for (int e = 0; e < CE; ++e) m.v[e] = o.v[e] > m.v[e] ? o.v[e] : m.v[e];
            }
            Bv[i] = m;
        }
        __syncthreads();
        // vertical 5-max: B -> A, and out
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int y = i / a.W, x = i - y * a.W;
            V m = Bv[i];
            for (int d = -2; d <= 2; ++d) {
                const int yy = y + d;
                if (d == 0 || yy < 0 || yy >= a.H) continue;
                const V o = Bv[yy * a.W + x];
#pragma unroll
                for (int e = 0; e < CE; ++e) m.v[e] = o.v[e] > m.v[e] ? o.v[e] : m.v[e];
            }
            A[i] = m;
            *reinterpret_cast<V *>(base + (long long)i * ld + pass * a.c) = m;
        }
        __syncthreads();
    }
}

static size_t pool_lds_bytes(const PoolArgs &a) { return (size_t)2 * a.H * a.W * 16; }

hipError_t pool_init_attributes() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sppf_pool_kernel<_Float16, 8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&sppf_pool_kernel<float, 4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_sppf_pool(const PoolArgs &a, int is_f16, hipStream_t stream) {
    const int ce = is_f16 ? 8 : 4;
    if (a.c % ce != 0) return hipErrorInvalidValue;
    const size_t lds = pool_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue; // map too large for the LDS-resident form
    const unsigned blocks = (unsigned)(a.N * (a.c / ce));
    if (is_f16)
        hipLaunchKernelGGL((sppf_pool_kernel<_Float16, 8>), dim3(blocks), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((sppf_pool_kernel<float, 4>), dim3(blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

} // namespace wtk
