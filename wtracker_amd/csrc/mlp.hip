// ResMLP trajectory predictor (wtracker/neural/mlp.py:144-188, eval mode) as small exact-fp32
// MFMA GEMMs.  One wavefront owns 16 samples; every Linear(+folded BatchNorm1d)+ReLU is
// D[16 samples x 16 outs] += X[16 x 4] * W^T[4 x 16] with v_mfma_f32_16x16x4_f32, which is
// bit-for-bit a k-ordered fp32 fma chain, so results stay within the reference's own
// batch-size variance (abs 2e-4 + rel 1e-5).  Activations ping-pong between two
// LDS buffers; the residual stream h lives in a third (h <- h + block(h), mlp.py:185-187).
// The whole parameter blob (33 KB / 105 KB for the two shipped predictors) and the layer table are copied
// into LDS first: the kernel is one dependent chain of ~20 tiny GEMMs, and fetching each weight element from
// global memory inside that chain cost ~10 % of a call (59 -> 54 us for the 26-layer predictor; the rest is the
// dependent MFMA / LDS / barrier chain itself: interleaving several output tiles per layer measured slower, 54 -> 70 us).
//
// The optional gather front-end is the batched form of MLPController.provide_movement_vector
// (wtracker/sim/sim_controllers/mlp_controllers.py:38-56): 7 boxes of the track at
// anchor + input_frames, NaN / out-of-range guard, x/y made relative to the first box's corner.
#include "wtk_kernels.h"

#include <type_traits>

namespace wtk {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kLdAct = kMlpMaxDim + 4; // LDS row stride (floats), +4 breaks the power-of-two stride
constexpr int kMlpLdsParams = 32768 - 64; // floats of the parameter blob staged in LDS (the fast path may read 63 floats past the blob)

// y[16][out_pad] = act(W x + b); src/dst are LDS [16][kLdAct]
// FAST (parameters in LDS): all fragments of a 64-deep K slice are fetched first — 32 LDS reads in flight instead of a
// read -> wait -> MFMA round trip per k step — then the MFMA chain runs (same k order: bit-identical results).  Reads past
// in_pad stay inside the activation row / the LDS parameter array and are never multiplied.
template <bool FAST, typename P> // P: const float * into LDS or global memory
__device__ __forceinline__ void mlp_layer(P params, const MlpLayerDev &L, const float *src, float *dst, int lane) {
    const int r = lane & 15, g = lane >> 4;
    P W = params + L.w_off;
    P bvec = params + L.b_off;
    for (int n0 = 0; n0 < L.out_pad; n0 += 16) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        P wrow = W + (n0 + r) * L.in_pad;
        if constexpr (FAST) {
            for (int kb = 0; kb < L.in_pad; kb += 64) {
                float xa[16], wb[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    xa[ks] = src[r * kLdAct + kb + 4 * ks + g];
                    wb[ks] = wrow[kb + 4 * ks + g];
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    if (kb + 4 * ks < L.in_pad) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks], wb[ks], acc, 0, 0, 0);
            }
        } else {
            for (int k0 = 0; k0 < L.in_pad; k0 += 4) {
                const float xa = src[r * kLdAct + k0 + g]; // A[row = sample r][k = k0 + g]
                const float wb = wrow[k0 + g];             // B[k = k0 + g][col = out n0 + r]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wb, acc, 0, 0, 0);
            }
        }
        // lane holds D[row = 4g + i][col = r]: sample 4g+i, out n0+r
        const float b = bvec[n0 + r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[i] + b;
            if (L.relu) v = fmaxf(v, 0.f);
            dst[(4 * g + i) * kLdAct + n0 + r] = v;
        }
    }
}

__global__ __launch_bounds__(64) void mlp_kernel(const MlpArgs a) {
    __shared__ float bufA[16 * kLdAct];
    __shared__ float bufB[16 * kLdAct];
    __shared__ float bufH[16 * kLdAct];
    __shared__ int s_valid[16];
    __shared__ __attribute__((aligned(16))) float wlds[kMlpLdsParams + 64];
    __shared__ MlpLayerDev s_layers[kMlpMaxLayers];
    const int lane = threadIdx.x;
    const int s0 = blockIdx.x * 16;
    const bool in_lds = a.n_params <= kMlpLdsParams; // block uniform
    if (in_lds) {
        const float4 *src4 = reinterpret_cast<const float4 *>(a.params); // hipMalloc'd: 16-byte aligned; blob padded to 4 floats
        float4 *dst4 = reinterpret_cast<float4 *>(wlds);
        for (int i = lane; i < (a.n_params + 3) / 4; i += 64) dst4[i] = src4[i];
    }
    for (int i = lane; i < a.n_layers * (int)(sizeof(MlpLayerDev) / sizeof(int)); i += 64)
        reinterpret_cast<int *>(s_layers)[i] = reinterpret_cast<const int *>(a.layers)[i];

    // ---- stage the 16 input rows (zero-padded) into bufA
    for (int i = lane; i < 16 * kLdAct; i += 64) bufA[i] = 0.f;
    __syncthreads();
    if (a.x) {
        for (int i = lane; i < 16 * a.in_dim; i += 64) {
            const int s = i / a.in_dim, k = i - s * a.in_dim;
            if (s0 + s < a.B) bufA[s * kLdAct + k] = a.x[(long long)(s0 + s) * a.in_dim + k];
        }
    } else {
        if (lane < 16) {
            const int s = s0 + lane;
            int ok = s < a.B;
            float x0 = 0.f, y0 = 0.f;
            if (ok) {
                const int t = a.anchor_frames[s];
                for (int j = 0; j < a.n_in; ++j) {
                    const int f = t + a.input_frames[j];
                    if (f < 0 || f >= a.n_frames) {
                        ok = 0;
                        break;
                    }
                    const float4 b = *reinterpret_cast<const float4 *>(a.track + (long long)f * 4);
                    if (!(isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w))) {
                        ok = 0;
                        break;
                    }
                    if (j == 0) {
                        x0 = b.x;
                        y0 = b.y;
                    }
                    float *d = &bufA[lane * kLdAct + j * 4];
                    d[0] = b.x - x0;
                    d[1] = b.y - y0;
                    d[2] = b.z;
                    d[3] = b.w;
                }
                if (!ok)
                    for (int k = 0; k < a.in_dim; ++k) bufA[lane * kLdAct + k] = 0.f;
            }
            s_valid[lane] = ok;
        }
    }
    __syncthreads();

    auto run = [&](auto fast_tag, auto params) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        int li = 0;
        // input layer -> h
        mlp_layer<FAST>(params, s_layers[li++], bufA, bufH, lane);
        __syncthreads();
        for (int b = 0; b < a.n_blocks; ++b) {
            const float *src = bufH;
            float *dst = bufA;
            for (int l = 0; l < a.layers_per_block; ++l) {
                mlp_layer<FAST>(params, s_layers[li++], src, dst, lane);
                __syncthreads();
                src = dst;
                dst = (dst == bufA) ? bufB : bufA;
            }
            // h <- h + block(h)
            for (int i = lane; i < 16 * kLdAct; i += 64) bufH[i] += src[i];
            __syncthreads();
        }
        mlp_layer<FAST>(params, s_layers[li], bufH, bufA, lane);
        __syncthreads();
    };
    if (in_lds)
        run(std::true_type{}, static_cast<const float *>(wlds));
    else
        run(std::false_type{}, a.params);
    for (int i = lane; i < 16 * a.out_dim; i += 64) {
        const int s = i / a.out_dim, k = i - s * a.out_dim;
        if (s0 + s < a.B) {
            float v = bufA[s * kLdAct + k];
            if (!a.x && !s_valid[s]) v = 0.f;
            a.y[(long long)(s0 + s) * a.out_dim + k] = v;
        }
    }
    if (!a.x && a.valid && lane < 16 && s0 + lane < a.B) a.valid[s0 + lane] = s_valid[lane];
}

hipError_t launch_mlp(const MlpArgs &a, hipStream_t stream) {
    if (a.B <= 0) return hipSuccess;
    if (a.in_dim > kMlpMaxDim || a.out_dim > kMlpMaxDim) return hipErrorInvalidValue;
    if (a.n_layers != 2 + a.n_blocks * a.layers_per_block || a.n_layers > kMlpMaxLayers) return hipErrorInvalidValue;
    if (!a.x && (a.n_in <= 0 || a.n_in > kMlpMaxInputFrames || a.n_in * 4 != a.in_dim)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mlp_kernel, dim3((a.B + 15) / 16), dim3(64), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
