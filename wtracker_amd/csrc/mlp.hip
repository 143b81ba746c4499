// ResMLP trajectory predictor (wtracker/neural/mlp.py:144-188, eval mode) as small exact-fp32
// MFMA GEMMs.  One wavefront owns 16 samples; every Linear(+folded BatchNorm1d)+ReLU is
// D[16 samples x 16 outs] += X[16 x 4] * W^T[4 x 16] with v_mfma_f32_16x16x4_f32, which is
// bit-for-bit a k-ordered fp32 fma chain, so results stay within the reference's own
// batch-size variance (abs 2e-4 + rel 1e-5).  Activations ping-pong between two
// LDS buffers; the residual stream h lives in a third (h <- h + block(h), mlp.py:185-187).
//
// The optional gather front-end is the batched form of MLPController.provide_movement_vector
// (wtracker/sim/sim_controllers/mlp_controllers.py:38-56): 7 boxes of the track at
// anchor + input_frames, NaN / out-of-range guard, x/y made relative to the first box's corner.
#include "wtk_kernels.h"

namespace wtk {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kLdAct = kMlpMaxDim + 4; // LDS row stride (floats), +4 breaks the power-of-two stride

// y[16][out_pad] = act(W x + b); src/dst are LDS [16][kLdAct]
__device__ __forceinline__ void mlp_layer(const float *__restrict__ params, const MlpLayerDev &L, const float *src, float *dst,
                                          int lane) {
    const int r = lane & 15, g = lane >> 4;
    const float *W = params + L.w_off;
    const float *bvec = params + L.b_off;
    for (int n0 = 0; n0 < L.out_pad; n0 += 16) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *wrow = W + (long long)(n0 + r) * L.in_pad;
        for (int k0 = 0; k0 < L.in_pad; k0 += 4) {
            const float xa = src[r * kLdAct + k0 + g]; // A[row = sample r][k = k0 + g]
            const float wb = wrow[k0 + g];             // B[k = k0 + g][col = out n0 + r]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wb, acc, 0, 0, 0);
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
    const int lane = threadIdx.x;
    const int s0 = blockIdx.x * 16;

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

    const float *params = a.params;
    int li = 0;
    // input layer -> h
    mlp_layer(params, a.layers[li++], bufA, bufH, lane);
    __syncthreads();
    for (int b = 0; b < a.n_blocks; ++b) {
        const float *src = bufH;
        float *dst = bufA;
        for (int l = 0; l < a.layers_per_block; ++l) {
            mlp_layer(params, a.layers[li++], src, dst, lane);
            __syncthreads();
            src = dst;
            dst = (dst == bufA) ? bufB : bufA;
        }
        // h <- h + block(h)
        for (int i = lane; i < 16 * kLdAct; i += 64) bufH[i] += src[i];
        __syncthreads();
    }
    mlp_layer(params, a.layers[li], bufH, bufA, lane);
    __syncthreads();
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
