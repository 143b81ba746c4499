// Implicit-GEMM convolution on CDNA4 matrix cores (gfx950).
//
// Computes, for NHWC activations,  out[p][co] = act( sum_k W[co][k] * im2col(in)[p][k] + b[co] ) (+ res)
// with k = (kh, kw, cin), cin fastest.  This is YOLOv8's fused Conv2d+BatchNorm2d+SiLU
// (ultralytics `Conv`) for every 1x1 / 3x3, stride 1 / 2 layer except the stem.
//
// MI355X mapping
//   * GEMM orientation: MFMA "A" operand = weights (rows = cout), "B" operand = pixels (cols), so each
//     lane's accumulators are a run of CONSECUTIVE output channels of ONE pixel -> NHWC stores are
//     16-byte vectors (the cout<->MFMA-row assignment is permuted so a lane owns 4*TC adjacent couts).
//   * 64-wide waves, 4 waves / block; wave tile 64 px x 64 cout (16 accumulator tiles of 16x16).
//   * K is consumed in 128-byte rows (64 fp16 / 32 fp32): both operand tiles are staged
//     global -> registers -> LDS as 16-byte chunks with an XOR swizzle that makes every
//     ds_read_b128 fragment read conflict-free (bank = (addr/4) % 64, 16-lane groups).
//   * register double-buffering: the global loads of step k+1 are issued before the MFMAs of step k
//     and written to the other LDS buffer after them: one barrier per K step.
//   * fp16 storage uses v_mfma_f32_16x16x32_f16 (fp32 accumulate); fp32 storage uses the exact
//     v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma chain) for the parity mode.
//   * zero padding / ragged tiles are handled by predicated 16-byte loads (no im2col buffer).
//   * pixel tiles are 2-D patches (tile_w x BM/tile_w) on large maps so the 3x3 halo is re-read from
//     L1/L2 rather than HBM, and block ids are remapped so neighbouring tiles share an XCD's L2.
#include "wtk_kernels.h"

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T> struct Elem;
template <> struct Elem<_Float16> {
    static constexpr int CE = 8; // elements per 16-byte chunk
};
template <> struct Elem<float> {
    static constexpr int CE = 4;
};

// SiLU with two transcendentals and three plain VALU ops (v_mul, v_exp, v_add, v_rcp, v_mul).  The obvious
// x / (1 + __expf(-x)) expands to ~35 instructions (IEEE division + range-checked exp) and made the
// epilogue, not the MFMA loop, the longest part of every conv.  v_exp/v_rcp are 1-ulp approximations.
__device__ __forceinline__ float silu_f(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// one 16-byte operand fragment pair -> MFMA(s)
__device__ __forceinline__ void mma_frag(const uint4 &wf, const uint4 &pf, floatx4 &acc, _Float16 *) {
    half8 a = __builtin_bit_cast(half8, wf);
    half8 b = __builtin_bit_cast(half8, pf);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_frag(const uint4 &wf, const uint4 &pf, floatx4 &acc, float *) {
    // lane (r, g) holds k = 4*(g + 4*khalf) + i, i = 0..3; MFMA #i contracts the i-th element of every
    // lane group: k set {i, 4+i, 8+i, 12+i} (+16*khalf) — same k on both operands.
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.x), __builtin_bit_cast(float, pf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.y), __builtin_bit_cast(float, pf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.z), __builtin_bit_cast(float, pf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.w), __builtin_bit_cast(float, pf.w), acc, 0, 0, 0);
}

template <int NV> __device__ __forceinline__ void load_run(const _Float16 *p, float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        half8 h = *reinterpret_cast<const half8 *>(p + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i + j] = (float)h[j];
    }
}
template <int NV> __device__ __forceinline__ void load_run(const float *p, float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        float4 f = *reinterpret_cast<const float4 *>(p + i);
        v[i] = f.x, v[i + 1] = f.y, v[i + 2] = f.z, v[i + 3] = f.w;
    }
}
template <int NV> __device__ __forceinline__ void store_run(_Float16 *p, const float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        half8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[i + j];
        *reinterpret_cast<half8 *>(p + i) = h;
    }
}
template <int NV> __device__ __forceinline__ void store_run(float *p, const float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) *reinterpret_cast<float4 *>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}

template <typename T, int BM, int BN, int WAVES_P, int WAVES_C>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int CE = Elem<T>::CE;
    constexpr int BKE = 8 * CE; // K elements per step = one 128-byte row
    constexpr int WP = BM / WAVES_P, WC = BN / WAVES_C;
    constexpr int TP = WP / 16, TC = WC / 16;
    constexpr int PR = BM / 32, WR = BN / 32; // staging rows per thread
    constexpr int NV = 4 * TC;                // consecutive couts owned by a lane
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(WAVES_P * WAVES_C == 4, "4 waves per block");
    static_assert(TP >= 1 && TC >= 1, "tile too small");

    // Two DISTINCT LDS objects (not one array indexed by `buf`): hipcc's waitcnt pass tracks pending
    // LDS-DMA writes per LDS object, so it can tell that the DMA filling one buffer does not alias the
    // ds_reads of the other and does not drain vmcnt before every fragment read.
    __shared__ __attribute__((aligned(16))) char smem0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char smem1[STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_p = wave / WAVES_C;
    const int wave_c = wave % WAVES_C;

    // ---- XCD-aware bijective remap: consecutive logical tiles share an XCD (and its L2)
    const int nwg = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nct = a.CoutPad / BN;
    const int ptile = L / nct;
    const int n0 = (L % nct) * BN;

    const int HoWo = a.Ho * a.Wo;
    const int tile_h = a.tile_w > 0 ? BM / a.tile_w : 0;
    const int tpi = a.tiles_x * a.tiles_y;

    auto pixel_coords = [&](int p, int &n, int &ho, int &wo) -> bool {
        if (a.tile_w == 0) {
            long long m = (long long)ptile * BM + p;
            if (m >= a.M) return false;
            n = (int)(m / HoWo);
            int rem = (int)(m - (long long)n * HoWo);
            ho = rem / a.Wo;
            wo = rem - ho * a.Wo;
            return true;
        } else {
            n = ptile / tpi;
            int t = ptile - n * tpi;
            int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
            int py = p / a.tile_w, px = p - py * a.tile_w;
            ho = ty * tile_h + py;
            wo = tx * a.tile_w + px;
            return ho < a.Ho && wo < a.Wo;
        }
    };

    // ---- staging assignment: thread -> 16-byte chunk `ch` of rows r0 + 32*i
    const int ch = tid & 7;
    const int r0 = tid >> 3;
    const T *in = reinterpret_cast<const T *>(a.in);
    const T *wgt = reinterpret_cast<const T *>(a.w);

    long long pbase[PR]; // element offset of the (hi0, wi0) input pixel of each staged row
    int phi0[PR], pwi0[PR];
#pragma unroll
    for (int i = 0; i < PR; ++i) {
        int n, ho, wo;
        bool ok = pixel_coords(r0 + 32 * i, n, ho, wo);
        if (ok) {
            phi0[i] = ho * a.stride - a.pad;
            pwi0[i] = wo * a.stride - a.pad;
            pbase[i] = (((long long)n * a.H + phi0[i]) * a.W + pwi0[i]) * a.in_ld + a.in_coff;
        } else {
            phi0[i] = -(1 << 28); // fails every bounds test below
            pwi0[i] = -(1 << 28);
            pbase[i] = 0;
        }
    }
    const T *wrow[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = r0 + 32 * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wrow[i] = wgt + (long long)(n0 + row) * a.Kpad + (ch ^ key) * CE;
    }

    // running (tap, c) of this thread's logical pixel chunk
    int kc = (ch ^ (r0 & 7)) * CE; // channel within the tap
    int tap = 0;
    while (kc >= a.Cin) {
        kc -= a.Cin;
        ++tap;
    }
    const int ntaps = a.KH * a.KW;

    // LDS-DMA staging (global_load_lds_dwordx4): one wave instruction fills 8 rows x 128 B of the LDS
    // image linearly (dest = wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane
    // SOURCE chunk instead: the lane that lands on physical chunk p of row r fetches logical chunk
    // p ^ key(r).  Rows r0 + 32*i of one thread share key(r) for the pixel tile (key = r & 7), so the
    // thread's logical K chunk — and with it its (tap, channel) walk — is the same for all its rows.
    // Out-of-image taps / ragged rows fetch from a zero page instead of being predicated (an inactive
    // lane would leave stale LDS bytes behind).
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);
    auto issue_stage = [&](int ks, char *pt) {
        char *wt = pt + BM * 128;
        int kh = tap / a.KW, kw = tap - kh * a.KW;
        const bool tap_ok = tap < ntaps;
        const long long delta = ((long long)kh * a.W + kw) * a.in_ld + kc;
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const int hi = phi0[i] + kh, wi = pwi0[i] + kw;
            const bool ok = tap_ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const char *src = ok ? reinterpret_cast<const char *>(in + pbase[i] + delta) : zero_page;
            char *dst = pt + (32 * i + 8 * wave) * 128; // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const char *src = reinterpret_cast<const char *>(wrow[i] + (long long)ks * BKE);
            char *dst = wt + (32 * i + 8 * wave) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        kc += BKE;
        while (kc >= a.Cin) {
            kc -= a.Cin;
            ++tap;
        }
    };

    floatx4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int lr = lane & 15, lg = lane >> 4;
    // fragment row addresses (bytes within a tile), constant over K
    int poff[TP], pkey[TP], woff[TC], wkey[TC];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        const int row = wave_p * WP + t * 16 + lr;
        poff[t] = row * 128;
        pkey[t] = row & 7;
    }
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int row = wave_c * WC + (lr >> 2) * NV + t * 4 + (lr & 3);
        woff[t] = row * 128;
        wkey[t] = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
    }

    auto compute_stage = [&](const char *pt) {
        const char *wt = pt + BM * 128;
#pragma unroll
        for (int kh2 = 0; kh2 < 2; ++kh2) {
            const int chunk = lg + 4 * kh2;
            uint4 pf[TP], wf[TC];
#pragma unroll
            for (int t = 0; t < TP; ++t) pf[t] = *reinterpret_cast<const uint4 *>(pt + poff[t] + ((chunk ^ pkey[t]) << 4));
#pragma unroll
            for (int t = 0; t < TC; ++t) wf[t] = *reinterpret_cast<const uint4 *>(wt + woff[t] + ((chunk ^ wkey[t]) << 4));
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_frag(wf[i], pf[j], acc[i][j], (T *)nullptr);
        }
    };

    const int nk = a.Kpad / BKE;
    issue_stage(0, smem0);
    __syncthreads(); // drains the LDS-DMA (vmcnt(0)) and publishes the tile
    for (int ks = 0; ks < nk; ks += 2) {
        // even step: compute smem0 while the DMA fills smem1 (last read before the previous barrier)
        if (ks + 1 < nk) issue_stage(ks + 1, smem1);
        compute_stage(smem0);
        __syncthreads();
        if (ks + 1 >= nk) break;
        if (ks + 2 < nk) issue_stage(ks + 2, smem0);
        compute_stage(smem1);
        __syncthreads();
    }

    // ---- epilogue: lane (pixel lr of tile j, group lg) owns couts cb .. cb+NV-1
    const int cb = n0 + wave_c * WC + lg * NV;
    if (cb + NV > a.Cout) return; // padded output channels (Cout < CoutPad) are never stored
    float bias[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) bias[i] = a.bias[cb + i];
    T *out = reinterpret_cast<T *>(a.out);
    T *out2 = reinterpret_cast<T *>(a.out2);
    const T *res = reinterpret_cast<const T *>(a.res);
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        int n, ho, wo;
        if (!pixel_coords(wave_p * WP + j * 16 + lr, n, ho, wo)) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r] + bias[i * 4 + r];
        if (a.act) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = silu_f(v[i]);
        }
        const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
        if (res) {
            float rv[NV];
            load_run<NV>(res + pix * a.res_ld + a.res_coff + cb, rv);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += rv[i];
        }
        store_run<NV>(out + pix * a.out_ld + a.out_coff + cb, v);
        if (out2) {
            const int Ho2 = a.Ho * 2, Wo2 = a.Wo * 2;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const long long pix2 = ((long long)n * Ho2 + (2 * ho + dy)) * Wo2 + (2 * wo + dx);
                    store_run<NV>(out2 + pix2 * a.out2_ld + a.out2_coff + cb, v);
                }
        }
    }
}

int conv_cfg_bm(int cfg) { return cfg == CFG_128x128 ? 128 : 256; }
int conv_cfg_bn(int cfg) { return cfg == CFG_128x128 ? 128 : (cfg == CFG_256x64 ? 64 : 32); }

hipError_t conv_init_attributes() { return hipSuccess; } // LDS is static: nothing to raise

template <typename T, int BM, int BN, int WAVES_P, int WAVES_C>
static hipError_t launch_t(const ConvArgs &a, hipStream_t stream) {
    long long ptiles;
    if (a.tile_w == 0)
        ptiles = (a.M + BM - 1) / BM;
    else
        ptiles = (long long)a.N * a.tiles_x * a.tiles_y;
    const long long blocks = ptiles * (a.CoutPad / BN);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvArgs &a, int cfg, int is_f16, hipStream_t stream) {
    // host-side shape checks: everything the kernel's indexing assumes
    const int ce = is_f16 ? 8 : 4;
    const int bn = conv_cfg_bn(cfg), bm = conv_cfg_bm(cfg);
    if (a.CoutPad % bn != 0 || a.Cout > a.CoutPad || a.Cout % 8 != 0 || a.Cin % ce != 0 || a.in_ld % ce != 0 || a.in_coff % ce != 0) return hipErrorInvalidValue;
    if (a.out_ld % ce != 0 || a.out_coff % ce != 0 || a.Kpad % (8 * ce) != 0 || a.Kpad < a.K) return hipErrorInvalidValue;
    if (a.K != a.KH * a.KW * a.Cin) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % ce != 0 || a.res_coff % ce != 0)) return hipErrorInvalidValue;
    if (a.out2 && (a.out2_ld % ce != 0 || a.out2_coff % ce != 0)) return hipErrorInvalidValue;
    if (a.tile_w != 0 && (bm % a.tile_w != 0)) return hipErrorInvalidValue;
    if (is_f16) {
        switch (cfg) {
        case CFG_128x128: return launch_t<_Float16, 128, 128, 2, 2>(a, stream);
        case CFG_256x64: return launch_t<_Float16, 256, 64, 4, 1>(a, stream);
        case CFG_256x32: return launch_t<_Float16, 256, 32, 4, 1>(a, stream);
        }
    } else {
        switch (cfg) {
        case CFG_128x128: return launch_t<float, 128, 128, 2, 2>(a, stream);
        case CFG_256x64: return launch_t<float, 256, 64, 4, 1>(a, stream);
        case CFG_256x32: return launch_t<float, 256, 32, 4, 1>(a, stream);
        }
    }
    return hipErrorInvalidValue;
}

} // namespace wtk
