// 3x3 / stride-1 convolution for thin layers (fp16, Cin = 32, Cout <= 96) — YOLOv8s' first C2f
// bottleneck (model.2.m.0.cv1/cv2: 32 -> 32 channels on the 160x160 map, 64 frames = 1.6 M pixels).
//
// These layers are HBM / latency bound (K = 288, N = 32: 18 MFMA-cycles per pixel row), so the design
// removes every per-tap synchronisation instead of chasing MFMA rate:
//   * same flat-strip window as conv3x3_halo.hip, but a window row is 64 bytes (32 fp16 channels) and
//     ALL nine taps' weights ([9][Cout][64 B] <= 55 KB) are staged once next to it -> one barrier per
//     block, then 9 x (ds_read_b128 + MFMA) with no further waits;
//   * 46-83 KB of LDS and 256 threads per block -> 2-3 blocks per CU hide the staging latency;
//   * 64-byte rows need their own XOR swizzles (bank = (addr/4) % 64 covers FOUR rows): pixel rows use
//     key = (row >> 1) & 3, weight rows key = (((row / NV) & 1) << 1) | ((row >> 1) & 1); both were
//     checked conflict-free for every ds_read_b128 lane group and every tap offset by enumeration.
#include "wtk_kernels.h"

namespace wtk {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 256;
constexpr int kRowsMax = kHaloRowsMax; // same strip geometry as the 128-byte-row kernel

__device__ __forceinline__ float silu_c(float x) {
    return wtk_silu_scaled(x); // x is the log2(e)-scaled pre-activation (wtk_kernels.h)
}

template <int BN> // BN = CoutPad: 32, 64 or 96; 4 waves, each 64 px x BN cout
__global__ __launch_bounds__(256) void conv3x3_c32_kernel(const HaloArgs a) {
    constexpr int TC = BN / 16, TP = 4, NV = 4 * TC;
    constexpr int WROWS = 9 * BN; // weight rows (tap, cout), 64 B each
    __shared__ __attribute__((aligned(16))) char win[kRowsMax * 64];
    __shared__ __attribute__((aligned(16))) char wts[WROWS * 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;

    unsigned t = blockIdx.x;
    unsigned q = fdiv(t, a.d_bps);
    const int o0 = (int)(t - q * (unsigned)a.blocks_per_strip) * kBM;
    t = q;
    q = fdiv(t, a.d_strips);
    const int xs = (int)(t - q * (unsigned)a.strips) * a.S;
    const int n = (int)q;
    const int pitch = a.pitch;
    const int halo_rows = kBM + 2 * pitch + 2;
    const int halo_pieces = (halo_rows + 15) >> 4; // 16 rows of 64 B per 1-KiB LDS-DMA piece

    const _Float16 *in = reinterpret_cast<const _Float16 *>(a.in) + (long long)n * a.H * a.W * a.in_ld + a.in_coff;
    const _Float16 *wgt = reinterpret_cast<const _Float16 *>(a.w);
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);

    // ---- stage the window: lane -> (row = piece*16 + lane/4, physical chunk = lane%4)
    for (int piece = wave; piece < halo_pieces; piece += 4) {
        const int hr = piece * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((hr >> 1) & 3);
        const unsigned flat = (unsigned)(o0 + hr);
        const unsigned r = fdiv(flat, a.d_pitch);
        const int cc = (int)(flat - r * (unsigned)pitch);
        const int iy = (int)r - 1, ix = xs + cc - 1;
        const bool ok = hr < halo_rows && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const char *src = ok ? reinterpret_cast<const char *>(in + ((long long)iy * a.W + ix) * a.in_ld + lc * 8) : zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(win + piece * 1024), 16, 0, 0);
    }
    // ---- stage all nine weight slabs: LDS row = tap*BN + cout
    for (int piece = wave; piece < (WROWS + 15) / 16; piece += 4) {
        const int row = piece * 16 + (lane >> 2);
        const int tap = row / BN, co = row - tap * BN;
        const int key = (((co / NV) & 1) << 1) | ((co >> 1) & 1);
        const int lc = (lane & 3) ^ key;
        const char *src = row < WROWS ? reinterpret_cast<const char *>(wgt + (long long)co * a.Kpad + tap * 32 + lc * 8) : zero_page;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wts + piece * 1024), 16, 0, 0);
    }

    // accumulators start at the bias of this lane's couts (rows exist up to CoutPad), fetched while the window is in flight
    const int cb = lg * NV;
    floatx4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const floatx4 b4 = (floatx4){a.bias[cb + i * 4 + 0], a.bias[cb + i * 4 + 1], a.bias[cb + i * 4 + 2], a.bias[cb + i * 4 + 3]};
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = b4;
    }

    const int wrow_l = (lr >> 2) * NV + (lr & 3); // + 4*i per cout tile, + tap*BN
    const int wkey_l = (((wrow_l / NV) & 1) << 1) | ((wrow_l >> 1) & 1);
    const unsigned wfrag0 = wrow_l * 64 + ((lg ^ wkey_l) << 4);
    const int prow0 = wave * 64 + lr;

    __syncthreads(); // the only barrier: window + weights have landed

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int base = prow0 + (tap / 3) * pitch + (tap % 3);
        const unsigned pa = base * 64 + ((lg ^ ((base >> 1) & 3)) << 4); // tiles j: + j*1024 (key unchanged)
        half8 pf[TP], wf[TC];
#pragma unroll
        for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const half8 *>(win + pa + j * 1024);
#pragma unroll
        for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const half8 *>(wts + tap * (BN * 64) + wfrag0 + i * 256);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: lane (pixel lr of tile j, group lg) owns couts lg*NV .. lg*NV+NV-1
    if (cb + NV > a.Cout) return;
    _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
    const _Float16 *res = reinterpret_cast<const _Float16 *>(a.res);
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const unsigned o = (unsigned)(o0 + wave * 64 + j * 16 + lr);
        const unsigned y = fdiv(o, a.d_pitch);
        const int x = (int)(o - y * (unsigned)pitch);
        if ((int)y >= a.H || x >= a.S || xs + x >= a.W) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
        if (a.act) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = silu_c(v[i]);
        }
        const long long pix = ((long long)n * a.H + (int)y) * a.W + xs + x;
        if (res) {
#pragma unroll
            for (int i = 0; i < NV; i += 8) {
                const half8 rv = *reinterpret_cast<const half8 *>(res + pix * a.res_ld + a.res_coff + cb + i);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i + e] += (float)rv[e];
            }
        }
#pragma unroll
        for (int i = 0; i < NV; i += 8) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[i + e];
            *reinterpret_cast<half8 *>(out + pix * a.out_ld + a.out_coff + cb + i) = h;
        }
    }
}

} // namespace

bool c32_eligible(int k, int stride, int cin, int cout_stored, int is_f16, bool has_out2) {
    return is_f16 && k == 3 && stride == 1 && cin == 32 && cout_stored % 8 == 0 && cout_stored <= 96 && !has_out2;
}

hipError_t launch_conv3x3_c32(HaloArgs a, hipStream_t stream) {
    if (a.Cin != 32 || a.Cout % 8 != 0 || a.Cout > a.CoutPad || a.CoutPad % 32 != 0 || a.CoutPad > 96 || a.out2) return hipErrorInvalidValue;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8 || a.Kpad < 9 * 32) return hipErrorInvalidValue;
    if (a.pitch != a.S + 2 || kBM + 2 * a.pitch + 2 > kRowsMax || a.strips * a.S < a.W) return hipErrorInvalidValue;
    if (a.blocks_per_strip * kBM < a.H * a.pitch) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % 8 || a.res_coff % 8)) return hipErrorInvalidValue;
    const long long blocks = (long long)a.N * a.strips * a.blocks_per_strip;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv((unsigned)a.strips);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    switch (a.CoutPad) {
    case 32: hipLaunchKernelGGL((conv3x3_c32_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
    case 64: hipLaunchKernelGGL((conv3x3_c32_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((conv3x3_c32_kernel<96>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
    }
    return hipGetLastError();
}

} // namespace wtk
