// Fused C2f tail for the first C2f block of YOLOv8s (model.2: hidden width 32, one bottleneck, shortcut), fp16:
//   m.0.cv1 (3x3, 32 -> 32) + m.0.cv2 (3x3, 32 -> 32, + residual) + cv2 (1x1 over [a | b | m], 96 -> 64)
// in ONE persistent kernel.  Input is the [a | b] tensor the fused front (or model.2.cv1) wrote into the concat
// buffer; output is the C2f result.  Run one by one these three layers move 105+105, 105+105+105 and 315+210 MB
// per 64 frames of 640^2 through HBM for 80 GFLOP (316 us, bandwidth/issue bound); fused, HBM sees [a | b] once
// (b with a 2-pixel halo) and the 64-channel result: 480 MB.
//
// One block (8 waves) = one 16 x 16 output tile.  LDS (155 KB):
//   B0/B1  b channels on the 20 x 20 halo window, flat rows (row = by*20 + bx, 64 B each), double buffered:
//          the next tile's window streams in by LDS-DMA while this one is computed               2 x 26 KB
//   T1     m.0.cv1 output on the flat 18-row strip (pitch 20, columns 18/19 are don't-care)          23 KB
//   A, M   a channels / bottleneck output, 256 compact rows, WAVE-LOCAL (a wave loads and writes exactly
//          the rows it later reads, so neither needs a barrier)                                   2 x 16 KB
//   weights of the three convs, resident                                                             48 KB
// Two barriers per tile (T1 complete; window landed + T1/window reads retired).  Every stage rounds to fp16 where the layer-by-layer
// kernels store fp16, walks K in the same order (taps 0..8; a, b, m) and uses the same MFMA + SiLU, so the
// result is bit-identical to conv3x3_c32_kernel x2 + conv_igemm_kernel.
#include "wtk_kernels.h"

namespace wtk {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kT = 16;             // output tile edge
constexpr int kP = kT + 4;         // 20: window edge = flat pitch
constexpr int kBRows = 416;        // 26 LDS-DMA pieces of 16 rows (400 window rows + finite padding)
constexpr int kT1Tiles = 23;       // flat strip of 18 rows x pitch 20 = 358 outputs -> 23 MFMA pixel tiles
constexpr int kT1Rows = kT1Tiles * 16;
constexpr int kWmBytes = 9 * 32 * 64; // [tap][cout] rows of 64 B
constexpr int kWcBytes = 3 * 64 * 64; // [k slab][cout] rows of 64 B
static_assert(2 * kBRows * 64 + kT1Rows * 64 + 2 * 256 * 64 + 2 * kWmBytes + kWcBytes <= 160 * 1024, "LDS budget");

__device__ __forceinline__ float pin_f32(float v) {
    asm("" : "+v"(v));
    return v;
}
// same SiLU as the stand-alone kernels; the product is pinned to fp32 so "(half)(x * r)" is never folded into a
// single-rounding v_fma_mixlo_f16 (see front_fused.hip)
__device__ __forceinline__ float silu_cf(float x) {
    return wtk_silu_scaled(x); // scaled domain, see wtk_kernels.h
}
// Raw barriers: s_waitcnt + s_barrier, with compiler-level memory clobbers so no LDS access is moved across them
// (the s_barrier intrinsic alone is IntrNoMem).  lds_barrier leaves global loads/stores and LDS-DMA in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void vm_lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0) lgkmcnt(0): this wave's LDS-DMA pieces have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned row64(int r, int chunk) { return (unsigned)(r * 64 + ((chunk ^ ((r >> 1) & 3)) << 4)); }

__global__ __launch_bounds__(512, 2) void c2f32_fused_kernel(const C2fArgs a) {
    __shared__ __attribute__((aligned(16))) char bwin0[kBRows * 64];
    __shared__ __attribute__((aligned(16))) char bwin1[kBRows * 64];
    __shared__ __attribute__((aligned(16))) char t1buf[kT1Rows * 64];
    __shared__ __attribute__((aligned(16))) char abuf[256 * 64];
    __shared__ __attribute__((aligned(16))) char mbuf[256 * 64];
    __shared__ __attribute__((aligned(16))) char wm1s[kWmBytes];
    __shared__ __attribute__((aligned(16))) char wm2s[kWmBytes];
    __shared__ __attribute__((aligned(16))) char wcs[kWcBytes];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);
    const _Float16 *cat = reinterpret_cast<const _Float16 *>(a.cat);
    const int tpi = a.tiles_x * a.tiles_y;

    auto tile_coords = [&](int tile, int &n, int &y0, int &x0) __attribute__((always_inline)) {
        n = (int)fdiv((unsigned)tile, a.d_tpi);
        const unsigned t = (unsigned)tile - (unsigned)n * (unsigned)tpi;
        const unsigned ty = fdiv(t, a.d_tilesx);
        y0 = (int)ty * kT;
        x0 = (int)(t - ty * (unsigned)a.tiles_x) * kT;
    };
    // window of b: LDS row r = by*20 + bx <-> image pixel (y0 - 2 + by, x0 - 2 + bx); outside -> zero page
    const int dma_row = lane >> 2;                 // row inside a 16-row piece
    auto issue_window = [&](char *dst, int tile) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        for (int piece = wave; piece < kBRows / 16; piece += 8) {
            const int r = piece * 16 + dma_row;
            const int by = (r * 3277) >> 16; // r / 20 for r < 416
            const int bx = r - by * kP;
            const int iy = y0 - 2 + by, ix = x0 - 2 + bx;
            const int lc = (lane & 3) ^ ((r >> 1) & 3);
            const bool ok = r < kP * kP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const char *src = ok ? reinterpret_cast<const char *>(cat + (((long long)n * a.H + iy) * a.W + ix) * a.cat_ld + a.b_coff + lc * 8) : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
        }
    };
    // a channels of this wave's own 32 pixels (tile rows 2w, 2w+1): compact row p = y*16 + x
    auto issue_a = [&](int tile) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = (2 * wave + j) * 16 + dma_row;
            const int iy = y0 + 2 * wave + j, ix = x0 + dma_row;
            const int lc = (lane & 3) ^ ((p >> 1) & 3);
            const bool ok = iy < a.H && ix < a.W;
            const char *src = ok ? reinterpret_cast<const char *>(cat + (((long long)n * a.H + iy) * a.W + ix) * a.cat_ld + a.a_coff + lc * 8) : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(abuf + (2 * wave + j) * 1024), 16, 0, 0);
        }
    };

    // ---- one-time: weights -> LDS
    {
        const _Float16 *w1 = reinterpret_cast<const _Float16 *>(a.w_m1);
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.w_m2);
        for (int piece = wave; piece < kWmBytes / 1024; piece += 8) { // 18 pieces of 16 rows; row = tap*32 + cout
            const int row = piece * 16 + dma_row;
            const int tap = row >> 5, co = row & 31;
            const int key = (((co >> 3) & 1) << 1) | ((co >> 1) & 1);
            const int lc = (lane & 3) ^ key;
            const long long off = (long long)co * a.Kpad_m + tap * 32 + lc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w1 + off),
                                             (__attribute__((address_space(3))) void *)(wm1s + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w2 + off),
                                             (__attribute__((address_space(3))) void *)(wm2s + piece * 1024), 16, 0, 0);
        }
        const _Float16 *wc = reinterpret_cast<const _Float16 *>(a.w_cv2);
        for (int piece = wave; piece < kWcBytes / 1024; piece += 8) { // 12 pieces; row = slab*64 + cout
            const int row = piece * 16 + dma_row;
            const int slab = row >> 6, co = row & 63;
            const int key = (((co >> 4) & 1) << 1) | ((co >> 1) & 1);
            const int lc = (lane & 3) ^ key;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wc + (long long)co * a.Kpad_cv2 + slab * 32 + lc * 8),
                                             (__attribute__((address_space(3))) void *)(wcs + piece * 1024), 16, 0, 0);
        }
    }

    float bias1[8], bias2[8], bias3[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias1[i] = a.b_m1[lg * 8 + i], bias2[i] = a.b_m2[lg * 8 + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias3[i] = a.b_cv2[lg * 16 + i];

    // stage-1 work list of this lane (tile invariant): pixel tile q = wave + 8*it covers flat outputs o = 16q + lr
    int s1_yx[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int o = (wave + 8 * it) * 16 + lr;
        const int oy = (o * 3277) >> 16; // o / 20
        s1_yx[it] = (oy << 8) | (o - oy * kP);
    }
    // weight fragment addresses (A operand): 3x3 convs NV = 8, cv2 NV = 16
    const int wrow_m = (lr >> 2) * 8 + (lr & 3);
    const unsigned wfrag_m = wrow_m * 64 + ((lg ^ ((((wrow_m >> 3) & 1) << 1) | ((wrow_m >> 1) & 1))) << 4);
    const int wrow_c = (lr >> 2) * 16 + (lr & 3);
    const unsigned wfrag_c = wrow_c * 64 + ((lg ^ ((((wrow_c >> 4) & 1) << 1) | ((wrow_c >> 1) & 1))) << 4);

#ifdef WTK_C2F_STAMPS // diagnostic builds only: per-wave cycle totals of each stage (s_memtime deltas)
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define C2F_STAMP(i)                                                 \
    {                                                                \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        st_sum[i] += now - st_prev;                                  \
        st_prev = now;                                               \
    }
#else
#define C2F_STAMP(i)
#endif
    auto do_tile = [&](const char *bcur, char *bnext, int tile, int next_tile) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        // next window + this wave's a rows stream in while stages 1 and 2 run
        issue_a(tile);
        C2F_STAMP(0);
        if (next_tile < a.total_tiles) issue_window(bnext, next_tile);

        // ======== stage 1: m.0.cv1 over the window -> T1 (flat, pitch 20)
        {
            floatx4 acc[3][2]; // accumulators start at the bias (as in every conv kernel of the library)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[q][i] = (floatx4){bias1[i * 4 + 0], bias1[i * 4 + 1], bias1[i * 4 + 2], bias1[i * 4 + 3]};
            const bool third = wave + 16 < kT1Tiles; // wave uniform: waves 0..6 own three pixel tiles, wave 7 two
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int shift = (tap / 3) * kP + tap % 3;
                half8 wf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const half8 *>(wm1s + tap * 2048 + wfrag_m + i * 256);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q == 2 && !third) break;
                    const int r = (wave + 8 * q) * 16 + lr + shift;
                    const half8 pf = *reinterpret_cast<const half8 *>(bcur + row64(r, lg));
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[q][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf, acc[q][i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q == 2 && !third) break;
                const int oy = s1_yx[q] >> 8, ox = s1_yx[q] & 0xff;
                // T1 outside the image is m.0.cv2's zero padding
                const bool inside = (unsigned)(y0 - 1 + oy) < (unsigned)a.H && (unsigned)(x0 - 1 + ox) < (unsigned)a.W;
                half8 hv;
                {
                    float t[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[i * 4 + r] = acc[q][i][r];
                    wtk_silu_scaled_run<8>(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)t[e];
                }
                uint4 bits = __builtin_bit_cast(uint4, hv);
                const uint32_t m = inside ? 0xffffffffu : 0u;
                bits.x &= m, bits.y &= m, bits.z &= m, bits.w &= m;
                const int o = (wave + 8 * q) * 16 + lr;
                *reinterpret_cast<uint4 *>(t1buf + row64(o, lg)) = bits;
            }
        }
        C2F_STAMP(1);
        lds_barrier(); // T1 complete
        C2F_STAMP(2);

        // ======== stage 2: m.0.cv2 over T1 (+ b) -> M.  Wave = tile rows 2w, 2w+1, all 32 couts
        half8 bfrag[2]; // the pixel's own b channels: residual here, and the SAME 16 bytes are cv2's B fragment of k-step 1
        {
            floatx4 acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = (floatx4){bias2[i * 4 + 0], bias2[i * 4 + 1], bias2[i * 4 + 2], bias2[i * 4 + 3]};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                half8 wf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const half8 *>(wm2s + tap * 2048 + wfrag_m + i * 256);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = (2 * wave + j + tap / 3) * kP + tap % 3 + lr;
                    const half8 pf = *reinterpret_cast<const half8 *>(t1buf + row64(r, lg));
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf, acc[j][i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = (2 * wave + j + 2) * kP + 2 + lr; // the pixel's own b row in the window
                bfrag[j] = *reinterpret_cast<const half8 *>(bcur + row64(rb, lg));
                const half8 res = bfrag[j];
                half8 hv;
                {
                    float t[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[i * 4 + r] = acc[j][i][r];
                    wtk_silu_scaled_run<8>(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)pin_f32(t[e] + (float)res[e]);
                }
                const int p = (2 * wave + j) * 16 + lr;
                *reinterpret_cast<half8 *>(mbuf + row64(p, lg)) = hv;
            }
        }

        // The next window (issued at the top) has had two stages to land; every wave confirms its own LDS-DMA pieces
        // (window + its a rows), then the barrier retires all reads of bcur / T1.  Stage 3 below touches only
        // wave-local LDS (A, M) and registers, so waves run on into the next tile without another barrier and this
        // tile's output stores stay in flight until the same point of the next tile.
        C2F_STAMP(3);
        vm_lds_barrier();
        C2F_STAMP(4);

        // ======== stage 3: cv2 over [a | b | m] -> global.  Same pixels as stage 2: M and A are wave-local
        {
            floatx4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (floatx4){bias3[i * 4 + 0], bias3[i * 4 + 1], bias3[i * 4 + 2], bias3[i * 4 + 3]};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                half8 wf[4], pf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8 *>(wcs + ks * 4096 + wfrag_c + i * 256);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = (2 * wave + j) * 16 + lr;
                    if (ks == 0)
                        pf[j] = *reinterpret_cast<const half8 *>(abuf + row64(p, lg));
                    else if (ks == 1)
                        pf[j] = bfrag[j];
                    else
                        pf[j] = *reinterpret_cast<const half8 *>(mbuf + row64(p, lg));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf[j], acc[i][j], 0, 0, 0);
            }
            _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = y0 + 2 * wave + j, ox = x0 + lr;
                if (oy < a.H && ox < a.W) {
                    _Float16 *o = out + (((long long)n * a.H + oy) * a.W + ox) * a.out_ld + a.out_coff + lg * 16;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        half8 hv;
                        {
                            float t[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int idx = c2 * 8 + e;
                                t[e] = acc[idx >> 2][j][idx & 3];
                            }
                            wtk_silu_scaled_run<8>(t);
#pragma unroll
                            for (int e = 0; e < 8; ++e) hv[e] = (_Float16)t[e];
                        }
                        *reinterpret_cast<half8 *>(o + c2 * 8) = hv;
                    }
                }
            }
        }
        C2F_STAMP(5);
    };

    int tile = blockIdx.x;
    if (tile < a.total_tiles) issue_window(bwin0, tile);
    __syncthreads(); // weights + first window landed
    const int step = gridDim.x;
    while (tile < a.total_tiles) {
        do_tile(bwin0, bwin1, tile, tile + step);
        tile += step;
        if (tile >= a.total_tiles) break;
        do_tile(bwin1, bwin0, tile, tile + step);
        tile += step;
    }
#ifdef WTK_C2F_STAMPS
    if (lane == 0 && a.dbg_stamps)
        for (int i = 0; i < 6; ++i) a.dbg_stamps[((long long)blockIdx.x * 8 + wave) * 6 + i] = st_sum[i];
#endif
}

} // namespace

bool c2f_fused_eligible(int is_f16, int c_hidden, int n_bottlenecks, int shortcut, int c_out) {
    return is_f16 && c_hidden == 32 && n_bottlenecks == 1 && shortcut && c_out == 64;
}

hipError_t launch_c2f_fused(C2fArgs a, int num_cus, hipStream_t stream) {
    if (a.N <= 0 || a.H <= 0 || a.W <= 0) return hipErrorInvalidValue;
    if (a.cat_ld % 8 || a.a_coff % 8 || a.b_coff % 8 || a.a_coff + 32 > a.cat_ld || a.b_coff + 32 > a.cat_ld) return hipErrorInvalidValue;
    if (a.out_ld % 8 || a.out_coff % 8 || a.out_coff + 64 > a.out_ld) return hipErrorInvalidValue;
    if (a.Kpad_m < 288 || a.Kpad_m % 8 || a.Kpad_cv2 < 96 || a.Kpad_cv2 % 8 || !a.zeros) return hipErrorInvalidValue;
    a.tiles_x = (a.W + kT - 1) / kT;
    a.tiles_y = (a.H + kT - 1) / kT;
    const long long total = (long long)a.N * a.tiles_x * a.tiles_y;
    if (total <= 0 || total > 0x3fffffffLL) return hipErrorInvalidValue;
    a.total_tiles = (int)total;
    a.d_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
    a.d_tilesx = make_fastdiv((unsigned)a.tiles_x);
    const unsigned grid = (unsigned)(total < num_cus ? total : num_cus);
    hipLaunchKernelGGL(c2f32_fused_kernel, dim3(grid), dim3(512), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
