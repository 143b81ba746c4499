// Fused network front (fp16, YOLOv8s widths): predictor preprocess + model.0 (stem, 3x3/s2, 3 -> 32)
// + model.1 (3x3/s2, 32 -> 64) + model.2.cv1 (1x1, 64 -> 64) in ONE persistent kernel.
//
// Why: these three layers live on the largest maps of the network (320^2 x 32 and 160^2 x 64 per 640^2
// frame).  Run one by one they move 419 MB (stem out) + 419 + 210 MB (model.1) + 210 + 210 MB (cv1) per
// 64 frames through HBM for 85 GFLOP — 470 us, all of it bandwidth/issue bound.  Fused, the only HBM
// traffic is the uint8 frames (26 MB) and the cv1 output (210 MB); every intermediate lives in LDS.
//
// One block (8 waves) = one 16x16 tile of the 1/4-resolution map, walked persistently:
//   raw patch (67 x 68 px, prefetched into registers one tile ahead)
//     -> P16   [67][68] x (R,G,B,0) fp16, /255                                   36 KB   (aliases O1)
//     -> S     stem output 33 x 33 px x 32 ch, 64-B rows, columns de-interleaved
//              by parity so a stride-2 tap reads 16 CONSECUTIVE rows              70 KB
//     -> O1    model.1 output 256 px x 64 ch, 128-B rows                          32 KB   (aliases P16)
//     -> cv1   -> global (NHWC slice view)
//   weights of model.1 ([9][64][64 B]) and cv1 ([64][128 B]) stay resident in LDS (45 KB), the stem's in
//   registers.  Each stage rounds to fp16 exactly where the layer-by-layer path stores fp16, walks K in the
//   same order and uses the same MFMA, so the result equals the unfused kernels' bit for bit.
// Barriers are raw s_barrier + lgkmcnt(0) so the next tile's patch loads and the previous tile's output
// stores stay in flight across them (a __syncthreads would drain vmcnt at every stage).
#include "wtk_kernels.h"

namespace wtk {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kT = 16;                 // output tile edge (1/4-resolution pixels)
constexpr int kSE = 2 * kT + 1;        // 33: stem-output tile edge
constexpr int kSEven = kT + 1;         // 17 even columns, then 16 odd ones
constexpr int kPR = 2 * kSE + 1;       // 67 patch rows
constexpr int kPC = 68;                // patch columns, starting at the 4-pixel aligned column 4*ox0 - 4
constexpr int kUnitsPerRow = kPC / 4;  // 17 units of 4 pixels
constexpr int kUnits = kPR * kUnitsPerRow; // 1139
constexpr int kUnitsPerThread = 3;     // 512 threads x 3 >= 1139
constexpr int kSRows = kSE * kSE;      // 1089
constexpr int kStemTiles = (kSRows + 15) / 16; // 69 MFMA pixel tiles

constexpr int kW1Bytes = 9 * 64 * 64;  // [tap][cout] rows of 64 B
constexpr int kW2Bytes = 64 * 128;     // [cout] rows of 128 B
constexpr int kSBytes = kSRows * 64;
constexpr int kPBytes = kPR * kPC * 8; // 36448
constexpr int kO1Bytes = 256 * 128;
constexpr int kPOBytes = ((kPBytes > kO1Bytes ? kPBytes : kO1Bytes) + 63) / 64 * 64;
static_assert(kW1Bytes + kW2Bytes + kSBytes + kPOBytes <= 160 * 1024, "LDS budget");

// (half)(b * (1/255.f)) == (half)(b / 255.f) for every byte b (checked exhaustively): the multiply replaces the
// ~10-instruction IEEE division of the stand-alone stem without changing a single fp16 result.
constexpr float kInv255 = 1.0f / 255.0f;

// Pins an fp32 value in a VGPR.  Without it hipcc folds "(half)(a * b)" into v_fma_mixlo_f16, which rounds the
// exact product ONCE to fp16; the stand-alone kernels round the product to fp32 first (v_mul_f32 +
// v_cvt_pk_f16_f32), and this kernel promises their bits.
__device__ __forceinline__ float pin_f32(float v) {
    asm("" : "+v"(v));
    return v;
}

__device__ __forceinline__ float silu_ff(float x) {
    return wtk_silu_scaled(x); // scaled domain, see wtk_kernels.h
}

__device__ __forceinline__ _Float16 norm_byte(uint32_t b) { return (_Float16)pin_f32((float)b * kInv255); }

// LDS-only barrier: waits for this wave's LDS traffic (lgkmcnt(0)), not for global loads/stores in flight.
// The asm clobbers keep the compiler from moving LDS accesses across it (the s_barrier intrinsic alone is IntrNoMem).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f); // vmcnt = 63 (no wait), expcnt = 7, lgkmcnt = 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool DBG>
__global__ __launch_bounds__(512, 2) void front_fused_kernel(const FrontArgs a) {
    __shared__ __attribute__((aligned(16))) char w1s[kW1Bytes];
    __shared__ __attribute__((aligned(16))) char w2s[kW2Bytes];
    __shared__ __attribute__((aligned(16))) char sbuf[kSBytes];
    __shared__ __attribute__((aligned(16))) char pobuf[kPOBytes];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int tpi = a.tiles_x * a.tiles_y;

    // ---- one-time: model.1 and cv1 weights -> LDS (LDS-DMA, swizzle applied on the source chunk)
    {
        const _Float16 *w1 = reinterpret_cast<const _Float16 *>(a.w1);
        for (int piece = wave; piece < kW1Bytes / 1024; piece += 8) { // 16 rows of 64 B per piece
            const int row = piece * 16 + (lane >> 2);
            const int tap = row >> 6, co = row & 63;
            const int key = (((co >> 4) & 1) << 1) | ((co >> 1) & 1);
            const int lc = (lane & 3) ^ key;
            const char *src = reinterpret_cast<const char *>(w1 + (long long)co * a.Kpad1 + tap * 32 + lc * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(w1s + piece * 1024), 16, 0, 0);
        }
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.w2);
        { // 8 pieces of 8 rows x 128 B: one per wave
            const int row = wave * 8 + (lane >> 3);
            const int key = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
            const int lc = (lane & 7) ^ key;
            const char *src = reinterpret_cast<const char *>(w2 + (long long)row * a.Kpad2 + lc * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(w2s + wave * 1024), 16, 0, 0);
        }
    }

    // ---- stem weights -> registers (same packing as stem_mfma_kernel: [cout][16 taps][4] fp16)
    half8 wf0[2][2];
    {
        const _Float16 *w0 = reinterpret_cast<const _Float16 *>(a.w0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = (lr >> 2) * 8 + i * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf0[i][ks] = *reinterpret_cast<const half8 *>(w0 + (co * 16 + ks * 8 + 2 * lg) * 4);
        }
    }
    // per-lane patch offsets of the two taps this lane feeds in each stem k-step.  Taps 9..15 have zero weights,
    // so they may read any finite value: offset 0
    int tap_off[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int tap = ks * 8 + 2 * lg + hh;
            tap_off[ks][hh] = tap < 9 ? ((tap * 11) >> 5) * kPC + (tap - 3 * ((tap * 11) >> 5)) : 0;
        }
    // stem work list of this lane, tile invariant: iteration it handles S pixel s = 16 * (wave + 8 * it) + lr
    constexpr int kStemIters = (kStemTiles + 7) / 8; // 9
    int st_pb[kStemIters], st_dst[kStemIters], st_yx[kStemIters];
#pragma unroll
    for (int it = 0; it < kStemIters; ++it) {
        const int s_raw = (wave + 8 * it) * 16 + lr;
        const int s = s_raw < kSRows ? s_raw : kSRows - 1;
        const int sy = (s * 1986) >> 16; // s / 33 for s < 1089
        const int sx = s - sy * kSE;
        const int R = sy * kSE + (sx & 1) * kSEven + (sx >> 1);
        st_pb[it] = (2 * sy) * kPC + 2 * sx + 1;
        st_dst[it] = s_raw < kSRows ? R * 64 + ((lg ^ ((R >> 1) & 3)) << 4) : -1;
        st_yx[it] = (sy << 8) | sx;
    }
    float bias0[8], bias1[16], bias2[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias0[i] = a.b0[lg * 8 + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias1[i] = a.b1[lg * 16 + i], bias2[i] = a.b2[lg * 16 + i];

    // ---- raw patch prefetch: unit u = 4 pixels = C dwords
    uint32_t raw[kUnitsPerThread][3];
    auto tile_coords = [&](int tile, int &n, int &oy0, int &ox0) __attribute__((always_inline)) {
        n = (int)fdiv((unsigned)tile, a.d_tpi);
        const unsigned t = (unsigned)tile - (unsigned)n * (unsigned)tpi;
        const unsigned ty = fdiv(t, a.d_tilesx);
        oy0 = (int)ty * kT;
        ox0 = (int)(t - ty * (unsigned)a.tiles_x) * kT;
    };
    auto load_patch = [&](int tile) __attribute__((always_inline)) {
        int n, oy0, ox0;
        tile_coords(tile, n, oy0, ox0);
        const uint8_t *img = a.frames + (long long)n * a.H * a.W * a.C;
#pragma unroll
        for (int k = 0; k < kUnitsPerThread; ++k) {
            const int u = tid + 512 * k;
            const int pr = (u * 241) >> 12; // u / 17 for u < 2048
            const int pu = u - pr * kUnitsPerRow;
            const int iy = 4 * oy0 - 3 + pr, ix = 4 * ox0 - 4 + 4 * pu;
            const bool ok = u < kUnits && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            raw[k][0] = raw[k][1] = raw[k][2] = 0u;
            if (ok) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(img + ((long long)iy * a.W + ix) * a.C);
                raw[k][0] = p[0];
                if (a.C == 3) raw[k][1] = p[1], raw[k][2] = p[2];
            }
        }
    };

    // The finished tile's output is kept in registers and stored one stage later (after the next tile's
    // patch has been consumed): vmcnt retires in order, so a store issued before that point would make the
    // wait for the patch registers also wait for the store acknowledgements.
    half8 pend[2][2];
    _Float16 *pend_ptr[2] = {nullptr, nullptr};
    auto flush_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (pend_ptr[j]) {
                *reinterpret_cast<half8 *>(pend_ptr[j]) = pend[j][0];
                *reinterpret_cast<half8 *>(pend_ptr[j] + 8) = pend[j][1];
            }
    };

    int tile = blockIdx.x;
    if (tile < a.total_tiles) load_patch(tile);
    __syncthreads(); // weights landed (drains vmcnt once)

    // diagnostic build only (WTK_FRONT_STAMPS): per-wave cycle totals of each stage, s_memtime deltas
#ifdef WTK_FRONT_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i)                                                   \
    {                                                              \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        st_sum[i] += now - st_prev;                                \
        st_prev = now;                                             \
    }
#else
#define STAMP(i)
#endif
    for (; tile < a.total_tiles; tile += gridDim.x) {
        int n, oy0, ox0;
        tile_coords(tile, n, oy0, ox0);

        // ======== A: raw registers -> P16 (normalised RGB0 fp16)
#pragma unroll
        for (int k = 0; k < kUnitsPerThread; ++k) {
            const int u = tid + 512 * k;
            if (u < kUnits) {
                half8 lo, hi; // pixels 0,1 and 2,3 of the unit
                if (a.C == 1) {
                    const uint32_t d = raw[k][0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const _Float16 v = norm_byte((d >> (8 * j)) & 0xffu);
                        half8 &dst = j < 2 ? lo : hi;
                        const int o = (j & 1) * 4;
                        dst[o] = v, dst[o + 1] = v, dst[o + 2] = v, dst[o + 3] = (_Float16)0.f;
                    }
                } else {
                    const uint64_t d01 = (uint64_t)raw[k][0] | ((uint64_t)raw[k][1] << 32);
                    const uint64_t d12 = (uint64_t)raw[k][1] | ((uint64_t)raw[k][2] << 32);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // pixel j = bytes 3j (B), 3j+1 (G), 3j+2 (R) of the 12-byte unit
                        const uint32_t bgr = j < 2 ? (uint32_t)(d01 >> (24 * j)) : (uint32_t)(d12 >> (24 * j - 32));
                        half8 &dst = j < 2 ? lo : hi;
                        const int o = (j & 1) * 4;
                        dst[o] = norm_byte((bgr >> 16) & 0xffu);
                        dst[o + 1] = norm_byte((bgr >> 8) & 0xffu);
                        dst[o + 2] = norm_byte(bgr & 0xffu);
                        dst[o + 3] = (_Float16)0.f;
                    }
                }
                char *dstp = pobuf + u * 32; // (pr*68 + 4*pu) * 8 = u * 32
                *reinterpret_cast<half8 *>(dstp) = lo;
                *reinterpret_cast<half8 *>(dstp + 16) = hi;
            }
        }
        STAMP(0);
        flush_pending(); // previous tile's output -> global
        lds_barrier();
        STAMP(1);

        // ======== B: prefetch the next tile's patch; stem: P16 -> S
        if (tile + (int)gridDim.x < a.total_tiles) load_patch(tile + gridDim.x);
        {
            const half4 *patch = reinterpret_cast<const half4 *>(pobuf);
            const int Hs = a.H >> 1, Ws = a.W >> 1;
#pragma unroll
            for (int it = 0; it < kStemIters; ++it) {
                if (wave + 8 * it >= kStemTiles) break; // wave uniform
                const int pbase = st_pb[it];
                const int sy = st_yx[it] >> 8, sx = st_yx[it] & 0xff;
                floatx4 acc[2] = {(floatx4){bias0[0], bias0[1], bias0[2], bias0[3]}, (floatx4){bias0[4], bias0[5], bias0[6], bias0[7]}}; // start at the bias
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8 pf;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const half4 v = patch[pbase + tap_off[ks][hh]];
                        pf[4 * hh + 0] = v.x, pf[4 * hh + 1] = v.y, pf[4 * hh + 2] = v.z, pf[4 * hh + 3] = v.w;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i][ks], pf, acc[i], 0, 0, 0);
                }
                const int gy = 2 * oy0 - 1 + sy, gx = 2 * ox0 - 1 + sx;
                const bool inside = (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
                // silu_ff pins its fp32 product, so rounding is product -> fp32 -> fp16 as in the stand-alone kernels;
                // pixels outside the stem map are model.1's zero padding: mask the packed halves (4 selects, not 8)
                half8 hv;
                {
                    float t[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[i * 4 + r] = acc[i][r];
                    wtk_silu_scaled_run<8>(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)t[e];
                }
                {
                    uint4 bits = __builtin_bit_cast(uint4, hv);
                    const uint32_t m = inside ? 0xffffffffu : 0u;
                    bits.x &= m, bits.y &= m, bits.z &= m, bits.w &= m;
                    hv = __builtin_bit_cast(half8, bits);
                }
                if (st_dst[it] >= 0) {
                    *reinterpret_cast<half8 *>(sbuf + st_dst[it]) = hv;
                    if (DBG && inside) // test hook: materialise the stem output (tiles overlap: same values)
                        *reinterpret_cast<half8 *>(reinterpret_cast<_Float16 *>(a.dbg_t0) + (((long long)n * Hs + gy) * Ws + gx) * 32 + lg * 8) = hv;
                }
            }
        }
        STAMP(2);
        lds_barrier();
        STAMP(3);

        // ======== C: model.1 (3x3 / stride 2 over S) -> O1.  Wave = tile rows 2w, 2w+1 x all 64 couts
        {
            floatx4 acc[4][2]; // accumulators start at the bias (as in every conv kernel of the library)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (floatx4){bias1[i * 4 + 0], bias1[i * 4 + 1], bias1[i * 4 + 2], bias1[i * 4 + 3]};
            const int wrow_l = (lr >> 2) * 16 + (lr & 3);
            const int wkey_l = (((wrow_l >> 4) & 1) << 1) | ((wrow_l >> 1) & 1);
            const unsigned wfrag0 = wrow_l * 64 + ((lg ^ wkey_l) << 4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                half8 pf[2], wf[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int R = (2 * (2 * wave + j) + ky) * kSE + (kx & 1) * kSEven + (kx >> 1) + lr;
                    pf[j] = *reinterpret_cast<const half8 *>(sbuf + R * 64 + ((lg ^ ((R >> 1) & 3)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8 *>(w1s + tap * 4096 + wfrag0 + i * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf[j], acc[i][j], 0, 0, 0);
            }
            STAMP(4);
            // O1 aliases P16, whose last readers passed the barrier above
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (2 * wave + j) * 16 + lr;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    half8 hv;
                    {
                        float t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int idx = c2 * 8 + e; // cout lg*16 + idx = tile idx>>2, row idx&3
                            t[e] = acc[idx >> 2][j][idx & 3];
                        }
                        wtk_silu_scaled_run<8>(t);
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)t[e];
                    }
                    const int c = 2 * lg + c2;
                    *reinterpret_cast<half8 *>(pobuf + p * 128 + ((c ^ (p & 7)) << 4)) = hv;
                    if (DBG && oy0 + 2 * wave + j < a.Ho && ox0 + lr < a.Wo) // test hook: model.1 output
                        *reinterpret_cast<half8 *>(reinterpret_cast<_Float16 *>(a.dbg_t1) +
                                                   (((long long)n * a.Ho + oy0 + 2 * wave + j) * a.Wo + ox0 + lr) * 64 + c * 8) = hv;
                }
            }
        }

        STAMP(5);
        // ======== D: cv1 (1x1, 64 -> 64) over O1 -> registers.  A wave reads back exactly the O1 rows it wrote
        // (same 32 pixels, all 64 channels), and a wave's LDS operations execute in order: no barrier between
        // C and D, so one wave's cv1 / epilogue overlaps its SIMD neighbour's model.1 MFMAs.
        {
            floatx4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (floatx4){bias2[i * 4 + 0], bias2[i * 4 + 1], bias2[i * 4 + 2], bias2[i * 4 + 3]};
            const int wrow_l = (lr >> 2) * 16 + (lr & 3);
            const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l >> 4) & 3) << 1);
            const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
            const int prow = 2 * wave * 16 + lr;
            const unsigned pfrag0 = prow * 128 + ((lg ^ (prow & 7)) << 4);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 pf[2], wf[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) pf[j] = *reinterpret_cast<const half8 *>(pobuf + (pfrag0 ^ (ks * 64u)) + j * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8 *>(w2s + (wfrag0 ^ (ks * 64u)) + i * 512);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], pf[j], acc[i][j], 0, 0, 0);
            }
            _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = oy0 + 2 * wave + j, ox = ox0 + lr;
                pend_ptr[j] = (oy < a.Ho && ox < a.Wo) ? out + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.out_ld + a.out_coff + lg * 16 : nullptr;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int idx = c2 * 8 + e;
                        t[e] = acc[idx >> 2][j][idx & 3];
                    }
                    wtk_silu_scaled_run<8>(t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) pend[j][c2][e] = (_Float16)t[e];
                }
            }
        }
        STAMP(6);
        lds_barrier(); // O1 fully consumed before the next tile's P16 overwrites it
        STAMP(7);
    }
    flush_pending();
#ifdef WTK_FRONT_STAMPS
    if (lane == 0 && a.dbg_stamps)
        for (int i = 0; i < 8; ++i) a.dbg_stamps[((long long)blockIdx.x * 8 + wave) * 8 + i] = st_sum[i];
#endif
}

} // namespace

bool front_fused_eligible(int is_f16, int c0, int c1, int c2_out) { return is_f16 && c0 == 32 && c1 == 64 && c2_out == 64; }

hipError_t launch_front_fused(FrontArgs a, int num_cus, hipStream_t stream) {
    if (a.C != 1 && a.C != 3) return hipErrorInvalidValue;
    if (a.H % 32 || a.W % 32 || a.H <= 0 || a.W <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(a.frames) % 4) return hipErrorInvalidValue; // rows are read as aligned dwords
    if (a.Kpad1 < 288 || a.Kpad1 % 8 || a.Kpad2 < 64 || a.Kpad2 % 8) return hipErrorInvalidValue;
    if (a.out_ld % 8 || a.out_coff % 8 || a.out_coff + 64 > a.out_ld) return hipErrorInvalidValue;
    a.Ho = a.H / 4, a.Wo = a.W / 4;
    a.tiles_x = (a.Wo + kT - 1) / kT;
    a.tiles_y = (a.Ho + kT - 1) / kT;
    const long long total = (long long)a.N * a.tiles_x * a.tiles_y;
    if (total <= 0 || total > 0x7fffffffLL) return hipErrorInvalidValue;
    a.total_tiles = (int)total;
    a.d_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
    a.d_tilesx = make_fastdiv((unsigned)a.tiles_x);
    const unsigned grid = (unsigned)(total < num_cus ? total : num_cus);
    if (a.dbg_t0 && a.dbg_t1)
        hipLaunchKernelGGL(front_fused_kernel<true>, dim3(grid), dim3(512), 0, stream, a);
    else
        hipLaunchKernelGGL(front_fused_kernel<false>, dim3(grid), dim3(512), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
