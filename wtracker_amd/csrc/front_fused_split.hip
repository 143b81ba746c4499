// Fused network front of the split-fp16 ("f16x3") handles, YOLOv8s widths: predictor preprocess + model.0 (stem, 3x3/s2, 3 -> 32,
// split operands — or, WTK_STEM_FP32=1, exact fp32 matrix instructions — as the stand-alone stem of these handles) + model.1 (3x3/s2, 32 -> 64) + model.2.cv1 (1x1, 64 -> 64)
// in ONE persistent kernel.
//
// Why: split tensors double the bytes of the largest maps of the network.  Run one by one these three layers move 839 MB (stem
// out) + 839 + 419 MB (model.1) + 419 + 419 MB (cv1) per 64 frames at 640^2 through HBM — 830 us of the 6.9 ms forward, all of it
// bandwidth / latency bound.  Fused, the HBM traffic is the uint8 frames (26 MB) and the cv1 output (419 MB).
//
// One block (8 waves) = one 16 x 4 tile of the 1/4-resolution map, walked persistently (the doubled rows and weights leave room
// for a quarter of front_fused_kernel's 16 x 16 tile: 145 KB of LDS):
//   raw patch (19 x 68 px, prefetched into registers one tile ahead)
//     -> P     [19][68] x (R,G,B,0) fp32, / 255                                  20 KB   (aliases O1)
//     -> S     stem output 9 x 33 px x 32 ch as split rows [hi32 | lo32] of 128 B,
//              columns de-interleaved by parity so a stride-2 tap reads 16 CONSECUTIVE rows  37 KB
//     -> O1    model.1 output 64 px x 64 ch = two split rows per pixel, block-major  16 KB   (aliases P)
//     -> cv1   -> global (split NHWC slice view)
//   weights of model.1 ([9][64] rows of 128 B) and cv1 ([2][64] rows) stay resident in LDS (88 KB), the stem's in registers.
// Every stage uses the operands, the K order and the instruction sequence of the layer-by-layer path (stem_mfma_kernel<fp16, 2, SPLIT> / <float, 2>,
// conv_igemm_kernel's split form: per K step hi.hi into `acc`, lo.hi then hi.lo into `acc1`; v = acc + acc1 * 2^-11; SiLU; split),
// so the result equals the unfused kernels' bit for bit (tests/test_gpu_f16x3.py switches the fusion off and on).
// Barriers are raw s_barrier + lgkmcnt(0) so the next tile's patch loads and the previous tile's output stores stay in flight.
#include "wtk_kernels.h"

namespace wtk {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kTW = 16, kTH = 4;          // output tile (1/4-resolution pixels)
constexpr int kSW = 2 * kTW + 1;          // 33 stem-output columns
constexpr int kSH = 2 * kTH + 1;          // 9 stem-output rows
constexpr int kSEven = kTW + 1;           // 17 even columns, then 16 odd ones
constexpr int kPR = 2 * kSH + 1;          // 19 patch rows
constexpr int kPC = 68;                   // patch columns, starting at the 4-pixel aligned column 4*ox0 - 4
constexpr int kUnitsPerRow = kPC / 4;     // 17 units of 4 pixels
constexpr int kUnits = kPR * kUnitsPerRow; // 323: one per thread
constexpr int kSRows = kSH * kSW;         // 297
constexpr int kStemTiles = (kSRows + 15) / 16; // 19 MFMA pixel tiles
constexpr int kStemIters = (kStemTiles + 7) / 8; // 3

constexpr int kW1Bytes = 9 * 64 * 128; // [tap][cout] rows [hi32 | lo32]
constexpr int kW2Bytes = 2 * 64 * 128; // [block of 32 input channels][cout] rows
constexpr int kSBytes = kSRows * 128;
constexpr int kPBytes = kPR * kPC * 16;   // fp32 form: (R,G,B,0) fp32 per pixel; split form: 8 bytes of hi halves per pixel, then the lo halves
constexpr int kPHalf = kPR * kPC * 8;
constexpr int kO1Bytes = 2 * 64 * 128; // [block of 32 channels][pixel] rows
constexpr int kPOBytes = ((kPBytes > kO1Bytes ? kPBytes : kO1Bytes) + 63) / 64 * 64;
static_assert(kW1Bytes + kW2Bytes + kSBytes + kPOBytes + 1024 <= 160 * 1024, "LDS budget");
static_assert(kUnits <= 512, "one patch unit per thread");

// LDS-only barrier: waits for this wave's LDS traffic (lgkmcnt(0)), not for global loads/stores in flight.
__device__ __forceinline__ void lds_barrier_s() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f); // vmcnt = 63 (no wait), expcnt = 7, lgkmcnt = 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Packed fp32 forms (v_pk_add_f32 / v_pk_mul_f32) in the epilogues: measured 574 us against 610 us for the scalar forms on the
// BASELINE shape (-DWTK_FFS_PACKED=0 builds the scalar, pinned forms; same IEEE operations, bit-identical results) — unlike the
// window kernels, whose epilogues run beside a wave issuing fp16 matrix instructions back to back, the stages here are VALU bound.
#ifndef WTK_FFS_PACKED
#define WTK_FFS_PACKED 1
#endif
__device__ __forceinline__ float ffs_pin(float v) {
#if WTK_FFS_PACKED
    return v;
#else
    return wtk_pin_f32(v);
#endif
}
// eight fp32 values -> their split halves (wtk_split_store's arithmetic)
__device__ __forceinline__ void split8(const float (&v)[8], half8 &hv, half8 &lv) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)v[e];
        hv[e] = h;
        lv[e] = (_Float16)ffs_pin(ffs_pin(v[e] - (float)h) * kSplitScale);
    }
}

// SS: the stem on split operands (default of the handles; FrontArgs::stem_split) instead of the fp32 matrix instructions (WTK_STEM_FP32=1)
template <bool DBG, bool SS>
__global__ __launch_bounds__(512, 2) void front_fused_split_kernel(const FrontArgs a) {
    __shared__ __attribute__((aligned(16))) char w1s[kW1Bytes];
    __shared__ __attribute__((aligned(16))) char w2s[kW2Bytes];
    __shared__ __attribute__((aligned(16))) char sbuf[kSBytes];
    __shared__ __attribute__((aligned(16))) char pobuf[kPOBytes];
    __shared__ float norm_lut[256]; // byte -> (float)byte / 255.0f, the division of the stand-alone stem, done once per block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int tpi = a.tiles_x * a.tiles_y;
    int total_tiles = a.total_tiles;
    if (a.n_dyn) total_tiles = min(max(*a.n_dyn, 0), a.N) * tpi; // dynamic batch: the tiles of the first *n_dyn images
    if ((int)blockIdx.x >= total_tiles) return;
    if (tid < 256) {
        const float x = (float)tid / 255.0f;
        if constexpr (SS) { // the split pair of the normalised byte: hi in the low half of the word, lo in the high half
            const _Float16 hi = (_Float16)x, lo = (_Float16)((x - (float)hi) * kSplitScale);
            norm_lut[tid] = __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16));
        } else {
            norm_lut[tid] = x;
        }
    }

    // ---- one-time: model.1 and cv1 weights -> LDS (LDS-DMA, swizzle applied on the source chunk); row key of cout co:
    // ((co >> 1) & 1) | (((co >> 3) & 3) << 1) — with the lane -> row map (lr >> 2) * 8 + (lr & 3) + 4 i of stages C / D the
    // slot function of (lr, lg) is conv3x3_halo_kernel's 64-cout one
    {
        const _Float16 *w1 = reinterpret_cast<const _Float16 *>(a.w1);
#pragma unroll
        for (int k = 0; k < kW1Bytes / 1024 / 8; ++k) { // 72 pieces of 8 rows, nine per wave
            const int piece = wave + 8 * k;
            const int row = piece * 8 + (lane >> 3);
            const int tap = row >> 6, co = row & 63;
            const int key = ((co >> 1) & 1) | (((co >> 3) & 3) << 1);
            const int lc = (lane & 7) ^ key;
            const char *src = reinterpret_cast<const char *>(w1 + (long long)co * a.Kpad1 + tap * 64 + lc * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(w1s + piece * 1024), 16, 0, 0);
        }
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.w2);
#pragma unroll
        for (int k = 0; k < 2; ++k) { // 16 pieces, two per wave: LDS row = block * 64 + cout
            const int piece = wave + 8 * k;
            const int row = piece * 8 + (lane >> 3);
            const int blk = row >> 6, co = row & 63;
            const int key = ((co >> 1) & 1) | (((co >> 3) & 3) << 1);
            const int lc = (lane & 7) ^ key;
            const char *src = reinterpret_cast<const char *>(w2 + (long long)co * a.Kpad2 + blk * 64 + lc * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(w2s + piece * 1024), 16, 0, 0);
        }
    }

    // ---- stem weights -> registers.  fp32 form: stem_mfma_kernel<float, 2>'s packing and lane map ([cout][9 taps][4] fp32, lane holds k = lg);
    // split form: stem_mfma_kernel<_Float16, 2, true>'s ([cout][16 taps][4] hi halves, then the lo halves; lane holds taps 2 lg, 2 lg + 1 of a k-step)
    float wf0[SS ? 1 : 2][SS ? 1 : 9];
    half8 wh0[SS ? 2 : 1][SS ? 2 : 1], wl0[SS ? 2 : 1][SS ? 2 : 1];
    int tap_off[2][2] = {{0, 0}, {0, 0}};
    if constexpr (SS) {
        const _Float16 *w0 = reinterpret_cast<const _Float16 *>(a.w0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = (lr >> 2) * 8 + i * 4 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                wh0[i][ks] = *reinterpret_cast<const half8 *>(w0 + (co * 16 + ks * 8 + 2 * lg) * 4);
                wl0[i][ks] = *reinterpret_cast<const half8 *>(w0 + 32 * 64 + (co * 16 + ks * 8 + 2 * lg) * 4);
            }
        }
        // per-lane patch offsets of the two taps this lane feeds in each k-step; taps 9..15 have zero weights and may read any finite value: offset 0
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int tap = ks * 8 + 2 * lg + hh;
                tap_off[ks][hh] = tap < 9 ? ((tap * 11) >> 5) * kPC + (tap - 3 * ((tap * 11) >> 5)) : 0;
            }
    } else {
        const float *w0 = reinterpret_cast<const float *>(a.w0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = (lr >> 2) * 8 + i * 4 + (lr & 3);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf0[i][tap] = w0[(co * 9 + tap) * 4 + lg];
        }
    }
    // stem work list of this lane, tile invariant: iteration `it` handles S pixel s = 16 * (wave + 8 * it) + lr
    int st_pb[kStemIters], st_dst[kStemIters], st_yx[kStemIters];
#pragma unroll
    for (int it = 0; it < kStemIters; ++it) {
        const int s_raw = (wave + 8 * it) * 16 + lr;
        const int s = s_raw < kSRows ? s_raw : kSRows - 1;
        const int sy = (s * 1986) >> 16; // s / 33 for s < 1089
        const int sx = s - sy * kSW;
        const int R = sy * kSW + (sx & 1) * kSEven + (sx >> 1);
        st_pb[it] = (2 * sy) * kPC + 2 * sx + 1;
        st_dst[it] = s_raw < kSRows ? R * 128 + ((lg ^ (R & 7)) << 4) : -1;
        st_yx[it] = (sy << 8) | sx;
    }
    // stages C / D: this wave's output row of the tile and its block of 32 couts; a lane owns couts c0 .. c0 + 7
    const int orow = wave & 3, ch = wave >> 2;
    const int c0 = ch * 32 + lg * 8;
    float bias0[8], bias1[8], bias2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias0[i] = a.b0[lg * 8 + i], bias1[i] = a.b1[c0 + i], bias2[i] = a.b2[c0 + i];
    const int wrow_l = ch * 32 + (lr >> 2) * 8 + (lr & 3); // + 4 i per cout tile, + 64 * (tap | block)
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l >> 3) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);

    // ---- raw patch prefetch: unit u = tid = 4 pixels = C dwords
    uint32_t raw[3];
    auto tile_coords = [&](int tile, int &n, int &oy0, int &ox0) __attribute__((always_inline)) {
        n = (int)fdiv((unsigned)tile, a.d_tpi);
        const unsigned t = (unsigned)tile - (unsigned)n * (unsigned)tpi;
        const unsigned ty = fdiv(t, a.d_tilesx);
        oy0 = (int)ty * kTH;
        ox0 = (int)(t - ty * (unsigned)a.tiles_x) * kTW;
    };
    const int u_pr = (tid * 241) >> 12; // tid / 17 for tid < 2048
    const int u_pu = tid - u_pr * kUnitsPerRow;
    auto load_patch = [&](int tile) __attribute__((always_inline)) {
        int n, oy0, ox0;
        tile_coords(tile, n, oy0, ox0);
        const uint8_t *img = a.frames + (long long)n * a.H * a.W * a.C;
        const int iy = 4 * oy0 - 3 + u_pr, ix = 4 * ox0 - 4 + 4 * u_pu;
        const bool ok = tid < kUnits && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        raw[0] = raw[1] = raw[2] = 0u;
        if (ok) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(img + ((long long)iy * a.W + ix) * a.C);
            raw[0] = p[0];
            if (a.C == 3) raw[1] = p[1], raw[2] = p[2];
        }
    };

    // the finished tile's output rows wait in registers until the next tile's patch has been consumed (vmcnt retires in order: a
    // store issued earlier would make the wait for the patch registers also wait for the store acknowledgements)
    half8 pend[2];
    _Float16 *pend_ptr = nullptr;
    auto flush_pending = [&]() __attribute__((always_inline)) {
        if (pend_ptr) {
            *reinterpret_cast<half8 *>(pend_ptr) = pend[0];
            *reinterpret_cast<half8 *>(pend_ptr + 32) = pend[1];
        }
    };

    int tile = blockIdx.x;
    load_patch(tile);
    __syncthreads(); // weights landed (drains vmcnt once)

    const int Hs = a.H >> 1, Ws = a.W >> 1;
    // diagnostic build only (WTK_FRONT_STAMPS, tools/front_split_stamps.hip): per-wave cycle totals of each stage, s_memtime deltas
#ifdef WTK_FRONT_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i)                                                     \
    {                                                                \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        st_sum[i] += now - st_prev;                                  \
        st_prev = now;                                               \
    }
#else
#define STAMP(i)
#endif
    for (; tile < total_tiles; tile += gridDim.x) {
        int n, oy0, ox0;
        tile_coords(tile, n, oy0, ox0);

        // ======== A: raw registers -> P.  fp32 form: (R,G,B,0) fp32, byte / 255 as the stand-alone stem divides; split form: the hi halves of the
        // four pixels (8 bytes each) in the first half of P, their lo halves in the second
        if (tid < kUnits) {
            uint32_t w3[4][3]; // table words of the (R, G, B) bytes of the unit's four pixels
            if (a.C == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w3[j][0] = w3[j][1] = w3[j][2] = __builtin_bit_cast(uint32_t, norm_lut[(raw[0] >> (8 * j)) & 0xffu]);
            } else {
                const uint64_t d01 = (uint64_t)raw[0] | ((uint64_t)raw[1] << 32);
                const uint64_t d12 = (uint64_t)raw[1] | ((uint64_t)raw[2] << 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) { // pixel j = bytes 3j (B), 3j+1 (G), 3j+2 (R) of the 12-byte unit
                    const uint32_t bgr = j < 2 ? (uint32_t)(d01 >> (24 * j)) : (uint32_t)(d12 >> (24 * j - 32));
                    w3[j][0] = __builtin_bit_cast(uint32_t, norm_lut[(bgr >> 16) & 0xffu]);
                    w3[j][1] = __builtin_bit_cast(uint32_t, norm_lut[(bgr >> 8) & 0xffu]);
                    w3[j][2] = __builtin_bit_cast(uint32_t, norm_lut[bgr & 0xffu]);
                }
            }
            if constexpr (SS) {
                uint4 hi[2], lo[2]; // pixel j: (R | G << 16, B) of the hi halves, the same of the lo halves
                uint32_t *hw = reinterpret_cast<uint32_t *>(hi), *lw = reinterpret_cast<uint32_t *>(lo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hw[2 * j] = (w3[j][0] & 0xffffu) | (w3[j][1] << 16), hw[2 * j + 1] = w3[j][2] & 0xffffu;
                    lw[2 * j] = (w3[j][0] >> 16) | (w3[j][1] & 0xffff0000u), lw[2 * j + 1] = w3[j][2] >> 16;
                }
                uint4 *dh = reinterpret_cast<uint4 *>(pobuf + tid * 32), *dl = reinterpret_cast<uint4 *>(pobuf + kPHalf + tid * 32);
                dh[0] = hi[0], dh[1] = hi[1], dl[0] = lo[0], dl[1] = lo[1];
            } else {
                floatx4 *dstp = reinterpret_cast<floatx4 *>(pobuf + tid * 64); // (pr*68 + 4*pu) * 16 = tid * 64
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dstp[j] = (floatx4){__builtin_bit_cast(float, w3[j][0]), __builtin_bit_cast(float, w3[j][1]), __builtin_bit_cast(float, w3[j][2]), 0.f};
            }
        }
        STAMP(0);
        flush_pending(); // previous tile's output -> global
        lds_barrier_s();
        STAMP(1);

        // ======== B: prefetch the next tile's patch; stem: P -> S (fp32 matrix instructions, one per tap and cout tile)
        if (tile + (int)gridDim.x < total_tiles) load_patch(tile + gridDim.x);
        {
            // all matrix instructions of the wave's (up to three) pixel tiles first — six independent accumulator chains — then the
            // SiLU + split epilogues, which run beside the matrix work of the SIMD's other wave
            const bool third = wave + 16 < kStemTiles; // wave uniform: waves 0..2 own a third tile
            floatx4 acc[kStemIters][2], acc1[SS ? kStemIters : 1][2];
#pragma unroll
            for (int it = 0; it < kStemIters; ++it)
                acc[it][0] = (floatx4){bias0[0], bias0[1], bias0[2], bias0[3]}, acc[it][1] = (floatx4){bias0[4], bias0[5], bias0[6], bias0[7]};
            if constexpr (SS) {
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                const half4 *ph4 = reinterpret_cast<const half4 *>(pobuf), *pl4 = reinterpret_cast<const half4 *>(pobuf + kPHalf);
#pragma unroll
                for (int it = 0; it < kStemIters; ++it) acc1[it][0] = acc1[it][1] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8 ph[kStemIters], pl[kStemIters];
#pragma unroll
                    for (int it = 0; it < kStemIters; ++it)
                        if (it < 2 || third) {
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                const half4 vh = ph4[st_pb[it] + tap_off[ks][hh]], vl = pl4[st_pb[it] + tap_off[ks][hh]];
                                ph[it][4 * hh + 0] = vh.x, ph[it][4 * hh + 1] = vh.y, ph[it][4 * hh + 2] = vh.z, ph[it][4 * hh + 3] = vh.w;
                                pl[it][4 * hh + 0] = vl.x, pl[it][4 * hh + 1] = vl.y, pl[it][4 * hh + 2] = vl.z, pl[it][4 * hh + 3] = vl.w;
                            }
                        }
#pragma unroll
                    for (int it = 0; it < kStemIters; ++it)
                        if (it < 2 || third) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                acc[it][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0[i][ks], ph[it], acc[it][i], 0, 0, 0);
                                acc1[it][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0[i][ks], ph[it], acc1[it][i], 0, 0, 0);
                            }
#pragma unroll
                            for (int i = 0; i < 2; ++i) acc1[it][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0[i][ks], pl[it], acc1[it][i], 0, 0, 0);
                        }
                }
            } else {
                const float *patch = reinterpret_cast<const float *>(pobuf);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int off = (tap / 3) * kPC + tap % 3;
                    float pv[kStemIters];
#pragma unroll
                    for (int it = 0; it < kStemIters; ++it)
                        if (it < 2 || third) pv[it] = patch[(st_pb[it] + off) * 4 + lg];
#pragma unroll
                    for (int it = 0; it < kStemIters; ++it)
                        if (it < 2 || third) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) acc[it][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[i][tap], pv[it], acc[it][i], 0, 0, 0);
                        }
                }
            }
#pragma unroll
            for (int it = 0; it < kStemIters; ++it) {
                if (it == 2 && !third) break; // wave uniform
                const int sy = st_yx[it] >> 8, sx = st_yx[it] & 0xff;
                const int gy = 2 * oy0 - 1 + sy, gx = 2 * ox0 - 1 + sx;
                const bool inside = (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
                float t[8];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (SS)
                            t[i * 4 + r] = wtk_split_value(acc[it][i][r], acc1[it][i][r]);
                        else
                            t[i * 4 + r] = acc[it][i][r];
                    }
                wtk_silu_scaled_run<8, !WTK_FFS_PACKED>(t);
                half8 hv, lv;
                split8(t, hv, lv);
                if (!inside) { // pixels outside the stem map are model.1's zero padding
                    hv = (half8){0, 0, 0, 0, 0, 0, 0, 0};
                    lv = hv;
                }
                if (st_dst[it] >= 0) {
                    *reinterpret_cast<half8 *>(sbuf + st_dst[it]) = hv;
                    *reinterpret_cast<half8 *>(sbuf + (st_dst[it] ^ 64)) = lv;
                    if (DBG && inside) { // test hook: materialise the stem output (tiles overlap: same values)
                        _Float16 *p = reinterpret_cast<_Float16 *>(a.dbg_t0) + (((long long)n * Hs + gy) * Ws + gx) * 64 + lg * 8;
                        *reinterpret_cast<half8 *>(p) = hv;
                        *reinterpret_cast<half8 *>(p + 32) = lv;
                    }
                }
            }
        }
        STAMP(2);
        lds_barrier_s();
        STAMP(3);

        // ======== C: model.1 (3x3 / stride 2 over S) -> O1.  Wave = output row `orow` of the tile x 32 couts
        {
            floatx4 acc[2], acc1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = (floatx4){bias1[i * 4 + 0], bias1[i * 4 + 1], bias1[i * 4 + 2], bias1[i * 4 + 3]};
                acc1[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int R = (2 * orow + ky) * kSW + (kx & 1) * kSEven + (kx >> 1) + lr;
                const unsigned pa = R * 128 + ((lg ^ (R & 7)) << 4);
                const half8 ph = *reinterpret_cast<const half8 *>(sbuf + pa);
                half8 wh[2], wl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wh[i] = *reinterpret_cast<const half8 *>(w1s + tap * 8192 + wfrag0 + i * 512);
#pragma unroll
                for (int i = 0; i < 2; ++i) wl[i] = *reinterpret_cast<const half8 *>(w1s + tap * 8192 + (wfrag0 ^ 64u) + i * 512);
                const half8 pl = *reinterpret_cast<const half8 *>(sbuf + (pa ^ 64u));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ph, acc[i], 0, 0, 0);
                    acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ph, acc1[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], pl, acc1[i], 0, 0, 0);
            }
            // O1 aliases P, whose last readers passed the barrier above
            float t[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) t[i * 4 + r] = wtk_split_value(acc[i][r], acc1[i][r]);
            wtk_silu_scaled_run<8, !WTK_FFS_PACKED>(t);
            half8 hv, lv;
            split8(t, hv, lv);
            const int hrow = ch * 64 + orow * 16 + lr;
            const unsigned oa = hrow * 128 + ((lg ^ (hrow & 7)) << 4);
            *reinterpret_cast<half8 *>(pobuf + oa) = hv;
            *reinterpret_cast<half8 *>(pobuf + (oa ^ 64u)) = lv;
            if (DBG && oy0 + orow < a.Ho && ox0 + lr < a.Wo) { // test hook: model.1 output
                _Float16 *p = reinterpret_cast<_Float16 *>(a.dbg_t1) + (((long long)n * a.Ho + oy0 + orow) * a.Wo + ox0 + lr) * 128 + ch * 64 + lg * 8;
                *reinterpret_cast<half8 *>(p) = hv;
                *reinterpret_cast<half8 *>(p + 32) = lv;
            }
        }
        STAMP(4);
        lds_barrier_s(); // a pixel's 64 channels come from two waves
        STAMP(5);

        // ======== D: cv1 (1x1, 64 -> 64) over O1 -> registers (stored one stage later)
        {
            floatx4 acc[2], acc1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = (floatx4){bias2[i * 4 + 0], bias2[i * 4 + 1], bias2[i * 4 + 2], bias2[i * 4 + 3]};
                acc1[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int hrow = blk * 64 + orow * 16 + lr;
                const unsigned pa = hrow * 128 + ((lg ^ (hrow & 7)) << 4);
                const half8 ph = *reinterpret_cast<const half8 *>(pobuf + pa);
                half8 wh[2], wl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wh[i] = *reinterpret_cast<const half8 *>(w2s + blk * 8192 + wfrag0 + i * 512);
#pragma unroll
                for (int i = 0; i < 2; ++i) wl[i] = *reinterpret_cast<const half8 *>(w2s + blk * 8192 + (wfrag0 ^ 64u) + i * 512);
                const half8 pl = *reinterpret_cast<const half8 *>(pobuf + (pa ^ 64u));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ph, acc[i], 0, 0, 0);
                    acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ph, acc1[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], pl, acc1[i], 0, 0, 0);
            }
            float t[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) t[i * 4 + r] = wtk_split_value(acc[i][r], acc1[i][r]);
            wtk_silu_scaled_run<8, !WTK_FFS_PACKED>(t);
            split8(t, pend[0], pend[1]);
            const int oy = oy0 + orow, ox = ox0 + lr;
            // pseudo-channel view: real channel c of a pixel lives at 64 * (c / 32) + c % 32 (hi) and + 32 (lo)
            pend_ptr = (oy < a.Ho && ox < a.Wo) ? reinterpret_cast<_Float16 *>(a.out) + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.out_ld + a.out_coff + ch * 64 + lg * 8
                                                : nullptr;
        }
        STAMP(6);
        lds_barrier_s(); // O1 fully consumed before the next tile's P overwrites it
        STAMP(7);
    }
    flush_pending();
#ifdef WTK_FRONT_STAMPS
    if (lane == 0 && a.dbg_stamps)
        for (int i = 0; i < 8; ++i) a.dbg_stamps[((long long)blockIdx.x * 8 + wave) * 8 + i] = st_sum[i];
#endif
}

} // namespace

bool front_fused_split_eligible(int c0, int c1, int c2_out) { return c0 == 32 && c1 == 64 && c2_out == 64; }

// a.out_ld / a.out_coff in pseudo-channels (2 x real); a.Kpad1 / a.Kpad2 pseudo as well (576 / 128); a.w0 the split stem packing (a.stem_split) or the fp32 one
hipError_t launch_front_fused_split(FrontArgs a, int num_cus, hipStream_t stream) {
    if (a.C != 1 && a.C != 3) return hipErrorInvalidValue;
    if (a.H % 32 || a.W % 32 || a.H <= 0 || a.W <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(a.frames) % 4) return hipErrorInvalidValue; // rows are read as aligned dwords
    if (a.Kpad1 != 9 * 64 || a.Kpad2 != 128) return hipErrorInvalidValue;
    if (a.out_ld % 64 || a.out_coff % 64 || a.out_coff + 128 > a.out_ld) return hipErrorInvalidValue;
    a.Ho = a.H / 4, a.Wo = a.W / 4;
    a.tiles_x = (a.Wo + kTW - 1) / kTW;
    a.tiles_y = (a.Ho + kTH - 1) / kTH;
    const long long total = (long long)a.N * a.tiles_x * a.tiles_y;
    if (total <= 0 || total > 0x7fffffffLL) return hipErrorInvalidValue;
    a.total_tiles = (int)total;
    a.d_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
    a.d_tilesx = make_fastdiv((unsigned)a.tiles_x);
    const unsigned grid = (unsigned)(total < num_cus ? total : num_cus);
    const bool dbg = a.dbg_t0 && a.dbg_t1;
    if (a.stem_split) {
        if (dbg)
            hipLaunchKernelGGL((front_fused_split_kernel<true, true>), dim3(grid), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((front_fused_split_kernel<false, true>), dim3(grid), dim3(512), 0, stream, a);
    } else {
        if (dbg)
            hipLaunchKernelGGL((front_fused_split_kernel<true, false>), dim3(grid), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((front_fused_split_kernel<false, false>), dim3(grid), dim3(512), 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace wtk
