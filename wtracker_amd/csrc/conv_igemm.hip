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

#ifndef WTK_IGEMM_ORDER
#define WTK_IGEMM_ORDER 0 // 1: issue the next stage's LDS-DMA between the two k-halves (measured 2 % slower: less time to land before the barrier)
#endif

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
    return wtk_silu_scaled(x); // x is the log2(e)-scaled pre-activation (wtk_kernels.h)
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

// LDS-DMA requests in buffer form (SGPR resource + wave-uniform byte offset + per-lane 32-bit offset) from inline asm: 5-10 % less
// wave time per request than the flat form (tools/lds_dma_rate.hip), no 64-bit address arithmetic and no zero-page select per row
// (a lane offset of 0xffffffff is out of range and lands zeros).  hipcc does not see these requests: every barrier that publishes a
// stage is preceded by an explicit s_waitcnt vmcnt(0).  The two-source (UP) loader keeps the builtin form.
#ifndef WTK_IGEMM_BUFFER_DMA
#define WTK_IGEMM_BUFFER_DMA 1
#endif
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *base) {
    const unsigned long long b = (unsigned long long)base;
    rsrc_t r;
    r.x = (int)(unsigned)(b & 0xffffffffu);
    r.y = (int)(unsigned)((b >> 32) & 0xffffu);
    r.z = (int)0xffffff00u;
    r.w = 0x00020000;
    return r;
}
template <bool NTL> __device__ __forceinline__ void dma_buf(const rsrc_t &rs, unsigned voff, unsigned soff, char *lds_dst) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
    if constexpr (NTL)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}

// K1: 1x1 stride-1 conv (no tap walk).  NT: the pixel operand is read by exactly one cout tile, so its
// LDS-DMA loads carry the non-temporal hint (measured: +8..17 % on the HBM-bound 1x1 layers, -15..25 % when a
// second cout tile re-reads the rows from L2 — hence only for CoutPad == BN).
// UP (K1 only): two-source input, channels [0, in2_split) come from a half-resolution tensor (see ConvArgs::in2).
// TAIL: fused 1x1 tail (ConvArgs::tail_w), fp16 128x128 tile only.
// SPLIT (T = fp16): operands are split-fp16 tensors (wtk_kernels.h): a 128-byte row is 32 channels as [hi32 | lo32], a K step is three MFMAs
// per tile pair (hi*hi into acc, hi*lo + lo*hi into acc1), the epilogue combines acc + 2^-11 acc1 and stores split (or fp32) values.
template <typename T, int BM, int BN, int WAVES_P, int WAVES_C, bool K1, bool NT, bool UP = false, bool TAIL = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * WAVES_P * WAVES_C, (2 * (BM + BN) * 128 <= 52 * 1024) ? 3 : 2) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(!SPLIT || (sizeof(T) == 2 && !TAIL), "split mode: fp16 storage, no fused tail");
    constexpr int CE = Elem<T>::CE;
    constexpr int BKE = 8 * CE; // K elements per step = one 128-byte row
    constexpr int WP = BM / WAVES_P, WC = BN / WAVES_C;
    constexpr int TP = WP / 16, TC = WC / 16;
    constexpr int NW = WAVES_P * WAVES_C;     // 4 waves (256 threads, 2 blocks / CU) or 8 waves (512, 1 block / CU)
    constexpr int RPP = 8 * NW;               // tile rows staged per pass of the whole block
    constexpr int PR = BM / RPP, WR = BN / RPP; // staging rows per thread
    constexpr int NV = 4 * TC;                // consecutive couts owned by a lane
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile must be a multiple of the staging pass");
    static_assert(TP >= 1 && TC >= 1, "tile too small");

    // Two DISTINCT LDS objects (not one array indexed by `buf`): hipcc's waitcnt pass tracks pending
    // LDS-DMA writes per LDS object, so it can tell that the DMA filling one buffer does not alias the
    // ds_reads of the other and does not drain vmcnt before every fragment read.
    __shared__ __attribute__((aligned(16))) char smem0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char smem1[STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave / WAVES_C;
    const int wave_c = wave % WAVES_C;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- persistent tile schedule.  Tiles (pixel tile major, cout tile minor) are cut into 8 contiguous
    // ranges, one per XCD label (blockIdx % 8 names the blocks that share an XCD and its L2 — a speed
    // heuristic only); the blocks of a label walk their range with stride = #blocks of that label, so at
    // any moment an XCD works on neighbouring tiles (shared 3x3 halos, shared weights).
    const int nct = a.CoutPad / BN;
    int ptiles_eff = a.ptiles;
    if (a.n_dyn) { // dynamic batch: only the pixel tiles that start inside the first *n_dyn images (rows behind them are scratch)
        const long long n_eff = min(max(*a.n_dyn, 0), a.N);
        const long long pt = a.tile_w == 0 ? (n_eff * a.Ho * a.Wo + BM - 1) / BM : n_eff * a.tiles_x * a.tiles_y;
        ptiles_eff = (int)min((long long)a.ptiles, pt);
    }
    const int total_tiles = ptiles_eff * nct;
    int t_begin, my_tiles, t_stride;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int nbx = (G - xcd + 7) >> 3;
        const int q = total_tiles >> 3, r = total_tiles & 7;
        const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int count = xcd < r ? q + 1 : q;
        t_begin = first + slot;
        t_stride = nbx;
        my_tiles = slot < count ? (count - slot + nbx - 1) / nbx : 0;
    }
    if (my_tiles == 0) return;

    const int HoWo = a.Ho * a.Wo;
    const int tile_h = a.tile_w > 0 ? BM / a.tile_w : 0;
    const int tpi = a.tiles_x * a.tiles_y;

    auto pixel_coords = [&](int ptile, int p, int &n, int &ho, int &wo) __attribute__((always_inline)) -> bool {
        if (a.tile_w == 0) {
            const long long m64 = (long long)ptile * BM + p;
            if (m64 >= a.M) return false;
            const unsigned m = (unsigned)m64;
            n = (int)fdiv(m, a.d_howo);
            const unsigned rem = m - (unsigned)n * (unsigned)HoWo;
            ho = (int)fdiv(rem, a.d_wo);
            wo = (int)(rem - (unsigned)ho * (unsigned)a.Wo);
            return true;
        } else {
            n = (int)fdiv((unsigned)ptile, a.d_tpi);
            const unsigned t = (unsigned)ptile - (unsigned)n * (unsigned)tpi;
            const unsigned ty = fdiv(t, a.d_tilesx), tx = t - ty * (unsigned)a.tiles_x;
            const unsigned py = fdiv((unsigned)p, a.d_tilew), px = (unsigned)p - py * (unsigned)a.tile_w;
            ho = (int)(ty * tile_h + py);
            wo = (int)(tx * a.tile_w + px);
            return ho < a.Ho && wo < a.Wo;
        }
    };

    // ---- staging assignment: thread -> 16-byte physical chunk `ch` of rows r0 + 32*i.
    // LDS-DMA (global_load_lds_dwordx4): one wave instruction fills 8 rows x 128 B of the LDS image
    // linearly (dest = wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
    // chunk: the lane that lands on physical chunk p of row r fetches logical chunk p ^ key(r).  Rows
    // r0 + 32*i of one thread share key(r) for the pixel tile (key = r & 7), so the thread's logical K
    // chunk — and with it its (tap, channel) walk — is the same for all its rows.  Out-of-image taps /
    // ragged rows fetch from a zero page instead of being predicated (an inactive lane would leave stale
    // LDS bytes behind).
    const int ch = tid & 7;
    const int r0 = tid >> 3;
    const int lchunk = ch ^ (r0 & 7);
    const T *in = reinterpret_cast<const T *>(a.in);
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);
    const int ntaps = a.KH * a.KW;
    const int nk = a.Kpad / BKE;

    unsigned wvoff[WR]; // per-lane byte offset inside a [BN][Kpad] weight slab, loop invariant
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = r0 + RPP * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wvoff[i] = (unsigned)(((long long)row * a.Kpad + (ch ^ key) * CE) * (long long)sizeof(T));
    }

    // loader state: the tile / K step being staged runs one stage ahead of the tile being computed
    long long pbase[PR]; // element offset of the (hi0, wi0) input pixel of each staged row (-1: no pixel)
    long long pbase2[UP ? PR : 1]; // UP: element offset of the half-resolution pixel (ho/2, wo/2) in in2
    const T *in2 = reinterpret_cast<const T *>(a.in2);
    int phi0[PR], pwi0[PR];
    const char *wslab = nullptr; // wave-uniform
    int kc = 0, tap = 0, ld_ks = 0, ld_i = 0;
    constexpr bool BUF = WTK_IGEMM_BUFFER_DMA;
    unsigned poff[PR]; // BUF: byte offset of the row's (first) input pixel from the tile's resource base (0xffffffff: no pixel)
    unsigned poff2[UP ? PR : 1]; // ... of the half-resolution pixel (ho/2, wo/2) in in2
    rsrc_t in_rs = {0, 0, 0, 0}, in2_rs = {0, 0, 0, 0}, w_rs = {0, 0, 0, 0};
    auto setup_loader = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int n0 = (tile - ptile * nct) * BN;
        int nb = 0; // BUF: image of the tile's first pixel (wave-uniform); offsets are relative to it
        if constexpr (BUF) {
            if (K1 && !UP && a.tile_w == 0) {
                in_rs = make_rsrc(in + (long long)ptile * BM * a.in_ld + a.in_coff);
            } else {
                int hb, wb;
                pixel_coords(ptile, 0, nb, hb, wb);
                nb = __builtin_amdgcn_readfirstlane(nb);
                in_rs = make_rsrc(in + (long long)nb * a.H * a.W * a.in_ld + a.in_coff);
                if (UP) in2_rs = make_rsrc(in2 + (long long)nb * (a.H >> 1) * (a.W >> 1) * a.in2_ld + a.in2_coff);
            }
        }
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            int n, ho, wo;
            if constexpr (BUF) {
                if (K1 && !UP && a.tile_w == 0) {
                    const long long m = (long long)ptile * BM + r0 + RPP * r;
                    poff[r] = m < a.M ? (unsigned)(((r0 + RPP * r) * a.in_ld + lchunk * CE) * (int)sizeof(T)) : 0xffffffffu;
                    continue;
                }
                const bool okb = pixel_coords(ptile, r0 + RPP * r, n, ho, wo);
                if (K1) {
                    poff[r] = okb ? (unsigned)((((((n - nb) * a.H + ho) * a.W + wo) * a.in_ld) + lchunk * CE) * (int)sizeof(T)) : 0xffffffffu;
                    if (UP)
                        poff2[r] = okb ? (unsigned)((((((n - nb) * (a.H >> 1) + (ho >> 1)) * (a.W >> 1) + (wo >> 1)) * a.in2_ld) + lchunk * CE) * (int)sizeof(T)) : 0xffffffffu;
                } else if (okb) {
                    phi0[r] = ho * a.stride - a.pad;
                    pwi0[r] = wo * a.stride - a.pad;
                    poff[r] = (unsigned)((((n - nb) * a.H + phi0[r]) * a.W + pwi0[r]) * a.in_ld * (int)sizeof(T)); // may wrap below 0: only used with an in-range tap
                } else {
                    phi0[r] = -(1 << 28);
                    pwi0[r] = -(1 << 28);
                    poff[r] = 0;
                }
                continue;
            }
            if (K1 && !UP && a.tile_w == 0) {
                // 1x1 / stride 1 over a linear pixel range: input pixel index = output pixel index m, no (n, ho, wo) needed.  The two
                // divisions per staged row this used to spend sit inside the MFMA loop of the previous tile (K = 128: two K steps per tile)
                const long long m = (long long)ptile * BM + r0 + RPP * r;
                pbase[r] = m < a.M ? m * a.in_ld + a.in_coff + lchunk * CE : -1;
                continue;
            }
            const bool ok = pixel_coords(ptile, r0 + RPP * r, n, ho, wo);
            if (K1) {
                pbase[r] = ok ? (((long long)n * a.H + ho) * a.W + wo) * a.in_ld + a.in_coff + lchunk * CE : -1;
                if (UP) pbase2[r] = ok ? (((long long)n * (a.H >> 1) + (ho >> 1)) * (a.W >> 1) + (wo >> 1)) * a.in2_ld + a.in2_coff + lchunk * CE : -1;
            } else if (ok) {
                phi0[r] = ho * a.stride - a.pad;
                pwi0[r] = wo * a.stride - a.pad;
                pbase[r] = (((long long)n * a.H + phi0[r]) * a.W + pwi0[r]) * a.in_ld + a.in_coff;
            } else {
                phi0[r] = -(1 << 28); // fails every bounds test below
                pwi0[r] = -(1 << 28);
                pbase[r] = 0;
            }
        }
        wslab = reinterpret_cast<const char *>(reinterpret_cast<const T *>(a.w) + (long long)n0 * a.Kpad);
        if constexpr (BUF) w_rs = make_rsrc(wslab);
        kc = lchunk * CE;
        tap = 0;
        if (!K1)
            while (kc >= a.Cin) {
                kc -= a.Cin;
                ++tap;
            }
        ld_ks = 0;
    };
    auto issue_stage = [&](char *pt) __attribute__((always_inline)) {
        char *wt = pt + BM * 128;
        if constexpr (BUF) {
            const unsigned so = (unsigned)(ld_ks * (BKE * (int)sizeof(T))); // wave-uniform K offset of the step
            if (K1) {
                const bool k_ok = kc < a.Cin; // K tail of the last step is zero
                if (UP && ld_ks * BKE < a.in2_split) { // block-uniform: in2_split is a multiple of the K step; low-resolution rows are shared by 4 pixels: no nt
#pragma unroll
                    for (int r = 0; r < PR; ++r) dma_buf<false>(in2_rs, poff2[r], so, pt + (RPP * r + 8 * wave) * 128);
                } else {
#pragma unroll
                    for (int r = 0; r < PR; ++r) dma_buf<NT>(in_rs, k_ok ? poff[r] : 0xffffffffu, so, pt + (RPP * r + 8 * wave) * 128);
                }
            } else {
                const int kh = a.KW == 3 ? (tap * 11) >> 5 : tap / a.KW, kw = tap - kh * a.KW; // tap/3 for tap < 32
                const bool tap_ok = tap < ntaps;
                const unsigned delta = (unsigned)((((kh * a.W + kw) * a.in_ld) + kc) * (int)sizeof(T));
#pragma unroll
                for (int r = 0; r < PR; ++r) {
                    const int hi = phi0[r] + kh, wi = pwi0[r] + kw;
                    const bool ok = tap_ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                    dma_buf<false>(in_rs, ok ? poff[r] + delta : 0xffffffffu, 0u, pt + (RPP * r + 8 * wave) * 128);
                }
            }
#pragma unroll
            for (int i = 0; i < WR; ++i) dma_buf<false>(w_rs, wvoff[i], so, wt + (RPP * i + 8 * wave) * 128);
        } else if (K1) {
            const bool k_ok = kc < a.Cin; // K tail of the last step is zero
            if (UP && ld_ks * BKE < a.in2_split) { // block-uniform: in2_split is a multiple of the K step
#pragma unroll
                for (int r = 0; r < PR; ++r) { // low-resolution rows are shared by 4 pixels: no non-temporal hint
                    const char *src = pbase[r] >= 0 ? reinterpret_cast<const char *>(in2 + pbase2[r] + ld_ks * BKE) : zero_page;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(pt + (RPP * r + 8 * wave) * 128), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < PR; ++r) {
                    const char *src = (k_ok && pbase[r] >= 0) ? reinterpret_cast<const char *>(in + pbase[r] + ld_ks * BKE) : zero_page;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(pt + (RPP * r + 8 * wave) * 128), 16, 0, NT ? 2 : 0);
                }
            }
        } else {
            const int kh = a.KW == 3 ? (tap * 11) >> 5 : tap / a.KW, kw = tap - kh * a.KW; // tap/3 for tap < 32
            const bool tap_ok = tap < ntaps;
            const long long delta = ((long long)kh * a.W + kw) * a.in_ld + kc;
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int hi = phi0[r] + kh, wi = pwi0[r] + kw;
                const bool ok = tap_ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const char *src = ok ? reinterpret_cast<const char *>(in + pbase[r] + delta) : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(pt + (RPP * r + 8 * wave) * 128), 16, 0, 0);
            }
        }
        if constexpr (!BUF) {
            const char *ub = wslab + (size_t)ld_ks * (BKE * sizeof(T));
#pragma unroll
            for (int i = 0; i < WR; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + wvoff[i]),
                                                 (__attribute__((address_space(3))) void *)(wt + (RPP * i + 8 * wave) * 128), 16, 0, 0);
        }
        // advance the loader; crossing into the next tile recomputes the row table
        kc += BKE;
        if (!K1)
            while (kc >= a.Cin) {
                kc -= a.Cin;
                ++tap;
            }
        if (++ld_ks == nk) {
            if (++ld_i < my_tiles) setup_loader(ld_i);
        }
    };

    // Accumulators START at the bias (the MFMA chain adds the products to it): saves one v_add per output value in the
    // epilogue, where the SiLU's VALU work — not the matrix pipe — bounds every layer with a short K (-2 % conv time).
    floatx4 acc[TC][TP];
    floatx4 acc1[SPLIT ? TC : 1][SPLIT ? TP : 1]; // split mode: the 2^-11 cross terms

    // fragment addresses: pixel tiles j are base + j*2048 (same swizzle key), cout tiles i are
    // base + i*512 (key independent of i); the second k-half is base ^ 64.
    const int prow_l = wave_p * WP + lr;
    const unsigned pfrag0 = prow_l * 128 + ((lg ^ (prow_l & 7)) << 4);
    const int wrow_l = wave_c * WC + (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = BM * 128 + wrow_l * 128 + ((lg ^ wkey_l) << 4);

    auto compute_half = [&](const char *pt, int kh2) __attribute__((always_inline)) {
        {
            const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0;
            const unsigned wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
            uint4 pf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(pt + pa + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(pt + wa + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_frag(wf[i], pf[j], acc[i][j], (T *)nullptr);
        }
    };

    // split mode: one K step = hi fragments (k-half 0) and lo fragments (k-half 1) of the same 32 channels
    auto compute_split = [&](const char *pt) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
            uint4 ph[TP], wh[TC], wl[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) ph[j] = *reinterpret_cast<const uint4 *>(pt + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wh[i] = *reinterpret_cast<const uint4 *>(pt + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) wl[i] = *reinterpret_cast<const uint4 *>(pt + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    mma_frag(wh[i], ph[j], acc[i][j], (T *)nullptr);
                    mma_frag(wl[i], ph[j], acc1[i][j], (T *)nullptr);
                }
            uint4 pl[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) pl[j] = *reinterpret_cast<const uint4 *>(pt + (pfrag0 ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_frag(wh[i], pl[j], acc1[i][j], (T *)nullptr);
        }
    };

    // bias of the NEXT tile to be multiplied (rows exist up to CoutPad): requested at the top of the current tile's
    // epilogue, consumed at its end when the accumulators are re-armed — the L2 round trip hides behind the SiLU work
    float bias_r[NV];
    auto load_bias = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * BN + wave_c * WC + lg * NV;
#pragma unroll
        for (int e = 0; e < NV; ++e) bias_r[e] = a.bias[cb + e];
    };
    auto arm_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                acc[i][j] = (floatx4){bias_r[i * 4 + 0], bias_r[i * 4 + 1], bias_r[i * 4 + 2], bias_r[i * 4 + 3]};
                if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
    };
    load_bias(0);
    arm_acc();

    // fused tail: this wave's slice of the tail weights (64 couts x 128 k as 4 k-steps x 4 cout tiles of MFMA A fragments)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 w2f[TAIL ? 4 : 1][TAIL ? 4 : 1];
    if constexpr (TAIL) {
        static_assert(sizeof(T) == 2 && BM == 128 && BN == 128 && WAVES_P == 2 && WAVES_C == 2 && TP == 4 && TC == 4, "fused tail: fp16, 128x128 tile");
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.tail_w);
        const int arow = wave_c * 64 + (lr >> 2) * 16 + (lr & 3); // + 4t: the lane ends up owning couts wave_c*64 + lg*16 .. +15
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) w2f[ks][t] = *reinterpret_cast<const h8 *>(w2 + (long long)(arow + 4 * t) * a.tail_kpad + ks * 32 + lg * 8);
    }

    // ---- epilogue of one finished tile: lane (pixel lr of tile j, group lg) owns couts cb .. cb+NV-1
    T *out = reinterpret_cast<T *>(a.out);
    T *out2 = reinterpret_cast<T *>(a.out2);
    const T *res = reinterpret_cast<const T *>(a.res);
    auto epilogue = [&](int i, char *cur) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * BN + wave_c * WC + lg * NV;
        if (nct > 1 && i + 1 < my_tiles) load_bias(i + 1); // with one cout tile every tile has the same bias
        if constexpr (TAIL) {
            // ---- fused 1x1 tail (128 -> 128).  A wave holds 64 pixels x ONE HALF of this conv's channels: the two cout-waves of a
            // pixel group exchange their SiLU'd fp16 tiles through the stage buffer that was just multiplied, then each of them
            // computes ITS half of the tail's couts for the group's 64 pixels over all 128 channels (k-steps 0,1 from the low-channel
            // tile, 2,3 from the high one: the K order of the stand-alone 1x1 kernel, same fp16 rounding of the intermediate:
            // bit-identical).  The wave's 64 x 128 slice of the tail weights stays in 64 VGPRs for the whole kernel (w2f): fetching
            // it per tile cost as much as the stand-alone 1x1 launch it replaces (a vector load holds its wave ~110 cycles).
            if constexpr (BUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // every wave is done reading `cur`
            char *mine = cur + wave * 8192;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = j * 16 + lr;
                float v[NV];
#pragma unroll
                for (int t = 0; t < TC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][j][r];
                if (a.act) wtk_silu_scaled_run<NV, (WTK_SILU_SCALAR_MASK & 2) != 0>(v);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    h8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[c2 * 8 + e];
                    *reinterpret_cast<h8 *>(mine + p * 128 + (((2 * lg + c2) ^ (p & 7)) << 4)) = hv;
                }
            }
            __syncthreads();
            {
                const float4 *bp = reinterpret_cast<const float4 *>(a.tail_bias + wave_c * 64 + lg * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 b = bp[t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = (floatx4){b.x, b.y, b.z, b.w};
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char *src = cur + (wave_p * 2 + (ks >> 1)) * 8192;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = j * 16 + lr;
                    const h8 pf = *reinterpret_cast<const h8 *>(src + p * 128 + ((((ks & 1) * 4 + lg) ^ (p & 7)) << 4));
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ks][t], pf, acc[t][j], 0, 0, 0);
                }
            }
            _Float16 *tout = reinterpret_cast<_Float16 *>(a.tail_out);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int n, ho, wo;
                if (!pixel_coords(ptile, wave_p * WP + j * 16 + lr, n, ho, wo)) continue;
                const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
                float v2[NV];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v2[t * 4 + r] = acc[t][j][r];
                if (a.tail_act) wtk_silu_scaled_run<NV>(v2);
                store_run<NV>(tout + pix * a.tail_ld + a.tail_coff + wave_c * 64 + lg * 16, v2);
            }
            arm_acc();
            return;
        }
        if (cb + NV <= a.Cout) { // padded output channels (Cout < CoutPad) are never stored
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int n = 0, ho = 0, wo = 0;
                long long pix;
                if (a.tile_w == 0 && !out2) { // linear pixel range: the output pixel index is m itself
                    pix = (long long)ptile * BM + wave_p * WP + j * 16 + lr;
                    if (pix >= a.M) continue;
                } else {
                    if (!pixel_coords(ptile, wave_p * WP + j * 16 + lr, n, ho, wo)) continue;
                    pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
                }
                float v[NV];
#pragma unroll
                for (int t = 0; t < TC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (SPLIT)
                            v[t * 4 + r] = wtk_split_value(acc[t][j][r], acc1[t][j][r]); // bias already inside acc
                        else
                            v[t * 4 + r] = acc[t][j][r]; // bias already inside
                    }
                if (a.act) {
                    wtk_silu_scaled_run<NV, (WTK_SILU_SCALAR_MASK & 2) != 0>(v);
                }
                if (res) {
                    float rv[NV];
                    if constexpr (SPLIT)
                        wtk_split_load<NV>(reinterpret_cast<const _Float16 *>(a.res) + pix * a.res_ld + a.res_coff, cb, rv);
                    else
                        load_run<NV>(res + pix * a.res_ld + a.res_coff + cb, rv);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] += rv[e];
                }
                if (sizeof(T) == 2 && a.out_f32)
                    store_run<NV>(reinterpret_cast<float *>(a.out) + pix * a.out_ld + a.out_coff + cb, v); // head logits stay fp32
                else if constexpr (SPLIT)
                    wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out) + pix * a.out_ld + a.out_coff, cb, v);
                else
                    store_run<NV>(out + pix * a.out_ld + a.out_coff + cb, v);
                if (out2) {
                    const int Ho2 = a.Ho * 2, Wo2 = a.Wo * 2;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const long long pix2 = ((long long)n * Ho2 + (2 * ho + dy)) * Wo2 + (2 * wo + dx);
                            if constexpr (SPLIT)
                                wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out2) + pix2 * a.out2_ld + a.out2_coff, cb, v);
                            else
                                store_run<NV>(out2 + pix2 * a.out2_ld + a.out2_coff + cb, v);
                        }
                }
            }
        }
        arm_acc();
    };

    // ---- flat pipeline over (tile, K step) stages: stage s+1 is in flight (LDS-DMA) while stage s is
    // multiplied and, at a tile's last K step, its epilogue runs — so the next tile's loads hide behind it.
    const int total_stages = my_tiles * nk;
    int cp_ks = 0, cp_i = 0;
    auto after_compute = [&](char *cur) __attribute__((always_inline)) {
        if (++cp_ks == nk) {
            epilogue(cp_i, cur);
            cp_ks = 0;
            ++cp_i;
        }
    };
    setup_loader(0);
    issue_stage(smem0);
    if constexpr (BUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // drains the LDS-DMA (vmcnt(0)) and publishes the tile
#ifdef WTK_IGEMM_STAMPS // diagnostic builds (tools/igemm_stamps_split.hip): per-wave cycle totals of a K step's request issue, multiply (+ epilogue), vmcnt wait, barrier wait
    unsigned long long st_sum[4] = {0, 0, 0, 0};
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define WTK_ST(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += t_ - st_prev; st_prev = t_; }
    unsigned long long st_prev = st_c0;
#else
#define WTK_ST(i)
#endif
    for (int s = 0; s < total_stages; s += 2) {
#if WTK_IGEMM_ORDER == 1
        compute_half(smem0, 0);
        if (s + 1 < total_stages) issue_stage(smem1); // smem1 was last read before the previous barrier
        compute_half(smem0, 1);
#else
        if (s + 1 < total_stages) issue_stage(smem1); // smem1 was last read before the previous barrier
        WTK_ST(0)
        if constexpr (SPLIT) {
            compute_split(smem0);
        } else {
            compute_half(smem0, 0);
            compute_half(smem0, 1);
        }
#endif
        after_compute(smem0);
#ifdef WTK_IGEMM_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        WTK_ST(1)
        if constexpr (BUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WTK_ST(2)
        __syncthreads();
        WTK_ST(3)
        if (s + 1 >= total_stages) break;
#if WTK_IGEMM_ORDER == 1
        compute_half(smem1, 0);
        if (s + 2 < total_stages) issue_stage(smem0);
        compute_half(smem1, 1);
#else
        if (s + 2 < total_stages) issue_stage(smem0);
        WTK_ST(0)
        if constexpr (SPLIT) {
            compute_split(smem1);
        } else {
            compute_half(smem1, 0);
            compute_half(smem1, 1);
        }
#endif
        after_compute(smem1);
#ifdef WTK_IGEMM_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        WTK_ST(1)
        if constexpr (BUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WTK_ST(2)
        __syncthreads();
        WTK_ST(3)
    }
#ifdef WTK_IGEMM_STAMPS
    if (lane == 0 && a.dbg_stamps) {
        unsigned long long *o = a.dbg_stamps + ((long long)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 4; ++i) o[i] = st_sum[i];
        o[4] = (unsigned long long)total_stages;
        o[5] = __builtin_amdgcn_s_memtime() - st_c0;
        o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
#undef WTK_ST
}

int conv_cfg_bm(int cfg) { return (cfg == CFG_128x128 || cfg == CFG_128x64) ? 128 : 256; }
int conv_cfg_bn(int cfg) { return cfg == CFG_128x128 ? 128 : ((cfg == CFG_256x64 || cfg == CFG_128x64) ? 64 : 32); }

hipError_t conv_init_attributes() { return hipSuccess; } // LDS is static: nothing to raise

template <typename T, int BM, int BN, int WAVES_P, int WAVES_C, bool SPLIT = false>
static hipError_t launch_t(ConvArgs a, hipStream_t stream) {
    long long ptiles;
    if (a.tile_w == 0)
        ptiles = (a.M + BM - 1) / BM;
    else
        ptiles = (long long)a.N * a.tiles_x * a.tiles_y;
    const long long tiles = ptiles * (a.CoutPad / BN);
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    a.ptiles = (int)ptiles;
    if (a.M > 0x7fffffffLL) return hipErrorInvalidValue;
    a.d_howo = make_fastdiv((unsigned)(a.Ho * a.Wo));
    a.d_wo = make_fastdiv((unsigned)a.Wo);
    a.d_tilew = make_fastdiv((unsigned)(a.tile_w > 0 ? a.tile_w : 1));
    a.d_tilesx = make_fastdiv((unsigned)(a.tiles_x > 0 ? a.tiles_x : 1));
    a.d_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y > 0 ? a.tiles_x * a.tiles_y : 1));
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / BN));
    const int g_num_cus = current_device_cus();
    if (g_num_cus <= 0) return hipErrorUnknown;
    constexpr int NW = WAVES_P * WAVES_C;
    const long long per_cu = NW == 8 ? 1 : ((2 * (BM + BN) * 128 <= 52 * 1024) ? 3 : 2); // LDS-limited residency
    const long long resident = per_cu * g_num_cus;
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    const bool k1 = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
    if (k1 && (a.Ho != a.H || a.Wo != a.W)) return hipErrorInvalidValue; // the 1x1 loader reads input pixel m for output pixel m
    if (a.tail_w) {
        if constexpr (BM == 128 && BN == 128 && sizeof(T) == 2 && !SPLIT) {
            if (k1 || a.in2 || a.res || a.out2 || a.Cout != BN || a.CoutPad != BN || !a.tail_bias || !a.tail_out || a.tail_kpad < BN || a.tail_kpad % 8 || a.tail_ld % 8 ||
                a.tail_coff % 8)
                return hipErrorInvalidValue; // built for the strided 3x3 -> 1x1 (128 -> 128) pair only
            hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, false, false, false, true>), dim3(grid), dim3(64 * NW), 0, stream, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (a.in2) {
        if constexpr (BM == 128 && BN == 128) {
            if (!k1) return hipErrorInvalidValue;
            if (a.CoutPad == BN)
                hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, true, true, true, false, SPLIT>), dim3(grid), dim3(64 * NW), 0, stream, a);
            else
                hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, true, false, true, false, SPLIT>), dim3(grid), dim3(64 * NW), 0, stream, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue; // the two-source loader is only built for the 128x128 tile
        }
    }
    if (k1 && a.CoutPad == BN)
        hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, true, true, false, false, SPLIT>), dim3(grid), dim3(64 * NW), 0, stream, a);
    else if (k1)
        hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, true, false, false, false, SPLIT>), dim3(grid), dim3(64 * NW), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WAVES_P, WAVES_C, false, false, false, false, SPLIT>), dim3(grid), dim3(64 * NW), 0, stream, a);
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
    if (a.in2 && (a.in2_split <= 0 || a.in2_split % (8 * ce) != 0 || a.in2_split > a.Cin || a.in2_ld % ce != 0 || a.in2_coff % ce != 0 ||
                  a.H % 2 != 0 || a.W % 2 != 0 || cfg != CFG_128x128))
        return hipErrorInvalidValue;
    if (is_f16) {
        switch (cfg) {
        case CFG_128x128: return launch_t<_Float16, 128, 128, 2, 2>(a, stream);
        case CFG_256x64: return launch_t<_Float16, 256, 64, 4, 1>(a, stream);
        case CFG_256x32: return launch_t<_Float16, 256, 32, 4, 1>(a, stream);
        case CFG_128x64: return launch_t<_Float16, 128, 64, 4, 1>(a, stream);
        }
    } else {
        switch (cfg) {
        case CFG_128x128: return launch_t<float, 128, 128, 2, 2>(a, stream);
        case CFG_256x64: return launch_t<float, 256, 64, 4, 1>(a, stream);
        case CFG_256x32: return launch_t<float, 256, 32, 4, 1>(a, stream);
        case CFG_128x64: return launch_t<float, 128, 64, 4, 1>(a, stream);
        }
    }
    return hipErrorInvalidValue;
}

// Split-fp16 operands: the fp16 instantiations with SPLIT = true; every channel-like argument of `a` is in pseudo-channels (see ConvArgs)
hipError_t launch_conv_split(const ConvArgs &a, int cfg, hipStream_t stream) {
    const int bn = conv_cfg_bn(cfg), bm = conv_cfg_bm(cfg);
    if (a.CoutPad % bn != 0 || a.Cout > a.CoutPad || a.Cout % 8 != 0 || a.Cin % 64 != 0 || a.in_ld % 64 != 0 || a.in_coff % 64 != 0) return hipErrorInvalidValue;
    if (a.Kpad % 64 != 0 || a.Kpad < a.K || a.K != a.KH * a.KW * a.Cin || a.tail_w) return hipErrorInvalidValue;
    if (!a.out_f32 && (a.out_ld % 64 != 0 || a.out_coff % 64 != 0)) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % 64 != 0 || a.res_coff % 64 != 0)) return hipErrorInvalidValue;
    if (a.out2 && (a.out2_ld % 64 != 0 || a.out2_coff % 64 != 0)) return hipErrorInvalidValue;
    if (a.tile_w != 0 && (bm % a.tile_w != 0)) return hipErrorInvalidValue;
    if (a.in2 && (a.in2_split <= 0 || a.in2_split % 64 != 0 || a.in2_split > a.Cin || a.in2_ld % 64 != 0 || a.in2_coff % 64 != 0 || a.H % 2 != 0 || a.W % 2 != 0 ||
                  cfg != CFG_128x128))
        return hipErrorInvalidValue;
    switch (cfg) {
    case CFG_128x128: return launch_t<_Float16, 128, 128, 2, 2, true>(a, stream);
    case CFG_256x64: return launch_t<_Float16, 256, 64, 4, 1, true>(a, stream);
    case CFG_256x32: return launch_t<_Float16, 256, 32, 4, 1, true>(a, stream);
    case CFG_128x64: return launch_t<_Float16, 128, 64, 4, 1, true>(a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace wtk
