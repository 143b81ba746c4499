// 3x3 / stride-1 convolution with an LDS-resident input window ("halo") on CDNA4 matrix cores.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-stages the im2col pixel operand for each of the
// 9 taps: 9x the L2->LDS traffic and 9x the LDS-DMA instructions of the input it actually needs, and
// on MI355X the DMA issue cost (~60-180 cycles per 1-KiB piece) then rivals the MFMA time.  Here the
// block's input window is staged ONCE per 64-channel chunk and the 9 taps read it at shifted offsets:
//
//   * the image is walked in column strips of S <= 85 px; inside a strip, pixels of the zero-padded
//     window (pitch = S + 2 columns) are numbered flat, o = y*pitch + x, so the input of output o for
//     tap (kh, kw) is simply window[o + kh*pitch + kw]: a 1-D shift.  A block owns 256 consecutive flat
//     outputs (the border columns are junk and are dropped in the epilogue); its window is the
//     contiguous run of 256 + 2*pitch + 2 window pixels, each 128 bytes (64 fp16 / 32 fp32 channels);
//   * the N images of a strip are STACKED vertically with one shared zero row between neighbours (the row
//     below image n is the row above image n+1), so the flat index runs over N*(H+1) rows and a block
//     boundary need not fall on an image boundary: a 20x20 map no longer rounds 440 outputs up to two
//     256-pixel blocks per image.  A map that fits one strip also shares ONE zero column between the
//     right border of a row and the left border of the next (pitch = W + 1).  Together: 13 % fewer
//     blocks on 20x20 maps, 6 % on 40x40 (time follows the executed MFMAs once two forward passes share
//     the chip);
//   * an MFMA pixel tile is 16 consecutive flat outputs = 16 consecutive window rows, so with the
//     row&7 XOR swizzle every ds_read_b128 fragment read is conflict-free for ANY tap offset;
//   * per tap only the [BN][64ch] weight slab is streamed (double buffered, LDS-DMA); the next channel
//     chunk's window is prefetched one 1-KiB piece per wave per tap underneath the MFMAs;
//   * 8 waves / block (2 per SIMD); wave tile 64 px x 64 cout (BN = 128) or 32 px x 64 cout (BN = 64);
//   * all four LDS buffers are distinct objects so hipcc's waitcnt pass does not drain the in-flight
//     LDS-DMA before each fragment read (see conv_igemm.hip).
#include "wtk_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

template <typename T> struct ElemH;
template <> struct ElemH<_Float16> {
    static constexpr int CE = 8;
};
template <> struct ElemH<float> {
    static constexpr int CE = 4;
};

// SiLU with two transcendentals and three plain VALU ops (v_mul, v_exp, v_add, v_rcp, v_mul).  The obvious
// x / (1 + __expf(-x)) expands to ~35 instructions (IEEE division + range-checked exp) and made the
// epilogue, not the MFMA loop, the longest part of every conv.  v_exp/v_rcp are 1-ulp approximations.
__device__ __forceinline__ float silu_h(float x) {
    return wtk_silu_scaled(x); // x is the log2(e)-scaled pre-activation (wtk_kernels.h)
}

__device__ __forceinline__ void mma_h(const uint4 &wf, const uint4 &pf, floatx4 &acc, _Float16 *) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wf), __builtin_bit_cast(half8, pf), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_h(const uint4 &wf, const uint4 &pf, floatx4 &acc, float *) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.x), __builtin_bit_cast(float, pf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.y), __builtin_bit_cast(float, pf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.z), __builtin_bit_cast(float, pf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf.w), __builtin_bit_cast(float, pf.w), acc, 0, 0, 0);
}

template <int NV> __device__ __forceinline__ void load_run_h(const _Float16 *p, float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        half8 h = *reinterpret_cast<const half8 *>(p + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i + j] = (float)h[j];
    }
}
template <int NV> __device__ __forceinline__ void load_run_h(const float *p, float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        float4 f = *reinterpret_cast<const float4 *>(p + i);
        v[i] = f.x, v[i + 1] = f.y, v[i + 2] = f.z, v[i + 3] = f.w;
    }
}
template <int NV> __device__ __forceinline__ void store_run_h(_Float16 *p, const float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        half8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[i + j];
        *reinterpret_cast<half8 *>(p + i) = h;
    }
}
template <int NV> __device__ __forceinline__ void store_run_h(float *p, const float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) *reinterpret_cast<float4 *>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}

constexpr int kBM = 256;

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at the wave-uniform LDS address).  RAW = true issues it from inline asm:
// hipcc then does not know an LDS write is pending and inserts no vmcnt wait of its own in front of later ds_reads —
// the three-slab schedule orders every read behind an explicit counted wait + barrier instead.  (With the builtin, the
// waitcnt pass tracks pending LDS-DMA per LDS object; once a few are in flight it gives up counting and drains with
// vmcnt(0) before the first fragment read of a slab, which is exactly the in-flight request the schedule relies on.)
template <bool RAW> __device__ __forceinline__ void lds_dma16(const char *src, char *lds_dst) {
    if constexpr (RAW) {
        const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
    } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
    }
}

// The same request in buffer form: SGPR resource (base, huge range) + wave-uniform byte offset + per-lane 32-bit offset.  Measured
// 5-10 % less wave time per request than the flat form (tools/lds_dma_rate.hip: 108 vs 120 cycles), no 64-bit address arithmetic
// per request, and a lane whose offset is 0xffffffff is out of range and lands ZEROS: no zero-page select for padding rows.
#ifndef WTK_HALO_BUFFER_DMA
#define WTK_HALO_BUFFER_DMA 1
#endif
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *base) {
    const unsigned long long b = (unsigned long long)base;
    rsrc_t r;
    r.x = (int)(unsigned)(b & 0xffffffffu);
    r.y = (int)(unsigned)((b >> 32) & 0xffffu); // stride 0: raw buffer
    r.z = (int)0xffffff00u;                     // num_records (bytes): everything a 32-bit offset can reach except the "invalid" marker
    r.w = 0x00020000;                           // DATA_FORMAT = 32-bit (gfx9 family raw-buffer word 3)
    return r;
}
__device__ __forceinline__ void lds_dma16_buf(const rsrc_t &rs, unsigned voff, unsigned soff, char *lds_dst) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Stacked geometry (header comment).  Window row `flat` (relative to the strip) -> input pixel: stacked row rho = flat / pitch
// holds image n = rho / (H+1), input row iy = rho % (H+1) - 1 (-1: the shared zero row); column ix = xs + flat % pitch - 1.
__device__ __forceinline__ bool halo_in_coords(const HaloArgs &a, int flat, int xs, int &n, int &iy, int &ix) {
    const int rho = (int)fdiv((unsigned)flat, a.d_pitch);
    const int cc = flat - rho * a.pitch;
    n = (int)fdiv((unsigned)rho, a.d_h1);
    iy = rho - n * (a.H + 1) - 1;
    ix = xs + cc - 1;
    return n < a.N && iy >= 0 && (unsigned)ix < (unsigned)a.W;
}
// Flat output index -> (image, row, column inside the strip); false for the junk row / junk columns / past the last image
__device__ __forceinline__ bool halo_out_coords(const HaloArgs &a, int o, int xs, int &n, int &y, int &x) {
    const int q = (int)fdiv((unsigned)o, a.d_pitch);
    x = o - q * a.pitch;
    n = (int)fdiv((unsigned)q, a.d_h1);
    y = q - n * (a.H + 1);
    return n < a.N && y < a.H && x < a.S && xs + x < a.W;
}

// Per-lane byte offsets of a wave's window pieces (wave w stages pieces w, w+8, ...: 8 rows x 8 chunks of 16 B each).  The 8 x KMAX
// rows of a wave are evaluated ONCE — lane L works out row L&7 of piece L>>3 — and handed to the lanes that need them with
// ds_bpermute, instead of every lane redoing the divisions for each of its pieces: the ~250 VALU instructions this took per
// block sat in front of the block's first LDS-DMA request (stamped: 0.5 us of a 10-18 us block).
template <typename T, int KMAX>
__device__ __forceinline__ void halo_piece_offsets(const HaloArgs &a, int o0, int xs, int n_base, int halo_rows, int wave, int lane, unsigned (&hoff)[KMAX],
                                                   unsigned &hvalid) {
    static_assert(KMAX <= 8, "one lane per (piece, row)");
    constexpr int CE = ElemH<T>::CE;
    const int hr_e = (wave + 8 * (lane >> 3)) * 8 + (lane & 7);
    int pn, iy, ix;
    const bool ok_e = halo_in_coords(a, o0 + hr_e, xs, pn, iy, ix) && hr_e < halo_rows;
    const unsigned row_e = ok_e ? (unsigned)(((((long long)(pn - n_base) * a.H + iy) * a.W + ix) * a.in_ld) * (long long)sizeof(T)) : 0xffffffffu;
    const unsigned lc_term = (unsigned)((((lane & 7) ^ ((lane >> 3) & 7)) * CE) * (int)sizeof(T)); // logical chunk landing on this lane's slot
    hvalid = 0;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
        const unsigned v = (unsigned)__builtin_amdgcn_ds_bpermute((q * 8 + (lane >> 3)) * 4, (int)row_e);
        const bool ok = v != 0xffffffffu;
        hoff[q] = ok ? v + lc_term : 0u;
        hvalid |= ok ? (1u << q) : 0u;
    }
}
// Output pixel of a wave's flat outputs o_first + L (L < 64), evaluated once per lane: pixel index (n*H + y)*W + xs + x, or -1 for
// junk rows / columns; `col` = xs + x.  The lanes of pixel tile j fetch theirs with ds_bpermute from lane j*16 + (lane & 15).
__device__ __forceinline__ void halo_out_pixel(const HaloArgs &a, int o_first, int xs, int lane, int &pix_e, int &col_e) {
    int n, y, x;
    const bool ok = halo_out_coords(a, o_first + lane, xs, n, y, x);
    col_e = xs + x;
    pix_e = ok ? (n * a.H + y) * a.W + col_e : -1;
}
__device__ __forceinline__ int lane_fetch(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane * 4, v); }

// HROWS: window rows one LDS buffer holds.  NWB: weight slabs in the ring (2, 3, or 6: the split 64-cout x 128-pixel tile of small handles, see the
// tap loop).  With NWB == 3 the slab of tap g+2 is
// requested while tap g is multiplied and a COUNTED s_waitcnt vmcnt leaves it in flight across the tap barrier
// (raw s_barrier): a slab has two full taps to arrive instead of one.  PMC on the two-slab kernel showed every wave
// waiting ~1/3 of its life in the vmcnt(0) that __syncthreads puts in front of each tap barrier (1 block per CU:
// nothing else hides the L2 latency of the slab requested at the top of the same tap).
// BMT: flat output pixels per block, 256 or 128 (small maps: twice the blocks, so a 20x20 map still fills the chip).
// TAIL: instantiation with the fused 1x1 tail (see the epilogue); a separate instantiation so that its extra registers do
// not touch the plain variant's allocation (the 64-cout variant lives at 2 blocks per CU = 128 VGPRs).
// SPLIT (T = fp16): split-fp16 operands (wtk_kernels.h, kSplitScale): a 128-byte row is 32 channels as [hi32 | lo32]; per tap and tile pair
// three MFMAs (hi*hi into acc, hi*lo + lo*hi into acc1); a.Cin / in_ld / out_ld / ... are pseudo-channel counts (2 x real), a.Cout is real.
template <typename T, int BN, int NHALO, int MINW, int NWB, int HROWS, int BMT = 256, bool TAIL = false, bool SPLIT = false>
__global__ __launch_bounds__(512, MINW) void conv3x3_halo_kernel(const HaloArgs a) {
    static_assert(!SPLIT || (sizeof(T) == 2 && BN != 192 && (MINW <= 2 || (BN == 64 && BMT == 128 && NHALO == 1))),
                  "split mode: fp16 storage, 64 / 128 couts, 256-register budget (128 for the two-blocks-per-CU form: 64 couts x 128 pixels, one window buffer)");
#ifdef WTK_HALO_STAMPS // diagnostic builds only: block start / main-loop start / main-loop end / block end, 100 MHz clock
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Every kernel argument the set-up needs, requested in ONE batch: hipcc otherwise loads them lazily in 4-5 dependent rounds of
    // s_load + s_waitcnt (~0.2 us each) in front of the block's first LDS-DMA request.
    asm volatile("" ::"s"(a.in), "s"(a.w), "s"(a.bias), "s"(a.zeros), "s"(a.in_ld), "s"(a.in_coff), "s"(a.N), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CoutPad),
                 "s"(a.Kpad), "s"(a.S), "s"(a.pitch), "s"(a.strips), "s"(a.d_strips.mul), "s"(a.d_strips.sh1), "s"(a.d_strips.sh2), "s"(a.d_pitch.mul),
                 "s"(a.d_pitch.sh1), "s"(a.d_pitch.sh2), "s"(a.d_nct.mul), "s"(a.d_nct.sh1), "s"(a.d_nct.sh2), "s"(a.d_h1.mul), "s"(a.d_h1.sh1), "s"(a.d_h1.sh2), "s"(a.grid));
    constexpr int CE = ElemH<T>::CE;
    constexpr int CCH = 8 * CE; // channels per 128-byte chunk
    // BN = 64: 8(P) x 1(C) waves of 32 px x 64 cout; BN = 128 / 192: 4(P) x 2(C) waves of 64 px x 64 / 96 cout
    constexpr int WAVES_C = BN == 64 ? 1 : 2, WAVES_P = 8 / WAVES_C;
    constexpr int WC = BN / WAVES_C;
    constexpr int WP = BMT / WAVES_P, TP = WP / 16, TC = WC / 16, NV = 4 * TC;
    constexpr int WR = BN / 64; // weight rows staged per thread per tap

    constexpr int kHaloBytesT = HROWS * 128;
    __shared__ __attribute__((aligned(16))) char halo0[kHaloBytesT];
    __shared__ __attribute__((aligned(16))) char halo1[NHALO == 2 ? kHaloBytesT : 16];
    static_assert(NWB == 2 || NWB == 3 || (NWB == 6 && NHALO == 2 && !TAIL && SPLIT), "slab ring: 2, 3 or 6 (split tiles, two window buffers, no fused tail)");
    constexpr bool kRing = NWB >= 3; // counted waits, raw LDS-DMA requests
    __shared__ __attribute__((aligned(16))) char wbuf0[NWB > 3 ? 16 : BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf1[NWB > 3 ? 16 : BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf2[NWB == 3 ? BN * 128 : 16];
    __shared__ __attribute__((aligned(16))) char wring[NWB > 3 ? NWB * BN * 128 : 16]; // the six-slab ring: slot = global tap index % 6

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- block -> (cout tile, row block, strip, image); XCD-aware bijective remap
    const int nct = a.CoutPad / BN;
    int nwg = a.grid;
    if (a.n_dyn) { // dynamic batch: only the row blocks that start inside the first *n_dyn images exist; the remap runs over them, so
                   // the surviving tiles stay spread over all XCDs (block-uniform exit for the rest)
        const int lim = min(max(*a.n_dyn, 0), a.N) * (a.H + 1) * a.pitch;
        const int live = ((lim + BMT - 1) / BMT) * a.strips * nct;
        if ((int)blockIdx.x >= live) return;
        nwg = min(nwg, live);
    }
    int L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // launch-invariant divisors go through FastDiv: a runtime integer division is ~40 instructions, and the ~25 of them this
    // kernel used to execute before its first LDS-DMA request cost every block 1.3-1.9 us (stamped) of an 11-33 us life
    const unsigned t = fdiv((unsigned)L, a.d_nct);
    const int n0 = (L - (int)t * nct) * BN;
    const int rb = (int)fdiv(t, a.d_strips); // row blocks major, strips minor: the strips of a row block share input rows (L2)
    const int strip = (int)t - rb * a.strips;
    const int o0 = rb * BMT;
    const int xs = strip * a.S;
    const int pitch = a.pitch;
    const int halo_rows = BMT + 2 * pitch + 2;
    const int halo_pieces = (halo_rows + 7) >> 3;
    // piece offsets are 32-bit and relative to the first image the window touches
    const int n_base = (int)fdiv(fdiv((unsigned)o0, a.d_pitch), a.d_h1);

    const T *in = reinterpret_cast<const T *>(a.in) + (long long)n_base * a.H * a.W * a.in_ld + a.in_coff;
    const T *wgt = reinterpret_cast<const T *>(a.w);
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);

    // ---- loop-invariant per-lane addressing (the inner loop must stay almost VALU-free: a wave64 VALU op
    // costs ~4 issue cycles against 16 per MFMA, so a few dozen address instructions per tap starve the
    // matrix pipe).
    // Window pieces: wave w stages pieces w, w+8, ... (<= kMaxPiecesPerWave); the window geometry does not
    // depend on the channel chunk, so each piece's per-lane byte offset inside the image is computed once.
    constexpr int kMaxPiecesPerWave = (HROWS / 8 + 7) / 8;
    unsigned hoff[kMaxPiecesPerWave];
    unsigned hvalid;
    halo_piece_offsets<T, kMaxPiecesPerWave>(a, o0, xs, n_base, halo_rows, wave, lane, hoff, hvalid);
    const char *img = reinterpret_cast<const char *>(in);
    auto issue_halo_piece = [&](char *buf, int q, int c) { // q static after unrolling
        const int piece = wave + 8 * q;
        if (piece >= halo_pieces) return; // wave-uniform
        if constexpr (kRing && WTK_HALO_BUFFER_DMA) {
            lds_dma16_buf(make_rsrc(img), ((hvalid >> q) & 1u) ? hoff[q] : 0xffffffffu, (unsigned)(c * (CCH * (int)sizeof(T))), buf + piece * 1024);
        } else {
            const char *src = ((hvalid >> q) & 1u) ? img + (size_t)c * (CCH * sizeof(T)) + hoff[q] : zero_page;
            lds_dma16<kRing>(src, buf + piece * 1024);
        }
    };
    // Same, but never skipped (see the three-slab schedule below).  Pieces past the window rows carry zeros into unused
    // rows; a piece index past the BUFFER (only the last q of the highest waves) re-requests the wave's previous piece.
    auto issue_halo_piece_always = [&](char *buf, int q, int c) __attribute__((always_inline)) { // q static after unrolling
        constexpr int kPieces = HROWS / 8;
        static_assert(kPieces >= 16, "window buffer too small");
        const bool back = q > 0 && wave + 8 * q >= kPieces; // wave-uniform
        const int piece = back ? wave + 8 * (q - 1) : wave + 8 * q;
        const unsigned off = back ? hoff[q > 0 ? q - 1 : 0] : hoff[q];
        const bool ok = back ? ((hvalid >> (q > 0 ? q - 1 : 0)) & 1u) : ((hvalid >> q) & 1u);
        if constexpr (kRing && WTK_HALO_BUFFER_DMA) {
            lds_dma16_buf(make_rsrc(img), ok ? off : 0xffffffffu, (unsigned)(c * (CCH * (int)sizeof(T))), buf + piece * 1024);
        } else {
            const char *src = ok ? img + (size_t)c * (CCH * sizeof(T)) + off : zero_page;
            lds_dma16<kRing>(src, buf + piece * 1024);
        }
    };
    // Weight slab of (tap, chunk c): rows = couts n0 .. n0+BN, 128 bytes each.  Uniform base + invariant
    // per-lane 32-bit offset (lets the compiler use the SGPR-base form of global_load_lds).
    const int wrow0 = tid >> 3, wp = tid & 7;
    unsigned wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = wrow0 + 64 * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wvoff[i] = (unsigned)(((long long)row * a.Kpad + (wp ^ key) * CE) * (long long)sizeof(T));
    }
    const char *wtile = reinterpret_cast<const char *>(wgt + (long long)n0 * a.Kpad);
    auto issue_weights = [&](char *buf, int tap, int c) {
        if constexpr (kRing && WTK_HALO_BUFFER_DMA) {
            const rsrc_t rs = make_rsrc(wtile);
            const unsigned so = (unsigned)((tap * a.Cin + c * CCH) * (int)sizeof(T)); // wave-uniform
#pragma unroll
            for (int i = 0; i < WR; ++i) lds_dma16_buf(rs, wvoff[i], so, buf + (64 * i + 8 * wave) * 128);
        } else {
            const char *ub = wtile + ((size_t)tap * a.Cin + (size_t)c * CCH) * sizeof(T); // wave-uniform
#pragma unroll
            for (int i = 0; i < WR; ++i) lds_dma16<kRing>(ub + wvoff[i], buf + (64 * i + 8 * wave) * 128);
        }
    };

    floatx4 acc[TC][TP];
    floatx4 acc1[SPLIT ? TC : 1][SPLIT ? TP : 1]; // split mode: the 2^-11 cross terms
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }

    // weight fragments: row(i) = wave_c*64 + (lr>>2)*16 + 4*i + (lr&3); the swizzle key does not depend on
    // i, so the four tiles are one base + immediates (i*512), and the second k-half is base ^ 64.
    const int wrow_l = wave_c * WC + (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
    const int prow0 = wave_p * WP + lr; // window row of this lane's pixel in tile 0 at tap (0,0)

    auto compute_tap = [&](const char *halo, const char *wb, int tapoff) {
        const int base = prow0 + tapoff;
        const unsigned pfrag0 = base * 128 + ((lg ^ (base & 7)) << 4); // tiles j: + j*2048 (key unchanged)
        if constexpr (SPLIT) { // k-half 0 = the hi halves of the row's 32 channels, k-half 1 = their lo halves
            uint4 ph[TP], wh[TC], wl[TC], pl[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) ph[j] = *reinterpret_cast<const uint4 *>(halo + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wh[i] = *reinterpret_cast<const uint4 *>(wb + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) wl[i] = *reinterpret_cast<const uint4 *>(wb + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int j = 0; j < TP; ++j) pl[j] = *reinterpret_cast<const uint4 *>(halo + (pfrag0 ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    mma_h(wh[i], ph[j], acc[i][j], (T *)nullptr);
                    mma_h(wl[i], ph[j], acc1[i][j], (T *)nullptr);
                }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wh[i], pl[j], acc1[i][j], (T *)nullptr);
            return;
        }
#pragma unroll
        for (int kh2 = 0; kh2 < 2; ++kh2) {
            const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0;
            const unsigned wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
            uint4 pf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(halo + pa + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(wb + wa + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wf[i], pf[j], acc[i][j], (T *)nullptr);
        }
    };

    // Six-slab ring (split tiles): the fragments of tap g + 1 are read from LDS BEFORE the MFMAs of tap g are issued, into the second of two register
    // sets (g & 1): with one barrier per tap and all eight waves in step, the ds_read phase (8 waves x 10-12 b128 reads = 80-96 KB per tap through a
    // 128 B/clk port) and the MFMA phase (two waves per SIMD) otherwise run one after the other.  Same MFMAs in the same order per accumulator.
    constexpr int kFr = (NWB > 3 && SPLIT) ? 2 * TP + 2 * TC : 1;
    uint4 fr[NWB > 3 ? 2 : 1][kFr];
    auto load_frags = [&](uint4(&f)[kFr], const char *halo, const char *wb, int tapoff) __attribute__((always_inline)) {
        if constexpr (NWB > 3 && SPLIT) {
            const int base = prow0 + tapoff;
            const unsigned pfrag0 = base * 128 + ((lg ^ (base & 7)) << 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) f[j] = *reinterpret_cast<const uint4 *>(halo + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) f[2 * TP + i] = *reinterpret_cast<const uint4 *>(wb + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) f[2 * TP + TC + i] = *reinterpret_cast<const uint4 *>(wb + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int j = 0; j < TP; ++j) f[TP + j] = *reinterpret_cast<const uint4 *>(halo + (pfrag0 ^ 64u) + j * 2048);
        }
    };
    auto mma_frags = [&](const uint4(&f)[kFr]) __attribute__((always_inline)) { // the order of compute_tap's split branch
        if constexpr (NWB > 3 && SPLIT) {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    mma_h(f[2 * TP + i], f[j], acc[i][j], (T *)nullptr);
                    mma_h(f[2 * TP + TC + i], f[j], acc1[i][j], (T *)nullptr);
                }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(f[2 * TP + i], f[TP + j], acc1[i][j], (T *)nullptr);
        }
    };

    const int nchunks = a.Cin / CCH;
    const int cb = n0 + wave_c * WC + lg * NV; // first of the NV consecutive couts this lane owns
    // the accumulators start at the bias (rows exist up to CoutPad): no v_add per output value in the epilogue
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const floatx4 b4 = (floatx4){a.bias[cb + i * 4 + 0], a.bias[cb + i * 4 + 1], a.bias[cb + i * 4 + 2], a.bias[cb + i * 4 + 3]};
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = b4;
    }

#ifdef WTK_HALO_STAMPS
    const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- prologue: whole window of chunk 0 + weights of tap 0 (and tap 1 with the three-slab ring)
#pragma unroll
    for (int q = 0; q < kMaxPiecesPerWave; ++q) issue_halo_piece(halo0, q, 0);
    if constexpr (NWB > 3) { // six-slab ring: the slabs of taps 0..4 (a layer has at least 9 taps)
#pragma unroll
        for (int t0 = 0; t0 < NWB - 1; ++t0) issue_weights(wring + t0 * (BN * 128), t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_weights(wbuf0, 0, 0);
        if (NWB == 3) {
            issue_weights(wbuf1, 1, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the raw requests are invisible to the compiler's own wait insertion
        }
    }
    __syncthreads();
    // The bias values above come from ordinary loads: without a use in front of the loop hipcc waits for them at their first use INSIDE it, and —
    // not knowing how many raw LDS-DMA requests are younger — does so with a vmcnt(0) on every trip (found in the 64-cout x 128-pixel variants:
    // one full drain per two chunks; tools/asm_loop_waits.py lists such waits).  Here the queue is empty anyway.
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) asm volatile("" : "+v"(acc[i][j]));
    if constexpr (NWB > 3) load_frags(fr[0], halo0, wring, 0);

#ifdef WTK_HALO_TAP_STAMPS // diagnostic builds: per-wave cycle totals of (issue + ds_read + MFMA), vmcnt wait, barrier wait
    unsigned long long tap_sum[3] = {0, 0, 0};
    unsigned long long tap_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long clk_c0 = tap_prev, clk_r0 = __builtin_amdgcn_s_memrealtime(); // in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
#endif
    // one channel chunk = 9 taps.  CP = parity of the chunk: window in halo[CP].
    // NWB == 2: tap t's weights in wbuf[(CP+t)&1], next tap's slab requested at the top of the tap, vmcnt(0) at its end.
    // NWB == 3: tap t's weights in wbuf[t % 3] (9 taps per chunk keep the ring aligned), slab of tap t+2 requested at the
    //           top of tap t, counted wait at its end: only the requests of THIS tap stay in flight across the barrier.
    auto chunk_body = [&](auto cp_tag, int c) {
        constexpr int CP = decltype(cp_tag)::value;
        const char *hcur = (NHALO == 2 && CP == 1) ? halo1 : halo0;
        char *hnext = (NHALO == 2 && CP == 0) ? halo1 : halo0;
        const bool more = c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if constexpr (NWB == 2) {
                char *wnext = ((CP + tap) & 1) ? wbuf0 : wbuf1;
                const char *wcur = ((CP + tap) & 1) ? wbuf1 : wbuf0;
                if (tap < 8)
                    issue_weights(wnext, tap + 1, c);
                else if (more)
                    issue_weights(wnext, 0, c + 1);
                if (NHALO == 2 && more && tap < kMaxPiecesPerWave) // next chunk's window, one piece per wave per
                    issue_halo_piece(hnext, tap, c + 1);           // tap, underneath the MFMAs
                compute_tap(hcur, wcur, (tap / 3) * pitch + (tap % 3));
                __syncthreads(); // vmcnt(0): everything issued above has landed; everyone is done reading wcur
            } else if constexpr (NWB > 3) {
                // Six-slab ring (64-cout split tiles: 2 x 54 + 6 x 8 = 156 KB).  The slab of global tap g = 9 c + tap lives in slot g % 6 = (3 CP + tap) % 6;
                // tap g requests the slab of tap g + 5 into the slot tap g - 1 was read from, so a slab has four taps to arrive (the three-slab ring:
                // one) and is there one tap BEFORE its tap: tap g reads the fragments of tap g + 1 first and multiplies its own — read during tap g - 1 —
                // underneath.  The next chunk's window goes out two pieces per tap at taps 0..3: it has landed by the wait of tap 7, in front of the
                // fragment reads of the next chunk's tap 0.  Write-after-read: the slot of tap g - 1 and the window of chunk c - 1 were last read one tap
                // before their last tap, a barrier earlier than the first request into them.  Same taps in the same order: bit-identical.
                constexpr int kSlab = BN * 128;
                static_assert(kMaxPiecesPerWave <= 8, "window pieces are requested at taps 0..3, two per tap");
                auto pcs = [](int t) constexpr { t = (t + 9) % 9; const int left = kMaxPiecesPerWave - 2 * t; return left < 0 ? 0 : (left > 2 ? 2 : left); };
                const int par = (CP + tap) & 1; // parity of the global tap index (a constant once the tap loop is unrolled)
                const char *wnext1 = wring + ((3 * CP + tap + 1) % NWB) * kSlab;
                char *wnext5 = wring + ((3 * CP + tap + 5) % NWB) * kSlab;
                if (tap < 8)
                    load_frags(fr[par ^ 1], hcur, wnext1, ((tap + 1) / 3) * pitch + ((tap + 1) % 3));
                else
                    load_frags(fr[par ^ 1], hnext, wnext1, 0); // (last chunk: values nobody uses)
                mma_frags(fr[par]);
                if (tap < 4)
                    issue_weights(wnext5, tap + 5, c);
                else
                    issue_weights(wnext5, tap - 4, more ? c + 1 : c); // last chunk: a slab nobody reads again (constant request count per tap)
                if (2 * tap < kMaxPiecesPerWave) issue_halo_piece_always(hnext, 2 * tap, more ? c + 1 : c);
                if (2 * tap + 1 < kMaxPiecesPerWave) issue_halo_piece_always(hnext, 2 * tap + 1, more ? c + 1 : c);
                // landed after this wait: the slab of tap g + 2 (requested at tap g - 3) and everything older; in flight: the window pieces of tap g - 3
                // and all requests of taps g - 2 .. g
                constexpr int WRc = BN / 64;
                const int in_flight = pcs(tap - 3) + (WRc + pcs(tap - 2)) + (WRc + pcs(tap - 1)) + (WRc + pcs(tap));
                wait_vmcnt(in_flight);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                const char *wcur = tap % 3 == 0 ? wbuf0 : (tap % 3 == 1 ? wbuf1 : wbuf2);
                char *wnext2 = (tap + 2) % 3 == 0 ? wbuf0 : ((tap + 2) % 3 == 1 ? wbuf1 : wbuf2);
                // The number of LDS-DMA instructions per tap is a compile-time constant (no branch around any of them):
                // hipcc tracks pending LDS-DMA per LDS object and, when it cannot count the younger requests, puts a
                // vmcnt(0) in front of the first fragment read of a slab — draining exactly what this schedule keeps
                // in flight.  Where nothing is needed (last chunk: no next slab / no next window) a harmless duplicate
                // is requested instead: a slab nobody reads again, or a window piece of the dead buffer.
                const int issued = WR + ((NHALO == 2 && tap < kMaxPiecesPerWave) ? 1 : 0); // constant after unrolling
                compute_tap(hcur, wcur, (tap / 3) * pitch + (tap % 3));
                // The requests go out AFTER the tap's fragment reads and MFMAs have been issued: an LDS-DMA instruction costs
                // 100-185 issue cycles next to ds_reads but far less in the quiet stretch before the barrier, where this wave
                // would otherwise only wait for its SIMD neighbour (stamped: ~290 of ~1530 cycles per tap)
                if (tap < 7)
                    issue_weights(wnext2, tap + 2, c);
                else
                    issue_weights(wnext2, tap - 7, more ? c + 1 : c);
                if (NHALO == 2 && tap < kMaxPiecesPerWave) issue_halo_piece_always(hnext, tap, more ? c + 1 : c); // tap is unrolled
                // everything requested in EARLIER taps (next tap's slab, older window pieces) has landed; this tap's
                // requests keep flying.  lgkmcnt(0): this wave's fragment reads of wcur / hcur are done.
#ifdef WTK_HALO_TAP_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
                wait_vmcnt(issued);
                const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
                tap_sum[0] += ts0 - tap_prev, tap_sum[1] += ts1 - ts0, tap_sum[2] += ts2 - ts1, tap_prev = ts2;
#else
                wait_vmcnt(issued);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#endif
                if constexpr (NHALO == 1) {
                    // ONE window buffer and more than one chunk (the two-blocks-per-CU form of the split kernel): the next chunk's window is requested when
                    // everybody is done with this one, and waited for on the spot — the OTHER block of the CU is what runs meanwhile
                    if (tap == 8 && more) {
#pragma unroll
                        for (int q = 0; q < kMaxPiecesPerWave; ++q) issue_halo_piece(halo0, q, c + 1);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                }
            }
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk_body(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, c + 1);
    }

#ifdef WTK_HALO_TAP_STAMPS
    if (lane == 0 && a.dbg_stamps)
    {
        for (int i = 0; i < 3; ++i) a.dbg_stamps[((long long)blockIdx.x * 8 + wave) * 4 + i] = tap_sum[i];
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk_c0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        a.dbg_stamps[((long long)blockIdx.x * 8 + wave) * 4 + 3] = (dc << 24) | (dr & 0xffffff);
    }
#endif
#ifdef WTK_HALO_STAMPS
    const unsigned long long st_t2 = __builtin_amdgcn_s_memrealtime();
    struct StampOnExit {
        unsigned long long *p, t0, t1, t2;
        bool on;
        __device__ ~StampOnExit() {
            if (on) p[0] = t0, p[1] = t1, p[2] = t2, p[3] = __builtin_amdgcn_s_memrealtime();
        }
    } stamp_on_exit{a.dbg_stamps + ((long long)blockIdx.x * 8 + wave) * 4, st_t0, st_t1, st_t2, a.dbg_stamps != nullptr && lane == 0};
#endif
    // ---- epilogue (the bias is already inside the accumulators).  Output pixels: one evaluation per lane, fetched per pixel tile
    // (before any lane leaves: ds_bpermute reads from active lanes only)
    int pix_e, col_e;
    halo_out_pixel(a, o0 + wave_p * WP, xs, lane, pix_e, col_e);
    int pixj[TAIL ? 1 : TP], colj[TAIL ? 1 : TP]; // the fused-tail variants have no padded couts: they fetch inside their loops
    if constexpr (!TAIL) {
#pragma unroll
        for (int j = 0; j < TP; ++j) pixj[j] = lane_fetch(j * 16 + lr, pix_e), colj[j] = lane_fetch(j * 16 + lr, col_e);
    }
    if (cb + NV > a.Cout) return;
    // ---- fused 1x1 tail (fp16, 64-cout tile: the wave owns ALL 64 output channels of its pixels).  The Detect box tower's
    // last conv (1x1, 64 -> 64, no activation) consumes this conv's output and nothing else does: instead of writing the
    // 64-channel tensor and launching a second kernel that reads it back, the SiLU'd fp16 values go to a wave-local LDS tile
    // (the window buffer is free after the last tap's barrier) and are multiplied by the 1x1 weights right here.  Same fp16
    // rounding of the intermediate, same K order (two 32-deep steps), same MFMA: bit-identical to the two-kernel path.
    if constexpr (TAIL && SPLIT && BN == 128) {
        // ---- split-fp16 form of the 128-cout tail (f16x3 handles, Detect class towers: 3x3 128 -> 128, then 1x1 128 -> nc stored as 32 padded
        // couts, fp32 logits).  A wave holds 64 pixels x ONE HALF of the channels; every wave writes its SiLU'd values as split rows — per pixel and
        // block of 32 channels one 128-byte row [hi32 | lo32], 2 x WP rows = 16 KB per wave — into a tile of its own, the two cout-waves of a pixel
        // group meet at one block barrier, and each then multiplies HALF of the group's pixels over all four channel blocks (blocks 0, 1 from the
        // low-channel wave's tile, 2, 3 from the high-channel one: the K order of the stand-alone split 1x1, per block hi.hi into acc2, lo.hi then
        // hi.lo into acc2l, value = acc2 + 2^-11 acc2l: bit-identical to the two-kernel path).  The eight tiles (128 KB) take both window buffers
        // (three each) and two of the three slab buffers — all free once the last tap's requests have landed (the last taps re-request a slab nobody
        // reads: it must not land on a tile, hence the drain in front of the first store).
        static_assert(WAVES_C == 2 && TC == 4 && TP % 2 == 0 && NHALO == 2 && NWB == 3 && kHaloBytesT >= 3 * 2 * WP * 128 && BN * 128 >= 2 * WP * 128,
                      "split class-tower tail: 4 x 2 waves, three tiles per window buffer, one per slab buffer");
        constexpr int kTile = 2 * WP * 128; // bytes of one wave's tile: rows blk * WP + p
        auto tile_of = [&](int w) __attribute__((always_inline)) -> char * {
            return w < 3 ? halo0 + w * kTile : (w < 6 ? halo1 + (w - 3) * kTile : (w == 6 ? wbuf0 : wbuf1));
        };
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.tail_w);
        const int arow = (lr >> 2) * 8 + (lr & 3); // + 4i: the lane ends up owning couts lg*8 .. lg*8+7 of the 32 stored ones
        half8 wh2[4][2], wl2[4][2];                // A fragments straight from global memory (16 KB of weights, L2 resident); requested first, used last
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const _Float16 *wr = w2 + (long long)(arow + 4 * i) * a.tail_kpad + kb * 64 + lg * 8;
                wh2[kb][i] = *reinterpret_cast<const half8 *>(wr);
                wl2[kb][i] = *reinterpret_cast<const half8 *>(wr + 32);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (also the weight fragments above: a few hundred cycles, once per block)
        __builtin_amdgcn_s_barrier();                     // every wave's requests have landed, every wave is past its last fragment read
        asm volatile("" ::: "memory");
        {
            char *mine = tile_of(wave) + (lg >> 1) * (WP * 128); // this lane's 16 couts lie in block lg / 2 of the wave's two, channels 16 (lg & 1) .. + 15 of it
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = j * 16 + lr;
                float v[NV];
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[i * 4 + r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
                if (a.act) wtk_silu_scaled_run<NV>(v);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    half8 hv, lv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = v[c2 * 8 + e];
                        const _Float16 hh = (_Float16)x;
                        hv[e] = hh;
                        lv[e] = (_Float16)((x - (float)hh) * kSplitScale);
                    }
                    const int c = 2 * (lg & 1) + c2; // chunk of the hi halves; the lo halves: + 4
                    *reinterpret_cast<half8 *>(mine + p * 128 + ((c ^ (p & 7)) << 4)) = hv;
                    *reinterpret_cast<half8 *>(mine + p * 128 + (((c + 4) ^ (p & 7)) << 4)) = lv;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int TPH = TP / 2; // pixel tiles of the group this wave finishes
        floatx4 acc2[2][TPH], acc2l[2][TPH];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const floatx4 b4 = (floatx4){a.tail_bias[lg * 8 + i * 4 + 0], a.tail_bias[lg * 8 + i * 4 + 1], a.tail_bias[lg * 8 + i * 4 + 2], a.tail_bias[lg * 8 + i * 4 + 3]};
#pragma unroll
            for (int j = 0; j < TPH; ++j) acc2[i][j] = b4, acc2l[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const char *src = tile_of(wave_p * 2 + (kb >> 1)) + (kb & 1) * (WP * 128); // channels 0..63 of the group from the low-channel wave, 64..127 from the other
#pragma unroll
            for (int j = 0; j < TPH; ++j) {
                const int p = (wave_c * TPH + j) * 16 + lr;
                const half8 ph = *reinterpret_cast<const half8 *>(src + p * 128 + ((lg ^ (p & 7)) << 4));
                const half8 pl = *reinterpret_cast<const half8 *>(src + p * 128 + (((lg + 4) ^ (p & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[kb][i], ph, acc2[i][j], 0, 0, 0);
                    acc2l[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl2[kb][i], ph, acc2l[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) acc2l[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[kb][i], pl, acc2l[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < TPH; ++j) {
            const long long pix = lane_fetch((wave_c * TPH + j) * 16 + lr, pix_e);
            if (pix < 0) continue;
            float v2[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v2[i * 4 + r] = wtk_split_value(acc2[i][j][r], acc2l[i][j][r]);
            if (lg * 8 < a.tail_cout) // padded couts are never stored; the class logits are an fp32 tensor (launch check)
                store_run_h<8>(reinterpret_cast<float *>(a.tail_out) + pix * a.tail_ld + a.tail_coff + lg * 8, v2);
        }
        return;
    } else if constexpr (TAIL && SPLIT) {
        // ---- split-fp16 form of the 64-cout tail (f16x3 handles, Detect box towers): the wave's SiLU'd values go to LDS as split rows — one 128-byte
        // row [hi32 | lo32] per pixel and block of 32 channels, block 0 in the first window buffer, block 1 in the second (both free after the last tap's
        // barrier), wave-local — and are multiplied by the split 1x1 weights as conv_igemm_kernel's split form does: two K steps (the blocks), per step
        // hi.hi into acc2, lo.hi then hi.lo into acc2l, value = acc2 + 2^-11 acc2l: bit-identical to the two-kernel path.
        static_assert(BN == 64 && TC == 4 && NHALO == 2, "split fused tail: 64 couts per wave, two window buffers");
        char *tile0 = halo0 + wave * (WP * 128), *tile1 = halo1 + wave * (WP * 128);
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.tail_w);
        const int arow = (lr >> 2) * 16 + (lr & 3); // + 4i: cout row of A fragment tile i (the lane ends up owning couts lg*16 .. lg*16+15)
        half8 wh2[2][4], wl2[2][4]; // A fragments straight from global memory (16 KB of weights, L2 resident); requested first, used last
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 *wr = w2 + (long long)(arow + 4 * i) * a.tail_kpad + blk * 64 + lg * 8;
                wh2[blk][i] = *reinterpret_cast<const half8 *>(wr);
                wl2[blk][i] = *reinterpret_cast<const half8 *>(wr + 32);
            }
        {
            char *mine = (lg >> 1) ? tile1 : tile0; // this lane's 16 couts lie in block lg / 2, channels 16 (lg & 1) .. + 15 of it
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = j * 16 + lr;
                float v[NV];
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[i * 4 + r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
                if (a.act) wtk_silu_scaled_run<NV>(v);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    half8 hv, lv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = v[c2 * 8 + e];
                        const _Float16 hh = (_Float16)x;
                        hv[e] = hh;
                        lv[e] = (_Float16)((x - (float)hh) * kSplitScale);
                    }
                    const int c = 2 * (lg & 1) + c2; // chunk of the hi halves; the lo halves: + 4
                    *reinterpret_cast<half8 *>(mine + p * 128 + ((c ^ (p & 7)) << 4)) = hv;
                    *reinterpret_cast<half8 *>(mine + p * 128 + (((c + 4) ^ (p & 7)) << 4)) = lv;
                }
            }
        }
        // the rows of a pixel come from all four lane groups of this wave: its LDS operations execute in order, no barrier
        floatx4 acc2[4][TP], acc2l[4][TP];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const floatx4 b4 = (floatx4){a.tail_bias[lg * 16 + i * 4 + 0], a.tail_bias[lg * 16 + i * 4 + 1], a.tail_bias[lg * 16 + i * 4 + 2],
                                         a.tail_bias[lg * 16 + i * 4 + 3]};
#pragma unroll
            for (int j = 0; j < TP; ++j) acc2[i][j] = b4, acc2l[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const char *tile = blk ? tile1 : tile0;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = j * 16 + lr;
                const half8 ph = *reinterpret_cast<const half8 *>(tile + p * 128 + ((lg ^ (p & 7)) << 4));
                const half8 pl = *reinterpret_cast<const half8 *>(tile + p * 128 + (((lg + 4) ^ (p & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[blk][i], ph, acc2[i][j], 0, 0, 0);
                    acc2l[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl2[blk][i], ph, acc2l[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2l[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh2[blk][i], pl, acc2l[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long long pix = lane_fetch(j * 16 + lr, pix_e);
            if (pix < 0) continue;
            float v2[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v2[i * 4 + r] = wtk_split_value(acc2[i][j][r], acc2l[i][j][r]);
            if (a.tail_f32)
                store_run_h<16>(reinterpret_cast<float *>(a.tail_out) + pix * a.tail_ld + a.tail_coff + lg * 16, v2); // head logits stay fp32
            else
                wtk_split_store<16>(reinterpret_cast<_Float16 *>(a.tail_out) + pix * a.tail_ld + a.tail_coff, lg * 16, v2);
        }
        return;
    } else if constexpr (TAIL && BN == 128) {
        // ---- 128-cout tile (the Detect class tower: 3x3 128 -> 128, then 1x1 128 -> nc stored as 32 padded channels).  A wave
        // holds 64 pixels x ONE HALF of the channels, so the two cout-waves of a pixel group exchange through LDS: both write
        // their SiLU'd fp16 tile (64 px x 64 ch), one block barrier, then each of them multiplies HALF of the group's pixels over
        // all 128 channels — k-steps 0,1 from the low-channel tile, 2,3 from the high-channel tile: the K order of the stand-alone
        // 1x1 kernel, so the result is bit-identical.  Both window buffers are free after the last tap's barrier.
        static_assert(sizeof(T) == 2 && WAVES_C == 2 && TC == 4 && TP % 2 == 0, "fused class-tower tail: fp16, 4 x 2 waves");
        const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.tail_w);
        const int arow = (lr >> 2) * 8 + (lr & 3); // + 4i: the lane ends up owning couts lg*8 .. lg*8+7 of the 32 stored ones
        half8 wf2[4][2];                           // A fragments straight from global memory (8 KB of weights); requested first, used last
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) wf2[ks][i] = *reinterpret_cast<const half8 *>(w2 + (long long)(arow + 4 * i) * a.tail_kpad + ks * 32 + lg * 8);
        auto tile_of = [&](int w) __attribute__((always_inline)) -> char * { return (w < 4 ? halo0 : halo1) + (w & 3) * (WP * 128); };
        char *mine = tile_of(wave);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int p = j * 16 + lr;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = c2 * 8 + e;
                    const float x = acc[idx >> 2][j][idx & 3];
                    hv[e] = (_Float16)(a.act ? silu_h(x) : x);
                }
                const int c = 2 * lg + c2;
                *reinterpret_cast<half8 *>(mine + p * 128 + ((c ^ (p & 7)) << 4)) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int TPH = TP / 2; // pixel tiles of the group this wave finishes
        floatx4 acc2[2][TPH];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const floatx4 b4 = (floatx4){a.tail_bias[lg * 8 + i * 4 + 0], a.tail_bias[lg * 8 + i * 4 + 1], a.tail_bias[lg * 8 + i * 4 + 2], a.tail_bias[lg * 8 + i * 4 + 3]};
#pragma unroll
            for (int j = 0; j < TPH; ++j) acc2[i][j] = b4;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char *src = tile_of(wave_p * 2 + (ks >> 1)); // channels 0..63 of the group, then 64..127
#pragma unroll
            for (int j = 0; j < TPH; ++j) {
                const int p = (wave_c * TPH + j) * 16 + lr;
                const half8 pf = *reinterpret_cast<const half8 *>(src + p * 128 + ((((ks & 1) * 4 + lg) ^ (p & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf2[ks][i], pf, acc2[i][j], 0, 0, 0);
            }
        }
        _Float16 *tout = reinterpret_cast<_Float16 *>(a.tail_out);
#pragma unroll
        for (int j = 0; j < TPH; ++j) {
            const long long pix = lane_fetch((wave_c * TPH + j) * 16 + lr, pix_e);
            if (pix < 0) continue;
            float v2[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v2[i * 4 + r] = acc2[i][j][r];
            if (lg * 8 < a.tail_cout) { // padded couts are never stored
                if (a.tail_f32)
                    store_run_h<8>(reinterpret_cast<float *>(a.tail_out) + pix * a.tail_ld + a.tail_coff + lg * 8, v2); // head logits stay fp32
                else
                    store_run_h<8>(tout + pix * a.tail_ld + a.tail_coff + lg * 8, v2);
            }
        }
        return;
    } else if constexpr (TAIL) {
        static_assert(!TAIL || (BN == 64 && sizeof(T) == 2), "fused tail: fp16, 64- or 128-cout tile");
        {
            static_assert(BN != 64 || TC == 4, "64 couts per wave");
            char *tile = halo0 + wave * (WP * 128); // WP rows of 128 B: 64 channels of the wave's pixels
            const _Float16 *w2 = reinterpret_cast<const _Float16 *>(a.tail_w);
            const int arow = (lr >> 2) * 16 + (lr & 3); // + 4i: cout row of A fragment tile i (same permutation as above)
            half8 wf2[2][4]; // A fragments straight from global memory (8 KB of weights, L2 resident); requested first, used last
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) wf2[ks][i] = *reinterpret_cast<const half8 *>(w2 + (long long)(arow + 4 * i) * a.tail_kpad + ks * 32 + lg * 8);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = j * 16 + lr; // row of the wave's tile
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    half8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int idx = c2 * 8 + e;
                        const float x = acc[idx >> 2][j][idx & 3];
                        hv[e] = (_Float16)(a.act ? silu_h(x) : x);
                    }
                    const int c = 2 * lg + c2;
                    *reinterpret_cast<half8 *>(tile + p * 128 + ((c ^ (p & 7)) << 4)) = hv;
                }
            }
            // the first conv's accumulators are dead now: the second set starts at the tail's bias
            floatx4 acc2[4][TP];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const floatx4 b4 = (floatx4){a.tail_bias[lg * 16 + i * 4 + 0], a.tail_bias[lg * 16 + i * 4 + 1], a.tail_bias[lg * 16 + i * 4 + 2],
                                             a.tail_bias[lg * 16 + i * 4 + 3]};
#pragma unroll
                for (int j = 0; j < TP; ++j) acc2[i][j] = b4;
            }
            // wave-local tile: a wave's LDS operations execute in order, no barrier
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int p = j * 16 + lr;
                    const half8 pf = *reinterpret_cast<const half8 *>(tile + p * 128 + (((ks * 4 + lg) ^ (p & 7)) << 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf2[ks][i], pf, acc2[i][j], 0, 0, 0);
                }
            _Float16 *tout = reinterpret_cast<_Float16 *>(a.tail_out);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const long long pix = lane_fetch(j * 16 + lr, pix_e);
                if (pix < 0) continue;
                float v2[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v2[i * 4 + r] = acc2[i][j][r];
                if (a.tail_f32)
                    store_run_h<16>(reinterpret_cast<float *>(a.tail_out) + pix * a.tail_ld + a.tail_coff + lg * 16, v2); // head logits stay fp32
                else
                    store_run_h<16>(tout + pix * a.tail_ld + a.tail_coff + lg * 16, v2);
            }
            return;
        }
    }
    T *out = reinterpret_cast<T *>(a.out);
    T *out2 = reinterpret_cast<T *>(a.out2);
    const T *res = reinterpret_cast<const T *>(a.res);
    // the residual of all pixel tiles is requested before any arithmetic (junk pixels read pixel 0): one exposed memory latency
    // per block instead of TP (variants with a 256-register budget only)
    constexpr bool kHoistRes = MINW <= 2 && sizeof(T) == 2 && BN != 192 && !TAIL && !SPLIT; // raw fp16 values: 8 VGPRs per pixel tile
    half8 rres[kHoistRes ? TP : 1][kHoistRes ? NV / 8 : 1];
    if constexpr (kHoistRes) {
        if (res) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const long long pr = pixj[j];
                const T *rp = res + (pr < 0 ? 0 : pr) * a.res_ld + a.res_coff + cb;
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) rres[j][q] = *reinterpret_cast<const half8 *>(rp + 8 * q);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const long long pix = pixj[TAIL ? 0 : j];
        const int col = colj[TAIL ? 0 : j];
        if (pix < 0) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (SPLIT)
                    v[i * 4 + r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
                else
                    v[i * 4 + r] = acc[i][j][r];
            }
        if (a.act) {
            wtk_silu_scaled_run<NV>(v);
        }
        if (res) {
            if constexpr (kHoistRes) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] += (float)rres[j][i >> 3][i & 7];
            } else {
                float rv[NV];
                if constexpr (SPLIT)
                    wtk_split_load<NV>(reinterpret_cast<const _Float16 *>(a.res) + pix * a.res_ld + a.res_coff, cb, rv);
                else
                    load_run_h<NV>(res + pix * a.res_ld + a.res_coff + cb, rv);
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] += rv[i];
            }
        }
        if constexpr (SPLIT)
            wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out) + pix * a.out_ld + a.out_coff, cb, v);
        else
            store_run_h<NV>(out + pix * a.out_ld + a.out_coff + cb, v);
        if (out2) { // 2x nearest upsample: pixel (n, 2y+dy, 2X+dx) of the [2H][2W] map = 4*pix - 2X + 2W*dy + dx
            const int W2 = a.W * 2;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const long long pix2 = 4 * pix - 2 * col + W2 * dy + dx;
                    if constexpr (SPLIT)
                        wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out2) + pix2 * a.out2_ld + a.out2_coff, cb, v);
                    else
                        store_run_h<NV>(out2 + pix2 * a.out2_ld + a.out2_coff + cb, v);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of the three-slab kernel (two window buffers, even number of 64-channel chunks, BN = 128 / 192).
// A block walks tiles v = blockIdx.x, + gridDim.x, ... and the tap pipeline never drains: during the LAST chunk of a
// tile the "next chunk" requests (window pieces at taps 0..6, weight slabs at taps 7 / 8) simply name chunk 0 of the
// NEXT tile, whose geometry (piece offsets, image, cout tile, bias) is computed while this tile is still multiplying.
// Per tile this removes the ~1.2 us of address setup and the ~1.5 us wait for the first window + slab that every
// block of the one-tile-per-block kernel spends with an idle matrix pipe (stamped: 8-25 % of a block's life).
// Arithmetic and K order are those of conv3x3_halo_kernel: bit-identical results.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BN, int HROWS>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_pkernel(const HaloArgs a) {
    asm volatile("" ::"s"(a.in), "s"(a.w), "s"(a.bias), "s"(a.zeros), "s"(a.in_ld), "s"(a.in_coff), "s"(a.N), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CoutPad),
                 "s"(a.Kpad), "s"(a.S), "s"(a.pitch), "s"(a.strips), "s"(a.d_strips.mul), "s"(a.d_strips.sh1), "s"(a.d_strips.sh2), "s"(a.d_pitch.mul),
                 "s"(a.d_pitch.sh1), "s"(a.d_pitch.sh2), "s"(a.d_nct.mul), "s"(a.d_nct.sh1), "s"(a.d_nct.sh2), "s"(a.d_h1.mul), "s"(a.d_h1.sh1), "s"(a.d_h1.sh2), "s"(a.grid),
                 "s"(a.blocks_per_strip)); // one batch of scalar loads (see conv3x3_halo_kernel)
    constexpr int CE = ElemH<T>::CE;
    constexpr int CCH = 8 * CE;
    constexpr int WAVES_C = 2, WAVES_P = 4;
    constexpr int WC = BN / WAVES_C;
    constexpr int WP = kBM / WAVES_P, TP = WP / 16, TC = WC / 16, NV = 4 * TC;
    constexpr int WR = BN / 64;
    constexpr int kHaloBytesT = HROWS * 128;
    constexpr int kPieces = HROWS / 8;
    constexpr int kMaxPiecesPerWave = (kPieces + 7) / 8;
    static_assert(kMaxPiecesPerWave <= 7, "window pieces are requested at taps 0..6");

    __shared__ __attribute__((aligned(16))) char halo0[kHaloBytesT];
    __shared__ __attribute__((aligned(16))) char halo1[kHaloBytesT];
    __shared__ __attribute__((aligned(16))) char wbuf0[BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf1[BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf2[BN * 128];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int lr = lane & 15, lg = lane >> 4;
    const int pitch = a.pitch;
    const int halo_rows = kBM + 2 * pitch + 2;
    const int nct = a.CoutPad / BN;
    const int nchunks = a.Cin / CCH; // even (launcher)
    const int total = a.strips * a.blocks_per_strip * nct;
    const int G = a.grid; // multiple of 8: a block's tiles stay on one XCD label
    const T *wgt = reinterpret_cast<const T *>(a.w);
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);

    // per-tile context.  Two copies (current / next) of plain scalars and small arrays: everything stays in registers.
    struct Tile {
        int o0, xs, n0;
        const char *img;   // base of the first image the window touches (input view)
        const char *wtile; // weight rows of this cout tile
        unsigned hoff[kMaxPiecesPerWave];
        unsigned hvalid;
        float bias[NV];
    };
    auto setup_tile = [&](int v, Tile &tc) __attribute__((always_inline)) {
        // XCD-aware bijective remap of the virtual block id (as conv3x3_halo_kernel, with nwg = total tiles)
        const int xcd = v & 7, q8 = total >> 3, r8 = total & 7;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        const unsigned t = fdiv((unsigned)L, a.d_nct);
        tc.n0 = (L - (int)t * nct) * BN;
        const int rb = (int)fdiv(t, a.d_strips); // row blocks major, strips minor (as conv3x3_halo_kernel)
        const int strip = (int)t - rb * a.strips;
        tc.o0 = rb * kBM;
        tc.xs = strip * a.S;
        const int n_base = (int)fdiv(fdiv((unsigned)tc.o0, a.d_pitch), a.d_h1);
        tc.img = reinterpret_cast<const char *>(reinterpret_cast<const T *>(a.in) + (long long)n_base * a.H * a.W * a.in_ld + a.in_coff);
        tc.wtile = reinterpret_cast<const char *>(wgt + (long long)tc.n0 * a.Kpad);
        halo_piece_offsets<T, kMaxPiecesPerWave>(a, tc.o0, tc.xs, n_base, halo_rows, wave, lane, tc.hoff, tc.hvalid);
        const int cb = tc.n0 + wave_c * WC + lg * NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) tc.bias[i] = a.bias[cb + i];
    };
    // window piece q of this wave (never skipped; see issue_halo_piece_always in conv3x3_halo_kernel)
    auto issue_piece = [&](char *buf, int q, const Tile &tc, int c) __attribute__((always_inline)) {
        const bool back = q > 0 && wave + 8 * q >= kPieces;
        const int qq = back ? q - 1 : q;
        const int piece = wave + 8 * qq;
        const unsigned off = back ? tc.hoff[q > 0 ? q - 1 : 0] : tc.hoff[q];
        const bool ok = back ? ((tc.hvalid >> (q > 0 ? q - 1 : 0)) & 1u) : ((tc.hvalid >> q) & 1u);
        if constexpr (WTK_HALO_BUFFER_DMA) {
            lds_dma16_buf(make_rsrc(tc.img), ok ? off : 0xffffffffu, (unsigned)(c * (CCH * (int)sizeof(T))), buf + piece * 1024);
        } else {
            const char *src = ok ? tc.img + (size_t)c * (CCH * sizeof(T)) + off : zero_page;
            lds_dma16<true>(src, buf + piece * 1024);
        }
    };
    const int wrow0 = tid >> 3, wp = tid & 7;
    unsigned wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = wrow0 + 64 * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wvoff[i] = (unsigned)(((long long)row * a.Kpad + (wp ^ key) * CE) * (long long)sizeof(T));
    }
    auto issue_weights = [&](char *buf, const char *wtile, int tap, int c) __attribute__((always_inline)) {
        if constexpr (WTK_HALO_BUFFER_DMA) {
            const rsrc_t rs = make_rsrc(wtile);
            const unsigned so = (unsigned)((tap * a.Cin + c * CCH) * (int)sizeof(T));
#pragma unroll
            for (int i = 0; i < WR; ++i) lds_dma16_buf(rs, wvoff[i], so, buf + (64 * i + 8 * wave) * 128);
        } else {
            const char *ub = wtile + ((size_t)tap * a.Cin + (size_t)c * CCH) * sizeof(T);
#pragma unroll
            for (int i = 0; i < WR; ++i) lds_dma16<true>(ub + wvoff[i], buf + (64 * i + 8 * wave) * 128);
        }
    };

    floatx4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int wrow_l = wave_c * WC + (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
    const int prow0 = wave_p * WP + lr;
    auto compute_tap = [&](const char *halo, const char *wb, int tapoff) __attribute__((always_inline)) {
        const int base = prow0 + tapoff;
        const unsigned pfrag0 = base * 128 + ((lg ^ (base & 7)) << 4);
#pragma unroll
        for (int kh2 = 0; kh2 < 2; ++kh2) {
            const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0;
            const unsigned wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
            uint4 pf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(halo + pa + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(wb + wa + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wf[i], pf[j], acc[i][j], (T *)nullptr);
        }
    };

    int v = blockIdx.x;
    if (v >= total) return;
    Tile cur, nxt;
    setup_tile(v, cur);
    nxt = cur;
    auto arm_acc = [&](const Tile &tc) __attribute__((always_inline)) { // accumulators start at the tile's bias
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = (floatx4){tc.bias[i * 4 + 0], tc.bias[i * 4 + 1], tc.bias[i * 4 + 2], tc.bias[i * 4 + 3]};
    };
    arm_acc(cur);
    // ---- prologue (once per block): whole window of chunk 0 + slabs of taps 0 and 1
#pragma unroll
    for (int q = 0; q < kMaxPiecesPerWave; ++q) issue_piece(halo0, q, cur, 0);
    issue_weights(wbuf0, cur.wtile, 0, 0);
    issue_weights(wbuf1, cur.wtile, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    T *out = reinterpret_cast<T *>(a.out);
    T *out2 = reinterpret_cast<T *>(a.out2);
    const T *res = reinterpret_cast<const T *>(a.res);

    // one channel chunk = 9 taps; CP = chunk parity inside the tile (chunk 0 of every tile is in halo0).
    // `last`: last chunk of the tile -> the "next chunk" requests go to chunk 0 of the next tile (or, on the final tile,
    // re-request data of the current one: the number of LDS-DMA instructions per tap stays constant)
    auto chunk_body = [&](auto cp_tag, int c, bool last, bool has_next) __attribute__((always_inline)) {
        constexpr int CP = decltype(cp_tag)::value;
        const char *hcur = CP == 1 ? halo1 : halo0;
        char *hnext = CP == 0 ? halo1 : halo0;
        const bool to_next = last && has_next;
        const int cn = last ? 0 : c + 1; // chunk the requests of this chunk are for
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char *wcur = tap % 3 == 0 ? wbuf0 : (tap % 3 == 1 ? wbuf1 : wbuf2);
            char *wnext2 = (tap + 2) % 3 == 0 ? wbuf0 : ((tap + 2) % 3 == 1 ? wbuf1 : wbuf2);
            const int issued = WR + (tap < kMaxPiecesPerWave ? 1 : 0);
            compute_tap(hcur, wcur, (tap / 3) * pitch + (tap % 3));
            // requests after the tap's reads and MFMAs (see conv3x3_halo_kernel)
            if (tap < 7)
                issue_weights(wnext2, cur.wtile, tap + 2, c);
            else
                issue_weights(wnext2, to_next ? nxt.wtile : cur.wtile, tap - 7, cn);
            if (tap < kMaxPiecesPerWave) {
                // select the geometry by value (no branch around the request)
                Tile sel;
                sel.img = to_next ? nxt.img : cur.img;
                sel.hvalid = to_next ? nxt.hvalid : cur.hvalid;
#pragma unroll
                for (int q = 0; q < kMaxPiecesPerWave; ++q) sel.hoff[q] = to_next ? nxt.hoff[q] : cur.hoff[q];
                issue_piece(hnext, tap, sel, cn);
            }
            wait_vmcnt(issued);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };

    while (true) {
        const bool has_next = v + G < total;
        if (has_next) setup_tile(v + G, nxt); // overlaps this tile's MFMAs
        for (int c = 0; c < nchunks; c += 2) {
            chunk_body(std::integral_constant<int, 0>{}, c, false, has_next);
            chunk_body(std::integral_constant<int, 1>{}, c + 1, c + 2 >= nchunks, has_next);
        }
        // ---- epilogue of the finished tile
        const int cb = cur.n0 + wave_c * WC + lg * NV;
        int pix_e, col_e; // output pixels: one evaluation per lane, fetched per pixel tile (all lanes active here)
        halo_out_pixel(a, cur.o0 + wave_p * WP, cur.xs, lane, pix_e, col_e);
        int pixj[TP], colj[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) pixj[j] = lane_fetch(j * 16 + lr, pix_e), colj[j] = lane_fetch(j * 16 + lr, col_e);
        if (cb + NV <= a.Cout) {
            // residual of all four pixel tiles, requested before any arithmetic (see conv3x3_halo_kernel); raw fp16: 8 VGPRs per tile
            constexpr bool kHoistRes = sizeof(T) == 2 && BN == 128;
            half8 rres[kHoistRes ? TP : 1][kHoistRes ? NV / 8 : 1];
            if constexpr (kHoistRes) {
                if (res) {
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const T *rp = res + (pixj[j] < 0 ? 0 : (long long)pixj[j]) * a.res_ld + a.res_coff + cb;
#pragma unroll
                        for (int q = 0; q < NV / 8; ++q) rres[j][q] = *reinterpret_cast<const half8 *>(rp + 8 * q);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const long long pix = pixj[j];
                if (pix < 0) continue;
                float vv[NV];
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[i * 4 + r] = acc[i][j][r];
                if (a.act) {
                    wtk_silu_scaled_run<NV>(vv);
                }
                if (res) {
                    if constexpr (kHoistRes) {
#pragma unroll
                        for (int i = 0; i < NV; ++i) vv[i] += (float)rres[j][i >> 3][i & 7];
                    } else {
                        float rv[NV];
                        load_run_h<NV>(res + pix * a.res_ld + a.res_coff + cb, rv);
#pragma unroll
                        for (int i = 0; i < NV; ++i) vv[i] += rv[i];
                    }
                }
                store_run_h<NV>(out + pix * a.out_ld + a.out_coff + cb, vv);
                if (out2) {
                    const int W2 = a.W * 2;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const long long pix2 = 4 * pix - 2 * colj[j] + W2 * dy + dx;
                            store_run_h<NV>(out2 + pix2 * a.out2_ld + a.out2_coff + cb, vv);
                        }
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        arm_acc(cur);
        v += G;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the final tile's duplicate requests must not outlive the block's LDS
}


// ---------------------------------------------------------------------------------------------------------------
// Weight-stationary window kernel for the 64 -> 64 channel 3x3 layers (fp16: the c = 64 bottlenecks on the 80x80 maps).
//
// All nine tap slabs of such a layer are 9 x 64 x 64 x 2 B = 72 KiB: they fit LDS next to two window buffers, so they are
// staged ONCE per persistent block and the main loop has no weight LDS-DMA and no per-tap barrier at all (the one change the
// ablation of conv3x3_halo_kernel showed to shorten a tap, profiles/r01_notes.md).  The block is two GROUPS of four waves (one
// wave of each group per SIMD).  A group owns a window buffer and walks its own tiles of 256 flat output pixels; per tile it
//   P: multiplies — 18 (tap, k-half) steps of 16 MFMAs per wave (64 px x 64 cout wave tile, 0.5 ds_read_b128 per MFMA), fragment
//      reads of step s+1 issued before the MFMAs of step s, no barrier, no vector-memory instruction in the stream;
//   Q: requests the next tile's window (LDS-DMA, buffer form), runs the SiLU / residual epilogue of the tile just finished while
//      those requests land, waits for them.
// The groups alternate: while one multiplies the other is in Q, one s_barrier per interval (= per 288 MFMAs of a wave instead of
// per 32).  The matrix pipe of a SIMD is fed by one wave at a time and never waits for an epilogue or a window.
// Geometry, fragment layouts, K order (tap-major, two 32-deep halves), bias-initialised accumulators and SiLU are those of
// conv3x3_halo_kernel<_Float16, 64, ...>: results are bit-identical (WTK_NO_WS64=1 switches back; tests compare).
// LDS: 73 728 (weights) + 2 x 44 032 (344-row windows: strips of <= 41 columns) = 161 792 B of 163 840.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWsRows = 344;                 // window rows per group buffer (43 pieces of 8 rows)
constexpr int kWsPiecesPerWave = 11;         // 43 pieces over the 4 waves of a group

// NWV (round 3): pixel tiles (of the four of a wave's 64 x 64 tile) whose epilogue is DEFERRED — their sums move to a second accumulator set
// when the group leaves its multiply phase and the SiLU / residual / store of those values is woven, one floatx4 at a time, between the
// MFMAs of the group's NEXT multiply phase (sched_group_barrier pins the interleave; stores go out as raw buffer stores whose junk lanes
// carry an out-of-range offset, so the phase has no branch in it).  Stamps of the round-2 kernel had Q (stage + epilogue, ~8 400 cycles)
// as the long pole of every interval against ~5 100 for the multiply: the partner's matrix time was wasted for a third of each interval,
// and with two waves per SIMD no amount of overlap BETWEEN waves gets under (P + Q) / 2 — only work moved INTO the multiplying wave's
// own instruction stream does.  NWV = 0 is the round-2 kernel (WTK_WS64_WEAVE=0), results are bit-identical for every NWV.
template <int NWV>
__global__ __launch_bounds__(512) void conv3x3_ws64_kernel(const HaloArgs a) {
    static_assert(NWV >= 0 && NWV <= 4, "woven pixel tiles");
    asm volatile("" ::"s"(a.in), "s"(a.w), "s"(a.bias), "s"(a.in_ld), "s"(a.in_coff), "s"(a.N), "s"(a.H), "s"(a.W), "s"(a.Kpad), "s"(a.S), "s"(a.pitch),
                 "s"(a.strips), "s"(a.d_strips.mul), "s"(a.d_strips.sh1), "s"(a.d_strips.sh2), "s"(a.d_pitch.mul), "s"(a.d_pitch.sh1), "s"(a.d_pitch.sh2),
                 "s"(a.d_h1.mul), "s"(a.d_h1.sh1), "s"(a.d_h1.sh2), "s"(a.grid), "s"(a.blocks_per_strip));
    using T = _Float16;
    constexpr int TP = 4, TC = 4, NV = 16, WP = 64;
    __shared__ __attribute__((aligned(16))) char wts[9 * 8192];
    __shared__ __attribute__((aligned(16))) char win0[kWsRows * 128];
    __shared__ __attribute__((aligned(16))) char win1[kWsRows * 128];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3; // group, wave inside the group (= pixel quarter of the group's tile)
    const int lr = lane & 15, lg = lane >> 4;
    const int pitch = a.pitch;
    const int halo_rows = 256 + 2 * pitch + 2;
    const int total = a.strips * a.blocks_per_strip; // one cout tile
    const int NG = 2 * a.grid;                      // groups in the launch
    const int g0 = 2 * (int)blockIdx.x;             // this block's first group id
    // tiles of a group: v = g, g + NG, ... < total
    const int nA = g0 < total ? (total - g0 + NG - 1) / NG : 0;
    const int nB = g0 + 1 < total ? (total - g0 - 1 + NG - 1) / NG : 0;
    const int n_mine = grp ? nB : nA;
    const int intervals = 2 * nA > 2 * nB + 1 ? 2 * nA : 2 * nB + 1;
    char *win = grp ? win1 : win0;

    // ---- per-tile geometry (flat origin, strip, image base, window piece offsets)
    struct Tile {
        int o0, xs;
        const char *img;
        unsigned hoff[kWsPiecesPerWave];
        unsigned hvalid;
    };
    auto setup_tile = [&](int k, Tile &tc) __attribute__((always_inline)) {
        const int v = g0 + grp + k * NG;
        const int xcd = v & 7, q8 = total >> 3, r8 = total & 7; // XCD-aware bijective remap (as conv3x3_halo_kernel)
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        const int rb = (int)fdiv((unsigned)L, a.d_strips);
        const int strip = L - rb * a.strips;
        tc.o0 = rb * 256;
        tc.xs = strip * a.S;
        const int n_base = (int)fdiv(fdiv((unsigned)tc.o0, a.d_pitch), a.d_h1);
        tc.img = reinterpret_cast<const char *>(reinterpret_cast<const T *>(a.in) + (long long)n_base * a.H * a.W * a.in_ld + a.in_coff);
        // window rows of this wave's pieces gw, gw + 4, ...: lane L evaluates row L & 7 of piece slot L >> 3 in two rounds (slots 0..7, 8..10)
        unsigned row_e[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int q = rr * 8 + (lane >> 3);
            const int hr = (gw + 4 * q) * 8 + (lane & 7);
            int pn = 0, iy = 0, ix = 0;
            const bool ok = q < kWsPiecesPerWave && hr < halo_rows && halo_in_coords(a, tc.o0 + hr, tc.xs, pn, iy, ix);
            // 32-bit offsets relative to the window's first image: pixel index < 2^24, bytes per pixel < 2^24 -> full-rate 24-bit multiplies
            const unsigned pixel = __umul24(__umul24((unsigned)(pn - n_base), (unsigned)a.H) + (unsigned)iy, (unsigned)a.W) + (unsigned)ix;
            row_e[rr] = ok ? __umul24(pixel, (unsigned)(a.in_ld * (int)sizeof(T))) : 0xffffffffu;
        }
        const unsigned lc_term = (unsigned)((((lane & 7) ^ ((lane >> 3) & 7)) * 8) * (int)sizeof(T));
        tc.hvalid = 0;
#pragma unroll
        for (int q = 0; q < kWsPiecesPerWave; ++q) {
            const unsigned v2 = (unsigned)__builtin_amdgcn_ds_bpermute(((q & 7) * 8 + (lane >> 3)) * 4, (int)row_e[q >> 3]);
            const bool ok = v2 != 0xffffffffu;
            tc.hoff[q] = ok ? v2 + lc_term : 0u;
            tc.hvalid |= ok ? (1u << q) : 0u;
        }
    };
    auto stage_window = [&](const Tile &tc) __attribute__((always_inline)) {
        const rsrc_t rs = make_rsrc(tc.img);
#pragma unroll
        for (int q = 0; q < kWsPiecesPerWave; ++q) {
            const int piece = gw + 4 * q;
            if (piece * 8 >= kWsRows) continue; // static after unrolling for q < 10; q == 10: waves 0..2 only (wave-uniform)
            lds_dma16_buf(rs, ((tc.hvalid >> q) & 1u) ? tc.hoff[q] : 0xffffffffu, 0u, win + piece * 1024);
        }
    };

    // ---- prologue: all nine weight slabs (every thread one 16-byte piece per tap) + group 0's first window
    {
        const int row = tid >> 3, wp = tid & 7;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        const unsigned wvoff = (unsigned)(((long long)row * a.Kpad + (wp ^ key) * 8) * (long long)sizeof(T));
        const rsrc_t rs = make_rsrc(a.w);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) lds_dma16_buf(rs, wvoff, (unsigned)(tap * 64 * (int)sizeof(T)), wts + tap * 8192 + (8 * wave) * 128);
    }
    Tile cur;
    cur.o0 = cur.xs = 0, cur.img = nullptr, cur.hvalid = 0;
    if (grp == 0 && n_mine > 0) {
        setup_tile(0, cur);
        stage_window(cur);
    }
    // accumulators start at the bias
    floatx4 acc[TC][TP];
    auto arm_acc = [&]() __attribute__((always_inline)) { // the bias is re-read per tile (64 B per lane, cache resident): 16 registers that need not live through the multiply phase
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const float4 b = *reinterpret_cast<const float4 *>(a.bias + lg * NV + i * 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = (floatx4){b.x, b.y, b.z, b.w};
        }
    };
    arm_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // fragment addressing (conv3x3_halo_kernel, BN = 64)
    const int wrow_l = (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
    const int prow0 = gw * WP + lr;
    T *out = reinterpret_cast<T *>(a.out);
    const T *res = reinterpret_cast<const T *>(a.res);
    const int cb = lg * NV;

    auto load_frags = [&](int step, uint4 (&pf)[TP], uint4 (&wf)[TC]) __attribute__((always_inline)) { // step static after unrolling
        const int tap = step >> 1, kh = step & 1;
        const int base = prow0 + (tap / 3) * pitch + (tap % 3);
        unsigned pa = base * 128 + ((lg ^ (base & 7)) << 4);
        unsigned wa = tap * 8192 + wfrag0;
        if (kh) pa ^= 64u, wa ^= 64u;
#pragma unroll
        for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(win + pa + j * 2048);
#pragma unroll
        for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(wts + wa + i * 512);
    };
    // ---- deferred (woven) part of the previous tile's epilogue: sums, residual values and store offsets of its last NWV pixel tiles
    constexpr int NB = NWV > 0 ? NWV : 1;
    constexpr int kPieces = 4 * NWV; // one floatx4 (4 couts of one pixel) per piece
    floatx4 accB[TC][NB];
    half8 rresB[NB][2];
    unsigned ooffB[NB]; // byte offset of the lane's 16 couts of that pixel in `out`; kJunkOff for junk pixels: beyond the descriptor's range, also after the
                        // + 16 of a pixel's second store (0xffffffff would wrap to 15 and land inside the tensor), so the hardware drops the store
    constexpr unsigned kJunkOff = 0xf0000000u, kOutRange = 0xe0000000u;
#pragma unroll
    for (int jw = 0; jw < NB; ++jw) {
        ooffB[jw] = kJunkOff;
#pragma unroll
        for (int i = 0; i < TC; ++i) accB[i][jw] = (floatx4){0.f, 0.f, 0.f, 0.f};
        rresB[jw][0] = rresB[jw][1] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.out) + (long long)a.out_coff * (long long)sizeof(T), 0,
                                                                            (int)kOutRange, 0x00020000);
    // step of the multiply phase that hosts piece p: the pieces are spread evenly over steps 0 .. 15
    auto compute_tile = [&]() __attribute__((always_inline)) {
        constexpr bool WV = NWV > 0; // the woven pieces run in EVERY multiply phase: a group's first tile has nothing pending and weaves zeros whose stores are
                                     // dropped (out-of-range offsets) — one instruction stream, no branch, no second copy of the 288-MFMA body
        // Two fragment register sets: the eight ds_read_b128 of step s+1 are issued BETWEEN the MFMAs of step s (one read per two
        // MFMAs), so a wave that has the SIMD's matrix pipe to itself never waits for LDS.  hipcc's scheduler otherwise sinks every
        // read to just before its first use (one register set, the LDS latency exposed 18 times per tile): the sched_barrier /
        // sched_group_barrier calls pin the order.
        uint4 pf[2][TP], wf[2][TC];
        uint2 packed[2]; // a pixel's 8 finished couts (two pieces) on their way to one 16-byte store
        if (a.slabs & 8) __builtin_amdgcn_s_setprio(3); // the multiplying wave wins the SIMD's issue arbitration; its partner (stage + epilogue) takes the gaps
        load_frags(0, pf[0], wf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) load_frags(s + 1, pf[(s + 1) & 1], wf[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wf[s & 1][i], pf[s & 1][j], acc[i][j], (T *)nullptr);
            // piece p = (pixel tile jw, cout quad i) of the deferred tile rides on this step when p * 16 / kPieces == s
            bool hosts = false;
            if constexpr (WV) {
#pragma unroll
                for (int p = 0; p < kPieces; ++p) {
                    if (p * 16 / kPieces != s) continue;
                    hosts = true;
                    const int jw = p >> 2, i = p & 3;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = accB[i][jw][r];
                    wtk_silu_scaled_run<4, true>(v); // scalar add / multiply: a packed fp32 instruction costs 27-32 issue cycles beside MFMAs
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = wtk_pin_f32(v[r] + (float)rresB[jw][i >> 1][(i & 1) * 4 + r]); // zeros when the layer has no residual
                    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                    const half4 h4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    packed[i & 1] = __builtin_bit_cast(uint2, h4);
                    if (i & 1) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 d = {packed[0].x, packed[0].y, packed[1].x, packed[1].y};
                        __builtin_amdgcn_raw_buffer_store_b128(d, out_rs, (int)(ooffB[jw] + (unsigned)((i >> 1) * 16)), 0, 0);
                    }
                }
            }
            if (s + 1 < 18) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); // 2 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // 1 DS read
                    if (WV && hosts) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); // 4 VALU of the woven piece
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.slabs & 8) __builtin_amdgcn_s_setprio(0);
    };
    // `defer_tail`: the last NWV pixel tiles are not finished here — their sums, residual values and store offsets go to the deferred set and
    // ride on the group's next multiply phase (only when the group HAS a next tile; the last tile of a group is finished whole)
    auto epilogue = [&](const Tile &tc, bool defer_tail) __attribute__((always_inline)) {
        int pix_e, col_e;
        halo_out_pixel(a, tc.o0 + gw * WP, tc.xs, lane, pix_e, col_e);
        long long pixj[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) pixj[j] = lane_fetch(j * 16 + lr, pix_e);
        // the residual of ALL four pixel tiles is requested before any arithmetic (junk pixels read pixel 0): one exposed memory
        // latency per tile instead of four — this phase is the long pole of an interval (stamped), every cycle it waits counts
        half8 rraw[TP][2];
        if (res) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const T *rp = res + (pixj[j] < 0 ? 0 : pixj[j]) * a.res_ld + a.res_coff + cb;
                rraw[j][0] = *reinterpret_cast<const half8 *>(rp);
                rraw[j][1] = *reinterpret_cast<const half8 *>(rp + 8);
            }
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            if (NWV > 0 && j >= TP - NWV && defer_tail) { // wave-uniform
                const int jw = j - (TP - NWV);
#pragma unroll
                for (int i = 0; i < TC; ++i) accB[i][jw] = acc[i][j];
                if (res) rresB[jw][0] = rraw[j][0], rresB[jw][1] = rraw[j][1]; // (no residual: they stay the zeros they were initialised to)
                ooffB[jw] = pixj[j] >= 0 ? (unsigned)((pixj[j] * a.out_ld + cb) * (long long)sizeof(T)) : kJunkOff;
                continue;
            }
            float v[NV];
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
            if (a.act) {
                wtk_silu_scaled_run<NV, (WTK_SILU_SCALAR_MASK & 1) != 0>(v);
            }
            if (res) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v[i] += (float)rraw[j][i >> 3][i & 7];
                    if constexpr ((WTK_SILU_SCALAR_MASK & 1) != 0) v[i] = wtk_pin_f32(v[i]);
                }
            }
            if (pixj[j] >= 0) store_run_h<NV>(out + pixj[j] * a.out_ld + a.out_coff + cb, v);
        }
    };

    // ---- interval schedule: group g multiplies in intervals i with (i & 1) == g, the other group is in Q; one barrier per interval
    Tile done; // tile whose accumulators are waiting for their epilogue
    done = cur;
#ifdef WTK_WS64_STAMPS
    const bool stamp_on = a.dbg_stamps != nullptr && blockIdx.x < 2 && lane == 0;
    unsigned long long *stamp = a.dbg_stamps + ((long long)blockIdx.x * 8 + wave) * 16 * 4;
    if (stamp_on) stamp[16 * 4 - 1] = __builtin_amdgcn_s_memtime(); // slot 15.3: loop entry
#endif
    for (int i = 0; i < intervals; ++i) {
#ifdef WTK_WS64_STAMPS
        if (stamp_on && i < 15) stamp[i * 4 + 0] = __builtin_amdgcn_s_memtime();
#endif
        if ((i & 1) == grp) {
            const int k = (i - grp) >> 1;
            if (k < n_mine) {
                compute_tile();
                done = cur;
            }
        } else {
            const int kprev = (i - 1 - grp) >> 1, knext = (i + 1 - grp) >> 1;
            const bool has_prev = i - 1 - grp >= 0 && kprev < n_mine, has_next = knext < n_mine;
            if (has_next) {
                setup_tile(knext, cur);
                stage_window(cur); // the group finished reading its window before the last barrier
            }
            if (has_prev) {
                epilogue(done, has_next);
                arm_acc();
            }
#ifdef WTK_WS64_STAMPS
            if (stamp_on && i < 15) stamp[i * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef WTK_WS64_STAMPS
        if (stamp_on && i < 15) stamp[i * 4 + 2] = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef WTK_WS64_STAMPS
        if (stamp_on && i < 15) stamp[i * 4 + 3] = __builtin_amdgcn_s_memtime();
#endif
    }
}

hipError_t launch_ws64(HaloArgs a, int num_cus, hipStream_t stream) {
    const long long tiles = (long long)a.strips * a.blocks_per_strip;
    if (tiles <= 0 || tiles > 0x3fffffffLL || num_cus < 8) return hipErrorInvalidValue;
    if (256 + 2 * a.pitch + 2 > kWsRows) return hipErrorInvalidValue;
    a.d_nct = make_fastdiv(1u);
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv((unsigned)a.strips);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    a.d_h1 = make_fastdiv((unsigned)(a.H + 1));
    const long long cap = num_cus / 8 * 8; // one block per CU (158 KiB of LDS)
    const long long want = (tiles + 1) / 2;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    a.grid = (int)grid;
    // (the woven-epilogue schedules of round 3 — NWV = 1 .. 3 pixel tiles of a wave's four riding on the next multiply phase — measured no gain and are no
    // longer instantiated: NWV = 0 is round 2's schedule)
    hipLaunchKernelGGL(conv3x3_ws64_kernel<0>, dim3(grid), dim3(512), 0, stream, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 / STRIDE-2 convolution with an LDS-resident window (fp16, 128-cout tile): model.5 / 7 / 16 / 19 of YOLOv8s.
//
// The implicit-GEMM kernel re-stages the pixel operand for each of the nine taps (one LDS-DMA request per 4 MFMAs and wave).
// Here the input is read as its four PARITY PLANES: plane (py, px) holds input pixels (2Y + py, 2X + px), i.e. one pixel per
// output pixel, and in plane coordinates the stride-2 conv is a stride-1 conv whose taps are 1-D shifts of the flat window —
// exactly the geometry of conv3x3_halo_kernel on the OUTPUT map (stacked images, one shared zero row, pitch = Wo + 1).
// Output (y, x) needs input rows 2y-1, 2y, 2y+1 = plane rows (y-1, py=1), (y, py=0), (y, py=1), columns alike, so
//   plane (1,1) serves 4 taps (kh, kw in {0, 2}), planes (0,1) and (1,0) two each, plane (0,0) one (kh = kw = 1): 9 in all.
// A 64-channel chunk of one plane is staged ONCE (its per-lane request addresses gather the plane out of the NHWC tensor:
// the plane's offset is a wave-uniform constant on top of plane (0,0)'s per-row offsets) and multiplied by 1 / 2 / 2 / 4 taps:
// 2.25 taps per staged window instead of 1.  Weight slabs: the three-slab ring of conv3x3_halo_kernel (slab of step g+2
// requested in step g, counted vmcnt).  Window pieces are requested ahead of the slab pieces of a step, so "all but the slab
// requests of this step" (vmcnt(WR)) at the last tap of a plane means the next plane's window has landed.
// K is walked chunk-major / plane-major (the implicit-GEMM kernel: tap-major), so results equal that kernel's up to fp32
// summation order, not bit for bit (WTK_NO_S2WIN=1 switches back; tests compare within one fp16 ulp of the activations' scale).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kS2Rows = 344; // window rows per buffer (43 pieces): BMT + (Wo + 1) + 2 <= 344  ->  Wo <= 85 at BMT = 256

// SPLIT: split-fp16 operands (see conv3x3_halo_kernel): pseudo-channel a.Cin / in_ld / out_ld, three MFMAs per tile pair and K step.
template <int BMT, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_kernel(const HaloArgs a) {
    asm volatile("" ::"s"(a.in), "s"(a.w), "s"(a.bias), "s"(a.in_ld), "s"(a.in_coff), "s"(a.N), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CoutPad), "s"(a.Kpad),
                 "s"(a.S), "s"(a.pitch), "s"(a.d_pitch.mul), "s"(a.d_pitch.sh1), "s"(a.d_pitch.sh2), "s"(a.d_nct.mul), "s"(a.d_nct.sh1), "s"(a.d_nct.sh2),
                 "s"(a.d_h1.mul), "s"(a.d_h1.sh1), "s"(a.d_h1.sh2), "s"(a.grid));
    using T = _Float16;
    constexpr int BN = 128, WAVES_C = 2, WAVES_P = 4, WC = 64, WP = BMT / WAVES_P, TP = WP / 16, TC = 4, NV = 16, WR = 2;
    constexpr int kPieces = kS2Rows / 8;              // 43
    constexpr int KW = (kPieces + 7) / 8;             // window pieces per wave (6)
    __shared__ __attribute__((aligned(16))) char win0[kS2Rows * 128];
    __shared__ __attribute__((aligned(16))) char win1[kS2Rows * 128];
    __shared__ __attribute__((aligned(16))) char wbuf0[BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf1[BN * 128];
    __shared__ __attribute__((aligned(16))) char wbuf2[BN * 128];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int lr = lane & 15, lg = lane >> 4;
    const int nct = a.CoutPad / BN;
    int nwg = a.grid;
    if (a.n_dyn) { // dynamic batch (see conv3x3_halo_kernel)
        const int lim = min(max(*a.n_dyn, 0), a.N) * (a.H + 1) * a.pitch;
        const int live = ((lim + BMT - 1) / BMT) * nct;
        if ((int)blockIdx.x >= live) return;
        nwg = min(nwg, live);
    }
    int L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const unsigned t = fdiv((unsigned)L, a.d_nct);
    const int n0 = (L - (int)t * nct) * BN;
    const int o0 = (int)t * BMT; // one strip: row blocks only
    const int pitch = a.pitch;   // Wo + 1
    const int halo_rows = BMT + pitch + 2;
    const int Hin = 2 * a.H, Win = 2 * a.W; // a.H, a.W: OUTPUT map (the geometry lives there)
    const int n_base = (int)fdiv(fdiv((unsigned)o0, a.d_pitch), a.d_h1);
    const char *img = reinterpret_cast<const char *>(reinterpret_cast<const T *>(a.in) + (long long)n_base * Hin * Win * a.in_ld + a.in_coff);
    const T *wgt = reinterpret_cast<const T *>(a.w);

    // ---- window rows of this wave's pieces (wave, wave + 8, ...): byte offset of plane (0,0)'s pixel (2Y, 2X), evaluated once
    unsigned hoff[KW];
    unsigned hvalid = 0;
    {
        const int hr_e = (wave + 8 * (lane >> 3)) * 8 + (lane & 7);
        int pn = 0, Y = 0, X = 0;
        const bool ok_e = (lane >> 3) < KW && hr_e < halo_rows && halo_in_coords(a, o0 + hr_e, 0, pn, Y, X);
        const unsigned pixel = __umul24(__umul24((unsigned)(pn - n_base), (unsigned)Hin) + (unsigned)(2 * Y), (unsigned)Win) + (unsigned)(2 * X);
        const unsigned row_e = ok_e ? __umul24(pixel, (unsigned)(a.in_ld * (int)sizeof(T))) : 0xffffffffu;
        const unsigned lc_term = (unsigned)((((lane & 7) ^ ((lane >> 3) & 7)) * 8) * (int)sizeof(T));
#pragma unroll
        for (int q = 0; q < KW; ++q) {
            const unsigned v = (unsigned)__builtin_amdgcn_ds_bpermute((q * 8 + (lane >> 3)) * 4, (int)row_e);
            const bool ok = v != 0xffffffffu;
            hoff[q] = ok ? v + lc_term : 0u;
            hvalid |= ok ? (1u << q) : 0u;
        }
    }
    const rsrc_t irs = make_rsrc(img);
    // window piece q of this wave for plane (py, px), channel chunk c (never skipped: constant request count per step)
    auto issue_piece = [&](char *buf, int q, int py, int px, int c) __attribute__((always_inline)) { // q static after unrolling
        const bool back = q > 0 && wave + 8 * q >= kPieces; // wave-uniform: the last slot of the highest waves re-requests their previous piece
        const int piece = __builtin_amdgcn_readfirstlane(back ? wave + 8 * (q - 1) : wave + 8 * q);
        const unsigned off = back ? hoff[q > 0 ? q - 1 : 0] : hoff[q];
        const bool ok = back ? ((hvalid >> (q > 0 ? q - 1 : 0)) & 1u) : ((hvalid >> q) & 1u);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((((py * Win + px) * a.in_ld) + c * 64) * (int)sizeof(T)); // wave-uniform
        lds_dma16_buf(irs, ok ? off : 0xffffffffu, so, buf + piece * 1024);
    };
    const int wrow0 = tid >> 3, wp = tid & 7;
    unsigned wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = wrow0 + 64 * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wvoff[i] = (unsigned)(((long long)row * a.Kpad + (wp ^ key) * 8) * (long long)sizeof(T));
    }
    const rsrc_t wrs = make_rsrc(wgt + (long long)n0 * a.Kpad);
    auto issue_weights = [&](char *buf, int tap, int c) __attribute__((always_inline)) {
        const unsigned so = (unsigned)((tap * a.Cin + c * 64) * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < WR; ++i) lds_dma16_buf(wrs, wvoff[i], so, buf + (64 * i + 8 * wave) * 128);
    };

    const int cb = n0 + wave_c * WC + lg * NV;
    floatx4 acc[TC][TP];
    floatx4 acc1[SPLIT ? TC : 1][SPLIT ? TP : 1];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const floatx4 b4 = (floatx4){a.bias[cb + i * 4 + 0], a.bias[cb + i * 4 + 1], a.bias[cb + i * 4 + 2], a.bias[cb + i * 4 + 3]};
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            acc[i][j] = b4;
            if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const int wrow_l = wave_c * WC + (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
    const int prow0 = wave_p * WP + lr;
    auto compute_tap = [&](const char *win, const char *wb, int tapoff) __attribute__((always_inline)) {
        const int base = prow0 + tapoff;
        const unsigned pfrag0 = base * 128 + ((lg ^ (base & 7)) << 4);
        if constexpr (SPLIT) {
            uint4 ph[TP], wh[TC], wl[TC], pl[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) ph[j] = *reinterpret_cast<const uint4 *>(win + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wh[i] = *reinterpret_cast<const uint4 *>(wb + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) wl[i] = *reinterpret_cast<const uint4 *>(wb + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int j = 0; j < TP; ++j) pl[j] = *reinterpret_cast<const uint4 *>(win + (pfrag0 ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    mma_h(wh[i], ph[j], acc[i][j], (T *)nullptr);
                    mma_h(wl[i], ph[j], acc1[i][j], (T *)nullptr);
                }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wh[i], pl[j], acc1[i][j], (T *)nullptr);
            return;
        }
#pragma unroll
        for (int kh2 = 0; kh2 < 2; ++kh2) {
            const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0;
            const unsigned wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
            uint4 pf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(win + pa + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(wb + wa + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_h(wf[i], pf[j], acc[i][j], (T *)nullptr);
        }
    };

    // ---- the nine steps of a channel chunk.  Plane order (1,1) [4 taps], (0,1) [2], (1,0) [2], (0,0) [1]; window buffers alternate per
    // plane (even number of planes per chunk: the parity is static).  kStep*: static tables, indexed by the unrolled step.
    //                         step:   0  1  2  3   4  5   6  7   8
    constexpr int kStepTap[9] =      { 0, 2, 6, 8,  3, 5,  1, 7,  4};              // kh * 3 + kw of the packed weights
    constexpr int kStepDY[9] =       {-1,-1, 0, 0,  0, 0, -1, 0,  0};
    constexpr int kStepDX[9] =       {-1, 0,-1, 0, -1, 0,  0, 0,  0};
    constexpr int kStepBuf[9] =      { 0, 0, 0, 0,  1, 1,  0, 0,  1};              // window buffer of the step's plane
    constexpr int kStepReq[9] =      { 2, 2, 2, 0,  KW,0,  KW,0,  KW};             // next plane's window pieces requested in this step
    constexpr int kStepReqFrom[9] =  { 0, 2, 4, 0,  0, 0,  0, 0,  0};              // first piece slot of that request
    constexpr int kNextPy[9] =       { 0, 0, 0, 0,  1, 1,  0, 0,  1};              // plane whose window the step requests:
    constexpr int kNextPx[9] =       { 1, 1, 1, 1,  0, 0,  0, 0,  1};              //   (0,1) (0,1) (0,1) - (1,0) - (0,0) - next chunk's (1,1)
    static_assert(KW == 6, "request schedule written for six window pieces per wave");
    const int nchunks = a.Cin / 64;

    // ---- prologue: window of chunk 0 / plane (1,1) + slabs of steps 0 and 1
#pragma unroll
    for (int q = 0; q < KW; ++q) issue_piece(win0, q, 1, 1, 0);
    issue_weights(wbuf0, kStepTap[0], 0);
    issue_weights(wbuf1, kStepTap[1], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one step; S is a compile-time constant so that every table entry, buffer choice and request count folds
    auto step = [&](auto s_tag, int c, bool more) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value;
        const char *wcur = S % 3 == 0 ? wbuf0 : (S % 3 == 1 ? wbuf1 : wbuf2);
        char *wnext2 = (S + 2) % 3 == 0 ? wbuf0 : ((S + 2) % 3 == 1 ? wbuf1 : wbuf2);
        const char *wcur_win = kStepBuf[S] ? win1 : win0;
        char *wnext_win = kStepBuf[S] ? win0 : win1;
        compute_tap(wcur_win, wcur, (kStepDY[S] + 1) * pitch + (kStepDX[S] + 1));
        // requests after the step's reads and MFMAs; window pieces BEFORE slab pieces (they are the older ones for the counted wait)
        const int cn = S == 8 ? (more ? c + 1 : c) : c;
        if constexpr (kStepReq[S] > 0) {
#pragma unroll
            for (int q = 0; q < kStepReq[S]; ++q) issue_piece(wnext_win, kStepReqFrom[S] + q, kNextPy[S], kNextPx[S], cn);
        }
        if constexpr (S < 7)
            issue_weights(wnext2, kStepTap[S + 2], c);
        else
            issue_weights(wnext2, kStepTap[S - 7], more ? c + 1 : c);
        // last step of a plane (3, 5, 7, 8): everything but this step's slab requests has landed -> the next plane's window is complete
        constexpr bool plane_end = S == 3 || S == 5 || S == 7 || S == 8;
        wait_vmcnt(plane_end ? WR : WR + kStepReq[S]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        step(std::integral_constant<int, 0>{}, c, more);
        step(std::integral_constant<int, 1>{}, c, more);
        step(std::integral_constant<int, 2>{}, c, more);
        step(std::integral_constant<int, 3>{}, c, more);
        step(std::integral_constant<int, 4>{}, c, more);
        step(std::integral_constant<int, 5>{}, c, more);
        step(std::integral_constant<int, 6>{}, c, more);
        step(std::integral_constant<int, 7>{}, c, more);
        step(std::integral_constant<int, 8>{}, c, more);
    }

    // ---- epilogue (bias already in the accumulators)
    int pix_e, col_e;
    halo_out_pixel(a, o0 + wave_p * WP, 0, lane, pix_e, col_e);
    long long pixj[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) pixj[j] = lane_fetch(j * 16 + lr, pix_e);
    T *out = reinterpret_cast<T *>(a.out);
    if (cb + NV <= a.Cout) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            if (pixj[j] < 0) continue;
            float v[NV];
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (SPLIT)
                        v[i * 4 + r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
                    else
                        v[i * 4 + r] = acc[i][j][r];
                }
            if (a.act) {
                wtk_silu_scaled_run<NV>(v);
            }
            if constexpr (SPLIT)
                wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out) + pixj[j] * a.out_ld + a.out_coff, cb, v);
            else
                store_run_h<NV>(out + pixj[j] * a.out_ld + a.out_coff + cb, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the last chunk's duplicate requests must not outlive the block's LDS
}

template <int BMT, bool SPLIT = false> hipError_t launch_s2(HaloArgs a, hipStream_t stream) {
    const long long blocks = (long long)a.blocks_per_strip * (a.CoutPad / 128);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / 128));
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv(1u);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    a.d_h1 = make_fastdiv((unsigned)(a.H + 1));
    a.grid = (int)blocks;
    hipLaunchKernelGGL((conv3x3_s2_kernel<BMT, SPLIT>), dim3((unsigned)blocks), dim3(512), 0, stream, a);
    return hipGetLastError();
}

template <typename T, int BN, int HROWS> hipError_t launch_hp(HaloArgs a, int num_cus, hipStream_t stream) {
    const long long tiles = (long long)a.strips * a.blocks_per_strip * (a.CoutPad / BN);
    if (tiles <= 0 || tiles > 0x3fffffffLL || num_cus < 8) return hipErrorInvalidValue;
    if (kBM + 2 * a.pitch + 2 > HROWS || (a.Cin / (8 * ElemH<T>::CE)) % 2 != 0) return hipErrorInvalidValue;
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / BN));
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv((unsigned)a.strips);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    a.d_h1 = make_fastdiv((unsigned)(a.H + 1));
    const long long cap = num_cus / 8 * 8; // one block per CU (156-160 KB of LDS); a multiple of 8 keeps a block's tiles on its XCD label
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    a.grid = (int)grid;
    hipLaunchKernelGGL((conv3x3_halo_pkernel<T, BN, HROWS>), dim3(grid), dim3(512), 0, stream, a);
    return hipGetLastError();
}

template <typename T, int BN, int NHALO, int MINW, int NWB, int HROWS, int BMT = 256, bool TAIL = false, bool SPLIT = false> hipError_t launch_h(HaloArgs a, hipStream_t stream) {
    const long long blocks = (long long)a.strips * a.blocks_per_strip * (a.CoutPad / BN);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (BMT + 2 * a.pitch + 2 > HROWS || (long long)a.blocks_per_strip * BMT < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / BN));
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv((unsigned)a.strips);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    a.d_h1 = make_fastdiv((unsigned)(a.H + 1));
    a.grid = (int)blocks;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, BN, NHALO, MINW, NWB, HROWS, BMT, TAIL, SPLIT>), dim3((unsigned)blocks), dim3(512), 0, stream, a);
    return hipGetLastError();
}

} // namespace

bool halo_eligible(int k, int stride, int cin, int is_f16) {
    const int cch = is_f16 ? 64 : 32;
    return k == 3 && stride == 1 && cin % cch == 0;
}

// split-fp16 operands: real channel counts here; 64- or 128-cout tiles only (two accumulator sets)
bool split_halo_eligible(int k, int stride, int cin, int cout) { return k == 3 && stride == 1 && cin % 32 == 0 && cout % 64 == 0; }
int split_halo_cout_tile(int cout_stored) { return cout_stored % 128 == 0 ? 128 : 64; }

// Every channel count / offset of `a` but Cout / CoutPad in pseudo-channels (2 x real); three weight slabs, one tile per block
hipError_t launch_conv3x3_halo_split(const HaloArgs &a, hipStream_t stream) {
    const int bn = (a.narrow && !a.tail_w) ? 64 : split_halo_cout_tile(a.Cout);
    if (a.Cin % 64 != 0 || a.CoutPad % bn != 0 || a.Cout != a.CoutPad || a.slabs == 2) return hipErrorInvalidValue;
    // fused 1x1 tail: 64 -> 64 couts (box towers), split weights [64][tail_kpad = 128 pseudo-channels]; 128 -> <= 32 stored couts with fp32 output
    // (class towers), split weights [32][tail_kpad = 256 pseudo-channels]
    if (a.tail_w && a.Cout == 128) {
        if (a.res || a.out2 || !a.tail_bias || !a.tail_out || a.tail_kpad != 256 || a.tail_cout < 1 || a.tail_cout > 32 || !a.tail_f32 || a.tail_ld % 4 || a.tail_coff % 4) return hipErrorInvalidValue;
    } else if (a.tail_w && (a.Cout != 64 || a.res || a.out2 || !a.tail_bias || !a.tail_out || a.tail_kpad != 128 || a.tail_cout != 64 || (a.tail_f32 ? (a.tail_ld % 4 || a.tail_coff % 4) : (a.tail_ld % 64 || a.tail_coff % 64))))
        return hipErrorInvalidValue;
    if (a.in_ld % 64 || a.in_coff % 64 || a.out_ld % 64 || a.out_coff % 64 || a.Kpad != 9 * a.Cin) return hipErrorInvalidValue;
    if (a.pitch != (a.strips == 1 ? a.S + 1 : a.S + 2) || kBM + 2 * a.pitch + 2 > kHaloRowsMax || a.strips * a.S < a.W || (a.strips == 1 && a.S != a.W))
        return hipErrorInvalidValue;
    const int bm = a.bm == 128 ? 128 : kBM;
    if ((long long)a.blocks_per_strip * bm < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % 64 || a.res_coff % 64)) return hipErrorInvalidValue;
    if (a.out2 && (a.out2_ld % 64 || a.out2_coff % 64)) return hipErrorInvalidValue;
    if (bn == 128 && a.tail_w) return bm == 128 ? launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 128, true, true>(a, stream) : launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 256, true, true>(a, stream);
    if (bn == 128) return bm == 128 ? launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 128, false, true>(a, stream) : launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 256, false, true>(a, stream);
    if (a.tail_w) return bm == 128 ? launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 128, true, true>(a, stream) : launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 256, true, true>(a, stream);
    // six-slab ring: the 128-pixel tile only (the 256-pixel tile, 24 MFMAs per wave and tap, measured the same with either ring)
    if (a.deep && bm == 128) return launch_h<_Float16, 64, 2, 1, 6, kHaloRowsMax, 128, false, true>(a, stream);
    return bm == 128 ? launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 128, false, true>(a, stream) : launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 256, false, true>(a, stream);
}

int halo_rows_max(int cout_stored, int slabs) { return (slabs == 3 && halo_cout_tile(cout_stored) == 192) ? kHaloRowsSmall : kHaloRowsMax; }

void halo_geometry(int H, int W, int rows_max, int *S, int *pitch, int *strips, int *blocks_per_strip, int bm) {
    const int smax = (rows_max - kBM - 2) / 2 - 2; // 256 + 2*(S+2) + 2 <= rows_max (the same strips for both block sizes)
    *strips = (W + smax - 1) / smax;
    *S = (W + *strips - 1) / *strips;
    *pitch = *S + 2;
    *blocks_per_strip = (H * *pitch + bm - 1) / bm;
}

void halo_geometry_stacked(int N, int H, int W, int rows_max, int *S, int *pitch, int *strips, int *blocks_per_strip, int bm) {
    const int smax = (rows_max - kBM - 2) / 2 - 2; // 256 + 2*(S+2) + 2 <= rows_max (the same strips for both block sizes)
    *strips = (W + smax - 1) / smax;
    *S = (W + *strips - 1) / *strips;
    *pitch = *strips == 1 ? *S + 1 : *S + 2; // one strip: the zero column right of a row is the one left of the next row
    *blocks_per_strip = (int)(((long long)N * (H + 1) * *pitch + bm - 1) / bm);
}

// stride-2 window kernel: a.H / a.W are the OUTPUT map; one strip (pitch = W + 1); a.bm = 128 selects the half-size block
bool s2win_eligible(int k, int stride, int cin, int cout, int cout_pad, int is_f16, int wo, bool plain) {
    // cin >= 128: the 64-channel strided conv (model.3) keeps the implicit-GEMM kernel, whose fused-tail form (model.3 + model.4.cv1) must stay
    // bit-identical to its stand-alone form (test_fused_kernels_equal_layer_by_layer switches the fusion off and on)
    return is_f16 && k == 3 && stride == 2 && cin % 64 == 0 && cin >= 128 && cout_pad % 128 == 0 && cout % 16 == 0 && plain && 256 + (wo + 1) + 2 <= kS2Rows;
}

hipError_t launch_conv3x3_s2(const HaloArgs &a, hipStream_t stream) {
    if (a.Cin % 64 || a.CoutPad % 128 || a.Cout > a.CoutPad || a.Cout % 16 || a.out2 || a.tail_w || a.res || a.Kpad < 9 * a.Cin || a.Kpad % 64) return hipErrorInvalidValue;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8) return hipErrorInvalidValue;
    const int bm = a.bm == 128 ? 128 : 256;
    if (a.strips != 1 || a.S != a.W || a.pitch != a.W + 1 || bm + a.pitch + 2 > kS2Rows) return hipErrorInvalidValue;
    if ((long long)a.blocks_per_strip * bm < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    return bm == 128 ? launch_s2<128>(a, stream) : launch_s2<256>(a, stream);
}

// split-fp16 operands: real channel counts here
bool split_s2win_eligible(int k, int stride, int cin, int cout, int cout_pad, int wo, bool plain) {
    return k == 3 && stride == 2 && cin % 32 == 0 && cout_pad % 128 == 0 && cout == cout_pad && plain && 256 + (wo + 1) + 2 <= kS2Rows;
}
// a.Cin / Kpad / in_ld / in_coff / out_ld / out_coff in pseudo-channels (2 x real), a.Cout real
hipError_t launch_conv3x3_s2_split(const HaloArgs &a, hipStream_t stream) {
    if (a.Cin % 64 || a.CoutPad % 128 || a.Cout != a.CoutPad || a.out2 || a.tail_w || a.res || a.Kpad != 9 * a.Cin) return hipErrorInvalidValue;
    if (a.in_ld % 64 || a.in_coff % 64 || a.out_ld % 64 || a.out_coff % 64) return hipErrorInvalidValue;
    const int bm = a.bm == 128 ? 128 : 256;
    if (a.strips != 1 || a.S != a.W || a.pitch != a.W + 1 || bm + a.pitch + 2 > kS2Rows) return hipErrorInvalidValue;
    if ((long long)a.blocks_per_strip * bm < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    return bm == 128 ? launch_s2<128, true>(a, stream) : launch_s2<256, true>(a, stream);
}

int ws64_rows_max() { return kWsRows; }

bool ws64_eligible(int k, int stride, int cin, int cout, int cout_pad, int is_f16, bool has_out2, bool has_tail) {
    return is_f16 && k == 3 && stride == 1 && cin == 64 && cout == 64 && cout_pad == 64 && !has_out2 && !has_tail;
}

hipError_t launch_conv3x3_ws64(const HaloArgs &a, int num_cus, hipStream_t stream) {
    if (a.Cin != 64 || a.Cout != 64 || a.CoutPad != 64 || a.out2 || a.tail_w || a.Kpad < 576 || a.Kpad % 64) return hipErrorInvalidValue;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % 8 || a.res_coff % 8)) return hipErrorInvalidValue;
    if (a.pitch != (a.strips == 1 ? a.S + 1 : a.S + 2) || a.strips * a.S < a.W || (a.strips == 1 && a.S != a.W)) return hipErrorInvalidValue;
    if ((long long)a.blocks_per_strip * 256 < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    return launch_ws64(a, num_cus, stream);
}

int halo_cout_tile(int cout_stored) { return cout_stored % 128 == 0 ? 128 : (cout_stored % 192 == 0 ? 192 : 64); }

hipError_t launch_conv3x3_halo(const HaloArgs &a, int is_f16, hipStream_t stream) {
    const int ce = is_f16 ? 8 : 4;
    const int cch = 8 * ce;
    const int bn = (a.narrow && !a.tail_w && a.CoutPad % 64 == 0) ? 64 : halo_cout_tile(a.Cout); // narrow: 64-cout tiles for a thin grid (small handles)
    if (a.Cin % cch != 0 || a.CoutPad % bn != 0 || a.Cout > a.CoutPad || a.Cout % (bn == 192 ? 24 : 16) != 0) return hipErrorInvalidValue;
    if (a.in_ld % ce || a.in_coff % ce || a.out_ld % ce || a.out_coff % ce || a.Kpad % cch || a.Kpad < 9 * a.Cin) return hipErrorInvalidValue;
    if (a.pitch != (a.strips == 1 ? a.S + 1 : a.S + 2) || kBM + 2 * a.pitch + 2 > kHaloRowsMax || a.strips * a.S < a.W || (a.strips == 1 && a.S != a.W))
        return hipErrorInvalidValue;
    if (a.tail_w && (!is_f16 || (bn != 64 && bn != 128) || a.Cout != bn || a.CoutPad != bn || a.res || a.out2 || !a.tail_bias || !a.tail_out ||
                     a.tail_kpad < bn || a.tail_kpad % 8 || a.tail_ld % 8 || a.tail_coff % 8 || a.slabs == 2))
        return hipErrorInvalidValue; // the fused 1x1 tail exists for the fp16 64-cout (-> 64) and 128-cout (-> 32) three-slab variants only
    const int bm = a.bm == 128 ? 128 : kBM;
    if ((long long)a.blocks_per_strip * bm < (long long)a.N * (a.H + 1) * a.pitch) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % ce || a.res_coff % ce)) return hipErrorInvalidValue;
    if (a.out2 && (a.out2_ld % ce || a.out2_coff % ce)) return hipErrorInvalidValue;
    const int nchunks = a.Cin / cch;
    // three weight slabs + counted vmcnt (default) or the two-slab / vmcnt(0) schedule (slabs == 2).  A 192-cout tile with three
    // slabs only leaves room for 352-row windows: the planner then cuts wide maps into strips of <= 45 columns.
    // persistent form only where a block gets to walk several tiles (measured: -5..-8 % at 6-7 tiles per CU, -1..2 % at 1.75, but
    // +4 % when every block has exactly one tile: its per-tile bookkeeping then buys nothing)
    const long long tiles = (long long)a.strips * a.blocks_per_strip * (a.CoutPad / bn);
    if (a.slabs != 2 && !a.tail_w && bm == kBM && a.persist_cus > 0 && 2 * tiles >= 3 * (long long)a.persist_cus && nchunks % 2 == 0 && (bn == 128 || bn == 192)) {
        if (is_f16) return bn == 128 ? launch_hp<_Float16, 128, kHaloRowsMax>(a, a.persist_cus, stream) : launch_hp<_Float16, 192, kHaloRowsSmall>(a, a.persist_cus, stream);
        return bn == 128 ? launch_hp<float, 128, kHaloRowsMax>(a, a.persist_cus, stream) : launch_hp<float, 192, kHaloRowsSmall>(a, a.persist_cus, stream);
    }
    if (a.tail_w && bn == 128) { // class tower: one tile per block (both window buffers hold the exchange tiles afterwards)
        if (bm == 128) return launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 128, true>(a, stream);
        return launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 256, true>(a, stream);
    }
    if (a.tail_w) { // validated above: fp16, 64-cout tile, three slabs
        if (bm == 128) return nchunks == 1 ? launch_h<_Float16, 64, 1, 4, 3, kHaloRowsMax, 128, true>(a, stream) : launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 128, true>(a, stream);
        return nchunks == 1 ? launch_h<_Float16, 64, 1, 4, 3, kHaloRowsMax, 256, true>(a, stream) : launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 256, true>(a, stream);
    }
    if (a.slabs != 2 && bm == 128) { // half-size blocks: small maps that would otherwise leave CUs without a block
        if (is_f16) {
            if (bn == 128) return launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax, 128>(a, stream);
            if (bn == 192) return launch_h<_Float16, 192, 2, 2, 3, kHaloRowsSmall, 128>(a, stream);
            if (nchunks == 1) return launch_h<_Float16, 64, 1, 4, 3, kHaloRowsMax, 128>(a, stream);
            return launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax, 128>(a, stream);
        }
        if (bn == 128) return launch_h<float, 128, 2, 2, 3, kHaloRowsMax, 128>(a, stream);
        if (bn == 192) return launch_h<float, 192, 2, 2, 3, kHaloRowsSmall, 128>(a, stream);
        if (nchunks == 1) return launch_h<float, 64, 1, 4, 3, kHaloRowsMax, 128>(a, stream);
        return launch_h<float, 64, 2, 2, 3, kHaloRowsMax, 128>(a, stream);
    }
    if (bm != kBM) return hipErrorInvalidValue;
    if (a.slabs != 2) {
        if (is_f16) {
            if (bn == 128) return launch_h<_Float16, 128, 2, 2, 3, kHaloRowsMax>(a, stream);
            if (bn == 192) return launch_h<_Float16, 192, 2, 2, 3, kHaloRowsSmall>(a, stream);
            if (nchunks == 1) return launch_h<_Float16, 64, 1, 4, 3, kHaloRowsMax>(a, stream);
            return launch_h<_Float16, 64, 2, 2, 3, kHaloRowsMax>(a, stream);
        }
        if (bn == 128) return launch_h<float, 128, 2, 2, 3, kHaloRowsMax>(a, stream);
        if (bn == 192) return launch_h<float, 192, 2, 2, 3, kHaloRowsSmall>(a, stream);
        if (nchunks == 1) return launch_h<float, 64, 1, 4, 3, kHaloRowsMax>(a, stream);
        return launch_h<float, 64, 2, 2, 3, kHaloRowsMax>(a, stream);
    }
    if (is_f16) {
        if (bn == 128) return launch_h<_Float16, 128, 2, 2, 2, kHaloRowsMax>(a, stream);
        if (bn == 192) return launch_h<_Float16, 192, 2, 2, 2, kHaloRowsMax>(a, stream);
        if (nchunks == 1) return launch_h<_Float16, 64, 1, 4, 2, kHaloRowsMax>(a, stream);
        return launch_h<_Float16, 64, 2, 2, 2, kHaloRowsMax>(a, stream);
    }
    if (bn == 128) return launch_h<float, 128, 2, 2, 2, kHaloRowsMax>(a, stream);
    if (bn == 192) return launch_h<float, 192, 2, 2, 2, kHaloRowsMax>(a, stream);
    if (nchunks == 1) return launch_h<float, 64, 1, 4, 2, kHaloRowsMax>(a, stream);
    return launch_h<float, 64, 2, 2, 2, kHaloRowsMax>(a, stream);
}

} // namespace wtk
