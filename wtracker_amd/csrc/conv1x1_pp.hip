// 1x1 / stride-1 convolution (a plain GEMM over pixels) as TWO FOUR-WAVE GROUPS THAT ALTERNATE inside one workgroup, gfx950.
//
// Why (round 4, stamped with tools/igemm_stamps_split.hip): conv_igemm_kernel runs two independent 128 x 128 blocks of four waves per CU.  A K step
// costs a wave ~3 300 cycles against 768 cycles of MFMAs: ~1 000 cycles ISSUING its eight LDS-DMA requests (the CU's texture-address unit takes a 1-KiB
// request every 16 cycles, and all eight waves of the CU queue there at once: 8 x 16 = 128 cycles per request and wave), ~1 700 multiplying (stretched:
// the SIMD partner — the other block's wave — multiplies at the same time), ~450 waiting for the stage it requested at the top of the step, ~140 at the
// block barrier.  The two blocks drift INTO phase (both are throttled by the same shared unit), so the matrix pipe is busy 46 % of a step.
//
// Here the two blocks are the two halves of one 512-thread workgroup (waves 0-3 = group A, waves 4-7 = group B: waves w and w + 4 share a SIMD), each
// with its own tile stream, stage buffers (2 x 32 KB) and accumulators, and the workgroup barrier keeps them HALF A STEP APART:
//     quiet phase:     wait for the stage requested a step ago; request the next stage (8 requests per wave); run the epilogue of a finished tile
//     barrier
//     multiply phase:  fragment reads + MFMAs of the current stage
//     barrier
// with group B one barrier behind group A.  While a wave multiplies, its SIMD partner issues requests / runs its SiLU epilogue, so the requests see
// half the queue (4 waves) and the MFMAs an uncontended pipe; a stage has a full step to land (requested in one quiet phase, awaited at the top of the
// next).  Arithmetic is conv_igemm_kernel's (K1 form): accumulators start at the bias, K walked upwards, the same MFMA sequence per accumulator, SiLU in
// the log2(e)-scaled domain, one rounding at the store: bit-identical outputs (WTK_NO_PP_1X1=1 switches back to conv_igemm_kernel).
#include "wtk_kernels.h"

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBM = 128, kBN = 128;           // tile of ONE group
constexpr int kStage = (kBM + kBN) * 128;     // 32 KB: 128 pixel rows + 128 weight rows of 128 bytes
constexpr int kTP = 4, kTC = 4, kNV = 16;     // wave tile 64 px x 64 cout

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

__device__ __forceinline__ void mma16(const uint4 &wf, const uint4 &pf, floatx4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wf), __builtin_bit_cast(half8, pf), acc, 0, 0, 0);
}

// SPLIT: split-fp16 operands (wtk_kernels.h): a 128-byte row is 32 channels as [hi32 | lo32]; channel-like arguments in pseudo-channels.
// NT: the pixel rows are read by exactly one cout tile (CoutPad == 128): non-temporal hint on their requests.
template <bool SPLIT, bool NT> __global__ __launch_bounds__(512, 2) void conv1x1_pp_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) char sa0[kStage];
    __shared__ __attribute__((aligned(16))) char sa1[kStage];
    __shared__ __attribute__((aligned(16))) char sb0[kStage];
    __shared__ __attribute__((aligned(16))) char sb1[kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2; // 0: group A, 1: group B (SIMD partners of group A's waves)
    const int gw = wave & 3;
    const int wave_p = gw >> 1, wave_c = gw & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int tg = tid & 255; // thread inside its group
    char *const st0 = grp ? sb0 : sa0;
    char *const st1 = grp ? sb1 : sa1;

    constexpr int BKE = 64; // K elements (halves) per step = one 128-byte row
    const int nk = a.Kpad / BKE;
    const int nct = a.CoutPad / kBN;
    int ptiles_eff = a.ptiles;
    if (a.n_dyn) { // dynamic batch: only the pixel tiles that start inside the first *n_dyn images
        const long long n_eff = min(max(*a.n_dyn, 0), a.N);
        ptiles_eff = (int)min((long long)a.ptiles, (n_eff * a.Ho * a.Wo + kBM - 1) / kBM);
    }
    const int total_tiles = ptiles_eff * nct;
    // tile schedule: every (block, group) is a virtual four-wave block; tiles (pixel tile major, cout tile minor) are cut into 8 contiguous ranges, one
    // per XCD label (blockIdx % 8), and the virtual blocks of a label walk their range with stride = their number (conv_igemm_kernel's order)
    auto schedule = [&](int g, int &t_begin, int &my_tiles, int &t_stride) __attribute__((always_inline)) {
        const int G = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = (bid >> 3) * 2 + g;
        const int nbx = ((G - xcd + 7) >> 3) * 2;
        const int q = total_tiles >> 3, r = total_tiles & 7;
        const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int count = xcd < r ? q + 1 : q;
        t_begin = first + slot;
        t_stride = nbx;
        my_tiles = slot < count ? (count - slot + nbx - 1) / nbx : 0;
    };
    int t_begin, my_tiles, t_stride, ob, other_tiles, os;
    schedule(grp, t_begin, my_tiles, t_stride);
    schedule(grp ^ 1, ob, other_tiles, os);
    const int my_total = my_tiles * nk;
    const int block_total = max(my_tiles, other_tiles) * nk; // both groups run this many phase pairs (the shorter one idles through the rest)
    if (block_total == 0) return;

    // ---- staging assignment inside the group (conv_igemm_kernel with four waves): thread -> 16-byte physical chunk `ch` of rows r0 + 32 i
    const int ch = tg & 7, r0 = tg >> 3;
    const int lchunk = ch ^ (r0 & 7);
    const _Float16 *in = reinterpret_cast<const _Float16 *>(a.in);
    unsigned wvoff[4], poff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        const int key = ((row >> 1) & 1) | (((row / kNV) & 3) << 1);
        wvoff[i] = (unsigned)(((long long)row * a.Kpad + (ch ^ key) * 8) * 2);
    }
    rsrc_t in_rs = {0, 0, 0, 0}, w_rs = {0, 0, 0, 0};
    int ld_ks = 0, ld_i = 0;
    auto setup_loader = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int n0 = (tile - ptile * nct) * kBN;
        in_rs = make_rsrc(in + (long long)ptile * kBM * a.in_ld + a.in_coff);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = (long long)ptile * kBM + r0 + 32 * r;
            poff[r] = m < a.M ? (unsigned)(((r0 + 32 * r) * a.in_ld + lchunk * 8) * 2) : 0xffffffffu;
        }
        w_rs = make_rsrc(reinterpret_cast<const _Float16 *>(a.w) + (long long)n0 * a.Kpad);
        ld_ks = 0;
    };
    auto issue_stage = [&](char *pt) __attribute__((always_inline)) {
        char *wt = pt + kBM * 128;
        const unsigned so = (unsigned)(ld_ks * (BKE * 2));
        const bool k_ok = lchunk * 8 + ld_ks * BKE < a.Cin; // K tail of the last step is zero
#pragma unroll
        for (int r = 0; r < 4; ++r) dma_buf<NT>(in_rs, k_ok ? poff[r] : 0xffffffffu, so, pt + (32 * r + 8 * gw) * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_buf<false>(w_rs, wvoff[i], so, wt + (32 * i + 8 * gw) * 128);
        if (++ld_ks == nk) {
            if (++ld_i < my_tiles) setup_loader(ld_i);
        }
    };

    floatx4 acc[kTC][kTP];
    floatx4 acc1[SPLIT ? kTC : 1][SPLIT ? kTP : 1];
    const int prow_l = wave_p * 64 + lr;
    const unsigned pfrag0 = prow_l * 128 + ((lg ^ (prow_l & 7)) << 4);
    const int wrow_l = wave_c * 64 + (lr >> 2) * kNV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / kNV) & 3) << 1);
    const unsigned wfrag0 = kBM * 128 + wrow_l * 128 + ((lg ^ wkey_l) << 4);

    auto compute = [&](const char *pt) __attribute__((always_inline)) {
        if constexpr (SPLIT) { // one K step = the hi fragments (k-half 0) and the lo fragments (k-half 1) of the same 32 channels
            uint4 ph[kTP], wh[kTC], wl[kTC];
#pragma unroll
            for (int j = 0; j < kTP; ++j) ph[j] = *reinterpret_cast<const uint4 *>(pt + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < kTC; ++i) wh[i] = *reinterpret_cast<const uint4 *>(pt + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < kTC; ++i) wl[i] = *reinterpret_cast<const uint4 *>(pt + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int i = 0; i < kTC; ++i)
#pragma unroll
                for (int j = 0; j < kTP; ++j) {
                    mma16(wh[i], ph[j], acc[i][j]);
                    mma16(wl[i], ph[j], acc1[i][j]);
                }
            uint4 pl[kTP];
#pragma unroll
            for (int j = 0; j < kTP; ++j) pl[j] = *reinterpret_cast<const uint4 *>(pt + (pfrag0 ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < kTC; ++i)
#pragma unroll
                for (int j = 0; j < kTP; ++j) mma16(wh[i], pl[j], acc1[i][j]);
        } else {
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2) {
                const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0;
                const unsigned wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
                uint4 pf[kTP], wf[kTC];
#pragma unroll
                for (int j = 0; j < kTP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(pt + pa + j * 2048);
#pragma unroll
                for (int i = 0; i < kTC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(pt + wa + i * 512);
#pragma unroll
                for (int i = 0; i < kTC; ++i)
#pragma unroll
                    for (int j = 0; j < kTP; ++j) mma16(wf[i], pf[j], acc[i][j]);
            }
        }
    };

    float bias_r[kNV];
    auto load_bias = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * kBN + wave_c * 64 + lg * kNV;
#pragma unroll
        for (int e = 0; e < kNV; ++e) bias_r[e] = a.bias[cb + e];
    };
    auto arm_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kTC; ++i)
#pragma unroll
            for (int j = 0; j < kTP; ++j) {
                acc[i][j] = (floatx4){bias_r[i * 4 + 0], bias_r[i * 4 + 1], bias_r[i * 4 + 2], bias_r[i * 4 + 3]};
                if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
    };

    _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
    auto epilogue = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * kBN + wave_c * 64 + lg * kNV;
        if (i + 1 < my_tiles) load_bias(i + 1); // (with one cout tile the values repeat; the loads hide behind the SiLU work either way)
        if (cb + kNV <= a.Cout) { // padded output channels are never stored
#pragma unroll
            for (int j = 0; j < kTP; ++j) {
                const long long pix = (long long)ptile * kBM + wave_p * 64 + j * 16 + lr;
                if (pix >= a.M) continue;
                float v[kNV];
#pragma unroll
                for (int t = 0; t < kTC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (SPLIT)
                            v[t * 4 + r] = wtk_split_value(acc[t][j][r], acc1[t][j][r]);
                        else
                            v[t * 4 + r] = acc[t][j][r];
                    }
                if (a.act) wtk_silu_scaled_run<kNV, (WTK_SILU_SCALAR_MASK & 2) != 0>(v);
                if constexpr (SPLIT) {
                    wtk_split_store<kNV>(out + pix * a.out_ld + a.out_coff, cb, v);
                } else {
                    _Float16 *o = out + pix * a.out_ld + a.out_coff + cb;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        half8 hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[h * 8 + e];
                        *reinterpret_cast<half8 *>(o + h * 8) = hv;
                    }
                }
            }
        }
        arm_acc();
    };

    // ---- prologue: stage 0 of this group's first tile
    if (my_tiles > 0) {
        setup_loader(0);
        load_bias(0);
        arm_acc();
        issue_stage(st0);
    }
    if (grp) { // group B runs one barrier behind group A
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    int cp_ks = 0, cp_i = 0;
    bool pend = false; // a finished tile waits for its epilogue (it runs in the next quiet phase, beside the partner's MFMAs)
    for (int p = 0; p < block_total; ++p) {
        // ---- quiet phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of stage p (requested a step ago) has landed; older stores have retired
        if (p + 1 < my_total) issue_stage((p & 1) ? st0 : st1); // that buffer was multiplied in the previous multiply phase
        if (pend) {
            epilogue(cp_i);
            ++cp_i;
            pend = false;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- multiply phase
        if (p < my_total) {
            compute((p & 1) ? st1 : st0);
            if (++cp_ks == nk) {
                cp_ks = 0;
                pend = true;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (!grp) { // group A is one barrier ahead: meet group B's last one
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (pend) epilogue(cp_i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

} // namespace

bool conv1x1_pp_eligible(const ConvArgs &a, int split) {
    const int mult = split ? 64 : 8;
    return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.in2 && !a.res && !a.out2 && !a.tail_w && !a.out_f32 && a.tile_w == 0 && a.CoutPad % kBN == 0 &&
           a.Cout % 16 == 0 && a.Kpad % 64 == 0 && a.H == a.Ho && a.W == a.Wo && a.in_ld % mult == 0 && a.in_coff % mult == 0 && a.out_ld % mult == 0 &&
           a.out_coff % mult == 0 && a.Cin % mult == 0;
}

hipError_t launch_conv1x1_pp(ConvArgs a, int split, hipStream_t stream) {
    if (!conv1x1_pp_eligible(a, split)) return hipErrorInvalidValue;
    if (a.Kpad < a.Cin || a.Cout > a.CoutPad || a.M <= 0 || a.M > 0x7fffffffLL) return hipErrorInvalidValue;
    const long long ptiles = (a.M + kBM - 1) / kBM;
    const long long tiles = ptiles * (a.CoutPad / kBN);
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    a.ptiles = (int)ptiles;
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / kBN));
    const int cus = current_device_cus();
    if (cus <= 0) return hipErrorUnknown;
    // two tile streams per block, 128 KB of LDS: one block per CU.  The tile ranges are per XCD label (blockIdx % 8): every label that owns tiles
    // needs a block, so at least min(tiles, 8) blocks
    long long want = (tiles + 1) / 2;
    if (want > cus) want = cus;
    const long long floor8 = tiles < 8 ? tiles : 8;
    const unsigned grid = (unsigned)(want < floor8 ? floor8 : want);
    const bool nt = a.CoutPad == kBN;
    if (split) {
        if (nt)
            hipLaunchKernelGGL((conv1x1_pp_kernel<true, true>), dim3(grid), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((conv1x1_pp_kernel<true, false>), dim3(grid), dim3(512), 0, stream, a);
    } else {
        if (nt)
            hipLaunchKernelGGL((conv1x1_pp_kernel<false, true>), dim3(grid), dim3(512), 0, stream, a);
        else
            hipLaunchKernelGGL((conv1x1_pp_kernel<false, false>), dim3(grid), dim3(512), 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace wtk
