// 1x1 / stride-1 convolution (a plain GEMM over pixels) with a wide tile and a three-stage LDS ring, fp16, gfx950.
//
// Why a second GEMM kernel: conv_igemm.hip stages 32 KB per 128 x 128 x 64 step (64 FLOP per staged byte) into two stage
// buffers, so with two blocks per CU at most 64 KB are in flight per CU while a step's requests take ~2 us to land from L2:
// PMC shows its waves parked in s_waitcnt / barriers 47 % of their cycles and the matrix pipe busy 20 %.  Here
//   * the block tile is 256 px x 128 cout with FOUR waves of 128 px x 64 cout (8 x 4 accumulator tiles, 0.375 ds_read_b128 per
//     MFMA): 85 FLOP per staged byte, and a pixel row is re-read from L2 by half as many cout tiles' worth of blocks;
//   * K is consumed in 32-element steps: operand rows are 64 B, a stage is 16 KB of pixels + 8 KB of weights, and THREE stages
//     (72 KB) still leave room for two blocks per CU: 96 KB per CU in flight;
//   * the stage of step s+2 is requested while step s is multiplied (inline-asm LDS-DMA, invisible to hipcc's own wait
//     insertion) and a COUNTED s_waitcnt vmcnt(6) leaves exactly those six requests of the wave in flight across the raw
//     s_barrier — the scheme of conv3x3_halo.hip's three weight slabs;
//   * 64-byte rows: a ds_read_b128 fragment read touches 16 rows x one 16-byte chunk per 16-lane group; rows r and r+4 share
//     banks, so the chunk is XOR-ed with key(row) = {0,2,3,1}[(row >> 2) & 3] (pixels) / [(row >> 4) & 3] (weights, whose
//     rows are read through the cout permutation below): every group then covers all 64 banks once.  The swizzle is applied on
//     the SOURCE side of the LDS-DMA (the lane landing on physical chunk p of a row fetches logical chunk p ^ key).
// Arithmetic is conv_igemm_kernel's: accumulators start at the bias, K is walked upwards in 32-element MFMA steps
// (v_mfma_f32_16x16x32_f16), SiLU in the log2(e)-scaled domain, one fp16 rounding at the store: bit-identical outputs.
#include "wtk_kernels.h"

namespace wtk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBM = 256, kBN = 128, kKS = 32;      // block tile, K elements per step
constexpr int kStageA = kBM * 64, kStageB = kBN * 64; // bytes
constexpr int kStage = kStageA + kStageB;            // 24 KB

// one 1-KiB LDS-DMA request in buffer form (SGPR resource + wave-uniform byte offset + per-lane 32-bit offset; a lane offset of
// 0xffffffff is out of range and lands zeros): 5-10 % less wave time per request than the flat form (tools/lds_dma_rate.hip)
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
template <bool NT> __device__ __forceinline__ void dma16(const rsrc_t &rs, unsigned voff, unsigned soff, char *lds_dst) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}

// NT: the pixel rows are read by exactly one cout tile (CoutPad == 128): non-temporal hint on their requests
template <bool NT> __global__ __launch_bounds__(256, 2) void conv1x1_wide_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) char st0[kStage];
    __shared__ __attribute__((aligned(16))) char st1[kStage];
    __shared__ __attribute__((aligned(16))) char st2[kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave >> 1, wave_c = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- persistent tile schedule (as conv_igemm_kernel): tiles = (pixel tile major, cout tile minor), cut into 8 contiguous
    // ranges, one per XCD label (blockIdx % 8); the blocks of a label walk their range with stride = #blocks of that label
    const int nct = a.CoutPad / kBN;
    const int total_tiles = a.ptiles * nct;
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

    const _Float16 *in = reinterpret_cast<const _Float16 *>(a.in);
    const int nk = a.Kpad / kKS;

    // ---- loader.  One LDS-DMA instruction fills 16 rows x 64 B: lane -> row lane >> 2, physical chunk lane & 3.
    // Pixel pieces of wave w: w, w+4, w+8, w+12 (rows piece*16 ..); weight pieces: w, w+4.
    const int prow = lane >> 2, pch = lane & 3;
    const int kperm = (0x78 >> (2 * ((prow >> 2) & 3))) & 3; // {0,2,3,1}[(row >> 2) & 3] packed as 0b01_11_10_00 (rows of a piece: row & 15 = prow)
    const int lchunk_a = pch ^ kperm;
    const int lchunk_b = pch ^ ((0x78 >> (2 * (wave & 3))) & 3); // weight piece p = wave + 4q: (row >> 4) & 3 = p & 3 = wave
    unsigned wvoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) wvoff[q] = (unsigned)((((long long)((wave + 4 * q) * 16 + prow)) * a.Kpad + lchunk_b * 8) * 2);

    unsigned avoff[4]; // byte offset of this lane's chunk in step 0 from the tile's first pixel (0xffffffff: past the last pixel)
    rsrc_t ars = {0, 0, 0, 0}, wrs = {0, 0, 0, 0};
    int ld_i = 0, ld_ks = 0;
    auto setup_loader = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int n0 = (tile - ptile * nct) * kBN;
        const long long m0 = (long long)ptile * kBM;
        ars = make_rsrc(in + m0 * a.in_ld + a.in_coff);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (wave + 4 * q) * 16 + prow;
            avoff[q] = m0 + row < a.M ? (unsigned)((row * a.in_ld + lchunk_a * 8) * 2) : 0xffffffffu;
        }
        wrs = make_rsrc(reinterpret_cast<const _Float16 *>(a.w) + (long long)n0 * a.Kpad);
        ld_ks = 0;
    };
    auto issue_stage = [&](char *st) __attribute__((always_inline)) {
        const bool k_ok = ld_ks * kKS < a.Cin; // steps past Cin (K padding) read zeros
        const unsigned so = (unsigned)(ld_ks * (kKS * 2));
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16<NT>(ars, k_ok ? avoff[q] : 0xffffffffu, so, st + (wave + 4 * q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16<false>(wrs, wvoff[q], so, st + kStageA + (wave + 4 * q) * 1024);
        if (++ld_ks == nk) {
            if (++ld_i < my_tiles) setup_loader(ld_i);
        }
    };

    // ---- fragment addresses.  Pixel tile j: rows wave_p*128 + j*16 + lr; weight tile i: rows wave_c*64 + (lr>>2)*16 + (lr&3) + 4i
    // (the permutation that gives a lane 16 consecutive couts); both keys reduce to {0,2,3,1}[(lr >> 2) & 3].
    const int fkey = (0x78 >> (2 * ((lr >> 2) & 3))) & 3;
    const unsigned pfrag0 = (unsigned)((wave_p * 128 + lr) * 64 + ((lg ^ fkey) << 4));
    const unsigned wfrag0 = (unsigned)(kStageA + (wave_c * 64 + (lr >> 2) * 16 + (lr & 3)) * 64 + ((lg ^ fkey) << 4));

    floatx4 acc[4][8];
    auto compute = [&](const char *st) __attribute__((always_inline)) {
        uint4 pf[8], wf[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = *reinterpret_cast<const uint4 *>(st + pfrag0 + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const uint4 *>(st + wfrag0 + i * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wf[i]), __builtin_bit_cast(half8, pf[j]), acc[i][j], 0, 0, 0);
    };

    // accumulators start at the bias of the tile (rows exist up to CoutPad)
    auto arm_acc = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * kBN + wave_c * 64 + lg * 16;
        const float4 *bp = reinterpret_cast<const float4 *>(a.bias + cb);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 b = bp[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] = (floatx4){b.x, b.y, b.z, b.w};
        }
    };

    _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
    auto epilogue = [&](int i) __attribute__((always_inline)) {
        const int tile = t_begin + i * t_stride;
        const int ptile = (int)fdiv((unsigned)tile, a.d_nct);
        const int cb = (tile - ptile * nct) * kBN + wave_c * 64 + lg * 16;
        if (cb + 16 > a.Cout) return; // padded output channels are never stored
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long pix = (long long)ptile * kBM + wave_p * 128 + j * 16 + lr;
            if (pix >= a.M) continue;
            float v[16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][j][r];
            if (a.act) wtk_silu_scaled_run<16, (WTK_SILU_SCALAR_MASK & 4) != 0>(v);
            _Float16 *o = out + pix * a.out_ld + a.out_coff + cb;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[h * 8 + e];
                *reinterpret_cast<half8 *>(o + h * 8) = hv;
            }
        }
    };

    // ---- flat pipeline over (tile, K step): ring slot of step s = s % 3
    const int total = my_tiles * nk;
    setup_loader(0);
    arm_acc(0);
    issue_stage(st0);
    if (total > 1) issue_stage(st1);
    if (total > 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // step 0 has landed, step 1 may still fly
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cp_ks = 0, cp_i = 0, slot = 0;
    for (int s = 0; s < total; ++s) {
        char *cur = slot == 0 ? st0 : (slot == 1 ? st1 : st2);
        char *nxt2 = slot == 0 ? st2 : (slot == 1 ? st0 : st1); // slot of step s+2 = slot of step s-1: free since the last barrier
        compute(cur);
        const bool more = s + 2 < total;
        if (more) issue_stage(nxt2);
        const bool tile_end = ++cp_ks == nk;
        if (tile_end) {
            epilogue(cp_i);
            cp_ks = 0;
            ++cp_i;
            if (cp_i < my_tiles) arm_acc(cp_i);
        }
        // everything older than this step's six requests has landed (step s+1's stage); at a tile end the stores of the
        // epilogue are younger than the requests, so the count no longer isolates them: drain
        if (more && !tile_end)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
    }
}

} // namespace

bool conv1x1_wide_eligible(const ConvArgs &a, int is_f16) {
    return is_f16 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.in2 && !a.res && !a.out2 && a.CoutPad % kBN == 0 && a.Cin % kKS == 0 &&
           a.Kpad % kKS == 0 && a.Cout % 16 == 0 && a.H == a.Ho && a.W == a.Wo;
}

hipError_t launch_conv1x1_wide(ConvArgs a, hipStream_t stream) {
    if (!conv1x1_wide_eligible(a, 1)) return hipErrorInvalidValue;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8 || a.Kpad < a.Cin || a.Cout > a.CoutPad) return hipErrorInvalidValue;
    if (a.M <= 0 || a.M > 0x7fffffffLL) return hipErrorInvalidValue;
    const long long ptiles = (a.M + kBM - 1) / kBM;
    const long long tiles = ptiles * (a.CoutPad / kBN);
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    a.ptiles = (int)ptiles;
    a.d_nct = make_fastdiv((unsigned)(a.CoutPad / kBN));
    const int cus = current_device_cus();
    if (cus <= 0) return hipErrorUnknown;
    const long long resident = 2LL * cus; // 72 KB of LDS per block
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    if (a.CoutPad == kBN)
        hipLaunchKernelGGL((conv1x1_wide_kernel<true>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv1x1_wide_kernel<false>), dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
