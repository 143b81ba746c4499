// Internal declarations shared by the HIP translation units of libwtk_hip.so.
// Everything here is gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, MFMA, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wtk {

// Unsigned division by a launch-invariant divisor (Granlund-Montgomery / libdivide branch-free form):
// q = (t + ((n - t) >> sh1)) >> sh2 with t = umulhi(n, mul) (Granlund-Montgomery; sh1 = min(l, 1), sh2 = max(l - 1, 0),
// l = ceil(log2 d)).  Exact for every 32-bit n and every d >= 1, branch-free (d = 1: mul = 1, both shifts 0 -> q = n); ~5 ops
// instead of the ~25-instruction software division hipcc emits for n / d with a runtime d.
struct FastDiv {
    unsigned mul, sh1, sh2;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    if (d == 0) d = 1;
    unsigned l = 0;
    while ((1ull << l) < d) ++l; // ceil(log2 d)
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.mul = (unsigned)m;
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
    const unsigned t = __umulhi(n, f.mul);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// CU count of the CURRENT HIP device, cached per device id (a process may hold handles on several GPUs).
inline int current_device_cus() {
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cache[dev] = prop.multiProcessorCount;
    }
    return cache[dev];
}

// Pins an fp32 value in a VGPR (no instruction).  hipcc folds "(half)(a * b)" into v_fma_mixlo_f16, which rounds the exact
// product ONCE to fp16, whenever the conversion is not paired into a v_cvt_pk_f16_f32 — which depends on the surrounding
// code.  Every SiLU of the library pins its product so that all kernels (stand-alone and fused) round product -> fp32 -> fp16
// and stay bit-identical to each other.
__device__ __forceinline__ float wtk_pin_f32(float v) {
    asm("" : "+v"(v));
    return v;
}

// Activations live in a SCALED domain: every SiLU layer's weights-and-bias are packed so that its accumulator holds
// a = log2(e) * x, and it stores y = a / (1 + 2^-a) = log2(e) * SiLU(x).  The consumer of a scaled tensor divides its weights by
// log2(e) when packing (for a SiLU layer fed by a SiLU layer the two factors cancel: only the bias is scaled; the stem scales its
// weights, the linear Detect outputs divide theirs), so logits and boxes come out unscaled.  What it buys: exp(-x) = 2^(-a) needs no
// multiply — SiLU is v_exp (with a free negate), v_add, v_rcp, v_mul: one VALU instruction less per output value in epilogues that
// are VALU bound (-1 % end to end).  Concatenation, residual adds, max-pooling and 2x upsampling commute with a positive scale.
constexpr float kActScale = 1.4426950408889634f; // log2(e)
__device__ __forceinline__ float wtk_silu_scaled(float a) {
    const float e = __builtin_amdgcn_exp2f(-a);
    return wtk_pin_f32(a * __builtin_amdgcn_rcpf(1.0f + e));
}

// SiLU of a run of values, two at a time: the add and the multiply go through v_pk_add_f32 / v_pk_mul_f32 (two IEEE fp32
// operations per instruction — the same results as the scalar form), so a pair costs 2 v_exp + 2 v_rcp + 2 packed ops instead of
// 2 v_exp + 2 v_rcp + 4 scalar ops in epilogues whose only work is this.
#ifndef WTK_SILU_SCALAR_MASK
#define WTK_SILU_SCALAR_MASK 0 // bit 0: conv3x3_ws64_kernel, bit 1: conv_igemm_kernel, bit 2: conv1x1_wide_kernel (A/B builds)
#endif
typedef float wtk_f2 __attribute__((ext_vector_type(2)));
// SCALAR: v_add_f32 / v_mul_f32 instead of the packed forms (the sums are pinned so that the SLP pass cannot re-pack them).  For epilogues
// that run BESIDE a wave issuing MFMAs: there a packed fp32 instruction costs 27-32 issue cycles against 6 alone (tools/ubench/valu_cost.hip).
// Same IEEE operations, bit-identical results.
template <int NV, bool SCALAR = false> __device__ __forceinline__ void wtk_silu_scaled_run(float (&v)[NV]) {
    static_assert(NV % 2 == 0, "pairs");
    if constexpr (SCALAR) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float e = __builtin_amdgcn_exp2f(-v[i]);
            const float d = wtk_pin_f32(1.0f + e);
            v[i] = wtk_pin_f32(v[i] * __builtin_amdgcn_rcpf(d));
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; i += 2) {
        const wtk_f2 a = {v[i], v[i + 1]};
        const wtk_f2 e = {__builtin_amdgcn_exp2f(-a.x), __builtin_amdgcn_exp2f(-a.y)};
        const wtk_f2 d = e + (wtk_f2){1.0f, 1.0f};
        const wtk_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const wtk_f2 y = a * r;
        v[i] = wtk_pin_f32(y.x);
        v[i + 1] = wtk_pin_f32(y.y);
    }
}

// ---------------------------------------------------------------------------------------------
// Split-fp16 storage ("f16x3" mode: fp32-grade results from the fp16 matrix pipe).  A value x is kept as the pair
//   hi = fp16(x),  lo = fp16((x - hi) * 2^11)          (x - hi is exact in fp32; the scale keeps lo out of the fp16 subnormals)
// and a product is x*w = hi_x*hi_w + 2^-11 (hi_x*lo_w + lo_x*hi_w) + O(2^-24 |x||w|): three v_mfma_f32_16x16x32_f16 (exact fp16
// products, fp32 accumulation; the 2^-11 terms in their own accumulator) instead of eight v_mfma_f32_16x16x4_f32 per 32 channels
// of K — 16 x the FLOP rate at 3 x the instructions.  Measured against float64 on wide-range data the error equals the plain
// fp32 dot product's (3.6e-7 vs 3.1e-7 of sum|x||w|, K = 1152).
// Layout: a tensor of C channels (C % 32 == 0) is an fp16 tensor of 2C pseudo-channels: per block of 32 channels 32 hi halves,
// then 32 lo halves (one 128-byte LDS row = the two k-halves the fp16 kernels already address).  Loaders, LDS-DMA and swizzles
// of the fp16 kernels work unchanged on pseudo-channels; only the MFMA pattern, the accumulators and the epilogue differ.
// Range: |x| must stay below the fp16 maximum (65504), as in the fp16 mode.
// ---------------------------------------------------------------------------------------------
constexpr float kSplitScale = 2048.0f, kSplitInv = 1.0f / 2048.0f;
// value of a split pair / of the two accumulators of a split product: hi + lo * 2^-11 as ONE fused multiply-add.  The product is exact (a power of
// two), so the single rounding of the fma is the rounding of the separate multiply + add this replaces — the same bits, one VALU op less per value
// in epilogues that are VALU bound (-ffp-contract=off keeps the compiler from fusing on its own, hence the explicit builtin)
__device__ __forceinline__ float wtk_split_value(float hi, float lo) { return __builtin_fmaf(lo, kSplitInv, hi); }
typedef _Float16 wtk_h8 __attribute__((ext_vector_type(8)));
// store NV (8 or 16) consecutive channels starting at real channel c (multiple of NV) of one pixel; `pix` = the pixel's pseudo-channel 0
template <int NV> __device__ __forceinline__ void wtk_split_store(_Float16 *pix, int c, const float (&v)[NV]) {
    static_assert(NV % 8 == 0 && NV <= 32, "runs of 8 channels inside one 32-channel block");
    _Float16 *p = pix + 64 * (c >> 5) + (c & 31);
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        wtk_h8 hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 h = (_Float16)v[i + j];
            hv[j] = h;
            lv[j] = (_Float16)((v[i + j] - (float)h) * kSplitScale);
        }
        *reinterpret_cast<wtk_h8 *>(p + i) = hv;
        *reinterpret_cast<wtk_h8 *>(p + 32 + i) = lv;
    }
}
template <int NV> __device__ __forceinline__ void wtk_split_load(const _Float16 *pix, int c, float (&v)[NV]) {
    static_assert(NV % 8 == 0 && NV <= 32, "runs of 8 channels inside one 32-channel block");
    const _Float16 *p = pix + 64 * (c >> 5) + (c & 31);
#pragma unroll
    for (int i = 0; i < NV; i += 8) {
        const wtk_h8 hv = *reinterpret_cast<const wtk_h8 *>(p + i), lv = *reinterpret_cast<const wtk_h8 *>(p + 32 + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i + j] = wtk_split_value((float)hv[j], (float)lv[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_igemm.hip).  Activations are NHWC; a tensor argument is a
// *channel-slice view* (base pointer, pixel stride `ld` in elements, first channel `coff`), so
// C2f / SPPF / FPN concatenations are never materialised: producers write into the slice of
// the consumer's buffer.
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const void *in;
    int in_ld, in_coff;
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int CoutPad; // Cout rounded up to the tile's BN; rows [Cout, CoutPad) of w/bias are zero
    int KH, KW, stride, pad;
    const void *w;     // packed weights [CoutPad][Kpad], K = (kh, kw, cin), cin fastest
    const float *bias; // [CoutPad]
    void *out;
    int out_ld, out_coff;
    void *out2; // optional second destination: 2x nearest-neighbour upsampled copy [N,2Ho,2Wo]
    int out2_ld, out2_coff;
    const void *res; // optional residual added after the activation (Bottleneck shortcut)
    int res_ld, res_coff;
    // optional second source of a 1x1 conv over a concat [up2x(low) | high]: input channels [0, in2_split) are read
    // from the half-resolution tensor `in2` at pixel (ho/2, wo/2) — nn.Upsample(2, "nearest") + Concat without
    // ever materialising the upsampled copy; channels >= in2_split come from `in` as usual
    const void *in2;
    int in2_ld, in2_coff, in2_split;
    int act; // 1: SiLU
    int K, Kpad;
    int tile_w; // 0: linear pixel order over N*Ho*Wo; >0: 2-D pixel tiles tile_w x (BM/tile_w)
    int tiles_x, tiles_y;
    long long M; // N*Ho*Wo
    int ptiles;  // pixel tiles (filled by launch_conv)
    FastDiv d_howo, d_wo, d_tilew, d_tilesx, d_tpi, d_nct; // filled by launch_conv
    const void *zeros; // >= 16 zero bytes in device memory (source of padded / out-of-range chunks)
    // optional fused 1x1 tail (fp16, 128x128 tile, Cout == CoutPad == 128, no residual / second output): the conv whose ONLY reader
    // is a 1x1 conv 128 -> 128 hands its SiLU'd fp16 tile over through LDS and the block multiplies it by tail_w right away;
    // `out` is then not written (model.3 -> model.4.cv1: 105 MB written and read back per 64 frames otherwise)
    const void *tail_w;     // packed [128][tail_kpad]
    const float *tail_bias; // [128]
    void *tail_out;
    int tail_kpad, tail_ld, tail_coff, tail_act;
    const int *n_dyn; // nullable: device-side image count <= N; pixel tiles that start beyond its last image are never visited
#ifdef WTK_IGEMM_STAMPS
    unsigned long long *dbg_stamps; // diagnostic builds only: [grid][waves][8] cycle sums
#endif
    int out_f32; // fp16 kernels only: `out` is an fp32 tensor (the six Detect output convs: head logits are never rounded to fp16)
    // split mode (launch_conv_split): in / in2 / res / out / out2 are split-fp16 tensors; in_ld, in_coff, Cin, K, Kpad, in2_*, res_*, out_ld,
    // out_coff are given in PSEUDO-channels (2 x the real ones); Cout / CoutPad stay real (an fp32 `out` keeps real out_ld / out_coff)
};

// Tile configurations (pixels x couts), all 4 waves / 256 threads
enum ConvCfg { CFG_128x128 = 0, CFG_256x64 = 1, CFG_256x32 = 2, CFG_128x64 = 4 };
int conv_cfg_bm(int cfg);
int conv_cfg_bn(int cfg);
// is_f16: 1 -> _Float16 storage + v_mfma_f32_16x16x32_f16; 0 -> fp32 + v_mfma_f32_16x16x4_f32
hipError_t launch_conv(const ConvArgs &a, int cfg, int is_f16, hipStream_t stream);
hipError_t launch_conv_split(const ConvArgs &a, int cfg, hipStream_t stream); // split-fp16 operands (see kSplitScale above)
hipError_t conv_init_attributes();
// Small-batch (latency) plan: split-K implicit GEMM, split-fp16 or fp32 operands (conv_sk.hip).  `a` as launch_conv_split / launch_conv take it.
int conv_sk_slices(int nk); // default K atoms of a layer of nk steps (32 channels of one tap each): a function of the layer alone
bool conv_sk_eligible(const ConvArgs &a, int split);
// K atoms of a layer for a handle whose calls bring up to M output pixels (cost model at M; the default conv_sk_slices(nk) unless clearly better)
int conv_sk_plan_atoms(long long M, int cout_pad, int nk, int num_cus, int split);
size_t conv_sk_partial_bytes(const ConvArgs &a, int split, int atoms = 0); // scratch of the op: [atoms][M][CoutPad] fp32 (0: one atom); atoms = 0: conv_sk_slices
// tickets: conv_sk_ticket_count(a) zero-initialised counters of this op (the last block of a tile combines the slabs in-kernel), or null (second launch: sk_finish_kernel)
size_t conv_sk_ticket_count(long long M, int cout_pad);
hipError_t launch_conv_sk(const ConvArgs &a, int split, int atoms, float *partial, unsigned *tickets, int num_cus, hipStream_t stream); // atoms = 0: conv_sk_slices(nk)
// ... and up to kSkGroupMax such convs that do not depend on each other as ONE launch (the latency plan's dependency levels, csrc/wtk_plan.hip: sk_schedule).
// force_tile 0..3 / force_form 0 (one block per atom), 1 (one block walks all atoms): test hooks, -1 = the cost model decides.  Bit-identical to n launches.
constexpr int kSkGroupMax = 4;
struct SkMember {
    ConvArgs a;
    int atoms;         // 0: conv_sk_slices(nk)
    float *partial;    // slab scratch of this conv (null: one atom)
    unsigned *tickets; // its arrival counters (null: sk_finish_kernel as a second launch)
};
// what the cost model decided for one launch at one batch size (kept by the caller: the model runs once, not per call)
struct SkChoice {
    int valid = 0;
    int separate = 0; // 1: the members' own launches, each with its own tile (many pixels: a common tile costs more than the launches)
    int tile = 3;     // the grouped launch's tile ...
    int S[kSkGroupMax] = {1, 1, 1, 1}; // ... and every member's form
    int m_tile[kSkGroupMax] = {3, 3, 3, 3}, m_S[kSkGroupMax] = {1, 1, 1, 1}; // separate launches
    double est_us = 0.0;
};
hipError_t launch_conv_sk_group(const SkMember *m, int n, int split, int num_cus, int force_tile, int force_form, hipStream_t stream, struct SkChoice *choice);
constexpr int kSkMaxCandidates = 4 * (1 << kSkGroupMax); // tiles x forms of a launch
int conv_sk_enumerate(const SkMember *m, int n, int split, int num_cus, int force_tile, int force_form, SkChoice *out, int cap);
// the kernel's view of one conv and of a grouped launch (here, not in conv_sk.hip, so that the host-side launch checker of tests/hostsan can read them)
struct SkArgs {
    const char *in;  // input tensor (slice view): byte pitch per pixel, byte offset of the first channel
    unsigned in_ldb, in_offb;
    const char *in2; // optional half-resolution source of the first in2_blocks 32-channel blocks (nn.Upsample(2x) + Concat, 1x1 only)
    unsigned in2_ldb, in2_offb;
    int in2_blocks;
    int N, H, W, Ho, Wo;
    int cpb;              // 32-channel blocks per tap
    int KW, stride, pad;  // square taps: KH == KW
    int nk, S;            // K steps in all (taps * cpb); blocks per tile along K: 1 or NA
    int NA;               // K atoms of the layer (conv_sk_slices): the unit of summation, see the header
    const char *w;        // [CoutPad][nk * 128 bytes]
    unsigned w_rowb;
    const float *bias;
    int Cout, CoutPad, act;
    void *out;
    int out_ld, out_coff, out_f32; // elements of the storage type (SPLIT: pseudo-channels unless out_f32), as ConvArgs
    const void *res;
    int res_ld, res_coff;
    float *partial; // S > 1: [S][M][CoutPad] fp32
    unsigned *tickets; // S > 1, nullable: one arrival counter per (pixel tile, cout tile), zero between launches — the block that arrives LAST combines the
                       // slabs itself (no second launch); null: sk_finish_kernel does it
    long long M;
    int ptiles, nct;
    FastDiv d_ptiles, d_nct, d_howo, d_wo, d_cpb, d_cg;
    const int *n_dyn;
};

struct SkGroupArgs {
    SkArgs p[kSkGroupMax];
    unsigned first[kSkGroupMax]; // first block of member i (first[0] = 0; unused members: 0xffffffff)
};
// fp16 1x1 / stride-1 convs with a 256 x 128 tile, 32-deep K steps and a three-stage LDS ring (conv1x1_wide.hip)
bool conv1x1_wide_eligible(const ConvArgs &a, int is_f16);
hipError_t launch_conv1x1_wide(ConvArgs a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS-resident input window (conv3x3_halo.hip).
// ---------------------------------------------------------------------------------------------
constexpr int kHaloRowsMax = 432;   // window rows (128 B each) one LDS buffer holds: 54 KiB
constexpr int kHaloRowsSmall = 352; // ... for the 192-cout tile with three weight slabs (2 x 44 KiB + 3 x 24 KiB = 160 KiB)
struct HaloArgs {
    const void *in;
    int in_ld, in_coff;
    int N, H, W, Cin;
    int Cout, CoutPad;
    const void *w;     // packed [CoutPad][Kpad], K = (tap, cin)
    const float *bias; // [CoutPad]
    void *out;
    int out_ld, out_coff;
    void *out2;
    int out2_ld, out2_coff;
    const void *res;
    int res_ld, res_coff;
    int act;
    int Kpad;
    int S, pitch, strips, blocks_per_strip; // column-strip geometry, see halo_geometry()
    const void *zeros;
    unsigned long long *dbg_stamps; // diagnostic builds (-DWTK_HALO_STAMPS) only
    // optional fused 1x1 tail (fp16, Cout == 64): out_tail[p][0..63] = tail_w[64][tail_kpad] . act(conv(p)) + tail_bias, no activation;
    // when set, `out` is NOT written (the intermediate stays on chip)
    const void *tail_w;
    const float *tail_bias;
    void *tail_out;
    int tail_kpad, tail_ld, tail_coff;
    int tail_cout; // channels actually stored by the tail (multiple of 8; 64 for the box tower, cls_ld for the class tower)
    int tail_f32;  // 1: tail_out is an fp32 tensor (head logits are kept in fp32 in both precision modes)
    int bm;          // flat output pixels per block of conv3x3_halo_kernel: 0 / 256 (default) or 128
    int persist_cus; // > 0: CU count; use the persistent form of the three-slab kernel where it exists (128 / 192-cout tiles, even chunk count)
    int slabs; // conv3x3_halo: 3 (default, also 0) = three weight slabs + counted vmcnt; 2 = two slabs, vmcnt(0) per tap
    int deep;       // split window kernel, 64-cout x 128-pixel tiles without a fused tail: 1 = six-slab weight ring + fragment prefetch (156 KB of LDS)
    int narrow; // split window kernel: 1 = 64-cout tiles for a layer of 128-multiple couts (twice the blocks; for grids that leave most CUs idle)
    int grid; // blocks of the launch (filled by the launchers: reading gridDim.x costs the set-up one more scalar-load round trip)
    const int *n_dyn; // nullable: device-side image count <= N (conv3x3_halo_kernel, conv3x3_s2_kernel): blocks whose tile starts beyond its last image exit at once
    FastDiv d_bps, d_strips, d_pitch, d_nct, d_h1; // filled by the launchers (d_h1: H + 1, conv3x3_halo.hip's stacked rows)
};
bool halo_eligible(int k, int stride, int cin, int is_f16);
bool split_halo_eligible(int k, int stride, int cin, int cout); // split-fp16 operands (real channel counts)
int split_halo_cout_tile(int cout_stored);
hipError_t launch_conv3x3_halo_split(const HaloArgs &a, hipStream_t stream); // channel counts / offsets but Cout in pseudo-channels
bool split_s2win_eligible(int k, int stride, int cin, int cout, int cout_pad, int wo, bool plain); // real channel counts
hipError_t launch_conv3x3_s2_split(const HaloArgs &a, hipStream_t stream);
int halo_rows_max(int cout_stored, int slabs); // window rows the kernel variant for this Cout can hold
void halo_geometry(int H, int W, int rows_max, int *S, int *pitch, int *strips, int *blocks_per_strip, int bm = 256); // per image (conv3x3_c32)
// conv3x3_halo.hip: the N images of a strip are stacked vertically with ONE shared zero row between neighbours, and a map that
// fits one strip shares ONE zero column between the right border of a row and the left border of the next (pitch = W + 1)
void halo_geometry_stacked(int N, int H, int W, int rows_max, int *S, int *pitch, int *strips, int *blocks_per_strip, int bm = 256);
int halo_cout_tile(int cout_stored);
hipError_t launch_conv3x3_halo(const HaloArgs &a, int is_f16, hipStream_t stream);
// 3x3 / stride-2 convs (fp16, 128-cout tiles) through an LDS-resident window of the four input parity planes; H / W of the args
// are the OUTPUT map, `plain` = no residual / second output / fused tail / second source
bool s2win_eligible(int k, int stride, int cin, int cout, int cout_pad, int is_f16, int wo, bool plain);
hipError_t launch_conv3x3_s2(const HaloArgs &a, hipStream_t stream);
// weight-stationary form for 64 -> 64 channel layers (fp16): all nine slabs resident in LDS, two wave groups alternating multiply / stage+epilogue
int ws64_rows_max();
bool ws64_eligible(int k, int stride, int cin, int cout, int cout_pad, int is_f16, bool has_out2, bool has_tail);
hipError_t launch_conv3x3_ws64(const HaloArgs &a, int num_cus, hipStream_t stream);
// thin fp16 layers (Cin = 32, Cout <= 96): conv3x3_c32.hip
bool c32_eligible(int k, int stride, int cin, int cout_stored, int is_f16, bool has_out2);
hipError_t launch_conv3x3_c32(HaloArgs a, hipStream_t stream);
// ... and the split-fp16 form of the 32 -> 32 channel layers (one 128-byte [hi32 | lo32] row per pixel); real channel counts here
bool c32_split_eligible(int k, int stride, int cin, int cout, bool plain_out);
int c32_split_rows_max();
hipError_t launch_conv3x3_c32_split(HaloArgs a, hipStream_t stream); // channel counts / offsets but Cout in pseudo-channels

// ---------------------------------------------------------------------------------------------
// Stem: uint8 frame -> (BGR->RGB, /255) -> 3x3 stride-2 conv (Cin=3) + bias + SiLU -> NHWC.
// Fuses ultralytics' preprocess with model.0 so the fp tensor [B,3,S,S] never exists.
// ---------------------------------------------------------------------------------------------
struct StemArgs {
    const uint8_t *frames; // [N][H][W][C], C = 1 or 3 (BGR)
    int N, H, W, C;
    const void *w; // packed: fp16 [Cout][16 taps][4] (taps 9..15 and channel 3 zero), fp32 [Cout][9][4]; in_split: see below
    const float *bias;
    void *out; // [N][H/2][W/2][Cout]
    int Cout;
    int Ho, Wo;
    int out_split; // fp32 kernel only: store split-fp16 pairs (Cout % 32 == 0)
    int in_split;  // with out_split: split-fp16 OPERANDS too (w = [Cout][16][4] hi halves, then the lo halves): three fp16 matrix instructions per k-step instead of the fp32 ones
    const int *n_dyn; // nullable: device-side image count <= N (wtk_yolo_set_dynamic_batch): blocks of images beyond it exit at once
};
hipError_t launch_stem(const StemArgs &a, int is_f16, hipStream_t stream);

// Fused front (front_fused.hip): preprocess + model.0 + model.1 + model.2.cv1, fp16, widths 32 / 64 / 64.
struct FrontArgs {
    const uint8_t *frames; // [N][H][W][C] network-size frames, C = 1 or 3 (BGR); 4-byte aligned
    int N, H, W, C;
    const void *w0; // stem, packed as StemArgs::w (fp16)
    const float *b0;
    const void *w1; // model.1, [64][Kpad1] fp16, K = tap*32 + channel
    const float *b1;
    int Kpad1;
    const void *w2; // model.2.cv1, [64][Kpad2] fp16
    const float *b2;
    int Kpad2;
    void *out; // [N][H/4][W/4] slice view (out_ld, out_coff), 64 channels
    int out_ld, out_coff;
    unsigned long long *dbg_stamps; // diagnostic builds (-DWTK_FRONT_STAMPS) only: [grid][8 waves][8 stages] cycle sums
    void *dbg_t0, *dbg_t1; // test hook (normally null): also materialise model.0 [N][H/2][W/2][32] / model.1 [N][H/4][W/4][64]
    const int *n_dyn;      // nullable: device-side image count <= N (front_fused_split_kernel: the tiles of images beyond it are not visited)
    int stem_split;        // front_fused_split_kernel: w0 = the split stem packing ([32][16][4] hi halves, then lo) instead of the fp32 one
    int Ho, Wo, tiles_x, tiles_y, total_tiles; // filled by the launcher
    FastDiv d_tpi, d_tilesx;
};
// Fused C2f tail (c2f_fused.hip): m.0.cv1 + m.0.cv2 (+ residual) + cv2 of a C2f block with hidden width 32 and one
// bottleneck (YOLOv8s model.2), fp16.
struct C2fArgs {
    const void *cat; // [N][H][W] concat buffer holding a (a_coff) and b (b_coff), 32 channels each
    int cat_ld, a_coff, b_coff;
    int N, H, W;
    const void *w_m1, *w_m2; // [32][Kpad_m] fp16, K = tap*32 + channel
    const float *b_m1, *b_m2;
    int Kpad_m;
    const void *w_cv2; // [64][Kpad_cv2] fp16, K = concat channel (a, b, m)
    const float *b_cv2;
    int Kpad_cv2;
    void *out; // [N][H][W] slice view, 64 channels
    int out_ld, out_coff;
    const void *zeros;
    unsigned long long *dbg_stamps; // diagnostic builds (-DWTK_C2F_STAMPS) only
    int tiles_x, tiles_y, total_tiles; // filled by the launcher
    FastDiv d_tpi, d_tilesx;
};
bool c2f_fused_eligible(int is_f16, int c_hidden, int n_bottlenecks, int shortcut, int c_out);
hipError_t launch_c2f_fused(C2fArgs a, int num_cus, hipStream_t stream);

bool front_fused_eligible(int is_f16, int c0, int c1, int c2_out);
// split-fp16 ("f16x3") handles: front_fused_split.hip.  w0 = the split stem packing (stem_split) or the fp32 one, w1 / w2 split rows, Kpad1 / Kpad2 / out_ld /
// out_coff in pseudo-channels (2 x real)
bool front_fused_split_eligible(int c0, int c1, int c2_out);
hipError_t launch_front_fused_split(FrontArgs a, int num_cus, hipStream_t stream);
hipError_t launch_front_fused(FrontArgs a, int num_cus, hipStream_t stream);

// Letterbox (ultralytics LetterBox, cv2.INTER_LINEAR fixed-point bilinear + pad 114) of uint8
// frames [N][H][W][C] into [N][S_h][S_w][C].
struct LetterboxArgs {
    const uint8_t *src;
    uint8_t *dst;
    int N, H, W, C;
    int Sh, Sw;       // destination size
    int new_h, new_w; // resized (unpadded) size
    int top, left;    // padding offsets
};
hipError_t launch_letterbox(const LetterboxArgs &a, hipStream_t stream);

struct CropArgs {
    const uint8_t *frames; // [N][H][W][C]
    const int *pos_xy;     // [N][2] platform position (x, y)
    uint8_t *views;        // [N][rows][cols][C]
    int N, H, W, C;
    int view_w, view_h;    // the (w, h) the reference passes to _custom_view
    int rows, cols;        // rows = w, cols = h (reference quirk, identical for square views)
};
hipError_t launch_crop_views(const CropArgs &a, hipStream_t stream);

// Camera view + letterbox in ONE pass: net input pixel <- cv2-style bilinear resize of the camera view
// (rows x cols window of the replicate-padded frame centred on the platform position) + constant-114 border.  The view is
// never materialised: each bilinear tap is fetched from the full frame through the crop's clamp.
struct ViewLetterboxArgs {
    const uint8_t *frames;  // [F][H][W][C] full frames
    const int *frame_index; // [N] frame of batch row n, or nullptr (row n = frame n)
    const int *pos_xy;      // [N][2] platform position (x, y)
    uint8_t *dst;           // [N][Sh][Sw][C] network input
    int N, H, W, C;
    int F;                  // frames in the stack (frame indices are clamped to [0, F))
    int view_w, view_h;     // the (w, h) the reference passes to _custom_view
    int rows, cols;         // view shape: rows = w, cols = h (view_controller.py:171)
    int Sh, Sw;
    int new_h, new_w, top, left; // letterbox geometry of a rows x cols image into Sh x Sw
};
hipError_t launch_view_letterbox(const ViewLetterboxArgs &a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// SPPF pooling: y1 = maxpool5(x), y2 = maxpool5(y1), y3 = maxpool5(y2) (stride 1, pad 2) on a
// channel slice of the SPPF concat buffer; x at channels [0,c), y_k at [k*c,(k+1)*c).
// ---------------------------------------------------------------------------------------------
struct PoolArgs {
    void *buf; // [N][H][W][4c]
    int N, H, W, c;
    FastDiv d_w; // filled by the launcher
    int split;   // fp32 kernel only: buf is a split-fp16 tensor (c % 32 == 0); the maxima are taken on the reconstructed fp32 values
};
hipError_t launch_sppf_pool(const PoolArgs &a, int is_f16, hipStream_t stream);
hipError_t pool_init_attributes();

// ---------------------------------------------------------------------------------------------
// Head: DFL decode + thresholded arg-max selection (max_det = 1) + scale_boxes + xyxy->xywh.
// ---------------------------------------------------------------------------------------------
struct HeadArgs {
    const void *box[3]; // per level [N][h*w][64] DFL logits (storage dtype)
    const void *cls[3]; // per level [N][h*w][cls_ld] class logits (storage dtype)
    int cls_ld;
    int nc;
    int lh[3], lw[3]; // level map sizes
    int N;
    float conf;
    // inverse letterbox (ultralytics scale_boxes): x = (x - pad_x) / gain, clipped to [0, W0]
    float gain, pad_x, pad_y;
    float img_w, img_h;
    float *out_xywh; // [N][4]
    float *out_conf; // [N] or null
    int *out_anchor; // [N] or null
    // decision margin of the frame in logit units, or null: min(best - second-best class logit over the anchors,
    // |best - logit(conf)|) — how far the fp32 result is from choosing another anchor or from flipping detection / NaN row
    float *out_margin;
    float conf_logit; // log(conf / (1 - conf))
    int *status;      // nullable: sticky flags of the handle in pinned HOST memory (bit 0: a non-finite head logit was read; wtk_yolo_status)
};
hipError_t launch_head(const HeadArgs &a, int is_f16, hipStream_t stream);

// General greedy IoU NMS (ultralytics non_max_suppression, class-aware, best class per anchor) for max_det >= 1: HeadArgs plus
// the per-image scratch and the multi-row outputs.  out_* rows beyond count[n] are NaN / 0 / -1.
struct NmsArgs {
    HeadArgs h;        // logits, geometry, conf; h.out_* unused
    float iou;
    int max_det;
    float *scratch_score; // [N][A]   candidate score, -inf once dead / never a candidate
    int *scratch_cls;     // [N][A]   best class of the anchor
    float *scratch_box;   // [N][A][4] net-space xyxy of the candidates
    float *out_xywh;      // [N][max_det][4] image pixels
    float *out_conf;      // [N][max_det]
    int *out_cls;         // [N][max_det] or null
    int *out_anchor;      // [N][max_det] or null
    int *out_count;       // [N] or null
};
hipError_t launch_head_nms(const NmsArgs &a, int is_f16, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// ResMLP (mlp.hip)
// ---------------------------------------------------------------------------------------------
struct MlpLayerDev {
    int in_dim, out_dim;
    int in_pad;  // in_dim rounded up to 4
    int out_pad; // out_dim rounded up to 16
    int relu;
    int w_off; // float offset of W[out_pad][in_pad] in the parameter blob
    int b_off; // float offset of b[out_pad]
    int pad_;
};
constexpr int kMlpMaxLayers = 64;
constexpr int kMlpMaxDim = 128; // widest activation the kernel supports
constexpr int kMlpMaxInputFrames = 16;
struct MlpArgs {
    const float *params; // 16-byte aligned blob, padded to a multiple of 4 floats
    int n_params;        // floats in the blob
    const MlpLayerDev *layers; // device array
    int n_layers, n_blocks, layers_per_block;
    int in_dim, out_dim;
    // direct mode
    const float *x; // [B][in_dim]
    // gather mode (x == nullptr)
    const float *track; // [n_frames][4]
    int n_frames;
    const int *anchor_frames; // [B]
    int input_frames[kMlpMaxInputFrames];
    int n_in;
    int *valid; // [B]
    float *y;   // [B][out_dim]
    int B;
};
hipError_t launch_mlp(const MlpArgs &a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Batched per-cycle predictors over a device track (track_ops.hip)
// ---------------------------------------------------------------------------------------------
constexpr int kTrackMaxWindow = 64; // imaging frames per cycle the median kernel sorts in registers
constexpr int kTrackMaxTimes = 16;  // sample times of a polynomial fit / input or target frames of a training pair
constexpr int kTrackMaxCoef = 8;    // polynomial degree + 1
struct TrackMedianArgs {
    const void *track; // [n_frames][4] xywh, float or double
    int n_frames;
    const int *cycles; // [n_samples] cycle numbers (device)
    int n_samples;
    int cycle_frame_num, imaging_frame_num;
    double *pred; // [n_samples][2] median centre (x, y), absolute coordinates
    int *valid;   // [n_samples] 0: no finite row in the window
};
hipError_t launch_track_median(const TrackMedianArgs &a, int track_f64, hipStream_t stream);
struct TrackPolyfitArgs {
    const void *track;
    int n_frames;
    const int *cycles;
    int n_samples;
    int cycle_frame_num;
    int times[kTrackMaxTimes];      // frame offsets from the start of the cycle (sorted, as PolyfitConfig leaves them)
    double weights[kTrackMaxTimes]; // weight of the i-th time
    int n_times, degree;
    double t_eval; // cycle_frame_num + imaging_frame_num // 2
    double *pred;  // [n_samples][2] extrapolated centre (x, y), absolute coordinates
    int *valid;    // [n_samples] 0: no usable sample
};
hipError_t launch_track_polyfit(const TrackPolyfitArgs &a, int track_f64, hipStream_t stream);
struct TrackPairsArgs {
    const void *track;
    int n_frames;
    int row0, n_rows; // candidate rows row0 .. row0 + n_rows - 1
    int in_frames[kTrackMaxTimes], out_frames[kTrackMaxTimes];
    int n_in, n_out;
    float *X; // [n_rows][4 * n_in]
    float *Y; // [n_rows][2 * n_out]
    int *keep; // [n_rows] 1: no NaN in the row
};
hipError_t launch_track_pairs(const TrackPairsArgs &a, int track_f64, hipStream_t stream);

// Second look at the frames with the weakest decisions (wtk_recheck_select / wtk_recheck_merge): the K smallest decision margins of a batch,
// and the merge of the K re-detected rows into the batch's rows
struct RecheckArgs {
    const float *margins; // [B] decision margins of the fast pass (wtk_yolo_margin_buffer)
    int B, K;
    int *slots;           // [K] batch rows of the K smallest margins (ties: lower row first), ascending
    float thr;            // merge: a re-detected row replaces the fast one where margins[row] < thr
    const float *src_xywh, *src_conf;
    const int *src_anchor;
    float *dst_xywh, *dst_conf;
    int *dst_anchor;
    int *n_replaced;      // nullable: += number of rows replaced
    int *n_weak;          // select, nullable: = min(K, number of rows with margin < thr) — the leading slots (the input of wtk_yolo_set_dynamic_batch)
    int *n_overflow;      // select, nullable: += max(rows with margin < thr - K, 0): weak rows that get NO second look because the ceiling K cut them off
};
hipError_t launch_recheck_select(const RecheckArgs &a, hipStream_t stream);
hipError_t launch_recheck_merge(const RecheckArgs &a, hipStream_t stream);

// Deferred second look (wtk_recheck_enqueue / wtk_recheck_scatter): the weak rows of SEVERAL fast passes are collected in a device-side
// queue — a copy of the frame plus the addresses of the row's three outputs — and looked at again in ONE full-precision pass, so the fixed
// cost of that pass (62 launches of a few tiles each: ~1.2 ms however few frames are live) is paid once per D batches instead of once per batch.
struct RecheckQueueArgs {
    const float *margins;       // [B] decision margins of the fast pass
    int B;
    float thr;                  // rows with margin < thr are queued
    const unsigned char *frames; // [B][frame_bytes] the fast pass's input batch
    long long frame_bytes;      // multiple of 16
    unsigned char *q_frames;    // [q_cap][frame_bytes]
    int q_cap;
    int *q_len;                 // rows queued so far (device counter)
    float **q_xywh, **q_conf;   // [q_cap] where the re-detected row goes (q_conf / q_anchor entries may be null)
    int **q_anchor;
    float *dst_xywh, *dst_conf; // this batch's output rows: row b at dst_xywh + 4 b, dst_conf + b, dst_anchor + b
    int *dst_anchor;
    int *pos;                   // [B] scratch: queue position of batch row b, or -1
    int *n_overflow;            // nullable: += weak rows that found the queue full (they keep their fast result)
    // scatter
    const float *src_xywh, *src_conf; // [q_cap] rows of the full-precision pass over q_frames
    const int *src_anchor;
    int *n_replaced;            // nullable: += rows written back
};
hipError_t launch_recheck_enqueue(const RecheckQueueArgs &a, hipStream_t stream);
hipError_t launch_recheck_scatter(const RecheckQueueArgs &a, hipStream_t stream);
// slots[i] (a batch row of the fast pass) -> frame index and view centre of row i of the second look (wtk_hybrid_predict_views)
hipError_t launch_recheck_gather_views(const int *slots, int K, const int *frame_index, const int *pos_xy, int *idx_out, int *pos_out, hipStream_t stream);

} // namespace wtk
