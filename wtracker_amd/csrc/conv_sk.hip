// Split-K implicit-GEMM convolution for the SMALL-BATCH (latency) plan of a detector handle (gfx950).
//
// Why it exists.  The reference calls the detector twice per cycle: one batch of cycle_frame_num frames (9 / 15) and ONE frame
// (yolo_controller.py:96-98, 108-109), at imgsz 384.  At that size the throughput kernels (conv_igemm / conv3x3_halo / conv3x3_s2)
// run every layer on a handful of blocks, each of which walks the whole K of its tile one staged step at a time: a step costs a full
// memory latency when nothing else is resident on the CU, so a layer takes 10-45 us whatever its arithmetic (profiles/r05_notes.md:
// B = 1 at 384^2, B = 15 at 384^2 and B = 1 at 640^2 take the same time per layer).  Here a layer is cut along K as well:
//   block = (pixel tile, cout tile, K atom — or all atoms, see below);  a step = one 128-byte operand row = 32 channels of one tap
// and a block requests its slice's operands through an NS-deep LDS ring with NS - 1 stages in flight from the first instruction
// (for a slice of <= NS - 1 steps: everything at once, ONE latency), multiplies, and leaves either the finished tile (S = 1) or an
// fp32 partial tile in its slice's slab.  The slabs of a tile are combined by the block of that tile that ARRIVES LAST (one ticket per tile: write-through
// `sc1` slab stores, every wave's vmcnt(0), block barrier, one agent-scope atomic add by one lane; the block whose add returns S - 1 reads all S slabs
// with `sc1` loads — MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads", first row — adds them in slice order and applies bias / SiLU / residual /
// store; it also re-arms the ticket).  sk_finish_kernel is the same combination as a second launch (WTK_SK_FINISH=1): a dependent launch costs
// ~4.7 us here, which is what a small layer's whole convolution costs.  Both forms add the slabs in the same order: bit-identical.
//
// Where the hand-off has been checked beyond that row's "one workgroup per CU" (ADVICE r05): a block of this kernel owns its CU against other blocks of
// this kernel (96-128 KB of LDS), not against blocks of OTHER kernels — the deferred track log runs a cycle batch on a second lane beside the single-frame
// call.  tests/test_gpu_latency.py::test_slab_hand_off_holds_beside_another_handles_kernels compares every conv tensor of 24 rounds, word for word, with
// the two-launch form while a second handle streams uneven batches on another stream: equal.  The loads are sc1 (served by L2, never by this CU's L1), the
// stores write through, every storing wave drains before the one ticket add — none of which depends on who else is resident; an agent acquire in the
// combining block (buffer_inv sc1 + vmcnt(0): ~1.5 us on each of the ~33 split layers of a forward, +10 %) would buy nothing that test can see.
//
// Grouped launches (round 6).  A launch carries up to kSkGroupMax convs that do not depend on each other — one dependency LEVEL of the latency plan
// (csrc/wtk_plan.hip: sk_schedule) — as one grid on the caller's stream: a block finds its conv by its index (SkGroupArgs::first).  One tile per launch,
// a form per member, both chosen by the cost model; neither enters the arithmetic.
//
// Determinism and batch invariance.  K is cut into ATOMS — fixed per layer and HANDLE, never per call (conv_sk_slices: <= 12 steps: one atom; else atoms
// of ~8 steps, at most 8; conv_sk_plan_atoms: the count the cost model below likes best for the handle's typical call) — and an output value is DEFINED as ((A_0 + A_1) + A_2) + ..., A_i = the MFMA chain over atom i's steps started from zero
// (split mode: acc + 2^-11 acc1 of that chain).  Two launch forms produce exactly that value:
//   S = NA  one block per (tile, atom): A_i goes to slab i, the combining block adds the slabs in atom order;
//   S = 1   one block walks all atoms through one continuous ring and folds A_i into its running total at every atom boundary
// so the form (like the tile shape) is chosen per LAUNCH from the batch — many pixels: S = 1, no slabs at all; few pixels: S = NA, every CU gets
// work — and a frame still gets the same logits at B = 1 and at B = 15, bit for bit.  No atomics on data.
//
// Operand layout: the igemm kernel's (conv_igemm.hip): NHWC activations as channel-slice views, weights [CoutPad][K] with K = (tap, channel);
// a step's two operand tiles are staged by LDS-DMA (buffer_load ... lds from inline asm, source-side XOR swizzle) and read back as
// conflict-free ds_read_b128 MFMA fragments.  Two storage modes share all of it because both spend 4 bytes per value:
//   SPLIT  split-fp16 pairs [hi32 | lo32] per 32 channels, three v_mfma_f32_16x16x32_f16 per product (wtk_kernels.h, kSplitScale)
//   fp32   32 floats per row, exact v_mfma_f32_16x16x4_f32
#include "wtk_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace wtk {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t sk_rsrc(const void *base) {
    const unsigned long long b = (unsigned long long)base;
    rsrc_t r;
    r.x = (int)(unsigned)(b & 0xffffffffu);
    r.y = (int)(unsigned)((b >> 32) & 0xffffu);
    r.z = (int)0xffffff00u; // num_records: a lane offset of 0xffffffff is out of range and lands zeros
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void sk_dma(const rsrc_t &rs, unsigned voff, unsigned soff, char *lds_dst) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}
// wait until all but the youngest n * PER vector-memory operations of this wave are done (n stages of PER requests each still in flight)
template <int PER> __device__ __forceinline__ void sk_wait_stages(int n) {
    static_assert(PER * 6 <= 63, "vmcnt is a 6-bit counter");
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PER) : "memory");
}

// 16-byte write-through store / L1-bypassing load (agent scope): the slab hand-off between the blocks of a tile
__device__ __forceinline__ void sk_store_sc1(float *p, const floatx4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void sk_load_sc1(floatx4 &v, const float *p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
// the loads above are invisible to hipcc's waitcnt pass: the wait names the registers so that no use can be scheduled in front of it
__device__ __forceinline__ void sk_wait_loads(floatx4 (&x)[16]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]),
                   "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])::"memory");
}

// what becomes of a finished run of NV consecutive couts (cb ..) of output pixel `pix`; v = the K sum WITHOUT the bias
template <int NV> __device__ __forceinline__ void sk_load_bias(const SkArgs &a, int cb, float (&b)[NV]) { // rows exist up to CoutPad
#pragma unroll
    for (int e = 0; e < NV; e += 4) {
        const float4 f = *reinterpret_cast<const float4 *>(a.bias + cb + e);
        b[e] = f.x, b[e + 1] = f.y, b[e + 2] = f.z, b[e + 3] = f.w;
    }
}
template <bool SPLIT, int NV> __device__ __forceinline__ void sk_store(const SkArgs &a, long long pix, int cb, float (&v)[NV], const float (&bias)[NV]) {
    if (cb + NV > a.Cout) return; // padded output channels are never stored (Cout is a multiple of 8, NV of 8)
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] += bias[e];
    if (a.act) wtk_silu_scaled_run<NV>(v);
    if (a.res) {
        float rv[NV];
        if constexpr (SPLIT) {
            wtk_split_load<NV>(reinterpret_cast<const _Float16 *>(a.res) + pix * a.res_ld + a.res_coff, cb, rv);
        } else {
            const float *rp = reinterpret_cast<const float *>(a.res) + pix * a.res_ld + a.res_coff + cb;
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const float4 f = *reinterpret_cast<const float4 *>(rp + e);
                rv[e] = f.x, rv[e + 1] = f.y, rv[e + 2] = f.z, rv[e + 3] = f.w;
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] += rv[e];
    }
    if (!SPLIT || a.out_f32) {
        float *op = reinterpret_cast<float *>(a.out) + pix * a.out_ld + a.out_coff + cb;
#pragma unroll
        for (int e = 0; e < NV; e += 4) *reinterpret_cast<float4 *>(op + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
    } else {
        wtk_split_store<NV>(reinterpret_cast<_Float16 *>(a.out) + pix * a.out_ld + a.out_coff, cb, v);
    }
}

// One block's work on one conv: `bid` = the block's index among THIS conv's blocks (a grouped launch carries several convs, see conv_sk_kernel below)
template <bool SPLIT, int BM, int BN, int WAVES_P, int WAVES_C, int NS>
__device__ __forceinline__ void sk_body(const SkArgs &a, const unsigned bid, char *smem) {
    constexpr int NW = WAVES_P * WAVES_C;
    constexpr int RPP = 8 * NW; // tile rows staged per pass of the whole block (one wave instruction = 8 rows x 128 B)
    constexpr int PR = BM / RPP, WR = BN / RPP;
    constexpr int WP = BM / WAVES_P, WC = BN / WAVES_C;
    constexpr int TP = WP / 16, TC = WC / 16;
    constexpr int NV = 4 * TC; // consecutive couts owned by a lane
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PER = PR + WR; // LDS-DMA requests per thread and stage
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(BM % RPP == 0 && BN % RPP == 0 && TP >= 1 && TC >= 2 && NV % 8 == 0, "tile shape");
    static_assert((NS & (NS - 1)) == 0 && NS >= 2 && NS <= 8 && NS * STAGE <= 160 * 1024, "ring");
    // the slab hand-off below is the form MI355X_MICROARCH.md measured for ONE workgroup per CU: more than half of the LDS per block guarantees it
    static_assert(NS * STAGE > 80 * 1024, "one block per CU");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_p = wave / WAVES_C, wave_c = wave % WAVES_C;
    const int lr = lane & 15, lg = lane >> 4;

    // block -> (pixel tile fastest, cout tile, K slice): the blocks that are neighbours in launch order share a weight slice
    const unsigned bq = fdiv(bid, a.d_ptiles);
    const int ptile = (int)(bid - bq * (unsigned)a.ptiles);
    const int slice = (int)fdiv(bq, a.d_nct);
    const int ctile = (int)(bq - (unsigned)slice * (unsigned)a.nct);
    // this block's atoms: all of them (S = 1) or atom `slice` (S = NA); atom i = steps [i nk / NA, (i + 1) nk / NA)
    int atom = a.S == 1 ? 0 : slice;
    // (nk <= a few hundred, NA <= 8: 32-bit products; the 64-bit quotient hipcc would otherwise expand costs more than a K step)
    const int ks0 = (int)((unsigned)(atom * a.nk) / (unsigned)a.NA), ks1 = a.S == 1 ? a.nk : (int)((unsigned)((atom + 1) * a.nk) / (unsigned)a.NA);
    const int nkb = ks1 - ks0;
    int atom_end = (int)((unsigned)((atom + 1) * a.nk) / (unsigned)a.NA) - ks0; // step index (from ks0) at which the current atom is complete
    const int HoWo = a.Ho * a.Wo;
    long long m_eff = a.M;
    if (a.n_dyn) m_eff = (long long)min(max(*a.n_dyn, 0), a.N) * HoWo; // dynamic batch: tiles that start beyond its last image are not computed
    const long long m0 = (long long)ptile * BM;
    if (m0 >= m_eff) return;
    const int n0 = ctile * BN;

    // ---- staging assignment (as conv_igemm.hip): thread -> 16-byte physical chunk `ch` of rows r0 + RPP * i; the swizzle is applied to the SOURCE chunk
    const int ch = tid & 7, r0 = tid >> 3;
    const int lchunk = ch ^ (r0 & 7);
    const int nb = (int)fdiv((unsigned)m0, a.d_howo); // image of the tile's first pixel: lane offsets are relative to it (32-bit)
    const rsrc_t in_rs = sk_rsrc(a.in + (long long)nb * a.H * a.W * a.in_ldb + a.in_offb);
    const rsrc_t in2_rs = sk_rsrc(a.in2 ? a.in2 + (long long)nb * (a.H >> 1) * (a.W >> 1) * a.in2_ldb + a.in2_offb : a.in);
    const rsrc_t w_rs = sk_rsrc(a.w + (long long)n0 * a.w_rowb);
    unsigned poff[PR], poff2[PR];
    int hi0[PR], wi0[PR];
#pragma unroll
    for (int i = 0; i < PR; ++i) {
        const long long m = m0 + r0 + RPP * i;
        if (m < a.M) {
            const unsigned mu = (unsigned)m;
            const int n = (int)fdiv(mu, a.d_howo);
            const unsigned rem = mu - (unsigned)n * (unsigned)HoWo;
            const int ho = (int)fdiv(rem, a.d_wo), wo = (int)(rem - (unsigned)ho * (unsigned)a.Wo);
            hi0[i] = ho * a.stride - a.pad;
            wi0[i] = wo * a.stride - a.pad;
            poff[i] = (unsigned)((((n - nb) * a.H + hi0[i]) * a.W + wi0[i]) * (int)a.in_ldb + lchunk * 16); // may wrap below 0: only used with an in-range tap
            poff2[i] = (unsigned)((((n - nb) * (a.H >> 1) + (ho >> 1)) * (a.W >> 1) + (wo >> 1)) * (int)a.in2_ldb + lchunk * 16);
        } else {
            hi0[i] = wi0[i] = -(1 << 28); // fails every bounds test
            poff[i] = 0;
            poff2[i] = 0xffffffffu;
        }
    }
    unsigned wvoff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = r0 + RPP * i;
        const int key = ((row >> 1) & 1) | (((row / NV) & 3) << 1);
        wvoff[i] = (unsigned)row * a.w_rowb + (unsigned)((ch ^ key) << 4);
    }
    auto issue = [&](int ks, char *buf) __attribute__((always_inline)) {
        const int tap = (int)fdiv((unsigned)ks, a.d_cpb), cb = ks - tap * a.cpb; // wave-uniform
        char *pt = buf + 8 * wave * 128, *wt = buf + (BM + 8 * wave) * 128;
        if (cb < a.in2_blocks) { // 1x1 over [up2x(low) | high]: these channels live in the half-resolution tensor
#pragma unroll
            for (int i = 0; i < PR; ++i) sk_dma(in2_rs, poff2[i], (unsigned)cb * 128u, pt + RPP * i * 128);
        } else {
            const int kh = a.KW == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * a.KW; // tap / 3 for tap < 32; a 1x1 has the one tap
            const unsigned delta = (unsigned)((kh * a.W + kw) * (int)a.in_ldb + cb * 128);
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                const bool ok = (unsigned)(hi0[i] + kh) < (unsigned)a.H && (unsigned)(wi0[i] + kw) < (unsigned)a.W;
                sk_dma(in_rs, ok ? poff[i] + delta : 0xffffffffu, 0u, pt + RPP * i * 128);
            }
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) sk_dma(w_rs, wvoff[i], (unsigned)ks * 128u, wt + RPP * i * 128);
    };

    // the bias of this lane's couts: requested before the first operand so that its round trip is over long before the epilogue (a small layer is
    // a chain of two or three memory latencies: every one that can run beside another counts)
    float bias_r[NV];
    sk_load_bias<NV>(a, n0 + wave_c * WC + lg * NV, bias_r);
    floatx4 acc[TC][TP];
    floatx4 acc1[SPLIT ? TC : 1][SPLIT ? TP : 1];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
    // fragment addresses inside a stage: pixel tiles j at +j*2048 (same swizzle key), cout tiles i at +i*512 (key independent of i); second k-half = ^64
    const int prow_l = wave_p * WP + lr;
    const unsigned pfrag0 = prow_l * 128 + ((lg ^ (prow_l & 7)) << 4);
    const int wrow_l = wave_c * WC + (lr >> 2) * NV + (lr & 3);
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l / NV) & 3) << 1);
    const unsigned wfrag0 = BM * 128 + wrow_l * 128 + ((lg ^ wkey_l) << 4);
    auto compute = [&](const char *pt) __attribute__((always_inline)) {
        if constexpr (SPLIT) { // hi fragments = k-half 0, lo fragments = k-half 1 of the same 32 channels
            uint4 ph[TP], wh[TC], wl[TC], pl[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) ph[j] = *reinterpret_cast<const uint4 *>(pt + pfrag0 + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wh[i] = *reinterpret_cast<const uint4 *>(pt + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) wl[i] = *reinterpret_cast<const uint4 *>(pt + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int j = 0; j < TP; ++j) pl[j] = *reinterpret_cast<const uint4 *>(pt + (pfrag0 ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const half8 xh = __builtin_bit_cast(half8, ph[j]), xl = __builtin_bit_cast(half8, pl[j]);
                    const half8 yh = __builtin_bit_cast(half8, wh[i]), yl = __builtin_bit_cast(half8, wl[i]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xh, acc[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, xh, acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xl, acc1[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2) {
                const unsigned pa = kh2 ? (pfrag0 ^ 64u) : pfrag0, wa = kh2 ? (wfrag0 ^ 64u) : wfrag0;
                uint4 pf[TP], wf[TC];
#pragma unroll
                for (int j = 0; j < TP; ++j) pf[j] = *reinterpret_cast<const uint4 *>(pt + pa + j * 2048);
#pragma unroll
                for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const uint4 *>(pt + wa + i * 512);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) { // the same k on both operands: element e of every lane group's chunk
                        floatx4 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf[i].x), __builtin_bit_cast(float, pf[j].x), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf[i].y), __builtin_bit_cast(float, pf[j].y), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf[i].z), __builtin_bit_cast(float, pf[j].z), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, wf[i].w), __builtin_bit_cast(float, pf[j].w), c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        }
    };

    // ---- the ring: stages ks0 .. ks0 + NS - 2 are requested before anything else; step s waits for ITS stage only (counted vmcnt), meets
    // the other waves (which also says that everybody is done reading the stage of step s - 1) and then re-fills that stage
    const int pre = min(NS - 1, nkb);
    for (int s = 0; s < pre; ++s) issue(ks0 + s, smem + s * STAGE);
    floatx4 tot[TC][TP]; // ((A_0 + A_1) + ...) of the atoms finished so far
    // (two nested loops — atoms outside, the atom's steps inside — rather than a fold under an `if` in one flat loop: with the flat form hipcc kept the
    // chains in AGPRs and copied all of them out and back at EVERY step; the ring itself runs on across the atom boundaries)
    int s = 0;
    for (bool first_atom = true; s < nkb; first_atom = false) {
        for (; s < atom_end; ++s) {
            sk_wait_stages<PER>(min(s + NS - 1, nkb) - s - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragment reads of step s - 1 are done
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + NS - 1 < nkb) issue(ks0 + s + NS - 1, smem + ((s + NS - 1) & (NS - 1)) * STAGE);
            compute(smem + (s & (NS - 1)) * STAGE);
        }
        // atom complete: its value joins the total, the chains start again from zero
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                floatx4 av;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (SPLIT)
                        av[r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
                    else
                        av[r] = acc[i][j][r];
                }
                tot[i][j] = first_atom ? av : tot[i][j] + av;
                acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT) acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
        ++atom;
        atom_end = min((int)((unsigned)((atom + 1) * a.nk) / (unsigned)a.NA) - ks0, nkb);
    }

    // ---- the tile: lane (pixel lr of tile j, group lg) owns couts cb .. cb + NV - 1
    const int cb = n0 + wave_c * WC + lg * NV;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const long long pix = m0 + wave_p * WP + j * 16 + lr;
        if (pix >= a.M) continue;
        float v[NV];
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t * 4 + r] = tot[t][j][r];
        if (a.S == 1) {
            sk_store<SPLIT, NV>(a, pix, cb, v, bias_r);
        } else {
            float *pp = a.partial + ((long long)slice * a.M + pix) * a.CoutPad + cb;
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const floatx4 f = {v[e], v[e + 1], v[e + 2], v[e + 3]};
                if (a.tickets)
                    sk_store_sc1(pp + e, f);
                else
                    *reinterpret_cast<floatx4 *>(pp + e) = f;
            }
        }
    }
    if (a.S == 1 || !a.tickets) return;
    // ---- the block of this tile that arrives last combines the slabs (see the header).  Every wave's slab stores are complete (vmcnt(0)) before the
    // barrier; one lane then takes the ticket; its value reaches the other waves through LDS (the ring is free: every wave is past its last read).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *flag = reinterpret_cast<int *>(smem);
    if (tid == 0) {
        unsigned *tk = a.tickets + (ptile + a.ptiles * ctile);
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(a.S - 1);
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // nobody else touches this ticket before the next launch
        *flag = last;
    }
    __syncthreads();
    if (!*reinterpret_cast<volatile int *>(flag)) return;
    const long long slab = a.M * a.CoutPad;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const long long pix = m0 + wave_p * WP + j * 16 + lr;
        if (pix >= a.M) continue;
        if (cb + NV > a.Cout) continue;
        const float *pp = a.partial + pix * a.CoutPad + cb;
        float v[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = 0.f;
        static_assert(NV == 8, "two float4 per slab and lane");
        for (int s0 = 0; s0 < a.S; s0 += 8) { // eight slabs per round trip (conv_sk_slices never makes more): slabs beyond S - 1 re-read the last one and are not added
            floatx4 x[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float *ps = pp + (long long)min(s0 + q, a.S - 1) * slab;
                sk_load_sc1(x[2 * q], ps);
                sk_load_sc1(x[2 * q + 1], ps + 4);
            }
            sk_wait_loads(x);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < a.S) { // slice order: v = ((p0 + p1) + p2) + ...
                    if (s0 + q == 0) {
                        v[0] = x[0].x, v[1] = x[0].y, v[2] = x[0].z, v[3] = x[0].w, v[4] = x[1].x, v[5] = x[1].y, v[6] = x[1].z, v[7] = x[1].w;
                    } else {
                        v[0] += x[2 * q].x, v[1] += x[2 * q].y, v[2] += x[2 * q].z, v[3] += x[2 * q].w;
                        v[4] += x[2 * q + 1].x, v[5] += x[2 * q + 1].y, v[6] += x[2 * q + 1].z, v[7] += x[2 * q + 1].w;
                    }
                }
        }
        sk_store<SPLIT, NV>(a, pix, cb, v, bias_r);
    }
}

// A launch carries up to kSkGroupMax convs that do not depend on each other (one dependency LEVEL of the latency plan: a Detect tower's box and class
// convs, a PAN layer next to the tower of the feature map before it — csrc/wtk_plan.hip: sk_schedule): the grid is the concatenation of the members'
// grids, a block finds its member by its index.  One stream, one dispatch per level — what round 5 spread over three streams and 61 dispatches.
template <bool SPLIT, int BM, int BN, int WAVES_P, int WAVES_C, int NS>
__global__ __launch_bounds__(64 * WAVES_P *WAVES_C, 1) void conv_sk_kernel(const SkGroupArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (BM + BN) * 128];
    int gi = 0; // block-uniform
#pragma unroll
    for (int i = 1; i < kSkGroupMax; ++i) gi += blockIdx.x >= g.first[i] ? 1 : 0;
    sk_body<SPLIT, BM, BN, WAVES_P, WAVES_C, NS>(g.p[gi], blockIdx.x - g.first[gi], smem);
}

// slabs of a split layer -> the layer's output: one thread per (pixel, run of 8 couts), slabs added in slice order
template <bool SPLIT> __global__ __launch_bounds__(256) void sk_finish_kernel(const SkArgs a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = a.Cout >> 3;
    long long m_eff = a.M;
    if (a.n_dyn) m_eff = (long long)min(max(*a.n_dyn, 0), a.N) * a.Ho * a.Wo;
    if (t >= m_eff * cg) return;
    const unsigned pix = fdiv((unsigned)t, a.d_cg);
    const int cb = (int)((unsigned)t - pix * (unsigned)cg) * 8;
    float v[8];
    const float *pp = a.partial + (long long)pix * a.CoutPad + cb;
    const long long slab = a.M * a.CoutPad;
    {
        const float4 x = *reinterpret_cast<const float4 *>(pp), y = *reinterpret_cast<const float4 *>(pp + 4);
        v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w, v[4] = y.x, v[5] = y.y, v[6] = y.z, v[7] = y.w;
    }
    for (int s = 1; s < a.S; ++s) {
        const float4 x = *reinterpret_cast<const float4 *>(pp + s * slab), y = *reinterpret_cast<const float4 *>(pp + s * slab + 4);
        v[0] += x.x, v[1] += x.y, v[2] += x.z, v[3] += x.w, v[4] += y.x, v[5] += y.y, v[6] += y.z, v[7] += y.w;
    }
    float bias[8];
    sk_load_bias<8>(a, cb, bias);
    sk_store<SPLIT, 8>(a, (long long)pix, cb, v, bias);
}

template <bool SPLIT, int BM, int BN, int WAVES_P, int WAVES_C, int NS> hipError_t sk_launch_t(SkGroupArgs &g, int n, hipStream_t st) {
    long long grid = 0;
    for (int i = 0; i < kSkGroupMax; ++i) {
        g.first[i] = 0xffffffffu;
        if (i >= n) continue;
        SkArgs &a = g.p[i];
        a.ptiles = (int)((a.M + BM - 1) / BM);
        a.nct = a.CoutPad / BN;
        a.d_ptiles = make_fastdiv((unsigned)a.ptiles);
        a.d_nct = make_fastdiv((unsigned)a.nct);
        g.first[i] = (unsigned)grid;
        grid += (long long)a.ptiles * a.nct * a.S;
        if (a.CoutPad % BN || grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL((conv_sk_kernel<SPLIT, BM, BN, WAVES_P, WAVES_C, NS>), dim3((unsigned)grid), dim3(64 * WAVES_P * WAVES_C), 0, st, g);
    return hipGetLastError();
}

} // namespace

// Default K atoms of a layer with nk steps of 32 channels: a function of the layer alone (see the header: batch invariance; conv_sk_plan_atoms refines it per handle)
// nk <= sk_single_max(): one block walks the whole K through the ring (no slabs, no hand-off); above it slices of ~sk_slice_steps() steps, at most 8
// slices (the combining block reads all slabs in ONE round trip).  WTK_SK_SINGLE_MAX / WTK_SK_SLICE_STEPS: tuning switches, read once per process.
static int sk_env(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return e && e[0] ? std::atoi(e) : dflt;
}
static int sk_single_max() {
    static const int v = sk_env("WTK_SK_SINGLE_MAX", 12);
    return v;
}
static int sk_slice_steps() {
    static const int v = sk_env("WTK_SK_SLICE_STEPS", 8);
    return v > 0 ? v : 8;
}
int conv_sk_slices(int nk) {
    if (nk <= sk_single_max()) return 1;
    const int s = (nk + sk_slice_steps() - 1) / sk_slice_steps();
    return s < 2 ? 2 : (s > 8 ? 8 : s);
}

// Form and tile of one launch.  A small cost model in microseconds, calibrated on profiles/r05_notes.md's forced-tile timelines: blocks run one per
// CU (96-128 KB of LDS) in rounds; a block costs a fixed latency chain, its operand bytes at what a CU's memory path delivers, and — split
// form — the hand-off (write-through slabs, ticket, slab reads) or the second launch.
static long long sk_inkernel_max() {
    static const long long v = (long long)sk_env("WTK_SK_INKERNEL_MAX_KB", 4096) * 1024;
    return v;
}
struct SkShape { // what the cost model needs to know of one conv
    long long M;
    int cout_pad, nk, NA;
};
struct SkTile {
    int bm, bn;
};
static const SkTile kSkTiles[4] = {{128, 128}, {128, 64}, {64, 64}, {64, 32}};
// (constants from the forced-tile / forced-form runs of profiles/r05_notes.md: a CU gets ~40 KB/us of operands through its LDS-DMA requests whether the
// chip is full or not, and with one block per CU nothing overlaps the multiply that follows — 16 cycles per v_mfma_f32_16x16x32_f16, three per tile pair in
// split mode, eight 32-cycle v_mfma_f32_16x16x4_f32 in fp32 mode, four SIMDs, ~2.2 GHz —; the in-kernel hand-off costs ~4 us + the combining block's slab
// reads at ~60 KB/us; a second launch ~3 us + the slabs at ~5 MB/us)
static void sk_member_cost(const SkShape &c, const SkTile &t, int S, int split, double *blocks, double *block_us, double *handoff_us) {
    const double tiles = (double)((c.M + t.bm - 1) / t.bm) * (c.cout_pad / t.bn);
    *blocks = tiles * S;
    const double steps = std::ceil((double)c.nk / S);
    const double stage_kb = (t.bm + t.bn) * 128.0 / 1024.0;
    const double mfma_us = (t.bm / 16) * (t.bn / 16) * (split ? 3.0 * 16.0 : 8.0 * 32.0) / 4.0 / 2200.0;
    *block_us = 1.9 + steps * (stage_kb / 40.0 + mfma_us);
    *handoff_us = 0.0;
    if (S > 1) {
        const double slab_bytes = (double)S * c.M * c.cout_pad * 4.0;
        const double tile_kb = t.bm * t.bn * 4.0 / 1024.0;
        *handoff_us = slab_bytes <= (double)sk_inkernel_max() ? 4.0 + S * tile_kb / 60.0 : 3.0 + slab_bytes / 5.0e6;
    }
}
// list scheduling of a grouped launch: member i contributes b[i] blocks of u[i] microseconds each, dispatched in member order, one block per CU at a time
static double sk_makespan(const double *b, const double *u, int n, int cus) {
    std::vector<double> heap((size_t)std::max(cus, 1), 0.0); // min-heap of the CUs' free times
    auto cmp = [](double x, double y) { return x > y; };
    double end = 0.0;
    for (int i = 0; i < n; ++i)
        for (long long k = 0; k < (long long)b[i]; ++k) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const double t = heap.back() + u[i];
            heap.back() = t;
            std::push_heap(heap.begin(), heap.end(), cmp);
            end = std::max(end, t);
        }
    return end;
}

// Form (per member) and tile (one per launch: the instantiation) of a launch of n independent convs: blocks run one per CU (96-128 KB of LDS) in rounds
// over the concatenated grid; a round lasts as long as its slowest block, the hand-offs of the members run side by side.  For n = 1 this is round 5's
// model of a single launch.  force_tile / force_form: the test hooks (0..3 / 0: always S = NA, 1: always S = 1; -1: free).
static double sk_choose(const SkShape *c, int n, int num_cus, int split, int force_tile, int force_form, int *tile, int *S_out) {
    double best_t = 1e30;
    *tile = 3;
    for (int i = 0; i < n; ++i) S_out[i] = c[i].NA;
    for (int ti = 0; ti < 4; ++ti) {
        const SkTile &t = kSkTiles[ti];
        bool fits = true, forced_fits = force_tile >= 0 && force_tile <= 3;
        for (int i = 0; i < n; ++i) {
            fits = fits && c[i].cout_pad % t.bn == 0;
            if (forced_fits) forced_fits = c[i].cout_pad % kSkTiles[force_tile].bn == 0;
        }
        if (!fits) continue;
        if (forced_fits && ti != force_tile) continue;
        for (int combo = 0; combo < (1 << n); ++combo) { // bit i: member i walks all its atoms in one block (S = 1)
            double blocks = 0, slowest = 0, handoff = 0;
            bool ok = true;
            int S[kSkGroupMax];
            for (int i = 0; i < n && ok; ++i) {
                S[i] = (combo >> i) & 1 ? 1 : c[i].NA;
                if (c[i].NA == 1 && ((combo >> i) & 1) == 0) ok = false; // (one atom: the two forms are the same launch — count it once)
                if (force_form == 0 && S[i] != c[i].NA) ok = false;
                if (force_form == 1 && S[i] != 1) ok = false;
                double b, bu, hu;
                sk_member_cost(c[i], t, S[i], split, &b, &bu, &hu);
                blocks += b, slowest = std::max(slowest, bu), handoff = std::max(handoff, hu);
            }
            if (!ok) continue;
            double tt;
            if (n == 1) {
                tt = std::ceil(blocks / (double)num_cus) * slowest + handoff; // identical blocks: whole rounds
            } else {
                // members of different block lengths share the CUs: blocks are handed out in launch order to whichever CU frees up first
                double b_i[kSkGroupMax], u_i[kSkGroupMax], hu;
                for (int i = 0; i < n; ++i) sk_member_cost(c[i], t, S[i], split, &b_i[i], &u_i[i], &hu);
                tt = sk_makespan(b_i, u_i, n, num_cus) + handoff;
            }
            if (tt < best_t) {
                best_t = tt, *tile = ti;
                for (int i = 0; i < n; ++i) S_out[i] = S[i];
            }
        }
    }
    return best_t;
}

// K atoms of a layer on a handle whose calls bring up to M output pixels: the count (1..8, atoms of at least four steps) the cost model likes best at M — 272
// blocks on 256 CUs are two rounds, 238 are one.  A function of the layer and the HANDLE (its max_batch), never of the call: batch invariance holds.
int conv_sk_plan_atoms(long long M, int cout_pad, int nk, int num_cus, int split) {
    const int dflt = conv_sk_slices(nk);
    int tile, form;
    SkShape c{M, cout_pad, nk, dflt};
    double best_t = sk_choose(&c, 1, num_cus, split, -1, -1, &tile, &form);
    int best = dflt;
    for (int na = 1; na <= 8; ++na) {
        if (na == dflt || (na > 1 && nk / na < 4)) continue;
        c.NA = na;
        const double t = sk_choose(&c, 1, num_cus, split, -1, -1, &tile, &form);
        if (t < best_t - 0.5) best_t = t, best = na; // (the default unless clearly better)
    }
    return best;
}

bool conv_sk_eligible(const ConvArgs &a, int split) {
    const int esz = split ? 2 : 4;
    const int cin_real = split ? a.Cin / 2 : a.Cin;
    if (a.KH != a.KW || (a.KH != 1 && a.KH != 3) || a.pad != a.KH / 2 || (a.stride != 1 && a.stride != 2)) return false;
    if (cin_real % 32 || cin_real <= 0 || (long long)a.Kpad * esz != (long long)a.KH * a.KW * cin_real * 4) return false; // rows of 32 channels, no K padding
    if (a.out2 || a.tail_w) return false; // (tile_w, the igemm kernel's 2-D pixel tiles, is ignored: pixels are walked linearly here)
    if (a.CoutPad % 32 || a.Cout % 8 || a.Cout > a.CoutPad) return false;
    if (a.in2 && (a.KH != 1 || a.stride != 1 || a.in2_split <= 0 || (a.in2_split * esz) % 128 || (a.H & 1) || (a.W & 1))) return false;
    if (a.M <= 0 || a.M > 0x7fffffffLL) return false;
    if ((long long)a.H * a.W * a.in_ld * esz * 2 > 0x7fffffffLL) return false; // lane offsets: two neighbouring images inside 31 bits
    return true;
}

size_t conv_sk_partial_bytes(const ConvArgs &a, int split, int atoms) {
    const int esz = split ? 2 : 4;
    const int nk = (int)((long long)a.Kpad * esz / 128);
    const int S = atoms > 0 ? atoms : conv_sk_slices(nk);
    return S > 1 ? (size_t)S * (size_t)a.M * a.CoutPad * sizeof(float) : 0;
}

size_t conv_sk_ticket_count(long long M, int cout_pad) { return (size_t)((M + 63) / 64) * (size_t)(cout_pad / 32); } // the smallest tile: 64 px x 32 couts

// fills the kernel's view of one member; `a` as the implicit-GEMM launchers take it (split: pseudo-channel arguments)
static hipError_t sk_fill(SkArgs &k, const SkMember &m, int split) {
    const ConvArgs &a = m.a;
    if (!conv_sk_eligible(a, split)) return hipErrorInvalidValue;
    const int esz = split ? 2 : 4;
    std::memset(&k, 0, sizeof(k));
    k.in = reinterpret_cast<const char *>(a.in), k.in_ldb = (unsigned)a.in_ld * esz, k.in_offb = (unsigned)a.in_coff * esz;
    if (a.in2) {
        k.in2 = reinterpret_cast<const char *>(a.in2), k.in2_ldb = (unsigned)a.in2_ld * esz, k.in2_offb = (unsigned)a.in2_coff * esz;
        k.in2_blocks = a.in2_split * esz / 128;
    }
    k.N = a.N, k.H = a.H, k.W = a.W, k.Ho = a.Ho, k.Wo = a.Wo;
    k.cpb = a.Cin * esz / 128;
    k.KW = a.KW, k.stride = a.stride, k.pad = a.pad;
    k.nk = (int)((long long)a.Kpad * esz / 128);
    k.NA = m.atoms > 0 ? m.atoms : conv_sk_slices(k.nk);
    if (k.NA > 8 || k.NA > k.nk) return hipErrorInvalidValue;
    k.S = k.NA;
    k.w = reinterpret_cast<const char *>(a.w), k.w_rowb = (unsigned)k.nk * 128u;
    k.bias = a.bias;
    k.Cout = a.Cout, k.CoutPad = a.CoutPad, k.act = a.act;
    k.out = a.out, k.out_ld = a.out_ld, k.out_coff = a.out_coff, k.out_f32 = a.out_f32;
    k.res = a.res, k.res_ld = a.res_ld, k.res_coff = a.res_coff;
    k.partial = m.partial;
    k.tickets = nullptr; // decided with the form
    k.M = a.M;
    k.n_dyn = a.n_dyn;
    k.d_howo = make_fastdiv((unsigned)(a.Ho * a.Wo));
    k.d_wo = make_fastdiv((unsigned)a.Wo);
    k.d_cpb = make_fastdiv((unsigned)k.cpb);
    k.d_cg = make_fastdiv((unsigned)(a.Cout / 8));
    if (k.NA > 1 && !m.partial) return hipErrorInvalidValue;
    if ((k.in2 != nullptr) != (k.in2_blocks > 0) || k.in2_blocks > k.cpb) return hipErrorInvalidValue;
    return hipSuccess;
}

// n <= kSkGroupMax convs that do not depend on each other, as ONE launch.  Form (per member) and tile (per launch) are chosen by the cost model: neither
// enters the arithmetic (see the header), so a grouped launch gives every member the bits its own launch would give it.
// one launch of g.p[0 .. n) with the given tile and forms (+ the second launches of members whose slabs are not combined in-kernel)
static hipError_t sk_launch_chosen(SkGroupArgs &g, const SkMember *m, int n, int tile, const int *S, int split, hipStream_t st) {
    for (int i = 0; i < n; ++i) {
        SkArgs &k = g.p[i];
        k.S = S[i];
        k.tickets = (k.S > 1 && (long long)k.S * k.M * k.CoutPad * 4 <= sk_inkernel_max()) ? m[i].tickets : nullptr;
    }
    for (int i = n; i < kSkGroupMax; ++i) std::memset(&g.p[i], 0, sizeof(SkArgs));
    hipError_t e;
    switch (tile) {
    case 0: e = split ? sk_launch_t<true, 128, 128, 2, 4, 4>(g, n, st) : sk_launch_t<false, 128, 128, 2, 4, 4>(g, n, st); break;
    case 1: e = split ? sk_launch_t<true, 128, 64, 2, 2, 4>(g, n, st) : sk_launch_t<false, 128, 64, 2, 2, 4>(g, n, st); break;
    case 2: e = split ? sk_launch_t<true, 64, 64, 4, 2, 8>(g, n, st) : sk_launch_t<false, 64, 64, 4, 2, 8>(g, n, st); break;
    default: e = split ? sk_launch_t<true, 64, 32, 4, 1, 8>(g, n, st) : sk_launch_t<false, 64, 32, 4, 1, 8>(g, n, st); break;
    }
    if (e != hipSuccess) return e;
    for (int i = 0; i < n; ++i) { // members whose slabs are combined by a second launch (slabs above the in-kernel limit, or no tickets: WTK_SK_FINISH=1)
        const SkArgs &k = g.p[i];
        if (k.S == 1 || k.tickets) continue;
        const long long threads = k.M * (k.Cout / 8);
        if (split)
            hipLaunchKernelGGL(sk_finish_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, k);
        else
            hipLaunchKernelGGL(sk_finish_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, k);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// n <= kSkGroupMax convs that do not depend on each other, as ONE launch.  Form (per member) and tile (per launch) are chosen by the cost model: neither
// enters the arithmetic (see the header), so a grouped launch gives every member the bits its own launch would give it.
hipError_t launch_conv_sk_group(const SkMember *m, int n, int split, int num_cus, int force_tile, int force_form, hipStream_t st, SkChoice *choice) {
    if (n < 1 || n > kSkGroupMax) return hipErrorInvalidValue;
    SkGroupArgs g;
    SkShape shapes[kSkGroupMax];
    for (int i = 0; i < n; ++i) {
        const hipError_t e = sk_fill(g.p[i], m[i], split);
        if (e != hipSuccess) return e;
        shapes[i] = {g.p[i].M, g.p[i].CoutPad, g.p[i].nk, g.p[i].NA};
    }
    SkChoice local;
    SkChoice &ch = choice ? *choice : local;
    if (!ch.valid) { // the cost model runs once per (launch, batch size) of a handle: the caller keeps the choice
        ch.est_us = sk_choose(shapes, n, num_cus, split, force_tile, force_form, &ch.tile, ch.S);
        ch.separate = 0;
        if (n > 1) { // many pixels: the members' own launches (each with the tile it likes best) can beat one grid with a common tile
            double sum = 0.0;
            for (int i = 0; i < n; ++i) {
                int s1[kSkGroupMax];
                sum += sk_choose(&shapes[i], 1, num_cus, split, force_tile, force_form, &ch.m_tile[i], s1) + 2.5; // + a dependent launch
                ch.m_S[i] = s1[0];
            }
            ch.separate = sum < ch.est_us + 2.5;
            // measured (profiles/r06_notes.md section 2): grouping gains at B = 1 .. 4 of imgsz 384 / 640 and loses 4 % at B = 15, where one member alone fills the
            // chip several times over — such levels run as their members' own launches whatever the model says
            for (int i = 0; i < n; ++i)
                if (shapes[i].M > 16384) ch.separate = 1;
        }
        ch.valid = 1;
        static const int verbose = sk_env("WTK_SK_VERBOSE", 0);
        if (verbose)
            for (int i = 0; i < n; ++i)
                std::fprintf(stderr, "conv_sk%s: M %lld cout %d nk %d atoms %d -> tile %d form S=%d (launch est %.1f us%s)\n", n > 1 ? " (grouped)" : "", g.p[i].M, g.p[i].CoutPad,
                             g.p[i].nk, g.p[i].NA, ch.separate ? ch.m_tile[i] : ch.tile, ch.separate ? ch.m_S[i] : ch.S[i], ch.est_us, ch.separate ? ", launched one by one" : "");
    }
    if (n > 1 && ch.separate) {
        for (int i = 0; i < n; ++i) {
            SkGroupArgs one;
            one.p[0] = g.p[i];
            const hipError_t e = sk_launch_chosen(one, &m[i], 1, ch.m_tile[i], &ch.m_S[i], split, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    return sk_launch_chosen(g, m, n, ch.tile, ch.S, split, st);
}

// Every (tile, forms) a launch of these members can run as — out[0] = the cost model's choice, then the others — for the handle's autotune pass
// (csrc/wtk_run.hip: sk_autotune): the model is calibrated on a few layers, the timer is not.  Test hooks and chip-filling members leave the one choice.
int conv_sk_enumerate(const SkMember *m, int n, int split, int num_cus, int force_tile, int force_form, SkChoice *out, int cap) {
    if (n < 1 || n > kSkGroupMax || cap < 1) return 0;
    SkGroupArgs g;
    SkShape shapes[kSkGroupMax];
    bool big = false;
    for (int i = 0; i < n; ++i) {
        if (sk_fill(g.p[i], m[i], split) != hipSuccess) return 0;
        shapes[i] = {g.p[i].M, g.p[i].CoutPad, g.p[i].nk, g.p[i].NA};
        big = big || shapes[i].M > 16384;
    }
    SkChoice &c0 = out[0];
    c0 = SkChoice();
    c0.est_us = sk_choose(shapes, n, num_cus, split, force_tile, force_form, &c0.tile, c0.S);
    c0.valid = 1;
    if (big) { // (launch_conv_sk_group's rule: the members' own launches)
        c0.separate = n > 1;
        for (int i = 0; i < n; ++i) {
            int s1[kSkGroupMax];
            (void)sk_choose(&shapes[i], 1, num_cus, split, force_tile, force_form, &c0.m_tile[i], s1);
            c0.m_S[i] = s1[0];
        }
        return 1;
    }
    int cnt = 1;
    if (force_tile >= 0 || force_form >= 0) return cnt;
    for (int ti = 0; ti < 4; ++ti) {
        bool fits = true;
        for (int i = 0; i < n; ++i) fits = fits && shapes[i].cout_pad % kSkTiles[ti].bn == 0;
        if (!fits) continue;
        for (int combo = 0; combo < (1 << n) && cnt < cap; ++combo) {
            SkChoice c;
            c.valid = 1, c.tile = ti;
            bool ok = true, same = ti == c0.tile;
            for (int i = 0; i < n; ++i) {
                c.S[i] = (combo >> i) & 1 ? 1 : shapes[i].NA;
                if (shapes[i].NA == 1 && ((combo >> i) & 1) == 0) ok = false; // (one atom: both forms are the same launch)
                same = same && c.S[i] == c0.S[i];
            }
            if (ok && !same) out[cnt++] = c;
        }
    }
    return cnt;
}

hipError_t launch_conv_sk(const ConvArgs &a, int split, int atoms, float *partial, unsigned *tickets, int num_cus, hipStream_t st) {
    const SkMember m{a, atoms, partial, tickets};
    return launch_conv_sk_group(&m, 1, split, num_cus, -1, -1, st, nullptr);
}

} // namespace wtk
