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

#include <algorithm>

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


// ---------------------------------------------------------------------------------------------------------------
// Split-fp16 form ("f16x3" handles, wtk_kernels.h kSplitScale): 32 -> 32 channels, a pixel is ONE 128-byte row [hi32 | lo32] and
// so is a (tap, cout) weight row.  Persistent: a block stages all nine weight slabs (36 KB) once and walks 256-pixel tiles; per
// tile the window (<= 352 rows, 44 KB) is requested in one go, then 9 taps x (12 ds_read_b128 + 24 MFMAs) per wave with no further
// waits: hi.hi in `acc`, the two 2^-11 cross terms in `acc1` (the accumulation order of conv3x3_halo_kernel's split form).
// The layer is HBM / latency bound (split rows double the bytes: 420 MB per 64 frames at 160 x 160), so the schedule is about
// keeping requests in flight: the NEXT tile's window (and its residual rows, into registers) is requested right after the
// barrier that ends the multiply phase, i.e. before this tile's SiLU + stores, and the second block of the CU (80 KB of LDS each)
// multiplies meanwhile.  Every XCD owns one contiguous run of tiles (vertical neighbours share window rows in its L2).
// Rows and swizzles are conv3x3_halo_kernel's: pixel rows key = row & 7; weight rows key = ((row >> 1) & 1) | (((row / 8) & 3) << 1)
// with the lane -> row map (lr >> 2) * 8 + (lr & 3) + 4 i, the same slot function of (lr, lg) as that kernel's 64-cout tile.
constexpr int kSplitRows = kHaloRowsSmall;        // 352 rows: 256 + 2 * pitch + 2 -> strips of <= 45 columns
constexpr int kSplitPiecesPerWave = kSplitRows / 8 / 4; // 11 LDS-DMA pieces (8 rows) per wave: a fixed count, rows behind the window read the zero page

__global__ __launch_bounds__(256, 2) void conv3x3_c32_split_kernel(const HaloArgs a) {
    constexpr int TC = 2, TP = 4, NV = 8;
    __shared__ __attribute__((aligned(16))) char win[kSplitRows * 128];
    __shared__ __attribute__((aligned(16))) char wts[9 * 32 * 128];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;

    // ---- this block's tiles: XCD x owns tiles [start, start + count), its blocks take them round-robin
    int ntiles = a.N * a.strips * a.blocks_per_strip;
    if (a.n_dyn) ntiles = min(max(*a.n_dyn, 0), a.N) * a.strips * a.blocks_per_strip; // dynamic batch: the tiles of the first *n_dyn images
    const int per = a.grid >> 3; // blocks per XCD (the launcher makes the grid a multiple of 8)
    int start, count;
    {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        count = q + (xcd < r ? 1 : 0);
    }
    int t = blockIdx.x >> 3;
    if (t >= count) return; // block-uniform

    const int pitch = a.pitch;
    const int halo_rows = kBM + 2 * pitch + 2;
    const _Float16 *wgt = reinterpret_cast<const _Float16 *>(a.w);
    const char *zero_page = reinterpret_cast<const char *>(a.zeros);
    _Float16 *out = reinterpret_cast<_Float16 *>(a.out);
    const _Float16 *res = reinterpret_cast<const _Float16 *>(a.res);
    const int cb = lg * NV; // this lane's eight couts

    struct Tile {
        int n, xs, o0;
    };
    auto geometry = [&](int tile) {
        unsigned v = (unsigned)tile;
        unsigned q = fdiv(v, a.d_bps);
        Tile g;
        g.o0 = (int)(v - q * (unsigned)a.blocks_per_strip) * kBM;
        v = q;
        q = fdiv(v, a.d_strips);
        g.xs = (int)(v - q * (unsigned)a.strips) * a.S;
        g.n = (int)q;
        return g;
    };
    // NHWC pixel index of this lane's pixel in tile j of the wave's 64 outputs, -1 for the junk positions of the flat strip
    auto pixel = [&](const Tile &g, int j) -> long long {
        const unsigned o = (unsigned)(g.o0 + wave * 64 + j * 16 + lr);
        const unsigned y = fdiv(o, a.d_pitch);
        const int x = (int)(o - y * (unsigned)pitch);
        if ((int)y >= a.H || x >= a.S || g.xs + x >= a.W) return -1;
        return ((long long)g.n * a.H + (int)y) * a.W + g.xs + x;
    };
    // the window: lane -> (row = piece*8 + lane/8, physical chunk = lane%8)
    auto request_window = [&](const Tile &g) {
        const _Float16 *in = reinterpret_cast<const _Float16 *>(a.in) + (long long)g.n * a.H * a.W * a.in_ld + a.in_coff;
#pragma unroll
        for (int k = 0; k < kSplitPiecesPerWave; ++k) {
            const int piece = wave + 4 * k;
            const int hr = piece * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ (hr & 7);
            const unsigned flat = (unsigned)(g.o0 + hr);
            const unsigned r = fdiv(flat, a.d_pitch);
            const int cc = (int)(flat - r * (unsigned)pitch);
            const int iy = (int)r - 1, ix = g.xs + cc - 1;
            const bool ok = hr < halo_rows && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const char *src = ok ? reinterpret_cast<const char *>(in + ((long long)iy * a.W + ix) * a.in_ld + lc * 8) : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(win + piece * 1024), 16, 0, 0);
        }
    };
    // the residual rows of a tile, raw (junk positions read pixel 0): [j][0] = eight hi halves, [j][1] = their lo halves
    auto request_residual = [&](const Tile &g, half8 (&rr)[TP][2]) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long long pix = pixel(g, j);
            const _Float16 *p = res + (pix < 0 ? 0 : pix) * a.res_ld + a.res_coff + cb;
            rr[j][0] = *reinterpret_cast<const half8 *>(p);
            rr[j][1] = *reinterpret_cast<const half8 *>(p + 32);
        }
    };

    // ---- all nine weight slabs, once: LDS row = tap*32 + cout (36 pieces, nine per wave)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int piece = wave + 4 * k;
        const int row = piece * 8 + (lane >> 3);
        const int tap = row >> 5, co = row & 31;
        const int key = ((co >> 1) & 1) | (((co >> 3) & 3) << 1);
        const int lc = (lane & 7) ^ key;
        const char *src = reinterpret_cast<const char *>(wgt + (long long)co * a.Kpad + tap * 64 + lc * 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wts + piece * 1024), 16, 0, 0);
    }
    Tile cur = geometry(start + t);
    half8 rcur[TP][2];
#pragma unroll
    for (int j = 0; j < TP; ++j) rcur[j][0] = rcur[j][1] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    if (res) request_residual(cur, rcur);
    request_window(cur);

    floatx4 bias4[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) bias4[i] = (floatx4){a.bias[cb + i * 4 + 0], a.bias[cb + i * 4 + 1], a.bias[cb + i * 4 + 2], a.bias[cb + i * 4 + 3]};

    const int wrow_l = (lr >> 2) * NV + (lr & 3); // + 4*i per cout tile, + tap*32
    const int wkey_l = ((wrow_l >> 1) & 1) | (((wrow_l >> 3) & 3) << 1);
    const unsigned wfrag0 = wrow_l * 128 + ((lg ^ wkey_l) << 4);
    const int prow0 = wave * 64 + lr;

    for (;;) {
        floatx4 acc[TC][TP], acc1[TC][TP];
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = bias4[i], acc1[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // this tile's window (and, first time round, the weights) have landed

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int base = prow0 + (tap / 3) * pitch + (tap % 3);
            const unsigned pa = base * 128 + ((lg ^ (base & 7)) << 4); // tiles j: + j*2048 (key unchanged); lo halves: ^ 64
            half8 ph[TP], pl[TP], wh[TC], wl[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j) ph[j] = *reinterpret_cast<const half8 *>(win + pa + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i) wh[i] = *reinterpret_cast<const half8 *>(wts + tap * 4096 + wfrag0 + i * 512);
#pragma unroll
            for (int i = 0; i < TC; ++i) wl[i] = *reinterpret_cast<const half8 *>(wts + tap * 4096 + (wfrag0 ^ 64u) + i * 512);
#pragma unroll
            for (int j = 0; j < TP; ++j) pl[j] = *reinterpret_cast<const half8 *>(win + (pa ^ 64u) + j * 2048);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ph[j], acc[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ph[j], acc1[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], pl[j], acc1[i][j], 0, 0, 0);
        }

        __syncthreads(); // every wave is done reading the window: the next tile's requests go out before this tile's epilogue
        t += per;
        const bool more = t < count; // block-uniform
        Tile nxt = cur;
        half8 rnext[TP][2];
#pragma unroll
        for (int j = 0; j < TP; ++j) rnext[j][0] = rcur[j][0], rnext[j][1] = rcur[j][1];
        if (more) {
            nxt = geometry(start + t);
            if (res) request_residual(nxt, rnext);
            request_window(nxt);
        }

        // ---- epilogue: lane (pixel lr of tile j, group lg) owns couts 8 lg .. 8 lg + 7 -> one 16-byte hi run and one 16-byte lo run
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const long long pix = pixel(cur, j);
            if (pix < 0) continue;
            float v[NV];
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = wtk_split_value(acc[i][j][r], acc1[i][j][r]);
            if (a.act) wtk_silu_scaled_run<NV>(v);
            if (res) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] += wtk_split_value((float)rcur[j][0][e], (float)rcur[j][1][e]);
            }
            wtk_split_store<NV>(out + pix * a.out_ld + a.out_coff, cb, v);
        }
        if (!more) break;
        cur = nxt;
#pragma unroll
        for (int j = 0; j < TP; ++j) rcur[j][0] = rnext[j][0], rcur[j][1] = rnext[j][1];
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

// split-fp16 operands: real channel counts here (the arguments of launch_conv3x3_c32_split are in pseudo-channels but Cout / CoutPad)
bool c32_split_eligible(int k, int stride, int cin, int cout, bool plain_out) { return k == 3 && stride == 1 && cin == 32 && cout == 32 && plain_out; }
int c32_split_rows_max() { return kSplitRows; }

hipError_t launch_conv3x3_c32_split(HaloArgs a, hipStream_t stream) {
    if (a.Cin != 64 || a.Cout != 32 || a.CoutPad != 32 || a.out2 || a.tail_w || a.Kpad != 9 * 64) return hipErrorInvalidValue;
    if (a.in_ld % 64 || a.in_coff % 64 || a.out_ld % 64 || a.out_coff % 64) return hipErrorInvalidValue;
    if (a.res && (a.res_ld % 64 || a.res_coff % 64)) return hipErrorInvalidValue;
    if (a.pitch != a.S + 2 || kBM + 2 * a.pitch + 2 > kSplitRows || a.strips * a.S < a.W) return hipErrorInvalidValue;
    if (a.blocks_per_strip * kBM < a.H * a.pitch) return hipErrorInvalidValue;
    const long long tiles = (long long)a.N * a.strips * a.blocks_per_strip;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    a.d_bps = make_fastdiv((unsigned)a.blocks_per_strip);
    a.d_strips = make_fastdiv((unsigned)a.strips);
    a.d_pitch = make_fastdiv((unsigned)a.pitch);
    // two persistent blocks per CU (80 KB of LDS each), a multiple of 8 so that every XCD gets the same number of them
    const long long cus = a.persist_cus > 0 ? a.persist_cus : 256;
    long long blocks = std::min(2 * cus, (tiles + 7) / 8 * 8);
    blocks = std::max(8LL, blocks / 8 * 8);
    a.grid = (int)blocks;
    hipLaunchKernelGGL(conv3x3_c32_split_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
