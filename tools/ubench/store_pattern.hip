// Store-issue microbenchmark (round 3): what does the lane -> address pattern of a conv epilogue's 16-byte stores cost?
// Every conv kernel of this library ends with "a lane owns 16 consecutive couts (32 B) of ONE pixel": a wave's store instruction then writes 64 pieces of
// 16 B to 16 different pixel rows (4 pieces per row, 32 B apart) — lanes that are neighbours in lane id never share a cache line.  Patterns:
//   0  as the kernels do it:     lane (lr = l & 15, lg = l >> 4): pixel lr, bytes lg*32 + h*16          (two instructions h = 0, 1 per 16 pixels x 128 B)
//   1  quad-contiguous:          lane l: pixel l >> 2, bytes (l & 3)*16 + h*64                         (4 neighbouring lanes = 64 contiguous bytes)
//   2  fully linear (reference): lane l: byte l*16 + h*1024
// Every block (512 threads, as the window kernels) writes its own tiles of 16 pixels x `ld` bytes; `reps` tiles per wave.  Prints us and GB/s per pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT> __global__ __launch_bounds__(512) void k(char *out, int ld, int reps, long long wave_stride) {
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *base = out + ((long long)blockIdx.x * 8 + wave) * wave_stride;
    const uint4 v = make_uint4(l, wave, blockIdx.x, 7);
    for (int r = 0; r < reps; ++r) {
        char *t = base + (long long)r * 16 * ld; // 16 pixels of `ld` bytes; the wave writes the first 128 B of each
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long long off;
            if (PAT == 0) off = (long long)(l & 15) * ld + (l >> 4) * 32 + h * 16;
            else if (PAT == 1) off = (long long)((l >> 2) ) * ld + (l & 3) * 16 + h * 64;
            else off = (long long)l * 16 + h * 1024;
            *reinterpret_cast<uint4 *>(t + off) = v;
        }
    }
}

int main() {
    const int ld = 384, reps = 256, blocks = 256 * 2; // ld: a C2f concat buffer's 192 channels
    const long long wave_stride = (long long)reps * 16 * ld;
    const long long bytes = wave_stride * 8 * blocks;
    char *buf;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const double payload = (double)blocks * 8 * reps * 2048;
    for (int round = 0; round < 3; ++round)
        for (int pat = 0; pat < 3; ++pat) {
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it) {
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, buf, ld, reps, wave_stride);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, buf, ld, reps, wave_stride);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, buf, ld, reps, wave_stride);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("round %d pattern %d: %.1f us per launch, %.0f GB/s of payload (%.1f B/clk/CU at 2.1 GHz)\n", round, pat, ms * 1000 / 5, payload / (ms / 5 * 1e-3) / 1e9,
                   payload / (ms / 5 * 1e-3) / 256 / 2.1e9);
        }
    return 0;
}
