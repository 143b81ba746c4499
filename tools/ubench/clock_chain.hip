// Shader clock and memory latency INSIDE short dependent kernels (the regime of the latency plan: ~60 launches of a few microseconds each).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_chain.hip -o /tmp/clock_chain && /tmp/clock_chain
// Each launch: 64 blocks x 256 threads; lane 0 of block 0 stamps s_memtime (shader cycles) / s_memrealtime (100 MHz) around
//   (a) a chain of 400 dependent v_fma (~1600 cycles)            -> clock the chip holds in such a kernel
//   (b) one dependent global load from a line no CU touched since the previous launch wrote it  -> kernel-to-kernel data latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k(float *chain_buf, unsigned long long *stamps, int it, float seed) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    // every thread writes a value the NEXT launch reads (producer side of (b))
    float *slot_w = chain_buf + (size_t)((it & 1) * 65536 + tid) * 32;
    const float *slot_r = chain_buf + (size_t)(((it + 1) & 1) * 65536 + ((tid * 97) & 16383)) * 32;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = *slot_r; // written by the previous launch (another CU): dependent load
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float x = seed + v;
#pragma unroll 1
    for (int i = 0; i < 400; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    asm volatile("" : "+v"(x));
    unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    *slot_w = x;
    if (tid == 0) {
        stamps[it * 4 + 0] = c1 - c0, stamps[it * 4 + 1] = r1 - r0, stamps[it * 4 + 2] = c2 - c1, stamps[it * 4 + 3] = r2 - r1;
    }
}

int main() {
    const int N = 400;
    float *buf;
    unsigned long long *st;
    hipMalloc(&buf, (size_t)2 * 65536 * 32 * 4);
    hipMemset(buf, 0, (size_t)2 * 65536 * 32 * 4);
    hipMalloc(&st, N * 4 * 8);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, buf, st, i, 1.0f);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(N * 4);
        hipMemcpy(h.data(), st, N * 4 * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz, lat_ns, lat_cyc;
        for (int i = 50; i < N; ++i) {
            mhz.push_back(100.0 * h[i * 4 + 2] / std::max<unsigned long long>(h[i * 4 + 3], 1));
            lat_ns.push_back(10.0 * h[i * 4 + 1]);
            lat_cyc.push_back((double)h[i * 4 + 0]);
        }
        std::sort(mhz.begin(), mhz.end()), std::sort(lat_ns.begin(), lat_ns.end()), std::sort(lat_cyc.begin(), lat_cyc.end());
        printf("rep %d: %.2f us per launch (eager, %d launches); clock in the fma chain median %.0f MHz (min %.0f max %.0f); dependent load of the previous launch's data: median %.0f ns = %.0f cycles (p90 %.0f ns)\n",
               rep, ms * 1e3 / N, N, mhz[mhz.size() / 2], mhz.front(), mhz.back(), lat_ns[lat_ns.size() / 2], lat_cyc[lat_cyc.size() / 2], lat_ns[lat_ns.size() * 9 / 10]);
    }
    return 0;
}
