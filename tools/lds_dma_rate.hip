// Micro-benchmark: how fast can a CU take data in through LDS-DMA?  Every block (one per CU, W waves) requests 1-KiB pieces in a
// loop from an L2-resident buffer (per-block region of `region` bytes, default 64 KiB = a weight tile re-read by every block)
// and waits with a counted vmcnt that keeps D pieces per wave in flight.  Variants:
//   0  global_load_lds_dwordx4, per-lane 64-bit addresses, piece = 8 rows x 128 B of a [rows][stride] matrix (the conv kernels' form)
//   1  global_load_lds_dwordx4, piece = 1 KiB contiguous
//   2  buffer_load_dwordx4 ... offen lds (SGPR resource + 32-bit per-lane offsets), piece = 8 rows x 128 B
//   3  buffer_load_dwordx4 ... offen lds, piece = 1 KiB contiguous
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_dma_rate.hip -o tools/_bin/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int rsrc_t __attribute__((ext_vector_type(4)));

template <int VAR, int D> __global__ __launch_bounds__(1024) void dma_kernel(const char *src, long long region, int stride, int iters, unsigned long long *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const char *base = src + (long long)blockIdx.x % 8 * region; // 8 distinct regions: L2 resident, shared by the blocks of an XCD label
    // per-lane offset inside a piece
    const unsigned rowform = (unsigned)((lane >> 3) * stride + (lane & 7) * 16);
    const unsigned lin = (unsigned)lane * 16;
    const unsigned loff = (VAR == 0 || VAR == 2) ? rowform : lin;
    const unsigned piece_bytes = (VAR == 0 || VAR == 2) ? 8u * stride : 1024u;
    const unsigned pieces = (unsigned)(region / piece_bytes);
    rsrc_t rs;
    {
        const unsigned long long b = (unsigned long long)base;
        rs.x = (int)(b & 0xffffffffu);
        rs.y = (int)((b >> 32) & 0xffff); // stride 0
        rs.z = (int)region;               // num_records (bytes)
        rs.w = 0x00020000;                // raw buffer: DATA_FORMAT = 32 (gfx9 family)
    }
    char *dst0 = lds + wave * (D + 1) * 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned p = (unsigned)wave % pieces; // (a start index past the region faulted in the first version)
    for (int it = 0; it < iters; ++it) {
        char *dst = dst0 + (it % (D + 1)) * 1024;
        const unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dst;
        const unsigned off = p * piece_bytes + loff;
        if (VAR < 2) {
            const char *a = base + off;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(ldsaddr) : "memory");
        } else {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rs), "s"(ldsaddr) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        p = (p + (unsigned)nw) % pieces;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

// Interference: waves 0..WD-1 request pieces as above (variant 2, buffer form, 3 in flight), the other waves of the block keep the CU's
// LDS busy with ds_read_b128 (MODE 1), its matrix pipes with MFMAs (MODE 2), or both (MODE 3: 8 reads + 16 MFMAs per iteration, the
// conv kernels' mix) until the requesters are done.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(1024) void dma_interf_kernel(const char *src, long long region, int stride, int iters, int WD, unsigned long long *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int done;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < WD) {
        const char *base = src + (long long)blockIdx.x % 8 * region;
        const unsigned loff = (unsigned)((lane >> 3) * stride + (lane & 7) * 16);
        const unsigned piece_bytes = 8u * stride;
        const unsigned pieces = (unsigned)(region / piece_bytes);
        rsrc_t rs;
        const unsigned long long b = (unsigned long long)base;
        rs.x = (int)(b & 0xffffffffu), rs.y = (int)((b >> 32) & 0xffff), rs.z = (int)region, rs.w = 0x00020000;
        char *dst0 = lds + wave * 4 * 1024;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned p = (unsigned)wave % pieces;
        for (int it = 0; it < iters; ++it) {
            char *dst = dst0 + (it & 3) * 1024;
            const unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dst;
            const unsigned off = p * piece_bytes + loff;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rs), "s"(ldsaddr) : "memory");
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            p = (p + (unsigned)WD) % pieces;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            out[blockIdx.x * 16 + wave] = t1 - t0;
            atomicAdd(&done, 1);
        }
    } else {
        // busy waves: read 8 KiB regions of LDS above the requesters' slots; bounded (iters * 64 iterations at most)
        const char *rb = lds + 64 * 1024 + (wave - WD) * 2048;
        floatx4 acc[4] = {};
        half8 fa = {}, fb = {};
        uint4 sink = {0, 0, 0, 0};
        for (int it = 0; it < iters * 64; ++it) {
            if (MODE & 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 v;
                    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)(rb + ((j & 1) * 1024) + lane * 16);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); // (keeps a few reads in flight; the last ones are collected below)
                    sink.x ^= v.x;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[j & 3], 0, 0, 0);
            }
            if ((it & 15) == 15 && *reinterpret_cast<volatile int *>(&done) >= WD) break;
        }
        if (sink.x == 0x12345678u || acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 1.2345f) out[0] = 1; // keep the work alive
    }
}

template <int MODE> int run_interf(const char *src, long long region, int stride, int wd, int wb, unsigned long long *dout, int ncu) {
    const int iters = 2000;
    const size_t shm = 64 * 1024 + (size_t)wb * 2048;
    CK(hipFuncSetAttribute((const void *)dma_interf_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((dma_interf_kernel<MODE>), dim3(ncu), dim3((wd + wb) * 64), shm, nullptr, src, region, stride, iters, wd, dout);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(ncu * 16);
    CK(hipMemcpy(st.data(), dout, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0; int n = 0;
    for (int b = 0; b < ncu; ++b) for (int w = 0; w < wd; ++w) { cyc += (double)st[b * 16 + w]; ++n; }
    cyc /= n;
    std::printf("requesters %d + busy waves %d (%s): %6.1f B/clk/CU, %5.0f cycles per piece per requesting wave\n", wd, wb,
                MODE == 1 ? "ds_read_b128" : MODE == 2 ? "MFMA" : MODE == 3 ? "8 ds_read_b128 + 16 MFMA" : "idle", (double)wd * iters * 1024 / cyc, cyc / iters);
    return 0;
}

template <int VAR, int D> int run(const char *src, long long region, int stride, int waves, unsigned long long *dout, int ncu) {
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shm = (size_t)waves * (D + 1) * 1024;
    hipLaunchKernelGGL((dma_kernel<VAR, D>), dim3(ncu), dim3(waves * 64), shm, nullptr, src, region, stride, iters, dout);
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((dma_kernel<VAR, D>), dim3(ncu), dim3(waves * 64), shm, nullptr, src, region, stride, iters, dout);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(ncu * 16);
    CK(hipMemcpy(st.data(), dout, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0; int n = 0;
    for (int b = 0; b < ncu; ++b) for (int w = 0; w < waves; ++w) { cyc += (double)st[b * 16 + w]; ++n; }
    cyc /= n;
    const double bytes_cu = (double)waves * iters * 1024;
    std::printf("variant %d  waves/CU %2d  in flight/wave %d: %6.1f B/clk/CU (%5.0f cycles per piece per wave), %6.2f TB/s chip\n", VAR, waves, D, bytes_cu / cyc, cyc / iters,
                bytes_cu * ncu / (ms * 1e-3) * 1e-12);
    return 0;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const long long region = argc > 1 ? std::atoll(argv[1]) : 65536;
    const int stride = argc > 2 ? std::atoi(argv[2]) : 1536; // row pitch in bytes of the "matrix" (Kpad * 2)
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    char *src; unsigned long long *dout;
    CK(hipMalloc(&src, region * 8 + 4096)); CK(hipMemset(src, 1, region * 8 + 4096)); CK(hipMalloc(&dout, ncu * 16 * 8));
    for (int waves : {2, 4, 8, 16}) {
        run<0, 3>(src, region, stride, waves, dout, ncu);
        run<1, 3>(src, region, stride, waves, dout, ncu);
        run<2, 3>(src, region, stride, waves, dout, ncu);
        run<3, 3>(src, region, stride, waves, dout, ncu);
    }
    for (int wd : {4, 8}) {
        run_interf<1>(src, region, stride, wd, 8, dout, ncu);
        run_interf<2>(src, region, stride, wd, 8, dout, ncu);
        run_interf<3>(src, region, stride, wd, 8, dout, ncu);
    }
    run<0, 1>(src, region, stride, 8, dout, ncu);
    run<0, 7>(src, region, stride, 8, dout, ncu);
    run<2, 7>(src, region, stride, 8, dout, ncu);
    return 0;
}
