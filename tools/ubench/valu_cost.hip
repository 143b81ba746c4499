// Issue cost of the epilogue's VALU instructions on gfx950, alone and beside a wave that keeps the SIMD's matrix pipe busy.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cost tools/ubench/valu_cost.hip && /tmp/valu_cost
// One block per CU, 8 waves: waves 0..3 (one per SIMD) run the measured stream between two s_memtime reads, waves 4..7 either
// idle (mode 0), run back-to-back v_mfma_f32_16x16x32_f16 (mode 1) or run the same measured stream (mode 2).  Prints shader-clock
// cycles per block of 16 values (or per 16 instructions) for every stream.  It measures the VALU wave only: what the extra VALU
// issues cost the multiplying partner is not seen here (end to end the scalar SiLU form gained nothing: profiles/r02_notes.md §6).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kIters = 64;

enum Stream { S_EXP, S_RCP, S_ADD, S_MUL, S_PKADD, S_PKMUL, S_CVT, S_SILU_PK, S_SILU_SC, S_SILU_MIX, S_SILU_FMA, S_MOV, S_COUNT };
static const char *kNames[S_COUNT] = {"16 v_exp_f32", "16 v_rcp_f32", "16 v_add_f32", "16 v_mul_f32", "8 v_pk_add_f32", "8 v_pk_mul_f32", "8 v_cvt_pk_f16_f32",
                                      "SiLU x16: exp, pk_add, rcp, pk_mul, cvt_pk (the kernels' form)", "SiLU x16: exp, add, rcp, mul, cvt_pk (scalar adds / muls)",
                                      "SiLU x16 scalar, interleaved per value", "SiLU x16 scalar without the conversions", "16 v_mov_b32"};

template <int S> __device__ __forceinline__ void body(float (&x)[16], float2v (&p)[8], unsigned (&h)[8]) {
    if constexpr (S == S_EXP) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, -%0" : "+v"(x[k]));
    } else if constexpr (S == S_RCP) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
    } else if constexpr (S == S_ADD) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x[k]));
    } else if constexpr (S == S_MUL) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(x[(k + 5) & 15]));
    } else if constexpr (S == S_MOV) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "v"(x[(k + 5) & 15]));
    } else if constexpr (S == S_PKADD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(p[k]));
    } else if constexpr (S == S_PKMUL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 3) & 7]));
    } else if constexpr (S == S_CVT) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(x[2 * k]), "v"(x[2 * k + 1]));
    } else if constexpr (S == S_SILU_PK) {
        float2v e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_exp_f32 %0, -%1" : "=v"(e[k].x) : "v"(p[k].x));
            asm volatile("v_exp_f32 %0, -%1" : "=v"(e[k].y) : "v"(p[k].y));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(e[k]));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(e[k].x));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(e[k].y));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(e[k]));
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(p[k].x), "v"(p[k].y));
    } else if constexpr (S == S_SILU_SC || S == S_SILU_FMA) {
        float e[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, -%1" : "=v"(e[k]) : "v"(x[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(e[k]));
        if constexpr (S == S_SILU_SC) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(x[2 * k]), "v"(x[2 * k + 1]));
        }
    } else if constexpr (S == S_SILU_MIX) {
        // software-pipelined per value: exp(k), add(k-4), rcp(k-8), mul(k-12): a transcendental every other instruction
        float e[16];
#pragma unroll
        for (int k = 0; k < 16 + 12; ++k) {
            if (k < 16) asm volatile("v_exp_f32 %0, -%1" : "=v"(e[k]) : "v"(x[k]));
            if (k >= 4 && k - 4 < 16) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[k - 4]));
            if (k >= 8 && k - 8 < 16) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[k - 8]));
            if (k >= 12 && k - 12 < 16) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k - 12]) : "v"(e[k - 12]));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[k]) : "v"(x[2 * k]), "v"(x[2 * k + 1]));
    }
}

template <int S> __device__ __forceinline__ unsigned long long run_stream(float seed, float *sink) {
    float x[16];
    float2v p[8];
    unsigned h[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = seed + 0.01f * k;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = (float2v){seed + 0.02f * k, seed - 0.02f * k}, h[k] = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < kIters; ++it) body<S>(x, p, h);
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += x[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += p[k].x + p[k].y + (float)h[k];
    if (acc == 123.456f) *sink = acc;
    return t1 - t0;
}

template <int S> __global__ __launch_bounds__(512) void bench_kernel(int mode, unsigned long long *out, float *sink, float seed, int mfma_iters) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((wave < 4 && mode != 3) || mode == 2) {
        const unsigned long long dt = run_stream<S>(seed, sink);
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = dt;
    } else if (mode == 1 || mode == 3) {
        floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        half8 u, v;
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = (_Float16)(seed * k), v[k] = (_Float16)(seed + k);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int it = 0; it < mfma_iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a3, 0, 0, 0);
        }
        const float s = a0[0] + a1[1] + a2[2] + a3[3];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s == 123.456f) *sink = s;
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    } else {
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = 0;
    }
}

constexpr int kMfmaIters = 12000; // x 4 MFMAs x 16 cycles = 768 k cycles: the partner outlasts every measured stream (64 passes)
template <int S> static void run_all(unsigned long long *d_out, float *d_sink, std::vector<double> (&res)[3]) {
    const int blocks = 256;
    std::vector<unsigned long long> h(blocks * 8);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { // second launch is the measured one
            hipLaunchKernelGGL(bench_kernel<S>, dim3(blocks), dim3(512), 0, 0, mode, d_out, d_sink, 0.37f, kMfmaIters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        int n = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < 4; ++w) sum += (double)h[b * 8 + w], ++n;
        res[mode].push_back(sum / n / kIters);
    }
}

int main() {
    unsigned long long *d_out;
    float *d_sink;
    (void)hipMalloc(&d_out, 256 * 8 * 8);
    (void)hipMalloc(&d_sink, 4);
    std::vector<double> res[3];
    run_all<S_EXP>(d_out, d_sink, res);
    run_all<S_RCP>(d_out, d_sink, res);
    run_all<S_ADD>(d_out, d_sink, res);
    run_all<S_MUL>(d_out, d_sink, res);
    run_all<S_PKADD>(d_out, d_sink, res);
    run_all<S_PKMUL>(d_out, d_sink, res);
    run_all<S_CVT>(d_out, d_sink, res);
    run_all<S_SILU_PK>(d_out, d_sink, res);
    run_all<S_SILU_SC>(d_out, d_sink, res);
    run_all<S_SILU_MIX>(d_out, d_sink, res);
    run_all<S_SILU_FMA>(d_out, d_sink, res);
    run_all<S_MOV>(d_out, d_sink, res);
    printf("%-72s %12s %12s %12s\n", "stream (shader-clock cycles per pass, one wave)", "partner idle", "partner MFMA", "partner same");
    for (int s = 0; s < S_COUNT; ++s) printf("%-72s %12.1f %12.1f %12.1f\n", kNames[s], res[0][s], res[1][s], res[2][s]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return 0;
}
