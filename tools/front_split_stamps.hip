// Diagnostic harness for front_fused_split_kernel (f16x3 handles): builds the kernel with -DWTK_FRONT_STAMPS and prints the
// average shader cycles each stage costs per tile and wave (s_memtime deltas).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWTK_FRONT_STAMPS -I wtracker_amd/csrc tools/front_split_stamps.hip -o /tmp/front_split_stamps
#include "../wtracker_amd/csrc/front_fused_split.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 64, S = argc > 2 ? std::atoi(argv[2]) : 640;
    const int Ho = S / 4;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint8_t> fr((size_t)N * S * S);
    for (auto &v : fr) v = (uint8_t)(std::rand() & 0xff);
    std::vector<float> w0(2048); // fp32 packing (1152 floats) or split packing (4096 halves): either fits, values are arbitrary small numbers
    std::vector<uint16_t> w1(64 * 576), w2(64 * 128);
    auto rh = []() { return (uint16_t)(0x2800 + (std::rand() & 0x3ff) + ((std::rand() & 1) << 15)); }; // ~ +-0.03..0.06
    for (auto &v : w0) { const uint16_t hb = rh(); const uint32_t wd = (uint32_t)hb | ((uint32_t)rh() << 16); std::memcpy(&v, &wd, 4); } // two small fp16 numbers per word = one small-ish fp32 number
    for (auto &v : w1) v = rh();
    for (auto &v : w2) v = rh();
    std::vector<float> b(64, 0.01f);
    uint8_t *dfr;
    void *dw0, *dw1, *dw2, *dout;
    float *db;
    unsigned long long *dst;
    CK(hipMalloc(&dfr, fr.size()));
    CK(hipMalloc(&dw0, w0.size() * 4));
    CK(hipMalloc(&dw1, w1.size() * 2));
    CK(hipMalloc(&dw2, w2.size() * 2));
    CK(hipMalloc(&db, 64 * 4));
    CK(hipMalloc(&dout, (size_t)N * Ho * Ho * 192 * 2));
    CK(hipMalloc(&dst, (size_t)cus * 8 * 8 * 8));
    CK(hipMemcpy(dfr, fr.data(), fr.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dw0, w0.data(), w0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 64 * 4, hipMemcpyHostToDevice));
    wtk::FrontArgs a{};
    a.frames = dfr, a.N = N, a.H = S, a.W = S, a.C = 1;
    a.w0 = dw0, a.b0 = db, a.w1 = dw1, a.b1 = db, a.Kpad1 = 576, a.w2 = dw2, a.b2 = db, a.Kpad2 = 128;
    a.stem_split = argc > 3 ? std::atoi(argv[3]) : 1;
    a.out = dout, a.out_ld = 192, a.out_coff = 0;
    a.dbg_stamps = dst;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(wtk::launch_front_fused_split(a, cus, nullptr));
    CK(hipEventRecord(e0, nullptr));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) CK(wtk::launch_front_fused_split(a, cus, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)cus * 64);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    const int tiles = N * ((Ho + 15) / 16) * ((Ho + 3) / 4);
    const double tiles_per_block = (double)tiles / cus;
    static const char *names[8] = {"A convert", "A flush+barrier", "B stem", "B barrier", "C model.1", "C barrier", "D cv1", "D barrier"};
    std::printf("kernel %.1f us, %d tiles, %.1f tiles/block\n", ms / reps * 1e3, tiles, tiles_per_block);
    double tot = 0;
    for (int i = 0; i < 8; ++i) {
        double sum = 0, mx = 0;
        for (int b2 = 0; b2 < cus; ++b2)
            for (int w = 0; w < 8; ++w) {
                const double v = (double)st[((size_t)b2 * 8 + w) * 8 + i];
                sum += v;
                if (v > mx) mx = v;
            }
        const double avg = sum / (cus * 8.0) / tiles_per_block;
        tot += avg;
        std::printf("  %-16s avg %8.0f cyc/tile/wave   (max wave total %.0f)\n", names[i], avg, mx);
    }
    std::printf("  total %.0f cyc/tile\n", tot);
    // per-wave view of block 0
    for (int w = 0; w < 8; ++w) {
        std::printf("  block0 wave%d:", w);
        for (int i = 0; i < 8; ++i) std::printf(" %7.0f", (double)st[(size_t)w * 8 + i] / tiles_per_block);
        std::printf("\n");
    }
    return 0;
}
