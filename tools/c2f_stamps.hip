// Diagnostic harness for c2f32_fused_kernel: average shader cycles per stage, tile and wave (s_memtime deltas).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWTK_C2F_STAMPS -I wtracker_amd/csrc tools/c2f_stamps.hip -o /tmp/c2f_stamps
#include "../wtracker_amd/csrc/c2f_fused.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 64, HW = 160;
    const size_t px = (size_t)N * HW * HW;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint16_t> cat(px * 96), w1(32 * 320), w2(32 * 320), wc(64 * 128);
    for (auto &v : cat) v = (uint16_t)(0x3000 + (std::rand() & 0x3ff));
    auto rh = []() { return (uint16_t)(0x2800 + (std::rand() & 0x3ff) + ((std::rand() & 1) << 15)); };
    for (auto &v : w1) v = rh(); for (auto &v : w2) v = rh(); for (auto &v : wc) v = rh();
    std::vector<float> b(64, 0.01f);
    void *dcat, *dw1, *dw2, *dwc, *dout, *dz; float *db; unsigned long long *dst;
    CK(hipMalloc(&dcat, cat.size() * 2)); CK(hipMalloc(&dw1, w1.size() * 2)); CK(hipMalloc(&dw2, w2.size() * 2)); CK(hipMalloc(&dwc, wc.size() * 2));
    CK(hipMalloc(&dout, px * 64 * 2)); CK(hipMalloc(&dz, 4096)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dst, (size_t)cus * 8 * 6 * 8));
    CK(hipMemset(dz, 0, 4096)); CK(hipMemset(dst, 0, (size_t)cus * 8 * 6 * 8));
    CK(hipMemcpy(dcat, cat.data(), cat.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwc, wc.data(), wc.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    wtk::C2fArgs a{};
    a.cat = dcat, a.cat_ld = 96, a.a_coff = 0, a.b_coff = 32, a.N = N, a.H = HW, a.W = HW;
    a.w_m1 = dw1, a.w_m2 = dw2, a.b_m1 = db, a.b_m2 = db, a.Kpad_m = 320, a.w_cv2 = dwc, a.b_cv2 = db, a.Kpad_cv2 = 128;
    a.out = dout, a.out_ld = 64, a.out_coff = 0, a.zeros = dz, a.dbg_stamps = dst;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(wtk::launch_c2f_fused(a, cus, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 10; ++i) CK(wtk::launch_c2f_fused(a, cus, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)cus * 48);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    const double tpb = (double)N * 100 / cus;
    static const char *names[6] = {"issue DMA", "stage1 (3x3 + epi)", "T1 barrier", "stage2 (3x3+res)", "vm barrier", "stage3 (1x1+store)"};
    std::printf("kernel %.1f us, %.1f tiles/block\n", ms / 10 * 1e3, tpb);
    double tot = 0;
    for (int i = 0; i < 6; ++i) {
        double sum = 0;
        for (int k = 0; k < cus * 8; ++k) sum += (double)st[(size_t)k * 6 + i];
        const double avg = sum / (cus * 8.0) / tpb; tot += avg;
        std::printf("  %-20s %8.0f cyc/tile/wave\n", names[i], avg);
    }
    std::printf("  total %.0f cyc/tile\n", tot);
    return 0;
}
