// Diagnostic harness: where a K step of the SPLIT (f16x3) 1x1 implicit GEMM (128 x 128 tile, two blocks of four waves per CU) spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWTK_IGEMM_STAMPS -I wtracker_amd/csrc tools/igemm_stamps_split.hip -o tools/bin/igemm_stamps_split
//   tools/bin/igemm_stamps_split HW CIN COUT [seconds under load]
#include "../wtracker_amd/csrc/conv_igemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int N = 64, HW = argc > 1 ? std::atoi(argv[1]) : 40, C = argc > 2 ? std::atoi(argv[2]) : 512, CO = argc > 3 ? std::atoi(argv[3]) : 256;
    const size_t px = (size_t)N * HW * HW;
    std::vector<uint16_t> in(px * C * 2), w((size_t)CO * C * 2);
    for (auto &v : in) v = (uint16_t)(0x3000 + (std::rand() & 0x3ff));
    for (auto &v : w) v = (uint16_t)(0x2000 + (std::rand() & 0x3ff) + ((std::rand() & 1) << 15));
    std::vector<float> b(CO, 0.01f);
    void *din, *dw, *dout, *dz; float *db; unsigned long long *dst;
    CK(hipMalloc(&din, in.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&dout, px * CO * 4)); CK(hipMalloc(&db, CO * 4)); CK(hipMalloc(&dz, 4096));
    CK(hipMemset(dz, 0, 4096));
    CK(hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), CO * 4, hipMemcpyHostToDevice));
    wtk::ConvArgs a{};
    a.in = din, a.in_ld = 2 * C, a.N = N, a.H = HW, a.W = HW, a.Cin = 2 * C, a.Ho = HW, a.Wo = HW, a.Cout = CO, a.CoutPad = CO, a.KH = a.KW = 1, a.stride = 1, a.pad = 0;
    a.w = dw, a.bias = db, a.out = dout, a.out_ld = 2 * CO, a.act = 1, a.K = 2 * C, a.Kpad = 2 * C, a.M = (long long)px, a.zeros = dz; // pseudo-channels
    const int cus = wtk::current_device_cus();
    const size_t nst = (size_t)2 * cus * 4 * 8;
    CK(hipMalloc(&dst, nst * 8)); CK(hipMemset(dst, 0, nst * 8));
    a.dbg_stamps = dst;
    for (int i = 0; i < 5; ++i) CK(wtk::launch_conv_split(a, wtk::CFG_128x128, nullptr));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 4) {
        CK(hipEventRecord(e0, nullptr));
        float ms = 0;
        while (ms < 1000.f * std::atof(argv[4])) {
            for (int i = 0; i < 200; ++i) CK(wtk::launch_conv_split(a, wtk::CFG_128x128, nullptr));
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
    }
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) CK(wtk::launch_conv_split(a, wtk::CFG_128x128, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%dx%d 1x1 C%d->%d: %.1f us per launch, %.0f TF/s algorithmic, %.0f GB/s algorithmic\n", HW, HW, C, CO, ms * 50, 2.0 * px * CO * C / (ms * 50e-6) * 1e-12,
                (double)px * (C + CO) * 4 / (ms * 50e-6) * 1e-9);
    std::vector<unsigned long long> st(nst);
    CK(hipMemcpy(st.data(), dst, nst * 8, hipMemcpyDeviceToHost));
    double s[4] = {0, 0, 0, 0}, steps = 0; long long n = 0; std::vector<double> clk;
    for (size_t i = 0; i < nst / 8; ++i) {
        const unsigned long long *o = &st[i * 8];
        if (!o[4]) continue;
        for (int k = 0; k < 4; ++k) s[k] += (double)o[k];
        steps += (double)o[4]; ++n;
        if (o[6]) clk.push_back(100.0 * (double)o[5] / (double)o[6]);
    }
    std::sort(clk.begin(), clk.end());
    std::printf("   per K step and wave (48 MFMAs = 768 cycles alone): request issue %.0f, reads + multiply (+ epilogue share) %.0f, vmcnt wait %.0f, barrier wait %.0f cycles; clock %.0f MHz; %.1f K steps per wave\n",
                s[0] / steps, s[1] / steps, s[2] / steps, s[3] / steps, clk.empty() ? 0.0 : clk[clk.size() / 2], steps / n);
    return 0;
}
