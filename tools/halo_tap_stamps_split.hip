// Diagnostic harness: where a tap of the SPLIT (f16x3) conv3x3_halo_kernel (three-slab form, one tile per block) spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWTK_HALO_TAP_STAMPS -I wtracker_amd/csrc tools/halo_tap_stamps_split.hip -o /tmp/halo_tap_split
#include "../wtracker_amd/csrc/conv3x3_halo.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int N = 64, HW = argc > 1 ? std::atoi(argv[1]) : 20, C = argc > 2 ? std::atoi(argv[2]) : 256, CO = argc > 3 ? std::atoi(argv[3]) : 256;
    const size_t px = (size_t)N * HW * HW;
    std::vector<uint16_t> in(px * C * 2), w((size_t)CO * 9 * C * 2); // split rows: [hi32 | lo32] per 32 channels (random halves are fine for timing)
    for (auto &v : in) v = (uint16_t)(0x3000 + (std::rand() & 0x3ff));
    for (auto &v : w) v = (uint16_t)(0x2000 + (std::rand() & 0x3ff) + ((std::rand() & 1) << 15));
    std::vector<float> b(CO, 0.01f);
    void *din, *dw, *dout, *dz; float *db; unsigned long long *dst;
    CK(hipMalloc(&din, in.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&dout, px * CO * 4)); CK(hipMalloc(&db, CO * 4)); CK(hipMalloc(&dz, 4096));
    CK(hipMemset(dz, 0, 4096));
    CK(hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), CO * 4, hipMemcpyHostToDevice));
    wtk::HaloArgs a{};
    a.in = din, a.in_ld = 2 * C, a.N = N, a.H = HW, a.W = HW, a.Cin = 2 * C, a.Cout = CO, a.CoutPad = CO, a.w = dw, a.bias = db, a.out = dout, a.out_ld = 2 * CO, a.act = 1, a.Kpad = 9 * 2 * C, a.zeros = dz, a.slabs = 3; // pseudo-channels
    wtk::halo_geometry_stacked(N, HW, HW, wtk::kHaloRowsMax, &a.S, &a.pitch, &a.strips, &a.blocks_per_strip);
    const int bn = wtk::split_halo_cout_tile(CO);
    const long long blocks = (long long)a.strips * a.blocks_per_strip * (CO / bn);
    CK(hipMalloc(&dst, blocks * 8 * 4 * 8)); CK(hipMemset(dst, 0, blocks * 8 * 4 * 8));
    a.dbg_stamps = dst;
    for (int i = 0; i < 5; ++i) CK(wtk::launch_conv3x3_halo_split(a, nullptr));
    CK(hipDeviceSynchronize());
    if (argc > 4) { // hold the chip under load for argv[4] seconds first: the clock it settles at is the one a long run sees
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, nullptr));
        float ms = 0;
        while (ms < 1000.f * std::atof(argv[4])) {
            for (int i = 0; i < 200; ++i) CK(wtk::launch_conv3x3_halo_split(a, nullptr));
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
    }
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 20; ++i) CK(wtk::launch_conv3x3_halo_split(a, nullptr));
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%dx%d C%d->%d: %.1f us per launch, %.0f TF/s\n", HW, HW, C, CO, ms * 50, 2.0 * px * CO * 9 * C / (ms * 50e-6) * 1e-12 /* algorithmic */);
    }
    std::vector<unsigned long long> st(blocks * 32);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    const double taps = 9.0 * (C / 32);
    double s0 = 0, s1 = 0, s2 = 0; long long n = 0;
    for (long long i = 0; i < blocks * 8; ++i) { if (!st[i * 4]) continue; s0 += st[i * 4]; s1 += st[i * 4 + 1]; s2 += st[i * 4 + 2]; ++n; }
    double wmin = 0, wmax = 0; long long nb = 0; // slowest / fastest wave of a block (work phase)
    for (long long b = 0; b < blocks; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w8 = 0; w8 < 8; ++w8) { const unsigned long long v = st[(b * 8 + w8) * 4]; if (!v) continue; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        if (hi) wmin += lo, wmax += hi, ++nb;
    }
    std::printf("   fastest wave of a block %.0f cyc/tap, slowest %.0f\n", wmin / nb / taps, wmax / nb / taps);
    {
        std::vector<double> clk;
        for (long long i = 0; i < blocks * 8; ++i) { const unsigned long long v = st[i * 4 + 3]; if (v & 0xffffff) clk.push_back(100.0 * (double)(v >> 24) / (double)(v & 0xffffff)); }
        if (!clk.empty()) { std::sort(clk.begin(), clk.end()); std::printf("   in-kernel clock (median over waves): %.0f MHz\n", clk[clk.size() / 2]); }
    }
    const double mf = (bn == 128 ? 48 : 24) * 16.0; // MFMAs per wave per tap x 16 cycles (three per tile pair)
    std::printf("%dx%d C%d->%d (bn %d): per tap and wave: work %.0f cyc (MFMA issue alone %.0f), vmcnt wait %.0f, barrier wait %.0f\n", HW, HW, C, CO, bn, s0 / n / taps, mf,
                s1 / n / taps, s2 / n / taps);
    return 0;
}
