// Diagnostic harness for conv3x3_halo_kernel: per-block start / end times (s_memrealtime, 100 MHz) of one layer shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWTK_HALO_STAMPS -I wtracker_amd/csrc tools/halo_stamps.hip -o /tmp/halo_stamps
#include "../wtracker_amd/csrc/conv3x3_halo.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int N = 64, HW = argc > 1 ? std::atoi(argv[1]) : 20, C = argc > 2 ? std::atoi(argv[2]) : 256, CO = argc > 3 ? std::atoi(argv[3]) : 256;
    const int slabs = argc > 4 ? std::atoi(argv[4]) : 3;
    const size_t px = (size_t)N * HW * HW;
    std::vector<uint16_t> in(px * C), w((size_t)CO * 9 * C);
    for (auto &v : in) v = (uint16_t)(0x3000 + (std::rand() & 0x3ff));
    for (auto &v : w) v = (uint16_t)(0x2000 + (std::rand() & 0x3ff) + ((std::rand() & 1) << 15));
    std::vector<float> b(CO, 0.01f);
    void *din, *dw, *dout, *dz;
    float *db;
    unsigned long long *dst;
    CK(hipMalloc(&din, in.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&dout, px * CO * 2)); CK(hipMalloc(&db, CO * 4)); CK(hipMalloc(&dz, 4096));
    CK(hipMemset(dz, 0, 4096));
    CK(hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), CO * 4, hipMemcpyHostToDevice));
    wtk::HaloArgs a{};
    a.in = din, a.in_ld = C, a.N = N, a.H = HW, a.W = HW, a.Cin = C, a.Cout = CO, a.CoutPad = CO, a.w = dw, a.bias = db, a.out = dout, a.out_ld = CO, a.act = 1, a.Kpad = 9 * C, a.zeros = dz, a.slabs = slabs;
    wtk::halo_geometry_stacked(N, HW, HW, wtk::halo_rows_max(CO, slabs), &a.S, &a.pitch, &a.strips, &a.blocks_per_strip);
    const int bn = wtk::halo_cout_tile(CO);
    const long long blocks = (long long)a.strips * a.blocks_per_strip * (CO / bn);
    CK(hipMalloc(&dst, blocks * 8 * 4 * 8));
    CK(hipMemset(dst, 0, blocks * 8 * 4 * 8));
    a.dbg_stamps = dst;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(wtk::launch_conv3x3_halo(a, 1, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) CK(wtk::launch_conv3x3_halo(a, 1, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(blocks * 32);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    std::vector<double> start, dur, pro, loop, epi;
    for (long long i = 0; i < blocks * 8; ++i) { if (!st[i * 4]) continue; t_min = std::min(t_min, st[i * 4]); t_max = std::max(t_max, st[i * 4 + 3]); }
    for (long long i = 0; i < blocks * 8; ++i) {
        if (!st[i * 4]) continue;
        start.push_back((st[i * 4] - t_min) * 0.01); dur.push_back((st[i * 4 + 3] - st[i * 4]) * 0.01);
        pro.push_back((st[i * 4 + 1] - st[i * 4]) * 0.01); loop.push_back((st[i * 4 + 2] - st[i * 4 + 1]) * 0.01); epi.push_back((st[i * 4 + 3] - st[i * 4 + 2]) * 0.01);
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    std::printf("shape %dx%d C%d->%d slabs %d: %lld blocks (bn %d, strips %d), kernel %.1f us (event avg), %.1f TF/s; first-start..last-end %.1f us\n", HW, HW, C, CO, slabs, blocks, bn,
                a.strips, ms / 20 * 1e3, 2.0 * px * CO * 9 * C / (ms / 20 * 1e-3) / 1e12, (t_max - t_min) * 0.01);
    std::printf("  wave start offset us: min %.1f med %.1f p90 %.1f max %.1f\n", q(start, 0), q(start, .5), q(start, .9), q(start, 1));
    std::printf("  wave lifetime us:     min %.1f med %.1f p90 %.1f max %.1f\n", q(dur, 0), q(dur, .5), q(dur, .9), q(dur, 1));
    std::printf("  setup->loop us med %.2f | main loop med %.1f p90 %.1f max %.1f | epilogue med %.2f p90 %.2f max %.2f\n", q(pro, .5), q(loop, .5), q(loop, .9), q(loop, 1), q(epi, .5), q(epi, .9), q(epi, 1));
    return 0;
}
