// The detector handle, part 1 (see wtk_internal.h): the YOLOv8 conv table, weight packing, graph planning (channel-slice views instead of concat / upsample
// tensors), the launch schedule of the latency plan, the stream pool, the status page, create and destroy.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include "wtk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace wtk;

namespace {
static int make_divisible8(double x) { return (int)(std::ceil(x / 8.0) * 8.0); }

static ModelDims model_dims(float width, float depth, int max_ch, int nc) {
    ModelDims d;
    const int base[5] = {64, 128, 256, 512, 1024};
    for (int i = 0; i < 5; ++i) d.c[i] = make_divisible8(std::min(base[i], max_ch) * (double)width);
    const int nb[4] = {3, 6, 6, 3};
    for (int i = 0; i < 4; ++i) d.n[i] = std::max((int)std::lround(nb[i] * (double)depth), 1);
    d.hb = std::max(std::max(16, d.c[2] / 4), 64);
    d.hc = std::max(d.c[2], std::min(nc, 100));
    d.nc = nc;
    return d;
}

static void c2f_specs(std::vector<ConvSpec> &v, const std::string &p, int c1, int c2, int n) {
    const int c = c2 / 2;
    v.push_back({p + ".cv1", 2 * c, c1, 1, 1, 1});
    v.push_back({p + ".cv2", c2, (2 + n) * c, 1, 1, 1});
    for (int i = 0; i < n; ++i) {
        v.push_back({p + ".m." + std::to_string(i) + ".cv1", c, c, 3, 1, 1});
        v.push_back({p + ".m." + std::to_string(i) + ".cv2", c, c, 3, 1, 1});
    }
}

// Fused convs in ultralytics module order
static std::vector<ConvSpec> conv_specs(const ModelDims &d) {
    std::vector<ConvSpec> v;
    const int *c = d.c;
    v.push_back({"model.0", c[0], 3, 3, 2, 1});
    v.push_back({"model.1", c[1], c[0], 3, 2, 1});
    c2f_specs(v, "model.2", c[1], c[1], d.n[0]);
    v.push_back({"model.3", c[2], c[1], 3, 2, 1});
    c2f_specs(v, "model.4", c[2], c[2], d.n[1]);
    v.push_back({"model.5", c[3], c[2], 3, 2, 1});
    c2f_specs(v, "model.6", c[3], c[3], d.n[2]);
    v.push_back({"model.7", c[4], c[3], 3, 2, 1});
    c2f_specs(v, "model.8", c[4], c[4], d.n[3]);
    v.push_back({"model.9.cv1", c[4] / 2, c[4], 1, 1, 1});
    v.push_back({"model.9.cv2", c[4], c[4] * 2, 1, 1, 1});
    c2f_specs(v, "model.12", c[4] + c[3], c[3], d.n[3]);
    c2f_specs(v, "model.15", c[3] + c[2], c[2], d.n[3]);
    v.push_back({"model.16", c[2], c[2], 3, 2, 1});
    c2f_specs(v, "model.18", c[2] + c[3], c[3], d.n[3]);
    v.push_back({"model.19", c[3], c[3], 3, 2, 1});
    c2f_specs(v, "model.21", c[3] + c[4], c[4], d.n[3]);
    const int ch[3] = {c[2], c[3], c[4]};
    for (int i = 0; i < 3; ++i) {
        const std::string p = "model.22.cv2." + std::to_string(i);
        v.push_back({p + ".0", d.hb, ch[i], 3, 1, 1});
        v.push_back({p + ".1", d.hb, d.hb, 3, 1, 1});
        v.push_back({p + ".2", 64, d.hb, 1, 1, 0});
    }
    for (int i = 0; i < 3; ++i) {
        const std::string p = "model.22.cv3." + std::to_string(i);
        v.push_back({p + ".0", d.hc, ch[i], 3, 1, 1});
        v.push_back({p + ".1", d.hc, d.hc, 3, 1, 1});
        v.push_back({p + ".2", d.nc, d.hc, 1, 1, 0});
    }
    return v;
}

static int find_spec(const std::vector<ConvSpec> &v, const std::string &name) {
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].name == name) return (int)i;
    return -1;
}

} // namespace

extern "C" int wtk_yolo_conv_count(float width_mult, float depth_mult, int32_t max_channels, int32_t nc) {
    if (nc < 1 || width_mult <= 0 || depth_mult <= 0 || max_channels < 8) return -1;
    return (int)conv_specs(model_dims(width_mult, depth_mult, max_channels, nc)).size();
}

extern "C" int wtk_yolo_conv_info(float width_mult, float depth_mult, int32_t max_channels, int32_t nc, int32_t index, int32_t *cout,
                                  int32_t *cin, int32_t *k, int32_t *stride, int32_t *act, char *name_out, size_t name_cap) {
    if (nc < 1 || width_mult <= 0 || depth_mult <= 0 || max_channels < 8) return fail("wtk_yolo_conv_info: bad model scale");
    const auto v = conv_specs(model_dims(width_mult, depth_mult, max_channels, nc));
    if (index < 0 || index >= (int)v.size()) return fail("wtk_yolo_conv_info: index out of range");
    const ConvSpec &s = v[index];
    if (cout) *cout = s.cout;
    if (cin) *cin = s.cin;
    if (k) *k = s.k;
    if (stride) *stride = s.stride;
    if (act) *act = s.act;
    if (name_out && name_cap) {
        std::snprintf(name_out, name_cap, "%s", s.name.c_str());
    }
    return 0;
}

// Streams come from a per-process pool and go back to it (never destroyed): the HIP runtime binds a stream to one of its few hardware
// queues when the stream is created, and after handles have come and gone the streams of a NEW handle can land on the queue of the
// caller's stream — the towers then run behind the PAN path instead of next to it (the whole benefit of the side streams, 14 %, was
// lost for the fourth workload of bench.py).  Reused streams keep the queues they got when the process was young.
// Lifetime protocol (round 6; tests/hostsan models it): a stream enters the pool only after it has drained (hipStreamSynchronize) and is handed
// out only when hipStreamIsCapturing says "none" — a stream that was the origin or a fork of a capture can never carry a capture state, or work
// of a destroyed handle, into the next handle.
static std::mutex g_stream_mu;
static std::vector<std::pair<int, hipStream_t>> g_free_streams; // (device, stream)
int wtk::stream_idle(hipStream_t s, const char *what) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(s, &cs));
    if (cs != hipStreamCaptureStatusNone) return fail(std::string("stream protocol violation: ") + what + " is still part of a stream capture");
    return 0;
}
int wtk::pooled_stream(int device, hipStream_t *s) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = 0; i < g_free_streams.size(); ++i)
            if (g_free_streams[i].first == device) {
                *s = g_free_streams[i].second;
                g_free_streams.erase(g_free_streams.begin() + (long)i);
                return stream_idle(*s, "a stream handed out by the pool");
            }
    }
    HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    return 0;
}
void wtk::unpool_stream(int device, hipStream_t s) {
    (void)hipStreamSynchronize(s); // nothing of the handle that held it is still queued on it (wtk_yolo_destroy has synchronised the device already: this returns at once)
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_free_streams.insert(g_free_streams.begin(), {device, s}); // LIFO: the next handle gets the streams of the last one destroyed
}

// ABI v3 symbol of the block cache of destroyed handles (an experiment of round 3, removed in round 6: it changed nothing measurable): nothing is cached
extern "C" int wtk_release_cached_memory(void) { return 0; }

static int dev_alloc(wtk_yolo *h, void **p, size_t bytes) {
    *p = nullptr;
    HIP_TRY(hipMalloc(p, bytes));
    h->dev_allocs.emplace_back(*p, bytes);
    return 0;
}

static void dev_release(wtk_yolo *h) {
    for (auto &a : h->dev_allocs) (void)hipFree(a.first);
    h->dev_allocs.clear();
}

static int pick_cfg(int cout, bool k1) {
    // 8-wave tiles (256x256, 256x128, also with a three-buffer ring) all measured 3..30 % slower than two independent
    // 4-wave 128x128 blocks per CU (profiles/r01_notes.md) and were removed.
    if (cout % 128 == 0) return CFG_128x128;
    if (k1 && cout % 64 == 0) return CFG_128x64; // 48 KB LDS, 123 VGPRs: 3 blocks per CU on the HBM-bound 1x1 layers (+8 %)
    if (cout % 64 == 0) return CFG_256x64;
    return CFG_256x32;
}

// pack [cout][k][k][cin] fp32 -> [cout_pad][Kpad] storage dtype (zero padded) on the device
// Scaled activation domain (wtk_kernels.h, kActScale): every conv but the stem reads log2(e)-scaled activations; SiLU layers
// produce scaled outputs.  w_scale = (act ? s : 1) / s, b_scale = act ? s : 1; for a SiLU layer w_scale is exactly 1.
static int pack_conv(wtk_yolo *h, Op &op, const std::vector<const float *> &w_parts, const std::vector<const float *> &b_parts,
                     const std::vector<int> &couts) {
    const int ce = h->is_f16 ? 8 : 4;
    op.K = op.k * op.k * op.cin;
    op.Kpad = (op.K + 8 * ce - 1) / (8 * ce) * (8 * ce);
    if (h->split) op.Kpad = op.K; // cin % 32 == 0 (checked at create): rows of 2 K halves, no K tail
    const int bn = op.halo == 2 ? 32 : (op.halo ? (h->split ? split_halo_cout_tile(op.cout) : halo_cout_tile(op.cout)) : conv_cfg_bn(op.cfg));
    op.cout_pad = (op.cout + bn - 1) / bn * bn;
    std::vector<float> wf((size_t)op.cout_pad * op.Kpad, 0.f), bf(op.cout_pad, 0.f);
    int row = 0;
    for (size_t p = 0; p < w_parts.size(); ++p) {
        for (int o = 0; o < couts[p]; ++o, ++row) {
            std::memcpy(&wf[(size_t)row * op.Kpad], w_parts[p] + (size_t)o * op.K, sizeof(float) * op.K);
            bf[row] = b_parts[p][o];
            if (op.act) {
                bf[row] = (float)((double)bf[row] * (double)kActScale);
            } else { // linear layer fed by scaled activations: take the scale out again
                for (int k = 0; k < op.K; ++k) wf[(size_t)row * op.Kpad + k] = (float)((double)wf[(size_t)row * op.Kpad + k] / (double)kActScale);
            }
        }
    }
    if (h->is_f16 || h->split) { // range guard of the fp16-storage modes (include/wtk_hip.h: wtk_yolo_status)
        for (size_t i = 0; i < wf.size(); ++i) {
            const float m = std::fabs(wf[i]);
            if (!(m <= 65504.0f))
                return fail("wtk_yolo_create: a folded weight of conv blob " + std::to_string(op.spec) + " (|w| = " + std::to_string(m) +
                            " in the library's scaled domain) is outside the fp16 range: this model needs dtype WTK_F32");
        }
    }
    for (float b : bf)
        if (!std::isfinite(b)) return fail("wtk_yolo_create: a bias of conv blob " + std::to_string(op.spec) + " is not finite");
    if (dev_alloc(h, (void **)&op.bias, bf.size() * sizeof(float))) return 1;
    HIP_TRY(hipMemcpy(op.bias, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
    if (h->split) {
        // [cout_pad][tap][block of 32 channels][hi32 | lo32]: 2 K halves per weight, the k order of a split activation row
        std::vector<uint16_t> wh(wf.size() * 2);
        const int taps = op.k * op.k, blocks = op.cin / 32;
        for (int r = 0; r < op.cout_pad; ++r)
            for (int t = 0; t < taps; ++t)
                for (int b = 0; b < blocks; ++b)
                    for (int c = 0; c < 32; ++c) {
                        const float x = wf[(size_t)r * op.Kpad + (size_t)t * op.cin + b * 32 + c];
                        const uint16_t hb = f32_to_f16_bits(x);
                        const float hi = (float)f16_bits_to_f32(hb);
                        const size_t o = ((size_t)r * op.Kpad + (size_t)t * op.cin + b * 32) * 2 + c;
                        wh[o] = hb;
                        wh[o + 32] = f32_to_f16_bits((x - hi) * kSplitScale);
                    }
        if (dev_alloc(h, &op.w, wh.size() * 2)) return 1;
        HIP_TRY(hipMemcpy(op.w, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    } else if (h->is_f16) {
        std::vector<uint16_t> wh(wf.size());
        for (size_t i = 0; i < wf.size(); ++i) wh[i] = f32_to_f16_bits(wf[i]);
        if (dev_alloc(h, &op.w, wh.size() * 2)) return 1;
        HIP_TRY(hipMemcpy(op.w, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    } else {
        if (dev_alloc(h, &op.w, wf.size() * 4)) return 1;
        HIP_TRY(hipMemcpy(op.w, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

namespace {
struct Planner {
    wtk_yolo *h;
    const std::vector<ConvSpec> &specs;
    const wtk_conv_blob *blobs;
    bool failed = false;

    int new_buf(int hh, int ww, int C) {
        Buf b;
        b.h = hh;
        b.w = ww;
        b.C = C;
        b.elems_per_image = (size_t)hh * ww * C;
        h->bufs.push_back(b);
        return (int)h->bufs.size() - 1;
    }
    // generic conv op from one or more blobs (concatenated along cout)
    void conv(const std::vector<std::string> &names, int in_buf, int in_coff, int out_buf, int out_coff, int out2_buf = -1,
              int out2_coff = 0, int res_buf = -1, int res_coff = 0, int cout_store_pad = 0, int in2_buf = -1, int in2_coff = 0,
              int in2_split = 0) {
        if (failed) return;
        Op op;
        op.kind = OP_CONV;
        std::vector<const float *> wp, bp;
        std::vector<int> couts;
        int cout = 0;
        for (auto &nm : names) {
            const int i = find_spec(specs, nm);
            if (i < 0) {
                failed = true;
                fail("internal: unknown conv " + nm);
                return;
            }
            const ConvSpec &s = specs[i];
            if (op.spec < 0) op.spec = i;
            op.cin = s.cin;
            op.k = s.k;
            op.stride = s.stride;
            op.act = s.act;
            wp.push_back(blobs[i].weight);
            bp.push_back(blobs[i].bias);
            couts.push_back(s.cout);
            cout += s.cout;
        }
        op.cout = std::max(cout, cout_store_pad); // channels actually stored (>= real cout, zero rows beyond)
        op.cfg = pick_cfg(op.cout, op.k == 1 && op.stride == 1);
        if (h->split && op.cfg == CFG_256x64) op.cfg = CFG_128x64; // the split 256x64 instantiation spills (two accumulator sets)
        op.in_buf = in_buf;
        op.in_coff = in_coff;
        op.out_buf = out_buf;
        op.out_coff = out_coff;
        op.out2_buf = out2_buf;
        op.out2_coff = out2_coff;
        op.res_buf = res_buf;
        op.res_coff = res_coff;
        op.in2_buf = in2_buf;
        op.in2_coff = in2_coff;
        op.in2_split = in2_split;
        if (in2_buf >= 0) {
            const Buf &lb = h->bufs[in2_buf];
            const Buf &hb = h->bufs[in_buf];
            if (op.cfg != CFG_128x128 || op.k != 1 || lb.h * 2 != hb.h || lb.w * 2 != hb.w || in2_coff + in2_split > lb.C || in2_split > op.cin) {
                failed = true;
                fail("internal: two-source conv " + names[0] + " does not fit the 128x128 loader");
                return;
            }
        }
        const Buf &ib = h->bufs[in_buf];
        const Buf &ob = h->bufs[out_buf];
        const int pad = op.k / 2;
        const int ho = (ib.h + 2 * pad - op.k) / op.stride + 1, wo = (ib.w + 2 * pad - op.k) / op.stride + 1;
        if (ho != ob.h || wo != ob.w || in_coff + op.cin > ib.C || out_coff + op.cout > ob.C) {
            failed = true;
            fail("internal: shape mismatch planning conv " + names[0]);
            return;
        }
        // 2-D pixel tiles where the map is large enough that a linear tile would be a thin strip
        const int bm = conv_cfg_bm(op.cfg);
        op.tile_w = 0;
        if (op.k == 3 && wo >= 64 && wo % 16 == 0 && ho % (bm / 16) == 0) op.tile_w = 16;
        op.halo = halo_eligible(op.k, op.stride, op.cin, h->is_f16) && h->use_halo ? 1 : 0;
        if (h->split) op.halo = split_halo_eligible(op.k, op.stride, op.cin, op.cout) && h->use_halo ? 1 : 0;
        if (!h->split && h->use_halo && c32_eligible(op.k, op.stride, op.cin, op.cout, h->is_f16, out2_buf >= 0)) op.halo = 2;
        if (h->split && h->use_halo && h->use_c32s && c32_split_eligible(op.k, op.stride, op.cin, op.cout, out2_buf < 0 && in2_buf < 0 && !ob.f32)) op.halo = 2;
        op.macs_per_image = (double)ho * wo * cout * op.k * op.k * op.cin;
        if (pack_conv(h, op, wp, bp, couts)) {
            failed = true;
            return;
        }
        h->ops.push_back(op);
    }
    // C2f block: input view -> output view.  Returns nothing; allocates its concat + temp buffers.
    void c2f(const std::string &p, int in_buf, int in_coff, int c2, int n, bool shortcut, int out_buf, int out_coff, int out2_buf = -1,
             int out2_coff = 0, int in2_buf = -1, int in2_coff = 0, int in2_split = 0) {
        if (failed) return;
        const Buf ib = h->bufs[in_buf];
        const int c = c2 / 2;
        const int cat = new_buf(ib.h, ib.w, (2 + n) * c);
        const int tmp = new_buf(ib.h, ib.w, c);
        conv({p + ".cv1"}, in_buf, in_coff, cat, 0, -1, 0, -1, 0, 0, in2_buf, in2_coff, in2_split);
        for (int i = 0; i < n; ++i) {
            const std::string m = p + ".m." + std::to_string(i);
            conv({m + ".cv1"}, cat, (1 + i) * c, tmp, 0);
            conv({m + ".cv2"}, tmp, 0, cat, (2 + i) * c, -1, 0, shortcut ? cat : -1, (1 + i) * c);
        }
        conv({p + ".cv2"}, cat, 0, out_buf, out_coff, out2_buf, out2_coff);
    }
};
} // namespace

// WTK_SEGV_BACKTRACE=1 (tests/conftest.py sets it): a SIGSEGV prints the native frames to stderr before the handler that was installed before this
// library was loaded runs (under pytest: Python's faulthandler, which adds the Python stack and re-raises).  Diagnostic only; off by default.
namespace {
struct sigaction g_old_segv;
void segv_backtrace(int sig, siginfo_t *info, void *ctx) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char msg[] = "\nwtracker_amd: SIGSEGV, native frames:\n";
    if (write(2, msg, sizeof(msg) - 1) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    if ((g_old_segv.sa_flags & SA_SIGINFO) && g_old_segv.sa_sigaction)
        g_old_segv.sa_sigaction(sig, info, ctx);
    else if (!(g_old_segv.sa_flags & SA_SIGINFO) && g_old_segv.sa_handler != SIG_DFL && g_old_segv.sa_handler != SIG_IGN)
        g_old_segv.sa_handler(sig);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) void install_segv_backtrace() {
    const char *e = std::getenv("WTK_SEGV_BACKTRACE");
    if (!e || e[0] != '1') return;
    void *warm[4];
    (void)backtrace(warm, 4); // the first call loads libgcc and allocates: done here, not inside the handler
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = segv_backtrace;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER; // (no SA_ONSTACK: nobody installs an alternate stack here)
    sigemptyset(&sa.sa_mask);
    (void)sigaction(SIGSEGV, &sa, &g_old_segv);
}
} // namespace

// Status words live in ONE pinned, device-mapped page per process, handed out by slot and never freed (a handle is a few hundred allocations already; pinning
// and unpinning host memory per handle — hundreds of times in a test run — is a driver operation that has no business on that path).
namespace {
std::mutex g_status_mu;
int *g_status_page = nullptr, *g_status_page_dev = nullptr;
std::vector<int> g_status_free;
constexpr int kStatusSlots = 4096;
int acquire_status_word(int **host, int **dev) {
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (!g_status_page) {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, kStatusSlots * sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return 1;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            (void)hipHostFree(hp);
            return 1;
        }
        g_status_page = reinterpret_cast<int *>(hp), g_status_page_dev = reinterpret_cast<int *>(dp);
        for (int i = kStatusSlots - 1; i >= 0; --i) g_status_free.push_back(i);
    }
    if (g_status_free.empty()) return 1;
    const int slot = g_status_free.back();
    g_status_free.pop_back();
    g_status_page[slot] = 0;
    *host = g_status_page + slot, *dev = g_status_page_dev + slot;
    return 0;
}
void release_status_word(int *host, int *) {
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (g_status_page && host >= g_status_page && host < g_status_page + kStatusSlots) g_status_free.push_back((int)(host - g_status_page));
}
} // namespace


// Launch schedule of a latency-plan handle.  The forward pass is a DAG: the P3 / P4 Detect towers hang off the PAN path, a tower's box and class branches
// off its first conv.  Round 5 spread it over three streams (61 dispatches for one frame, ~45 on the critical path, the rest beside it on streams the
// capture forks into).  Here every op gets its dependency LEVEL — one more than the deepest earlier op it conflicts with (read-after-write,
// write-after-write, write-after-read, at whole-buffer granularity: conservative) — and the split-K convs of one level become ONE launch
// (conv_sk.hip: launch_conv_sk_group): 48 dispatches for YOLOv8s, all on the caller's stream, nothing to fork, nothing to join.
static void sk_schedule(wtk_yolo *h) {
    h->lat_sched.clear();
    const size_t n = h->ops.size();
    if (!h->latency || n <= 3) return;
    auto reads = [&](const Op &o, std::vector<int> &r) {
        r.clear();
        for (int b : {o.in_buf, o.res_buf, o.in2_buf})
            if (b >= 0) r.push_back(b);
    };
    auto writes = [&](const Op &o, std::vector<int> &w) {
        w.clear();
        if (o.kind == OP_POOL) w.push_back(o.in_buf); // the pool reads and writes slices of the SPPF concat buffer
        for (int b : {o.out_buf, o.out2_buf})
            if (b >= 0) w.push_back(b);
        if (o.tail_op >= 0 && h->ops[o.tail_op].out_buf >= 0) w.push_back(h->ops[o.tail_op].out_buf);
    };
    auto meets = [](const std::vector<int> &a, const std::vector<int> &b) {
        for (int x : a)
            for (int y : b)
                if (x == y) return true;
        return false;
    };
    std::vector<int> level(n, 0), ri, wi, rj, wj;
    int deepest = 0;
    for (size_t i = 0; i < n; ++i) {
        if (h->ops[i].folded) continue;
        reads(h->ops[i], ri), writes(h->ops[i], wi);
        for (size_t j = 0; j < i; ++j) {
            if (h->ops[j].folded) continue;
            reads(h->ops[j], rj), writes(h->ops[j], wj);
            if (meets(wj, ri) || meets(wj, wi) || meets(rj, wi)) level[i] = std::max(level[i], level[j] + 1);
        }
        deepest = std::max(deepest, level[i]);
    }
    for (int lv = 0; lv <= deepest; ++lv) {
        std::vector<int> group;
        for (size_t i = 3; i < n; ++i) { // ops[0 .. 2] are the (fused) front's: launched first, as before
            const Op &o = h->ops[i];
            if (o.folded || level[i] != lv) continue;
            if (o.kind == OP_CONV && o.sk) {
                group.push_back((int)i);
                if ((int)group.size() == kSkGroupMax) h->lat_sched.push_back(group), group.clear();
            } else {
                h->lat_sched.push_back({(int)i});
            }
        }
        if (!group.empty()) h->lat_sched.push_back(group);
    }
}

// Order of release (round 6: one protocol, checked by tests/hostsan): (1) the device drains — no kernel, copy or graph replay of this handle is in
// flight; (2) the graph execs go, BEFORE the events and streams they were captured through; (3) the events; (4) the streams go back to the pool,
// idle and outside any capture; (5) device memory.
extern "C" void wtk_yolo_destroy(wtk_yolo *h) {
    if (!h) return;
    DeviceGuard guard(h->device); // the synchronise and the releases below are about the HANDLE's device, whatever the caller's current device is
    (void)hipDeviceSynchronize();
    drop_graphs(h);
    for (int i = 0; i < h->ev_created; ++i) (void)hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->tune_ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i)
        if (h->feat_ev[i]) (void)hipEventDestroy(h->feat_ev[i]);
    for (int i = 0; i < wtk_yolo::kSideStreams; ++i) {
        if (h->side_done[i]) (void)hipEventDestroy(h->side_done[i]);
    }
    if (h->host_stream) unpool_stream(h->device, h->host_stream);
    dev_release(h);
    (void)hipFree(h->frames_dev);
    (void)hipFree(h->lb_dev);
    (void)hipFree(h->nms_score);
    (void)hipFree(h->nms_box);
    (void)hipFree(h->nms_cls);
    if (h->status_host) release_status_word(h->status_host, h->status_dev);
    delete h;
}

extern "C" int wtk_yolo_create(wtk_yolo **out, const wtk_yolo_desc *d) { return wtk_yolo_create_planned(out, d, WTK_PLAN_AUTO); }
extern "C" int wtk_yolo_plan(wtk_yolo *h) { return h ? (h->latency ? WTK_PLAN_LATENCY : WTK_PLAN_THROUGHPUT) : -1; }

extern "C" int wtk_yolo_create_planned(wtk_yolo **out, const wtk_yolo_desc *d, int32_t plan) {
    if (!out || !d || !d->convs) return fail("wtk_yolo_create: null argument");
    if (plan != WTK_PLAN_AUTO && plan != WTK_PLAN_THROUGHPUT && plan != WTK_PLAN_LATENCY) return fail("wtk_yolo_create_planned: plan must be WTK_PLAN_AUTO, _THROUGHPUT or _LATENCY");
    if (plan == WTK_PLAN_LATENCY && d->dtype == WTK_F16) return fail("wtk_yolo_create_planned: the latency plan is built for WTK_F32 and WTK_F16X3 handles");
    if (d->dtype != WTK_F32 && d->dtype != WTK_F16 && d->dtype != WTK_F16X3) return fail("wtk_yolo_create: dtype must be WTK_F32, WTK_F16 or WTK_F16X3");
    if (d->imgsz_h <= 0 || d->imgsz_w <= 0 || d->imgsz_h % 32 || d->imgsz_w % 32) return fail("wtk_yolo_create: imgsz must be a positive multiple of 32");
    if (d->max_batch <= 0) return fail("wtk_yolo_create: max_batch must be positive");
    // nc <= 32: the class towers' last 1x1 runs inside the 3x3 before it (32 stored couts); 33..80 (a stock 80-class YOLOv8 head): the same conv as a
    // launch of its own (implicit GEMM over cls_ld = nc rounded up to 8 couts).  The reference trains single_cls (yolo/yolo_train_config.yaml:27).
    if (d->nc < 1 || d->nc > 80) return fail("wtk_yolo_create: nc must be in [1, 80]");
    if (wtk_device_count() <= d->device) return fail("wtk_yolo_create: no such HIP device (is a GPU visible?)");
    const ModelDims dims = model_dims(d->width_mult, d->depth_mult, d->max_channels, d->nc);
    const std::vector<ConvSpec> specs = conv_specs(dims);
    if ((int)specs.size() != d->n_convs) return fail("wtk_yolo_create: n_convs does not match the model scale");
    for (size_t i = 0; i < specs.size(); ++i) {
        const wtk_conv_blob &b = d->convs[i];
        const ConvSpec &s = specs[i];
        if (b.cout != s.cout || b.cin != s.cin || b.k != s.k || b.stride != s.stride || b.act != s.act || !b.weight || !b.bias)
            return fail("wtk_yolo_create: conv blob " + std::to_string(i) + " (" + s.name + ") does not match the expected shape");
    }
    for (int i = 0; i < 5; ++i)
        if (dims.c[i] % 16 != 0) return fail("wtk_yolo_create: channel widths must be multiples of 16 for this build");
    if (dims.hb % 16 || dims.hc % 16) return fail("wtk_yolo_create: head widths must be multiples of 16");
    if (d->dtype == WTK_F16X3) {
        for (int i = 0; i < 5; ++i)
            if (dims.c[i] % 64 != 0 && !(i == 0 && dims.c[0] == 32)) return fail("wtk_yolo_create: WTK_F16X3 needs channel widths in multiples of 64 (stem: 32)");
        if (dims.hb % 32 || dims.hc % 32) return fail("wtk_yolo_create: WTK_F16X3 needs head widths in multiples of 32");
    }
    DEVICE_GUARD(d);
    if (ensure_attributes(d->device)) return 1;

    wtk_yolo *h = new wtk_yolo();
    h->device = d->device;
    h->is_f16 = d->dtype == WTK_F16;
    h->split = d->dtype == WTK_F16X3;
    h->esize = h->is_f16 ? 2 : 4;
    h->S_h = d->imgsz_h;
    h->S_w = d->imgsz_w;
    h->max_batch = d->max_batch;
    h->dims = dims;
    if (const char *e = std::getenv("WTK_NO_HALO")) h->use_halo = !(e[0] == '1');
    // Side streams are for LARGE batches.  A handle for the reference's own calls (max_batch <= 16: one frame, one cycle batch) runs on the caller's stream
    // alone: its launches last 5-50 us, and a dependency between two streams costs microseconds when the runtime has put them on different hardware
    // queues, nothing when they share one — so with side streams the same controller loop ran at 7.6 k or 4.4 k frames/s (throughput plan), 9.0 k or
    // 11.2 k (deferred log) depending on GPU_MAX_HW_QUEUES and on which streams the process had created before; on one stream it runs at the same rate
    // in every such environment (profiles/r06_notes.md section 4).  wtk_yolo_set_side_streams(h, 2) turns them on for such a handle explicitly.
    if (d->max_batch <= 16) h->use_side = 0, h->side_streams = 0;
    if (const char *e = std::getenv("WTK_NO_SIDE_STREAM")) h->use_side = h->use_side && !(e[0] == '1');
    if (const char *e = std::getenv("WTK_HALO_SLABS")) h->halo_slabs = e[0] == '2' ? 2 : 3;
    if (const char *e = std::getenv("WTK_HALO_PERSIST")) h->halo_persist = e[0] != '0';
    if (const char *e = std::getenv("WTK_HALO_SMALL_BLOCKS")) h->halo_small_blocks = e[0] != '0';
    if (const char *e = std::getenv("WTK_NO_FUSED_TAIL")) h->use_tail = e[0] != '1';
    if (const char *e = std::getenv("WTK_NO_SPLIT_CLS_TAIL")) h->use_tail_cls_split = e[0] != '1';
    if (const char *e = std::getenv("WTK_NO_WIDE_1X1")) h->use_wide = e[0] != '1';
    if (const char *e = std::getenv("WTK_NO_WS64")) h->use_ws64 = e[0] != '1';
    if (const char *e = std::getenv("WTK_NO_S2WIN")) h->use_s2win = e[0] != '1';
    if (const char *e = std::getenv("WTK_NO_C32S")) h->use_c32s = e[0] != '1';
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, d->device));
        h->num_cus = prop.multiProcessorCount;
    }
    h->latency = d->max_batch <= 4 && !h->is_f16;
    if (const char *e = std::getenv("WTK_LATENCY_PLAN")) h->latency = e[0] == '1' && !h->is_f16;
    if (plan != WTK_PLAN_AUTO) h->latency = plan == WTK_PLAN_LATENCY; // the caller's word beats the rule and the environment
    // Replayed captures (hipGraph) are OPT-IN since round 6: WTK_GRAPH=1 (every form), or WTK_GRAPH_HOST=1 (the *_host entry points) / WTK_GRAPH_VIEWS=1
    // (caller buffers, captured the second time an argument set is met), read when the handle is created.  Round 5 replayed by default on latency-plan
    // handles (0.53 against 0.57 ms per single-frame call); a capture of this forward pass FORKS into the side streams, a graph exec instantiated from
    // a forked capture runs its branches on streams the runtime creates for it, and that machinery is where the two open problems of round 5 lived (an
    // intermittent host fault in the first capturing call of a handle after > 100 handles in the process, and replays that ran 2-4 x slower for later
    // handles of a process with eight hardware queues): profiles/r06_notes.md section 1.  The latency plan of round 6 runs on ONE stream in launches
    // grouped per dependency level (sk_schedule), so eager launches no longer pay for the fork either.
    {
        const char *g = std::getenv("WTK_GRAPH");
        h->graph_host = h->graph_views = g && g[0] == '1';
    }
    if (h->latency) h->use_tail = 0;
    if (const char *e = std::getenv("WTK_GRAPH_MAX_BATCH")) h->graph_max_batch = std::atoi(e);
    if (const char *e = std::getenv("WTK_GRAPH_HOST")) h->graph_host = e[0] == '1';
    if (const char *e = std::getenv("WTK_GRAPH_VIEWS")) h->graph_views = e[0] == '1';

    Planner P{h, specs, d->convs};
    const int *c = dims.c;
    const int H = h->S_h, W = h->S_w;
    auto hw = [&](int s, int &hh, int &ww) { hh = H / s, ww = W / s; };
    int h2, w2, h4, w4, h8, w8, h16, w16, h32, w32;
    hw(2, h2, w2), hw(4, h4, w4), hw(8, h8, w8), hw(16, h16, w16), hw(32, h32, w32);

    // ---- buffers that hold more than one logical tensor (concat-free FPN/PAN)
    const int t0 = P.new_buf(h2, w2, c[0]);
    const int t1 = P.new_buf(h4, w4, c[1]);
    const int t2 = P.new_buf(h4, w4, c[1]);
    const int t3 = P.new_buf(h8, w8, c[2]);
    const int cat14 = P.new_buf(h8, w8, c[3] + c[2]);   // [up(t12) | t4]
    const int t5 = P.new_buf(h16, w16, c[3]);
    const int cat11 = P.new_buf(h16, w16, c[4] + c[3]); // [up(t9) | t6]
    const int t7 = P.new_buf(h32, w32, c[4]);
    const int t8 = P.new_buf(h32, w32, c[4]);
    const int sppcat = P.new_buf(h32, w32, 2 * c[4]);   // [x | y1 | y2 | y3], each c4/2
    const int cat20 = P.new_buf(h32, w32, c[3] + c[4]); // [t19 | t9]
    const int cat17 = P.new_buf(h16, w16, c[2] + c[3]); // [t16 | t12]
    const int t15 = P.new_buf(h8, w8, c[2]);
    const int t18 = P.new_buf(h16, w16, c[3]);
    const int t21 = P.new_buf(h32, w32, c[4]);

    // ---- backbone
    {
        Op op;
        op.kind = OP_STEM;
        op.out_buf = t0;
        op.cout = c[0];
        op.macs_per_image = (double)h2 * w2 * c[0] * 27;
        const int i0 = find_spec(specs, "model.0");
        op.spec = i0;
        // repack [cout][3][3][3(RGB)] -> K = tap*4 + channel (see stem_mfma_kernel)
        const float *w0 = d->convs[i0].weight;
        // split mode: split-fp16 operands like every other conv of the handle (pixels / 255 and the weights as hi + lo pairs)
        const bool stem_split = h->split;
        const int taps = (h->is_f16 || stem_split) ? 16 : 9;
        std::vector<float> wp((size_t)c[0] * taps * 4, 0.f);
        for (int co = 0; co < c[0]; ++co)
            for (int tap = 0; tap < 9; ++tap)
                for (int ch = 0; ch < 3; ++ch) // the stem reads unscaled pixels and produces scaled activations
                    wp[((size_t)co * taps + tap) * 4 + ch] = (float)((double)w0[((size_t)co * 9 + tap) * 3 + ch] * (double)kActScale);
        if (h->is_f16 || stem_split)
            for (float x : wp) {
                if (!(std::fabs(x) <= 65504.0f)) {
                    wtk_yolo_destroy(h);
                    return fail("wtk_yolo_create: a folded weight of conv blob " + std::to_string(i0) + " (model.0) is outside the fp16 range: this model needs dtype WTK_F32");
                }
            }
        std::vector<float> stem_bias(c[0]);
        for (int co = 0; co < c[0]; ++co) stem_bias[co] = (float)((double)d->convs[i0].bias[co] * (double)kActScale);
        void *wdev;
        float *bdev;
        std::vector<uint16_t> wh;
        const void *src = wp.data();
        size_t bytes = wp.size() * 4;
        if (h->is_f16) {
            wh.resize(wp.size());
            for (size_t i = 0; i < wp.size(); ++i) wh[i] = f32_to_f16_bits(wp[i]);
            src = wh.data();
            bytes = wh.size() * 2;
        } else if (stem_split) { // [cout][16][4] hi halves, then [cout][16][4] lo halves
            wh.resize(2 * wp.size());
            for (size_t i = 0; i < wp.size(); ++i) {
                const uint16_t hb = f32_to_f16_bits(wp[i]);
                wh[i] = hb;
                wh[wp.size() + i] = f32_to_f16_bits((wp[i] - f16_bits_to_f32(hb)) * kSplitScale);
            }
            src = wh.data();
            bytes = wh.size() * 2;
        }
        if (dev_alloc(h, &wdev, bytes) || dev_alloc(h, (void **)&bdev, sizeof(float) * c[0])) {
            wtk_yolo_destroy(h);
            return 1;
        }
        if (hipMemcpy(wdev, src, bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(bdev, stem_bias.data(), sizeof(float) * c[0], hipMemcpyHostToDevice) != hipSuccess) {
            wtk_yolo_destroy(h);
            return fail("wtk_yolo_create: stem weight upload failed");
        }
        op.w = wdev;
        op.bias = bdev;
        h->ops.push_back(op);
    }
    P.conv({"model.1"}, t0, 0, t1, 0);
    P.c2f("model.2", t1, 0, c[1], dims.n[0], true, t2, 0);
    P.conv({"model.3"}, t2, 0, t3, 0);
    P.c2f("model.4", t3, 0, c[2], dims.n[1], true, cat14, c[3]);
    P.conv({"model.5"}, cat14, c[3], t5, 0);
    P.c2f("model.6", t5, 0, c[3], dims.n[2], true, cat11, c[4]);
    P.conv({"model.7"}, cat11, c[4], t7, 0);
    P.c2f("model.8", t7, 0, c[4], dims.n[3], true, t8, 0);
    // SPPF
    P.conv({"model.9.cv1"}, t8, 0, sppcat, 0);
    if (!P.failed) {
        Op op;
        op.kind = OP_POOL;
        op.in_buf = sppcat;
        op.cin = c[4] / 2;
        h->ops.push_back(op);
    }
    // nn.Upsample(2x nearest) + Concat: the consumer's 1x1 conv reads the half-resolution producer directly (two-source
    // loader of the 128x128 tile), so the 4x larger upsampled copy is never written.  Narrow scales whose cv1 does not
    // use that tile (and WTK_MATERIALIZE_UPSAMPLE=1) keep the materialised copy in the concat buffer.
    bool lazy_up = c[2] % 128 == 0 && c[3] % 128 == 0;
    if (const char *e = std::getenv("WTK_MATERIALIZE_UPSAMPLE")) lazy_up = lazy_up && e[0] != '1';
    if (lazy_up) {
        P.conv({"model.9.cv2"}, sppcat, 0, cat20, c[3]); // t9 -> cat20 slice
        P.c2f("model.12", cat11, 0, c[3], dims.n[3], false, cat17, c[2], -1, 0, cat20, c[3], c[4]); // [up(t9) | t6]; t12 -> cat17 slice
        P.c2f("model.15", cat14, 0, c[2], dims.n[3], false, t15, 0, -1, 0, cat17, c[2], c[3]);      // [up(t12) | t4]
    } else {
        P.conv({"model.9.cv2"}, sppcat, 0, cat20, c[3], cat11, 0); // t9 -> cat20 slice, upsampled copy -> cat11
        P.c2f("model.12", cat11, 0, c[3], dims.n[3], false, cat17, c[2], cat14, 0); // t12 -> cat17 slice, up -> cat14
        P.c2f("model.15", cat14, 0, c[2], dims.n[3], false, t15, 0);
    }
    if (!P.failed) h->ops.back().signal_feat = 0; // P3 feature map complete
    P.conv({"model.16"}, t15, 0, cat17, 0);
    P.c2f("model.18", cat17, 0, c[3], dims.n[3], false, t18, 0);
    if (!P.failed) h->ops.back().signal_feat = 1; // P4 feature map complete
    P.conv({"model.19"}, t18, 0, cat20, 0);
    P.c2f("model.21", cat20, 0, c[4], dims.n[3], false, t21, 0);
    // ---- Detect: both towers' first 3x3 share one conv (weights concatenated along cout)
    const int feat[3] = {t15, t18, t21};
    const int fh[3] = {h8, h16, h32}, fw[3] = {w8, w16, w32};
    h->cls_ld = (d->nc + 7) / 8 * 8; // class logits are stored in 16-byte groups: nc = 1 costs 16 B per anchor, not 64
    for (int i = 0; i < 3 && !P.failed; ++i) {
        const std::string b = "model.22.cv2." + std::to_string(i), cl = "model.22.cv3." + std::to_string(i);
        const int d1 = P.new_buf(fh[i], fw[i], dims.hb + dims.hc);
        const int d2b = P.new_buf(fh[i], fw[i], dims.hb);
        const int d2c = P.new_buf(fh[i], fw[i], dims.hc);
        h->box_buf[i] = P.new_buf(fh[i], fw[i], 64);
        h->cls_buf[i] = P.new_buf(fh[i], fw[i], h->cls_ld);
        h->bufs[h->box_buf[i]].f32 = h->bufs[h->cls_buf[i]].f32 = 1;
        h->lh[i] = fh[i];
        h->lw[i] = fw[i];
        const size_t first_op = h->ops.size();
        P.conv({b + ".0", cl + ".0"}, feat[i], 0, d1, 0);
        P.conv({b + ".1"}, d1, 0, d2b, 0);
        P.conv({cl + ".1"}, d1, dims.hb, d2c, 0);
        P.conv({b + ".2"}, d2b, 0, h->box_buf[i], 0);
        P.conv({cl + ".2"}, d2c, 0, h->cls_buf[i], 0, -1, 0, -1, 0, h->cls_ld);
        if (!P.failed && h->use_tail) { // box tower: the last 1x1 runs in the epilogue of the 3x3 before it (fp16, 64 channels)
            Op &b1 = h->ops[first_op + 1], &b2 = h->ops[first_op + 3];
            if ((h->is_f16 || h->split) && b1.halo == 1 && b1.cout == 64 && b1.cout_pad == 64 && b2.k == 1 && b2.cin == 64 && b2.cout == 64 && !b2.act &&
                b2.in_buf == b1.out_buf && b2.res_buf < 0 && b2.out2_buf < 0 && b1.res_buf < 0 && b1.out2_buf < 0 && h->halo_slabs == 3) {
                b1.tail_op = (int)first_op + 3;
                b2.folded = 1;
            }
            // class tower: 3x3 (128 -> 128) then 1x1 (128 -> nc, stored as cls_ld = 8, 16, 24 or 32 channels)
            Op &c1 = h->ops[first_op + 2], &c2 = h->ops[first_op + 4];
            if ((h->is_f16 || (h->split && h->use_tail_cls_split)) && c1.halo == 1 && c1.cout == 128 && c1.cout_pad == 128 && c2.k == 1 && c2.cin == 128 && c2.cout <= 32 && c2.cout_pad == 32 &&
                !c2.act && c2.in_buf == c1.out_buf && c2.res_buf < 0 && c2.out2_buf < 0 && c1.res_buf < 0 && c1.out2_buf < 0 && h->halo_slabs == 3 &&
                c2.cout == h->cls_ld) {
                c1.tail_op = (int)first_op + 4;
                c2.folded = 1;
            }
        }
        if (!P.failed && i < 2) { // P3 and P4 towers only need t15 / t18: independent of the rest of the PAN path
            for (size_t k = first_op; k < h->ops.size(); ++k) h->ops[k].side = i == 1 ? 2 : 1; // P4 tower: side stream 2 (folded onto stream 1 at launch time when the handle runs with one side stream)
            h->ops[first_op].wait_feat = i;
        }
    }
    // A strided 3x3 conv (128 couts, implicit GEMM, fp16) whose ONLY reader is the 1x1 conv 128 -> 128 right behind it (model.3 ->
    // model.4.cv1 in YOLOv8s): the 1x1 runs in the 3x3's epilogue, its input never reaches HBM.  WTK_NO_IGEMM_TAIL=1 switches it off.
    if (!P.failed && h->is_f16 && !(std::getenv("WTK_NO_IGEMM_TAIL") && std::getenv("WTK_NO_IGEMM_TAIL")[0] == '1')) {
        for (size_t i = 0; i + 1 < h->ops.size(); ++i) {
            Op &c3 = h->ops[i], &c1 = h->ops[i + 1];
            if (c3.kind != OP_CONV || c1.kind != OP_CONV || c3.halo || c3.k != 3 || c3.stride != 2 || c3.cfg != CFG_128x128 || c3.cout != 128 || c3.cout_pad != 128 ||
                !c3.act || c3.res_buf >= 0 || c3.out2_buf >= 0 || c3.in2_buf >= 0 || c3.tail_op >= 0 || c3.folded)
                continue;
            if (c1.k != 1 || c1.stride != 1 || c1.cin != 128 || c1.cout != 128 || c1.cout_pad != 128 || c1.in_buf != c3.out_buf || c1.in_coff != c3.out_coff ||
                c1.res_buf >= 0 || c1.out2_buf >= 0 || c1.in2_buf >= 0 || c1.folded || c1.tail_op >= 0 || h->bufs[c3.out_buf].C != 128)
                continue;
            bool other_reader = false;
            for (size_t j = 0; j < h->ops.size(); ++j) {
                const Op &o = h->ops[j];
                if (j != i + 1 && (o.in_buf == c3.out_buf || o.res_buf == c3.out_buf || o.in2_buf == c3.out_buf)) other_reader = true;
            }
            if (other_reader) continue;
            c3.tail_op = (int)i + 1;
            c1.folded = 1;
        }
    }
    if (P.failed) {
        wtk_yolo_destroy(h);
        return 1;
    }
    h->anchors = h8 * w8 + h16 * w16 + h32 * w32;
    for (const Op &op : h->ops) h->macs_per_frame += op.macs_per_image;
    // Which convs the split-K kernel (conv_sk.hip) takes, and their slab scratch.  Latency plan: everything with rows of 32 input channels.
    // Throughput plan of a SMALL handle (max_batch <= 16, fp32 / f16x3: what a controller's cycle batch of 9 / 15 frames runs on, yolo_controller.py:108-109):
    // the layers whose whole batch is at most 4 096 output pixels — the 12 x 12 maps of imgsz 384 — where the window / implicit-GEMM kernels run ~40-block
    // grids that walk K serially (model.8's bottlenecks 40 us, split over K 24 us: profiles/r05_notes.md section 5); the choice is fixed per handle, so a
    // frame's result still does not depend on its batch.  WTK_NO_SK_MIXED=1 switches the second rule off (A/B).
    const bool sk_mixed = !h->latency && !h->is_f16 && h->max_batch <= 16 && !(std::getenv("WTK_NO_SK_MIXED") && std::getenv("WTK_NO_SK_MIXED")[0] == '1');
    // (f16x3: the 12 x 12 maps of imgsz 384; fp32, whose window kernels are 2.5 x slower per tap, gains on the 24 x 24 maps too — profiles/r05_notes.md section 5)
    const long long sk_mixed_max_px = std::getenv("WTK_SK_MIXED_MAX_PX") ? std::atoll(std::getenv("WTK_SK_MIXED_MAX_PX")) : (h->split ? 4096 : 10000);
    h->small_narrow = !h->is_f16 && h->max_batch <= 16 && !(std::getenv("WTK_SMALL_NARROW") && std::getenv("WTK_SMALL_NARROW")[0] == '0');
    {
        const int deep = std::getenv("WTK_HALO_DEEP") ? std::atoi(std::getenv("WTK_HALO_DEEP")) : 1;
        h->halo_deep = h->split && (deep == 2 || (deep == 1 && h->max_batch <= 16));
    }
    if (h->latency || sk_mixed) {
        for (size_t i = 3; i < h->ops.size(); ++i) { // ops[0..2] stay the fused front's
            Op &op = h->ops[i];
            if (op.kind != OP_CONV || op.folded || op.tail_op >= 0 || op.out2_buf >= 0 || op.cin % 32 || (op.k != 1 && op.k != 3) || op.cout_pad % 32 || op.cout % 8) continue;
            if (op.in2_buf >= 0 && (op.k != 1 || op.in2_split % 32)) continue;
            {
                const Buf &ibx = h->bufs[op.in_buf]; // conv_sk_kernel addresses a tile's pixels by 32-bit lane offsets from its first image: two images inside 31 bits
                if (2LL * ibx.h * ibx.w * ibx.C * 4 > 0x7fffffffLL) continue;
            }
            if (sk_mixed && (long long)h->max_batch * h->bufs[op.out_buf].h * h->bufs[op.out_buf].w > sk_mixed_max_px) continue;
            op.sk = 1;
            const Buf &ob = h->bufs[op.out_buf];
            // K atoms: the count the launcher's cost model likes best for what this handle is for — a small throughput-plan handle's largest call (a cycle
            // batch's 12 x 12 maps: eight atoms x 34 tiles are 272 blocks = two rounds on 256 CUs, seven are one round), a latency-plan handle's single frame
            // (0.529 -> 0.515 ms at 384 x 384) — and the layer's default where the model sees no difference.  Fixed per handle, so a frame's result does not
            // depend on its batch.  WTK_SK_PLAN_ATOMS=0: the default count everywhere.
            const int nk_op = op.k * op.k * op.cin / 32;
            const bool plan_atoms = !(std::getenv("WTK_SK_PLAN_ATOMS") && std::getenv("WTK_SK_PLAN_ATOMS")[0] == '0');
            const long long plan_px = (long long)(h->latency ? 1 : h->max_batch) * ob.h * ob.w;
            op.sk_atoms = plan_atoms ? conv_sk_plan_atoms(plan_px, op.cout_pad, nk_op, h->num_cus, h->split) : conv_sk_slices(nk_op);
            const int S = op.sk_atoms;
            if (S > 1 && dev_alloc(h, (void **)&op.sk_partial, (size_t)S * h->max_batch * ob.h * ob.w * op.cout_pad * sizeof(float))) {
                wtk_yolo_destroy(h);
                return 1;
            }
            const bool two_launches = std::getenv("WTK_SK_FINISH") && std::getenv("WTK_SK_FINISH")[0] == '1'; // A/B switch: slabs combined by a second launch
            if (S > 1 && !two_launches) {
                const size_t nt = conv_sk_ticket_count((long long)h->max_batch * ob.h * ob.w, op.cout_pad) * sizeof(unsigned);
                if (dev_alloc(h, (void **)&op.sk_tickets, nt)) {
                    wtk_yolo_destroy(h);
                    return 1;
                }
                if (hipMemset(op.sk_tickets, 0, nt) != hipSuccess) {
                    wtk_yolo_destroy(h);
                    return fail("wtk_yolo_create: hipMemset failed");
                }
            }
        }
    }
    if (const char *e = std::getenv("WTK_SK_GROUP")) h->sk_group = e[0] != '0';
    if (const char *e = std::getenv("WTK_SK_AUTOTUNE")) h->sk_autotune = e[0] != '0';
    if (const char *e = std::getenv("WTK_SK_TILE")) h->sk_force_tile = std::atoi(e) >= 0 && std::atoi(e) <= 3 ? std::atoi(e) : -1;
    if (const char *e = std::getenv("WTK_SK_FORM")) h->sk_force_form = std::atoi(e) == 0 || std::atoi(e) == 1 ? std::atoi(e) : -1;
    sk_schedule(h);
    // ops[0..2] are stem, model.1, model.2.cv1 by construction; fuse them when the widths match the kernel
    {
        const char *e = std::getenv("WTK_NO_FUSED_FRONT");
        const bool off = e && e[0] == '1';
        if (const char *dbg = std::getenv("WTK_FRONT_DEBUG")) h->front_debug = dbg[0] == '1';
        h->use_front = !off && h->ops.size() > 3 && h->ops[0].kind == OP_STEM && h->ops[1].kind == OP_CONV && h->ops[2].kind == OP_CONV &&
                       h->ops[1].k == 3 && h->ops[1].stride == 2 && h->ops[2].k == 1 && h->ops[1].act && h->ops[2].act &&
                       h->ops[2].out2_buf < 0 && h->ops[2].res_buf < 0 &&
                       (front_fused_eligible(h->is_f16, h->ops[0].cout, h->ops[1].cout, h->ops[2].cout) ||
                        (h->split && h->ops[1].cin == 32 && front_fused_split_eligible(h->ops[0].cout, h->ops[1].cout, h->ops[2].cout)));
        // ops[3..5] are the first C2f's bottleneck convs and cv2 (dims.n[0] == 1)
        const char *e2 = std::getenv("WTK_NO_FUSED_C2F");
        const bool off2 = e2 && e2[0] == '1';
        if (!off2 && h->ops.size() > 6 && dims.n[0] == 1) {
            const Op &m1 = h->ops[3], &m2 = h->ops[4], &cv2 = h->ops[5], &cv1 = h->ops[2];
            h->use_c2f = m1.kind == OP_CONV && m2.kind == OP_CONV && cv2.kind == OP_CONV && m1.k == 3 && m2.k == 3 && cv2.k == 1 &&
                         m1.stride == 1 && m2.stride == 1 && m1.act && m2.act && cv2.act && m1.in_buf == cv1.out_buf &&
                         m2.res_buf == cv1.out_buf && m2.res_coff == m1.in_coff && cv2.in_buf == cv1.out_buf && cv2.in_coff == cv1.out_coff &&
                         m1.in_coff == cv1.out_coff + 32 && m2.out_coff == cv1.out_coff + 64 && cv2.cin == 96 && m1.Kpad == m2.Kpad &&
                         cv2.out2_buf < 0 && cv2.res_buf < 0 && m1.cout == 32 && m2.cout == 32 &&
                         c2f_fused_eligible(h->is_f16, m1.cin, dims.n[0], m2.res_buf >= 0, cv2.cout);
        }
    }

    // ---- activation workspace: every tensor gets its own allocation (288 GB HBM: no liveness reuse needed)
    for (Buf &b : h->bufs) {
        if (dev_alloc(h, &b.ptr, b.elems_per_image * (size_t)h->max_batch * (b.f32 ? 4 : h->esize))) {
            wtk_yolo_destroy(h);
            return 1;
        }
    }
    if (dev_alloc(h, (void **)&h->o_xywh, sizeof(float) * 4 * h->max_batch) || dev_alloc(h, (void **)&h->o_conf, sizeof(float) * h->max_batch) ||
        dev_alloc(h, (void **)&h->o_anchor, sizeof(int) * h->max_batch) || dev_alloc(h, (void **)&h->o_margin, sizeof(float) * h->max_batch)) {
        wtk_yolo_destroy(h);
        return 1;
    }
    if (acquire_status_word(&h->status_host, &h->status_dev)) {
        h->status_host = nullptr;
        wtk_yolo_destroy(h);
        return fail("wtk_yolo_create: no pinned status word (hipHostMalloc failed, or more than 4096 live handles)");
    }
    if (dev_alloc(h, &h->zero_page, 256)) {
        wtk_yolo_destroy(h);
        return 1;
    }
    if (hipMemset(h->zero_page, 0, 256) != hipSuccess) {
        wtk_yolo_destroy(h);
        return fail("wtk_yolo_create: hipMemset failed");
    }
    // streams (side streams, the host entry points' stream) are taken from the process pool at first use: a handle that never runs
    // with side streams (the hybrid's second look) or never sees a host call does not occupy a hardware queue slot
    *out = h;
    return 0;
}

extern "C" int wtk_yolo_status(wtk_yolo *h, int32_t *flags, int32_t clear) {
    if (!h || !flags) return fail("wtk_yolo_status: null argument");
    *flags = h->status_static | (h->status_host ? __atomic_load_n(h->status_host, __ATOMIC_RELAXED) : 0);
    if (clear && h->status_host) __atomic_store_n(h->status_host, 0, __ATOMIC_RELAXED);
    return 0;
}

extern "C" int wtk_yolo_workload(wtk_yolo *h, double *macs_per_frame, int32_t *anchors) {
    if (!h) return fail("wtk_yolo_workload: null handle");
    if (macs_per_frame) *macs_per_frame = h->macs_per_frame;
    if (anchors) *anchors = h->anchors;
    return 0;
}

extern "C" int wtk_yolo_set_profiling(wtk_yolo *h, int32_t enabled) {
    if (!h) return fail("wtk_yolo_set_profiling: null handle");
    if (enabled && !h->ev_created) {
        for (int i = 0; i < wtk_yolo::kProfEvents; ++i) {
            HIP_TRY(hipEventCreate(&h->ev[i]));
            h->ev_created = i + 1;
        }
    }
    h->profiling = enabled ? 1 : 0;
    for (int i = 0; i < wtk_yolo::kProfKernels; ++i) h->prof_ms[i] = 0, h->prof_flops[i] = 0, h->prof_launches[i] = 0;
    return 0;
}

extern "C" int wtk_yolo_get_kernel_profile(wtk_yolo *h, int32_t kernel_id, double *total_ms, int64_t *launches, double *flops) {
    if (!h || kernel_id < 0 || kernel_id >= wtk_yolo::kProfKernels) return fail("wtk_yolo_get_kernel_profile: bad argument");
    if (total_ms) *total_ms = h->prof_ms[kernel_id];
    if (launches) *launches = h->prof_launches[kernel_id];
    if (flops) *flops = h->prof_flops[kernel_id];
    return 0;
}

extern "C" int wtk_yolo_get_profile(wtk_yolo *h, int32_t kernel_class, double *total_ms, int64_t *launches) {
    if (!h || kernel_class < 0 || kernel_class > 3) return fail("wtk_yolo_get_profile: bad argument");
    double ms = h->prof_ms[kernel_class];
    long long n = h->prof_launches[kernel_class];
    if (kernel_class == 1)
        for (int k = 4; k < wtk_yolo::kProfKernels; ++k) ms += h->prof_ms[k], n += h->prof_launches[k];
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return 0;
}

