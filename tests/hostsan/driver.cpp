// TEST INFRASTRUCTURE (tests/hostsan): drives the C ABI of libwtk_hip.so — its HOST side, compiled with -fsanitize=address,undefined and linked against
// hip_stub.cpp instead of the HIP runtime — over the shape matrix of the GPU suite: model scales n / s / m, network sizes 96 x 160 ... 1280 x 1280,
// batches 1 ... 64, every dtype, both launch plans, nc 1 / 32 / 80, every entry point a controller or a test calls, handles created and destroyed out of
// phase.  No kernel runs; what is exercised is planning, weight packing, every launcher's geometry, the allocation sizes against the extents the kernels
// will address (launch_checks.inc), and the stream / event / graph-capture lifetime protocol (hip_stub.cpp).  Exit code 0 = no sanitizer report and no
// protocol violation.
#include "../../include/wtk_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

extern "C" {
int stub_violations(void);
size_t stub_peak_device_bytes(void);
size_t stub_live_device_bytes(void);
int stub_live_streams(void);
int stub_live_events(void);
int stub_live_execs(void);
int stub_open_captures(void);
unsigned long long stub_kernel_launches(const char *substr);
void stub_set_verbose(int);
// the slice of the HIP API the driver itself needs (device buffers of a caller)
int hipMalloc(void **, size_t);
int hipFree(void *);
int hipStreamCreateWithFlags(void **, unsigned);
int hipStreamBeginCapture(void *, int);
int hipStreamEndCapture(void *, void **);
int hipStreamWaitEvent(void *, void *, unsigned);
int hipStreamSynchronize(void *);
int hipEventCreateWithFlags(void **, unsigned);
int hipEventRecord(void *, void *);
int hipEventDestroy(void *);
int hipGraphInstantiate(void **, void *, void *, char *, size_t);
int hipGraphDestroy(void *);
int hipGraphExecDestroy(void *);
int hipGraphLaunch(void *, void *);
int hipMemcpy(void *, const void *, size_t, int);
int hipMemset(void *, int, size_t);
}

static int g_fail = 0;
#define CHECK(cond, ...)                                                                                                       \
    do {                                                                                                                       \
        if (!(cond)) {                                                                                                         \
            std::fprintf(stderr, "driver FAIL %s:%d: ", __FILE__, __LINE__);                                                   \
            std::fprintf(stderr, __VA_ARGS__);                                                                                 \
            std::fprintf(stderr, " [last error: %s]\n", wtk_last_error());                                                     \
            ++g_fail;                                                                                                          \
        }                                                                                                                      \
    } while (0)

struct Rng {
    unsigned long long s;
    explicit Rng(unsigned long long seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    unsigned next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (unsigned)(s >> 33);
    }
    float uni() { return (float)(next() & 0xffffff) / (float)0x1000000 - 0.5f; }
};

struct Scale {
    const char *name;
    float width, depth;
    int maxch;
};
static const Scale kScales[] = {{"n", 0.25f, 0.33f, 1024}, {"s", 0.50f, 0.33f, 1024}, {"m", 0.75f, 0.67f, 768}};

struct Model {
    Scale sc;
    int nc;
    std::vector<std::vector<float>> w, b;
    std::vector<wtk_conv_blob> blobs;
};
static Model make_model(const Scale &sc, int nc, unsigned seed) {
    Model m;
    m.sc = sc, m.nc = nc;
    const int n = wtk_yolo_conv_count(sc.width, sc.depth, sc.maxch, nc);
    CHECK(n > 0, "conv_count(%s)", sc.name);
    Rng r(seed);
    m.w.resize(n), m.b.resize(n), m.blobs.resize(n);
    for (int i = 0; i < n; ++i) {
        int32_t co, ci, k, s, a;
        char name[96];
        CHECK(wtk_yolo_conv_info(sc.width, sc.depth, sc.maxch, nc, i, &co, &ci, &k, &s, &a, name, sizeof(name)) == 0, "conv_info %d", i);
        m.w[i].resize((size_t)co * k * k * ci);
        m.b[i].resize(co);
        const float g = 1.0f / std::sqrt((float)(ci * k * k));
        for (float &x : m.w[i]) x = r.uni() * g;
        for (float &x : m.b[i]) x = r.uni() * 0.1f;
        wtk_conv_blob &bl = m.blobs[i];
        std::memset(&bl, 0, sizeof(bl));
        bl.cout = co, bl.cin = ci, bl.k = k, bl.stride = s, bl.act = a, bl.weight = m.w[i].data(), bl.bias = m.b[i].data();
    }
    return m;
}

static wtk_yolo *create(const Model &m, int H, int W, int max_batch, int dtype, int plan, bool expect_ok = true) {
    wtk_yolo_desc d;
    std::memset(&d, 0, sizeof(d));
    d.device = 0, d.dtype = (wtk_dtype)dtype, d.imgsz_h = H, d.imgsz_w = W, d.max_batch = max_batch, d.nc = m.nc;
    d.width_mult = m.sc.width, d.depth_mult = m.sc.depth, d.max_channels = m.sc.maxch, d.n_convs = (int)m.blobs.size(), d.convs = m.blobs.data();
    wtk_yolo *h = nullptr;
    const int rc = wtk_yolo_create_planned(&h, &d, plan);
    if (expect_ok) CHECK(rc == 0 && h, "create %s %dx%d B%d dtype %d plan %d", m.sc.name, H, W, max_batch, dtype, plan);
    return rc == 0 ? h : nullptr;
}

struct DevBufs { // what a caller with device-resident frames holds
    void *frames = nullptr, *xywh = nullptr, *conf = nullptr, *anchor = nullptr, *idx = nullptr, *pos = nullptr, *cls = nullptr, *count = nullptr;
    void alloc(size_t frame_bytes, int B, int max_det) {
        hipMalloc(&frames, frame_bytes);
        hipMalloc(&xywh, (size_t)B * max_det * 16), hipMalloc(&conf, (size_t)B * max_det * 4), hipMalloc(&anchor, (size_t)B * max_det * 4);
        hipMalloc(&cls, (size_t)B * max_det * 4), hipMalloc(&count, (size_t)B * 4);
        hipMalloc(&idx, (size_t)B * 4), hipMalloc(&pos, (size_t)B * 8);
    }
    void release() {
        for (void *p : {frames, xywh, conf, anchor, idx, pos, cls, count}) hipFree(p);
    }
};

// everything a controller / the GPU tests do with one handle
static void exercise(wtk_yolo *h, int H, int W, int max_batch, bool small, void *user_stream, Rng &rng) {
    const int Bs[3] = {1, std::min(3, max_batch), max_batch};
    std::vector<uint8_t> host((size_t)max_batch * H * W * 3);
    for (size_t i = 0; i < host.size(); i += 97) host[i] = (uint8_t)rng.next();
    std::vector<float> xywh((size_t)max_batch * 4), conf(max_batch);
    std::vector<int32_t> anchor(max_batch);
    // host entry point: first call of the handle (staging allocation, host stream, capture on latency-plan handles), same arguments again (replay), other batches
    for (int rep = 0; rep < 2; ++rep)
        for (int B : Bs) {
            CHECK(wtk_yolo_predict_host(h, host.data(), B, H, W, 1, 0.1f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host gray B%d", B);
            CHECK(wtk_yolo_predict_host(h, host.data(), B, H, W, 3, 0.1f, 0.7f, 1, xywh.data(), nullptr, nullptr) == 0, "predict_host BGR B%d", B);
        }
    // letterboxed frame sizes (the reference's 360 -> 384 case): smaller and non-square sources
    if (H >= 64 && W >= 64) {
        CHECK(wtk_yolo_predict_host(h, host.data(), Bs[1], H - 24, W - 8, 1, 0.25f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host letterbox");
        CHECK(wtk_yolo_predict_host(h, host.data(), 1, H / 2, W - 8, 3, 0.25f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host letterbox 2");
    }
    // a larger frame than the staging buffer was sized for: re-allocation under live graphs
    {
        std::vector<uint8_t> big((size_t)(H + 40) * (W + 56) * 3);
        CHECK(wtk_yolo_predict_host(h, big.data(), 1, H + 40, W + 56, 3, 0.1f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host larger frame");
        CHECK(wtk_yolo_predict_host(h, host.data(), 1, H, W, 1, 0.1f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host after re-allocation");
    }
    // caller-owned device buffers on the caller's stream: met once (eager), met again (captured on latency-plan handles), replayed
    DevBufs d;
    const int F = max_batch + 2, FH = H + 32, FW = W + 48; // full frames for the views entry point
    d.alloc((size_t)F * FH * FW, max_batch, 5);
    for (int rep = 0; rep < 3; ++rep)
        for (int B : Bs) CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, B, H, W, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, user_stream) == 0, "predict dev B%d", B);
    CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, 1, H, W, 1, 0.1f, 0.7f, 1, (float *)d.xywh, nullptr, nullptr, nullptr) == 0, "predict on the null stream");
    for (int rep = 0; rep < 3; ++rep)
        for (int B : Bs) {
            CHECK(wtk_yolo_predict_views(h, (const uint8_t *)d.frames, F, FH, FW, 1, (const int32_t *)d.idx, (const int32_t *)d.pos, B, H - 24, H - 24, 0.1f, 0.7f, 1, (float *)d.xywh,
                                         (float *)d.conf, (int32_t *)d.anchor, user_stream) == 0, "predict_views B%d", B);
        }
    CHECK(wtk_yolo_predict_views(h, (const uint8_t *)d.frames, F, FH, FW, 1, nullptr, (const int32_t *)d.pos, 1, W, H, 0.1f, 0.7f, 1, (float *)d.xywh, nullptr, nullptr, user_stream) == 0, "predict_views no index");
    CHECK(wtk_yolo_predict_nms(h, (const uint8_t *)d.frames, Bs[1], H, W, 1, 0.1f, 0.7f, 5, (float *)d.xywh, (float *)d.conf, (int32_t *)d.cls, (int32_t *)d.anchor, (int32_t *)d.count, user_stream) == 0, "predict_nms");
    // stream layout changes drop the captured launches
    for (int n : {1, 0, 2}) {
        CHECK(wtk_yolo_set_side_streams(h, n) == 0, "set_side_streams %d", n);
        for (int rep = 0; rep < 2; ++rep) CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, 1, H, W, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, user_stream) == 0, "predict side %d", n);
        CHECK(wtk_yolo_predict_host(h, host.data(), 1, H, W, 1, 0.1f, 0.7f, 1, xywh.data(), conf.data(), anchor.data()) == 0, "predict_host side %d", n);
    }
    // the hybrid's device-side batch count
    CHECK(wtk_yolo_set_dynamic_batch(h, (const int32_t *)d.count) == 0, "set_dynamic_batch");
    for (int rep = 0; rep < 2; ++rep) CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, max_batch, H, W, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, user_stream) == 0, "predict dynamic");
    CHECK(wtk_yolo_set_dynamic_batch(h, nullptr) == 0, "set_dynamic_batch off");
    // profiling brackets
    CHECK(wtk_yolo_set_profiling(h, 1) == 0, "profiling on");
    CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, Bs[1], H, W, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, user_stream) == 0, "predict profiled");
    for (int k = 0; k < 7; ++k) {
        double ms, fl;
        int64_t n;
        CHECK(wtk_yolo_get_kernel_profile(h, k, &ms, &n, &fl) == 0, "kernel profile");
    }
    CHECK(wtk_yolo_set_profiling(h, 0) == 0, "profiling off");
    // test hooks
    int32_t flags = -1;
    CHECK(wtk_yolo_status(h, &flags, 1) == 0 && flags == 0, "status");
    double macs;
    int32_t A;
    CHECK(wtk_yolo_workload(h, &macs, &A) == 0 && A == (H / 8) * (W / 8) + (H / 16) * (W / 16) + (H / 32) * (W / 32), "workload");
    if (small) {
        const int B = Bs[1];
        std::vector<float> mg(B);
        CHECK(wtk_yolo_last_margins_host(h, B, mg.data()) == 0, "margins");
        for (int l = 0; l < 3; ++l) {
            const size_t Al = (size_t)(H >> (3 + l)) * (W >> (3 + l));
            std::vector<float> bx((size_t)B * Al * 64), cl((size_t)B * Al * 80);
            CHECK(wtk_yolo_debug_head(h, l, B, bx.data(), cl.data()) == 0, "debug_head");
        }
        int covered = 0;
        int32_t shp[3];
        for (int ci = 0; ci < 80; ++ci) { // (the class towers' first convs ride in the box towers' ops: no tensor of their own)
            if (wtk_yolo_debug_tensor(h, ci, B, nullptr, 0, shp) != 0) continue;
            std::vector<float> t((size_t)B * shp[0] * shp[1] * shp[2]);
            CHECK(wtk_yolo_debug_tensor(h, ci, B, t.data(), t.size(), shp) == 0, "debug_tensor %d", ci);
            ++covered;
        }
        CHECK(covered >= 57, "debug_tensor covered %d convs", covered);
        std::vector<float> box((size_t)B * A * 64, 0.f), cls((size_t)B * A * 80, -3.f);
        CHECK(wtk_yolo_decode_host(h, box.data(), cls.data(), B, H, W, 0.1f, xywh.data(), conf.data(), anchor.data()) == 0, "decode_host");
        std::vector<float> ox((size_t)B * 5 * 4), oc((size_t)B * 5);
        std::vector<int32_t> ok((size_t)B * 5), oa((size_t)B * 5), on(B);
        CHECK(wtk_yolo_decode_nms_host(h, box.data(), cls.data(), B, H, W, 0.1f, 0.7f, 5, ox.data(), oc.data(), ok.data(), oa.data(), on.data()) == 0, "decode_nms_host");
    }
    // argument errors must be errors, not crashes
    CHECK(wtk_yolo_predict_host(h, host.data(), max_batch + 1, H, W, 1, 0.1f, 0.7f, 1, xywh.data(), nullptr, nullptr) != 0, "batch > max_batch accepted");
    CHECK(wtk_yolo_predict_host(h, host.data(), 0, H, W, 1, 0.1f, 0.7f, 1, xywh.data(), nullptr, nullptr) != 0, "empty batch accepted");
    CHECK(wtk_yolo_predict_host(h, host.data(), 1, H, W, 2, 0.1f, 0.7f, 1, xywh.data(), nullptr, nullptr) != 0, "C = 2 accepted");
    CHECK(wtk_yolo_predict_host(h, host.data(), 1, H, W, 1, 0.1f, 0.7f, 3, xywh.data(), nullptr, nullptr) != 0, "max_det = 3 accepted by predict");
    d.release();
}

struct Case {
    int scale; // index into kScales
    int H, W, max_batch, dtype, plan, nc;
};

static void run_matrix(const char *label, const std::vector<Case> &cases, int keep_alive) {
    std::fprintf(stderr, "[hostsan] %s: %zu handles\n", label, cases.size());
    void *user_stream = nullptr;
    hipStreamCreateWithFlags(&user_stream, 1);
    std::deque<wtk_yolo *> alive; // destruction out of phase with creation, like a test process whose handles die when the garbage collector gets to them
    Rng rng(7);
    std::vector<Model> models;
    auto model_of = [&](int scale, int nc) -> const Model & {
        for (const Model &m : models)
            if (m.sc.name == kScales[scale].name && m.nc == nc) return m;
        models.push_back(make_model(kScales[scale], nc, 11 + scale + 7 * nc));
        return models.back();
    };
    models.reserve(16);
    for (const Case &c : cases) {
        const Model &m = model_of(c.scale, c.nc);
        const bool splitless = c.dtype == WTK_F16X3 && c.scale != 1; // n: widths 16 .. 256, m: 48 .. 576 — not multiples of 64: refused by design
        wtk_yolo *h = create(m, c.H, c.W, c.max_batch, c.dtype, c.plan, !splitless);
        if (splitless) {
            CHECK(h == nullptr, "f16x3 on scale %s must be refused", m.sc.name);
            continue;
        }
        if (!h) continue;
        const size_t act_px = (size_t)c.H * c.W * c.max_batch;
        exercise(h, c.H, c.W, c.max_batch, act_px <= (size_t)256 * 256 * 4, user_stream, rng);
        alive.push_back(h);
        while ((int)alive.size() > keep_alive) {
            wtk_yolo_destroy(alive.front());
            alive.pop_front();
        }
    }
    while (!alive.empty()) {
        wtk_yolo_destroy(alive.back()); // ... and the rest in the other order
        alive.pop_back();
    }
}

static void hybrid_and_misc() {
    std::fprintf(stderr, "[hostsan] hybrid, ResMLP, track ops\n");
    const Model m = make_model(kScales[1], 1, 3);
    void *st = nullptr;
    hipStreamCreateWithFlags(&st, 1);
    for (int defer : {1, 4}) {
        wtk_yolo *fast = create(m, 128, 128, 8, WTK_F16, WTK_PLAN_AUTO), *exact = create(m, 128, 128, 8, WTK_F16X3, WTK_PLAN_THROUGHPUT);
        wtk_hybrid *hy = nullptr;
        CHECK(wtk_hybrid_create(&hy, fast, exact, 0.08f, 0, defer) == 0, "hybrid_create");
        DevBufs d;
        d.alloc((size_t)10 * 160 * 176, 8, 1);
        for (int i = 0; i < 9; ++i) CHECK(wtk_hybrid_predict(hy, (const uint8_t *)d.frames, 8, 128, 128, 1, 0.1f, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, st) == 0, "hybrid_predict");
        if (defer == 1)
            CHECK(wtk_hybrid_predict_views(hy, (const uint8_t *)d.frames, 10, 160, 176, 1, (const int32_t *)d.idx, (const int32_t *)d.pos, 8, 120, 120, 0.1f, (float *)d.xywh, (float *)d.conf,
                                           (int32_t *)d.anchor, st) == 0, "hybrid_predict_views");
        CHECK(wtk_hybrid_flush(hy, st) == 0, "hybrid_flush");
        int64_t r, o;
        CHECK(wtk_hybrid_counters(hy, &r, &o) == 0, "hybrid_counters");
        CHECK(wtk_hybrid_hold(hy, 1) == 0 && wtk_hybrid_hold(hy, 0) == 0, "hybrid_hold");
        wtk_hybrid_destroy(hy);
        wtk_yolo_destroy(exact);
        wtk_yolo_destroy(fast);
        d.release();
    }
    // ResMLP (the reference's two models' shapes)
    for (int hidden : {40, 60}) {
        const int nb = hidden == 40 ? 4 : 6, lpb = 4;
        const int dims[4] = {hidden == 40 ? 10 : 20, hidden == 40 ? 4 : 8, hidden == 40 ? 10 : 20, hidden};
        std::vector<wtk_mlp_layer> L;
        std::vector<std::vector<float>> store;
        auto add = [&](int in, int out, int relu) {
            store.emplace_back((size_t)in * out, 0.01f);
            store.emplace_back((size_t)out, 0.0f);
            wtk_mlp_layer l;
            std::memset(&l, 0, sizeof(l));
            l.in_dim = in, l.out_dim = out, l.relu = relu, l.weight = store[store.size() - 2].data(), l.bias = store.back().data();
            L.push_back(l);
        };
        store.reserve(64);
        add(28, hidden, 1);
        for (int b = 0; b < nb; ++b) {
            int in = hidden;
            for (int l = 0; l < lpb; ++l) add(in, dims[l], 1), in = dims[l];
        }
        add(hidden, 2, 0);
        wtk_mlp_desc d;
        std::memset(&d, 0, sizeof(d));
        d.device = 0, d.n_layers = (int)L.size(), d.n_blocks = nb, d.layers_per_block = lpb, d.layers = L.data();
        wtk_mlp *mlp = nullptr;
        CHECK(wtk_mlp_create(&mlp, &d) == 0, "mlp_create");
        for (int B : {1, 7, 300}) {
            std::vector<float> x((size_t)B * 28, 0.5f), y((size_t)B * 2);
            CHECK(wtk_mlp_forward_host(mlp, x.data(), B, y.data()) == 0, "mlp_forward_host B%d", B);
        }
        void *track = nullptr, *anchors = nullptr, *pred = nullptr, *valid = nullptr;
        hipMalloc(&track, 1000 * 16), hipMalloc(&anchors, 100 * 4), hipMalloc(&pred, 100 * 8), hipMalloc(&valid, 100 * 4);
        const int32_t inf[7] = {0, -2, -9, -11, -18, -20, -27};
        CHECK(wtk_mlp_predict_track(mlp, (const float *)track, 1000, (const int32_t *)anchors, 100, inf, 7, (float *)pred, (int32_t *)valid, st) == 0, "mlp_predict_track");
        void *p64 = nullptr;
        hipMalloc(&p64, 100 * 16);
        CHECK(wtk_track_median_centers(track, 0, 1000, (const int32_t *)anchors, 100, 9, 6, (double *)p64, (int32_t *)valid, st) == 0, "track_median");
        const int32_t times[4] = {0, 2, 4, 5};
        const double wts[4] = {1, 1, 2, 3};
        CHECK(wtk_track_polyfit(track, 0, 1000, (const int32_t *)anchors, 100, 9, times, wts, 4, 2, 12.0, (double *)p64, (int32_t *)valid, st) == 0, "track_polyfit");
        void *X = nullptr, *Y = nullptr, *keep = nullptr;
        hipMalloc(&X, 500 * 28 * 4), hipMalloc(&Y, 500 * 2 * 4), hipMalloc(&keep, 500 * 4);
        const int32_t predf[1] = {9};
        CHECK(wtk_track_training_pairs(track, 0, 1000, 27, 500, inf, 7, predf, 1, (float *)X, (float *)Y, (int32_t *)keep, st) == 0, "track_pairs");
        for (void *p : {track, anchors, pred, valid, p64, X, Y, keep}) hipFree(p);
        wtk_mlp_destroy(mlp);
    }
}

// The latency plan's launch schedule (csrc/wtk_plan.hip: sk_schedule): one frame of YOLOv8s is 48 dependency levels — the fused front, 45 levels of split-K
// convs (a Detect tower's box and class convs, a PAN layer and the tower of the feature map before it share a level), the pool, the head — on ONE stream.
static void schedule_shape() {
    std::fprintf(stderr, "[hostsan] latency-plan schedule\n");
    const Model m = make_model(kScales[1], 1, 9);
    void *st = nullptr;
    hipStreamCreateWithFlags(&st, 1);
    wtk_yolo *h = create(m, 384, 384, 4, WTK_F16X3, WTK_PLAN_LATENCY);
    if (!h) return;
    DevBufs d;
    d.alloc((size_t)4 * 384 * 384, 4, 1);
    // (the first eager call at a batch size is preceded by the autotune's timing passes — the same forward pass many times: counted on the second call)
    CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, 1, 384, 384, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, st) == 0, "predict (tuning)");
    const int streams_before = stub_live_streams();
    auto count = [&](const char *k) { return stub_kernel_launches(k); };
    const unsigned long long sk0 = count("conv_sk_kernel"), fr0 = count("front_fused"), po0 = count("sppf_pool"), he0 = count("head_select"), all0 = count(nullptr), fin0 = count("sk_finish");
    CHECK(wtk_yolo_predict(h, (const uint8_t *)d.frames, 1, 384, 384, 1, 0.1f, 0.7f, 1, (float *)d.xywh, (float *)d.conf, (int32_t *)d.anchor, st) == 0, "predict");
    const unsigned long long sk = count("conv_sk_kernel") - sk0, fin = count("sk_finish") - fin0, all = count(nullptr) - all0;
    const bool grouped = !(std::getenv("WTK_SK_GROUP") && std::getenv("WTK_SK_GROUP")[0] == '0');
    if (grouped) {
        CHECK(sk == 45, "%llu grouped split-K launches per frame, expected 45 levels", sk);
        CHECK(all == 48 + fin, "%llu launches per frame, expected 48 levels + %llu second-launch combinations", all, fin);
    } else {
        CHECK(sk == 57, "%llu split-K launches per frame without grouping, expected one per conv op behind the front (57: 63 convs, 3 in the front, 3 x 2 Detect first convs as 3 ops)", sk);
    }
    CHECK(count("front_fused") - fr0 == 1 && count("sppf_pool") - po0 == 1 && count("head_select") - he0 == 1, "front / pool / head once each");
    CHECK(stub_live_streams() == streams_before, "a latency-plan forward created %d stream(s): it must run on the caller's stream alone", stub_live_streams() - streams_before);
    wtk_yolo_destroy(h);
    d.release();
}

// two host threads, a handle each, both capturing through the process-wide pair of side streams (ctypes releases the GIL: TrackPipeline lanes)
static void two_threads() {
    std::fprintf(stderr, "[hostsan] two host threads\n");
    const Model m = make_model(kScales[1], 1, 5);
    auto worker = [&](int id) {
        void *st = nullptr;
        hipStreamCreateWithFlags(&st, 1);
        Rng rng(100 + id);
        for (int round = 0; round < 3; ++round) {
            wtk_yolo *h = create(m, 128, 160, 4, id ? WTK_F16X3 : WTK_F32, WTK_PLAN_LATENCY);
            if (!h) return;
            std::vector<uint8_t> host((size_t)4 * 128 * 160);
            std::vector<float> xywh(16);
            for (int i = 0; i < 4; ++i) CHECK(wtk_yolo_predict_host(h, host.data(), 1 + (i & 1), 128, 160, 1, 0.1f, 0.7f, 1, xywh.data(), nullptr, nullptr) == 0, "threaded predict_host");
            wtk_yolo_destroy(h);
        }
    };
    std::thread a(worker, 0), b(worker, 1);
    a.join(), b.join();
}

// The launch layer must SEE what it claims to see: each of these is a protocol violation committed on purpose; the mode passes when every one is flagged.
static int selftest() {
    int flagged = 0, expected = 0;
    auto expect = [&](const char *what, int before) {
        ++expected;
        if (stub_violations() > before)
            ++flagged;
        else
            std::fprintf(stderr, "selftest: NOT flagged: %s\n", what);
    };
    void *s0 = nullptr, *s1 = nullptr, *ev = nullptr, *ev2 = nullptr, *g = nullptr, *x = nullptr, *buf = nullptr;
    hipStreamCreateWithFlags(&s0, 1), hipStreamCreateWithFlags(&s1, 1);
    hipEventCreateWithFlags(&ev, 2), hipEventCreateWithFlags(&ev2, 2);
    hipMalloc(&buf, 256);
    int v = stub_violations();
    char host[512] = {0};
    hipMemcpy(buf, host, 300, 1 /* H2D */);
    expect("copy past the end of a device allocation", v);
    v = stub_violations();
    hipMemset((char *)buf + 250, 0, 16);
    expect("memset past the end of a device allocation", v);
    // fork without join
    hipStreamBeginCapture(s0, 1 /* thread local */);
    hipEventRecord(ev, s0);
    hipStreamWaitEvent(s1, ev, 0);
    hipMemcpy(buf, host, 16, 1);
    v = stub_violations() - 1; // (the synchronous copy inside the capture is one)
    expect("synchronous copy by a capturing thread", v);
    v = stub_violations();
    hipStreamSynchronize(s1);
    expect("hipStreamSynchronize on a captured stream", v);
    v = stub_violations();
    hipEventDestroy(ev);
    expect("destroying an event recorded in an open capture", v);
    v = stub_violations();
    hipStreamEndCapture(s0, &g);
    expect("end of an invalidated capture is an error", stub_violations() > v || g == nullptr ? v - 1 : v);
    // a clean capture, then: replay after its buffer is freed, double destroy, wait on a stale captured event
    void *buf2 = nullptr;
    hipMalloc(&buf2, 64);
    hipStreamBeginCapture(s0, 1);
    hipEventRecord(ev2, s0);
    hipStreamEndCapture(s0, &g);
    if (g) hipGraphInstantiate(&x, g, nullptr, nullptr, 0), hipGraphDestroy(g);
    v = stub_violations();
    hipStreamWaitEvent(s1, ev2, 0);
    expect("wait on an event whose capture has ended", v);
    v = stub_violations();
    hipEventDestroy(ev2);
    hipEventRecord(ev2, s0);
    expect("record on a destroyed event", v);
    if (x) {
        hipGraphExecDestroy(x);
        v = stub_violations();
        hipGraphLaunch(x, s0);
        expect("launch of a destroyed graph exec", v);
        v = stub_violations();
        hipGraphExecDestroy(x);
        expect("double destroy of a graph exec", v);
    }
    v = stub_violations();
    hipFree(buf2), hipFree(buf2);
    expect("double free", v);
    hipFree(buf);
    std::fprintf(stderr, "[hostsan] selftest: %d of %d deliberate violations flagged\n", flagged, expected);
    return flagged == expected ? 0 : 1;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "quick";
    if (mode == "selftest") return selftest();
    if (std::getenv("WTK_STUB_VERBOSE")) stub_set_verbose(1);
    CHECK(wtk_abi_version() == WTK_ABI_VERSION, "abi version");
    CHECK(wtk_device_count() == 1, "device count");
    const int F32 = WTK_F32, F16 = WTK_F16, X3 = WTK_F16X3, AUTO = WTK_PLAN_AUTO, THR = WTK_PLAN_THROUGHPUT, LAT = WTK_PLAN_LATENCY;
    std::vector<Case> cases;
    // the two configurations of the round-5 crashes first (scale n, B = 3, tiny maps; a latency-plan fp32 handle of capacity 16 and a throughput-plan fp16 handle)
    cases.push_back({0, 160, 160, 16, F32, LAT, 1});
    cases.push_back({0, 96, 160, 3, F16, THR, 1});
    cases.push_back({0, 96, 160, 3, F32, AUTO, 1});
    for (int dtype : {F32, F16, X3})
        for (int plan : {AUTO, THR, LAT}) {
            if (plan == LAT && dtype == F16) continue;
            cases.push_back({0, 128, 128, 3, dtype, plan, 1});
            cases.push_back({1, 128, 128, 2, dtype, plan, 1});
            cases.push_back({1, 96, 160, 3, dtype, plan, 1});
            cases.push_back({1, 384, 384, 16, dtype, plan, 1});
            if (mode == "full") {
                cases.push_back({0, 160, 160, 2, dtype, plan, 2});
                cases.push_back({1, 256, 256, 3, dtype, plan, 32});
                cases.push_back({1, 352, 224, 3, dtype, plan, 1});
                cases.push_back({1, 384, 384, 1, dtype, plan, 1});
                cases.push_back({1, 384, 384, 4, dtype, plan, 1});
                cases.push_back({1, 640, 640, 1, dtype, plan, 1});
                cases.push_back({1, 640, 640, 64, dtype, plan, 1});
                cases.push_back({1, 640, 640, 8, dtype, plan, 80});
                cases.push_back({2, 160, 160, 2, dtype, plan, 3});
                cases.push_back({1, 1280, 1280, 4, dtype, plan, 1});
            }
        }
    if (mode == "full") {
        cases.push_back({1, 1280, 1280, 64, F16, THR, 1});
        cases.push_back({1, 640, 640, 256, F16, THR, 1});
        cases.push_back({1, 32, 32, 1, F32, AUTO, 1}); // the smallest legal network: 4 x 4, 2 x 2, 1 x 1 maps
        cases.push_back({0, 32, 64, 5, F16, AUTO, 1});
        cases.push_back({1, 64, 32, 17, X3, AUTO, 1});
    }
    run_matrix("shape matrix", cases, 3);
    hybrid_and_misc();
    schedule_shape();
    two_threads();
    const int v = stub_violations();
    std::fprintf(stderr, "[hostsan] kernels launched: %llu (conv_sk %llu, window %llu, igemm %llu, front %llu, head %llu); peak device memory %.1f GB; violations %d; driver failures %d\n",
                 stub_kernel_launches(nullptr), stub_kernel_launches("conv_sk_kernel"), stub_kernel_launches("conv3x3_halo"), stub_kernel_launches("conv_igemm_kernel"),
                 stub_kernel_launches("front_fused"), stub_kernel_launches("head_"), (double)stub_peak_device_bytes() / 1e9, v, g_fail);
    // everything the handles took must be back: events and graph execs destroyed, no capture left open, only the process-wide status page and nothing else alive
    CHECK(stub_live_events() == 0, "%d events leaked", stub_live_events());
    CHECK(stub_live_execs() == 0, "%d graph execs leaked", stub_live_execs());
    CHECK(stub_open_captures() == 0, "%d captures left open", stub_open_captures());
    CHECK(stub_live_device_bytes() == 0, "%zu bytes of device memory leaked", stub_live_device_bytes());
    std::fprintf(stderr, "[hostsan] live streams at exit (pooled, by design): %d\n", stub_live_streams());
    return (v || g_fail) ? 1 : 0;
}
