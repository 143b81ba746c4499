// The detector handle, part 2 (see wtk_internal.h): one forward pass — letterbox / view cut, the fused front, the conv ops (one grouped split-K launch per
// dependency level on latency-plan handles), pool, head —, the opt-in replay of a captured pass, the predict entry points and the test hooks.
#include "wtk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace wtk;

#ifdef WTK_WS64_STAMPS // diagnostic builds: per-wave interval stamps of the kernel under study (tools/gpu_sessions/ws64_stamps.py)
static unsigned long long *g_dbg_stamps = nullptr;
constexpr size_t kDbgStampBytes = 1 << 20;
extern "C" int wtk_debug_stamps(unsigned long long *host, size_t n_words) {
    if (!g_dbg_stamps || n_words * 8 > kDbgStampBytes) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpy(host, g_dbg_stamps, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

// ultralytics LetterBox geometry (auto=False: pad to exactly imgsz) + scale_boxes inverse
static void letterbox_geom(int H, int W, int Sh, int Sw, int &new_h, int &new_w, int &top, int &left, float &gain, float &pad_x, float &pad_y) {
    const double r = std::min((double)Sh / H, (double)Sw / W);
    new_w = (int)std::nearbyint(W * r);
    new_h = (int)std::nearbyint(H * r);
    const double dw = (Sw - new_w) / 2.0, dh = (Sh - new_h) / 2.0;
    top = (int)std::nearbyint(dh - 0.1);
    left = (int)std::nearbyint(dw - 0.1);
    // scale_boxes recomputes gain/pad from the two shapes
    gain = (float)std::min((double)Sh / H, (double)Sw / W);
    pad_x = (float)std::nearbyint((Sw - W * (double)gain) / 2.0 - 0.1);
    pad_y = (float)std::nearbyint((Sh - H * (double)gain) / 2.0 - 0.1);
}

// outputs of the general NMS path (max_det >= 1 rows per image)
struct NmsOut {
    float iou;
    int max_det;
    int *out_cls, *out_count;
};
static int run_head(wtk_yolo *h, int B, int H, int W, float conf, float *out_xywh, float *out_conf, int *out_anchor, hipStream_t st,
                    const NmsOut *nms = nullptr) {
    HeadArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < 3; ++i) {
        a.box[i] = h->bufs[h->box_buf[i]].ptr;
        a.cls[i] = h->bufs[h->cls_buf[i]].ptr;
        a.lh[i] = h->lh[i];
        a.lw[i] = h->lw[i];
    }
    a.cls_ld = h->cls_ld;
    a.nc = h->dims.nc;
    a.N = B;
    a.conf = conf;
    int nh, nw, top, left;
    letterbox_geom(H, W, h->S_h, h->S_w, nh, nw, top, left, a.gain, a.pad_x, a.pad_y);
    a.img_w = (float)W;
    a.img_h = (float)H;
    a.out_xywh = out_xywh;
    a.out_conf = out_conf;
    a.out_anchor = out_anchor;
    a.out_margin = h->o_margin;
    a.status = h->status_dev;
    a.conf_logit = conf > 0.f && conf < 1.f ? std::log(conf / (1.f - conf)) : (conf <= 0.f ? -INFINITY : INFINITY);
    if (nms) {
        NmsArgs q;
        q.h = a;
        q.iou = nms->iou, q.max_det = nms->max_det;
        q.scratch_score = h->nms_score, q.scratch_cls = h->nms_cls, q.scratch_box = h->nms_box;
        q.out_xywh = out_xywh, q.out_conf = out_conf, q.out_anchor = out_anchor, q.out_cls = nms->out_cls, q.out_count = nms->out_count;
        HIP_TRY(launch_head_nms(q, 0, st)); // the Detect outputs are fp32 tensors in both modes
        return 0;
    }
    HIP_TRY(launch_head(a, 0, st)); // the Detect outputs are fp32 tensors in both modes
    return 0;
}

static int ensure_nms_scratch(wtk_yolo *h, hipStream_t st) {
    if (h->nms_score) return 0;
    HIP_TRY(hipStreamSynchronize(st));
    const size_t n = (size_t)h->max_batch * h->anchors;
    HIP_TRY(hipMalloc(&h->nms_score, n * sizeof(float)));
    HIP_TRY(hipMalloc(&h->nms_cls, n * sizeof(int)));
    HIP_TRY(hipMalloc(&h->nms_box, n * 4 * sizeof(float)));
    return 0;
}

// Enqueue one forward pass (letterbox, stem, convs, pool, head) on `st`.  No allocation, no synchronisation
// (profiling mode excepted): safe inside stream capture.
// The pair of side streams is shared by every handle of the process on a device (ensure_side_streams).  Handles driven from different host
// threads (ctypes releases the GIL) must not interleave on it: a stream capture in one thread (the graph path of wtk_yolo_predict pulls the side
// streams into a hipStreamCaptureModeThreadLocal capture through the event waits) would swallow or reject the other thread's launches.  Every
// enqueue that touches the shared pair, and the whole capture bracket, holds this lock; a single-threaded caller (the bench, the controllers)
// never contends on it.
static std::recursive_mutex g_side_mu;

// side streams and their events, taken at the first forward pass that uses them
static int ensure_side_streams(wtk_yolo *h) {
    // ONE pair of side streams per process and device, shared by every handle and never destroyed.  The HIP runtime multiplexes streams onto its
    // hardware queues (four by default); with two lanes (two caller streams) a pair per handle made six streams, and which of them shared a queue
    // depended on the order in which streams had been created in the process: the same workload ran at 24.5 .. 27 k frames/s (fp16) or 14.8 .. 17.7 k
    // (hybrid) depending on what had run before it (tools/gpu_sessions/order_probe.py).  Two callers + one shared pair = four streams: every stream
    // has a queue of its own, and the rate no longer depends on the history of the process.  The towers of different handles then run one after the
    // other on a side stream; lanes are out of phase, nothing is lost (26.8 k / 17.6 k).
    for (int i = 1; i <= 2; ++i) {
        if (!h->side_stream[i]) {
            static std::mutex mu;
            static std::vector<std::pair<int, hipStream_t>> g_shared[2]; // per slot: (device, stream)
            std::lock_guard<std::mutex> lk(mu);
            for (auto &e : g_shared[i - 1])
                if (e.first == h->device) h->side_stream[i] = e.second;
            if (!h->side_stream[i]) {
                HIP_TRY(hipStreamCreateWithFlags(&h->side_stream[i], hipStreamNonBlocking));
                g_shared[i - 1].emplace_back(h->device, h->side_stream[i]);
            }
        }
        if (!h->side_done[i]) HIP_TRY(hipEventCreateWithFlags(&h->side_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i)
        if (!h->feat_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->feat_ev[i], hipEventDisableTiming));
    return 0;
}

// `vs` != nullptr: the batch rows are camera views of full frames (wtk_yolo_predict_views) — crop + letterbox in one kernel.
struct ViewSrc {
    const int32_t *pos_xy, *frame_index;
    int view_w, view_h, n_frames;
};
static int yolo_enqueue_pass(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float *out_xywh,
                             float *out_conf, int32_t *out_anchor, hipStream_t st, const ViewSrc *vs, const NmsOut *nms) {
    const uint8_t *net_in = frames_dev;
    if (vs) {
        ViewLetterboxArgs va;
        va.frames = frames_dev, va.frame_index = vs->frame_index, va.pos_xy = vs->pos_xy, va.dst = h->lb_dev;
        va.N = B, va.H = H, va.W = W, va.C = C;
        va.F = vs->n_frames;
        va.view_w = vs->view_w, va.view_h = vs->view_h;
        va.rows = vs->view_w, va.cols = vs->view_h; // frame[y : y + w, x : x + h], view_controller.py:171
        va.Sh = h->S_h, va.Sw = h->S_w;
        float g, px, py;
        letterbox_geom(va.rows, va.cols, h->S_h, h->S_w, va.new_h, va.new_w, va.top, va.left, g, px, py);
        HIP_TRY(launch_view_letterbox(va, st));
        net_in = h->lb_dev;
        H = va.rows, W = va.cols; // from here on the "image" is the view: scale_boxes maps back to view pixels
    } else if (H != h->S_h || W != h->S_w) {
        LetterboxArgs la;
        la.src = frames_dev;
        la.dst = h->lb_dev;
        la.N = B, la.H = H, la.W = W, la.C = C;
        la.Sh = h->S_h, la.Sw = h->S_w;
        float g, px, py;
        letterbox_geom(H, W, h->S_h, h->S_w, la.new_h, la.new_w, la.top, la.left, g, px, py);
        HIP_TRY(launch_letterbox(la, st));
        net_in = h->lb_dev;
    }

    int cur_class = -1, nev = 0;
    int ev_class[wtk_yolo::kProfEvents];
    auto mark = [&](int cls) -> int {
        if (!h->profiling || cls == cur_class) return 0;
        if (nev >= wtk_yolo::kProfEvents - 1) return 0;
        HIP_TRY(hipEventRecord(h->ev[nev], st));
        ev_class[nev] = cls;
        ++nev;
        cur_class = cls;
        return 0;
    };
    long long launches[wtk_yolo::kProfKernels] = {};
    double flops[wtk_yolo::kProfKernels] = {};
    auto op_flops = [&](const Op &o) { return 2.0 * B * o.macs_per_image; }; // algorithmic: 2 x output pixels x cout x (cin x k x k)

    // Two lanes: the caller's stream runs backbone + PAN + the P5 tower; the P3 / P4 Detect towers run on
    // the side stream as soon as their feature map is complete and fill the tails of the small PAN kernels.
    // Profiling keeps everything on one stream so the per-class event brackets stay meaningful.
    // latency-plan handles (round 6): everything on the caller's stream, independent convs grouped per dependency level into one launch each
    const bool grouped = h->latency && h->sk_group && !h->lat_sched.empty();
    if (!grouped && h->use_side && h->side_streams > 0 && !h->profiling && ensure_side_streams(h)) return 1;
    const bool two_lanes = !grouped && h->use_side && h->side_streams > 0 && h->side_stream[1] && !h->profiling;
    std::unique_lock<std::recursive_mutex> side_lock;
    if (two_lanes) side_lock = std::unique_lock<std::recursive_mutex>(g_side_mu);
    unsigned side_used = 0; // bit i: side_stream[i] carries work of this pass
    hipStream_t main_st = st;
    size_t first_op = 0;
    if (h->use_front && reinterpret_cast<uintptr_t>(net_in) % 4 == 0) {
        if (mark(5)) return 1;
        const Op &o0 = h->ops[0], &o1 = h->ops[1], &o2 = h->ops[2];
        FrontArgs f;
        std::memset(&f, 0, sizeof(f));
        f.frames = net_in;
        f.N = B, f.H = h->S_h, f.W = h->S_w, f.C = C;
        f.w0 = o0.w, f.b0 = o0.bias;
        f.w1 = o1.w, f.b1 = o1.bias, f.Kpad1 = o1.Kpad;
        f.w2 = o2.w, f.b2 = o2.bias, f.Kpad2 = o2.Kpad;
        f.out = h->bufs[o2.out_buf].ptr;
        f.out_ld = h->bufs[o2.out_buf].C;
        f.out_coff = o2.out_coff;
        if (h->front_debug) f.dbg_t0 = h->bufs[o0.out_buf].ptr, f.dbg_t1 = h->bufs[o1.out_buf].ptr;
        if (h->split) { // pseudo-channels (see the conv path below)
            f.Kpad1 *= 2, f.Kpad2 *= 2, f.out_ld *= 2, f.out_coff *= 2;
            f.n_dyn = h->n_dyn;
            f.stem_split = 1;
            HIP_TRY(launch_front_fused_split(f, h->num_cus, st));
        } else
            HIP_TRY(launch_front_fused(f, h->num_cus, st));
        ++launches[5];
        flops[5] += op_flops(o0) + op_flops(o1) + op_flops(o2);
        first_op = 3;
    }
    // the conv of `op` as the implicit-GEMM / split-K launchers take it (split handles: pseudo-channel arguments)
    auto conv_args = [&](const Op &op) -> ConvArgs {
        const Buf &ib = h->bufs[op.in_buf];
        const Buf &ob = h->bufs[op.out_buf];
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.in = ib.ptr;
        a.in_ld = ib.C;
        a.in_coff = op.in_coff;
        a.N = B, a.H = ib.h, a.W = ib.w, a.Cin = op.cin;
        a.Ho = ob.h, a.Wo = ob.w, a.Cout = op.cout;
        a.CoutPad = op.cout_pad;
        a.KH = a.KW = op.k;
        a.stride = op.stride;
        a.pad = op.k / 2;
        a.w = op.w;
        a.bias = op.bias;
        a.out = ob.ptr;
        a.out_ld = ob.C;
        a.out_coff = op.out_coff;
        a.out_f32 = ob.f32;
        a.n_dyn = h->n_dyn;
        if (op.out2_buf >= 0) {
            a.out2 = h->bufs[op.out2_buf].ptr;
            a.out2_ld = h->bufs[op.out2_buf].C;
            a.out2_coff = op.out2_coff;
        }
        if (op.in2_buf >= 0) {
            a.in2 = h->bufs[op.in2_buf].ptr;
            a.in2_ld = h->bufs[op.in2_buf].C;
            a.in2_coff = op.in2_coff;
            a.in2_split = op.in2_split;
        }
        if (op.res_buf >= 0) {
            a.res = h->bufs[op.res_buf].ptr;
            a.res_ld = h->bufs[op.res_buf].C;
            a.res_coff = op.res_coff;
        }
        a.act = op.act;
        a.K = op.K;
        a.Kpad = op.Kpad;
        a.M = (long long)B * ob.h * ob.w;
        a.tile_w = op.tile_w;
        a.zeros = h->zero_page;
        if (op.tile_w) {
            const int th = conv_cfg_bm(op.cfg) / op.tile_w;
            a.tiles_x = (ob.w + op.tile_w - 1) / op.tile_w;
            a.tiles_y = (ob.h + th - 1) / th;
        }
        if (h->split) {
            // pseudo-channels: every channel count / offset of a split tensor doubles (an fp32 output keeps its real layout)
            a.in_ld *= 2, a.in_coff *= 2, a.Cin *= 2, a.K *= 2, a.Kpad *= 2;
            a.in2_ld *= 2, a.in2_coff *= 2, a.in2_split *= 2;
            a.res_ld *= 2, a.res_coff *= 2, a.out2_ld *= 2, a.out2_coff *= 2;
            if (!a.out_f32) a.out_ld *= 2, a.out_coff *= 2;
        }
        return a;
    };
    auto run_op = [&](size_t oi) -> int {
        const Op &op = h->ops[oi];
        if (h->use_c2f && (oi == 3 || oi == 4)) return 0; // folded into the fused C2f tail launched at op 5
        if (op.folded) return 0;                          // runs in the epilogue of the op that names it as tail_op
        if (h->use_c2f && oi == 5) {
            if (mark(5)) return 1;
            const Op &m1 = h->ops[3], &m2 = h->ops[4];
            const Buf &cb = h->bufs[op.in_buf];
            C2fArgs c;
            std::memset(&c, 0, sizeof(c));
            c.cat = cb.ptr, c.cat_ld = cb.C, c.a_coff = op.in_coff, c.b_coff = m1.in_coff;
            c.N = B, c.H = cb.h, c.W = cb.w;
            c.w_m1 = m1.w, c.b_m1 = m1.bias, c.w_m2 = m2.w, c.b_m2 = m2.bias, c.Kpad_m = m1.Kpad;
            c.w_cv2 = op.w, c.b_cv2 = op.bias, c.Kpad_cv2 = op.Kpad;
            c.out = h->bufs[op.out_buf].ptr, c.out_ld = h->bufs[op.out_buf].C, c.out_coff = op.out_coff;
            c.zeros = h->zero_page;
            HIP_TRY(launch_c2f_fused(c, h->num_cus, main_st));
            ++launches[5];
            flops[5] += op_flops(m1) + op_flops(m2) + op_flops(op);
            return 0;
        }
        st = main_st;
        if (two_lanes && op.side) {
            const int sidx = std::min(op.side, h->side_streams); // wtk_yolo_set_side_streams(1): both towers on side stream 1
            st = h->side_stream[sidx];
            if (op.wait_feat >= 0) HIP_TRY(hipStreamWaitEvent(st, h->feat_ev[op.wait_feat], 0));
            side_used |= 1u << sidx;
        }
        if (op.kind == OP_STEM) {
            if (mark(0)) return 1;
            StemArgs a;
            a.frames = net_in;
            a.N = B, a.H = h->S_h, a.W = h->S_w, a.C = C;
            a.w = op.w;
            a.bias = op.bias;
            a.out = h->bufs[op.out_buf].ptr;
            a.Cout = op.cout;
            a.Ho = h->S_h / 2, a.Wo = h->S_w / 2;
            a.out_split = h->split; // split store
            a.in_split = h->split;  // split operands
            a.n_dyn = h->n_dyn;
            HIP_TRY(launch_stem(a, h->is_f16, st));
            ++launches[0];
            flops[0] += op_flops(op);
        } else if (op.kind == OP_POOL) {
            if (mark(2)) return 1;
            const Buf &b = h->bufs[op.in_buf];
            PoolArgs a;
            a.buf = b.ptr;
            a.N = B, a.H = b.h, a.W = b.w, a.c = op.cin;
            a.split = h->split;
            HIP_TRY(launch_sppf_pool(a, h->is_f16, st));
            ++launches[2];
        } else {
            const int kid = op.sk ? 1 : (op.halo == 2 ? 6 : (op.halo ? 4 : 1));
            if (mark(kid)) return 1;
            const Buf &ib = h->bufs[op.in_buf];
            const Buf &ob = h->bufs[op.out_buf];
            ConvArgs a = conv_args(op);
            if (op.sk) {
                a.tile_w = 0;
                if (!conv_sk_eligible(a, h->split)) return fail("internal: conv " + std::to_string(oi) + " of the latency plan does not fit conv_sk_kernel");
                const SkMember one{a, op.sk_atoms, op.sk_partial, op.sk_tickets};
                HIP_TRY(launch_conv_sk_group(&one, 1, h->split, h->num_cus, h->sk_force_tile, h->sk_force_form, st, &h->sk_choices[((long long)(oi + 100000) << 24) | (long long)B]));
            } else if (h->split && !op.halo && h->use_s2win && ib.h == 2 * ob.h && ib.w == 2 * ob.w &&
                split_s2win_eligible(op.k, op.stride, op.cin, op.cout, op.cout_pad, ob.w, op.res_buf < 0 && op.out2_buf < 0 && op.in2_buf < 0 && !ob.f32)) {
                // strided 3x3, split operands: the parity-plane window kernel on pseudo-channels
                HaloArgs g;
                std::memset(&g, 0, sizeof(g));
                g.in = a.in, g.in_ld = a.in_ld, g.in_coff = a.in_coff;
                g.N = B, g.H = ob.h, g.W = ob.w, g.Cin = a.Cin;
                g.Cout = op.cout, g.CoutPad = op.cout_pad;
                g.w = op.w, g.bias = op.bias;
                g.out = a.out, g.out_ld = a.out_ld, g.out_coff = a.out_coff;
                g.act = op.act, g.Kpad = a.Kpad;
                g.n_dyn = h->n_dyn;
                g.S = ob.w, g.pitch = ob.w + 1, g.strips = 1;
                g.bm = 256;
                g.blocks_per_strip = (int)(((long long)B * (ob.h + 1) * g.pitch + 255) / 256);
                if (2LL * g.blocks_per_strip * (op.cout_pad / 128) <= h->num_cus) {
                    g.bm = 128;
                    g.blocks_per_strip = (int)(((long long)B * (ob.h + 1) * g.pitch + 127) / 128);
                }
                g.zeros = h->zero_page;
                HIP_TRY(launch_conv3x3_s2_split(g, st));
            } else if (h->split && !op.halo) {
                int cfg = op.cfg;
                // a small handle's 128 x 128-tile layer whose grid leaves a third of the CUs idle: 64-cout tiles, twice the blocks (same K order: same bits)
                if (h->small_narrow && h->split && cfg == CFG_128x128 && !a.in2 && !a.tile_w &&
                    3 * ((a.M + 127) / 128) * (a.CoutPad / 128) <= 2LL * h->num_cus)
                    cfg = CFG_128x64;
                HIP_TRY(launch_conv_split(a, cfg, st));
            } else if (op.halo) {
                HaloArgs g;
                std::memset(&g, 0, sizeof(g));
                g.in = a.in, g.in_ld = a.in_ld, g.in_coff = a.in_coff;
                g.N = B, g.H = ib.h, g.W = ib.w, g.Cin = op.cin;
                g.Cout = op.cout, g.CoutPad = op.cout_pad;
                g.w = op.w, g.bias = op.bias;
                g.out = a.out, g.out_ld = a.out_ld, g.out_coff = a.out_coff;
                g.out2 = a.out2, g.out2_ld = a.out2_ld, g.out2_coff = a.out2_coff;
                g.res = a.res, g.res_ld = a.res_ld, g.res_coff = a.res_coff;
                g.act = op.act, g.Kpad = op.Kpad;
                g.n_dyn = h->n_dyn;
                g.slabs = h->halo_slabs;
                if (h->split) g.Cin = a.Cin, g.Kpad = a.Kpad, g.slabs = 3; // pseudo-channels
                if (op.tail_op >= 0) {
                    const Op &t = h->ops[op.tail_op];
                    g.tail_w = t.w, g.tail_bias = t.bias, g.tail_kpad = t.Kpad;
                    g.tail_out = h->bufs[t.out_buf].ptr, g.tail_ld = h->bufs[t.out_buf].C, g.tail_coff = t.out_coff;
                    g.tail_cout = t.cout;
                    g.tail_f32 = h->bufs[t.out_buf].f32;
                    if (h->split) { // pseudo-channels for the split weights (and for a split output; the fp32 head logits keep their real layout)
                        g.tail_kpad *= 2;
                        if (!g.tail_f32) g.tail_ld *= 2, g.tail_coff *= 2;
                    }
                }
                g.persist_cus = h->halo_persist ? h->num_cus : 0;
                const int rows_max = (op.halo == 2 && h->split) ? c32_split_rows_max() : (op.halo == 2 || h->split) ? kHaloRowsMax : halo_rows_max(op.cout, h->halo_slabs);
                bool ws64 = false;
                if (op.halo == 1 && h->use_ws64 && h->halo_slabs == 3 &&
                    ws64_eligible(op.k, op.stride, op.cin, op.cout, op.cout_pad, h->is_f16, op.out2_buf >= 0, op.tail_op >= 0)) {
                    halo_geometry_stacked(B, ib.h, ib.w, ws64_rows_max(), &g.S, &g.pitch, &g.strips, &g.blocks_per_strip);
                    // worth it when every group of a persistent block gets at least two tiles (weights are staged once per block)
                    ws64 = (long long)g.strips * g.blocks_per_strip >= 4LL * h->num_cus;
                }
                if (ws64) {
                    g.zeros = h->zero_page;
                    g.bm = 0; // (the weave schedules of round 3 lost: the round-2 schedule)
#ifdef WTK_WS64_STAMPS
                    if (std::getenv("WTK_WS64_STAMPS")) {
                        if (!g_dbg_stamps) HIP_TRY(hipMalloc(&g_dbg_stamps, kDbgStampBytes));
                        g.dbg_stamps = g_dbg_stamps;
                    }
#endif
                    HIP_TRY(launch_conv3x3_ws64(g, h->num_cus, st));
                } else if (op.halo == 2) {
                    halo_geometry(ib.h, ib.w, rows_max, &g.S, &g.pitch, &g.strips, &g.blocks_per_strip);
                } else {
                    halo_geometry_stacked(B, ib.h, ib.w, rows_max, &g.S, &g.pitch, &g.strips, &g.blocks_per_strip);
                    if (h->halo_slabs == 3 || h->split) {
                        // small maps: halve the blocks when 256-pixel blocks leave at least half of the CUs without work
                        const long long tiles = (long long)g.strips * g.blocks_per_strip * (op.cout_pad / (h->split ? split_halo_cout_tile(op.cout) : halo_cout_tile(op.cout)));
                        if (h->halo_small_blocks && 2 * tiles <= h->num_cus) {
                            g.bm = 128;
                            halo_geometry_stacked(B, ib.h, ib.w, rows_max, &g.S, &g.pitch, &g.strips, &g.blocks_per_strip, 128);
                            // still under half of the CUs with 128-pixel blocks (a small handle's cycle batch on the 24 x 24 maps): 64-cout tiles as well —
                            // each block then walks the same taps over half the couts
                            if (h->small_narrow && h->split && op.tail_op < 0 && op.cout_pad % 128 == 0 &&
                                2LL * g.strips * g.blocks_per_strip * (op.cout_pad / 128) <= h->num_cus)
                                g.narrow = 1;
                        }
                    }
                }
                // fp32 handles: the exact-fp32 matrix instructions make these layers arithmetic bound, so a grid on under three quarters of the CUs (a small
                // handle's 48 x 48 maps: 141-150 blocks of 128 / 192 couts) is cut into 64-cout tiles (Detect P3 first convs 205 us, class tower 139 us before)
                if (h->small_narrow && !h->split && !ws64 && op.halo == 1 && op.tail_op < 0 && op.cout_pad % 64 == 0 && halo_cout_tile(op.cout) != 64 &&
                    4LL * g.strips * g.blocks_per_strip * (op.cout_pad / halo_cout_tile(op.cout)) <= 3LL * h->num_cus)
                    g.narrow = 1;
                // Small f16x3 handles: the 64-cout window tiles on the six-slab ring with fragment prefetch (conv3x3_halo.hip; bit-identical to the
                // three-slab kernel).  A cycle batch's 24 x 24 layers 19.4 -> 15.9 us each; the 256-pixel tiles and the large handles measure the
                // same either way (profiles/r05_notes.md section 7), so those keep the three-slab kernel.  WTK_HALO_DEEP: 0 off, 1 small handles
                // (default), 2 every handle; read when the handle is created.
                if (h->halo_deep && op.halo == 1 && !ws64) g.deep = 1;
                g.zeros = h->zero_page;
                if (ws64) {
                } else if (h->split && op.halo == 2)
                    HIP_TRY(launch_conv3x3_c32_split(g, st));
                else if (h->split)
                    HIP_TRY(launch_conv3x3_halo_split(g, st));
                else if (op.halo == 2)
                    HIP_TRY(launch_conv3x3_c32(g, st));
                else
                    HIP_TRY(launch_conv3x3_halo(g, h->is_f16, st));
            } else if (h->use_s2win && op.tail_op < 0 &&
                       s2win_eligible(op.k, op.stride, op.cin, op.cout, op.cout_pad, h->is_f16, ob.w, op.res_buf < 0 && op.out2_buf < 0 && op.in2_buf < 0) &&
                       ib.h == 2 * ob.h && ib.w == 2 * ob.w) {
                // strided 3x3: parity-plane window kernel; the geometry lives on the OUTPUT map (stacked images, one strip)
                HaloArgs g;
                std::memset(&g, 0, sizeof(g));
                g.in = a.in, g.in_ld = a.in_ld, g.in_coff = a.in_coff;
                g.N = B, g.H = ob.h, g.W = ob.w, g.Cin = op.cin;
                g.Cout = op.cout, g.CoutPad = op.cout_pad;
                g.w = op.w, g.bias = op.bias;
                g.out = a.out, g.out_ld = a.out_ld, g.out_coff = a.out_coff;
                g.act = op.act, g.Kpad = op.Kpad;
                g.n_dyn = h->n_dyn;
                g.S = ob.w, g.pitch = ob.w + 1, g.strips = 1;
                g.bm = 256;
                g.blocks_per_strip = (int)(((long long)B * (ob.h + 1) * g.pitch + 255) / 256);
                if (2LL * g.blocks_per_strip * (op.cout_pad / 128) <= h->num_cus) { // small maps: half-size blocks fill the chip
                    g.bm = 128;
                    g.blocks_per_strip = (int)(((long long)B * (ob.h + 1) * g.pitch + 127) / 128);
                }
                g.zeros = h->zero_page;
                HIP_TRY(launch_conv3x3_s2(g, st));
            } else if (op.tail_op >= 0) { // implicit GEMM with the 1x1 behind it fused into its epilogue
                const Op &t = h->ops[op.tail_op];
                a.tail_w = t.w, a.tail_bias = t.bias, a.tail_kpad = t.Kpad, a.tail_act = t.act;
                a.tail_out = h->bufs[t.out_buf].ptr, a.tail_ld = h->bufs[t.out_buf].C, a.tail_coff = t.out_coff;
                HIP_TRY(launch_conv(a, op.cfg, h->is_f16, st));
            } else if (h->use_wide && conv1x1_wide_eligible(a, h->is_f16) && a.CoutPad >= 256 && ((a.M + 255) / 256) * (a.CoutPad / 128) >= 384) {
                HIP_TRY(launch_conv1x1_wide(a, st));
            } else {
                HIP_TRY(launch_conv(a, op.cfg, h->is_f16, st));
            }
            ++launches[kid];
            flops[kid] += op_flops(op) + (op.tail_op >= 0 ? op_flops(h->ops[op.tail_op]) : 0.0);
            if (two_lanes && op.signal_feat >= 0) HIP_TRY(hipEventRecord(h->feat_ev[op.signal_feat], main_st));
        }
            return 0;
    };
    if (grouped) {
        // latency plan: ONE stream, one launch per dependency level (sk_schedule): ops[0 .. 2] (the front, when it did not run fused) first
        for (size_t oi = first_op; oi < 3 && oi < h->ops.size(); ++oi)
            if (run_op(oi)) return 1;
        for (size_t li = 0; li < h->lat_sched.size(); ++li) {
            const std::vector<int> &L = h->lat_sched[li];
            if (!(h->ops[L[0]].kind == OP_CONV && h->ops[L[0]].sk)) { // (the pool, or a conv that does not fit the split-K kernel: one op, its own launch)
                if (run_op((size_t)L[0])) return 1;
                continue;
            }
            if (mark(1)) return 1;
            SkMember m[kSkGroupMax];
            for (size_t k = 0; k < L.size(); ++k) {
                const Op &op = h->ops[L[k]];
                m[k] = SkMember{conv_args(op), op.sk_atoms, op.sk_partial, op.sk_tickets};
                m[k].a.tile_w = 0;
                if (!conv_sk_eligible(m[k].a, h->split)) return fail("internal: conv " + std::to_string(L[k]) + " of the latency plan does not fit conv_sk_kernel");
                flops[1] += op_flops(op);
            }
            const long long key = ((long long)li << 24) | (long long)B;
            if (h->tune_pass >= 0) { // a timing pass of sk_autotune: this launch as its candidate number (pass mod candidates), between two events
                std::vector<SkChoice> &cands = h->sk_cands[key];
                if (cands.empty()) {
                    cands.resize(kSkMaxCandidates);
                    cands.resize((size_t)std::max(conv_sk_enumerate(m, (int)L.size(), h->split, h->num_cus, h->sk_force_tile, h->sk_force_form, cands.data(), kSkMaxCandidates), 0));
                    if (cands.empty()) return fail("internal: no launch candidate for level " + std::to_string(li));
                    h->tune_ms[key].assign(cands.size(), 1e30f);
                }
                SkChoice c = cands[(size_t)h->tune_pass % cands.size()];
                HIP_TRY(hipEventRecord(h->tune_ev[2 * li], main_st));
                HIP_TRY(launch_conv_sk_group(m, (int)L.size(), h->split, h->num_cus, h->sk_force_tile, h->sk_force_form, main_st, &c));
                HIP_TRY(hipEventRecord(h->tune_ev[2 * li + 1], main_st));
                h->tune_key[li] = key;
            } else {
                HIP_TRY(launch_conv_sk_group(m, (int)L.size(), h->split, h->num_cus, h->sk_force_tile, h->sk_force_form, main_st, &h->sk_choices[key]));
            }
            ++launches[1];
        }
    } else {
        for (size_t oi = first_op; oi < h->ops.size(); ++oi)
            if (run_op(oi)) return 1;
    }
    st = main_st;
    for (int i = 1; i < wtk_yolo::kSideStreams; ++i)
        if (side_used & (1u << i)) {
            HIP_TRY(hipEventRecord(h->side_done[i], h->side_stream[i]));
            HIP_TRY(hipStreamWaitEvent(main_st, h->side_done[i], 0));
        }
    if (mark(3)) return 1;
    if (run_head(h, B, H, W, conf, out_xywh, out_conf, out_anchor, st, nms)) return 1;
    ++launches[3];
    if (h->profiling) {
        if (nev < wtk_yolo::kProfEvents) {
            HIP_TRY(hipEventRecord(h->ev[nev], st));
            ev_class[nev] = -1;
            ++nev;
        }
        HIP_TRY(hipEventSynchronize(h->ev[nev - 1]));
        for (int i = 0; i + 1 < nev; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
            h->prof_ms[ev_class[i]] += ms;
        }
        for (int i = 0; i < wtk_yolo::kProfKernels; ++i) h->prof_launches[i] += launches[i], h->prof_flops[i] += flops[i];
    }
    return 0;
}

// Autotune of a latency-plan handle: the first EAGER forward pass at a batch size is preceded by timing passes — the same forward pass (real frames, real
// activations), every grouped launch between two events, launch i running its candidate number (pass mod candidates_i) — and every launch keeps the
// (tile, forms) that took the least time.  The cost model that otherwise decides is calibrated on a few layers of one network at one size; the choice
// enters no arithmetic (tests/test_gpu_latency.py: every tile and form gives the same bits), so timing noise can cost microseconds, never results.
static int sk_autotune(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float *out_xywh, float *out_conf,
                       int32_t *out_anchor, hipStream_t st, const ViewSrc *vs, const NmsOut *nms) {
    const size_t n = h->lat_sched.size();
    while (h->tune_ev.size() < 2 * n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->tune_ev.push_back(e);
    }
    h->tune_key.assign(n, -1);
    constexpr int kReps = 3;
    size_t most = 1;
    int rc = 0;
    for (size_t pass = 0; pass < most * kReps && !rc; ++pass) {
        h->tune_pass = (int)pass;
        rc = yolo_enqueue_pass(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st, vs, nms);
        h->tune_pass = -1;
        if (rc) break;
        if (hipStreamSynchronize(st) != hipSuccess) {
            rc = fail("sk_autotune: hipStreamSynchronize failed");
            break;
        }
        for (size_t li = 0; li < n; ++li) {
            const long long key = h->tune_key[li];
            if (key < 0) continue;
            const std::vector<SkChoice> &cands = h->sk_cands[key];
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->tune_ev[2 * li], h->tune_ev[2 * li + 1]) != hipSuccess) continue;
            float &best = h->tune_ms[key][pass % cands.size()];
            best = std::min(best, ms);
            most = std::max(most, cands.size());
        }
    }
    if (rc) return rc;
    for (size_t li = 0; li < n; ++li) {
        const long long key = h->tune_key[li];
        if (key < 0) continue;
        const std::vector<SkChoice> &cands = h->sk_cands[key];
        const std::vector<float> &ms = h->tune_ms[key];
        size_t b = 0;
        for (size_t i = 1; i < cands.size(); ++i)
            if (ms[i] < ms[b] * 0.97f) b = i; // (the model's choice unless another is clearly faster)
        h->sk_choices[key] = cands[b];
    }
    h->sk_tuned.push_back(B);
    return 0;
}

static int yolo_enqueue(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float *out_xywh, float *out_conf,
                        int32_t *out_anchor, hipStream_t st, const ViewSrc *vs = nullptr, const NmsOut *nms = nullptr) {
    if (h->latency && h->sk_group && h->sk_autotune && !h->lat_sched.empty() && !h->profiling && h->tune_pass < 0 &&
        std::find(h->sk_tuned.begin(), h->sk_tuned.end(), (int)B) == h->sk_tuned.end()) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) { // (a capture cannot be timed: it keeps the cost model's choices)
            if (sk_autotune(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st, vs, nms)) return 1;
        }
    }
    return yolo_enqueue_pass(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st, vs, nms);
}

// Captured launches carry the stream layout / dynamic-batch pointer they were captured with: drop them all.  Each exec is destroyed only after its last
// replay has finished (its own event; no device-wide synchronise: other lanes keep running, and a global-mode capture open in another thread stays
// legal); argument sets met once are forgotten too.
static void destroy_graph_entry(wtk_yolo::GraphEntry &g) {
    if (g.done) {
        (void)hipEventSynchronize(g.done);
        (void)hipEventDestroy(g.done);
    }
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    g.exec = nullptr, g.done = nullptr;
}
void wtk::drop_graphs(wtk_yolo *h) {
    for (auto &g : h->graphs) destroy_graph_entry(g);
    h->graphs.clear();
    h->seen_once.clear();
}

// Replay the captured forward pass of this argument set, or capture it now (the whole launch sequence incl. the side streams).
static int graph_replay_or_capture(wtk_yolo *h, wtk_yolo::GraphEntry key, hipStream_t st, const ViewSrc *vs) {
    for (auto &g : h->graphs)
        if (g.same_args(key)) {
            HIP_TRY(hipGraphLaunch(g.exec, st));
            HIP_TRY(hipEventRecord(g.done, st));
            return 0;
        }
    hipGraph_t graph = nullptr;
    const bool forks = !(h->latency && h->sk_group && !h->lat_sched.empty()) && h->use_side && h->side_streams > 0;
    if (forks && ensure_side_streams(h)) return 1; // streams and events exist before the capture starts
    std::unique_lock<std::recursive_mutex> capture_lock(g_side_mu); // no other thread may touch the shared side streams while they are captured
    // protocol: the origin and the streams the capture will fork into are outside any capture when it begins (a stream left inside one by a failed
    // bracket, here or in the caller's code, must not be captured again: fail loudly instead)
    if (stream_idle(st, "the stream a capture is about to begin on")) return 1;
    for (int i = 1; i < wtk_yolo::kSideStreams; ++i)
        if (h->side_stream[i] && stream_idle(h->side_stream[i], "a side stream")) return 1;
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = yolo_enqueue(h, reinterpret_cast<const uint8_t *>(key.frames), key.B, key.H, key.W, key.C, key.conf, reinterpret_cast<float *>(key.o_xywh),
                                reinterpret_cast<float *>(key.o_conf), reinterpret_cast<int32_t *>(key.o_anchor), st, vs);
    const hipError_t ec = hipStreamEndCapture(st, &graph);
    capture_lock.unlock();
    if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return 1;
    }
    if (ec != hipSuccess) return fail_hip("hipStreamEndCapture", ec);
    const hipError_t ei = hipGraphInstantiate(&key.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) return fail_hip("hipGraphInstantiate", ei);
    if (hipEventCreateWithFlags(&key.done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGraphExecDestroy(key.exec);
        return fail("hipEventCreateWithFlags failed");
    }
    if (h->graphs.size() >= 16) { // bounded cache: callers that rotate buffers would otherwise grow it without limit
        destroy_graph_entry(h->graphs.front());
        h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back(key);
    HIP_TRY(hipGraphLaunch(key.exec, st));
    HIP_TRY(hipEventRecord(key.done, st));
    return 0;
}

extern "C" int wtk_yolo_predict(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float iou,
                                int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_anchor, void *stream) {
    (void)iou; // with max_det == 1 the IoU threshold cannot change the survivor
    if (!h || !frames_dev || !out_xywh) return fail("wtk_yolo_predict: null argument");
    if (B <= 0) return fail("wtk_yolo_predict: empty batch (the reference asserts len(frames) > 0, yolo_controller.py:65)");
    if (B > h->max_batch) return fail("wtk_yolo_predict: batch exceeds max_batch");
    if (C != 1 && C != 3) return fail("wtk_yolo_predict: frames must have 1 (gray) or 3 (BGR) channels");
    if (max_det != 1) return fail("wtk_yolo_predict: max_det must be 1 (yolo_controller.py:76 hard-wires it); wtk_yolo_predict_nms takes max_det > 1");
    if (H <= 0 || W <= 0) return fail("wtk_yolo_predict: bad frame size");
    DEVICE_GUARD(h);
    hipStream_t st = (hipStream_t)stream;
    if ((H != h->S_h || W != h->S_w) && h->lb_cap == 0) { // letterbox staging image, allocated once
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMalloc(&h->lb_dev, (size_t)h->max_batch * h->S_h * h->S_w * 3));
        h->lb_cap = (size_t)h->max_batch * h->S_h * h->S_w * 3;
    }
    // Opt-in (WTK_GRAPH / WTK_GRAPH_HOST / WTK_GRAPH_VIEWS, see wtk_yolo_create_planned): replay a captured hipGraph of the forward pass.  The handle's own
    // staging buffers (the *_host entry points) never change address, so one capture per (B, H, W, C, conf) is replayed forever; a caller's argument set
    // is captured the second time it is met, so a caller that rotates its buffers never pays for a capture.
    const bool own_buffers = frames_dev == h->frames_dev && out_xywh == h->o_xywh;
    const bool use_graph = ((own_buffers && h->graph_host) || (!own_buffers && h->graph_views)) && st != nullptr && !h->profiling && h->graph_max_batch > 0 && B <= h->graph_max_batch;
    if (!use_graph) return yolo_enqueue(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st);
    wtk_yolo::GraphEntry key{frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, nullptr};
    if (!own_buffers) {
        bool known = false;
        for (auto &g : h->graphs) known = known || g.same_args(key);
        for (auto &g : h->seen_once) known = known || g.same_args(key);
        if (!known) {
            if (h->seen_once.size() >= 16) h->seen_once.erase(h->seen_once.begin());
            h->seen_once.push_back(key);
            return yolo_enqueue(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st);
        }
    }
    return graph_replay_or_capture(h, key, st, nullptr);
}

extern "C" int wtk_yolo_predict_nms(wtk_yolo *h, const uint8_t *frames_dev, int32_t B, int32_t H, int32_t W, int32_t C, float conf, float iou,
                                    int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_cls, int32_t *out_anchor, int32_t *out_count,
                                    void *stream) {
    if (!h || !frames_dev || !out_xywh) return fail("wtk_yolo_predict_nms: null argument");
    if (B <= 0) return fail("wtk_yolo_predict_nms: empty batch (the reference asserts len(frames) > 0, yolo_controller.py:65)");
    if (B > h->max_batch) return fail("wtk_yolo_predict_nms: batch exceeds max_batch");
    if (C != 1 && C != 3) return fail("wtk_yolo_predict_nms: frames must have 1 (gray) or 3 (BGR) channels");
    if (max_det < 1 || max_det > 30000) return fail("wtk_yolo_predict_nms: max_det must be in [1, 30000]");
    if (!(iou >= 0.f && iou <= 1.f)) return fail("wtk_yolo_predict_nms: iou must be in [0, 1]");
    if (H <= 0 || W <= 0) return fail("wtk_yolo_predict_nms: bad frame size");
    DEVICE_GUARD(h);
    hipStream_t st = (hipStream_t)stream;
    if ((H != h->S_h || W != h->S_w) && h->lb_cap == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMalloc(&h->lb_dev, (size_t)h->max_batch * h->S_h * h->S_w * 3));
        h->lb_cap = (size_t)h->max_batch * h->S_h * h->S_w * 3;
    }
    if (ensure_nms_scratch(h, st)) return 1;
    const NmsOut nms{iou, max_det, out_cls, out_count};
    return yolo_enqueue(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st, nullptr, &nms);
}

extern "C" int wtk_yolo_predict_views(wtk_yolo *h, const uint8_t *frames_dev, int32_t n_frames, int32_t H, int32_t W, int32_t C,
                                      const int32_t *frame_index_dev, const int32_t *pos_xy_dev, int32_t B, int32_t view_w, int32_t view_h, float conf,
                                      float iou, int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_anchor, void *stream) {
    (void)iou;
    if (!h || !frames_dev || !pos_xy_dev || !out_xywh) return fail("wtk_yolo_predict_views: null argument");
    if (B <= 0) return fail("wtk_yolo_predict_views: empty batch (the reference asserts len(frames) > 0, yolo_controller.py:65)");
    if (B > h->max_batch) return fail("wtk_yolo_predict_views: batch exceeds max_batch");
    if (C != 1 && C != 3) return fail("wtk_yolo_predict_views: frames must have 1 (gray) or 3 (BGR) channels");
    if (max_det != 1) return fail("wtk_yolo_predict_views: max_det must be 1 (yolo_controller.py:76 hard-wires it)");
    if (H <= 0 || W <= 0 || view_w <= 0 || view_h <= 0 || n_frames <= 0) return fail("wtk_yolo_predict_views: bad frame / view size");
    if (!frame_index_dev && B > n_frames) return fail("wtk_yolo_predict_views: without frame_index the batch rows are frames 0..B-1");
    DEVICE_GUARD(h);
    hipStream_t st = (hipStream_t)stream;
    if (h->lb_cap == 0) { // staging image of the network input, allocated once
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMalloc(&h->lb_dev, (size_t)h->max_batch * h->S_h * h->S_w * 3));
        h->lb_cap = (size_t)h->max_batch * h->S_h * h->S_w * 3;
    }
    const ViewSrc vs{pos_xy_dev, frame_index_dev, view_w, view_h, n_frames};
    // The reference's operating point is this call at B = 1 and B = one cycle (9 / 15 views), once per cycle each (yolo_controller.py:95-109).  With
    // WTK_GRAPH_VIEWS=1 a caller that comes back with the SAME device addresses (frames, view table, output rows — HipYoloController keeps them per
    // batch size) gets the captured forward replayed; an argument set is captured the second time it is met (a caller that rotates its buffers never
    // pays for a capture).  OFF by default: measured in round 4 (bench.py `closed_loop`, 384 x 384 views) the replay changes a B = 1 call from 1.13 to
    // 1.12 ms and a B = 15 call from 1.38 to 1.35 ms — these calls are bound by the ~60 dependent kernels' own latencies (18 us each on grids of a few
    // blocks), not by the host's launch rate — while a replay costs its fixed 10-16 us.
    const bool use_graph = h->graph_views && st != nullptr && !h->profiling && h->graph_max_batch > 0 && B <= h->graph_max_batch;
    if (use_graph) {
        wtk_yolo::GraphEntry key{frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, nullptr};
        key.idx = frame_index_dev, key.pos = pos_xy_dev, key.vw = view_w, key.vh = view_h, key.nf = n_frames;
        bool known = false;
        for (auto &g : h->graphs) known = known || g.same_args(key);
        for (auto &g : h->seen_once) known = known || g.same_args(key);
        if (known) return graph_replay_or_capture(h, key, st, &vs);
        if (h->seen_once.size() >= 16) h->seen_once.erase(h->seen_once.begin());
        h->seen_once.push_back(key);
    }
    return yolo_enqueue(h, frames_dev, B, H, W, C, conf, out_xywh, out_conf, out_anchor, st, &vs);
}

extern "C" int wtk_yolo_predict_host(wtk_yolo *h, const uint8_t *frames_host, int32_t B, int32_t H, int32_t W, int32_t C, float conf,
                                     float iou, int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_anchor) {
    if (!h || !frames_host || !out_xywh) return fail("wtk_yolo_predict_host: null argument");
    if (B <= 0) return fail("wtk_yolo_predict_host: empty batch (the reference asserts len(frames) > 0, yolo_controller.py:65)");
    if (B > h->max_batch) return fail("wtk_yolo_predict_host: batch exceeds max_batch");
    if (H <= 0 || W <= 0 || (C != 1 && C != 3)) return fail("wtk_yolo_predict_host: bad frame shape");
    DEVICE_GUARD(h);
    const size_t need = (size_t)B * H * W * C;
    if (need > h->frames_cap) {
        (void)hipFree(h->frames_dev);
        h->frames_dev = nullptr;
        h->frames_cap = 0;
        const size_t cap = std::max(need, (size_t)h->max_batch * H * W * C);
        HIP_TRY(hipMalloc(&h->frames_dev, cap));
        h->frames_cap = cap;
    }
    if (!h->host_stream && pooled_stream(h->device, &h->host_stream)) return 1;
    hipStream_t st = h->host_stream;
    HIP_TRY(hipMemcpyAsync(h->frames_dev, frames_host, need, hipMemcpyHostToDevice, st));
    if (wtk_yolo_predict(h, h->frames_dev, B, H, W, C, conf, iou, max_det, h->o_xywh, h->o_conf, h->o_anchor, st)) return 1;
    HIP_TRY(hipMemcpyAsync(out_xywh, h->o_xywh, sizeof(float) * 4 * B, hipMemcpyDeviceToHost, st));
    if (out_conf) HIP_TRY(hipMemcpyAsync(out_conf, h->o_conf, sizeof(float) * B, hipMemcpyDeviceToHost, st));
    if (out_anchor) HIP_TRY(hipMemcpyAsync(out_anchor, h->o_anchor, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int wtk_yolo_set_side_streams(wtk_yolo *h, int32_t n) {
    if (!h || n < 0 || n > 2) return fail("wtk_yolo_set_side_streams: n must be 0, 1 or 2");
    DEVICE_GUARD(h);
    drop_graphs(h); // captured launches (host stream or a caller's) carry the old stream layout
    h->side_streams = n;
    h->use_side = n > 0;
    return 0;
}

extern "C" int wtk_yolo_set_dynamic_batch(wtk_yolo *h, const int32_t *n_dev) {
    if (!h) return fail("wtk_yolo_set_dynamic_batch: null handle");
    h->n_dyn = n_dev;
    if (!h->graphs.empty() || !h->seen_once.empty()) { // captured launches carry the old pointer
        DEVICE_GUARD(h);
        drop_graphs(h);
    }
    return 0;
}

extern "C" int wtk_yolo_margin_buffer(wtk_yolo *h, const float **margins_dev) {
    if (!h || !margins_dev) return fail("wtk_yolo_margin_buffer: null argument");
    *margins_dev = h->o_margin;
    return 0;
}

extern "C" int wtk_yolo_last_margins_host(wtk_yolo *h, int32_t B, float *margins_host) {
    if (!h || !margins_host || B <= 0 || B > h->max_batch) return fail("wtk_yolo_last_margins_host: bad argument");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(margins_host, h->o_margin, sizeof(float) * B, hipMemcpyDeviceToHost));
    return 0;
}

static void to_f32(const void *src, float *dst, size_t n, int is_f16) {
    if (!is_f16) {
        std::memcpy(dst, src, n * 4);
        return;
    }
    const _Float16 *s = reinterpret_cast<const _Float16 *>(src);
    for (size_t i = 0; i < n; ++i) dst[i] = (float)s[i];
}

extern "C" int wtk_yolo_debug_head(wtk_yolo *h, int32_t level, int32_t B, float *box_host, float *cls_host) {
    if (!h || level < 0 || level > 2 || B <= 0 || B > h->max_batch) return fail("wtk_yolo_debug_head: bad argument");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t A = (size_t)h->lh[level] * h->lw[level];
    if (box_host) {
        const size_t n = (size_t)B * A * 64;
        HIP_TRY(hipMemcpy(box_host, h->bufs[h->box_buf[level]].ptr, n * 4, hipMemcpyDeviceToHost)); // fp32 in both modes
    }
    if (cls_host) {
        const size_t n = (size_t)B * A * h->cls_ld;
        std::vector<float> full(n);
        HIP_TRY(hipMemcpy(full.data(), h->bufs[h->cls_buf[level]].ptr, n * 4, hipMemcpyDeviceToHost)); // fp32 in both modes
        for (size_t i = 0; i < (size_t)B * A; ++i)
            for (int k = 0; k < h->dims.nc; ++k) cls_host[i * h->dims.nc + k] = full[i * h->cls_ld + k];
    }
    return 0;
}

extern "C" int wtk_yolo_debug_tensor(wtk_yolo *h, int32_t conv_index, int32_t B, float *out_host, size_t out_cap, int32_t *shape_hwc) {
    if (!h || B <= 0 || B > h->max_batch) return fail("wtk_yolo_debug_tensor: bad argument");
    const Op *op = nullptr;
    for (const Op &o : h->ops)
        if (o.spec == conv_index && o.out_buf >= 0) op = &o;
    if (!op) return fail("wtk_yolo_debug_tensor: no op computes conv " + std::to_string(conv_index));
    const Buf &b = h->bufs[op->out_buf];
    if (shape_hwc) shape_hwc[0] = b.h, shape_hwc[1] = b.w, shape_hwc[2] = op->cout;
    if (!out_host) return 0;
    const size_t px = (size_t)B * b.h * b.w;
    if (out_cap < px * op->cout) return fail("wtk_yolo_debug_tensor: output buffer too small");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<char> tmp(px * b.C * (b.f32 ? 4 : h->esize));
    HIP_TRY(hipMemcpy(tmp.data(), b.ptr, tmp.size(), hipMemcpyDeviceToHost));
    std::vector<float> full(px * b.C);
    if (h->split && !b.f32) {
        const _Float16 *sp = reinterpret_cast<const _Float16 *>(tmp.data());
        for (size_t i = 0; i < px; ++i)
            for (int c = 0; c < b.C; ++c) {
                const size_t o = i * 2 * b.C + 64 * (c >> 5) + (c & 31);
                full[i * b.C + c] = (float)sp[o] + (float)sp[o + 32] * kSplitInv;
            }
    } else {
        to_f32(tmp.data(), full.data(), full.size(), b.f32 ? 0 : h->is_f16);
    }
    const float unscale = op->act ? 1.0f / kActScale : 1.0f; // SiLU outputs are stored log2(e)-scaled
    for (size_t i = 0; i < px; ++i)
        for (int k = 0; k < op->cout; ++k) out_host[i * op->cout + k] = full[i * b.C + op->out_coff + k] * unscale;
    return 0;
}

static int upload_head_logits(wtk_yolo *h, const float *box_host, const float *cls_host, int32_t B);

extern "C" int wtk_yolo_decode_nms_host(wtk_yolo *h, const float *box_host, const float *cls_host, int32_t B, int32_t H, int32_t W, float conf, float iou,
                                        int32_t max_det, float *out_xywh, float *out_conf, int32_t *out_cls, int32_t *out_anchor, int32_t *out_count) {
    if (!h || !box_host || !cls_host || !out_xywh || B <= 0 || B > h->max_batch || max_det < 1) return fail("wtk_yolo_decode_nms_host: bad argument");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    if (upload_head_logits(h, box_host, cls_host, B)) return 1;
    if (ensure_nms_scratch(h, nullptr)) return 1;
    const size_t rows = (size_t)B * max_det;
    float *d_xywh = nullptr, *d_conf = nullptr;
    int *d_cls = nullptr, *d_anchor = nullptr, *d_count = nullptr;
    hipError_t e = hipSuccess;
    if ((e = hipMalloc(&d_xywh, rows * 16)) != hipSuccess || (e = hipMalloc(&d_conf, rows * 4)) != hipSuccess || (e = hipMalloc(&d_cls, rows * 4)) != hipSuccess ||
        (e = hipMalloc(&d_anchor, rows * 4)) != hipSuccess || (e = hipMalloc(&d_count, (size_t)B * 4)) != hipSuccess) {
        (void)hipFree(d_xywh), (void)hipFree(d_conf), (void)hipFree(d_cls), (void)hipFree(d_anchor), (void)hipFree(d_count);
        return fail_hip("wtk_yolo_decode_nms_host: hipMalloc", e);
    }
    const NmsOut nms{iou, max_det, d_cls, d_count};
    int rc = run_head(h, B, H, W, conf, d_xywh, d_conf, d_anchor, nullptr, &nms);
    if (!rc) {
        if ((e = hipMemcpy(out_xywh, d_xywh, rows * 16, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail_hip("wtk_yolo_decode_nms_host: copy", e);
        if (!rc && out_conf && (e = hipMemcpy(out_conf, d_conf, rows * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail_hip("copy", e);
        if (!rc && out_cls && (e = hipMemcpy(out_cls, d_cls, rows * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail_hip("copy", e);
        if (!rc && out_anchor && (e = hipMemcpy(out_anchor, d_anchor, rows * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail_hip("copy", e);
        if (!rc && out_count && (e = hipMemcpy(out_count, d_count, (size_t)B * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail_hip("copy", e);
    }
    (void)hipFree(d_xywh), (void)hipFree(d_conf), (void)hipFree(d_cls), (void)hipFree(d_anchor), (void)hipFree(d_count);
    return rc;
}

extern "C" int wtk_yolo_decode_host(wtk_yolo *h, const float *box_host, const float *cls_host, int32_t B, int32_t H, int32_t W, float conf,
                                    float *out_xywh, float *out_conf, int32_t *out_anchor) {
    if (!h || !box_host || !cls_host || !out_xywh || B <= 0 || B > h->max_batch) return fail("wtk_yolo_decode_host: bad argument");
    DEVICE_GUARD(h);
    HIP_TRY(hipDeviceSynchronize());
    if (upload_head_logits(h, box_host, cls_host, B)) return 1;
    if (run_head(h, B, H, W, conf, h->o_xywh, h->o_conf, h->o_anchor, nullptr)) return 1;
    HIP_TRY(hipMemcpy(out_xywh, h->o_xywh, sizeof(float) * 4 * B, hipMemcpyDeviceToHost));
    if (out_conf) HIP_TRY(hipMemcpy(out_conf, h->o_conf, sizeof(float) * B, hipMemcpyDeviceToHost));
    if (out_anchor) HIP_TRY(hipMemcpy(out_anchor, h->o_anchor, sizeof(int) * B, hipMemcpyDeviceToHost));
    return 0;
}

// scatter concatenated [B][A][.] fp32 logits into the per-level head buffers (storage dtype): the test hook behind the two
// decode entry points
static int upload_head_logits(wtk_yolo *h, const float *box_host, const float *cls_host, int32_t B) {
    // scatter the concatenated [B][A][.] logits into the per-level head buffers (storage dtype)
    const int A = h->anchors;
    size_t a0 = 0;
    for (int l = 0; l < 3; ++l) {
        const size_t Al = (size_t)h->lh[l] * h->lw[l];
        std::vector<float> bx((size_t)B * Al * 64), cl((size_t)B * Al * h->cls_ld, 0.f);
        for (int n = 0; n < B; ++n)
            for (size_t j = 0; j < Al; ++j) {
                std::memcpy(&bx[((size_t)n * Al + j) * 64], &box_host[((size_t)n * A + a0 + j) * 64], 64 * sizeof(float));
                for (int k = 0; k < h->dims.nc; ++k) cl[((size_t)n * Al + j) * h->cls_ld + k] = cls_host[((size_t)n * A + a0 + j) * h->dims.nc + k];
            }
        HIP_TRY(hipMemcpy(h->bufs[h->box_buf[l]].ptr, bx.data(), bx.size() * 4, hipMemcpyHostToDevice)); // fp32 in both modes
        HIP_TRY(hipMemcpy(h->bufs[h->cls_buf[l]].ptr, cl.data(), cl.size() * 4, hipMemcpyHostToDevice));
        a0 += Al;
    }
    return 0;
}

