// Internal declarations shared by the HOST-side translation units of libwtk_hip.so (round 6: csrc/wtk_api.hip was one file of 2 500 lines):
//   wtk_api.hip     errors, versions, ResMLP / track / recheck / crop entry points
//   wtk_plan.hip    the detector handle: model table, weight packing, graph planning (channel-slice views), launch schedule of the latency plan,
//                   stream pool, status page, create / destroy
//   wtk_run.hip     one forward pass: enqueue, opt-in graph replay, the predict entry points, test hooks
//   wtk_hybrid.hip  the look-twice detector composed from the entry points above
// Kernel-side declarations live in wtk_kernels.h.
#pragma once
#include "../../include/wtk_hip.h"
#include "wtk_kernels.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace wtk {
// thread-local error message of the C ABI (wtk_last_error); both return 1
int fail(const std::string &msg);
int fail_hip(const char *what, hipError_t e);
int ensure_attributes(int device); // kernel attributes, once per device
}
#define HIP_TRY(expr)                                                                                                          \
    do {                                                                                                                       \
        hipError_t _e = (expr);                                                                                                \
        if (_e != hipSuccess) return wtk::fail_hip(#expr, _e);                                                                 \
    } while (0)

// Stream entry points launch on the handle's device whatever the caller's current device is, and leave the caller's
// current device as they found it (PyTorch tracks the same thread-local HIP state).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define DEVICE_GUARD(h)                                                                                                        \
    DeviceGuard _guard((h)->device);                                                                                           \
    if (_guard.err != hipSuccess) return wtk::fail_hip("selecting the handle's device", _guard.err)

inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f; // round-to-nearest-even, host compiler builtin
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

inline float f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}


// ---------------------------------------------------------------------------------------------
// the detector handle
// ---------------------------------------------------------------------------------------------
struct ConvSpec {
    std::string name;
    int cout, cin, k, stride, act;
};

struct ModelDims {
    int c[5];  // channel widths of P1..P5
    int n[4];  // C2f repeats of layers 2,4,6,8
    int hb, hc; // Detect hidden widths (box tower, cls tower)
    int nc;
};

struct Buf {
    size_t elems_per_image = 0; // h*w*C
    int h = 0, w = 0, C = 0;
    int f32 = 0; // 1: stored as fp32 whatever the handle's dtype (the Detect outputs: head logits are never rounded to fp16)
    void *ptr = nullptr;
};

enum OpKind { OP_STEM, OP_CONV, OP_POOL };

struct Op {
    OpKind kind;
    // conv
    int in_buf = -1, in_coff = 0, cin = 0;
    int out_buf = -1, out_coff = 0;
    int out2_buf = -1, out2_coff = 0;
    int res_buf = -1, res_coff = 0;
    int tail_op = -1; // index of a 1x1 op (64 -> 64, no activation) computed in this op's epilogue (conv3x3_halo fused tail)
    int folded = 0;   // 1: this op runs inside another op's kernel
    int in2_buf = -1, in2_coff = 0, in2_split = 0; // half-resolution source of the first in2_split input channels (ConvArgs::in2)
    int cout = 0, cout_pad = 0, k = 1, stride = 1, act = 1;
    int cfg = 0;
    int K = 0, Kpad = 0;
    int tile_w = 0;
    int halo = 0; // 1: conv3x3_halo kernel, 2: conv3x3_c32 kernel
    int side = 0;        // 1: runs on the handle's side stream (Detect towers of P3 / P4)
    int wait_feat = -1;  // side ops: feature event (0: P3 ready, 1: P4 ready) to wait for before the first one
    int signal_feat = -1; // main ops: record this feature event after the op
    void *w = nullptr; // packed device weights
    float *bias = nullptr;
    double macs_per_image = 0;
    int spec = -1; // index of the (first) conv blob this op computes, for wtk_yolo_debug_tensor
    int sk = 0;                  // latency plan: this conv runs on conv_sk_kernel (split-K implicit GEMM, conv_sk.hip)
    int sk_atoms = 0;            // ... with this many K atoms (conv_sk_slices(nk), or conv_sk_plan_atoms on a small throughput-plan handle)
    float *sk_partial = nullptr; // ... and this is its slab scratch ([slices][max_batch * ho * wo][cout_pad] fp32; null: one slice)
    unsigned *sk_tickets = nullptr; // ... and the arrival counters of its tiles (zero between launches; null: one slice, or WTK_SK_FINISH=1)
};


struct wtk_yolo {
    int device = 0;
    int is_f16 = 1;
    int esize = 2;
    // WTK_F16X3: split-fp16 storage (wtk_kernels.h, kSplitScale).  Planned like the fp32 mode (is_f16 = 0, esize = 4: a split tensor
    // takes the same 4 bytes per value), launched on the SPLIT instantiations of the fp16 kernels with pseudo-channel arguments.
    int split = 0;
    const int *n_dyn = nullptr; // wtk_yolo_set_dynamic_batch: device-side count of the batch rows that matter (<= B of the call)
    int S_h = 0, S_w = 0, max_batch = 0;
    ModelDims dims;
    std::vector<Buf> bufs;
    std::vector<Op> ops;
    std::vector<std::pair<void *, size_t>> dev_allocs; // (pointer, bytes) of every dev_alloc
    int box_buf[3] = {-1, -1, -1}, cls_buf[3] = {-1, -1, -1};
    int lh[3] = {0, 0, 0}, lw[3] = {0, 0, 0};
    int cls_ld = 32;
    double macs_per_frame = 0;
    int anchors = 0;
    // staging for the host entry points and for letterboxing
    uint8_t *frames_dev = nullptr;
    size_t frames_cap = 0;
    uint8_t *lb_dev = nullptr;
    size_t lb_cap = 0;
    void *zero_page = nullptr;
    float *o_xywh = nullptr, *o_conf = nullptr;
    float *o_margin = nullptr; // decision margin of every frame of the last max_det = 1 call (wtk_yolo_last_margins_host / wtk_yolo_margin_buffer)
    int *o_anchor = nullptr;
    // scratch of the general NMS (max_det > 1), allocated at its first use
    float *nms_score = nullptr, *nms_box = nullptr;
    int *nms_cls = nullptr;
    // profiling
    int use_halo = 1;
    int front_debug = 0; // WTK_FRONT_DEBUG=1: the fused front also writes the model.0 / model.1 tensors (test hook)
    int use_tail = 1; // WTK_NO_FUSED_TAIL=1: Detect box.2 as its own launch (A/B switch)
    int use_tail_cls_split = 1; // WTK_NO_SPLIT_CLS_TAIL=1: f16x3 handles launch the class towers' last 1x1 on its own (A/B switch; fp16 handles: WTK_NO_FUSED_TAIL)
    int halo_small_blocks = 1; // WTK_HALO_SMALL_BLOCKS=0: always 256-pixel blocks (A/B switch)
    int halo_persist = 1; // WTK_HALO_PERSIST=0: one tile per block (A/B switch)
    int halo_slabs = 3; // WTK_HALO_SLABS=2: two-slab / vmcnt(0) schedule of conv3x3_halo_kernel (A/B switch)
    int use_c32s = 1;  // WTK_NO_C32S=1: the 32 -> 32 channel 3x3 layers of a split (f16x3) handle through conv_igemm_kernel (A/B switch)
    int use_s2win = 1; // WTK_NO_S2WIN=1: strided 3x3 convs through conv_igemm_kernel instead of the parity-plane window kernel (A/B switch)
    int use_ws64 = 1;  // WTK_NO_WS64=1: 64 -> 64 channel 3x3 layers through conv3x3_halo_kernel instead of the weight-stationary kernel (A/B switch)
    int use_wide = 1;  // WTK_NO_WIDE_1X1=1: every 1x1 conv through conv_igemm_kernel (A/B switch)
    int use_c2f = 0;   // ops[3..5] (model.2.m.0.cv1, m.0.cv2, model.2.cv2) run as ONE fused kernel (c2f_fused.hip)
    int use_front = 0; // ops[0..2] (stem, model.1, model.2.cv1) run as ONE fused kernel (front_fused.hip)
    int num_cus = 0;
    // Latency plan (small batches: the reference's own operating point, one B = cycle_frame_num call and one B = 1 call per cycle,
    // yolo_controller.py:96-98,108-109).  Chosen when the handle is created — max_batch <= 4 and a reference-precision dtype, WTK_LATENCY_PLAN=0/1, or the caller's word (wtk_yolo_create_planned) —
    // and NOT per call: every conv behind the fused front then runs on conv_sk_kernel whatever the batch of the call, so a frame's logits do not
    // depend on the batch it arrives in.  The Detect towers' 1x1 tails are launches of their own in this plan.
    int latency = 0;
    // latency plan, round 6: the convs of one dependency level run as ONE grouped split-K launch on the caller's stream (sk_schedule).
    int sk_group = 1;                      // WTK_SK_GROUP=0: one launch per conv, in op order (test hook: the grouped launches must give the same bits)
    int sk_force_tile = -1, sk_force_form = -1; // WTK_SK_TILE / WTK_SK_FORM, read when the handle is created (test hooks: every tile and form gives the same bits)
    std::map<long long, wtk::SkChoice> sk_choices; // (launch or op, batch) -> what the split-K cost model chose (it runs once per key, not per call)
    // autotune of the latency plan (wtk_run.hip: sk_autotune): the first eager call at a batch size times every (tile, forms) candidate of every launch inside the
    // real forward pass and keeps the fastest; neither enters the arithmetic.  WTK_SK_AUTOTUNE=0: the cost model's choice.
    int sk_autotune = 1;
    int tune_pass = -1;                                     // >= 0: a timing pass is being enqueued (candidate = pass % candidates of the launch)
    std::map<long long, std::vector<wtk::SkChoice>> sk_cands; // (launch, batch) -> candidates, [0] = the cost model's
    std::map<long long, std::vector<float>> tune_ms;         // ... and the best time seen of each
    std::vector<hipEvent_t> tune_ev;                        // two per launch of lat_sched
    std::vector<long long> tune_key;                        // key of the launch timed through tune_ev[2 i], [2 i + 1] in this pass (-1: none)
    std::vector<int> sk_tuned;                              // batch sizes that have been tuned
    std::vector<std::vector<int>> lat_sched; // launches behind ops[0..2] in order: one op, or up to kSkGroupMax split-K ops that do not depend on each other
    int small_narrow = 0; // a small handle (max_batch <= 16, f16x3) runs window / implicit-GEMM layers whose grid leaves most CUs idle on 64-cout tiles (WTK_SMALL_NARROW=0: off)
    int halo_deep = 0;    // f16x3: the 64-cout x 128-pixel window tiles on the six-slab ring (small handles; WTK_HALO_DEEP)
    int *status_host = nullptr; // pinned, device-visible: sticky run-time flags written by the head kernels (wtk_yolo_status); a slot of the process-wide page
    int *status_dev = nullptr;  // ... and the device's address of the same word
    int status_static = 0;      // flags fixed at create time (none today)
    int profiling = 0;
    // kernel ids of the profile: 0 stem, 1 conv_igemm, 2 pool, 3 head, 4 conv3x3_halo (+ fused tails), 5 fused front / C2f tail,
    // 6 conv3x3_c32; the public class 1 ("conv") of wtk_yolo_get_profile is the sum of 1, 4, 5, 6
    static constexpr int kProfKernels = 7, kProfEvents = 96;
    hipEvent_t ev[kProfEvents];
    // concurrency: the P3 / P4 Detect towers run on a side stream next to the PAN path
    // Side streams of one forward pass (op.side = index, 0 = the caller's stream): 1 / 2 = P3 / P4 Detect towers (they only need t15 / t18).  The pair is
    // process-wide (ensure_side_streams); wtk_yolo_set_side_streams(1) folds both towers onto stream 1, (0) keeps everything on the caller's stream.
    static constexpr int kSideStreams = 3;
    hipStream_t side_stream[kSideStreams] = {};
    hipEvent_t feat_ev[2] = {nullptr, nullptr}, side_done[kSideStreams] = {};
    int use_side = 1;
    int side_streams = 2;
    // launch-bound regime (small batches): the whole forward is captured once per argument set and replayed
    struct GraphEntry {
        const void *frames;
        int B, H, W, C;
        float conf;
        void *o_xywh, *o_conf, *o_anchor;
        hipGraphExec_t exec;
        hipEvent_t done = nullptr; // recorded behind every replay: waited for before the exec is destroyed (a replay may still be in flight; the handle's OWN event, because
                                   // the stream of the last replay is the caller's and may be gone by then)
        // views form (wtk_yolo_predict_views): the view table's device addresses and the view shape are part of the key
        const void *idx = nullptr, *pos = nullptr;
        int vw = 0, vh = 0, nf = 0;
        bool same_args(const GraphEntry &o) const {
            return frames == o.frames && B == o.B && H == o.H && W == o.W && C == o.C && conf == o.conf && o_xywh == o.o_xywh && o_conf == o.o_conf &&
                   o_anchor == o.o_anchor && idx == o.idx && pos == o.pos && vw == o.vw && vh == o.vh && nf == o.nf;
        }
    };
    std::vector<GraphEntry> graphs;
    std::vector<GraphEntry> seen_once; // caller-buffer argument sets met once (exec == nullptr): captured when they come back
    int graph_max_batch = 16; // WTK_GRAPH_MAX_BATCH; 0 disables
    int graph_host = 0;       // WTK_GRAPH=1 / WTK_GRAPH_HOST=1: the *_host entry points replay captures (own staging buffers)
    int graph_views = 0;      // WTK_GRAPH=1 / WTK_GRAPH_VIEWS=1: caller-buffer argument sets are captured when they come back, then replayed
    hipStream_t host_stream = nullptr; // stream of the *_host entry points (graph capture needs a non-null stream)
    int ev_created = 0;
    double prof_ms[kProfKernels] = {};
    double prof_flops[kProfKernels] = {};
    long long prof_launches[kProfKernels] = {};
};

namespace wtk {
// stream pool and its protocol checks (wtk_plan.hip)
int stream_idle(hipStream_t s, const char *what);
int pooled_stream(int device, hipStream_t *s);
void unpool_stream(int device, hipStream_t s);
// captured launches of a handle (wtk_run.hip)
void drop_graphs(wtk_yolo *h);
}
