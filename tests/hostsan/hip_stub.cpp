// TEST INFRASTRUCTURE (tests/hostsan): a host-only stand-in for libamdhip64 — the "no-op launch layer" behind which the HOST side of libwtk_hip.so
// (planning, weight packing, launchers, the stream / event / graph lifetime protocol of csrc/wtk_plan.hip / wtk_run.hip) runs in the build container under
// AddressSanitizer + UndefinedBehaviorSanitizer.  Nothing here is shipped or measured; no kernel runs.
//
// What it models, and what it reports as a violation (stderr line + counter read by the driver through stub_violations()):
//  * device memory: every hipMalloc is a tracked region; hipMemcpy* / hipMemset* ranges must lie inside ONE live region (large regions are PROT_NONE
//    reservations: a stray host access faults); hipFree of an unknown / already freed pointer;
//  * streams, events, graphs and graph execs are heap objects with a magic word: use after destroy is caught by ASan (freed memory) or by the magic;
//  * stream capture as CUDA / HIP define it: one origin, forks through hipStreamWaitEvent on an event recorded inside the capture, every fork joined
//    before hipStreamEndCapture; calls that are illegal for a capturing thread (hipMalloc, hipFree, synchronisations, synchronous copies) while a
//    thread-local capture is open; a wait on an event whose last record belongs to a capture that has ended; hipEventDestroy of an event recorded in an
//    OPEN capture; hipGraphLaunch / eager work on a capturing stream from outside; begin-capture on a busy stream;
//  * launches: the kernel must have been registered, grid and block dimensions positive and within the device's limits, dynamic LDS within the
//    function's attribute; pointers of the known argument structs (ConvArgs, HaloArgs, ...) are checked by tests/hostsan/launch_checks.inc.
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

std::mutex &mu_ref() { // the stub's own state; the library may call from several threads (constructed on first use: hipcc's module constructors run first)
    static std::mutex *m = new std::mutex();
    return *m;
}
#define g_mu mu_ref()
int g_violations = 0;
bool g_verbose = false;

void violation(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::fprintf(stderr, "hip_stub VIOLATION: ");
    std::vfprintf(stderr, fmt, ap);
    std::fprintf(stderr, "\n");
    va_end(ap);
    ++g_violations;
    if (const char *e = std::getenv("WTK_STUB_ABORT"))
        if (e[0] == '1') std::abort();
}

// ---- device memory ------------------------------------------------------------------------------------------------------------
struct Region {
    size_t bytes;
    bool reserved; // PROT_NONE reservation (no host access possible)
    uint64_t id;   // unique per allocation: a captured launch names the ALLOCATION it was captured with, not just an address
};
uint64_t g_next_region = 1;
std::set<uint64_t> g_live_region_ids;
std::map<uintptr_t, Region> g_regions;
constexpr size_t kReserveAbove = 8u << 20; // regions above 8 MiB are address-space reservations
size_t g_live_bytes = 0, g_peak_bytes = 0;

const std::pair<const uintptr_t, Region> *find_region(const void *p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto it = g_regions.upper_bound(a);
    if (it == g_regions.begin()) return nullptr;
    --it;
    if (a >= it->first + it->second.bytes && !(it->second.bytes == 0 && a == it->first)) return nullptr;
    return &*it;
}

// a device-side range of a copy / memset: inside one live region.  Returns 1 when the bytes can be touched by the host (plain heap region).
int check_dev_range(const void *p, size_t n, const char *what) {
    if (n == 0) return 0;
    const auto *r = find_region(p);
    if (!r) {
        violation("%s: %p (+%zu) is not inside a live device allocation", what, p, n);
        return 0;
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (a + n > r->first + r->second.bytes) {
        violation("%s: [%p, +%zu) runs %zu bytes past the end of its device allocation (%zu bytes)", what, p, n, a + n - (r->first + r->second.bytes), r->second.bytes);
        return 0;
    }
    return r->second.reserved ? 0 : 1;
}

// ---- streams / events / graphs --------------------------------------------------------------------------------------------------
constexpr uint32_t kStreamMagic = 0x5354524du, kEventMagic = 0x45564e54u, kGraphMagic = 0x47525048u, kExecMagic = 0x45584543u, kDead = 0xdeadbeefu;

struct Capture {
    uint64_t id;
    struct StubStream *origin;
    std::thread::id thread;
    int mode;
    int nodes = 0;
    std::set<struct StubStream *> forks; // streams pulled in (origin excluded)
    bool invalid = false;
    std::set<uint64_t> regions; // device allocations the captured launches point into
};

struct StubStream {
    uint32_t magic = kStreamMagic;
    int device = 0;
    unsigned flags = 0;
    Capture *cap = nullptr; // open capture this stream belongs to
    uint64_t ops = 0;       // work items enqueued (eager or captured)
    uint64_t joined_ops = 0; // forks: value of `ops` covered by the last join into the origin's side
    uint64_t last_eager_op = 0;
};
struct StubEvent {
    uint32_t magic = kEventMagic;
    unsigned flags = 0;
    bool recorded = false;
    uint64_t capture_id = 0;        // != 0: the last record happened inside this capture
    StubStream *capture_stream = nullptr;
    uint64_t stream_ops = 0;        // `ops` of the recording stream at the record
};
struct StubGraph {
    uint32_t magic = kGraphMagic;
    int nodes = 0;
    std::set<uint64_t> regions;
};
struct StubExec {
    uint32_t magic = kExecMagic;
    int nodes = 0;
    uint64_t launches = 0;
    std::set<uint64_t> regions;
};

std::set<StubStream *> g_streams;
std::set<StubEvent *> g_events;
std::set<StubExec *> g_execs;
std::map<uint64_t, Capture *> g_captures; // open captures by id
uint64_t g_next_capture = 1;
StubStream g_null_stream; // the legacy default stream
int g_cur_device = 0;
thread_local int t_device = 0;
thread_local hipError_t t_last_error = hipSuccess;

StubStream *S(hipStream_t s, const char *what) {
    if (!s) return &g_null_stream;
    StubStream *p = reinterpret_cast<StubStream *>(s);
    if (!g_streams.count(p)) {
        violation("%s: stream %p is not a live stream", what, (void *)s);
        return nullptr;
    }
    if (p->magic != kStreamMagic) violation("%s: stream %p has a corrupted header", what, (void *)s);
    return p;
}
StubEvent *E(hipEvent_t e, const char *what) {
    StubEvent *p = reinterpret_cast<StubEvent *>(e);
    if (!p || !g_events.count(p)) {
        violation("%s: event %p is not a live event (destroyed or never created)", what, (void *)e);
        return nullptr;
    }
    return p;
}

// the calling thread has a thread-local (or global) capture open: synchronising / allocating calls are illegal for it
Capture *capture_of_this_thread() {
    for (auto &kv : g_captures)
        if (kv.second->thread == std::this_thread::get_id() && kv.second->mode != hipStreamCaptureModeRelaxed) return kv.second;
    return nullptr;
}
void unsafe_call(const char *what) {
    if (Capture *c = capture_of_this_thread()) {
        violation("%s called by a thread with an open stream capture (id %llu): prohibited, invalidates the capture", what, (unsigned long long)c->id);
        c->invalid = true;
    }
}

hipError_t ret(hipError_t e) {
    if (e != hipSuccess) t_last_error = e;
    return e;
}

// enqueue of one work item on a stream (launch, async copy, memset, graph launch)
void enqueue(StubStream *s, const char *what) {
    ++s->ops;
    if (s->cap) {
        ++s->cap->nodes;
        if (s->cap->thread != std::this_thread::get_id() && s->cap->mode == hipStreamCaptureModeThreadLocal)
            violation("%s on stream %p from a thread other than the one that captures it (thread-local capture %llu)", what, (void *)s, (unsigned long long)s->cap->id);
    } else {
        s->last_eager_op = s->ops;
    }
}

// ---- kernels ------------------------------------------------------------------------------------------------------------------
struct KernelInfo {
    std::string name;
    int max_dyn_lds = 64 * 1024;
    uint64_t launches = 0;
};
std::map<const void *, KernelInfo> &kernels_ref() {
    static auto *m = new std::map<const void *, KernelInfo>();
    return *m;
}
#define g_kernels kernels_ref()
struct CallConfig {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
};
thread_local std::vector<CallConfig> t_configs;

} // namespace

// launch_checks.inc: argument-struct checks of the library's own kernels (needs the region table above)
namespace stubchk {
bool dev_ptr_ok(const void *p, size_t bytes, const char *kernel, const char *field) {
    if (!p) {
        violation("%s: argument %s is null", kernel, field);
        return false;
    }
    const auto *r = find_region(p);
    if (!r) {
        violation("%s: argument %s = %p is not inside a live device allocation", kernel, field, p);
        return false;
    }
    if (Capture *c = capture_of_this_thread()) c->regions.insert(r->second.id);
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (a + bytes > r->first + r->second.bytes) {
        violation("%s: argument %s = %p needs %zu bytes but its allocation ends %zu bytes earlier (allocation %zu bytes)", kernel, field, p, bytes,
                  a + bytes - (r->first + r->second.bytes), r->second.bytes);
        return false;
    }
    return true;
}
void fail(const char *kernel, const char *msg) { violation("%s: %s", kernel, msg); }
} // namespace stubchk
#include "launch_checks.inc"

extern "C" {

int stub_violations(void) { return g_violations; }
size_t stub_peak_device_bytes(void) { return g_peak_bytes; }
size_t stub_live_device_bytes(void) { return g_live_bytes; }
int stub_live_streams(void) { return (int)g_streams.size(); }
int stub_live_events(void) { return (int)g_events.size(); }
int stub_live_execs(void) { return (int)g_execs.size(); }
int stub_open_captures(void) { return (int)g_captures.size(); }
uint64_t stub_kernel_launches(const char *substr) {
    std::lock_guard<std::mutex> lk(g_mu);
    uint64_t n = 0;
    for (auto &kv : g_kernels)
        if (!substr || kv.second.name.find(substr) != std::string::npos) n += kv.second.launches;
    return n;
}
void stub_set_verbose(int v) { g_verbose = v != 0; }

// ---- registration (what hipcc's host stubs call at load time) ---------------------------------------------------------------
void **__hipRegisterFatBinary(const void *) {
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned, void *, void *, void *, void *, int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[hostFunction].name = deviceName ? deviceName : "?";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_configs.push_back({grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    if (t_configs.empty()) return hipErrorInvalidValue;
    const CallConfig c = t_configs.back();
    t_configs.pop_back();
    *grid = c.grid, *block = c.block, *shmem = c.shmem, *stream = c.stream;
    return hipSuccess;
}

// ---- devices / errors --------------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = t_device;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) {
    if (d != 0) return ret(hipErrorInvalidDevice);
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device) {
    if (device != 0) return ret(hipErrorInvalidDevice);
    std::memset(prop, 0, sizeof(*prop));
    std::snprintf(prop->name, sizeof(prop->name), "hip_stub gfx950");
    std::snprintf(prop->gcnArchName, sizeof(prop->gcnArchName), "gfx950:sramecc+:xnack-");
    prop->multiProcessorCount = std::getenv("WTK_STUB_CUS") ? std::atoi(std::getenv("WTK_STUB_CUS")) : 256;
    prop->warpSize = 64;
    prop->maxThreadsPerBlock = 1024;
    prop->sharedMemPerBlock = 64 * 1024;
    prop->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    prop->totalGlobalMem = 288ull << 30;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) {
    static thread_local char buf[64];
    std::snprintf(buf, sizeof(buf), "hip_stub error %d", (int)e);
    return buf;
}
hipError_t hipGetLastError(void) {
    const hipError_t e = t_last_error;
    t_last_error = hipSuccess;
    return e;
}
hipError_t hipDeviceSynchronize(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipDeviceSynchronize");
    if (!g_captures.empty()) {
        for (auto &kv : g_captures)
            if (kv.second->mode == hipStreamCaptureModeGlobal) violation("hipDeviceSynchronize while a global-mode capture is open in the process");
    }
    return hipSuccess;
}

// ---- memory -------------------------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipMalloc");
    if (!p) return ret(hipErrorInvalidValue);
    if (bytes == 0) {
        *p = nullptr;
        return hipSuccess;
    }
    void *q = nullptr;
    bool reserved = false;
    if (bytes > kReserveAbove) {
        q = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (q == MAP_FAILED) return ret(hipErrorOutOfMemory);
        reserved = true;
    } else {
        q = std::malloc(bytes);
        if (!q) return ret(hipErrorOutOfMemory);
        std::memset(q, 0xa5, bytes); // fresh device memory is NOT zero
    }
    g_regions[reinterpret_cast<uintptr_t>(q)] = {bytes, reserved, g_next_region};
    g_live_region_ids.insert(g_next_region++);
    g_live_bytes += bytes;
    if (g_live_bytes > g_peak_bytes) g_peak_bytes = g_live_bytes;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!p) return hipSuccess;
    unsafe_call("hipFree");
    auto it = g_regions.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_regions.end()) {
        violation("hipFree(%p): not the base of a live device allocation (double free, or a pointer into a block)", p);
        return ret(hipErrorInvalidValue);
    }
    g_live_bytes -= it->second.bytes;
    g_live_region_ids.erase(it->second.id);
    if (it->second.reserved)
        munmap(p, it->second.bytes);
    else
        std::free(p);
    g_regions.erase(it);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipHostMalloc");
    void *q = std::malloc(bytes ? bytes : 1);
    if (!q) return ret(hipErrorOutOfMemory);
    g_regions[reinterpret_cast<uintptr_t>(q)] = {bytes, false, g_next_region}; // device-visible
    g_live_region_ids.insert(g_next_region++);
    *p = q;
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) {
    *dev = host;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_regions.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_regions.end()) {
        violation("hipHostFree(%p): unknown pointer", p);
        return ret(hipErrorInvalidValue);
    }
    g_live_region_ids.erase(it->second.id);
    std::free(p);
    g_regions.erase(it);
    return hipSuccess;
}

static hipError_t copy_common(void *dst, const void *src, size_t n, hipMemcpyKind kind, const char *what) {
    bool dst_dev = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice, src_dev = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
    if (kind == hipMemcpyDefault) dst_dev = find_region(dst) != nullptr, src_dev = find_region(src) != nullptr;
    if (n == 0) return hipSuccess;
    if (!dst || !src) {
        violation("%s: null pointer (dst %p, src %p, %zu bytes)", what, dst, src, n);
        return ret(hipErrorInvalidValue);
    }
    int touch = 1;
    if (dst_dev) touch &= check_dev_range(dst, n, what);
    if (src_dev) touch &= check_dev_range(src, n, what);
    // host sides are read / written for real, so that ASan sees a host buffer that is too small; a reserved device side cannot be touched:
    // the host side is then read / written against a scratch buffer
    if (touch)
        std::memmove(dst, src, n);
    else {
        std::vector<unsigned char> scratch(n, 0x5a);
        if (!src_dev) std::memcpy(scratch.data(), src, n); // read the whole host source
        if (!dst_dev) std::memcpy(dst, scratch.data(), n); // write the whole host destination
    }
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipMemcpy");
    return copy_common(dst, src, n, kind, "hipMemcpy");
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipMemcpyAsync");
    if (!s) return ret(hipErrorInvalidHandle);
    enqueue(s, "hipMemcpyAsync");
    return copy_common(dst, src, n, kind, "hipMemcpyAsync");
}
hipError_t hipMemset(void *p, int v, size_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipMemset");
    if (check_dev_range(p, n, "hipMemset")) std::memset(p, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipMemsetAsync");
    if (!s) return ret(hipErrorInvalidHandle);
    enqueue(s, "hipMemsetAsync");
    if (check_dev_range(p, n, "hipMemsetAsync")) std::memset(p, v, n);
    return hipSuccess;
}

// ---- streams ------------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t *out, unsigned flags) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsafe_call("hipStreamCreateWithFlags");
    StubStream *s = new StubStream();
    s->flags = flags;
    s->device = t_device;
    g_streams.insert(s);
    *out = reinterpret_cast<hipStream_t>(s);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamDestroy");
    if (!s || s == &g_null_stream) return ret(hipErrorInvalidHandle);
    if (s->cap) violation("hipStreamDestroy of a stream that is part of an open capture");
    g_streams.erase(s);
    s->magic = kDead;
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamSynchronize");
    if (!s) return ret(hipErrorInvalidHandle);
    if (s->cap) {
        violation("hipStreamSynchronize on a capturing stream");
        s->cap->invalid = true;
        return ret(hipErrorStreamCaptureUnsupported);
    }
    unsafe_call("hipStreamSynchronize");
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t stream, hipStreamCaptureStatus *st) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamIsCapturing");
    if (!s) return ret(hipErrorInvalidHandle);
    *st = s->cap ? (s->cap->invalid ? hipStreamCaptureStatusInvalidated : hipStreamCaptureStatusActive) : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode mode) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamBeginCapture");
    if (!s) return ret(hipErrorInvalidHandle);
    if (s == &g_null_stream) {
        violation("hipStreamBeginCapture on the legacy default stream");
        return ret(hipErrorStreamCaptureUnsupported);
    }
    if (s->cap) {
        violation("hipStreamBeginCapture on a stream that is already capturing (capture %llu)", (unsigned long long)s->cap->id);
        return ret(hipErrorIllegalState);
    }
    if (capture_of_this_thread() && mode != hipStreamCaptureModeRelaxed) violation("hipStreamBeginCapture: this thread already has a capture open (nested captures of one thread)");
    Capture *c = new Capture();
    c->id = g_next_capture++;
    c->origin = s;
    c->thread = std::this_thread::get_id();
    c->mode = (int)mode;
    g_captures[c->id] = c;
    s->cap = c;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t stream, hipGraph_t *graph) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamEndCapture");
    if (graph) *graph = nullptr;
    if (!s) return ret(hipErrorInvalidHandle);
    Capture *c = s->cap;
    if (!c) {
        violation("hipStreamEndCapture on a stream that is not capturing");
        return ret(hipErrorIllegalState);
    }
    if (c->origin != s) {
        violation("hipStreamEndCapture on a stream that did not begin the capture");
        return ret(hipErrorStreamCaptureUnmatched);
    }
    if (c->thread != std::this_thread::get_id() && c->mode != hipStreamCaptureModeRelaxed) violation("hipStreamEndCapture from another thread than hipStreamBeginCapture");
    hipError_t rc = hipSuccess;
    for (StubStream *f : c->forks) {
        if (f->joined_ops != f->ops) {
            violation("hipStreamEndCapture: forked stream %p has captured work that was never joined back into the origin (unjoined)", (void *)f);
            rc = hipErrorStreamCaptureUnjoined;
        }
        f->cap = nullptr;
    }
    s->cap = nullptr;
    for (StubEvent *e : g_events) // HIP: the events of the capture go back to their idle state
        if (e->capture_id == c->id) e->capture_stream = nullptr;
    if (c->invalid && rc == hipSuccess) rc = hipErrorStreamCaptureInvalidated;
    if (rc == hipSuccess && graph) {
        StubGraph *g = new StubGraph();
        g->nodes = c->nodes;
        g->regions = c->regions;
        *graph = reinterpret_cast<hipGraph_t>(g);
    }
    g_captures.erase(c->id);
    delete c;
    return ret(rc);
}
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipStreamWaitEvent");
    StubEvent *e = E(event, "hipStreamWaitEvent");
    if (!s || !e) return ret(hipErrorInvalidHandle);
    if (e->capture_id) { // the event's last record was a captured one
        auto it = g_captures.find(e->capture_id);
        if (it == g_captures.end()) {
            // CUDA: undefined / error; HIP resets the event at end of capture and treats it as never recorded.  Either way the caller's ordering intent is lost.
            violation("hipStreamWaitEvent: event %p was last recorded inside capture %llu, which has ended: the wait orders nothing", (void *)event, (unsigned long long)e->capture_id);
            return hipSuccess;
        }
        Capture *c = it->second;
        if (s->cap && s->cap != c) {
            violation("hipStreamWaitEvent: stream %p belongs to capture %llu but the event to capture %llu (isolation)", (void *)s, (unsigned long long)s->cap->id, (unsigned long long)c->id);
            s->cap->invalid = c->invalid = true;
            return ret(hipErrorStreamCaptureIsolation);
        }
        if (!s->cap) { // fork: the waiting stream joins the capture
            if (s == &g_null_stream) {
                violation("hipStreamWaitEvent: the legacy default stream cannot join a capture");
                return ret(hipErrorStreamCaptureImplicit);
            }
            if (c->thread != std::this_thread::get_id() && c->mode == hipStreamCaptureModeThreadLocal) violation("a stream is pulled into thread-local capture %llu from another thread", (unsigned long long)c->id);
            s->cap = c;
            c->forks.insert(s);
            s->joined_ops = s->ops; // nothing captured on it yet
        }
        // join bookkeeping: a wait on an event recorded on fork F covers F's work up to the record — when the WAITING stream is the origin, or a stream
        // that itself is joined later (transitive joins are accepted: the waiting stream takes over the obligation)
        StubStream *f = e->capture_stream;
        if (f && f != s && c->forks.count(f)) {
            if (e->stream_ops > f->joined_ops) f->joined_ops = e->stream_ops;
            if (s != c->origin) ++s->ops; // the dependency is work of the waiting fork: it must be joined in turn
        }
        ++c->nodes;
        return hipSuccess;
    }
    if (s->cap) {
        if (e->recorded) {
            violation("hipStreamWaitEvent: capturing stream %p waits on event %p recorded OUTSIDE the capture (isolation; invalidates the capture)", (void *)s, (void *)event);
            s->cap->invalid = true;
            return ret(hipErrorStreamCaptureIsolation);
        }
        return hipSuccess; // never recorded: a no-op
    }
    return hipSuccess;
}

// ---- events -------------------------------------------------------------------------------------------------------------------
hipError_t hipEventCreateWithFlags(hipEvent_t *out, unsigned flags) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubEvent *e = new StubEvent();
    e->flags = flags;
    g_events.insert(e);
    *out = reinterpret_cast<hipEvent_t>(e);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *out) { return hipEventCreateWithFlags(out, 0); }
hipError_t hipEventDestroy(hipEvent_t event) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubEvent *e = E(event, "hipEventDestroy");
    if (!e) return ret(hipErrorInvalidHandle);
    if (e->capture_id && g_captures.count(e->capture_id)) violation("hipEventDestroy: event %p was recorded inside capture %llu, which is still open", (void *)event, (unsigned long long)e->capture_id);
    g_events.erase(e);
    e->magic = kDead;
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubStream *s = S(stream, "hipEventRecord");
    StubEvent *e = E(event, "hipEventRecord");
    if (!s || !e) return ret(hipErrorInvalidHandle);
    if (s->cap) {
        if (s->cap->thread != std::this_thread::get_id() && s->cap->mode == hipStreamCaptureModeThreadLocal) violation("hipEventRecord on a stream captured by another thread");
        e->capture_id = s->cap->id;
        e->capture_stream = s;
        e->stream_ops = s->ops;
        e->recorded = false;
        ++s->cap->nodes;
    } else {
        e->capture_id = 0;
        e->capture_stream = nullptr;
        e->recorded = true;
        e->stream_ops = s->ops;
    }
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t event) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubEvent *e = E(event, "hipEventSynchronize");
    if (!e) return ret(hipErrorInvalidHandle);
    if (e->capture_id && g_captures.count(e->capture_id)) {
        violation("hipEventSynchronize on an event recorded inside an open capture");
        return ret(hipErrorCapturedEvent);
    }
    unsafe_call("hipEventSynchronize");
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubEvent *ea = E(a, "hipEventElapsedTime"), *eb = E(b, "hipEventElapsedTime");
    if (!ea || !eb) return ret(hipErrorInvalidHandle);
    if (!ea->recorded || !eb->recorded) {
        violation("hipEventElapsedTime on an event that was never recorded eagerly");
        return ret(hipErrorInvalidHandle);
    }
    if ((ea->flags | eb->flags) & hipEventDisableTiming) violation("hipEventElapsedTime on an event created with hipEventDisableTiming");
    *ms = 0.001f;
    return hipSuccess;
}

// ---- graphs -------------------------------------------------------------------------------------------------------------------
hipError_t hipGraphInstantiate(hipGraphExec_t *out, hipGraph_t graph, hipGraphNode_t *, char *, size_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubGraph *g = reinterpret_cast<StubGraph *>(graph);
    if (!g || g->magic != kGraphMagic) {
        violation("hipGraphInstantiate: not a live graph");
        return ret(hipErrorInvalidValue);
    }
    unsafe_call("hipGraphInstantiate");
    StubExec *x = new StubExec();
    x->nodes = g->nodes;
    x->regions = g->regions;
    g_execs.insert(x);
    *out = reinterpret_cast<hipGraphExec_t>(x);
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t graph) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubGraph *g = reinterpret_cast<StubGraph *>(graph);
    if (!g || g->magic != kGraphMagic) {
        violation("hipGraphDestroy: not a live graph");
        return ret(hipErrorInvalidValue);
    }
    g->magic = kDead;
    delete g;
    return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubExec *x = reinterpret_cast<StubExec *>(exec);
    if (!x || !g_execs.count(x)) {
        violation("hipGraphExecDestroy: %p is not a live graph exec (null, destroyed twice, or never instantiated)", (void *)exec);
        return ret(hipErrorInvalidValue);
    }
    g_execs.erase(x);
    x->magic = kDead;
    delete x;
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    StubExec *x = reinterpret_cast<StubExec *>(exec);
    StubStream *s = S(stream, "hipGraphLaunch");
    if (!x || !g_execs.count(x)) {
        violation("hipGraphLaunch: %p is not a live graph exec", (void *)exec);
        return ret(hipErrorInvalidValue);
    }
    if (!s) return ret(hipErrorInvalidHandle);
    if (s->cap) violation("hipGraphLaunch on a capturing stream (would become a child graph: not what the library means)");
    for (uint64_t id : x->regions)
        if (!g_live_region_ids.count(id)) {
            violation("hipGraphLaunch: the graph was captured with a device allocation (#%llu) that has been freed since: its kernels would read / write freed memory", (unsigned long long)id);
            break;
        }
    ++x->launches;
    enqueue(s, "hipGraphLaunch");
    return hipSuccess;
}

// ---- launches -----------------------------------------------------------------------------------------------------------------
hipError_t hipFuncSetAttribute(const void *func, hipFuncAttribute attr, int value) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_kernels.find(func);
    if (it == g_kernels.end()) {
        violation("hipFuncSetAttribute: %p is not a registered kernel", func);
        return ret(hipErrorInvalidDeviceFunction);
    }
    if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (value > 160 * 1024) {
            violation("hipFuncSetAttribute(%s): %d bytes of dynamic LDS exceed the CU's 160 KiB", it->second.name.c_str(), value);
            return ret(hipErrorInvalidValue);
        }
        it->second.max_dyn_lds = value;
    }
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *func, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_kernels.find(func);
    if (it == g_kernels.end()) {
        violation("hipLaunchKernel: %p is not a registered kernel", func);
        return ret(hipErrorInvalidDeviceFunction);
    }
    KernelInfo &k = it->second;
    StubStream *s = S(stream, k.name.c_str());
    if (!s) return ret(hipErrorInvalidHandle);
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || threads == 0 || threads > 1024 || grid.y > 65535 || grid.z > 65535 || grid.x > 0x7fffffffu) {
        violation("%s: launch configuration grid (%u, %u, %u) block (%u, %u, %u) is invalid", k.name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z);
        return ret(hipErrorInvalidConfiguration);
    }
    if ((long long)shmem > k.max_dyn_lds) {
        violation("%s: %zu bytes of dynamic LDS exceed the function's limit of %d (hipFuncSetAttribute missing?)", k.name.c_str(), shmem, k.max_dyn_lds);
        return ret(hipErrorInvalidValue);
    }
    ++k.launches;
    if (g_verbose) std::fprintf(stderr, "launch %s grid %u block %u lds %zu%s\n", k.name.c_str(), grid.x, block.x, shmem, s->cap ? " [captured]" : "");
    stubchk::check_launch(k.name, grid, block, args);
    enqueue(s, k.name.c_str());
    return hipSuccess;
}

} // extern "C"
