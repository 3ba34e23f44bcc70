// Error state, version, and the unit executor (recorded op list -> per-iteration replay / hipGraph).
#include "rdo_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace rdo {

static thread_local std::string g_err;
static thread_local Recorder g_rec;

Recorder& recorder() { return g_rec; }

int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {
struct TuneDef { const char* key; const char* env; int dflt; };
constexpr TuneDef kTune[T_COUNT] = {{"wgrad_x6_w8", "RDO_WGX6_W8", 1}, {"conv_x6", "RDO_CONV_X6", 1},
                                    {"fwd_x6_ver", "RDO_X6_VER", 6}, {"xcd", "RDO_XCD", 1},
                                    {"x6p_ablate", "RDO_X6P_ABLATE", 0}, {"graph_unroll", "RDO_GRAPH_UNROLL", 8},
                                    {"tail_grid", "RDO_TAIL_GRID", 1024}, {"x6p_halo", "RDO_X6P_HALO", 1},
                                    {"wgrad_p3_row", "RDO_WGRAD_P3_ROW", 1}, {"thin_mfma", "RDO_THIN_MFMA", 1},
                                    {"h2_stagger", "RDO_H2_STAGGER", 1}, {"ada_w1_min", "RDO_ADA_W1_MIN", 128},
                                    {"h2_k32", "RDO_H2_K32", 3}, {"wgrad_sub", "RDO_WGRAD_SUB", 13}, {"h2_n48", "RDO_H2_N48", 1}};
std::atomic<int> g_tune[T_COUNT];
std::once_flag g_tune_once;
void tune_init() {
    std::call_once(g_tune_once, [] {
        for (int i = 0; i < T_COUNT; ++i) {
            const char* e = getenv(kTune[i].env);
            int v = e ? atoi(e) : kTune[i].dflt;
#ifndef RDO_DIAG   // values that select code compiled only into diagnostic builds are ignored (a stray variable must change nothing)
            if (i == T_X6P_ABLATE || (i == T_FWD_X6_VER && v < 5) || (i == T_WGRAD_X6_W8 && v == 0)) v = kTune[i].dflt;
#endif
            g_tune[i].store(v);
        }
    });
}
int tune_index(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < T_COUNT; ++i)
        if (!strcmp(key, kTune[i].key)) return i;
    return -1;
}
}  // namespace

int tuning(Tune t) {
    tune_init();
    return g_tune[t].load(std::memory_order_relaxed);
}

// rdo_h2_bind_flag: the caller's own flag word for the launches of this thread (nullptr: the per-device default below)
static thread_local int* t_h2_bound = nullptr;

// rdo_iter_bind_publish: where the NEXT loss / tail launch of this thread leaves the iteration number it read (nullptr: nowhere)
static thread_local int32_t* t_iter_pub = nullptr;
int32_t* take_iter_publish() {
    int32_t* p = t_iter_pub;
    t_iter_pub = nullptr;
    return p;
}

int* h2_overflow_flag() {
    if (t_h2_bound) return t_h2_bound;
    static std::mutex mu;
    static int* ptr[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    int*& p = ptr[dev & 63];
    if (!p) {
        if (hipMalloc(reinterpret_cast<void**>(&p), 64) != hipSuccess) { p = nullptr; return nullptr; }
        (void)hipMemset(p, 0, 64);
    }
    return p;
}

}  // namespace rdo

struct rdo_plan {
    std::vector<rdo::Op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph_k = nullptr;        // kUnroll iterations in one graph: no inter-graph gap between them
    hipGraphExec_t exec_k = nullptr;
    int unroll_k = 0;                    // iterations inside exec_k (the tuning value at ITS capture, not the current one)
    hipStream_t cap_stream = nullptr;
    bool recording = false;
    // rdo_plan_run_then: this plan's ops followed by another plan's in ONE graph (cached for that partner)
    const rdo_plan* then_with = nullptr;
    unsigned long then_gen = 0;          // the partner's recording generation at capture (a re-recorded partner invalidates the graph)
    unsigned long generation = 0;        // process-wide recording serial, set by every rdo_plan_begin_record (unique across plans)
    hipGraph_t graph_then = nullptr;
    hipGraphExec_t exec_then = nullptr;
};

extern "C" {

#ifdef RDO_DIAG
const char* rdo_version(void) { return "rdo-ptq-hip 0.1 (gfx950) DIAG"; }     // diagnostic build: ablation masks / stamps compiled in
#else
const char* rdo_version(void) { return "rdo-ptq-hip 0.1 (gfx950)"; }
#endif
const char* rdo_last_error(void) { return rdo::g_err.c_str(); }

int rdo_set_tuning(const char* key, int32_t value) {
    const int i = rdo::tune_index(key);
    RDO_REQUIRE(i >= 0, "rdo_set_tuning: unknown key '%s'", key ? key : "(null)");
#ifndef RDO_DIAG
    RDO_REQUIRE(i != rdo::T_X6P_ABLATE || value == 0, "rdo_set_tuning: 'x6p_ablate' exists only in a diagnostic build (make DIAG=1)");
    RDO_REQUIRE(i != rdo::T_FWD_X6_VER || value >= 5, "rdo_set_tuning: 'fwd_x6_ver' %d exists only in a diagnostic build (make DIAG=1)", value);
    RDO_REQUIRE(i != rdo::T_WGRAD_X6_W8 || value != 0, "rdo_set_tuning: the four-wave weight gradient exists only in a diagnostic build (make DIAG=1)");
#endif
    rdo::tune_init();
    rdo::g_tune[i].store(value);
    return RDO_OK;
}

int rdo_get_tuning(const char* key) {
    const int i = rdo::tune_index(key);
    if (i < 0) return -1;
    rdo::tune_init();
    return rdo::g_tune[i].load();
}

int rdo_iter_bind_publish(int32_t* publish) {
    const int pending = rdo::t_iter_pub != nullptr;      // 1: an earlier binding was never consumed (no loss launch followed it)
    rdo::t_iter_pub = publish;
    return pending;
}

int rdo_h2_bind_flag(int32_t* flag) {
    rdo::t_h2_bound = flag;
    return RDO_OK;
}

int rdo_h2_overflow(int reset) {
    int* p = rdo::h2_overflow_flag();
    if (!p) return -1;
    int v[2] = {0, 0};
    if (hipMemcpy(v, p, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset && (v[0] | v[1])) (void)hipMemset(p, 0, sizeof v);
    return (v[0] | v[1]) != 0;
}

rdo_plan* rdo_plan_create(void) { return new rdo_plan(); }

void rdo_plan_destroy(rdo_plan* p) {
    if (!p) return;
    if (p->recording) { rdo::recorder().active = false; rdo::recorder().sink = nullptr; }
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->exec_k) (void)hipGraphExecDestroy(p->exec_k);
    if (p->graph_k) (void)hipGraphDestroy(p->graph_k);
    if (p->exec_then) (void)hipGraphExecDestroy(p->exec_then);
    if (p->graph_then) (void)hipGraphDestroy(p->graph_then);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    delete p;
}

// Suspend (on != 0) / resume recording on this thread: calls in between are LAUNCHED, not recorded -- one-time preparation of constants
// (a frozen weight's derived layouts) discovered while a plan is being recorded.  Returns the previous state (1 = was suspended), < 0 on error.
int rdo_plan_suspend_record(int on) {
    static thread_local bool suspended = false;
    rdo::Recorder& r = rdo::recorder();
    const int was = suspended ? 1 : 0;
    if (on && !suspended) {
        if (!r.active) return 0;                         // nothing is recording: nothing to suspend
        r.active = false;
        suspended = true;
    } else if (!on && suspended) {
        r.active = true;
        suspended = false;
    }
    return was;
}

int rdo_plan_begin_record(rdo_plan* p) {
    RDO_REQUIRE(p != nullptr, "plan is null");
    RDO_REQUIRE(!rdo::recorder().active, "another plan is already recording on this thread");
    p->ops.clear();
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
    if (p->exec_k) { (void)hipGraphExecDestroy(p->exec_k); p->exec_k = nullptr; }
    if (p->graph_k) { (void)hipGraphDestroy(p->graph_k); p->graph_k = nullptr; }
    if (p->exec_then) { (void)hipGraphExecDestroy(p->exec_then); p->exec_then = nullptr; }
    if (p->graph_then) { (void)hipGraphDestroy(p->graph_then); p->graph_then = nullptr; }
    p->then_with = nullptr;
    static std::atomic<unsigned long> serial{0};
    p->generation = ++serial;
    p->recording = true;
    rdo::recorder().active = true;
    rdo::recorder().sink = &p->ops;
    return RDO_OK;
}

int rdo_plan_end_record(rdo_plan* p) {
    RDO_REQUIRE(p != nullptr && p->recording, "plan is not recording");
    p->recording = false;
    rdo::recorder().active = false;
    rdo::recorder().sink = nullptr;
    return RDO_OK;
}

int rdo_plan_num_ops(const rdo_plan* p) { return p ? (int)p->ops.size() : 0; }

static int run_ops(rdo_plan* p, hipStream_t s) {
    for (auto& op : p->ops) {
        int rc = op.fn(s);
        if (rc != RDO_OK) return rc;
    }
    return RDO_OK;
}

static int plan_run_impl(rdo_plan* p, int n_iters, int use_graph, void* stream, bool launch);

int rdo_plan_run(rdo_plan* p, int n_iters, int use_graph, void* stream) { return plan_run_impl(p, n_iters, use_graph, stream, true); }

// Capture and instantiate the graphs a later rdo_plan_run(p, n_iters, 1, .) will replay, without launching anything: graph building
// is set-up work (like recording); a caller that times its loop does it here instead of inside the first timed call.
int rdo_plan_prepare(rdo_plan* p, int n_iters) { return plan_run_impl(p, n_iters, 1, nullptr, false); }

static int plan_run_impl(rdo_plan* p, int n_iters, int use_graph, void* stream, bool launch) {
    RDO_REQUIRE(p != nullptr && !p->recording, "plan is null or still recording");
    RDO_REQUIRE(n_iters >= 0, "n_iters < 0");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!use_graph) {
        for (int i = 0; i < n_iters; ++i) {
            int rc = run_ops(p, s);
            if (rc != RDO_OK) return rc;
        }
        return RDO_OK;
    }
    auto capture = [&](int reps, hipGraph_t* g, hipGraphExec_t* ex) -> int {
        // Capture on a private stream (the caller's stream may be the legacy default stream).
        if (!p->cap_stream && hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipStreamCreate failed");
        if (hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipStreamBeginCapture failed");
        int rc = RDO_OK;
        for (int r = 0; r < reps && rc == RDO_OK; ++r) rc = run_ops(p, p->cap_stream);
        hipError_t e = hipStreamEndCapture(p->cap_stream, g);
        if (rc != RDO_OK) return rc;
        if (e != hipSuccess) return rdo::set_error(RDO_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(ex, *g, nullptr, nullptr, 0);
        if (e != hipSuccess) return rdo::set_error(RDO_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        return RDO_OK;
    };
    // Long runs replay a graph of kUnroll iterations (every per-iteration scalar comes from device tables indexed by the device
    // iteration counter, so consecutive iterations can live in one graph): the ~2 us gap between two graph launches and the per-
    // launch host cost are paid once per kUnroll iterations -- it shows on the small units (45-160 us per iteration).
    const int unroll = rdo::tuning(rdo::T_GRAPH_UNROLL);
    int i = 0;
    if (unroll > 1 && n_iters >= 2 * unroll) {
        if (p->exec_k && p->unroll_k != unroll) {      // "graph_unroll" changed since the capture: the old graph holds the old count
            (void)hipGraphExecDestroy(p->exec_k);
            (void)hipGraphDestroy(p->graph_k);
            p->exec_k = nullptr;
            p->graph_k = nullptr;
        }
        if (!p->exec_k) {
            if (int rc = capture(unroll, &p->graph_k, &p->exec_k)) return rc;
            p->unroll_k = unroll;
        }
        for (; i + unroll <= n_iters; i += unroll) {
            if (!launch) continue;
            hipError_t e = hipGraphLaunch(p->exec_k, s);
            if (e != hipSuccess) return rdo::set_error(RDO_EHIP, "hipGraphLaunch: %s", hipGetErrorString(e));
        }
    }
    if (i < n_iters && !p->exec)
        if (int rc = capture(1, &p->graph, &p->exec)) return rc;
    for (; i < n_iters && launch; ++i) {
        hipError_t e = hipGraphLaunch(p->exec, s);
        if (e != hipSuccess) return rdo::set_error(RDO_EHIP, "hipGraphLaunch: %s", hipGetErrorString(e));
    }
    return RDO_OK;
}

// One iteration of `p` followed by one iteration of `q` as ONE graph launch (or eagerly): the data-parallel host loop enqueues
// "apply of iteration i + forward/backward of iteration i + 1" with one call between two collectives.
int rdo_plan_run_then(rdo_plan* p, rdo_plan* q, int use_graph, void* stream) {
    RDO_REQUIRE(p != nullptr && q != nullptr && !p->recording && !q->recording, "rdo_plan_run_then: plan is null or still recording");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!use_graph) {
        if (int rc = run_ops(p, s)) return rc;
        return run_ops(q, s);
    }
    if (p->exec_then && (p->then_with != q || p->then_gen != q->generation)) {
        (void)hipGraphExecDestroy(p->exec_then);
        (void)hipGraphDestroy(p->graph_then);
        p->exec_then = nullptr;
        p->graph_then = nullptr;
    }
    if (!p->exec_then) {
        if (!p->cap_stream && hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipStreamCreate failed");
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(p->cap_stream, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone)
            return rdo::set_error(RDO_EHIP, "rdo_plan_run_then: the plan's capture stream is still capturing (an earlier capture failed)");
        if (hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipStreamBeginCapture failed");
        int rc = run_ops(p, p->cap_stream);
        if (rc == RDO_OK) rc = run_ops(q, p->cap_stream);
        hipError_t e = hipStreamEndCapture(p->cap_stream, &p->graph_then);     // always ended: the stream must leave capture mode
        auto drop = [&]() {                                                     // every error path: no half-built graph is kept
            if (p->exec_then) { (void)hipGraphExecDestroy(p->exec_then); p->exec_then = nullptr; }
            if (p->graph_then) { (void)hipGraphDestroy(p->graph_then); p->graph_then = nullptr; }
        };
        if (rc != RDO_OK) { drop(); return rc; }
        if (e != hipSuccess) { drop(); return rdo::set_error(RDO_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
        e = hipGraphInstantiate(&p->exec_then, p->graph_then, nullptr, nullptr, 0);
        if (e != hipSuccess) { drop(); return rdo::set_error(RDO_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
        p->then_with = q;
        p->then_gen = q->generation;
    }
    hipError_t e = hipGraphLaunch(p->exec_then, s);
    if (e != hipSuccess) return rdo::set_error(RDO_EHIP, "hipGraphLaunch: %s", hipGetErrorString(e));
    return RDO_OK;
}

int rdo_plan_op_info(const rdo_plan* p, int i, const char** tag, double* flops, double* bytes) {
    RDO_REQUIRE(p && i >= 0 && i < (int)p->ops.size(), "rdo_plan_op_info: index out of range");
    if (tag) *tag = p->ops[i].tag;
    if (flops) *flops = p->ops[i].flops;
    if (bytes) *bytes = p->ops[i].bytes;
    return RDO_OK;
}

// One eager iteration with a hipEvent pair around every op: ms[i] = device time of op i on `stream`.  Synchronises.
int rdo_plan_profile(rdo_plan* p, float* ms, void* stream) {
    RDO_REQUIRE(p != nullptr && !p->recording && ms != nullptr, "rdo_plan_profile: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t n = p->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return rdo::set_error(RDO_EHIP, "hipEventCreate failed");
    int rc = RDO_OK;
    (void)hipEventRecord(ev[0], s);
    for (size_t i = 0; i < n && rc == RDO_OK; ++i) {
        rc = p->ops[i].fn(s);
        (void)hipEventRecord(ev[i + 1], s);
    }
    if (hipStreamSynchronize(s) != hipSuccess && rc == RDO_OK) rc = rdo::set_error(RDO_EHIP, "hipStreamSynchronize failed");
    if (rc == RDO_OK)
        for (size_t i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
