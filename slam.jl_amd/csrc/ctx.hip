// ctx.hip -- context, scratch management, error reporting, filter coefficients.
#include "common.hpp"
#include <mutex>
#include <cstdlib>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdarg>

thread_local std::string g_slam_err;

int slam_fail(slam_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_slam_err = buf;
    return code;
}

static int grow(slam_ctx *ctx, void **p, size_t *have, size_t want, bool pinned)
{
    if (*have >= want) return SLAM_OK;
    if (*p) { HIP_TRY(ctx, slam_stream_wait(ctx->stream)); if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *have = 0; }
    size_t sz = want + want / 4 + 4096;
    if (pinned) HIP_TRY(ctx, hipHostMalloc(p, sz, hipHostMallocMapped | hipHostMallocCoherent));   // device-visible, fine-grained
    else HIP_TRY(ctx, hipMalloc(p, sz));
    *have = sz;
    return SLAM_OK;
}
int slam_scratch(slam_ctx *ctx, size_t bytes, void **out) { int rc = grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes, false); *out = ctx->scratch; return rc; }
int slam_scratch2(slam_ctx *ctx, size_t bytes, void **out) { int rc = grow(ctx, &ctx->scratch2, &ctx->scratch2_bytes, bytes, false); *out = ctx->scratch2; return rc; }
int slam_pinned(slam_ctx *ctx, size_t bytes, void **out) { int rc = grow(ctx, &ctx->pinned, &ctx->pinned_bytes, bytes, true); *out = ctx->pinned; return rc; }

static hipEvent_t prof_event(slam_ctx *c)
{
    if (!c->prof_pool.empty()) { hipEvent_t e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
static int prof_id(slam_ctx *c, const char *name)
{
    for (size_t i = 0; i < c->prof_names.size(); i++) if (c->prof_names[i] == name) return (int)i;
    c->prof_names.push_back(name); c->prof_ms.push_back(0.0); c->prof_cnt.push_back(0);
    return (int)c->prof_names.size() - 1;
}
ProfScope::ProfScope(slam_ctx *ctx, const char *name) : c(ctx), idx(-1)
{
    if (!c || !c->prof_on) return;
    slam_ctx::ProfSpan sp; sp.id = prof_id(c, name); sp.a = prof_event(c); sp.b = prof_event(c);
    (void)hipEventRecord(sp.a, c->stream);
    c->prof_pending.push_back(sp); idx = (int)c->prof_pending.size() - 1;
}
ProfScope::~ProfScope()
{
    if (idx >= 0) (void)hipEventRecord(c->prof_pending[idx].b, c->stream);
}
static void prof_collect(slam_ctx *c)
{
    (void)slam_stream_wait(c->stream);
    for (auto &sp : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { c->prof_ms[sp.id] += ms; c->prof_cnt[sp.id]++; }
        c->prof_pool.push_back(sp.a); c->prof_pool.push_back(sp.b);
    }
    c->prof_pending.clear();
}

hipError_t slam_stream_wait(hipStream_t s)
{
    static const long spin_us = [] { const char *v = getenv("SLAMHIP_SPIN_US"); return v ? atol(v) : 2000L; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0;; k++) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if ((k & 63) == 63 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    return hipStreamSynchronize(s);
}

extern "C" {

int slam_prof_enable(slam_ctx *ctx, int on)
{
    ARG_TRY(ctx, ctx != nullptr);
    if (!on && ctx->prof_on) prof_collect(ctx);
    ctx->prof_on = on != 0;
    return SLAM_OK;
}
int slam_prof_reset(slam_ctx *ctx)
{
    ARG_TRY(ctx, ctx != nullptr);
    prof_collect(ctx);
    for (auto &v : ctx->prof_ms) v = 0.0;
    for (auto &v : ctx->prof_cnt) v = 0;
    return SLAM_OK;
}
int slam_prof_get(slam_ctx *ctx, const char *name, double *total_ms, int64_t *count)
{
    ARG_TRY(ctx, ctx != nullptr && name != nullptr);
    prof_collect(ctx);
    double ms = 0.0; long long n = 0;
    for (size_t i = 0; i < ctx->prof_names.size(); i++) if (ctx->prof_names[i] == name) { ms = ctx->prof_ms[i]; n = ctx->prof_cnt[i]; }
    if (total_ms) *total_ms = ms;
    if (count) *count = n;
    return SLAM_OK;
}

// Streams are PARKED, not destroyed, when their context goes (and handed to the next context of the same device and scheduling
// class): another library sharing the process may still hold events it recorded on a context's stream -- PyTorch's pinned-memory
// allocator does, for every non-blocking copy issued on a torch.cuda.ExternalStream of ours, and queries them on LATER allocations; a
// query of an event whose stream had been destroyed failed once in ~17 full bench runs with hipErrorCapturedEvent.  Until
// round 5 the rule "retire foreign events before slam_ctx_destroy" lived in INTEGRATION prose; now no stream a caller has seen ever
// dies before the process does.  (CU-masked streams are not pooled: their mask is part of their identity.)  SLAMHIP_NO_STREAM_POOL=1
// restores destroy-on-close.
namespace {
struct ParkedStream { int device, prio; hipStream_t st; };
std::mutex g_park_mu;
std::vector<ParkedStream> g_parked;
bool stream_pool_on() { static const bool on = getenv("SLAMHIP_NO_STREAM_POOL") == nullptr; return on; }
hipStream_t take_parked(int device, int prio)
{
    if (!stream_pool_on()) return nullptr;
    std::lock_guard<std::mutex> lk(g_park_mu);
    for (size_t i = 0; i < g_parked.size(); i++)
        if (g_parked[i].device == device && g_parked[i].prio == prio) { hipStream_t s = g_parked[i].st; g_parked.erase(g_parked.begin() + (long)i); return s; }
    return nullptr;
}
}  // namespace

// the context's second stream (created on first use, in the scheduling class of the first; CU-masked contexts have none) and its two events
hipStream_t ctx_aux_stream(slam_ctx *c)
{
    if (c->stream2) return c->stream2;
    if (c->cus > 0) return nullptr;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int p = c->pool_class == 1 ? greatest : c->pool_class == -1 ? least : 0;
    if (e == hipSuccess) e = c->pool_class == 0 || c->pool_class == 99 ? hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) : hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, p);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGetLastError(); if (c->stream2) (void)hipStreamDestroy(c->stream2); c->stream2 = nullptr; return nullptr; }
    return c->stream2;
}

// What the device is, once per context: the kernels whose workgroups wait for one another (k_ba_window's two halves, the twisted band
// solve, k_cum_fused's row segments) need to know how many workgroups the chip really holds and that it is the architecture their
// memory-side hand-overs were validated on (no hard-coded 256, no assumption about other architectures).
void ctx_probe_device(slam_ctx *c)
{
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, c->device) != hipSuccess) { c->dev_cus = 0; c->arch_ok = c->xwg_ok = false; return; }
    c->dev_cus = pr.multiProcessorCount;
    const bool arch = strncmp(pr.gcnArchName, "gfx950", 6) == 0 || strncmp(pr.gcnArchName, "gfx942", 6) == 0;
    c->arch_ok = arch;
    c->xwg_ok = arch && c->cus == 0 && c->dev_cus >= 16;
}

int slam_ctx_create(int device, slam_ctx **out)
{
    if (!out) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create: device %d out of range [0,%d)", device, n);
    slam_ctx *c = new slam_ctx();
    c->device = device;
    e = hipSetDevice(device);
    c->pool_class = 0;
    if (e == hipSuccess && (c->stream = take_parked(device, 0)) == nullptr) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create: %s", hipGetErrorString(e)); }
    ctx_probe_device(c);
    *out = c;
    return SLAM_OK;
}

// A context whose stream may only use the compute units set in `cu_mask` (bit i of word i / 32: CU i of the device's
// enumeration): lets two concurrently running stages -- the bandwidth-bound pyramid builds and the latency-bound tracking
// kernels -- each keep a partition of the chip instead of evicting one another from the CUs (LDS and wave slots).
int slam_ctx_create_cumask(int device, const uint32_t *cu_mask, int n_words, slam_ctx **out)
{
    if (!out || !cu_mask || n_words <= 0) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create_cumask: bad argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create_cumask: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create_cumask: device %d out of range [0,%d)", device, n);
    slam_ctx *c = new slam_ctx();
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)n_words, cu_mask);
    for (int w = 0; w < n_words; w++) c->cus += __builtin_popcount(cu_mask[w]);
    if (e != hipSuccess) { delete c; return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create_cumask: %s", hipGetErrorString(e)); }
    ctx_probe_device(c);
    *out = c;
    return SLAM_OK;
}

// A context on a stream of its own scheduling class: priority > 0 high, < 0 low, 0 the default class.  The runtime maps
// streams of one class onto a small shared set of hardware queues, and the branches of a replayed hipGraph are placed on that
// same set: work enqueued behind a graph branch that is waiting for its predecessors waits with it (measured: the tracking
// kernels of a step sat behind the pyramid graph's small-level branch for the whole build).  A stream of another class has
// its own hardware queue, and the dispatcher prefers it -- what the latency-critical tracking stream of a pipeline wants.
int slam_ctx_create_priority(int device, int priority, slam_ctx **out)
{
    if (!out) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create_priority: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create_priority: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_ctx_create_priority: device %d out of range [0,%d)", device, n);
    slam_ctx *c = new slam_ctx();
    c->device = device;
    e = hipSetDevice(device);
    int least = 0, greatest = 0;                      // numerically: greatest priority <= 0 <= least priority
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    int p = priority > 0 ? greatest : priority < 0 ? least : 0;
    c->pool_class = priority > 0 ? 1 : priority < 0 ? -1 : 0;
    if (e == hipSuccess && (c->stream = take_parked(device, c->pool_class)) == nullptr) e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, p);
    if (e != hipSuccess) { delete c; return slam_fail(nullptr, SLAM_ERR_HIP, "slam_ctx_create_priority: %s", hipGetErrorString(e)); }
    ctx_probe_device(c);
    *out = c;
    return SLAM_OK;
}

extern "C" void ba_forget_jobs(slam_ctx *ctx);          // ba_batch.hip: a batch job still in flight on the context is waited for
int slam_ctx_destroy(slam_ctx *ctx)
{
    if (!ctx) return SLAM_OK;
    ba_forget_jobs(ctx);
    (void)hipSetDevice(ctx->device);
    (void)slam_stream_wait(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->wait_event) (void)hipEventDestroy(ctx->wait_event);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    for (auto &sp : ctx->prof_pending) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (stream_pool_on() && ctx->pool_class != 99) { std::lock_guard<std::mutex> lk(g_park_mu); g_parked.push_back({ctx->device, ctx->pool_class, ctx->stream}); }
    else (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SLAM_OK;
}

int slam_ctx_synchronize(slam_ctx *ctx)
{
    ARG_TRY(ctx, ctx != nullptr);
    HIP_TRY(ctx, slam_stream_wait(ctx->stream));
    return SLAM_OK;
}

void *slam_ctx_stream(slam_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int slam_ctx_wait_for(slam_ctx *ctx, slam_ctx *other)
{
    ARG_TRY(ctx, ctx != nullptr && other != nullptr && ctx->device == other->device);
    if (ctx == other) return SLAM_OK;
    if (!ctx->wait_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wait_event, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->wait_event, other->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->wait_event, 0));
    return SLAM_OK;
}

// Markers: "what `ctx` has enqueued up to here" as a handle another context can wait on later, after more work has been
// enqueued on `ctx` (slam_ctx_wait_for can only name the current tail of the other stream).
struct slam_event { hipEvent_t ev; int device; };
int slam_event_create(slam_ctx *ctx, slam_event **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slam_event *e = new slam_event();
    e->device = ctx->device;
    hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (rc != hipSuccess) { delete e; return slam_fail(ctx, SLAM_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(rc)); }
    *out = e;
    return SLAM_OK;
}
// event with timing enabled: a pair of them brackets a stage on a context's stream (slam_event_elapsed_ms)
int slam_event_create_timed(slam_ctx *ctx, slam_event **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slam_event *e = new slam_event();
    e->device = ctx->device;
    hipError_t rc = hipEventCreate(&e->ev);
    if (rc != hipSuccess) { delete e; return slam_fail(ctx, SLAM_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(rc)); }
    *out = e;
    return SLAM_OK;
}
// milliseconds between two recorded timed events (waits for `b`)
int slam_event_elapsed_ms(slam_event *a, slam_event *b, double *ms)
{
    if (!a || !b || !ms) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_event_elapsed_ms: NULL argument");
    hipError_t rc = hipEventSynchronize(b->ev);
    float f = 0.f;
    if (rc == hipSuccess) rc = hipEventElapsedTime(&f, a->ev, b->ev);
    if (rc != hipSuccess) return slam_fail(nullptr, SLAM_ERR_HIP, "slam_event_elapsed_ms: %s", hipGetErrorString(rc));
    *ms = (double)f;
    return SLAM_OK;
}
int slam_event_record(slam_ctx *ctx, slam_event *e)
{
    ARG_TRY(ctx, ctx != nullptr && e != nullptr && e->device == ctx->device);
    HIP_TRY(ctx, hipEventRecord(e->ev, ctx->stream));
    return SLAM_OK;
}
int slam_ctx_wait_event(slam_ctx *ctx, slam_event *e)
{
    ARG_TRY(ctx, ctx != nullptr && e != nullptr && e->device == ctx->device);
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, e->ev, 0));
    return SLAM_OK;
}
int slam_event_destroy(slam_event *e)
{
    if (!e) return SLAM_OK;
    (void)hipSetDevice(e->device);
    (void)hipEventDestroy(e->ev);
    delete e;
    return SLAM_OK;
}

const char *slam_last_error(slam_ctx *ctx) { return ctx ? ctx->err.c_str() : g_slam_err.c_str(); }

const char *slam_version(void) { return "slamhip 0.1 gfx950"; }

} // extern "C"

// KernelFactors.IIRGaussian(sigma) -> TriggsSdika coefficients.  Host-side, same
// expression order as ImageFiltering's constructor so the constants are the
// ones Julia would compute.
IIRCoef slam_iir_coef(double sigma)
{
    const double m0 = 1.16680, m1 = 1.10783, m2 = 1.40586;
    double q = 1.31564 * (std::sqrt(1 + 0.490811 * sigma * sigma) - 1);
    double ascale = (m0 + q) * (m1 * m1 + m2 * m2 + 2 * m1 * q + q * q);
    double B = (m0 * (m1 * m1 + m2 * m2) / ascale);
    B = B * B;
    double a1 = q * (2 * m0 * m1 + m1 * m1 + m2 * m2 + (2 * m0 + 4 * m1) * q + 3 * q * q) / ascale;
    double a2 = -q * q * (m0 + 2 * m1 + 3 * q) / ascale;
    double a3 = q * q * q / ascale;
    IIRCoef k;
    k.a1 = a1; k.a2 = a2; k.a3 = a3; k.scale = B;
    double Md = (1 + a1 - a2 + a3) * (1 - a1 - a2 - a3) * (1 + a2 + (a1 - a3) * a3);
    double M[9] = {
        -a3 * a1 + 1 - a3 * a3 - a2, (a3 + a1) * (a2 + a3 * a1), a3 * (a1 + a3 * a2),
        a1 + a3 * a2, -(a2 - 1) * (a2 + a3 * a1), -(a3 * a1 + a3 * a3 + a2 - 1) * a3,
        a3 * a1 + a2 + a1 * a1 - a2 * a2,
        a1 * a2 + a3 * a2 * a2 - a1 * a3 * a3 - a3 * a3 * a3 - a3 * a2 + a3,
        a3 * (a1 + a3 * a2)};
    for (int i = 0; i < 9; i++) k.M[i] = M[i] / Md;
    double asum = (a1 + a2) + a3;
    k.inv1masum = 1 - asum;
    k.inv1mbsum = 1 - asum;
    return k;
}

int slam_gaussian_taps(double sigma, double *w)
{
    int l = 4 * (int)std::ceil(sigma) + 1;
    int hw = l >> 1;
    double s = 0.0;
    for (int i = 0; i < l; i++) {
        double x = (double)(i - hw);
        w[i] = std::exp(-(x * x) / (2.0 * (sigma * sigma)));
        s += w[i];
    }
    for (int i = 0; i < l; i++) w[i] = w[i] / s;
    return l;
}
