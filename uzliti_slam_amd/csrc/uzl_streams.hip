// uzl_streams.hip — streams that do not stand in each other's way, chosen once per process.
//
// A HIP stream is served by one of the runtime's hardware queues (four per priority by default: GPU_MAX_HW_QUEUES), handed out at
// hipStreamCreate, least used first, and the driver puts the queues on the GPU's four compute pipes in the order they were first made.
// Streams on one queue run strictly one behind the other.  Streams on two queues of one PIPE do overlap for a single kernel - but two
// chains of dependent kernels, one on each, take 2.4x the time of one chain (measured: tests/diag/stream_overlap.py), worse than
// running them one after the other.  Which streams of a process collide either way depends on every stream it has made before: a batch
// whose rebuild stream sat on the solver stream's pipe lost 15 - 45 % (16 chain-like graphs 14.3 -> 20.9 ms), two launch sequences on one
// pipe ran 2x slower than one sequence.  Nothing tells a process where a stream landed, but it can be measured: a chain of 5 dependent
// 7-us kernels on one stream alone, then the same chain on both at once, TIMED ON THE DEVICE (first link's start to last link's
// end).  Round 6, three boxes, > 200 pairs (tests/diag/stream_overlap.py; links of one workgroup per CU, so that up to four chains fit
// the chip side by side): independent pairs 1.04 - 1.48x; one hardware queue 2.02x; two queues in each other's way 2.5 - 2.6x; nothing in
// between - the threshold is 1.75x, best of three (anything else on the GPU can only hold a run back).  (Links of 1000 workgroups: 1.01 -
// 1.17x / 2.01x / 2.4x.)  Rounds 4 - 5 timed 32 4-us links from the host: two chains cost the host twice the launches, so independent pairs
// read 1.05 - 1.56x depending on the box's CPU and the verdict was noise-driven (round 5's red GPU test).  A pair costs 0.1 - 0.2 ms
// (independent: one run, one host synchronize) to 0.5 ms (three runs); a candidate is measured against ALL its unmeasured partners at
// once first (one run if they are all independent); the chain alone is measured once per process.  The first launch on a fresh stream
// (the runtime sets its queue up: ~1 ms) is not counted as measuring.
//
// Round 4 measured at every uzl_pgo_batch_create and threw the rejected streams away.  Now the streams that have to run side by side
// (a solver handle's solver / rebuild pair, a batch's launch sequences and their rebuild streams) come from ONE POOL PER DEVICE that
// lives as long as the process:
//   * a pair of streams is measured at most once; the verdict is remembered, so the second batch of a process gets the first one's
//     streams back without a single probe launch, and a process's choices do not change while it runs;
//   * rejected streams stay in the pool (they keep their hardware queue, so the next stream lands elsewhere - and they are the first
//     candidates of the next lease with other partners);
//   * the search is bounded: at most kNewPerLease new streams per lease and kPoolMax per device; a lease that is `required` and finds
//     nothing returns nullptr and the caller takes the layout that needs no such stream (a batch: one launch sequence) - the same
//     layout UZL_STREAM_PROBE=0 selects without any measurement, for deployments that share the GPU (a probe under foreign load proves
//     nothing);
//   * the streams the other handles make for themselves (estimator, gate, places, radius) are registered, so that uzl_stream_stats
//     shows every long-lived stream of the library, and a lease prefers streams independent of the estimator's as well.
#include "uzl_streams.hpp"

#include <algorithm>
#include <chrono>

namespace uzl {
namespace {

constexpr int kPoolMax = 16;            // streams per device
constexpr int kNewPerLease = 6;         // new streams a single lease may add (three of the priority asked for, three of the other)
constexpr int kMaxDevices = 16;
constexpr int kKeepFree = 32;           // idle streams the pool keeps

// One link of a chain: every workgroup spins for `ticks` of the 100 MHz wall clock.  The chain's first link leaves its start in
// stamp[2 slot] and its last link its latest end in stamp[2 slot + 1] (DEVICE memory: a plain store by one lane, an atomicMax by
// every 64th workgroup - the clock only grows, so the end needs no reset), so a measurement is taken ON THE DEVICE: neither the
// host's launch rate (two chains cost it twice the launches: on a slow host that alone read as 1.4 - 1.56x with host timing) nor the
// latency of the synchronize is in it.
__global__ void chain_kernel(unsigned ticks, unsigned long long* stamp, int where, int slot)
{
    const unsigned long long t0 = wall_clock64();            // 100 MHz
    if (where == 1 && blockIdx.x == 0 && threadIdx.x == 0) stamp[2 * slot] = t0;
    while (wall_clock64() - t0 < ticks) {}
    if (where == 2 && (blockIdx.x & 63) == 63 && threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
}

// the stamps handed to the host without a copy engine and without a stream synchronize: eight system-scope stores into pinned memory,
// a fence, then the sequence word the host spins on (a hipMemcpyAsync + hipStreamSynchronize pair cost the host ~100 us per measurement)
__global__ void stamps_out_kernel(const unsigned long long* __restrict__ stamp, unsigned long long* __restrict__ host, unsigned long long seq, int n)
{
    if ((int)threadIdx.x < n) __hip_atomic_store(&host[threadIdx.x], stamp[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&host[n], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int kChainLen = 5;            // links per chain
constexpr unsigned kChainTicks = 700;   // 7 us a link: the host (2.8 us a launch) stays ahead of two chains
constexpr int kChainWgs = 256;             // one workgroup per CU: four chains side by side fit the chip (a group measurement)
// Independent iff two chains side by side take < kIndependentBelow x one chain, best of kProbeTries (classes: file header).
constexpr double kIndependentBelow = 1.75;
constexpr int kProbeTries = 3;

constexpr int kGroupMax = 4;            // chains side by side in one measurement
struct Probe {                          // per device: the stamps and the time of one chain alone (measured once per process)
    unsigned long long* stamp = nullptr;                        // device: {start, end} per chain
    unsigned long long* host = nullptr;                         // pinned: where they are read (+ the sequence word behind them)
    unsigned long long seq = 0;
    hipEvent_t ev[kGroupMax] = {nullptr, nullptr, nullptr, nullptr};
    double alone = 0.;
    double warm_ms = 0.;                                        // first launches on fresh streams (the runtime sets their queues up: ~1 ms each)
    std::vector<hipStream_t> warmed;
};

// ticks from the earliest first link's start to the latest last link's end, one chain on each of the n (<= kGroupMax) streams of q.
// ONE host synchronize per measurement: the other chains are ordered in front of the first stream's read-back by events.
double chain_ticks(Probe& pr, const hipStream_t* q, int n)
{
    if (!pr.stamp) {
        // once per process and device: two small allocations, four events and the first launch of this file's kernels (with which the
        // runtime loads the library's code object if nothing has been launched yet: ~8 ms that the first real kernel would pay otherwise).
        // Counted as set-up, not as measuring.
        const auto t_init = std::chrono::steady_clock::now();
        struct Done { Probe& p; std::chrono::steady_clock::time_point t; ~Done() { p.warm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } done{pr, t_init};
        if (hipMalloc((void**)&pr.stamp, 2 * kGroupMax * sizeof(unsigned long long)) != hipSuccess) { pr.stamp = nullptr; return -1.; }
        if (hipMemset(pr.stamp, 0, 2 * kGroupMax * sizeof(unsigned long long)) != hipSuccess) return -1.;
        if (hipHostMalloc((void**)&pr.host, (2 * kGroupMax + 1) * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { pr.host = nullptr; return -1.; }
        pr.host[2 * kGroupMax] = 0;
        for (hipEvent_t& e : pr.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return -1.; }
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, q[0], 1u, pr.stamp, 0, 0);
        hipLaunchKernelGGL(stamps_out_kernel, dim3(1), dim3(64), 0, q[0], pr.stamp, pr.host, 0ull, 2 * kGroupMax);
        (void)hipStreamSynchronize(q[0]);
        pr.warmed.push_back(q[0]);
    }
    if (!pr.host || n < 1 || n > kGroupMax) return -1.;
    for (int i = 0; i < n; i++)                                 // (the first launch on a stream sets its queue up: whoever uses the stream first pays that)
        if (std::find(pr.warmed.begin(), pr.warmed.end(), q[i]) == pr.warmed.end()) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, q[i], 1u, pr.stamp, 0, 0);
            (void)hipStreamSynchronize(q[i]);
            pr.warmed.push_back(q[i]);
            pr.warm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    for (int k = 0; k < kChainLen; k++) {
        const int where = k == 0 ? 1 : (k == kChainLen - 1 ? 2 : 0);
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(chain_kernel, dim3(kChainWgs), dim3(256), 0, q[i], kChainTicks, pr.stamp, where, i);
    }
    bool ok = true;
    for (int i = 1; i < n; i++) ok = ok && hipEventRecord(pr.ev[i], q[i]) == hipSuccess && hipStreamWaitEvent(q[0], pr.ev[i], 0) == hipSuccess;
    const unsigned long long seq = ++pr.seq;
    if (ok) hipLaunchKernelGGL(stamps_out_kernel, dim3(1), dim3(64), 0, q[0], pr.stamp, pr.host, seq, 2 * kGroupMax);
    ok = ok && hipGetLastError() == hipSuccess;
    if (ok) {                                                   // spin on the sequence word; a stream synchronize if it does not come (50 ms)
        const auto t0 = std::chrono::steady_clock::now();
        volatile unsigned long long* flag = pr.host + 2 * kGroupMax;
        for (long spin = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; spin++)
            if ((spin & 4095) == 4095 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 50.) {
                (void)hipStreamSynchronize(q[0]);
                ok = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
                break;
            }
    }
    if (!ok) { for (int i = 0; i < n; i++) (void)hipStreamSynchronize(q[i]); }
    (void)hipGetLastError();
    if (!ok) return -1.;
    const unsigned long long* t = pr.host;
    unsigned long long start = t[0], end = t[1];
    for (int i = 1; i < n; i++) { start = std::min(start, t[2 * i]); end = std::max(end, t[2 * i + 1]); }
    return end > start ? (double)(end - start) : -1.;
}

// n chains of dependent kernels side by side against one chain alone: the smallest of up to kProbeTries ratios (anything else on the
// GPU can only hold a run back), stopping at the first one below the threshold
double group_over_single(Probe& pr, const hipStream_t* q, int n)
{
    if (pr.alone <= 0.) pr.alone = std::min(chain_ticks(pr, q, 1), chain_ticks(pr, q, 1));
    if (pr.alone <= 0.) return 1e9;
    double best = 1e9;
    for (int t = 0; t < kProbeTries && best >= kIndependentBelow; t++) {
        const double p = chain_ticks(pr, q, n);
        if (p > 0.) best = std::min(best, p / pr.alone);
    }
    return best;
}
double pair_over_single(Probe& pr, hipStream_t a, hipStream_t b)
{
    const hipStream_t q[2] = {a, b};
    return group_over_single(pr, q, 2);
}

struct Pool {
    std::mutex mu;
    struct Entry { hipStream_t s; int prio; bool leased; };
    std::vector<Entry> pooled;
    struct Ext { hipStream_t s; bool beside; };
    std::vector<Ext> registered;
    std::map<std::pair<hipStream_t, hipStream_t>, bool> verdict;       // (lower pointer, higher pointer) -> independent
    StreamPoolStats st;
    Probe probe;

    bool known(hipStream_t a, hipStream_t b) const
    {
        return a == b || verdict.count(a < b ? std::make_pair(a, b) : std::make_pair(b, a)) != 0;
    }
    bool independent(hipStream_t a, hipStream_t b, double* ratio = nullptr)
    {
        if (a == b) return false;
        const auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
        auto it = verdict.find(key);
        if (it != verdict.end()) return it->second;
        const auto t0 = std::chrono::steady_clock::now();
        const double w0 = probe.warm_ms;
        const double r = pair_over_single(probe, a, b);
        const double took = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - (probe.warm_ms - w0);
        st.probe_ms += took;                                     // (queue set-up of a fresh stream is not the measurement's)
        const bool ok = r < kIndependentBelow;
        st.pairs_measured++; if (ok) st.pairs_independent++;
        static const bool dbg = diag_flag("UZL_STREAM_DBG");
        if (dbg) fprintf(stderr, "[uzl] stream pair %p %p: pair / single chain %.2f -> %s (%.3f ms, queue set-up %.3f ms)\n", (void*)a, (void*)b, r, ok ? "independent" : "in each other's way", took, probe.warm_ms - w0);
        if (ratio) *ratio = r;
        verdict[key] = ok;
        return ok;
    }
    void forget(hipStream_t s)
    {
        for (auto it = verdict.begin(); it != verdict.end();) it = (it->first.first == s || it->first.second == s) ? verdict.erase(it) : std::next(it);
        probe.warmed.erase(std::remove(probe.warmed.begin(), probe.warmed.end(), s), probe.warmed.end());
    }
};

Pool& pool_of(int device)
{
    static Pool pools[kMaxDevices];
    return pools[std::min(std::max(device, 0), kMaxDevices - 1)];
}

bool probing_enabled()
{
    static const bool on = [] { const char* v = getenv("UZL_STREAM_PROBE"); return !(v && v[0] == '0'); }();
    return on;
}

}  // namespace

hipStream_t stream_lease(int device, int priority, const std::vector<hipStream_t>& apart_from, bool required)
{
    Pool& P = pool_of(device);
    std::lock_guard<std::mutex> lock(P.mu);
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    std::vector<hipStream_t> hard;
    for (hipStream_t o : apart_from) if (o) hard.push_back(o);
    auto take = [&](Pool::Entry& e) { e.leased = true; return e.s; };
    auto any = [&]() -> hipStream_t {                          // not required: whatever the pool has, or a fresh stream nobody measured
        for (Pool::Entry& e : P.pooled) if (!e.leased && e.prio == priority) return take(e);
        for (Pool::Entry& e : P.pooled) if (!e.leased) return take(e);
        hipStream_t q = nullptr;                               // (kPoolMax bounds the search for independent streams, not the pool)
        if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, priority) != hipSuccess) return nullptr;
        P.pooled.push_back({q, priority, true});
        return q;
    };
    // the estimator's streams run long launch sequences beside a solve: avoided when that costs no more than the measurement
    std::vector<hipStream_t> soft;
    for (const Pool::Ext& x : P.registered) if (x.beside && soft.size() < 2 && std::find(hard.begin(), hard.end(), x.s) == hard.end()) soft.push_back(x.s);
    if (!probing_enabled() || (hard.empty() && soft.empty())) {
        if (required && !hard.empty()) { P.st.fallbacks++; return nullptr; }
        return any();
    }
    // a candidate against its hard partners: those it has not been measured with all at once (one chain on each, <= kGroupMax side by
    // side): if together they take no longer than one chain, every one of those pairs is independent - the common case costs one
    // measurement instead of one per partner; otherwise pair by pair
    auto fits = [&](hipStream_t q) {
        for (hipStream_t o : hard) if (P.known(o, q) && !P.independent(o, q)) return false;
        std::vector<hipStream_t> grp(1, q);
        for (hipStream_t o : hard) if (o != q && !P.known(o, q) && (int)grp.size() < kGroupMax) grp.push_back(o);
        if (grp.size() >= 3) {
            const auto t0 = std::chrono::steady_clock::now();
            const double w0 = P.probe.warm_ms;
            const double r = group_over_single(P.probe, grp.data(), (int)grp.size());
            P.st.probe_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - (P.probe.warm_ms - w0);
            if (r < kIndependentBelow)
                for (size_t i = 1; i < grp.size(); i++) { P.verdict[q < grp[i] ? std::make_pair(q, grp[i]) : std::make_pair(grp[i], q)] = true; P.st.pairs_measured++; P.st.pairs_independent++; }
        }
        for (hipStream_t o : hard) if (!P.independent(o, q)) return false;
        return true;
    };
    // (a registered stream belongs to another handle, whose worker may be enqueueing on it right now: a measurement under its load
    //  proves nothing and would stall it, so an unmeasured pair with a busy stream stays unmeasured and counts as no hit)
    auto soft_hits = [&](hipStream_t q) {
        int n = 0;
        for (hipStream_t o : soft) {
            if (!P.known(o, q) && hipStreamQuery(o) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (!P.independent(o, q)) n++;
        }
        return n;
    };
    int best = -1, best_hits = 1 << 30;
    auto consider = [&](int i) {
        if (!fits(P.pooled[(size_t)i].s)) return false;
        const int hits = soft_hits(P.pooled[(size_t)i].s);
        if (hits < best_hits) { best = i; best_hits = hits; }
        return hits == 0;
    };
    // 1) what the pool holds, the priority asked for first, in the order the streams were made (a process's choices repeat)
    bool done = false;
    for (int pass = 0; pass < 2 && !done; pass++)
        for (int i = 0; i < (int)P.pooled.size() && !done; i++)
            if (!P.pooled[(size_t)i].leased && (P.pooled[(size_t)i].prio == priority) == (pass == 0)) done = consider(i);
    // 2) new streams: the two priorities' queues sit on different pipes more often than not
    for (int attempt = 0; attempt < kNewPerLease && !done && (int)P.pooled.size() < kPoolMax; attempt++) {
        const int pr = attempt < kNewPerLease / 2 ? priority : (priority == 0 ? -1 : 0);
        hipStream_t q = nullptr;
        if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, pr) != hipSuccess) break;
        P.pooled.push_back({q, pr, false});
        done = consider((int)P.pooled.size() - 1);
        if (best >= 0 && !soft.empty() && attempt >= 1) break;       // (a stream that fits the hard set is enough after one more try for the soft one)
    }
    if (best >= 0) return take(P.pooled[(size_t)best]);
    P.st.fallbacks++;
    return required ? nullptr : any();
}

void stream_release(int device, hipStream_t s)
{
    if (!s) return;
    Pool& P = pool_of(device);
    std::lock_guard<std::mutex> lock(P.mu);
    size_t at = P.pooled.size(), free_now = 0;
    for (size_t i = 0; i < P.pooled.size(); i++) {
        if (P.pooled[i].s == s) at = i;
        else if (!P.pooled[i].leased) free_now++;
    }
    if (at == P.pooled.size()) return;
    // a process that once had hundreds of handles alive does not keep their streams for ever: beyond kKeepFree idle streams a returned
    // one is destroyed (its verdicts with it)
    if (free_now >= (size_t)kKeepFree) {
        P.forget(s);
        P.pooled.erase(P.pooled.begin() + (long)at);
        (void)hipSetDevice(device);
        (void)hipStreamDestroy(s);
        return;
    }
    P.pooled[at].leased = false;
}

void stream_register(int device, hipStream_t s, bool beside_solver)
{
    if (!s) return;
    Pool& P = pool_of(device);
    std::lock_guard<std::mutex> lock(P.mu);
    P.registered.push_back({s, beside_solver});
}

void stream_unregister(int device, hipStream_t s)
{
    if (!s) return;
    Pool& P = pool_of(device);
    std::lock_guard<std::mutex> lock(P.mu);
    for (size_t i = 0; i < P.registered.size(); i++)
        if (P.registered[i].s == s) { P.registered.erase(P.registered.begin() + (long)i); break; }
    P.forget(s);                                                // (the runtime may hand the address to another stream)
}

StreamPoolStats stream_pool_stats(int device)
{
    Pool& P = pool_of(device);
    std::lock_guard<std::mutex> lock(P.mu);
    StreamPoolStats s = P.st;
    s.pooled = (int32_t)P.pooled.size();
    s.leased = 0; for (const Pool::Entry& e : P.pooled) if (e.leased) s.leased++;
    s.registered = (int32_t)P.registered.size();
    return s;
}

}  // namespace uzl

extern "C" int uzl_stream_stats(int32_t device, int32_t* n_pooled, int32_t* n_leased, int32_t* n_registered, int32_t* pairs_measured,
                                int32_t* pairs_independent, int32_t* fallbacks, double* probe_ms)
{
    if (device < 0 || device >= uzl::kMaxDevices) return UZL_ERR_BAD_ARG;
    const uzl::StreamPoolStats s = uzl::stream_pool_stats(device);
    if (n_pooled) *n_pooled = s.pooled;
    if (n_leased) *n_leased = s.leased;
    if (n_registered) *n_registered = s.registered;
    if (pairs_measured) *pairs_measured = s.pairs_measured;
    if (pairs_independent) *pairs_independent = s.pairs_independent;
    if (fallbacks) *fallbacks = s.fallbacks;
    if (probe_ms) *probe_ms = s.probe_ms;
    return UZL_OK;
}

// test hook of the diagnostic build (tests/diag/stream_overlap.py, tests/test_zz_streams_gpu.py): n fresh streams - of priority
// `priority`, or of priorities 0 and -1 in turn for priority = 200 - go through the POOL'S OWN decision, one measurement per unordered
// pair: verdict[i * n + j] = 1 independent / 0 in each other's way / -1 on the diagonal, ratio100[i * n + j] = 100 x (two chains side
// by side / one chain) as that decision saw it (both symmetric by construction); probe_ms = what the measurements cost
#ifdef UZL_DIAG
extern "C" UZL_DIAG_EXPORT int uzl_debug_stream_pairs(int n, int priority, int32_t* verdict, int32_t* ratio100, double* probe_ms)
{
    if (n < 2 || n > 16 || !verdict) return UZL_ERR_BAD_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return UZL_ERR_HIP;
    uzl::Pool& P = uzl::pool_of(dev);
    std::lock_guard<std::mutex> lock(P.mu);
    std::vector<hipStream_t> q((size_t)n, nullptr);
    for (int i = 0; i < n; i++)
        if (hipStreamCreateWithPriority(&q[i], hipStreamNonBlocking, priority == 200 ? -(i & 1) : priority) != hipSuccess) return UZL_ERR_HIP;
    const double ms0 = P.st.probe_ms;
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            double r = -0.01;
            const int v = (i == j) ? -1 : (P.independent(q[i], q[j], &r) ? 1 : 0);
            verdict[i * n + j] = verdict[j * n + i] = v;
            if (ratio100) ratio100[i * n + j] = ratio100[j * n + i] = (int32_t)(100. * r + (r < 0 ? 0 : 0.5));
        }
    if (probe_ms) *probe_ms = P.st.probe_ms - ms0;
    for (hipStream_t s : q) { P.forget(s); (void)hipStreamDestroy(s); }
    return UZL_OK;
}
#endif
