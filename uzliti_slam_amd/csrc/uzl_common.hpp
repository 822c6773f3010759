// uzl_common.hpp — shared host-side plumbing of libuzl_mi355x.so (HIP runtime only, no torch).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/uzl_mi355x.h"

namespace uzl {

// A/B switches (alternative kernels, preconditioner variants, scheduling constants) exist for measurements and for the tests that
// prove the variants agree.  They are read from the environment ONLY in the diagnostic build (`make diag` -> libuzl_mi355x_diag.so,
// -DUZL_DIAG; tests and tests/diag scripts select it with UZL_LIB); in the shipped library every one of them is a compile-time
// constant.  Two run-time switches remain in both builds because a profiler / a debugging session needs them on the product:
// UZL_NO_GRAPH=1 (eager launches: rocprofv3's kernel tracer cannot follow hipGraph replays on this image) and UZL_VERBOSE=1.
// Test hooks (uzl_debug_*) exist in the diagnostic build only; the product library exports include/uzl_mi355x.h and nothing else.
#define UZL_DIAG_EXPORT __attribute__((visibility("default")))
#ifdef UZL_DIAG
inline bool diag_flag(const char* name) { return getenv(name) != nullptr; }
inline int diag_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
inline double diag_double(const char* name, double dflt) { const char* v = getenv(name); return v ? atof(v) : dflt; }
#else
constexpr bool diag_flag(const char*) { return false; }
constexpr int diag_int(const char*, int dflt) { return dflt; }
constexpr double diag_double(const char*, double dflt) { return dflt; }
#endif

struct HipError {
    hipError_t code;
    const char* what;
    const char* file;
    int line;
};

#define UZL_HIP(expr)                                                            \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) throw ::uzl::HipError{e_, #expr, __FILE__, __LINE__}; \
    } while (0)

// Format a caught HipError into a handle's last_error string; returns the status code.
inline int report(std::string& last_error, const HipError& e)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d in %s", (int)e.code, hipGetErrorString(e.code),
             e.file, e.line, e.what);
    last_error = buf;
    return (e.code == hipErrorOutOfMemory) ? UZL_ERR_OOM
           : (e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice) ? UZL_ERR_NO_DEVICE
                                                                             : UZL_ERR_HIP;
}

// Growable device buffer (never shrinks).  Plain hipMalloc: 288 GB of HBM, nothing is paged.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void reserve(size_t n, bool keep = false, hipStream_t s = nullptr)
    {
        if (n <= cap) return;
        size_t ncap = cap ? cap : 256;
        while (ncap < n) ncap *= 2;
        T* np = nullptr;
        UZL_HIP(hipMalloc((void**)&np, ncap * sizeof(T)));
        if (keep && p && cap) {
            UZL_HIP(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s));
            UZL_HIP(hipStreamSynchronize(s));
        }
        if (p) UZL_HIP(hipFree(p));
        p = np;
        cap = ncap;
    }
};

// Pinned host staging buffer.
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    void reserve(size_t n, unsigned flags = hipHostMallocDefault)
    {
        if (n <= cap) return;
        size_t ncap = cap ? cap : 256;
        while (ncap < n) ncap *= 2;
        if (p) UZL_HIP(hipHostFree(p));
        p = nullptr;
        UZL_HIP(hipHostMalloc((void**)&p, ncap * sizeof(T), flags));
        cap = ncap;
    }
};

// Per-kernel timing with HIP events on the handle's own stream (bench.py's `roofline.achieved`
// uses these; torch.cuda.Event would only see torch's current stream).
class KernelTimer {
public:
    bool on = false;
    ~KernelTimer()
    {
        for (auto& e : pool_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
    void reset()
    {
        used_ = 0;
        acc_.clear();
    }
    void begin(const char* name, hipStream_t s)
    {
        if (!on) return;
        if (used_ == pool_.size()) {
            Pair p;
            UZL_HIP(hipEventCreate(&p.a));
            UZL_HIP(hipEventCreate(&p.b));
            pool_.push_back(p);
        }
        pool_[used_].name = name;
        UZL_HIP(hipEventRecord(pool_[used_].a, s));
    }
    void end(hipStream_t s)
    {
        if (!on) return;
        UZL_HIP(hipEventRecord(pool_[used_].b, s));
        ++used_;
    }
    // Events for hipExtLaunchKernelGGL(..., start, stop, ...): they take the dispatch's own begin / end timestamps, i.e.
    // the kernel's execution time as rocprofv3 reports it (hipEventRecord pairs around a ~10 us kernel also count the
    // dispatch latency and read 1.4-2x too long).  Returns false when profiling is off.
    bool pair(const char* name, hipEvent_t* a, hipEvent_t* b)
    {
        if (!on) return false;
        if (used_ == pool_.size()) {
            Pair p;
            UZL_HIP(hipEventCreate(&p.a));
            UZL_HIP(hipEventCreate(&p.b));
            pool_.push_back(p);
        }
        pool_[used_].name = name;
        *a = pool_[used_].a; *b = pool_[used_].b;
        ++used_;
        return true;
    }
    // call after the stream has been synchronised
    void resolve()
    {
        if (!on) return;
        for (size_t i = 0; i < used_; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pool_[i].a, pool_[i].b) == hipSuccess) {
                auto& a = acc_[pool_[i].name];
                a.first += ms;
                a.second += 1;
            }
        }
        used_ = 0;
    }
    int report(int cap, const char** names, double* ms, int32_t* launches) const
    {
        int n = 0;
        for (auto& kv : acc_) {
            if (n >= cap) break;
            names[n] = kv.first;
            ms[n] = kv.second.first;
            launches[n] = kv.second.second;
            ++n;
        }
        return n;
    }

private:
    struct Pair { hipEvent_t a, b; const char* name; };
    std::vector<Pair> pool_;
    size_t used_ = 0;
    struct CStrLess { bool operator()(const char* x, const char* y) const { return strcmp(x, y) < 0; } };
    std::map<const char*, std::pair<double, int32_t>, CStrLess> acc_;
};


}  // namespace uzl
