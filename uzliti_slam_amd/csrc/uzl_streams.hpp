// uzl_streams.hpp — the library's long-lived HIP streams: one process-wide pool per device (uzl_streams.hip).
#pragma once
#include "uzl_common.hpp"

namespace uzl {

// A stream from the device's pool that does not stand in the way of any stream of `apart_from` (a pair is measured once per process and
// remembered), of priority `priority` (0 or -1) if one can be had, else of the other one.  The stream is the caller's alone until
// stream_release.  `required`: nullptr when no such stream exists within the pool's budget (the caller falls back to a layout that does
// not need one); otherwise any stream of the pool is better than none and the pair is simply served one behind the other.
hipStream_t stream_lease(int device, int priority, const std::vector<hipStream_t>& apart_from, bool required);
void stream_release(int device, hipStream_t s);

// Streams the other handles make for themselves (estimator, gate, places, radius) are entered so that the pool knows every
// long-lived stream of the library; `beside_solver`: long launch sequences run on it while a solve is in flight (the estimator's), so
// leases prefer streams that are independent of it too when that costs nothing.
void stream_register(int device, hipStream_t s, bool beside_solver);
void stream_unregister(int device, hipStream_t s);

struct StreamPoolStats { int32_t pooled = 0, leased = 0, registered = 0, pairs_measured = 0, pairs_independent = 0, fallbacks = 0; double probe_ms = 0.; };
StreamPoolStats stream_pool_stats(int device);

}  // namespace uzl
