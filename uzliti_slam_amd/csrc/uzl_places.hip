// uzl_places.hip — appearance-based candidate producer (kernels + host + C ABI uzl_places_*).
//
// Mirrors FastLshSet / LshSetRecognizer (place_recognition/src/lsh_set_recognizer.cpp:46-310) and the filters of
// PlaceRecognizer (place_recognizer.cpp:71-215).  HBM layout: per table an open-addressing hash (keys u64 + list head
// i32, <= 50 % load, rebuilt at twice the size when it fills); entries {place, next} in one append-only arena shared
// by all tables.  Kernels: one lane per (descriptor row, table) - key = key_width descriptor bytes; "count" walks
// the key's entry list and bumps the per-place counters with integer atomics (order-free, so exact); "insert" claims
// the slot with a 64-bit CAS and prepends an entry with an atomic exchange; "unlink" marks a place's entries dead.
// Byte / integer work bound by dependent HBM/L2 accesses (hash probe -> list walk); nothing to tile.
#include "uzl_common.hpp"
#include "uzl_streams.hpp"

#include <algorithm>
#include <cmath>
#include <new>
#include <unordered_set>

namespace uzl {

constexpr int kPlBlk = 256;
constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;       // a real all-ones key lives in the extra slot `cap`

struct PlEntry { int32_t place; int32_t next; };

struct PlTable {
    unsigned long long* keys;     // [cap + 1]
    int32_t* head;                // [cap + 1]  first entry of the key's list, -1 = none
    uint32_t mask;                // cap - 1
    int32_t start_byte;
};

struct PlArgs {
    PlTable tab[8];
    int32_t nt, rows, bytes, key_width, id, popcount_min;      // popcount_min: keys with fewer set bits + 1 are skipped (-1: none)
    const uint8_t* desc;
    PlEntry* entries;
    int32_t* n_entries;           // [1]
    int32_t* used;                // [8] occupied slots per table
    int32_t* counts;              // [places]
};

namespace {

__device__ __forceinline__ unsigned long long pl_mix(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ unsigned long long pl_key(const PlArgs& a, int row, int t)
{
    const uint8_t* d = a.desc + (size_t)row * a.bytes + a.tab[t].start_byte;
    unsigned long long k = 0;
    for (int i = 0; i < a.key_width; i++) k |= (unsigned long long)d[i] << (8 * i);     // long_long_array_u (:181-186)
    return k;
}
// slot of `key` in table t, -1 when absent
__device__ __forceinline__ int pl_find(const PlTable& T, unsigned long long key)
{
    if (key == kEmptyKey) return (T.head[T.mask + 1] >= 0 || T.keys[T.mask + 1] == 0ull) ? (int)(T.mask + 1) : -1;
    uint32_t s = (uint32_t)pl_mix(key) & T.mask;
    for (;;) {
        const unsigned long long k = T.keys[s];
        if (k == key) return (int)s;
        if (k == kEmptyKey) return -1;
        s = (s + 1) & T.mask;
    }
}

}  // namespace

// FastLshTable::match (:199-212) / the matching half of matchAndAdd (:214-233)
__global__ __launch_bounds__(kPlBlk) void places_count_kernel(PlArgs a)
{
    const int i = blockIdx.x * kPlBlk + threadIdx.x;
    if (i >= a.rows * a.nt) return;
    const int row = i / a.nt, t = i % a.nt;
    const unsigned long long key = pl_key(a, row, t);
    if (a.popcount_min >= 0 && !(__popcll(key) > a.popcount_min)) return;
    const int s = pl_find(a.tab[t], key);
    if (s < 0) return;
    for (int e = a.tab[t].head[s]; e >= 0; e = a.entries[e].next) {
        const int p = a.entries[e].place;
        if (p >= 0) atomicAdd(&a.counts[p], 1);
    }
}

// FastLshTable::add (:188-197) / the adding half of matchAndAdd
__global__ __launch_bounds__(kPlBlk) void places_insert_kernel(PlArgs a)
{
    const int i = blockIdx.x * kPlBlk + threadIdx.x;
    if (i >= a.rows * a.nt) return;
    const int row = i / a.nt, t = i % a.nt;
    const unsigned long long key = pl_key(a, row, t);
    if (a.popcount_min >= 0 && !(__popcll(key) > a.popcount_min)) return;
    const PlTable& T = a.tab[t];
    uint32_t s;
    if (key == kEmptyKey) {
        s = T.mask + 1;
        T.keys[s] = 0ull;                                         // marks the extra slot as in use (see pl_find)
    } else {
        s = (uint32_t)pl_mix(key) & T.mask;
        for (;;) {
            const unsigned long long old = atomicCAS(&T.keys[s], kEmptyKey, key);
            if (old == kEmptyKey) { atomicAdd(&a.used[t], 1); break; }
            if (old == key) break;
            s = (s + 1) & T.mask;
        }
    }
    const int e = atomicAdd(a.n_entries, 1);
    a.entries[e].place = a.id;
    a.entries[e].next = atomicExch(&T.head[s], e);
}

// FastLshTable::remove (:235-249): every entry of place `id` under the row's keys
__global__ __launch_bounds__(kPlBlk) void places_unlink_kernel(PlArgs a)
{
    const int i = blockIdx.x * kPlBlk + threadIdx.x;
    if (i >= a.rows * a.nt) return;
    const int row = i / a.nt, t = i % a.nt;
    const int s = pl_find(a.tab[t], pl_key(a, row, t));
    if (s < 0) return;
    for (int e = a.tab[t].head[s]; e >= 0; e = a.entries[e].next)
        if (a.entries[e].place == a.id) a.entries[e].place = -1;
}

// rebuild of one table at a larger size: every occupied slot moves with its list head
__global__ __launch_bounds__(kPlBlk) void places_rehash_kernel(PlTable from, PlTable to)
{
    const uint32_t i = blockIdx.x * kPlBlk + threadIdx.x;
    if (i > from.mask + 1) return;
    if (i == from.mask + 1) { to.keys[to.mask + 1] = from.keys[i]; to.head[to.mask + 1] = from.head[i]; return; }
    const unsigned long long key = from.keys[i];
    if (key == kEmptyKey) return;
    uint32_t s = (uint32_t)pl_mix(key) & to.mask;
    for (;;) {
        if (atomicCAS(&to.keys[s], kEmptyKey, key) == kEmptyKey) break;
        s = (s + 1) & to.mask;
    }
    to.head[s] = from.head[i];
}

}  // namespace uzl

using namespace uzl;

struct uzl_places {
    std::mutex mu;
    std::string last_error;
    uzl_places_cfg cfg;
    hipStream_t stream = nullptr;
    int nt = 0;
    struct Tab { DevBuf<unsigned long long> keys; DevBuf<int32_t> head; uint32_t cap = 0; int32_t used = 0; };
    Tab tab[8];
    DevBuf<PlEntry> entries; size_t entry_cap = 0; int64_t n_entries = 0;
    DevBuf<int32_t> d_n_entries, d_used, d_counts;
    DevBuf<uint8_t> d_desc;
    PinBuf<int32_t> h_counts, h_small;
    std::vector<int64_t> stamp; std::vector<uint8_t> alive;
    std::unordered_set<uint64_t> checked;
    std::vector<int32_t> last_counts;
};

namespace {

int fail(uzl_places* h, int code, const char* msg) { h->last_error = msg; return code; }

void alloc_table(uzl_places* h, uzl_places::Tab& T, uint32_t cap)
{
    T.keys.reserve((size_t)cap + 1); T.head.reserve((size_t)cap + 1);
    UZL_HIP(hipMemsetAsync(T.keys.p, 0xFF, ((size_t)cap + 1) * 8, h->stream));
    UZL_HIP(hipMemsetAsync(T.head.p, 0xFF, ((size_t)cap + 1) * 4, h->stream));       // -1
    T.cap = cap;
}

void fill_args(uzl_places* h, PlArgs& a, int rows, int bytes, int id, int popcount_min)
{
    memset(&a, 0, sizeof(a));
    const int kw = h->cfg.key_width;
    for (int t = 0; t < h->nt; t++) { a.tab[t].keys = h->tab[t].keys.p; a.tab[t].head = h->tab[t].head.p; a.tab[t].mask = h->tab[t].cap - 1; a.tab[t].start_byte = t * kw; }
    a.nt = h->nt; a.rows = rows; a.bytes = bytes; a.key_width = kw; a.id = id; a.popcount_min = popcount_min;
    a.desc = h->d_desc.p; a.entries = h->entries.p; a.n_entries = h->d_n_entries.p; a.used = h->d_used.p; a.counts = h->d_counts.p;
}

// room for `rows` more keys per table and rows * nt more entries
void ensure_room(uzl_places* h, int rows)
{
    hipStream_t s = h->stream;
    for (int t = 0; t < h->nt; t++) {
        uzl_places::Tab& T = h->tab[t];
        if ((size_t)T.used + (size_t)rows <= T.cap / 2) continue;
        uint32_t ncap = T.cap;
        while ((size_t)T.used + (size_t)rows > ncap / 2) ncap *= 2;
        uzl_places::Tab N;
        alloc_table(h, N, ncap);
        PlTable from{T.keys.p, T.head.p, T.cap - 1, 0}, to{N.keys.p, N.head.p, ncap - 1, 0};
        hipLaunchKernelGGL(places_rehash_kernel, dim3((T.cap + 1 + kPlBlk) / kPlBlk), dim3(kPlBlk), 0, s, from, to);
        UZL_HIP(hipStreamSynchronize(s));
        std::swap(T.keys.p, N.keys.p); std::swap(T.keys.cap, N.keys.cap);
        std::swap(T.head.p, N.head.p); std::swap(T.head.cap, N.head.cap);
        T.cap = ncap;
    }
    const size_t need = (size_t)h->n_entries + (size_t)rows * h->nt;
    if (need > h->entry_cap) {
        size_t ncap = std::max<size_t>(h->entry_cap, 1 << 16);
        while (ncap < need) ncap *= 2;
        h->entries.reserve(ncap, true, s);
        h->entry_cap = ncap;
    }
}

void upload_desc(uzl_places* h, const uint8_t* desc, int rows, int bytes)
{
    h->d_desc.reserve(std::max<size_t>((size_t)rows * bytes, 1));
    if (rows > 0) UZL_HIP(hipMemcpyAsync(h->d_desc.p, desc, (size_t)rows * bytes, hipMemcpyHostToDevice, h->stream));
}

void run_count(uzl_places* h, int rows, int bytes, int n_counts, int popcount_min)
{
    hipStream_t s = h->stream;
    h->d_counts.reserve(std::max(n_counts, 1)); h->h_counts.reserve(std::max(n_counts, 1));
    UZL_HIP(hipMemsetAsync(h->d_counts.p, 0, (size_t)std::max(n_counts, 1) * 4, s));
    if (rows > 0) {
        PlArgs a; fill_args(h, a, rows, bytes, -1, popcount_min);
        hipLaunchKernelGGL(places_count_kernel, dim3((rows * h->nt + kPlBlk - 1) / kPlBlk), dim3(kPlBlk), 0, s, a);
    }
    UZL_HIP(hipMemcpyAsync(h->h_counts.p, h->d_counts.p, (size_t)std::max(n_counts, 1) * 4, hipMemcpyDeviceToHost, s));
}

void run_insert(uzl_places* h, int rows, int bytes, int id, int popcount_min)
{
    if (rows <= 0) return;
    hipStream_t s = h->stream;
    ensure_room(h, rows);
    PlArgs a; fill_args(h, a, rows, bytes, id, popcount_min);
    hipLaunchKernelGGL(places_insert_kernel, dim3((rows * h->nt + kPlBlk - 1) / kPlBlk), dim3(kPlBlk), 0, s, a);
    UZL_HIP(hipMemcpyAsync(h->h_small.p, h->d_used.p, 8 * 4, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipMemcpyAsync(h->h_small.p + 8, h->d_n_entries.p, 4, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipStreamSynchronize(s));
    UZL_HIP(hipGetLastError());
    for (int t = 0; t < h->nt; t++) h->tab[t].used = h->h_small.p[t];
    h->n_entries = h->h_small.p[8];
}

// thresholds, sort and the self / time / knn / reported-once filters (lsh_set_recognizer.cpp:73-92, place_recognizer.cpp:87-114)
int32_t finish(uzl_places* h, int32_t nc, int64_t stamp_q, int32_t id_q, int32_t cap, int32_t* out)
{
    h->last_counts.assign(h->h_counts.p, h->h_counts.p + nc);
    std::vector<std::pair<int32_t, float>> m;
    for (int32_t i = 0; i < nc; i++) if (h->h_counts.p[i] > 0) {
        const float sim = (float)h->h_counts.p[i] / (float)h->nt;
        if ((double)sim >= h->cfg.T) m.push_back({i, sim});
    }
    std::stable_sort(m.begin(), m.end(), [](const std::pair<int32_t, float>& a, const std::pair<int32_t, float>& b) { return a.second > b.second; });
    int32_t n_out = 0, pr = 0;
    for (const auto& x : m) {
        const int32_t nb = x.first;
        if (nb >= (int32_t)h->alive.size() || !h->alive[nb]) continue;
        if (!(std::fabs((double)(h->stamp[nb] - stamp_q) * 1e-9) > h->cfg.min_time_gap)) continue;
        pr++;
        const uint64_t pair = ((uint64_t)(uint32_t)nb << 32) | (uint32_t)id_q;
        if (h->checked.insert(pair).second) { if (n_out < cap && out) out[n_out] = nb; n_out++; }
        if (pr >= h->cfg.k_nearest_neighbors) break;
    }
    return n_out;
}

int check_desc(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes)
{
    if (rows < 0 || (rows > 0 && !desc)) return fail(h, UZL_ERR_BAD_ARG, "null descriptors");
    if (bytes < 32 && rows > 0) return fail(h, UZL_ERR_BAD_ARG, "descriptors must be at least 32 bytes (tables cover byte offsets below 32)");
    return UZL_OK;
}

}  // namespace

#define UZL_GUARD_BEGIN(h)                       \
    if (!(h)) return UZL_ERR_BAD_ARG;            \
    std::lock_guard<std::mutex> lock_((h)->mu);  \
    try {
#define UZL_GUARD_END(h)                                                             \
    } catch (const ::uzl::HipError& e) { return ::uzl::report((h)->last_error, e); } \
    catch (const std::bad_alloc&) { (h)->last_error = "host out of memory"; return UZL_ERR_OOM; } \
    catch (...) { (h)->last_error = "unexpected exception"; return UZL_ERR_HIP; }

extern "C" {

void uzl_places_cfg_default(uzl_places_cfg* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->key_width = 8; c->min_rows_to_add = 150; c->T = 10.0; c->k_nearest_neighbors = 10; c->device = 0; c->min_time_gap = 5.0;
}

int uzl_places_create(const uzl_places_cfg* cfg, uzl_places** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_places_cfg c;
    if (cfg) c = *cfg; else uzl_places_cfg_default(&c);
    if (c.key_width < 1 || c.key_width > 8) return UZL_ERR_BAD_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return UZL_ERR_NO_DEVICE;     // no CPU fallback
    if (c.device < 0 || c.device >= count) return UZL_ERR_NO_DEVICE;
    uzl_places* h = new (std::nothrow) uzl_places();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    for (int i = 0; i < 32 - c.key_width + 1; i += c.key_width) h->nt++;                     // FastLshSet::clear :258-263
    if (h->nt > 8) h->nt = 8;
    try {
        UZL_HIP(hipSetDevice(c.device));
        UZL_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        stream_register(c.device, h->stream, false);
        for (int t = 0; t < h->nt; t++) alloc_table(h, h->tab[t], 1u << 16);
        h->d_n_entries.reserve(1); h->d_used.reserve(8); h->h_small.reserve(16);
        UZL_HIP(hipMemsetAsync(h->d_n_entries.p, 0, 4, h->stream));
        UZL_HIP(hipMemsetAsync(h->d_used.p, 0, 32, h->stream));
        h->entries.reserve(1 << 16); h->entry_cap = 1 << 16;
        UZL_HIP(hipStreamSynchronize(h->stream));
    } catch (...) { delete h; return UZL_ERR_HIP; }
    *out = h;
    return UZL_OK;
}

void uzl_places_destroy(uzl_places* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_unregister(h->cfg.device, h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

const char* uzl_places_last_error(uzl_places* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_places_search_and_add(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t cap,
                              int32_t* neighbors, int32_t* n_neighbors, int32_t* place_index)
{
    UZL_GUARD_BEGIN(h)
    if (int rc = check_desc(h, desc, rows, bytes)) return rc;
    if (!n_neighbors || cap < 0 || (cap > 0 && !neighbors)) return fail(h, UZL_ERR_BAD_ARG, "bad outputs");
    UZL_HIP(hipSetDevice(h->cfg.device));
    const int32_t id = (int32_t)h->stamp.size();
    upload_desc(h, desc, rows, bytes);
    const bool index_it = rows > h->cfg.min_rows_to_add;                       // :66-70
    const int pc = index_it ? 3 * h->cfg.key_width : -1;                        // matchAndAdd skips sparse keys (:222), match does not
    run_count(h, rows, bytes, id + 1, pc);
    if (index_it) run_insert(h, rows, bytes, id, pc);                           // (entries of this frame only ever hit its own counter)
    UZL_HIP(hipStreamSynchronize(h->stream));
    h->stamp.push_back(stamp_ns); h->alive.push_back(1);                        // place_id_map_.insert, place_count_++ (:84-85)
    *n_neighbors = finish(h, id + 1, stamp_ns, id, cap, neighbors);
    if (place_index) *place_index = id;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_places_add(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t* place_index)
{
    UZL_GUARD_BEGIN(h)
    if (int rc = check_desc(h, desc, rows, bytes)) return rc;
    UZL_HIP(hipSetDevice(h->cfg.device));
    const int32_t id = (int32_t)h->stamp.size();
    if (rows > h->cfg.min_rows_to_add) {                                        // addPlaceImpl :111-114
        upload_desc(h, desc, rows, bytes);
        run_insert(h, rows, bytes, id, -1);
    }
    h->stamp.push_back(stamp_ns); h->alive.push_back(1);
    if (place_index) *place_index = id;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_places_search(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t query_place,
                      int32_t cap, int32_t* neighbors, int32_t* n_neighbors)
{
    UZL_GUARD_BEGIN(h)
    if (int rc = check_desc(h, desc, rows, bytes)) return rc;
    if (!n_neighbors || cap < 0 || (cap > 0 && !neighbors)) return fail(h, UZL_ERR_BAD_ARG, "bad outputs");
    *n_neighbors = 0;
    const int32_t n = (int32_t)h->stamp.size();
    if (n == 0) return UZL_OK;                                                  // place_recognizer.cpp:152-155
    UZL_HIP(hipSetDevice(h->cfg.device));
    upload_desc(h, desc, rows, bytes);
    run_count(h, rows, bytes, n, -1);
    UZL_HIP(hipStreamSynchronize(h->stream));
    *n_neighbors = finish(h, n, stamp_ns, query_place, cap, neighbors);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_places_remove(uzl_places* h, int32_t id, const uint8_t* desc, int32_t rows, int32_t bytes)
{
    UZL_GUARD_BEGIN(h)
    if (int rc = check_desc(h, desc, rows, bytes)) return rc;
    if (id < 0 || id >= (int32_t)h->stamp.size() || !h->alive[id]) return UZL_OK;   // "tried to remove a non-existing place"
    UZL_HIP(hipSetDevice(h->cfg.device));
    if (rows > 0) {
        upload_desc(h, desc, rows, bytes);
        PlArgs a; fill_args(h, a, rows, bytes, id, -1);
        hipLaunchKernelGGL(places_unlink_kernel, dim3((rows * h->nt + kPlBlk - 1) / kPlBlk), dim3(kPlBlk), 0, h->stream, a);
        UZL_HIP(hipStreamSynchronize(h->stream));
    }
    h->alive[id] = 0;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_places_count(uzl_places* h)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return (int)h->stamp.size();
}

int uzl_places_last_counts(uzl_places* h, int32_t cap, int32_t* counts)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    for (size_t i = 0; i < h->last_counts.size() && (int32_t)i < cap; i++) counts[i] = h->last_counts[i];
    return (int)h->last_counts.size();
}

}  // extern "C"
