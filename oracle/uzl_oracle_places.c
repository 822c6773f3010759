/*
 * uzl_oracle_places.c — CPU ORACLE (test infrastructure, NOT product code), see uzl_oracle.h.
 *
 * Restatement of the appearance-based candidate producer:
 *   FastLshTable / FastLshSet            place_recognition/src/lsh_set_recognizer.cpp:180-310
 *   LshSetRecognizer::{searchAndAddPlaceImpl, addPlaceImpl, searchImpl, removePlaceImpl}   :46-178
 *   PlaceRecognizer::{searchAndAddPlace, addPlace, searchPlace, removePlace}               place_recognizer.cpp:71-215
 * Tables: one per byte offset 0, kw, 2 kw, ... < 32 - kw + 1 (:258-263; descriptors are taken to be >= 32 bytes), key =
 * kw descriptor bytes as a little-endian u64, exact-match buckets of place indices (a place appears once per
 * descriptor that produced the key).
 * PARITY UNPINNED (no reference tests).  Unspecified in the reference: the order of equal similarities after the
 * unstable std::sort (:88, :148) -> (similarity desc, place index asc) here.
 */
#include "uzl_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct bucket { uint64_t key; int32_t* ids; int32_t n, cap; int used; } bucket;
typedef struct table { bucket* b; size_t cap, used; int start_byte; } table;

struct uzlo_places {
    uzlo_places_cfg cfg;
    table* tabs; int nt;
    int32_t place_count;
    int64_t* stamp; uint8_t* alive; int32_t cap_places;
    uint64_t* checked; size_t n_checked, cap_checked;      /* (neighbor << 32 | id) pairs already reported */
    int32_t* last_counts; int32_t n_last;
};

void uzlo_places_cfg_default(uzlo_places_cfg* c)
{
    c->key_width = 8; c->min_rows_to_add = 150; c->T = 10.0; c->k_nearest_neighbors = 10; c->min_time_gap = 5.0; c->device = 0;
}

static void table_init(table* t, int start) { t->cap = 1024; t->used = 0; t->start_byte = start; t->b = (bucket*)calloc(t->cap, sizeof(bucket)); }

uzlo_places* uzlo_places_create(const uzlo_places_cfg* cfg)
{
    uzlo_places* h = (uzlo_places*)calloc(1, sizeof(*h));
    if (cfg) h->cfg = *cfg; else uzlo_places_cfg_default(&h->cfg);
    const int kw = h->cfg.key_width;
    for (int i = 0; i < 32 - kw + 1; i += kw) h->nt++;                          /* FastLshSet::clear :258-263 */
    h->tabs = (table*)calloc((size_t)(h->nt > 0 ? h->nt : 1), sizeof(table));
    for (int i = 0; i < h->nt; i++) table_init(&h->tabs[i], i * kw);
    return h;
}

static void table_free(table* t) { for (size_t i = 0; i < t->cap; i++) free(t->b[i].ids); free(t->b); }

void uzlo_places_destroy(uzlo_places* h)
{
    if (!h) return;
    for (int i = 0; i < h->nt; i++) table_free(&h->tabs[i]);
    free(h->tabs); free(h->stamp); free(h->alive); free(h->checked); free(h->last_counts); free(h);
}

static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

static bucket* table_find(table* t, uint64_t key, int create)
{
    if (create && 2 * (t->used + 1) > t->cap) {
        table n; n.cap = t->cap * 2; n.used = 0; n.start_byte = t->start_byte; n.b = (bucket*)calloc(n.cap, sizeof(bucket));
        for (size_t i = 0; i < t->cap; i++) if (t->b[i].used) {
            size_t s = mix(t->b[i].key) & (n.cap - 1);
            while (n.b[s].used) s = (s + 1) & (n.cap - 1);
            n.b[s] = t->b[i]; n.used++;
        }
        free(t->b); *t = n;
    }
    size_t s = mix(key) & (t->cap - 1);
    while (t->b[s].used) { if (t->b[s].key == key) return &t->b[s]; s = (s + 1) & (t->cap - 1); }
    if (!create) return NULL;
    t->b[s].used = 1; t->b[s].key = key; t->b[s].ids = NULL; t->b[s].n = 0; t->b[s].cap = 0; t->used++;
    return &t->b[s];
}

static uint64_t key_of(const uint8_t* d, int start, int kw)                   /* long_long_array_u (:181-186) */
{
    uint64_t k = 0;
    for (int i = 0; i < kw; i++) k |= (uint64_t)d[start + i] << (8 * i);
    return k;
}

static void bucket_push(bucket* b, int32_t id)
{
    if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 4; b->ids = (int32_t*)realloc(b->ids, sizeof(int32_t) * (size_t)b->cap); }
    b->ids[b->n++] = id;
}

static void grow_places(uzlo_places* h)
{
    if (h->place_count < h->cap_places) return;
    h->cap_places = h->cap_places ? 2 * h->cap_places : 256;
    h->stamp = (int64_t*)realloc(h->stamp, sizeof(int64_t) * (size_t)h->cap_places);
    h->alive = (uint8_t*)realloc(h->alive, (size_t)h->cap_places);
}

/* thresholds, sort, self / time / knn / checked filters (lsh_set_recognizer.cpp:73-92, place_recognizer.cpp:87-114) */
static int32_t finish(uzlo_places* h, const int32_t* counts, int32_t nc, int64_t stamp_q, int32_t id_q, int32_t cap, int32_t* out)
{
    typedef struct { int32_t i; float s; } ms;
    ms* m = (ms*)malloc(sizeof(ms) * (size_t)(nc + 1));
    int32_t nm = 0;
    for (int32_t i = 0; i < nc; i++) if (counts[i] > 0) {
        const float sim = (float)counts[i] / (float)h->nt;
        if ((double)sim >= h->cfg.T) { m[nm].i = i; m[nm].s = sim; nm++; }
    }
    for (int32_t a = 1; a < nm; a++) {                                          /* similarity descending, index ascending */
        ms x = m[a]; int32_t b = a - 1;
        while (b >= 0 && (m[b].s < x.s)) { m[b + 1] = m[b]; b--; }
        m[b + 1] = x;
    }
    int32_t n_out = 0, pr = 0;
    for (int32_t a = 0; a < nm; a++) {
        const int32_t nb = m[a].i;
        if (nb >= h->place_count || !h->alive[nb]) continue;                    /* place_id_map_.left.find */
        if (!(fabs((double)(h->stamp[nb] - stamp_q) * 1e-9) > h->cfg.min_time_gap)) continue;
        pr++;
        {                                                                       /* checked_ (:104-110) */
            const uint64_t pair = ((uint64_t)(uint32_t)nb << 32) | (uint32_t)id_q;
            int seen = 0;
            for (size_t c = 0; c < h->n_checked; c++) if (h->checked[c] == pair) { seen = 1; break; }
            if (!seen) {
                if (h->n_checked == h->cap_checked) { h->cap_checked = h->cap_checked ? 2 * h->cap_checked : 256; h->checked = (uint64_t*)realloc(h->checked, 8 * h->cap_checked); }
                h->checked[h->n_checked++] = pair;
                if (n_out < cap) out[n_out] = nb;
                n_out++;
            }
        }
        if (pr >= h->cfg.k_nearest_neighbors) break;                            /* :95-98 */
    }
    free(m);
    return n_out;
}

static void keep_counts(uzlo_places* h, const int32_t* counts, int32_t n)
{
    h->last_counts = (int32_t*)realloc(h->last_counts, sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(h->last_counts, counts, sizeof(int32_t) * (size_t)n);
    h->n_last = n;
}

/* PlaceRecognizer::searchAndAddPlace (:71-117) */
int32_t uzlo_places_search_and_add(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t cap,
                                   int32_t* neighbors, int32_t* place_index)
{
    const int kw = h->cfg.key_width;
    const int32_t id = h->place_count;
    int32_t* counts = (int32_t*)calloc((size_t)id + 1, sizeof(int32_t));
    if (rows > h->cfg.min_rows_to_add) {                                        /* matchAndAdd (:214-233, :288-297) */
        for (int32_t r = 0; r < rows; r++) for (int t = 0; t < h->nt; t++) {
            const uint64_t key = key_of(desc + (size_t)r * bytes, h->tabs[t].start_byte, kw);
            if (__builtin_popcountll(key) > 3 * kw) {
                bucket* b = table_find(&h->tabs[t], key, 1);
                for (int32_t q = 0; q < b->n; q++) counts[b->ids[q]]++;
                bucket_push(b, id);
            }
        }
    } else {                                                                    /* match (:199-212) */
        for (int32_t r = 0; r < rows; r++) for (int t = 0; t < h->nt; t++) {
            bucket* b = table_find(&h->tabs[t], key_of(desc + (size_t)r * bytes, h->tabs[t].start_byte, kw), 0);
            if (b) for (int32_t q = 0; q < b->n; q++) counts[b->ids[q]]++;
        }
    }
    grow_places(h);
    h->stamp[id] = stamp_ns; h->alive[id] = 1; h->place_count++;                /* :84-85 */
    keep_counts(h, counts, id + 1);
    const int32_t n = finish(h, counts, id + 1, stamp_ns, id, cap, neighbors);
    free(counts);
    if (place_index) *place_index = id;
    return n;
}

/* PlaceRecognizer::addPlace (:131-142) -> addPlaceImpl (:96-118) */
int32_t uzlo_places_add(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns)
{
    const int kw = h->cfg.key_width;
    const int32_t id = h->place_count;
    if (rows > h->cfg.min_rows_to_add)
        for (int32_t r = 0; r < rows; r++) for (int t = 0; t < h->nt; t++)
            bucket_push(table_find(&h->tabs[t], key_of(desc + (size_t)r * bytes, h->tabs[t].start_byte, kw), 1), id);
    grow_places(h);
    h->stamp[id] = stamp_ns; h->alive[id] = 1; h->place_count++;
    return id;
}

/* PlaceRecognizer::searchPlace (:149-190) -> searchImpl (:120-157); id_q = place index of the querying node (for checked_) */
int32_t uzlo_places_search(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t id_q,
                           int32_t cap, int32_t* neighbors)
{
    const int kw = h->cfg.key_width;
    if (h->place_count == 0) return 0;
    int32_t* counts = (int32_t*)calloc((size_t)h->place_count, sizeof(int32_t));
    for (int32_t r = 0; r < rows; r++) for (int t = 0; t < h->nt; t++) {
        bucket* b = table_find(&h->tabs[t], key_of(desc + (size_t)r * bytes, h->tabs[t].start_byte, kw), 0);
        if (b) for (int32_t q = 0; q < b->n; q++) counts[b->ids[q]]++;
    }
    keep_counts(h, counts, h->place_count);
    const int32_t n = finish(h, counts, h->place_count, stamp_ns, id_q, cap, neighbors);
    free(counts);
    return n;
}

/* PlaceRecognizer::removePlace (:199-203) -> removePlaceImpl (:160-176), FastLshTable::remove (:235-249) */
void uzlo_places_remove(uzlo_places* h, int32_t id, const uint8_t* desc, int32_t rows, int32_t bytes)
{
    const int kw = h->cfg.key_width;
    if (id < 0 || id >= h->place_count || !h->alive[id]) return;
    for (int32_t r = 0; r < rows; r++) for (int t = 0; t < h->nt; t++) {
        bucket* b = table_find(&h->tabs[t], key_of(desc + (size_t)r * bytes, h->tabs[t].start_byte, kw), 0);
        if (!b) continue;
        int32_t w = 0;
        for (int32_t q = 0; q < b->n; q++) if (b->ids[q] != id) b->ids[w++] = b->ids[q];
        b->n = w;                                   /* an emptied bucket stays as an empty slot: same behaviour as erase */
    }
    h->alive[id] = 0;
}

int32_t uzlo_places_count(const uzlo_places* h) { return h->place_count; }
int32_t uzlo_places_num_tables(const uzlo_places* h) { return h->nt; }
int32_t uzlo_places_last_counts(const uzlo_places* h, int32_t cap, int32_t* counts)
{
    for (int32_t i = 0; i < h->n_last && i < cap; i++) counts[i] = h->last_counts[i];
    return h->n_last;
}
