/*
 * uzl_oracle_filter.c — CPU ORACLE (test infrastructure, NOT product code), see uzl_oracle.h.
 *
 * Restatement of TransformationFilter / EdgeCluster
 *   transformation_estimation/src/transformation_filter.cpp:43-350
 *   transformation_estimation/include/transformation_estimation/transformation_filter.h:28-106
 * as called from G2oOptimizer::addGraphImpl (graph_optimization/src/g2o_optimizer.cpp:74-103).
 *
 * PARITY UNPINNED (no reference tests / fixtures for it; not buildable here).  Deliberate choices where
 * the reference is unspecified:
 *   - EdgeCluster::edges_ is an unordered_map: iteration order (P/Q column order at :253-268, the order
 *     of valid_cluster_edges at :299-303) is unspecified -> insertion order here (an id inserted again
 *     keeps its position, like a map slot);
 *   - std::sort at :313,:321 is unstable -> stable here (equal scores keep cluster order).  Both sorts
 *     use EdgeScoreSort in the reference (:321 says "timestamp" but passes EdgeScoreSort) - kept;
 *   - EdgeCluster::changed_ is not initialised by the constructor (:43-59) -> false here; it cannot matter
 *     for min_size >= 2 (a cluster grows only through addEdge/merge, which set it);
 *   - sensor_transforms_[name] of an unknown name default-constructs an (uninitialised) Isometry3d ->
 *     index -1 = identity here;
 *   - std::rand -> the counter-based stream of uzl_oracle_match.c, job id = cluster uid * 2^20 + evaluation
 *     counter of that cluster (uids count clusters in creation order from 0).
 */
#include "uzl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct edata {
    uint64_t key;
    double pos_from[12], pos_to[12];
    int64_t t_from, t_to;
    /* the stored SlamEdge */
    double score; int32_t edge_valid; int32_t sensor_from, sensor_to;
    double transform[12], disp_from[12], disp_to[12];
    uint8_t valid;                               /* EdgeData::valid_ */
} edata;

typedef struct cluster {
    uint64_t uid;
    int64_t fs, fe, ts, te;                      /* cluster_from_start_/end_, cluster_to_start_/end_ */
    int32_t changed, consensus, evals;
    edata* e; int32_t n, cap;
    double* lastP; double* lastQ; double lastT[12]; int32_t last_n, last_ransac;
} cluster;

typedef struct keyrec { uint64_t key; cluster** cl; int32_t n, cap; } keyrec;

struct uzlo_filter {
    uzlo_filter_cfg cfg;
    cluster** clusters; int32_t nc, capc;
    keyrec* keys; int32_t nk, capk;              /* sorted by key: TransformationFilter::edges_ */
    double* sensors; int32_t n_sensors;
    uint64_t next_uid;
};

static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

/* Isometry3d product / inverse in a fixed operation order (the HIP kernel repeats it) */
static void iso_mul(const double* A, const double* B, double* C)
{
    double o[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            o[r * 4 + c] = (A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c];
        o[r * 4 + 3] = ((A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7]) + A[r * 4 + 2] * B[11]) + A[r * 4 + 3];
    }
    memcpy(C, o, sizeof(o));
}

static void iso_inv(const double* A, double* C)
{
    double o[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) o[r * 4 + c] = A[c * 4 + r];
        o[r * 4 + 3] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11]);
    }
    memcpy(C, o, sizeof(o));
}

void uzlo_filter_cfg_default(uzlo_filter_cfg* c)
{
    c->max_dt = 5.0; c->min_size = 8.0; c->max_cluster_size = 100; c->ransac_iterations = 200;
    c->max_error = 0.3; c->min_time_span = 2.0; c->max_edges = 5; c->device = 0; c->seed = 0;
}

uzlo_filter* uzlo_filter_create(const uzlo_filter_cfg* cfg)
{
    uzlo_filter* f = (uzlo_filter*)calloc(1, sizeof(*f));
    if (cfg) f->cfg = *cfg; else uzlo_filter_cfg_default(&f->cfg);
    return f;
}

static void cluster_free(cluster* c)
{
    free(c->e); free(c->lastP); free(c->lastQ); free(c);
}

void uzlo_filter_destroy(uzlo_filter* f)
{
    if (!f) return;
    /* clusters are owned by clusters_; an erased-but-referenced cluster cannot exist (see remove) */
    for (int32_t i = 0; i < f->nc; i++) cluster_free(f->clusters[i]);
    for (int32_t i = 0; i < f->nk; i++) free(f->keys[i].cl);
    free(f->clusters); free(f->keys); free(f->sensors); free(f);
}

void uzlo_filter_set_sensors(uzlo_filter* f, int32_t n, const double* sensors)
{
    free(f->sensors);
    f->sensors = (double*)malloc(sizeof(double) * 12 * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(f->sensors, sensors, sizeof(double) * 12 * (size_t)n);
    f->n_sensors = n;
}

/* ---- key table (edges_) ---- */
static int32_t key_find(const uzlo_filter* f, uint64_t key, int* found)
{
    int32_t lo = 0, hi = f->nk;
    while (lo < hi) { int32_t mid = (lo + hi) / 2; if (f->keys[mid].key < key) lo = mid + 1; else hi = mid; }
    *found = lo < f->nk && f->keys[lo].key == key;
    return lo;
}

static keyrec* key_insert(uzlo_filter* f, int32_t pos, uint64_t key)
{
    if (f->nk == f->capk) { f->capk = f->capk ? 2 * f->capk : 64; f->keys = (keyrec*)realloc(f->keys, sizeof(keyrec) * (size_t)f->capk); }
    memmove(f->keys + pos + 1, f->keys + pos, sizeof(keyrec) * (size_t)(f->nk - pos));
    f->nk++;
    keyrec* k = &f->keys[pos];
    k->key = key; k->cl = NULL; k->n = 0; k->cap = 0;
    return k;
}

static void key_push(keyrec* k, cluster* c)
{
    if (k->n == k->cap) { k->cap = k->cap ? 2 * k->cap : 2; k->cl = (cluster**)realloc(k->cl, sizeof(cluster*) * (size_t)k->cap); }
    k->cl[k->n++] = c;
}

/* ---- EdgeCluster ---- */
static int32_t cl_find(const cluster* c, uint64_t key)
{
    for (int32_t i = 0; i < c->n; i++) if (c->e[i].key == key) return i;
    return -1;
}

static void edata_fill(edata* d, const uzlo_filter_edge* e, int64_t tf, int64_t tt)
{
    d->key = e->key;
    memcpy(d->pos_from, e->pose_from, sizeof(d->pos_from)); memcpy(d->pos_to, e->pose_to, sizeof(d->pos_to));
    d->t_from = tf; d->t_to = tt;
    d->score = e->matching_score; d->edge_valid = e->valid; d->sensor_from = e->sensor_from; d->sensor_to = e->sensor_to;
    memcpy(d->transform, e->transform, sizeof(d->transform));
    memcpy(d->disp_from, e->displacement_from, sizeof(d->disp_from));
    memcpy(d->disp_to, e->displacement_to, sizeof(d->disp_to));
    d->valid = e->valid ? 1 : 0;
}

/* edges_[edge.id_] = data  (:55-56, :72-73): a new id appends, a present id is overwritten in place */
static void cl_put(cluster* c, const uzlo_filter_edge* e, int64_t tf, int64_t tt)
{
    int32_t at = cl_find(c, e->key);
    if (at < 0) {
        if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 8; c->e = (edata*)realloc(c->e, sizeof(edata) * (size_t)c->cap); }
        at = c->n++;
    }
    edata_fill(&c->e[at], e, tf, tt);
    if (e->valid) c->consensus++;                                            /* :57-59, :74-76 */
}

static cluster* cl_new(uzlo_filter* f, const uzlo_filter_edge* e, int64_t tf, int64_t tt)   /* :43-60 */
{
    cluster* c = (cluster*)calloc(1, sizeof(*c));
    c->uid = f->next_uid++;
    c->fs = c->fe = tf; c->ts = c->te = tt;
    memcpy(c->lastT, I12, sizeof(I12));
    cl_put(c, e, tf, tt);
    return c;
}

static void cl_add(cluster* c, const uzlo_filter_edge* e, int64_t tf, int64_t tt)           /* :62-77 */
{
    if (tf < c->fs) c->fs = tf;
    if (tf > c->fe) c->fe = tf;
    if (tt < c->ts) c->ts = tt;
    if (tt > c->te) c->te = tt;
    c->changed = 1;
    cl_put(c, e, tf, tt);
}

static double dsec(int64_t a, int64_t b) { return (double)(a - b) * 1e-9; }  /* (ros::Time - ros::Time).toSec() */

static int cl_is_part(const cluster* c, int64_t tf, int64_t tt, double max_dt)              /* :109-115 */
{
    return dsec(tf, c->fs) > -max_dt && dsec(tf, c->fe) < max_dt && dsec(tt, c->ts) > -max_dt && dsec(tt, c->te) < max_dt;
}

static void cl_merge(cluster* a, const cluster* b)                                          /* :98-107 */
{
    if (b->fs < a->fs) a->fs = b->fs;
    if (b->fe > a->fe) a->fe = b->fe;
    if (b->ts < a->ts) a->ts = b->ts;
    if (b->te > a->te) a->te = b->te;
    a->changed = 1;
    a->consensus += b->consensus;
    for (int32_t i = 0; i < b->n; i++) {                 /* unordered_map::insert(range): present ids are kept */
        if (cl_find(a, b->e[i].key) >= 0) continue;
        if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->e = (edata*)realloc(a->e, sizeof(edata) * (size_t)a->cap); }
        a->e[a->n++] = b->e[i];
    }
}

static void clusters_erase(uzlo_filter* f, int32_t idx)
{
    memmove(f->clusters + idx, f->clusters + idx + 1, sizeof(cluster*) * (size_t)(f->nc - idx - 1));
    f->nc--;
}

/* TransformationFilter::add (:138-207) */
void uzlo_filter_add(uzlo_filter* f, int32_t n_edges, const uzlo_filter_edge* edges)
{
    for (int32_t q = 0; q < n_edges; q++) {
        const uzlo_filter_edge* e = &edges[q];
        int found;
        int32_t pos = key_find(f, e->key, &found);
        if (found) {                                                          /* :140-146 updateEdge (:79-87) */
            keyrec* k = &f->keys[pos];
            for (int32_t i = 0; i < k->n; i++) {
                int32_t at = cl_find(k->cl[i], e->key);
                if (at < 0) continue;
                edata* d = &k->cl[i]->e[at];
                memcpy(d->pos_from, e->pose_from, sizeof(d->pos_from)); memcpy(d->pos_to, e->pose_to, sizeof(d->pos_to));
                d->score = e->matching_score; d->edge_valid = e->valid; d->sensor_from = e->sensor_from; d->sensor_to = e->sensor_to;
                memcpy(d->transform, e->transform, sizeof(d->transform));
                memcpy(d->disp_from, e->displacement_from, sizeof(d->disp_from));
                memcpy(d->disp_to, e->displacement_to, sizeof(d->disp_to));
            }
            continue;
        }
        for (int32_t a = 0; a < e->n_stamps_from; a++) {
            for (int32_t b = 0; b < e->n_stamps_to; b++) {
                const int64_t tf = e->stamps_from_ns[a], tt = e->stamps_to_ns[b];
                int32_t* matched = (int32_t*)malloc(sizeof(int32_t) * (size_t)(f->nc + 1));
                int32_t nm = 0;
                for (int32_t i = 0; i < f->nc; i++)                           /* :152-159 */
                    if (f->clusters[i]->n < f->cfg.max_cluster_size && cl_is_part(f->clusters[i], tf, tt, f->cfg.max_dt)) matched[nm++] = i;
                int fnd;
                int32_t kp = key_find(f, e->key, &fnd);
                keyrec* k = fnd ? &f->keys[kp] : key_insert(f, kp, e->key);
                if (nm == 0) {                                                /* :162-165 */
                    cluster* c = cl_new(f, e, tf, tt);
                    if (f->nc == f->capc) { f->capc = f->capc ? 2 * f->capc : 64; f->clusters = (cluster**)realloc(f->clusters, sizeof(cluster*) * (size_t)f->capc); }
                    f->clusters[f->nc++] = c;
                    key_push(k, c);
                } else {
                    cluster* c0 = f->clusters[matched[0]];
                    cl_add(c0, e, tf, tt);                                    /* :168 */
                    key_push(k, c0);                                          /* :169 */
                    for (int32_t i = nm - 1; i >= 1; i--) {                   /* :172-199 */
                        cluster* ci = f->clusters[matched[i]];
                        if (c0->n + ci->n < f->cfg.max_cluster_size) {
                            /* :175-182 repoints the FIRST listing of ci in each of its edges' cluster lists; a second
                             * listing (same edge put into ci by two stamp pairs) keeps the merged-away object alive in
                             * the reference, where it is no longer in clusters_ and can never influence a result.
                             * Repointing every listing to c0 is observably the same (remove() skips repeated listings). */
                            for (int32_t m = 0; m < ci->n; m++) {
                                int f2;
                                int32_t p2 = key_find(f, ci->e[m].key, &f2);
                                if (!f2) continue;
                                keyrec* k2 = &f->keys[p2];
                                for (int32_t u = 0; u < k2->n; u++) if (k2->cl[u] == ci) k2->cl[u] = c0;
                            }
                            cl_merge(c0, ci);
                            clusters_erase(f, matched[i]);
                            cluster_free(ci);
                        }
                    }
                }
                free(matched);
            }
        }
    }
}

/* TransformationFilter::remove (:209-220), EdgeCluster::removeEdge (:89-96) */
void uzlo_filter_remove(uzlo_filter* f, int32_t n_keys, const uint64_t* keys)
{
    for (int32_t q = 0; q < n_keys; q++) {
        int found;
        int32_t pos = key_find(f, keys[q], &found);
        if (!found) continue;
        keyrec k = f->keys[pos];
        for (int32_t i = 0; i < k.n; i++) {
            cluster* c = k.cl[i];
            int dup = 0;
            for (int32_t j = 0; j < i; j++) if (k.cl[j] == c) dup = 1;          /* second listing of a cluster: nothing left to do */
            if (dup) continue;
            int32_t at = cl_find(c, keys[q]);
            if (at >= 0) {
                if (c->e[at].valid) c->consensus--;
                memmove(c->e + at, c->e + at + 1, sizeof(edata) * (size_t)(c->n - at - 1));
                c->n--;
            }
            if (c->n == 0) {
                for (int32_t u = 0; u < f->nc; u++) if (f->clusters[u] == c) { clusters_erase(f, u); break; }
                cluster_free(c);
            }
        }
        free(k.cl);
        memmove(f->keys + pos, f->keys + pos + 1, sizeof(keyrec) * (size_t)(f->nk - pos - 1));
        f->nk--;
    }
}

int32_t uzlo_filter_all_edges(const uzlo_filter* f, int32_t cap, uint64_t* keys)            /* :343-350 */
{
    for (int32_t i = 0; i < f->nk && i < cap; i++) keys[i] = f->keys[i].key;
    return f->nk;
}

/* the two world-frame end poses of one edge (:253-262); only the translations are used */
static void edge_points(const uzlo_filter* f, const edata* d, double p[3], double q[3])
{
    const double* Sf = (d->sensor_from >= 0 && d->sensor_from < f->n_sensors) ? f->sensors + 12 * (size_t)d->sensor_from : I12;
    const double* St = (d->sensor_to >= 0 && d->sensor_to < f->n_sensors) ? f->sensors + 12 * (size_t)d->sensor_to : I12;
    double a[12], inv[12];
    iso_mul(d->pos_from, d->disp_from, a);
    iso_mul(a, Sf, a);
    iso_mul(a, d->transform, a);
    iso_inv(St, inv);
    iso_mul(a, inv, a);
    p[0] = a[3]; p[1] = a[7]; p[2] = a[11];
    iso_mul(d->pos_to, d->disp_to, a);
    q[0] = a[3]; q[1] = a[7]; q[2] = a[11];
}

/* TransformationFilter::calcValidEdges (:222-291) */
int32_t uzlo_filter_calc_valid_edges(uzlo_filter* f)
{
    int32_t evaluated = 0;
    for (int32_t ci = 0; ci < f->nc; ci++) {
        cluster* c = f->clusters[ci];
        if ((double)c->n < f->cfg.min_size) continue;                          /* :233 */
        if (!c->changed) continue;                                             /* :236 */
        if (fabs(dsec(c->fs, c->fe)) < f->cfg.min_time_span || fabs(dsec(c->ts, c->te)) < f->cfg.min_time_span) continue;   /* :240-244 */
        c->changed = 0;                                                        /* :247 */
        const int32_t m = c->n;
        c->lastP = (double*)realloc(c->lastP, sizeof(double) * 3 * (size_t)m);
        c->lastQ = (double*)realloc(c->lastQ, sizeof(double) * 3 * (size_t)m);
        for (int32_t k = 0; k < m; k++) edge_points(f, &c->e[k], c->lastP + 3 * (size_t)k, c->lastQ + 3 * (size_t)k);
        double T[12], mse;
        int32_t cons;
        uint8_t* set = (uint8_t*)malloc((size_t)m);
        uzlo_prosac(c->lastP, c->lastQ, m, f->cfg.max_error, f->cfg.ransac_iterations, 1.0, 0, f->cfg.seed,
                    (c->uid << 20) + (uint64_t)c->evals, T, &cons, &mse, NULL, NULL, NULL);                 /* :270-273 */
        c->last_ransac = cons;
        cons = uzlo_consensus3d(c->lastP, c->lastQ, m, T, f->cfg.max_error, set);                             /* :275-276 */
        memcpy(c->lastT, T, sizeof(T)); c->last_n = m;
        c->evals++;
        evaluated++;
        if ((double)cons >= f->cfg.min_size && cons >= c->consensus) {         /* :279-284 */
            c->consensus = cons;
            for (int32_t k = 0; k < m; k++) c->e[k].valid = set[k];
        }
        free(set);
    }
    return evaluated;
}

static int cmp_u64(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* TransformationFilter::validEdges (:293-337): returns the number of keys (sorted, unique) */
int32_t uzlo_filter_valid_edges(const uzlo_filter* f, int32_t cap, uint64_t* keys)
{
    size_t total = 0;
    for (int32_t i = 0; i < f->nc; i++) total += (size_t)f->clusters[i]->n;
    uint64_t* out = (uint64_t*)malloc(sizeof(uint64_t) * (total + 1));
    size_t no = 0;
    const int32_t max_edges = f->cfg.max_edges;
    for (int32_t ci = 0; ci < f->nc; ci++) {
        const cluster* c = f->clusters[ci];
        int32_t nv = 0;
        int32_t* v = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->n + 1));
        for (int32_t k = 0; k < c->n; k++) if (c->e[k].valid) v[nv++] = k;     /* :299-303 */
        if (nv > 2 * max_edges) {                                              /* :311 */
            for (int32_t a = 1; a < nv; a++) {                                 /* stable insertion sort, score descending (:313) */
                int32_t x = v[a], b = a - 1;
                while (b >= 0 && c->e[v[b]].score < c->e[x].score) { v[b + 1] = v[b]; b--; }
                v[b + 1] = x;
            }
            for (int32_t i = 0; i < max_edges; i++) out[no++] = c->e[v[i]].key;            /* :316-318 */
            /* :321 sorts with EdgeScoreSort again: order unchanged */
            const double increment = (double)nv / (double)max_edges;                        /* :324 */
            for (int32_t i = 0; i < max_edges - 1; i++) out[no++] = c->e[v[(int32_t)floor(increment * i)]].key;   /* :325-327 */
            out[no++] = c->e[v[nv - 1]].key;                                                /* :328 */
        } else {
            for (int32_t i = 0; i < nv; i++) out[no++] = c->e[v[i]].key;                    /* :331-333 */
        }
        free(v);
    }
    qsort(out, no, sizeof(uint64_t), cmp_u64);                                 /* std::set<std::string> */
    size_t nu = 0;
    for (size_t i = 0; i < no; i++) if (i == 0 || out[i] != out[i - 1]) out[nu++] = out[i];
    for (size_t i = 0; i < nu && (int32_t)i < cap; i++) keys[i] = out[i];
    free(out);
    return (int32_t)nu;
}

/* ---- introspection (parity tests) ---- */
int32_t uzlo_filter_cluster_count(const uzlo_filter* f) { return f->nc; }

void uzlo_filter_cluster_info(const uzlo_filter* f, int32_t idx, uzlo_cluster_info* o)
{
    const cluster* c = f->clusters[idx];
    o->uid = c->uid; o->from_start_ns = c->fs; o->from_end_ns = c->fe; o->to_start_ns = c->ts; o->to_end_ns = c->te;
    o->size = c->n; o->consensus = c->consensus; o->changed = c->changed; o->evaluations = c->evals;
}

void uzlo_filter_cluster_edges(const uzlo_filter* f, int32_t idx, uint64_t* keys, uint8_t* valid)
{
    const cluster* c = f->clusters[idx];
    for (int32_t k = 0; k < c->n; k++) { keys[k] = c->e[k].key; valid[k] = c->e[k].valid; }
}

int32_t uzlo_filter_cluster_last_eval(const uzlo_filter* f, int32_t idx, double* P, double* Q, double* T, int32_t* ransac_consensus)
{
    const cluster* c = f->clusters[idx];
    if (c->last_n > 0) { memcpy(P, c->lastP, sizeof(double) * 3 * (size_t)c->last_n); memcpy(Q, c->lastQ, sizeof(double) * 3 * (size_t)c->last_n); }
    memcpy(T, c->lastT, sizeof(c->lastT));
    *ransac_consensus = c->last_ransac;
    return c->last_n;
}
