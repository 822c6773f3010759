/*
 * uzl_oracle_match.c — CPU ORACLE (test infrastructure, not product code; see uzl_oracle.h).
 * Restates transformation_estimation/src/feature_transformation_estimator.cpp:32-347.
 * PARITY UNPINNED (no reference golden vectors exist; see uzl_oracle.h header).
 *
 * Build with -ffp-contract=off: the float pose recipe must round after every operation.
 */
#include "uzl_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * M1  cv::BFMatcher(NORM_HAMMING).knnMatch(query=to, train=from, k=2)
 *     feature_transformation_estimator.cpp:38,58.  [EXT] OpenCV 2.4 batchDistance keeps the K
 *     best per query with a strict '<' insertion, so equal distances keep the lower train index
 *     first: the order is lexicographic in (distance, trainIdx).
 * ------------------------------------------------------------------------------------------ */
static inline int32_t hamming_bytes(const uint8_t* a, const uint8_t* b, int32_t bytes)
{
    int32_t d = 0, i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    for (; i < bytes; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

void uzlo_knn2(const uint8_t* query, int32_t nq, const uint8_t* train, int32_t nt, int32_t bytes,
               int32_t* idx0, int32_t* d0, int32_t* idx1, int32_t* d1)
{
    for (int32_t q = 0; q < nq; q++) {
        const uint8_t* qd = query + (size_t)q * bytes;
        int32_t b0 = -1, b1 = -1, e0 = INT32_MAX, e1 = INT32_MAX;
        for (int32_t t = 0; t < nt; t++) {
            int32_t d = hamming_bytes(qd, train + (size_t)t * bytes, bytes);
            if (d < e0) { e1 = e0; b1 = b0; e0 = d; b0 = t; }
            else if (d < e1) { e1 = d; b1 = t; }
        }
        idx0[q] = b0; d0[q] = (b0 >= 0) ? e0 : -1;
        idx1[q] = b1; d1[q] = (b1 >= 0) ? e1 : -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * M2  ratio test  match_pair.size()==2 && d0 < 0.99*d1   (:65-71)   (float distance -> double)
 * M4  valid_3d filter (:101-112) and std::sort by DMatch::operator< (distance) (:114).
 *     std::sort is unstable; the build fixes the order to (distance, queryIdx).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t d, q, t; } match_t;
static int cmp_match(const void* a, const void* b)
{
    const match_t* x = (const match_t*)a; const match_t* y = (const match_t*)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    if (x->q != y->q) return x->q < y->q ? -1 : 1;
    return 0;
}

int32_t uzlo_filter_sort(int32_t nq, const int32_t* idx0, const int32_t* d0, const int32_t* idx1,
                         const int32_t* d1, const uint8_t* valid_train, const uint8_t* valid_query,
                         int32_t* out_query, int32_t* out_train, int32_t* out_dist, int32_t* n_ratio)
{
    match_t* m = (match_t*)malloc(sizeof(match_t) * (size_t)(nq > 0 ? nq : 1));
    int32_t cnt = 0, ratio = 0;
    for (int32_t q = 0; q < nq; q++) {
        if (idx0[q] < 0 || idx1[q] < 0) continue;               /* match_pair.size() != 2 */
        if ((double)(float)d0[q] < 0.99 * (double)(float)d1[q]) {
            ratio++;
            if (valid_train[idx0[q]] && valid_query[q]) {
                m[cnt].d = d0[q]; m[cnt].q = q; m[cnt].t = idx0[q]; cnt++;
            }
        }
    }
    qsort(m, (size_t)cnt, sizeof(match_t), cmp_match);
    for (int32_t i = 0; i < cnt; i++) { out_query[i] = m[i].q; out_train[i] = m[i].t; out_dist[i] = m[i].d; }
    free(m);
    if (n_ratio) *n_ratio = ratio;
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * M6a RNG.  Reference: std::random_shuffle(idx.begin(), idx.begin()+n_i) on a persistent
 *     permutation, then sample = idx[0..2] (:217-225).  Because the prefixes only grow, positions
 *     [0,n_i) always hold exactly {0..n_i-1}; a uniform shuffle therefore makes idx[0..2] a uniform
 *     ordered 3-subset of [0,n_i), independent of earlier iterations (positions >= n_i are still
 *     the identity).  The build draws that directly with a counter-based hash so that every
 *     iteration is independent of the others (parallel on the GPU) and reproducible.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31; return x;
}
static inline uint64_t stream_key(uint64_t seed, uint64_t job)
{
    return mix64(seed ^ mix64(job + 0x9e3779b97f4a7c15ULL));
}
static inline uint32_t draw_below(uint64_t key, uint32_t iter, uint32_t k, uint32_t n)
{
    uint64_t h = mix64(key + 0x9e3779b97f4a7c15ULL * (uint64_t)(iter * 4u + k + 1u));
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

int32_t uzlo_prosac_prefix(int32_t iter, int32_t iterations, int32_t m)
{
    int32_t n = (int32_t)ceil(((iter + 3.) / iterations) * m);   /* :217 */
    return n < m ? n : m;
}

void uzlo_sample3(uint64_t seed, uint64_t job_id, int32_t iter, int32_t iterations, int32_t m,
                  int32_t do_prosac, int32_t out[3])
{
    const uint64_t key = stream_key(seed, job_id);
    const int32_t n = do_prosac ? uzlo_prosac_prefix(iter, iterations, m) : m;
    /* forward Fisher-Yates over the virtual identity array, first three steps only */
    int32_t opos[6], oval[6], no = 0;
    for (int32_t s = 0; s < 3; s++) {
        if (n - s >= 2) {
            int32_t j = s + (int32_t)draw_below(key, (uint32_t)iter, (uint32_t)s, (uint32_t)(n - s));
            int32_t vs = s, vj = j;
            for (int32_t k = 0; k < no; k++) { if (opos[k] == s) vs = oval[k]; }
            for (int32_t k = 0; k < no; k++) { if (opos[k] == j) vj = oval[k]; }
            opos[no] = s; oval[no] = vj; no++;
            opos[no] = j; oval[no] = vs; no++;
        }
        int32_t v = s;
        for (int32_t k = 0; k < no; k++) { if (opos[k] == s) v = oval[k]; }
        out[s] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * M7  estimatePoseSVD (:299-314) -> pcl::TransformationFromCorrespondences [EXT], float.
 *     add():   alpha = w/accW; d1 = p-mean1; d2 = q-mean2;
 *              cov = (1-alpha)*(cov + alpha*(d2*d1^T)); mean1 += alpha*d1; mean2 += alpha*d2
 *     (weight is always 1: `weight = 1./weight` at :308 divides 1 by 1)
 *     getTransformation(): JacobiSVD(cov) -> R = U*diag(1,1,sign(det U * det V))*V^T,
 *                          t = mean2 - R*mean1
 *     The SVD is a two-sided Jacobi iteration in the manner of Eigen's JacobiSVD [EXT]; its exact
 *     operation order below is this build's recipe (the HIP kernel repeats it operation for
 *     operation).  Only + - * / sqrt fabs and comparisons are used.
 * ------------------------------------------------------------------------------------------ */
static inline float det3f(const float* m)
{
    float t0 = m[0] * (m[4] * m[8] - m[5] * m[7]);
    float t1 = m[1] * (m[3] * m[8] - m[5] * m[6]);
    float t2 = m[2] * (m[3] * m[7] - m[4] * m[6]);
    return (t0 - t1) + t2;
}

void uzlo_svd3f(const float A[9], float U[9], float S[3], float V[9])
{
    float W[9];
    float scale = 0.f;
    for (int i = 0; i < 9; i++) { float a = fabsf(A[i]); if (a > scale) scale = a; }
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < 9; i++) W[i] = A[i] / scale;
    for (int i = 0; i < 9; i++) { U[i] = (i % 4 == 0) ? 1.f : 0.f; V[i] = U[i]; }

    const float precision = 2.f * FLT_EPSILON;
    const float tiny = FLT_MIN;
    float maxdiag = fabsf(W[0]);
    if (fabsf(W[4]) > maxdiag) maxdiag = fabsf(W[4]);
    if (fabsf(W[8]) > maxdiag) maxdiag = fabsf(W[8]);

    for (int sweep = 0; sweep < 64; sweep++) {
        int rotated = 0;
        for (int p = 1; p < 3; p++) {
            for (int q = 0; q < p; q++) {
                float thr = precision * maxdiag;
                if (thr < tiny) thr = tiny;
                if (!(fabsf(W[p * 3 + q]) > thr || fabsf(W[q * 3 + p]) > thr)) continue;
                rotated = 1;
                /* 2x2 block m = [[a,b],[c,d]] on rows/cols (p,q) */
                float a = W[p * 3 + p], b = W[p * 3 + q], c = W[q * 3 + p], d = W[q * 3 + q];
                /* step 1: left rotation that symmetrises the block */
                float c1, s1;
                float t = a + d, dd = c - b;
                if (fabsf(dd) < tiny) { c1 = 1.f; s1 = 0.f; }
                else { float u = t / dd; float tmp = sqrtf(1.f + u * u); s1 = 1.f / tmp; c1 = u / tmp; }
                float x = c1 * a + s1 * c;
                float y = c1 * b + s1 * d;
                float z = -s1 * b + c1 * d;
                /* step 2: Jacobi rotation that diagonalises [[x,y],[y,z]] */
                float cj, sj;
                if (fabsf(y) < tiny) { cj = 1.f; sj = 0.f; }
                else {
                    float tau = (z - x) / (2.f * y);
                    float w = sqrtf(tau * tau + 1.f);
                    float tt = (tau >= 0.f) ? 1.f / (tau + w) : -1.f / (w - tau);
                    cj = 1.f / sqrtf(tt * tt + 1.f);
                    sj = tt * cj;
                }
                /* left rotation L = J^T * R1 = [[cl, sl],[-sl, cl]] */
                float cl = cj * c1 + sj * s1;
                float sl = cj * s1 - sj * c1;
                /* W <- L_pq * W  (rows p,q) */
                for (int k = 0; k < 3; k++) {
                    float wp = W[p * 3 + k], wq = W[q * 3 + k];
                    W[p * 3 + k] = cl * wp + sl * wq;
                    W[q * 3 + k] = cl * wq - sl * wp;
                }
                /* U <- U * L^T  (cols p,q) */
                for (int k = 0; k < 3; k++) {
                    float up = U[k * 3 + p], uq = U[k * 3 + q];
                    U[k * 3 + p] = cl * up + sl * uq;
                    U[k * 3 + q] = cl * uq - sl * up;
                }
                /* W <- W * J_pq  (cols p,q), J = [[cj, sj],[-sj, cj]] */
                for (int k = 0; k < 3; k++) {
                    float wp = W[k * 3 + p], wq = W[k * 3 + q];
                    W[k * 3 + p] = cj * wp - sj * wq;
                    W[k * 3 + q] = sj * wp + cj * wq;
                }
                /* V <- V * J_pq */
                for (int k = 0; k < 3; k++) {
                    float vp = V[k * 3 + p], vq = V[k * 3 + q];
                    V[k * 3 + p] = cj * vp - sj * vq;
                    V[k * 3 + q] = sj * vp + cj * vq;
                }
                float mp = fabsf(W[p * 3 + p]), mq = fabsf(W[q * 3 + q]);
                if (mp > maxdiag) maxdiag = mp;
                if (mq > maxdiag) maxdiag = mq;
            }
        }
        if (!rotated) break;
    }
    /* singular values = |diag|; negative ones flip the column of U */
    for (int i = 0; i < 3; i++) {
        float s = W[i * 3 + i];
        if (s < 0.f) { s = -s; for (int k = 0; k < 3; k++) U[k * 3 + i] = -U[k * 3 + i]; }
        S[i] = s * scale;
    }
    /* sort descending (selection, first maximum wins) */
    for (int i = 0; i < 2; i++) {
        int pos = i;
        for (int k = i + 1; k < 3; k++) if (S[k] > S[pos]) pos = k;
        if (pos != i) {
            float ts = S[i]; S[i] = S[pos]; S[pos] = ts;
            for (int k = 0; k < 3; k++) {
                float tu = U[k * 3 + i]; U[k * 3 + i] = U[k * 3 + pos]; U[k * 3 + pos] = tu;
                float tv = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + pos]; V[k * 3 + pos] = tv;
            }
        }
    }
}

void uzlo_pose_svd(const double* P, const double* Q, const int32_t* idx, int32_t k, double T[12])
{
    float mean1[3] = {0.f, 0.f, 0.f}, mean2[3] = {0.f, 0.f, 0.f}, cov[9] = {0.f};
    float accw = 0.f;
    for (int32_t i = 0; i < k; i++) {
        const int32_t c = idx ? idx[i] : i;
        float p[3], q[3], d1[3], d2[3];
        for (int r = 0; r < 3; r++) { p[r] = (float)P[3 * (size_t)c + r]; q[r] = (float)Q[3 * (size_t)c + r]; }
        accw += 1.f;
        const float alpha = 1.f / accw;
        const float om = 1.f - alpha;
        for (int r = 0; r < 3; r++) { d1[r] = p[r] - mean1[r]; d2[r] = q[r] - mean2[r]; }
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++)
                cov[r * 3 + cc] = om * (cov[r * 3 + cc] + alpha * (d2[r] * d1[cc]));
        for (int r = 0; r < 3; r++) { mean1[r] += alpha * d1[r]; mean2[r] += alpha * d2[r]; }
    }
    float U[9], S[3], V[9];
    uzlo_svd3f(cov, U, S, V);
    const float sg = (det3f(U) * det3f(V) < 0.f) ? -1.f : 1.f;
    float R[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            R[r * 3 + c] = (U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1]) + (U[r * 3 + 2] * sg) * V[c * 3 + 2];
    for (int r = 0; r < 3; r++) {
        float rm = (R[r * 3 + 0] * mean1[0] + R[r * 3 + 1] * mean1[1]) + R[r * 3 + 2] * mean1[2];
        float t = mean2[r] - rm;
        T[r * 4 + 0] = (double)R[r * 3 + 0];
        T[r * 4 + 1] = (double)R[r * 3 + 1];
        T[r * 4 + 2] = (double)R[r * 3 + 2];
        T[r * 4 + 3] = (double)t;
    }
}

/* ------------------------------------------------------------------------------------------
 * M8  consensus3D (:337-347): P' = T*[P;1]; set[i] = ||P'_i - Q_i|| < thresh (double, strict).
 * ------------------------------------------------------------------------------------------ */
/* Vote recipes.  0 (default, the one the HIP kernels repeat operation for operation): fused multiply-adds, innermost first.
 * 1 = the reference binary's evaluation: its package is built with -msse2 -msse3 -mssse3 -O3 (transformation_estimation/
 * CMakeLists.txt:9, no FMA), Eigen 3.2 evaluates `T * P.colwise().homogeneous()` as linear() * P (3 x 3 times 3 x M through the
 * general product: acc += R(r,k) * p(k) for k = 0, 1, 2, every product and sum rounded) and then adds the translation;
 * `(P - Q).colwise().norm()` is sqrt((dx*dx + dy*dy) + dz*dz).  The two differ by an ulp or two of the distance, i.e. only a
 * point within ~1e-16 m of the threshold can vote differently; tests/test_oracle_match.py counts how often that happens. */
static int g_vote_recipe = 0;
void uzlo_set_vote_recipe(int32_t r) { g_vote_recipe = r; }

static inline double point_dist_fused(const double* p, const double* q, const double T[12])
{
    double x = fma(T[0], p[0], fma(T[1], p[1], fma(T[2], p[2], T[3])));
    double y = fma(T[4], p[0], fma(T[5], p[1], fma(T[6], p[2], T[7])));
    double z = fma(T[8], p[0], fma(T[9], p[1], fma(T[10], p[2], T[11])));
    double dx = x - q[0], dy = y - q[1], dz = z - q[2];
    return sqrt(fma(dx, dx, fma(dy, dy, dz * dz)));
}
static inline double point_dist_reference_order(const double* p, const double* q, const double T[12])
{
    /* -ffp-contract=off: none of this is fused */
    double x = ((T[0] * p[0] + T[1] * p[1]) + T[2] * p[2]) + T[3];
    double y = ((T[4] * p[0] + T[5] * p[1]) + T[6] * p[2]) + T[7];
    double z = ((T[8] * p[0] + T[9] * p[1]) + T[10] * p[2]) + T[11];
    double dx = x - q[0], dy = y - q[1], dz = z - q[2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
}
static inline double point_dist(const double* p, const double* q, const double T[12])
{
    return g_vote_recipe ? point_dist_reference_order(p, q, T) : point_dist_fused(p, q, T);
}

/* Every hypothesis of a PROSAC run (no early exit) voted under both recipes: number of (hypothesis, point) tests and of tests
 * whose verdict differs; min_margin = the smallest | distance - threshold | met (fused recipe). */
void uzlo_vote_recipe_diff(const double* P, const double* Q, int32_t m, double max_error, int32_t iterations, int32_t do_prosac,
                           uint64_t seed, uint64_t job_id, int64_t* n_tests, int64_t* n_diff, double* min_margin)
{
    int64_t nt = 0, nd = 0;
    double mm = 1e300;
    if (m >= 3) {
        for (int32_t i = 0; i < iterations; i++) {
            int32_t s[3];
            double Tt[12];
            uzlo_sample3(seed, job_id, i, iterations, m, do_prosac, s);
            uzlo_pose_svd(P, Q, s, 3, Tt);
            for (int32_t k = 0; k < m; k++) {
                const double a = point_dist_fused(P + 3 * (size_t)k, Q + 3 * (size_t)k, Tt);
                const double b = point_dist_reference_order(P + 3 * (size_t)k, Q + 3 * (size_t)k, Tt);
                nd += (a < max_error) != (b < max_error);
                const double g = fabs(a - max_error);
                if (g < mm) mm = g;
                nt++;
            }
        }
    }
    *n_tests = nt; *n_diff = nd; *min_margin = mm;
}

int32_t uzlo_consensus3d(const double* P, const double* Q, int32_t m, const double T[12],
                         double thresh, uint8_t* set)
{
    int32_t cnt = 0;
    for (int32_t i = 0; i < m; i++) {
        uint8_t in = point_dist(P + 3 * (size_t)i, Q + 3 * (size_t)i, T) < thresh;
        if (set) set[i] = in;
        cnt += in;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * M6  prosac (:186-297) with minCorrespondenceCount = 3 (estimateSVD, :178-184).
 * ------------------------------------------------------------------------------------------ */
void uzlo_prosac(const double* P, const double* Q, int32_t m, double max_error, int32_t iterations,
                 double break_percentage, int32_t do_prosac, uint64_t seed, uint64_t job_id,
                 double T[12], int32_t* consensus, double* mse, uint8_t* mask,
                 int32_t* iterations_run, int32_t* best_iteration)
{
    static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    uint8_t* set = (uint8_t*)malloc((size_t)(m > 0 ? m : 1));
    uint8_t* best = (uint8_t*)calloc((size_t)(m > 0 ? m : 1), 1);
    int32_t max_cons = 0, it_run = 0, best_it = -1;
    double Tt[12];
    memcpy(T, I12, sizeof(I12));
    if (m >= 3) {
        for (int32_t i = 0; i < iterations; i++) {                           /* :214 */
            int32_t s[3];
            uzlo_sample3(seed, job_id, i, iterations, m, do_prosac, s);      /* :216-225 */
            uzlo_pose_svd(P, Q, s, 3, Tt);                                   /* :227 */
            int32_t c = uzlo_consensus3d(P, Q, m, Tt, max_error, set);       /* :230 */
            it_run = i + 1;
            if (c > max_cons) {                                              /* :233 */
                max_cons = c; best_it = i;
                memcpy(best, set, (size_t)m);
                memcpy(T, Tt, sizeof(Tt));
                if (max_cons >= 3 && (double)max_cons > break_percentage * (double)m) break;   /* :239 */
            }
        }
    }
    double err = 0.;
    if (max_cons >= 3) {                                                     /* :246 */
        int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_cons);
        int32_t k = 0;
        for (int32_t i = 0; i < m; i++) if (best[i]) sel[k++] = i;
        uzlo_pose_svd(P, Q, sel, max_cons, T);                               /* :257 */
        max_cons = uzlo_consensus3d(P, Q, m, T, max_error, best);            /* :258 */
        for (int32_t i = 0; i < m; i++)                                      /* :285-289 */
            if (best[i]) err += point_dist(P + 3 * (size_t)i, Q + 3 * (size_t)i, T);
        err /= max_cons;                                                     /* :290 */
        free(sel);
    } else {                                                                 /* :291-294 */
        max_cons = 0;
        memcpy(T, I12, sizeof(I12));
        memset(best, 0, (size_t)(m > 0 ? m : 1));
    }
    *consensus = max_cons;
    *mse = err;
    if (mask) memcpy(mask, best, (size_t)(m > 0 ? m : 0));
    if (iterations_run) *iterations_run = it_run;
    if (best_iteration) *best_iteration = best_it;
    free(set); free(best);
}

/* M9 information matrix (:133-137) */
void uzlo_information(int32_t consensus, double mse, double info[36])
{
    for (int i = 0; i < 36; i++) info[i] = (i % 7 == 0) ? 1. : 0.;
    if (consensus > 0 && mse > 0) {
        const double s = 0.1 * consensus / mse;
        for (int i = 0; i < 36; i++) info[i] *= s;
        for (int r = 3; r < 6; r++) for (int c = 3; c < 6; c++) info[r * 6 + c] *= 100.;
    }
}

/* ------------------------------------------------------------------------------------------
 * estimateEdgeDirect (:32-159)
 * ------------------------------------------------------------------------------------------ */
void uzlo_estimate_edge(const uzlo_frame* from, int32_t n_from, const uzlo_frame* to, int32_t n_to,
                        double ransac_threshold, int32_t ransac_iteration, double break_percentage,
                        int32_t do_prosac, uint64_t seed, uint64_t job_id,
                        uzlo_edge_result* res, int32_t max_corr,
                        int32_t* corr_query, int32_t* corr_train, int32_t* corr_dist, uint8_t* mask)
{
    static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memset(res, 0, sizeof(*res));
    res->frame_from = -1; res->frame_to = -1; res->best_iteration = -1;
    memcpy(res->T, I12, sizeof(I12));
    uzlo_information(0, 0., res->information);

    double best_score = -1;                                                  /* :35 */
    int32_t bf = -1, bt = -1, best_m = 0;
    int32_t *bq = NULL, *btr = NULL, *bd = NULL;
    for (int32_t f = 0; f < n_from; f++) {                                   /* :40 */
        for (int32_t t = 0; t < n_to; t++) {                                 /* :42 */
            const uzlo_frame* ff = &from[f]; const uzlo_frame* ft = &to[t];
            if (!(ff->n >= 7 && ft->n >= 7 && ff->feature_type == ft->feature_type &&
                  ff->sensor_frame == ft->sensor_frame && ff->bytes_per_desc == ft->bytes_per_desc)) continue;   /* :47-49 */
            int32_t nq = ft->n;
            int32_t* i0 = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)nq);
            int32_t *d0 = i0 + nq, *i1 = d0 + nq, *d1 = i1 + nq;
            uzlo_knn2(ft->desc, nq, ff->desc, ff->n, ff->bytes_per_desc, i0, d0, i1, d1);   /* :58 */
            int32_t* oq = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nq);
            int32_t *ot = oq + nq, *od = ot + nq;
            int32_t n_ratio = 0;
            int32_t m = uzlo_filter_sort(nq, i0, d0, i1, d1, ff->valid3d, ft->valid3d, oq, ot, od, &n_ratio);
            double score = (double)n_ratio;                                  /* :78 */
            if (score > best_score) {                                        /* :81 */
                best_score = score; bf = f; bt = t; best_m = m;
                free(bq); bq = oq; btr = ot; bd = od;
            } else free(oq);
            free(i0);
        }
    }
    if (best_score == -1) { free(bq); return; }                              /* :93 */
    res->n_matches = (int32_t)best_score;
    res->n_corr = best_m;
    res->frame_from = bf; res->frame_to = bt;
    for (int32_t i = 0; i < best_m && i < max_corr; i++) {
        if (corr_query) corr_query[i] = bq[i];
        if (corr_train) corr_train[i] = btr[i];
        if (corr_dist) corr_dist[i] = bd[i];
    }
    if (best_m >= 3) {                                                       /* :118 */
        double* Xd = (double*)malloc(sizeof(double) * 6 * (size_t)best_m);
        double* Pd = Xd + 3 * (size_t)best_m;
        for (int32_t i = 0; i < best_m; i++)                                  /* :121-124 */
            for (int r = 0; r < 3; r++) {
                Xd[3 * i + r] = from[bf].pos_xyz[3 * (size_t)btr[i] + r];
                Pd[3 * i + r] = to[bt].pos_xyz[3 * (size_t)bq[i] + r];
            }
        uint8_t* mk = (uint8_t*)malloc((size_t)best_m);
        uzlo_prosac(Pd, Xd, best_m, ransac_threshold, ransac_iteration, break_percentage, do_prosac,
                    seed, job_id, res->T, &res->consensus, &res->mse, mk,
                    &res->iterations_run, &res->best_iteration);              /* :130 */
        uzlo_information(res->consensus, res->mse, res->information);         /* :133-137 */
        if (mask) for (int32_t i = 0; i < best_m && i < max_corr; i++) mask[i] = mk[i];
        res->ok = 1;                                                          /* :156 */
        free(mk); free(Xd);
    }
    free(bq);
}


void uzlo_estimate_edge_batch(int32_t n_pairs, const uzlo_frame* from, const uzlo_frame* to,
                              double ransac_threshold, int32_t ransac_iteration, double break_percentage,
                              int32_t do_prosac, uint64_t seed, uint64_t job_id0, int32_t threads, uzlo_edge_result* results)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (int32_t k = 0; k < n_pairs; k++)
        uzlo_estimate_edge(from + k, 1, to + k, 1, ransac_threshold, ransac_iteration, break_percentage, do_prosac, seed,
                           job_id0 + (uint64_t)k, results + k, 0, NULL, NULL, NULL, NULL);
}
