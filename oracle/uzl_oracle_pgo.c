/*
 * uzl_oracle_pgo.c — CPU ORACLE (test infrastructure, not product code; see uzl_oracle.h).
 * Restates graph_optimization/src/g2o_optimizer.cpp:55-349 and the g2o semantics it delegates to
 * (EdgeSE3, RobustKernelHuber, BlockSolver<6,3>, OptimizationAlgorithmLevenberg,
 * LinearSolverCSparse = sparse direct Cholesky) [EXT], with the SE(3) <-> vector maps taken from the
 * in-tree g2o excerpt graph_slam_common/thirdparty/src/isometry3d_mappings.cpp.
 * PARITY UNPINNED (no reference golden vectors exist; see uzl_oracle.h header).
 */
#include "uzl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---------------- SE(3) helpers on 12-double row-major [R|t] ---------------- */
#define R_(T, r, c) ((T)[(r) * 4 + (c)])
#define t_(T, r) ((T)[(r) * 4 + 3])

static void se3_mul(const double A[12], const double B[12], double C[12])
{
    double o[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            o[r * 4 + c] = R_(A, r, 0) * R_(B, 0, c) + R_(A, r, 1) * R_(B, 1, c) + R_(A, r, 2) * R_(B, 2, c);
        o[r * 4 + 3] = R_(A, r, 0) * t_(B, 0) + R_(A, r, 1) * t_(B, 1) + R_(A, r, 2) * t_(B, 2) + t_(A, r);
    }
    memcpy(C, o, sizeof(o));
}
static void se3_inv(const double A[12], double C[12])
{
    double o[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) o[r * 4 + c] = R_(A, c, r);
        o[r * 4 + 3] = -(R_(A, 0, r) * t_(A, 0) + R_(A, 1, r) * t_(A, 1) + R_(A, 2, r) * t_(A, 2));
    }
    memcpy(C, o, sizeof(o));
}
static void se3_rot(const double T[12], double R[9])
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r * 3 + c] = R_(T, r, c);
}

/* Eigen::Quaterniond(Matrix3d) [EXT Eigen 3.2 quaternionbase_assign_impl<Other,3,3>] -> (w,x,y,z) */
void uzlo_quat_from_R(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[7] - m[5]) * t;
        q[2] = (m[2] - m[6]) * t;
        q[3] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[1 + j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[1 + k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

/* Eigen::Quaterniond::toRotationMatrix [EXT] */
void uzlo_R_from_quat(const double q[4], double R[9])
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* normalize(q): ||q||=1, w>=0  (isometry3d_mappings.cpp:38-44) */
static void quat_normalize_pos(double q[4])
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
    if (q[0] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
}

/* toVectorMQT (isometry3d_mappings.cpp:94-99) via toCompactQuaternion (:77-82) */
void uzlo_to_vector_mqt(const double T[12], double v[6])
{
    double R[9], q[4];
    se3_rot(T, R);
    uzlo_quat_from_R(R, q);
    quat_normalize_pos(q);
    v[0] = t_(T, 0); v[1] = t_(T, 1); v[2] = t_(T, 2);
    v[3] = q[1]; v[4] = q[2]; v[5] = q[3];
}

/* fromVectorMQT (:117-122) via fromCompactQuaternion (:84-91) */
void uzlo_from_vector_mqt(const double v[6], double T[12])
{
    double R[9];
    double w = 1 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
    } else {
        double q[4] = {sqrt(w), v[3], v[4], v[5]};
        uzlo_R_from_quat(q, R);
    }
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R_(T, r, c) = R[r * 3 + c]; t_(T, r) = v[r]; }
}

/* toEuler (:47-57), fromEuler (:59-75) */
void uzlo_to_euler(const double R[9], double rpy[3])
{
    double q[4];
    uzlo_quat_from_R(R, q);
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    rpy[0] = atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
    rpy[1] = asin(2 * (q0 * q2 - q3 * q1));
    rpy[2] = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
}
void uzlo_from_euler(const double v[3], double R[9])
{
    double sy = sin(v[2] * 0.5), cy = cos(v[2] * 0.5);
    double sp = sin(v[1] * 0.5), cp = cos(v[1] * 0.5);
    double sr = sin(v[0] * 0.5), cr = cos(v[0] * 0.5);
    double q[4];
    q[0] = cr * cp * cy + sr * sp * sy;
    q[1] = sr * cp * cy - cr * sp * sy;
    q[2] = cr * sp * cy + sr * cp * sy;
    q[3] = cr * cp * sy - sr * sp * cy;
    uzlo_R_from_quat(q, R);
}

/* optimize_xy_only projection (g2o_optimizer.cpp:164-170, :231-237, :282-288) */
static void project_xy(double T[12])
{
    double R[9], rpy[3];
    se3_rot(T, R);
    uzlo_to_euler(R, rpy);
    rpy[0] = 0; rpy[1] = 0;
    uzlo_from_euler(rpy, R);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R_(T, r, c) = R[r * 3 + c];
    t_(T, 2) = 0;
}

/* ---------------- G3: EdgeSE3::computeError [EXT]  e = toVectorMQT(Z^-1 * Xi^-1 * Xj) ---------------- */
void uzlo_edge_error(const double Xi[12], const double Xj[12], const double Z[12], double e[6])
{
    double Zi[12], Xii[12], B[12], E[12];
    se3_inv(Z, Zi);
    se3_inv(Xi, Xii);
    se3_mul(Xii, Xj, B);
    se3_mul(Zi, B, E);
    uzlo_to_vector_mqt(E, e);
}

/* ---------------- G4: EdgeSE3::linearizeOplus [EXT] --------------------------------------------------
 * Analytic derivative of e w.r.t. the right-multiplicative increments X <- X * fromVectorMQT(d) at d=0.
 * With A = Z^-1, B = Xi^-1 Xj, E = A B:
 *   d t_E / d t_i = -Ra            d t_E / d q_i = 2 Ra [t_b]x
 *   d t_E / d t_j =  R_E           d t_E / d q_j = 0
 *   d q_E / d q_j = w_E I + [v_E]x
 *   d q_E / d q_i = -s * ( -v_b v_a^T + (w_b I - [v_b]x)(w_a I + [v_a]x) ),   q_E = s * (q_a (x) q_b), s = +-1
 * (first-order quaternion increment (1, dq); the factor 2 comes from angle = 2|dq|).  g2o's own
 * implementation reaches the same derivative through a generated dq/dR table; the tests check both
 * against central differences. */
static void skew(const double v[3], double S[9])
{
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    double o[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        o[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
    memcpy(C, o, sizeof(o));
}

void uzlo_edge_jacobians(const double Xi[12], const double Xj[12], const double Z[12],
                         double Ji[36], double Jj[36])
{
    double A[12], Xii[12], B[12], E[12];
    se3_inv(Z, A);
    se3_inv(Xi, Xii);
    se3_mul(Xii, Xj, B);
    se3_mul(A, B, E);
    double Ra[9], Rb[9], Re[9], qa[4], qb[4], qe[4];
    se3_rot(A, Ra); se3_rot(B, Rb); se3_rot(E, Re);
    uzlo_quat_from_R(Ra, qa); quat_normalize_pos(qa);
    uzlo_quat_from_R(Rb, qb); quat_normalize_pos(qb);
    uzlo_quat_from_R(Re, qe); quat_normalize_pos(qe);
    /* sign s with q_E = s * (q_a (x) q_b) */
    double wab = qa[0] * qb[0] - (qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3]);
    double s = (wab * qe[0] >= 0) ? 1. : -1.;
    if (fabs(qe[0]) < 1e-12) {   /* w_E ~ 0: decide by the vector part */
        double vab0 = qa[0] * qb[1] + qb[0] * qa[1] + (qa[2] * qb[3] - qa[3] * qb[2]);
        double vab1 = qa[0] * qb[2] + qb[0] * qa[2] + (qa[3] * qb[1] - qa[1] * qb[3]);
        double vab2 = qa[0] * qb[3] + qb[0] * qa[3] + (qa[1] * qb[2] - qa[2] * qb[1]);
        s = (vab0 * qe[1] + vab1 * qe[2] + vab2 * qe[3] >= 0) ? 1. : -1.;
    }
    memset(Ji, 0, 36 * sizeof(double));
    memset(Jj, 0, 36 * sizeof(double));
    double tb[3] = {t_(B, 0), t_(B, 1), t_(B, 2)};
    double S[9], M[9];
    skew(tb, S);
    mat3_mul(Ra, S, M);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        Ji[r * 6 + c] = -Ra[r * 3 + c];
        Ji[r * 6 + 3 + c] = 2 * M[r * 3 + c];
        Jj[r * 6 + c] = Re[r * 3 + c];
    }
    /* d q_E / d q_j = w_E I + [v_E]x */
    double Se[9];
    skew(&qe[1], Se);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        Jj[(3 + r) * 6 + 3 + c] = ((r == c) ? qe[0] : 0.) + Se[r * 3 + c];
    /* d q_E / d q_i */
    double Sa[9], Sb[9], L[9], Rm[9], P[9];
    skew(&qa[1], Sa); skew(&qb[1], Sb);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        L[r * 3 + c] = ((r == c) ? qb[0] : 0.) - Sb[r * 3 + c];
        Rm[r * 3 + c] = ((r == c) ? qa[0] : 0.) + Sa[r * 3 + c];
    }
    mat3_mul(L, Rm, P);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        Ji[(3 + r) * 6 + 3 + c] = -s * (P[r * 3 + c] - qb[1 + r] * qa[1 + c]);
}

/* ---------------- G5: RobustKernelHuber::robustify [EXT] ---------------- */
void uzlo_huber(double e2, double delta, double rho[3])
{
    double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
    else {
        double sqrte = sqrt(e2);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e2;
    }
}

/* ---------------- G1: addGraphImpl flattening ---------------- */
static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

/* g2o::OdomConvert::convertToVelocity followed by convertToMotion(vel, l = 1) [EXT]
 * (g2o/types/sclam2d/odometry_measurement.cpp; restated from the published source):
 *   to velocity: |theta| > 1e-7:  px2 = (0,10); px3 = (x,y); px4 = Rot(theta) px2 + px3;
 *                                 R = y2 (x3 y4 - y3 x4) / (y2 (x3 - x4));  w = |dt| > 1e-7 ? theta/dt : 0;
 *                                 vl = (2 R w - w)/2;  vr = w + vl
 *                else             vl = vr = |dt| > 1e-7 ? hypot(x,y)/dt : 0
 *   to motion:   |vr - vl| > 1e-7: R = l/2 (vl+vr)/(vr-vl); w = (vr-vl)/l; theta = w dt;
 *                                 (x,y) = Rot(theta) (0,-R) + (0,R)
 *                else             tv = (vr+vl)/2; theta = 0; x = tv dt; y = 0 */
void uzlo_odom_convert(double x, double y, double theta, double dt, double out[3])
{
    double vl, vr;
    if (fabs(theta) > 1e-7) {
        const double c = cos(theta), s = sin(theta);
        const double y2 = 10.;
        const double x3 = x, y3 = y;
        const double x4 = (c * 0. - s * y2) + x3, y4 = (s * 0. + c * y2) + y3;
        const double R = (y2 * (x3 * y4 - y3 * x4)) / (y2 * (x3 - x4));
        const double w = (fabs(dt) > 1e-7) ? theta / dt : 0.;
        vl = (2. * R * w - w) / 2.;
        vr = w + vl;
    } else {
        vl = vr = (fabs(dt) > 1e-7) ? hypot(x, y) / dt : 0.;
    }
    const double l = 1.;
    if (fabs(vr - vl) > 1e-7) {
        const double R = l * 0.5 * ((vl + vr) / (vr - vl));
        const double w = (vr - vl) / l;
        const double th = w * dt;
        const double c = cos(th), s = sin(th);
        out[0] = (c * 0. - s * (-R)) + 0.;
        out[1] = (s * 0. + c * (-R)) + R;
        out[2] = th;
    } else {
        const double tv = 0.5 * (vr + vl);
        out[0] = tv * dt; out[1] = 0.; out[2] = 0.;
    }
}

int32_t uzlo_flatten_graph(int32_t n_nodes, const uzlo_node* nodes, int32_t n_edges, const uzlo_edge* edges,
                           int32_t n_sensors, const double* sensors, int32_t optimize_xy_only,
                           int32_t use_odometry_parameters,
                           double* poses, uint8_t* fixed, int32_t* ij, double* meas, double* info,
                           uint8_t* robust, int32_t* src_edge)
{
    for (int32_t v = 0; v < n_nodes; v++) {                                   /* addVertex :160-188 */
        memcpy(poses + 12 * (size_t)v, nodes[v].pose, 12 * sizeof(double));
        if (optimize_xy_only) project_xy(poses + 12 * (size_t)v);             /* :164-170 */
        fixed[v] = nodes[v].fixed ? 1 : 0;
    }
    int32_t ne = 0;
    /* the reference adds odometry edges while iterating (:78-79) and the filtered feature edges after
       (:100-103); g2o's result does not depend on that order, but keep it: odometry first. */
    for (int pass = 0; pass < 2; pass++) {
        for (int32_t k = 0; k < n_edges; k++) {
            const uzlo_edge* ed = &edges[k];
            if (ed->from < 0 || ed->to < 0 || ed->from >= n_nodes || ed->to >= n_nodes) continue;   /* :77 */
            const int is_odom = (ed->type == 104);                              /* TYPE_2D_WHEEL_ODOMETRY */
            if ((pass == 0) != is_odom) continue;
            double Zm[12], tmp[12], inv[12];
            if (is_odom) {                                                    /* addOdometryEdge :190-259 */
                if (nodes[ed->from].fixed && !nodes[ed->to].fixed) continue;  /* :203-206 */
                double odom[12];
                memcpy(odom, ed->transform, sizeof(odom));
                if (use_odometry_parameters) {                                /* :209-227 */
                    double R[9], rpy[3], m[3];
                    se3_rot(odom, R);
                    uzlo_to_euler(R, rpy);
                    uzlo_odom_convert(t_(odom, 0), t_(odom, 1), rpy[2], fabs(ed->diff_time), m);
                    t_(odom, 0) = m[0]; t_(odom, 1) = m[1];
                    rpy[2] = m[2];
                    uzlo_from_euler(rpy, R);
                    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R_(odom, r, c) = R[r * 3 + c];
                }
                se3_mul(ed->displacement_from, odom, tmp);                    /* :229 */
                se3_inv(ed->displacement_to, inv);
                se3_mul(tmp, inv, Zm);
                robust[ne] = 0;
            } else {                                                          /* addFeatureEdge :261-299 */
                if (!ed->valid) continue;                                     /* not in validEdges() :98 */
                if (nodes[ed->from].fixed && nodes[ed->to].fixed) continue;   /* :270-274 */
                const double* Sf = (ed->sensor_from >= 0 && ed->sensor_from < n_sensors) ? sensors + 12 * (size_t)ed->sensor_from : I12;
                const double* St = (ed->sensor_to >= 0 && ed->sensor_to < n_sensors) ? sensors + 12 * (size_t)ed->sensor_to : I12;
                se3_mul(ed->displacement_from, Sf, tmp);                      /* :281 */
                se3_mul(tmp, ed->transform, tmp);
                se3_inv(St, inv);
                se3_mul(tmp, inv, tmp);
                se3_inv(ed->displacement_to, inv);
                se3_mul(tmp, inv, Zm);
                robust[ne] = 1;                                               /* :292-294 */
            }
            if (optimize_xy_only) project_xy(Zm);                             /* :231-237, :282-288 */
            memcpy(meas + 12 * (size_t)ne, Zm, sizeof(Zm));
            memcpy(info + 36 * (size_t)ne, ed->information, 36 * sizeof(double));
            ij[2 * ne] = ed->from; ij[2 * ne + 1] = ed->to;
            if (src_edge) src_edge[ne] = k;
            ne++;
        }
    }
    return ne;
}

/* ---------------- G2: setFixedNodes (:301-349) ---------------- */
static int32_t uf_find(int32_t* p, int32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

int32_t uzlo_set_fixed_nodes(int32_t n, uint8_t* fixed, int32_t e, const int32_t* ij)
{
    int32_t* p = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; i++) p[i] = i;
    for (int32_t k = 0; k < e; k++) {
        int32_t a = uf_find(p, ij[2 * k]), b = uf_find(p, ij[2 * k + 1]);
        if (a != b) { if (a < b) p[b] = a; else p[a] = b; }      /* root = smallest index of the component */
    }
    uint8_t* has = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int32_t i = 0; i < n; i++) if (fixed[i]) has[uf_find(p, i)] = 1;
    int32_t cnt = 0;
    for (int32_t i = 0; i < n; i++) {
        int32_t r = uf_find(p, i);
        if (!has[r]) { fixed[r] = 1; has[r] = 1; cnt++; }          /* r is the smallest index (:338) */
    }
    free(p); free(has);
    return cnt;
}

/* ---------------- per-edge linearisation shared by LM and the dense builder ---------------- */
static void mat6_mul_AtB(const double A[36], const double B[36], double C[36])   /* C = A^T B */
{
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += A[k * 6 + r] * B[k * 6 + c];
        C[r * 6 + c] = s;
    }
}
static void mat6_mul(const double A[36], const double B[36], double C[36])
{
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += A[r * 6 + k] * B[k * 6 + c];
        C[r * 6 + c] = s;
    }
}

typedef struct {
    double e[6], Ji[36], Jj[36], Om[36];   /* Om = rho1 * Omega */
    double rho0;
} lin_t;

static void linearize_edge(const double* poses, const int32_t* ij, const double* meas, const double* info,
                           const uint8_t* robust, double delta, int32_t k, int want_jac, lin_t* L)
{
    const double* Xi = poses + 12 * (size_t)ij[2 * k];
    const double* Xj = poses + 12 * (size_t)ij[2 * k + 1];
    const double* Z = meas + 12 * (size_t)k;
    const double* Om = info + 36 * (size_t)k;
    uzlo_edge_error(Xi, Xj, Z, L->e);
    double chi = 0;
    for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += Om[r * 6 + c] * L->e[c]; chi += L->e[r] * s; }
    double rho[3] = {chi, 1., 0.};
    if (robust[k]) uzlo_huber(chi, delta, rho);
    L->rho0 = rho[0];
    for (int i = 0; i < 36; i++) L->Om[i] = rho[1] * Om[i];                /* robustInformation = rho1 * Omega */
    if (want_jac) uzlo_edge_jacobians(Xi, Xj, Z, L->Ji, L->Jj);
}

/* Threads of the "all cores" baseline build (-fopenmp, oracle/Makefile target `native`): g2o parallelises over the edges
 * (computeActiveErrors / linearizeOplus under OpenMP when built so) and leaves CSparse serial; the same split here.  Every
 * per-edge quantity is computed in parallel and ACCUMULATED SERIALLY IN EDGE ORDER, so results are bit-identical to the
 * single-thread build for any thread count. */
static int g_threads = 1;
void uzlo_set_threads(int32_t t) { g_threads = t > 0 ? t : 1; }
int32_t uzlo_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

double uzlo_chi2(int32_t n, const double* poses, int32_t e, const int32_t* ij, const double* meas,
                 const double* info, const uint8_t* robust, double huber_delta)
{
    double s = 0;
#ifdef _OPENMP
    if (g_threads > 1 && e > 256) {
        double* r0 = (double*)malloc(sizeof(double) * (size_t)e);
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int32_t k = 0; k < e; k++) { lin_t L; linearize_edge(poses, ij, meas, info, robust, huber_delta, k, 0, &L); r0[k] = L.rho0; }
        for (int32_t k = 0; k < e; k++) s += r0[k];
        free(r0);
        return s;
    }
#endif
    lin_t L;
    for (int32_t k = 0; k < e; k++) { linearize_edge(poses, ij, meas, info, robust, huber_delta, k, 0, &L); s += L.rho0; }
    return s;
}

void uzlo_edge_error_norms(int32_t n, const double* poses, int32_t e, const int32_t* ij,
                           const double* meas, double* err)
{
    for (int32_t k = 0; k < e; k++) {
        double v[6];
        uzlo_edge_error(poses + 12 * (size_t)ij[2 * k], poses + 12 * (size_t)ij[2 * k + 1], meas + 12 * (size_t)k, v);
        double s = 0; for (int r = 0; r < 6; r++) s += v[r] * v[r];
        err[k] = sqrt(s);
    }
}

void uzlo_build_dense(int32_t n, const double* poses, const uint8_t* fixed, int32_t e, const int32_t* ij,
                      const double* meas, const double* info, const uint8_t* robust, double huber_delta,
                      double* H, double* b)
{
    const size_t D = 6 * (size_t)n;
    memset(H, 0, sizeof(double) * D * D);
    memset(b, 0, sizeof(double) * D);
    lin_t L;
    double OJi[36], OJj[36], T[36];
    for (int32_t k = 0; k < e; k++) {
        linearize_edge(poses, ij, meas, info, robust, huber_delta, k, 1, &L);
        const int32_t vi = ij[2 * k], vj = ij[2 * k + 1];
        mat6_mul(L.Om, L.Ji, OJi); mat6_mul(L.Om, L.Jj, OJj);
        double Oe[6];
        for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Om[r * 6 + c] * L.e[c]; Oe[r] = s; }
        const int fi = !fixed[vi], fj = !fixed[vj];
        if (fi) {
            mat6_mul_AtB(L.Ji, OJi, T);
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) H[(6 * (size_t)vi + r) * D + 6 * vi + c] += T[r * 6 + c];
            for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Ji[c * 6 + r] * Oe[c]; b[6 * vi + r] -= s; }
        }
        if (fj) {
            mat6_mul_AtB(L.Jj, OJj, T);
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) H[(6 * (size_t)vj + r) * D + 6 * vj + c] += T[r * 6 + c];
            for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Jj[c * 6 + r] * Oe[c]; b[6 * vj + r] -= s; }
        }
        if (fi && fj) {
            mat6_mul_AtB(L.Ji, OJj, T);
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
                H[(6 * (size_t)vi + r) * D + 6 * vj + c] += T[r * 6 + c];
                H[(6 * (size_t)vj + c) * D + 6 * vi + r] += T[r * 6 + c];
            }
        }
    }
}

/* ==========================================================================================
 *  G8: sparse direct Cholesky on 6x6 blocks (mirrors g2o LinearSolverCSparse: fill-reducing
 *  ordering on the block structure, symbolic factorisation cached across LM trials, numeric
 *  factorisation per trial) [EXT].  Ordering = greedy minimum degree on the block graph.
 * ========================================================================================== */
typedef struct {
    int32_t nb;            /* number of free (non-fixed) blocks */
    int32_t* perm;         /* position -> block */
    int32_t* pos;          /* block -> position */
    int64_t* colptr;       /* per position: start into rowpos/val of the below-diagonal blocks */
    int32_t* rowpos;       /* row positions (sorted ascending) */
    double* diag;          /* nb * 36 */
    double* val;           /* nnz * 36, block (row,col) row-major = L_rc */
    int64_t nnz;
} chol_t;

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

/* adjacency given as CSR over free blocks (no self loops, symmetric, unique) */
static void chol_analyze(chol_t* C, int32_t nb, const int64_t* adjptr, const int32_t* adjidx)
{
    C->nb = nb;
    C->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    C->pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    int32_t** adj = (int32_t**)calloc((size_t)(nb + 1), sizeof(int32_t*));
    int32_t* deg = (int32_t*)calloc((size_t)(nb + 1), sizeof(int32_t));
    int32_t* cap = (int32_t*)calloc((size_t)(nb + 1), sizeof(int32_t));
    uint8_t* done = (uint8_t*)calloc((size_t)(nb + 1), 1);
    int64_t* mark = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb + 1));
    int64_t stamp = 0;
    for (int32_t i = 0; i < nb; i++) mark[i] = -1;
    for (int32_t i = 0; i < nb; i++) {
        deg[i] = (int32_t)(adjptr[i + 1] - adjptr[i]);
        cap[i] = deg[i] > 4 ? deg[i] : 4;
        adj[i] = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap[i]);
        memcpy(adj[i], adjidx + adjptr[i], sizeof(int32_t) * (size_t)deg[i]);
    }
    /* column structures recorded at elimination time */
    int32_t** cstruct = (int32_t**)calloc((size_t)(nb + 1), sizeof(int32_t*));
    int32_t* clen = (int32_t*)calloc((size_t)(nb + 1), sizeof(int32_t));
    /* degree buckets for O(1) min selection */
    int32_t* bhead = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    int32_t* bnext = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    int32_t* bprev = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    for (int32_t i = 0; i <= nb; i++) bhead[i] = -1;
#define BUCKET_INSERT(v) do { int32_t d_ = deg[v]; bnext[v] = bhead[d_]; bprev[v] = -1; if (bhead[d_] >= 0) bprev[bhead[d_]] = (v); bhead[d_] = (v); } while (0)
#define BUCKET_REMOVE(v) do { int32_t d_ = deg[v]; if (bprev[v] >= 0) bnext[bprev[v]] = bnext[v]; else bhead[d_] = bnext[v]; if (bnext[v] >= 0) bprev[bnext[v]] = bprev[v]; } while (0)
    for (int32_t i = nb - 1; i >= 0; i--) BUCKET_INSERT(i);
    int32_t mind = 0;
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    for (int32_t step = 0; step < nb; step++) {
        while (mind <= nb && bhead[mind] < 0) mind++;
        int32_t v = bhead[mind];
        BUCKET_REMOVE(v);
        done[v] = 1;
        C->perm[step] = v; C->pos[v] = step;
        int32_t nv = deg[v];
        int32_t* Nv = adj[v];
        cstruct[v] = Nv; clen[v] = nv;          /* ownership moves to cstruct */
        adj[v] = NULL;
        /* clique update */
        for (int32_t a = 0; a < nv; a++) {
            int32_t u = Nv[a];
            BUCKET_REMOVE(u);
            /* new adj[u] = (adj[u] U Nv) \ {u, v} */
            int32_t cnt = 0;
            stamp++;
            for (int32_t k = 0; k < deg[u]; k++) { int32_t w = adj[u][k]; if (w != v) { mark[w] = stamp; tmp[cnt++] = w; } }
            for (int32_t k = 0; k < nv; k++) { int32_t w = Nv[k]; if (w != u && mark[w] != stamp) { mark[w] = stamp; tmp[cnt++] = w; } }
            if (cnt > cap[u]) { cap[u] = cnt + cnt / 2 + 4; free(adj[u]); adj[u] = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap[u]); }
            memcpy(adj[u], tmp, sizeof(int32_t) * (size_t)cnt);
            deg[u] = cnt;
            BUCKET_INSERT(u);
            if (cnt < mind) mind = cnt;
        }
    }
    /* build column pointers in elimination order, rows as positions, sorted */
    C->colptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb + 1));
    int64_t nnz = 0;
    for (int32_t p = 0; p < nb; p++) { C->colptr[p] = nnz; nnz += clen[C->perm[p]]; }
    C->colptr[nb] = nnz;
    C->nnz = nnz;
    C->rowpos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    for (int32_t p = 0; p < nb; p++) {
        int32_t v = C->perm[p];
        int32_t* dst = C->rowpos + C->colptr[p];
        for (int32_t k = 0; k < clen[v]; k++) dst[k] = C->pos[cstruct[v][k]];
        qsort(dst, (size_t)clen[v], sizeof(int32_t), cmp_i32);
        free(cstruct[v]);
    }
    C->diag = (double*)malloc(sizeof(double) * 36 * (size_t)(nb > 0 ? nb : 1));
    C->val = (double*)malloc(sizeof(double) * 36 * (size_t)(nnz > 0 ? nnz : 1));
    for (int32_t i = 0; i < nb; i++) free(adj[i]);
    free(adj); free(deg); free(cap); free(done); free(mark); free(cstruct); free(clen);
    free(bhead); free(bnext); free(bprev); free(tmp);
}

static void chol_free(chol_t* C)
{
    free(C->perm); free(C->pos); free(C->colptr); free(C->rowpos); free(C->diag); free(C->val);
    memset(C, 0, sizeof(*C));
}

static inline int64_t chol_find(const chol_t* C, int32_t colp, int32_t rowp)
{
    int64_t lo = C->colptr[colp], hi = C->colptr[colp + 1] - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        int32_t r = C->rowpos[mid];
        if (r == rowp) return mid;
        if (r < rowp) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* dense 6x6 lower Cholesky in place (row-major, lower triangle valid). returns 0 on failure */
static int chol6(double* A)
{
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0)) return 0;
        d = sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
        for (int c = j + 1; c < 6; c++) A[j * 6 + c] = 0;
    }
    return 1;
}

/* numeric right-looking block Cholesky. diag/val must already hold the (permuted) lower part of A. */
static int chol_factor(chol_t* C)
{
    const int32_t nb = C->nb;
    for (int32_t k = 0; k < nb; k++) {
        double* Lkk = C->diag + 36 * (size_t)k;
        if (!chol6(Lkk)) return 0;
        const int64_t s = C->colptr[k], e = C->colptr[k + 1];
        /* L_ik = A_ik * Lkk^-T : solve X Lkk^T = A_ik row by row */
        for (int64_t a = s; a < e; a++) {
            double* X = C->val + 36 * (size_t)a;
            for (int r = 0; r < 6; r++) {
                for (int c = 0; c < 6; c++) {
                    double v = X[r * 6 + c];
                    for (int m = 0; m < c; m++) v -= X[r * 6 + m] * Lkk[c * 6 + m];
                    X[r * 6 + c] = v / Lkk[c * 6 + c];
                }
            }
        }
        /* trailing update: A_ij -= L_ik L_jk^T for i >= j in struct(k) */
        for (int64_t b = s; b < e; b++) {
            const int32_t j = C->rowpos[b];
            const double* Ljk = C->val + 36 * (size_t)b;
            double* Djj = C->diag + 36 * (size_t)j;
            for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) {
                double v = 0;
                for (int m = 0; m < 6; m++) v += Ljk[r * 6 + m] * Ljk[c * 6 + m];
                Djj[r * 6 + c] -= v;
            }
            int64_t hint = C->colptr[j];
            for (int64_t a = b + 1; a < e; a++) {
                const int32_t i = C->rowpos[a];
                const double* Lik = C->val + 36 * (size_t)a;
                /* locate (i, j): rows of column j are sorted, and i increases with a */
                int64_t q = hint, qe = C->colptr[j + 1];
                while (q < qe && C->rowpos[q] < i) q++;
                if (q >= qe || C->rowpos[q] != i) return 0;   /* symbolic structure violated */
                hint = q + 1;
                double* Aij = C->val + 36 * (size_t)q;
                for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
                    double v = 0;
                    for (int m = 0; m < 6; m++) v += Lik[r * 6 + m] * Ljk[c * 6 + m];
                    Aij[r * 6 + c] -= v;
                }
            }
        }
    }
    return 1;
}

/* solve L L^T x = b (b, x in permuted block order, in place) */
static void chol_solve(const chol_t* C, double* x)
{
    const int32_t nb = C->nb;
    for (int32_t k = 0; k < nb; k++) {
        const double* Lkk = C->diag + 36 * (size_t)k;
        double* xk = x + 6 * (size_t)k;
        for (int r = 0; r < 6; r++) {
            double v = xk[r];
            for (int m = 0; m < r; m++) v -= Lkk[r * 6 + m] * xk[m];
            xk[r] = v / Lkk[r * 6 + r];
        }
        for (int64_t a = C->colptr[k]; a < C->colptr[k + 1]; a++) {
            const double* L = C->val + 36 * (size_t)a;
            double* xi = x + 6 * (size_t)C->rowpos[a];
            for (int r = 0; r < 6; r++) {
                double v = 0;
                for (int m = 0; m < 6; m++) v += L[r * 6 + m] * xk[m];
                xi[r] -= v;
            }
        }
    }
    for (int32_t k = nb - 1; k >= 0; k--) {
        const double* Lkk = C->diag + 36 * (size_t)k;
        double* xk = x + 6 * (size_t)k;
        for (int64_t a = C->colptr[k]; a < C->colptr[k + 1]; a++) {
            const double* L = C->val + 36 * (size_t)a;
            const double* xi = x + 6 * (size_t)C->rowpos[a];
            for (int c = 0; c < 6; c++) {
                double v = 0;
                for (int m = 0; m < 6; m++) v += L[m * 6 + c] * xi[m];
                xk[c] -= v;
            }
        }
        for (int r = 5; r >= 0; r--) {
            double v = xk[r];
            for (int m = r + 1; m < 6; m++) v -= Lkk[m * 6 + r] * xk[m];
            xk[r] = v / Lkk[r * 6 + r];
        }
    }
}

/* ==========================================================================================
 *  G6 + G7 + G9: BlockSolver::buildSystem, OptimizationAlgorithmLevenberg::solve, oplus [EXT]
 * ========================================================================================== */
typedef struct { int32_t a, b, k; } epair_t;   /* free-block pair (a<b) and the edge it came from */
static int cmp_epair(const void* x, const void* y)
{
    const epair_t* p = (const epair_t*)x; const epair_t* q = (const epair_t*)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    return (p->k > q->k) - (p->k < q->k);
}

int32_t uzlo_pgo_optimize(int32_t n, double* poses, const uint8_t* fixed, int32_t e, const int32_t* ij,
                          const double* meas, const double* info, const uint8_t* robust,
                          double huber_delta, int32_t iterations, uzlo_pgo_stats* st)
{
    uzlo_pgo_stats S;
    memset(&S, 0, sizeof(S));
    const double t_begin = now_ms();
    /* free-block numbering (hessian index) in vertex order */
    int32_t* blk = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t nb = 0;
    for (int32_t v = 0; v < n; v++) blk[v] = fixed[v] ? -1 : nb++;
    S.n_vertices = n; S.n_edges = e;

    /* ---- structure: unique off-diagonal block pairs ---- */
    double t0 = now_ms();
    epair_t* ep = (epair_t*)malloc(sizeof(epair_t) * (size_t)(e > 0 ? e : 1));
    int32_t npair_e = 0;
    for (int32_t k = 0; k < e; k++) {
        int32_t a = blk[ij[2 * k]], b = blk[ij[2 * k + 1]];
        if (a < 0 || b < 0 || a == b) continue;
        if (a > b) { int32_t t = a; a = b; b = t; }
        ep[npair_e].a = a; ep[npair_e].b = b; ep[npair_e].k = k; npair_e++;
    }
    qsort(ep, (size_t)npair_e, sizeof(epair_t), cmp_epair);
    /* unique pairs -> offdiag index; edge -> offdiag index */
    int32_t* e2off = (int32_t*)malloc(sizeof(int32_t) * (size_t)(e > 0 ? e : 1));
    for (int32_t k = 0; k < e; k++) e2off[k] = -1;
    int32_t noff = 0;
    int32_t* offa = (int32_t*)malloc(sizeof(int32_t) * (size_t)(npair_e > 0 ? npair_e : 1));
    int32_t* offb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(npair_e > 0 ? npair_e : 1));
    for (int32_t i = 0; i < npair_e; i++) {
        if (i == 0 || ep[i].a != ep[i - 1].a || ep[i].b != ep[i - 1].b) { offa[noff] = ep[i].a; offb[noff] = ep[i].b; noff++; }
        e2off[ep[i].k] = noff - 1;
    }
    /* symmetric adjacency CSR */
    int64_t* adjptr = (int64_t*)calloc((size_t)nb + 2, sizeof(int64_t));
    for (int32_t i = 0; i < noff; i++) { adjptr[offa[i] + 1]++; adjptr[offb[i] + 1]++; }
    for (int32_t i = 0; i < nb; i++) adjptr[i + 1] += adjptr[i];
    int32_t* adjidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * (size_t)noff + 1));
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb + 1));
    memcpy(fill, adjptr, sizeof(int64_t) * (size_t)(nb + 1));
    for (int32_t i = 0; i < noff; i++) { adjidx[fill[offa[i]]++] = offb[i]; adjidx[fill[offb[i]]++] = offa[i]; }
    free(fill);
    chol_t C;
    memset(&C, 0, sizeof(C));
    chol_analyze(&C, nb, adjptr, adjidx);
    S.t_order_ms = now_ms() - t0;
    S.factor_blocks = C.nnz + nb;
    /* where each H off-diagonal block lands in the factor: (rowpos > colpos), transposed or not */
    t0 = now_ms();
    int64_t* offslot = (int64_t*)malloc(sizeof(int64_t) * (size_t)(noff > 0 ? noff : 1));
    uint8_t* offtr = (uint8_t*)malloc((size_t)(noff > 0 ? noff : 1));
    for (int32_t i = 0; i < noff; i++) {
        int32_t pa = C.pos[offa[i]], pb = C.pos[offb[i]];
        /* H block stored as (a,b) = Ja^T O Jb with a<b in block numbering. Factor needs lower: row>col in position */
        if (pb > pa) { offslot[i] = chol_find(&C, pa, pb); offtr[i] = 1; }   /* L(pb,pa) = H(b,a) = H(a,b)^T */
        else { offslot[i] = chol_find(&C, pb, pa); offtr[i] = 0; }            /* L(pa,pb) = H(a,b)          */
    }
    S.t_symbolic_ms = now_ms() - t0;

    double* Hd = (double*)malloc(sizeof(double) * 36 * (size_t)(nb > 0 ? nb : 1));      /* diagonal blocks */
    double* Ho = (double*)malloc(sizeof(double) * 36 * (size_t)(noff > 0 ? noff : 1));  /* off-diagonal (a<b) */
    double* bv = (double*)malloc(sizeof(double) * 6 * (size_t)(nb > 0 ? nb : 1));
    double* xv = (double*)malloc(sizeof(double) * 6 * (size_t)(nb > 0 ? nb : 1));
    double* xp = (double*)malloc(sizeof(double) * 6 * (size_t)(nb > 0 ? nb : 1));
    double* backup = (double*)malloc(sizeof(double) * 12 * (size_t)(n > 0 ? n : 1));

    double lambda = 0, ni = 2;
    int32_t it_done = 0;
    double chi_final = 0;
    for (int32_t it = 0; it < iterations; it++) {
        /* computeActiveErrors + activeRobustChi2 + buildSystem */
        double tl = now_ms();
        memset(Hd, 0, sizeof(double) * 36 * (size_t)nb);
        memset(Ho, 0, sizeof(double) * 36 * (size_t)noff);
        memset(bv, 0, sizeof(double) * 6 * (size_t)nb);
        double current_chi = 0;
        lin_t L;
        double OJi[36], OJj[36], T[36];
        lin_t* Lall = NULL;
#ifdef _OPENMP
        if (g_threads > 1 && e > 256) {                                  /* errors + Jacobians of all edges in parallel */
            Lall = (lin_t*)malloc(sizeof(lin_t) * (size_t)e);
#pragma omp parallel for num_threads(g_threads) schedule(static)
            for (int32_t k = 0; k < e; k++) linearize_edge(poses, ij, meas, info, robust, huber_delta, k, 1, &Lall[k]);
        }
#endif
        for (int32_t k = 0; k < e; k++) {
            if (Lall) L = Lall[k]; else linearize_edge(poses, ij, meas, info, robust, huber_delta, k, 1, &L);
            current_chi += L.rho0;
            const int32_t a = blk[ij[2 * k]], b = blk[ij[2 * k + 1]];
            double Oe[6];
            for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Om[r * 6 + c] * L.e[c]; Oe[r] = s; }
            mat6_mul(L.Om, L.Ji, OJi); mat6_mul(L.Om, L.Jj, OJj);
            if (a >= 0) {
                mat6_mul_AtB(L.Ji, OJi, T);
                for (int q = 0; q < 36; q++) Hd[36 * (size_t)a + q] += T[q];
                for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Ji[c * 6 + r] * Oe[c]; bv[6 * a + r] -= s; }
            }
            if (b >= 0) {
                mat6_mul_AtB(L.Jj, OJj, T);
                for (int q = 0; q < 36; q++) Hd[36 * (size_t)b + q] += T[q];
                for (int r = 0; r < 6; r++) { double s = 0; for (int c = 0; c < 6; c++) s += L.Jj[c * 6 + r] * Oe[c]; bv[6 * b + r] -= s; }
            }
            if (a >= 0 && b >= 0 && a != b) {
                const int32_t o = e2off[k];
                if (a < b) mat6_mul_AtB(L.Ji, OJj, T); else mat6_mul_AtB(L.Jj, OJi, T);
                for (int q = 0; q < 36; q++) Ho[36 * (size_t)o + q] += T[q];
            }
        }
        free(Lall);
        S.t_linearize_ms += now_ms() - tl;
        if (it == 0) {
            S.chi2_initial = current_chi;
            /* computeLambdaInit: tau * max |H_jj| */
            double maxd = 0;
            for (int32_t a = 0; a < nb; a++) for (int r = 0; r < 6; r++) { double d = fabs(Hd[36 * (size_t)a + r * 7]); if (d > maxd) maxd = d; }
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        double temp_chi = current_chi;
        do {
            memcpy(backup, poses, sizeof(double) * 12 * (size_t)n);       /* push */
            /* (H + lambda I) in permuted factor storage */
            double tn = now_ms();
            memset(C.val, 0, sizeof(double) * 36 * (size_t)C.nnz);
            for (int32_t a = 0; a < nb; a++) {
                double* D = C.diag + 36 * (size_t)C.pos[a];
                memcpy(D, Hd + 36 * (size_t)a, 36 * sizeof(double));
                for (int r = 0; r < 6; r++) D[r * 7] += lambda;
            }
            for (int32_t o = 0; o < noff; o++) {
                double* dst = C.val + 36 * (size_t)offslot[o];
                const double* src = Ho + 36 * (size_t)o;
                if (offtr[o]) { for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) dst[r * 6 + c] = src[c * 6 + r]; }
                else memcpy(dst, src, 36 * sizeof(double));
            }
            int ok2 = chol_factor(&C);
            if (ok2) {
                for (int32_t a = 0; a < nb; a++) memcpy(xp + 6 * (size_t)C.pos[a], bv + 6 * (size_t)a, 6 * sizeof(double));
                chol_solve(&C, xp);
                for (int32_t a = 0; a < nb; a++) memcpy(xv + 6 * (size_t)a, xp + 6 * (size_t)C.pos[a], 6 * sizeof(double));
            } else memset(xv, 0, sizeof(double) * 6 * (size_t)nb);
            S.t_numeric_ms += now_ms() - tn;
            S.lm_trials++;
            /* update: oplus X <- X * fromVectorMQT(dx) (VertexSE3::oplusImpl) */
            for (int32_t v = 0; v < n; v++) {
                if (blk[v] < 0) continue;
                double inc[12];
                uzlo_from_vector_mqt(xv + 6 * (size_t)blk[v], inc);
                se3_mul(poses + 12 * (size_t)v, inc, poses + 12 * (size_t)v);
            }
            temp_chi = uzlo_chi2(n, poses, e, ij, meas, info, robust, huber_delta);
            if (!ok2) temp_chi = 1.7976931348623157e308;
            rho = current_chi - temp_chi;
            double scale = 0;                                               /* computeScale */
            for (int32_t q = 0; q < 6 * nb; q++) scale += xv[q] * (lambda * xv[q] + bv[q]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                if (alpha > 2. / 3.) alpha = 2. / 3.;
                double sf = alpha < 1. / 3. ? 1. / 3. : alpha;
                lambda *= sf;
                ni = 2;
                current_chi = temp_chi;
            } else {
                lambda *= ni;
                ni *= 2;
                memcpy(poses, backup, sizeof(double) * 12 * (size_t)n);   /* pop */
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        chi_final = current_chi;
        it_done = it + 1;
        if (qmax == 10 || rho == 0) { S.terminated_early = 1; break; }     /* Terminate */
    }
    if (iterations <= 0) chi_final = uzlo_chi2(n, poses, e, ij, meas, info, robust, huber_delta);
    S.iterations_done = it_done;
    S.chi2_final = chi_final;
    S.lambda_final = lambda;
    S.t_total_ms = now_ms() - t_begin;
    if (st) *st = S;
    chol_free(&C);
    free(blk); free(ep); free(e2off); free(offa); free(offb); free(adjptr); free(adjidx);
    free(offslot); free(offtr); free(Hd); free(Ho); free(bv); free(xv); free(xp); free(backup);
    return it_done;
}
