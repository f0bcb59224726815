/*
 * lsc_oracle.c -- CPU restatement of the lsc_planner replanning tick (see lsc_oracle.h).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Written from the behaviour of the reference (file:line cited per function); no source is
 * copied.  float32 is used exactly where the reference holds octomap::point3d (float[3]) and
 * float64 exactly where it uses double, because the LSC margins depend on those roundings.
 * Compile with -ffp-contract=off (the reference's x86-64 build has no FMA contraction).
 */
#include "lsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------------------------------
 * Constants
 * ---------------------------------------------------------------------------------------- */

int orc_segments(void) { return ORC_M; }

static int binom(int n, int k)
{
    if (k < 0 || k > n) return 0;
    int r = 1;
    for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
    return r;
}

/* include/polynomial.hpp:415-426 : B(i,j) = C(n,i) C(n-i,n-j) (-1)^(j-i), j>=i */
void orc_bernstein_basis(double B[ORC_NC * ORC_NC])
{
    for (int i = 0; i < ORC_NC; i++)
        for (int j = 0; j < ORC_NC; j++)
            B[i * ORC_NC + j] = (j >= i) ? binom(ORC_N, i) * binom(ORC_N - i, ORC_N - j) * (((j - i) & 1) ? -1.0 : 1.0)
                                         : 0.0;
}

/* include/polynomial.hpp:224-234 : n (n-1) ... (n-phi+1), 0 when n < phi */
static int falling(int n, int phi)
{
    if (n < phi) return 0;
    int c = 1;
    for (int i = 0; i < phi; i++) c *= n - i;
    return c;
}

/* src/traj_optimizer.cpp:169-184 with phi_n = 1: Q = B Z B^T dt^(-2 phi + 1) */
void orc_qbase(double dt, double Q[ORC_NC * ORC_NC])
{
    double B[ORC_NC * ORC_NC], Z[ORC_NC * ORC_NC], T[ORC_NC * ORC_NC];
    orc_bernstein_basis(B);
    const int k = ORC_PHI;
    for (int i = 0; i < ORC_NC; i++)
        for (int j = 0; j < ORC_NC; j++) {
            int den = i + j - 2 * k + 1;
            Z[i * ORC_NC + j] = (den > 0) ? (double)falling(i, k) * falling(j, k) / den : 0.0;
        }
    for (int i = 0; i < ORC_NC; i++)
        for (int j = 0; j < ORC_NC; j++) {
            double s = 0;
            for (int l = 0; l < ORC_NC; l++) s += B[i * ORC_NC + l] * Z[l * ORC_NC + j];
            T[i * ORC_NC + j] = s;
        }
    double scale = pow(dt, -2 * k + 1);
    for (int i = 0; i < ORC_NC; i++)
        for (int j = 0; j < ORC_NC; j++) {
            double s = 0;
            for (int l = 0; l < ORC_NC; l++) s += T[i * ORC_NC + l] * B[j * ORC_NC + l];
            Q[i * ORC_NC + j] = s * scale;
        }
}

/* src/traj_optimizer.cpp:186-236 : rows 0..2 initial pos/vel/acc, then 3 continuity rows per junction */
void orc_aeq_base(double dt, double A[(ORC_PHI * ORC_M) * ORC_SEGV])
{
    /* derivative stencils at the start / end of a degree-5 Bernstein segment */
    static const double start[3][ORC_NC] = {{1, 0, 0, 0, 0, 0}, {-1, 1, 0, 0, 0, 0}, {1, -2, 1, 0, 0, 0}};
    static const double end[3][ORC_NC] = {{0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, -1, 1}, {0, 0, 0, 1, -2, 1}};
    memset(A, 0, sizeof(double) * (ORC_PHI * ORC_M) * ORC_SEGV);
    int nn = 1;
    for (int j = 0; j < ORC_PHI; j++) {
        for (int i = 0; i < ORC_NC; i++) A[j * ORC_SEGV + i] = pow(dt, -j) * nn * start[j][i];
        nn *= ORC_N - j;
    }
    for (int m = 1; m < ORC_M; m++) {
        nn = 1;
        for (int j = 0; j < ORC_PHI; j++) {
            int r = ORC_PHI + ORC_PHI * (m - 1) + j;
            for (int i = 0; i < ORC_NC; i++) {
                A[r * ORC_SEGV + ORC_NC * (m - 1) + i] = pow(dt, -j) * nn * end[j][i];
                A[r * ORC_SEGV + ORC_NC * m + i] = -pow(dt, -j) * nn * start[j][i];
            }
            nn *= ORC_N - j;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * GJK between conv(pts) and the origin.  Follows src/openGJK/openGJK.cpp:
 *   sub-distance predicates :181-241, S1D :243-256, S2D :259-313, S3D :315-631,
 *   support :633-655, main loop :674-780.
 * Vertex slots keep the reference's order (slot nv-1 is the vertex just added) because
 * the sub-algorithms' decisions and the order after a reduction depend on it.
 * ---------------------------------------------------------------------------------------- */

typedef double v3[3];

static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cpy3(double *d, const double *s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
static inline void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* origin projected on the line through p and q (:149-161) */
static void foot_line(const double *p, const double *q, double *v)
{
    v3 pq = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    double t = dot3(p, pq) / dot3(pq, pq);
    for (int i = 0; i < 3; i++) v[i] = p[i] - pq[i] * t;
}

/* origin projected on the plane through p, q, r (:163-179) */
static void foot_plane(const double *p, const double *q, const double *r, double *v)
{
    v3 pq, pr, nrm;
    for (int i = 0; i < 3; i++) pq[i] = p[i] - q[i];
    for (int i = 0; i < 3; i++) pr[i] = p[i] - r[i];
    cross3(pq, pr, nrm);
    double t = dot3(nrm, p) / dot3(nrm, nrm);
    for (int i = 0; i < 3; i++) v[i] = nrm[i] * t;
}

/* does the origin project inside edge p->q, seen from p?  (:181-193) */
static int edge_keeps(const double *p, const double *q)
{
    double t = 0;
    for (int i = 0; i < 3; i++) t += (p[i] * p[i] - p[i] * q[i]);
    return t > 0;
}

/* (:195-218): 1 when the origin lies on the far side of edge pq within triangle pqr's plane */
static int edge_rejects(const double *p, const double *q, const double *r)
{
    v3 pq, pr, a, b;
    for (int i = 0; i < 3; i++) pq[i] = q[i] - p[i];
    for (int i = 0; i < 3; i++) pr[i] = r[i] - p[i];
    cross3(pq, pr, a);
    cross3(pq, a, b);
    double t = 0;
    for (int i = 0; i < 3; i++) t = t + (p[i] * b[i]);
    return t < 0;
}

/* (:220-241): 0 when p.(pq x pr) > 0 */
static int face_side(const double *p, const double *q, const double *r)
{
    v3 pq, pr, a;
    for (int i = 0; i < 3; i++) pq[i] = q[i] - p[i];
    for (int i = 0; i < 3; i++) pr[i] = r[i] - p[i];
    cross3(pq, pr, a);
    double t = 0;
    for (int i = 0; i < 3; i++) t = t + (p[i] * a[i]);
    return (t > 0) ? 0 : 1;
}

typedef struct {
    int nv;
    v3 w[4];
} gjk_simplex;

static void sub_1d(gjk_simplex *s, double *v)
{
    const double *a = s->w[1], *b = s->w[0];
    if (edge_keeps(a, b)) {
        foot_line(a, b, v);
    } else {
        cpy3(v, a);
        s->nv = 1;
        cpy3(s->w[0], a);
    }
}

static void sub_2d(gjk_simplex *s, double *v)
{
    const double *a = s->w[2], *b = s->w[1], *c = s->w[0];
    int ab = edge_keeps(a, b);
    int ac = edge_keeps(a, c);
    int in_bc = !edge_rejects(a, b, c);
    int in_cb = !edge_rejects(a, c, b);
    enum { FACE, EDGE_AB, EDGE_AC, VERT } pick;
    if (ab) {
        if (in_bc) pick = (ac && !in_cb) ? EDGE_AC : FACE;
        else pick = EDGE_AB;
    } else if (ac) {
        pick = in_cb ? FACE : EDGE_AC;
    } else {
        pick = VERT;
    }
    switch (pick) {
    case FACE:
        foot_plane(a, b, c, v);
        break;
    case EDGE_AB: /* slots become [a, b] */
        foot_line(a, b, v);
        s->nv = 2;
        cpy3(s->w[0], s->w[2]);
        break;
    case EDGE_AC: /* slots become [c, a] */
        foot_line(a, c, v);
        s->nv = 2;
        cpy3(s->w[1], s->w[2]);
        break;
    case VERT:
        cpy3(v, a);
        s->nv = 1;
        cpy3(s->w[0], s->w[2]);
        break;
    }
}

/* keep {a, p, q} as a triangle: slots [q, p, a] */
static void keep_tri(gjk_simplex *s, const double *a, const double *p, const double *q)
{
    v3 ta, tp, tq;
    cpy3(ta, a); cpy3(tp, p); cpy3(tq, q);
    s->nv = 3;
    cpy3(s->w[2], ta); cpy3(s->w[1], tp); cpy3(s->w[0], tq);
}
/* keep {a, p} as an edge: slots [p, a] */
static void keep_edge(gjk_simplex *s, const double *a, const double *p)
{
    v3 ta, tp;
    cpy3(ta, a); cpy3(tp, p);
    s->nv = 2;
    cpy3(s->w[1], ta); cpy3(s->w[0], tp);
}

static void sub_3d(gjk_simplex *s, double *v)
{
    v3 a, P[3]; /* P[2], P[1], P[0] = the three older vertices in slot order 2,1,0 */
    cpy3(a, s->w[3]);
    for (int t = 0; t < 3; t++) cpy3(P[t], s->w[t]);
    v3 e2, e1, e0;
    for (int t = 0; t < 3; t++) { e2[t] = P[2][t] - a[t]; e1[t] = P[1][t] - a[t]; e0[t] = P[0][t] - a[t]; }

    int keep[3];
    keep[2] = edge_keeps(a, P[2]);
    keep[1] = edge_keeps(a, P[1]);
    keep[0] = edge_keeps(a, P[0]);
    int nkeep = keep[2] + keep[1] + keep[0];
    if (nkeep == 0) {
        cpy3(v, a);
        s->nv = 1;
        cpy3(s->w[0], a);
        return;
    }

    /* orientation of the tetrahedron: e1 . (e0 x e2) evaluated as a cofactor expansion (:138-140) */
    double det = e1[0] * ((e0[1] * e2[2]) - (e2[1] * e0[2])) - e1[1] * (e0[0] * e2[2] - e2[0] * e0[2]) +
                 e1[2] * (e0[0] * e2[1] - e2[0] * e0[1]);
    int flip = (det > 0) ? 0 : 1;
    int f2 = face_side(a, P[1], P[0]) - flip; f2 *= f2; /* face opposite slot 2 */
    int f1 = face_side(a, P[0], P[2]) - flip; f1 *= f1; /* face opposite slot 1 */
    int f0 = face_side(a, P[2], P[1]) - flip; f0 *= f0; /* face opposite slot 0 */

    int k, i, j;
    switch (f2 + f1 + f0) {
    case 3: /* origin inside the tetrahedron */
        v[0] = v[1] = v[2] = 0;
        s->nv = 4;
        return;
    case 2: /* exactly one face sees the origin: drop the opposite vertex, keep slot order */
        s->nv = 3;
        if (!f2) {
            cpy3(s->w[2], a);
        } else if (!f1) {
            cpy3(s->w[1], P[2]);
            cpy3(s->w[2], a);
        } else {
            cpy3(s->w[0], P[1]);
            cpy3(s->w[1], P[2]);
            cpy3(s->w[2], a);
        }
        sub_2d(s, v);
        return;
    case 1: {
        s->nv = 3;
        if (f2) { k = 2; i = 1; j = 0; }
        else if (f1) { k = 1; i = 0; j = 2; }
        else { k = 0; i = 2; j = 1; }
        const double *pi = P[i], *pj = P[j], *pk = P[k];
        if (nkeep == 1) {
            if (keep[k]) {
                if (!edge_rejects(a, pk, pi)) { keep_tri(s, a, pi, pk); foot_plane(a, pi, pk, v); }
                else if (!edge_rejects(a, pk, pj)) { keep_tri(s, a, pj, pk); foot_plane(a, pj, pk, v); }
                else { keep_edge(s, a, pk); foot_line(a, pk, v); }
            } else if (keep[i]) {
                if (!edge_rejects(a, pi, pk)) { keep_tri(s, a, pi, pk); foot_plane(a, pi, pk, v); }
                else { keep_edge(s, a, pi); foot_line(a, pi, v); }
            } else {
                if (!edge_rejects(a, pj, pk)) { keep_tri(s, a, pj, pk); foot_plane(a, pj, pk, v); }
                else { keep_edge(s, a, pj); foot_line(a, pj, v); }
            }
        } else if (nkeep == 2) {
            if (keep[i]) {
                if (!edge_rejects(a, pk, pi)) {
                    if (!edge_rejects(a, pi, pk)) { keep_tri(s, a, pi, pk); foot_plane(a, pi, pk, v); }
                    else { keep_edge(s, a, pk); foot_line(a, pk, v); }
                } else {
                    if (!edge_rejects(a, pk, pj)) { keep_tri(s, a, pj, pk); foot_plane(a, pj, pk, v); }
                    else { keep_edge(s, a, pk); foot_line(a, pk, v); }
                }
            } else if (keep[j]) {
                if (!edge_rejects(a, pk, pj)) {
                    if (!edge_rejects(a, pj, pk)) { keep_tri(s, a, pj, pk); foot_plane(a, pj, pk, v); }
                    else { keep_edge(s, a, pj); foot_line(a, pj, v); }
                } else {
                    if (!edge_rejects(a, pk, pi)) { keep_tri(s, a, pi, pk); foot_plane(a, pi, pk, v); }
                    else { keep_edge(s, a, pk); foot_line(a, pk, v); }
                }
            }
            /* else: reference leaves v and the 3 oldest slots untouched (:497-499) */
        } else {
            int r_ik = edge_rejects(a, pi, pk);
            int r_jk = edge_rejects(a, pj, pk);
            int r_ki = edge_rejects(a, pk, pi);
            int r_kj = edge_rejects(a, pk, pj);
            if (r_ki && r_kj) { keep_edge(s, a, pk); foot_line(a, pk, v); }
            else if (r_ki) {
                if (r_jk) { keep_edge(s, a, pj); foot_line(a, pj, v); }
                else { keep_tri(s, a, pj, pk); foot_plane(a, pk, pj, v); }
            } else {
                if (r_ik) { keep_edge(s, a, pi); foot_line(a, pi, v); }
                else { keep_tri(s, a, pi, pk); foot_plane(a, pk, pi, v); }
            }
        }
        return;
    }
    case 0:
        if (nkeep == 1) {
            if (keep[1]) { k = 2; i = 1; j = 0; }
            else if (keep[0]) { k = 1; i = 0; j = 2; }
            else { k = 0; i = 2; j = 1; }
            const double *pi = P[i], *pj = P[j], *pk = P[k];
            if (!edge_rejects(a, pi, pj)) { keep_tri(s, a, pi, pj); foot_plane(a, pi, pj, v); }
            else if (!edge_rejects(a, pi, pk)) { keep_tri(s, a, pi, pk); foot_plane(a, pi, pk, v); }
            else { keep_edge(s, a, pi); foot_line(a, pi, v); }
        } else if (nkeep == 2) {
            s->nv = 3;
            if (!keep[1]) { k = 2; i = 1; j = 0; }
            else if (!keep[0]) { k = 1; i = 0; j = 2; }
            else { k = 0; i = 2; j = 1; }
            const double *pi = P[i], *pj = P[j], *pk = P[k];
            if (!edge_rejects(a, pj, pk)) {
                if (!edge_rejects(a, pk, pj)) { keep_tri(s, a, pj, pk); foot_plane(a, pj, pk, v); }
                else if (!edge_rejects(a, pk, pi)) { keep_tri(s, a, pi, pk); foot_plane(a, pk, pi, v); }
                else { keep_edge(s, a, pk); foot_line(a, pk, v); }
            } else if (!edge_rejects(a, pj, pi)) { keep_tri(s, a, pi, pj); foot_plane(a, pi, pj, v); }
            else { keep_edge(s, a, pj); foot_line(a, pj, v); }
        }
        /* nkeep == 3: reference does nothing; simplex stays at 4 vertices and the loop ends */
        return;
    }
}

double orc_gjk_origin(const double (*pts)[3], int npts, double v[3], int *nvrtx, int *iters)
{
    const double eps_rel = 1e-10, eps_tot = 1e-12;
    gjk_simplex s;
    v3 sup; /* current support point of the hull */
    double wmax2 = 0;
    int k = 0;

    cpy3(v, pts[0]);
    for (int t = 0; t < 3; t++) v[t] = pts[0][t] - 0.0;
    s.nv = 1;
    cpy3(s.w[0], v);
    cpy3(sup, pts[0]);

    do {
        k++;
        v3 neg = {-v[0], -v[1], -v[2]};
        /* support(:633-655): first strict improvement over the previous support's score */
        double best = dot3(sup, neg);
        int better = -1;
        for (int i = 0; i < npts; i++) {
            double sc = dot3(pts[i], neg);
            if (sc > best) { best = sc; better = i; }
        }
        if (better >= 0) cpy3(sup, pts[better]);
        v3 w = {sup[0] - 0.0, sup[1] - 0.0, sup[2] - 0.0};

        double vv = dot3(v, v);
        double gap = vv - dot3(v, w);
        if (gap <= eps_rel * vv || gap < eps_tot) break;
        if (vv < eps_rel * eps_rel) break;

        cpy3(s.w[s.nv], w);
        s.nv++;
        switch (s.nv) {
        case 4: sub_3d(&s, v); break;
        case 3: sub_2d(&s, v); break;
        case 2: sub_1d(&s, v); break;
        }
        for (int j = 0; j < s.nv; j++) {
            double t = dot3(s.w[j], s.w[j]);
            if (t > wmax2) wmax2 = t;
        }
        if (dot3(v, v) <= eps_tot * eps_tot * wmax2) break;
    } while (s.nv != 4 && k != 25);

    if (nvrtx) *nvrtx = s.nv;
    if (iters) *iters = k;
    return sqrt(dot3(v, v));
}

/* ------------------------------------------------------------------------------------------
 * float32 point semantics (octomath::Vector3: float storage, float arithmetic, double norm)
 * ---------------------------------------------------------------------------------------- */

static inline float f32_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void f32_normalize(float *a)
{
    float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    double len = sqrt((double)n2);
    if (len > 0) {
        float l = (float)len;
        a[0] /= l; a[1] /= l; a[2] /= l;
    }
}

static double f32_dist(const float *a, const float *b)
{
    float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    return sqrt((double)n2);
}

/* src/traj_planner.cpp:2030-2043 + include/geometry.hpp:364-394 */
void orc_normal_between_polys(const float (*pa)[3], const float (*po)[3], int npts, float normal[3])
{
    double rel[8][3];
    for (int i = 0; i < npts; i++)
        for (int t = 0; t < 3; t++) {
            float r = pa[i][t] - po[i][t];
            rel[i][t] = (double)r;
        }
    double v[3];
    orc_gjk_origin((const double(*)[3])rel, npts, v, NULL, NULL);
    for (int t = 0; t < 3; t++) normal[t] = 0.0f + (float)v[t];
    f32_normalize(normal);
}

/* src/traj_planner.cpp:829-864 (agent branch) and :997-1016 */
void orc_shift_traj(const float *prev, float *out)
{
    for (int k = 0; k < 3; k++) {
        const float *p = prev + k * ORC_SEGV;
        float *o = out + k * ORC_SEGV;
        for (int m = 0; m < ORC_M - 1; m++)
            for (int i = 0; i < ORC_NC; i++) o[m * ORC_NC + i] = p[(m + 1) * ORC_NC + i];
        for (int i = 0; i < ORC_NC; i++) o[(ORC_M - 1) * ORC_NC + i] = p[(ORC_M - 1) * ORC_NC + ORC_N];
    }
}

/* src/traj_planner.cpp:699-712 / :1030-1037: pos + vel * m_intp * dt, point3d * double -> float factor */
void orc_const_vel_traj(const float pos[3], const float vel[3], double dt, float *out)
{
    for (int m = 0; m < ORC_M; m++)
        for (int i = 0; i < ORC_NC; i++) {
            double m_intp = m + (double)i / ORC_N;
            for (int k = 0; k < 3; k++) {
                float a = vel[k] * (float)m_intp;
                float b = a * (float)dt;
                out[k * ORC_SEGV + m * ORC_NC + i] = pos[k] + b;
            }
        }
}

/* src/traj_planner.cpp:1336-1404 (AGENT obstacle, previous-solution prediction) */
void orc_lsc_pair(const float *init_traj, const float *obs_traj, double r_a, double r_o, double dw_a,
                  double dw_o, float normal[ORC_M][3], double d[ORC_M][ORC_NC])
{
    double downwash = (dw_a * r_a + dw_o * r_o) / (r_a + r_o);
    for (int m = 0; m < ORC_M; m++) {
        float pa[ORC_NC][3], po[ORC_NC][3];
        for (int i = 0; i < ORC_NC; i++) {
            int c = m * ORC_NC + i;
            pa[i][0] = init_traj[c]; pa[i][1] = init_traj[ORC_SEGV + c];
            pa[i][2] = (float)((double)init_traj[2 * ORC_SEGV + c] / downwash); /* util.hpp:231-240 */
            po[i][0] = obs_traj[c]; po[i][1] = obs_traj[ORC_SEGV + c];
            po[i][2] = (float)((double)obs_traj[2 * ORC_SEGV + c] / downwash);
        }
        float nv[3];
        orc_normal_between_polys((const float(*)[3])pa, (const float(*)[3])po, ORC_NC, nv);
        double collision_dist = r_o + r_a;
        for (int i = 0; i < ORC_NC; i++) {
            float rel[3] = {pa[i][0] - po[i][0], pa[i][1] - po[i][1], pa[i][2] - po[i][2]};
            d[m][i] = 0.5 * (collision_dist + (double)f32_dot(rel, nv));
        }
        nv[2] = (float)((double)nv[2] / downwash);
        normal[m][0] = nv[0]; normal[m][1] = nv[1]; normal[m][2] = nv[2];
    }
}

/* ------------------------------------------------------------------------------------------
 * QP assembly (src/traj_optimizer.cpp:261-548)
 * ---------------------------------------------------------------------------------------- */

/* :541-548 */
int orc_terminal_segments(const float goal[3], const float pos[3], double v_nom, double dt)
{
    double flight = f32_dist(goal, pos) / v_nom;
    int t = (int)((ORC_M * dt - flight + 1e-9) / dt);
    return t > 1 ? t : 1;
}

static inline int vidx(int k, int m, int i) { return k * ORC_SEGV + m * ORC_NC + i; }

/* The default-mode assembly (LSC planner, no slack variables, every segment constrained) is the general routine of
 * lsc_oracle_modes.c with its switches off: one restatement of populatebyrow, not two.  The number of variables is
 * orc_qp_nvars(prm): dim * M * (n + 1) with dim = world/dimension (src/traj_optimizer.cpp:8, 264-266), i.e. 60 in a planar
 * world; P has that leading dimension. */
int orc_qp_nvars(const orc_params *prm) { return (prm->world_dimension == 2 ? 2 : 3) * ORC_SEGV; }

int orc_qp_assemble(const orc_params *prm, const float state[9], const float goal[3], double v_nom,
                    const double vmax[3], const double amax[3], int n_obs, const float *obs_traj,
                    const float *normal, const double *d, const float *sfc, double *P, double *c, double *cst,
                    double *lo, double *hi, orc_row *rows)
{
    const orc_modes md = {0, 0, 0.0, -1, 0.0};
    int nv = 0;
    return orc_qp_assemble_ex(prm, &md, state, goal, v_nom, vmax, amax, n_obs, obs_traj, normal, d, sfc, NULL, &nv, P, c, cst, lo,
                              hi, rows);
}

/* ------------------------------------------------------------------------------------------
 * Exact convex QP solve in fp64: equality constraints removed with an orthonormal null-space
 * basis (Householder QR of Aeq^T), inequalities handled by a Mehrotra predictor-corrector
 * interior-point method.  Stands where CPLEX's dual simplex stood (src/traj_optimizer.cpp:76).
 * ---------------------------------------------------------------------------------------- */

static int chol_factor(double *A, int n)
{
    for (int j = 0; j < n; j++) {
        double s = A[j * n + j];
        for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0)) return -1;
        double l = sqrt(s);
        A[j * n + j] = l;
        for (int i = j + 1; i < n; i++) {
            double t = A[i * n + j];
            for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / l;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; i++) {
        double t = b[i];
        for (int k = 0; k < i; k++) t -= L[i * n + k] * b[k];
        b[i] = t / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double t = b[i];
        for (int k = i + 1; k < n; k++) t -= L[k * n + i] * b[k];
        b[i] = t / L[i * n + i];
    }
}

/* Quadruple-precision fallback of the Newton system (round 4).  Close to a near-degenerate optimum the weights z/s of the rows span
 * twenty decades and K = Hy + Z'G' diag(z/s) G Z, assembled in double, is no longer numerically positive definite: the Cholesky
 * factorisation fails, and a diagonal shift -- the previous answer -- damps the Newton steps in exactly the directions that matter
 * (the solver then stalled up to 3.6e-6 relative above the optimum of slack-mode QPs: tests/golden/fuzz_found_6800157.npz, HiGHS and
 * the kernel agreeing with each other).  Assembled and factored (LDL', no square roots) in __float128 the matrix keeps its definiteness;
 * only the iterations that need it pay for it.  Returns 0 and the factor in Kq (unit lower L below the diagonal, D on it). */
typedef __float128 orc_q;
static int quad_build_factor(int NV, int ny, int R, const orc_row *G, const double *w, const double *Z, const double *Hy, orc_q *Kq,
                             orc_q *Kxq, orc_q *KxZq)
{
    for (size_t i = 0; i < (size_t)NV * NV; i++) Kxq[i] = 0;
    for (int r = 0; r < R; r++) {
        const orc_q wr = w[r];
        for (int a = 0; a < G[r].nnz; a++)
            for (int b = 0; b < G[r].nnz; b++) Kxq[G[r].idx[a] * NV + G[r].idx[b]] += wr * (orc_q)G[r].val[a] * (orc_q)G[r].val[b];
    }
    for (size_t i = 0; i < (size_t)NV * ny; i++) KxZq[i] = 0;
    for (int i = 0; i < NV; i++)
        for (int l = 0; l < NV; l++) {
            const orc_q p = Kxq[i * NV + l];
            if (p == 0) continue;
            for (int j = 0; j < ny; j++) KxZq[i * ny + j] += p * (orc_q)Z[l * ny + j];
        }
    for (int i = 0; i < ny * ny; i++) Kq[i] = Hy[i];
    for (int i = 0; i < NV; i++)
        for (int a = 0; a < ny; a++) {
            const double z = Z[i * ny + a];
            if (z == 0) continue;
            for (int b = 0; b < ny; b++) Kq[a * ny + b] += (orc_q)z * KxZq[i * ny + b];
        }
    /* LDL' in place (lower part) */
    for (int j = 0; j < ny; j++) {
        orc_q d = Kq[j * ny + j];
        for (int k = 0; k < j; k++) d -= Kq[j * ny + k] * Kq[j * ny + k] * Kq[k * ny + k];
        if (!(d > 0)) return -1;
        Kq[j * ny + j] = d;
        for (int i = j + 1; i < ny; i++) {
            orc_q t = Kq[i * ny + j];
            for (int k = 0; k < j; k++) t -= Kq[i * ny + k] * Kq[j * ny + k] * Kq[k * ny + k];
            Kq[i * ny + j] = t / d;
        }
    }
    return 0;
}
static void quad_solve(const orc_q *Kq, int n, double *b, orc_q *w)
{
    for (int i = 0; i < n; i++) {
        orc_q t = b[i];
        for (int k = 0; k < i; k++) t -= Kq[i * n + k] * w[k];
        w[i] = t;
    }
    for (int i = 0; i < n; i++) w[i] /= Kq[i * n + i];
    for (int i = n - 1; i >= 0; i--) {
        orc_q t = w[i];
        for (int k = i + 1; k < n; k++) t -= Kq[k * n + i] * w[k];
        w[i] = t;
    }
    for (int i = 0; i < n; i++) b[i] = (double)w[i];
}

/* Dense long-double Gaussian elimination with partial pivoting: solves M x = rhs in place (M n x n row-major, destroyed); 0 ok, 1 singular. */
static int orc_ld_solve(int n, long double *Mx, long double *rhs)
{
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++) if (fabsl(Mx[r * n + c]) > fabsl(Mx[piv * n + c])) piv = r;
        if (fabsl(Mx[piv * n + c]) < 1e-300L) return 1;
        if (piv != c) {
            for (int j = 0; j < n; j++) { long double t = Mx[c * n + j]; Mx[c * n + j] = Mx[piv * n + j]; Mx[piv * n + j] = t; }
            long double t = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = t;
        }
        for (int r = c + 1; r < n; r++) {
            const long double f = Mx[r * n + c] / Mx[c * n + c];
            if (f == 0) continue;
            for (int j = c; j < n; j++) Mx[r * n + j] -= f * Mx[c * n + j];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        long double t = rhs[r];
        for (int j = r + 1; j < n; j++) t -= Mx[r * n + j] * rhs[j];
        rhs[r] = t / Mx[r * n + r];
    }
    return 0;
}

/* Dual active-set solve (Goldfarb & Idnani, Math. Programming 27, 1983) of  min 1/2 y'H y + g'y  s.t.  A y <= b  (H positive definite,
 * n unknowns, R rows, A dense row-major) from the unconstrained optimum with an empty working set.  Test infrastructure: no factor is
 * updated, every step solves its small systems afresh in long double.  Returns 0 with the optimum in yout (every row satisfied to
 * 1e-11 of its scale) and the multipliers in zout, 1 when no admissible step exists (infeasible rows) or after 500 changes of the working set. */
static int orc_gi_polish(int n, const double *H, const double *g, int R, const double *A, const double *b, double *yout, double *zout, int *changes_out)
{
    long double *Hi = (long double *)malloc(sizeof(long double) * (size_t)n * n);       /* H^-1 by Gauss-Jordan on [H | I] with partial pivoting */
    long double *Mx = (long double *)malloc(sizeof(long double) * (size_t)n * 2 * n), *col = NULL;
    int rc = 1, changes = 0;
    {
        const int w = 2 * n;
        int sing = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < w; j++) Mx[i * w + j] = j < n ? (long double)H[i * n + j] : (j - n == i ? 1.0L : 0.0L);
        for (int c = 0; c < n && !sing; c++) {
            int piv = c;
            for (int r = c + 1; r < n; r++) if (fabsl(Mx[r * w + c]) > fabsl(Mx[piv * w + c])) piv = r;
            if (fabsl(Mx[piv * w + c]) < 1e-300L) { sing = 1; break; }
            if (piv != c) for (int j = 0; j < w; j++) { long double t = Mx[c * w + j]; Mx[c * w + j] = Mx[piv * w + j]; Mx[piv * w + j] = t; }
            const long double d = Mx[c * w + c];
            for (int j = 0; j < w; j++) Mx[c * w + j] /= d;
            for (int r = 0; r < n; r++) {
                if (r == c) continue;
                const long double f = Mx[r * w + c];
                if (f != 0) for (int j = c; j < w; j++) Mx[r * w + j] -= f * Mx[c * w + j];
            }
        }
        if (sing) { free(Hi); free(Mx); *changes_out = 0; return 1; }
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Hi[i * n + j] = Mx[i * w + n + j];
    }
    long double *y = (long double *)malloc(sizeof(long double) * (size_t)n), *hn = (long double *)malloc(sizeof(long double) * (size_t)n),
                *zv = (long double *)malloc(sizeof(long double) * (size_t)n);
    int *W = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    long double *u = (long double *)malloc(sizeof(long double) * (size_t)(n + 1)), *rr = (long double *)malloc(sizeof(long double) * (size_t)(n + 1));
    long double *Y = (long double *)malloc(sizeof(long double) * (size_t)(n + 1) * n);   /* H^-1 a_w for the rows of the working set */
    long double *S = (long double *)malloc(sizeof(long double) * (size_t)(n + 1) * (n + 1));
    char *inw = (char *)calloc((size_t)R, 1);
    int q = 0;
    for (int i = 0; i < n; i++) { long double t = 0; for (int j = 0; j < n; j++) t += Hi[i * n + j] * g[j]; y[i] = -t; }
    for (;;) {
        int p = -1;
        long double worst = 1e-11L;
        for (int r = 0; r < R; r++) {
            if (inw[r]) continue;
            long double t = -b[r];
            for (int j = 0; j < n; j++) t += A[(size_t)r * n + j] * y[j];
            t /= 1.0L + fabsl((long double)b[r]);
            if (t > worst) { worst = t; p = r; }
        }
        if (p < 0) { rc = 0; break; }
        const double *np = A + (size_t)p * n;
        long double viol = -b[p], up = 0;
        for (int j = 0; j < n; j++) viol += np[j] * y[j];
        for (int i = 0; i < n; i++) { long double t = 0; for (int j = 0; j < n; j++) t += Hi[i * n + j] * np[j]; hn[i] = t; }
        int fail = 0;
        for (;;) {
            /* r = S^-1 A_W H^-1 n,  z = H^-1 n - Y' r */
            for (int a = 0; a < q; a++) {
                long double t = 0;
                for (int j = 0; j < n; j++) t += A[(size_t)W[a] * n + j] * hn[j];
                rr[a] = t;
                for (int c = 0; c < q; c++) { long double v = 0; for (int j = 0; j < n; j++) v += A[(size_t)W[a] * n + j] * Y[c * n + j]; S[a * q + c] = v; }
            }
            if (q > 0 && orc_ld_solve(q, S, rr)) { fail = 1; break; }
            for (int i = 0; i < n; i++) { long double t = hn[i]; for (int a = 0; a < q; a++) t -= Y[a * n + i] * rr[a]; zv[i] = t; }
            long double zn = 0, nph = 0;
            for (int j = 0; j < n; j++) { zn += np[j] * zv[j]; nph += np[j] * hn[j]; }
            long double t1 = INFINITY;
            int jd = -1;
            for (int a = 0; a < q; a++) if (rr[a] > 1e-18L && u[a] / rr[a] < t1) { t1 = u[a] / rr[a]; jd = a; }
            const long double t2 = zn > 1e-14L * nph ? viol / zn : INFINITY;
            if (!isfinite((double)t1) && !isfinite((double)t2)) { fail = 1; break; }
            const long double t = t1 < t2 ? t1 : t2;
            if (isfinite((double)t2)) { for (int i = 0; i < n; i++) y[i] -= t * zv[i]; viol -= t * zn; }
            for (int a = 0; a < q; a++) { u[a] -= t * rr[a]; if (u[a] < 0) u[a] = 0; }
            up += t;
            if (++changes > 500) { fail = 1; break; }
            if (t2 <= t1) {
                if (q == n) { fail = 1; break; }
                W[q] = p; u[q] = up; inw[p] = 1;
                for (int i = 0; i < n; i++) Y[q * n + i] = hn[i];
                q++;
                break;
            }
            inw[W[jd]] = 0;
            for (int a = jd; a + 1 < q; a++) { W[a] = W[a + 1]; u[a] = u[a + 1]; for (int i = 0; i < n; i++) Y[a * n + i] = Y[(a + 1) * n + i]; }
            q--;
        }
        if (fail) break;
    }
    if (rc == 0) {
        for (int i = 0; i < n; i++) yout[i] = (double)y[i];
        for (int r = 0; r < R; r++) zout[r] = 0.0;                 /* multipliers: those of the working set, zero elsewhere */
        for (int a = 0; a < q; a++) zout[W[a]] = (double)u[a];
    }
    *changes_out = changes;
    free(Hi); free(Mx); free(col); free(y); free(hn); free(zv); free(W); free(u); free(rr); free(Y); free(S); free(inw);
    return rc;
}

int orc_qp_solve(const double *P, const double *c, double cst, const double *lo, const double *hi,
                 const orc_row *rows, int nrows, double *x, double *cost, int *iters, double *kkt)
{
    return orc_qp_solve_n(ORC_NV, P, c, cst, lo, hi, rows, nrows, x, cost, iters, kkt);
}

/* The same solver for NV variables (the 90 control-point coordinates plus the slack variables of the alternate modes,
 * src/traj_optimizer.cpp:306-326): nothing in it depends on what the variables mean. */
int orc_qp_solve_n(const int NV, const double *P, const double *c, double cst, const double *lo, const double *hi,
                   const orc_row *rows, int nrows, double *x, double *cost, int *iters, double *kkt)
{
    /* ---- split rows ---- */
    int neq = 0, nin = 0;
    for (int r = 0; r < nrows; r++) (rows[r].sense == 0) ? neq++ : nin++;
    int nb = 0;
    for (int j = 0; j < NV; j++) { if (isfinite(lo[j])) nb++; if (isfinite(hi[j])) nb++; }
    const int R = nin + nb;

    /* inequalities as g'x <= h, sparse */
    orc_row *G = (orc_row *)malloc(sizeof(orc_row) * (size_t)(R > 0 ? R : 1));
    int gi = 0;
    for (int r = 0; r < nrows; r++) {
        if (rows[r].sense == 0) continue;
        G[gi] = rows[r];
        if (rows[r].sense == 1) {
            for (int j = 0; j < G[gi].nnz; j++) G[gi].val[j] = -G[gi].val[j];
            G[gi].rhs = -G[gi].rhs;
        }
        G[gi].sense = 2;
        gi++;
    }
    for (int j = 0; j < NV; j++) {
        if (isfinite(hi[j])) { G[gi].nnz = 1; G[gi].idx[0] = j; G[gi].val[0] = 1; G[gi].rhs = hi[j]; G[gi].sense = 2; gi++; }
        if (isfinite(lo[j])) { G[gi].nnz = 1; G[gi].idx[0] = j; G[gi].val[0] = -1; G[gi].rhs = -lo[j]; G[gi].sense = 2; gi++; }
    }

    /* ---- Householder QR of Aeq^T (NV x neq) ---- */
    double *W = (double *)calloc((size_t)NV * neq, sizeof(double));
    double *beq = (double *)calloc((size_t)neq, sizeof(double));
    {
        int e = 0;
        for (int r = 0; r < nrows; r++) {
            if (rows[r].sense != 0) continue;
            for (int j = 0; j < rows[r].nnz; j++) W[rows[r].idx[j] * neq + e] = rows[r].val[j];
            beq[e] = rows[r].rhs;
            e++;
        }
    }
    double *HV = (double *)calloc((size_t)NV * neq, sizeof(double)); /* reflector k in column k */
    for (int k = 0; k < neq; k++) {
        double nrm = 0;
        for (int i = k; i < NV; i++) nrm += W[i * neq + k] * W[i * neq + k];
        nrm = sqrt(nrm);
        double alpha = (W[k * neq + k] > 0) ? -nrm : nrm;
        double vn = 0;
        for (int i = k; i < NV; i++) {
            double t = W[i * neq + k] - ((i == k) ? alpha : 0.0);
            HV[i * neq + k] = t;
            vn += t * t;
        }
        vn = sqrt(vn);
        if (vn > 0) for (int i = k; i < NV; i++) HV[i * neq + k] /= vn;
        for (int j = k; j < neq; j++) {
            double s = 0;
            for (int i = k; i < NV; i++) s += HV[i * neq + k] * W[i * neq + j];
            for (int i = k; i < NV; i++) W[i * neq + j] -= 2 * s * HV[i * neq + k];
        }
    }
    /* W now holds R (upper neq x neq).  Q e_j = H_0 ... H_{neq-1} e_j */
    const int ny = NV - neq;
    double *Z = (double *)calloc((size_t)NV * ny, sizeof(double));
    double *xp = (double *)calloc(NV, sizeof(double));
    {
        /* xp = Q [R^-T beq ; 0] */
        double *t = (double *)calloc(NV, sizeof(double));
        for (int i = 0; i < neq; i++) {
            double s = beq[i];
            for (int k = 0; k < i; k++) s -= W[k * neq + i] * t[k];
            t[i] = s / W[i * neq + i];
        }
        for (int k = neq - 1; k >= 0; k--) {
            double s = 0;
            for (int i = k; i < NV; i++) s += HV[i * neq + k] * t[i];
            for (int i = k; i < NV; i++) t[i] -= 2 * s * HV[i * neq + k];
        }
        memcpy(xp, t, sizeof(double) * NV);
        for (int j = 0; j < ny; j++) {
            memset(t, 0, sizeof(double) * NV);
            t[neq + j] = 1;
            for (int k = neq - 1; k >= 0; k--) {
                double s = 0;
                for (int i = k; i < NV; i++) s += HV[i * neq + k] * t[i];
                for (int i = k; i < NV; i++) t[i] -= 2 * s * HV[i * neq + k];
            }
            for (int i = 0; i < NV; i++) Z[i * ny + j] = t[i];
        }
        free(t);
    }

    /* ---- reduced cost: Hy = Z'PZ, gy = Z'(P xp + c) ---- */
    double *PZ = (double *)calloc((size_t)NV * ny, sizeof(double));
    double *Hy = (double *)calloc((size_t)ny * ny, sizeof(double));
    double *gy = (double *)calloc(ny, sizeof(double));
    for (int i = 0; i < NV; i++)
        for (int l = 0; l < NV; l++) {
            double p = P[i * NV + l];
            if (p == 0) continue;
            for (int j = 0; j < ny; j++) PZ[i * ny + j] += p * Z[l * ny + j];
        }
    for (int i = 0; i < NV; i++)
        for (int a = 0; a < ny; a++) {
            double z = Z[i * ny + a];
            for (int b = 0; b < ny; b++) Hy[a * ny + b] += z * PZ[i * ny + b];
        }
    {
        double t[NV];
        for (int i = 0; i < NV; i++) {
            double s = c[i];
            for (int l = 0; l < NV; l++) s += P[i * NV + l] * xp[l];
            t[i] = s;
        }
        for (int a = 0; a < ny; a++) {
            double s = 0;
            for (int i = 0; i < NV; i++) s += Z[i * ny + a] * t[i];
            gy[a] = s;
        }
    }

    double *y = (double *)calloc(ny, sizeof(double));
    double *s = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *z = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *ds = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *dz = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *rp = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *u = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));
    double *K = (double *)calloc((size_t)ny * ny, sizeof(double));
    orc_q *Kq = NULL, *Kxq = NULL, *KxZq = NULL, *wq = NULL, *txq = NULL;      /* quadruple-precision fallback: allocated when first needed */
    int everq = 0, stall = 0;
    double gap0 = 0.0;
    double obj_prev = INFINITY;
    double *Kx = (double *)calloc((size_t)NV * NV, sizeof(double));
    double xx[NV], dx[NV], tx[NV];
    double *rd = (double *)calloc(ny, sizeof(double));
    double *g0 = (double *)calloc(ny, sizeof(double));
    double *dy = (double *)calloc(ny, sizeof(double));
    double *KxZ = (double *)calloc((size_t)NV * ny, sizeof(double));

#define X_FROM_Y(yy, out, with_xp)                                                    \
    for (int i_ = 0; i_ < NV; i_++) {                                                 \
        double s_ = (with_xp) ? xp[i_] : 0.0;                                         \
        for (int a_ = 0; a_ < ny; a_++) s_ += Z[i_ * ny + a_] * (yy)[a_];             \
        (out)[i_] = s_;                                                               \
    }
#define ROWDOT(r_, vec) ({ double s_ = 0; for (int j_ = 0; j_ < G[r_].nnz; j_++) s_ += G[r_].val[j_] * (vec)[G[r_].idx[j_]]; s_; })

    /* build K = Hy + Z' (sum w_r g_r g_r') Z */
    #define BUILD_K(wvec)                                                             \
    do {                                                                              \
        memset(Kx, 0, sizeof(double) * (size_t)NV * NV);                              \
        for (int r_ = 0; r_ < R; r_++) {                                              \
            double w_ = (wvec)[r_];                                                   \
            for (int a_ = 0; a_ < G[r_].nnz; a_++)                                    \
                for (int b_ = 0; b_ < G[r_].nnz; b_++)                                \
                    Kx[G[r_].idx[a_] * NV + G[r_].idx[b_]] += w_ * G[r_].val[a_] * G[r_].val[b_]; \
        }                                                                             \
        memset(KxZ, 0, sizeof(double) * NV * ny);                                     \
        for (int i_ = 0; i_ < NV; i_++)                                               \
            for (int l_ = 0; l_ < NV; l_++) {                                         \
                double p_ = Kx[i_ * NV + l_];                                         \
                if (p_ == 0) continue;                                                \
                for (int j_ = 0; j_ < ny; j_++) KxZ[i_ * ny + j_] += p_ * Z[l_ * ny + j_]; \
            }                                                                         \
        memcpy(K, Hy, sizeof(double) * ny * ny);                                      \
        for (int i_ = 0; i_ < NV; i_++)                                               \
            for (int a_ = 0; a_ < ny; a_++) {                                         \
                double z_ = Z[i_ * ny + a_];                                          \
                if (z_ == 0) continue;                                                \
                for (int b_ = 0; b_ < ny; b_++) K[a_ * ny + b_] += z_ * KxZ[i_ * ny + b_]; \
            }                                                                         \
    } while (0)

    /* Z' G' uvec -> out (ny) */
    #define GT_APPLY(uvec, out)                                                       \
    do {                                                                              \
        memset(tx, 0, sizeof(tx));                                                    \
        for (int r_ = 0; r_ < R; r_++)                                                \
            for (int j_ = 0; j_ < G[r_].nnz; j_++) tx[G[r_].idx[j_]] += G[r_].val[j_] * (uvec)[r_]; \
        for (int a_ = 0; a_ < ny; a_++) {                                             \
            double s_ = 0;                                                            \
            for (int i_ = 0; i_ < NV; i_++) s_ += Z[i_ * ny + a_] * tx[i_];           \
            (out)[a_] = s_;                                                           \
        }                                                                             \
    } while (0)

    int status = 1, it = 0;
    double hmax = 1.0;
    for (int r = 0; r < R; r++) { double a = fabs(G[r].rhs); if (a > hmax) hmax = a; }

    /* ---- initial point: least-squares start (H + A'A) y = -g + A'h ---- */
    for (int r = 0; r < R; r++) u[r] = 1.0;
    BUILD_K(u);
    if (chol_factor(K, ny) != 0) goto done;
    {
        X_FROM_Y(y, xx, 1); /* y = 0 -> xx = xp */
        for (int r = 0; r < R; r++) u[r] = G[r].rhs - ROWDOT(r, xx);
        GT_APPLY(u, dy);
        for (int a = 0; a < ny; a++) dy[a] -= gy[a];
        chol_solve(K, ny, dy);
        memcpy(y, dy, sizeof(double) * ny);
        X_FROM_Y(y, xx, 1);
        double mins = INFINITY, minz = INFINITY;
        for (int r = 0; r < R; r++) {
            double sl = G[r].rhs - ROWDOT(r, xx);
            s[r] = sl; z[r] = -sl;
            if (s[r] < mins) mins = s[r];
            if (z[r] < minz) minz = z[r];
        }
        if (mins <= 0) for (int r = 0; r < R; r++) s[r] += 1.0 - mins;
        if (minz <= 0) for (int r = 0; r < R; r++) z[r] += 1.0 - minz;
    }

    for (it = 0; it < 80; it++) {
        X_FROM_Y(y, xx, 1);
        /* residuals */
        GT_APPLY(z, rd);
        double rdn = 0, rpn = 0, gap = 0;
        for (int a = 0; a < ny; a++) {
            double t = gy[a];
            for (int b = 0; b < ny; b++) t += Hy[a * ny + b] * y[b];
            g0[a] = t;                                    /* cost gradient alone (right-hand sides below) */
            rd[a] += t;
        }
        if (everq) {
            /* once K has needed quadruple precision the multipliers of the degenerate rows are of any size (the complementarity products
             * stay tiny): Z'G'z in double is then round-off only, and with it the stationarity test.  The same sum in quadruple precision. */
            for (int i = 0; i < NV; i++) txq[i] = 0;
            for (int r = 0; r < R; r++)
                for (int j = 0; j < G[r].nnz; j++) txq[G[r].idx[j]] += (orc_q)G[r].val[j] * (orc_q)z[r];
            for (int a = 0; a < ny; a++) {
                orc_q t = g0[a];
                for (int i = 0; i < NV; i++) t += (orc_q)Z[i * ny + a] * txq[i];
                rd[a] = (double)t;
            }
        }
        for (int a = 0; a < ny; a++) if (fabs(rd[a]) > rdn) rdn = fabs(rd[a]);
        for (int r = 0; r < R; r++) {
            rp[r] = ROWDOT(r, xx) + s[r] - G[r].rhs;
            if (fabs(rp[r]) > rpn) rpn = fabs(rp[r]);
            gap += s[r] * z[r];
        }
        double mu = (R > 0) ? gap / R : 0.0;
        double obj = cst;
        for (int i = 0; i < NV; i++) {
            double t = 0;
            for (int l = 0; l < NV; l++) t += P[i * NV + l] * xx[l];
            obj += xx[i] * (0.5 * t + c[i]);
        }
        /* optimal: primal residual at round-off, duality gap 1e-9 relative, stationarity small.
         * (the normal-equation solve loses digits as z/s grows, so rd is held to a looser bound) */
        int gap_ok = gap <= 1e-9 * (1.0 + fabs(obj));
        if (rpn <= 1e-9 * hmax && rdn <= 1e-8 * (1.0 + fabs(obj)) && gap_ok) {
            status = 0;
            break;
        }
        /* last resort (round 4): feasible, complementary to 1e-7 (the bound of the other fallback below) and the objective unchanged to
         * 1e-9 for three iterations in a row -- a run that has converged but passes neither test above.  Two ways there, both found by
         * fuzzing after the right-hand sides were corrected: a slack-mode QP whose optima form a FACE (the iterates drift along it, the
         * Newton step never gets small, the multipliers of the degenerate rows have no limit: tests/golden/fuzz_found_7301082.npz), and a
         * run whose steps are blocked with the gap a few per cent above 1e-9 (1 + |f|) (tick 12 of the forest parity test).  Left
         * running, either loses its multipliers to round-off within a few iterations and ends "infeasible" at the iteration cap.  A run
         * that converges normally goes from 1e-7 to 1e-9 in one or two iterations and never gets here. */
        if (rpn <= 1e-9 * hmax && gap <= 1e-7 * (1.0 + fabs(obj)) && fabs(obj - obj_prev) <= 1e-9 * (1.0 + fabs(obj))) stall++; else stall = 0;
        obj_prev = obj;
        if (stall >= 3) { status = 0; break; }
        if (getenv("ORC_DEBUG")) fprintf(stderr, "it %d rp %.3e rd %.3e gap %.3e obj %.9g\n", it, rpn, rdn, gap, obj);
        if (!isfinite(rdn) || !isfinite(rpn) || !isfinite(mu)) break;
        /* divergence: on an infeasible QP the multipliers run away and the gap grows without bound; a convergent run never exceeds its
         * starting gap by orders of magnitude (the kernel's test, lsc_kernels.hip).  Without it an infeasible QP burns the iteration cap
         * -- since round 4 with a quadruple-precision factorisation in most of those iterations. */
        if (it == 0) gap0 = gap;
        else if (gap > 1e8 * gap0) break;

        for (int r = 0; r < R; r++) u[r] = z[r] / s[r];
        BUILD_K(u);
        int useq = 0;
        const int cfail = chol_factor(K, ny) != 0;
        if (cfail) {
            if (!Kq) {
                Kq = (orc_q *)malloc(sizeof(orc_q) * (size_t)ny * ny); Kxq = (orc_q *)malloc(sizeof(orc_q) * (size_t)NV * NV);
                KxZq = (orc_q *)malloc(sizeof(orc_q) * (size_t)NV * ny); wq = (orc_q *)malloc(sizeof(orc_q) * (size_t)ny);
                txq = (orc_q *)malloc(sizeof(orc_q) * (size_t)NV);
            }
            useq = quad_build_factor(NV, ny, R, G, u, Z, Hy, Kq, Kxq, KxZq) == 0;
            everq = 1;
            if (getenv("ORC_DEBUG")) fprintf(stderr, "chol failed, quadruple precision: %d\n", useq);
        }
        if (cfail && !useq) {
            /* (not positive definite in quadruple precision either.)  K lost definiteness to round-off (z/s spans ~20 decades close
             * to a degenerate optimum).  A solver that stands where CPLEX stood must not call that "infeasible": shift the diagonal by
             * the smallest amount that factors (1e-13 .. 1e-7 of the largest diagonal entry; an inexact Newton step, the residuals stay
             * exact) and go on.  Found by fuzzing the alternate modes (round 2): two such instances in 7.9 k agent-ticks. */
            int fixed = 0;
            double dmax = 0.0;
            for (double shift = 1e-13; shift <= 1e-7 && !fixed; shift *= 100.0) {
                BUILD_K(u);
                if (dmax == 0.0) for (int a = 0; a < ny; a++) if (K[a * ny + a] > dmax) dmax = K[a * ny + a];
                for (int a = 0; a < ny; a++) K[a * ny + a] += shift * dmax;
                fixed = chol_factor(K, ny) == 0;
            }
            if (getenv("ORC_DEBUG")) fprintf(stderr, "chol failed, regularised: %d\n", fixed);
            if (!fixed) {
                if (rpn <= 1e-8 * hmax && gap <= 1e-7 * (1.0 + fabs(obj))) status = 0;
                break;
            }
        }

        /* predictor (sigma = 0): rc = s.z.  The right-hand side is -rd - Z'G'((z rp - s z)/s); the multipliers themselves cancel between
         * the two terms, so it is formed as -(Hy y + gy) - Z'G'(z rp / s) -- with z in the 1e10s close to a degenerate optimum the
         * difference of the two sums had no digits left (round 4; the kernel has always formed it this way). */
        double *rhs = dy;
        for (int r = 0; r < R; r++) u[r] = z[r] * rp[r] / s[r];
        GT_APPLY(u, rhs);
        for (int a = 0; a < ny; a++) rhs[a] = -g0[a] - rhs[a];
        if (useq) quad_solve(Kq, ny, rhs, wq); else chol_solve(K, ny, rhs);
        X_FROM_Y(dy, dx, 0);
        {
            /* Newton-step test: with the gap and the primal residual at tolerance, the affine (pure Newton)
             * step measures the distance to the optimum; the stationarity residual itself can stall at the
             * round-off level of the ill-conditioned normal equations when z/s is huge. */
            double dxn = 0, xn = 1.0;
            for (int i = 0; i < NV; i++) { if (fabs(dx[i]) > dxn) dxn = fabs(dx[i]); if (fabs(xx[i]) > xn) xn = fabs(xx[i]); }
            if (getenv("ORC_DEBUG")) fprintf(stderr, "      newton step %.3e (x %.3e) useq %d\n", dxn, xn, useq);
            if (rpn <= 1e-9 * hmax && gap_ok && dxn <= 1e-9 * xn) { status = 0; break; }
        }
        double alpha = 1.0;
        for (int r = 0; r < R; r++) {
            double adx = ROWDOT(r, dx);
            ds[r] = -rp[r] - adx;
            dz[r] = (-s[r] * z[r] + z[r] * rp[r] + z[r] * adx) / s[r];
            if (ds[r] < 0) { double t = -s[r] / ds[r]; if (t < alpha) alpha = t; }
            if (dz[r] < 0) { double t = -z[r] / dz[r]; if (t < alpha) alpha = t; }
        }
        double mu_aff = 0;
        for (int r = 0; r < R; r++) mu_aff += (s[r] + alpha * ds[r]) * (z[r] + alpha * dz[r]);
        mu_aff = (R > 0) ? mu_aff / R : 0.0;
        double sigma = (mu > 0) ? pow(mu_aff / mu, 3.0) : 0.0;

        /* corrector: rc = s.z + ds_aff.dz_aff - sigma mu */
        for (int r = 0; r < R; r++) {
            double rc = s[r] * z[r] + ds[r] * dz[r] - sigma * mu;
            u[r] = (z[r] * rp[r] - (ds[r] * dz[r] - sigma * mu)) / s[r];      /* (z rp - rc)/s + z: see the predictor */
            ds[r] = rc; /* stash rc */
        }
        GT_APPLY(u, rhs);
        for (int a = 0; a < ny; a++) rhs[a] = -g0[a] - rhs[a];
        if (useq) quad_solve(Kq, ny, rhs, wq); else chol_solve(K, ny, rhs);
        X_FROM_Y(dy, dx, 0);
        alpha = 1.0;
        double amax_ = INFINITY;
        for (int r = 0; r < R; r++) {
            double rc = ds[r];
            double adx = ROWDOT(r, dx);
            ds[r] = -rp[r] - adx;
            dz[r] = (-rc + z[r] * rp[r] + z[r] * adx) / s[r];
            if (ds[r] < 0) { double t = -s[r] / ds[r]; if (t < amax_) amax_ = t; }
            if (dz[r] < 0) { double t = -z[r] / dz[r]; if (t < amax_) amax_ = t; }
        }
        alpha = 0.99 * amax_;
        if (alpha > 1.0) alpha = 1.0;
        for (int a = 0; a < ny; a++) y[a] += alpha * dy[a];
        for (int r = 0; r < R; r++) { s[r] += alpha * ds[r]; z[r] += alpha * dz[r]; }
    }

done:
    /* Polish (round 5).  The interior point stops at a duality gap of 1e-9 (1 + |f|); along directions in which the cost is flat that
     * leaves the plan up to ~1e-4 m from the optimum (tick 28 of the 20-agent circle: HiGHS 4e-6 m from the product's active-set
     * solve, 8.9e-5 m from this solver, whose multipliers there are too rough to read an active set off them -- the fourth time since
     * round 2 that the yardstick was the weaker instrument).  So the optimum is finished EXACTLY by a second, independent method: a dual
     * active-set solve (Goldfarb & Idnani 1983) of the same reduced problem  min 1/2 y'Hy y + gy'y  s.t.  Ay y <= by  from the
     * unconstrained optimum, dense algebra in long double, nothing kept between changes (orc_gi_polish below).  Its point is ACCEPTED
     * only if it satisfies every row to 1e-10 and its objective is not above the interior point's (beyond 1e-9 (1 + |f|)); otherwise the
     * interior point's iterate stands as before.  The STATUS is the interior point's either way.  ORC_NO_POLISH=1 switches it off. */
    if (status == 0 && R > 0 && !getenv("ORC_NO_POLISH")) {
        double *Ay = (double *)malloc(sizeof(double) * (size_t)R * ny), *by = (double *)malloc(sizeof(double) * (size_t)R);
        for (int r = 0; r < R; r++) {
            double off = 0;
            for (int b2 = 0; b2 < ny; b2++) Ay[(size_t)r * ny + b2] = 0;
            for (int j = 0; j < G[r].nnz; j++) {
                const int i = G[r].idx[j];
                off += G[r].val[j] * xp[i];
                for (int b2 = 0; b2 < ny; b2++) Ay[(size_t)r * ny + b2] += G[r].val[j] * Z[i * ny + b2];
            }
            by[r] = G[r].rhs - off;
        }
        double *ynew = (double *)malloc(sizeof(double) * (size_t)ny), *znew = (double *)malloc(sizeof(double) * (size_t)R);
        int changes = 0;
        if (orc_gi_polish(ny, Hy, gy, R, Ay, by, ynew, znew, &changes) == 0) {
            double f_old = 0, f_new = 0;
            for (int a2 = 0; a2 < ny; a2++) {
                double t_old = gy[a2], t_new = gy[a2];
                for (int b2 = 0; b2 < ny; b2++) { t_old += 0.5 * Hy[a2 * ny + b2] * y[b2]; t_new += 0.5 * Hy[a2 * ny + b2] * ynew[b2]; }
                f_old += y[a2] * t_old; f_new += ynew[a2] * t_new;
            }
            if (getenv("ORC_DEBUG")) fprintf(stderr, "polish: %d changes, reduced objective %.12g -> %.12g\n", changes, f_old, f_new);
            if (f_new <= f_old + 1e-9 * (1.0 + fabs(f_old))) {
                memcpy(y, ynew, sizeof(double) * (size_t)ny);
                memcpy(z, znew, sizeof(double) * (size_t)R);
                X_FROM_Y(y, tx, 1);
                for (int r = 0; r < R; r++) { const double sl = G[r].rhs - ROWDOT(r, tx); s[r] = sl > 0 ? sl : 0; }
            }
        } else if (getenv("ORC_DEBUG")) fprintf(stderr, "polish: gave up after %d changes\n", changes);
        free(Ay); free(by); free(ynew); free(znew);
    }
    X_FROM_Y(y, xx, 1);
    memcpy(x, xx, sizeof(double) * NV);
    {
        double obj = cst;
        for (int i = 0; i < NV; i++) {
            double t = 0;
            for (int l = 0; l < NV; l++) t += P[i * NV + l] * xx[l];
            obj += xx[i] * (0.5 * t + c[i]);
        }
        if (cost) *cost = obj;
    }
    if (iters) *iters = it;
    if (kkt) {
        /* stationarity in the null space of Aeq, primal / dual infeasibility, complementarity */
        double st = 0, pf = 0, df = 0, cp = 0;
        GT_APPLY(z, rd);
        for (int a = 0; a < ny; a++) {
            double t = gy[a];
            for (int b = 0; b < ny; b++) t += Hy[a * ny + b] * y[b];
            if (fabs(t + rd[a]) > st) st = fabs(t + rd[a]);
        }
        for (int r = 0; r < R; r++) {
            double sl = G[r].rhs - ROWDOT(r, xx);
            if (-sl > pf) pf = -sl;
            if (-z[r] > df) df = -z[r];
            if (fabs(sl * z[r]) > cp) cp = fabs(sl * z[r]);
        }
        for (int r = 0; r < nrows; r++) {
            if (rows[r].sense != 0) continue;
            double t = -rows[r].rhs;
            for (int j = 0; j < rows[r].nnz; j++) t += rows[r].val[j] * xx[rows[r].idx[j]];
            if (fabs(t) > pf) pf = fabs(t);
        }
        kkt[0] = st; kkt[1] = pf; kkt[2] = df; kkt[3] = cp;
    }
    free(G); free(W); free(beq); free(HV); free(Z); free(xp); free(PZ); free(Hy); free(gy);
    free(y); free(s); free(z); free(ds); free(dz); free(rp); free(u); free(K); free(Kq); free(Kxq); free(KxZq); free(wq); free(txq); free(rd); free(g0); free(dy); free(KxZ); free(Kx);
    return status;
}

/* ------------------------------------------------------------------------------------------
 * State propagation: getStateFromControlPoints at t = dt (include/polynomial.hpp:63-97).
 * t/dt = 1 -> segment 1, local parameter 0: Bernstein weights are exactly (1,0,0,...).
 * ---------------------------------------------------------------------------------------- */
void orc_next_state(const float *traj, double dt, float state[9])
{
    const float fn = (float)ORC_N, fn1 = (float)(ORC_N - 1), finv = (float)pow(dt, -1);
    for (int k = 0; k < 3; k++) {
        const float *c1 = traj + k * ORC_SEGV + ORC_NC; /* segment 1 */
        float v0 = ((c1[1] - c1[0]) * fn) * finv;
        float v1 = ((c1[2] - c1[1]) * fn) * finv;
        float a0 = ((v1 - v0) * fn1) * finv;
        double px = 0.0 + (double)c1[0] * 1.0;
        double vx = 0.0 + (double)v0 * 1.0;
        double ax = 0.0 + (double)a0 * 1.0;
        state[k] = (float)px;
        state[3 + k] = (float)vx;
        state[6 + k] = (float)ax;
    }
}

/* ------------------------------------------------------------------------------------------
 * Goal planning, mode/goal = prior_based, on a map WITHOUT a distance field
 * (TrajPlanner::goalPlanningWithPriority src/traj_planner.cpp:540-608).  The grid A* result
 * (src/grid_based_planner.cpp:53-70) only enters through findLOSFreeGoal (:350-407), whose line-of-sight
 * test passes for every path point when there are neither static obstacles nor a distmap: the LOS goal
 * is then the desired goal clamped to goal_radius from the end of the initial trajectory, so the A*
 * search has no observable effect on empty maps and is not restated here.
 * ---------------------------------------------------------------------------------------- */
void orc_goal_prior_based(int N, int qi, const float *state, const float *desired_goal, const float *prev_traj,
                          int planner_seq, double dt, double goal_threshold, double priority_dist_threshold,
                          double goal_radius, float out_goal[3])
{
    const float *pos = state + 9 * qi;
    const float *goal_i = desired_goal + 3 * qi;
    const double dist_to_goal = f32_dist(pos, goal_i);
    double min_dist_to_obs = 1e9;
    int closest = -1;
    for (int qj = 0; qj < N; qj++) {
        if (qj == qi) continue;
        const float *opos = state + 9 * qj;                       /* obstacle.pose = ideal next state of qj */
        const float *ogoal = desired_goal + 3 * qj;               /* obstacle.goal_point = getDesiredGoalPosition() */
        const double obs_dist_to_goal = f32_dist(opos, ogoal);
        const double dist_to_obs = f32_dist(opos, pos);
        if (obs_dist_to_goal < goal_threshold) continue;          /* :560-562 */
        const float *pt = prev_traj + (size_t)qj * ORC_NV;        /* obs_prev_trajs[oi] = qj's previous plan (unshifted) */
        float a[3], b[3];
        for (int k = 0; k < 3; k++) {
            float last = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N], first = pt[k * ORC_SEGV + ORC_N];
            a[k] = last - first;
            b[k] = first - pos[k];
        }
        if (dist_to_goal > goal_threshold && (double)f32_dot(a, b) > 0) continue;   /* same direction :564-566 */
        if (dist_to_goal < goal_threshold || obs_dist_to_goal < dist_to_goal) {      /* :569-575 */
            if (dist_to_obs < min_dist_to_obs) { min_dist_to_obs = dist_to_obs; closest = qj; }
        }
    }
    const double dist_keep = priority_dist_threshold + 0.1;
    if (min_dist_to_obs < priority_dist_threshold) {              /* retreat :580-587 */
        const float *opos = state + 9 * closest;
        float dir[3] = {opos[0] - pos[0], opos[1] - pos[1], opos[2] - pos[2]};
        f32_normalize(dir);
        for (int k = 0; k < 3; k++) { float s = dir[k] * (float)dist_keep; out_goal[k] = pos[k] - s; }
        return;
    }
    /* findLOSFreeGoal(initial_traj[M-1][n], desired_goal, ...) on an empty map */
    float end[3];
    if (planner_seq < 2) {
        float tmp[ORC_NV];
        orc_const_vel_traj(pos, pos + 3, dt, tmp);
        for (int k = 0; k < 3; k++) end[k] = tmp[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    } else {
        const float *pt = prev_traj + (size_t)qi * ORC_NV;
        for (int k = 0; k < 3; k++) end[k] = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    }
    float delta[3] = {goal_i[0] - end[0], goal_i[1] - end[1], goal_i[2] - end[2]};
    float n2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
    if (sqrt((double)n2) > goal_radius) {
        f32_normalize(delta);
        for (int k = 0; k < 3; k++) { float s = delta[k] * (float)goal_radius; out_goal[k] = end[k] + s; }
    } else {
        for (int k = 0; k < 3; k++) out_goal[k] = goal_i[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * One synchronous tick (src/multi_sync_simulator.cpp:249-337 + src/traj_planner.cpp:344-425)
 * ---------------------------------------------------------------------------------------- */
static const orc_edt *g_edt = NULL;
static double g_wres = 0.1;
static int *g_sfc_init = NULL;
void orc_tick_set_map(const void *edt, double world_res, int *sfc_init_flags)
{
    g_edt = (const orc_edt *)edt; g_wres = world_res; g_sfc_init = sfc_init_flags;
}

int orc_tick(const orc_params *prm, int N, const float *state, const float *goal, const float *prev_traj,
             int planner_seq, const double *radius, const double *downwash, const double *vmax,
             const double *amax, const double *vnom, float *stale_traj, float *sfc_io, float *out_traj,
             double *out_cost, int *out_status, int *out_iters, float *out_normal, double *out_d, int nthreads)
{
    const int n_obs = N - 1;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int qi = 0; qi < N; qi++) {
        float init_traj[ORC_NV];
        float *obs_traj = (float *)malloc(sizeof(float) * ORC_NV * (size_t)(n_obs > 0 ? n_obs : 1));
        float *nrm = (float *)malloc(sizeof(float) * 3 * ORC_M * (size_t)(n_obs > 0 ? n_obs : 1));
        double *dd = (double *)malloc(sizeof(double) * ORC_NC * ORC_M * (size_t)(n_obs > 0 ? n_obs : 1));
        orc_row *rows = (orc_row *)malloc(sizeof(orc_row) * (size_t)(51 + 27 * n_obs + 252 + 162));
        double *P = (double *)malloc(sizeof(double) * ORC_NV * ORC_NV);
        double c[ORC_NV], lo[ORC_NV], hi[ORC_NV], x[ORC_NV], cst;

        /* TrajPlanner::currentStateCallback (src/traj_planner.cpp:304-314): in a planar world the agent's OWN position is taken
         * at z = world/z_2d whatever the message says (the other agents' messages are used as they come) */
        float own[9];
        memcpy(own, state + 9 * qi, sizeof own);
        if (prm->world_dimension == 2) own[2] = (float)prm->world_z_2d;
        /* own initial trajectory :997-1016 (current-velocity model while planner_seq < 2) */
        if (planner_seq < 2) orc_const_vel_traj(own, own + 3, prm->dt, init_traj);
        else orc_shift_traj(prev_traj + (size_t)qi * ORC_NV, init_traj);

        int oi = 0;
        for (int qj = 0; qj < N; qj++) {
            if (qj == qi) continue;
            float *ot = obs_traj + (size_t)oi * ORC_NV;
            if (planner_seq < 2) orc_const_vel_traj(state + 9 * qj, state + 9 * qj + 3, prm->dt, ot);
            else orc_shift_traj(prev_traj + (size_t)qj * ORC_NV, ot);
            double r_o = prm->obs_f32 ? (double)(float)radius[qj] : radius[qj];
            double dw_o = prm->obs_f32 ? (double)(float)downwash[qj] : downwash[qj];
            orc_lsc_pair(init_traj, ot, radius[qi], r_o, downwash[qi], dw_o,
                         (float(*)[3])(nrm + (size_t)oi * ORC_M * 3), (double(*)[ORC_NC])(dd + (size_t)oi * ORC_M * ORC_NC));
            oi++;
        }
        if (out_normal) memcpy(out_normal + (size_t)qi * n_obs * ORC_M * 3, nrm, sizeof(float) * 3 * ORC_M * (size_t)n_obs);
        if (out_d) memcpy(out_d + (size_t)qi * n_obs * ORC_M * ORC_NC, dd, sizeof(double) * ORC_NC * ORC_M * (size_t)n_obs);

        const float *sfc = (prm->use_sfc && sfc_io) ? sfc_io + (size_t)qi * ORC_M * 6 : NULL;
        int sfc_rc = 0;
        if (sfc && g_edt && g_sfc_init)   /* generateSFC, src/traj_planner.cpp:1242-1250 */
            sfc_rc = orc_update_sfc(prm, g_edt, g_wres, own, goal + 3 * qi, prev_traj + (size_t)qi * ORC_NV,
                                    radius[qi], sfc_io + (size_t)qi * ORC_M * 6, &g_sfc_init[qi]);
        int nr = orc_qp_assemble(prm, own, goal + 3 * qi, vnom[qi], vmax + 3 * qi, amax + 3 * qi, n_obs,
                                 obs_traj, nrm, dd, sfc, P, c, &cst, lo, hi, rows);
        double cost;
        int iters = 0;
        const int nv = orc_qp_nvars(prm);
        int st = orc_qp_solve_n(nv, P, c, cst, lo, hi, rows, nr, x, &cost, &iters, NULL);
        /* planar world: only x and y are variables, the stored height is world/z_2d (src/traj_optimizer.cpp:87-90) */
        for (int j = nv; j < ORC_NV; j++) x[j] = (double)(float)prm->world_z_2d;
        if (sfc_rc) st = 4;   /* seed box blocked: the reference throws out of plan(); reported, stale trajectory kept */
        float *o = out_traj + (size_t)qi * ORC_NV;
        float *stale = stale_traj + (size_t)qi * ORC_NV;
        if (st == 0) {
            for (int j = 0; j < ORC_NV; j++) { o[j] = (float)x[j]; stale[j] = o[j]; }
            out_cost[qi] = cost;
        } else {
            /* src/traj_planner.cpp:1553-1584: exception swallowed, optimiser's stale trajectory reused */
            for (int j = 0; j < ORC_NV; j++) o[j] = stale[j];
#ifdef _OPENMP
#pragma omp atomic write
#endif
            rc = 1;
        }
        out_status[qi] = st;
        if (out_iters) out_iters[qi] = iters;
        free(obs_traj); free(nrm); free(dd); free(rows); free(P);
    }
    return rc;
}
