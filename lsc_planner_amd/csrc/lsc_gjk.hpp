// lsc_gjk.hpp -- register-resident GJK for "origin vs. convex hull of 6 points" on gfx950.
//
// Replaces the reference's generic openGJK call made once per (agent, obstacle, segment):
//   closestPointsBetweenPointAndConvexHull  include/geometry.hpp:364-394
//   gjk / support / S1D / S2D / S3D          src/openGJK/openGJK.cpp:674-780, 633-655, 243-631
// Specialised for what the hot path actually passes: body 1 = 6 relative control points
// (float32 values widened to double), body 2 = the origin.  One lane runs one hull; the simplex lives
// in named registers (no dynamically indexed arrays -> no scratch memory), support is a 6-way select.
//
// The arithmetic order of every predicate and projection is the reference's, and FP contraction is
// switched off for this header, so results are bit-identical to the CPU reference (the reference
// build has no FMA).  Vertex slots keep the reference's order because later decisions depend on it.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define LSC_HD __host__ __device__ __forceinline__
#else
#define LSC_HD inline
#endif

#pragma clang fp contract(off)

namespace lsc {

struct D3 {
    double x, y, z;
};

LSC_HD double dot(const D3 &a, const D3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LSC_HD D3 sub(const D3 &a, const D3 &b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LSC_HD D3 cross(const D3 &a, const D3 &b)
{
    return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// foot of the origin on line(p,q)   -- openGJK.cpp:149-161
LSC_HD D3 foot_on_line(const D3 &p, const D3 &q)
{
    D3 pq = sub(p, q);
    double t = dot(p, pq) / dot(pq, pq);
    return D3{p.x - pq.x * t, p.y - pq.y * t, p.z - pq.z * t};
}

// foot of the origin on plane(p,q,r) -- openGJK.cpp:163-179
LSC_HD D3 foot_on_plane(const D3 &p, const D3 &q, const D3 &r)
{
    D3 n = cross(sub(p, q), sub(p, r));
    double t = dot(n, p) / dot(n, n);
    return D3{n.x * t, n.y * t, n.z * t};
}

// openGJK.cpp:181-193
LSC_HD bool keeps_edge(const D3 &p, const D3 &q)
{
    double t = 0.0;
    t += (p.x * p.x - p.x * q.x);
    t += (p.y * p.y - p.y * q.y);
    t += (p.z * p.z - p.z * q.z);
    return t > 0.0;
}

// openGJK.cpp:195-218
LSC_HD bool rejects_edge(const D3 &p, const D3 &q, const D3 &r)
{
    D3 pq = sub(q, p), pr = sub(r, p);
    D3 b = cross(pq, cross(pq, pr));
    double t = 0.0;
    t = t + (p.x * b.x);
    t = t + (p.y * b.y);
    t = t + (p.z * b.z);
    return t < 0.0;
}

// openGJK.cpp:220-241 : 0 when p.(pq x pr) > 0, else 1
LSC_HD int side_of_face(const D3 &p, const D3 &q, const D3 &r)
{
    D3 a = cross(sub(q, p), sub(r, p));
    double t = 0.0;
    t = t + (p.x * a.x);
    t = t + (p.y * a.y);
    t = t + (p.z * a.z);
    return (t > 0.0) ? 0 : 1;
}

struct Simplex {
    D3 w0, w1, w2, w3;  // slot nv-1 holds the vertex added last
    int nv;
};

LSC_HD void set_tri(Simplex &s, const D3 &a, const D3 &p, const D3 &q) { s.nv = 3; s.w2 = a; s.w1 = p; s.w0 = q; }
LSC_HD void set_edge(Simplex &s, const D3 &a, const D3 &p) { s.nv = 2; s.w1 = a; s.w0 = p; }

// 2-simplex: openGJK.cpp:243-256
LSC_HD D3 reduce_edge(Simplex &s)
{
    const D3 a = s.w1, b = s.w0;
    if (keeps_edge(a, b)) return foot_on_line(a, b);
    s.nv = 1;
    s.w0 = a;
    return a;
}

// 3-simplex: openGJK.cpp:259-313
LSC_HD D3 reduce_triangle(Simplex &s)
{
    const D3 a = s.w2, b = s.w1, c = s.w0;
    const bool ab = keeps_edge(a, b);
    const bool ac = keeps_edge(a, c);
    const bool in_bc = !rejects_edge(a, b, c);
    const bool in_cb = !rejects_edge(a, c, b);
    // 0 face, 1 edge ab, 2 edge ac, 3 vertex
    int pick;
    if (ab) pick = in_bc ? ((ac && !in_cb) ? 2 : 0) : 1;
    else if (ac) pick = in_cb ? 0 : 2;
    else pick = 3;
    if (pick == 0) return foot_on_plane(a, b, c);
    if (pick == 1) { s.nv = 2; s.w0 = a; return foot_on_line(a, b); }          // slots [a, b]
    if (pick == 2) { s.nv = 2; s.w1 = a; return foot_on_line(a, c); }          // slots [c, a]
    s.nv = 1;
    s.w0 = a;
    return a;
}

LSC_HD D3 pick3(int idx, const D3 &p0, const D3 &p1, const D3 &p2) { return idx == 0 ? p0 : (idx == 1 ? p1 : p2); }

// 4-simplex: openGJK.cpp:315-631.  v is left untouched on the paths where the reference leaves it.
LSC_HD void reduce_tetra(Simplex &s, D3 &v)
{
    const D3 a = s.w3, p2 = s.w2, p1 = s.w1, p0 = s.w0;
    const bool k2 = keeps_edge(a, p2), k1 = keeps_edge(a, p1), k0 = keeps_edge(a, p0);
    const int nkeep = int(k2) + int(k1) + int(k0);
    if (nkeep == 0) { s.nv = 1; s.w0 = a; v = a; return; }

    const D3 e2 = sub(p2, a), e1 = sub(p1, a), e0 = sub(p0, a);
    const double det = e1.x * ((e0.y * e2.z) - (e2.y * e0.z)) - e1.y * (e0.x * e2.z - e2.x * e0.z) +
                       e1.z * (e0.x * e2.y - e2.x * e0.y);
    const int flip = (det > 0.0) ? 0 : 1;
    int f2 = side_of_face(a, p1, p0) - flip; f2 *= f2;
    int f1 = side_of_face(a, p0, p2) - flip; f1 *= f1;
    int f0 = side_of_face(a, p2, p1) - flip; f0 *= f0;
    const int nface = f2 + f1 + f0;

    if (nface == 3) { v = D3{0.0, 0.0, 0.0}; s.nv = 4; return; }
    if (nface == 2) {
        s.nv = 3;
        if (!f2) { s.w2 = a; }
        else if (!f1) { s.w1 = p2; s.w2 = a; }
        else { s.w0 = p1; s.w1 = p2; s.w2 = a; }
        v = reduce_triangle(s);
        return;
    }
    int k, i, j;
    if (nface == 1) {
        s.nv = 3;
        if (f2) { k = 2; i = 1; j = 0; }
        else if (f1) { k = 1; i = 0; j = 2; }
        else { k = 0; i = 2; j = 1; }
        const D3 pi = pick3(i, p0, p1, p2), pj = pick3(j, p0, p1, p2), pk = pick3(k, p0, p1, p2);
        const bool kk = (k == 2) ? k2 : (k == 1 ? k1 : k0);
        const bool ki = (i == 2) ? k2 : (i == 1 ? k1 : k0);
        const bool kj = (j == 2) ? k2 : (j == 1 ? k1 : k0);
        if (nkeep == 1) {
            if (kk) {
                if (!rejects_edge(a, pk, pi)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pi, pk); }
                else if (!rejects_edge(a, pk, pj)) { set_tri(s, a, pj, pk); v = foot_on_plane(a, pj, pk); }
                else { set_edge(s, a, pk); v = foot_on_line(a, pk); }
            } else if (ki) {
                if (!rejects_edge(a, pi, pk)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pi, pk); }
                else { set_edge(s, a, pi); v = foot_on_line(a, pi); }
            } else {
                if (!rejects_edge(a, pj, pk)) { set_tri(s, a, pj, pk); v = foot_on_plane(a, pj, pk); }
                else { set_edge(s, a, pj); v = foot_on_line(a, pj); }
            }
        } else if (nkeep == 2) {
            if (ki) {
                if (!rejects_edge(a, pk, pi)) {
                    if (!rejects_edge(a, pi, pk)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pi, pk); }
                    else { set_edge(s, a, pk); v = foot_on_line(a, pk); }
                } else {
                    if (!rejects_edge(a, pk, pj)) { set_tri(s, a, pj, pk); v = foot_on_plane(a, pj, pk); }
                    else { set_edge(s, a, pk); v = foot_on_line(a, pk); }
                }
            } else if (kj) {
                if (!rejects_edge(a, pk, pj)) {
                    if (!rejects_edge(a, pj, pk)) { set_tri(s, a, pj, pk); v = foot_on_plane(a, pj, pk); }
                    else { set_edge(s, a, pj); v = foot_on_line(a, pj); }
                } else {
                    if (!rejects_edge(a, pk, pi)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pi, pk); }
                    else { set_edge(s, a, pk); v = foot_on_line(a, pk); }
                }
            }
        } else {
            const bool r_ik = rejects_edge(a, pi, pk), r_jk = rejects_edge(a, pj, pk);
            const bool r_ki = rejects_edge(a, pk, pi), r_kj = rejects_edge(a, pk, pj);
            if (r_ki && r_kj) { set_edge(s, a, pk); v = foot_on_line(a, pk); }
            else if (r_ki) {
                if (r_jk) { set_edge(s, a, pj); v = foot_on_line(a, pj); }
                else { set_tri(s, a, pj, pk); v = foot_on_plane(a, pk, pj); }
            } else {
                if (r_ik) { set_edge(s, a, pi); v = foot_on_line(a, pi); }
                else { set_tri(s, a, pi, pk); v = foot_on_plane(a, pk, pi); }
            }
        }
        return;
    }
    // nface == 0
    if (nkeep == 1) {
        if (k1) { k = 2; i = 1; j = 0; }
        else if (k0) { k = 1; i = 0; j = 2; }
        else { k = 0; i = 2; j = 1; }
        const D3 pi = pick3(i, p0, p1, p2), pj = pick3(j, p0, p1, p2), pk = pick3(k, p0, p1, p2);
        if (!rejects_edge(a, pi, pj)) { set_tri(s, a, pi, pj); v = foot_on_plane(a, pi, pj); }
        else if (!rejects_edge(a, pi, pk)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pi, pk); }
        else { set_edge(s, a, pi); v = foot_on_line(a, pi); }
    } else if (nkeep == 2) {
        s.nv = 3;
        if (!k1) { k = 2; i = 1; j = 0; }
        else if (!k0) { k = 1; i = 0; j = 2; }
        else { k = 0; i = 2; j = 1; }
        const D3 pi = pick3(i, p0, p1, p2), pj = pick3(j, p0, p1, p2), pk = pick3(k, p0, p1, p2);
        if (!rejects_edge(a, pj, pk)) {
            if (!rejects_edge(a, pk, pj)) { set_tri(s, a, pj, pk); v = foot_on_plane(a, pj, pk); }
            else if (!rejects_edge(a, pk, pi)) { set_tri(s, a, pi, pk); v = foot_on_plane(a, pk, pi); }
            else { set_edge(s, a, pk); v = foot_on_line(a, pk); }
        } else if (!rejects_edge(a, pj, pi)) { set_tri(s, a, pi, pj); v = foot_on_plane(a, pi, pj); }
        else { set_edge(s, a, pj); v = foot_on_line(a, pj); }
    }
    // nkeep == 3: the reference does nothing; 4 vertices remain and its loop ends.
}

// Closest point of conv{q0..q5} to the origin (witness vector v); returns |v|.  openGJK.cpp:674-780.
LSC_HD double gjk_origin_hull6(const D3 &q0, const D3 &q1, const D3 &q2, const D3 &q3, const D3 &q4, const D3 &q5,
                               D3 &v, int &nv_out)
{
    const double eps_rel = 1e-10, eps_tot = 1e-12;
    Simplex s;
    v = D3{q0.x - 0.0, q0.y - 0.0, q0.z - 0.0};
    s.nv = 1;
    s.w0 = v; s.w1 = v; s.w2 = v; s.w3 = v;
    D3 sup = q0;
    double wmax2 = 0.0;
    int it = 0;
    do {
        ++it;
        const D3 neg = D3{-v.x, -v.y, -v.z};
        // support: first strict improvement over the previous support's score (openGJK.cpp:633-655)
        double best = dot(sup, neg);
        double sc;
        D3 cand = sup;
        sc = dot(q0, neg); if (sc > best) { best = sc; cand = q0; }
        sc = dot(q1, neg); if (sc > best) { best = sc; cand = q1; }
        sc = dot(q2, neg); if (sc > best) { best = sc; cand = q2; }
        sc = dot(q3, neg); if (sc > best) { best = sc; cand = q3; }
        sc = dot(q4, neg); if (sc > best) { best = sc; cand = q4; }
        sc = dot(q5, neg); if (sc > best) { best = sc; cand = q5; }
        sup = cand;
        const D3 w = D3{sup.x - 0.0, sup.y - 0.0, sup.z - 0.0};

        const double vv = dot(v, v);
        const double gap = vv - dot(v, w);
        if (gap <= eps_rel * vv || gap < eps_tot) break;
        if (vv < eps_rel * eps_rel) break;

        if (s.nv == 1) { s.w1 = w; s.nv = 2; v = reduce_edge(s); }
        else if (s.nv == 2) { s.w2 = w; s.nv = 3; v = reduce_triangle(s); }
        else { s.w3 = w; s.nv = 4; reduce_tetra(s, v); }

        double t = dot(s.w0, s.w0);
        if (t > wmax2) wmax2 = t;
        if (s.nv > 1) { t = dot(s.w1, s.w1); if (t > wmax2) wmax2 = t; }
        if (s.nv > 2) { t = dot(s.w2, s.w2); if (t > wmax2) wmax2 = t; }
        if (s.nv > 3) { t = dot(s.w3, s.w3); if (t > wmax2) wmax2 = t; }
        if (dot(v, v) <= eps_tot * eps_tot * wmax2) break;
    } while (s.nv != 4 && it != 25);
    nv_out = s.nv;
    return sqrt(dot(v, v));
}

// ---- octomath::Vector3 (float32) semantics used around the GJK call -------------------------------
struct F3 {
    float x, y, z;
};

// Vector3::normalized(): norm_sq in float, sqrt in double, divide by (float)len, zero stays zero
LSC_HD F3 normalized_f32(F3 a)
{
    float n2 = a.x * a.x + a.y * a.y + a.z * a.z;
    double len = sqrt((double)n2);
    if (len > 0.0) {
        float l = (float)len;
        a.x /= l; a.y /= l; a.z /= l;
    }
    return a;
}

// One (agent, obstacle, segment) LSC: src/traj_planner.cpp:1338-1404 + :2030-2043.
//   pa[i], po[i]: own initial / obstacle predicted control points (float32, UNSCALED)
//   out: normal (de-scaled, z / downwash) and the six margins d_i
LSC_HD void lsc_segment(const F3 pa[6], const F3 po[6], double downwash, double collision_dist, F3 &normal, double d[6])
{
    F3 ra[6];
    D3 q[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        // coordinateTransform (include/util.hpp:231-240): z <- (float)(z / downwash)
        float az = (float)((double)pa[i].z / downwash);
        float oz = (float)((double)po[i].z / downwash);
        ra[i] = F3{pa[i].x - po[i].x, pa[i].y - po[i].y, az - oz};
        q[i] = D3{(double)ra[i].x, (double)ra[i].y, (double)ra[i].z};
    }
    D3 v;
    int nv;
    gjk_origin_hull6(q[0], q[1], q[2], q[3], q[4], q[5], v, nv);
    F3 n = normalized_f32(F3{0.0f + (float)v.x, 0.0f + (float)v.y, 0.0f + (float)v.z});
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float dp = ra[i].x * n.x + ra[i].y * n.y + ra[i].z * n.z;
        d[i] = 0.5 * (collision_dist + (double)dp);
    }
    n.z = (float)((double)n.z / downwash);
    normal = n;
}

}  // namespace lsc

#pragma clang fp contract(fast)
