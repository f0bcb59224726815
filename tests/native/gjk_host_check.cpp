// Test-only: compiles the product's device GJK header for the HOST so that the CPU test-suite can
// check its arithmetic against the oracle without a GPU.  Not part of the product library.
#include <cmath>
#include "../../lsc_planner_amd/csrc/lsc_gjk.hpp"

extern "C" void gjkhdr_batch(const double* pts, int count, double* v, double* dist, int* nv)
{
    for (int c = 0; c < count; c++) {
        lsc::D3 q[6];
        for (int i = 0; i < 6; i++) q[i] = lsc::D3{pts[(c * 6 + i) * 3], pts[(c * 6 + i) * 3 + 1], pts[(c * 6 + i) * 3 + 2]};
        lsc::D3 w;
        int n;
        dist[c] = lsc::gjk_origin_hull6(q[0], q[1], q[2], q[3], q[4], q[5], w, n);
        v[3 * c] = w.x; v[3 * c + 1] = w.y; v[3 * c + 2] = w.z;
        nv[c] = n;
    }
}

extern "C" void lschdr_segment(const float* pa, const float* po, double downwash, double cdist, float* normal, double* d)
{
    lsc::F3 a[6], o[6];
    for (int i = 0; i < 6; i++) { a[i] = lsc::F3{pa[3 * i], pa[3 * i + 1], pa[3 * i + 2]}; o[i] = lsc::F3{po[3 * i], po[3 * i + 1], po[3 * i + 2]}; }
    lsc::F3 n;
    lsc::lsc_segment(a, o, downwash, cdist, n, d);
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
}
