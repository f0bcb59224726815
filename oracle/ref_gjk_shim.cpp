// extern "C" shim around the REFERENCE's openGJK (compiled from /root/reference in place by
// oracle/Makefile; never copied).  Lets tests call gjk() with flat arrays via ctypes.
#include "openGJK/openGJK.hpp"

extern "C" double ref_gjk(const double* pts1, int n1, const double* pts2, int n2, double* v, int* nvrtx)
{
    bd b1, b2;
    b1.numpoints = n1;
    b2.numpoints = n2;
    for (int i = 0; i < n1; i++) b1.coord.push_back({pts1[3 * i], pts1[3 * i + 1], pts1[3 * i + 2]});
    for (int i = 0; i < n2; i++) b2.coord.push_back({pts2[3 * i], pts2[3 * i + 1], pts2[3 * i + 2]});
    simplex s;
    s.nvrtx = 0;
    double d = gjk(b1, b2, &s, v);
    if (nvrtx) *nvrtx = s.nvrtx;
    return d;
}
