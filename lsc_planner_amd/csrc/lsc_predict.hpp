// lsc_predict.hpp -- the predicted control points every kernel reads of an agent (device code only; shared by lsc_kernels.hip and lsc_neigh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lsc_gjk.hpp"
#include "lsc_model.hpp"

namespace lsc {

// Predicted / initial control points of agent q for segment m.
//   planner_seq < 2 : pos + vel * m_intp * dt   (float32, src/traj_planner.cpp:699-712, 1030-1037)
//   else            : previous plan shifted by one segment, last segment = 6 x previous end point
__device__ __forceinline__ void load_segment(const float *__restrict__ state, const float *__restrict__ traj_prev, int q,
                                             int m, int planner_seq, float dtf, F3 out[6])
{
#pragma clang fp contract(off)   // float32 semantics of octomath::Vector3: no fused multiply-add
    if (planner_seq < 2) {
        const float *s = state + 9 * q;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float mi = (float)((double)m + (double)i / (double)DEG);
            float ax = (s[3] * mi) * dtf, ay = (s[4] * mi) * dtf, az = (s[5] * mi) * dtf;
            out[i] = F3{s[0] + ax, s[1] + ay, s[2] + az};
        }
    } else {
        const float *t = traj_prev + (size_t)q * NV;
        if (m < M - 1) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                int c = (m + 1) * NC + i;
                out[i] = F3{t[c], t[SEGV + c], t[2 * SEGV + c]};
            }
        } else {
            int c = (M - 1) * NC + DEG;
            F3 e = F3{t[c], t[SEGV + c], t[2 * SEGV + c]};
#pragma unroll
            for (int i = 0; i < 6; i++) out[i] = e;
        }
    }
}

}  // namespace lsc
