// lsc_wave.hpp -- wave-wide reductions of gfx950 shared by the kernels (device code only).
#pragma once
#include <hip/hip_runtime.h>

namespace lsc {

// Wave-wide reductions without LDS traffic: four DPP row_shr steps inside each 16-lane row, then the four row
// results (lanes 15/31/47/63) are combined through v_readlane.  (__shfl_xor on a double costs two ds_bpermute
// round trips per step; six dependent steps were ~1000 cycles per value.)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v, double identity)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    const int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    // (identity 0.0 = bound_ctrl: a lane without a source reads zero, and the destination needs no initialising move)
    const bool zero_id = ilo == 0 && ihi == 0;
    lo = zero_id ? __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(ilo, lo, CTRL, 0xf, 0xf, false);
    hi = zero_id ? __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(ihi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane_const)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_const);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_const);
    return __hiloint2double(hi, lo);
}
// OP: 0 sum, 1 max, 2 min
template <int OP>
__device__ __forceinline__ double red_op(double a, double b) { return OP == 0 ? a + b : (OP == 1 ? fmax(a, b) : fmin(a, b)); }
// max / min: the moved copy needs no identity.  A lane without a DPP source keeps what its temporary held before -- a value of
// this same reduction (its own, or an earlier stage's) -- and taking an element twice does not change a maximum.  Two moves and
// one v_max_f64 / v_min_f64 per stage instead of four moves, two canonicalisations and the operation.
template <int OP>
__device__ __forceinline__ double hw_extreme(double a, double b)       // v_max_f64 / v_min_f64 as they are (no NaN quieting round trip)
{
    double r;
    if (OP == 1) asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int OP>
__device__ __forceinline__ double wave_extreme(double v)
{
    static_assert(OP == 1 || OP == 2, "max or min");
    int tlo = __double2loint(v), thi = __double2hiint(v);
#define LSC_STAGE(CTRL)                                                                                   \
    tlo = __builtin_amdgcn_update_dpp(tlo, __double2loint(v), CTRL, 0xf, 0xf, false);                     \
    thi = __builtin_amdgcn_update_dpp(thi, __double2hiint(v), CTRL, 0xf, 0xf, false);                     \
    v = hw_extreme<OP>(v, __hiloint2double(thi, tlo));
    LSC_STAGE(0x111) LSC_STAGE(0x112) LSC_STAGE(0x114) LSC_STAGE(0x118)
#undef LSC_STAGE
    return hw_extreme<OP>(hw_extreme<OP>(lane_value(v, 15), lane_value(v, 31)), hw_extreme<OP>(lane_value(v, 47), lane_value(v, 63)));
}
template <int OP>
__device__ __forceinline__ double wave_reduce(double v)
{
    if constexpr (OP != 0) return wave_extreme<OP>(v);
    const double id = 0.0;
    v = red_op<OP>(v, dpp_move<0x111>(v, id));   // row_shr:1
    v = red_op<OP>(v, dpp_move<0x112>(v, id));   // row_shr:2
    v = red_op<OP>(v, dpp_move<0x114>(v, id));   // row_shr:4
    v = red_op<OP>(v, dpp_move<0x118>(v, id));   // row_shr:8  -> lane 15 of every row holds the row result
    return red_op<OP>(red_op<OP>(lane_value(v, 15), lane_value(v, 31)), red_op<OP>(lane_value(v, 47), lane_value(v, 63)));
}
// Up to five values at once, stage by stage: the moves of one value fill the wait states behind another's (a DPP move may not follow the
// instruction that wrote its source by less than two wait states; value by value, a five-value reduction carried ~40 s_nop).
// op[i]: 0 sum, 1 max, 2 min, < 0 unused (compile-time constants at every call site: the selects fold).
template <int CTRL>
__device__ __forceinline__ void red5_stage(double (&v)[5], int (&tlo)[5], int (&thi)[5], const int (&op)[5])
{
    double m[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (op[i] < 0) continue;
        if (op[i] == 0) m[i] = dpp_move<CTRL>(v[i], 0.0);
        else {
            tlo[i] = __builtin_amdgcn_update_dpp(tlo[i], __double2loint(v[i]), CTRL, 0xf, 0xf, false);
            thi[i] = __builtin_amdgcn_update_dpp(thi[i], __double2hiint(v[i]), CTRL, 0xf, 0xf, false);
            m[i] = __hiloint2double(thi[i], tlo[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (op[i] < 0) continue;
        v[i] = op[i] == 0 ? v[i] + m[i] : (op[i] == 1 ? hw_extreme<1>(v[i], m[i]) : hw_extreme<2>(v[i], m[i]));
    }
}
__device__ __forceinline__ void wave_reduce5(double (&v)[5], const int (&op)[5])
{
    int tlo[5], thi[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { tlo[i] = __double2loint(v[i]); thi[i] = __double2hiint(v[i]); }
    red5_stage<0x111>(v, tlo, thi, op);
    red5_stage<0x112>(v, tlo, thi, op);
    red5_stage<0x114>(v, tlo, thi, op);
    red5_stage<0x118>(v, tlo, thi, op);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (op[i] < 0) { v[i] = 0.0; continue; }
        const double a = lane_value(v[i], 15), b = lane_value(v[i], 31), c = lane_value(v[i], 47), d = lane_value(v[i], 63);
        v[i] = op[i] == 0 ? (a + b) + (c + d) : (op[i] == 1 ? hw_extreme<1>(hw_extreme<1>(a, b), hw_extreme<1>(c, d)) : hw_extreme<2>(hw_extreme<2>(a, b), hw_extreme<2>(c, d)));
    }
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<0>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<1>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<2>(v); }


}  // namespace lsc
