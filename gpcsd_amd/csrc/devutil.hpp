// Wavefront / workgroup reduction helpers shared by the eigensolver kernels (64-lane waves, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace gpcsd {

#ifndef GPCSD_EIG_MAXN_BUILD                          // (A/B builds only: -DGPCSD_EIG_MAXN_BUILD=1024 through GPCSD_CXXFLAGS)
#define GPCSD_EIG_MAXN_BUILD 4096
#endif
constexpr int EIG_MAXN = GPCSD_EIG_MAXN_BUILD;       // LDS vectors of the eigensolver are sized for this (= GPCSD_MAX_EIG_N)
constexpr int MAX_BATCH = 4;                         // independent eigenproblems sharing launches
constexpr double EPS_U = 1.1102230246251565e-16;     // unit roundoff (LAPACK dlamch('E'))

// reciprocal to ~2 ulp: hardware seed + two Newton steps (5 dependent operations instead of the ~12 of an IEEE division)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Wave-wide reductions, result valid in every lane.  __shfl_xor on a double compiles to two ds_bpermute_b32 per stage
// (12 LDS-crossbar round trips per reduction, ~600 cycles); the DPP forms below stay in the VALU: four v_mov_dpp
// stages fold each 16-lane row, then the four row totals are combined through v_readlane (8x fewer cycles, same
// fixed association for every call -> bit-reproducible).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every control used here has a source lane for every lane, so `old` is never read: with bound_ctrl set and a
    // constant old the compiler drops the v_mov that otherwise copies the source into the tied destination first
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() waits for EVERY outstanding memory operation of the wave,
// so a global load issued in front of it as a prefetch has landed before the barrier lets anybody through; here the loads stay
// in flight (the compiler still waits for them where their registers are first used).  For phases that communicate through
// LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double lane_get(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// DPP controls: 0xB1 = quad_perm [1,0,3,2], 0x4E = quad_perm [2,3,0,1], 0x141 = row_half_mirror, 0x140 = row_mirror
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
__device__ __forceinline__ double wave_prod(double v) {
    v *= dpp_mov<0xB1>(v);
    v *= dpp_mov<0x4E>(v);
    v *= dpp_mov<0x141>(v);
    v *= dpp_mov<0x140>(v);
    return (lane_get(v, 0) * lane_get(v, 16)) * (lane_get(v, 32) * lane_get(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x141>(v));
    v = fmax(v, dpp_mov<0x140>(v));
    return fmax(fmax(lane_get(v, 0), lane_get(v, 16)), fmax(lane_get(v, 32), lane_get(v, 48)));
}

// Reductions over a workgroup of NW waves; the result is valid in every thread.  red: >= NW doubles of LDS.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double *red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmax(s, red[i]);
    return s;
}
__device__ __forceinline__ double block_sum256(double v, double *red) { return block_sum<4>(v, red); }
__device__ __forceinline__ double block_max256(double v, double *red) { return block_max<4>(v, red); }

}  // namespace gpcsd
