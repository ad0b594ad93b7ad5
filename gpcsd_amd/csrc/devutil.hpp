// Wavefront / workgroup reduction helpers shared by the eigensolver kernels (64-lane waves, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace gpcsd {

constexpr int EIG_MAXN = 1024;                       // LDS vectors of the eigensolver are sized for this
constexpr int MAX_BATCH = 4;                         // independent eigenproblems sharing launches
constexpr double EPS_U = 1.1102230246251565e-16;     // unit roundoff (LAPACK dlamch('E'))

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_prod(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
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
