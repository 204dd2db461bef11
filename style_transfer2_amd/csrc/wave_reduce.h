// Wave64 / workgroup reductions shared by the HBM-bound passes (gfx950: wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace st2 {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a 256-thread workgroup; result valid in thread 0.  `scratch` holds >= 4 floats per value.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * 4 + wave] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (scratch[i * 4] + scratch[i * 4 + 1]) + (scratch[i * 4 + 2] + scratch[i * 4 + 3]);
    }
}

// Deterministic sum of per-block partials, in double, by one 256-thread workgroup (thread 0 gets it).
__device__ __forceinline__ double sum_partials(const float* part, int n, double* scratch /*[256]*/)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
    scratch[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) scratch[threadIdx.x] += scratch[threadIdx.x + s];
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

__host__ __device__ inline int reduce_grid(size_t n, int per_block, int cap)
{
    size_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

}  // namespace st2
