// reduce_partials.h -- out[i] = sum over b of part[b][i] for a few hundred to a few thousand values and up to a few thousand per-workgroup
// partials, in ONE fixed order (deterministic) but not as one dependent chain: a workgroup owns 16 values, sixteen 16-lane groups deal the
// partials (group g takes b = g, g + 16, ...) with four independent accumulators each, and the sixteen group sums meet in LDS in a fixed
// tree.  (The first form -- one thread per value looping over all partials -- was a chain of ~1 000 dependent L2 round trips: 170-280 us per
// call, 2.3 ms of a training step spent in "finish" kernels.)
#pragma once
#include <hip/hip_runtime.h>

namespace ps {

template <class T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const T* __restrict__ part, int n_part, int nv, T* __restrict__ out)
{
    __shared__ T red[16][17];
    const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + v;
    T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (i < nv) {
        int b = g;
        for (; b + 48 < n_part; b += 64) {
            a0 += part[(size_t)b * nv + i];
            a1 += part[(size_t)(b + 16) * nv + i];
            a2 += part[(size_t)(b + 32) * nv + i];
            a3 += part[(size_t)(b + 48) * nv + i];
        }
        if (b < n_part) a0 += part[(size_t)b * nv + i];
        if (b + 16 < n_part) a1 += part[(size_t)(b + 16) * nv + i];
        if (b + 32 < n_part) a2 += part[(size_t)(b + 32) * nv + i];
    }
    red[g][v] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < nv) {
        T s = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) s += (red[k][v] + red[k + 1][v]) + (red[k + 2][v] + red[k + 3][v]);
        out[i] = s;
    }
}

// the same reduction with the result split over two arrays: values [0, n0) -> out0, [n0, nv) -> out1 (a weight matrix and its bias vector)
template <class T>
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const T* __restrict__ part, int n_part, int nv, int n0, T* __restrict__ out0, T* __restrict__ out1)
{
    __shared__ T red[16][17];
    const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + v;
    T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (i < nv) {
        int b = g;
        for (; b + 48 < n_part; b += 64) {
            a0 += part[(size_t)b * nv + i];
            a1 += part[(size_t)(b + 16) * nv + i];
            a2 += part[(size_t)(b + 32) * nv + i];
            a3 += part[(size_t)(b + 48) * nv + i];
        }
        if (b < n_part) a0 += part[(size_t)b * nv + i];
        if (b + 16 < n_part) a1 += part[(size_t)(b + 16) * nv + i];
        if (b + 32 < n_part) a2 += part[(size_t)(b + 32) * nv + i];
    }
    red[g][v] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < nv) {
        T s = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) s += (red[k][v] + red[k + 1][v]) + (red[k + 2][v] + red[k + 3][v]);
        if (i < n0) out0[i] = s;
        else out1[i - n0] = s;
    }
}

}  // namespace ps
