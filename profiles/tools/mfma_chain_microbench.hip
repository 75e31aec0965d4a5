// Cost of a DEPENDENT chain of MFMAs (acc = mfma(a, b, acc) back to back) against NA independent accumulators issued round-robin,
// one wave per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -w -o mcb mfma_chain_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int KIND, int NA>
__global__ void k(float* out, int iters)
{
    float v = threadIdx.x * 1e-3f, w = 1.0001f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)0.5f; }
    f32x16 A[NA];
    f32x4 C[NA];
    for (int i = 0; i < NA; ++i) {
        for (int r = 0; r < 16; ++r) A[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) C[i][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16 / NA; ++s)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (KIND == 0) A[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, w, A[i], 0, 0, 0);
                if (KIND == 1) A[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, A[i], 0, 0, 0);
                if (KIND == 2) C[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, w, C[i], 0, 0, 0);
            }
    }
    float r = 0.f;
    for (int i = 0; i < NA; ++i) r += A[i][0] + A[i][7] + C[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int NA>
void run(const char* name, int pipe_cycles)
{
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NA>), dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NA>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %d accumulator(s): %6.1f cycles per MFMA at a nominal 2.4 GHz (pipe time %d)\n", name, NA, ms * 1e-3 * 2.4e9 / iters / 16, pipe_cycles);
}

int main()
{
    run<0, 1>("f32 32x32x2", 64); run<0, 2>("f32 32x32x2", 64); run<0, 4>("f32 32x32x2", 64);
    run<1, 1>("bf16 32x32x16", 32); run<1, 2>("bf16 32x32x16", 32); run<1, 4>("bf16 32x32x16", 32);
    run<2, 1>("f32 16x16x4", 32); run<2, 2>("f32 16x16x4", 32); run<2, 4>("f32 16x16x4", 32); run<2, 8>("f32 16x16x4", 32);
    return 0;
}
