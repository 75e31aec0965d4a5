// Do a wave's MFMAs overlap with ANOTHER wave's VALU work on the same SIMD?  Each wave alternates a phase of NM dependent
// v_mfma_f32_32x32x16_bf16 (or 32x32x2 f32) with a phase of NV dependent v_fma_f32; waves per SIMD = blockDim / 256 (one block per CU).
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_microbench overlap_microbench.hip ; run: ./overlap_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NM, int NV, bool F32>
__global__ void k(float* out, int iters, int stagger)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v = threadIdx.x * 1e-3f, w = 1.0001f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)0.5f; }
    const int wave = threadIdx.x >> 6;
    if (stagger && (wave & 4)) {  // half of the waves start with the VALU phase
#pragma unroll
        for (int i = 0; i < NV; ++i) v = __builtin_fmaf(v, w, 0.25f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, w, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v = __builtin_fmaf(v, w, 0.25f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + acc[0] + acc[5];
}

template <int NM, int NV, bool F32>
void run(const char* name)
{
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 2000;
    for (int stagger = 0; stagger < 2; ++stagger)
        for (int wps : {1, 2, 4}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((k<NM, NV, F32>), dim3(256), dim3(256 * wps), 0, 0, d, 10, stagger);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<NM, NV, F32>), dim3(256), dim3(256 * wps), 0, 0, d, iters, stagger);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * 2.4e9 / iters;  // cycles per iteration per SIMD (nominal 2.4 GHz)
            printf("%-34s %d wave(s)/SIMD stagger %d: %8.0f cycles / iteration  (MFMA alone %d, VALU alone ~%d per wave)\n", name, wps, stagger, cyc,
                   NM * (F32 ? 64 : 32), NV * 4);
        }
}

int main()
{
    run<16, 0, false>("16 bf16 MFMA, no VALU");
    run<0, 128, false>("no MFMA, 128 VALU");
    run<16, 128, false>("16 bf16 MFMA + 128 VALU");
    run<16, 64, false>("16 bf16 MFMA + 64 VALU");
    run<8, 128, true>("8 fp32 MFMA + 128 VALU");
    return 0;
}
