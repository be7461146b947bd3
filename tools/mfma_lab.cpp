// Calibration: raw v_mfma_f32_32x32x2_f32 issue rate (dependent chain vs independent accumulators,
// 1 or 2 waves per SIMD) so GEMM loop efficiency can be judged against what the pipe can do.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int grid, int block, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(block), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(block), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_wave = 16.0 * iters;
    int waves_per_simd = block / 256 > 0 ? block / 256 : 1;
    double ns_per_mfma_simd = ms * 1e6 / (mfma_per_wave * waves_per_simd);
    double tf = (double)grid * (block / 64) * mfma_per_wave * 4096 / (ms * 1e-3) / 1e12;
    printf("nacc=%d grid=%d block=%d: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cyc @2.4GHz), %.1f TFLOP/s\n", NACC, grid, block, ms,
           ns_per_mfma_simd, ns_per_mfma_simd * 2.4, tf);
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    int iters = 20000;
    run<1>(1, 256, iters, out); run<1>(1, 512, iters, out); run<2>(1, 256, iters, out); run<4>(1, 256, iters, out);
    run<1>(256, 256, iters, out); run<1>(256, 512, iters, out); run<4>(256, 256, iters, out); run<1>(512, 512, iters, out);
    run<4>(1024, 256, iters, out);
    return 0;
}
