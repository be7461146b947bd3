// Calibration 2: MFMA chain fed from LDS fragment reads (ds_read_b128 per 4 MFMAs), no barriers,
// no fills — isolates the LDS-read -> MFMA path of the GEMM k-loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* A = lds + ((wave >> 2) * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
    const float* B = lds + 4096 + ((wave & 3) * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
    f32x4 a[2], b[2];
    a[0] = *(const f32x4*)A; b[0] = *(const f32x4*)B;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (MODE == 0) { a[(kc + 1) & 1] = *(const f32x4*)(A + ((kc + 1) & 3) * 8); b[(kc + 1) & 1] = *(const f32x4*)(B + ((kc + 1) & 3) * 8); }
            if (MODE == 2) { a[(kc + 1) & 1] = *(const f32x4*)(A + ((kc + 1) & 3) * 8); b[(kc+1)&1] = b[kc&1]; }
            if (MODE == 1) { a[(kc + 1) & 1] = a[kc & 1]; b[(kc + 1) & 1] = b[kc & 1]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc & 1][e], b[kc & 1][e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(int grid, int block, int iters, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int wps = block / 256;
    double ns = ms * 1e6 / (16.0 * iters * wps);
    printf("mode=%d grid=%d block=%d: %.3f ms, %.1f cyc per MFMA per SIMD @2.4GHz\n", MODE, grid, block, ms, ns * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 24);
    int iters = 4000;
    run<1>(1, 512, iters, out); run<0>(1, 512, iters, out); run<2>(1, 512, iters, out);
    run<1>(1, 256, iters, out); run<0>(1, 256, iters, out);
    run<0>(256, 512, iters, out);
    return 0;
}
