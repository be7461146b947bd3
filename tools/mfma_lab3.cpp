// Calibration 3: lab2 + (a) one s_barrier per 16 MFMAs, (b) 3 LDS-DMA pieces per wave per 16 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <bool BAR, int NDMA, int DPOS>
__global__ __launch_bounds__(512) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 16384 floats frag area + 3*6144 fill area
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* A = lds + ((wave >> 2) * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
    const float* B = lds + 4096 + ((wave & 3) * 32 + (lane & 31)) * 36 + (lane >> 5) * 4;
    float* fill = lds + 16384;
    const float* g = src + (size_t)blockIdx.x * 65536 + wave * 256 + lane * 4;
    f32x4 a[2], b[2];
    a[0] = *(const f32x4*)A; b[0] = *(const f32x4*)B;
    for (int it = 0; it < iters; ++it) {
        if (BAR) { if (NDMA && DPOS != 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (NDMA && kc == (DPOS == 9 ? (wave >= 4 ? 2 : 0) : (DPOS == 8 ? 0 : DPOS))) {
                const int st = it % 3;
#pragma unroll
                for (int j = 0; j < NDMA; ++j) dma16(g + ((it * NDMA + j) & 15) * 4096, fill + st * 6144 + (j * 8 + wave) * 256);
            }
            a[(kc + 1) & 1] = *(const f32x4*)(A + ((kc + 1) & 3) * 8); b[(kc + 1) & 1] = *(const f32x4*)(B + ((kc + 1) & 3) * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc & 1][e], b[kc & 1][e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = fill[threadIdx.x]; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool BAR, int NDMA, int DPOS> void run(int grid, int iters, float* out, const float* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    size_t lds = (16384 + 3 * 6144) * 4;
    hipLaunchKernelGGL((k<BAR, NDMA, DPOS>), dim3(grid), dim3(512), lds, 0, out, src, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<BAR, NDMA, DPOS>), dim3(grid), dim3(512), lds, 0, out, src, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / (16.0 * iters * 2);
    printf("bar=%d ndma=%d dpos=%d grid=%d: %.3f ms, %.1f cyc per MFMA per SIMD @2.4GHz\n", BAR, NDMA, DPOS, grid, ms, ns * 2.4);
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&src, (size_t)256 * 65536 * 4 + (1 << 20)); (void)hipMemset(src, 0, (size_t)256 * 65536 * 4 + (1 << 20));
    int iters = 4000;
    run<false, 0, 0>(1, iters, out, src); run<true, 0, 0>(1, iters, out, src);
    run<false, 3, 0>(1, iters, out, src); run<true, 3, 0>(1, iters, out, src); run<true, 3, 2>(1, iters, out, src);
    run<true, 3, 8>(1, iters, out, src); run<true, 6, 8>(1, iters, out, src);
    run<true, 0, 0>(256, iters, out, src); run<true, 3, 0>(256, iters, out, src); run<true, 3, 9>(256, iters, out, src);
    return 0;
}
