// Calibration for a split-operand GEMM loop (fp32 values carried as P bf16 planes, NPROD bf16 MFMAs per fp32 product):
// per k-tile (K=32) and wave (64x32 wave tile): P*3 fragment reads per k-step, NPROD*2 MFMAs per k-step, NDMA LDS-DMA
// pieces, one barrier.  Prints cycles per k-tile per SIMD (2 waves per SIMD) against the fp32 kernel's measured 4525.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int P, int NPROD, int NDMA, bool BAR>
__global__ __launch_bounds__(512) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 24576 floats frag area (96 KB) + 3 * 4096 fill area
    for (int i = threadIdx.x; i < 24576; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i) for (int s = 0; s < 3; ++s) for (int r = 0; r < 16; ++r) acc[i][s][r] = 0.f;
    // plane p of A tile i / B: rows of 64 B (16 floats), swizzled chunk
    const int r = lane & 31, h = lane >> 5;
    auto frag = [&](int base_row, int plane, int ks) -> const f32x4* {
        const int row = base_row + r;
        const int chunk = (ks * 2 + h) ^ ((row >> 2) & 3);
        return (const f32x4*)(lds + plane * 4096 + row * 16 + chunk * 4);
    };
    float* fill = lds + 24576;
    const float* g = src + (size_t)blockIdx.x * 65536 + wave * 256 + lane * 4;
    for (int it = 0; it < iters; ++it) {
        if (BAR) { if (NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        if (NDMA) {
            const int st = it % 3;
#pragma unroll
            for (int j = 0; j < NDMA; ++j) dma16(g + ((it * NDMA + j) & 15) * 4096, fill + st * 4096 + ((j * 8 + wave) & 15) * 256);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 a[2][P], b[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                a[0][p] = *frag((wave >> 2) * 64, p, ks);
                a[1][p] = *frag((wave >> 2) * 64 + 32, p, ks);
                b[p] = *frag(128 + (wave & 3) * 32, P + p, ks);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // products ordered by significance: (0,0) | (0,1),(1,0) | (0,2),(2,0),(1,1)
                constexpr int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1}, sl[6] = {0, 1, 1, 2, 2, 2};
#pragma unroll
                for (int q = 0; q < NPROD; ++q) {
                    const int ia = pa[q] < P ? pa[q] : 0, ib = pb[q] < P ? pb[q] : 0;
                    acc[i][sl[q]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][ia]), __builtin_bit_cast(bf16x8, b[ib]), acc[i][sl[q]], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = fill[threadIdx.x];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 3; ++q) for (int r2 = 0; r2 < 16; ++r2) s += acc[i][q][r2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int P, int NPROD, int NDMA, bool BAR> void run(int grid, int iters, float* out, const float* src) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    size_t lds = (24576 + 3 * 4096) * 4;
    hipLaunchKernelGGL((k<P, NPROD, NDMA, BAR>), dim3(grid), dim3(512), lds, 0, out, src, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<P, NPROD, NDMA, BAR>), dim3(grid), dim3(512), lds, 0, out, src, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / iters;
    double ideal = 2.0 * 2 * 2 * NPROD * 32;   // 2 waves/SIMD x 2 k-steps x 2 tiles x NPROD MFMAs x 32 cycles
    printf("P=%d NPROD=%d ndma=%d bar=%d grid=%d: %.0f 'cycles@2.4GHz' per k-tile(K=32) per SIMD (MFMA-ideal %.0f; fp32 kernel: 4525) -> x%.2f vs fp32\n",
           P, NPROD, NDMA, BAR, grid, cyc, ideal, 4525.0 / cyc);
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&src, (size_t)256 * 65536 * 4 + (1 << 20)); (void)hipMemset(src, 0, (size_t)256 * 65536 * 4 + (1 << 20));
    int iters = 4000;
    run<3, 6, 0, false>(1, iters, out, src);
    run<3, 6, 0, true>(256, iters, out, src);
    run<3, 6, 6, true>(1, iters, out, src);
    run<3, 6, 6, true>(256, iters, out, src);
    run<2, 3, 4, true>(256, iters, out, src);
    run<1, 1, 2, true>(256, iters, out, src);
    return 0;
}
