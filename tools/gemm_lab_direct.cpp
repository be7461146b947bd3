// Lab: "direct-fragment" fp32 MFMA GEMM — no LDS, no barriers.  Every wave owns a
// (32*MI)x(32*NI) output tile and loads its A/B fragments straight from row-major global memory
// (lane (r,h) reads 16 B at row r, k = 8c+4h), PF chunks ahead, in registers.
//   gemm_lab_direct M N K
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MI, int NI, int WM, int WN, int U>
__global__ __launch_bounds__(WM* WN * 64) void gemm_direct(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y, int M, int N, int K,
                                                           int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = (tile_m * WM + wm) * 32 * MI, n0 = (tile_n * WN + wn) * 32 * NI;
    const int fr = lane & 31, fh = lane >> 5;
    const float* ap[MI];
    const float* bp[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ap[i] = X + (size_t)min(m0 + i * 32 + fr, M - 1) * K + fh * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) bp[j] = W + (size_t)min(n0 + j * 32 + fr, N - 1) * K + fh * 4;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 a[U][MI], b[U][NI];
    const int nc = K / 8;
    // prologue: chunks 0..U-2
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[u][i] = *(const f32x4*)(ap[i] + u * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[u][j] = *(const f32x4*)(bp[j] + u * 8);
    }
    for (int c0 = 0; c0 < nc; c0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cl = min(c0 + u + U - 1, nc - 1);  // clamp: tail re-loads the last chunk (unused)
            constexpr int dummy = 0;
            (void)dummy;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[(u + U - 1) % U][i] = *(const f32x4*)(ap[i] + cl * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[(u + U - 1) % U][j] = *(const f32x4*)(bp[j] + cl * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][e], b[u][j][e], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M && col < N) Y[(size_t)row * N + col] = acc[i][j][r] + bv;
            }
    }
}

template <int MI, int NI, int WM, int WN, int U>
void run(const char* name, const float* x, const float* w, const float* b, float* y, int M, int N, int K, std::vector<float>& ref,
         std::vector<float>& hy) {
    const int tm = (M + 32 * MI * WM - 1) / (32 * MI * WM), tn = (N + 32 * NI * WN - 1) / (32 * NI * WN);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_direct<MI, NI, WM, WN, U>), dim3(tm * tn), dim3(WM * WN * 64), 0, 0, x, w, b, y, M, N, K, tn);
    (void)hipEventRecord(e0, 0);
    const int reps = 30;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_direct<MI, NI, WM, WN, U>), dim3(tm * tn), dim3(WM * WN * 64), 0, 0, x, w, b, y, M, N, K, tn);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < ref.size(); ++i) { maxerr = fmax(maxerr, fabs(hy[i] - ref[i])); maxref = fmax(maxref, fabs(ref[i])); }
    double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-28s M=%d N=%d K=%d grid=%d: %.2f us %.1f TF (%.1f%%)  err(first rows)=%.2e\n", name, M, N, K, tm * tn, us, tf, tf / 1.573, maxerr / maxref);
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 2048;
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N), hy((size_t)M * N);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX);
    const int RR = 8;  // reference rows
    std::vector<float> ref((size_t)RR * N);
    for (int r = 0; r < RR; ++r)
        for (int n = 0; n < N; ++n) {
            double s = hb[n];
            for (int k = 0; k < K; ++k) s += (double)hx[(size_t)r * K + k] * hw[(size_t)n * K + k];
            ref[(size_t)r * N + n] = (float)s;
        }
    float *x, *w, *b, *y;
    (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, N * 4); (void)hipMalloc(&y, hy.size() * 4);
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    run<1, 2, 2, 2, 4>("32x64 wave, 2x2 waves, U4", x, w, b, y, M, N, K, ref, hy);
    run<1, 2, 2, 2, 8>("32x64 wave, 2x2 waves, U8", x, w, b, y, M, N, K, ref, hy);
    run<2, 1, 2, 2, 4>("64x32 wave, 2x2 waves, U4", x, w, b, y, M, N, K, ref, hy);
    run<2, 2, 2, 2, 4>("64x64 wave, 2x2 waves, U4", x, w, b, y, M, N, K, ref, hy);
    run<2, 2, 2, 4, 4>("64x64 wave, 2x4 waves, U4", x, w, b, y, M, N, K, ref, hy);
    run<1, 2, 2, 4, 4>("32x64 wave, 2x4 waves, U4", x, w, b, y, M, N, K, ref, hy);
    run<1, 1, 2, 4, 8>("32x32 wave, 2x4 waves, U8", x, w, b, y, M, N, K, ref, hy);
    run<1, 2, 1, 4, 4>("32x64 wave, 1x4 waves, U4", x, w, b, y, M, N, K, ref, hy);
    return 0;
}
