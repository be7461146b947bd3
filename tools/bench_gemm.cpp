// Standalone micro-benchmark of vita_linear_f32 through the C ABI (no Python), for rocprofv3
// counter passes and quick A/B of kernel variants:  bench_gemm M N K ln act gated reps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/vita_hip.h"
#ifdef VITA_STAMP
extern "C" int vita_lab_read_stamps(unsigned long long*, int);
#endif

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    int ln = argc > 4 ? atoi(argv[4]) : 0, act = argc > 5 ? atoi(argv[5]) : 0, gated = argc > 6 ? atoi(argv[6]) : 0;
    int reps = argc > 7 ? atoi(argv[7]) : 50;
    int padx = argc > 8 ? atoi(argv[8]) : 0, padw = argc > 9 ? atoi(argv[9]) : 0;
    const size_t ldx = K + padx, ldw = K + padw;
    std::vector<float> hx((size_t)M * ldx), hw((size_t)N * ldw), hv(N > K ? N : K);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hv) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
    float *x, *w, *b, *y, *pm, *pa, *g;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMalloc(&y, (size_t)M * N * 4)); CK(hipMalloc(&pm, K * 4)); CK(hipMalloc(&pa, K * 4)); CK(hipMalloc(&g, N * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hv.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pm, hv.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pa, hv.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hv.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0, (size_t)M * N * 4));
    vita_linear_args a = {};
    a.X = x; a.ldx = ldx; a.W = w; a.ldw = ldw; a.bias = b; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.act = act;
    if (ln) { a.pro = VITA_PRO_LN_MOD; a.ln_eps = 1e-6f; a.pro_mul = pm; a.pro_add = pa; }
    if (gated) { a.gate = g; a.resid = y; a.ld_resid = N; }
    if (getenv("SPLITK") && atoi(getenv("SPLITK"))) {   // hand the library its split-K scratch, as the orchestrators do
        void* ws; CK(hipMalloc(&ws, vita_splitk_workspace_bytes()));
        a.splitk_ws = ws; a.splitk_ws_bytes = vita_splitk_workspace_bytes();
    }
    for (int i = 0; i < 5; ++i) if (vita_linear_f32(&a, 0) != 0) { printf("err: %s\n", vita_last_error()); return 1; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) vita_linear_f32(&a, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
#ifdef VITA_STAMP
    {
        int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        if (tiles < 512) tiles = ((M + 63) / 64) * ((N + 127) / 128);
        if (getenv("TILES")) tiles = atoi(getenv("TILES"));
        if (tiles > 4096) tiles = 4096;
        std::vector<unsigned long long> st(tiles * 8);
        CK(hipDeviceSynchronize());
        vita_lab_read_stamps(st.data(), tiles * 8);
        double pro = 0, loop = 0, epi = 0, clk = 0, e_wr = 0, e_bar = 0, e_row = 0; unsigned long long t0 = ~0ull, t3 = 0, r0 = ~0ull, r1 = 0;
        for (int i = 0; i < tiles; ++i) {
            e_wr += st[i * 8 + 6] - st[i * 8 + 2]; e_bar += st[i * 8 + 7] - st[i * 8 + 6]; e_row += st[i * 8 + 3] - st[i * 8 + 7];
            pro += st[i * 8 + 1] - st[i * 8]; loop += st[i * 8 + 2] - st[i * 8 + 1]; epi += st[i * 8 + 3] - st[i * 8 + 2];
            if (st[i * 8] < t0) t0 = st[i * 8]; if (st[i * 8 + 3] > t3) t3 = st[i * 8 + 3];
            clk += (double)(st[i * 8 + 3] - st[i * 8]) / (double)(st[i * 8 + 5] - st[i * 8 + 4]) * 0.1;   // GHz
            if (st[i * 8 + 4] < r0) r0 = st[i * 8 + 4]; if (st[i * 8 + 5] > r1) r1 = st[i * 8 + 5];
        }
        printf("epilogue split: drain+barrier+acc->LDS %.0f | barrier %.0f | row phase %.0f\n", e_wr / tiles, e_bar / tiles, e_row / tiles);
        printf("stamps (shader cycles) tiles=%d prologue=%.0f loop=%.0f epilogue=%.0f | in-kernel clock %.3f GHz | last launch first-WG-start..last-WG-end %.2f us\n",
               tiles, pro / tiles, loop / tiles, epi / tiles, clk / tiles, (r1 - r0) * 0.01);
    }
#endif
    printf("padx=%d padw=%d ", padx, padw);
    printf("M=%d N=%d K=%d ln=%d act=%d gated=%d : %.2f us  %.2f TFLOP/s  %.1f%% of 157.3\n", M, N, K, ln, act, gated, us, tf, tf / 157.3 * 100);
    return 0;
}
