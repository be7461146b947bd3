// Calibration 4 (round 3): what the fp32 k-loop's STRUCTURE can deliver on RANDOM operands, all CUs, with the in-kernel clock.
// Labs 1-3 ran on near-constant operands (the chip holds a higher clock on them: MI355X_MICROARCH.md "DVFS give-back").
//   bare    : MFMAs only, operands in registers                      (32x32x2 and 16x16x4, constant / random data)
//   ldsfed  : + fragment ds_read_b128 at the fc1 tile's ratio (5 reads per 16 MFMAs), random LDS image
//   loop    : the fc1 tile's whole k-loop (256x128x32 stages, 3-stage LDS-DMA ring, one s_barrier per k-tile, XOR-swizzled
//             chunks, fragments one chunk ahead) on random global operands; variants by template flags
// Prints TFLOP/s (HIP events, second of two launches) and the median in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long t_shader() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ unsigned long long t_real() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return (float)(int)(x & 0xffff) * (1.0f / 32768.0f) - 1.0f; }
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct Clk { unsigned long long s0, r0, s1, r1; };

// ---- bare -----------------------------------------------------------------------------------------------------------
template <int SHAPE, bool RAND>
__global__ __launch_bounds__(512) void bare(float* out, Clk* clk, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = RAND ? rnd(threadIdx.x * 16 + i + blockIdx.x * 8192) : 1.0f; b[i] = RAND ? rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 8192) : 0.5f; }
    f32x16 acc[4]; f32x4 acs[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acs[i][r] = 0.f;
    Clk c; if (threadIdx.x == 0) { c.s0 = t_shader(); c.r0 = t_real(); }
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 7], b[(u >> 1) & 7], acc[u & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) acs[u & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 7], b[(u >> 1) & 7], acs[u & 15], 0, 0, 0);
        }
    }
    if (threadIdx.x == 0) { c.s1 = t_shader(); c.r1 = t_real(); clk[blockIdx.x] = c; }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acs[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- the fc1 tile's k-loop ------------------------------------------------------------------------------------------
constexpr int BM = 256, BN = 128, BK = 32, NS = 3, STAGE = (BM + BN) * BK;
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

// DMA: 0 none (ldsfed: stages filled once), 1 production (all 6 issues behind the barrier)
// BAR: 1 barrier per k-tile, 0 none (timing only)
// PRIO: waves 4-7 at s_setprio 1
// SHAPE: 32 -> v_mfma_f32_32x32x2_f32 (wave tile 128 x 32: 4 A + 1 B fragments per 8-k chunk)
//        16 -> v_mfma_f32_16x16x4_f32 (same wave tile: 8 A + 2 B fragments per 16-k chunk)
template <int SHAPE, int DMA, int BAR, int PRIO>
__global__ __launch_bounds__(512, 2) void loop(float* out, Clk* clk, const float* X, const float* W, int K, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = (blockIdx.x >> 4) * BM, n0 = (blockIdx.x & 15) * BN;
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // DMA sources: instruction j of a wave writes rows (j*8+wave)*8 .. +8, lane l -> row +l/8, chunk position l%8
    const float* a_src[4]; const float* b_src[2];
    for (int j = 0; j < 4; ++j) { const int r = (j * 8 + wave) * 8 + lane / 8; a_src[j] = X + (size_t)(m0 + r) * K + (((lane % 8) ^ swz(r)) * 4); }
    for (int j = 0; j < 2; ++j) { const int r = (j * 8 + wave) * 8 + lane / 8; b_src[j] = W + (size_t)(n0 + r) * K + (((lane % 8) ^ swz(r)) * 4); }
    const int nk = K / BK;
    auto issue = [&](int kt, int stage) {
        float* sa = smem + stage * STAGE; float* sb = sa + BM * BK;
        const int kb = (kt % nk) * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(a_src[j] + kb, sa + (j * 8 + wave) * 8 * BK);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(b_src[j] + kb, sb + (j * 8 + wave) * 8 * BK);
    };
    if (DMA == 0) {
        for (int i = tid; i < NS * STAGE; i += 512) smem[i] = rnd(i + blockIdx.x * 77777);
    } else {
        issue(0, 0); issue(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int total = nk * reps;
    Clk c; if (tid == 0) { c.s0 = t_shader(); c.r0 = t_real(); }
    float s = 0;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int fr = lane & 31, fh = lane >> 5;
        int a_off[4], a_sw[4];
        for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + fr; a_off[i] = r * BK; a_sw[i] = swz(r); }
        const int rb = wn * 32 + fr, b_off = BM * BK + rb * BK, b_sw = swz(rb);
        f32x4 af[2][4], bf[2];
        auto rd = [&](int stage, int kc, f32x4 (&a)[4], f32x4& b) {
            const float* st = smem + stage * STAGE; const int ch = kc * 2 + fh;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + ((ch ^ a_sw[i]) << 2));
            b = *reinterpret_cast<const f32x4*>(st + b_off + ((ch ^ b_sw) << 2));
        };
        rd(0, 0, af[0], bf[0]);
        int stage = 0;
        for (int kt = 0; kt < total; ++kt) {
            const int stage1 = stage + 1 == NS ? 0 : stage + 1;
            const int stage_new = stage + 2 >= NS ? stage + 2 - NS : stage + 2;
            if (kt > 0) {
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (BAR) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (DMA == 1) issue(kt + 2, stage_new);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                if (kc + 1 < 4) rd(stage, kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
                else rd(stage1, 0, af[0], bf[0]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][i][e], bf[kc & 1][e], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage = stage1;
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8][2];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int fr = lane & 15, fq = lane >> 4;
        int a_off[8], a_sw[8], b_off[2], b_sw[2];
        for (int i = 0; i < 8; ++i) { const int r = wm * 128 + i * 16 + fr; a_off[i] = r * BK; a_sw[i] = swz(r); }
        for (int j = 0; j < 2; ++j) { const int r = wn * 32 + j * 16 + fr; b_off[j] = BM * BK + r * BK; b_sw[j] = swz(r); }
        f32x4 af[2][8], bf[2][2];
        auto rd = [&](int stage, int kc, f32x4 (&a)[8], f32x4 (&b)[2]) {   // kc: 16-k chunk 0..1
            const float* st = smem + stage * STAGE; const int ch = kc * 4 + fq;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + ((ch ^ a_sw[i]) << 2));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + ((ch ^ b_sw[j]) << 2));
        };
        rd(0, 0, af[0], bf[0]);
        int stage = 0;
        for (int kt = 0; kt < total; ++kt) {
            const int stage1 = stage + 1 == NS ? 0 : stage + 1;
            const int stage_new = stage + 2 >= NS ? stage + 2 - NS : stage + 2;
            if (kt > 0) {
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (BAR) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (DMA == 1) issue(kt + 2, stage_new);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                if (kc + 1 < 2) rd(stage, kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
                else rd(stage1, 0, af[0], bf[0]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kc & 1][i][e], bf[kc & 1][j][e], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage = stage1;
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (tid == 0) { c.s1 = t_shader(); c.r1 = t_real(); clk[blockIdx.x] = c; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = s;
}


// ---- whole tiles back to back: prologue + k-loop + epilogue (accumulators -> LDS -> rows -> bias + tanh-GELU -> 16-byte stores) ----
// One workgroup runs `reps` tiles of BM x BN x K one after the other (what a persistent phase loop does, minus the hand-off).
// WGS workgroups of NWM * NWN waves per CU: <256,128,32,3, 2,4> x 1 = the production fc1 tile; <128,128,16,5, 2,2> x 2 = two
// half-size, independent workgroups per CU whose epilogues may run under each other's k-loops.
template <int BK_> __device__ __forceinline__ int swzk(int r) { return BK_ == 32 ? (r >> 1) & 7 : (r >> 2) & 3; }
__device__ __forceinline__ float gelu_tanh(float x) {
    const float x3 = x * x * x, inner = 0.7978845608028654f * (x + 0.044715f * x3);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(inner * -2.8853900817779268f));
}
template <int TBM, int TBN, int TBK, int TNS, int NWM, int NWN, int OCC, int STAG>
__global__ __launch_bounds__(NWM * NWN * 64, OCC) void tilek(float* Y, Clk* clk, const float* X, const float* W, const float* bias, int K, int reps, int mtiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NWM * NWN * 64, NW = NWM * NWN;
    constexpr int CH = TBK / 4, RPI = 64 / CH, A_DMA = TBM / RPI / NW, B_DMA = TBN / RPI / NW, DPT = A_DMA + B_DMA;
    constexpr int TM = TBM / NWM, TN = TBN / NWN, MI = TM / 32, NI = TN / 32, NCH = TBK / 8, TSTAGE = (TBM + TBN) * TBK, S = TBN + 4;
    static_assert(TBM * S <= TNS * TSTAGE, "epilogue buffer");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    if (STAG && (__builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6) != 0))  /* HW_REG_LDS_ALLOC.LDS_BASE != 0: the CU's second workgroup */ { for (int i = 0; i < STAG; ++i) __builtin_amdgcn_s_sleep(127); }
    const int nk = K / TBK;
    const int fr = lane & 31, fh = lane >> 5;
    Clk c; if (tid == 0) { c.s0 = t_shader(); c.r0 = t_real(); }
    for (int rep = 0; rep < reps; ++rep) {
        const int t = (blockIdx.x + rep * 7) % (mtiles * (2048 / TBN));
        const int m0 = (t / (2048 / TBN)) * TBM, n0 = (t % (2048 / TBN)) * TBN;
        const float* a_src[A_DMA]; const float* b_src[B_DMA];
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) { const int r = (j * NW + wave) * RPI + lane / CH; a_src[j] = X + (size_t)(m0 + r) * K + (((lane % CH) ^ swzk<TBK>(r)) * 4); }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) { const int r = (j * NW + wave) * RPI + lane / CH; b_src[j] = W + (size_t)(n0 + r) * K + (((lane % CH) ^ swzk<TBK>(r)) * 4); }
        auto issue = [&](int kt, int stage) {
            float* sa = smem + stage * TSTAGE; float* sb = sa + TBM * TBK;
            const int kb = (kt < nk ? kt : 0) * TBK;
#pragma unroll
            for (int j = 0; j < A_DMA; ++j) dma16(a_src[j] + kb, sa + (j * NW + wave) * RPI * TBK);
#pragma unroll
            for (int j = 0; j < B_DMA; ++j) dma16(b_src[j] + kb, sb + (j * NW + wave) * RPI * TBK);
        };
#pragma unroll
        for (int s_ = 0; s_ < TNS - 1; ++s_) issue(s_, s_);
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int a_off[MI], a_sw[MI], b_off[NI], b_sw[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) { const int r = wm * TM + i * 32 + fr; a_off[i] = r * TBK; a_sw[i] = swzk<TBK>(r); }
#pragma unroll
        for (int j = 0; j < NI; ++j) { const int r = wn * TN + j * 32 + fr; b_off[j] = TBM * TBK + r * TBK; b_sw[j] = swzk<TBK>(r); }
        f32x4 af[2][MI], bf[2][NI];
        auto rd = [&](int stage, int kc, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
            const float* st = smem + stage * TSTAGE; const int ch = kc * 2 + fh;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + ((ch ^ a_sw[i]) << 2));
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + ((ch ^ b_sw[j]) << 2));
        };
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TNS - 2) * DPT) : "memory");
        __syncthreads();
        rd(0, 0, af[0], bf[0]);
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int stage1 = stage + 1 == TNS ? 0 : stage + 1;
            int stage_new = stage + TNS - 1; stage_new = stage_new >= TNS ? stage_new - TNS : stage_new;
            if (kt > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TNS - 3) * DPT) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            issue(kt + TNS - 1, stage_new);
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) {
                if (kc + 1 < NCH) rd(stage, kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
                else rd(stage1, 0, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][i][e], bf[kc & 1][j][e], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage = stage1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) smem[(wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * S + wn * TN + j * 32 + fr] = acc[i][j][r];
        __syncthreads();
        constexpr int C4 = TBN / 4, RSTEP = NT / C4, ITERS = TBM / RSTEP;
        const int lc = (tid % C4) * 4, lr0 = tid / C4;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n0 + lc);
#pragma unroll 4
        for (int it = 0; it < ITERS; ++it) {
            const int lr = lr0 + it * RSTEP;
            f32x4 a = *reinterpret_cast<const f32x4*>(smem + lr * S + lc);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = gelu_tanh(a[k] + b4[k]);
            *reinterpret_cast<f32x4*>(Y + (size_t)(m0 + lr) * 2048 + n0 + lc) = a;
        }
        __syncthreads();
    }
    if (tid == 0) { c.s1 = t_shader(); c.r1 = t_real(); clk[blockIdx.x] = c; }
}

static double g_last_start_us, g_span_us, g_first_end_us;
static double median_clock(Clk* d_clk, int grid) {
    std::vector<Clk> h(grid);
    (void)hipMemcpy(h.data(), d_clk, grid * sizeof(Clk), hipMemcpyDeviceToHost);
    std::vector<double> g;
    for (auto& c : h) g.push_back((double)(c.s1 - c.s0) / (double)(c.r1 - c.r0) * 0.1);
    std::sort(g.begin(), g.end());
    unsigned long long r0max = 0, r1min = ~0ull, r0min = ~0ull, r1max = 0;
    for (auto& c : h) { r0max = std::max(r0max, c.r0); r1min = std::min(r1min, c.r1); r0min = std::min(r0min, c.r0); r1max = std::max(r1max, c.r1); }
    g_last_start_us = (double)(r0max - r0min) * 0.01; g_span_us = (double)(r1max - r0min) * 0.01; g_first_end_us = (double)(r1min - r0min) * 0.01;
    return g[g.size() / 2];
}

static float g_warm_ms = 2500.f;
template <class F> static void timed(const char* name, double flop, Clk* clk, int grid, F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // >= 2 s of back-to-back launches first: the chip's clock settles over seconds (MI355X_MICROARCH.md, DVFS give-back (6))
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int warm = (int)(g_warm_ms / ms) + 1;
    for (int w = 0; w < warm; ++w) launch();
    (void)hipEventRecord(e0, 0);
    for (int w = 0; w < 10; ++w) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); ms *= 0.1f;
    const double tf = flop / (ms * 1e-3) / 1e12, ghz = median_clock(clk, grid);
    // cycles per 32x32x2-equivalent MFMA (4096 FLOP) per SIMD
    const double cyc = (ms * 1e-3 * ghz * 1e9) / (flop / 4096.0 / 1024.0);
    printf("{\"lab\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f, \"frac\": %.3f, \"ghz\": %.3f, \"cyc_per_mfma_simd\": %.1f, \"last_block_start_us\": %.1f, \"first_block_end_us\": %.1f, \"span_us\": %.1f}\n", name, ms, tf, tf / 157.3, ghz, cyc, g_last_start_us, g_first_end_us, g_span_us);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) g_warm_ms = (float)atof(argv[1]);
    const int grid = 256, K = 512, reps = 400;
    float *out, *X, *W; Clk* clk;
    (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&clk, 4096 * sizeof(Clk));
    (void)hipMalloc(&X, (size_t)4096 * K * 4); (void)hipMalloc(&W, (size_t)2048 * K * 4);
    {
        std::vector<float> h((size_t)4096 * K);
        srand(1);
        for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        (void)hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < (size_t)2048 * K; ++i) h[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        (void)hipMemcpy(W, h.data(), (size_t)2048 * K * 4, hipMemcpyHostToDevice);
    }
    const bool only_tiles = argc > 2;
    const int bi = 40000;
    if (!only_tiles) {
    const double bflop = (double)grid * 8 * bi * 16 * 4096.0;
    timed("bare 32x32x2 const", bflop, clk, grid, [&] { hipLaunchKernelGGL((bare<32, false>), dim3(grid), dim3(512), 0, 0, out, clk, bi); });
    timed("bare 32x32x2 random", bflop, clk, grid, [&] { hipLaunchKernelGGL((bare<32, true>), dim3(grid), dim3(512), 0, 0, out, clk, bi); });
    timed("bare 16x16x4 const", bflop, clk, grid, [&] { hipLaunchKernelGGL((bare<16, false>), dim3(grid), dim3(512), 0, 0, out, clk, bi); });
    timed("bare 16x16x4 random", bflop, clk, grid, [&] { hipLaunchKernelGGL((bare<16, true>), dim3(grid), dim3(512), 0, 0, out, clk, bi); });
    const size_t lds = (size_t)NS * STAGE * 4;
    const double lflop = 2.0 * 4096 * 2048 * K * reps;
#define LOOP(S, D, B, P, name) \
    (void)hipFuncSetAttribute((const void*)loop<S, D, B, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    timed(name, lflop, clk, grid, [&] { hipLaunchKernelGGL((loop<S, D, B, P>), dim3(grid), dim3(512), lds, 0, out, clk, X, W, K, reps); });
    LOOP(32, 0, 0, 0, "ldsfed 32x32x2 (no dma, no barrier)");
    LOOP(16, 0, 0, 0, "ldsfed 16x16x4 (no dma, no barrier)");
    LOOP(32, 0, 1, 0, "ldsfed 32x32x2 + barrier");
    LOOP(32, 1, 1, 0, "loop 32x32x2 production structure");
    LOOP(16, 1, 1, 0, "loop 16x16x4 production structure");
    LOOP(32, 1, 0, 0, "loop 32x32x2 no barrier (timing only)");
    LOOP(32, 1, 1, 1, "loop 32x32x2 waves 4-7 prio 1");
    LOOP(16, 1, 1, 1, "loop 16x16x4 waves 4-7 prio 1");
    }

    {
        float *Y, *bias; (void)hipMalloc(&Y, (size_t)4096 * 2048 * 4); (void)hipMalloc(&bias, 2048 * 4); (void)hipMemset(bias, 0, 2048 * 4);
        const int treps = 240;
        const double tflop = 256.0 * treps * 2.0 * 256 * 128 * K;
#define TILE(BM_, BN_, BK_, NS_, WM_, WN_, OCC_, WGS_, STAG_, name) { \
        const size_t l = (size_t)NS_ * (BM_ + BN_) * BK_ * 4; \
        (void)hipFuncSetAttribute((const void*)tilek<BM_, BN_, BK_, NS_, WM_, WN_, OCC_, STAG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
        timed(name, tflop, clk, 256 * WGS_, [&] { hipLaunchKernelGGL((tilek<BM_, BN_, BK_, NS_, WM_, WN_, OCC_, STAG_>), dim3(256 * WGS_), dim3(WM_ * WN_ * 64), l, 0, Y, clk, X, W, bias, K, treps * (256 * 128) / (BM_ * BN_) / WGS_, 4096 / BM_); }); }
        TILE(256, 128, 32, 3, 2, 4, 2, 1, 0, "tiles 256x128 (fc1 tile), 1 workgroup of 8 waves per CU");
        TILE(128, 128, 16, 5, 2, 2, 2, 2, 0, "tiles 128x128, 2 workgroups of 4 waves per CU");
        TILE(128, 128, 16, 5, 2, 2, 2, 2, 2, "tiles 128x128, 2 workgroups of 4 waves per CU, the CU's second one starts 2 x s_sleep 127 late");
        TILE(128, 128, 16, 5, 2, 2, 2, 2, 4, "tiles 128x128, 2 workgroups of 4 waves per CU, the CU's second one starts 4 x s_sleep 127 late");
        TILE(128, 128, 16, 5, 2, 2, 2, 2, 8, "tiles 128x128, 2 workgroups of 4 waves per CU, the CU's second one starts 8 x s_sleep 127 late");
        TILE(128, 128, 16, 5, 2, 2, 2, 1, 0, "tiles 128x128, 1 workgroup of 4 waves per CU");
        TILE(128, 128, 32, 4, 2, 4, 2, 1, 0, "tiles 128x128 BK=32 NS=4, 1 workgroup of 8 waves per CU");
    }
    return 0;
}
