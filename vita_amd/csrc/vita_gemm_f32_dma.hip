// fp32 MFMA GEMM, LDS-DMA variant — the fast path of vita_linear_f32 for 16-byte-aligned
// operands (every GEMM of the sampler except obs_encoder's K=533).
//
//   Y[M,N] = resid + gate * act( pro(X)[M,K] · W[N,K]^T + bias )
//
// Why a second kernel: staging fp32 tiles through VGPRs costs a ds_write_b128 per 1 KiB
// (13 LDS cycles each) on the same LDS port the fragment reads use; measured, those writes
// and their vmcnt waits took ~20 % of the k-loop.  Here both operand tiles go global -> LDS
// directly (global_load_lds_dwordx4, no VGPRs, no ds_write), NS stages deep:
//
//   stage image : [rows][BK floats], UNPADDED (an LDS-DMA wave-instruction writes 1 KiB
//                 linearly: 8 rows x 128 B at BK=32, 4 rows x 256 B at BK=64), with the 16-byte
//                 chunks of row r XOR-swizzled (chunk c stored at position c ^ f(r)) so the
//                 ds_read_b128 fragment reads (32 rows x 16 B per half-wave) hit 16 distinct
//                 16-byte bank slots.  The swizzle is applied on the per-lane GLOBAL source
//                 address (the LDS side of a DMA is lane-linear) and again on the read.
//   k-loop      : per tile ONE counted s_waitcnt vmcnt + ONE raw s_barrier; the DMAs of the
//                 next NS-2 tiles stay in flight across it; fragment reads run one 8-deep chunk
//                 ahead of the MFMAs, including across the barrier (tile kt+1 is complete one
//                 iteration early).
//   LayerNorm   : cannot be applied while staging (the data never passes through registers),
//                 so it is applied to the A fragments after the LDS read: (a-mean)*rstd from
//                 per-lane row statistics, then *mul[k] + add[k] from LDS tables — VALU work
//                 that hides under the 64-cycle fp32 MFMAs.
#include <stdio.h>
#include <type_traits>

#include "vita_common.h"

namespace vita {
#ifdef VITA_STAMP
// Diagnostic build only (tools/lab_variant.sh): per-workgroup s_memtime stamps, read back by vita_lab_read_stamps.
__device__ unsigned long long g_stamps[4096 * 8];
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t__;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");               \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (i)] = t__;          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#define RSTAMP(i)                                                                                 \
    do {                                                                                          \
        unsigned long long t__;                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");           \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (i)] = t__;          \
    } while (0)
#else
#define STAMP(i)
#define RSTAMP(i)
#endif

}  // namespace vita
#define VITA_EPI_STAMP(i) STAMP(i)

#include "vita_gemm_epilogue.h"
#include "vita_host.h"

namespace vita {

struct GemmParams;  // defined in vita_gemm_f32.hip; the layout is repeated here

struct GemmParamsDma {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;
    float* Y; int64_t ldy;
    int M, N, K;
    int act, pro;
    float ln_eps;
    const float* pro_mul;
    const float* pro_add;
    const float* gate; int64_t ld_gate;
    int use_gate_scalar; float gate_scalar;
    const float* resid; int64_t ld_resid;
    float* preact; int64_t ld_preact;
    int tiles_m, tiles_n;
    int sk_splits, sk_k;   // split-K: number of splits, k elements per split
    int xcd_nbr, xcd_nbc;  // the 8 XCDs as an nbr x nbc grid of tile blocks (0: linear ranges of tile ids)
    EpiParams epi;
};

constexpr int DNT = 512;

__device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// chunk swizzle: BK=32 -> 8 chunks/row, rows alternate halves of the 256-B bank row, f = (r>>1)&7
//                BK=64 -> 16 chunks/row, one row per bank row,                        f = r&15
template <int BK>
__device__ __forceinline__ int swz(int r) {
    if constexpr (BK == 32) return (r >> 1) & 7;
    else return r & 15;
}

// OCC = waves per SIMD the register allocation must allow (2 = one 512-thread workgroup per CU, 4 = two)
// BF16: the operand bytes are bf16 (p.K / p.ldx / p.ldw count 4-byte units = bf16 pairs, BK 4-byte units = 2*BK bf16 of
// reduction): same DMA / LDS image, and the 16-byte fragment a lane reads IS the 8-deep bf16 operand of one
// v_mfma_f32_32x32x16_bf16 (lanes 0-31: k 0-7, lanes 32-63: k 8-15), so one MFMA replaces the four fp32 ones.
// KG: in-workgroup split-K.  The 8 waves form KG groups of 8/KG; every group covers the WHOLE output tile (so a wave owns
// KG times more of it: 64x64 instead of 32x32 at 64x128 / KG=4 -> half the LDS fragment reads per MFMA, which is what
// bounds the small tile) and reduces 1/KG of each k-tile's chunks; the epilogue sums the groups through LDS.
// BWD: the epilogue may apply act'(act_grad_z) (the backward GEMMs of the bf16-operand training step).
template <int BM, int BN, int BK, int NS, bool LN, int OCC, bool BF16 = false, int KG = 1, bool BWD = false>
__global__ __launch_bounds__(DNT, OCC) void gemm_f32_dma_kernel(GemmParamsDma p) {
    constexpr int CH = BK / 4;                 // 16-byte chunks per tile row
    constexpr int RPI = 64 / CH;               // rows covered by one DMA wave-instruction
    constexpr int A_DMA = BM / RPI / 8;        // DMA instructions per wave per tile (A)
    constexpr int B_DMA = BN / RPI / 8;
    constexpr int DPT = A_DMA + B_DMA;
    constexpr int WGN = KG == 1 ? 4 : 2, WGM = 8 / KG / WGN;   // waves of one k-group along n / m
    constexpr int TM = BM / WGM, TN = BN / WGN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int NCH = BK / 8 / KG;           // 8-deep k chunks of a k-tile that one wave reduces
    constexpr int STAGE = (BM + BN) * BK;      // floats per stage
    static_assert(A_DMA >= 1 && B_DMA >= 1 && MI >= 1 && NI >= 1 && NCH >= 2 && NCH % 2 == 0 && NS >= 3 && WGM >= 1, "bad tile");
    static_assert(KG * BM * (BN + 4) <= NS * STAGE, "the output tile(s) of the epilogue must fit the stage buffers");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [NS][ A: BM x BK | B: BN x BK ]  then (LN only) mul[K], add[K], stats[BM][2]
    float* lnm = smem + NS * STAGE;
    float* lna = lnm + (LN ? p.K : 0);
    float* stat = lna + (LN ? p.K : 0);
    float* sfold = stat + (LN ? 2 * BM : 0);   // folded LayerNorm (consumer): (mean, rstd) of the tile's rows, 2*BM floats

#ifdef VITA_STAGGER   // lab: delay half of the workgroups so co-resident pairs are out of phase
    if ((blockIdx.x >> VITA_STAGGER_SHIFT) & 1) {
#pragma unroll 1
        for (int i = 0; i < VITA_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    RSTAMP(4);
    STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WGM * WGN);
    const int wm = (wave % (WGM * WGN)) / WGN, wn = wave % WGN;

    const int nwg = p.tiles_m * p.tiles_n;
    const int split = p.sk_splits > 1 ? (int)blockIdx.x / nwg : 0;
    int bid = blockIdx.x - split * nwg;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    if (p.xcd_nbr > 0) {
        // workgroup ids go round-robin over the XCDs: XCD x = id & 7 owns block (x / nbc, x % nbc) of the tile grid, so the
        // rows of X and W its L2 has to fetch are the block's, not a full strip's (fc1: 42 -> 33.5 MB fetched per launch)
        const int x = (int)(blockIdx.x - split * nwg) & 7, l = (int)(blockIdx.x - split * nwg) >> 3;
        const int br = p.tiles_m / p.xcd_nbr, bc = p.tiles_n / p.xcd_nbc;
        tile_m = (x / p.xcd_nbc) * br + l / bc;
        tile_n = (x % p.xcd_nbc) * bc + l % bc;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbase = split * p.sk_k;                                   // this workgroup reduces k in [kbase, K)
    const int K = p.sk_splits > 1 ? min(p.K, kbase + p.sk_k) : p.K;
    const int nk = (K - kbase + BK - 1) / BK;

    // ---- per-lane DMA sources: lane l of instruction j writes row (j*8+wave)*RPI + l/CH,
    //      chunk position l%CH, so it must fetch chunk (l%CH) ^ swz(row) of that row
    const float* a_src[A_DMA];
    int a_k[A_DMA];
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
        const int r = (j * 8 + wave) * RPI + lane / CH;
        a_k[j] = ((lane % CH) ^ swz<BK>(r)) * 4;
        a_src[j] = p.X + (int64_t)min(m0 + r, p.M - 1) * p.ldx + a_k[j];
    }
    const float* b_src[B_DMA];
    int b_k[B_DMA];
#pragma unroll
    for (int j = 0; j < B_DMA; ++j) {
        const int r = (j * 8 + wave) * RPI + lane / CH;
        b_k[j] = ((lane % CH) ^ swz<BK>(r)) * 4;
        b_src[j] = p.W + (int64_t)min(n0 + r, p.N - 1) * p.ldw + b_k[j];
    }
    auto issue_tile = [&](int kt, int stage) {
        float* sa = smem + stage * STAGE;
        float* sb = sa + BM * BK;
        const int kb = kbase + kt * BK;
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const float* g = (kb + a_k[j] < K) ? a_src[j] + kb : g_zero_chunk;
            dma16(g, sa + (j * 8 + wave) * RPI * BK);
        }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) {
            const float* g = (kb + b_k[j] < K) ? b_src[j] + kb : g_zero_chunk;
            dma16(g, sb + (j * 8 + wave) * RPI * BK);
        }
    };

    // prologue DMAs first: they fly while the LayerNorm statistics are computed
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_tile(s, s);

    // folded LayerNorm, consumer side: combine the producer's partial row statistics while the first DMAs fly; the barrier
    // in front of the k-loop publishes them for the epilogue
    if (p.epi.ln_stats != nullptr && tid < BM && m0 + tid < p.M) fold_row_stats(p.epi, m0 + tid, 128, sfold + tid * 2);

    // ---- LayerNorm: tables into LDS, row statistics (two-pass) ---------------------------
    float mean[MI], rstd[MI];
    if constexpr (LN) {
        const bool mod = p.pro == VITA_PRO_LN_MOD;
        for (int k = tid * 4; k < K; k += DNT * 4) {
            f32x4 m = *reinterpret_cast<const f32x4*>(p.pro_mul + k);
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.pro_add + k);
            if (mod) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = m[j] + 1.0f;
            }
            *reinterpret_cast<f32x4*>(lnm + k) = m;
            *reinterpret_cast<f32x4*>(lna + k) = a;
        }
        // wave w owns rows w*BM/8 ..; a row is read 64 lanes x 16 B at a time
        constexpr int RW = BM / 8;
        constexpr int RB = 8;  // rows in flight
        const int nseg = (K + 255) / 256;
        for (int rb = 0; rb < RW; rb += RB) {
            float s[RB];
            f32x4 v0[RB], v1[RB];  // first two segments stay in registers (K <= 512: single read)
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float* row = p.X + (int64_t)min(m0 + wave * RW + rb + u, p.M - 1) * p.ldx;
                const int k0 = lane * 4, k1 = 256 + lane * 4;
                v0[u] = *reinterpret_cast<const f32x4*>(k0 < K ? row + k0 : g_zero_chunk);
                v1[u] = *reinterpret_cast<const f32x4*>(k1 < K ? row + k1 : g_zero_chunk);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                s[u] = ((v0[u][0] + v0[u][1]) + (v0[u][2] + v0[u][3])) + ((v1[u][0] + v1[u][1]) + (v1[u][2] + v1[u][3]));
                const float* row = p.X + (int64_t)min(m0 + wave * RW + rb + u, p.M - 1) * p.ldx;
                for (int sg = 2; sg < nseg; ++sg) {
                    const int k = sg * 256 + lane * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(k < K ? row + k : g_zero_chunk);
                    s[u] += (v[0] + v[1]) + (v[2] + v[3]);
                }
                s[u] = wave_sum(s[u]);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float mu = s[u] / (float)K;
                float q = 0.f;
                const int k0 = lane * 4, k1 = 256 + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d0 = (k0 + j < K) ? v0[u][j] - mu : 0.f;
                    const float d1 = (k1 + j < K) ? v1[u][j] - mu : 0.f;
                    q = fmaf(d0, d0, q);
                    q = fmaf(d1, d1, q);
                }
                const float* row = p.X + (int64_t)min(m0 + wave * RW + rb + u, p.M - 1) * p.ldx;
                for (int sg = 2; sg < nseg; ++sg) {
                    const int k = sg * 256 + lane * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(k < K ? row + k : g_zero_chunk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = (k + j < K) ? v[j] - mu : 0.f;
                        q = fmaf(d, d, q);
                    }
                }
                q = wave_sum(q);
                if (lane == 0) {
                    stat[(wave * RW + rb + u) * 2 + 0] = mu;
                    stat[(wave * RW + rb + u) * 2 + 1] = 1.0f / sqrtf(q / (float)K + p.ln_eps);
                }
            }
        }
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addressing ---------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[MI], a_sw[MI], b_off[NI], b_sw[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * TM + i * 32 + fr;
        a_off[i] = r * BK;
        a_sw[i] = swz<BK>(r);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = wn * TN + j * 32 + fr;
        b_off[j] = BM * BK + r * BK;
        b_sw[j] = swz<BK>(r);
    }
    f32x4 af[2][MI], bf[2][NI], lm[2], la[2];
    auto read_frags = [&](int stage, int kt, int kc, f32x4 (&a)[MI], f32x4 (&b)[NI], f32x4& m, f32x4& ad) {
        const float* st = smem + stage * STAGE;
        const int c = (kg * NCH + kc) * 2 + fh;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + ((c ^ a_sw[i]) << 2));
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + ((c ^ b_sw[j]) << 2));
        if constexpr (LN) {
            int k = kt * BK + c * 4;
            k = k < K ? k : 0;  // beyond K the W chunk is zero: any finite value will do
            m = *reinterpret_cast<const f32x4*>(lnm + k);
            ad = *reinterpret_cast<const f32x4*>(lna + k);
        }
    };
    auto ln_apply = [&](f32x4 (&a)[MI], const f32x4& m, const f32x4& ad) {
        if constexpr (LN) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = (a[i][e] - mean[i]) * rstd[i];
                    a[i][e] = n * m[e] + ad[e];
                }
        }
    };

    // tile 0 (and 1) landed for everyone; LN tables / stats visible
    wait_vmcnt<(NS - 3) * DPT>();
    __syncthreads();
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            mean[i] = stat[(wm * TM + i * 32 + fr) * 2 + 0];
            rstd[i] = stat[(wm * TM + i * 32 + fr) * 2 + 1];
        }
    }
    read_frags(0, 0, 0, af[0], bf[0], lm[0], la[0]);
    STAMP(1);

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage1 = stage + 1 == NS ? 0 : stage + 1;
        int stage_new = stage + NS - 1;
        stage_new = stage_new >= NS ? stage_new - NS : stage_new;
        if (kt > 0) {
            // this wave's share of tile kt+1 has landed ...
            wait_vmcnt<(NS - 3) * DPT>();
            // ... and after the barrier everyone's has, and everyone is done reading tile kt-1
#ifndef VITA_LAB_NOBAR     // timing-only lab builds: results are wrong without the barrier / the refills
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        }
        // all of the tile's DMAs right behind the barrier: spreading them between the MFMA groups of the k-chunks was
        // measured 7 % slower (fc1 86 -> 93.6 us, fc2 85 -> 91.4 us); letting only one wave per SIMD issue them, or BK=32
        // with 5-6 stages, changed nothing.  Lab decomposition (timing only): without the barrier the kernel takes the
        // same time; without the refills fc1 73.8 us / fc2 68.2 us at 2.22 GHz instead of 87.5 / 85.0 us at 2.03 GHz, i.e.
        // the operand traffic costs both stall cycles (fc1 +4.6 %, fc2 +12.5 % over the MFMA-only loop) and ~9 % of clock
#ifndef VITA_LAB_NODMA
        issue_tile(kt + NS - 1, stage_new);   // refills the stage tile kt-1 lived in (past the last tile: dummy reads of a
                                              // zero chunk; branching around them costs more than they do: fc2 loop +4 %)
#endif
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            if (kc + 1 < NCH) read_frags(stage, kt, kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1], lm[(kc + 1) & 1], la[(kc + 1) & 1]);
            else read_frags(stage1, kt + 1, 0, af[0], bf[0], lm[0], la[0]);
            ln_apply(af[kc & 1], lm[kc & 1], la[kc & 1]);
            if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kc & 1][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[kc & 1][j]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][i][e], bf[kc & 1][j][e], acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the prefetched fragments must be in registers before the next barrier lets anyone
        // overwrite... (they belong to tile kt+1, which is not overwritten until kt+2) — only
        // the reads of tile kt must have retired, and they fed the MFMAs above.
        stage = stage1;
    }
    STAMP(2);
    wait_vmcnt<0>();  // drain the dummy tail DMAs before the wave exits / LDS is released

    // ---- epilogue: through the (now dead) stage buffers, see vita_gemm_epilogue.h ------------
    __syncthreads();   // every wave is done reading fragments, every DMA has landed
    EpiParams e = p.epi;
    if (e.partial != nullptr) e.partial += (int64_t)split * p.M * e.ld_partial;
    epilogue_dispatch<BM, BN, MI, NI, DNT, BWD, KG>(acc, smem, e, m0, n0, wm, wn, lane, tid, kg, sfold);
    STAMP(3);
    RSTAMP(5);
}

// what the last LDS-DMA launch of this host thread dispatched (bench.py labels its roofline kernel from it)
static thread_local char g_last_cfg[96] = "";
const char* last_dma_config() { return g_last_cfg; }

template <int BM, int BN, int BK, int NS>
static int launch_dma_bf16(const GemmParamsDma& p0, hipStream_t stream) {
    GemmParamsDma p = p0;
    snprintf(g_last_cfg, sizeof(g_last_cfg), "gemm_f32_dma_kernel<%d,%d,%d,%d,bf16>", BM, BN, BK, NS);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = (size_t)NS * (BM + BN) * BK * sizeof(float);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (p.sk_splits > 1 ? p.sk_splits : 1)));
    if (p.epi.act_grad_z != nullptr)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, 2, true, 1, true>), grid, dim3(DNT), lds, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, 2, true>), grid, dim3(DNT), lds, stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

template <int BM, int BN, int BK, int NS, int OCC = 2, int KG = 1>
static int launch_dma(const GemmParamsDma& p0, hipStream_t stream) {
    GemmParamsDma p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    // XCD grid: minimise the operand bytes the eight L2s fetch, M * nbc + N * nbr, over the factorizations of 8 that divide
    // the tile grid evenly (nbr = 8, nbc = 1 is what linear id ranges give)
    {
        long best = -1;
        for (int nbr = 1; nbr <= 8; nbr *= 2) {
            const int nbc = 8 / nbr;
            if (p.tiles_m % nbr != 0 || p.tiles_n % nbc != 0) continue;
            const long cost = (long)p.M * nbc + (long)p.N * nbr;
            if (best < 0 || cost < best) { best = cost; p.xcd_nbr = nbr; p.xcd_nbc = nbc; }
        }
#ifdef VITA_XCD_LINEAR
        p.xcd_nbr = 0;
#endif
    }
    const bool ln = p.pro != VITA_PRO_NONE;
    snprintf(g_last_cfg, sizeof(g_last_cfg), "gemm_f32_dma_kernel<%d,%d,%d,%d,KG=%d%s> xcd %dx%d", BM, BN, BK, NS, KG, ln ? ",LN" : "",
             p.xcd_nbr, p.xcd_nbc);
    const size_t lds = ((size_t)NS * (BM + BN) * BK + (ln ? 2 * (size_t)p.K + 2 * BM : 0) + (p.epi.ln_stats ? 2 * BM : 0)) * sizeof(float);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (p.sk_splits > 1 ? p.sk_splits : 1))), block(DNT);
    if (ln) hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, true, OCC, false, KG>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, OCC, false, KG>), grid, block, lds, stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// Eligibility is decided by the caller (vita_gemm_f32.hip): 16-byte aligned X/W rows, K % 4 == 0,
// LayerNorm vectors broadcast over rows (ld == 0) and K small enough for the LDS tables.
int gemm_f32_dma(const vita_linear_args& a, hipStream_t stream, const SplitK* sk, const EpiFold* fold) {
    GemmParamsDma p{};
    p.X = a.X; p.ldx = a.ldx; p.W = a.W; p.ldw = a.ldw; p.bias = a.bias;
    p.Y = a.Y; p.ldy = a.ldy; p.M = a.M; p.N = a.N; p.K = a.K;
    p.act = a.act; p.pro = a.pro; p.ln_eps = a.ln_eps;
    p.pro_mul = a.pro_mul; p.pro_add = a.pro_add;
    p.gate = a.gate; p.ld_gate = a.ld_gate;
    p.use_gate_scalar = a.use_gate_scalar; p.gate_scalar = a.gate_scalar;
    p.resid = a.resid; p.ld_resid = a.ld_resid;
    p.preact = a.preact; p.ld_preact = a.ld_preact;
    p.epi.Y = a.Y; p.epi.ldy = a.ldy; p.epi.M = a.M; p.epi.N = a.N; p.epi.bias = a.bias; p.epi.act = a.act;
    p.epi.gate = a.gate; p.epi.ld_gate = a.ld_gate; p.epi.use_gate_scalar = a.use_gate_scalar;
    p.epi.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    p.epi.resid = a.resid; p.epi.ld_resid = a.ld_resid; p.epi.preact = a.preact; p.epi.ld_preact = a.ld_preact;
    p.epi.act_grad_z = a.act_grad_z; p.epi.ld_act_grad_z = a.ld_act_grad_z;
    p.epi.vec = epi_vec_ok(p.epi);
    if (fold) {
        auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
        VITA_REQUIRE(p.epi.vec && a.M % 256 == 0 && a.N % 128 == 0 && sk == nullptr && al(fold->aux_out) && al(fold->aux_mul) &&
                         al(fold->ln_c1) && fold->ld_aux % 4 == 0,
                     "folded LayerNorm epilogue needs interior, 16-byte aligned tiles (M=%d N=%d)", a.M, a.N);
        VITA_REQUIRE((fold->aux_out == nullptr) == (fold->stat_out == nullptr) && (fold->aux_out == nullptr) == (fold->aux_mul == nullptr) &&
                         (fold->ln_stats == nullptr) == (fold->ln_c1 == nullptr) && (fold->ln_stats == nullptr || fold->ln_nseg >= 1),
                     "folded LayerNorm epilogue: inconsistent arguments");
        p.epi.aux_out = fold->aux_out; p.epi.ld_aux = fold->ld_aux; p.epi.aux_mul = fold->aux_mul; p.epi.aux_add_one = fold->aux_add_one;
        p.epi.stat_out = fold->stat_out; p.epi.ln_stats = fold->ln_stats; p.epi.ln_c1 = fold->ln_c1; p.epi.ln_nseg = fold->ln_nseg;
        p.epi.ln_fold_eps = fold->ln_fold_eps;
    }
    if (sk && sk->splits > 1) { p.sk_splits = sk->splits; p.sk_k = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial; }
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
#ifndef VITA_DMA_CFG
#define VITA_DMA_CFG 5
#endif
#if VITA_DMA_CFG == 0
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    return launch_dma<64, 128, 32, 4>(p, stream);
#elif VITA_DMA_CFG == 1   // two co-resident workgroups per CU everywhere (72 KB LDS each)
    return launch_dma<64, 128, 32, 3, 4>(p, stream);
#elif VITA_DMA_CFG == 2   // two per CU only for the small-N GEMMs
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    return launch_dma<64, 128, 32, 3, 4>(p, stream);
#elif VITA_DMA_CFG == 3
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    return launch_dma<64, 128, 32, 3>(p, stream);
#elif VITA_DMA_CFG == 4   // deeper k-tiles for the 64-row tile (one barrier per 32 MFMAs per wave): fc2 95.4 -> 89.6 us
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    return launch_dma<64, 128, 64, 3>(p, stream);
#else   // 5: 256x128 tiles when they still give every CU a tile (6 DMA issues per 64 MFMAs per wave instead of 4 per 32)
    if ((long)((a.M + 255) / 256) * ((a.N + 127) / 128) >= 256 && sk == nullptr) return launch_dma<256, 128, 32, 3>(p, stream);
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    // 64-row tile: 4 k-groups of 2 waves, a wave owns 64x64 outputs over a quarter of each k-tile (fc2 88.9 -> 84 us; 2 groups: 85 us)
    return launch_dma<64, 128, 64, 3, 2, 4>(p, stream);
#endif
}

// bf16 operands: a.X / a.W point at bf16 data, a.K / a.ldx / a.ldw count bf16 elements (multiples of 8, 16-byte aligned
// rows); everything on the output side is as in the fp32 call, plus the optional bf16 copy of Y.
// sk: split-K plan in the kernel's 4-byte k units (bf16 pairs), or NULL.  act_grad_z: as in the transposed fp32 GEMMs.
int gemm_bf16_dma(const vita_linear_args& a, uint16_t* ybf, int64_t ld_ybf, hipStream_t stream, const SplitK* sk) {
    GemmParamsDma p{};
    p.X = a.X; p.ldx = a.ldx / 2; p.W = a.W; p.ldw = a.ldw / 2; p.bias = a.bias;
    p.Y = a.Y; p.ldy = a.ldy; p.M = a.M; p.N = a.N; p.K = a.K / 2;
    p.act = a.act; p.pro = VITA_PRO_NONE;
    p.epi.Y = a.Y; p.epi.ldy = a.ldy; p.epi.M = a.M; p.epi.N = a.N; p.epi.bias = a.bias; p.epi.act = a.act;
    p.epi.gate = a.gate; p.epi.ld_gate = a.ld_gate; p.epi.use_gate_scalar = a.use_gate_scalar;
    p.epi.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    p.epi.resid = a.resid; p.epi.ld_resid = a.ld_resid; p.epi.preact = a.preact; p.epi.ld_preact = a.ld_preact;
    p.epi.act_grad_z = a.act_grad_z; p.epi.ld_act_grad_z = a.ld_act_grad_z;
    p.epi.ybf = ybf; p.epi.ld_ybf = ld_ybf;
    p.epi.vec = epi_vec_ok(p.epi);
    if (sk && sk->splits > 1) { p.sk_splits = sk->splits; p.sk_k = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial; }
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (t128 >= 512) return launch_dma_bf16<128, 128, 32, 4>(p, stream);
    return launch_dma_bf16<64, 128, 32, 4>(p, stream);
}

bool gemm_f32_dma_eligible(const vita_linear_args& a) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (a.K % 4 != 0 || a.ldx % 4 != 0 || a.ldw % 4 != 0 || !al(a.X) || !al(a.W)) return false;
    if (a.pro != VITA_PRO_NONE) {
        if (a.ld_pro_mul != 0 || a.ld_pro_add != 0) return false;
        if (!al(a.pro_mul) || !al(a.pro_add)) return false;
        if (a.K > 2048) return false;
    }
    return true;
}

}  // namespace vita

#ifdef VITA_STAMP
extern "C" int vita_lab_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vita::g_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif
