// The LDS-DMA GEMM tile (device code shared by vita_gemm_f32_dma.hip's stand-alone kernel and the persistent Euler sampler
// of vita_ode_persistent.hip).  See vita_gemm_f32_dma.hip for the design notes.
#pragma once
#include <type_traits>

#include "vita_common.h"
#include "vita_gemm_epilogue.h"

namespace vita {

#ifndef STAMP
#define STAMP(i)
#define RSTAMP(i)
#endif

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

static __device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

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
// One output tile of the LDS-DMA GEMM: the whole body of gemm_f32_dma_kernel, callable from a persistent kernel as well.
// smem: the workgroup's dynamic LDS (layout as in the kernel's header comment); (tile_m, tile_n, split) choose the tile.
// SC1: the epilogue stores write through to the fabric (`sc1`), the producer half of an in-launch hand-off to other
// workgroups (vita_ode_persistent.hip).  Sync: a functor run between the prefetch of the tile's first WEIGHT stages (which
// depend on no other workgroup) and every load of the A operand / epilogue operands; NoSync for stand-alone launches.
// FACT >= 0: the activation is known at compile time (only that epilogue is instantiated).
// Returns false, uniformly over the workgroup, when the functor gave up (a bounded wait ran out).
struct NoSync {
    __device__ __forceinline__ bool operator()() const { return true; }
};

template <int BM, int BN, int BK, int NS, bool LN, bool BF16, int KG, bool BWD, bool SC1, int FACT = -1, class Sync = NoSync>
__device__ __forceinline__ bool dma_tile(const GemmParamsDma& p, float* smem, int tile_m, int tile_n, int split, Sync sync) {
    constexpr bool SYNCED = !std::is_same<Sync, NoSync>::value;

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
    auto issue_a = [&](int kt, int stage) {
        float* sa = smem + stage * STAGE;
        const int kb = kbase + kt * BK;
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const float* g = (kb + a_k[j] < K) ? a_src[j] + kb : g_zero_chunk;
            dma16(g, sa + (j * 8 + wave) * RPI * BK);
        }
    };
    auto issue_b = [&](int kt, int stage) {
        float* sb = smem + stage * STAGE + BM * BK;
        const int kb = kbase + kt * BK;
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) {
            const float* g = (kb + b_k[j] < K) ? b_src[j] + kb : g_zero_chunk;
            dma16(g, sb + (j * 8 + wave) * RPI * BK);
        }
    };
    auto issue_tile = [&](int kt, int stage) {
        issue_a(kt, stage);
        issue_b(kt, stage);
    };

    // prologue DMAs first: they fly while the LayerNorm statistics are computed
    if constexpr (SYNCED) {
        // the weight halves of the first stages depend on nobody: they fly while the workgroup waits for its producers
        static_assert(NS == 3, "the split prologue relies on the k-loop's first wait being vmcnt(0)");
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_b(s, s);
        if (!sync()) {
            wait_vmcnt<0>();      // the prefetched stages must have landed before the wave may leave
            return false;
        }
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_a(s, s);
    } else {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_tile(s, s);
    }

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
#ifndef VITA_LAB_NOMFMA    // timing-only lab builds: operand delivery alone
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
#endif
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
    epilogue_dispatch<BM, BN, MI, NI, DNT, BWD, KG, SC1, FACT>(acc, smem, e, m0, n0, wm, wn, lane, tid, kg, sfold);
    STAMP(3);
    RSTAMP(5);
    return true;
}

}  // namespace vita
