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

#include "vita_gemm_dma_tile.h"
#include "vita_host.h"

namespace vita {

template <int BM, int BN, int BK, int NS, bool LN, int OCC, bool BF16 = false, int KG = 1, bool BWD = false>
__global__ __launch_bounds__(DNT, OCC) void gemm_f32_dma_kernel(GemmParamsDma p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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
    dma_tile<BM, BN, BK, NS, LN, BF16, KG, BWD, false, -1>(p, smem, tile_m, tile_n, split, NoSync{});
}

// what the last LDS-DMA launch of this host thread dispatched (bench.py labels its roofline kernel from it)
static thread_local char g_last_cfg[96] = "";
const char* last_dma_config() { return g_last_cfg; }

template <int BM, int BN, int BK, int NS, int OCC = 2, int KG = 1>
static int launch_dma_bf16(const GemmParamsDma& p0, hipStream_t stream) {
    GemmParamsDma p = p0;
    snprintf(g_last_cfg, sizeof(g_last_cfg), "gemm_f32_dma_kernel<%d,%d,%d,%d,bf16>", BM, BN, BK, NS);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = ((size_t)NS * (BM + BN) * BK + (p.epi.ln_stats ? 2 * BM : 0)) * sizeof(float);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (p.sk_splits > 1 ? p.sk_splits : 1)));
    if (p.epi.act_grad_z != nullptr)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, OCC, true, KG, true>), grid, dim3(DNT), lds, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, OCC, true, KG>), grid, dim3(DNT), lds, stream, p);
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
    else if (p.epi.act_grad_z != nullptr)   // input-gradient GEMM on transposed weights: epilogue multiplies by act'(z)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, BK, NS, false, OCC, false, KG, true>), grid, block, lds, stream, p);
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
        VITA_REQUIRE(p.epi.vec && a.M % 256 == 0 && a.N % 128 == 0 && al(fold->aux_out) && al(fold->aux_mul) &&
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
#ifdef VITA_LAB_KG      // lab: the k-group count of the 64-row tile for short reductions
    if (a.K <= 512 && (long)((a.M + 255) / 256) * ((a.N + 127) / 128) < 256) return launch_dma<64, 128, 64, 3, 2, VITA_LAB_KG>(p, stream);
#endif
    if ((fold == nullptr || a.M % 4096 != 0) && (sk == nullptr || sk->splits <= 1)) {
        // (calls with the folded-LayerNorm epilogue at the batches the persistent sampler takes -- multiples of 4096 rows -- keep the
        // rule below: it walks the same tiles and must give the same bits, tests/test_gpu_persistent.py)
        // One workgroup per CU at a time: a launch costs ceil(tiles / 256) rounds of its tile's time.  Tile times relative to the
        // 64-row tile, from the stand-alone kernels at 4096 rows (25.7 / 45 / 83 us per round): 1, 1.8, 3.4.  B = 4096: 256-row tiles
        // (one round); 2048: 128-row (one round instead of two 64-row ones: sampler 2.99 -> 2.87 ms); 3072: 64-row (three rounds
        // beat 1.5 -> 2 rounds of 128-row tiles: 3.96 against 4.23 ms).
        const long nt = (a.N + 127) / 128;
        auto rounds = [&](int rows) { return (double)((((long)(a.M + rows - 1) / rows) * nt + 255) / 256); };
        const double c64 = rounds(64) * 1.0, c128 = rounds(128) * 1.8, c256 = rounds(256) * 3.4;
        if (c256 <= c128 && c256 <= c64 && (long)((a.M + 255) / 256) * nt >= 256) return launch_dma<256, 128, 32, 3>(p, stream);
        if (c128 < c64) return launch_dma<128, 128, 32, 4>(p, stream);
        return launch_dma<64, 128, 64, 3, 2, 4>(p, stream);
    }
    if ((long)((a.M + 255) / 256) * ((a.N + 127) / 128) >= 256 && sk == nullptr) return launch_dma<256, 128, 32, 3>(p, stream);
    if (t128 >= 512) return launch_dma<128, 128, 32, 4>(p, stream);
    // 64-row tile: 4 k-groups of 2 waves, a wave owns 64x64 outputs over a quarter of each k-tile (fc2 88.9 -> 84 us; 2 groups: 85 us)
    return launch_dma<64, 128, 64, 3, 2, 4>(p, stream);
#endif
}

// bf16 operands: a.X / a.W point at bf16 data, a.K / a.ldx / a.ldw count bf16 elements (multiples of 8, 16-byte aligned
// rows); everything on the output side is as in the fp32 call, plus the optional bf16 copy of Y.
// sk: split-K plan in the kernel's 4-byte k units (bf16 pairs), or NULL.  act_grad_z: as in the transposed fp32 GEMMs.
int gemm_bf16_dma(const vita_linear_args& a, uint16_t* ybf, int64_t ld_ybf, hipStream_t stream, const SplitK* sk, const EpiFold* fold) {
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
    if (fold) {     // LayerNorm folded into the neighbouring GEMMs, as in gemm_f32_dma: interior 128-row / 128-column tiles only
        auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
        VITA_REQUIRE(p.epi.vec && a.M % 128 == 0 && a.N % 128 == 0 && sk == nullptr && al(fold->aux_out) && al(fold->aux_bf) && al(fold->aux_mul) &&
                         al(fold->ln_c1) && fold->ld_aux % 4 == 0 && fold->ld_aux_bf % 4 == 0,
                     "folded LayerNorm epilogue (bf16) needs interior, 16-byte aligned tiles (M=%d N=%d)", a.M, a.N);
        p.epi.aux_out = fold->aux_out; p.epi.ld_aux = fold->ld_aux; p.epi.aux_bf = fold->aux_bf; p.epi.ld_aux_bf = fold->ld_aux_bf;
        p.epi.aux_mul = fold->aux_mul; p.epi.aux_add_one = fold->aux_add_one;
        p.epi.stat_out = fold->stat_out; p.epi.ln_stats = fold->ln_stats; p.epi.ln_c1 = fold->ln_c1; p.epi.ln_nseg = fold->ln_nseg;
        p.epi.ln_fold_eps = fold->ln_fold_eps;
    }
    if (sk && sk->splits > 1) { p.sk_splits = sk->splits; p.sk_k = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial; }
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (t128 >= 512) return launch_dma_bf16<128, 128, 32, 4>(p, stream);
    // (round 3 lab, same box, B = 4096 bf16 step with this tile / <64,128,64,3> / <64,128,64,3,KG=4>: 9.88 / 10.02 / 10.12 ms; alone the
    // KG = 4 form is 10 % faster on warm operands (18.3 against 20.8 us at K = 2048), in the step it is not.  Timing-only builds without
    // the refills / without the MFMAs take 12.5 / 13.0 us of the 19.2: delivery and compute add up instead of overlapping -- a wave
    // that issues its 3 DMA pieces per k-tile has only 4 MFMAs = 128 cycles to put beside them)
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
