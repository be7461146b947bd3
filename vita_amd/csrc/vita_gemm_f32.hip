// fp32 MFMA GEMM with fused prologue / epilogue — the workhorse of the VITA hot path.
//
//   Y[M,N] = resid + gate * act( pro(X)[M,K] · W[N,K]^T + bias )
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fmaf chain (no TF32/xf32 on
// gfx950), so the result differs from the reference's fp32 GEMM only by summation order.
//
// Tiling (one workgroup = 512 threads = 8 waves, 2 per SIMD):
//   block tile BM x BN, waves laid out 2 (M) x 4 (N), wave tile (BM/2) x (BN/4) made of
//   32x32 MFMA accumulators; K is walked in BK-deep tiles through a ring of 3 LDS buffers:
//     global -> registers (two tiles ahead) -> [LayerNorm/modulation prologue on the A rows]
//            -> LDS (one tile ahead) -> fragment registers (one 8-deep chunk ahead) -> MFMA
//   one barrier per K tile, and no LDS or global wait on the MFMA stream's critical path.  LDS rows are padded to BK+4 floats: the ds_read_b128 fragment
//   reads (32 rows x 16 B per half-wave) and the ds_write_b128 staging writes are then
//   bank-conflict-free (row stride = 4*odd dwords).
//   A ds_read_b128 gives a lane 4 consecutive k of its row; lanes 0-31 take k = 8c..8c+3,
//   lanes 32-63 k = 8c+4..8c+7, so register j of the A and B fragments pair k = 8c+j with
//   8c+4+j in one 32x32x2 MFMA: 4 MFMAs per pair of 16-byte LDS reads.
#include <type_traits>

#include "vita_gemm_epilogue.h"
#include "vita_host.h"

namespace vita {

#define VITA_TRY_RC(expr) do { int rc__ = (expr); if (rc__ != VITA_OK) return rc__; } while (0)

struct GemmParams {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;
    float* Y; int64_t ldy;
    int M, N, K;
    int act, pro;
    float ln_eps;
    const float* pro_mul; int64_t ld_pro_mul;
    const float* pro_add; int64_t ld_pro_add;
    const float* gate; int64_t ld_gate;
    int use_gate_scalar; float gate_scalar;
    const float* resid; int64_t ld_resid;
    float* preact; int64_t ld_preact;
    int tiles_m, tiles_n;
    int vec_x, vec_w, vec_p;  // 16-byte vector loads legal for X / W / pro vectors
    EpiParams epi;
};

constexpr int NT = 512;

// A page of zeros: loads whose k lies beyond K are redirected here by an ADDRESS select, so the
// loaded registers are not touched until they are written to LDS (a data select right after the
// load would make the compiler wait for the load at issue time and serialise the pipeline).
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

// Guarded load of 4 consecutive k of one row.  Rows beyond M / N are clamped by the caller to
// the last valid row; their products are discarded at the store.
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ row, int k, int K) {
    f32x4 v;
    if constexpr (VEC) {  // K % 4 == 0, 16-byte aligned rows
        const float* src = (k < K) ? row + k : g_zero_page;
        v = *reinterpret_cast<const f32x4*>(src);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* src = (k + j) < K ? row + k + j : g_zero_page;
            v[j] = *src;
        }
    }
    return v;
}

template <int BM, int BN, int BK, bool LN, bool VEC>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmParams p) {
    constexpr int S = BK + 4;             // padded LDS row stride (floats)
    constexpr int C4 = BK / 4;            // float4 slots per tile row
    constexpr int A_SLOTS = BM * C4 / NT;
    constexpr int B_SLOTS = BN * C4 / NT;
    constexpr int TM = BM / 2, TN = BN / 4;
    constexpr int MI = TM / 32, NI = TN / 32;
    static_assert(A_SLOTS >= 1 && B_SLOTS >= 1 && MI >= 1 && NI >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [3][BM][S]
    float* Bs = smem + 3 * BM * S;         // [3][BN][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give
    // each XCD a contiguous run of tiles so its L2 sees each A row-tile from one place.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = p.K;
    const int nk = (K + BK - 1) / BK;

    // per-thread staging slots
    int a_row[A_SLOTS], a_c[A_SLOTS];
    const float* a_ptr[A_SLOTS];
    int a_grow[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int s = tid + i * NT;
        a_row[i] = s / C4;
        a_c[i] = (s % C4) * 4;
        a_grow[i] = min(m0 + a_row[i], p.M - 1);
        a_ptr[i] = p.X + (int64_t)a_grow[i] * p.ldx;
    }
    int b_row[B_SLOTS], b_c[B_SLOTS];
    const float* b_ptr[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int s = tid + i * NT;
        b_row[i] = s / C4;
        b_c[i] = (s % C4) * 4;
        b_ptr[i] = p.W + (int64_t)min(n0 + b_row[i], p.N - 1) * p.ldw;
    }

    // ---- LayerNorm statistics of this block's rows (two-pass, over the whole K extent) ----
    // Loads are issued SB at a time before any is consumed so they overlap; for K <= SB*BK the
    // row stays in registers between the mean pass and the variance pass.
    float mean[A_SLOTS], rstd[A_SLOTS];
    if constexpr (LN) {
        constexpr int SB = 16;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            f32x4 v[SB];
            float s = 0.f;
            for (int kb = 0; kb < nk; kb += SB) {
#pragma unroll
                for (int u = 0; u < SB; ++u) v[u] = load4<VEC>(a_ptr[i], (kb + u) * BK + a_c[i], K);
#pragma unroll
                for (int u = 0; u < SB; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            }
#pragma unroll
            for (int o = C4 / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mu = s / (float)K;
            float q = 0.f;
            for (int kb = 0; kb < nk; kb += SB) {
                if (nk > SB) {
#pragma unroll
                    for (int u = 0; u < SB; ++u) v[u] = load4<VEC>(a_ptr[i], (kb + u) * BK + a_c[i], K);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int k = (kb + u) * BK + a_c[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = (k + j < K) ? v[u][j] - mu : 0.f;
                        q = fmaf(d, d, q);
                    }
                }
            }
#pragma unroll
            for (int o = C4 / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            mean[i] = mu;
            rstd[i] = 1.0f / sqrtf(q / (float)K + p.ln_eps);
        }
    }

    f32x4 a_reg[A_SLOTS], b_reg[B_SLOTS];
    f32x4 pm_reg[A_SLOTS], pa_reg[A_SLOTS];

    auto issue_loads = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int k = kt * BK + a_c[i];
            a_reg[i] = load4<VEC>(a_ptr[i], k, K);
            if constexpr (LN) {
                const int64_t r = a_grow[i];
                pm_reg[i] = load4<VEC>(p.pro_mul + r * p.ld_pro_mul, k, K);
                pa_reg[i] = load4<VEC>(p.pro_add + r * p.ld_pro_add, k, K);
            }
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            b_reg[i] = load4<VEC>(b_ptr[i], kt * BK + b_c[i], K);
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            f32x4 v = a_reg[i];
            if constexpr (LN) {
                const bool mod = p.pro == VITA_PRO_LN_MOD;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float n = (v[j] - mean[i]) * rstd[i];
                    const float mul = mod ? (pm_reg[i][j] + 1.0f) : pm_reg[i][j];
                    v[j] = n * mul + pa_reg[i][j];
                }
            }
            *reinterpret_cast<f32x4*>(&As[(buf * BM + a_row[i]) * S + a_c[i]]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + b_row[i]) * S + b_c[i]]) = b_reg[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop: 3 LDS buffers, fragment reads software-pipelined one chunk ahead -------
    // iteration kt:  MFMAs read tile kt from buf[kt%3]; tile kt+1 is already complete in
    // buf[(kt+1)%3] (written during kt-1, barrier passed), so the last chunk prefetches its
    // first fragments across the barrier; the registers holding tile kt+2 (loaded during kt-1)
    // are written to buf[(kt+2)%3] and re-used for the global loads of tile kt+3.
    constexpr int NCH = BK / 8;
    static_assert(NCH % 2 == 0, "fragment double-buffer parity");
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    const float* Afr = &As[(wm * TM + frag_row) * S + frag_k];
    const float* Bfr = &Bs[(wn * TN + frag_row) * S + frag_k];
    f32x4 af[2][MI], bf[2][NI];
    auto read_frags = [&](int buf, int kc, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Afr + (buf * BM + i * 32) * S + kc * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bfr + (buf * BN + j * 32) * S + kc * 8);
    };

    issue_loads(0);
    write_lds(0);
    if (nk > 1) {
        issue_loads(1);
        write_lds(1);
    }
    if (nk > 2) issue_loads(2);
    __syncthreads();
    read_frags(0, 0, af[0], bf[0]);

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf1 = buf == 2 ? 0 : buf + 1;
        const int buf2 = buf1 == 2 ? 0 : buf1 + 1;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            // The body is branch-free on purpose (a conditional prefetch gets sunk below the MFMAs
            // by the compiler and exposes the LDS latency after every barrier): past the last
            // tile the prefetch reads a stale-but-valid buffer, the staging write lands in a
            // buffer nobody reads again, and load4 redirects k >= K to the zero page.
            // LDS ops retire in order, so the staging writes go after the first MFMA group:
            // nothing but already-landed fragments gates the first MFMA after the barrier.
#if !defined(VITA_LAB) || (VITA_LAB != 2 && VITA_LAB != 4)
            if (kc == 1) write_lds(buf2);
#endif
#if !defined(VITA_LAB) || (VITA_LAB != 3 && VITA_LAB != 4)
            if (kc == 2) issue_loads(kt + 3);
#endif
            if (kc + 1 < NCH) read_frags(buf, kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
            else read_frags(buf1, 0, af[0], bf[0]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc & 1][i][e], bf[kc & 1][j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if !defined(VITA_LAB) || VITA_LAB != 1
        __syncthreads();
#endif
        buf = buf1;
    }

    // ---- epilogue: through the (now dead) staging buffers, see vita_gemm_epilogue.h (the k-loop ended on a barrier)
    epilogue_dispatch<BM, BN, MI, NI, NT, false>(acc, smem, p.epi, m0, n0, wm, wn, lane, tid);
}

template <int BM, int BN, int BK>
static int launch_cfg(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = (size_t)3 * (BM + BN) * (BK + 4) * sizeof(float);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(NT);
    const bool vec = p.vec_x && p.vec_w && (p.pro == VITA_PRO_NONE || p.vec_p);
    if (p.pro != VITA_PRO_NONE) {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, false>), grid, block, lds, stream, p);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false, false>), grid, block, lds, stream, p);
    }
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int gemm_f32_dma(const vita_linear_args& a, hipStream_t stream, const SplitK* sk, const EpiFold* fold = nullptr);
int gemm_f32_t(const vita_linear_args& a, hipStream_t stream, const SplitK* sk);
int gemm_f32_t_batch(const vita_linear_args* a, int n, hipStream_t stream);
int gemm_f32_t_check(const vita_linear_args& a);
int gemm_f32_small(const vita_linear_args& a, hipStream_t stream, const SplitK* sk);
bool gemm_f32_small_eligible(const vita_linear_args& a, bool have_splitk_ws);
bool gemm_f32_dma_eligible(const vita_linear_args& a);

// ---- split-K: second pass ------------------------------------------------------------------------
// Y = epilogue( sum_{s=0..S-1} partial[s] ) with the splits added in index order (deterministic).
constexpr int SPLITK_MAX_TILES = 128;     // split only when a launch would fill at most half of the 256 CUs
constexpr int SPLITK_MIN_K = 128;         // k elements per split (a multiple of every kernel's BK)

template <int ACT, bool BWD>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(EpiParams e, const float* __restrict__ part, int64_t ldp, int S) {
    const int c4 = (e.N + 3) / 4;
    const int64_t n = (int64_t)e.M * c4, stride = (int64_t)gridDim.x * 256;
    const bool gate_vec = (e.gate != nullptr) && (e.ld_gate == 0);
    const bool gate_mat = (e.gate != nullptr) && (e.ld_gate != 0);
    const bool has_gate = e.use_gate_scalar || (e.gate != nullptr);
    const bool has_resid = e.resid != nullptr;
    const int64_t split_stride = (int64_t)e.M * ldp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int row = (int)(i / c4), col = (int)(i - (int64_t)row * c4) * 4;
        const float* src = part + (int64_t)row * ldp + col;
        // partials are requested 8 at a time before any is added (a latency-bound kernel: one round trip per 8 splits
        // instead of one per split); the additions keep the index order
        f32x4 a = *reinterpret_cast<const f32x4*>(src);
        for (int s0 = 1; s0 < S; s0 += 8) {
            f32x4 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < S) b[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(s0 + u) * split_stride);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < S) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = a[k] + b[u][k];
                }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = col + k;
            if (cc >= e.N) break;
            const float b = e.bias ? e.bias[cc] : 0.f;
            float zin = 0.f;
            if constexpr (BWD) zin = e.act_grad_z[(int64_t)row * e.ld_act_grad_z + cc];
            const float g = gate_vec ? e.gate[cc] : (gate_mat ? e.gate[(int64_t)row * e.ld_gate + cc] : e.gate_scalar);
            const float rs = has_resid ? e.resid[(int64_t)row * e.ld_resid + cc] : 0.f;
            float p_ = 0.f;
            const float v = epi_elem<ACT, BWD>(a[k], b, zin, g, has_gate, rs, has_resid, e.preact ? &p_ : nullptr);
            if (e.preact) e.preact[(int64_t)row * e.ld_preact + cc] = p_;
            e.Y[(int64_t)row * e.ldy + cc] = v;
        }
    }
}

// Second pass of a split-K GEMM that carries the folded-LayerNorm epilogue (vita_gemm_epilogue.h, EpiParams: producer fields aux_* /
// stat_out, consumer fields ln_stats / ln_c1): sums the partials in split order and applies what the tile epilogue applies — the
// consumer's finish rstd * (acc - mean * c1), bias, gate, residual, then the producer's aux = out * mul and the per-row, per-128-column
// (mean, M2) — so the sampler's 64-row GEMMs can split K at batches where they would otherwise leave most CUs without a tile.
// Layout: one thread per float4, N % 128 == 0: the 32 lanes of a half-wave hold one row's 128-column segment, as in the tile epilogue.
// M * N / 4 must be a multiple of 256 (every lane of every wave active: the row statistics use DPP reductions).
__global__ __launch_bounds__(256) void splitk_reduce_fold_kernel(EpiParams e, const float* __restrict__ part, int64_t ldp, int S) {
    const int c4 = e.N / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(i / c4), col = (int)(i - (int64_t)row * c4) * 4;
    const int lane = threadIdx.x & 63;
    const int64_t split_stride = (int64_t)e.M * ldp;
    const float* src = part + (int64_t)row * ldp + col;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    for (int s0 = 1; s0 < S; s0 += 8) {
        f32x4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) b[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(s0 + u) * split_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = a[k] + b[u][k];
            }
    }
    if (e.ln_stats) {      // consumer: finish the normalisation on the summed accumulators
        float st2[2];
        fold_row_stats(e, row, 128, st2);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(e.ln_c1 + col);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = st2[1] * fmaf(-st2[0], c1[k], a[k]);
    }
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, g = {e.gate_scalar, e.gate_scalar, e.gate_scalar, e.gate_scalar}, rs = bias, out;
    if (e.bias) bias = *reinterpret_cast<const f32x4*>(e.bias + col);
    const bool has_gate = e.use_gate_scalar || (e.gate != nullptr);
    if (e.gate) g = *reinterpret_cast<const f32x4*>(e.gate + (e.ld_gate ? (int64_t)row * e.ld_gate : 0) + col);
    if (e.resid) rs = *reinterpret_cast<const f32x4*>(e.resid + (int64_t)row * e.ld_resid + col);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = epi_elem<VITA_ACT_NONE, false>(a[k], bias[k], 0.f, g[k], has_gate, rs[k], e.resid != nullptr, nullptr);
    *reinterpret_cast<f32x4*>(e.Y + (int64_t)row * e.ldy + col) = out;
    if (e.aux_out != nullptr) {
        f32x4 am = *reinterpret_cast<const f32x4*>(e.aux_mul + col);
        if (e.aux_add_one) am = am + 1.0f;
        *reinterpret_cast<f32x4*>(e.aux_out + (int64_t)row * e.ld_aux + col) = out * am;
        // (mean, M2) of the row over its 128-column segment: the tile epilogue's shifted one-pass sums, in the same order
        const float xa = __builtin_amdgcn_readlane(out[0], 0), xb = __builtin_amdgcn_readlane(out[0], 32);
        const float x0 = lane < 32 ? xa : xb;
        float sm = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = out[k] - x0; sm += d; q = fmaf(d, d, q); }
        sm = half_wave_sum(sm);
        q = half_wave_sum(q);
        if ((lane & 31) == 0) {
            const int64_t so = ((int64_t)(col / 128) * e.M + row) * 2;
            e.stat_out[so] = x0 + sm * (1.0f / 128.0f);
            e.stat_out[so + 1] = q - sm * sm * (1.0f / 128.0f);
        }
    }
}
static bool splitk_fold_ok(const vita_linear_args& a) {
    return a.act == VITA_ACT_NONE && a.N % 128 == 0 && ((int64_t)a.M * (a.N / 4)) % 256 == 0 && a.preact == nullptr && a.act_grad_z == nullptr;
}

static int launch_splitk_reduce(const EpiParams& e, const float* part, int64_t ldp, int S, hipStream_t stream) {
    const int64_t n = (int64_t)e.M * ((e.N + 3) / 4);
    int blocks = (int)((n + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
#define VITA_RED(A, B) hipLaunchKernelGGL((splitk_reduce_kernel<A, B>), dim3(blocks), dim3(256), 0, stream, e, part, ldp, S)
    const bool bwd = e.act_grad_z != nullptr;
    switch (e.act) {
        case VITA_ACT_RELU: if (bwd) VITA_RED(VITA_ACT_RELU, true); else VITA_RED(VITA_ACT_RELU, false); break;
        case VITA_ACT_GELU_ERF: if (bwd) VITA_RED(VITA_ACT_GELU_ERF, true); else VITA_RED(VITA_ACT_GELU_ERF, false); break;
        case VITA_ACT_GELU_TANH: if (bwd) VITA_RED(VITA_ACT_GELU_TANH, true); else VITA_RED(VITA_ACT_GELU_TANH, false); break;
        case VITA_ACT_SILU: if (bwd) VITA_RED(VITA_ACT_SILU, true); else VITA_RED(VITA_ACT_SILU, false); break;
        case VITA_ACT_MISH: if (bwd) VITA_RED(VITA_ACT_MISH, true); else VITA_RED(VITA_ACT_MISH, false); break;
        default: VITA_RED(VITA_ACT_NONE, false); break;
    }
#undef VITA_RED
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// Decide whether (and how far) to split the reduction of this call.  Returns splits == 1 for "do not".
static SplitK plan_splitk(const vita_linear_args& a) {
    SplitK sk;
    if (a.splitk_ws == nullptr || (reinterpret_cast<uintptr_t>(a.splitk_ws) & 255u) != 0 || a.pro != VITA_PRO_NONE) return sk;
    const long tiles = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    if (tiles > SPLITK_MAX_TILES || a.K < 2 * SPLITK_MIN_K) return sk;
    long S = 256 / tiles;
    const long by_k = a.K / SPLITK_MIN_K;
    if (S > by_k) S = by_k;
    const int64_t ldp = ((int64_t)a.N + 3) & ~int64_t(3);
    const long by_ws = (long)(a.splitk_ws_bytes / ((size_t)a.M * ldp * sizeof(float)));
    if (S > by_ws) S = by_ws;
    if (S < 2) return sk;
    int kps = (int)((a.K + S - 1) / S);
    kps = (kps + SPLITK_MIN_K - 1) / SPLITK_MIN_K * SPLITK_MIN_K;
    sk.splits = (a.K + kps - 1) / kps;       // every split is non-empty
    sk.k_per_split = kps;
    sk.partial = static_cast<float*>(a.splitk_ws);
    sk.ld_partial = ldp;
    if (sk.splits < 2) sk.splits = 1;
    return sk;
}

static EpiParams epi_of(const vita_linear_args& a) {
    EpiParams e{};
    e.Y = a.Y; e.ldy = a.ldy; e.M = a.M; e.N = a.N; e.bias = a.bias; e.act = a.act;
    e.gate = a.gate; e.ld_gate = a.ld_gate; e.use_gate_scalar = a.use_gate_scalar;
    e.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    e.resid = a.resid; e.ld_resid = a.ld_resid; e.preact = a.preact; e.ld_preact = a.ld_preact;
    e.act_grad_z = a.act_grad_z; e.ld_act_grad_z = a.ld_act_grad_z;
    e.vec = 0;
    return e;
}

// rows of reduction from which a weight-gradient GEMM (both operands reduced over their row index) takes the row-major LDS-DMA
// kernel instead of the register-staged one; VITA_F32_TN_MIN_ROWS overrides (lab / A-B measurements)
static int tn_min_rows() {
    static const int v = [] { const char* e = getenv("VITA_F32_TN_MIN_ROWS"); return e != nullptr ? atoi(e) : 2048; }();
    return v;
}

// dW = dY^T X from ROW-MAJOR bf16 copies of dY (R, M) and X (R, N) (vita_gemm_bf16_tn.hip): the reduction over the R stacked rows
// is split over workgroups until ~2 workgroups per CU are in flight, then summed by the deterministic second pass.
// a = wgrad(...) of the fp32 call: Y / ldy / M / N / K = R and the split-K scratch; a.X / a.W are not read.
int gemm_bf16_wgrad_tn(const uint16_t* dYb, int64_t ld_dy, const uint16_t* Xb, int64_t ldx, const vita_linear_args& a, hipStream_t stream) {
    SplitK sk;
    const long tiles = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.splitk_ws != nullptr && (reinterpret_cast<uintptr_t>(a.splitk_ws) & 255u) == 0 && tiles < 512) {
        long S = (512 + tiles - 1) / tiles;
        const long by_r = a.K / 256;                       // at least 8 k-tiles of 32 rows per split
        if (S > by_r) S = by_r;
        const int64_t ldp = ((int64_t)a.N + 3) & ~int64_t(3);
        const long by_ws = (long)(a.splitk_ws_bytes / ((size_t)a.M * ldp * sizeof(float)));
        if (S > by_ws) S = by_ws;
        if (S >= 2) {
            int rps = (int)((a.K + S - 1) / S);
            rps = (rps + 31) & ~31;
            sk.splits = (a.K + rps - 1) / rps;
            sk.k_per_split = rps;
            sk.partial = static_cast<float*>(a.splitk_ws);
            sk.ld_partial = ldp;
            if (sk.splits < 2) sk.splits = 1;
        }
    }
    VITA_TRY_RC(gemm_bf16_tn(dYb, ld_dy, Xb, ldx, a, stream, sk.splits > 1 ? &sk : nullptr));
    if (sk.splits > 1) return launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream);
    return VITA_OK;
}

// bf16-operand route (training step with vita_loss_config.gemm_bf16): stage A(m,k) and B(n,k) as bf16 rows (transposing
// column operands), run the bf16 LDS-DMA kernel with the call's own epilogue (incl. act'(z) of the backward GEMMs and
// split-K).  *taken = false: the call is not eligible and takes the fp32 kernels.
static int gemm_f32_via_bf16(const vita_linear_args& a, hipStream_t stream, bool* taken) {
    *taken = false;
    Bf16Stage& st = bf16_stage();
    // whatever happens below, this call overwrites a.Y: a bf16 copy of its old contents is stale from here on, and a pending
    // emission request is for THIS output or for nobody
    st.shadow_drop(a.Y);
    uint16_t* emit_dst = (st.emit_y != nullptr && st.emit_y == a.Y) ? st.emit_dst : nullptr;
    const int64_t emit_ld = st.emit_ld;
    const bool emit_only = st.emit_only;
    st.emit_y = nullptr;
    st.emit_only = false;
    if (st.mode == 0 || st.ptr == nullptr || a.pro != VITA_PRO_NONE || a.M <= 0 || a.N <= 0 || a.K <= 0) return VITA_OK;
    if (st.mode == 1 && 2.0 * (double)a.M * (double)a.N * (double)a.K < kBf16MinFlops) return VITA_OK;
    const int Kp = (a.K + 7) & ~7;
    const size_t abytes = ((size_t)a.M * Kp * 2 + 255) & ~size_t(255), bbytes = ((size_t)a.N * Kp * 2 + 255) & ~size_t(255);
    if (abytes + bbytes > st.bytes) return VITA_OK;      // (checked before the cache lookup: an uncached call needs both)
    VITA_REQUIRE(a.ldx >= (a.a_col ? a.M : a.K) && a.ldw >= (a.b_col ? a.N : a.K) && a.ldy >= a.N, "vita_linear (bf16 route): leading dimension too small");
    VITA_REQUIRE(a.act >= VITA_ACT_NONE && a.act <= VITA_ACT_MISH, "vita_linear (bf16 route): unknown activation %d", a.act);
    uint16_t* xa = static_cast<uint16_t*>(st.ptr);
    uint16_t* wb = reinterpret_cast<uint16_t*>(static_cast<char*>(st.ptr) + abytes);
    // a weight the step packed once (Bf16Stage::wc): forward GEMMs read its row form, input-gradient GEMMs its transposed form
    const uint16_t* wcached = nullptr;
    if (!a.a_col) {
        for (int i = 0; i < st.nwc && wcached == nullptr; ++i) {
            const Bf16WEntry& e = st.wc[i];
            if (e.w != a.W) continue;
            if (!a.b_col && a.N == e.out_f && a.K == e.in_f && a.ldw == e.in_f) wcached = e.rows;
            else if (a.b_col && a.N == e.in_f && a.K == e.out_f && a.ldw == e.in_f) wcached = e.t;
        }
    }
    // the A operand's bf16 copy may already exist: written by the kernel that produced the fp32 matrix
    const uint16_t* xshadow = nullptr;
    if (!a.a_col) {
        for (int i = 0; i < st.nsh; ++i)
            if (st.sh[i].src == a.X && st.sh[i].ld == Kp && st.sh[i].rows >= a.M && st.sh[i].cols == a.K) { xshadow = st.sh[i].bf; break; }
        VITA_REQUIRE(xshadow != nullptr || !st.shadow_is_only(a.X), "vita_linear (bf16 step): the A operand exists only as a bf16 copy of another shape");
        if (xshadow != nullptr) st.shadow_drop(a.X);       // consumed
    }
    if (xshadow != nullptr) xa = const_cast<uint16_t*>(xshadow);
    else if (a.a_col) VITA_TRY_RC(launch_pack_bf16_t(a.X, a.ldx, xa, Kp, a.K, a.M, stream));
    else VITA_TRY_RC(launch_pack_bf16(a.X, a.ldx, xa, Kp, a.M, a.K, stream));
    if (wcached != nullptr) wb = const_cast<uint16_t*>(wcached);
    else if (a.b_col) VITA_TRY_RC(launch_pack_bf16_t(a.W, a.ldw, wb, Kp, a.K, a.N, stream));
    else VITA_TRY_RC(launch_pack_bf16(a.W, a.ldw, wb, Kp, a.N, a.K, stream));
    vita_linear_args b = a;
    b.X = reinterpret_cast<const float*>(xa); b.ldx = Kp; b.W = reinterpret_cast<const float*>(wb); b.ldw = Kp; b.K = Kp;
    b.a_col = 0; b.b_col = 0;
    if (!(a.a_col || a.b_col)) { b.act_grad_z = nullptr; }
    vita_linear_args u = b;
    u.K = Kp / 2;                               // the kernel counts k in 4-byte units (bf16 pairs)
    const SplitK sk = plan_splitk(u);
    const bool emit = emit_dst != nullptr && sk.splits <= 1 && emit_ld >= a.N && emit_ld % 8 == 0;   // a split GEMM's epilogue is the reduce pass
    if (emit && emit_only) { b.Y = nullptr; b.ldy = 0; }     // the caller reads only the bf16 copy
    VITA_TRY_RC(gemm_bf16_dma(b, emit ? emit_dst : nullptr, emit ? emit_ld : 0, stream, sk.splits > 1 ? &sk : nullptr));
    if (sk.splits > 1) VITA_TRY_RC(launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream));
    if (emit) st.shadow_add(a.Y, emit_dst, emit_ld, a.M, a.N, emit_only);
    *taken = true;
    return VITA_OK;
}

// Few-row GEMMs (vita_gemm_f32_small.hip): the in-workgroup reduction covers K <= 1024; longer reductions are split over
// workgroups (<= 512 k each, at most 8 splits) and finished by the deterministic second pass.
static int run_small(const vita_linear_args& a, hipStream_t stream) {
    SplitK sk;
#ifndef VITA_SMALL_NOSPLIT_K
#define VITA_SMALL_NOSPLIT_K 1024   // 2048 (no grid split for fc2-sized reductions) measured slower: B=128 train step 3.34 -> 3.89 ms
#endif
    if (a.K > VITA_SMALL_NOSPLIT_K) {
        int S = (a.K + 511) / 512;
        if (S > 8) S = 8;
        const int64_t ldp = ((int64_t)a.N + 3) & ~int64_t(3);
        const long by_ws = (long)(a.splitk_ws_bytes / ((size_t)a.M * ldp * sizeof(float)));
        if (S > by_ws) S = (int)by_ws;
        if (S >= 2) {
            int kps = ((a.K + S - 1) / S + 7) & ~7;
            sk.splits = (a.K + kps - 1) / kps;
            sk.k_per_split = kps;
            sk.partial = static_cast<float*>(a.splitk_ws);
            sk.ld_partial = ldp;
            if (sk.splits < 2) sk.splits = 1;
            // the orchestrator's scratch with zeroed tile counters: the kernel finishes the reduction itself
            const SplitkScratch& sc = splitk_scratch();
            const long tiles = (long)((a.M + 31) / 32) * ((a.N + 31) / 32);
            if (sk.splits > 1 && sc.counters != nullptr && a.splitk_ws == sc.ptr && tiles <= kSplitkCounters &&
                (size_t)sk.splits * a.M * ldp * sizeof(float) <= sc.bytes)
                sk.counters = sc.counters;
        }
    }
    VITA_TRY_RC(gemm_f32_small(a, stream, sk.splits > 1 ? &sk : nullptr));
    if (sk.splits > 1 && sk.counters == nullptr) return launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream);
    return VITA_OK;
}
static bool small_ws_ok(const vita_linear_args& a) {
    return a.splitk_ws != nullptr && (reinterpret_cast<uintptr_t>(a.splitk_ws) & 255u) == 0 &&
           a.splitk_ws_bytes >= (size_t)2 * a.M * (((size_t)a.N + 3) & ~size_t(3)) * sizeof(float);
}

int gemm_f32_fold(const vita_linear_args& a, const EpiFold& fold, hipStream_t stream) {
    VITA_REQUIRE(a.X && a.W && a.Y && !a.a_col && !a.b_col && a.pro == VITA_PRO_NONE && a.preact == nullptr && a.act_grad_z == nullptr &&
                     gemm_f32_dma_eligible(a),
                 "gemm_f32_fold: the call is not eligible for the folded-LayerNorm epilogue");
    // few output tiles (the 64-row GEMMs below ~3072 rows): split the reduction over workgroups; the second pass applies the
    // folded epilogue (round 3 -- before, the fold could not split K and started at 3072 rows)
    const SplitK sk = splitk_fold_ok(a) ? plan_splitk(a) : SplitK{};
    if (sk.splits <= 1 || fold.aux_bf != nullptr) return gemm_f32_dma(a, stream, nullptr, &fold);   // (the second pass has no bf16 output)
    VITA_TRY_RC(gemm_f32_dma(a, stream, &sk, &fold));
    EpiParams e = epi_of(a);
    e.aux_out = fold.aux_out; e.ld_aux = fold.ld_aux; e.aux_mul = fold.aux_mul; e.aux_add_one = fold.aux_add_one;
    e.stat_out = fold.stat_out; e.ln_stats = fold.ln_stats; e.ln_c1 = fold.ln_c1; e.ln_nseg = fold.ln_nseg; e.ln_fold_eps = fold.ln_fold_eps;
    const int64_t n = (int64_t)a.M * (a.N / 4);
    hipLaunchKernelGGL(splitk_reduce_fold_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, stream, e, sk.partial, sk.ld_partial, sk.splits);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// Weight gradients that are ready together (fc1 + fc2 of a flow layer; a whole decoder) at few stacked rows: each alone has too few
// tiles for the chip and would split K and pay a second pass; together, unsplit, they fill it.  Calls that do not fit that picture
// (bf16-operand step, many rows, epilogue options) and groups of fewer than 64 tiles run one by one.
int gemm_f32_wgrad_batch(const vita_linear_args* a, int n, hipStream_t stream) {
    auto plain = [](const vita_linear_args& x) {
        return x.a_col && x.b_col && x.X && x.W && x.Y && x.M > 0 && x.N > 0 && x.K >= 64 && x.K < tn_min_rows() && x.bias == nullptr &&
               x.act_grad_z == nullptr && x.resid == nullptr && !x.use_gate_scalar && x.act == VITA_ACT_NONE && x.pro == VITA_PRO_NONE &&
               x.gate == nullptr && x.preact == nullptr;
    };
    auto tiles = [](const vita_linear_args& x) { return (long)((x.M + 63) / 64) * ((x.N + 127) / 128); };
    int i = 0;
    while (i < n) {
        // greedy groups of up to 8 consecutive eligible calls, at most 1024 tiles
        int j = i;
        long t = 0;
        if (bf16_stage().mode == 0)
            while (j < n && j - i < 8 && plain(a[j]) && t + tiles(a[j]) <= 1024) t += tiles(a[j++]);
        if (j - i >= 2 && t >= 64) {
            VITA_TRY_RC(gemm_f32_t_batch(a + i, j - i, stream));
            i = j;
        } else {
            VITA_TRY_RC(gemm_f32(a[i], stream));
            ++i;
        }
    }
    return VITA_OK;
}
int gemm_f32_wgrad_pair(const vita_linear_args& a0, const vita_linear_args& a1, hipStream_t stream) {
    const vita_linear_args a[2] = {a0, a1};
    return gemm_f32_wgrad_batch(a, 2, stream);
}

int gemm_f32(const vita_linear_args& a, hipStream_t stream) {
    if (a.M == 0 && a.N > 0 && a.K > 0) return VITA_OK;  // empty batch: nothing to do (pointers may be null)
    VITA_REQUIRE(a.X && a.W && a.Y, "vita_linear_f32: X, W and Y must be non-null");
    if (bf16_stage().mode != 0) {
        bool taken = false;
        VITA_TRY_RC(gemm_f32_via_bf16(a, stream, &taken));
        if (taken) return VITA_OK;
        // (a producer that was told to skip its fp32 output relies on this call taking the bf16 route as well)
        VITA_REQUIRE(!bf16_stage().shadow_is_only(a.X), "vita_linear (bf16 step): the A operand exists only as its bf16 copy, but this call cannot take the bf16 route");
    }
    if (!a.a_col && a.b_col && wt_cache().n > 0 && a.M >= kWtMinRows) {
        // dX = dY . W with W^T registered for this step (WtScope): an ordinary row-operand GEMM on the LDS-DMA kernels
        const WtCache& c = wt_cache();
        for (int i = 0; i < c.n; ++i) {
            if (c.e[i].w != a.W || c.e[i].in_f != a.N || c.e[i].out_f != a.K || a.ldw != a.N) continue;
            vita_linear_args b = a;
            b.W = c.e[i].wt; b.ldw = a.K; b.b_col = 0;
            if (!gemm_f32_dma_eligible(b)) break;
            VITA_TRY_RC(gemm_f32_t_check(a));
            const SplitK sk = plan_splitk(b);
            VITA_TRY_RC(gemm_f32_dma(b, stream, sk.splits > 1 ? &sk : nullptr));
            if (sk.splits > 1) return launch_splitk_reduce(epi_of(b), sk.partial, sk.ld_partial, sk.splits, stream);
            return VITA_OK;
        }
    }
    if (a.a_col || a.b_col) {
        if (!a.a_col && gemm_f32_small_eligible(a, small_ws_ok(a))) {
            VITA_TRY_RC(gemm_f32_t_check(a));
            return run_small(a, stream);
        }
        if (a.K >= tn_min_rows() && gemm_f32_tn_eligible(a)) {
            // dW = dY^T X over many stacked rows: the tiles go to LDS as they lie in memory (vita_gemm_f32_tn.hip)
            VITA_TRY_RC(gemm_f32_t_check(a));
            SplitK sk;
            const long tiles = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
            if (a.splitk_ws != nullptr && (reinterpret_cast<uintptr_t>(a.splitk_ws) & 255u) == 0 && tiles < 256) {
                long S = (256 + tiles - 1) / tiles;
                const long by_r = a.K / 256;                       // at least 8 k-tiles of 32 rows per split
                if (S > by_r) S = by_r;
                const int64_t ldp = ((int64_t)a.N + 3) & ~int64_t(3);
                const long by_ws = (long)(a.splitk_ws_bytes / ((size_t)a.M * ldp * sizeof(float)));
                if (S > by_ws) S = by_ws;
                if (S >= 2) {
                    int rps = (int)((a.K + S - 1) / S);
                    rps = (rps + 31) & ~31;
                    sk.splits = (a.K + rps - 1) / rps;
                    sk.k_per_split = rps;
                    sk.partial = static_cast<float*>(a.splitk_ws);
                    sk.ld_partial = ldp;
                    if (sk.splits < 2) sk.splits = 1;
                }
            }
            VITA_TRY_RC(gemm_f32_tn(a, stream, sk.splits > 1 ? &sk : nullptr));
            if (sk.splits > 1) return launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream);
            return VITA_OK;
        }
        const SplitK sk = (a.M > 0 && a.K > 0) ? plan_splitk(a) : SplitK{};
        VITA_TRY_RC(gemm_f32_t(a, stream, sk.splits > 1 ? &sk : nullptr));
        if (sk.splits > 1) return launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream);
        return VITA_OK;
    }
    VITA_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "vita_linear_f32: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    VITA_REQUIRE(a.preact == nullptr || a.ld_preact >= a.N, "vita_linear_f32: ld_preact must be >= N");
    VITA_REQUIRE(a.act_grad_z == nullptr, "vita_linear_f32: act_grad_z is a backward-pass (a_col/b_col) option");
    VITA_REQUIRE(a.ldx >= a.K && a.ldw >= a.K && a.ldy >= a.N,
                 "vita_linear_f32: leading dimension smaller than the row (ldx=%lld ldw=%lld ldy=%lld, K=%d N=%d)",
                 (long long)a.ldx, (long long)a.ldw, (long long)a.ldy, a.K, a.N);
    VITA_REQUIRE(a.act >= VITA_ACT_NONE && a.act <= VITA_ACT_MISH, "vita_linear_f32: unknown activation %d", a.act);
    VITA_REQUIRE(a.pro >= VITA_PRO_NONE && a.pro <= VITA_PRO_LN_AFFINE, "vita_linear_f32: unknown prologue %d", a.pro);
    if (a.pro != VITA_PRO_NONE) {
        VITA_REQUIRE(a.pro_mul && a.pro_add, "vita_linear_f32: LayerNorm prologue needs pro_mul and pro_add");
        VITA_REQUIRE((a.ld_pro_mul == 0 || a.ld_pro_mul >= a.K) && (a.ld_pro_add == 0 || a.ld_pro_add >= a.K),
                     "vita_linear_f32: prologue vector strides must be 0 or >= K");
    }
    VITA_REQUIRE(a.gate == nullptr || a.ld_gate == 0 || a.ld_gate >= a.N, "vita_linear_f32: ld_gate must be 0 or >= N");
    VITA_REQUIRE(a.resid == nullptr || a.ld_resid >= a.N, "vita_linear_f32: ld_resid must be >= N");
    VITA_REQUIRE((const void*)a.Y != (const void*)a.X && (const void*)a.Y != (const void*)a.W,
                 "vita_linear_f32: Y must not alias X or W");
    if (gemm_f32_small_eligible(a, small_ws_ok(a))) return run_small(a, stream);
    if (gemm_f32_dma_eligible(a)) {
        const SplitK sk = plan_splitk(a);
        VITA_TRY_RC(gemm_f32_dma(a, stream, sk.splits > 1 ? &sk : nullptr));
        if (sk.splits > 1) return launch_splitk_reduce(epi_of(a), sk.partial, sk.ld_partial, sk.splits, stream);
        return VITA_OK;
    }

    GemmParams p{};
    p.X = a.X; p.ldx = a.ldx; p.W = a.W; p.ldw = a.ldw; p.bias = a.bias;
    p.Y = a.Y; p.ldy = a.ldy; p.M = a.M; p.N = a.N; p.K = a.K;
    p.act = a.act; p.pro = a.pro; p.ln_eps = a.ln_eps;
    p.pro_mul = a.pro_mul; p.ld_pro_mul = a.ld_pro_mul;
    p.pro_add = a.pro_add; p.ld_pro_add = a.ld_pro_add;
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
    p.vec_x = (a.K % 4 == 0) && (a.ldx % 4 == 0) && aligned16(a.X);
    p.vec_w = (a.K % 4 == 0) && (a.ldw % 4 == 0) && aligned16(a.W);
    p.vec_p = (a.pro != VITA_PRO_NONE) && (a.K % 4 == 0) && (a.ld_pro_mul % 4 == 0) && (a.ld_pro_add % 4 == 0) &&
              aligned16(a.pro_mul) && aligned16(a.pro_add);

    // Tile choice: 128x128 when that still yields >= 2 tiles per CU, else 64x128 so that a
    // 4096 x 512 output still spreads over all 256 CUs.
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (t128 >= 512) return launch_cfg<128, 128, 32>(p, stream);
    return launch_cfg<64, 128, 32>(p, stream);
}

}  // namespace vita
