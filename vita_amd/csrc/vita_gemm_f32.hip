// fp32 MFMA GEMM with fused prologue / epilogue — the workhorse of the VITA hot path.
//
//   Y[M,N] = resid + gate * act( pro(X)[M,K] · W[N,K]^T + bias )
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fmaf chain (no TF32/xf32 on
// gfx950), so the result differs from the reference's fp32 GEMM only by summation order.
//
// Tiling (one workgroup = 512 threads = 8 waves, 2 per SIMD):
//   block tile BM x BN, waves laid out 2 (M) x 4 (N), wave tile (BM/2) x (BN/4) made of
//   32x32 MFMA accumulators; K is walked in BK-deep tiles, double-buffered in LDS:
//     global -> registers (issued before the MFMAs of the current tile)
//            -> [LayerNorm/modulation prologue on the A rows] -> LDS (after the MFMAs)
//   one barrier per K tile.  LDS rows are padded to BK+4 floats: the ds_read_b128 fragment
//   reads (32 rows x 16 B per half-wave) and the ds_write_b128 staging writes are then
//   bank-conflict-free (row stride = 4*odd dwords).
//   A ds_read_b128 gives a lane 4 consecutive k of its row; lanes 0-31 take k = 8c..8c+3,
//   lanes 32-63 k = 8c+4..8c+7, so register j of the A and B fragments pair k = 8c+j with
//   8c+4+j in one 32x32x2 MFMA: 4 MFMAs per pair of 16-byte LDS reads.
#include "vita_common.h"

namespace vita {

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
    int tiles_m, tiles_n;
    int vec_x, vec_w, vec_p;  // 16-byte vector loads legal for X / W / pro vectors
};

constexpr int NT = 512;

__device__ __forceinline__ f32x4 load4(const float* __restrict__ row, int k, int K, bool row_ok, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
        if (vec) {
            if (k < K) v = *reinterpret_cast<const f32x4*>(row + k);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < K) v[j] = row[k + j];
        }
    }
    return v;
}

template <int BM, int BN, int BK, bool LN>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmParams p) {
    constexpr int S = BK + 4;             // padded LDS row stride (floats)
    constexpr int C4 = BK / 4;            // float4 slots per tile row
    constexpr int A_SLOTS = BM * C4 / NT;
    constexpr int B_SLOTS = BN * C4 / NT;
    constexpr int TM = BM / 2, TN = BN / 4;
    constexpr int MI = TM / 32, NI = TN / 32;
    static_assert(A_SLOTS >= 1 && B_SLOTS >= 1 && MI >= 1 && NI >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM][S]
    float* Bs = smem + 2 * BM * S;         // [2][BN][S]

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
    bool a_ok[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int s = tid + i * NT;
        a_row[i] = s / C4;
        a_c[i] = (s % C4) * 4;
        a_ok[i] = (m0 + a_row[i]) < p.M;
        a_ptr[i] = p.X + (int64_t)(m0 + a_row[i]) * p.ldx;
    }
    int b_row[B_SLOTS], b_c[B_SLOTS];
    const float* b_ptr[B_SLOTS];
    bool b_ok[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int s = tid + i * NT;
        b_row[i] = s / C4;
        b_c[i] = (s % C4) * 4;
        b_ok[i] = (n0 + b_row[i]) < p.N;
        b_ptr[i] = p.W + (int64_t)(n0 + b_row[i]) * p.ldw;
    }

    // ---- LayerNorm statistics of this block's rows (two-pass, over the whole K extent) ----
    float mean[A_SLOTS], rstd[A_SLOTS];
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            float s = 0.f;
            for (int kt = 0; kt < nk; ++kt) {
                f32x4 v = load4(a_ptr[i], kt * BK + a_c[i], K, a_ok[i], p.vec_x);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
#pragma unroll
            for (int o = C4 / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mu = s / (float)K;
            float q = 0.f;
            for (int kt = 0; kt < nk; ++kt) {
                const int k = kt * BK + a_c[i];
                f32x4 v = load4(a_ptr[i], k, K, a_ok[i], p.vec_x);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = (k + j < K) ? v[j] - mu : 0.f;
                    q = fmaf(d, d, q);
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
            a_reg[i] = load4(a_ptr[i], k, K, a_ok[i], p.vec_x);
            if constexpr (LN) {
                const int64_t r = m0 + a_row[i];
                pm_reg[i] = load4(p.pro_mul + r * p.ld_pro_mul, k, K, a_ok[i], p.vec_p);
                pa_reg[i] = load4(p.pro_add + r * p.ld_pro_add, k, K, a_ok[i], p.vec_p);
            }
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            b_reg[i] = load4(b_ptr[i], kt * BK + b_c[i], K, b_ok[i], p.vec_w);
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

    issue_loads(0);
    write_lds(0);
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_loads(kt + 1);
        const float* Ab = &As[(buf * BM + wm * TM + frag_row) * S + frag_k];
        const float* Bb = &Bs[(buf * BN + wn * TN + frag_row) * S + frag_k];
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * S + kc * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * S + kc * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) write_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, activation, gate, residual --------------------------------------
    const int act = p.act;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * TN + j * 32 + (lane & 31);
        const bool col_ok = col < p.N;
        const float bias = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
        float gcol = p.use_gate_scalar ? p.gate_scalar : 1.0f;
        const bool gate_vec = (p.gate != nullptr) && (p.ld_gate == 0);
        const bool gate_mat = (p.gate != nullptr) && (p.ld_gate != 0);
        if (gate_vec && col_ok) gcol = p.gate[col];
        const bool has_gate = p.use_gate_scalar || (p.gate != nullptr);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M && col_ok) {
                    float v = acc[i][j][r] + bias;
                    v = apply_act(v, act);
                    if (has_gate) {
                        const float g = gate_mat ? p.gate[(int64_t)row * p.ld_gate + col] : gcol;
                        v = v * g;
                    }
                    if (p.resid != nullptr) v = p.resid[(int64_t)row * p.ld_resid + col] + v;
                    p.Y[(int64_t)row * p.ldy + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int BK>
static int launch_cfg(const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(NT);
    if (p.pro != VITA_PRO_NONE)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, false>), grid, block, lds, stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int gemm_f32(const vita_linear_args& a, hipStream_t stream) {
    if (a.M == 0 && a.N > 0 && a.K > 0) return VITA_OK;  // empty batch: nothing to do (pointers may be null)
    VITA_REQUIRE(a.X && a.W && a.Y, "vita_linear_f32: X, W and Y must be non-null");
    VITA_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "vita_linear_f32: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
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
    GemmParams p{};
    p.X = a.X; p.ldx = a.ldx; p.W = a.W; p.ldw = a.ldw; p.bias = a.bias;
    p.Y = a.Y; p.ldy = a.ldy; p.M = a.M; p.N = a.N; p.K = a.K;
    p.act = a.act; p.pro = a.pro; p.ln_eps = a.ln_eps;
    p.pro_mul = a.pro_mul; p.ld_pro_mul = a.ld_pro_mul;
    p.pro_add = a.pro_add; p.ld_pro_add = a.ld_pro_add;
    p.gate = a.gate; p.ld_gate = a.ld_gate;
    p.use_gate_scalar = a.use_gate_scalar; p.gate_scalar = a.gate_scalar;
    p.resid = a.resid; p.ld_resid = a.ld_resid;
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
