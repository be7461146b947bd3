// fp32 MFMA GEMM with transposed ("column") operands — the backward passes of nn.Linear.
//
//   C[M,N] = resid + gate_scalar * act'(Z[M,N]) * ( sum_k A(m,k) * B(n,k) + bias )      (act' only when act_grad_z is given)
//     a_col = 0 : A(m,k) = X[m*ldx + k]        a_col = 1 : A(m,k) = X[k*ldx + m]
//     b_col = 0 : B(n,k) = W[n*ldw + k]        b_col = 1 : B(n,k) = W[k*ldw + n]
//
//   dgrad  dX[B,in]   = dY[B,out] . W[out,in]          -> a_col=0, b_col=1, K=out
//   wgrad  dW[out,in] += dY[B,out]^T . X[B,in]         -> a_col=1, b_col=1, K=B, resid = Y = dW
//
// Same skeleton as vita_gemm_f32.hip (512 threads, 2x4 waves of 32x32 MFMA tiles, ring of three
// LDS buffers, fragments one chunk ahead).  A column operand is staged as [BK][rows+4]: the
// global rows are contiguous along m (or n), so the staging loads stay 16-byte coalesced, and
// the fragment read becomes four ds_read_b32 (32 consecutive dwords per half-wave:
// conflict-free) with the same k pairing (8c + 4h + j) the row operand's ds_read_b128 gives.
#include <type_traits>

#include "vita_gemm_epilogue.h"

namespace vita {

struct GemmParamsT {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;
    float* Y; int64_t ldy;
    int M, N, K;
    int use_gate_scalar; float gate_scalar;
    const float* resid; int64_t ld_resid;
    const float* act_grad_z; int64_t ld_act_grad_z; int act;   // epilogue: *= act'(z[row,col])
    int tiles_m, tiles_n;
    int sk_splits, sk_k;
    EpiParams epi;
};

constexpr int TNT = 512;
__device__ __attribute__((aligned(16))) float g_zero_t[4] = {0.f, 0.f, 0.f, 0.f};

// Stage descriptor of one operand tile: ROWS output rows (m or n) x BK reduction elements.
template <int ROWS, int BK, bool COL>
struct OpTile {
    static constexpr int S = COL ? ROWS + 4 : BK + 4;                  // LDS row stride (floats)
    static constexpr int BUF = COL ? BK * S : ROWS * S;                // floats per buffer
    static constexpr int PER_ROW = COL ? ROWS / 4 : BK / 4;            // float4 slots per LDS row
    static constexpr int SLOTS = ROWS * BK / 4 / TNT;
};

template <int BM, int BN, int BK, bool ACOL, bool BCOL, bool VEC>
__device__ __forceinline__ void gemm_f32_t_body(const GemmParamsT& p, const int block) {
    using TA = OpTile<BM, BK, ACOL>;
    using TB = OpTile<BN, BK, BCOL>;
    constexpr int TM = BM / 2, TN = BN / 4;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int NCH = BK / 8;
    static_assert(TA::SLOTS >= 1 && TB::SLOTS >= 1 && MI >= 1 && NI >= 1 && NCH % 2 == 0, "bad tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [3][TA::BUF]
    float* Bs = smem + 3 * TA::BUF;     // [3][TB::BUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const int split = p.sk_splits > 1 ? block / nwg : 0;
    int bid = block - split * nwg;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbase = split * p.sk_k;
    const int K = p.sk_splits > 1 ? min(p.K, kbase + p.sk_k) : p.K;
    const int nk = (K - kbase + BK - 1) / BK;

    // slot geometry: (r, c) = LDS row / float offset in the row
    auto slot_rc = [&](int s, int per_row, int& r, int& c) { r = s / per_row; c = (s % per_row) * 4; };

    f32x4 a_reg[TA::SLOTS], b_reg[TB::SLOTS];
    auto load_tile = [&](auto tile_tag, const float* base, int64_t ld, int o0, int lim, int kt, auto& regs) {
        using T = decltype(tile_tag);
        constexpr bool COL = std::is_same<T, TA>::value ? ACOL : BCOL;
#pragma unroll
        for (int i = 0; i < T::SLOTS; ++i) {
            int r, c;
            slot_rc(tid + i * TNT, T::PER_ROW, r, c);
            f32x4 v;
            if constexpr (COL) {      // LDS row r = k, c = offset along the output dim
                const int k = kbase + kt * BK + r, o = o0 + c;
                const float* src = base + (int64_t)k * ld + o;
                if constexpr (VEC) {
                    v = *reinterpret_cast<const f32x4*>((k < K && o < lim) ? src : g_zero_t);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = *((k < K && o + j < lim) ? src + j : g_zero_t);
                }
            } else {                  // LDS row r = output row, c = k offset
                const int k = kbase + kt * BK + c;
                const float* src = base + (int64_t)min(o0 + r, lim - 1) * ld + k;
                if constexpr (VEC) {
                    v = *reinterpret_cast<const f32x4*>(k < K ? src : g_zero_t);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = *((k + j < K) ? src + j : g_zero_t);
                }
            }
            regs[i] = v;
        }
    };
    auto store_tile = [&](auto tile_tag, float* lds, const auto& regs) {
        using T = decltype(tile_tag);
#pragma unroll
        for (int i = 0; i < T::SLOTS; ++i) {
            int r, c;
            slot_rc(tid + i * TNT, T::PER_ROW, r, c);
            *reinterpret_cast<f32x4*>(lds + r * T::S + c) = regs[i];
        }
    };
    auto issue_loads = [&](int kt) {
        load_tile(TA{}, p.X, p.ldx, m0, p.M, kt, a_reg);
        load_tile(TB{}, p.W, p.ldw, n0, p.N, kt, b_reg);
    };
    auto write_lds = [&](int buf) {
        store_tile(TA{}, As + buf * TA::BUF, a_reg);
        store_tile(TB{}, Bs + buf * TB::BUF, b_reg);
    };

    const int fr = lane & 31, fh = lane >> 5;
    f32x4 af[2][MI], bf[2][NI];
    auto read_frags = [&](int buf, int kc, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
        const float* A = As + buf * TA::BUF;
        const float* B = Bs + buf * TB::BUF;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * TM + i * 32 + fr;
            if constexpr (ACOL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[i][e] = A[(kc * 8 + fh * 4 + e) * TA::S + row];
            } else {
                a[i] = *reinterpret_cast<const f32x4*>(A + row * TA::S + kc * 8 + fh * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = wn * TN + j * 32 + fr;
            if constexpr (BCOL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[j][e] = B[(kc * 8 + fh * 4 + e) * TB::S + row];
            } else {
                b[j] = *reinterpret_cast<const f32x4*>(B + row * TB::S + kc * 8 + fh * 4);
            }
        }
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
    issue_loads(1);   // beyond K every element comes from the zero page
    write_lds(1);
    issue_loads(2);
    __syncthreads();
    read_frags(0, 0, af[0], bf[0]);

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf1 = buf == 2 ? 0 : buf + 1;
        const int buf2 = buf1 == 2 ? 0 : buf1 + 1;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            if (kc == 1) write_lds(buf2);
            if (kc == 2) issue_loads(kt + 3);
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
        __syncthreads();
        buf = buf1;
    }

    // the k-loop ended on a barrier: the staging buffers are dead (vita_gemm_epilogue.h)
    EpiParams e = p.epi;
    if (e.partial != nullptr) e.partial += (int64_t)split * p.M * e.ld_partial;
    epilogue_dispatch<BM, BN, MI, NI, TNT, true>(acc, smem, e, m0, n0, wm, wn, lane, tid);
}

template <int BM, int BN, int BK, bool ACOL, bool BCOL, bool VEC>
__global__ __launch_bounds__(TNT) void gemm_f32_t_kernel(GemmParamsT p) {
    gemm_f32_t_body<BM, BN, BK, ACOL, BCOL, VEC>(p, (int)blockIdx.x);
}

// Several independent problems in one launch (the weight gradients of a layer, or of a whole decoder, at few stacked rows: each
// alone has too few tiles for the chip and would split K and pay a second pass): workgroups [start[i], start[i+1]) take problem i.
constexpr int kBatchT = 8;
struct GemmParamsTB { GemmParamsT p[kBatchT]; int start[kBatchT + 1]; int n; };
template <int BM, int BN, int BK, bool ACOL, bool BCOL, bool VEC>
__global__ __launch_bounds__(TNT) void gemm_f32_t_batch_kernel(GemmParamsTB q) {
    int which = 0;
#pragma unroll
    for (int i = 1; i < kBatchT; ++i)
        if (i < q.n && (int)blockIdx.x >= q.start[i]) which = i;
    gemm_f32_t_body<BM, BN, BK, ACOL, BCOL, VEC>(q.p[which], (int)blockIdx.x - q.start[which]);
}

template <bool ACOL, bool BCOL>
static int launch_t(GemmParamsT p, bool vec, hipStream_t stream) {
    constexpr int BM = 64, BN = 128, BK = 32;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const size_t lds = (size_t)3 * (OpTile<BM, BK, ACOL>::BUF + OpTile<BN, BK, BCOL>::BUF) * sizeof(float);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (p.sk_splits > 1 ? p.sk_splits : 1))), block(TNT);
    if (vec) hipLaunchKernelGGL((gemm_f32_t_kernel<BM, BN, BK, ACOL, BCOL, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((gemm_f32_t_kernel<BM, BN, BK, ACOL, BCOL, false>), grid, block, lds, stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

static GemmParamsT make_params_t(const vita_linear_args& a, const SplitK* sk, bool* vec_out);

int gemm_f32_t_check(const vita_linear_args& a) {
    const char* who = "vita_linear_f32 (transposed operands)";
    VITA_REQUIRE(a.X && a.W && a.Y, "%s: X, W and Y must be non-null", who);
    VITA_REQUIRE(a.M > 0 && a.N > 0 && a.K >= 0, "%s: bad shape M=%d N=%d K=%d", who, a.M, a.N, a.K);
    VITA_REQUIRE(a.ldx >= (a.a_col ? a.M : a.K) && a.ldw >= (a.b_col ? a.N : a.K) && a.ldy >= a.N, "%s: leading dimension too small", who);
    VITA_REQUIRE(a.pro == VITA_PRO_NONE && a.gate == nullptr && a.preact == nullptr,
                 "%s: prologue / vector gate / preact are not available with a_col or b_col", who);
    VITA_REQUIRE(a.act >= VITA_ACT_NONE && a.act <= VITA_ACT_MISH, "%s: unknown activation %d", who, a.act);
    VITA_REQUIRE(a.act == VITA_ACT_NONE || a.act_grad_z != nullptr, "%s: with a_col/b_col `act` names the derivative applied from act_grad_z, which is null", who);
    VITA_REQUIRE(a.act_grad_z == nullptr || a.ld_act_grad_z >= a.N, "%s: ld_act_grad_z must be >= N", who);
    VITA_REQUIRE(a.resid == nullptr || a.ld_resid >= a.N, "%s: ld_resid must be >= N", who);
    VITA_REQUIRE((const void*)a.Y != (const void*)a.X && (const void*)a.Y != (const void*)a.W, "%s: Y must not alias X or W", who);
    return VITA_OK;
}

int gemm_f32_t(const vita_linear_args& a, hipStream_t stream, const SplitK* sk) {
    const char* who = "vita_linear_f32 (transposed operands)";
    {
        const int rc = gemm_f32_t_check(a);
        if (rc != VITA_OK) return rc;
    }
    if (a.K == 0 && a.resid == nullptr) {   // empty reduction: the result is the bias (wgrad of an empty batch is zero)
        VITA_REQUIRE(a.bias == nullptr, "%s: K == 0 with a bias is not supported", who);
        VITA_CHECK_HIP(hipMemset2DAsync(a.Y, (size_t)a.ldy * 4, 0, (size_t)a.N * 4, (size_t)a.M, stream));
        return VITA_OK;
    }
    bool vec = false;
    const GemmParamsT p = make_params_t(a, sk, &vec);
    if (a.a_col && a.b_col) return launch_t<true, true>(p, vec, stream);
    if (a.a_col) return launch_t<true, false>(p, vec, stream);
    return launch_t<false, true>(p, vec, stream);
}

// dW_i = dY_i^T X_i, i < n <= 8, in ONE launch, no split-K (each problem's tiles reduce their whole K)
int gemm_f32_t_batch(const vita_linear_args* a, int n, hipStream_t stream) {
    const char* who = "vita_linear_f32 (batched weight gradients)";
    VITA_REQUIRE(a != nullptr && n >= 1 && n <= kBatchT, "%s: 1 to %d problems", who, kBatchT);
    constexpr int BM = 64, BN = 128, BK = 32;
    GemmParamsTB q{};
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        VITA_REQUIRE(a[i].a_col && a[i].b_col && a[i].K > 0, "%s: every call must be a weight gradient over at least one row", who);
        { const int rc = gemm_f32_t_check(a[i]); if (rc != VITA_OK) return rc; }
        bool v = false;
        q.p[i] = make_params_t(a[i], nullptr, &v);
        vec = vec && v;
        q.p[i].tiles_m = (q.p[i].M + BM - 1) / BM; q.p[i].tiles_n = (q.p[i].N + BN - 1) / BN;
        q.start[i + 1] = q.start[i] + q.p[i].tiles_m * q.p[i].tiles_n;
    }
    q.n = n;
    const size_t lds = (size_t)3 * (OpTile<BM, BK, true>::BUF + OpTile<BN, BK, true>::BUF) * sizeof(float);
    const dim3 grid((unsigned)q.start[n]), block(TNT);
    if (vec) hipLaunchKernelGGL((gemm_f32_t_batch_kernel<BM, BN, BK, true, true, true>), grid, block, lds, stream, q);
    else hipLaunchKernelGGL((gemm_f32_t_batch_kernel<BM, BN, BK, true, true, false>), grid, block, lds, stream, q);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

static GemmParamsT make_params_t(const vita_linear_args& a, const SplitK* sk, bool* vec_out) {
    GemmParamsT p{};
    p.X = a.X; p.ldx = a.ldx; p.W = a.W; p.ldw = a.ldw; p.bias = a.bias; p.Y = a.Y; p.ldy = a.ldy;
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.use_gate_scalar = a.use_gate_scalar; p.gate_scalar = a.gate_scalar;
    p.resid = a.resid; p.ld_resid = a.ld_resid;
    p.act_grad_z = a.act_grad_z; p.ld_act_grad_z = a.ld_act_grad_z; p.act = a.act;
    p.epi.Y = a.Y; p.epi.ldy = a.ldy; p.epi.M = a.M; p.epi.N = a.N; p.epi.bias = a.bias; p.epi.act = a.act;
    p.epi.gate = a.gate; p.epi.ld_gate = a.ld_gate; p.epi.use_gate_scalar = a.use_gate_scalar;
    p.epi.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    p.epi.resid = a.resid; p.epi.ld_resid = a.ld_resid; p.epi.preact = a.preact; p.epi.ld_preact = a.ld_preact;
    p.epi.act_grad_z = a.act_grad_z; p.epi.ld_act_grad_z = a.ld_act_grad_z;
    p.epi.vec = epi_vec_ok(p.epi);
    if (sk && sk->splits > 1) { p.sk_splits = sk->splits; p.sk_k = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial; }
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool vx = al(a.X) && a.ldx % 4 == 0 && (a.a_col ? a.M % 4 == 0 : a.K % 4 == 0);
    const bool vw = al(a.W) && a.ldw % 4 == 0 && (a.b_col ? a.N % 4 == 0 : a.K % 4 == 0);
    *vec_out = vx && vw;
    return p;
}

}  // namespace vita
