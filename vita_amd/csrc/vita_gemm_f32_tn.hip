// fp32 weight-gradient GEMM on ROW-MAJOR operands, the reference's arithmetic (fp32 MFMA, fp32 accumulate):
//
//     C[m][n] = sum_r A[r][m] * B[r][n]        A = dY (R, M), B = X (R, N), C = dW (M, N)
//
// (autograd of F.linear w.r.t. its weight, simple_flow_net.py:27-47 / action_ae.py:226-279; R = the rows stacked over every
// use of the weight in the step).  Rounds 1-2 ran this through the register-staged transposed-operand kernel
// (vita_gemm_f32_t.hip); measured on the flow net's stacked weight gradients at B = 4096 (28672 rows, 2048 x 512, 60 GFLOP):
// 530 us there, 506 us here = 0.755 of the fp32 MFMA peak (profiles/r03_train_b4096_f32_kernel_stats.csv); the step as a
// whole 22.14 -> 21.80 ms.  Below 2048 rows the register-staged kernel stays (B = 128: 2.83 ms against 2.89 ms with this one).
// v_mfma_f32_32x32x2_f32 takes ONE value per
// lane and operand -- lane l supplies A[k = l / 32][m = l % 32] -- so operands stored [r][m] need no transpose at all: the
// tiles go to LDS as they lie in HBM ([32 r][128 columns], 512-byte rows, one global_load_lds_dwordx4 per 2 rows) and a
// fragment is one ds_read_b32 whose 32-lane halves read 128 contiguous bytes of rows k and k + 1: conflict-free, no swizzle.
//
// Workgroup: 512 threads = 8 waves as 2 x 4, each wave 64 x 32 of a 128 x 128 tile (as the row-operand LDS-DMA kernel's
// 128 x 128 tile, whose epilogue it shares); 4 stages of 32 KB, one barrier and a counted vmcnt per k-tile.  The reduction
// over R is split over workgroups and finished by the deterministic second pass.
#include <hip/hip_runtime.h>

#include "vita_common.h"
#include "vita_gemm_dma_tile.h"
#include "vita_host.h"

namespace vita {
namespace {

struct GemmParamsTNF {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    int M, N, R;
    int tiles_m, tiles_n;
    int sk_splits, sk_r;     // split over r: number of splits, rows per split (multiple of 32)
    EpiParams epi;
};

constexpr int FBM = 128, FBN = 128, FBR = 32, FNS = 4;
constexpr int FSTAGE = (FBM + FBN) * FBR;                     // floats per stage
static_assert(FBM * (FBN + 4) <= FNS * FSTAGE, "the epilogue's output tile must fit the stage buffers");

__global__ __launch_bounds__(DNT, 1) void gemm_f32_tn_kernel(GemmParamsTNF p) {
    extern __shared__ __attribute__((aligned(256))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // (split, tile) ids in contiguous ranges per XCD, as in vita_gemm_bf16_tn.hip
    const int ntile = p.tiles_m * p.tiles_n;
    const int total = gridDim.x;
    const int bid = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
    const int split = bid / ntile, t = bid % ntile;
    const int m0 = (t / p.tiles_n) * FBM, n0 = (t % p.tiles_n) * FBN;
    const int rbase = split * p.sk_r;
    const int R = p.sk_splits > 1 ? min(p.R, rbase + p.sk_r) : p.R;
    const int nk = (R - rbase + FBR - 1) / FBR;

    // DMA: an instruction covers 2 rows x 32 chunks of 16 bytes; wave w issues instructions w and w + 8 of each operand's 16
    constexpr int DPT = 4;
    const int drow = lane >> 5, dch = lane & 31;
    const float* a_src[2];
    const float* b_src[2];
    int d_r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 2 + drow;              // row within the k-tile
        d_r[j] = r;
        const int ma = m0 + dch * 4, nb = n0 + dch * 4;
        a_src[j] = ma < p.M ? p.A + (int64_t)r * p.lda + ma : nullptr;     // (M, N multiples of 4: a chunk is inside or outside)
        b_src[j] = nb < p.N ? p.B + (int64_t)r * p.ldb + nb : nullptr;
    }
    auto issue_tile = [&](int kt, int stage) {
        float* sa = smem + stage * FSTAGE;
        float* sb = sa + FBM * FBR;
        const int r0 = rbase + kt * FBR;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = r0 + d_r[j] < R;
            const float* ga = (in && a_src[j] != nullptr) ? a_src[j] + (int64_t)r0 * p.lda : g_zero_chunk;
            const float* gb = (in && b_src[j] != nullptr) ? b_src[j] + (int64_t)r0 * p.ldb : g_zero_chunk;
            dma16(ga, sa + (wave + 8 * j) * 256);
            dma16(gb, sb + (wave + 8 * j) * 256);
        }
    };
#pragma unroll
    for (int s = 0; s < FNS - 1; ++s) issue_tile(s, s);

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    // fragment addresses (floats within a stage): k-step ks reads rows 2 ks + lane / 32
    const int fr = lane & 31, fh = lane >> 5;
    const int a_off = fh * FBM + wm * 64 + fr;
    const int b_off = FBM * FBR + fh * FBN + wn * 32 + fr;

    wait_vmcnt<(FNS - 3) * DPT>();
    __syncthreads();
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        int stage_new = stage + FNS - 1;
        stage_new = stage_new >= FNS ? stage_new - FNS : stage_new;
        if (kt > 0) {
            wait_vmcnt<(FNS - 3) * DPT>();        // this wave's share of tile kt has landed ...
            __builtin_amdgcn_s_barrier();         // ... everyone's has, and everyone is done reading tile kt - 1
            asm volatile("" ::: "memory");
        }
        issue_tile(kt + FNS - 1, stage_new);      // past the last tile: zero chunks (r >= R)
        const float* st = smem + stage * FSTAGE;
        float a0[2], a1[2], b[2];
        a0[0] = st[a_off]; a1[0] = st[a_off + 32]; b[0] = st[b_off];
#pragma unroll
        for (int ks = 0; ks < FBR / 2; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < FBR / 2) {
                a0[nxt] = st[a_off + (ks + 1) * 2 * FBM];
                a1[nxt] = st[a_off + (ks + 1) * 2 * FBM + 32];
                b[nxt] = st[b_off + (ks + 1) * 2 * FBN];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b[cur], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b[cur], acc[1][0], 0, 0, 0);
        }
        stage = stage + 1 == FNS ? 0 : stage + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();      // every wave is done reading fragments, every DMA has landed: the stages become the output tile
    EpiParams e = p.epi;
    if (e.partial != nullptr) e.partial += (int64_t)split * p.M * e.ld_partial;
    epilogue_dispatch<FBM, FBN, 2, 1, DNT, false>(acc, smem, e, m0, n0, wm, wn, lane, tid);
}

}  // namespace

bool gemm_f32_tn_eligible(const vita_linear_args& a) {
    auto al = [](const void* q) { return q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return a.a_col && a.b_col && al(a.X) && al(a.W) && a.ldx % 4 == 0 && a.ldw % 4 == 0 && a.ldx >= a.M && a.ldw >= a.N && a.M > 0 && a.N > 0 &&
           a.K > 0 && a.M % 4 == 0 && a.N % 4 == 0 && a.pro == VITA_PRO_NONE && a.gate == nullptr && a.preact == nullptr &&
           a.act_grad_z == nullptr && a.act == VITA_ACT_NONE;
}

// a = wgrad(...): X = dY (R, M), W = X (R, N), K = R; sk: the split over r, or null
int gemm_f32_tn(const vita_linear_args& a, hipStream_t stream, const SplitK* sk) {
    VITA_REQUIRE(gemm_f32_tn_eligible(a), "gemm_f32_tn: the call is not eligible (M=%d N=%d R=%d)", a.M, a.N, a.K);
    VITA_REQUIRE(a.Y != nullptr && a.ldy >= a.N, "gemm_f32_tn: bad output");
    GemmParamsTNF p{};
    p.A = a.X; p.lda = a.ldx; p.B = a.W; p.ldb = a.ldw; p.M = a.M; p.N = a.N; p.R = a.K;
    p.tiles_m = (a.M + FBM - 1) / FBM; p.tiles_n = (a.N + FBN - 1) / FBN;
    p.epi.Y = a.Y; p.epi.ldy = a.ldy; p.epi.M = a.M; p.epi.N = a.N; p.epi.bias = a.bias; p.epi.act = VITA_ACT_NONE;
    p.epi.use_gate_scalar = a.use_gate_scalar; p.epi.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    p.epi.resid = a.resid; p.epi.ld_resid = a.ld_resid;
    p.epi.vec = epi_vec_ok(p.epi);
    int splits = 1;
    if (sk && sk->splits > 1) {
        VITA_REQUIRE(sk->k_per_split % FBR == 0 && sk->partial != nullptr, "gemm_f32_tn: rows per split must be a multiple of %d", FBR);
        splits = sk->splits; p.sk_splits = splits; p.sk_r = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial;
    }
    const long blocks = (long)p.tiles_m * p.tiles_n * splits;
    hipLaunchKernelGGL(gemm_f32_tn_kernel, dim3((unsigned)blocks), dim3(DNT), FNS * FSTAGE * sizeof(float), stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita
