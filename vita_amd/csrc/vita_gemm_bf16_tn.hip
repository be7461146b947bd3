// Weight-gradient GEMM of the bf16-operand training step on ROW-MAJOR operands:
//
//     C[m][n] = sum_r A[r][m] * B[r][n]        A = dY (R, M) bf16, B = X (R, N) bf16, C = dW (M, N) fp32
//
// (autograd of F.linear w.r.t. its weight, reference flow net: simple_flow_net.py:27-47; R = (1 + K) * B stacked rows).  Both
// operands are reduced over their ROW index, so the k-contiguous fragments v_mfma_f32_32x32x16_bf16 wants do not exist in
// memory.  Round 2/3 made them with a transposed pack launch per operand (pack_bf16_t: 0.9 of the 10.7 ms step at B = 4096);
// here the tiles go to LDS as they lie in HBM -- [32 r-rows][128 columns] per operand and k-tile, 256-byte rows, one
// global_load_lds_dwordx4 per 4 rows -- and the fragments come out of LDS transposed by the hardware, ds_read_b64_tr_b16:
// per 16-lane group a 4-row x 16-column block, lane i receiving column i of the 4 rows, i.e. 4 consecutive k of its m.
// Two such reads (rows +0..3, +4..7) are one lane's 8-deep operand of the 32x32x16 MFMA.
//
// LDS image: plain 256-byte rows with the 16-byte chunks XOR-swizzled, chunk ch of row r at 256 r + 16 (ch ^ f(r)),
// f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the DMA writes are linear (lane l of an instruction lands at base + 16 l), so the
// lane at chunk POSITION p of row r fetches global chunk p ^ f(r); the transposed reads then spread the 4 rows of a block
// over different 64-byte bank groups and are conflict-free (MI355X guide, LDS transposed read).
//
// Workgroup: 256 threads = 4 waves as 2 x 2, each wave 64 x 64 of a 128 x 128 tile (4 accumulators of 16 registers);
// 4 stages of 16 KB, one barrier and a counted vmcnt per k-tile, as the row-operand LDS-DMA kernel; two workgroups per CU
// cover each other's barrier and LDS latency.  The reduction over R is split over workgroups (SplitK, multiples of 32 rows)
// and finished by the same deterministic second pass as every other split GEMM here.
#include <hip/hip_runtime.h>

#include "vita_common.h"
#include "vita_gemm_dma_tile.h"
#include "vita_host.h"

namespace vita {
namespace {

struct GemmParamsTN {
    const uint16_t* A; int64_t lda;
    const uint16_t* B; int64_t ldb;
    int M, N, R;
    int tiles_m, tiles_n;
    int sk_splits, sk_r;     // split over r: number of splits, rows per split (multiple of 32)
    EpiParams epi;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
struct S16x8 { s16x4 lo, hi; };

constexpr int TBM = 128, TBN = 128, TBR = 32, TNS = 4, TNT = 256;
constexpr int TSTAGE = (TBM + TBN) * TBR * 2;                 // bytes per stage
constexpr int TLDS = (TNS * TSTAGE > TBM * (TBN + 4) * 4) ? TNS * TSTAGE : TBM * (TBN + 4) * 4;

__device__ __forceinline__ int tn_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// ds_read_b64_tr_b16 as inline asm, with the waits written by hand.  Through the builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16)
// hipcc orders every such read behind ALL pending LDS-DMA of the wave -- an s_waitcnt vmcnt(0) in the middle of the k-loop, right
// after the refills were issued -- which serialises operand delivery and compute (measured: 98 us for the 60-GFLOP weight gradient
// against 59 us without the refills and 56 us with ONLY the refills).  The kernel's own counted vmcnt + barrier is what orders a
// stage's reads behind its DMA.  `addr` is an LDS byte address.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Frag { u32x2 lo, hi; };     // 8 bf16: one 32x32x16 operand
template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// wait until at most N LDS reads of this wave are outstanding; the fragments are in/out operands so that nothing that uses them
// (the MFMAs, a register copy the allocator might insert) is scheduled above the wait
template <int N>
__device__ __forceinline__ void tr_wait(Frag (&a)[2], Frag (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo), "+v"(b[1].hi)
                 : "n"(N)
                 : "memory");
}

// LAB (timing only, results wrong): 1 = no operand refills inside the loop, 2 = refills only (no fragment reads, no MFMAs)
template <int LAB>
__global__ __launch_bounds__(TNT, 2) void gemm_bf16_tn_kernel(GemmParamsTN p) {
    extern __shared__ __attribute__((aligned(256))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // workgroup ids go round-robin over the 8 XCDs: give each XCD a CONTIGUOUS range of (split, tile) ids, so that its L2
    // fetches one split's rows (with 8 splits: 1/8 of both operands) instead of all of them
    const int ntile = p.tiles_m * p.tiles_n;
    const int total = gridDim.x;
    const int bid = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
    const int split = bid / ntile, t = bid % ntile;
    const int m0 = (t / p.tiles_n) * TBM, n0 = (t % p.tiles_n) * TBN;
    const int rbase = split * p.sk_r;
    const int R = p.sk_splits > 1 ? min(p.R, rbase + p.sk_r) : p.R;
    const int nk = (R - rbase + TBR - 1) / TBR;

    // DMA: an instruction covers 4 rows x 16 chunks; wave w issues instructions w and w + 4 of each operand's 8
    constexpr int DPT = 4;
    const int drow = lane >> 4, dpos = lane & 15;
    const uint16_t* a_src[2];
    const uint16_t* b_src[2];
    int d_r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 4 * j) * 4 + drow;              // row within the k-tile
        const int ch = dpos ^ tn_f(r);                        // global chunk this lane fetches
        d_r[j] = r;
        const int ma = m0 + ch * 8, nb = n0 + ch * 8;
        a_src[j] = ma < p.M ? p.A + (int64_t)r * p.lda + ma : nullptr;     // (M, N multiples of 8: a chunk is inside or outside)
        b_src[j] = nb < p.N ? p.B + (int64_t)r * p.ldb + nb : nullptr;
    }
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_chunk);
    auto issue_tile = [&](int kt, int stage) {
        char* sa = lds + stage * TSTAGE;
        char* sb = sa + TBM * TBR * 2;
        const int r0 = rbase + kt * TBR;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = r0 + d_r[j] < R;
            const uint16_t* ga = (in && a_src[j] != nullptr) ? a_src[j] + (int64_t)r0 * p.lda : zero;
            const uint16_t* gb = (in && b_src[j] != nullptr) ? b_src[j] + (int64_t)r0 * p.ldb : zero;
            dma16(reinterpret_cast<const float*>(ga), reinterpret_cast<float*>(sa + (wave + 4 * j) * 1024));
            dma16(reinterpret_cast<const float*>(gb), reinterpret_cast<float*>(sb + (wave + 4 * j) * 1024));
        }
    };
#pragma unroll
    for (int s = 0; s < TNS - 1; ++s) issue_tile(s, s);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read addresses: group g = lane / 16 reads the block of rows 8 (g >> 1) + 4 h .. + 3 (h = 0, 1: the two halves
    // of the 8-deep operand; + 16 for the tile's second MFMA k-step), columns cb + 16 (g & 1) .. + 15; lane 4 q + pp of the group
    // supplies row q, columns 4 pp .. 4 pp + 3 of the block
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int a_off[2][2], b_off[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * (g >> 1) + 4 * h + q;
            const int f = tn_f(row);
            const int ca = (wm * 64 + i * 32 + 16 * (g & 1)) / 8 + (pp >> 1);
            const int cb = (wn * 64 + i * 32 + 16 * (g & 1)) / 8 + (pp >> 1);
            a_off[i][h] = 256 * row + 16 * (ca ^ f) + 8 * (pp & 1);
            b_off[i][h] = TBM * TBR * 2 + 256 * row + 16 * (cb ^ f) + 8 * (pp & 1);
        }

    // fragments of one MFMA k-step (16 rows): 2 A + 2 B operands of 8 bf16, two transposed reads each (8 reads, issued in this order)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    auto read_step0 = [&](int stage, Frag (&af)[2], Frag (&bf)[2]) {
        const unsigned sb = lds_base + stage * TSTAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            tr_issue<0>(af[i].lo, sb + a_off[i][0]);
            tr_issue<0>(af[i].hi, sb + a_off[i][1]);
            tr_issue<0>(bf[i].lo, sb + b_off[i][0]);
            tr_issue<0>(bf[i].hi, sb + b_off[i][1]);
        }
    };
    auto read_step1 = [&](int stage, Frag (&af)[2], Frag (&bf)[2]) {
        const unsigned sb = lds_base + stage * TSTAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            tr_issue<4096>(af[i].lo, sb + a_off[i][0]);
            tr_issue<4096>(af[i].hi, sb + a_off[i][1]);
            tr_issue<4096>(bf[i].lo, sb + b_off[i][0]);
            tr_issue<4096>(bf[i].hi, sb + b_off[i][1]);
        }
    };
    auto mfma_step = [&](const Frag (&af)[2], const Frag (&bf)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
    };

    wait_vmcnt<(TNS - 3) * DPT>();
    __syncthreads();
    // software pipeline: the fragments of k-step s + 1 are requested before the MFMAs of k-step s, across k-tiles too -- after
    // the barrier of iteration kt tile kt + 1 has landed for everyone (at most one tile is outstanding per wave), so its first
    // fragments may be read while tile kt's second k-step multiplies.  LDS reads return in order: with the next step's 8 reads
    // behind them, `lgkmcnt(8)` means this step's 8 have arrived.
    Frag af0[2], bf0[2], af1[2], bf1[2];
    read_step0(0, af0, bf0);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        int stage_new = stage + TNS - 1;
        stage_new = stage_new >= TNS ? stage_new - TNS : stage_new;
        const int stage1 = stage + 1 == TNS ? 0 : stage + 1;
        if (kt > 0) {
            wait_vmcnt<(TNS - 3) * DPT>();        // this wave's share of tile kt + 1 has landed ...
            __builtin_amdgcn_s_barrier();         // ... everyone's has, and everyone is done reading tile kt - 1
            asm volatile("" ::: "memory");
        }
        if constexpr (LAB != 1) issue_tile(kt + TNS - 1, stage_new);      // past the last tile: zero chunks (r >= R)
        if constexpr (LAB == 2) { stage = stage1; continue; }
        read_step1(stage, af1, bf1);
        tr_wait<8>(af0, bf0);
        mfma_step(af0, bf0);
        read_step0(stage1, af0, bf0);      // (past the last tile: a zero-filled or stale stage, never multiplied)
        tr_wait<8>(af1, bf1);
        mfma_step(af1, bf1);
        stage = stage1;
    }
    tr_wait<0>(af0, bf0);     // the last prefetch: nothing of this wave may be in flight when the stages are reused
    wait_vmcnt<0>();
    __syncthreads();      // every wave is done reading fragments, every DMA has landed: the stages become the output tile
    EpiParams e = p.epi;
    if (e.partial != nullptr) e.partial += (int64_t)split * p.M * e.ld_partial;
    epilogue_dispatch<TBM, TBN, 2, 2, TNT, false>(acc, smem, e, m0, n0, wm, wn, lane, tid);
}


// (Tried and dropped, round 3: ONE 8-wave workgroup per CU with waves 4-7 only issuing the refills and waves 0-3 only multiplying,
// 6 stages: 85 us against 78 us for this kernel on the same box -- the refills are not held up by the waves that issue them, the
// delivery rate itself is the bound.)
}  // namespace

bool gemm_bf16_tn_eligible(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int R) {
    auto al = [](const void* q) { return q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return al(A) && al(B) && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && M > 0 && N > 0 && R > 0 && M % 8 == 0 && N % 8 == 0;
}

// a: Y / ldy / M / N / K (= R) and the epilogue options of the fp32 call (a.X / a.W are not read); sk: the split over r, or null
int gemm_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const vita_linear_args& a, hipStream_t stream, const SplitK* sk) {
    VITA_REQUIRE(gemm_bf16_tn_eligible(A, lda, B, ldb, a.M, a.N, a.K), "gemm_bf16_tn: operands must be 16-byte aligned bf16 rows, M and N multiples of 8 (M=%d N=%d R=%d)",
                 a.M, a.N, a.K);
    VITA_REQUIRE(a.Y != nullptr && a.ldy >= a.N, "gemm_bf16_tn: bad output");
    GemmParamsTN p{};
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = a.M; p.N = a.N; p.R = a.K;
    p.tiles_m = (a.M + TBM - 1) / TBM; p.tiles_n = (a.N + TBN - 1) / TBN;
    p.epi.Y = a.Y; p.epi.ldy = a.ldy; p.epi.M = a.M; p.epi.N = a.N; p.epi.bias = a.bias; p.epi.act = a.act;
    p.epi.gate = a.gate; p.epi.ld_gate = a.ld_gate; p.epi.use_gate_scalar = a.use_gate_scalar;
    p.epi.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    p.epi.resid = a.resid; p.epi.ld_resid = a.ld_resid; p.epi.preact = a.preact; p.epi.ld_preact = a.ld_preact;
    p.epi.vec = epi_vec_ok(p.epi);
    int splits = 1;
    if (sk && sk->splits > 1) {
        VITA_REQUIRE(sk->k_per_split % TBR == 0 && sk->partial != nullptr, "gemm_bf16_tn: rows per split must be a multiple of %d", TBR);
        splits = sk->splits; p.sk_splits = splits; p.sk_r = sk->k_per_split; p.epi.partial = sk->partial; p.epi.ld_partial = sk->ld_partial;
    }
    const long blocks = (long)p.tiles_m * p.tiles_n * splits;
    static const int lab = [] { const char* e = getenv("VITA_TN_LAB"); return e ? atoi(e) : 0; }();
    if (lab == 1) hipLaunchKernelGGL(gemm_bf16_tn_kernel<1>, dim3((unsigned)blocks), dim3(TNT), TLDS, stream, p);
    else if (lab == 2) hipLaunchKernelGGL(gemm_bf16_tn_kernel<2>, dim3((unsigned)blocks), dim3(TNT), TLDS, stream, p);
    else hipLaunchKernelGGL(gemm_bf16_tn_kernel<0>, dim3((unsigned)blocks), dim3(TNT), TLDS, stream, p);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita
