// fp32 MFMA GEMM for few-row problems (M <= 512: the B = 128 training step, serving batches, the time path).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),   A(m,k) = X[m*ldx + k],   B(n,k) = W[n*ldw + k]  (b_col: W[k*ldw + n])
//
// At these sizes the LDS-DMA kernel's fixed cost (three-stage prologue, LDS-transposing epilogue: ~20 k cycles) is most of
// its run time, and filling the chip needs a split-K second pass.  Here one workgroup owns a 32 x 32 output tile and its NW
// waves split the reduction: wave w reduces k in [w*kw, (w+1)*kw) with operands loaded global -> VGPR directly in MFMA
// fragment layout (lane = row | 32*half reads the 16 bytes k = 8c + 4*half .. +3 of its row: the k pairing of the other
// kernels), every load of a batch of chunks issued before the first MFMA; the NW partial tiles are summed through LDS in
// wave order (deterministic) and 256 threads apply the epilogue on row-major float4s.  No LDS staging, no pipeline, and the
// in-workgroup reduction replaces the split-K pass for K <= 1024 (above that a grid-level split feeds the usual
// splitk_reduce_kernel).  A column operand (dgrad: B(n,k) = W[k][n]) is read with four coalesced dword loads per chunk.
#include <stdlib.h>

#include "vita_gemm_dma_tile.h"
#include "vita_splitk_fused.h"

namespace vita {

struct SmallParams {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    int M, N, K;
    int tiles_n;
    int kw;                  // k elements per wave (multiple of 8)
    int sk_splits, sk_k;     // grid-level split-K (k elements per split, multiple of 8)
    int* sk_counters;        // non-null: one zeroed counter per output tile; the last split to arrive finishes the tile (vita_splitk_fused.h)
    // LayerNorm prologue of the 8-wave LDS kernel (K = the whole row, 8 x 64): A' = ((A - mean) * rstd) * m + add, m = mul (+ 1 for LN_MOD)
    int pro_mod;             // 1: LN_MOD (m = 1 + mul), 0: LN_AFFINE
    float ln_eps;
    const float* pro_mul;    // (K): one modulation row for all rows of the call
    const float* pro_add;
    EpiParams epi;
};

constexpr int SM_TS = 36;    // LDS row stride of a partial tile (floats): 16-byte aligned rows

template <bool BCOL, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_f32_small_kernel(SmallParams p) {
    constexpr int CB = 16;   // chunks (of 8 k) loaded per batch
    __shared__ __attribute__((aligned(16))) float red[NW * 32 * SM_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int ntile = p.tiles_n * ((p.M + 31) / 32);
    const int split = p.sk_splits > 1 ? (int)blockIdx.x / ntile : 0;
    const int bid = (int)blockIdx.x - split * ntile;
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int kbase = split * p.sk_k;
    const int Kend = p.sk_splits > 1 ? min(p.K, kbase + p.sk_k) : p.K;
    const int k0 = kbase + wave * p.kw;
    const int k1 = min(Kend, k0 + p.kw);
    const int nch = k1 > k0 ? (k1 - k0 + 7) / 8 : 0;

    const float* arow = p.X + (int64_t)min(m0 + fr, p.M - 1) * p.ldx;
    const int ncol = min(n0 + fr, p.N - 1);
    const float* brow = BCOL ? p.W + ncol : p.W + (int64_t)ncol * p.ldw;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // epilogue operands of this thread's output float4, requested before the reduction so their latency hides behind it
    const int lr = (tid & 255) >> 3, lc = (tid & 7) * 4;
    const int row = m0 + lr, col = n0 + lc;
    const EpiParams& e = p.epi;
    const bool bwd = e.act_grad_z != nullptr;
    const bool gate_vec = (e.gate != nullptr) && (e.ld_gate == 0);
    const bool gate_mat = (e.gate != nullptr) && (e.ld_gate != 0);
    const bool has_gate = e.use_gate_scalar || (e.gate != nullptr);
    const bool fused = e.partial != nullptr && p.sk_counters != nullptr;
    const bool fast = tid < 256 && (e.partial == nullptr || fused) && e.vec && row < e.M && col + 3 < e.N;
    f32x4 pbias = {0.f, 0.f, 0.f, 0.f}, pzin = pbias, prs = pbias, pg = {e.gate_scalar, e.gate_scalar, e.gate_scalar, e.gate_scalar};
    if (fast) {
        if (e.bias) pbias = *reinterpret_cast<const f32x4*>(e.bias + col);
        if (bwd) pzin = *reinterpret_cast<const f32x4*>(e.act_grad_z + (int64_t)row * e.ld_act_grad_z + col);
        if (gate_vec) pg = *reinterpret_cast<const f32x4*>(e.gate + col);
        else if (gate_mat) pg = *reinterpret_cast<const f32x4*>(e.gate + (int64_t)row * e.ld_gate + col);
        if (e.resid) prs = *reinterpret_cast<const f32x4*>(e.resid + (int64_t)row * e.ld_resid + col);
    }

    for (int c0 = 0; c0 < nch; c0 += CB) {
        f32x4 a[CB], b[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int kk = k0 + (c0 + c) * 8 + fh * 4;
            const bool ok = (c0 + c) < nch && kk < k1;       // K % 4 == 0 and kk % 4 == 0: the four k exist together
            a[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            b[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) {
                a[c] = *reinterpret_cast<const f32x4*>(arow + kk);
                if constexpr (BCOL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[c][e] = brow[(int64_t)(kk + e) * p.ldw];
                } else {
                    b[c] = *reinterpret_cast<const f32x4*>(brow + kk);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (c0 + c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[c][e], acc, 0, 0, 0);
            }
        }
    }

    // partial tile of this wave -> LDS (accumulator register r of a lane: row (r&3) + 8*(r>>2) + 4*half, column lane&31)
    float* mine = red + wave * (32 * SM_TS);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * fh) * SM_TS + fr] = acc[r];
    __syncthreads();
    if (tid >= 256 && !fused) return;
    const bool inb = tid < 256 && row < e.M && col < e.N;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (tid < 256) {
        s = *reinterpret_cast<const f32x4*>(red + lr * SM_TS + lc);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(red + w * (32 * SM_TS) + lr * SM_TS + lc);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += t[k];
        }
    }
    if (fused) {                    // grid-level split-K finished here by the last workgroup of the tile
        __shared__ int last_flag;
        const unsigned split_bytes = (unsigned)e.M * (unsigned)e.ld_partial * 4u;
        const __amdgpu_buffer_rsrc_t rp = skf_rsrc(e.partial, split_bytes * (unsigned)p.sk_splits);
        const unsigned poff = ((unsigned)row * (unsigned)e.ld_partial + (unsigned)col) * 4u;
        if (inb) skf_store16(rp, (unsigned)split * split_bytes + poff, s);      // ld_partial % 4 == 0: the padded row takes the float4
        if (!skf_arrive_last(p.sk_counters + bid, p.sk_splits, &last_flag)) return;
        if (!inb) return;
        s = skf_sum(rp, poff, split_bytes, p.sk_splits);
    } else {
        if (!inb) return;
        if (e.partial != nullptr) {     // grid-level split-K: raw sums (ld_partial % 4 == 0, 16-byte aligned buffer)
            float* dst = e.partial + ((int64_t)split * e.M + row) * e.ld_partial + col;
            if (col + 3 < e.N) *reinterpret_cast<f32x4*>(dst) = s;
            else for (int k = 0; k < 4 && col + k < e.N; ++k) dst[k] = s[k];
            return;
        }
    }
    if (fast) {
        const f32x4 bias = pbias, zin = pzin, g = pg, rs = prs;
        f32x4 pre, out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z = s[k] + bias[k];
            pre[k] = z;
            float v = bwd ? z * act_grad(zin[k], e.act) : apply_act(z, e.act);
            if (has_gate) v = v * g[k];
            if (e.resid) v = rs[k] + v;
            out[k] = v;
        }
        if (e.preact) *reinterpret_cast<f32x4*>(e.preact + (int64_t)row * e.ld_preact + col) = pre;
        *reinterpret_cast<f32x4*>(e.Y + (int64_t)row * e.ldy + col) = out;
    } else {
        for (int k = 0; k < 4 && col + k < e.N; ++k) {
            const int cc = col + k;
            const float z = s[k] + (e.bias ? e.bias[cc] : 0.f);
            if (e.preact) e.preact[(int64_t)row * e.ld_preact + cc] = z;
            float v = bwd ? z * act_grad(e.act_grad_z[(int64_t)row * e.ld_act_grad_z + cc], e.act) : apply_act(z, e.act);
            const float g = gate_vec ? e.gate[cc] : (gate_mat ? e.gate[(int64_t)row * e.ld_gate + cc] : e.gate_scalar);
            if (has_gate) v = v * g;
            if (e.resid) v = e.resid[(int64_t)row * e.ld_resid + cc] + v;
            e.Y[(int64_t)row * e.ldy + cc] = v;
        }
    }
}

// The same tile with the operands staged through LDS (4 waves, k span <= 512).  In the kernel above adjacent lanes load
// adjacent ROWS (the MFMA fragment layout), so every 4 lanes touch 4 cache lines and the loads run at a third of the rate
// of a coalesced stream (measured in vita_ode_resident.hip: 28 against 70+ GB/s per CU); at M = 128 that is most of the
// kernel.  Here every wave copies ITS k-slice of the two 32-row operand tiles into a private 32 KB LDS region with
// global_load_lds (an instruction = 8 rows x 128 contiguous bytes; 16-byte chunks XOR-swizzled by (row >> 1) & 7 so that
// the fragment reads are conflict-free), waits for its own DMAs block by block (no workgroup barrier: the region is the
// wave's own) and reads the fragments back with ds_read_b128.  Same MFMAs in the same order on the same values as the
// kernel above: the same bits.  A column operand (BCOL) keeps its four coalesced dword loads per chunk.
// NW = 8 (k span 257..512: 64 k per wave, two blocks) halves the MFMA chain behind the DMA latency of a one-round launch.
// LN: the A rows are LayerNorm-ed (+ modulated) on the way from LDS to the MFMAs (NW = 8, K = 512 = the whole row, no grid
// split): every wave sums its 64-k slice per row, the 8 partial sums meet in LDS (two passes: mean, then centred squares, as
// ln_mod_kernel does), and each lane normalises its own fragments in registers.  Replaces a 4 us LayerNorm launch in front of
// every few-row fc1 / out_proj of the sampler (one modulation row for the batch) by ~0.4 us inside it.
template <bool BCOL, int NW, bool LN = false>
__global__ __launch_bounds__(NW * 64) void gemm_f32_small_lds_kernel(SmallParams p) {
    static_assert(!LN || (NW == 8 && !BCOL), "the LayerNorm prologue is built for the 8-wave row-operand kernel");
    constexpr int KB = 32, NBLK = 16 / NW;                    // 32-k blocks per wave: kw <= 128 (NW = 4) or 64 (NW = 8)
    constexpr int WREG = 2 * NBLK * 32 * KB;                  // floats of a wave's region
    extern __shared__ __attribute__((aligned(16))) float stage[];     // [NW][A: NBLK x 32 x 32 | B: NBLK x 32 x 32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntile = p.tiles_n * ((p.M + 31) / 32);
    const int split = p.sk_splits > 1 ? (int)blockIdx.x / ntile : 0;
    const int bid = (int)blockIdx.x - split * ntile;
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int kbase = split * p.sk_k;
    const int Kend = p.sk_splits > 1 ? min(p.K, kbase + p.sk_k) : p.K;
    const int k0 = kbase + wave * p.kw;
    const int k1 = min(Kend, k0 + p.kw);
    const int nch = k1 > k0 ? (k1 - k0 + 7) / 8 : 0;
    float* sa = stage + wave * WREG;
    float* sb = sa + NBLK * 32 * KB;

    // epilogue operands of this thread's output float4, requested first (they return first: the DMA waits below count DMAs only)
    const bool epi_thread = tid < 256;
    const int lr = (tid & 255) >> 3, lc = (tid & 7) * 4;
    const int row = m0 + lr, col = n0 + lc;
    const EpiParams& e = p.epi;
    const bool bwd = e.act_grad_z != nullptr;
    const bool gate_vec = (e.gate != nullptr) && (e.ld_gate == 0);
    const bool gate_mat = (e.gate != nullptr) && (e.ld_gate != 0);
    const bool has_gate = e.use_gate_scalar || (e.gate != nullptr);
    const bool fused = e.partial != nullptr && p.sk_counters != nullptr;
    const bool fast = epi_thread && (e.partial == nullptr || fused) && e.vec && row < e.M && col + 3 < e.N;
    f32x4 pbias = {0.f, 0.f, 0.f, 0.f}, pzin = pbias, prs = pbias, pg = {e.gate_scalar, e.gate_scalar, e.gate_scalar, e.gate_scalar};
    if (fast) {
        if (e.bias) pbias = *reinterpret_cast<const f32x4*>(e.bias + col);
        if (bwd) pzin = *reinterpret_cast<const f32x4*>(e.act_grad_z + (int64_t)row * e.ld_act_grad_z + col);
        if (gate_vec) pg = *reinterpret_cast<const f32x4*>(e.gate + col);
        else if (gate_mat) pg = *reinterpret_cast<const f32x4*>(e.gate + (int64_t)row * e.ld_gate + col);
        if (e.resid) prs = *reinterpret_cast<const f32x4*>(e.resid + (int64_t)row * e.ld_resid + col);
    }
    // (NW = 8: waves 4..7 issue no epilogue loads, so every wave's DMA counts below start from the same zero or are only ahead)
    // LayerNorm prologue: this lane's modulation fragments (k = k0 + 8 c + 4 fh .. +3 of row fr), requested before the DMAs too
    f32x4 lnm[LN ? 4 * NBLK : 1], lna[LN ? 4 * NBLK : 1];
    if constexpr (LN) {
        const float* pm = p.pro_mul + k0 + fh * 4;
        const float* pa = p.pro_add + k0 + fh * 4;
#pragma unroll
        for (int c = 0; c < 4 * NBLK; ++c) {
            lnm[c] = *reinterpret_cast<const f32x4*>(pm + c * 8);
            lna[c] = *reinterpret_cast<const f32x4*>(pa + c * 8);
        }
    }

    // ---- DMA sources: lane l of the instruction for rows 8 j .. 8 j + 7 writes row 8 j + l / 8, chunk position l % 8,
    //      so it fetches chunk (l % 8) ^ swz(row) of that row's 128-byte piece
    const int dr = lane >> 3, dslot = lane & 7;
    const float* asrc[4];
    const float* bsrc[4];
    int dchunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + dr;
        dchunk[j] = (dslot ^ swz<KB>(r)) * 4;
        asrc[j] = p.X + (int64_t)min(m0 + r, p.M - 1) * p.ldx;
        bsrc[j] = BCOL ? nullptr : p.W + (int64_t)min(n0 + r, p.N - 1) * p.ldw;
    }
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + KB * kb + dchunk[j];
            const bool ok = k < k1;                           // K % 4 == 0 and k % 4 == 0: the four k exist together
            dma16(ok ? asrc[j] + k : g_zero_chunk, sa + (kb * 32 + 8 * j) * KB);
            if constexpr (!BCOL) dma16(ok ? bsrc[j] + k : g_zero_chunk, sb + (kb * 32 + 8 * j) * KB);
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int fsw = swz<KB>(fr);
    if constexpr (BCOL) {
        // column operand: registers, as in the kernel above (coalesced across lanes); then everything has landed at once
        const int ncol = min(n0 + fr, p.N - 1);
        const float* brow = p.W + ncol;
        f32x4 b[4 * NBLK];
#pragma unroll
        for (int c = 0; c < 4 * NBLK; ++c) {
            const int kk = k0 + c * 8 + fh * 4;
            b[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nch && kk < k1) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) b[c][e4] = brow[(int64_t)(kk + e4) * p.ldw];
            }
        }
        wait_vmcnt<0>();
#pragma unroll
        for (int c = 0; c < 4 * NBLK; ++c) {
            if (c < nch) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sa + ((c >> 2) * 32 + fr) * KB + ((((c & 3) * 2 + fh) ^ fsw) * 4));
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e4], b[c][e4], acc, 0, 0, 0);
            }
        }
    } else {
        auto block = [&](int kb) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = kb * 4 + cc;
                if (c < nch) {
                    const int off = (kb * 32 + fr) * KB + (((cc * 2 + fh) ^ fsw) * 4);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sa + off);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sb + off);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e4], b[e4], acc, 0, 0, 0);
                }
            }
        };
        if constexpr (LN) {
            __shared__ float ln_part[2][NW][32];
            wait_vmcnt<0>();
            f32x4 af[4 * NBLK];
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 4 * NBLK; ++c) {
                af[c] = *reinterpret_cast<const f32x4*>(sa + ((c >> 2) * 32 + fr) * KB + ((((c & 3) * 2 + fh) ^ fsw) * 4));
                sm += (af[c][0] + af[c][1]) + (af[c][2] + af[c][3]);
            }
            sm += __shfl_xor(sm, 32, 64);
            if (fh == 0) ln_part[0][wave][fr] = sm;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += ln_part[0][w][fr];
            const float mu = tot / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 4 * NBLK; ++c) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) { const float d = af[c][e4] - mu; q = fmaf(d, d, q); }
            }
            q += __shfl_xor(q, 32, 64);
            if (fh == 0) ln_part[1][wave][fr] = q;
            __syncthreads();
            float qt = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) qt += ln_part[1][w][fr];
            const float rstd = 1.0f / sqrtf(qt / (float)p.K + p.ln_eps);
#pragma unroll
            for (int c = 0; c < 4 * NBLK; ++c) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float nrm = (af[c][e4] - mu) * rstd;
                    const float m = p.pro_mod ? (lnm[c][e4] + 1.0f) : lnm[c][e4];
                    af[c][e4] = nrm * m + lna[c][e4];
                }
            }
#pragma unroll
            for (int c = 0; c < 4 * NBLK; ++c) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(sb + ((c >> 2) * 32 + fr) * KB + ((((c & 3) * 2 + fh) ^ fsw) * 4));
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][e4], b[e4], acc, 0, 0, 0);
            }
        } else
        // the wave's DMAs complete in order: block kb is there once at most 8 * (NBLK - 1 - kb) of them are still out
        if constexpr (NBLK == 4) {
            wait_vmcnt<24>(); block(0);
            wait_vmcnt<16>(); block(1);
            wait_vmcnt<8>();  block(2);
            wait_vmcnt<0>();  block(3);
        } else {
            wait_vmcnt<8>();  block(0);
            wait_vmcnt<0>();  block(1);
        }
    }

    // the wave's partial tile goes into its own (now consumed) staging region; the tile is summed in wave order
    float* mine = sa;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * fh) * SM_TS + fr] = acc[r];
    __syncthreads();
    if (!epi_thread && !fused) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (epi_thread) {
        s = *reinterpret_cast<const f32x4*>(stage + lr * SM_TS + lc);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(stage + w * WREG + lr * SM_TS + lc);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += t[k];
        }
    }
    const bool inb = epi_thread && row < e.M && col < e.N;
    if (fused) {                    // grid-level split-K finished here by the last workgroup of the tile
        __shared__ int last_flag;
        const unsigned split_bytes = (unsigned)e.M * (unsigned)e.ld_partial * 4u;
        const __amdgpu_buffer_rsrc_t rp = skf_rsrc(e.partial, split_bytes * (unsigned)p.sk_splits);
        const unsigned poff = ((unsigned)row * (unsigned)e.ld_partial + (unsigned)col) * 4u;
        if (inb) skf_store16(rp, (unsigned)split * split_bytes + poff, s);      // ld_partial % 4 == 0: the padded row takes the float4
        if (!skf_arrive_last(p.sk_counters + bid, p.sk_splits, &last_flag)) return;
        if (!inb) return;
        s = skf_sum(rp, poff, split_bytes, p.sk_splits);
    } else {
        if (!inb) return;
        if (e.partial != nullptr) {     // grid-level split-K: raw sums (ld_partial % 4 == 0, 16-byte aligned buffer)
            float* dst = e.partial + ((int64_t)split * e.M + row) * e.ld_partial + col;
            if (col + 3 < e.N) *reinterpret_cast<f32x4*>(dst) = s;
            else for (int k = 0; k < 4 && col + k < e.N; ++k) dst[k] = s[k];
            return;
        }
    }
    if (fast) {
        const f32x4 bias = pbias, zin = pzin, g = pg, rs = prs;
        f32x4 pre, out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z = s[k] + bias[k];
            pre[k] = z;
            float v = bwd ? z * act_grad(zin[k], e.act) : apply_act(z, e.act);
            if (has_gate) v = v * g[k];
            if (e.resid) v = rs[k] + v;
            out[k] = v;
        }
        if (e.preact) *reinterpret_cast<f32x4*>(e.preact + (int64_t)row * e.ld_preact + col) = pre;
        *reinterpret_cast<f32x4*>(e.Y + (int64_t)row * e.ldy + col) = out;
    } else {
        for (int k = 0; k < 4 && col + k < e.N; ++k) {
            const int cc = col + k;
            const float z = s[k] + (e.bias ? e.bias[cc] : 0.f);
            if (e.preact) e.preact[(int64_t)row * e.ld_preact + cc] = z;
            float v = bwd ? z * act_grad(e.act_grad_z[(int64_t)row * e.ld_act_grad_z + cc], e.act) : apply_act(z, e.act);
            const float g = gate_vec ? e.gate[cc] : (gate_mat ? e.gate[(int64_t)row * e.ld_gate + cc] : e.gate_scalar);
            if (has_gate) v = v * g;
            if (e.resid) v = e.resid[(int64_t)row * e.ld_resid + cc] + v;
            e.Y[(int64_t)row * e.ldy + cc] = v;
        }
    }
}

// Shapes this kernel is meant for.  k_split = 0: the caller has no split-K scratch.
// The LayerNorm prologue this file builds: the 8-wave LDS kernel on a whole 512-long row (K = 8 x 64), row operands, no grid split.
bool gemm_f32_small_ln_eligible(const vita_linear_args& a) {
#ifdef VITA_NO_SMALL_GEMM
    return false;
#endif
    static const bool off = (getenv("VITA_SMALL_NO_LDS") != nullptr && atoi(getenv("VITA_SMALL_NO_LDS")) != 0) ||
                            (getenv("VITA_NO_SMALL_LN") != nullptr && atoi(getenv("VITA_NO_SMALL_LN")) != 0);
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (off || a.a_col || a.b_col || (a.pro != VITA_PRO_LN_MOD && a.pro != VITA_PRO_LN_AFFINE)) return false;
    if (a.M <= 0 || a.M > 512 || a.N < 32 || a.K != 512 || a.ldx % 4 != 0 || a.ldw % 4 != 0 || !al(a.X) || !al(a.W)) return false;
    if (!a.pro_mul || !a.pro_add || !al(a.pro_mul) || !al(a.pro_add) || a.ld_pro_mul % 4 != 0 || a.ld_pro_add % 4 != 0) return false;
    if (a.ld_pro_mul != 0 || a.ld_pro_add != 0) return false;     // one modulation row for the batch (the sampler); per-row tables measured no gain
    const double traffic = ((double)((a.M + 31) / 32) * a.N + (double)((a.N + 31) / 32) * a.M) * a.K * 4.0;
    return traffic <= 96e6;
}

bool gemm_f32_small_eligible(const vita_linear_args& a, bool have_splitk_ws) {
#ifdef VITA_NO_SMALL_GEMM
    return false;
#endif
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (a.pro != VITA_PRO_NONE) return gemm_f32_small_ln_eligible(a);
    if (a.a_col || a.M <= 0 || a.M > 512 || a.N < 32 || a.K < 64) return false;
    if (a.K % 4 != 0 || a.ldx % 4 != 0 || !al(a.X)) return false;
    if (!a.b_col && (a.ldw % 4 != 0 || !al(a.W))) return false;
    // one 32 x 32 tile per workgroup re-reads operands (M/32 + N/32) times: keep the L2 traffic of a launch bounded
    const double traffic = ((double)((a.M + 31) / 32) * a.N + (double)((a.N + 31) / 32) * a.M) * a.K * 4.0;
    if (traffic > 96e6) return false;
    if (a.K > 2048 && !have_splitk_ws) return false;
    return true;
}

// sk: grid-level split (k_per_split multiple of 8) or NULL.
int gemm_f32_small(const vita_linear_args& a, hipStream_t stream, const SplitK* sk) {
    SmallParams p{};
    p.X = a.X; p.ldx = a.ldx; p.W = a.W; p.ldw = a.ldw; p.M = a.M; p.N = a.N; p.K = a.K;
    p.tiles_n = (a.N + 31) / 32;
    EpiParams& e = p.epi;
    e.Y = a.Y; e.ldy = a.ldy; e.M = a.M; e.N = a.N; e.bias = a.bias; e.act = a.act;
    e.gate = a.gate; e.ld_gate = a.ld_gate; e.use_gate_scalar = a.use_gate_scalar; e.gate_scalar = a.use_gate_scalar ? a.gate_scalar : 1.0f;
    e.resid = a.resid; e.ld_resid = a.ld_resid; e.preact = a.preact; e.ld_preact = a.ld_preact;
    e.act_grad_z = a.act_grad_z; e.ld_act_grad_z = a.ld_act_grad_z;
    e.vec = epi_vec_ok(e);
    int kspan = a.K;
    if (sk && sk->splits > 1) {
        p.sk_splits = sk->splits; p.sk_k = sk->k_per_split; e.partial = sk->partial; e.ld_partial = sk->ld_partial;
        p.sk_counters = sk->counters;
        kspan = sk->k_per_split;
    }
#ifndef VITA_SMALL_NW8_MIN_K
#define VITA_SMALL_NW8_MIN_K 257     // k span from which 8 waves split the reduction (64 k each at 512)
#endif
    const int nw = kspan < VITA_SMALL_NW8_MIN_K ? 4 : 8;     // 16 waves of 32 k each: measured slower (B = 128 step 2.98 -> 3.00 ms)
    p.kw = ((kspan + nw - 1) / nw + 7) & ~7;
    const unsigned grid = (unsigned)(p.tiles_n * ((a.M + 31) / 32) * (p.sk_splits > 1 ? p.sk_splits : 1));
    static const bool no_lds = getenv("VITA_SMALL_NO_LDS") != nullptr && atoi(getenv("VITA_SMALL_NO_LDS")) != 0;
    if (a.pro != VITA_PRO_NONE) {       // LayerNorm prologue (gemm_f32_small_ln_eligible): 8 waves x 64 k, no grid split
        VITA_REQUIRE(gemm_f32_small_ln_eligible(a) && p.sk_splits <= 1 && nw == 8 && p.kw == 64, "few-row GEMM: LayerNorm prologue on an unsupported call");
        p.pro_mod = a.pro == VITA_PRO_LN_MOD ? 1 : 0; p.ln_eps = a.ln_eps;
        p.pro_mul = a.pro_mul; p.pro_add = a.pro_add;
        constexpr size_t lds_ln = (size_t)4 * 2 * 4 * 32 * 32 * sizeof(float);
        hipLaunchKernelGGL((gemm_f32_small_lds_kernel<false, 8, true>), dim3(grid), dim3(512), lds_ln, stream, p);
        VITA_CHECK_HIP(hipGetLastError());
        return VITA_OK;
    }
    if (p.kw <= 512 / nw && !no_lds) {      // operands through LDS (coalesced DMA): the same bits, faster loads
        constexpr size_t lds = (size_t)4 * 2 * 4 * 32 * 32 * sizeof(float);      // 128 KB either way: NW x NBLK = 16 blocks of 2 x 4 KB
        if (nw == 4) {
            if (a.b_col) hipLaunchKernelGGL((gemm_f32_small_lds_kernel<true, 4>), dim3(grid), dim3(256), lds, stream, p);
            else hipLaunchKernelGGL((gemm_f32_small_lds_kernel<false, 4>), dim3(grid), dim3(256), lds, stream, p);
        } else {
            if (a.b_col) hipLaunchKernelGGL((gemm_f32_small_lds_kernel<true, 8>), dim3(grid), dim3(512), lds, stream, p);
            else hipLaunchKernelGGL((gemm_f32_small_lds_kernel<false, 8>), dim3(grid), dim3(512), lds, stream, p);
        }
        VITA_CHECK_HIP(hipGetLastError());
        return VITA_OK;
    }
    if (a.b_col) {
        if (nw == 4) hipLaunchKernelGGL((gemm_f32_small_kernel<true, 4>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_f32_small_kernel<true, 8>), dim3(grid), dim3(512), 0, stream, p);
    } else {
        if (nw == 4) hipLaunchKernelGGL((gemm_f32_small_kernel<false, 4>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_f32_small_kernel<false, 8>), dim3(grid), dim3(512), 0, stream, p);
    }
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita
