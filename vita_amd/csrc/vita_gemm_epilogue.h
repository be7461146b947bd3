// Shared GEMM epilogue: accumulators -> LDS (transposed to row-major) -> 16-byte coalesced global stores.
//
// A 32x32 MFMA accumulator gives a lane ONE column and 16 rows, so storing from registers means 4-byte
// stores of 128-byte row segments: measured 16.7k cycles for a 128x128 fp32 tile (store-issue bound,
// ~4 B/clk/CU), 18 % of the tile's time.  The stage buffers are dead once the k-loop ends, so the tile
// is bounced through them: each wave writes its accumulators (ds_write_b32, conflict-free), and after one
// barrier every thread reads float4s of whole rows (ds_read_b128, conflict-free with the +4 pad) and issues
// global_store_dwordx4 covering 512 contiguous bytes per row.  Bias / activation / gate / residual /
// pre-activation output / activation-derivative input are applied on the float4 side, with 16-byte loads.
//
//   z = acc + bias ; [preact = z] ; v = act_grad_z ? z * act'(zin) : act(z) ; v *= gate ; v = resid + v ; Y = v
#pragma once
#include <type_traits>

#include "vita_common.h"

#ifndef VITA_EPI_STAMP
#define VITA_EPI_STAMP(i)
#endif

namespace vita {

struct EpiParams {
    float* Y; int64_t ldy;
    int M, N;
    const float* bias;
    int act;
    const float* gate; int64_t ld_gate;       // vector (ld 0) or matrix
    int use_gate_scalar; float gate_scalar;
    const float* resid; int64_t ld_resid;
    float* preact; int64_t ld_preact;
    const float* act_grad_z; int64_t ld_act_grad_z;
    int vec;                                    // every pointer 16-byte aligned, every stride a multiple of 4
    // split-K: when `partial` is set the tile's raw accumulators go to partial[(split*M + row)*ld_partial + col] and
    // splitk_reduce_kernel applies the epilogue after summing the splits in a fixed order
    float* partial; int64_t ld_partial;
    // optional bf16 copy of Y (the next bf16 GEMM's A operand); Y itself may then be NULL
    uint16_t* ybf; int64_t ld_ybf;
    // ---- LayerNorm folded into the neighbouring GEMMs (sampler at large batch, vita_api.hip; interior tiles only) ----
    // producer side: besides Y = h, write aux_out = h * (aux_mul [+ 1]) (the next GEMM's A operand) and, per row and per
    // 128-column tile, (mean, M2) of h over the tile's columns into stat_out[(tile_n * M + row) * 2]
    float* aux_out; int64_t ld_aux; const float* aux_mul; int aux_add_one;
    uint16_t* aux_bf; int64_t ld_aux_bf;      // the same product as bf16 (the bf16-operand sampler's next A operand); either or both
    float* stat_out;
    // consumer side: acc <- rstd_row * (acc - mean_row * ln_c1[col]) before bias/activation, with (mean, rstd) of the row
    // combined from ln_nseg partials of ln_seg columns each (Chan's formula), rstd = 1/sqrt(M2/H + ln_eps)
    const float* ln_stats; const float* ln_c1; int ln_nseg; float ln_fold_eps;
};

// k-range of a split-K launch: blockIdx.x = split * tiles + tile; the split covers k in [split*k_per_split, ...)
struct SplitK {
    int splits = 1;
    int k_per_split = 0;      // multiple of 128 (so of every kernel's BK)
    float* partial = nullptr;
    int64_t ld_partial = 0;
    int* counters = nullptr;   // kernels that can: the last split to arrive at counters[tile] reduces and applies the epilogue (no second pass)
};

inline bool epi_aligned(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
inline int epi_vec_ok(const EpiParams& e) {
    auto ok = [](const void* q, int64_t ld) { return q == nullptr || (epi_aligned(q) && ld % 4 == 0); };
    return ok(e.Y, e.ldy) && ok(e.bias, 0) && ok(e.gate, e.ld_gate) && ok(e.resid, e.ld_resid) && ok(e.preact, e.ld_preact) &&
           ok(e.act_grad_z, e.ld_act_grad_z) && ok(e.ybf, e.ld_ybf);
}

// Folded LayerNorm, consumer side: (mean, rstd) of row `row` from the producer's per-segment (mean, M2) partials, Chan's
// combination of ln_nseg segments of `seg` columns each.
__device__ __forceinline__ void fold_row_stats(const EpiParams& e, int64_t row, int seg, float* out2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float* sp = e.ln_stats + row * 2;
    const int64_t seg_stride = (int64_t)e.M * 2;
    f32x2 part[16];   // every segment's (mean, M2) requested before the first use: one memory round trip (ln_nseg <= 16)
#pragma unroll
    for (int sg = 0; sg < 16; ++sg) {
        part[sg] = f32x2{0.f, 0.f};
        if (sg < e.ln_nseg) part[sg] = *reinterpret_cast<const f32x2*>(sp + sg * seg_stride);
    }
    float mu = 0.f;
#pragma unroll
    for (int sg = 0; sg < 16; ++sg) if (sg < e.ln_nseg) mu += part[sg][0];
    mu = mu / (float)e.ln_nseg;
    float m2 = 0.f;
#pragma unroll
    for (int sg = 0; sg < 16; ++sg) {
        if (sg < e.ln_nseg) {
            const float d = part[sg][0] - mu;
            m2 += part[sg][1] + (float)seg * (d * d);
        }
    }
    out2[0] = mu;
    out2[1] = 1.0f / sqrtf(m2 / (float)(e.ln_nseg * seg) + e.ln_fold_eps);
}

// sum over each 32-lane half of the wave, result in all of its lanes: four DPP steps inside the 16-lane rows (quad
// permutes, half mirror, mirror) and one cross-row exchange
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v + __shfl_xor(v, 16, 64);
}

template <int ACT, bool BWD>
__device__ __forceinline__ float epi_elem(float acc, float bias, float zin, float g, bool has_gate, float r, bool has_resid, float* pre) {
    const float z = acc + bias;
    if (pre) *pre = z;
    float v;
    if constexpr (BWD) v = z * act_grad(zin, ACT);
    else v = apply_act_t<ACT>(z);
    if (has_gate) v = v * g;
    if (has_resid) v = r + v;
    return v;
}

// obuf: at least BM * (BN + 4) floats of LDS that no wave reads any more (caller barriers before the call).
// KG > 1: the workgroup's waves form KG groups that each reduced a slice of every k-tile over the SAME output tile
// (in-workgroup split-K); group kg parks its partial tile in region kg of obuf and the row phase sums the regions in
// index order (deterministic).  obuf then needs KG * BM * (BN + 4) floats.
// 16-byte / 8-byte stores that write through to the fabric (`sc1`): the payload stores of an in-launch hand-off to other
// workgroups (MI355X: per-XCD L2s are not coherent with each other; a write-through store needs no L2 write-back release).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base) {
    // the whole 4 GiB window above `base`: the callers' offsets are validated on the host (every buffer is far smaller)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void store16_wt(__amdgpu_buffer_rsrc_t r, int64_t elem_off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)(elem_off * 4), 0, 16);
}
__device__ __forceinline__ void store8_wt(__amdgpu_buffer_rsrc_t r, int64_t elem_off, float a, float b) {
    u32x2_t v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(elem_off * 4), 0, 16);
}

__device__ __forceinline__ void store8_bf_wt(__amdgpu_buffer_rsrc_t r, int64_t elem_off, const u16x4& v) {   // 4 bf16
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, (int)(elem_off * 2), 0, 16);
}

template <int BM, int BN, int MI, int NI, int NT_, int ACT, bool BWD, int KG = 1, bool SC1 = false>
__device__ __forceinline__ void epilogue_tile(const f32x16 (&acc)[MI][NI], float* obuf, const EpiParams& e, int m0, int n0, int wm, int wn,
                                              int lane, int tid, int kg = 0, const float* srow = nullptr) {
    constexpr int S = BN + 4;
    constexpr int TM = MI * 32, TN = NI * 32;
    constexpr int C4 = BN / 4;
    constexpr int ITERS = BM * C4 / NT_;
    static_assert(BM * C4 % NT_ == 0, "tile / thread-count mismatch");
    // The residual (and, in the backward GEMMs, the saved pre-activation) rows this thread will combine with are
    // requested BEFORE the accumulators go through LDS, so their latency overlaps the LDS round trip and the barrier
    // instead of sitting in front of every store (fc2: epilogue 10.0k -> see DESIGN.md).
    constexpr bool PREFETCH = ITERS <= 8;
    f32x4 pre_r[PREFETCH ? ITERS : 1], pre_z[(PREFETCH && BWD) ? ITERS : 1];
    const bool pf = PREFETCH && e.vec && e.partial == nullptr;
    if (pf) {
#pragma unroll
        for (int it = 0; it < (PREFETCH ? ITERS : 0); ++it) {
            const int c = tid + it * NT_;
            const int row = m0 + c / C4, col = n0 + (c % C4) * 4;
            const bool ok = row < e.M && col + 3 < e.N;
            if (e.resid != nullptr && ok) pre_r[it] = *reinterpret_cast<const f32x4*>(e.resid + (int64_t)row * e.ld_resid + col);
            if constexpr (BWD) {
                if (ok) pre_z[it] = *reinterpret_cast<const f32x4*>(e.act_grad_z + (int64_t)row * e.ld_act_grad_z + col);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * TN + j * 32 + (lane & 31);
                obuf[kg * (BM * S) + row * S + col] = acc[i][j][r];
            }
    VITA_EPI_STAMP(6);
    __syncthreads();
    VITA_EPI_STAMP(7);
    auto tile_row4 = [&](int lr, int lc) -> f32x4 {
        f32x4 a = *reinterpret_cast<const f32x4*>(obuf + lr * S + lc);
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(obuf + g * (BM * S) + lr * S + lc);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += b[k];
        }
        return a;
    };
    const bool gate_vec = (e.gate != nullptr) && (e.ld_gate == 0);
    const bool gate_mat = (e.gate != nullptr) && (e.ld_gate != 0);
    const bool has_gate = e.use_gate_scalar || (e.gate != nullptr);
    const bool has_resid = e.resid != nullptr;
    if (e.partial != nullptr) {   // split-K: raw accumulators only (ld_partial is a multiple of 4, the buffer 16-byte aligned)
#pragma unroll 4
        for (int it = 0; it < ITERS; ++it) {
            const int c = tid + it * NT_;
            const int lr = c / C4, lc = (c % C4) * 4;
            const int row = m0 + lr, col = n0 + lc;
            if (row >= e.M || col >= e.N) continue;
            *reinterpret_cast<f32x4*>(e.partial + (int64_t)row * e.ld_partial + col) = tile_row4(lr, lc);
        }
        return;
    }
    // Interior tiles with 16-byte-aligned operands (every tile of the sampler's GEMMs): a thread keeps its column for all
    // ITERS rows, so the bias / gate vectors are loaded once and nothing is bounds-checked.
    if constexpr (NT_ % C4 == 0) {
        if (e.vec && m0 + BM <= e.M && n0 + BN <= e.N) {
            constexpr int RSTEP = NT_ / C4;
            const int lc = (tid % C4) * 4, lr0 = tid / C4;
            const int col = n0 + lc;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            f32x4 gv = {e.gate_scalar, e.gate_scalar, e.gate_scalar, e.gate_scalar};
            if (e.bias) b = *reinterpret_cast<const f32x4*>(e.bias + col);
            if (gate_vec) gv = *reinterpret_cast<const f32x4*>(e.gate + col);
            f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, am = c1;
            __amdgpu_buffer_rsrc_t rs_y, rs_aux, rs_stat, rs_auxb, rs_ybf;
            if constexpr (SC1) {
                rs_y = wt_rsrc(e.Y); rs_aux = wt_rsrc(e.aux_out); rs_stat = wt_rsrc(e.stat_out);
                rs_auxb = wt_rsrc(e.aux_bf); rs_ybf = wt_rsrc(e.ybf);
            }
            if (e.ln_stats) c1 = *reinterpret_cast<const f32x4*>(e.ln_c1 + col);
            const bool has_aux = e.aux_out != nullptr || e.aux_bf != nullptr;
            if (has_aux) {
                am = *reinterpret_cast<const f32x4*>(e.aux_mul + col);
                if (e.aux_add_one) am = am + 1.0f;
            }
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lr = lr0 + it * RSTEP;
                const int64_t row = m0 + lr;
                f32x4 a = tile_row4(lr, lc);
                if (e.ln_stats) {   // folded LayerNorm: finish the normalisation on the accumulators; srow = (mean, rstd)
                                    // of the tile's rows, combined by the kernel's prologue (fold_row_stats)
                    const float mu = srow[lr * 2], rstd = srow[lr * 2 + 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = rstd * fmaf(-mu, c1[k], a[k]);
                }
                f32x4 zin = {0.f, 0.f, 0.f, 0.f}, rs = zin, g = gv, pre, out;
                if constexpr (BWD) {
                    if (pf) zin = pre_z[PREFETCH ? it : 0];
                    else zin = *reinterpret_cast<const f32x4*>(e.act_grad_z + row * e.ld_act_grad_z + col);
                }
                if (gate_mat) g = *reinterpret_cast<const f32x4*>(e.gate + row * e.ld_gate + col);
                if (has_resid) {
                    if (pf) rs = pre_r[PREFETCH ? it : 0];
                    else rs = *reinterpret_cast<const f32x4*>(e.resid + row * e.ld_resid + col);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float p_ = 0.f;
                    out[k] = epi_elem<ACT, BWD>(a[k], b[k], zin[k], g[k], has_gate, rs[k], has_resid, e.preact ? &p_ : nullptr);
                    pre[k] = p_;
                }
                if (e.preact) *reinterpret_cast<f32x4*>(e.preact + row * e.ld_preact + col) = pre;
                if constexpr (SC1) {
                    if (e.Y) store16_wt(rs_y, row * e.ldy + col, out);
                } else {
                    if (e.Y) *reinterpret_cast<f32x4*>(e.Y + row * e.ldy + col) = out;
                }
                if (e.ybf) {
                    u16x4 hb;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hb[k] = f32_to_bf16(out[k]);
                    if constexpr (SC1) store8_bf_wt(rs_ybf, row * e.ld_ybf + col, hb);
                    else *reinterpret_cast<u16x4*>(e.ybf + row * e.ld_ybf + col) = hb;
                }
                if (has_aux) {
                    const f32x4 hm = out * am;
                    if (e.aux_out) {
                        if constexpr (SC1) store16_wt(rs_aux, row * e.ld_aux + col, hm);
                        else *reinterpret_cast<f32x4*>(e.aux_out + row * e.ld_aux + col) = hm;
                    }
                    if (e.aux_bf) {
                        u16x4 hb;
#pragma unroll
                        for (int k = 0; k < 4; ++k) hb[k] = f32_to_bf16(hm[k]);
                        if constexpr (SC1) store8_bf_wt(rs_auxb, row * e.ld_aux_bf + col, hb);
                        else *reinterpret_cast<u16x4*>(e.aux_bf + row * e.ld_aux_bf + col) = hb;
                    }
                    if constexpr (C4 == 32) {
                        // a row's BN = 128 outputs sit in 32 consecutive lanes.  (mean, M2) in ONE shuffle chain: sums of
                        // d = x - x0 and d^2 with x0 the row's first element (shifted one-pass: no cancellation against the mean)
                        const float xa = __builtin_amdgcn_readlane(out[0], 0), xb = __builtin_amdgcn_readlane(out[0], 32);
                        const float x0 = lane < 32 ? xa : xb;
                        float sm = 0.f, q = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const float d = out[k] - x0; sm += d; q = fmaf(d, d, q); }
                        sm = half_wave_sum(sm);
                        q = half_wave_sum(q);
                        if ((tid & 31) == 0) {
                            const int64_t so = ((int64_t)(n0 / BN) * e.M + row) * 2;
                            const float s0 = x0 + sm * (1.0f / (float)BN), s1 = q - sm * sm * (1.0f / (float)BN);
                            if constexpr (SC1) store8_wt(rs_stat, so, s0, s1);
                            else { e.stat_out[so] = s0; e.stat_out[so + 1] = s1; }
                        }
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = tid + it * NT_;
        const int lr = c / C4, lc = (c % C4) * 4;
        const int row = m0 + lr, col = n0 + lc;
        if (row >= e.M || col >= e.N) continue;
        const f32x4 a = tile_row4(lr, lc);
        if (e.vec && col + 3 < e.N) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f}, zin = b, g = b, rs = b, pre, out;
            if (e.bias) b = *reinterpret_cast<const f32x4*>(e.bias + col);
            if constexpr (BWD) {
                if (pf) zin = pre_z[PREFETCH ? it : 0];
                else zin = *reinterpret_cast<const f32x4*>(e.act_grad_z + (int64_t)row * e.ld_act_grad_z + col);
            }
            if (gate_vec) g = *reinterpret_cast<const f32x4*>(e.gate + col);
            else if (gate_mat) g = *reinterpret_cast<const f32x4*>(e.gate + (int64_t)row * e.ld_gate + col);
            else g = f32x4{e.gate_scalar, e.gate_scalar, e.gate_scalar, e.gate_scalar};
            if (has_resid) {
                if (pf) rs = pre_r[PREFETCH ? it : 0];
                else rs = *reinterpret_cast<const f32x4*>(e.resid + (int64_t)row * e.ld_resid + col);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float p_ = 0.f;
                out[k] = epi_elem<ACT, BWD>(a[k], b[k], zin[k], g[k], has_gate, rs[k], has_resid, e.preact ? &p_ : nullptr);
                pre[k] = p_;
            }
            if (e.preact) *reinterpret_cast<f32x4*>(e.preact + (int64_t)row * e.ld_preact + col) = pre;
            if (e.Y) *reinterpret_cast<f32x4*>(e.Y + (int64_t)row * e.ldy + col) = out;
            if (e.ybf) {
                u16x4 hb;
#pragma unroll
                for (int k = 0; k < 4; ++k) hb[k] = f32_to_bf16(out[k]);
                *reinterpret_cast<u16x4*>(e.ybf + (int64_t)row * e.ld_ybf + col) = hb;
            }
        } else {
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
                if (e.Y) e.Y[(int64_t)row * e.ldy + cc] = v;
                if (e.ybf) e.ybf[(int64_t)row * e.ld_ybf + cc] = f32_to_bf16(v);
            }
        }
    }
}

// run-time activation -> compile-time dispatch.  BACKWARD = false: forward GEMMs (act = the activation applied);
// BACKWARD = true: the transposed-operand GEMMs (act = the activation whose derivative is applied from act_grad_z).
template <int BM, int BN, int MI, int NI, int NT_, bool BACKWARD, int KG = 1, bool SC1 = false, int FACT = -1>
__device__ __forceinline__ void epilogue_dispatch(const f32x16 (&acc)[MI][NI], float* obuf, const EpiParams& e, int m0, int n0, int wm, int wn,
                                                  int lane, int tid, int kg = 0, const float* srow = nullptr) {
#define VITA_EPI(A) epilogue_tile<BM, BN, MI, NI, NT_, A, BACKWARD, KG, SC1>(acc, obuf, e, m0, n0, wm, wn, lane, tid, kg, srow)
    if constexpr (FACT >= 0) {      // activation fixed at compile time (persistent sampler): one instantiation, no split-K form
        epilogue_tile<BM, BN, MI, NI, NT_, FACT, BACKWARD, KG, SC1>(acc, obuf, e, m0, n0, wm, wn, lane, tid, kg, srow);
        return;
    }
    if (e.partial != nullptr || (BACKWARD && e.act_grad_z == nullptr)) {
        epilogue_tile<BM, BN, MI, NI, NT_, VITA_ACT_NONE, false, KG>(acc, obuf, e, m0, n0, wm, wn, lane, tid, kg);
        return;
    }
    switch (e.act) {
        case VITA_ACT_RELU: VITA_EPI(VITA_ACT_RELU); break;
        case VITA_ACT_GELU_ERF: VITA_EPI(VITA_ACT_GELU_ERF); break;
        case VITA_ACT_GELU_TANH: VITA_EPI(VITA_ACT_GELU_TANH); break;
        case VITA_ACT_SILU: VITA_EPI(VITA_ACT_SILU); break;
        case VITA_ACT_MISH: VITA_EPI(VITA_ACT_MISH); break;
        default: VITA_EPI(VITA_ACT_NONE); break;
    }
#undef VITA_EPI
}

}  // namespace vita
