// TorchFlowMatcher.sample (flow_matchers.py:41-79) at SERVING batches (B <= 64 rows, in halves of 32: eval_n_envs = 25 / 5,
// default_policy.yaml:48; the pusht batch of 64) as ONE launch whose velocity-field weights stay resident in LDS across all K Euler steps.
//
// At these batches a solve is ~60 dependent GEMMs of a few MFLOP each: the per-launch path is bound by kernel boundaries
// and by re-reading 34 MB of weights per step (0.89 ms per 25-row solve).  Here:
//   * the grid is 256 workgroups, one per CU; workgroup c owns the SAME output columns of every layer for the whole solve:
//     columns [2c, 2c+2) of input_proj / fc2 / out_proj (N = 512) and [8c, 8c+8) of fc1 (N = 2048).  Its slices of all
//     weight matrices (2*512 + L*(8*512 + 2*2048) + 2*512 floats = 136 KB at L = 4) are copied into LDS once and never
//     re-read: 256 x 136 KB = the whole 34 MB net, resident on chip for all K steps (simple_flow_net.py:39-48,99-108).
//   * a phase = one GEMM of the chain on all rows.  Every workgroup needs the whole A operand (B x K floats); it streams it
//     global -> registers in MFMA operand layout and multiplies with v_mfma_f32_4x4x1_16B.  Adjacent lanes of any MFMA operand
//     hold adjacent ROWS, which on a row-major matrix makes every 4 lanes touch 4 cache lines (measured: 28 GB/s per CU,
//     7 us for fc2's A); so the hand-off buffers (x, h, u) are kept in a layout of 64-byte blocks of 4 rows x 4 k,
//     [row / 4][k / 4][row % 4][k % 4], in which a wave's 16-byte loads cover whole 256-byte runs:
//     the 16 blocks of one instruction are 4 row-quads x 4 k-residues, a lane holds A[row][4 consecutive k] (one 16-byte
//     load covers it) and W[col][the same 4 k] (one ds_read_b128), the 8 waves split K, partial sums meet in LDS in a
//     fixed order (deterministic).
//   * the Euler state x and the residual stream h of a workgroup's own two columns never leave its registers
//     ("x += dt * v in-register"); what the other workgroups need is stored write-through for the next phase.
//   * LayerNorm (+ modulation) is applied by every workgroup on the A fragments it has just loaded (two-pass statistics
//     across the 8 waves through LDS).
//
// Hand-off (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", first table row): every handed-off
// byte is stored `sc1` (8- / 16-byte buffer stores), every storing wave waits vmcnt(0), a workgroup barrier, then ONE lane
// adds to its shard of an 8-shard counter (agent-scope atomic); the consumer's wave 0 polls EVERY shard with sc1 loads,
// joins a workgroup barrier, and EVERY load of handed-off bytes is a 16-byte `buffer_load ... sc1` to registers -- in that
// form (hipMalloc memory, one workgroup per CU) no L1 invalidate is needed.  With `acquire` set (memory of another kind:
// the allocation-free entry point's __device__ array; or VITA_RESIDENT_ACQUIRE=1) the polling wave also runs the agent-scope
// acquire.  Every spin is bounded; on a timeout the workgroup records the phase and leaves (vita_ode_sample_status).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "vita_common.h"
#include "vita_host.h"
#include "vita_ode_sync.h"

namespace vita {

constexpr int RS_D = 512, RS_FF = 2048;          // D = H = 512, FF = 2048: the shapes whose slices fill LDS
constexpr int RS_WG = 256, RS_NT = 512, RS_NW = RS_NT / 64, RS_ROWS = 32, RS_MAXB = 64;
constexpr int RS_KP1 = RS_D + 16, RS_KP2 = RS_FF + 16;   // LDS row strides: = 16 (mod 64) floats -> conflict-free ds_read_b128
constexpr int RS_C2 = RS_D / RS_WG, RS_C8 = RS_FF / RS_WG;
static_assert(RS_C2 == 2 && RS_C8 == 8, "slice widths");
constexpr int RS_RED = RS_NW * RS_ROWS * 8;      // per wave: 32 rows x 8 columns of partial sums
constexpr int RS_ST = 2 * RS_NW * RS_ROWS;       // LayerNorm partials: sums, then centred squares
constexpr size_t rs_weight_floats(int L) { return (size_t)4 * RS_KP1 + (size_t)L * (2 * RS_KP2 + 8 * RS_KP1); }
constexpr size_t rs_smem_floats(int L) { return rs_weight_floats(L) + RS_RED + RS_ST + 4; }

typedef unsigned int u32x4_r __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_r __attribute__((ext_vector_type(2)));

struct ResidentArgs {
    vita_flownet_weights w;
    const float* x0; int64_t ldx0;       // step 0 reads it
    float* x_out; int64_t ldo;           // every step writes it (in place from step 1 on) unless traj
    float* traj;                         // NULL or (K+1, B, D): step k reads slab k, writes slab k+1
    float *h, *u, *xi;                   // hand-off buffers h (B, H), u (B, FF), x (B, D), blocked: rows padded to a multiple of 4
    float* mod;                          // (K, L, 3H): gamma | scale | shift per layer and step (time path)
    int tp;                              // != 0: the time path runs in this launch too (K <= 32, TE = 256); scratch below, blocked
    float *thid, *tsilu;                 // (K, 4 TE), (K, H)
    unsigned* sync;
    int B, K;
    float dt;
    unsigned spin_limit;
    int acquire;
    // the action decoder behind the last step (SimpleActionDecoder.forward, action_ae.py:267-279): dec != 0
    int dec;
    vita_decoder_weights dw;
    float *da, *db;                      // blocked hand-off buffers (B, HD)
    float* actions; int64_t lda;         // (B, OUT) row-major
    // the observation encoder ahead of the first step (obs_encoder = nn.Linear(O, D), vita_policy.py:77,194): obs_w != NULL;
    // then x0 is not read
    const float *obs_w, *obs_b, *obs; int64_t ldobs; int O;
    unsigned long long* stamps;   // diagnostic: NULL, or [workgroup][phase][6] 100 MHz clocks: phase start, wait over, A landed, sums in LDS, published
};

// the consumer half of the grid-wide hand-off: all 8 shards must have counted `target` arrivals
struct GridSync {
    unsigned* sync;
    unsigned target;      // 0: nothing to wait for
    unsigned limit, code;
    int acquire;
    float* flag;
    __device__ __forceinline__ bool operator()() const {
        if (target == 0) return true;
        const int tid = threadIdx.x;
        if (tid < 64) {
            unsigned* tmo = sync + kMaxTeams * kTeamStride;
            unsigned ok = 1, spins = 0;
            for (;;) {
                unsigned v = target;
                if (tid < 8) v = __hip_atomic_load(sync + tid * kTeamStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v < target) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > limit || ((spins & 255u) == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = 0;
                    break;
                }
            }
            if (!ok && tid == 0) {
                __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + kStickyWord, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (acquire) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the loads below the poll
            }
            if (tid == 0) *reinterpret_cast<volatile unsigned*>(flag) = ok;
        }
        __syncthreads();
        return *reinterpret_cast<volatile unsigned*>(flag) != 0;
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 16-byte load that bypasses this CU's L1 (`sc1`); offsets past the descriptor's size return zeros (rows >= B)
__device__ __forceinline__ f32x4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ float load4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_r, v), r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void store8_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float a, float b) {
    u32x2_r v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)byte_off, 0, 16);
}

// float offset of element (row, k) in a blocked hand-off buffer of K columns: [row / 4][k / 4][row % 4][k % 4]
__device__ __forceinline__ unsigned rs_blk(int row, int k, int K) {
    return ((unsigned)(row >> 2) * (unsigned)(K >> 2) + (unsigned)(k >> 2)) * 16u + (unsigned)(row & 3) * 4u + (unsigned)(k & 3);
}

// sum over the four k-residue lanes of a 16-lane row (lane bits 2-3): the same bits in all four
__device__ __forceinline__ float kq_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
    return v;
}

// What one phase multiplies.
struct PhaseIn {
    __amdgpu_buffer_rsrc_t ra;            // A operand (ASRC 0: blocked hand-off buffer; 1: the caller's row-major x0)
    unsigned lda_bytes;                   // ASRC 1: row stride
    int row0;                             // first of the 32 rows this call multiplies (a multiple of 32: batches above 32 rows go in halves)
    int steps;                            // ASRC 2: A = the sinusoidal embedding of t_k = k / steps, row k < steps (pos_embed.py:11-18)
    const float* wl;                      // the weight slice in LDS, column c at wl + c * (KDIM + 16); GW: the (N, KDIM) matrix in global memory
    int gw_col0, gw_n;                    // GW: this workgroup's first column, and N (columns >= N read as zeros)
    __amdgpu_buffer_rsrc_t rln;           // LNM 1: the modulation table, (1 + mul) and add at these byte offsets (sc1 loads: the time
    unsigned ln_mul_off, ln_add_off;      //        path of this very launch may have written it)
    const float *ln_w, *ln_b;             // LNM 2: nn.LayerNorm's affine vectors (weights: plain loads)
};

// One GEMM of the chain on this workgroup's column slice: out[row][c] = sum_k A'[row][k] * W[c][k] for row < 32,
// c < 4 * CG, with A' = A (LNM = 0), LN(A) * (1 + mul) + add (LNM = 1, eps 1e-6) or LN(A) * w + b (LNM = 2, eps 1e-5).
//   KDIM : the reduction length (256, 512, 1024, 2048); each of the 8 waves owns KDIM / 8 consecutive k
//   GW   : the weights are read from global memory into registers BEFORE the hand-off wait -- they depend on nobody, so
//          their latency hides behind the wait; for matrices used once per solve (time path, decoder), which LDS has no room for
// On return red[w][row][8] holds wave w's partial sums (behind a workgroup barrier); false = the hand-off wait ran out.
template <int KDIM, int CG, int LNM, int ASRC, bool GW>
__device__ __forceinline__ bool rs_phase(const PhaseIn& in, float* red, float* st, const GridSync& sy, unsigned long long* stamp) {
    constexpr int KP = KDIM + 16;
    constexpr int NS = KDIM / 128;               // 16-k pieces of this wave's k range
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 3, kq = (lane >> 2) & 3, rq = lane >> 4;
    const int kb = w * (KDIM / 8) + 4 * kq;      // + 16 * s
    f32x4 lm[LNM ? NS : 1], la[LNM ? NS : 1];
    if constexpr (LNM != 0) {
        static_assert(LNM == 0 || KDIM == RS_D, "LayerNorm phases have K = H");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (LNM == 1) {
                lm[s] = load16_sc1(in.rln, in.ln_mul_off + (unsigned)(kb + 16 * s) * 4u) + 1.0f;
                la[s] = load16_sc1(in.rln, in.ln_add_off + (unsigned)(kb + 16 * s) * 4u);
            } else {
                lm[s] = *reinterpret_cast<const f32x4*>(in.ln_w + kb + 16 * s);
                la[s] = *reinterpret_cast<const f32x4*>(in.ln_b + kb + 16 * s);
            }
        }
    }
    f32x4 wreg[GW ? CG : 1][GW ? NS : 1];
    if constexpr (GW) {
        constexpr unsigned KB = (unsigned)KDIM * 4u;      // bytes per weight row
        const __amdgpu_buffer_rsrc_t rw = rs_rsrc(in.wl, (unsigned)in.gw_n * KB);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                wreg[cg][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                            rw, (int)((unsigned)(in.gw_col0 + 4 * cg + i) * KB + (unsigned)(kb + 16 * s) * 4u), 0, 0));
    }
    f32x4 a[2][NS];
    if constexpr (ASRC == 2) {       // sinusoidal embedding, the arithmetic of sinusoid_kernel (vita_elementwise.hip); waves 0-3 hold the
                                     // sines, 4-7 the cosines (KDIM = 2 * half, a wave's k range lies in one half)
        constexpr int half = KDIM / 2;
        const float emb = logf(10000.0f) / (float)(half - 1);
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            const int row = 16 * R + 4 * rq + i;
            const float t = (float)((double)row / (double)in.steps);      // flow_matchers.py:69
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a[R][s] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (16 * R < in.steps && row < in.steps) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = kb + 16 * s + r;
                        const float e = t * expf((float)(k & (half - 1)) * -emb);
                        a[R][s][r] = (w < RS_NW / 2) ? sinf(e) : cosf(e);
                    }
                }
            }
        }
    }
    if (!sy()) return false;
    if (stamp && tid == 0) stamp[1] = clock64_real();
    if constexpr (ASRC != 2) {
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            if constexpr (ASRC == 1) {
                const unsigned off = (unsigned)(in.row0 + 16 * R + 4 * rq + i) * in.lda_bytes + (unsigned)kb * 4u;
#pragma unroll
                for (int s = 0; s < NS; ++s) a[R][s] = load16_sc1(in.ra, off + 64u * s);
            } else {      // block (row group 4 R + rq, k chunk kb / 4 + 4 s): 64 bytes, this lane's row at + 16 i
                const unsigned off = ((unsigned)((in.row0 >> 2) + 4 * R + rq) * (unsigned)(KDIM / 4) + (unsigned)(kb >> 2)) * 64u + 16u * i;
#pragma unroll
                for (int s = 0; s < NS; ++s) a[R][s] = load16_sc1(in.ra, off + 256u * s);
            }
        }
    }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) stamp[2] = clock64_real();
    }
    if constexpr (LNM != 0) {
        // two-pass statistics; written on whole f32x4 values so that the packed fp32 instructions (v_pk_add / mul / fma_f32)
        // carry them
        constexpr float inv_n = 1.0f / (float)RS_D;
        const float eps = LNM == 1 ? 1e-6f : 1e-5f;
        float p[2];
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            f32x4 t4 = a[R][0];
#pragma unroll
            for (int s = 1; s < NS; ++s) t4 += a[R][s];
            p[R] = kq_sum((t4[0] + t4[1]) + (t4[2] + t4[3]));
        }
        if (kq == 0) {
            st[w * RS_ROWS + 4 * rq + i] = p[0];
            st[w * RS_ROWS + 16 + 4 * rq + i] = p[1];
        }
        __syncthreads();
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < RS_NW; ++ww) t += st[ww * RS_ROWS + 16 * R + 4 * rq + i];
            const float mean = t * inv_n;
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a[R][s] -= mean;
                q4 = __builtin_elementwise_fma(a[R][s], a[R][s], q4);
            }
            p[R] = kq_sum((q4[0] + q4[1]) + (q4[2] + q4[3]));
        }
        float* st2 = st + RS_NW * RS_ROWS;
        if (kq == 0) {
            st2[w * RS_ROWS + 4 * rq + i] = p[0];
            st2[w * RS_ROWS + 16 + 4 * rq + i] = p[1];
        }
        __syncthreads();
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < RS_NW; ++ww) t += st2[ww * RS_ROWS + 16 * R + 4 * rq + i];
            const float var = t * inv_n + eps;
            float rstd = __builtin_amdgcn_rsqf(var);                 // 1 ulp; one Newton step on top
            rstd = rstd * (1.5f - 0.5f * var * rstd * rstd);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 y = a[R][s] * rstd;
                y = y * lm[s] + la[s];       // LNM 1: lm = 1 + scale (simple_flow_net.py:44-45); LNM 2: nn.LayerNorm's weight (:81,106)
                a[R][s] = y;
            }
        }
    }
    if (stamp && tid == 0) stamp[5] = clock64_real();
    f32x4 acc[2][CG];
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[R][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 wv[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            if constexpr (GW) wv[cg] = wreg[cg][s];
            else wv[cg] = *reinterpret_cast<const f32x4*>(in.wl + (4 * cg + i) * KP + kb + 16 * s);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int R = 0; R < 2; ++R)
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) acc[R][cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[R][s][r], wv[cg][r], acc[R][cg], 0, 0, 0);
    }
    // block (rq, kq) of an accumulator = rows 4 rq .. 4 rq + 3 (register v) x columns 4 cg + (lane & 3), over the k of residue kq
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            f32x4 c = acc[R][cg];
#pragma unroll
            for (int v = 0; v < 4; ++v) c[v] = kq_sum(c[v]);
            const float val = kq == 0 ? c[0] : kq == 1 ? c[1] : kq == 2 ? c[2] : c[3];     // lane kq publishes row 4 rq + kq
            red[(w * RS_ROWS + 16 * R + 4 * rq + kq) * 8 + 4 * cg + i] = val;
        }
    __syncthreads();
    if (stamp && tid == 0) stamp[3] = clock64_real();
    return true;
}

// The observation encoder's GEMM: A (nrows, Kact) and W (N, Kact) row-major with Kact not a multiple of 4 and rows that are not
// 16-byte aligned (O = 533: 512 camera features + 21 joint positions), so both operands are read dword by dword.  Nothing of it is
// written in this launch: plain loads, no wait.  Same lane layout and `red` result as rs_phase<1024, 1, ...>.
__device__ __forceinline__ void rs_phase_unaligned(const float* __restrict__ A, int64_t lda, int row0, int nrows, int Kact,
                                                   const float* __restrict__ W, int col0, int N, float* red) {
    constexpr int KDIM = 1024, NS = KDIM / 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 3, kq = (lane >> 2) & 3, rq = lane >> 4;
    const int kb = w * (KDIM / 8) + 4 * kq;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int col = col0 + i;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (w * (KDIM / 8) + 16 * s >= Kact) break;          // wave-uniform: this 16-k piece lies past the operands' end
        f32x4 a[2], wv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kb + 16 * s + r;
            const bool kin = k < Kact;
            wv[r] = (kin && col < N) ? W[(int64_t)col * Kact + k] : 0.f;
#pragma unroll
            for (int R = 0; R < 2; ++R) {
                const int row = row0 + 16 * R + 4 * rq + i;
                a[R][r] = (kin && row < nrows) ? A[(int64_t)row * lda + k] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int R = 0; R < 2; ++R) acc[R] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[R][r], wv[r], acc[R], 0, 0, 0);
    }
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        f32x4 c = acc[R];
#pragma unroll
        for (int v = 0; v < 4; ++v) c[v] = kq_sum(c[v]);
        const float val = kq == 0 ? c[0] : kq == 1 ? c[1] : kq == 2 ? c[2] : c[3];
        red[(w * RS_ROWS + 16 * R + 4 * rq + kq) * 8 + i] = val;
    }
    __syncthreads();
}

// sum of the 8 waves' partials, in wave order
__device__ __forceinline__ f32x4 rs_sum4(const float* red, int row, int c0) {
    f32x4 s = *reinterpret_cast<const f32x4*>(red + row * 8 + c0);
#pragma unroll
    for (int ww = 1; ww < RS_NW; ++ww) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(red + (ww * RS_ROWS + row) * 8 + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += t[k];
    }
    return s;
}

__global__ __launch_bounds__(RS_NT) void ode_resident_kernel(ResidentArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;                     // the column slice this workgroup owns
    const int L = q.w.L, B = q.B;
    constexpr int D = RS_D, H = RS_D, FF = RS_FF;
    float* w_in = smem;
    float* w_out = w_in + 2 * RS_KP1;
    float* w_fc2 = w_out + 2 * RS_KP1;
    float* w_fc1 = w_fc2 + (size_t)L * 2 * RS_KP2;
    float* red = w_fc1 + (size_t)L * 8 * RS_KP1;
    float* st = red + RS_RED;
    float* flag = st + RS_ST;

    // ---- the weight slices, once per solve (plain loads: nobody writes weights in this launch)
    {
        auto copy_rows = [&](const float* src, int row0, int nrows, int K, float* dst, int KP) {
            const int per_row = K / 4;
            for (int e = tid; e < nrows * per_row; e += RS_NT) {
                const int r = e / per_row, k4 = e - r * per_row;
                *reinterpret_cast<f32x4*>(dst + r * KP + 4 * k4) = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + r) * K + 4 * k4);
            }
        };
        copy_rows(q.w.in_w, 2 * c, 2, D, w_in, RS_KP1);
        copy_rows(q.w.out_w, 2 * c, 2, H, w_out, RS_KP1);
        for (int l = 0; l < L; ++l) {
            copy_rows(q.w.fc2_w[l], 2 * c, 2, FF, w_fc2 + (size_t)l * 2 * RS_KP2, RS_KP2);
            copy_rows(q.w.fc1_w[l], 8 * c, 8, H, w_fc1 + (size_t)l * 8 * RS_KP1, RS_KP1);
        }
    }
    // ---- this workgroup's two columns of the state: thread t < B keeps row t of x and of h in registers for the whole solve.
    // Batches above 32 rows go through every phase in halves of 32 rows (nh = 2): half hf is multiplied by one rs_phase call,
    // its sums are turned into outputs by threads [32 hf, 32 hf + 32) (fc1: [64 hf, 64 hf + 64)), one wait and one publish per phase.
    const int row = tid;                          // meaningful for tid < RS_MAXB
    const bool mine = tid < RS_MAXB && row < B;
    const int nh = (B + RS_ROWS - 1) / RS_ROWS;
    const int col2 = 2 * c;
    float x0v = 0.f, x1v = 0.f, h0v = 0.f, h1v = 0.f;
    {
        const float* xs0 = q.traj ? q.traj : q.x0;
        const int64_t ld0 = q.traj ? D : q.ldx0;
        if (mine && q.obs_w == nullptr) { x0v = xs0[(int64_t)row * ld0 + col2]; x1v = xs0[(int64_t)row * ld0 + col2 + 1]; }
    }
    const unsigned rgs = (unsigned)(B + 3) >> 2;              // row groups that hold a row < B: loads past them return zeros
    const __amdgpu_buffer_rsrc_t rh = rs_rsrc(q.h, rgs * H * 16u), ru = rs_rsrc(q.u, rgs * FF * 16u), rxi = rs_rsrc(q.xi, rgs * D * 16u);
    unsigned* my_cnt = q.sync + (blockIdx.x & 7) * kTeamStride;
    GridSync sy{q.sync, 0u, q.spin_limit, 0u, q.acquire, flag};
    unsigned phase = 0;
    const int nph = q.K * (2 + 2 * L) + (q.dec ? 2 + 2 * q.dw.NL : 0) + (q.tp ? 3 : 0) + ((q.obs_w && !q.tp) ? 1 : 0);
    unsigned long long* stp = q.stamps ? q.stamps + (size_t)blockIdx.x * nph * 6 : nullptr;
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its write-through stores have left
        __syncthreads();                                      // ... all of them; `red` / `st` are free again
        if (tid == 0) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stp) {
            const unsigned long long now = tid == 0 ? clock64_real() : 0ull;
            if (tid == 0) stp[4] = now;
            stp += 6;
            if (tid == 0 && phase + 1 < (unsigned)nph) stp[0] = now;
        }
        ++phase;
        sy.target = (RS_WG / 8) * phase;
        sy.code = phase + 1;
    };
    const __amdgpu_buffer_rsrc_t rmod = rs_rsrc(q.mod, (unsigned)((size_t)q.K * L * 3 * H * 4));
    __syncthreads();                                          // the slices are in LDS
    if (stp && tid == 0) stp[0] = clock64_real();

    if (q.obs_w != nullptr) {
        // ---- obs_encoder (vita_policy.py:194): the start of the flow; this workgroup's two columns of it are its x
        const float b0 = q.obs_b[col2], b1 = q.obs_b[col2 + 1];
        for (int hf = 0; hf < nh; ++hf) {
            rs_phase_unaligned(q.obs, q.ldobs, RS_ROWS * hf, B, q.O, q.obs_w, col2, D, red);
            if ((tid >> 5) == hf) {
                const f32x4 s = rs_sum4(red, row & 31, 0);
                x0v = s[0] + b0; x1v = s[1] + b1;
                if (mine) store8_sc1(rxi, rs_blk(row, col2, D) * 4u, x0v, x1v);
            }
            if (hf + 1 < nh) __syncthreads();
        }
        if (!q.tp) publish();        // otherwise the time path's first publish carries these stores too
        else __syncthreads();        // `red` is rewritten by the next phase
    }
    if (q.tp) {
        // ---- the time path for all K steps (simple_flow_net.py:66-71, 27-30, 41): rows = time steps.  sinusoid -> Linear + Mish ->
        // Linear (+ SiLU, the only form the modulators consume) -> per-layer modulator Linear into the (K, L, 3H) table
        const int T = q.K;
        const unsigned tgs = (unsigned)(T + 3) >> 2;
        const bool trow = tid < RS_ROWS && row < T;
        const __amdgpu_buffer_rsrc_t rthid = rs_rsrc(q.thid, tgs * 1024u * 16u), rtsilu = rs_rsrc(q.tsilu, tgs * 512u * 16u);
        {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (tid < RS_ROWS) b = *reinterpret_cast<const f32x4*>(q.w.te1_b + 4 * c);
            PhaseIn in{};
            in.steps = T; in.wl = q.w.te1_w; in.gw_col0 = 4 * c; in.gw_n = 1024;
            if (!rs_phase<256, 1, 0, 2, true>(in, red, st, sy, stp)) return;
            if (tid < RS_ROWS) {
                f32x4 s = rs_sum4(red, row, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = act_mish(s[e] + b[e]);
                if (trow) store16_sc1(rthid, rs_blk(row, 4 * c, 1024) * 4u, s);
            }
            publish();
        }
        {
            const float b0 = q.w.te3_b[col2], b1 = q.w.te3_b[col2 + 1];
            PhaseIn in{};
            in.ra = rthid; in.wl = q.w.te3_w; in.gw_col0 = col2; in.gw_n = 512;
            if (!rs_phase<1024, 1, 0, 0, true>(in, red, st, sy, stp)) return;
            if (tid < RS_ROWS) {
                const f32x4 s = rs_sum4(red, row, 0);
                if (trow) store8_sc1(rtsilu, rs_blk(row, col2, 512) * 4u, act_silu(s[0] + b0), act_silu(s[1] + b1));
            }
            publish();
        }
        for (int l = 0; l < L; ++l) {       // 3H = 1536 columns per layer: six per workgroup (two 4-column groups, the last two unused)
            float b[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) b[e] = tid < RS_ROWS ? q.w.mod_b[l][6 * c + e] : 0.f;
            PhaseIn in{};
            in.ra = rtsilu; in.wl = q.w.mod_w[l]; in.gw_col0 = 6 * c; in.gw_n = 3 * H;
            if (!rs_phase<RS_D, 2, 0, 0, true>(in, red, st, sy, stp)) return;
            sy.target = 0;                   // the layers' modulators all read the same activations: one wait serves them all
            if (tid < RS_ROWS) {
                const f32x4 sa = rs_sum4(red, row, 0), sb = rs_sum4(red, row, 4);
                if (trow) {
                    const unsigned o = (unsigned)(((size_t)row * L + l) * 3 * H + 6 * c) * 4u;
                    store8_sc1(rmod, o, sa[0] + b[0], sa[1] + b[1]);
                    store8_sc1(rmod, o + 8u, sa[2] + b[2], sa[3] + b[3]);
                    store8_sc1(rmod, o + 16u, sb[0] + b[4], sb[1] + b[5]);
                }
            }
            if (l + 1 < L) __syncthreads();  // `red` is rewritten by the next layer's modulator
        }
        publish();
    }

    for (int k = 0; k < q.K; ++k) {
        const float* xs;
        float* xd;
        int64_t lds_, ldd_;
        if (q.traj) {
            xs = q.traj + (size_t)k * B * D; lds_ = D;
            xd = q.traj + (size_t)(k + 1) * B * D; ldd_ = D;
        } else {
            xs = k == 0 ? q.x0 : q.x_out; lds_ = k == 0 ? q.ldx0 : q.ldo;
            xd = q.x_out; ldd_ = q.ldo;
        }
        // ---- input_proj (simple_flow_net.py:100)
        {
            const float b0 = q.w.in_b[col2], b1 = q.w.in_b[col2 + 1];
            PhaseIn in{};
            in.wl = w_in;
            for (int hf = 0; hf < nh; ++hf) {
                in.row0 = RS_ROWS * hf;
                if (k == 0 && q.obs_w == nullptr) {       // the caller's row-major x0
                    in.ra = rs_rsrc(xs, (unsigned)(((int64_t)(B - 1) * lds_ + D) * 4));
                    in.lda_bytes = (unsigned)lds_ * 4u;
                    if (!rs_phase<RS_D, 1, 0, 1, false>(in, red, st, sy, stp)) return;
                } else {
                    in.ra = rxi;
                    if (!rs_phase<RS_D, 1, 0, 0, false>(in, red, st, sy, stp)) return;
                }
                sy.target = 0;
                if ((tid >> 5) == hf) {
                    const f32x4 s = rs_sum4(red, row & 31, 0);
                    h0v = s[0] + b0; h1v = s[1] + b1;
                    if (mine) store8_sc1(rh, rs_blk(row, col2, H) * 4u, h0v, h1v);
                }
                if (hf + 1 < nh) __syncthreads();
            }
            publish();
        }
        for (int l = 0; l < L; ++l) {
            const unsigned gamma_off = (unsigned)(((size_t)k * L + l) * 3 * H) * 4u;     // byte offset of (gamma | scale | shift) in the table
            // ---- fc1 + tanh-GELU on LN(h) * (1 + scale) + shift (:44-45, 19-24): 8 columns, thread (row, half) stores 4
            {
                const int r8 = tid >> 1, c4 = (tid & 1) * 4;      // thread (row r8, half of the 8 columns) for tid < 2 * RS_MAXB
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (tid < 2 * RS_MAXB) b = *reinterpret_cast<const f32x4*>(q.w.fc1_b[l] + 8 * c + c4);
                PhaseIn in{};
                in.ra = rh; in.wl = w_fc1 + (size_t)l * 8 * RS_KP1;
                in.rln = rmod; in.ln_mul_off = gamma_off + H * 4u; in.ln_add_off = gamma_off + 2 * H * 4u;
                for (int hf = 0; hf < nh; ++hf) {
                    in.row0 = RS_ROWS * hf;
                    if (!rs_phase<RS_D, 2, 1, 0, false>(in, red, st, sy, stp)) return;
                    sy.target = 0;
                    if ((tid >> 6) == hf) {
                        f32x4 s = rs_sum4(red, r8 & 31, c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] = act_gelu_tanh(s[e] + b[e]);
                        if (r8 < B) store16_sc1(ru, rs_blk(r8, 8 * c + c4, FF) * 4u, s);
                    }
                    if (hf + 1 < nh) __syncthreads();
                }
                publish();
            }
            // ---- fc2, gate, residual (:46): h += gamma * fc2(u)
            {
                const float b0 = q.w.fc2_b[l][col2], b1 = q.w.fc2_b[l][col2 + 1];
                const float g0 = load4_sc1(rmod, gamma_off + (unsigned)col2 * 4u), g1 = load4_sc1(rmod, gamma_off + (unsigned)col2 * 4u + 4u);
                PhaseIn in{};
                in.ra = ru; in.wl = w_fc2 + (size_t)l * 2 * RS_KP2;
                for (int hf = 0; hf < nh; ++hf) {
                    in.row0 = RS_ROWS * hf;
                    if (!rs_phase<RS_FF, 1, 0, 0, false>(in, red, st, sy, stp)) return;
                    sy.target = 0;
                    if ((tid >> 5) == hf) {
                        const f32x4 s = rs_sum4(red, row & 31, 0);
                        const float v0 = (s[0] + b0) * g0, v1 = (s[1] + b1) * g1;
                        h0v = h0v + v0; h1v = h1v + v1;
                        if (mine) store8_sc1(rh, rs_blk(row, col2, H) * 4u, h0v, h1v);
                    }
                    if (hf + 1 < nh) __syncthreads();
                }
                publish();
            }
        }
        // ---- final LayerNorm (affine) + out_proj + the Euler update x + v * dt (:106-107, flow_matchers.py:71)
        {
            const float b0 = q.w.out_b[col2], b1 = q.w.out_b[col2 + 1];
            PhaseIn in{};
            in.ra = rh; in.wl = w_out; in.ln_w = q.w.norm_w; in.ln_b = q.w.norm_b;
            for (int hf = 0; hf < nh; ++hf) {
                in.row0 = RS_ROWS * hf;
                if (!rs_phase<RS_D, 1, 2, 0, false>(in, red, st, sy, stp)) return;
                sy.target = 0;
                if ((tid >> 5) == hf) {
                    const f32x4 s = rs_sum4(red, row & 31, 0);
                    const float v0 = (s[0] + b0) * q.dt, v1 = (s[1] + b1) * q.dt;
                    x0v = x0v + v0; x1v = x1v + v1;
                    if (mine) {
                        if (k + 1 < q.K || q.dec) store8_sc1(rxi, rs_blk(row, col2, D) * 4u, x0v, x1v);   // the next step's (or the decoder's) A operand
                        if (q.traj || k + 1 == q.K) {                                                  // what the caller sees: row-major
                            const __amdgpu_buffer_rsrc_t rxd = rs_rsrc(xd, (unsigned)(((int64_t)(B - 1) * ldd_ + D) * 4));
                            store8_sc1(rxd, (unsigned)((int64_t)row * ldd_ + col2) * 4u, x0v, x1v);
                        }
                    }
                }
                if (hf + 1 < nh) __syncthreads();
            }
            publish();
        }
    }
    if (!q.dec) return;
    // ---- SimpleActionDecoder.forward (action_ae.py:267-279) on the final latent: input_proj, NL x Mlp(fc1, exact GELU, fc2),
    // output_proj; 512-wide like the flow net, its weights used once per solve: read straight from global memory
    {
        const vita_decoder_weights& dw = q.dw;
        const __amdgpu_buffer_rsrc_t rda = rs_rsrc(q.da, rgs * 512u * 16u), rdb = rs_rsrc(q.db, rgs * 512u * 16u);
        {
            const float b0 = dw.in_b[col2], b1 = dw.in_b[col2 + 1];
            PhaseIn in{};
            in.ra = rxi; in.wl = dw.in_w; in.gw_col0 = col2; in.gw_n = 512;
            for (int hf = 0; hf < nh; ++hf) {
                in.row0 = RS_ROWS * hf;
                if (!rs_phase<RS_D, 1, 0, 0, true>(in, red, st, sy, stp)) return;
                sy.target = 0;
                if ((tid >> 5) == hf) {
                    const f32x4 s = rs_sum4(red, row & 31, 0);
                    if (mine) store8_sc1(rda, rs_blk(row, col2, 512) * 4u, s[0] + b0, s[1] + b1);
                }
                if (hf + 1 < nh) __syncthreads();
            }
            publish();
        }
        for (int l = 0; l < dw.NL; ++l) {
            {
                const float b0 = dw.fc1_b[l][col2], b1 = dw.fc1_b[l][col2 + 1];
                PhaseIn in{};
                in.ra = rda; in.wl = dw.fc1_w[l]; in.gw_col0 = col2; in.gw_n = 512;
                for (int hf = 0; hf < nh; ++hf) {
                    in.row0 = RS_ROWS * hf;
                    if (!rs_phase<RS_D, 1, 0, 0, true>(in, red, st, sy, stp)) return;
                    sy.target = 0;
                    if ((tid >> 5) == hf) {
                        const f32x4 s = rs_sum4(red, row & 31, 0);
                        if (mine) store8_sc1(rdb, rs_blk(row, col2, 512) * 4u, act_gelu_erf(s[0] + b0), act_gelu_erf(s[1] + b1));
                    }
                    if (hf + 1 < nh) __syncthreads();
                }
                publish();
            }
            {
                const float b0 = dw.fc2_b[l][col2], b1 = dw.fc2_b[l][col2 + 1];
                PhaseIn in{};
                in.ra = rdb; in.wl = dw.fc2_w[l]; in.gw_col0 = col2; in.gw_n = 512;
                for (int hf = 0; hf < nh; ++hf) {
                    in.row0 = RS_ROWS * hf;
                    if (!rs_phase<RS_D, 1, 0, 0, true>(in, red, st, sy, stp)) return;
                    sy.target = 0;
                    if ((tid >> 5) == hf) {
                        const f32x4 s = rs_sum4(red, row & 31, 0);
                        if (mine) store8_sc1(rda, rs_blk(row, col2, 512) * 4u, s[0] + b0, s[1] + b1);
                    }
                    if (hf + 1 < nh) __syncthreads();
                }
                publish();
            }
        }
        {
            const bool have = col2 < dw.OUT;          // OUT is even: a workgroup has both of its columns or none
            float b0 = 0.f, b1 = 0.f;
            if (have) { b0 = dw.out_b[col2]; b1 = dw.out_b[col2 + 1]; }
            PhaseIn in{};
            in.ra = rda; in.wl = dw.out_w; in.gw_col0 = col2; in.gw_n = dw.OUT;
            for (int hf = 0; hf < nh; ++hf) {
                in.row0 = RS_ROWS * hf;
                if (!rs_phase<RS_D, 1, 0, 0, true>(in, red, st, sy, stp)) return;
                sy.target = 0;
                if ((tid >> 5) == hf && mine && have) {
                    const f32x4 s = rs_sum4(red, row & 31, 0);
                    float* o = q.actions + (int64_t)row * q.lda + col2;       // nobody reads it in this launch: plain stores
                    o[0] = s[0] + b0; o[1] = s[1] + b1;
                }
                if (hf + 1 < nh) __syncthreads();
            }
        }
    }
}

// ---- the failure path of both in-launch samplers ---------------------------------------------------------------------
// A workgroup whose bounded wait ran out has left with its outputs unwritten (vita_ode_persistent.hip TeamSync, GridSync above):
// this kernel, enqueued behind the launch, reads the call's sticky status word and overwrites the outputs with NaN when it is
// set, so nobody can consume undefined actions without noticing (the word itself needs a host read: vita_*_status).
__global__ void poison_on_timeout_kernel(const unsigned* sync, unsigned* buf, int64_t ld, int64_t rows, int64_t cols) {
    if (sync[kStickyWord] == 0) return;      // across a kernel boundary: a plain load reads the launch's stores
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
        for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) buf[r * ld + c] = 0x7FC07FC0u;
}
int launch_poison_on_timeout(const void* sync, void* buf, int64_t ld_words, int64_t rows, int64_t cols_words, hipStream_t s) {
    if (buf == nullptr || rows <= 0 || cols_words <= 0) return VITA_OK;
    const unsigned grid = (unsigned)(rows < 512 ? rows : 512);
    hipLaunchKernelGGL(poison_on_timeout_kernel, dim3(grid), dim3(256), 0, s, static_cast<const unsigned*>(sync), static_cast<unsigned*>(buf),
                       ld_words, rows, cols_words);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
static std::atomic<unsigned> g_spin_limit{1u << 22};      // ~ seconds: only a lost workgroup gets there
unsigned ode_spin_limit() { return g_spin_limit.load(std::memory_order_relaxed); }
void set_ode_spin_limit(unsigned n) { g_spin_limit.store(n == 0 ? (1u << 22) : n, std::memory_order_relaxed); }

// ---- host side -------------------------------------------------------------------------------------------------------
static std::atomic<int> g_cus[64];
int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
        g_cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

bool ode_resident_dims(const vita_flownet_weights* w, int B) {
    return w->D == RS_D && w->H == RS_D && w->FF == RS_FF && w->L >= 1 && w->L <= 4 && B >= 1 && B <= RS_MAXB;
}

bool ode_resident_eligible(const vita_flownet_weights* w, int B, const float* x0, int64_t ldx0, const float* x_out, int64_t ldo) {
    if (!ode_persistent_enabled() || !ode_resident_dims(w, B)) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!al(x0) || !al(x_out) || ldx0 % 4 || ldo % 4) return false;
    if (!al(w->in_w) || !al(w->out_w) || !al(w->norm_w) || !al(w->norm_b)) return false;
    for (int l = 0; l < w->L; ++l)
        if (!al(w->fc1_w[l]) || !al(w->fc2_w[l]) || !al(w->fc1_b[l])) return false;
    if ((int64_t)B * (ldx0 > ldo ? ldx0 : ldo) * 4 >= (int64_t(1) << 31)) return false;
    return device_cus() >= RS_WG;     // one workgroup per CU, all co-resident
}

bool ode_resident_decoder_ok(const vita_decoder_weights* dw, const float* actions, int64_t lda) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (dw->D != RS_D || dw->HD != RS_D || dw->NL < 0 || dw->NL > 4 || dw->OUT < 2 || dw->OUT > 2 * RS_WG || dw->OUT % 2) return false;
    if (!al(dw->in_w) || !al(dw->out_w) || lda < dw->OUT || actions == nullptr) return false;
    for (int l = 0; l < dw->NL; ++l)
        if (!al(dw->fc1_w[l]) || !al(dw->fc2_w[l])) return false;
    return true;
}

// Whether the launch can run the time path itself (then the caller enqueues no time-path kernel and passes thid / tsilu).
bool ode_resident_time_path(const vita_flownet_weights* w, int K) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (w->TE != 256 || K < 1 || K > RS_ROWS || !al(w->te1_w) || !al(w->te3_w) || !al(w->te1_b)) return false;
    for (int l = 0; l < w->L; ++l)
        if (!al(w->mod_w[l])) return false;
    return true;
}

// h, u, xi: ((B + 3) / 4 * 4) x (H, FF, D) floats of scratch; thid / tsilu: 32 x (1024, 512) floats, or NULL with mod (K, L, 3H) already enqueued on `s`; sync = the workspace's hand-off words.
int launch_ode_resident(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, float* traj, int B, int K,
                        float dt, float* h, float* u, float* xi, float* mod, float* thid, float* tsilu, void* sync, bool acquire, hipStream_t s,
                        const vita_decoder_weights* dw, float* da, float* db, float* actions, int64_t lda, const ResidentObs* obs) {
    static const bool force_acquire = getenv("VITA_RESIDENT_ACQUIRE") != nullptr && atoi(getenv("VITA_RESIDENT_ACQUIRE")) != 0;
    ResidentArgs a{};
    a.w = *w;
    a.x0 = x0; a.ldx0 = ldx0; a.x_out = x_out; a.ldo = ldo; a.traj = traj;
    a.h = h; a.u = u; a.xi = xi; a.mod = mod;
    a.tp = thid != nullptr ? 1 : 0; a.thid = thid; a.tsilu = tsilu;
    a.sync = static_cast<unsigned*>(sync);
    a.B = B; a.K = K; a.dt = dt;
    a.spin_limit = ode_spin_limit();
    a.acquire = (acquire || force_acquire) ? 1 : 0;
    a.stamps = static_cast<unsigned long long*>(ode_persistent_stamps());
    if (obs != nullptr) { a.obs_w = obs->w; a.obs_b = obs->b; a.obs = obs->x; a.ldobs = obs->ldx; a.O = obs->O; }
    if (dw != nullptr) {      // the decoder rides behind the last step (the caller has checked ode_resident_decoder_ok)
        a.dec = 1; a.dw = *dw; a.da = da; a.db = db; a.actions = actions; a.lda = lda;
    }
    VITA_CHECK_HIP(hipMemsetAsync(sync, 0, ode_sync_words_bytes(), s));      // counters, timeout word, status word
    hipLaunchKernelGGL(ode_resident_kernel, dim3(RS_WG), dim3(RS_NT), rs_smem_floats(w->L) * sizeof(float), s, a);
    VITA_CHECK_HIP(hipGetLastError());
    // a timed-out launch leaves NaN, not undefined values, in everything the caller reads
    VITA_TRY(launch_poison_on_timeout(sync, x_out, ldo, B, w->D, s));
    if (traj) VITA_TRY(launch_poison_on_timeout(sync, traj, w->D, (int64_t)(K + 1) * B, w->D, s));
    if (dw != nullptr) VITA_TRY(launch_poison_on_timeout(sync, actions, lda, B, dw->OUT, s));
    return VITA_OK;
}

}  // namespace vita
