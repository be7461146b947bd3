// HBM-bound elementwise / reduction kernels of the VITA hot path: losses and their backward,
// CFM interpolation, (un)normalisation, sinusoidal time embedding.
// All are grid-stride, 16-byte vectorised where alignment allows, wave64 shuffle reductions,
// and reduce in two deterministic stages (per-block partial -> one-block final) so a loss
// value is bitwise reproducible run to run (no float atomics).
#include "vita_common.h"

namespace vita {

constexpr int EW_THREADS = 256;
constexpr int LOSS_BLOCKS = 1024;  // = vita_loss_partials_count()

static inline int ew_grid(int64_t work_items, int cap = 2048) {
    int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- stage 1: per-block partial of sum f(a-b) ---------------------------------------------
template <bool L1>
__global__ __launch_bounds__(EW_THREADS) void diff_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               int64_t n, int vec, float* __restrict__ partials) {
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    const int64_t gid = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
        const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
        for (int64_t i = gid; i < n4; i += stride) {
            const f32x4 x = a4[i], y = b4[i];
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = x[j] - y[j];
                t += L1 ? fabsf(d) : d * d;
            }
            s += t;
        }
        for (int64_t i = (n4 << 2) + gid; i < n; i += stride) {
            const float d = a[i] - b[i];
            s += L1 ? fabsf(d) : d * d;
        }
    } else {
        for (int64_t i = gid; i < n; i += stride) {
            const float d = a[i] - b[i];
            s += L1 ? fabsf(d) : d * d;
        }
    }
    s = wave_sum(s);
    __shared__ float ws[EW_THREADS / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < EW_THREADS / 64; ++i) t += ws[i];
        partials[blockIdx.x] = t;
    }
}

// ---- stage 2: one block folds the partials in double, divides by n -------------------------
__global__ __launch_bounds__(EW_THREADS) void finalize_mean_kernel(const float* __restrict__ partials, int count, int64_t n,
                                                                 float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += EW_THREADS) s += (double)partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[EW_THREADS / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < EW_THREADS / 64; ++i) t += ws[i];
        out[0] = (float)(t / (double)n);
    }
}

template <bool L1>
__global__ __launch_bounds__(EW_THREADS) void diff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                            const float* __restrict__ gout, float* __restrict__ ga,
                                                            float* __restrict__ gb, int accumulate) {
    const float g = gout[0];
    const float inv_n = 1.0f / (float)n;
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - b[i];
        float v;
        if (L1) v = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_n * g;   // sign(d)/n * gout
        else v = (2.0f * d) * inv_n * g;                                    // 2 d / n * gout
        if (ga) ga[i] = accumulate ? ga[i] + v : v;
        if (gb) gb[i] = accumulate ? gb[i] - v : -v;
    }
}

template <bool L1>
static int loss_fwd(const float* a, const float* b, int64_t n, float* partials, float* out, hipStream_t s, const char* name) {
    VITA_REQUIRE(a && b && partials && out, "%s: null pointer", name);
    VITA_REQUIRE(n > 0, "%s: n must be > 0 (got %lld); the mean of an empty tensor is undefined", name, (long long)n);
    const int vec = al16(a) && al16(b);
    const int blocks = ew_grid((n + 3) / 4, LOSS_BLOCKS);
    hipLaunchKernelGGL((diff_reduce_kernel<L1>), dim3(blocks), dim3(EW_THREADS), 0, s, a, b, n, vec, partials);
    hipLaunchKernelGGL(finalize_mean_kernel, dim3(1), dim3(EW_THREADS), 0, s, partials, blocks, n, out);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
template <bool L1>
static int loss_bwd(const float* a, const float* b, int64_t n, const float* gout, float* ga, float* gb, int accumulate, hipStream_t s,
                    const char* name) {
    VITA_REQUIRE(a && b && gout, "%s: null pointer", name);
    VITA_REQUIRE(n > 0, "%s: n must be > 0", name);
    if (!ga && !gb) return VITA_OK;
    hipLaunchKernelGGL((diff_bwd_kernel<L1>), dim3(ew_grid(n)), dim3(EW_THREADS), 0, s, a, b, n, gout, ga, gb, accumulate);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// ---- CFM interpolation ---------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void cfm_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                       const float* __restrict__ t, const int64_t* __restrict__ pi,
                                                       const int64_t* __restrict__ pj, float* __restrict__ xt, float* __restrict__ ut,
                                                       int B, int D) {
    const int64_t n = (int64_t)B * D;
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const int64_t r0 = pi ? pi[r] : r;
        const int64_t r1 = pj ? pj[r] : r;
        const float a = x0[r0 * D + c], b = x1[r1 * D + c], tt = t[r];
        xt[i] = tt * b + (1.0f - tt) * a;   // torchcfm: t*x1 + (1-t)*x0  (sigma = 0)
        ut[i] = b - a;
    }
}

// torchcfm 1.0.5 sample_location_and_conditional_flow for the three Gaussian-path matchers the reference registers
// (flow_matchers.py:82-99, 137-145), op for op (RESTATED from the published source, UNPINNED: torchcfm is not in the image):
//   conditional / exact   mu = t*x1 + (1-t)*x0 ; xt = mu + sigma*eps                 ; ut = x1 - x0
//   target                mu = t*x1            ; st = 1 - (1-sigma)*t ; xt = mu + st*eps ; ut = (x1 - (1-sigma)*xt) / st
//   schrodinger           mu as conditional    ; st = sigma*sqrt(t*(1-t)) ; xt = mu + st*eps ;
//                         ut = (1-2t)/(2*t*(1-t) + 1e-8) * (xt - mu) + x1 - x0
// eps is the randn_like(x0) draw, indexed by output row; NULL only for conditional with sigma = 0 (then + 0*eps is exact).
template <int KIND>
__global__ __launch_bounds__(EW_THREADS) void fm_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ t,
                                                      const float* __restrict__ eps, const int64_t* __restrict__ pi,
                                                      const int64_t* __restrict__ pj, float* __restrict__ xt, float* __restrict__ ut, int B, int D,
                                                      float sigma, float oms) {
    const int64_t n = (int64_t)B * D;
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const int64_t r0 = pi ? pi[r] : r;
        const int64_t r1 = pj ? pj[r] : r;
        const float a = x0[r0 * D + c], b = x1[r1 * D + c], tt = t[r];
        const float e = eps ? eps[i] : 0.f;
        if constexpr (KIND == VITA_FM_CONDITIONAL) {
            const float mu = tt * b + (1.0f - tt) * a;
            xt[i] = eps ? mu + sigma * e : mu;
            ut[i] = b - a;
        } else if constexpr (KIND == VITA_FM_TARGET) {
            const float mu = tt * b;
            const float st = 1.0f - oms * tt;
            const float x = mu + st * e;
            xt[i] = x;
            ut[i] = (b - oms * x) / st;
        } else {
            const float mu = tt * b + (1.0f - tt) * a;
            const float st = sigma * sqrtf(tt * (1.0f - tt));
            const float x = mu + st * e;
            const float rr = (1.0f - 2.0f * tt) / (2.0f * tt * (1.0f - tt) + 1e-8f);
            xt[i] = x;
            ut[i] = rr * (x - mu) + b - a;
        }
    }
}

// ---- (un)normalise ---------------------------------------------------------------------------
template <bool UN>
__global__ __launch_bounds__(EW_THREADS) void norm_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ sd, float* __restrict__ y, int64_t n, int C, int64_t inner) {
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / inner) % C);
        if (UN) y[i] = x[i] * sd[c] + mean[c];
        else y[i] = (x[i] - mean[c]) / (sd[c] + 1e-8f);
    }
}

// ---- LayerNorm (+ modulation / affine), one wave per row --------------------------------------
// y = ((x - mean) * rstd) * m + a,  m = mul (+1 when MOD), two-pass statistics held in registers.
// The row (H floats) is read once: lane l holds float4 chunks l, l+64, ... (H <= 64*4*LN_MAXC).
constexpr int LN_MAXC = 8;
// NC = float4 chunks per lane (H <= 256*NC), ROWS = rows per wave: all of a wave's loads are issued before the first
// reduction so several rows are in flight per wave (at one row per wave the kernel ran at 2.1 TB/s, latency-bound).
// SV: the modulation vectors are shared by all rows (ld = 0) and 16-byte loadable: held once per wave (ROWS x fewer registers
// and L1 requests for them)
template <int NC, int ROWS, bool SV = false>
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                     int M, int H, float eps, int mod, const float* __restrict__ mul, int64_t ld_mul,
                                                     const float* __restrict__ add, int64_t ld_add, int vec, uint16_t* __restrict__ ybf,
                                                     int64_t ld_ybf) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= M) return;
    f32x4 v[ROWS][NC];
    float s[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, M - 1);
        const float* xr = x + (int64_t)row * ldx;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = (c * 64 + lane) * 4;
            v[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < H) {
                if (vec) v[r][c] = *reinterpret_cast<const f32x4*>(xr + k);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < H) v[r][c][j] = xr[k + j];
                }
            }
        }
    }
    // the modulation vectors are requested together with the rows (not after the two reductions): one memory round trip
    // instead of two on the critical path of a latency-bound kernel
    // (vectors shared by all rows — the sampler's per-step modulation, ld = 0 — are loaded once per wave, not once per row:
    // at 4 rows per wave the repeated loads were two thirds of the kernel's L1 requests)
    constexpr int VR = SV ? 1 : ROWS;
    f32x4 m4[VR][NC], a4[VR][NC];
    if (vec) {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int row = min(row0 + r, M - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = (c * 64 + lane) * 4;
                if (k < H) {
                    m4[r][c] = *reinterpret_cast<const f32x4*>(mul + (int64_t)row * ld_mul + k);
                    a4[r][c] = *reinterpret_cast<const f32x4*>(add + (int64_t)row * ld_add + k);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) t += (v[r][c][0] + v[r][c][1]) + (v[r][c][2] + v[r][c][3]);
        s[r] = wave_sum(t);
    }
    float mu[ROWS], rstd[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        mu[r] = s[r] / (float)H;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = (c * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = (k + j < H) ? v[r][c][j] - mu[r] : 0.f;
                q = fmaf(d, d, q);
            }
        }
        q = wave_sum(q);
        rstd[r] = 1.0f / sqrtf(q / (float)H + eps);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float* yr = y + (int64_t)row * ldy;
        const float* mr = mul + (int64_t)row * ld_mul;
        const float* ar = add + (int64_t)row * ld_add;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = (c * 64 + lane) * 4;
            if (k < H) {
                f32x4 o;
                if (vec) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float n = (v[r][c][j] - mu[r]) * rstd[r];
                        const float m = mod ? (m4[SV ? 0 : r][c][j] + 1.0f) : m4[SV ? 0 : r][c][j];
                        o[j] = n * m + a4[SV ? 0 : r][c][j];
                    }
                    if (y) *reinterpret_cast<f32x4*>(yr + k) = o;
                    if (ybf) {   // bf16 copy: the next bf16 GEMM's A operand
                        u16x4 hb;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hb[j] = f32_to_bf16(o[j]);
                        *reinterpret_cast<u16x4*>(ybf + (int64_t)row * ld_ybf + k) = hb;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k + j < H) {
                            const float n = (v[r][c][j] - mu[r]) * rstd[r];
                            const float m = mod ? (mr[k + j] + 1.0f) : mr[k + j];
                            const float o1 = n * m + ar[k + j];
                            if (y) yr[k + j] = o1;
                            if (ybf) ybf[(int64_t)row * ld_ybf + k + j] = f32_to_bf16(o1);
                        }
                    }
                }
            }
        }
    }
}

int launch_ln_mod(const float* x, int64_t ldx, float* y, int64_t ldy, int M, int H, float eps, int mode, const float* mul,
                  int64_t ld_mul, const float* add, int64_t ld_add, hipStream_t s, uint16_t* ybf, int64_t ld_ybf) {
    VITA_REQUIRE(x && (y || ybf) && mul && add, "vita_layernorm_modulate_f32: null pointer");
    VITA_REQUIRE(ybf == nullptr || (ld_ybf >= H && ld_ybf % 4 == 0 && (reinterpret_cast<uintptr_t>(ybf) & 7u) == 0),
                 "vita_layernorm_modulate: bf16 output needs ld >= H, ld %% 4 == 0 and 8-byte alignment");
    VITA_REQUIRE(M >= 0 && H > 0 && H <= 64 * 4 * LN_MAXC, "vita_layernorm_modulate_f32: H=%d outside (0, %d]", H, 64 * 4 * LN_MAXC);
    VITA_REQUIRE(ldx >= H && (y == nullptr || ldy >= H) && (ld_mul == 0 || ld_mul >= H) && (ld_add == 0 || ld_add >= H),
                 "vita_layernorm_modulate_f32: bad leading dimension");
    VITA_REQUIRE(mode == VITA_PRO_LN_MOD || mode == VITA_PRO_LN_AFFINE, "vita_layernorm_modulate_f32: mode must be LN_MOD or LN_AFFINE");
    if (M == 0) return VITA_OK;
    const int vec = (H % 4 == 0) && (ldx % 4 == 0) && (y == nullptr || (ldy % 4 == 0 && al16(y))) && al16(x) && al16(mul) && al16(add) &&
                    (ld_mul % 4 == 0) && (ld_add % 4 == 0);
    const int md = mode == VITA_PRO_LN_MOD ? 1 : 0;
    const bool sv = vec && ld_mul == 0 && ld_add == 0;
#define VITA_LN(NC, ROWS)                                                                                                       \
    do {                                                                                                                        \
        if (sv && ROWS > 1)                                                                                                     \
            hipLaunchKernelGGL((ln_mod_kernel<NC, ROWS, true>), dim3((M + 4 * ROWS - 1) / (4 * ROWS)), dim3(256), 0, s, x, ldx, y, ldy, M, H, \
                               eps, md, mul, ld_mul, add, ld_add, vec, ybf, ld_ybf);                                             \
        else                                                                                                                    \
            hipLaunchKernelGGL((ln_mod_kernel<NC, ROWS>), dim3((M + 4 * ROWS - 1) / (4 * ROWS)), dim3(256), 0, s, x, ldx, y, ldy, M, H, eps,  \
                               md, mul, ld_mul, add, ld_add, vec, ybf, ld_ybf);                                                  \
    } while (0)
#ifndef VITA_LN_ROWS
#define VITA_LN_ROWS 4
#endif
    // few rows: one row per wave (a wave's rows share its shuffle chains: 4 rows = 48 dependent cross-lane steps, which is
    // what a launch at M <= 1024 spends its time on); many rows: several rows in flight per wave to cover the memory latency
    const bool few = M <= 1024;
    if (H <= 256) { if (few) VITA_LN(1, 1); else VITA_LN(1, VITA_LN_ROWS); }
    else if (H <= 512) { if (few) VITA_LN(2, 1); else VITA_LN(2, VITA_LN_ROWS); }
    else if (H <= 1024) { if (few) VITA_LN(4, 1); else VITA_LN(4, 2); }
    else VITA_LN(8, 1);
#undef VITA_LN
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// ---- tables of the folded LayerNorm (vita_api.hip) --------------------------------------------
// One wave per weight row j of one "layer" (blockIdx.y): rs = sum_i W[j][i]; for each of the K modulation vectors
// c1[k][j] = sum_i mul_k[i] W[j][i] (+ rs), c2[k][j] = sum_i add_k[i] W[j][i] + bias[j].  The weight row stays in registers
// (H <= 2048), the per-lane partial dot products of 4 vectors are formed before their reductions.  Deterministic.
struct FoldTableJob { const float* W; const float* bias; const float* mul; const float* add; float* c1; float* c2; int N; int K; int add_one; int64_t ld_vec, ld_c;
                      const uint16_t* Wb; int64_t ldwb; };   // Wb != NULL: the weight rows are bf16 (row stride ldwb): the bf16-operand sampler's tables
struct FoldTableJobs { FoldTableJob j[VITA_MAX_LAYERS + 1]; };

__global__ __launch_bounds__(256) void ln_fold_tables_kernel(FoldTableJobs jobs, int H) {
    const FoldTableJob& jb = jobs.j[blockIdx.y];
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= jb.N) return;
    const float* wr = jb.W + (int64_t)j * H;
    const int nch = (H + 255) / 256;           // float4 chunks per lane
    f32x4 w[8];
    float rs = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        w[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int i = (c * 64 + lane) * 4;
        if (c < nch && i < H) {
            if (jb.Wb) {
                const u16x4 wb = *reinterpret_cast<const u16x4*>(jb.Wb + (int64_t)j * jb.ldwb + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[c][e] = bf16_to_f32(wb[e]);
            } else {
                w[c] = *reinterpret_cast<const f32x4*>(wr + i);
            }
        }
        rs += (w[c][0] + w[c][1]) + (w[c][2] + w[c][3]);
    }
    rs = wave_sum(rs);
    const float bj = jb.bias ? jb.bias[j] : 0.f;
    for (int k0 = 0; k0 < jb.K; k0 += 4) {
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u >= jb.K) break;
            const float* mk = jb.mul + (int64_t)(k0 + u) * jb.ld_vec;
            const float* ak = jb.add + (int64_t)(k0 + u) * jb.ld_vec;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int i = (c * 64 + lane) * 4;
                if (c < nch && i < H) {
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(mk + i), a4 = *reinterpret_cast<const f32x4*>(ak + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[u] = fmaf(m4[e], w[c][e], a[u]); b[u] = fmaf(a4[e], w[c][e], b[u]); }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = wave_sum(a[u]); b[u] = wave_sum(b[u]); }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u >= jb.K) break;
                jb.c1[(int64_t)(k0 + u) * jb.ld_c + j] = jb.add_one ? a[u] + rs : a[u];
                jb.c2[(int64_t)(k0 + u) * jb.ld_c + j] = b[u] + bj;
            }
        }
    }
}

// all layers' fc1 tables + the final (norm -> out_proj) table in one launch.  H % 4 == 0, 16-byte aligned rows.
int launch_ln_fold_tables_all(const vita_flownet_weights* w, const float* mod, int T, float* c1, float* c2, float* c1f, float* c2f, hipStream_t s,
                              const vita_bf16_weights* bw) {
    const int H = w->H, FF = w->FF, L = w->L;
    VITA_REQUIRE(H <= 2048 && H % 4 == 0, "ln_fold_tables: H=%d out of range", H);
    FoldTableJobs jobs{};
    int nmax = w->D;
    for (int l = 0; l < L; ++l) {
        jobs.j[l] = FoldTableJob{w->fc1_w[l], w->fc1_b[l], mod + (size_t)l * 3 * H + H, mod + (size_t)l * 3 * H + 2 * H,
                                 c1 + (size_t)l * FF, c2 + (size_t)l * FF, FF, T, 1, (int64_t)L * 3 * H, (int64_t)L * FF,
                                 bw ? bw->fc1_w[l] : nullptr, (int64_t)((H + 7) & ~7)};
        if (FF > nmax) nmax = FF;
    }
    jobs.j[L] = FoldTableJob{w->out_w, w->out_b, w->norm_w, w->norm_b, c1f, c2f, w->D, 1, 0, 0, (int64_t)w->D,
                             bw ? bw->out_w : nullptr, (int64_t)((H + 7) & ~7)};
    hipLaunchKernelGGL(ln_fold_tables_kernel, dim3((nmax + 3) / 4, L + 1), dim3(256), 0, s, jobs, H);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// ---- sinusoidal embedding --------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int64_t ldo,
                                                            int B, int dim) {
    const int half = dim / 2;
    const float emb = logf(10000.0f) / (float)(half - 1);
    const int64_t n = (int64_t)B * half;
    const int64_t stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / half), j = (int)(i - (int64_t)r * half);
        const float f = expf((float)j * -emb);
        const float e = t[r] * f;
        out[r * ldo + j] = sinf(e);
        out[r * ldo + half + j] = cosf(e);
    }
}

// step times k/K for the sampler: ts[k] = 1.0f * (float)(k / K)  (flow_matchers.py:69)
__global__ void step_times_kernel(float* __restrict__ ts, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) ts[k] = (float)((double)k / (double)K);
}

// fp32 (rows, cols) -> bf16 (rows, ld_dst), columns cols..ld_dst-1 zero-filled (so a GEMM may reduce over ld_dst)
__global__ __launch_bounds__(EW_THREADS) void pack_bf16_kernel(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                             int64_t ld_dst, int64_t rows, int cols) {
    const int64_t n = rows * ld_dst, stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / ld_dst;
        const int c = (int)(i - r * ld_dst);
        dst[i] = c < cols ? f32_to_bf16(src[r * ld_src + c]) : (uint16_t)0;
    }
}

// ---- operand staging of the bf16-operand training GEMMs (gemm_f32 under a Bf16Scope, vita_gemm_f32.hip) ----
// fp32 (rows, cols) -> bf16 (rows, ld_dst), eight elements per thread (two 16-byte loads, one 16-byte store); columns
// cols..ld_dst-1 zero-filled.  Needs 16-byte aligned src rows (ld_src % 4 == 0) and ld_dst % 8 == 0.
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(EW_THREADS) void pack_bf16_vec8_kernel(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                                  int64_t ld_dst, int64_t rows, int cols) {
    const int64_t c8 = ld_dst / 8, n = rows * c8, stride = (int64_t)gridDim.x * EW_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const float* sp = src + r * ld_src + c;
        u16x8 o;
        if (c + 8 <= cols) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = f32_to_bf16(a[j]); o[4 + j] = f32_to_bf16(b[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (c + j < cols) ? f32_to_bf16(sp[j]) : (uint16_t)0;
        }
        *reinterpret_cast<u16x8*>(dst + r * ld_dst + c) = o;
    }
}

// fp32 src (K rows, M cols) -> bf16 dst (M rows, ld_dst), dst[m][k] = src[k][m], k in [K, ld_dst) zero-filled: a column
// operand of a backward GEMM (dY^T, X^T, W^T) becomes the k-contiguous rows the LDS-DMA kernel reads.  64 x 64 tiles
// through LDS: 16-byte loads along m, 8-byte stores along k.
template <bool VEC>
__global__ __launch_bounds__(256) void pack_bf16_t_kernel(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                         int64_t ld_dst, int K, int M) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, m0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, kr = idx >> 4, mc = (idx & 15) * 4;
        const int k = k0 + kr, m = m0 + mc;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const float* sp = src + (int64_t)k * ld_src + m;
            if constexpr (VEC) {
                if (m < M) v = *reinterpret_cast<const f32x4*>(sp);      // M % 4 == 0: the four columns exist together
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (m + j < M) v[j] = sp[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[kr][mc + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, kq = (idx & 15) * 4, mr = idx >> 4;
        const int m = m0 + mr, k = k0 + kq;
        if (m < M && k < ld_dst) {      // ld_dst % 8 == 0 and k % 4 == 0: the four k exist together
            u16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f32_to_bf16(tile[kq + j][mr]);
            *reinterpret_cast<u16x4*>(dst + (int64_t)m * ld_dst + k) = o;
        }
    }
}

int launch_pack_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return VITA_OK;
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 &&
                     (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
    if (vec) hipLaunchKernelGGL(pack_bf16_vec8_kernel, dim3(ew_grid(rows * (ld_dst / 8), 4096)), dim3(EW_THREADS), 0, s, src, ld_src, dst, ld_dst, rows, cols);
    else hipLaunchKernelGGL(pack_bf16_kernel, dim3(ew_grid(rows * ld_dst, 4096)), dim3(EW_THREADS), 0, s, src, ld_src, dst, ld_dst, rows, cols);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
// src (K, M) row stride ld_src -> dst (M, ld_dst) bf16, ld_dst % 8 == 0, dst 16-byte aligned
int launch_pack_bf16_t(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int K, int M, hipStream_t s) {
    if (K <= 0 || M <= 0) return VITA_OK;
    VITA_REQUIRE(ld_dst % 8 == 0 && ld_dst >= K && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0, "pack_bf16_t: bad destination");
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((ld_dst + 63) / 64));
    VITA_REQUIRE(grid.y <= 65535u, "pack_bf16_t: K too large (%d)", K);
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && ld_src % 4 == 0 && M % 4 == 0;
    if (vec) hipLaunchKernelGGL((pack_bf16_t_kernel<true>), grid, dim3(256), 0, s, src, ld_src, dst, ld_dst, K, M);
    else hipLaunchKernelGGL((pack_bf16_t_kernel<false>), grid, dim3(256), 0, s, src, ld_src, dst, ld_dst, K, M);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int launch_sinusoid(const float* t, float* out, int64_t ldo, int B, int dim, hipStream_t s) {
    VITA_REQUIRE(t && out, "vita_sinusoidal_embed_f32: null pointer");
    VITA_REQUIRE(dim >= 4 && dim % 2 == 0 && ldo >= dim, "vita_sinusoidal_embed_f32: dim must be even and >= 4, ldo >= dim");
    if (B <= 0) return VITA_OK;
    hipLaunchKernelGGL(sinusoid_kernel, dim3(ew_grid((int64_t)B * (dim / 2))), dim3(EW_THREADS), 0, s, t, out, ldo, B, dim);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int launch_step_times(float* ts, int K, hipStream_t s) {
    hipLaunchKernelGGL(step_times_kernel, dim3((K + 63) / 64), dim3(64), 0, s, ts, K);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

extern "C" {

int32_t vita_loss_partials_count(void) { return LOSS_BLOCKS; }

int vita_mse_loss_fwd_f32(const float* a, const float* b, int64_t n, float* partials, float* out, vita_stream_t stream) {
    return loss_fwd<false>(a, b, n, partials, out, (hipStream_t)stream, "vita_mse_loss_fwd_f32");
}
int vita_l1_loss_fwd_f32(const float* a, const float* b, int64_t n, float* partials, float* out, vita_stream_t stream) {
    return loss_fwd<true>(a, b, n, partials, out, (hipStream_t)stream, "vita_l1_loss_fwd_f32");
}
int vita_mse_loss_bwd_f32(const float* a, const float* b, int64_t n, const float* gout, float* ga, float* gb, int32_t accumulate,
                          vita_stream_t stream) {
    return loss_bwd<false>(a, b, n, gout, ga, gb, accumulate, (hipStream_t)stream, "vita_mse_loss_bwd_f32");
}
int vita_l1_loss_bwd_f32(const float* a, const float* b, int64_t n, const float* gout, float* ga, float* gb, int32_t accumulate,
                         vita_stream_t stream) {
    return loss_bwd<true>(a, b, n, gout, ga, gb, accumulate, (hipStream_t)stream, "vita_l1_loss_bwd_f32");
}

int vita_cfm_interpolate_f32(const float* x0, const float* x1, const float* t, const int64_t* pair_i, const int64_t* pair_j, float* xt,
                             float* ut, int32_t B, int32_t D, vita_stream_t stream) {
    VITA_REQUIRE(x0 && x1 && t && xt && ut, "vita_cfm_interpolate_f32: null pointer");
    VITA_REQUIRE(B >= 0 && D > 0, "vita_cfm_interpolate_f32: bad shape B=%d D=%d", B, D);
    if (B == 0) return VITA_OK;
    hipLaunchKernelGGL(cfm_kernel, dim3(ew_grid((int64_t)B * D)), dim3(EW_THREADS), 0, (hipStream_t)stream, x0, x1, t, pair_i, pair_j, xt,
                       ut, B, D);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_fm_interpolate_f32(int32_t kind, double sigma, const float* x0, const float* x1, const float* t, const float* eps,
                            const int64_t* pair_i, const int64_t* pair_j, float* xt, float* ut, int32_t B, int32_t D, vita_stream_t stream) {
    const char* who = "vita_fm_interpolate_f32";
    VITA_REQUIRE(kind >= VITA_FM_CONDITIONAL && kind <= VITA_FM_SCHRODINGER, "%s: unknown matcher kind %d", who, kind);
    VITA_REQUIRE(x0 && x1 && t && xt && ut, "%s: null pointer", who);
    VITA_REQUIRE(B >= 0 && D > 0, "%s: bad shape B=%d D=%d", who, B, D);
    VITA_REQUIRE(sigma >= 0.0, "%s: sigma must be >= 0", who);
    VITA_REQUIRE(kind != VITA_FM_SCHRODINGER || sigma > 0.0, "%s: Sigma must be strictly positive, got %g.", who, sigma);
    VITA_REQUIRE(eps != nullptr || (kind == VITA_FM_CONDITIONAL && sigma == 0.0), "%s: this matcher adds noise to xt: eps (B, D) is required", who);
    if (B == 0) return VITA_OK;
    const dim3 grid(ew_grid((int64_t)B * D)), block(EW_THREADS);
    const float sg = (float)sigma, oms = (float)(1.0 - sigma);
    hipStream_t s = (hipStream_t)stream;
    if (kind == VITA_FM_CONDITIONAL) hipLaunchKernelGGL((fm_kernel<VITA_FM_CONDITIONAL>), grid, block, 0, s, x0, x1, t, eps, pair_i, pair_j, xt, ut, B, D, sg, oms);
    else if (kind == VITA_FM_TARGET) hipLaunchKernelGGL((fm_kernel<VITA_FM_TARGET>), grid, block, 0, s, x0, x1, t, eps, pair_i, pair_j, xt, ut, B, D, sg, oms);
    else hipLaunchKernelGGL((fm_kernel<VITA_FM_SCHRODINGER>), grid, block, 0, s, x0, x1, t, eps, pair_i, pair_j, xt, ut, B, D, sg, oms);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_normalize_f32(const float* x, const float* mean, const float* sd, float* y, int64_t rows, int32_t C, int64_t inner,
                       vita_stream_t stream) {
    VITA_REQUIRE(x && mean && sd && y, "vita_normalize_f32: null pointer");
    VITA_REQUIRE(rows >= 0 && C > 0 && inner >= 1, "vita_normalize_f32: bad shape");
    if (rows == 0) return VITA_OK;
    hipLaunchKernelGGL((norm_kernel<false>), dim3(ew_grid(rows * C * inner)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, mean, sd, y,
                       rows * C * inner, C, inner);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int vita_unnormalize_f32(const float* x, const float* mean, const float* sd, float* y, int64_t rows, int32_t C, int64_t inner,
                         vita_stream_t stream) {
    VITA_REQUIRE(x && mean && sd && y, "vita_unnormalize_f32: null pointer");
    VITA_REQUIRE(rows >= 0 && C > 0 && inner >= 1, "vita_unnormalize_f32: bad shape");
    if (rows == 0) return VITA_OK;
    hipLaunchKernelGGL((norm_kernel<true>), dim3(ew_grid(rows * C * inner)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, mean, sd, y,
                       rows * C * inner, C, inner);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_layernorm_modulate_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t M, int32_t H, float eps, int32_t mode,
                                const float* mul, int64_t ld_mul, const float* add, int64_t ld_add, vita_stream_t stream) {
    return launch_ln_mod(x, ldx, y, ldy, M, H, eps, mode, mul, ld_mul, add, ld_add, (hipStream_t)stream, nullptr, 0);
}

int vita_pack_bf16_f32(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int32_t cols, vita_stream_t stream) {
    VITA_REQUIRE(src && dst && rows >= 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "vita_pack_bf16_f32: bad arguments");
    if (rows == 0) return VITA_OK;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(ew_grid(rows * ld_dst, 4096)), dim3(EW_THREADS), 0, (hipStream_t)stream, src, ld_src, dst,
                       ld_dst, rows, cols);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_sinusoidal_embed_f32(const float* t, float* out, int64_t ldo, int32_t B, int32_t dim, vita_stream_t stream) {
    return launch_sinusoid(t, out, ldo, B, dim, (hipStream_t)stream);
}

}  // extern "C"
