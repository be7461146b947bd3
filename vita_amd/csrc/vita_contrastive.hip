// Symmetric InfoNCE between two sets of latents (compute_contrastive_loss, contrastive_loss.py:5-22) with its backward,
// and the diagonal-Gaussian KL term (DiagonalGaussianDistribution.kl + .mean(), dists.py:23-31, vita_policy.py:148).
// The B x B logits are two fp32 MFMA GEMMs (an.bn^T and bn.an^T, so both cross-entropies are ROW-wise: coalesced, one
// wave per row); everything else is HBM-bound row kernels with wave64 shuffle reductions; all sums are deterministic.
#include <math.h>

#include "vita_host.h"

namespace vita {

namespace {

constexpr int CT = 256;

// y = x / max(||x||_2, 1e-12) (F.normalize, dim=1); inv[row] = 1 / max(||x||, eps).  One wave per row.
__global__ __launch_bounds__(CT) void rownorm_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, float* __restrict__ inv,
                                                    int B, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* xr = x + (int64_t)row * ldx;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(xr[k], xr[k], s);
    s = wave_sum(s);
    const float n = sqrtf(s);
    const float iv = 1.0f / (n > 1e-12f ? n : 1e-12f);
    for (int k = lane; k < D; k += 64) y[(int64_t)row * D + k] = xr[k] * iv;
    if (lane == 0) inv[row] = iv;
}
// dx (+)= (dy - y * <y, dy>) * inv     (backward of y = x * inv(x))
__global__ __launch_bounds__(CT) void rownorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ inv,
                                                        float* __restrict__ dx, int64_t ld_dx, int B, int D, int accumulate) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* yr = y + (int64_t)row * D;
    const float* gr = dy + (int64_t)row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s = fmaf(yr[k], gr[k], s);
    s = wave_sum(s);
    const float iv = inv[row];
    for (int k = lane; k < D; k += 64) {
        const float g = (gr[k] - yr[k] * s) * iv;
        float* o = dx + (int64_t)row * ld_dx + k;
        *o = accumulate ? *o + g : g;
    }
}
// cross_entropy(logits, arange(B)) per row: loss[i] = logsumexp(l[i,:]) - l[i,i]; lse[i] kept for the backward
__global__ __launch_bounds__(CT) void ce_diag_kernel(const float* __restrict__ l, int B, float* __restrict__ loss, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* lr = l + (int64_t)row * B;
    float m = -INFINITY;
    for (int k = lane; k < B; k += 64) m = fmaxf(m, lr[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int k = lane; k < B; k += 64) s += expf(lr[k] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float e = m + logf(s);
        lse[row] = e;
        loss[row] = e - lr[row];
    }
}
// G[i,j] = (softmax(l[i,:])[j] - [i == j]) * scale[0] * c      in place over the logits
__global__ __launch_bounds__(CT) void ce_diag_bwd_kernel(float* __restrict__ l, const float* __restrict__ lse, int B, const float* __restrict__ scale,
                                                        float c) {
    const int64_t n = (int64_t)B * B, stride = (int64_t)gridDim.x * CT;
    const float sc = scale[0] * c;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / B), col = (int)(i - (int64_t)r * B);
        l[i] = (expf(l[i] - lse[r]) - (r == col ? 1.0f : 0.0f)) * sc;
    }
}
// out[0] = (sum(a[0..n)) + sum(b[0..n))) * c     (one block, fp64 fold: deterministic)
__global__ __launch_bounds__(CT) void pair_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, double c, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += CT) s += (double)a[i] + (b ? (double)b[i] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[CT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) * c);
}
// per-row KL of N(mean, exp(logvar)) against N(0, 1): 0.5 * sum(mean^2 + exp(logvar) - 1 - logvar)
__global__ __launch_bounds__(CT) void kl_rows_kernel(const float* __restrict__ mean, int64_t ldm, const float* __restrict__ logvar, int64_t ldv,
                                                    int B, int Cc, float* __restrict__ rows) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float s = 0.f;
    for (int k = lane; k < Cc; k += 64) {
        const float mu = mean[(int64_t)row * ldm + k], lv = logvar[(int64_t)row * ldv + k];
        s += mu * mu + expf(lv) - 1.0f - lv;
    }
    s = wave_sum(s);
    if (lane == 0) rows[row] = 0.5f * s;
}
__global__ __launch_bounds__(CT) void kl_bwd_kernel(const float* __restrict__ mean, int64_t ldm, const float* __restrict__ logvar, int64_t ldv,
                                                   int B, int Cc, const float* __restrict__ gout, float* __restrict__ dmean, int64_t ld_dm,
                                                   float* __restrict__ dlogvar, int64_t ld_dv) {
    const int64_t n = (int64_t)B * Cc, stride = (int64_t)gridDim.x * CT;
    const float g = gout[0] / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / Cc), c = (int)(i - (int64_t)r * Cc);
        if (dmean) dmean[(int64_t)r * ld_dm + c] = mean[(int64_t)r * ldm + c] * g;
        if (dlogvar) dlogvar[(int64_t)r * ld_dv + c] = 0.5f * (expf(logvar[(int64_t)r * ldv + c]) - 1.0f) * g;
    }
}

// DiagonalGaussianDistribution.sample (models/dists.py:16-21) on params rows [mean | logvar]: z = mean + exp(0.5*logvar) * eps;
// eps == NULL is the deterministic posterior (:12-18): z = mean.
__global__ __launch_bounds__(CT) void gauss_sample_kernel(const float* __restrict__ params, int64_t ldp, const float* __restrict__ eps,
                                                         float* __restrict__ z, int64_t ldz, int B, int D) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * CT;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const float mu = params[(int64_t)r * ldp + c];
        float v = mu;
        if (eps) {
            const float sd = expf(0.5f * params[(int64_t)r * ldp + D + c]);
            v = mu + sd * eps[i];
        }
        z[(int64_t)r * ldz + c] = v;
    }
}
// its backward joined with the KL term's: dmean = dz + ks*mean, dlogvar = dz*eps*std*0.5 + ks*0.5*(exp(logvar) - 1),
// ks = kl_weight / B (0: no KL term).  Deterministic posterior: dmean = dz, dlogvar = 0.
__global__ __launch_bounds__(CT) void gauss_sample_bwd_kernel(const float* __restrict__ params, int64_t ldp, const float* __restrict__ eps,
                                                             const float* __restrict__ dz, int64_t ld_dz, float ks, float* __restrict__ dparams,
                                                             int64_t ld_dp, int B, int D) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * CT;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const float g = dz[(int64_t)r * ld_dz + c];
        float dm = g, dv = 0.f;
        if (eps) {
            const float mu = params[(int64_t)r * ldp + c], lv = params[(int64_t)r * ldp + D + c];
            const float sd = expf(0.5f * lv);
            dv = (g * eps[i]) * sd * 0.5f;
            if (ks != 0.f) {
                dm = dm + mu * ks;
                dv = dv + 0.5f * (expf(lv) - 1.0f) * ks;
            }
        }
        dparams[(int64_t)r * ld_dp + c] = dm;
        dparams[(int64_t)r * ld_dp + D + c] = dv;
    }
}

__global__ __launch_bounds__(CT) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * CT;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < n; i += stride) y[i] += x[i];
}

}  // namespace

int launch_gauss_sample(const float* params, int64_t ldp, const float* eps, float* z, int64_t ldz, int B, int D, hipStream_t s) {
    if (B <= 0) return VITA_OK;
    const int64_t blocks = ((int64_t)B * D + CT - 1) / CT;
    hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(CT), 0, s, params, ldp, eps, z, ldz, B, D);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int launch_gauss_sample_bwd(const float* params, int64_t ldp, const float* eps, const float* dz, int64_t ld_dz, float kl_scale, float* dparams,
                            int64_t ld_dp, int B, int D, hipStream_t s) {
    if (B <= 0) return VITA_OK;
    const int64_t blocks = ((int64_t)B * D + CT - 1) / CT;
    hipLaunchKernelGGL(gauss_sample_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(CT), 0, s, params, ldp, eps, dz, ld_dz,
                       kl_scale, dparams, ld_dp, B, D);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int launch_axpy(const float* x, float* y, int64_t n, hipStream_t s) {
    if (n <= 0) return VITA_OK;
    const int64_t blocks = (n + CT - 1) / CT;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(CT), 0, s, x, y, n);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

struct NceWs {
    float *an, *bn, *inva, *invb, *L, *LT, *l1, *l2, *lse1, *lse2, *dan, *dbn;
    size_t bytes;
};
static NceWs carve_nce(void* ws, size_t B, size_t D) {
    Carver c(ws);
    NceWs w;
    w.an = c.take(B * D); w.bn = c.take(B * D); w.inva = c.take(B); w.invb = c.take(B);
    w.L = c.take(B * B); w.LT = c.take(B * B);
    w.l1 = c.take(B); w.l2 = c.take(B); w.lse1 = c.take(B); w.lse2 = c.take(B);
    w.dan = c.take(B * D); w.dbn = c.take(B * D);
    w.bytes = c.off;
    return w;
}

size_t nce_workspace_bytes(int B, int D) { return carve_nce(nullptr, (size_t)B, (size_t)D).bytes; }

int nce_forward(const float* a, int64_t lda, const float* b, int64_t ldb, int B, int D, float temperature, float* loss, void* ws, hipStream_t s) {
    NceWs w = carve_nce(ws, (size_t)B, (size_t)D);
    const dim3 rows((B + 3) / 4), blk(CT);
    hipLaunchKernelGGL(rownorm_kernel, rows, blk, 0, s, a, lda, w.an, w.inva, B, D);
    hipLaunchKernelGGL(rownorm_kernel, rows, blk, 0, s, b, ldb, w.bn, w.invb, B, D);
    const float it = 1.0f / temperature;
    vita_linear_args g1 = lin(w.an, D, w.bn, D, nullptr, w.L, B, B, B, D, VITA_ACT_NONE);
    g1.use_gate_scalar = 1; g1.gate_scalar = it;
    VITA_TRY(gemm_f32(g1, s));                                   // logits           (contrastive_loss.py:15)
    vita_linear_args g2 = lin(w.bn, D, w.an, D, nullptr, w.LT, B, B, B, D, VITA_ACT_NONE);
    g2.use_gate_scalar = 1; g2.gate_scalar = it;
    VITA_TRY(gemm_f32(g2, s));                                   // logits.T
    hipLaunchKernelGGL(ce_diag_kernel, rows, blk, 0, s, w.L, B, w.l1, w.lse1);    // loss_i2a, :19
    hipLaunchKernelGGL(ce_diag_kernel, rows, blk, 0, s, w.LT, B, w.l2, w.lse2);   // loss_a2i, :20
    hipLaunchKernelGGL(pair_mean_kernel, dim3(1), blk, 0, s, w.l1, w.l2, B, 0.5 / (double)B, loss);   // :22
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// After nce_forward on the same workspace: da (+)= gout[0] * d loss / d a, db likewise.
int nce_backward(int B, int D, float temperature, const float* gout, float* da, int64_t ld_da, int acc_a, float* db, int64_t ld_db, int acc_b,
                 void* ws, hipStream_t s) {
    NceWs w = carve_nce(ws, (size_t)B, (size_t)D);
    const float c = 0.5f / (float)B / temperature;
    const int eb = (int)(((int64_t)B * B + CT - 1) / CT);
    hipLaunchKernelGGL(ce_diag_bwd_kernel, dim3(eb > 4096 ? 4096 : eb), dim3(CT), 0, s, w.L, w.lse1, B, gout, c);    // G1 = d/d logits
    hipLaunchKernelGGL(ce_diag_bwd_kernel, dim3(eb > 4096 ? 4096 : eb), dim3(CT), 0, s, w.LT, w.lse2, B, gout, c);   // G2 = d/d logits.T
    // d an = G1 . bn + G2^T . bn ;  d bn = G1^T . an + G2 . an
    VITA_TRY(gemm_f32(dgrad(w.L, B, w.bn, D, w.dan, D, B, D, B), s));
    vita_linear_args t1 = wgrad(w.LT, B, w.bn, D, w.dan, D, B, D, B);
    t1.resid = w.dan; t1.ld_resid = D;
    VITA_TRY(gemm_f32(t1, s));
    VITA_TRY(gemm_f32(dgrad(w.LT, B, w.an, D, w.dbn, D, B, D, B), s));
    vita_linear_args t2 = wgrad(w.L, B, w.an, D, w.dbn, D, B, D, B);
    t2.resid = w.dbn; t2.ld_resid = D;
    VITA_TRY(gemm_f32(t2, s));
    const dim3 rows((B + 3) / 4), blk(CT);
    if (da) hipLaunchKernelGGL(rownorm_bwd_kernel, rows, blk, 0, s, w.an, w.dan, w.inva, da, ld_da, B, D, acc_a);
    if (db) hipLaunchKernelGGL(rownorm_bwd_kernel, rows, blk, 0, s, w.bn, w.dbn, w.invb, db, ld_db, B, D, acc_b);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

extern "C" {

size_t vita_contrastive_loss_workspace_bytes(int32_t B, int32_t D) {
    if (B <= 0 || D <= 0) return 0;
    return kSplitkBytes + nce_workspace_bytes(B, D);
}

int vita_contrastive_loss_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t B, int32_t D, float temperature, float* loss,
                              const float* gout, float* grad_a, int64_t ld_ga, float* grad_b, int64_t ld_gb, void* workspace,
                              size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_contrastive_loss_f32";
    VITA_REQUIRE(a && b && loss && B >= 1 && D >= 1 && lda >= D && ldb >= D, "%s: bad arguments", who);
    VITA_REQUIRE(temperature > 0.f, "%s: temperature must be > 0", who);
    VITA_REQUIRE(gout != nullptr || (grad_a == nullptr && grad_b == nullptr), "%s: gradients need gout", who);
    VITA_REQUIRE((grad_a == nullptr || ld_ga >= D) && (grad_b == nullptr || ld_gb >= D), "%s: bad gradient strides", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_contrastive_loss_workspace_bytes(B, D), who));
    hipStream_t s = (hipStream_t)stream;
    SplitkScope sk(workspace, kSplitkBytes);
    void* ws = static_cast<char*>(workspace) + kSplitkBytes;
    VITA_TRY(nce_forward(a, lda, b, ldb, B, D, temperature, loss, ws, s));
    if (gout && (grad_a || grad_b)) VITA_TRY(nce_backward(B, D, temperature, gout, grad_a, ld_ga, 0, grad_b, ld_gb, 0, ws, s));
    return VITA_OK;
}

int vita_gaussian_kl_fwd_f32(const float* mean, int64_t ld_mean, const float* logvar, int64_t ld_logvar, int32_t B, int32_t Cc, float* row_scratch,
                             float* out, vita_stream_t stream) {
    VITA_REQUIRE(mean && logvar && row_scratch && out && B >= 1 && Cc >= 1 && ld_mean >= Cc && ld_logvar >= Cc, "vita_gaussian_kl_fwd_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(kl_rows_kernel, dim3((B + 3) / 4), dim3(CT), 0, s, mean, ld_mean, logvar, ld_logvar, B, Cc, row_scratch);
    hipLaunchKernelGGL(pair_mean_kernel, dim3(1), dim3(CT), 0, s, row_scratch, (const float*)nullptr, B, 1.0 / (double)B, out);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_gaussian_kl_bwd_f32(const float* mean, int64_t ld_mean, const float* logvar, int64_t ld_logvar, int32_t B, int32_t Cc, const float* gout,
                             float* grad_mean, int64_t ld_gm, float* grad_logvar, int64_t ld_gv, vita_stream_t stream) {
    VITA_REQUIRE(mean && logvar && gout && B >= 1 && Cc >= 1 && ld_mean >= Cc && ld_logvar >= Cc, "vita_gaussian_kl_bwd_f32: bad arguments");
    VITA_REQUIRE((grad_mean == nullptr || ld_gm >= Cc) && (grad_logvar == nullptr || ld_gv >= Cc), "vita_gaussian_kl_bwd_f32: bad gradient strides");
    if (!grad_mean && !grad_logvar) return VITA_OK;
    const int blocks = (int)(((int64_t)B * Cc + CT - 1) / CT);
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(CT), 0, (hipStream_t)stream, mean, ld_mean, logvar, ld_logvar, B, Cc,
                       gout, grad_mean, ld_gm, grad_logvar, ld_gv);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // extern "C"
