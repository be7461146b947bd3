// Alternate matchers' INFERENCE paths: SimpleMeanFlowNet.forward (simple_mean_flow_net.py:63-81) and
// MeanFlowMatcher.sample (mean_flow_matcher.py:201-233) as orchestrations of the fused linear, plus the small kernels the
// Consistency-FM sampler (vita_api.hip) needs.  Training of these matchers (JVP through the net, adaptive / dispersive
// losses) is not built.
#include <math.h>

#include "vita_host.h"

namespace vita {

namespace {

constexpr int MT = 256;

__global__ void step_times_affine_kernel(float* __restrict__ ts, int K, double scale, double offset) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) ts[k] = (float)((double)k / (double)K * scale + offset);
}
// z_new = (z + (v + coef * (c_v * v - c_z * z)) * dt) + s_noise * noise     (consistency_flow_matcher.py:94-96, same op order)
__global__ __launch_bounds__(MT) void sde_update_kernel(const float* z, int64_t ldz, const float* __restrict__ v, const float* __restrict__ noise,
                                                       float* out, int64_t ldo, int B, int D, float c_v, float c_z, float coef, float dt,
                                                       float s_noise) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * MT;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const float zz = z[(int64_t)r * ldz + c], vv = v[i];
        const float pred = vv + coef * (c_v * vv - c_z * zz);
        out[(int64_t)r * ldo + c] = (zz + pred * dt) + s_noise * noise[i];
    }
}
// out[m, :] = x[m, :] + r[m * ldr + :]   (ldr = 0: one row for the whole batch)
__global__ __launch_bounds__(MT) void add_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ r, int64_t ldr,
                                                     float* __restrict__ out, int64_t ldo, int M, int N) {
    const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * MT;
    for (int64_t i = (int64_t)blockIdx.x * MT + threadIdx.x; i < n; i += stride) {
        const int m = (int)(i / N), c = (int)(i - (int64_t)m * N);
        out[(int64_t)m * ldo + c] = x[(int64_t)m * ldx + c] + r[(int64_t)m * ldr + c];
    }
}
// per-step (t, h) of the MeanFlow sampler: t = 1 - k/K, h = t - (1 - (k+1)/K), in double as the reference's python floats
__global__ void meanflow_times_kernel(float* __restrict__ t, float* __restrict__ h, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) {
        const double ts = 1.0 - (double)k / (double)K, rs = 1.0 - (double)(k + 1) / (double)K;
        t[k] = (float)ts;
        h[k] = (float)(ts - rs);
    }
}

inline unsigned mgrid(int64_t n) {
    const int64_t b = (n + MT - 1) / MT;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

int launch_step_times_affine(float* ts, int K, double scale, double offset, hipStream_t s) {
    hipLaunchKernelGGL(step_times_affine_kernel, dim3((K + 63) / 64), dim3(64), 0, s, ts, K, scale, offset);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int launch_sde_update(const float* z, int64_t ldz, const float* v, const float* noise, float* out, int64_t ldo, int B, int D, float c_v,
                      float c_z, float coef, float dt, float s_noise, hipStream_t s) {
    hipLaunchKernelGGL(sde_update_kernel, dim3(mgrid((int64_t)B * D)), dim3(MT), 0, s, z, ldz, v, noise, out, ldo, B, D, c_v, c_z, coef, dt, s_noise);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
int launch_add_rows(const float* x, int64_t ldx, const float* r, int64_t ldr, float* out, int64_t ldo, int M, int N, hipStream_t s) {
    hipLaunchKernelGGL(add_rows_kernel, dim3(mgrid((int64_t)M * N)), dim3(MT), 0, s, x, ldx, r, ldr, out, ldo, M, N);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

struct MfWs {
    float *sk, *t, *h, *e1, *tc, *x, *xc, *u;
    size_t bytes;
};
// T = rows of the time conditioning (B for per-row (t, h), num_steps when sampling)
static MfWs carve_mf(void* ws, const vita_meanflow_weights* w, size_t B, size_t T) {
    Carver c(ws);
    MfWs m;
    m.sk = c.take(kSplitkBytes / sizeof(float));
    m.t = c.take(T); m.h = c.take(T);
    m.e1 = c.take(T * w->H);
    m.tc = c.take(T * w->H);
    m.x = c.take(B * w->H);
    m.xc = c.take(B * w->H);
    m.u = c.take(B * w->FF);
    m.bytes = c.off;
    return m;
}

static int check_mf(const vita_meanflow_weights* w, const char* who) {
    VITA_REQUIRE(w != nullptr, "%s: null weights", who);
    VITA_REQUIRE(w->D > 0 && w->H > 0 && w->FF > 0 && w->L >= 0 && w->L <= VITA_MAX_LAYERS, "%s: bad dims D=%d H=%d FF=%d L=%d", who, w->D, w->H,
                 w->FF, w->L);
    VITA_REQUIRE(w->t1_w && w->t1_b && w->t2_w && w->t2_b && w->h1_w && w->h1_b && w->h2_w && w->h2_b && w->in_w && w->in_b && w->u_w && w->u_b,
                 "%s: null weight pointer", who);
    for (int l = 0; l < w->L; ++l) VITA_REQUIRE(w->fc1_w[l] && w->fc1_b[l] && w->fc2_w[l] && w->fc2_b[l], "%s: null weight pointer in layer %d", who, l);
    return VITA_OK;
}

// time_cond = t_embed(t) + h_embed(h) for T rows (simple_mean_flow_net.py:64-67); t, h device vectors of T floats
static int mf_time_cond(const vita_meanflow_weights* w, const MfWs& m, const float* t, const float* h, int T, hipStream_t s) {
    const int H = w->H;
    VITA_TRY(gemm_f32(lin(t, 1, w->t1_w, 1, w->t1_b, m.e1, H, T, H, 1, VITA_ACT_SILU), s));
    VITA_TRY(gemm_f32(lin(m.e1, H, w->t2_w, H, w->t2_b, m.tc, H, T, H, H, VITA_ACT_NONE), s));
    VITA_TRY(gemm_f32(lin(h, 1, w->h1_w, 1, w->h1_b, m.e1, H, T, H, 1, VITA_ACT_SILU), s));
    vita_linear_args a = lin(m.e1, H, w->h2_w, H, w->h2_b, m.tc, H, T, H, H, VITA_ACT_NONE);
    a.resid = m.tc; a.ld_resid = H;
    return gemm_f32(a, s);
}

// trunk: x = input_layer(x_in); for each layer x = fc2(silu(fc1(x + tc))) + x   (:69-75).  tc row stride ld_tc (0: shared row)
static int mf_trunk(const vita_meanflow_weights* w, const MfWs& m, const float* x_in, int64_t ldx, const float* tc, int64_t ld_tc, int B,
                    hipStream_t s) {
    const int D = w->D, H = w->H, FF = w->FF;
    VITA_TRY(gemm_f32(lin(x_in, ldx, w->in_w, D, w->in_b, m.x, H, B, H, D, VITA_ACT_NONE), s));
    for (int l = 0; l < w->L; ++l) {
        VITA_TRY(launch_add_rows(m.x, H, tc, ld_tc, m.xc, H, B, H, s));
        VITA_TRY(gemm_f32(lin(m.xc, H, w->fc1_w[l], H, w->fc1_b[l], m.u, FF, B, FF, H, VITA_ACT_SILU), s));
        vita_linear_args b = lin(m.u, FF, w->fc2_w[l], FF, w->fc2_b[l], m.x, H, B, H, FF, VITA_ACT_NONE);
        b.resid = m.x; b.ld_resid = H;
        VITA_TRY(gemm_f32(b, s));
    }
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

extern "C" {

size_t vita_meanflow_forward_workspace_bytes(const vita_meanflow_weights* w, int32_t B) {
    if (!w || B <= 0) return 0;
    return carve_mf(nullptr, w, (size_t)B, (size_t)B).bytes;
}

int vita_meanflow_forward_f32(const vita_meanflow_weights* w, const float* x, int64_t ldx, const float* t, const float* h, float* u, int64_t ldu,
                              float* v, int64_t ldv, int32_t B, void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_meanflow_forward_f32";
    VITA_TRY(check_mf(w, who));
    VITA_REQUIRE(B >= 0, "%s: B < 0", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(x && t && h && u && ldx >= w->D && ldu >= w->D, "%s: bad x/t/h/u", who);
    VITA_REQUIRE(v == nullptr || (w->v_w && w->v_b && ldv >= w->D), "%s: v output needs v_output_layer weights and ldv >= D", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_meanflow_forward_workspace_bytes(w, B), who));
    hipStream_t s = (hipStream_t)stream;
    MfWs m = carve_mf(workspace, w, (size_t)B, (size_t)B);
    SplitkScope sk(m.sk, kSplitkBytes);
    VITA_TRY(mf_time_cond(w, m, t, h, B, s));
    VITA_TRY(mf_trunk(w, m, x, ldx, m.tc, w->H, B, s));
    VITA_TRY(gemm_f32(lin(m.x, w->H, w->u_w, w->H, w->u_b, u, ldu, B, w->D, w->H, VITA_ACT_NONE), s));              // :77
    if (v) VITA_TRY(gemm_f32(lin(m.x, w->H, w->v_w, w->H, w->v_b, v, ldv, B, w->D, w->H, VITA_ACT_NONE), s));      // :78
    return VITA_OK;
}

size_t vita_meanflow_sample_workspace_bytes(const vita_meanflow_weights* w, int32_t B, int32_t num_steps) {
    if (!w || B <= 0 || num_steps <= 0) return 0;
    return carve_mf(nullptr, w, (size_t)B, (size_t)num_steps).bytes;
}

int vita_meanflow_sample_f32(const vita_meanflow_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int32_t B,
                             int32_t num_steps, void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_meanflow_sample_f32";
    VITA_TRY(check_mf(w, who));
    VITA_REQUIRE(num_steps >= 1, "%s: num_steps must be >= 1 (got %d)", who, num_steps);
    VITA_REQUIRE(B >= 0, "%s: B < 0", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(x0 && x_out && ldx0 >= w->D && ldo >= w->D, "%s: bad x0/x_out", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_meanflow_sample_workspace_bytes(w, B, num_steps), who));
    hipStream_t s = (hipStream_t)stream;
    const int K = num_steps, H = w->H;
    MfWs m = carve_mf(workspace, w, (size_t)B, (size_t)K);
    SplitkScope sk(m.sk, kSplitkBytes);
    hipLaunchKernelGGL(meanflow_times_kernel, dim3((K + 63) / 64), dim3(64), 0, s, m.t, m.h, K);
    VITA_CHECK_HIP(hipGetLastError());
    VITA_TRY(mf_time_cond(w, m, m.t, m.h, K, s));      // every row of the batch shares (t, h): one conditioning row per step
    const float* src = x0;
    int64_t lds = ldx0;
    for (int k = 0; k < K; ++k) {
        VITA_TRY(mf_trunk(w, m, src, lds, m.tc + (size_t)k * H, 0, B, s));
        // x <- x - h * u   (mean_flow_matcher.py:222), fused into u_output_layer's epilogue
        const double hk = (1.0 - (double)k / (double)K) - (1.0 - (double)(k + 1) / (double)K);
        vita_linear_args o = lin(m.x, H, w->u_w, H, w->u_b, x_out, ldo, B, w->D, H, VITA_ACT_NONE);
        o.use_gate_scalar = 1; o.gate_scalar = -(float)hk; o.resid = src; o.ld_resid = lds;
        VITA_TRY(gemm_f32(o, s));
        src = x_out;
        lds = ldo;
    }
    return VITA_OK;
}

}  // extern "C"
