// bf16-operand variants of the inference entry points (north star (b): "bf16/fp32 MFMA GEMMs"; BASELINE configs[2]).
// GEMM operands (weights, and the activations feeding a GEMM) are bf16; accumulation, bias, activations, the gate /
// residual stream h, the Euler state x, LayerNorm statistics and the whole time path stay fp32.  Every GEMM input is
// produced in bf16 by the kernel that computes it (LayerNorm kernel, GEMM epilogues), so nothing is converted twice.
// v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, which moves these GEMMs from MFMA-bound to LDS/HBM-bound.
// Stated tolerance vs the fp32 reference: see tests/test_gpu_bf16.py.
#include <stdlib.h>

#include "vita_host.h"

namespace vita {

static inline int r8(int v) { return (v + 7) & ~7; }

static vita_linear_args lin_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* b, float* Y, int64_t ldy, int M,
                                 int N, int Kp, int act) {
    vita_linear_args a{};
    a.X = reinterpret_cast<const float*>(X); a.ldx = ldx; a.W = reinterpret_cast<const float*>(W); a.ldw = ldw; a.bias = b;
    a.Y = Y; a.ldy = ldy; a.M = M; a.N = N; a.K = Kp; a.act = act;
    return a;
}

struct FlowWsB {
    float *ts, *sin, *hid, *silu, *mod, *h;
    uint16_t *hnb, *ub, *xb0, *xb1;
    float *stat, *c1, *c2, *c1f, *c2f;   // LayerNorm folded into the GEMM epilogues (as in vita_api.hip), tables from the bf16 weights
    float* sync;                         // hand-off state + phase table of the persistent launch (vita_ode_persistent.hip)
    size_t bytes;
};
// the fold needs interior 128 x 128 tiles; VITA_NO_LN_FOLD=1 keeps the LayerNorm kernels
static bool use_ln_fold_b(const vita_flownet_weights* w, int B) {
    static const bool off = getenv("VITA_NO_LN_FOLD") != nullptr && atoi(getenv("VITA_NO_LN_FOLD")) != 0;
    return !off && B >= 1024 && B % 128 == 0 && w->L >= 1 && w->H % 128 == 0 && w->H <= 2048 && w->FF % 128 == 0 && w->D % 128 == 0;
}
static uint16_t* take16(Carver& c, size_t n) { return reinterpret_cast<uint16_t*>(c.take((n + 1) / 2)); }
static FlowWsB carve_flow_b(void* ws, const vita_flownet_weights* w, int B, int T) {
    Carver c(ws);
    FlowWsB f;
    f.ts = c.take((size_t)T);
    f.sin = c.take((size_t)T * w->TE);
    f.hid = c.take((size_t)T * 4 * w->TE);
    f.silu = c.take((size_t)T * w->H);
    f.mod = c.take((size_t)T * w->L * 3 * w->H);
    f.h = c.take((size_t)B * w->H);
    f.hnb = take16(c, (size_t)B * r8(w->H));
    f.ub = take16(c, (size_t)B * r8(w->FF));
    f.xb0 = take16(c, (size_t)B * r8(w->D));
    f.xb1 = take16(c, (size_t)B * r8(w->D));
    f.stat = f.c1 = f.c2 = f.c1f = f.c2f = f.sync = nullptr;
    if (use_ln_fold_b(w, B)) {
        f.sync = c.take(ode_persistent_sync_bytes(T, w->L) / sizeof(float));
        f.stat = c.take((size_t)(w->H / 128) * B * 2);
        f.c1 = c.take((size_t)T * w->L * w->FF);
        f.c2 = c.take((size_t)T * w->L * w->FF);
        f.c1f = c.take((size_t)w->D);
        f.c2f = c.take((size_t)w->D);
    }
    f.bytes = c.off;
    return f;
}

// fp32 time path, as in vita_api.hip (K rows: negligible work)
static int time_path_b(const vita_flownet_weights* w, const FlowWsB& f, int T, hipStream_t s) {
    const int TE = w->TE, H = w->H, L = w->L;
    VITA_TRY(launch_sinusoid(f.ts, f.sin, TE, T, TE, s));
    VITA_TRY(gemm_f32(lin(f.sin, TE, w->te1_w, TE, w->te1_b, f.hid, 4 * TE, T, 4 * TE, TE, VITA_ACT_MISH), s));
    VITA_TRY(gemm_f32(lin(f.hid, 4 * TE, w->te3_w, 4 * TE, w->te3_b, f.silu, H, T, H, 4 * TE, VITA_ACT_SILU), s));
    for (int l = 0; l < L; ++l)
        VITA_TRY(gemm_f32(lin(f.silu, H, w->mod_w[l], H, w->mod_b[l], f.mod + (size_t)l * 3 * H, (int64_t)L * 3 * H, T, 3 * H, H,
                              VITA_ACT_NONE), s));
    return VITA_OK;
}

static int check_bf16(const vita_flownet_weights* w, const vita_bf16_weights* bw, const char* who) {
    VITA_REQUIRE(bw != nullptr && bw->in_w && bw->out_w, "%s: null bf16 weights", who);
    for (int l = 0; l < w->L; ++l) VITA_REQUIRE(bw->fc1_w[l] && bw->fc2_w[l], "%s: null bf16 weight in layer %d", who, l);
    VITA_REQUIRE(w->H <= 2048, "%s: hidden_dim <= 2048", who);
    return VITA_OK;
}

// x (fp32, ld D) + xb (bf16, ld r8(D)) -> x_out / xb_out = x + dt * v(x, t_k)
static int flow_step_b(const vita_flownet_weights* w, const vita_bf16_weights* bw, const FlowWsB& f, const float* x, int64_t ldx,
                       const uint16_t* xb, const float* mod, float* x_out, int64_t ldo, uint16_t* xb_out, float dt, int B, hipStream_t s) {
    const int D = w->D, H = w->H, FF = w->FF, L = w->L;
    const int Dp = r8(D), Hp = r8(H), FFp = r8(FF);
    VITA_TRY(gemm_bf16_dma(lin_bf16(xb, Dp, bw->in_w, Dp, w->in_b, f.h, H, B, H, Dp, VITA_ACT_NONE), nullptr, 0, s));
    for (int l = 0; l < L; ++l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        VITA_TRY(launch_ln_mod(f.h, H, nullptr, 0, B, H, 1e-6f, VITA_PRO_LN_MOD, gamma + H, 0, gamma + 2 * H, 0, s, f.hnb, Hp));
        VITA_TRY(gemm_bf16_dma(lin_bf16(f.hnb, Hp, bw->fc1_w[l], Hp, w->fc1_b[l], nullptr, 0, B, FF, Hp, VITA_ACT_GELU_TANH), f.ub, FFp, s));
        vita_linear_args b = lin_bf16(f.ub, FFp, bw->fc2_w[l], FFp, w->fc2_b[l], f.h, H, B, H, FFp, VITA_ACT_NONE);
        b.gate = gamma; b.ld_gate = 0; b.resid = f.h; b.ld_resid = H;
        VITA_TRY(gemm_bf16_dma(b, nullptr, 0, s));
    }
    VITA_TRY(launch_ln_mod(f.h, H, nullptr, 0, B, H, 1e-5f, VITA_PRO_LN_AFFINE, w->norm_w, 0, w->norm_b, 0, s, f.hnb, Hp));
    vita_linear_args o = lin_bf16(f.hnb, Hp, bw->out_w, Hp, w->out_b, x_out, ldo, B, D, Hp, VITA_ACT_NONE);
    o.use_gate_scalar = 1; o.gate_scalar = dt; o.resid = x; o.ld_resid = ldx;
    return gemm_bf16_dma(o, xb_out, Dp, s);
}

// flow_step_b with the LayerNorms folded (vita_api.hip: flow_eval_folded): the producer of h also writes bf16(h * (1 + scale)) (or
// h * norm_w) and per-tile row statistics, the consumer finishes the normalisation on its accumulators; k indexes the c-tables.
static int flow_step_b_folded(const vita_flownet_weights* w, const vita_bf16_weights* bw, const FlowWsB& f, const float* x, int64_t ldx,
                              const uint16_t* xb, const float* mod, int k, float* x_out, int64_t ldo, uint16_t* xb_out, float dt, int B,
                              hipStream_t s) {
    const int D = w->D, H = w->H, FF = w->FF, L = w->L, nseg = H / 128;
    auto producer = [&](const vita_linear_args& a, const float* next_mul, int add_one) -> int {
        EpiFold e;
        e.aux_bf = f.hnb; e.ld_aux_bf = H; e.aux_mul = next_mul; e.aux_add_one = add_one; e.stat_out = f.stat;
        return gemm_bf16_dma(a, nullptr, 0, s, nullptr, &e);
    };
    VITA_TRY(producer(lin_bf16(xb, D, bw->in_w, D, w->in_b, f.h, H, B, H, D, VITA_ACT_NONE), mod + H, 1));
    for (int l = 0; l < L; ++l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        EpiFold c;
        c.ln_stats = f.stat; c.ln_c1 = f.c1 + ((size_t)k * L + l) * FF; c.ln_nseg = nseg; c.ln_fold_eps = 1e-6f;
        VITA_TRY(gemm_bf16_dma(lin_bf16(f.hnb, H, bw->fc1_w[l], H, f.c2 + ((size_t)k * L + l) * FF, nullptr, 0, B, FF, H, VITA_ACT_GELU_TANH),
                               f.ub, FF, s, nullptr, &c));
        vita_linear_args b = lin_bf16(f.ub, FF, bw->fc2_w[l], FF, w->fc2_b[l], f.h, H, B, H, FF, VITA_ACT_NONE);
        b.gate = gamma; b.ld_gate = 0; b.resid = f.h; b.ld_resid = H;
        if (l + 1 < L) VITA_TRY(producer(b, mod + (size_t)(l + 1) * 3 * H + H, 1));
        else VITA_TRY(producer(b, w->norm_w, 0));
    }
    vita_linear_args o = lin_bf16(f.hnb, H, bw->out_w, H, f.c2f, x_out, ldo, B, D, H, VITA_ACT_NONE);
    o.use_gate_scalar = 1; o.gate_scalar = dt; o.resid = x; o.ld_resid = ldx;
    EpiFold c;
    c.ln_stats = f.stat; c.ln_c1 = f.c1f; c.ln_nseg = nseg; c.ln_fold_eps = 1e-5f;
    return gemm_bf16_dma(o, xb_out, D, s, nullptr, &c);
}

// Columns D..r8(D)-1 of the bf16 state must be zero for the next in_proj: they are, because the packers zero-fill and
// an epilogue never writes past N = D (the workspace is zeroed once per call when D % 8 != 0).
static int ode_sample_b(const vita_flownet_weights* w, const vita_bf16_weights* bw, const float* x0, int64_t ldx0, const uint16_t* x0b,
                        float* x_out, int64_t ldo, uint16_t* xb_final, int B, int K, void* ws, hipStream_t s) {
    const int H = w->H, L = w->L, D = w->D;
    FlowWsB f = carve_flow_b(ws, w, B, K);
    if (r8(D) != D || r8(H) != H || r8(w->FF) != w->FF)
        VITA_CHECK_HIP(hipMemsetAsync(f.hnb, 0, (size_t)((char*)ws + f.bytes - (char*)f.hnb), s));
    VITA_TRY(launch_step_times(f.ts, K, s));
    VITA_TRY(time_path_b(w, f, K, s));
    const bool fold = f.stat != nullptr;
    if (fold) VITA_TRY(launch_ln_fold_tables_all(w, f.mod, K, f.c1, f.c2, f.c1f, f.c2f, s, bw));
    const float dt = (float)(1.0 / (double)K);
    if (fold && ode_persistent_eligible(w, B, x0, ldx0, x_out, ldo))      // all K steps in one persistent launch
        return launch_ode_persistent_bf16(w, bw, x0, ldx0, x0b, x_out, ldo, f.xb0, xb_final, B, K, dt, f.h, f.hnb, f.ub, f.stat, f.mod, f.c1,
                                          f.c2, f.c1f, f.c2f, f.sync, s);
    const float* src = x0;
    int64_t lds = ldx0;
    const uint16_t* srcb = x0b;
    for (int k = 0; k < K; ++k) {
        uint16_t* dstb = (k == K - 1 && xb_final) ? xb_final : ((k & 1) ? f.xb1 : f.xb0);
        if (fold) VITA_TRY(flow_step_b_folded(w, bw, f, src, lds, srcb, f.mod + (size_t)k * L * 3 * H, k, x_out, ldo, dstb, dt, B, s));
        else VITA_TRY(flow_step_b(w, bw, f, src, lds, srcb, f.mod + (size_t)k * L * 3 * H, x_out, ldo, dstb, dt, B, s));
        src = x_out; lds = ldo; srcb = dstb;
    }
    return VITA_OK;
}

struct DecWsB { uint16_t *a, *b; size_t bytes; };
static DecWsB carve_dec_b(void* ws, const vita_decoder_weights* w, int B) {
    Carver c(ws);
    DecWsB d;
    d.a = take16(c, (size_t)B * r8(w->HD));
    d.b = take16(c, (size_t)B * r8(w->HD));
    d.bytes = c.off;
    return d;
}
static int decoder_b(const vita_decoder_weights* w, const vita_bf16_weights* bw, const uint16_t* zb, float* out, int64_t ldo, int B, void* ws,
                     hipStream_t s) {
    const int Dp = r8(w->D), HDp = r8(w->HD);
    DecWsB d = carve_dec_b(ws, w, B);
    if (HDp != w->HD) VITA_CHECK_HIP(hipMemsetAsync(d.a, 0, d.bytes, s));
    VITA_TRY(gemm_bf16_dma(lin_bf16(zb, Dp, bw->dec_in_w, Dp, w->in_b, nullptr, 0, B, w->HD, Dp, VITA_ACT_NONE), d.a, HDp, s));
    for (int i = 0; i < w->NL; ++i) {
        VITA_TRY(gemm_bf16_dma(lin_bf16(d.a, HDp, bw->dec_fc1_w[i], HDp, w->fc1_b[i], nullptr, 0, B, w->HD, HDp, VITA_ACT_GELU_ERF), d.b, HDp, s));
        VITA_TRY(gemm_bf16_dma(lin_bf16(d.b, HDp, bw->dec_fc2_w[i], HDp, w->fc2_b[i], nullptr, 0, B, w->HD, HDp, VITA_ACT_NONE), d.a, HDp, s));
    }
    return gemm_bf16_dma(lin_bf16(d.a, HDp, bw->dec_out_w, HDp, w->out_b, out, ldo, B, w->OUT, HDp, VITA_ACT_NONE), nullptr, 0, s);
}

}  // namespace vita

using namespace vita;

extern "C" {

int vita_linear_bf16(const vita_linear_args* a, uint16_t* y_bf16, int64_t ld_ybf, vita_stream_t stream) {
    const char* who = "vita_linear_bf16";
    VITA_REQUIRE(a != nullptr, "%s: null args", who);
    VITA_REQUIRE(a->X && a->W && (a->Y || y_bf16), "%s: X, W and one of Y / y_bf16 must be non-null", who);
    VITA_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "%s: bad shape", who);
    if (a->M == 0) return VITA_OK;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    VITA_REQUIRE(a->K % 8 == 0 && a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->ldx >= a->K && a->ldw >= a->K && al(a->X) && al(a->W),
                 "%s: bf16 operands need K, ldx, ldw multiples of 8 and 16-byte aligned rows (pack with vita_pack_bf16_f32)", who);
    VITA_REQUIRE(a->pro == VITA_PRO_NONE && !a->a_col && !a->b_col && a->act_grad_z == nullptr, "%s: prologue / transposed operands are fp32-only", who);
    VITA_REQUIRE(a->act >= VITA_ACT_NONE && a->act <= VITA_ACT_MISH, "%s: unknown activation", who);
    VITA_REQUIRE(a->Y == nullptr || a->ldy >= a->N, "%s: ldy < N", who);
    VITA_REQUIRE(y_bf16 == nullptr || ld_ybf >= a->N, "%s: ld_ybf < N", who);
    VITA_REQUIRE(a->gate == nullptr || a->ld_gate == 0 || a->ld_gate >= a->N, "%s: bad ld_gate", who);
    VITA_REQUIRE(a->resid == nullptr || a->ld_resid >= a->N, "%s: bad ld_resid", who);
    return gemm_bf16_dma(*a, y_bf16, ld_ybf, (hipStream_t)stream);
}

int vita_linear_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N, int32_t R,
                        void* splitk_ws, size_t splitk_ws_bytes, vita_stream_t stream) {
    const char* who = "vita_linear_bf16_tn";
    VITA_REQUIRE(A && B && C, "%s: null argument", who);
    VITA_REQUIRE(M > 0 && N > 0 && R > 0 && ldc >= N, "%s: bad shape M=%d N=%d R=%d ldc=%lld", who, M, N, R, (long long)ldc);
    VITA_REQUIRE(gemm_bf16_tn_eligible(A, lda, B, ldb, M, N, R),
                 "%s: operands must be 16-byte aligned bf16 rows with lda >= M, ldb >= N multiples of 8, and M, N multiples of 8", who);
    VITA_REQUIRE(splitk_ws == nullptr || (reinterpret_cast<uintptr_t>(splitk_ws) & 255u) == 0, "%s: splitk_ws must be 256-byte aligned", who);
    vita_linear_args a{};
    a.Y = C; a.ldy = ldc; a.M = M; a.N = N; a.K = R; a.act = VITA_ACT_NONE;
    a.splitk_ws = splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    return gemm_bf16_wgrad_tn(A, lda, B, ldb, a, (hipStream_t)stream);
}

size_t vita_generate_actions_bf16_workspace_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw, int32_t O, int32_t B,
                                                  int32_t num_steps) {
    if (!fw || !dw || B <= 0 || num_steps <= 0 || O <= 0) return 0;
    Carver c(nullptr);
    c.take(kSplitkBytes / sizeof(float));
    c.take((size_t)B * fw->D);
    take16(c, (size_t)B * r8(fw->D));
    take16(c, (size_t)B * r8(O));
    const size_t a = carve_flow_b(nullptr, fw, B, num_steps).bytes, b = carve_dec_b(nullptr, dw, B).bytes;
    return c.off + (a > b ? a : b);
}

int vita_generate_actions_bf16(const float* obs_b, int32_t O, const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                               const vita_bf16_weights* bw, const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                               float* latents_out, int32_t B, int32_t num_steps, void* workspace, size_t workspace_bytes,
                               vita_stream_t stream) {
    const char* who = "vita_generate_actions_bf16";
    VITA_TRY(check_flow_weights(fw, who));
    VITA_TRY(check_dec_weights(dw, who));
    VITA_TRY(check_bf16(fw, bw, who));
    VITA_REQUIRE(bw->obs_w && bw->dec_in_w && bw->dec_out_w, "%s: null bf16 obs/decoder weights", who);
    for (int i = 0; i < dw->NL; ++i) VITA_REQUIRE(bw->dec_fc1_w[i] && bw->dec_fc2_w[i], "%s: null bf16 decoder weight %d", who, i);
    VITA_REQUIRE(fw->D == dw->D && num_steps >= 1 && B >= 0 && O > 0, "%s: bad dims", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(obs_b && obs_features && actions && ldobs >= O && lda >= dw->OUT, "%s: bad pointers/strides", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_generate_actions_bf16_workspace_bytes(fw, dw, O, B, num_steps), who));
    hipStream_t s = (hipStream_t)stream;
    const int D = fw->D, Dp = r8(D), Op = r8(O);
    Carver c(workspace);
    float* sk = c.take(kSplitkBytes / sizeof(float));
    SplitkScope skscope(sk, kSplitkBytes);     // the fp32 time-path GEMMs
    float* lat = c.take((size_t)B * D);
    uint16_t* latb = take16(c, (size_t)B * Dp);
    uint16_t* obsb = take16(c, (size_t)B * Op);
    void* rest = static_cast<char*>(workspace) + c.off;
    if (Dp != D) VITA_CHECK_HIP(hipMemsetAsync(latb, 0, (size_t)B * Dp * 2, s));
    VITA_TRY(vita_pack_bf16_f32(obs_features, ldobs, obsb, Op, B, O, s));
    VITA_TRY(gemm_bf16_dma(lin_bf16(obsb, Op, bw->obs_w, Op, obs_b, lat, D, B, D, Op, VITA_ACT_NONE), latb, Dp, s));
    VITA_TRY(ode_sample_b(fw, bw, lat, D, latb, lat, D, latb, B, num_steps, rest, s));
    if (latents_out) VITA_CHECK_HIP(hipMemcpyAsync(latents_out, lat, (size_t)B * D * 4, hipMemcpyDeviceToDevice, s));
    return decoder_b(dw, bw, latb, actions, lda, B, rest, s);
}


int vita_generate_actions_bf16_status(const vita_flownet_weights* fw, int32_t O, int32_t B, int32_t num_steps, const void* workspace,
                                      int32_t* status) {
    VITA_REQUIRE(fw && workspace && status && B > 0 && num_steps > 0 && O > 0, "vita_generate_actions_bf16_status: bad arguments");
    *status = 0;
    Carver c(const_cast<void*>(workspace));      // the carve of vita_generate_actions_bf16
    c.take(kSplitkBytes / sizeof(float));
    c.take((size_t)B * fw->D);
    take16(c, (size_t)B * r8(fw->D));
    take16(c, (size_t)B * r8(O));
    FlowWsB f = carve_flow_b(static_cast<char*>(const_cast<void*>(workspace)) + c.off, fw, B, num_steps);
    if (f.sync == nullptr) return VITA_OK;
    unsigned v = 0;
    VITA_CHECK_HIP(hipMemcpy(&v, reinterpret_cast<const char*>(f.sync) + ode_persistent_status_offset(), sizeof(v), hipMemcpyDeviceToHost));
    *status = (int32_t)v;
    return VITA_OK;
}

}  // extern "C"
