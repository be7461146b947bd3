// C-ABI entry points that chain the fused GEMM kernels into the reference's operators:
// SimpleFlowNet.forward, TorchFlowMatcher.sample, SimpleActionDecoder.forward and
// VitaPolicy.generate_actions.  Host code only validates shapes, carves the caller's
// workspace and enqueues launches on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "vita_host.h"

namespace vita {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Size of the split-K region at the head of every carve below: kSplitkBytes for the workspace-taking entry points, smaller
// inside the allocation-free entry point (vita_generate_actions_noalloc_f32), whose whole scratch is a fixed array.
static thread_local size_t tl_skb = kSplitkBytes;
// set while the allocation-free entry point runs: the workspace is a __device__ array of the code object, not hipMalloc memory
// (the in-launch hand-off then keeps its agent-scope acquire, vita_ode_resident.hip)
static thread_local bool tl_ws_symbol = false;
struct SkbScope {
    size_t saved;
    explicit SkbScope(size_t n) : saved(tl_skb) { tl_skb = n; }
    ~SkbScope() { tl_skb = saved; }
};

struct FlowWs {
    float *sk, *ts, *sin, *hid, *silu, *mod, *h, *hn, *u, *x;
    float *stat, *c1, *c2, *c1f, *c2f;   // folded LayerNorm (sampling at large batch): row statistics and the c-tables
    float* sync;                         // hand-off state of the persistent sampler (team counters, timeout / status words)
    float* xi;                           // weights-resident sampler: the Euler state in the blocked hand-off layout
    float *da, *db;                      // ... and the decoder's activations when it rides behind the last step
    float *thid, *tsilu;                 // ... and the time path's, when it runs in the same launch
    size_t bytes;
};

// LayerNorm folded into the GEMM epilogues instead of 5 LayerNorm launches per Euler step.  With m = 1 + scale:
//   fc1(LN(h) * m + shift) = rstd * (hm . W1^T - mean * c1) + c2,   hm = h * m,  c1 = W1 m,  c2 = W1 shift + b1,
// so the producer of h (in_proj / fc2) also writes hm and per-tile row statistics, the consumer (fc1 / out_proj) finishes
// the normalisation on its accumulators, and c1 / c2 are (K, L, FF) tables built once per solve (the modulation depends
// only on the step).  Needs every tile interior: B % 256 == 0 and 128-multiples for H, FF, D.  VITA_NO_LN_FOLD=1 disables.
static bool use_ln_fold(const vita_flownet_weights* w, int B) {
    static const bool off = getenv("VITA_NO_LN_FOLD") != nullptr && atoi(getenv("VITA_NO_LN_FOLD")) != 0;
    // round 2: below 3072 rows the fold's 64-row GEMMs (no split-K then) left most CUs without a tile (B = 1024 2.85 ms folded
    // against 1.85 ms unfolded).  Round 3: they split K and a second pass applies the folded epilogue (gemm_f32_fold), so the
    // fold starts at 1024 rows
    static const int min_rows = getenv("VITA_LN_FOLD_MIN_ROWS") != nullptr ? atoi(getenv("VITA_LN_FOLD_MIN_ROWS")) : 1024;
    return !off && B >= min_rows && B % 256 == 0 && w->L >= 1 && w->H % 128 == 0 && w->H <= 2048 && w->FF % 128 == 0 && w->D % 128 == 0;
}
// T = number of distinct timesteps (num_steps when sampling, B for per-row t); sampling: one modulation row per step serves the
// whole batch (the Euler sampler), so the in-launch samplers' tables and hand-off state may be needed
static FlowWs carve_flow(void* ws, const vita_flownet_weights* w, int B, int T, bool need_x, bool sampling) {
    Carver c(ws);
    FlowWs f;
    f.sk = c.take(tl_skb / sizeof(float));
    f.ts = c.take((size_t)T);
    f.sin = c.take((size_t)T * w->TE);
    f.hid = c.take((size_t)T * 4 * w->TE);
    f.silu = c.take((size_t)T * w->H);
    f.mod = c.take((size_t)T * w->L * 3 * w->H);
    const size_t Bp = ((size_t)B + 3) & ~size_t(3);   // the blocked hand-off layout of the resident sampler pads rows to 4
    f.h = c.take(Bp * w->H);
    f.hn = c.take((size_t)B * w->H);
    f.u = c.take(Bp * w->FF);
    f.x = need_x ? c.take((size_t)B * w->D) : nullptr;
    f.stat = f.c1 = f.c2 = f.c1f = f.c2f = f.sync = f.xi = f.da = f.db = f.thid = f.tsilu = nullptr;
    if (sampling && ode_resident_dims(w, B)) {
        f.sync = c.take(ode_persistent_sync_bytes(0, 0) / sizeof(float));   // hand-off words only
        f.xi = c.take(Bp * w->D);
        f.da = c.take(Bp * w->D);        // decoder hidden width == D on that path (ode_resident_decoder_ok)
        f.db = c.take(Bp * w->D);
        f.thid = c.take((size_t)32 * 4 * w->TE);
        f.tsilu = c.take((size_t)32 * w->H);
    }
    if (sampling && use_ln_fold(w, B)) {   // sampling only (per-row t has per-row modulation: no shared tables)
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

int check_flow_weights(const vita_flownet_weights* w, const char* who) {
    VITA_REQUIRE(w != nullptr, "%s: null weights", who);
    VITA_REQUIRE(w->D > 0 && w->H > 0 && w->FF > 0 && w->TE >= 4 && w->TE % 2 == 0,
                 "%s: bad dims D=%d H=%d FF=%d TE=%d", who, w->D, w->H, w->FF, w->TE);
    VITA_REQUIRE(w->L >= 0 && w->L <= VITA_MAX_LAYERS, "%s: L=%d outside [0,%d]", who, w->L, VITA_MAX_LAYERS);
    VITA_REQUIRE(w->in_w && w->in_b && w->te1_w && w->te1_b && w->te3_w && w->te3_b && w->norm_w && w->norm_b && w->out_w &&
                     w->out_b,
                 "%s: null weight pointer", who);
    for (int l = 0; l < w->L; ++l)
        VITA_REQUIRE(w->fc1_w[l] && w->fc1_b[l] && w->fc2_w[l] && w->fc2_b[l] && w->mod_w[l] && w->mod_b[l],
                     "%s: null weight pointer in layer %d", who, l);
    return VITA_OK;
}

// Time path for T timesteps already in f.ts: sinusoid -> Linear+Mish -> Linear (+SiLU, the only
// form the modulators consume) -> per-layer modulator Linear into f.mod (T, L, 3H).
// simple_flow_net.py:66-71 and :27-30,41.
static int time_path(const vita_flownet_weights* w, const FlowWs& f, int T, hipStream_t s) {
    const int TE = w->TE, H = w->H, L = w->L;
    VITA_TRY(launch_sinusoid(f.ts, f.sin, TE, T, TE, s));
    VITA_TRY(gemm_f32(lin(f.sin, TE, w->te1_w, TE, w->te1_b, f.hid, 4 * TE, T, 4 * TE, TE, VITA_ACT_MISH), s));
    VITA_TRY(gemm_f32(lin(f.hid, 4 * TE, w->te3_w, 4 * TE, w->te3_b, f.silu, H, T, H, 4 * TE, VITA_ACT_SILU), s));
    for (int l = 0; l < L; ++l)
        VITA_TRY(gemm_f32(lin(f.silu, H, w->mod_w[l], H, w->mod_b[l], f.mod + (size_t)l * 3 * H, (int64_t)L * 3 * H, T, 3 * H, H,
                              VITA_ACT_NONE), s));
    return VITA_OK;
}

// One velocity-field evaluation given the modulation rows.
//   mod: pointer to this evaluation's (gamma|scale|shift) block of layer 0; ld_mod = 0 when the
//   same row serves the whole batch (sampling), L*3H when each row has its own t (training).
//   Output: Y = resid + gate_scalar * v   when resid != NULL (the Euler update, flow_matchers.py:71)
//           Y = v                          otherwise.
static int flow_eval(const vita_flownet_weights* w, const FlowWs& f, const float* x, int64_t ldx, const float* mod, int64_t ld_mod,
                     float* Y, int64_t ldy, const float* resid, int64_t ld_resid, float dt, int B, hipStream_t s) {
    const int D = w->D, H = w->H, FF = w->FF, L = w->L;
    VITA_TRY(gemm_f32(lin(x, ldx, w->in_w, D, w->in_b, f.h, H, B, H, D, VITA_ACT_NONE), s));        // :100
    for (int l = 0; l < L; ++l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        const float* scale = gamma + H;
        const float* shift = gamma + 2 * H;
        // LN + modulation as its own one-pass kernel (:18,44-45): fused into the GEMM's A staging it
        // cost every one of the FF/128 column tiles a statistics prologue (+20..40 us per GEMM).
        vita_linear_args a = lin(f.h, H, w->fc1_w[l], H, w->fc1_b[l], f.u, FF, B, FF, H, VITA_ACT_GELU_TANH);
        a.pro = VITA_PRO_LN_MOD; a.ln_eps = 1e-6f;
        a.pro_mul = scale; a.ld_pro_mul = ld_mod; a.pro_add = shift; a.ld_pro_add = ld_mod;
        if (H <= 2048 && !gemm_f32_small_ln_eligible(a)) {
            VITA_TRY(launch_ln_mod(f.h, H, f.hn, H, B, H, 1e-6f, VITA_PRO_LN_MOD, scale, ld_mod, shift, ld_mod, s));
            VITA_TRY(gemm_f32(lin(f.hn, H, w->fc1_w[l], H, w->fc1_b[l], f.u, FF, B, FF, H, VITA_ACT_GELU_TANH), s));
        } else {  // few rows: LayerNorm inside the few-row GEMM; very wide nets: fused into the generic GEMM's A staging
            VITA_TRY(gemm_f32(a, s));
        }
        vita_linear_args b = lin(f.u, FF, w->fc2_w[l], FF, w->fc2_b[l], f.h, H, B, H, FF, VITA_ACT_NONE);
        b.gate = gamma; b.ld_gate = ld_mod; b.resid = f.h; b.ld_resid = H;                            // :46
        VITA_TRY(gemm_f32(b, s));
    }
    vita_linear_args o = lin(f.h, H, w->out_w, H, w->out_b, Y, ldy, B, D, H, VITA_ACT_NONE);                       // :107
    o.pro = VITA_PRO_LN_AFFINE; o.ln_eps = 1e-5f;
    o.pro_mul = w->norm_w; o.ld_pro_mul = 0; o.pro_add = w->norm_b; o.ld_pro_add = 0;
    if (resid) { o.use_gate_scalar = 1; o.gate_scalar = dt; o.resid = resid; o.ld_resid = ld_resid; }
    if (H <= 2048 && !gemm_f32_small_ln_eligible(o)) {
        VITA_TRY(launch_ln_mod(f.h, H, f.hn, H, B, H, 1e-5f, VITA_PRO_LN_AFFINE, w->norm_w, 0, w->norm_b, 0, s));  // :81,106
        o.X = f.hn; o.pro = VITA_PRO_NONE; o.pro_mul = o.pro_add = nullptr;
    }
    VITA_TRY(gemm_f32(o, s));
    return VITA_OK;
}

static int fold_tables(const vita_flownet_weights* w, const FlowWs& f, int T, hipStream_t s) {
    return launch_ln_fold_tables_all(w, f.mod, T, f.c1, f.c2, f.c1f, f.c2f, s);
}

// flow_eval with the LayerNorms folded (sampling: one modulation row for the batch); k = step index into the c-tables
static int flow_eval_folded(const vita_flownet_weights* w, const FlowWs& f, const float* x, int64_t ldx, const float* mod, int k, float* Y,
                            int64_t ldy, const float* resid, int64_t ld_resid, float dt, int B, hipStream_t s) {
    const int D = w->D, H = w->H, FF = w->FF, L = w->L, nseg = H / 128;
    auto producer = [&](vita_linear_args a, const float* next_mul, int add_one) -> int {
        EpiFold e;
        e.aux_out = f.hn; e.ld_aux = H; e.aux_mul = next_mul; e.aux_add_one = add_one; e.stat_out = f.stat;
        return gemm_f32_fold(a, e, s);
    };
    VITA_TRY(producer(lin(x, ldx, w->in_w, D, w->in_b, f.h, H, B, H, D, VITA_ACT_NONE), mod + H, 1));            // h, h*(1+scale_0), stats
    for (int l = 0; l < L; ++l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        EpiFold c;
        c.ln_stats = f.stat; c.ln_c1 = f.c1 + ((size_t)k * L + l) * FF; c.ln_nseg = nseg; c.ln_fold_eps = 1e-6f;
        VITA_TRY(gemm_f32_fold(lin(f.hn, H, w->fc1_w[l], H, f.c2 + ((size_t)k * L + l) * FF, f.u, FF, B, FF, H, VITA_ACT_GELU_TANH), c, s));
        vita_linear_args b = lin(f.u, FF, w->fc2_w[l], FF, w->fc2_b[l], f.h, H, B, H, FF, VITA_ACT_NONE);
        b.gate = gamma; b.ld_gate = 0; b.resid = f.h; b.ld_resid = H;
        if (l + 1 < L) VITA_TRY(producer(b, mod + (size_t)(l + 1) * 3 * H + H, 1));
        else VITA_TRY(producer(b, w->norm_w, 0));
    }
    vita_linear_args o = lin(f.hn, H, w->out_w, H, f.c2f, Y, ldy, B, D, H, VITA_ACT_NONE);
    if (resid) { o.use_gate_scalar = 1; o.gate_scalar = dt; o.resid = resid; o.ld_resid = ld_resid; }
    EpiFold c;
    c.ln_stats = f.stat; c.ln_c1 = f.c1f; c.ln_nseg = nseg; c.ln_fold_eps = 1e-5f;
    return gemm_f32_fold(o, c, s);
}

// dec / actions / lda: the action decoder to run behind the last step when the weights-resident launch takes the solve;
// *dec_done says whether it did (otherwise the caller runs decoder_forward).
static int ode_sample(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int B, int K,
                      float* traj, void* ws, hipStream_t s, const vita_decoder_weights* dec = nullptr, float* actions = nullptr,
                      int64_t lda = 0, bool* dec_done = nullptr, const ResidentObs* obs = nullptr) {
    const int D = w->D, H = w->H, L = w->L;
    FlowWs f = carve_flow(ws, w, B, K, false, true);
    SplitkScope sk(f.sk, tl_skb);
    const float dt = (float)(1.0 / (double)K);  // python float 1.0/num_steps, rounded to fp32 at the multiply
    if (traj)
        VITA_CHECK_HIP(hipMemcpy2DAsync(traj, (size_t)D * 4, x0, (size_t)ldx0 * 4, (size_t)D * 4, (size_t)B, hipMemcpyDeviceToDevice, s));
    if (f.sync && f.xi && ode_resident_eligible(w, B, x0, ldx0, x_out, ldo)) {
        // serving batches: one launch, the net's weights resident in LDS for all K steps (vita_ode_resident.hip); the time
        // path and the decoder ride in the same launch where their shapes allow
        const bool tp = ode_resident_time_path(w, K);
        if (!tp) {
            VITA_TRY(launch_step_times(f.ts, K, s));
            VITA_TRY(time_path(w, f, K, s));
        }
        const bool fuse_dec = dec != nullptr && dec->D == w->D && ode_resident_decoder_ok(dec, actions, lda);
        VITA_TRY(launch_ode_resident(w, x0, ldx0, x_out, ldo, traj, B, K, dt, f.h, f.u, f.xi, f.mod, tp ? f.thid : nullptr,
                                     tp ? f.tsilu : nullptr, f.sync, tl_ws_symbol, s, fuse_dec ? dec : nullptr, f.da, f.db, actions, lda, obs));
        if (dec_done) *dec_done = fuse_dec;
        if (traj)
            VITA_CHECK_HIP(hipMemcpy2DAsync(x_out, (size_t)ldo * 4, traj + (size_t)K * B * D, (size_t)D * 4, (size_t)D * 4, (size_t)B,
                                            hipMemcpyDeviceToDevice, s));
        return VITA_OK;
    }
    VITA_TRY(launch_step_times(f.ts, K, s));
    VITA_TRY(time_path(w, f, K, s));
    const bool fold = f.stat != nullptr;
    if (fold) VITA_TRY(fold_tables(w, f, K, s));
    if (fold && ode_persistent_eligible(w, B, x0, ldx0, x_out, ldo)) {
        // all K steps in one persistent launch (vita_ode_persistent.hip): same tiles, same epilogues, same bits
        VITA_TRY(launch_ode_persistent(w, x0, ldx0, x_out, ldo, traj, B, K, dt, f.h, f.hn, f.u, f.stat, f.mod, f.c1, f.c2, f.c1f, f.c2f,
                                       f.sync, s));
        if (traj)
            VITA_CHECK_HIP(hipMemcpy2DAsync(x_out, (size_t)ldo * 4, traj + (size_t)K * B * D, (size_t)D * 4, (size_t)D * 4, (size_t)B,
                                            hipMemcpyDeviceToDevice, s));
        return VITA_OK;
    }
    if (f.sync)     // per-launch path on a workspace that could hold a persistent solve's state: its status word reads "completed"
        VITA_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(f.sync) + ode_persistent_status_offset(), 0, 128, s));
    // few-row batches: fc2's reduction (K = FF) is split over workgroups; the kernels finish it themselves (vita_splitk_fused.h)
    static const bool no_fused = getenv("VITA_NO_FUSED_SPLITK") != nullptr && atoi(getenv("VITA_NO_FUSED_SPLITK")) != 0;
    if (B <= 512 && !no_fused && !tl_ws_symbol) VITA_REQUIRE(sk.enable_fused(s) == 0, "vita_ode_sample_f32: could not zero the split-K counters");
    const float* src = x0;
    int64_t lds = ldx0;
    for (int k = 0; k < K; ++k) {
        float* dst = traj ? traj + (size_t)(k + 1) * B * D : x_out;
        const int64_t ldd = traj ? D : ldo;
        const float* mod = f.mod + (size_t)k * L * 3 * H;
        if (fold) VITA_TRY(flow_eval_folded(w, f, src, lds, mod, k, dst, ldd, src, lds, dt, B, s));
        else VITA_TRY(flow_eval(w, f, src, lds, mod, 0, dst, ldd, src, lds, dt, B, s));
        src = dst;
        lds = ldd;
    }
    if (traj)
        VITA_CHECK_HIP(hipMemcpy2DAsync(x_out, (size_t)ldo * 4, src, (size_t)D * 4, (size_t)D * 4, (size_t)B, hipMemcpyDeviceToDevice, s));
    return VITA_OK;
}

// ConsistencyFlowMatcher.sample (consistency_flow_matcher.py:72-106) over the same velocity net: step times
// t_i = i/K*(1-eps)+eps; with sigma_t = (1-t)*sigma_var > 0 the update is the SDE-flavoured one of :93-96 with the caller's
// noise, else the Euler update of :98 (fused into out_proj's epilogue like ode_sample).
static int consistency_sample(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int B, int K,
                              float eps, float sigma_var, float noise_scale, const float* noise, void* ws, hipStream_t s) {
    const int D = w->D, H = w->H, L = w->L;
    FlowWs f = carve_flow(ws, w, B, K, true, false);
    SplitkScope sk(f.sk, kSplitkBytes);
    VITA_TRY(launch_step_times_affine(f.ts, K, 1.0 - (double)eps, (double)eps, s));
    VITA_TRY(time_path(w, f, K, s));
    const double dt = 1.0 / (double)K;
    const float* src = x0;
    int64_t lds = ldx0;
    for (int k = 0; k < K; ++k) {
        const float* mod = f.mod + (size_t)k * L * 3 * H;
        const double t = (double)k / (double)K * (1.0 - (double)eps) + (double)eps;
        const double sigma_t = (1.0 - t) * (double)sigma_var;
        if (sigma_t > 0.0) {
            VITA_TRY(flow_eval(w, f, src, lds, mod, 0, f.x, D, nullptr, 0, 0.f, B, s));
            const double coef = (sigma_t * sigma_t) / (2.0 * ((double)noise_scale * (double)noise_scale) * ((1.0 - t) * (1.0 - t)));
            VITA_TRY(launch_sde_update(src, lds, f.x, noise + (size_t)k * B * D, x_out, ldo, B, D, (float)(0.5 * t * (1.0 - t)),
                                       (float)(0.5 * (2.0 - t)), (float)coef, (float)dt, (float)(sigma_t * sqrt(dt)), s));
        } else {
            VITA_TRY(flow_eval(w, f, src, lds, mod, 0, x_out, ldo, src, lds, (float)dt, B, s));
        }
        src = x_out;
        lds = ldo;
    }
    return VITA_OK;
}

struct DecWs { float *sk, *a, *b; size_t bytes; };
static DecWs carve_dec(void* ws, const vita_decoder_weights* w, int B) {
    Carver c(ws);
    DecWs d;
    d.sk = c.take(tl_skb / sizeof(float));
    d.a = c.take((size_t)B * w->HD);
    d.b = c.take((size_t)B * w->HD);
    d.bytes = c.off;
    return d;
}
int check_dec_weights(const vita_decoder_weights* w, const char* who) {
    VITA_REQUIRE(w != nullptr, "%s: null decoder weights", who);
    VITA_REQUIRE(w->D > 0 && w->HD > 0 && w->OUT > 0 && w->NL >= 0 && w->NL <= VITA_MAX_LAYERS, "%s: bad decoder dims", who);
    VITA_REQUIRE(w->in_w && w->in_b && w->out_w && w->out_b, "%s: null decoder weight pointer", who);
    for (int i = 0; i < w->NL; ++i)
        VITA_REQUIRE(w->fc1_w[i] && w->fc1_b[i] && w->fc2_w[i] && w->fc2_b[i], "%s: null decoder weight pointer in layer %d", who, i);
    return VITA_OK;
}
static int decoder_forward(const vita_decoder_weights* w, const float* z, int64_t ldz, float* out, int64_t ldo, int B, void* ws,
                           hipStream_t s) {
    const int D = w->D, HD = w->HD;
    DecWs d = carve_dec(ws, w, B);
    SplitkScope sk(d.sk, tl_skb);
    VITA_TRY(gemm_f32(lin(z, ldz, w->in_w, D, w->in_b, d.a, HD, B, HD, D, VITA_ACT_NONE), s));           // action_ae.py:274
    for (int i = 0; i < w->NL; ++i) {                                                                     // :275-276
        VITA_TRY(gemm_f32(lin(d.a, HD, w->fc1_w[i], HD, w->fc1_b[i], d.b, HD, B, HD, HD, VITA_ACT_GELU_ERF), s));
        VITA_TRY(gemm_f32(lin(d.b, HD, w->fc2_w[i], HD, w->fc2_b[i], d.a, HD, B, HD, HD, VITA_ACT_NONE), s));
    }
    VITA_TRY(gemm_f32(lin(d.a, HD, w->out_w, HD, w->out_b, out, ldo, B, w->OUT, HD, VITA_ACT_NONE), s)); // :277
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

// ---- allocation-free entry point (vita_generate_actions_noalloc_f32): its workspace is this array -------------------
// Every GPU the library runs on holds its own copy of `scratch` at its own address until the process ends.  In host code the
// name is only a stand-in for the symbol: the device address comes from hipGetSymbolAddress, once per device, and that
// pointer is what the carves, kernels and copies use.
static __device__ __align__(256) unsigned char scratch[VITA_NOALLOC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
static std::atomic<void*> g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

static int noalloc_scratch(void** out, const char* who) {
    int dev = -1;
    VITA_CHECK_HIP(hipGetDevice(&dev));
    VITA_REQUIRE(dev >= 0 && dev < kMaxDevices, "%s: device ordinal %d outside [0,%d)", who, dev, kMaxDevices);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (p == nullptr) {
            VITA_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));   // the current device's copy
            VITA_REQUIRE(p != nullptr && (reinterpret_cast<uintptr_t>(p) & 255u) == 0, "%s: bad scratch symbol address %p", who, p);
            g_scratch_ptr[dev].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return VITA_OK;
}

extern "C" {

int vita_abi_version(void) { return VITA_ABI_VERSION; }
const char* vita_last_error(void) { return g_err; }

size_t vita_splitk_workspace_bytes(void) { return size_t(32) << 20; }
const char* vita_debug_last_gemm_config(void) { return last_dma_config(); }

int vita_linear_f32(const vita_linear_args* args, vita_stream_t stream) {
    VITA_REQUIRE(args != nullptr, "vita_linear_f32: null args");
    return gemm_f32(*args, (hipStream_t)stream);
}

size_t vita_flownet_forward_workspace_bytes(const vita_flownet_weights* w, int32_t B) {
    if (!w || B <= 0) return 0;
    return carve_flow(nullptr, w, B, B, false, false).bytes;
}

int vita_flownet_forward_f32(const vita_flownet_weights* w, const float* x, int64_t ldx, const float* t, float* v, int64_t ldv,
                             int32_t B, void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    VITA_TRY(check_flow_weights(w, "vita_flownet_forward_f32"));
    VITA_REQUIRE(B >= 0, "vita_flownet_forward_f32: B < 0");
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(x && t && v && ldx >= w->D && ldv >= w->D, "vita_flownet_forward_f32: bad x/t/v");
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_flownet_forward_workspace_bytes(w, B), "vita_flownet_forward_f32"));
    hipStream_t s = (hipStream_t)stream;
    FlowWs f = carve_flow(workspace, w, B, B, false, false);
    SplitkScope sk(f.sk, kSplitkBytes);
    VITA_CHECK_HIP(hipMemcpyAsync(f.ts, t, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    VITA_TRY(time_path(w, f, B, s));
    return flow_eval(w, f, x, ldx, f.mod, (int64_t)w->L * 3 * w->H, v, ldv, nullptr, 0, 0.f, B, s);
}

size_t vita_ode_sample_workspace_bytes(const vita_flownet_weights* w, int32_t B, int32_t num_steps) {
    if (!w || B <= 0 || num_steps <= 0) return 0;
    return carve_flow(nullptr, w, B, num_steps, false, true).bytes;
}

int vita_ode_sample_f32(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int32_t B,
                        int32_t num_steps, float* traj, void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    VITA_TRY(check_flow_weights(w, "vita_ode_sample_f32"));
    VITA_REQUIRE(num_steps >= 1, "vita_ode_sample_f32: num_steps must be >= 1 (got %d)", num_steps);
    VITA_REQUIRE(B >= 0, "vita_ode_sample_f32: B < 0");
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(x0 && x_out && ldx0 >= w->D && ldo >= w->D, "vita_ode_sample_f32: bad x0/x_out");
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_ode_sample_workspace_bytes(w, B, num_steps), "vita_ode_sample_f32"));
    return ode_sample(w, x0, ldx0, x_out, ldo, B, num_steps, traj, workspace, (hipStream_t)stream);
}

int vita_ode_sample_set_persistent(int32_t on) { return set_ode_persistent(on); }
int vita_ode_sample_takes_persistent(const vita_flownet_weights* w, int32_t B) {
    if (!w || B <= 0) return 0;
    if (ode_resident_eligible(w, B, reinterpret_cast<const float*>(uintptr_t(256)), w->D, reinterpret_cast<const float*>(uintptr_t(256)), w->D))
        return 2;     // the weights-resident launch of serving batches
    if (!use_ln_fold(w, B)) return 0;
    return ode_persistent_eligible(w, B, reinterpret_cast<const float*>(uintptr_t(256)), w->D, reinterpret_cast<const float*>(uintptr_t(256)), w->D) ? 1 : 0;
}
int vita_debug_ode_stamps(void* device_buf) { set_ode_persistent_stamps(device_buf); return VITA_OK; }

int vita_ode_sample_status(const vita_flownet_weights* w, int32_t B, int32_t num_steps, const void* workspace, int32_t* status) {
    VITA_REQUIRE(w && workspace && status && B > 0 && num_steps > 0, "vita_ode_sample_status: bad arguments");
    *status = 0;
    FlowWs f = carve_flow(const_cast<void*>(workspace), w, B, num_steps, false, true);
    if (f.sync == nullptr) return VITA_OK;      // shapes that never take the persistent launch
    unsigned v = 0;
    VITA_CHECK_HIP(hipMemcpy(&v, reinterpret_cast<const char*>(f.sync) + ode_persistent_status_offset(), sizeof(v), hipMemcpyDeviceToHost));
    *status = (int32_t)v;
    return VITA_OK;
}

int vita_generate_actions_status(const vita_flownet_weights* fw, int32_t B, int32_t num_steps, const void* workspace, int32_t* status) {
    VITA_REQUIRE(fw && workspace && status && B > 0 && num_steps > 0, "vita_generate_actions_status: bad arguments");
    // the solve's carve starts behind the latent rows (generate_actions_run)
    const char* rest = static_cast<const char*>(workspace) + (((size_t)B * fw->D * 4 + 255) & ~size_t(255));
    return vita_ode_sample_status(fw, B, num_steps, rest, status);
}

int vita_debug_set_spin_limit(uint32_t spins) { set_ode_spin_limit(spins); return VITA_OK; }

size_t vita_consistency_sample_workspace_bytes(const vita_flownet_weights* w, int32_t B, int32_t num_steps) {
    if (!w || B <= 0 || num_steps <= 0) return 0;
    return carve_flow(nullptr, w, B, num_steps, true, false).bytes;
}

int vita_consistency_sample_f32(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int32_t B,
                                int32_t num_steps, float eps, float sigma_var, float noise_scale, const float* noise, void* workspace,
                                size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_consistency_sample_f32";
    VITA_TRY(check_flow_weights(w, who));
    VITA_REQUIRE(num_steps >= 1, "%s: num_steps must be >= 1 (got %d)", who, num_steps);
    VITA_REQUIRE(eps >= 0.f && eps < 1.f && sigma_var >= 0.f, "%s: eps must be in [0,1) and sigma_var >= 0", who);
    VITA_REQUIRE(sigma_var == 0.f || (noise != nullptr && noise_scale > 0.f), "%s: sigma_var > 0 needs the noise tensor and noise_scale > 0", who);
    VITA_REQUIRE(B >= 0, "%s: B < 0", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(x0 && x_out && ldx0 >= w->D && ldo >= w->D, "%s: bad x0/x_out", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_consistency_sample_workspace_bytes(w, B, num_steps), who));
    return consistency_sample(w, x0, ldx0, x_out, ldo, B, num_steps, eps, sigma_var, noise_scale, noise, workspace, (hipStream_t)stream);
}

size_t vita_decoder_forward_workspace_bytes(const vita_decoder_weights* w, int32_t B) {
    if (!w || B <= 0) return 0;
    return carve_dec(nullptr, w, B).bytes;
}

int vita_decoder_forward_f32(const vita_decoder_weights* w, const float* z, int64_t ldz, float* actions, int64_t lda, int32_t B,
                             void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    VITA_TRY(check_dec_weights(w, "vita_decoder_forward_f32"));
    VITA_REQUIRE(B >= 0, "vita_decoder_forward_f32: B < 0");
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(z && actions && ldz >= w->D && lda >= w->OUT, "vita_decoder_forward_f32: bad z/actions");
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_decoder_forward_workspace_bytes(w, B), "vita_decoder_forward_f32"));
    return decoder_forward(w, z, ldz, actions, lda, B, workspace, (hipStream_t)stream);
}

size_t vita_generate_actions_workspace_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw, int32_t B,
                                             int32_t num_steps) {
    if (!fw || !dw || B <= 0 || num_steps <= 0) return 0;
    const size_t lat = ((size_t)B * fw->D * 4 + 255) & ~size_t(255);
    const size_t a = carve_flow(nullptr, fw, B, num_steps, false, true).bytes;
    const size_t b = carve_dec(nullptr, dw, B).bytes;
    return lat + (a > b ? a : b);
}

static int generate_actions_checked(const char* who, const float* obs_w, const float* obs_b, int32_t O, const vita_flownet_weights* fw,
                                    const vita_decoder_weights* dw, const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                                    int32_t B, int32_t num_steps) {
    VITA_TRY(check_flow_weights(fw, who));
    VITA_TRY(check_dec_weights(dw, who));
    VITA_REQUIRE(fw->D == dw->D, "%s: flow-net D=%d != decoder D=%d", who, fw->D, dw->D);
    VITA_REQUIRE(num_steps >= 1, "%s: num_steps must be >= 1 (got %d)", who, num_steps);
    VITA_REQUIRE(B >= 0 && O > 0, "%s: bad B/O", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(obs_w && obs_b && obs_features && actions && ldobs >= O && lda >= dw->OUT, "%s: bad pointers/strides", who);
    return VITA_OK;
}

// obs_encoder -> Euler sampler -> decoder on `workspace`, whose carves start with a split-K region of tl_skb bytes
static int generate_actions_run(const float* obs_w, const float* obs_b, int32_t O, const vita_flownet_weights* fw,
                                const vita_decoder_weights* dw, const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                                float* latents_out, int32_t B, int32_t num_steps, void* workspace, hipStream_t s) {
    const int D = fw->D;
    float* lat = static_cast<float*>(workspace);
    char* rest = static_cast<char*>(workspace) + (((size_t)B * D * 4 + 255) & ~size_t(255));
    // serving batches: obs_encoder, the solve and the decoder are one launch (vita_ode_resident.hip); the condition is the one
    // ode_sample evaluates
    const ResidentObs oe{obs_w, obs_b, obs_features, ldobs, O};
    const bool fuse_obs = O <= 1024 && ode_resident_eligible(fw, B, lat, D, lat, D);
    if (!fuse_obs) {
        SplitkScope sk(rest, tl_skb);   // the first tl_skb bytes of either carve below
        VITA_TRY(gemm_f32(lin(obs_features, ldobs, obs_w, O, obs_b, lat, D, B, D, O, VITA_ACT_NONE), s));  // vita_policy.py:194
    }
    bool dec_done = false;
    VITA_TRY(ode_sample(fw, lat, D, lat, D, B, num_steps, nullptr, rest, s, dw, actions, lda, &dec_done, fuse_obs ? &oe : nullptr));   // :196-203
    if (latents_out)
        VITA_CHECK_HIP(hipMemcpyAsync(latents_out, lat, (size_t)B * D * 4, hipMemcpyDeviceToDevice, s));
    if (dec_done) return VITA_OK;                                                                       // serving batches: :206 ran in the same launch
    return decoder_forward(dw, lat, D, actions, lda, B, rest, s);                                       // :206
}

int vita_generate_actions_f32(const float* obs_w, const float* obs_b, int32_t O, const vita_flownet_weights* fw,
                              const vita_decoder_weights* dw, const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                              float* latents_out, int32_t B, int32_t num_steps, void* workspace, size_t workspace_bytes,
                              vita_stream_t stream) {
    VITA_TRY(generate_actions_checked("vita_generate_actions_f32", obs_w, obs_b, O, fw, dw, obs_features, ldobs, actions, lda, B, num_steps));
    if (B == 0) return VITA_OK;
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_generate_actions_workspace_bytes(fw, dw, B, num_steps), "vita_generate_actions_f32"));
    return generate_actions_run(obs_w, obs_b, O, fw, dw, obs_features, ldobs, actions, lda, latents_out, B, num_steps, workspace,
                                (hipStream_t)stream);
}

// ---- no counterpart in the reference: the same call for callers that cannot allocate (HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc).  No hipMalloc / hipFree / allocating routine below this line.
size_t vita_noalloc_scratch_bytes(void) { return (size_t)VITA_NOALLOC_SCRATCH_BYTES; }

int vita_noalloc_init(void) {
    void* p = nullptr;
    return noalloc_scratch(&p, "vita_noalloc_init");
}

size_t vita_generate_actions_noalloc_need_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw, int32_t B, int32_t num_steps) {
    SkbScope skb((size_t)VITA_NOALLOC_SPLITK_BYTES);
    return vita_generate_actions_workspace_bytes(fw, dw, B, num_steps);
}

int vita_generate_actions_noalloc_f32(const float* obs_w, const float* obs_b, int32_t O, const vita_flownet_weights* fw,
                                      const vita_decoder_weights* dw, const float* obs_features, int64_t ldobs, float* actions,
                                      int64_t lda, float* latents_out, int32_t B, int32_t num_steps, vita_stream_t stream) {
    const char* who = "vita_generate_actions_noalloc_f32";
    VITA_TRY(generate_actions_checked(who, obs_w, obs_b, O, fw, dw, obs_features, ldobs, actions, lda, B, num_steps));
    if (B == 0) return VITA_OK;
    SkbScope skb((size_t)VITA_NOALLOC_SPLITK_BYTES);
    const size_t need = vita_generate_actions_workspace_bytes(fw, dw, B, num_steps);
    if (need > (size_t)VITA_NOALLOC_SCRATCH_BYTES) {
        set_error("%s: B=%d, num_steps=%d with these dims need %zu bytes of scratch, the built-in array holds %zu: use "
                  "vita_generate_actions_f32 with a caller workspace", who, B, num_steps, need, (size_t)VITA_NOALLOC_SCRATCH_BYTES);
        return VITA_ERR_WORKSPACE;
    }
    void* ws = nullptr;
    VITA_TRY(noalloc_scratch(&ws, who));
    struct SymbolScope { SymbolScope() { tl_ws_symbol = true; } ~SymbolScope() { tl_ws_symbol = false; } } sym;
    return generate_actions_run(obs_w, obs_b, O, fw, dw, obs_features, ldobs, actions, lda, latents_out, B, num_steps, ws,
                                (hipStream_t)stream);
}


// status word of the last vita_generate_actions_noalloc_f32 call with these shapes on the current device (synchronises that copy)
int vita_generate_actions_noalloc_status(const vita_flownet_weights* fw, int32_t B, int32_t num_steps, int32_t* status) {
    const char* who = "vita_generate_actions_noalloc_status";
    VITA_REQUIRE(fw && status && B > 0 && num_steps > 0, "%s: bad arguments", who);
    void* ws = nullptr;
    VITA_TRY(noalloc_scratch(&ws, who));
    SkbScope skb((size_t)VITA_NOALLOC_SPLITK_BYTES);      // the carve of the call it reports on
    return vita_generate_actions_status(fw, B, num_steps, ws, status);
}

}  // extern "C"
