// Host-side helpers shared by the C-ABI orchestrators (vita_api.hip, vita_loss_step.hip):
// workspace carving, vita_linear_args builders, early-return macro and the internal launchers.
#pragma once
#include "vita_common.h"

namespace vita {

int gemm_f32(const vita_linear_args& a, hipStream_t stream);
const char* last_dma_config();   // "<BM,BN,BK,NS,KG>" of the calling thread's last LDS-DMA launch ("" = none yet)
// LayerNorm folded into the neighbouring GEMMs' epilogues (see EpiParams in vita_gemm_epilogue.h).  Producer fields:
// aux_out/aux_mul/stat_out; consumer fields: ln_stats/ln_c1.  Requires the LDS-DMA kernel, M % 256 == 0, N % 128 == 0 and
// 16-byte aligned operands (every tile interior); returns an error otherwise.
struct EpiFold {
    float* aux_out = nullptr; int64_t ld_aux = 0; const float* aux_mul = nullptr; int aux_add_one = 0;
    uint16_t* aux_bf = nullptr; int64_t ld_aux_bf = 0;     // bf16 copy of the product (bf16-operand sampler)
    float* stat_out = nullptr;
    const float* ln_stats = nullptr; const float* ln_c1 = nullptr; int ln_nseg = 0; float ln_fold_eps = 0.f;
};
int gemm_f32_fold(const vita_linear_args& a, const EpiFold& fold, hipStream_t stream);
// c-tables of the folded LayerNorm for T steps: c1[(k*L+l)*FF + j] = sum_i (1 + scale_kl[i]) W1_l[j][i], c2 likewise from the shifts
// (+ b1); c1f/c2f (D) for norm -> out_proj.  One launch (vita_elementwise.hip).
int launch_ln_fold_tables_all(const vita_flownet_weights* w, const float* mod, int T, float* c1, float* c2, float* c1f, float* c2f, hipStream_t s,
                              const vita_bf16_weights* bw = nullptr);   // bw: take the GEMM weights from the bf16 copies (row stride r8(H))
// the persistent Euler sampler (vita_ode_persistent.hip): one launch for all K steps at chip-filling batches
bool ode_persistent_eligible(const vita_flownet_weights* w, int B, const float* x0, int64_t ldx0, const float* x_out, int64_t ldo);
size_t ode_persistent_sync_bytes(int K, int L);
int set_ode_persistent(int on);   // returns the previous setting
void set_ode_persistent_stamps(void* device_buf);
void* ode_persistent_stamps();
size_t ode_persistent_status_offset();
// In-launch samplers' failure path (round 3): behind the launch, words != 0 in the sticky status word turn `buf` (rows x cols 32-bit
// words, row stride ld words) into NaN (0x7FC07FC0: a NaN as one fp32 and as two bf16), so a bounded hand-off wait that ran out
// travels with the data instead of leaving undefined values; no host synchronisation.
int launch_poison_on_timeout(const void* sync, void* buf, int64_t ld_words, int64_t rows, int64_t cols_words, hipStream_t s);
unsigned ode_spin_limit();          // bound of every hand-off spin (vita_debug_set_spin_limit; default 1 << 22)
void set_ode_spin_limit(unsigned n);
int device_cus();                  // compute units of the current device (cached per device; <= 0: unknown)
bool ode_persistent_enabled();     // the process-wide switch (VITA_NO_PERSISTENT / vita_ode_sample_set_persistent)
// the weights-resident Euler sampler for serving batches (vita_ode_resident.hip): one launch, weights in LDS for all K steps
bool ode_resident_dims(const vita_flownet_weights* w, int B);
bool ode_resident_eligible(const vita_flownet_weights* w, int B, const float* x0, int64_t ldx0, const float* x_out, int64_t ldo);
struct ResidentObs { const float *w, *b, *x; int64_t ldx; int O; };     // obs_encoder ahead of the first step (O <= 1024)
int launch_ode_resident(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, float* traj, int B, int K,
                        float dt, float* h, float* u, float* xi, float* mod, float* thid, float* tsilu, void* sync, bool acquire, hipStream_t s,
                        const vita_decoder_weights* dw = nullptr, float* da = nullptr, float* db = nullptr, float* actions = nullptr,
                        int64_t lda = 0, const ResidentObs* obs = nullptr);
bool ode_resident_time_path(const vita_flownet_weights* w, int K);
bool ode_resident_decoder_ok(const vita_decoder_weights* dw, const float* actions, int64_t lda);
int launch_ode_persistent(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, float* traj, int B, int K,
                          float dt, float* h, float* hn, float* u, float* stat, const float* mod, const float* c1, const float* c2,
                          const float* c1f, const float* c2f, void* sync, hipStream_t s);
int launch_ode_persistent_bf16(const vita_flownet_weights* w, const vita_bf16_weights* bw, const float* x0, int64_t ldx0, const uint16_t* x0b,
                               float* x_out, int64_t ldo, uint16_t* xb_mid, uint16_t* xb_final, int B, int K, float dt, float* h, uint16_t* hnb,
                               uint16_t* ub, float* stat, const float* mod, const float* c1, const float* c2, const float* c1f, const float* c2f,
                               void* sync, hipStream_t s);
int launch_sinusoid(const float* t, float* out, int64_t ldo, int B, int dim, hipStream_t s);
int launch_step_times(float* ts, int K, hipStream_t s);
int launch_ln_mod(const float* x, int64_t ldx, float* y, int64_t ldy, int M, int H, float eps, int mode, const float* mul,
                  int64_t ld_mul, const float* add, int64_t ld_add, hipStream_t s, uint16_t* ybf = nullptr, int64_t ld_ybf = 0);
struct SplitK;
int gemm_bf16_dma(const vita_linear_args& a, uint16_t* ybf, int64_t ld_ybf, hipStream_t stream, const SplitK* sk = nullptr,
                  const EpiFold* fold = nullptr);
// C (M, N) = A^T B for row-major bf16 A (R, M), B (R, N): the weight-gradient GEMM without transposed operand copies
// (vita_gemm_bf16_tn.hip); gemm_bf16_wgrad_tn plans the split over R and runs the second pass (vita_gemm_f32.hip)
// fp32 form (vita_gemm_f32_tn.hip): a = wgrad(...) with a_col = b_col = 1; eligible: 16-byte aligned rows, M, N, ldx, ldw % 4 == 0
// wgrad() calls that are ready together: one unsplit launch per group of up to 8 when that fills the chip, else one by one
int gemm_f32_wgrad_batch(const vita_linear_args* a, int n, hipStream_t stream);
int gemm_f32_wgrad_pair(const vita_linear_args& a0, const vita_linear_args& a1, hipStream_t stream);
bool gemm_f32_tn_eligible(const vita_linear_args& a);
int gemm_f32_tn(const vita_linear_args& a, hipStream_t stream, const SplitK* sk);
bool gemm_bf16_tn_eligible(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int R);
int gemm_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const vita_linear_args& a, hipStream_t stream, const SplitK* sk);
int gemm_bf16_wgrad_tn(const uint16_t* dYb, int64_t ld_dy, const uint16_t* Xb, int64_t ldx, const vita_linear_args& a, hipStream_t stream);
int launch_pack_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int cols, hipStream_t s);
int launch_pack_bf16_t(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int K, int M, hipStream_t s);
// vita_layernorm_modulate_bwd_f32 with an optional fused gate backward on the new dx (the next kernel of the flow net's backward
// chain): dz[row] = dx_new[row] * g[row], dg[row] = dx_new[row] * z[row]; z / dz are contiguous (M, H) and may be the same buffer
struct LnGateTail { const float* z; const float* g; int64_t ld_g; float* dz; float* dg; int64_t ld_dg; };
int launch_ln_mod_bwd(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* mul, int64_t ld_mul, int M, int H, float eps,
                      int mode, float* dx, int64_t ld_dx, int dx_accumulate, float* dmul_rows, float* dadd_rows, int64_t ld_drows, hipStream_t s,
                      const LnGateTail* tail);
// vita_gate_bwd_f32 with an optional bf16 copy of dz (row stride ld_bf16) for the bf16-operand step
int launch_gate_bwd(const float* dy, const float* z, const float* gate, int64_t ld_gate, float gate_scalar, float* dz, float* dgate_rows,
                    int64_t ld_dgate, int M, int N, hipStream_t s, uint16_t* dz_bf16, int64_t ld_bf16);
int launch_set_scalars(float* out, const float* host_vals, int n, hipStream_t st);
int launch_weighted_sum(const float* vals, const float* host_w, int n, float* out, hipStream_t st);

// Bump allocator over the caller's workspace; with base == nullptr it only measures.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    float* take(size_t nfloats) {
        float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += (nfloats * sizeof(float) + 255) & ~size_t(255);
        return r;
    }
};

// true when gemm_f32() will run this call's LayerNorm prologue (a.pro) inside the few-row kernel — instead of a LayerNorm launch
// in front of it (vita_gemm_f32_small.hip); callers that get false keep launch_ln_mod + a plain GEMM
bool gemm_f32_small_ln_eligible(const vita_linear_args& a);

// Column sums (bias gradients) of several few-row matrices in one launch; matrices of more than 2048 rows go straight to
// vita_colsum_f32.  flush() before the results are needed and before a source buffer is overwritten.
constexpr int kColsumJobs = 16;
struct ColsumJob { const float* x; int64_t ldx; int M, N; float* out; };
struct ColsumJobs { ColsumJob j[kColsumJobs]; };
struct ColsumBatch {
    ColsumJobs jobs;
    int n = 0, max_n = 0;
    float* cs_part;
    hipStream_t st;
    ColsumBatch(float* part, hipStream_t s) : cs_part(part), st(s) {}
    int add(const float* x, int64_t ldx, int M, int N, float* out);
    int flush();
};

// Split-K scratch of the running orchestrator (thread-local: entry points are re-entrant per stream/thread).  Every
// vita_linear_args built by lin()/dgrad()/wgrad() inside the scope carries it.
// counters: NULL, or kSplitkCounters zeroed ints at the tail of the scratch for the kernels that finish a split-K reduction
// themselves (vita_splitk_fused.h); zeroed by enable_fused on the scope's stream, kept at zero by the kernels.
struct SplitkScratch { void* ptr = nullptr; size_t bytes = 0; int* counters = nullptr; };
constexpr int kSplitkCounters = 2048;
inline SplitkScratch& splitk_scratch() {
    static thread_local SplitkScratch s;
    return s;
}
struct SplitkScope {
    SplitkScratch saved;
    SplitkScope(void* p, size_t n) : saved(splitk_scratch()) { splitk_scratch().ptr = p; splitk_scratch().bytes = n; splitk_scratch().counters = nullptr; }
    ~SplitkScope() { splitk_scratch() = saved; }
    // one memset per entry-point call instead of one reduce launch per split GEMM; every GEMM of the scope must run on `st`
    int enable_fused(hipStream_t st) {
        SplitkScratch& s = splitk_scratch();
        const size_t tail = (size_t)kSplitkCounters * sizeof(int);
        if (s.ptr == nullptr || s.bytes < (size_t(1) << 20)) return 0;
        s.bytes -= tail;
        s.counters = reinterpret_cast<int*>(static_cast<char*>(s.ptr) + s.bytes);
        if (hipMemsetAsync(s.counters, 0, tail, st) != hipSuccess) { s.counters = nullptr; return -1; }
        return 0;
    }
};
// bf16-operand training GEMMs (vita_loss_config.gemm_bf16): inside the scope gemm_f32() stages both operands of an
// eligible call as k-contiguous bf16 rows in `ptr` (stream-ordered reuse) and runs the bf16 LDS-DMA kernel; mode 1 only
// for calls of at least kBf16MinFlops, mode 2 for every call without a LayerNorm prologue.
// wc: bf16 copies of the step's weights, packed ONCE per step (row form (out, r8(in)) for the forward GEMMs, transposed form
// (in, r8(out)) for the input-gradient GEMMs) instead of once per GEMM call: a weight is an operand of 1 + K forward and 1 + K
// input-gradient GEMMs of a step (round 3).
struct Bf16WEntry { const float* w; const uint16_t* rows; const uint16_t* t; int out_f, in_f; };
// shadows: bf16 row-major copies of fp32 activation matrices written by the kernel that PRODUCED them (a GEMM epilogue's ybf, the
// LayerNorm kernel's bf16 output) instead of by a pack launch in front of the GEMM that consumes them.  emit_*: a request to the
// next bf16-routed GEMM whose output is emit_y to write such a copy.  A shadow is dropped when it is consumed, when its fp32
// source is overwritten by a routed GEMM, and at the end of the scope.
struct Bf16Shadow { const float* src; const uint16_t* bf; int64_t ld; int rows, cols; bool only = false; };   // only: the fp32 matrix was NOT written (emit_only)
struct Bf16Stage {
    void* ptr = nullptr; size_t bytes = 0; int mode = 0; Bf16WEntry wc[80]; int nwc = 0;
    Bf16Shadow sh[4]; int nsh = 0;
    const float* emit_y = nullptr; uint16_t* emit_dst = nullptr; int64_t emit_ld = 0;
    bool emit_only = false;     // with emit_y: when the bf16 copy IS written, skip the fp32 output (nobody reads it)
    uint16_t* arena = nullptr; size_t arena_halves = 0;      // shadow slots of the running step (vita_loss_step.hip)
    void shadow_add(const float* src, const uint16_t* bf, int64_t ld, int rows, int cols, bool only = false) {
        for (int i = 0; i < nsh; ++i)
            if (sh[i].src == src) { sh[i] = Bf16Shadow{src, bf, ld, rows, cols, only}; return; }
        if (nsh < 4) sh[nsh++] = Bf16Shadow{src, bf, ld, rows, cols, only};
    }
    // true when `src` exists ONLY as its bf16 copy: a GEMM that would read the fp32 matrix must not run
    bool shadow_is_only(const float* src) const {
        for (int i = 0; i < nsh; ++i)
            if (sh[i].src == src && sh[i].only) return true;
        return false;
    }
    bool has_shadow(const float* src, const uint16_t* bf) const {
        for (int i = 0; i < nsh; ++i)
            if (sh[i].src == src && sh[i].bf == bf) return true;
        return false;
    }
    void shadow_drop(const float* src) {
        for (int i = 0; i < nsh; ++i)
            if (sh[i].src == src) { sh[i] = sh[--nsh]; return; }
    }
};
inline Bf16Stage& bf16_stage() {
    static thread_local Bf16Stage s;
    return s;
}
struct Bf16Scope {
    Bf16Stage saved;
    Bf16Scope(void* p, size_t n, int mode) : saved(bf16_stage()) { bf16_stage().ptr = p; bf16_stage().bytes = n; bf16_stage().mode = mode; bf16_stage().nwc = 0; bf16_stage().nsh = 0; bf16_stage().emit_y = nullptr; bf16_stage().emit_only = false; bf16_stage().arena = nullptr; bf16_stage().arena_halves = 0; }
    ~Bf16Scope() { bf16_stage() = saved; }
};
// Transposed copies of weights for the input-gradient GEMMs of a large-batch training step: dX = dY . W with W (out, in)
// contracts over W's ROWS (a column operand, register-staged kernel); with W^T (in, out) materialised once per step it is
// an ordinary row-operand GEMM and takes the LDS-DMA kernels of the forward pass.  Inside the scope gemm_f32() reroutes
// every b_col call whose W is registered.  Thread-local like the other scopes.
struct WtEntry { const float* w; const float* wt; int out_f, in_f; };
struct WtCache { WtEntry e[80]; int n = 0; };
inline WtCache& wt_cache() {
    static thread_local WtCache c;
    return c;
}
struct WtScope {
    int saved;
    WtScope() : saved(wt_cache().n) {}
    ~WtScope() { wt_cache().n = saved; }
    void add(const float* w, const float* wt, int out_f, int in_f) {
        WtCache& c = wt_cache();
        if (c.n < 80) c.e[c.n++] = WtEntry{w, wt, out_f, in_f};
    }
};
constexpr int kWtMinRows = 1024;     // batch from which the step keeps transposed weights
int launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t s);   // (rows, cols) -> (cols, rows), contiguous

constexpr double kBf16MinFlops = 2147483648.0;   // 2^31: below this the two staging launches cost more than the MFMA rate gains
constexpr size_t kSplitkBytes = size_t(32) << 20;   // == vita_splitk_workspace_bytes()

inline vita_linear_args lin(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* b, float* Y, int64_t ldy, int M,
                            int N, int K, int act) {
    vita_linear_args a{};
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = b; a.Y = Y; a.ldy = ldy;
    a.M = M; a.N = N; a.K = K; a.act = act;
    a.splitk_ws = splitk_scratch().ptr; a.splitk_ws_bytes = splitk_scratch().bytes;
    return a;
}
// dX[M, in] = dY[M, out] . W[out, in]  (autograd of F.linear w.r.t. its input)
inline vita_linear_args dgrad(const float* dY, int64_t ld_dy, const float* W, int64_t ldw, float* dX, int64_t ld_dx, int M, int in_f,
                              int out_f) {
    vita_linear_args a = lin(dY, ld_dy, W, ldw, nullptr, dX, ld_dx, M, in_f, out_f, VITA_ACT_NONE);
    a.b_col = 1;
    return a;
}
// dW[out, in] = dY[R, out]^T . X[R, in]  (w.r.t. its weight), R = rows stacked over every use of W
inline vita_linear_args wgrad(const float* dY, int64_t ld_dy, const float* X, int64_t ldx, float* dW, int64_t ld_dw, int R, int in_f,
                              int out_f) {
    vita_linear_args a = lin(dY, ld_dy, X, ldx, nullptr, dW, ld_dw, out_f, in_f, R, VITA_ACT_NONE);
    a.a_col = 1;
    a.b_col = 1;
    return a;
}

#define VITA_TRY(expr)                    \
    do {                                  \
        int rc__ = (expr);                \
        if (rc__ != VITA_OK) return rc__; \
    } while (0)

inline int check_ws(const void* ws, size_t have, size_t need, const char* who) {
    if (need == 0) return VITA_OK;
    if (ws == nullptr || have < need) {
        set_error("%s: workspace too small or null (have %zu bytes, need %zu)", who, have, need);
        return VITA_ERR_WORKSPACE;
    }
    if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0) {
        set_error("%s: workspace must be 256-byte aligned", who);
        return VITA_ERR_WORKSPACE;
    }
    return VITA_OK;
}

// InfoNCE pieces (vita_contrastive.hip): forward keeps what the backward needs in `ws` (nce_workspace_bytes)
size_t nce_workspace_bytes(int B, int D);
int nce_forward(const float* a, int64_t lda, const float* b, int64_t ldb, int B, int D, float temperature, float* loss, void* ws, hipStream_t s);
int nce_backward(int B, int D, float temperature, const float* gout, float* da, int64_t ld_da, int acc_a, float* db, int64_t ld_db, int acc_b,
                 void* ws, hipStream_t s);
int launch_axpy(const float* x, float* y, int64_t n, hipStream_t s);   // y += x
// variational action encoder (action_vae.py:224-229, models/dists.py): params rows are [mean | logvar]
int launch_gauss_sample(const float* params, int64_t ldp, const float* eps, float* z, int64_t ldz, int B, int D, hipStream_t s);
int launch_gauss_sample_bwd(const float* params, int64_t ldp, const float* eps, const float* dz, int64_t ld_dz, float kl_scale, float* dparams,
                            int64_t ld_dp, int B, int D, hipStream_t s);
// alternate samplers (vita_meanflow.hip)
int launch_step_times_affine(float* ts, int K, double scale, double offset, hipStream_t s);   // ts[k] = (float)(k/K*scale + offset)
int launch_sde_update(const float* z, int64_t ldz, const float* v, const float* noise, float* out, int64_t ldo, int B, int D, float c_v,
                      float c_z, float coef, float dt, float s_noise, hipStream_t s);
int launch_add_rows(const float* x, int64_t ldx, const float* r, int64_t ldr, float* out, int64_t ldo, int M, int N, hipStream_t s);

int check_flow_weights(const vita_flownet_weights* w, const char* who);
int check_dec_weights(const vita_decoder_weights* w, const char* who);

}  // namespace vita
