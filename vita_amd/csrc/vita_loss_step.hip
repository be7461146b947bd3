// vita_compute_loss_f32 — the training forward + backward of the VITA hot path, orchestrated on
// the host side of the C ABI: VitaPolicy.compute_loss (vita_policy.py:112-189) and the autograd
// pass loss.backward() runs over it (policy_trainer.py:82).
//
// Design (MI355X-first, not a tape): the graph is static, so the backward is written out by hand
// and every buffer it needs is carved from the caller's workspace.
//   * The flow net is evaluated E = 1 + K times per step (one flow-matching evaluation with a
//     per-row t, K Euler steps with a per-step t).  Activations of all E evaluations are saved
//     STACKED, [layer][eval][B rows], so that after the E input-gradient chains have run, every
//     weight gets ONE wgrad GEMM over E*B rows (dW = sum_e dY_e^T X_e = [dY_0;..]^T [X_0;..]) and one
//     column-sum for its bias: 7x fewer wgrad launches and 7x less dW read-modify-write traffic than
//     accumulating per evaluation.
//   * dU / dY2 overwrite the saved pre-activations in place (the dgrad GEMM's epilogue multiplies by
//     act'(z) reading the element it then overwrites), so the backward adds no (B, FF) buffers.
//   * Both decoder calls (dec(pred), dec(z1)) run as one 2B-row batch.
//   * Modulation rows: the time path is evaluated once for B + K timesteps; per-row (gamma, scale,
//     shift) gradients of the flow-matching evaluation land directly in the time path's gradient
//     table, those of Euler step k are column-summed into row B + k.
//   * Gradients are written, not accumulated: each weight's gradient is produced by exactly one GEMM.
#include <stdlib.h>

#include "vita_host.h"

namespace vita {

namespace {

// ------------------------------------------------------------------------------------------------
// Generic "input_proj -> NL x timm Mlp(exact GELU) -> [output_proj]" stack: SimpleActionDecoder
// (action_ae.py:226-279) and SimpleActionEncoder (:111-170, per-timestep rows).
// ------------------------------------------------------------------------------------------------
struct Stack {
    int IN = 0, TR = 0, FF = 0, NL = 0, OUT = 0;  // OUT == 0: no output projection
    const float *in_w = nullptr, *in_b = nullptr;
    const float *fc1_w[VITA_MAX_LAYERS] = {}, *fc1_b[VITA_MAX_LAYERS] = {};
    const float *fc2_w[VITA_MAX_LAYERS] = {}, *fc2_b[VITA_MAX_LAYERS] = {};
    const float *out_w = nullptr, *out_b = nullptr;
};
struct StackWs {
    float *A = nullptr;    // (NL+1, M, TR) trunk activations
    float *P = nullptr;    // (NL, M, FF) fc1 pre-activations, overwritten by their gradients
    float *G = nullptr;    // (NL, M, FF) gelu(P)
    float *out = nullptr;  // (M, OUT)
    float *dOut = nullptr; // (M, OUT)
    float *dA[VITA_MAX_LAYERS + 1] = {};  // (M, TR) gradient of trunk activation i, i = 0..NL: each kept until its wgrad has run
    // bf16-operand step at large batch (carve_stack's `tn`): row-major bf16 copies of A, G, dP and dA, written by the GEMM that
    // produces each matrix; the next GEMM reads them as its A operand and the weight-gradient GEMMs read them as they lie
    // (vita_gemm_bf16_tn.hip).  Row strides r8(TR) / r8(FF).  NULL: operands are packed per call.
    uint16_t *Ab = nullptr, *Gb = nullptr, *dPb = nullptr;
    uint16_t *dAb[VITA_MAX_LAYERS + 1] = {};
    bool tn() const { return Ab != nullptr; }
};
inline int r8i(int v) { return (v + 7) & ~7; }
// after a GEMM that was asked to emit the bf16 copy `dst` of its output y: make sure the copy exists (a split or fp32-routed
// GEMM writes none) and is registered as y's shadow for the next GEMM
int emit_or_pack(const float* y, int64_t ldy, uint16_t* dst, int64_t ld, int M, int N, hipStream_t st) {
    Bf16Stage& bs = bf16_stage();
    if (bs.has_shadow(y, dst)) return VITA_OK;
    VITA_TRY(launch_pack_bf16(y, ldy, dst, ld, M, N, st));
    bs.shadow_add(y, dst, ld, M, N);
    return VITA_OK;
}
inline void emit_request(const float* y, uint16_t* dst, int64_t ld, bool only = false) {
    Bf16Stage& bs = bf16_stage();
    bs.emit_y = y; bs.emit_dst = dst; bs.emit_ld = ld; bs.emit_only = only;
}
Stack stack_of(const vita_decoder_weights& w) {
    Stack s;
    s.IN = w.D; s.TR = w.HD; s.FF = w.HD; s.NL = w.NL; s.OUT = w.OUT;
    s.in_w = w.in_w; s.in_b = w.in_b; s.out_w = w.out_w; s.out_b = w.out_b;
    for (int i = 0; i < w.NL; ++i) { s.fc1_w[i] = w.fc1_w[i]; s.fc1_b[i] = w.fc1_b[i]; s.fc2_w[i] = w.fc2_w[i]; s.fc2_b[i] = w.fc2_b[i]; }
    return s;
}
Stack stack_of_simple_encoder(const vita_encoder_weights& w) {
    Stack s;
    s.IN = w.A; s.TR = w.C; s.FF = 4 * w.C; s.NL = w.NL; s.OUT = 0;
    s.in_w = w.proj_w; s.in_b = w.proj_b;
    for (int i = 0; i < w.NL; ++i) { s.fc1_w[i] = w.w1[i]; s.fc1_b[i] = w.b1[i]; s.fc2_w[i] = w.w2[i]; s.fc2_b[i] = w.b2[i]; }
    return s;
}
// whether the stack's GEMMs over M rows all take the bf16 route and its dims suit the row-major weight-gradient kernel
bool stack_tn_ok(const Stack& s, size_t M, int mode) {
    return mode != 0 && M >= (size_t)kWtMinRows && s.TR % 8 == 0 && s.FF % 8 == 0 && s.NL > 0 &&
           (mode == 2 || 2.0 * (double)M * s.TR * (s.TR < s.FF ? s.TR : s.FF) >= kBf16MinFlops);
}
StackWs carve_stack(Carver& c, const Stack& s, size_t M, bool train, bool tn = false) {
    StackWs w;
    w.A = c.take((size_t)(s.NL + 1) * M * s.TR);
    w.P = c.take((size_t)s.NL * M * s.FF);
    w.G = c.take((size_t)s.NL * M * s.FF);
    if (s.OUT) w.out = c.take(M * s.OUT);
    if (train) {
        if (s.OUT) w.dOut = c.take(M * s.OUT);
        for (int i = 0; i <= s.NL; ++i) w.dA[i] = c.take(M * s.TR);
    }
    if (train && tn) {
        auto halves = [&](size_t n) { return reinterpret_cast<uint16_t*>(c.take((n + 1) / 2)); };
        const size_t tr8 = r8i(s.TR), ff8 = r8i(s.FF);
        w.Ab = halves((size_t)(s.NL + 1) * M * tr8);
        w.Gb = halves((size_t)s.NL * M * ff8);
        w.dPb = halves((size_t)s.NL * M * ff8);
        for (int i = 0; i <= s.NL; ++i) w.dAb[i] = halves(M * tr8);
    }
    return w;
}
// last_trunk: when non-null the final trunk activation A[NL] is written there instead (contiguous (M, TR))
int stack_fwd(const Stack& s, const StackWs& w, const float* x, int64_t ldx, int M, float* last_trunk, hipStream_t st) {
    const size_t tr = (size_t)M * s.TR, ff = (size_t)M * s.FF;
    auto trunk = [&](int i) { return (i == s.NL && last_trunk) ? last_trunk : w.A + (size_t)i * tr; };
    const bool tn = w.tn() && bf16_stage().mode != 0;
    const int tr8 = r8i(s.TR), ff8 = r8i(s.FF);
    auto Ab = [&](int i) { return w.Ab + (size_t)i * M * tr8; };
    auto Gb = [&](int i) { return w.Gb + (size_t)i * M * ff8; };
    if (tn) { bf16_stage().nsh = 0; emit_request(trunk(0), Ab(0), tr8); }
    VITA_TRY(gemm_f32(lin(x, ldx, s.in_w, s.IN, s.in_b, trunk(0), s.TR, M, s.TR, s.IN, VITA_ACT_NONE), st));
    if (tn) VITA_TRY(emit_or_pack(trunk(0), s.TR, Ab(0), tr8, M, s.TR, st));
    for (int i = 0; i < s.NL; ++i) {
        vita_linear_args a = lin(trunk(i), s.TR, s.fc1_w[i], s.TR, s.fc1_b[i], w.G + i * ff, s.FF, M, s.FF, s.TR, VITA_ACT_GELU_ERF);
        a.preact = w.P + i * ff; a.ld_preact = s.FF;
        if (tn) emit_request(w.G + i * ff, Gb(i), ff8, true);      // (nothing reads G in fp32 then: fc2 and the weight gradient take the copy)
        VITA_TRY(gemm_f32(a, st));
        if (tn) { VITA_TRY(emit_or_pack(w.G + i * ff, s.FF, Gb(i), ff8, M, s.FF, st)); emit_request(trunk(i + 1), Ab(i + 1), tr8); }
        VITA_TRY(gemm_f32(lin(w.G + i * ff, s.FF, s.fc2_w[i], s.FF, s.fc2_b[i], trunk(i + 1), s.TR, M, s.TR, s.FF, VITA_ACT_NONE), st));
        if (tn) VITA_TRY(emit_or_pack(trunk(i + 1), s.TR, Ab(i + 1), tr8, M, s.TR, st));
    }
    if (s.OUT) VITA_TRY(gemm_f32(lin(trunk(s.NL), s.TR, s.out_w, s.TR, s.out_b, w.out, s.OUT, M, s.OUT, s.TR, VITA_ACT_NONE), st));
    if (tn) bf16_stage().nsh = 0;
    return VITA_OK;
}
// Backward of the stack in two parts so that a caller may run them on different streams: the input-gradient chain
// (stack_chain_bwd: what the rest of the backward pass waits for) and the weight / bias gradients (stack_wgrads: needs only
// what the chain left behind).  d_top: gradient of the stack's result — (M, OUT) w.dOut when OUT > 0, else the gradient of
// A[NL] (M, TR).  dx: NULL or (M, IN) receiving the input gradient.  The chain leaves d A[i] in w.dA[i] (i < NL; i = NL
// too when OUT > 0) and dP_i in place of the saved pre-activation P_i.
int stack_chain_bwd(const Stack& s, const StackWs& w, int M, const float* d_top, int64_t ld_top, float* dx, int64_t ld_dx, hipStream_t st) {
    const size_t ff = (size_t)M * s.FF;
    const float* dA = d_top;
    int64_t ld_dA = ld_top;
    const bool tn = w.tn() && bf16_stage().mode != 0;
    const int tr8 = r8i(s.TR), ff8 = r8i(s.FF);
    if (tn) bf16_stage().nsh = 0;
    if (s.OUT) {
        if (tn) emit_request(w.dA[s.NL], w.dAb[s.NL], tr8);
        VITA_TRY(gemm_f32(dgrad(d_top, ld_top, s.out_w, s.TR, w.dA[s.NL], s.TR, M, s.TR, s.OUT), st));
        dA = w.dA[s.NL]; ld_dA = s.TR;
    }
    // (without an output projection the top gradient comes from the caller in fp32: one row pack, shared by the GEMM below and
    // the weight gradient)
    if (tn) VITA_TRY(emit_or_pack(dA, ld_dA, w.dAb[s.NL], tr8, M, s.TR, st));
    for (int i = s.NL - 1; i >= 0; --i) {
        float* P = w.P + i * ff;
        vita_linear_args a = dgrad(dA, ld_dA, s.fc2_w[i], s.FF, P, s.FF, M, s.FF, s.TR);   // dP = (dA . W2) * gelu'(P), in place
        a.act = VITA_ACT_GELU_ERF; a.act_grad_z = P; a.ld_act_grad_z = s.FF;
        uint16_t* dPb = tn ? w.dPb + (size_t)i * M * ff8 : nullptr;
        if (tn) emit_request(P, dPb, ff8);
        VITA_TRY(gemm_f32(a, st));
        if (tn) { VITA_TRY(emit_or_pack(P, s.FF, dPb, ff8, M, s.FF, st)); emit_request(w.dA[i], w.dAb[i], tr8); }
        VITA_TRY(gemm_f32(dgrad(P, s.FF, s.fc1_w[i], s.TR, w.dA[i], s.TR, M, s.TR, s.FF), st));
        if (tn) VITA_TRY(emit_or_pack(w.dA[i], s.TR, w.dAb[i], tr8, M, s.TR, st));
        dA = w.dA[i]; ld_dA = s.TR;
    }
    if (dx) VITA_TRY(gemm_f32(dgrad(dA, ld_dA, s.in_w, s.IN, dx, ld_dx, M, s.IN, s.TR), st));
    if (tn) bf16_stage().nsh = 0;
    return VITA_OK;
}
inline float* mut(const float* p) { return const_cast<float*>(p); }
// g: gradient buffers in the same descriptor; x: the stack's input rows; last_trunk as in stack_fwd.
int stack_wgrads(const Stack& s, const Stack& g, const StackWs& w, const float* x, int64_t ldx, int M, const float* last_trunk,
                 const float* d_top, int64_t ld_top, float* cs_part, hipStream_t st) {
    const size_t tr = (size_t)M * s.TR, ff = (size_t)M * s.FF;
    auto trunk = [&](int i) -> const float* { return (i == s.NL && last_trunk) ? last_trunk : w.A + (size_t)i * tr; };
    auto dA = [&](int i) -> const float* { return (i == s.NL && !s.OUT) ? d_top : w.dA[i]; };
    auto ld_dA = [&](int i) -> int64_t { return (i == s.NL && !s.OUT) ? ld_top : s.TR; };
    ColsumBatch cb(cs_part, st);      // the bias gradients: one launch for the stack at few rows
    const bool tn = w.tn() && bf16_stage().mode != 0;
    if (!tn) {
        // every weight gradient of the stack is ready here: at few rows they share launches (gemm_f32_wgrad_batch)
        vita_linear_args jobs[2 * VITA_MAX_LAYERS + 2];
        int nj = 0;
        if (s.OUT) {
            jobs[nj++] = wgrad(d_top, ld_top, trunk(s.NL), s.TR, mut(g.out_w), s.TR, M, s.TR, s.OUT);
            VITA_TRY(cb.add(d_top, ld_top, M, s.OUT, mut(g.out_b)));
        }
        for (int i = s.NL - 1; i >= 0; --i) {
            const float* P = w.P + i * ff;
            jobs[nj++] = wgrad(dA(i + 1), ld_dA(i + 1), w.G + i * ff, s.FF, mut(g.fc2_w[i]), s.FF, M, s.FF, s.TR);
            jobs[nj++] = wgrad(P, s.FF, trunk(i), s.TR, mut(g.fc1_w[i]), s.TR, M, s.TR, s.FF);
            VITA_TRY(cb.add(dA(i + 1), ld_dA(i + 1), M, s.TR, mut(g.fc2_b[i])));
            VITA_TRY(cb.add(P, s.FF, M, s.FF, mut(g.fc1_b[i])));
        }
        jobs[nj++] = wgrad(dA(0), ld_dA(0), x, ldx, mut(g.in_w), s.IN, M, s.IN, s.TR);
        VITA_TRY(cb.add(dA(0), ld_dA(0), M, s.TR, mut(g.in_b)));
        VITA_TRY(gemm_f32_wgrad_batch(jobs, nj, st));
        return cb.flush();
    }
    // bf16-operand step at large batch: the layers' operands exist as row-major bf16 copies, as the forward / backward GEMMs left them
    if (s.OUT) {
        VITA_TRY(gemm_f32(wgrad(d_top, ld_top, trunk(s.NL), s.TR, mut(g.out_w), s.TR, M, s.TR, s.OUT), st));
        VITA_TRY(cb.add(d_top, ld_top, M, s.OUT, mut(g.out_b)));
    }
    const int tr8 = r8i(s.TR), ff8 = r8i(s.FF);
    for (int i = s.NL - 1; i >= 0; --i) {
        const float* P = w.P + i * ff;
        VITA_TRY(gemm_bf16_wgrad_tn(w.dAb[i + 1], tr8, w.Gb + (size_t)i * M * ff8, ff8,
                                    wgrad(dA(i + 1), ld_dA(i + 1), w.G + i * ff, s.FF, mut(g.fc2_w[i]), s.FF, M, s.FF, s.TR), st));
        VITA_TRY(gemm_bf16_wgrad_tn(w.dPb + (size_t)i * M * ff8, ff8, w.Ab + (size_t)i * M * tr8, tr8,
                                    wgrad(P, s.FF, trunk(i), s.TR, mut(g.fc1_w[i]), s.TR, M, s.TR, s.FF), st));
        VITA_TRY(cb.add(dA(i + 1), ld_dA(i + 1), M, s.TR, mut(g.fc2_b[i])));
        VITA_TRY(cb.add(P, s.FF, M, s.FF, mut(g.fc1_b[i])));
    }
    VITA_TRY(gemm_f32(wgrad(dA(0), ld_dA(0), x, ldx, mut(g.in_w), s.IN, M, s.IN, s.TR), st));
    VITA_TRY(cb.add(dA(0), ld_dA(0), M, s.TR, mut(g.in_b)));
    return cb.flush();
}
int stack_bwd(const Stack& s, const Stack& g, const StackWs& w, const float* x, int64_t ldx, int M, const float* last_trunk,
              const float* d_top, int64_t ld_top, float* dx, int64_t ld_dx, float* cs_part, hipStream_t st) {
    VITA_TRY(stack_chain_bwd(s, w, M, d_top, ld_top, dx, ld_dx, st));
    return stack_wgrads(s, g, w, x, ldx, M, last_trunk, d_top, ld_top, cs_part, st);
}

// ------------------------------------------------------------------------------------------------
// CNNActionEncoder (action_ae.py:282-327) on channels-last rows: Conv1d(k5,s2,p2) = im2col + GEMM.
// ------------------------------------------------------------------------------------------------
struct CnnWs {
    float *col[VITA_MAX_LAYERS] = {};  // (B*Lout_i, Cin_i*5)
    float *out[VITA_MAX_LAYERS] = {};  // (B*Lout_i, C) post-ReLU, channels-last
    float *flat = nullptr;             // (B, C*Lf) in (C, L) order when Lf > 1
    float *d0 = nullptr, *d1 = nullptr; // gradient ping-pong, each max(B*Lout_0*C, B*C*Lf, B*L_i*C_i)
};
inline int conv_len(int T, int i) { return T >> i; }  // requires T % 2^NL == 0
inline int proj_out(const vita_encoder_weights& w) { return w.variational ? 2 * w.D : w.D; }   // action_vae.py:213
CnnWs carve_cnn(Carver& c, const vita_encoder_weights& w, size_t B, bool train) {
    CnnWs s;
    size_t dmax = 0;
    for (int i = 0; i < w.NL; ++i) {
        const size_t cin = i == 0 ? w.A : w.C, lout = conv_len(w.T, i + 1);
        s.col[i] = c.take(B * lout * cin * 5);
        s.out[i] = c.take(B * lout * w.C);
        dmax = dmax > B * lout * w.C ? dmax : B * lout * w.C;
        dmax = dmax > B * conv_len(w.T, i) * cin ? dmax : B * conv_len(w.T, i) * cin;
    }
    const size_t lf = conv_len(w.T, w.NL);
    if (lf > 1) s.flat = c.take(B * w.C * lf);
    dmax = dmax > B * w.C * lf ? dmax : B * w.C * lf;
    if (train) { s.d0 = c.take(dmax); s.d1 = c.take(dmax); }
    return s;
}
int cnn_fwd(const vita_encoder_weights& w, const CnnWs& s, const float* actions, float* z, int64_t ldz, int B, hipStream_t st) {
    const float* x = actions;
    for (int i = 0; i < w.NL; ++i) {
        const int cin = i == 0 ? w.A : w.C, lin_ = conv_len(w.T, i), lout = conv_len(w.T, i + 1);
        VITA_TRY(vita_im2col_k5s2_f32(x, s.col[i], B, lin_, cin, st));
        VITA_TRY(gemm_f32(lin(s.col[i], cin * 5, w.w1[i], cin * 5, w.b1[i], s.out[i], w.C, B * lout, w.C, cin * 5, VITA_ACT_RELU), st));
        x = s.out[i];
    }
    const int lf = conv_len(w.T, w.NL);
    const float* flat = w.NL ? s.out[w.NL - 1] : actions;
    const int cf = w.NL ? w.C : w.A;
    if (lf > 1) {
        VITA_TRY(vita_nlc_ncl_f32(flat, s.flat, B, lf, cf, 1, st));   // x.view(B,-1) of the (B,C,L) tensor, :323
        flat = s.flat;
    }
    return gemm_f32(lin(flat, cf * lf, w.proj_w, cf * lf, w.proj_b, z, ldz, B, proj_out(w), cf * lf, VITA_ACT_NONE), st);
}
int cnn_bwd(const vita_encoder_weights& w, const vita_encoder_weights& g, const CnnWs& s, const float* actions, const float* dz,
            int64_t ld_dz, int B, float* cs_part, hipStream_t st) {
    const int lf = conv_len(w.T, w.NL);
    const int cf = w.NL ? w.C : w.A;
    const float* flat = lf > 1 ? s.flat : (w.NL ? s.out[w.NL - 1] : actions);
    const int PD = proj_out(w);
    VITA_TRY(gemm_f32(wgrad(dz, ld_dz, flat, cf * lf, mut(g.proj_w), cf * lf, B, cf * lf, PD), st));
    VITA_TRY(vita_colsum_f32(dz, ld_dz, B, PD, cs_part, mut(g.proj_b), 0, st));
    if (w.NL == 0) return VITA_OK;
    float* d_out = s.d0;
    float* d_alt = s.d1;
    VITA_TRY(gemm_f32(dgrad(dz, ld_dz, w.proj_w, cf * lf, d_out, cf * lf, B, cf * lf, PD), st));
    if (lf > 1) {
        VITA_TRY(vita_nlc_ncl_f32(d_out, d_alt, B, lf, cf, 0, st));
        float* t = d_out; d_out = d_alt; d_alt = t;
    }
    for (int i = w.NL - 1; i >= 0; --i) {
        const int cin = i == 0 ? w.A : w.C, lin_ = conv_len(w.T, i), lout = conv_len(w.T, i + 1);
        const int rows = B * lout;
        // ReLU backward from the saved output (out > 0 <=> pre-activation > 0)
        VITA_TRY(vita_act_bwd_f32(s.out[i], d_out, d_alt, (int64_t)rows * w.C, VITA_ACT_RELU, st));
        VITA_TRY(gemm_f32(wgrad(d_alt, w.C, s.col[i], cin * 5, mut(g.w1[i]), cin * 5, rows, cin * 5, w.C), st));
        VITA_TRY(vita_colsum_f32(d_alt, w.C, rows, w.C, cs_part, mut(g.b1[i]), 0, st));
        if (i > 0) {
            VITA_TRY(gemm_f32(dgrad(d_alt, w.C, w.w1[i], cin * 5, s.col[i], cin * 5, rows, cin * 5, w.C), st));  // dcol reuses col
            VITA_TRY(vita_col2im_k5s2_f32(s.col[i], d_out, B, lin_, cin, st));
        }
    }
    return VITA_OK;
}

// ------------------------------------------------------------------------------------------------
// SimpleFlowNet with saved activations
// ------------------------------------------------------------------------------------------------
// Up to this many rows per batch Euler step 0 is evaluated together with the flow-matching rows (launch-bound regime); above,
// the doubled per-row time path would cost more than the saved launches (B = 4096: 23.0 -> 23.8 ms per step).
constexpr int kRideAlongMaxRows = 1024;

struct FlowTrainWs {
    // time path over T = B + K rows (flow-matching rows first, then one row per Euler step)
    float *ts, *sin, *hidpre, *hid, *tepre, *silu, *mod, *dmod, *dsilu, *dte, *dhid;
    // evaluations e = 0 (flow matching), 1..K (Euler step e-1)
    float *X, *Hs, *HN, *U, *G, *Y2, *DV, *DH0, *DHNF, *RN;
    float *dhn, *DM;
    // bf16-operand step at large batch: stacked ROW-MAJOR bf16 copies of HN, G, dU and dY2 (same (l, e) order as the fp32 buffers,
    // row strides r8(H) / r8(FF)), written by the kernels that produce the fp32 matrices and read by the next GEMM and, as they
    // lie, by the weight-gradient GEMM on row-major operands (vita_gemm_bf16_tn.hip).  NULL: the transposed-pack route.
    uint16_t *HNb, *Gb, *dUb, *Y2b;
};
FlowTrainWs carve_flow_train(Carver& c, const vita_flownet_weights& w, size_t B, size_t K, bool train, bool tn = false) {
    FlowTrainWs f{};
    // time-path rows: the B per-row timesteps of the flow-matching evaluation, B more for Euler step 0 (t = 0; it is evaluated
    // together with the flow-matching rows, see vita_compute_loss_f32), one per later step
    const size_t T = K > 0 ? (B <= (size_t)kRideAlongMaxRows ? 2 * B + K - 1 : B + K) : B, E = 1 + K, H = w.H, FF = w.FF, D = w.D, L = w.L, TE = w.TE;
    f.ts = c.take(T);
    f.sin = c.take(T * TE);
    f.hidpre = c.take(T * 4 * TE);
    f.hid = c.take(T * 4 * TE);
    f.tepre = c.take(T * H);
    f.silu = c.take(T * H);
    f.mod = c.take(T * L * 3 * H);
    f.X = c.take(E * B * D);
    f.Hs = c.take((L + 1) * E * B * H);
    f.HN = c.take((L + 1) * E * B * H);
    f.U = c.take(L * E * B * FF);
    f.G = c.take(L * E * B * FF);
    f.Y2 = c.take(L * E * B * H);
    if (train) {
        f.dmod = c.take(T * L * 3 * H);
        f.dsilu = c.take(T * H);
        f.dte = c.take(T * H);
        f.dhid = c.take(T * 4 * TE);
        f.DV = c.take(E * B * D);
        f.DH0 = c.take(E * B * H);
        f.DHNF = c.take(E * B * H);
        f.RN = c.take(E * B * H);
        f.dhn = c.take(2 * B * H);
        f.DM = c.take(B * L * 3 * H);
    }
    if (train && tn) {
        const size_t H8 = (H + 7) & ~size_t(7), FF8 = (FF + 7) & ~size_t(7);
        uint16_t* hn = reinterpret_cast<uint16_t*>(c.take((L * E * B * H8 + 1) / 2));
        uint16_t* g = reinterpret_cast<uint16_t*>(c.take((L * E * B * FF8 + 1) / 2));
        uint16_t* du = reinterpret_cast<uint16_t*>(c.take((L * E * B * FF8 + 1) / 2));
        // (when only measuring every pointer is null: the flag travels in the pointers, so mark them)
        uint16_t* y2 = reinterpret_cast<uint16_t*>(c.take((L * E * B * H8 + 1) / 2));
        f.HNb = hn; f.Gb = g; f.dUb = du; f.Y2b = y2;
    }
    return f;
}

struct FlowCtx {
    const vita_flownet_weights* w;
    FlowTrainWs f;
    int B, K, E;
    hipStream_t st;
    size_t bh() const { return (size_t)B * w->H; }
    size_t bf() const { return (size_t)B * w->FF; }
    float* Hs(int l, int e) const { return f.Hs + ((size_t)l * E + e) * bh(); }
    float* HN(int l, int e) const { return f.HN + ((size_t)l * E + e) * bh(); }
    float* U(int l, int e) const { return f.U + ((size_t)l * E + e) * bf(); }
    float* G(int l, int e) const { return f.G + ((size_t)l * E + e) * bf(); }
    float* Y2(int l, int e) const { return f.Y2 + ((size_t)l * E + e) * bh(); }
    float* X(int e) const { return f.X + (size_t)e * B * w->D; }
    float* DV(int e) const { return f.DV + (size_t)e * B * w->D; }
    float* DH0(int e) const { return f.DH0 + (size_t)e * bh(); }
    float* DHNF(int e) const { return f.DHNF + (size_t)e * bh(); }
    float* RN(int e) const { return f.RN + (size_t)e * bh(); }
    int H8() const { return (w->H + 7) & ~7; }
    int FF8() const { return (w->FF + 7) & ~7; }
    bool tn() const { return f.Gb != nullptr; }
    uint16_t* HNb(int l, int e) const { return f.HNb + ((size_t)l * E + e) * B * H8(); }
    uint16_t* Gb(int l, int e) const { return f.Gb + ((size_t)l * E + e) * B * FF8(); }
    uint16_t* Y2b(int l, int e) const { return f.Y2b + ((size_t)l * E + e) * B * H8(); }
    uint16_t* dUb(int l, int e) const { return f.dUb + ((size_t)l * E + e) * B * FF8(); }
};

// simple_flow_net.py:66-71,27-30 for the T rows of f.ts, keeping the pre-activations
int time_path_fwd(const FlowCtx& c, int T) {
    const vita_flownet_weights* w = c.w;
    const int TE = w->TE, H = w->H, L = w->L;
    VITA_TRY(launch_sinusoid(c.f.ts, c.f.sin, TE, T, TE, c.st));
    vita_linear_args a = lin(c.f.sin, TE, w->te1_w, TE, w->te1_b, c.f.hid, 4 * TE, T, 4 * TE, TE, VITA_ACT_MISH);
    a.preact = c.f.hidpre; a.ld_preact = 4 * TE;
    VITA_TRY(gemm_f32(a, c.st));
    vita_linear_args b = lin(c.f.hid, 4 * TE, w->te3_w, 4 * TE, w->te3_b, c.f.silu, H, T, H, 4 * TE, VITA_ACT_SILU);
    b.preact = c.f.tepre; b.ld_preact = H;
    VITA_TRY(gemm_f32(b, c.st));
    for (int l = 0; l < L; ++l)
        VITA_TRY(gemm_f32(lin(c.f.silu, H, w->mod_w[l], H, w->mod_b[l], c.f.mod + (size_t)l * 3 * H, (int64_t)L * 3 * H, T, 3 * H, H,
                              VITA_ACT_NONE), c.st));
    return VITA_OK;
}

int time_path_bwd(const FlowCtx& c, const vita_flownet_weights& g, int T, float* cs_part) {
    const vita_flownet_weights* w = c.w;
    const int TE = w->TE, H = w->H, L = w->L;
    const int64_t ldm = (int64_t)L * 3 * H;
    if (L == 0) {  // the time embedding feeds only the modulators: without layers its gradient is zero
        VITA_CHECK_HIP(hipMemsetAsync(mut(g.te1_w), 0, (size_t)4 * TE * TE * 4, c.st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(g.te1_b), 0, (size_t)4 * TE * 4, c.st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(g.te3_w), 0, (size_t)H * 4 * TE * 4, c.st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(g.te3_b), 0, (size_t)H * 4, c.st));
        return VITA_OK;
    }
    ColsumBatch cb(cs_part, c.st);    // the bias gradients: one launch at the end (dmod, dte and dhid stay as they are until then)
    {   // the modulators' weight gradients need only dmod and silu: ready now, one shared launch at few rows
        vita_linear_args jobs[VITA_MAX_LAYERS];
        for (int l = 0; l < L; ++l) jobs[l] = wgrad(c.f.dmod + (size_t)l * 3 * H, ldm, c.f.silu, H, mut(g.mod_w[l]), H, T, H, 3 * H);
        VITA_TRY(gemm_f32_wgrad_batch(jobs, L, c.st));
    }
    for (int l = 0; l < L; ++l) {
        const float* dm = c.f.dmod + (size_t)l * 3 * H;
        vita_linear_args a = dgrad(dm, ldm, w->mod_w[l], H, c.f.dsilu, H, T, H, 3 * H);
        if (l > 0) { a.resid = c.f.dsilu; a.ld_resid = H; }
        VITA_TRY(gemm_f32(a, c.st));
        VITA_TRY(cb.add(dm, ldm, T, 3 * H, mut(g.mod_b[l])));
    }
    VITA_TRY(vita_act_bwd_f32(c.f.tepre, c.f.dsilu, c.f.dte, (int64_t)T * H, VITA_ACT_SILU, c.st));
    vita_linear_args a = dgrad(c.f.dte, H, w->te3_w, 4 * TE, c.f.dhid, 4 * TE, T, 4 * TE, H);
    a.act = VITA_ACT_MISH; a.act_grad_z = c.f.hidpre; a.ld_act_grad_z = 4 * TE;
    VITA_TRY(gemm_f32(a, c.st));
    // (dte and dhid stay as they are: the two embedding weight gradients share a launch at few rows)
    VITA_TRY(gemm_f32_wgrad_pair(wgrad(c.f.dte, H, c.f.hid, 4 * TE, mut(g.te3_w), 4 * TE, T, 4 * TE, H),
                                 wgrad(c.f.dhid, 4 * TE, c.f.sin, TE, mut(g.te1_w), TE, T, TE, 4 * TE), c.st));
    VITA_TRY(cb.add(c.f.dte, H, T, H, mut(g.te3_b)));
    VITA_TRY(cb.add(c.f.dhid, 4 * TE, T, 4 * TE, mut(g.te1_b)));
    return cb.flush();
}

// SimpleFlowNet.forward (simple_flow_net.py:99-108) up to the final norm for the M rows that start at evaluation e (M = B, or
// n*B for n consecutive evaluations: their rows are adjacent in every stacked buffer), saving what the backward needs.
int flow_trunk_fwd(const FlowCtx& c, int e, int M, const float* mod, int64_t ld_mod) {
    const vita_flownet_weights* w = c.w;
    const int D = w->D, H = w->H, FF = w->FF, L = w->L, B = M;
    // bf16-operand step: the LayerNorm kernel and fc1's epilogue write the bf16 copies the next GEMM reads (Bf16Stage shadows)
    // instead of a pack launch in front of it; only where both GEMMs of a layer take the bf16 route
    Bf16Stage& bs = bf16_stage();
    bs.nsh = 0;
    const int H8 = (H + 7) & ~7, FF8 = (FF + 7) & ~7;
    const bool tn = c.tn() && bs.mode != 0;      // stacked copies kept for the weight-gradient GEMMs
    const bool sh = tn || (bs.mode != 0 && bs.arena != nullptr && (size_t)B * (H8 + FF8) <= bs.arena_halves &&
                           (bs.mode == 2 || 2.0 * B * (double)FF * H >= kBf16MinFlops));
    VITA_TRY(gemm_f32(lin(c.X(e), D, w->in_w, D, w->in_b, c.Hs(0, e), H, B, H, D, VITA_ACT_NONE), c.st));
    for (int l = 0; l < L; ++l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        const float* scale = gamma + H;
        const float* shift = gamma + 2 * H;
        if (sh) {
            uint16_t* slot_hn = tn ? c.HNb(l, e) : bs.arena;
            uint16_t* slot_g = tn ? c.Gb(l, e) : bs.arena + (size_t)B * H8;
            VITA_TRY(launch_ln_mod(c.Hs(l, e), H, c.HN(l, e), H, B, H, 1e-6f, VITA_PRO_LN_MOD, scale, ld_mod, shift, ld_mod, c.st, slot_hn, H8));
            bs.shadow_add(c.HN(l, e), slot_hn, H8, B, H);
            vita_linear_args a = lin(c.HN(l, e), H, w->fc1_w[l], H, w->fc1_b[l], c.G(l, e), FF, B, FF, H, VITA_ACT_GELU_TANH);
            a.preact = c.U(l, e); a.ld_preact = FF;
            bs.emit_y = c.G(l, e); bs.emit_dst = slot_g; bs.emit_ld = FF8;
            bs.emit_only = tn;      // with the stacked copies nothing reads G in fp32: fc2 and the weight gradient take the bf16 one
            VITA_TRY(gemm_f32(a, c.st));
            // the weight-gradient GEMM reads the stacked copy whatever route this call took (a split or fp32-routed GEMM emits none)
            if (tn && !bs.has_shadow(c.G(l, e), slot_g)) VITA_TRY(launch_pack_bf16(c.G(l, e), FF, slot_g, FF8, B, FF, c.st));
            vita_linear_args b = lin(c.G(l, e), FF, w->fc2_w[l], FF, w->fc2_b[l], c.Hs(l + 1, e), H, B, H, FF, VITA_ACT_NONE);
            b.preact = c.Y2(l, e); b.ld_preact = H;
            b.gate = gamma; b.ld_gate = ld_mod; b.resid = c.Hs(l, e); b.ld_resid = H;
            VITA_TRY(gemm_f32(b, c.st));
            bs.nsh = 0;
            continue;
        }
        // (the LayerNorm prologue of the few-row GEMM, vita_gemm_f32_small.hip, is not used here: with per-row modulation and the
        // normalised rows to be kept for the backward pass it measured no gain, B = 128 step 2.89 against 2.88 ms)
        VITA_TRY(launch_ln_mod(c.Hs(l, e), H, c.HN(l, e), H, B, H, 1e-6f, VITA_PRO_LN_MOD, scale, ld_mod, shift, ld_mod, c.st));
        vita_linear_args a = lin(c.HN(l, e), H, w->fc1_w[l], H, w->fc1_b[l], c.G(l, e), FF, B, FF, H, VITA_ACT_GELU_TANH);
        a.preact = c.U(l, e); a.ld_preact = FF;
        VITA_TRY(gemm_f32(a, c.st));
        vita_linear_args b = lin(c.G(l, e), FF, w->fc2_w[l], FF, w->fc2_b[l], c.Hs(l + 1, e), H, B, H, FF, VITA_ACT_NONE);
        b.preact = c.Y2(l, e); b.ld_preact = H;
        b.gate = gamma; b.ld_gate = ld_mod; b.resid = c.Hs(l, e); b.ld_resid = H;
        VITA_TRY(gemm_f32(b, c.st));
    }
    return launch_ln_mod(c.Hs(L, e), H, c.HN(L, e), H, B, H, 1e-5f, VITA_PRO_LN_AFFINE, w->norm_w, 0, w->norm_b, 0, c.st);
}
// out_proj of evaluation e (B rows): Y = resid + dt * v when resid != NULL (flow_matchers.py:71), else Y = v.
int flow_out_fwd(const FlowCtx& c, int e, float* Y, int64_t ldy, const float* resid, int64_t ld_resid, float dt) {
    const vita_flownet_weights* w = c.w;
    const int D = w->D, H = w->H, L = w->L, B = c.B;
    vita_linear_args o = lin(c.HN(L, e), H, w->out_w, H, w->out_b, Y, ldy, B, D, H, VITA_ACT_NONE);
    if (resid) { o.use_gate_scalar = 1; o.gate_scalar = dt; o.resid = resid; o.ld_resid = ld_resid; }
    return gemm_f32(o, c.st);
}
int flow_eval_fwd(const FlowCtx& c, int e, const float* mod, int64_t ld_mod, float* Y, int64_t ldy, const float* resid,
                  int64_t ld_resid, float dt) {
    VITA_TRY(flow_trunk_fwd(c, e, c.B, mod, ld_mod));
    return flow_out_fwd(c, e, Y, ldy, resid, ld_resid, dt);
}

// Input-gradient chain of evaluation e from DV(e) = d loss / d v.  Leaves dU in U(l,e), dY2 in Y2(l,e), dH0 in DH0(e),
// dHNF / RN for the final norm, the modulation-row gradients in dm_rows (B, L*3H), and
//   dx = resid + DH0 . W_in      (resid: the identity branch of the Euler update, or NULL)
// flow_chain_bwd: everything but the last GEMM, for the M rows that start at evaluation e (M = B or n*B, as flow_trunk_fwd);
// flow_in_bwd: dx of evaluation e (B rows).
int flow_chain_bwd(const FlowCtx& c, int e, int M, const float* mod, int64_t ld_mod, float* dm_rows) {
    const vita_flownet_weights* w = c.w;
    const int D = w->D, H = w->H, FF = w->FF, L = w->L, B = M;
    const int64_t ldm = (int64_t)L * 3 * H;
    Bf16Stage& bs = bf16_stage();      // as in flow_trunk_fwd: dU's bf16 copy comes out of the GEMM that makes dU
    bs.nsh = 0;
    const int FF8 = (FF + 7) & ~7;
    const bool tn = c.tn() && bs.mode != 0;
    const bool sh = tn || (bs.mode != 0 && bs.arena != nullptr && (size_t)B * FF8 <= bs.arena_halves &&
                           (bs.mode == 2 || 2.0 * B * (double)FF * H >= kBf16MinFlops));
    VITA_TRY(gemm_f32(dgrad(c.DV(e), D, w->out_w, H, c.DHNF(e), H, B, H, D), c.st));
    // x_{l+1} = x_l + y2 * gamma: the gate backward of layer l reads the dx the LayerNorm backward above it has just made, so (fp32
    // step) it runs as that kernel's tail -- one launch less per layer
    auto gate_tail = [&](int l) {
        return LnGateTail{c.Y2(l, e), mod + (size_t)l * 3 * H, ld_mod, c.Y2(l, e), dm_rows + (size_t)l * 3 * H, ldm};
    };
    const bool fuse_gate = !tn && L > 0;
    {
        const LnGateTail t = fuse_gate ? gate_tail(L - 1) : LnGateTail{};
        VITA_TRY(launch_ln_mod_bwd(c.Hs(L, e), H, c.DHNF(e), H, w->norm_w, 0, B, H, 1e-5f, VITA_PRO_LN_AFFINE, c.DH0(e), H, 0, c.RN(e), nullptr, H,
                                   c.st, fuse_gate ? &t : nullptr));
    }
    for (int l = L - 1; l >= 0; --l) {
        const float* gamma = mod + (size_t)l * 3 * H;
        const float* scale = gamma + H;
        float* dmr = dm_rows + (size_t)l * 3 * H;
        if (tn) {      // dY2's bf16 copy comes out of the kernel that makes dY2: the next GEMM's A operand, and the weight gradient's
            VITA_TRY(launch_gate_bwd(c.DH0(e), c.Y2(l, e), gamma, ld_mod, 0.f, c.Y2(l, e), dmr, ldm, B, H, c.st, c.Y2b(l, e), c.H8()));
            bs.shadow_add(c.Y2(l, e), c.Y2b(l, e), c.H8(), B, H);
        } else if (!fuse_gate) {
            VITA_TRY(vita_gate_bwd_f32(c.DH0(e), c.Y2(l, e), gamma, ld_mod, 0.f, c.Y2(l, e), dmr, ldm, B, H, c.st));
        }
        vita_linear_args a = dgrad(c.Y2(l, e), H, w->fc2_w[l], FF, c.U(l, e), FF, B, FF, H);   // dU = (dY2 . W2) * gelu_tanh'(U)
        a.act = VITA_ACT_GELU_TANH; a.act_grad_z = c.U(l, e); a.ld_act_grad_z = FF;
        uint16_t* slot_du = tn ? c.dUb(l, e) : bs.arena;
        if (sh) { bs.emit_y = c.U(l, e); bs.emit_dst = slot_du; bs.emit_ld = FF8; }
        VITA_TRY(gemm_f32(a, c.st));
        if (tn && !bs.has_shadow(c.U(l, e), slot_du)) VITA_TRY(launch_pack_bf16(c.U(l, e), FF, slot_du, FF8, B, FF, c.st));
        VITA_TRY(gemm_f32(dgrad(c.U(l, e), FF, w->fc1_w[l], H, c.f.dhn, H, B, H, FF), c.st));
        bs.nsh = 0;
        {
            const bool tl = fuse_gate && l > 0;
            const LnGateTail t = tl ? gate_tail(l - 1) : LnGateTail{};
            VITA_TRY(launch_ln_mod_bwd(c.Hs(l, e), H, c.f.dhn, H, scale, ld_mod, B, H, 1e-6f, VITA_PRO_LN_MOD, c.DH0(e), H, 1, dmr + H, dmr + 2 * H, ldm,
                                       c.st, tl ? &t : nullptr));
        }
    }
    return VITA_OK;
}
int flow_in_bwd(const FlowCtx& c, int e, float* dx, int64_t ld_dx, const float* resid, int64_t ld_resid) {
    const vita_flownet_weights* w = c.w;
    vita_linear_args a = dgrad(c.DH0(e), w->H, w->in_w, w->D, dx, ld_dx, c.B, w->D, w->H);
    if (resid) { a.resid = resid; a.ld_resid = ld_resid; }
    return gemm_f32(a, c.st);
}
int flow_eval_bwd(const FlowCtx& c, int e, const float* mod, int64_t ld_mod, float* dm_rows, float* dx, int64_t ld_dx,
                  const float* resid, int64_t ld_resid) {
    VITA_TRY(flow_chain_bwd(c, e, c.B, mod, ld_mod, dm_rows));
    return flow_in_bwd(c, e, dx, ld_dx, resid, ld_resid);
}

// One wgrad GEMM + one column-sum per weight over the E*B stacked rows.
// layer_ready: NULL, or L events; entry l is recorded when layers.l.mlp.{fc1,fc2}.{weight,bias} gradients are final
// parts: kWgOuter = output / input projections and the final norm, kWgLayers = the hidden layers (both by default)
enum { kWgOuter = 1, kWgLayers = 2 };
int flow_weight_grads(const FlowCtx& c, const vita_flownet_weights& g, float* cs_part, void* const* layer_ready = nullptr,
                      int parts = kWgOuter | kWgLayers) {
    const vita_flownet_weights* w = c.w;
    const int D = w->D, H = w->H, FF = w->FF, L = w->L, R = c.E * c.B;
    // bias-like gradients (column sums of buffers the chains left behind) first, batched: one launch per part at few rows, and
    // every layer's event below still covers that layer's biases
    ColsumBatch cb(cs_part, c.st);
    if (parts & kWgOuter) {
        VITA_TRY(cb.add(c.f.DV, D, R, D, mut(g.out_b)));
        VITA_TRY(cb.add(c.f.RN, H, R, H, mut(g.norm_w)));
        VITA_TRY(cb.add(c.f.DHNF, H, R, H, mut(g.norm_b)));
        VITA_TRY(cb.add(c.f.DH0, H, R, H, mut(g.in_b)));
    }
    for (int l = 0; l < L && (parts & kWgLayers); ++l) {
        VITA_TRY(cb.add(c.Y2(l, 0), H, R, H, mut(g.fc2_b[l])));
        VITA_TRY(cb.add(c.U(l, 0), FF, R, FF, mut(g.fc1_b[l])));
    }
    VITA_TRY(cb.flush());
    if (parts & kWgOuter) {
        VITA_TRY(gemm_f32_wgrad_pair(wgrad(c.f.DV, D, c.HN(L, 0), H, mut(g.out_w), H, R, H, D),
                                     wgrad(c.f.DH0, H, c.f.X, D, mut(g.in_w), D, R, D, H), c.st));
    }
    Bf16Stage& bs = bf16_stage();
    const int H8 = c.H8(), FF8 = c.FF8();
    const bool tn = c.tn() && bs.mode != 0;
    for (int l = 0; l < L && (parts & kWgLayers); ++l) {
        if (tn) {
            // row-major bf16 operands as the forward / backward kernels left them: no packs, no transposed copies
            VITA_TRY(gemm_bf16_wgrad_tn(c.Y2b(l, 0), H8, c.Gb(l, 0), FF8, wgrad(c.Y2(l, 0), H, c.G(l, 0), FF, mut(g.fc2_w[l]), FF, R, FF, H), c.st));
            VITA_TRY(gemm_bf16_wgrad_tn(c.dUb(l, 0), FF8, c.HNb(l, 0), H8, wgrad(c.U(l, 0), FF, c.HN(l, 0), H, mut(g.fc1_w[l]), H, R, H, FF), c.st));
            if (layer_ready && layer_ready[l]) VITA_CHECK_HIP(hipEventRecord((hipEvent_t)layer_ready[l], c.st));
            continue;
        }
        VITA_TRY(gemm_f32_wgrad_pair(wgrad(c.Y2(l, 0), H, c.G(l, 0), FF, mut(g.fc2_w[l]), FF, R, FF, H),
                                     wgrad(c.U(l, 0), FF, c.HN(l, 0), H, mut(g.fc1_w[l]), H, R, H, FF), c.st));
        if (layer_ready && layer_ready[l]) VITA_CHECK_HIP(hipEventRecord((hipEvent_t)layer_ready[l], c.st));
    }
    return VITA_OK;
}

// ------------------------------------------------------------------------------------------------
// Side streams for the launch-bound regime (B <= 1024), opt-in (vita_loss_config.side_streams).  The backward pass has three pieces that nothing later in the
// step waits for — the decoder's weight gradients (ready once its input-gradient chain has run, while the K sampler steps
// are still to be unrolled), the time path's backward and the encoders' backward (both independent of the flow net's
// stacked weight gradients) — and at these sizes every kernel leaves most of the chip idle.  They are enqueued on three
// non-blocking side streams between a fork event recorded on the caller's stream and join events the caller's stream
// waits for before the call returns its last event, each branch with its own split-K and column-sum scratch.
// The streams and events are created on the first such call of a host thread on a device and kept (the only objects this
// library ever creates); the fork / join pattern is the one HIP graph capture records as parallel branches.
// Measured (rocprofv3 kernel trace, profiles/): the branches' kernels do not overlap the main chain's on this runtime —
// the caller's stream stalls while a side queue drains — so the B = 128 step does not get shorter (3.34 -> 3.44 ms) and
// the default stays single-stream.  The split of the stack's backward into chain + weight gradients is kept either way.
// ------------------------------------------------------------------------------------------------
struct Side {
    hipStream_t s[3] = {};
    hipEvent_t fork[2] = {}, join[3] = {};
    bool init = false, ok = false;
};
Side* side_for_current_device() {
    static thread_local Side sides[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Side& sd = sides[dev];
    if (!sd.init) {
        sd.init = true;
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i) ok = hipStreamCreateWithFlags(&sd.s[i], hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&sd.fork[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreateWithFlags(&sd.join[i], hipEventDisableTiming) == hipSuccess;
        sd.ok = ok;
        if (!ok) (void)hipGetLastError();
    }
    return sd.ok ? &sd : nullptr;
}
constexpr int kSideMaxRows = 1024;

// ------------------------------------------------------------------------------------------------
// whole-step workspace
// ------------------------------------------------------------------------------------------------
// every weight whose input-gradient GEMM dX = dY . W runs over >= kWtMinRows rows in a large-batch step: fn(W, out_f, in_f)
template <class F>
static void for_each_dgrad_weight(const vita_flownet_weights* fl, const vita_meanflow_weights* mf, const vita_decoder_weights& dec,
                                  const vita_encoder_weights& enc, F fn) {
    if (fl) {
        fn(fl->in_w, fl->H, fl->D);
        fn(fl->te3_w, fl->H, 4 * fl->TE);
        for (int l = 0; l < fl->L; ++l) {
            fn(fl->fc1_w[l], fl->FF, fl->H);
            fn(fl->fc2_w[l], fl->H, fl->FF);
            fn(fl->mod_w[l], 3 * fl->H, fl->H);
        }
        fn(fl->out_w, fl->D, fl->H);
    }
    if (mf) {
        fn(mf->in_w, mf->H, mf->D);
        for (int l = 0; l < mf->L; ++l) {
            fn(mf->fc1_w[l], mf->FF, mf->H);
            fn(mf->fc2_w[l], mf->H, mf->FF);
        }
        fn(mf->u_w, mf->D, mf->H);
        if (mf->v_w) fn(mf->v_w, mf->D, mf->H);
    }
    fn(dec.in_w, dec.HD, dec.D);
    for (int i = 0; i < dec.NL; ++i) {
        fn(dec.fc1_w[i], dec.HD, dec.HD);
        fn(dec.fc2_w[i], dec.HD, dec.HD);
    }
    fn(dec.out_w, dec.OUT, dec.HD);
    if (enc.kind == VITA_ENC_CNN) {
        for (int i = 1; i < enc.NL; ++i) fn(enc.w1[i], enc.C, enc.C * 5);
        fn(enc.proj_w, enc.D * (enc.variational ? 2 : 1), enc.C * (enc.T >> enc.NL));
    }
}
static size_t wt_floats(const vita_flownet_weights* fl, const vita_meanflow_weights* mf, const vita_decoder_weights& dec,
                        const vita_encoder_weights& enc) {
    size_t n = 0;
    for_each_dgrad_weight(fl, mf, dec, enc, [&](const float*, int o, int i) { n += (((size_t)o * i) + 63) & ~size_t(63); });
    return n;
}
// transposes every such weight into `buf` (stream-ordered, once per step) and registers the copies with the scope
static int stage_wt(WtScope& sc, float* buf, const vita_flownet_weights* fl, const vita_meanflow_weights* mf, const vita_decoder_weights& dec,
                    const vita_encoder_weights& enc, hipStream_t st) {
    int rc = VITA_OK;
    size_t off = 0;
    for_each_dgrad_weight(fl, mf, dec, enc, [&](const float* w, int o, int i) {
        float* dst = buf + off;
        off += (((size_t)o * i) + 63) & ~size_t(63);
        if (rc != VITA_OK || w == nullptr) return;
        rc = launch_transpose_f32(w, o, i, dst, st);
        sc.add(w, dst, o, i);
    });
    return rc;
}

// bf16 copies of every such weight, row form + transposed form, packed once per step (Bf16Stage::wc)
static size_t wb_halves(const vita_flownet_weights* fl, const vita_decoder_weights& dec, const vita_encoder_weights& enc) {
    size_t n = 0;
    for_each_dgrad_weight(fl, nullptr, dec, enc, [&](const float*, int o, int i) {
        n += (((size_t)o * ((i + 7) & ~7)) + 127) & ~size_t(127);
        n += (((size_t)i * ((o + 7) & ~7)) + 127) & ~size_t(127);
    });
    return n;
}
static int stage_wb(uint16_t* buf, const vita_flownet_weights* fl, const vita_decoder_weights& dec, const vita_encoder_weights& enc, bool train,
                    hipStream_t st) {
    int rc = VITA_OK;
    size_t off = 0;
    Bf16Stage& bs = bf16_stage();
    bs.nwc = 0;
    for_each_dgrad_weight(fl, nullptr, dec, enc, [&](const float* w, int o, int i) {
        uint16_t* rows = buf + off;
        off += (((size_t)o * ((i + 7) & ~7)) + 127) & ~size_t(127);
        uint16_t* t = buf + off;
        off += (((size_t)i * ((o + 7) & ~7)) + 127) & ~size_t(127);
        if (rc != VITA_OK || w == nullptr || bs.nwc >= 80) return;
        // only what a bf16-routed call of this step will ask for: the forward form always, the transposed form with gradients
        rc = launch_pack_bf16(w, i, rows, (i + 7) & ~7, o, i, st);
        if (rc == VITA_OK && train) rc = launch_pack_bf16_t(w, i, t, (o + 7) & ~7, o, i, st);
        bs.wc[bs.nwc++] = Bf16WEntry{w, rows, train ? t : nullptr, o, i};
    });
    return rc;
}

struct StepWs {
    float* wt;                        // transposed weights of the input-gradient GEMMs (train, B >= kWtMinRows)
    uint16_t* wb;                     // bf16 copies of the weights (cfg.gemm_bf16, B >= kWtMinRows)
    uint16_t* arena; size_t arena_halves;   // bf16 shadows of activations written by their producers (Bf16Stage)
    float *sk, *consts, *loss_part, *cs_part;
    float *sk_side[3], *cs_side[3];   // split-K / column-sum scratch of the side-stream branches
    float *z0, *ZZ, *dz0, *dZZ;       // ZZ = [pred; z1] (2B, D): the decoder's input rows
    float *ut, *vt, *dut, *dxt, *dxa, *dxb;
    float *params, *dparams, *kl_rows;  // variational encoder: [mean | logvar] rows (B, 2D), their gradient, per-row KL
    void* stage; size_t stage_bytes;    // bf16 operand staging (cfg.gemm_bf16)
    float *nce_enc, *nce_flow, *dz0c;   // InfoNCE state (vita_contrastive.hip) and its d z0, added to dz0 at the end
    FlowTrainWs flow;
    StackWs dec;
    StackWs senc;   // simple encoder
    CnnWs cnn;
    size_t bytes;
};

int max_colsum_n(const vita_policy_weights& w) {
    int n = w.flow.L * 3 * w.flow.H;
    auto up = [&](int v) { if (v > n) n = v; };
    up(w.flow.FF); up(w.flow.H); up(w.flow.D); up(4 * w.flow.TE);
    up(w.dec.HD); up(w.dec.OUT); up(w.dec.D);
    up(w.enc.C * 4); up(w.enc.D * (w.enc.variational ? 2 : 1));
    return n;
}

// Upper bound of the bf16 staging one GEMM of the step needs: both operands as (rows, r8(k)) bf16.  The largest calls are
// the stacked weight-gradient GEMMs (k = (1 + K) * B rows, operands of the two widest feature dims) and the forward /
// input-gradient GEMMs over the same rows; a call that would not fit takes the fp32 kernels (gemm_f32_via_bf16).
size_t bf16_stage_bytes(const vita_policy_weights& w, size_t B, size_t K) {
    size_t rows = (1 + K) * B;
    auto upr = [&](size_t v) { if (v > rows) rows = v; };
    upr(2 * B);
    upr(w.enc.kind == VITA_ENC_SIMPLE ? B * w.enc.T : B * (size_t)(w.enc.T / 2 > 0 ? w.enc.T / 2 : 1));
    size_t d1 = 0, d2 = 0;   // the two widest feature dims any GEMM pairs
    auto upd = [&](size_t v) { if (v > d1) { d2 = d1; d1 = v; } else if (v > d2) { d2 = v; } };
    upd(w.flow.FF); upd(w.flow.H); upd(w.flow.D); upd(w.dec.HD); upd(w.dec.HD); upd(w.dec.OUT); upd(w.O);
    upd((size_t)w.enc.C * (w.enc.kind == VITA_ENC_SIMPLE ? 4 : 5)); upd(w.enc.C); upd((size_t)w.enc.D * (w.enc.variational ? 2 : 1));
    const size_t r8rows = (rows + 7) & ~size_t(7), a = (d1 + 7) & ~size_t(7), b = (d2 + 7) & ~size_t(7);
    const size_t by_rows = (a + b) * r8rows * 2;                 // wgrad: (d1, rows) + (d2, rows); fwd/dgrad: (rows, d) + (d, d')
    const size_t by_w = (r8rows * a + a * a) * 2;
    return ((by_rows > by_w ? by_rows : by_w) + 1023) & ~size_t(255);
}

// lab switch: VITA_WGRAD_TN=0 keeps the transposed-pack route of the weight-gradient GEMMs (A/B measurements)
bool wgrad_tn_enabled() {
    static const bool on = [] { const char* e = getenv("VITA_WGRAD_TN"); return e == nullptr || e[0] != '0'; }();
    return on;
}

StepWs carve_step(void* ws, const vita_policy_weights& w, const vita_loss_config& cfg, size_t B, bool train) {
    Carver c(ws);
    StepWs s{};
    const size_t D = w.flow.D;
    const bool samples = cfg.decode_flow_latents && !cfg.freeze_encoder && !cfg.freeze_decoder;   // vita_policy.py:153
    const size_t K = samples ? (size_t)cfg.num_steps : 0;
    s.sk = c.take(kSplitkBytes / sizeof(float));
    s.consts = c.take(16);
    s.loss_part = c.take((size_t)vita_loss_partials_count());
    s.cs_part = c.take((size_t)vita_colsum_partials_count(max_colsum_n(w)));
    for (int i = 0; i < 3; ++i) {
        const bool side = train && B <= (size_t)kSideMaxRows;
        s.sk_side[i] = side ? c.take(kSplitkBytes / sizeof(float)) : nullptr;
        s.cs_side[i] = side ? c.take((size_t)vita_colsum_partials_count(max_colsum_n(w))) : nullptr;
    }
    s.z0 = c.take(B * D);
    s.ZZ = c.take(2 * B * D);
    s.ut = c.take(B * D);
    s.vt = c.take(B * D);
    if (train) {
        s.dz0 = c.take(B * D);
        s.dZZ = c.take(2 * B * D);
        s.dut = c.take(B * D);
        s.dxt = c.take(B * D);
        s.dxa = c.take(B * D);
        s.dxb = c.take(B * D);
    }
    const bool nce_e = cfg.enc_contrastive_weight > 0.f, nce_f = samples && cfg.flow_contrastive_weight > 0.f;
    if (nce_e) s.nce_enc = c.take(nce_workspace_bytes((int)B, (int)D) / sizeof(float));
    if (nce_f) s.nce_flow = c.take(nce_workspace_bytes((int)B, (int)D) / sizeof(float));
    if (train && (nce_e || nce_f)) s.dz0c = c.take(B * D);
    if (w.enc.variational) {
        s.params = c.take(B * 2 * D);
        s.kl_rows = c.take(B);
        if (train) s.dparams = c.take(B * 2 * D);
    }
    if (cfg.gemm_bf16) {
        s.stage_bytes = bf16_stage_bytes(w, B, K);
        s.stage = c.take(s.stage_bytes / sizeof(float));
    }
    s.wt = (train && B >= (size_t)kWtMinRows) ? c.take(wt_floats(&w.flow, nullptr, w.dec, w.enc)) : nullptr;
    s.wb = (cfg.gemm_bf16 && B >= (size_t)kWtMinRows) ? reinterpret_cast<uint16_t*>(c.take((wb_halves(&w.flow, w.dec, w.enc) + 1) / 2)) : nullptr;
    const bool tn = train && cfg.gemm_bf16 && B >= (size_t)kWtMinRows && w.flow.H % 8 == 0 && w.flow.FF % 8 == 0 && wgrad_tn_enabled();
    if (cfg.gemm_bf16 && B >= (size_t)kWtMinRows && !tn) {      // (the same predicate as wb: when only measuring, every pointer is null)
        s.arena_halves = 2 * B * ((((size_t)w.flow.H + 7) & ~size_t(7)) + (((size_t)w.flow.FF + 7) & ~size_t(7)));
        s.arena = reinterpret_cast<uint16_t*>(c.take((s.arena_halves + 1) / 2));
    }
    s.flow = carve_flow_train(c, w.flow, B, K, train, tn);
    if (!cfg.external_decoder)
        s.dec = carve_stack(c, stack_of(w.dec), 2 * B, train, train && wgrad_tn_enabled() && stack_tn_ok(stack_of(w.dec), 2 * B, cfg.gemm_bf16));
    if (w.enc.kind == VITA_ENC_SIMPLE)
        s.senc = carve_stack(c, stack_of_simple_encoder(w.enc), B * w.enc.T, train,
                             train && wgrad_tn_enabled() && stack_tn_ok(stack_of_simple_encoder(w.enc), B * w.enc.T, cfg.gemm_bf16));
    else if (w.enc.kind == VITA_ENC_CNN) s.cnn = carve_cnn(c, w.enc, B, train);
    s.bytes = c.off;
    return s;
}

int check_enc_weights(const vita_encoder_weights* w, const char* who) {
    if (w->kind == VITA_ENC_EXTERNAL) {      // the caller's own action encoder: proj_w = its output (B, D), nothing else is read
        VITA_REQUIRE(w->A > 0 && w->T > 0 && w->D > 0 && w->proj_w != nullptr && w->variational == 0,
                     "%s: an external encoder needs A, T, D, its latents in proj_w, and variational = 0", who);
        return VITA_OK;
    }
    VITA_REQUIRE(w->kind == VITA_ENC_CNN || w->kind == VITA_ENC_SIMPLE, "%s: unknown encoder kind %d", who, w->kind);
    VITA_REQUIRE(w->A > 0 && w->T > 0 && w->C > 0 && w->D > 0 && w->NL >= 0 && w->NL <= VITA_MAX_LAYERS, "%s: bad encoder dims", who);
    VITA_REQUIRE(w->proj_w && w->proj_b, "%s: null encoder projection", who);
    VITA_REQUIRE(w->variational == 0 || w->variational == 1, "%s: encoder `variational` must be 0 or 1", who);
    VITA_REQUIRE(!w->variational || w->kind == VITA_ENC_CNN, "%s: the variational encoder is the CNN one (action_vae.py:400-418)", who);
    if (w->kind == VITA_ENC_CNN) {
        VITA_REQUIRE((w->T >> w->NL) >= 1 && (w->T % (1 << w->NL)) == 0,
                     "%s: cnn encoder needs pred_horizon %% 2^num_layers == 0 (T=%d, NL=%d; action_ae.py:311)", who, w->T, w->NL);
        for (int i = 0; i < w->NL; ++i) VITA_REQUIRE(w->w1[i] && w->b1[i], "%s: null conv weight %d", who, i);
    } else {
        VITA_REQUIRE(w->C * w->T == w->D, "%s: simple encoder needs action_latent_dim * pred_horizon == latent_dim", who);
        for (int i = 0; i < w->NL; ++i) VITA_REQUIRE(w->w1[i] && w->b1[i] && w->w2[i] && w->b2[i], "%s: null encoder weight %d", who, i);
    }
    return VITA_OK;
}

int check_policy(const vita_policy_weights* w, const char* who, bool external_decoder = false) {
    VITA_REQUIRE(w != nullptr, "%s: null weights", who);
    VITA_REQUIRE(w->O > 0 && w->obs_w && w->obs_b, "%s: bad obs_encoder", who);
    VITA_TRY(check_flow_weights(&w->flow, who));
    if (external_decoder) VITA_REQUIRE(w->dec.D > 0 && w->dec.OUT > 0 && w->dec.in_w != nullptr,
                                       "%s: an external decoder needs D, OUT and the (2B, D) latents buffer in dec.in_w", who);
    else VITA_TRY(check_dec_weights(&w->dec, who));
    VITA_TRY(check_enc_weights(&w->enc, who));
    VITA_REQUIRE(w->flow.D == w->dec.D && w->flow.D == w->enc.D, "%s: latent dims differ (flow %d, decoder %d, encoder %d)", who, w->flow.D,
                 w->dec.D, w->enc.D);
    VITA_REQUIRE(w->dec.OUT == w->enc.T * w->enc.A, "%s: decoder OUT=%d != T*A=%d", who, w->dec.OUT, w->enc.T * w->enc.A);
    VITA_REQUIRE(w->flow.H <= 2048, "%s: training supports hidden_dim <= 2048 (got %d)", who, w->flow.H);
    return VITA_OK;
}

int zero_encoder_grads(const vita_encoder_weights& g, hipStream_t st) {
    auto z = [&](const float* p, size_t n) -> int {
        VITA_CHECK_HIP(hipMemsetAsync(mut(p), 0, n * sizeof(float), st));
        return VITA_OK;
    };
    if (g.kind == VITA_ENC_EXTERNAL) return VITA_OK;     // d loss / d latents is zeroed by the caller of this (it knows B)
    if (g.kind == VITA_ENC_SIMPLE) {
        VITA_TRY(z(g.proj_w, (size_t)g.C * g.A)); VITA_TRY(z(g.proj_b, (size_t)g.C));
        for (int i = 0; i < g.NL; ++i) {
            VITA_TRY(z(g.w1[i], (size_t)4 * g.C * g.C)); VITA_TRY(z(g.b1[i], (size_t)4 * g.C));
            VITA_TRY(z(g.w2[i], (size_t)4 * g.C * g.C)); VITA_TRY(z(g.b2[i], (size_t)g.C));
        }
    } else {
        for (int i = 0; i < g.NL; ++i) {
            VITA_TRY(z(g.w1[i], (size_t)g.C * (i == 0 ? g.A : g.C) * 5)); VITA_TRY(z(g.b1[i], (size_t)g.C));
        }
        VITA_TRY(z(g.proj_w, (size_t)proj_out(g) * (g.NL ? g.C : g.A) * (g.T >> g.NL))); VITA_TRY(z(g.proj_b, (size_t)proj_out(g)));
    }
    return VITA_OK;
}

// the caller's own decoder (vita_loss_config.external_decoder): a descriptor whose decoder owns nothing
vita_policy_weights without_decoder(const vita_policy_weights& w) {
    vita_policy_weights wl = w;
    wl.dec = vita_decoder_weights{};
    wl.dec.D = w.dec.D;
    wl.dec.OUT = w.dec.OUT;
    return wl;
}

int encoder_fwd(const vita_encoder_weights& w, const StepWs& s, const float* actions, float* z1, int64_t ldz, int B, hipStream_t st) {
    if (w.kind == VITA_ENC_EXTERNAL) {       // the latents were computed by the caller's module (contiguous (B, D) rows)
        VITA_CHECK_HIP(hipMemcpy2DAsync(z1, (size_t)ldz * 4, w.proj_w, (size_t)w.D * 4, (size_t)w.D * 4, (size_t)B, hipMemcpyDeviceToDevice, st));
        return VITA_OK;
    }
    if (w.kind == VITA_ENC_SIMPLE) {
        VITA_REQUIRE(ldz == w.D, "simple encoder output must be contiguous");
        return stack_fwd(stack_of_simple_encoder(w), s.senc, actions, w.A, B * w.T, z1, st);
    }
    return cnn_fwd(w, s.cnn, actions, z1, ldz, B, st);
}

}  // namespace

}  // namespace vita

using namespace vita;

extern "C" {

size_t vita_compute_loss_workspace_bytes(const vita_policy_weights* w, const vita_loss_config* cfg, int32_t B) {
    if (!w || !cfg || B <= 0) return 0;
    if (cfg->external_decoder) {
        const vita_policy_weights wl = without_decoder(*w);
        return carve_step(nullptr, wl, *cfg, (size_t)B, true).bytes;
    }
    return carve_step(nullptr, *w, *cfg, (size_t)B, true).bytes;
}

int vita_compute_loss_f32(const vita_policy_weights* w, const vita_policy_weights* grads, const vita_loss_config* cfg,
                          const float* obs_features, int64_t ldobs, const float* actions, const float* t, const int64_t* pair_i,
                          const int64_t* pair_j, const float* fm_noise, const float* enc_noise, int32_t B, float* losses, float* d_obs_features, int64_t ld_dobs,
                          void* const* grad_ready, void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_compute_loss_f32";
    VITA_REQUIRE(cfg != nullptr, "%s: null config", who);
    const bool ext_dec = cfg->external_decoder != 0, resume = cfg->resume_backward != 0;
    VITA_TRY(check_policy(w, who, ext_dec));
    if (grads) VITA_TRY(check_policy(grads, "vita_compute_loss_f32 (grads)", ext_dec));
    VITA_REQUIRE(!ext_dec || cfg->gemm_bf16 == 0, "%s: an external decoder is built for the fp32 step only", who);
    VITA_REQUIRE(ext_dec ? (resume == (grads != nullptr)) : !resume,
                 "%s: with an external decoder the call is forward-only (grads NULL) or resume_backward with grads; resume_backward needs one", who);
    // external decoder: dec.in_w of the two descriptors carries [pred; z1] out and d loss / d [pred; z1] in; nothing else of it is used
    float* const lat_out = ext_dec ? const_cast<float*>(w->dec.in_w) : nullptr;
    const float* const dlat_in = (ext_dec && grads) ? grads->dec.in_w : nullptr;
    vita_policy_weights w_local, g_local;
    if (ext_dec) {
        w_local = without_decoder(*w);
        w = &w_local;
        if (grads) { g_local = without_decoder(*grads); grads = &g_local; }
    }
    VITA_REQUIRE(!cfg->decode_flow_latents || cfg->num_steps >= 1, "%s: num_steps must be >= 1 (got %d)", who, cfg->num_steps);
    VITA_REQUIRE(cfg->consistency_weight >= 0.f && cfg->enc_recon_weight >= 0.f && cfg->flow_recon_weight >= 0.f, "%s: negative loss weight", who);
    VITA_REQUIRE(cfg->enc_contrastive_weight >= 0.f && cfg->flow_contrastive_weight >= 0.f && cfg->temperature >= 0.f,
                 "%s: negative contrastive weight / temperature", who);
    VITA_REQUIRE(B >= 1, "%s: B must be >= 1 (the mean over an empty batch is undefined)", who);
    VITA_REQUIRE(cfg->kl_weight >= 0.f && cfg->gemm_bf16 >= 0 && cfg->gemm_bf16 <= 2, "%s: bad kl_weight / gemm_bf16", who);
    VITA_REQUIRE(cfg->fm_kind >= VITA_FM_CONDITIONAL && cfg->fm_kind <= VITA_FM_SCHRODINGER && cfg->fm_sigma >= 0.f, "%s: bad fm_kind / fm_sigma", who);
    VITA_REQUIRE(enc_noise == nullptr || w->enc.variational, "%s: enc_noise is the variational encoder's draw; this encoder is not variational", who);
    VITA_REQUIRE(obs_features && actions && t && losses && ldobs >= w->O, "%s: bad obs_features/actions/t/losses", who);
    VITA_REQUIRE(d_obs_features == nullptr || (grads != nullptr && ld_dobs >= w->O), "%s: d_obs_features needs grads and ld_dobs >= O", who);
    if (grads) {
        VITA_REQUIRE(grads->O == w->O && grads->flow.D == w->flow.D && grads->flow.H == w->flow.H && grads->flow.FF == w->flow.FF &&
                         grads->flow.L == w->flow.L && grads->flow.TE == w->flow.TE && grads->dec.HD == w->dec.HD &&
                         grads->dec.NL == w->dec.NL && grads->dec.OUT == w->dec.OUT && grads->enc.kind == w->enc.kind &&
                         grads->enc.C == w->enc.C && grads->enc.NL == w->enc.NL && grads->enc.A == w->enc.A && grads->enc.T == w->enc.T &&
                         grads->enc.variational == w->enc.variational,
                     "%s: the gradient descriptor's dims differ from the weights'", who);
    }
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_compute_loss_workspace_bytes(w, cfg, B), who));
    hipStream_t st = (hipStream_t)stream;
    const bool train = grads != nullptr;
    StepWs s = carve_step(workspace, *w, *cfg, (size_t)B, true);
    // the carve on the real base must be the carve that was measured (a region sized by a pointer test would not be: when measuring
    // every pointer is null): refuse, rather than write past the workspace
    VITA_REQUIRE(s.bytes <= workspace_bytes, "%s: internal error: the workspace carve (%zu bytes) exceeds the measured size (%zu)", who, s.bytes,
                 workspace_bytes);
    const int D = w->flow.D, H = w->flow.H, L = w->flow.L, OUT = w->dec.OUT;
    const bool fe = cfg->freeze_encoder != 0, fd = cfg->freeze_decoder != 0;
    const bool do_sample = cfg->decode_flow_latents != 0 && !fe && !fd;                                                // :153
    const int K = do_sample ? cfg->num_steps : 0;
    const bool ride = do_sample && B <= kRideAlongMaxRows;   // Euler step 0 rides along with the flow-matching rows
    // time-path rows: [flow-matching t (B) | step 0: zeros (B) | steps 1..K-1] when riding along, else [t (B) | steps 0..K-1]
    const int T = K > 0 ? (ride ? 2 * B + K - 1 : B + K) : B;
    const int step_row0 = ride ? 2 * B - 1 : B;                // time-path row of Euler step k: step_row0 + k
    const bool do_cons = do_sample && cfg->consistency_weight > 0.f;
    const bool dec_pred = do_sample && cfg->flow_recon_weight > 0.f && !ext_dec;   // external decoder: its terms are the caller's
    const bool dec_enc = cfg->enc_recon_weight > 0.f && !fd && !ext_dec;                                               // :183
    const bool nce_enc = cfg->enc_contrastive_weight > 0.f;
    const bool nce_flow = do_sample && cfg->flow_contrastive_weight > 0.f;
    const float tau = cfg->temperature > 0.f ? cfg->temperature : 0.07f;
    const int64_t ldm = (int64_t)L * 3 * H;
    float* pred = s.ZZ;
    float* z1 = s.ZZ + (size_t)B * D;
    FlowCtx fc{&w->flow, s.flow, B, K, 1 + K, st};
    SplitkScope skscope(s.sk, kSplitkBytes);
    // few-row GEMMs with a split reduction finish it in the kernel (vita_splitk_fused.h): one memset here instead of ~50 reduce launches
    static const bool no_fused = getenv("VITA_NO_FUSED_SPLITK") != nullptr && atoi(getenv("VITA_NO_FUSED_SPLITK")) != 0;
    if (!no_fused) VITA_REQUIRE(skscope.enable_fused(st) == 0, "%s: could not zero the split-K counters", who);
    Bf16Scope bfscope(s.stage, s.stage_bytes, cfg->gemm_bf16);
    WtScope wtscope;
    if (s.wt && cfg->gemm_bf16 != 2) VITA_TRY(stage_wt(wtscope, s.wt, &w->flow, nullptr, w->dec, w->enc, st));
    if (s.wb && !resume) VITA_TRY(stage_wb(s.wb, &w->flow, w->dec, w->enc, train, st));
    bf16_stage().arena = s.arena; bf16_stage().arena_halves = s.arena_halves;
    // the bf16 staging region is reused call after call in stream order: one stream only in that mode
    Side* side = (train && B <= kSideMaxRows && cfg->gemm_bf16 == 0 && cfg->side_streams != 0) ? side_for_current_device() : nullptr;
    const bool vae = w->enc.variational != 0;
    const bool do_kl = vae && !fe && !(fe && fd) && cfg->kl_weight > 0.f;                                               // :142-146

    auto step_mod = [&](int k) { return s.flow.mod + (size_t)(step_row0 + k) * ldm; };   // one broadcast row per step
    const bool plain_cfm = cfg->fm_kind == VITA_FM_CONDITIONAL && cfg->fm_sigma == 0.f;
    const float dt = K > 0 ? (float)(1.0 / (double)K) : 0.f;
    // both decoder calls as one batch of rows [r0, r0 + M) of ZZ = [pred; z1]      :174-187
    const Stack dec = stack_of(w->dec);
    const int r0 = dec_pred ? 0 : B;
    const int M = (dec_pred ? B : 0) + (dec_enc ? B : 0);
    const int enc_row = dec_pred ? B : 0;   // row of the decoder batch where dec(z1) starts
    const int64_t n_rec = (int64_t)B * OUT;

    // resume_backward: the forward ran in an earlier call on this workspace (everything the backward reads is still there)
    if (!resume) {
    // loss weights as device scalars (seeds of the backward pass) ; zero the outputs
    const float seeds[6] = {1.0f, cfg->consistency_weight, cfg->flow_recon_weight, cfg->enc_recon_weight, cfg->enc_contrastive_weight,
                            cfg->flow_contrastive_weight};
    VITA_TRY(launch_set_scalars(s.consts, seeds, 6, st));
    VITA_CHECK_HIP(hipMemsetAsync(losses, 0, sizeof(float) * VITA_LOSS_COUNT, st));

    // ---- forward ------------------------------------------------------------------------------
    VITA_TRY(gemm_f32(lin(obs_features, ldobs, w->obs_w, w->O, w->obs_b, s.z0, D, B, D, w->O, VITA_ACT_NONE), st));   // :114
    if (vae) {                                                                                                        // :121
        VITA_TRY(encoder_fwd(w->enc, s, actions, s.params, 2 * D, B, st));
        VITA_TRY(launch_gauss_sample(s.params, 2 * D, enc_noise, z1, D, B, D, st));
        // deterministic posterior: kl() is zeros (dists.py:25-26) and the loss entry stays 0
        if (do_kl && enc_noise) VITA_TRY(vita_gaussian_kl_fwd_f32(s.params, 2 * D, s.params + D, 2 * D, B, D, s.kl_rows, losses + VITA_LOSS_ACTION_KL, st));
    } else {
        VITA_TRY(encoder_fwd(w->enc, s, actions, z1, D, B, st));                                                      // :123
    }
    // time path for the B per-row timesteps of the flow-matching draw and the K Euler steps
    VITA_CHECK_HIP(hipMemcpyAsync(s.flow.ts, t, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    if (K > 0) {
        if (ride && B > 1) VITA_CHECK_HIP(hipMemsetAsync(s.flow.ts + B, 0, (size_t)(B - 1) * 4, st));
        VITA_TRY(launch_step_times(s.flow.ts + step_row0, K, st));   // ts[step_row0 + k] = k/K (k = 0 closes the zero block)
    }
    VITA_TRY(time_path_fwd(fc, T));
    // flow-matching loss, flow_matchers.py:30-36
    if (plain_cfm) VITA_TRY(vita_cfm_interpolate_f32(s.z0, z1, t, pair_i, pair_j, fc.X(0), s.ut, B, D, st));
    else VITA_TRY(vita_fm_interpolate_f32(cfg->fm_kind, (double)cfg->fm_sigma, s.z0, z1, t, fm_noise, pair_i, pair_j, fc.X(0), s.ut, B, D, st));
    // The flow-matching evaluation and Euler step 0 are independent and their rows are adjacent in every stacked buffer:
    // they run as ONE 2B-row pass with per-row modulation (one evaluation's worth of launches less per step, forward and
    // backward; at the default B = 128 the step is launch-bound).  Only out_proj differs (v vs x + dt*v) and stays per evaluation.
    if (do_sample) VITA_CHECK_HIP(hipMemcpyAsync(fc.X(1), s.z0, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
    if (ride) {
        VITA_TRY(flow_trunk_fwd(fc, 0, 2 * B, s.flow.mod, ldm));
        VITA_TRY(flow_out_fwd(fc, 0, s.vt, D, nullptr, 0, 0.f));
        VITA_TRY(flow_out_fwd(fc, 1, K == 1 ? pred : fc.X(2), D, fc.X(1), D, dt));
    } else {
        VITA_TRY(flow_eval_fwd(fc, 0, s.flow.mod, ldm, s.vt, D, nullptr, 0, 0.f));
    }
    VITA_TRY(vita_mse_loss_fwd_f32(s.vt, s.ut, (int64_t)B * D, s.loss_part, losses + VITA_LOSS_FLOW, st));
    if (nce_enc) VITA_TRY(nce_forward(s.z0, D, z1, D, B, D, tau, losses + VITA_LOSS_ENC_CONTRASTIVE, s.nce_enc, st));              // :134-139
    // Euler sampler with grad, vita_policy.py:153-160 / flow_matchers.py:55-79
    if (do_sample) {
        for (int k = ride ? 1 : 0; k < K; ++k) {
            const int e = k + 1;
            float* dst = (k == K - 1) ? pred : fc.X(e + 1);
            VITA_TRY(flow_eval_fwd(fc, e, step_mod(k), 0, dst, D, fc.X(e), D, dt));
        }
        if (do_cons) VITA_TRY(vita_mse_loss_fwd_f32(pred, z1, (int64_t)B * D, s.loss_part, losses + VITA_LOSS_CONSISTENCY, st));   // :162-165
        if (nce_flow) VITA_TRY(nce_forward(s.z0, D, pred, D, B, D, tau, losses + VITA_LOSS_FLOW_CONTRASTIVE, s.nce_flow, st));   // :167-172
    }
    if (ext_dec) {   // hand [pred; z1] to the caller's decoder (pred rows are zeros when the sampler branch is off)
        VITA_CHECK_HIP(hipMemcpyAsync(lat_out, s.ZZ, (size_t)2 * B * D * 4, hipMemcpyDeviceToDevice, st));
        if (!do_sample) VITA_CHECK_HIP(hipMemsetAsync(lat_out, 0, (size_t)B * D * 4, st));
    }
    if (M > 0) {
        VITA_TRY(stack_fwd(dec, s.dec, s.ZZ + (size_t)r0 * D, D, M, nullptr, st));
        auto recon = cfg->recon_l1 ? vita_l1_loss_fwd_f32 : vita_mse_loss_fwd_f32;
        if (dec_pred) VITA_TRY(recon(s.dec.out, actions, n_rec, s.loss_part, losses + VITA_LOSS_FLOW_RECON, st));
        if (dec_enc) VITA_TRY(recon(s.dec.out + (size_t)enc_row * OUT, actions, n_rec, s.loss_part, losses + VITA_LOSS_ENC_RECON, st));
    }
    const float wsum[8] = {0.f, 1.0f, do_cons ? cfg->consistency_weight : 0.f, dec_pred ? cfg->flow_recon_weight : 0.f,
                           dec_enc ? cfg->enc_recon_weight : 0.f, nce_enc ? cfg->enc_contrastive_weight : 0.f,
                           nce_flow ? cfg->flow_contrastive_weight : 0.f, do_kl ? cfg->kl_weight : 0.f};
    VITA_TRY(launch_weighted_sum(losses, wsum, 8, losses + VITA_LOSS_TOTAL, st));
    }   // !resume
    if (!train) return VITA_OK;

    // ---- backward -----------------------------------------------------------------------------
    const vita_policy_weights& g = *grads;
    auto record = [&](int i) -> int {
        if (grad_ready && grad_ready[i]) VITA_CHECK_HIP(hipEventRecord((hipEvent_t)grad_ready[i], st));
        return VITA_OK;
    };
    void* const* layer_ready = (grad_ready && cfg->grad_ready_layers) ? grad_ready + 3 : nullptr;
    // decoder: dZZ = d loss / d [pred; z1] through the reconstruction terms
    if (ext_dec) {   // d (the caller's decoder terms) / d [pred; z1]
        VITA_CHECK_HIP(hipMemcpyAsync(s.dZZ, dlat_in, (size_t)2 * B * D * 4, hipMemcpyDeviceToDevice, st));
    } else {
        if (!dec_pred) VITA_CHECK_HIP(hipMemsetAsync(s.dZZ, 0, (size_t)B * D * 4, st));
        if (!dec_enc) VITA_CHECK_HIP(hipMemsetAsync(s.dZZ + (size_t)B * D, 0, (size_t)B * D * 4, st));
    }
    if (ext_dec) {
        // nothing of the decoder's is computed here
    } else if (M > 0) {
        auto recon_bwd = cfg->recon_l1 ? vita_l1_loss_bwd_f32 : vita_mse_loss_bwd_f32;
        if (dec_pred) VITA_TRY(recon_bwd(s.dec.out, actions, n_rec, s.consts + 2, s.dec.dOut, nullptr, 0, st));
        if (dec_enc)
            VITA_TRY(recon_bwd(s.dec.out + (size_t)enc_row * OUT, actions, n_rec, s.consts + 3, s.dec.dOut + (size_t)enc_row * OUT, nullptr, 0, st));
        VITA_TRY(stack_chain_bwd(dec, s.dec, M, s.dec.dOut, OUT, s.dZZ + (size_t)r0 * D, D, st));
        if (side) {   // the decoder's weight gradients run beside the sampler's backward
            VITA_CHECK_HIP(hipEventRecord(side->fork[0], st));
            VITA_CHECK_HIP(hipStreamWaitEvent(side->s[0], side->fork[0], 0));
            {
                SplitkScope sc(s.sk_side[0], kSplitkBytes);
                VITA_TRY(stack_wgrads(dec, stack_of(g.dec), s.dec, s.ZZ + (size_t)r0 * D, D, M, nullptr, s.dec.dOut, OUT, s.cs_side[0], side->s[0]));
            }
            if (grad_ready && grad_ready[0]) VITA_CHECK_HIP(hipEventRecord((hipEvent_t)grad_ready[0], side->s[0]));
            VITA_CHECK_HIP(hipEventRecord(side->join[0], side->s[0]));
        } else {
            VITA_TRY(stack_wgrads(dec, stack_of(g.dec), s.dec, s.ZZ + (size_t)r0 * D, D, M, nullptr, s.dec.dOut, OUT, s.cs_part, st));
        }
    } else {  // the decoder takes no part in the loss
        const Stack gd = stack_of(g.dec);
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.in_w), 0, (size_t)gd.TR * gd.IN * 4, st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.in_b), 0, (size_t)gd.TR * 4, st));
        for (int i = 0; i < gd.NL; ++i) {
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc1_w[i]), 0, (size_t)gd.FF * gd.TR * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc1_b[i]), 0, (size_t)gd.FF * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc2_w[i]), 0, (size_t)gd.TR * gd.FF * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc2_b[i]), 0, (size_t)gd.TR * 4, st));
        }
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.out_w), 0, (size_t)gd.OUT * gd.TR * 4, st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.out_b), 0, (size_t)gd.OUT * 4, st));
    }
    const bool dec_on_side = side != nullptr && M > 0;
    if (!dec_on_side) VITA_TRY(record(0));
    float* dpred = s.dZZ;
    float* dz1 = s.dZZ + (size_t)B * D;
    if (do_cons) VITA_TRY(vita_mse_loss_bwd_f32(pred, z1, (int64_t)B * D, s.consts + 1, dpred, dz1, 1, st));
    // InfoNCE terms: d pred / d z1 join dZZ now (before the sampler is unrolled); their d z0 waits in dz0c because the
    // sampler's first step OVERWRITES dz0 below
    if (nce_flow) VITA_TRY(nce_backward(B, D, tau, s.consts + 5, s.dz0c, D, 0, dpred, D, 1, s.nce_flow, st));
    if (nce_enc) VITA_TRY(nce_backward(B, D, tau, s.consts + 4, s.dz0c, D, nce_flow ? 1 : 0, dz1, D, 1, s.nce_enc, st));
    // flow-matching evaluation
    VITA_TRY(vita_mse_loss_bwd_f32(s.vt, s.ut, (int64_t)B * D, s.consts + 0, fc.DV(0), s.dut, 0, st));
    // Euler steps, last to first: dx_k = dx_{k+1} + J_k^T (dt * dx_{k+1}); step 0 shares its pass with the flow-matching rows
    if (do_sample) {
        const float* dxn = dpred;
        for (int k = K - 1; k >= (ride ? 1 : 0); --k) {
            const int e = k + 1;
            VITA_TRY(vita_gate_bwd_f32(dxn, nullptr, nullptr, 0, dt, fc.DV(e), nullptr, 0, B, D, st));
            float* dst = (k == 0) ? s.dz0 : ((k & 1) ? s.dxa : s.dxb);
            VITA_TRY(flow_eval_bwd(fc, e, step_mod(k), 0, s.flow.DM, dst, D, dxn, D));
            if (L > 0) VITA_TRY(vita_colsum_f32(s.flow.DM, ldm, B, L * 3 * H, s.cs_part, s.flow.dmod + (size_t)(step_row0 + k) * ldm, 0, st));
            dxn = dst;
        }
        if (ride) {
            VITA_TRY(vita_gate_bwd_f32(dxn, nullptr, nullptr, 0, dt, fc.DV(1), nullptr, 0, B, D, st));
            VITA_TRY(flow_chain_bwd(fc, 0, 2 * B, s.flow.mod, ldm, s.flow.dmod));   // per-row modulation gradients: table rows 0..2B-1
            VITA_TRY(flow_in_bwd(fc, 0, s.dxt, D, nullptr, 0));
            VITA_TRY(flow_in_bwd(fc, 1, s.dz0, D, dxn, D));
        } else {
            VITA_TRY(flow_eval_bwd(fc, 0, s.flow.mod, ldm, s.flow.dmod, s.dxt, D, nullptr, 0));
        }
    } else {
        VITA_TRY(flow_eval_bwd(fc, 0, s.flow.mod, ldm, s.flow.dmod, s.dxt, D, nullptr, 0));
        VITA_CHECK_HIP(hipMemsetAsync(s.dz0, 0, (size_t)B * D * 4, st));
    }
    if (plain_cfm) VITA_TRY(vita_cfm_interpolate_bwd_f32(s.dxt, s.dut, t, pair_i, pair_j, s.dz0, dz1, B, D, 1, st));
    else VITA_TRY(vita_fm_interpolate_bwd_f32(cfg->fm_kind, (double)cfg->fm_sigma, s.dxt, s.dut, t, pair_i, pair_j, s.dz0, dz1, B, D, 1, st));
    if (nce_enc || nce_flow) VITA_TRY(launch_axpy(s.dz0c, s.dz0, (int64_t)B * D, st));
    // encoders (on their own stream when forking)
    auto encoders_bwd = [&](float* cs, hipStream_t es) -> int {
        if (w->enc.kind == VITA_ENC_EXTERNAL) {   // hand d loss / d latents back: the caller's autograd continues into its module
            if (fe) VITA_CHECK_HIP(hipMemsetAsync(mut(g.enc.proj_w), 0, (size_t)B * D * 4, es));
            else VITA_CHECK_HIP(hipMemcpyAsync(mut(g.enc.proj_w), dz1, (size_t)B * D * 4, hipMemcpyDeviceToDevice, es));
        } else if (fe) {   // :119 the action latents were computed without grad
            VITA_TRY(zero_encoder_grads(g.enc, es));
        } else if (w->enc.kind == VITA_ENC_SIMPLE) {
            VITA_TRY(stack_bwd(stack_of_simple_encoder(w->enc), stack_of_simple_encoder(g.enc), s.senc, actions, w->enc.A, B * w->enc.T, z1, dz1,
                               w->enc.C, nullptr, 0, cs, es));
        } else if (vae) {
            const float ks = (do_kl && enc_noise) ? cfg->kl_weight / (float)B : 0.f;
            VITA_TRY(launch_gauss_sample_bwd(s.params, 2 * D, enc_noise, dz1, D, ks, s.dparams, 2 * D, B, D, es));
            VITA_TRY(cnn_bwd(w->enc, g.enc, s.cnn, actions, s.dparams, 2 * D, B, cs, es));
        } else {
            VITA_TRY(cnn_bwd(w->enc, g.enc, s.cnn, actions, dz1, D, B, cs, es));
        }
        VITA_TRY(gemm_f32(wgrad(s.dz0, D, obs_features, ldobs, mut(g.obs_w), w->O, B, w->O, D), es));
        VITA_TRY(vita_colsum_f32(s.dz0, D, B, D, cs, mut(g.obs_b), 0, es));
        if (d_obs_features) VITA_TRY(gemm_f32(dgrad(s.dz0, D, w->obs_w, w->O, d_obs_features, ld_dobs, B, w->O, D), es));
        return VITA_OK;
    };
    if (side) {
        VITA_CHECK_HIP(hipEventRecord(side->fork[1], st));
        VITA_CHECK_HIP(hipStreamWaitEvent(side->s[1], side->fork[1], 0));
        VITA_CHECK_HIP(hipStreamWaitEvent(side->s[2], side->fork[1], 0));
        {
            SplitkScope sc(s.sk_side[1], kSplitkBytes);
            FlowCtx fcb = fc;
            fcb.st = side->s[1];
            VITA_TRY(time_path_bwd(fcb, g.flow, T, s.cs_side[1]));
            VITA_CHECK_HIP(hipEventRecord(side->join[1], side->s[1]));
        }
        {
            SplitkScope sc(s.sk_side[2], kSplitkBytes);
            VITA_TRY(encoders_bwd(s.cs_side[2], side->s[2]));
            VITA_CHECK_HIP(hipEventRecord(side->join[2], side->s[2]));
        }
        if (layer_ready) {   // the order of the single-stream branch below: layers, encoders, time path, outer projections
            VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part, layer_ready, kWgLayers));
            VITA_CHECK_HIP(hipStreamWaitEvent(st, side->join[2], 0));
            VITA_TRY(record(2));
            VITA_CHECK_HIP(hipStreamWaitEvent(st, side->join[1], 0));
            VITA_TRY(record(1));
            VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part, nullptr, kWgOuter));
            VITA_TRY(record(3 + L));
        } else {
            VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part));
            VITA_CHECK_HIP(hipStreamWaitEvent(st, side->join[1], 0));
            VITA_TRY(record(1));
            VITA_CHECK_HIP(hipStreamWaitEvent(st, side->join[2], 0));
        }
        if (dec_on_side) VITA_CHECK_HIP(hipStreamWaitEvent(st, side->join[0], 0));
        if (!layer_ready) VITA_TRY(record(2));
    } else if (layer_ready) {
        // 4 + L events, ordered for a data-parallel caller's link: the sections that make the most gradient bytes per unit of
        // compute first (the layers' stacked weight gradients: 33.6 MB in 0.27 ms at B = 128; the encoders: 18 MB in 0.2 ms; the
        // time path with the modulators: 15.7 MB in 0.2 ms), the 2 MB of the outer projections last — whatever the link's
        // speed, little is left to send when the backward ends (event order 0, 3 .. 3+L-1, 2, 1, 3+L)
        VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part, layer_ready, kWgLayers));
        VITA_TRY(encoders_bwd(s.cs_part, st));
        VITA_TRY(record(2));
        VITA_TRY(time_path_bwd(fc, g.flow, T, s.cs_part));
        VITA_TRY(record(1));
        VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part, nullptr, kWgOuter));
        VITA_TRY(record(3 + L));
    } else {
        VITA_TRY(flow_weight_grads(fc, g.flow, s.cs_part));
        VITA_TRY(time_path_bwd(fc, g.flow, T, s.cs_part));
        VITA_TRY(record(1));
        VITA_TRY(encoders_bwd(s.cs_part, st));
        VITA_TRY(record(2));
    }
    return VITA_OK;
}

size_t vita_action_encoder_workspace_bytes(const vita_encoder_weights* w, int32_t B) {
    if (!w || B <= 0) return 0;
    Carver c(nullptr);
    c.take(kSplitkBytes / sizeof(float));
    if (w->kind == VITA_ENC_SIMPLE) carve_stack(c, stack_of_simple_encoder(*w), (size_t)B * w->T, false);
    else carve_cnn(c, *w, (size_t)B, false);
    return c.off;
}

int vita_action_encoder_forward_f32(const vita_encoder_weights* w, const float* actions, float* latents, int64_t ldl, int32_t B,
                                    void* workspace, size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_action_encoder_forward_f32";
    VITA_REQUIRE(w != nullptr, "%s: null weights", who);
    VITA_TRY(check_enc_weights(w, who));
    VITA_REQUIRE(B >= 0, "%s: B < 0", who);
    if (B == 0) return VITA_OK;
    VITA_REQUIRE(actions && latents && ldl >= proj_out(*w), "%s: bad actions/latents (a variational encoder writes 2*D columns)", who);
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_action_encoder_workspace_bytes(w, B), who));
    Carver c(workspace);
    StepWs s{};
    SplitkScope skscope(c.take(kSplitkBytes / sizeof(float)), kSplitkBytes);
    if (w->kind == VITA_ENC_SIMPLE) s.senc = carve_stack(c, stack_of_simple_encoder(*w), (size_t)B * w->T, false);
    else s.cnn = carve_cnn(c, *w, (size_t)B, false);
    return encoder_fwd(*w, s, actions, latents, ldl, B, (hipStream_t)stream);
}

}  // extern "C"

// =================================================================================================
// MeanFlow training step: VitaPolicy.compute_loss with flow_matcher.name = "mean" over SimpleMeanFlowNet
// (mean_flow_matcher.py:109-199, use_imf = False; simple_mean_flow_net.py:63-81; vita_policy.py:112-189), forward + backward.
//
//   t, r given (sample_t_r is the host's draw, :87-107); h = t - r; z_t = (1-t)*x1 + t*x0; v = x0 - x1           :123-132
//   (u, dudt) = jvp(u(z, t, r), (z_t, t, r), (v, 1, 0))                                                           :171-176
//   err = u - stopgrad(v - h*dudt); loss_mf = mean_b( stopgrad(1 / (mean_d err^2 + c)^(1-gamma)) * mean_d err^2 ) :178-180, 10-18
//   pred = MeanFlow sampler from x0 with grad (x <- x - h*u(x, t_k, h_k), :201-233); consistency / reconstruction terms as in
//   the default policy.
// The target is detached, so only the PRIMAL evaluation is differentiated: the tangent pass is forward-only and rides the
// same weights — every Linear once more on the tangent rows without its bias, every SiLU as silu'(primal pre-activation) *
// tangent (vita_act_bwd_f32) — and the backward is the ordinary one of a residual MLP.  As in the default step the 1 + K
// evaluations are stacked so that each weight gets one wgrad GEMM; the gradient of the time conditioning telescopes:
// sum_l d x_cond_l = d x_0 - d x_L.
// =================================================================================================
namespace vita {
namespace {

constexpr int MFT = 256;
// tv/hv rows: [0,B) = (t, t - r); B + k = (1 - k/K, (1 - k/K) - (1 - (k+1)/K)) in double as the reference's python floats; ones[b] = 1
// with imf: B more rows (t, h = t - t = 0) for the no-grad evaluation that supplies the tangent (:138-140)
__global__ __launch_bounds__(MFT) void mf_prep_kernel(const float* __restrict__ t, const float* __restrict__ r, float* __restrict__ tv,
                                                     float* __restrict__ hv, float* __restrict__ ones, int B, int K, int imf) {
    const int i = blockIdx.x * MFT + threadIdx.x;
    if (i < B) { tv[i] = t[i]; hv[i] = t[i] - r[i]; ones[i] = 1.0f; }
    else if (i < B + K) {
        const int k = i - B;
        const double ts = 1.0 - (double)k / (double)K, rs = 1.0 - (double)(k + 1) / (double)K;
        tv[i] = (float)ts;
        hv[i] = (float)(ts - rs);
    } else if (imf && i < 2 * B + K) {
        const float tb = t[i - B - K];
        tv[i] = tb;
        hv[i] = tb - tb;
    }
}
// improved MeanFlow (:153-155, 21-24), one wave per row: e1 = (u + h*dudt) - v, e2 = vpred - v; ps = sum_d e^2;
// w = (ps + eps)^p; rows = ps / w; mse rows = ps / D
__global__ __launch_bounds__(MFT) void imf_loss_rows_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ dudt,
                                                           const float* __restrict__ vpred, const float* __restrict__ hv, int B, int D, float p,
                                                           float eps, float* __restrict__ r1, float* __restrict__ r2, float* __restrict__ m1,
                                                           float* __restrict__ m2, float* __restrict__ w1, float* __restrict__ w2) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float h = hv[row];
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < D; k += 64) {
        const int64_t i = (int64_t)row * D + k;
        const float e1 = (u[i] + h * dudt[i]) - v[i], e2 = vpred[i] - v[i];
        s1 += e1 * e1;
        s2 += e2 * e2;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        const float a1 = powf(s1 + eps, p), a2 = powf(s2 + eps, p);
        r1[row] = s1 / a1; r2[row] = s2 / a2;
        m1[row] = s1 / (float)D; m2[row] = s2 / (float)D;
        w1[row] = a1; w2[row] = a2;
    }
}
// du = (1/B) * 2*e1 / w1_b ; dvp = aux_w * (1/B) * 2*e2 / w2_b
__global__ __launch_bounds__(MFT) void imf_loss_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ dudt,
                                                          const float* __restrict__ vpred, const float* __restrict__ hv, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int B, int D, float aux_w, float* __restrict__ du,
                                                          float* __restrict__ dvp) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * MFT;
    for (int64_t i = (int64_t)blockIdx.x * MFT + threadIdx.x; i < n; i += stride) {
        const int row = (int)(i / D);
        const float e1 = (u[i] + hv[row] * dudt[i]) - v[i], e2 = vpred[i] - v[i];
        du[i] = ((2.0f * e1) / w1[row]) / (float)B;
        dvp[i] = aux_w * (((2.0f * e2) / w2[row]) / (float)B);
    }
}
// one wave per row: err = u - (v - h*dudt); delta = mean_d err^2; w = 1 / (delta + c)^p; rows[b] = w * delta, wrow[b] = w
__global__ __launch_bounds__(MFT) void mf_loss_rows_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ dudt,
                                                          const float* __restrict__ hv, int B, int D, float p, float c, float* __restrict__ rows,
                                                          float* __restrict__ wrow) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float h = hv[row];
    float s = 0.f;
    for (int k = lane; k < D; k += 64) {
        const int64_t i = (int64_t)row * D + k;
        const float e = u[i] - (v[i] - h * dudt[i]);
        s += e * e;
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float delta = s / (float)D;
        const float w = 1.0f / powf(delta + c, p);
        rows[row] = w * delta;
        wrow[row] = w;
    }
}
__global__ __launch_bounds__(MFT) void mf_mean_kernel(const float* __restrict__ a, int n, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += MFT) s += (double)a[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[MFT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) / (double)n);
}
// du = (1/B) * w_b * (2/D) * err      (w and the target detached)
__global__ __launch_bounds__(MFT) void mf_loss_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ dudt,
                                                         const float* __restrict__ hv, const float* __restrict__ wrow, int B, int D,
                                                         float* __restrict__ du) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * MFT;
    for (int64_t i = (int64_t)blockIdx.x * MFT + threadIdx.x; i < n; i += stride) {
        const int row = (int)(i / D);
        const float e = u[i] - (v[i] - hv[row] * dudt[i]);
        du[i] = (wrow[row] / (float)B) * ((2.0f / (float)D) * e);
    }
}
__global__ __launch_bounds__(MFT) void mf_sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * MFT;
    for (int64_t i = (int64_t)blockIdx.x * MFT + threadIdx.x; i < n; i += stride) out[i] = a[i] - b[i];
}
__global__ __launch_bounds__(MFT) void mf_sub_inplace_kernel(float* out, const float* __restrict__ b, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * MFT;
    for (int64_t i = (int64_t)blockIdx.x * MFT + threadIdx.x; i < n; i += stride) out[i] = out[i] - b[i];
}
// ---- dispersive loss (mean_flow_matcher.py:27-39) on one (B, H) feature matrix z, from its Gram matrix G = z z^T -------------
// dist_ij = G_ii + G_jj - 2 G_ij (clamped at 0; the diagonal is exactly 0); one wave per row: G <- dist, rowmax
__global__ __launch_bounds__(MFT) void disp_dist_kernel(float* __restrict__ G, int B, float* __restrict__ rowmax) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float ni = G[(int64_t)row * B + row];
    float mx = 0.f;
    for (int j = lane; j < B; j += 64) {
        const float nj = G[(int64_t)j * B + j];
        float d = (ni + nj) - 2.0f * G[(int64_t)row * B + j];
        d = (j == row || d < 0.f) ? 0.f : d;
        mx = fmaxf(mx, d);
        if (j != row) G[(int64_t)row * B + j] = d;      // the diagonal keeps the squared norm until every row has read it
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) rowmax[row] = mx;
}
// scal[0] = max(rowmax) + 1e-8
__global__ __launch_bounds__(MFT) void disp_max_kernel(const float* __restrict__ rowmax, int B, float* __restrict__ scal) {
    float mx = 0.f;
    for (int i = threadIdx.x; i < B; i += MFT) mx = fmaxf(mx, rowmax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __shared__ float ws[MFT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) scal[0] = fmaxf(fmaxf(ws[0], ws[1]), fmaxf(ws[2], ws[3])) + 1e-8f;
}
// G <- e_ij = exp(-(dist_ij / M') / tau) (diagonal: exp(0) = 1), rowsum_i = sum_j e_ij
__global__ __launch_bounds__(MFT) void disp_exp_kernel(float* __restrict__ G, int B, const float* __restrict__ scal, float tau, float* __restrict__ rowsum) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float M = scal[0];
    float s = 0.f;
    for (int j = lane; j < B; j += 64) {
        const float d = j == row ? 0.f : G[(int64_t)row * B + j];
        const float e = expf(-(d / M) / tau);
        G[(int64_t)row * B + j] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) rowsum[row] = s;
}
// scal[1] = S = mean_ij e_ij; total[0] += log(S)
__global__ __launch_bounds__(MFT) void disp_log_kernel(const float* __restrict__ rowsum, int B, float* __restrict__ scal, float* __restrict__ total) {
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += MFT) s += (double)rowsum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[MFT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S = (float)((((ws[0] + ws[1]) + (ws[2] + ws[3])) / (double)B) / (double)B);
        scal[1] = S;
        total[0] = total[0] + logf(S);
    }
}
// d (w * log S) / d z_i = 4 c (rowsum_i z_i - (E z)_i),  c = -w / (M' tau B^2 S)
__global__ __launch_bounds__(MFT) void disp_grad_kernel(const float* __restrict__ z, const float* __restrict__ ez, const float* __restrict__ rowsum,
                                                       const float* __restrict__ scal, float w, float tau, int B, int H, float* __restrict__ dz) {
    const int64_t n = (int64_t)B * H, stride = (int64_t)gridDim.x * MFT;
    const float c = -w / (scal[0] * tau * (float)B * (float)B * scal[1]);
    for (int64_t i = (int64_t)blockIdx.x * MFT + threadIdx.x; i < n; i += stride) {
        const int row = (int)(i / H);
        dz[i] = 4.0f * c * (rowsum[row] * z[i] - ez[i]);
    }
}
__global__ void mf_addw_kernel(float* __restrict__ losses, int dst, int src, float w, int keep) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float base = losses[dst];
        if (keep >= 0) losses[keep] = base;
        losses[dst] = base + w * losses[src];
    }
}

inline unsigned mfgrid(int64_t n) {
    const int64_t b = (n + MFT - 1) / MFT;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

struct MfTrainWs {
    float* wt;                                                      // transposed weights (B >= kWtMinRows), see StepWs
    float *sk, *consts, *loss_part, *cs_part;
    float *tv, *hv, *ones, *At, *St, *Ah, *Sh, *tc;              // conditioning rows T = B + K
    float *dAt, *dSt, *dtc_tan;                                    // tangent of the conditioning (B rows)
    float *X0, *Xs, *Xc, *U1, *S;                                  // stacked evaluations
    float *xd0, *xd1, *xcd, *u1d, *sd, *dudt;                      // tangent pass (B rows)
    float *z0, *ZZ, *v, *u0, *rows, *wrow;
    float *vpred, *vnet, *DV, *rows2, *wrow2, *mrow1, *mrow2;      // improved MeanFlow
    float *dispG, *dispEZ, *dispD, *dispRow, *dispSum, *dispScal;    // dispersive term: (B,B) Gram / exp matrix, E.z, per-level gradients (L,B,H)
    float *DU, *DX, *dz0, *dZZ, *dxa, *dxb, *dzt, *dtc, *dtmp, *dA1;
    float *nce_enc, *nce_flow, *dz0c;                              // InfoNCE state and its d z0 (added to dz0 at the end)
    float *params, *dparams, *kl_rows;                             // variational encoder: [mean | logvar] rows (B, 2D), their gradient, per-row KL
    StackWs dec, senc;
    CnnWs cnn;
    size_t bytes;
};
int max_colsum_n_mf(const vita_meanflow_policy_weights& w) {
    int n = w.flow.FF;
    auto up = [&](int v) { if (v > n) n = v; };
    up(w.flow.H); up(w.flow.D); up(w.dec.HD); up(w.dec.OUT); up(w.dec.D); up(w.enc.C * 4);
    up(w.enc.D * (w.enc.variational ? 2 : 1));   // the variational latent_proj's bias gradient is a (B, 2D) column sum
    return n;
}
MfTrainWs carve_mf_train(void* ws, const vita_meanflow_policy_weights& w, const vita_meanflow_loss_config& cfg, size_t B) {
    Carver c(ws);
    MfTrainWs s{};
    // improved MeanFlow: one more (forward-only) evaluation slot and B more conditioning rows
    const size_t K = (cfg.decode_flow_latents && !cfg.freeze_encoder && !cfg.freeze_decoder) ? (size_t)cfg.num_steps : 0, imf = cfg.use_imf ? 1 : 0, T = B + K + imf * B, E = 1 + K + imf;
    const size_t D = w.flow.D, H = w.flow.H, FF = w.flow.FF, L = w.flow.L;
    s.wt = B >= (size_t)kWtMinRows ? c.take(wt_floats(nullptr, &w.flow, w.dec, w.enc)) : nullptr;
    s.sk = c.take(kSplitkBytes / sizeof(float));
    s.consts = c.take(16);
    s.loss_part = c.take((size_t)vita_loss_partials_count());
    s.cs_part = c.take((size_t)vita_colsum_partials_count(max_colsum_n_mf(w)));
    s.tv = c.take(T); s.hv = c.take(T); s.ones = c.take(B);
    s.At = c.take(T * H); s.St = c.take(T * H); s.Ah = c.take(T * H); s.Sh = c.take(T * H); s.tc = c.take(T * H);
    s.dAt = c.take(B * H); s.dSt = c.take(B * H); s.dtc_tan = c.take(B * H);
    s.X0 = c.take(E * B * D);
    s.Xs = c.take((L + 1) * E * B * H);
    s.Xc = c.take((L ? L : 1) * E * B * H);
    s.U1 = c.take((L ? L : 1) * E * B * FF);
    s.S = c.take((L ? L : 1) * E * B * FF);
    s.xd0 = c.take(B * H); s.xd1 = c.take(B * H); s.xcd = c.take(B * H); s.u1d = c.take(B * FF); s.sd = c.take(B * FF); s.dudt = c.take(B * D);
    s.z0 = c.take(B * D); s.ZZ = c.take(2 * B * D); s.v = c.take(B * D); s.u0 = c.take(B * D); s.rows = c.take(B); s.wrow = c.take(B);
    if (cfg.dispersive_loss_weight > 0.f && L > 0 && B > 1) {
        s.dispG = c.take(B * B); s.dispEZ = c.take(B * H); s.dispD = c.take(L * B * H);
        s.dispRow = c.take(B); s.dispSum = c.take(B); s.dispScal = c.take(4);
    }
    if (imf) {
        s.vpred = c.take(B * D); s.vnet = c.take(B * D); s.DV = c.take(B * D);
        s.rows2 = c.take(B); s.wrow2 = c.take(B); s.mrow1 = c.take(B); s.mrow2 = c.take(B);
    }
    s.DU = c.take(E * B * D);
    s.DX = c.take((L + 1) * E * B * H);
    s.dz0 = c.take(B * D); s.dZZ = c.take(2 * B * D); s.dxa = c.take(B * D); s.dxb = c.take(B * D); s.dzt = c.take(B * D);
    s.dtc = c.take(T * H); s.dtmp = c.take(B * H); s.dA1 = c.take(T * H);
    if (cfg.enc_contrastive_weight > 0.f) s.nce_enc = c.take(nce_workspace_bytes((int)B, (int)D) / sizeof(float));
    if (K > 0 && cfg.flow_contrastive_weight > 0.f) s.nce_flow = c.take(nce_workspace_bytes((int)B, (int)D) / sizeof(float));
    if (s.nce_enc || s.nce_flow) s.dz0c = c.take(B * D);
    if (w.enc.variational) { s.params = c.take(B * 2 * D); s.dparams = c.take(B * 2 * D); s.kl_rows = c.take(B); }
    s.dec = carve_stack(c, stack_of(w.dec), 2 * B, true);
    if (w.enc.kind == VITA_ENC_SIMPLE) s.senc = carve_stack(c, stack_of_simple_encoder(w.enc), B * w.enc.T, true);
    else s.cnn = carve_cnn(c, w.enc, B, true);
    s.bytes = c.off;
    return s;
}

struct MfCtx {
    const vita_meanflow_weights* w;
    const MfTrainWs* s;
    int B, E;
    hipStream_t st;
    size_t bh() const { return (size_t)B * w->H; }
    size_t bf() const { return (size_t)B * w->FF; }
    float* X0(int e) const { return s->X0 + (size_t)e * B * w->D; }
    float* Xs(int l, int e) const { return s->Xs + ((size_t)l * E + e) * bh(); }
    float* Xc(int l, int e) const { return s->Xc + ((size_t)l * E + e) * bh(); }
    float* U1(int l, int e) const { return s->U1 + ((size_t)l * E + e) * bf(); }
    float* S(int l, int e) const { return s->S + ((size_t)l * E + e) * bf(); }
    float* DU(int e) const { return s->DU + (size_t)e * B * w->D; }
    float* DX(int l, int e) const { return s->DX + ((size_t)l * E + e) * bh(); }
};

// trunk of evaluation e with saved activations; tc: conditioning rows (ld_tc = H per row, 0 = one shared row)
int mf_trunk_fwd(const MfCtx& c, int e, const float* tc, int64_t ld_tc) {
    const vita_meanflow_weights* w = c.w;
    const int D = w->D, H = w->H, FF = w->FF, B = c.B;
    VITA_TRY(gemm_f32(lin(c.X0(e), D, w->in_w, D, w->in_b, c.Xs(0, e), H, B, H, D, VITA_ACT_NONE), c.st));
    for (int l = 0; l < w->L; ++l) {
        VITA_TRY(launch_add_rows(c.Xs(l, e), H, tc, ld_tc, c.Xc(l, e), H, B, H, c.st));
        vita_linear_args a = lin(c.Xc(l, e), H, w->fc1_w[l], H, w->fc1_b[l], c.S(l, e), FF, B, FF, H, VITA_ACT_SILU);
        a.preact = c.U1(l, e); a.ld_preact = FF;
        VITA_TRY(gemm_f32(a, c.st));
        vita_linear_args b = lin(c.S(l, e), FF, w->fc2_w[l], FF, w->fc2_b[l], c.Xs(l + 1, e), H, B, H, FF, VITA_ACT_NONE);
        b.resid = c.Xs(l, e); b.ld_resid = H;
        VITA_TRY(gemm_f32(b, c.st));
    }
    return VITA_OK;
}
// tangent pass of evaluation 0: xdot through the same weights, no biases; returns dudt (B, D)
int mf_tangent_fwd(const MfCtx& c, const float* dz, const float* dtc) {
    const vita_meanflow_weights* w = c.w;
    const MfTrainWs& s = *c.s;
    const int D = w->D, H = w->H, FF = w->FF, B = c.B;
    float* xd = s.xd0;
    float* xn = s.xd1;
    VITA_TRY(gemm_f32(lin(dz, D, w->in_w, D, nullptr, xd, H, B, H, D, VITA_ACT_NONE), c.st));
    for (int l = 0; l < w->L; ++l) {
        VITA_TRY(launch_add_rows(xd, H, dtc, H, s.xcd, H, B, H, c.st));
        VITA_TRY(gemm_f32(lin(s.xcd, H, w->fc1_w[l], H, nullptr, s.u1d, FF, B, FF, H, VITA_ACT_NONE), c.st));
        VITA_TRY(vita_act_bwd_f32(c.U1(l, 0), s.u1d, s.sd, (int64_t)B * FF, VITA_ACT_SILU, c.st));      // sdot = silu'(u1) * u1dot
        vita_linear_args b = lin(s.sd, FF, w->fc2_w[l], FF, nullptr, xn, H, B, H, FF, VITA_ACT_NONE);
        b.resid = xd; b.ld_resid = H;
        VITA_TRY(gemm_f32(b, c.st));
        float* t_ = xd; xd = xn; xn = t_;
    }
    return gemm_f32(lin(xd, H, w->u_w, H, nullptr, s.dudt, D, B, D, H, VITA_ACT_NONE), c.st);
}
// input-gradient chain of evaluation e from DU(e); leaves DX(l, e) for every l, dU1 in U1(l, e); dx = resid + DX(0,e) . W_in
// extra: NULL, or (L, B, H) additive gradients of the layer outputs x_1..x_L of this evaluation (the dispersive term)
int mf_chain_bwd(const MfCtx& c, int e, float* dx, int64_t ld_dx, const float* resid, int64_t ld_resid, const float* dv = nullptr,
                 const float* extra = nullptr) {
    const vita_meanflow_weights* w = c.w;
    const int D = w->D, H = w->H, FF = w->FF, B = c.B, L = w->L;
    VITA_TRY(gemm_f32(dgrad(c.DU(e), D, w->u_w, H, c.DX(L, e), H, B, H, D), c.st));
    if (dv) {   // the v head's gradient joins d x_L (improved MeanFlow's auxiliary loss)
        vita_linear_args a = dgrad(dv, D, w->v_w, H, c.DX(L, e), H, B, H, D);
        a.resid = c.DX(L, e); a.ld_resid = H;
        VITA_TRY(gemm_f32(a, c.st));
    }
    for (int l = L - 1; l >= 0; --l) {
        if (extra) VITA_TRY(launch_axpy(extra + (size_t)l * c.bh(), c.DX(l + 1, e), (int64_t)B * H, c.st));   // x_{l+1} is internal_features[l]
        vita_linear_args a = dgrad(c.DX(l + 1, e), H, w->fc2_w[l], FF, c.U1(l, e), FF, B, FF, H);        // dU1 = (dY . W2) * silu'(U1), in place
        a.act = VITA_ACT_SILU; a.act_grad_z = c.U1(l, e); a.ld_act_grad_z = FF;
        VITA_TRY(gemm_f32(a, c.st));
        vita_linear_args b = dgrad(c.U1(l, e), FF, w->fc1_w[l], H, c.DX(l, e), H, B, H, FF);             // dx_l = dx_{l+1} + dU1 . W1
        b.resid = c.DX(l + 1, e); b.ld_resid = H;
        VITA_TRY(gemm_f32(b, c.st));
    }
    vita_linear_args a = dgrad(c.DX(0, e), H, w->in_w, D, dx, ld_dx, B, D, H);
    if (resid) { a.resid = resid; a.ld_resid = ld_resid; }
    return gemm_f32(a, c.st);
}
int check_mfw(const vita_meanflow_weights* w, const char* who) {
    VITA_REQUIRE(w->D > 0 && w->H > 0 && w->FF > 0 && w->L >= 0 && w->L <= VITA_MAX_LAYERS, "%s: bad mean-flow dims", who);
    VITA_REQUIRE(w->t1_w && w->t1_b && w->t2_w && w->t2_b && w->h1_w && w->h1_b && w->h2_w && w->h2_b && w->in_w && w->in_b && w->u_w && w->u_b,
                 "%s: null mean-flow weight pointer", who);
    for (int l = 0; l < w->L; ++l) VITA_REQUIRE(w->fc1_w[l] && w->fc1_b[l] && w->fc2_w[l] && w->fc2_b[l], "%s: null weight pointer in layer %d", who, l);
    return VITA_OK;
}

}  // namespace
}  // namespace vita

extern "C" {

size_t vita_compute_loss_meanflow_workspace_bytes(const vita_meanflow_policy_weights* w, const vita_meanflow_loss_config* cfg, int32_t B) {
    if (!w || !cfg || B <= 0) return 0;
    return carve_mf_train(nullptr, *w, *cfg, (size_t)B).bytes;
}

int vita_compute_loss_meanflow_f32(const vita_meanflow_policy_weights* w, const vita_meanflow_policy_weights* grads,
                                   const vita_meanflow_loss_config* cfg, const float* obs_features, int64_t ldobs, const float* actions,
                                   const float* t, const float* r, const float* enc_noise, int32_t B, float* losses,
                                   float* d_obs_features, int64_t ld_dobs, void* const* grad_ready, void* workspace,
                                   size_t workspace_bytes, vita_stream_t stream) {
    const char* who = "vita_compute_loss_meanflow_f32";
    VITA_REQUIRE(d_obs_features == nullptr || (grads != nullptr && w != nullptr && ld_dobs >= w->O), "%s: d_obs_features needs grads and ld_dobs >= O", who);
    VITA_REQUIRE(w && cfg, "%s: null weights / config", who);
    VITA_REQUIRE(w->O > 0 && w->obs_w && w->obs_b, "%s: bad obs_encoder", who);
    VITA_TRY(check_mfw(&w->flow, who));
    VITA_TRY(check_dec_weights(&w->dec, who));
    VITA_TRY(check_enc_weights(&w->enc, who));
    VITA_REQUIRE(w->enc.kind != VITA_ENC_EXTERNAL, "%s: an external action encoder is built for vita_compute_loss_f32 only", who);
    VITA_REQUIRE(!w->enc.variational || w->enc.kind == VITA_ENC_CNN, "%s: the variational encoder is the CNN one", who);
    VITA_REQUIRE(cfg->kl_weight >= 0.f, "%s: kl_weight must be >= 0", who);
    VITA_REQUIRE(w->flow.D == w->dec.D && w->flow.D == w->enc.D && w->dec.OUT == w->enc.T * w->enc.A, "%s: latent / action dims differ", who);
    VITA_REQUIRE(B >= 1 && obs_features && actions && t && r && losses && ldobs >= w->O, "%s: bad obs_features/actions/t/r/losses", who);
    VITA_REQUIRE(!cfg->decode_flow_latents || cfg->num_steps >= 1, "%s: num_steps must be >= 1", who);
    VITA_REQUIRE(cfg->consistency_weight >= 0.f && cfg->enc_recon_weight >= 0.f && cfg->flow_recon_weight >= 0.f && cfg->adaptive_loss_c > 0.f &&
                     cfg->enc_contrastive_weight >= 0.f && cfg->flow_contrastive_weight >= 0.f,
                 "%s: bad loss weights", who);
    if (grads) {
        VITA_TRY(check_mfw(&grads->flow, "vita_compute_loss_meanflow_f32 (grads)"));
        VITA_TRY(check_dec_weights(&grads->dec, who));
        VITA_TRY(check_enc_weights(&grads->enc, who));
        VITA_REQUIRE(grads->obs_w && grads->obs_b && grads->O == w->O && grads->flow.D == w->flow.D && grads->flow.H == w->flow.H &&
                         grads->flow.FF == w->flow.FF && grads->flow.L == w->flow.L && grads->dec.HD == w->dec.HD && grads->dec.NL == w->dec.NL &&
                         grads->dec.OUT == w->dec.OUT && grads->enc.kind == w->enc.kind && grads->enc.C == w->enc.C && grads->enc.NL == w->enc.NL &&
                         grads->enc.A == w->enc.A && grads->enc.T == w->enc.T,
                     "%s: the gradient descriptor's dims differ from the weights'", who);
    }
    VITA_TRY(check_ws(workspace, workspace_bytes, vita_compute_loss_meanflow_workspace_bytes(w, cfg, B), who));
    hipStream_t st = (hipStream_t)stream;
    MfTrainWs s = carve_mf_train(workspace, *w, *cfg, (size_t)B);
    const vita_meanflow_weights* fw = &w->flow;
    const int D = fw->D, H = fw->H, FF = fw->FF, L = fw->L, OUT = w->dec.OUT;
    const bool fe = cfg->freeze_encoder != 0, fd = cfg->freeze_decoder != 0;          // vita_policy.py:119, 142-143, 153, 183
    const bool do_sample = cfg->decode_flow_latents != 0 && !fe && !fd;
    const bool imf = cfg->use_imf != 0;
    VITA_REQUIRE(!imf || (fw->v_w && fw->v_b && (!grads || (grads->flow.v_w && grads->flow.v_b))), "%s: improved MeanFlow needs the v head", who);
    VITA_REQUIRE(!imf || (cfg->norm_eps > 0.f && cfg->aux_v_loss_weight >= 0.f), "%s: bad norm_eps / aux_v_loss_weight", who);
    // conditioning rows: [B per-row (t, h) | K sampler steps | imf: B rows (t, 0)]; evaluation slots: 0 = JVP primal, 1..K sampler,
    // imf: slot 1 + K = the forward-only evaluation that supplies the tangent.  Tc = rows that carry gradient, Eg likewise.
    const int K = do_sample ? cfg->num_steps : 0, Tc = B + K, T = Tc + (imf ? B : 0), Eg = 1 + K, E = Eg + (imf ? 1 : 0);
    const bool do_cons = do_sample && cfg->consistency_weight > 0.f;
    const bool dec_pred = do_sample && cfg->flow_recon_weight > 0.f;
    const bool dec_enc = cfg->enc_recon_weight > 0.f && !fd;
    const bool nce_enc = cfg->enc_contrastive_weight > 0.f;                           // vita_policy.py:134-139
    const bool nce_flow = do_sample && cfg->flow_contrastive_weight > 0.f;            // :167-172
    const float tau = cfg->temperature > 0.f ? cfg->temperature : 0.07f;
    const bool vae = w->enc.variational != 0;
    const bool do_kl = vae && !fe && cfg->kl_weight > 0.f && enc_noise != nullptr;    // :142-150; deterministic posterior: kl() is zeros
    float* pred = s.ZZ;
    float* z1 = s.ZZ + (size_t)B * D;
    MfCtx mc{fw, &s, B, E, st};
    SplitkScope skscope(s.sk, kSplitkBytes);
    static const bool no_fused = getenv("VITA_NO_FUSED_SPLITK") != nullptr && atoi(getenv("VITA_NO_FUSED_SPLITK")) != 0;
    if (!no_fused) VITA_REQUIRE(skscope.enable_fused(st) == 0, "%s: could not zero the split-K counters", who);   // as in vita_compute_loss_f32
    WtScope wtscope;
    if (s.wt && grads) VITA_TRY(stage_wt(wtscope, s.wt, nullptr, &w->flow, w->dec, w->enc, st));
    const float seeds[6] = {1.0f, cfg->consistency_weight, cfg->flow_recon_weight, cfg->enc_recon_weight, cfg->enc_contrastive_weight,
                            cfg->flow_contrastive_weight};
    VITA_TRY(launch_set_scalars(s.consts, seeds, 6, st));
    VITA_CHECK_HIP(hipMemsetAsync(losses, 0, sizeof(float) * VITA_MF_LOSS_COUNT, st));

    // ---- forward ------------------------------------------------------------------------------
    VITA_TRY(gemm_f32(lin(obs_features, ldobs, w->obs_w, w->O, w->obs_b, s.z0, D, B, D, w->O, VITA_ACT_NONE), st));
    {
        StepWs es{};
        es.senc = s.senc; es.cnn = s.cnn;
        if (vae) {                                                                                                    // vita_policy.py:121
            VITA_TRY(encoder_fwd(w->enc, es, actions, s.params, 2 * D, B, st));
            VITA_TRY(launch_gauss_sample(s.params, 2 * D, enc_noise, z1, D, B, D, st));
            if (do_kl) VITA_TRY(vita_gaussian_kl_fwd_f32(s.params, 2 * D, s.params + D, 2 * D, B, D, s.kl_rows, losses + VITA_MF_LOSS_ACTION_KL, st));
        } else {
            VITA_TRY(encoder_fwd(w->enc, es, actions, z1, D, B, st));
        }
    }
    hipLaunchKernelGGL(mf_prep_kernel, dim3((T + MFT - 1) / MFT), dim3(MFT), 0, st, t, r, s.tv, s.hv, s.ones, B, K, imf ? 1 : 0);
    VITA_CHECK_HIP(hipGetLastError());
    // time conditioning for the T rows, keeping the pre-activations              simple_mean_flow_net.py:64-67
    {
        vita_linear_args a = lin(s.tv, 1, fw->t1_w, 1, fw->t1_b, s.St, H, T, H, 1, VITA_ACT_SILU);
        a.preact = s.At; a.ld_preact = H;
        VITA_TRY(gemm_f32(a, st));
        VITA_TRY(gemm_f32(lin(s.St, H, fw->t2_w, H, fw->t2_b, s.tc, H, T, H, H, VITA_ACT_NONE), st));
        vita_linear_args b = lin(s.hv, 1, fw->h1_w, 1, fw->h1_b, s.Sh, H, T, H, 1, VITA_ACT_SILU);
        b.preact = s.Ah; b.ld_preact = H;
        VITA_TRY(gemm_f32(b, st));
        vita_linear_args d = lin(s.Sh, H, fw->h2_w, H, fw->h2_b, s.tc, H, T, H, H, VITA_ACT_NONE);
        d.resid = s.tc; d.ld_resid = H;
        VITA_TRY(gemm_f32(d, st));
    }
    // its tangent along (tdot, rdot) = (1, 0): tdot = 1 and hdot = 1
    {
        VITA_TRY(gemm_f32(lin(s.ones, 1, fw->t1_w, 1, nullptr, s.dAt, H, B, H, 1, VITA_ACT_NONE), st));
        VITA_TRY(vita_act_bwd_f32(s.At, s.dAt, s.dSt, (int64_t)B * H, VITA_ACT_SILU, st));
        VITA_TRY(gemm_f32(lin(s.dSt, H, fw->t2_w, H, nullptr, s.dtc_tan, H, B, H, H, VITA_ACT_NONE), st));
        VITA_TRY(gemm_f32(lin(s.ones, 1, fw->h1_w, 1, nullptr, s.dAt, H, B, H, 1, VITA_ACT_NONE), st));
        VITA_TRY(vita_act_bwd_f32(s.Ah, s.dAt, s.dSt, (int64_t)B * H, VITA_ACT_SILU, st));
        vita_linear_args d = lin(s.dSt, H, fw->h2_w, H, nullptr, s.dtc_tan, H, B, H, H, VITA_ACT_NONE);
        d.resid = s.dtc_tan; d.ld_resid = H;
        VITA_TRY(gemm_f32(d, st));
    }
    // z_t = (1-t)*x1 + t*x0, v = x0 - x1: the CFM interpolation with the roles of the end points swapped
    VITA_TRY(vita_cfm_interpolate_f32(z1, s.z0, t, nullptr, nullptr, mc.X0(0), s.v, B, D, st));
    VITA_TRY(mf_trunk_fwd(mc, 0, s.tc, H));
    VITA_TRY(gemm_f32(lin(mc.Xs(L, 0), H, fw->u_w, H, fw->u_b, s.u0, D, B, D, H, VITA_ACT_NONE), st));
    if (imf) {
        // tangent = the v head at (z_t, t, r = t), no grad (:138-140): forward-only slot E - 1 with the (t, 0) conditioning rows
        VITA_CHECK_HIP(hipMemcpyAsync(mc.X0(E - 1), mc.X0(0), (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
        VITA_TRY(mf_trunk_fwd(mc, E - 1, s.tc + (size_t)Tc * H, H));
        VITA_TRY(gemm_f32(lin(mc.Xs(L, E - 1), H, fw->v_w, H, fw->v_b, s.vnet, D, B, D, H, VITA_ACT_NONE), st));
        VITA_TRY(gemm_f32(lin(mc.Xs(L, 0), H, fw->v_w, H, fw->v_b, s.vpred, D, B, D, H, VITA_ACT_NONE), st));
        VITA_TRY(mf_tangent_fwd(mc, s.vnet, s.dtc_tan));
        hipLaunchKernelGGL(imf_loss_rows_kernel, dim3((B + 3) / 4), dim3(MFT), 0, st, s.u0, s.v, s.dudt, s.vpred, s.hv, B, D, cfg->norm_p, cfg->norm_eps,
                           s.rows, s.rows2, s.mrow1, s.mrow2, s.wrow, s.wrow2);
        hipLaunchKernelGGL(mf_mean_kernel, dim3(1), dim3(MFT), 0, st, s.rows, B, losses + VITA_MF_LOSS_IMF);
        hipLaunchKernelGGL(mf_mean_kernel, dim3(1), dim3(MFT), 0, st, s.rows2, B, losses + VITA_MF_LOSS_AUX_V);
        hipLaunchKernelGGL(mf_mean_kernel, dim3(1), dim3(MFT), 0, st, s.mrow1, B, losses + VITA_MF_LOSS_IMF_MSE);
        hipLaunchKernelGGL(mf_mean_kernel, dim3(1), dim3(MFT), 0, st, s.mrow2, B, losses + VITA_MF_LOSS_AUX_V_MSE);
        VITA_CHECK_HIP(hipGetLastError());
        const float fsum[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 1.0f, cfg->aux_v_loss_weight};
        VITA_TRY(launch_weighted_sum(losses, fsum, 7, losses + VITA_MF_LOSS_FLOW, st));     // :157
    } else {
        VITA_TRY(mf_tangent_fwd(mc, s.v, s.dtc_tan));
        const float pexp = 1.0f - cfg->adaptive_loss_gamma;
        hipLaunchKernelGGL(mf_loss_rows_kernel, dim3((B + 3) / 4), dim3(MFT), 0, st, s.u0, s.v, s.dudt, s.hv, B, D, pexp, cfg->adaptive_loss_c, s.rows, s.wrow);
        hipLaunchKernelGGL(mf_mean_kernel, dim3(1), dim3(MFT), 0, st, s.rows, B, losses + VITA_MF_LOSS_FLOW);
        VITA_CHECK_HIP(hipGetLastError());
    }
    // dispersive term on the hidden layers' outputs at (z_t, t, r) — evaluation 0's own activations                 :185-195
    const bool disp = cfg->dispersive_loss_weight > 0.f && L > 0 && B > 1;
    VITA_REQUIRE(!(cfg->dispersive_loss_weight > 0.f) || cfg->dispersive_loss_tau > 0.f, "%s: dispersive_loss_tau must be > 0", who);
    if (disp) {
        for (int l = 0; l < L; ++l) {
            const float* z = mc.Xs(l + 1, 0);
            VITA_TRY(gemm_f32(lin(z, H, z, H, nullptr, s.dispG, B, B, B, H, VITA_ACT_NONE), st));
            hipLaunchKernelGGL(disp_dist_kernel, dim3((B + 3) / 4), dim3(MFT), 0, st, s.dispG, B, s.dispRow);
            hipLaunchKernelGGL(disp_max_kernel, dim3(1), dim3(MFT), 0, st, s.dispRow, B, s.dispScal);
            hipLaunchKernelGGL(disp_exp_kernel, dim3((B + 3) / 4), dim3(MFT), 0, st, s.dispG, B, s.dispScal, cfg->dispersive_loss_tau, s.dispSum);
            hipLaunchKernelGGL(disp_log_kernel, dim3(1), dim3(MFT), 0, st, s.dispSum, B, s.dispScal, losses + VITA_MF_LOSS_DISPERSIVE);
            VITA_CHECK_HIP(hipGetLastError());
            if (grads) {
                VITA_TRY(gemm_f32(dgrad(s.dispG, B, z, H, s.dispEZ, H, B, H, B), st));      // E . z
                hipLaunchKernelGGL(disp_grad_kernel, dim3(mfgrid((int64_t)B * H)), dim3(MFT), 0, st, z, s.dispEZ, s.dispSum, s.dispScal,
                                   cfg->dispersive_loss_weight, cfg->dispersive_loss_tau, B, H, s.dispD + (size_t)l * B * H);
                VITA_CHECK_HIP(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(mf_addw_kernel, dim3(1), dim3(64), 0, st, losses, (int)VITA_MF_LOSS_FLOW, (int)VITA_MF_LOSS_DISPERSIVE, cfg->dispersive_loss_weight,
                           (int)VITA_MF_LOSS_FLOW_BASE);
        VITA_CHECK_HIP(hipGetLastError());
    }
    // MeanFlow sampler with grad                                                  mean_flow_matcher.py:201-233
    auto hstep = [&](int k) { return (float)((1.0 - (double)k / (double)K) - (1.0 - (double)(k + 1) / (double)K)); };
    if (do_sample) {
        VITA_CHECK_HIP(hipMemcpyAsync(mc.X0(1), s.z0, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
        for (int k = 0; k < K; ++k) {
            const int e = k + 1;
            VITA_TRY(mf_trunk_fwd(mc, e, s.tc + (size_t)(B + k) * H, 0));
            float* dst = (k == K - 1) ? pred : mc.X0(e + 1);
            vita_linear_args o = lin(mc.Xs(L, e), H, fw->u_w, H, fw->u_b, dst, D, B, D, H, VITA_ACT_NONE);
            o.use_gate_scalar = 1; o.gate_scalar = -hstep(k); o.resid = mc.X0(e); o.ld_resid = D;
            VITA_TRY(gemm_f32(o, st));
        }
        if (do_cons) VITA_TRY(vita_mse_loss_fwd_f32(pred, z1, (int64_t)B * D, s.loss_part, losses + VITA_LOSS_CONSISTENCY, st));
        if (nce_flow) VITA_TRY(nce_forward(s.z0, D, pred, D, B, D, tau, losses + VITA_MF_LOSS_FLOW_CONTRASTIVE, s.nce_flow, st));
    }
    if (nce_enc) VITA_TRY(nce_forward(s.z0, D, z1, D, B, D, tau, losses + VITA_MF_LOSS_ENC_CONTRASTIVE, s.nce_enc, st));
    const Stack dec = stack_of(w->dec);
    const int r0 = dec_pred ? 0 : B;
    const int M = (dec_pred ? B : 0) + (dec_enc ? B : 0);
    const int enc_row = dec_pred ? B : 0;
    const int64_t n_rec = (int64_t)B * OUT;
    if (M > 0) {
        VITA_TRY(stack_fwd(dec, s.dec, s.ZZ + (size_t)r0 * D, D, M, nullptr, st));
        auto recon = cfg->recon_l1 ? vita_l1_loss_fwd_f32 : vita_mse_loss_fwd_f32;
        if (dec_pred) VITA_TRY(recon(s.dec.out, actions, n_rec, s.loss_part, losses + VITA_LOSS_FLOW_RECON, st));
        if (dec_enc) VITA_TRY(recon(s.dec.out + (size_t)enc_row * OUT, actions, n_rec, s.loss_part, losses + VITA_LOSS_ENC_RECON, st));
    }
    const float wsum[5] = {0.f, 1.0f, do_cons ? cfg->consistency_weight : 0.f, dec_pred ? cfg->flow_recon_weight : 0.f,
                           dec_enc ? cfg->enc_recon_weight : 0.f};
    VITA_TRY(launch_weighted_sum(losses, wsum, 5, losses + VITA_LOSS_TOTAL, st));
    if (nce_enc)
        hipLaunchKernelGGL(mf_addw_kernel, dim3(1), dim3(64), 0, st, losses, (int)VITA_MF_LOSS_TOTAL, (int)VITA_MF_LOSS_ENC_CONTRASTIVE,
                           cfg->enc_contrastive_weight, -1);
    if (nce_flow)
        hipLaunchKernelGGL(mf_addw_kernel, dim3(1), dim3(64), 0, st, losses, (int)VITA_MF_LOSS_TOTAL, (int)VITA_MF_LOSS_FLOW_CONTRASTIVE,
                           cfg->flow_contrastive_weight, -1);
    if (do_kl)
        hipLaunchKernelGGL(mf_addw_kernel, dim3(1), dim3(64), 0, st, losses, (int)VITA_MF_LOSS_TOTAL, (int)VITA_MF_LOSS_ACTION_KL, cfg->kl_weight, -1);
    VITA_CHECK_HIP(hipGetLastError());
    if (!grads) return VITA_OK;

    // ---- backward -----------------------------------------------------------------------------
    const vita_meanflow_policy_weights& g = *grads;
    const vita_meanflow_weights& gf = g.flow;
    if (!dec_pred) VITA_CHECK_HIP(hipMemsetAsync(s.dZZ, 0, (size_t)B * D * 4, st));
    if (!dec_enc) VITA_CHECK_HIP(hipMemsetAsync(s.dZZ + (size_t)B * D, 0, (size_t)B * D * 4, st));
    if (M > 0) {
        auto recon_bwd = cfg->recon_l1 ? vita_l1_loss_bwd_f32 : vita_mse_loss_bwd_f32;
        if (dec_pred) VITA_TRY(recon_bwd(s.dec.out, actions, n_rec, s.consts + 2, s.dec.dOut, nullptr, 0, st));
        if (dec_enc)
            VITA_TRY(recon_bwd(s.dec.out + (size_t)enc_row * OUT, actions, n_rec, s.consts + 3, s.dec.dOut + (size_t)enc_row * OUT, nullptr, 0, st));
        VITA_TRY(stack_bwd(dec, stack_of(g.dec), s.dec, s.ZZ + (size_t)r0 * D, D, M, nullptr, s.dec.dOut, OUT, s.dZZ + (size_t)r0 * D, D,
                           s.cs_part, st));
    } else {
        const Stack gd = stack_of(g.dec);
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.in_w), 0, (size_t)gd.TR * gd.IN * 4, st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.in_b), 0, (size_t)gd.TR * 4, st));
        for (int i = 0; i < gd.NL; ++i) {
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc1_w[i]), 0, (size_t)gd.FF * gd.TR * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc1_b[i]), 0, (size_t)gd.FF * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc2_w[i]), 0, (size_t)gd.TR * gd.FF * 4, st));
            VITA_CHECK_HIP(hipMemsetAsync(mut(gd.fc2_b[i]), 0, (size_t)gd.TR * 4, st));
        }
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.out_w), 0, (size_t)gd.OUT * gd.TR * 4, st));
        VITA_CHECK_HIP(hipMemsetAsync(mut(gd.out_b), 0, (size_t)gd.OUT * 4, st));
    }
    auto record = [&](int i) -> int {      // as in vita_compute_loss_f32: 0 decoder, 1 flow net, 2 everything
        if (grad_ready && grad_ready[i]) VITA_CHECK_HIP(hipEventRecord((hipEvent_t)grad_ready[i], st));
        return VITA_OK;
    };
    VITA_TRY(record(0));
    float* dpred = s.dZZ;
    float* dz1 = s.dZZ + (size_t)B * D;
    if (do_cons) VITA_TRY(vita_mse_loss_bwd_f32(pred, z1, (int64_t)B * D, s.consts + 1, dpred, dz1, 1, st));
    // InfoNCE terms: d pred / d z1 join dZZ now; their d z0 waits in dz0c (the sampler's first step overwrites dz0)
    if (nce_flow) VITA_TRY(nce_backward(B, D, tau, s.consts + 5, s.dz0c, D, 0, dpred, D, 1, s.nce_flow, st));
    if (nce_enc) VITA_TRY(nce_backward(B, D, tau, s.consts + 4, s.dz0c, D, nce_flow ? 1 : 0, dz1, D, 1, s.nce_enc, st));
    // MeanFlow loss -> DU(0); primal chain of evaluation 0 -> d z_t
    if (imf)
        hipLaunchKernelGGL(imf_loss_bwd_kernel, dim3(mfgrid((int64_t)B * D)), dim3(MFT), 0, st, s.u0, s.v, s.dudt, s.vpred, s.hv, s.wrow, s.wrow2, B, D,
                           cfg->aux_v_loss_weight, mc.DU(0), s.DV);
    else
        hipLaunchKernelGGL(mf_loss_bwd_kernel, dim3(mfgrid((int64_t)B * D)), dim3(MFT), 0, st, s.u0, s.v, s.dudt, s.hv, s.wrow, B, D, mc.DU(0));
    VITA_CHECK_HIP(hipGetLastError());
    VITA_TRY(mf_chain_bwd(mc, 0, s.dzt, D, nullptr, 0, imf ? s.DV : nullptr, disp ? s.dispD : nullptr));
    // conditioning gradients, eval 0: per row, dtc[b] = DX(0,0)[b] - DX(L,0)[b]  (sum over layers of d x_cond telescopes)
    hipLaunchKernelGGL(mf_sub_kernel, dim3(mfgrid((int64_t)B * H)), dim3(MFT), 0, st, mc.DX(0, 0), mc.DX(L, 0), s.dtc, (int64_t)B * H);
    // the gradients injected at x_1..x_{L-1} enter d x_l but are not part of any d x_cond: take them out of the telescoped sum
    for (int l = 0; disp && l + 1 < L; ++l)
        hipLaunchKernelGGL(mf_sub_inplace_kernel, dim3(mfgrid((int64_t)B * H)), dim3(MFT), 0, st, s.dtc, s.dispD + (size_t)l * B * H, (int64_t)B * H);
    VITA_CHECK_HIP(hipGetLastError());
    if (do_sample) {
        const float* dxn = dpred;
        for (int k = K - 1; k >= 0; --k) {
            const int e = k + 1;
            VITA_TRY(vita_gate_bwd_f32(dxn, nullptr, nullptr, 0, -hstep(k), mc.DU(e), nullptr, 0, B, D, st));
            float* dst = (k == 0) ? s.dz0 : ((k & 1) ? s.dxa : s.dxb);
            VITA_TRY(mf_chain_bwd(mc, e, dst, D, dxn, D));
            hipLaunchKernelGGL(mf_sub_kernel, dim3(mfgrid((int64_t)B * H)), dim3(MFT), 0, st, mc.DX(0, e), mc.DX(L, e), s.dtmp, (int64_t)B * H);
            VITA_CHECK_HIP(hipGetLastError());
            VITA_TRY(vita_colsum_f32(s.dtmp, H, B, H, s.cs_part, s.dtc + (size_t)(B + k) * H, 0, st));
            dxn = dst;
        }
    } else {
        VITA_CHECK_HIP(hipMemsetAsync(s.dz0, 0, (size_t)B * D * 4, st));
    }
    // z_t = t*x0 + (1-t)*x1 (v is detached): d x1 += (1-t) d z_t, d x0 += t d z_t
    VITA_TRY(vita_cfm_interpolate_bwd_f32(s.dzt, nullptr, t, nullptr, nullptr, dz1, s.dz0, B, D, 1, st));
    if (nce_enc || nce_flow) VITA_TRY(launch_axpy(s.dz0c, s.dz0, (int64_t)B * D, st));
    // weight gradients over the stacked E*B rows.  With cfg->grad_ready_layers the sections are ordered for a data-parallel caller's
    // link as in vita_compute_loss_f32: the hidden layers first (one event each: the bulk of the gradient bytes), then the time /
    // interval embeddings (event 1), the input layer and the heads (event 3 + L); the encoders follow (event 2)
    const bool per_layer = cfg->grad_ready_layers != 0 && grad_ready != nullptr;
    const int R = Eg * B;      // the forward-only slot of improved MeanFlow carries no gradient
    auto outer_wgrads = [&]() -> int {
        VITA_TRY(gemm_f32(wgrad(s.DU, D, mc.Xs(L, 0), H, mut(gf.u_w), H, R, H, D), st));
        VITA_TRY(vita_colsum_f32(s.DU, D, R, D, s.cs_part, mut(gf.u_b), 0, st));
        VITA_TRY(gemm_f32(wgrad(mc.DX(0, 0), H, s.X0, D, mut(gf.in_w), D, R, D, H), st));
        VITA_TRY(vita_colsum_f32(mc.DX(0, 0), H, R, H, s.cs_part, mut(gf.in_b), 0, st));
        if (imf) {      // the v head is used by evaluation 0 only
            VITA_TRY(gemm_f32(wgrad(s.DV, D, mc.Xs(L, 0), H, mut(gf.v_w), H, B, H, D), st));
            VITA_TRY(vita_colsum_f32(s.DV, D, B, D, s.cs_part, mut(gf.v_b), 0, st));
        } else {        // the v head takes no part (use_imf = False)
            if (gf.v_w) VITA_CHECK_HIP(hipMemsetAsync(mut(gf.v_w), 0, (size_t)D * H * 4, st));
            if (gf.v_b) VITA_CHECK_HIP(hipMemsetAsync(mut(gf.v_b), 0, (size_t)D * 4, st));
        }
        return VITA_OK;
    };
    if (!per_layer) VITA_TRY(outer_wgrads());
    for (int l = 0; l < L; ++l) {
        VITA_TRY(gemm_f32_wgrad_pair(wgrad(mc.DX(l + 1, 0), H, mc.S(l, 0), FF, mut(gf.fc2_w[l]), FF, R, FF, H),
                                     wgrad(mc.U1(l, 0), FF, mc.Xc(l, 0), H, mut(gf.fc1_w[l]), H, R, H, FF), st));
        VITA_TRY(vita_colsum_f32(mc.DX(l + 1, 0), H, R, H, s.cs_part, mut(gf.fc2_b[l]), 0, st));
        VITA_TRY(vita_colsum_f32(mc.U1(l, 0), FF, R, FF, s.cs_part, mut(gf.fc1_b[l]), 0, st));
        if (per_layer) VITA_TRY(record(3 + l));
    }
    // conditioning path: tc = t2(silu(t1 t)) + h2(silu(h1 h)) over the T rows
    {
        struct Br { const float *in, *A, *Sx, *w1, *w2; const float *gw1, *gb1, *gw2, *gb2; };
        const Br brs[2] = {{s.tv, s.At, s.St, fw->t1_w, fw->t2_w, gf.t1_w, gf.t1_b, gf.t2_w, gf.t2_b},
                           {s.hv, s.Ah, s.Sh, fw->h1_w, fw->h2_w, gf.h1_w, gf.h1_b, gf.h2_w, gf.h2_b}};
        for (const Br& b : brs) {
            VITA_TRY(gemm_f32(wgrad(s.dtc, H, b.Sx, H, mut(b.gw2), H, Tc, H, H), st));
            VITA_TRY(vita_colsum_f32(s.dtc, H, Tc, H, s.cs_part, mut(b.gb2), 0, st));
            vita_linear_args a = dgrad(s.dtc, H, b.w2, H, s.dA1, H, Tc, H, H);
            a.act = VITA_ACT_SILU; a.act_grad_z = b.A; a.ld_act_grad_z = H;
            VITA_TRY(gemm_f32(a, st));
            VITA_TRY(gemm_f32(wgrad(s.dA1, H, b.in, 1, mut(b.gw1), 1, Tc, 1, H), st));
            VITA_TRY(vita_colsum_f32(s.dA1, H, Tc, H, s.cs_part, mut(b.gb1), 0, st));
        }
    }
    VITA_TRY(record(1));
    if (per_layer) {
        VITA_TRY(outer_wgrads());
        VITA_TRY(record(3 + L));
    }
    // encoders
    {
        StepWs es{};
        es.senc = s.senc; es.cnn = s.cnn;
        if (fe)         // :119 the action latents were computed without grad
            VITA_TRY(zero_encoder_grads(g.enc, st));
        else if (w->enc.kind == VITA_ENC_SIMPLE)
            VITA_TRY(stack_bwd(stack_of_simple_encoder(w->enc), stack_of_simple_encoder(g.enc), s.senc, actions, w->enc.A, B * w->enc.T, z1, dz1,
                               w->enc.C, nullptr, 0, s.cs_part, st));
        else if (vae) {
            VITA_TRY(launch_gauss_sample_bwd(s.params, 2 * D, enc_noise, dz1, D, do_kl ? cfg->kl_weight / (float)B : 0.f, s.dparams, 2 * D, B, D, st));
            VITA_TRY(cnn_bwd(w->enc, g.enc, s.cnn, actions, s.dparams, 2 * D, B, s.cs_part, st));
        } else
            VITA_TRY(cnn_bwd(w->enc, g.enc, s.cnn, actions, dz1, D, B, s.cs_part, st));
    }
    VITA_TRY(gemm_f32(wgrad(s.dz0, D, obs_features, ldobs, mut(g.obs_w), w->O, B, w->O, D), st));
    VITA_TRY(vita_colsum_f32(s.dz0, D, B, D, s.cs_part, mut(g.obs_b), 0, st));
    // what flows on into a trainable observer (vita_policy.py:113-114): d obs_features = d z0 . obs_w
    if (d_obs_features) VITA_TRY(gemm_f32(dgrad(s.dz0, D, w->obs_w, w->O, d_obs_features, ld_dobs, B, w->O, D), st));
    VITA_TRY(record(2));
    return VITA_OK;
}

}  // extern "C"
