/* vita_hip.h — C ABI of libvita_hip.so: the MI355X (gfx950) kernels behind the VITA
 * vision-latent -> action-latent flow-matching hot path.
 *
 * The reference (ucd-dare/VITA, /root/reference) is pure PyTorch and has no FFI of its own;
 * every entry point below names the reference function whose arithmetic it replaces
 * (file:line under /root/reference/flare).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - all tensor arguments are DEVICE pointers to fp32, row-major; `ld*` are row strides in
 *     elements; weights keep the PyTorch nn.Linear layout (out_features, in_features), i.e.
 *     the checkpoint tensors are passed as they are;
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only enqueues work on
 *     it: no allocation, no synchronisation, no host<->device copy inside;
 *   - scratch memory comes from the caller: ask vita_*_workspace_bytes(), pass a device
 *     buffer at least that large and 256-byte aligned;
 *   - return value 0 = ok, <0 = error (VITA_ERR_*); vita_last_error() gives the message of the
 *     calling thread's last failure.  Shapes are validated on the host before any launch.
 */
#ifndef VITA_HIP_H
#define VITA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITA_ABI_VERSION 2
#define VITA_MAX_LAYERS 16

typedef void* vita_stream_t; /* hipStream_t */

enum { VITA_OK = 0, VITA_ERR_INVALID = -1, VITA_ERR_HIP = -2, VITA_ERR_WORKSPACE = -3 };

/* activations fused into GEMM epilogues */
enum {
    VITA_ACT_NONE = 0,
    VITA_ACT_RELU = 1,      /* CNNActionEncoder, action_ae.py:306 */
    VITA_ACT_GELU_ERF = 2,  /* timm Mlp default act in SimpleActionDecoder/Encoder, action_ae.py:139-146,245-252 */
    VITA_ACT_GELU_TANH = 3, /* FlowNetLayer approx_gelu, simple_flow_net.py:16 */
    VITA_ACT_SILU = 4,      /* FlowNetLayer.time_modulator[0], simple_flow_net.py:28 */
    VITA_ACT_MISH = 5       /* SimpleFlowNet.time_embed[2], simple_flow_net.py:69 */
};

/* A-operand prologues (applied while the activation tile is staged into LDS) */
enum {
    VITA_PRO_NONE = 0,
    VITA_PRO_LN_MOD = 1,   /* LN(no affine) then x*(1+scale)+shift — simple_flow_net.py:44-45 */
    VITA_PRO_LN_AFFINE = 2 /* LN then x*weight+bias — SimpleFlowNet.norm, simple_flow_net.py:81,106 */
};

int vita_abi_version(void);
const char* vita_last_error(void);

/* ---------------------------------------------------------------------------------------
 * vita_linear_f32 — one fused dense layer on the fp32 MFMA path:
 *     A'  = prologue(X)                                  (M,K)
 *     Z   = act(A' · Wᵀ + bias)                          (M,N)
 *     Y   = resid + Z * gate        when gate/gate_scalar/resid are given
 * Replaces every nn.Linear (+ adjacent elementwise ops) on the path: obs_encoder
 * (vita_policy.py:77,114), SimpleFlowNet.input_proj/out_proj and timm Mlp fc1/fc2
 * (simple_flow_net.py:19-24,65,82,100-107), the time path (simple_flow_net.py:66-71,27-30),
 * SimpleActionDecoder (action_ae.py:274-277) and CNNActionEncoder.latent_proj (:325).
 * gate: per-column vector (ld_gate = 0, the sampling case) or per-row matrix (ld_gate > 0,
 * per-row t in training): `x + mlp(x_norm).mul_(gamma)` simple_flow_net.py:46.  With
 * use_gate_scalar the gate is the scalar `gate_scalar`: `x + vt*dt` flow_matchers.py:71.
 * Y may alias resid.  Y must not alias X or W.
 * With a_col / b_col the same kernel family serves the backward passes of every nn.Linear on the
 * path (autograd of F.linear): dX = dY.W (b_col), dW += dY^T.X (a_col and b_col, resid = Y = dW).
 * --------------------------------------------------------------------------------------- */
typedef struct vita_linear_args {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;          /* (N) or NULL */
    float* Y; int64_t ldy;
    int32_t M, N, K;
    int32_t act;                /* VITA_ACT_* */
    int32_t pro;                /* VITA_PRO_* ; LN runs over the whole K extent of a row */
    float ln_eps;
    const float* pro_mul; int64_t ld_pro_mul; /* (K) if ld==0 else (M,K): scale (LN_MOD) or weight (LN_AFFINE) */
    const float* pro_add; int64_t ld_pro_add; /* shift / bias */
    const float* gate; int64_t ld_gate;       /* NULL = none */
    int32_t use_gate_scalar; float gate_scalar;
    const float* resid; int64_t ld_resid;     /* NULL = none */
    /* training-time extras (all zero/NULL for inference) */
    float* preact; int64_t ld_preact;         /* optional (M,N): receives A'.W^T + bias BEFORE act/gate (saved for backward) */
    int32_t a_col;  /* 1: X is stored (K,M) row-major, i.e. the GEMM contracts over X's ROWS (wgrad: dW = dY^T . X) */
    int32_t b_col;  /* 1: W is stored (K,N) row-major, i.e. contracts over W's ROWS (dgrad: dX = dY . W)          */
    /* with a_col/b_col only: the result is multiplied by act'(act_grad_z[m,n]) — the backward of the activation that
     * followed the previous layer, fused into this layer's dgrad (`act` names the activation).  Y may alias act_grad_z. */
    const float* act_grad_z; int64_t ld_act_grad_z;
    /* optional scratch for split-K (NULL = never split).  With few output tiles and a long reduction (M <= ~512 rows:
     * serving batches, per-GPU training batches) the k range is cut across workgroups; the partial tiles are summed in a
     * fixed order by a second kernel that applies the epilogue, so results stay bitwise reproducible for a given shape.
     * vita_splitk_workspace_bytes() is always enough; 256-byte aligned. */
    void* splitk_ws; size_t splitk_ws_bytes;
} vita_linear_args;

int vita_linear_f32(const vita_linear_args* args, vita_stream_t stream);
size_t vita_splitk_workspace_bytes(void);
/* diagnostic: the tile configuration "<BM,BN,BK,NS,KG> xcd RxC" that the calling thread's last LDS-DMA GEMM launch
 * dispatched ("" if the last call took another kernel family); bench.py labels its roofline kernel with it */
const char* vita_debug_last_gemm_config(void);

/* ---------------------------------------------------------------------------------------
 * Weight descriptors: device pointers to the checkpoint tensors, unchanged.
 * --------------------------------------------------------------------------------------- */
typedef struct vita_flownet_weights { /* SimpleFlowNet, simple_flow_net.py:52-84 */
    int32_t D;   /* input_dim = output_dim (latent) */
    int32_t H;   /* hidden_dim */
    int32_t FF;  /* int(H*mlp_ratio) */
    int32_t L;   /* num_layers <= VITA_MAX_LAYERS */
    int32_t TE;  /* time_embed_dim (256) */
    const float *in_w, *in_b;     /* flow_net.input_proj   (H,D),(H) */
    const float *te1_w, *te1_b;   /* flow_net.time_embed.1 (4TE,TE) */
    const float *te3_w, *te3_b;   /* flow_net.time_embed.3 (H,4TE) */
    const float *fc1_w[VITA_MAX_LAYERS], *fc1_b[VITA_MAX_LAYERS]; /* layers.l.mlp.fc1 (FF,H) */
    const float *fc2_w[VITA_MAX_LAYERS], *fc2_b[VITA_MAX_LAYERS]; /* layers.l.mlp.fc2 (H,FF) */
    const float *mod_w[VITA_MAX_LAYERS], *mod_b[VITA_MAX_LAYERS]; /* layers.l.time_modulator.1 (3H,H) */
    const float *norm_w, *norm_b; /* flow_net.norm (H) */
    const float *out_w, *out_b;   /* flow_net.out_proj (D,H) */
} vita_flownet_weights;

typedef struct vita_decoder_weights { /* SimpleActionDecoder, action_ae.py:226-279 */
    int32_t D;    /* latent_dim */
    int32_t HD;   /* dec_hidden_dim */
    int32_t NL;   /* num_layers <= VITA_MAX_LAYERS */
    int32_t OUT;  /* pred_horizon * action_dim */
    const float *in_w, *in_b;   /* action_decoder.input_proj (HD,D) */
    const float *fc1_w[VITA_MAX_LAYERS], *fc1_b[VITA_MAX_LAYERS]; /* layers.i.fc1 (HD,HD) */
    const float *fc2_w[VITA_MAX_LAYERS], *fc2_b[VITA_MAX_LAYERS]; /* layers.i.fc2 (HD,HD) */
    const float *out_w, *out_b; /* action_decoder.output_proj (OUT,HD) */
} vita_decoder_weights;

/* ---------------------------------------------------------------------------------------
 * vita_flownet_forward_f32 — SimpleFlowNet.forward(x, t) with a per-row timestep
 * (simple_flow_net.py:99-108; the training-time call at flow_matchers.py:35).
 *   x (B,D), t (B) -> v (B,D).
 * --------------------------------------------------------------------------------------- */
size_t vita_flownet_forward_workspace_bytes(const vita_flownet_weights* w, int32_t B);
int vita_flownet_forward_f32(const vita_flownet_weights* w, const float* x, int64_t ldx,
                             const float* t, float* v, int64_t ldv, int32_t B,
                             void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * vita_ode_sample_f32 — TorchFlowMatcher.sample with `start` given (flow_matchers.py:55-79):
 * num_steps explicit Euler steps x <- x + v(x, k/K) * (1/K) over SimpleFlowNet, all enqueued by
 * this one call.  The time path (sinusoid, time_embed, per-layer modulators) depends only on
 * k, so it is evaluated once per call for all K steps as a (K, L, 3H) table.
 *   x0 (B,D) -> x_out (B,D);  traj: NULL or (num_steps+1, B, D) receiving x_0..x_K
 *   (return_traces=True, flow_matchers.py:63-66,73-75).  x_out may alias x0.
 * --------------------------------------------------------------------------------------- */
size_t vita_ode_sample_workspace_bytes(const vita_flownet_weights* w, int32_t B, int32_t num_steps);
int vita_ode_sample_f32(const vita_flownet_weights* w, const float* x0, int64_t ldx0,
                        float* x_out, int64_t ldo, int32_t B, int32_t num_steps, float* traj,
                        void* workspace, size_t workspace_bytes, vita_stream_t stream);
/* Diagnostic, SYNCHRONISES (a blocking 4-byte copy; call it after the stream has been synchronised): at chip-filling
 * batches vita_ode_sample_f32 runs all K steps as one persistent launch whose workgroups hand activations to each other
 * through bounded waits.  *status = 0: the last solve on this workspace completed; != 0: a wait ran out in phase
 * (*status - 1) and x_out is NOT valid (cannot happen while the launch's <= 256 workgroups are resident; the bound only
 * guarantees that the kernel always ends).  VITA_NO_PERSISTENT=1 in the environment selects the per-launch path. */
int vita_ode_sample_status(const vita_flownet_weights* w, int32_t B, int32_t num_steps, const void* workspace, int32_t* status);
/* A timed-out solve cannot be missed (round 3): behind every in-launch sampler the library enqueues a one-pass kernel that reads
 * the same status word and, when it is set, overwrites x_out / traj / the fused decoder's actions with NaN -- so the failure travels
 * with the data and needs no host synchronisation to be seen; the decoder that runs behind a poisoned solve yields NaN actions.
 * VitaPolicy.select_action / validate also read the word and raise (vita_amd/policy.py). */
/* the same word for the workspace of vita_generate_actions_f32 (its solve runs in one launch at serving batches, B <= 64, and at
 * chip-filling ones): != 0 means that call's actions are NOT valid */
int vita_generate_actions_status(const vita_flownet_weights* fw, int32_t B, int32_t num_steps, const void* workspace, int32_t* status);
/* process-wide switch between the persistent launch (1, default unless VITA_NO_PERSISTENT=1) and the per-launch path (0);
 * returns the previous setting.  Both produce the same bits. */
int vita_ode_sample_set_persistent(int32_t on);
/* 1 when vita_ode_sample_f32 on a contiguous, 16-byte aligned (B, D) state takes the persistent launch under the current switch */
int vita_ode_sample_takes_persistent(const vita_flownet_weights* w, int32_t B);
/* diagnostic: NULL (default), or a device buffer of 256 * num_steps * (2 + 2L) * 6 uint64 that the persistent launches of this
 * process fill with 100 MHz clock stamps per workgroup and phase: phase start, wait over, tiles done, published, then the shader clock at start and publish (tools/persistent_stamps.py) */
int vita_debug_ode_stamps(void* device_buf);
/* diagnostic (tests/test_gpu_timeout.py): the bound of every hand-off spin of the in-launch samplers, in polls; 0 restores the
 * default (1 << 22, seconds).  A tiny bound forces the time-out path: status != 0 and NaN outputs. */
int vita_debug_set_spin_limit(uint32_t spins);

/* ---------------------------------------------------------------------------------------
 * vita_decoder_forward_f32 — SimpleActionDecoder.forward (action_ae.py:267-279):
 *   z (B,D) -> actions (B, OUT) (the caller views it as (B,T,A)).
 * --------------------------------------------------------------------------------------- */
size_t vita_decoder_forward_workspace_bytes(const vita_decoder_weights* w, int32_t B);
int vita_decoder_forward_f32(const vita_decoder_weights* w, const float* z, int64_t ldz,
                             float* actions, int64_t lda, int32_t B,
                             void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * vita_generate_actions_f32 — VitaPolicy.generate_actions from observer features on
 * (vita_policy.py:191-208): obs_encoder (O->D), vita_ode_sample, decoder.  This is the
 * quantity BasePolicy.validate times as chunks_per_sec (base_policy.py:65-87).
 *   obs_features (B,O) -> actions (B, OUT);  latents_out: NULL or (B,D) receiving x_K.
 * --------------------------------------------------------------------------------------- */
size_t vita_generate_actions_workspace_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                                             int32_t B, int32_t num_steps);
int vita_generate_actions_f32(const float* obs_w, const float* obs_b, int32_t O,
                              const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                              const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                              float* latents_out, int32_t B, int32_t num_steps,
                              void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * vita_generate_actions_noalloc_f32 — NO COUNTERPART IN THE REFERENCE.  vita_generate_actions_f32 for callers that cannot
 * allocate or size a workspace: code under HIP graph capture, tight serving loops (select_action at eval_n_envs rows),
 * harnesses that forbid hipMalloc.  Same kernels, same arguments minus the workspace: the scratch (latents, time-path
 * tables, activations, split-K partials) lives in a library-owned `static __device__` array of
 * VITA_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.  The call allocates nothing
 * and synchronises nothing.  Shapes whose scratch need (vita_generate_actions_noalloc_need_bytes) exceeds the array are
 * rejected with VITA_ERR_WORKSPACE before any launch (at the default dims: about B <= 512 rows at num_steps <= 16).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one
 * stream are ordered.  Nothing in the array has to start at zero.  vita_noalloc_init() resolves the array's device
 * address for the CURRENT device (hipGetSymbolAddress, once per device); the entry point does the same on its first call
 * on a device, so call vita_noalloc_init() beforehand when that first call would fall inside a graph capture.
 * Its split-K region is VITA_NOALLOC_SPLITK_BYTES (not vita_splitk_workspace_bytes()), so a GEMM may be cut into fewer
 * k-ranges than under vita_generate_actions_f32: results agree to summation order (1e-6), not always to the bit.
 * --------------------------------------------------------------------------------------- */
#define VITA_NOALLOC_SCRATCH_BYTES (16u << 20)
#define VITA_NOALLOC_SPLITK_BYTES (8u << 20)
size_t vita_noalloc_scratch_bytes(void);
int vita_noalloc_init(void);
size_t vita_generate_actions_noalloc_need_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                                                int32_t B, int32_t num_steps);
int vita_generate_actions_noalloc_f32(const float* obs_w, const float* obs_b, int32_t O,
                                      const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                                      const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                                      float* latents_out, int32_t B, int32_t num_steps, vita_stream_t stream);
/* vita_generate_actions_status for the built-in array on the current device (the caller never sees its address); SYNCHRONISES
 * (a blocking 4-byte copy).  No counterpart in the reference. */
int vita_generate_actions_noalloc_status(const vita_flownet_weights* fw, int32_t B, int32_t num_steps, int32_t* status);

/* ---------------------------------------------------------------------------------------
 * Loss kernels (HBM-bound elementwise + two-stage deterministic reduction).
 * `partials` is caller scratch of vita_loss_partials_count() floats; `out` is one device float.
 * --------------------------------------------------------------------------------------- */
int32_t vita_loss_partials_count(void);
/* mean((a-b)^2): F.mse_loss (vita_policy.py:163) and torch.mean((vt-ut)**2) (flow_matchers.py:36) */
int vita_mse_loss_fwd_f32(const float* a, const float* b, int64_t n, float* partials, float* out, vita_stream_t stream);
/* grad_a = 2(a-b)/n * gout[0]; grad_b = -grad_a (either may be NULL); accumulate != 0 adds into grad_a / grad_b */
int vita_mse_loss_bwd_f32(const float* a, const float* b, int64_t n, const float* gout,
                          float* grad_a, float* grad_b, int32_t accumulate, vita_stream_t stream);
/* mean(|a-b|): F.l1_loss (vita_policy.py:67,176,185) */
int vita_l1_loss_fwd_f32(const float* a, const float* b, int64_t n, float* partials, float* out, vita_stream_t stream);
/* grad_a = sign(a-b)/n * gout[0] */
int vita_l1_loss_bwd_f32(const float* a, const float* b, int64_t n, const float* gout,
                         float* grad_a, float* grad_b, int32_t accumulate, vita_stream_t stream);
/* Fused CFM interpolation (torchcfm sample_location_and_conditional_flow, called at
 * flow_matchers.py:34, sigma = 0): xt = t*x1[j] + (1-t)*x0[i], ut = x1[j] - x0[i]; pair_i / pair_j
 * (int64, B entries each, or NULL = identity) are the minibatch-OT re-pairing of the `exact`
 * matcher (drawn with replacement, so an index may repeat).  x0,x1,xt,ut (B,D); t (B). */
int vita_cfm_interpolate_f32(const float* x0, const float* x1, const float* t, const int64_t* pair_i, const int64_t* pair_j,
                             float* xt, float* ut, int32_t B, int32_t D, vita_stream_t stream);
/* its backward: dx0[i] += (1-t)*dxt - dut, dx1[j] += t*dxt + dut (float atomics when an index
 * array is given — repeated indices make the summation order, hence the last bits, run-dependent).
 * accumulate == 0: dx0 / dx1 are overwritten (zeroed first when indexed); != 0: added into. */
int vita_cfm_interpolate_bwd_f32(const float* dxt, const float* dut, const float* t, const int64_t* pair_i, const int64_t* pair_j,
                                 float* dx0, float* dx1, int32_t B, int32_t D, int32_t accumulate, vita_stream_t stream);

/* The same for every Gaussian-path matcher of torchcfm==1.0.5 that the reference registers (flow_matchers.py:82-99,137-145;
 * RESTATED from the published source, UNPINNED — torchcfm is absent from the image and no reference fixture covers it):
 *   VITA_FM_CONDITIONAL ('conditional', 'exact'): xt = t*x1 + (1-t)*x0 + sigma*eps,               ut = x1 - x0
 *   VITA_FM_TARGET      ('target'):               xt = t*x1 + st*eps, st = 1 - (1-sigma)*t,        ut = (x1 - (1-sigma)*xt) / st
 *   VITA_FM_SCHRODINGER ('schrodinger', sigma>0): xt = t*x1 + (1-t)*x0 + sigma*sqrt(t(1-t))*eps,   ut = (1-2t)/(2t(1-t)+1e-8)*(xt - mu) + x1 - x0
 * eps (B,D): the randn_like(x0) draw of sample_location_and_conditional_flow, indexed by output row; may be NULL only for
 * VITA_FM_CONDITIONAL with sigma == 0.  The backward follows the accumulate / pairing rules of vita_cfm_interpolate_bwd_f32. */
enum { VITA_FM_CONDITIONAL = 0, VITA_FM_TARGET = 1, VITA_FM_SCHRODINGER = 2 };
int vita_fm_interpolate_f32(int32_t kind, double sigma, const float* x0, const float* x1, const float* t, const float* eps,
                            const int64_t* pair_i, const int64_t* pair_j, float* xt, float* ut, int32_t B, int32_t D, vita_stream_t stream);
int vita_fm_interpolate_bwd_f32(int32_t kind, double sigma, const float* dxt, const float* dut, const float* t, const int64_t* pair_i,
                                const int64_t* pair_j, float* dx0, float* dx1, int32_t B, int32_t D, int32_t accumulate, vita_stream_t stream);

/* (x - mean) / (std + 1e-8) and x*std + mean for x viewed as (rows, C, inner) with per-C statistics:
 * Normalize.forward / Unnormalize.forward, normalize.py:38-46,62-70.  inner = 1 for state / action vectors
 * (stats of shape (C)); inner = H*W for images with stats of shape (C,1,1). */
int vita_normalize_f32(const float* x, const float* mean, const float* std, float* y, int64_t rows, int32_t C, int64_t inner,
                       vita_stream_t stream);
int vita_unnormalize_f32(const float* x, const float* mean, const float* std, float* y, int64_t rows, int32_t C, int64_t inner,
                         vita_stream_t stream);

/* LayerNorm over the last dim H fused with its following elementwise op, one pass over x:
 *   mode VITA_PRO_LN_MOD   : y = LN(x; no affine, eps) * (1 + mul) + add   — FlowNetLayer, simple_flow_net.py:44-45
 *   mode VITA_PRO_LN_AFFINE: y = LN(x; eps) * mul + add                    — SimpleFlowNet.norm, simple_flow_net.py:81,106
 * mul/add are (H) vectors (ld = 0) or (M,H) matrices (per-row t).  H <= 2048. */
int vita_layernorm_modulate_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t M, int32_t H, float eps, int32_t mode,
                                const float* mul, int64_t ld_mul, const float* add, int64_t ld_add, vita_stream_t stream);

/* Sinusoidal time embedding, pos_embed.py:11-18: t (B) -> (B, dim) = cat(sin(t f), cos(t f)). */
int vita_sinusoidal_embed_f32(const float* t, float* out, int64_t ldo, int32_t B, int32_t dim, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Backward-pass and optimizer kernels (autograd of the ops above; the reference gets these from
 * torch.autograd: loss.backward() at policy_trainer.py:82).
 * --------------------------------------------------------------------------------------- */
/* y = act(z), and dz = dy * act'(z) from the saved pre-activation */
int vita_act_fwd_f32(const float* z, float* y, int64_t n, int32_t act, vita_stream_t stream);
int vita_act_bwd_f32(const float* z, const float* dy, float* dz, int64_t n, int32_t act, vita_stream_t stream);
/* y = r + z*g  (simple_flow_net.py:46, flow_matchers.py:71):  dz = dy*g (contiguous (M,N), may alias z),
 * dgate_rows = dy*z (M,N) with row stride ld_dgate.
 * gate: (N) vector (ld_gate 0), (M,N) matrix, or NULL = the scalar gate_scalar.  dz / dgate_rows may be NULL. */
int vita_gate_bwd_f32(const float* dy, const float* z, const float* gate, int64_t ld_gate, float gate_scalar, float* dz,
                      float* dgate_rows, int64_t ld_dgate, int32_t M, int32_t N, vita_stream_t stream);
/* backward of vita_layernorm_modulate_f32: dx (M,H) (added into when dx_accumulate != 0: the residual
 * branch), and per-row dmul_rows = dy*n, dadd_rows = dy (row stride ld_drows; either may be NULL; reduce
 * them with vita_colsum_f32 when mul/add are vectors). */
int vita_layernorm_modulate_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* mul, int64_t ld_mul,
                                    int32_t M, int32_t H, float eps, int32_t mode, float* dx, int64_t ld_dx, int32_t dx_accumulate,
                                    float* dmul_rows, float* dadd_rows, int64_t ld_drows, vita_stream_t stream);
/* out[n] (+)= sum_m x[m,n] — bias gradients and broadcast-parameter gradients; deterministic.
 * partials: caller scratch of vita_colsum_partials_count(N) floats. */
int64_t vita_colsum_partials_count(int32_t N);
int vita_colsum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* partials, float* out, int32_t accumulate, vita_stream_t stream);
/* Conv1d(k=5, s=2, p=2) of CNNActionEncoder (action_ae.py:301-306) as a GEMM on channels-last data:
 * x (B,Lin,C) -> col (B*Lout, C*5), Lout = (Lin-1)/2+1, col[(b,lo), c*5+kk] = x[b, 2lo+kk-2, c];
 * the conv weight (Cout,Cin,5) is then the (Cout, Cin*5) matrix of vita_linear_f32.  col2im is its adjoint. */
int vita_im2col_k5s2_f32(const float* x, float* col, int32_t B, int32_t Lin, int32_t C, vita_stream_t stream);
int vita_col2im_k5s2_f32(const float* dcol, float* dx, int32_t B, int32_t Lin, int32_t C, vita_stream_t stream);
/* (B,L,C) -> (B,C,L) when to_ncl else the inverse: the x.view(B,-1) order of action_ae.py:323 */
int vita_nlc_ncl_f32(const float* in, float* out, int32_t B, int32_t L, int32_t C, int32_t to_ncl, vita_stream_t stream);
/* sum of squares of a flat buffer (gradient norm^2), deterministic; partials: vita_loss_partials_count() floats */
int vita_sumsq_f32(const float* x, int64_t n, float* partials, float* out, vita_stream_t stream);
/* One fused pass over flat fp32 buffers: g *= grad_scale * min(1, max_norm/(sqrt(*gnorm_sq)*grad_scale + 1e-6))
 * (clip_grad_norm_, policy_trainer.py:83-87; gnorm_sq NULL or max_norm <= 0 = no clipping), AdamW
 * (vita_policy.py:222-229; `step` is 1-based), then ema -= (1-ema_decay)*(ema - p) when ema != NULL
 * (diffusers EMAModel.step, base_policy.py:45-49; policy_trainer.py:90-94).  Hyper-parameters are doubles: torch
 * forms 1-beta, 1-lr*wd, lr/bias_correction in python floats and rounds to fp32 once. */
int vita_adamw_ema_step_f32(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gnorm_sq, float max_norm,
                            float grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step,
                            double ema_decay, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Dataset -> device batches on a replay buffer resident in HBM.
 * AVAlohaDataset.__getitem__ (gym_av_aloha/datasets/av_aloha_dataset.py:349-372) + default_collate for a batch of frame
 * indices: for b < B, j < T, with i = idx[b], ep = ep_index[i] (replay_buffer['episode_index'], :350) and
 * (lo, hi) = (ep_from[ep], ep_to[ep]) (episode_data_index, :308-309):
 *     q = max(lo, min(hi - 1, i + delta[j]))                         _get_query_indices, :310-313
 *     out[b, j] = src[q]
 *     is_pad[b, j] = (i + delta[j] < lo) | (i + delta[j] >= hi)      :314-319   (is_pad may be NULL)
 * delta: the key's delta indices, round(delta_timestamp * fps) (lerobot get_delta_indices).  Every pointer is device memory;
 * indices are int64 as torch.LongTensor holds them.
 * Bounds (round 3): num_frames = N, num_episodes = E.  As numpy / zarr indexing does, idx[b] < 0 wraps once (idx + N); an index
 * still outside [0, N), or an episode id outside [0, E), never leaves the buffer: the row is read from frame 0, flagged as
 * padding, and *err (NULL, or one device word the caller zeroed) is set to 1 -- the reference raises IndexError there, and so
 * does vita_amd.data.DeviceReplayDataset.batch (before the launch for host-side index lists, from err for device tensors
 * when asked to check).
 * vita_gather_window_f32: src (N, R) float rows with row stride ld_src (states, actions, precomputed camera latents) ->
 *     out (B, T, R).
 * vita_gather_window_u8_nchw_f32: src (N, H, W, C) uint8 images -> out (B, T, C, H, W) float32 = value / 255, the
 *     `.type(torch.float32).permute(0, 3, 1, 2) / 255.0` of :355-356 (C in {1, 3, 4}; B * T <= 65535 per call).
 * --------------------------------------------------------------------------------------- */
int vita_gather_window_f32(const float* src, int64_t ld_src, int32_t R, const int64_t* idx, int32_t B, const int64_t* delta, int32_t T,
                           const int64_t* ep_index, const int64_t* ep_from, const int64_t* ep_to, int64_t num_frames, int64_t num_episodes,
                           float* out, uint8_t* is_pad, int32_t* err, vita_stream_t stream);
int vita_gather_window_u8_nchw_f32(const uint8_t* src, int32_t H, int32_t W, int32_t C, const int64_t* idx, int32_t B, const int64_t* delta,
                                   int32_t T, const int64_t* ep_index, const int64_t* ep_from, const int64_t* ep_to, int64_t num_frames,
                                   int64_t num_episodes, float* out, uint8_t* is_pad, int32_t* err, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The observer's elementwise tail.  ResNetObserver's backbone (flare/policies/observers/resnet_observer.py:40-44:
 * resnet18 with norm_layer = torchvision FrozenBatchNorm2d) follows every convolution with  x * scale[c] + shift[c]  (scale =
 * weight * rsqrt(running_var + eps), shift = bias - running_mean * scale), the residual blocks add the shortcut, and a ReLU follows.
 * x, res, y: (rows / C, C, HW) contiguous float32 (NCHW with HW = H * W); res may be NULL; relu in {0, 1}.
 *     y = act((x * scale[c] + shift[c]) + res),   act = ReLU (NaN propagates, as torch.relu) or identity
 * One pass instead of torch's three to four; the arithmetic order is torch's, so the result is bitwise the unfused one.
 * _bwd: dx = g * scale[c], dres = g (NULL: not wanted) with g = 0 where y <= 0 else dy (relu; torch's threshold_backward) or g = dy; y is the forward's output
 * (may be NULL when relu == 0).  scale and shift carry no gradient (frozen statistics).  rows * HW < 2^32.
 * --------------------------------------------------------------------------------------- */
int vita_channel_affine_relu_f32(const float* x, const float* res, const float* scale, const float* shift, float* y, int64_t rows, int32_t C,
                                 int32_t HW, int32_t relu, vita_stream_t stream);
int vita_channel_affine_relu_bwd_f32(const float* dy, const float* y, const float* scale, float* dx, float* dres, int64_t rows, int32_t C,
                                     int32_t HW, int32_t relu, vita_stream_t stream);
/* The stem (conv1 -> bn1 -> relu -> maxpool of torchvision's resnet18): frozen norm + ReLU + MaxPool2d(kernel 3, stride 2, padding 1)
 * in one pass.  x: (planes / C, C, H, W); y: (planes / C, C, Ho, Wo) with Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1; idx: one byte per
 * element of y, the winning window position kh * 3 + kw (the first maximum in row-major order, NaN wins: torch's rule), or 255 where
 * the window's maximum is <= 0 (the ReLU passes no gradient there).
 * _bwd: dx[h, w] = scale[c] * sum of dy over the windows that (h, w) won (idx != 255), windows in (row, column) order — bitwise
 * torch's max_pool2d backward followed by the ReLU's and the norm's.  planes <= 65535 per call. */
int vita_affine_relu_maxpool3s2_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, int64_t planes, int32_t C,
                                    int32_t H, int32_t W, vita_stream_t stream);
int vita_affine_relu_maxpool3s2_bwd_f32(const float* dy, const uint8_t* idx, const float* scale, float* dx, int64_t planes, int32_t C, int32_t H,
                                        int32_t W, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Training forward + backward of the whole hot path in one call.
 * --------------------------------------------------------------------------------------- */
enum { VITA_ENC_CNN = 0, VITA_ENC_SIMPLE = 1,
       /* the caller's own action encoder (any module: the reference's TransformerActionEncoder, action_ae.py:36-108, a VAE variant ...):
        * vita_compute_loss_f32 only.  In the weights descriptor set A, T, D and proj_w = the encoder's output, contiguous (B, D) rows;
        * in the gradient descriptor proj_w = the (B, D) buffer that receives d loss / d latents (zeros under freeze_encoder), from
        * which the caller's autograd continues into its module.  Every other field is ignored; variational must be 0 (a variational
        * module samples, and adds its KL term, on the caller's side). */
       VITA_ENC_EXTERNAL = 2 };

typedef struct vita_encoder_weights {
    int32_t kind; /* VITA_ENC_CNN: CNNActionEncoder (action_ae.py:282-327); VITA_ENC_SIMPLE: SimpleActionEncoder (:111-170) */
    int32_t A;    /* action_dim */
    int32_t T;    /* pred_horizon */
    int32_t C;    /* cnn: enc_hidden_dim (conv channels); simple: action_latent_dim = D / T */
    int32_t NL;   /* num_layers <= VITA_MAX_LAYERS; cnn needs T >> NL >= 1 (action_ae.py:311) */
    int32_t D;    /* latent_dim */
    /* cnn:    w1[i] = action_encoder.encoder.{2i}.weight (C, Cin, 5), b1[i] its bias; w2/b2 unused
     * simple: w1[i]/b1[i] = layers.i.fc1 (4C, C), w2[i]/b2[i] = layers.i.fc2 (C, 4C) */
    const float *w1[VITA_MAX_LAYERS], *b1[VITA_MAX_LAYERS];
    const float *w2[VITA_MAX_LAYERS], *b2[VITA_MAX_LAYERS];
    /* cnn: latent_proj (D, C*(T>>NL));  simple: input_proj (C, A) */
    const float *proj_w, *proj_b;
    /* 1: the variational CNNActionEncoder of action_vae.py:179-229 (policy.action_ae.use_variational): latent_proj is
     * (2D, C*(T>>NL)) and its output rows are [mean | logvar] of a DiagonalGaussianDistribution (models/dists.py:4-21);
     * cnn only */
    int32_t variational;
} vita_encoder_weights;

typedef struct vita_policy_weights {
    int32_t O;                      /* observer feature dim */
    const float *obs_w, *obs_b;     /* obs_encoder (D, O), vita_policy.py:77 */
    vita_flownet_weights flow;
    vita_encoder_weights enc;
    vita_decoder_weights dec;
} vita_policy_weights;

typedef struct vita_loss_config { /* flare/configs/policy/vita.yaml:15-25,41-46 */
    int32_t num_steps;            /* flow_matcher.num_sampling_steps */
    int32_t recon_l1;             /* 1: recon_loss_type l1 (default), 0: l2 */
    int32_t decode_flow_latents;  /* vita_policy.py:153 */
    float consistency_weight;     /* FLC, :162-165 */
    float enc_recon_weight;       /* AE,  :183-187 */
    float flow_recon_weight;      /* FLD, :174-178 */
    float enc_contrastive_weight;  /* InfoNCE(obs_latents, action_latents), :134-139 (0 in vita.yaml) */
    float flow_contrastive_weight; /* InfoNCE(obs_latents, action_latents_pred), :167-172 (0 in vita.yaml) */
    float temperature;             /* contrastive_loss.py:5 default 0.07; 0 here selects that default */
    int32_t freeze_encoder;        /* action_ae.freeze_encoder: action latents carry no gradient (:119), no sampler branch (:153) */
    int32_t freeze_decoder;        /* action_ae.freeze_decoder: no reconstruction terms (:174,183), no sampler branch (:153);
                                      frozen modules' gradient buffers are written as zeros */
    float kl_weight;               /* action_ae.kl_weight: weight of posterior.kl().mean() (:146-150); variational encoders only */
    int32_t gemm_bf16;             /* 0: fp32 MFMA GEMMs (the reference's arithmetic).  1: GEMMs of >= 2^31 FLOP run on bf16
                                      operands (fp32 -> bf16 staged per call, fp32 accumulate / epilogue / master weights /
                                      gradients); 2: every GEMM does.  NOT bit-comparable with fp32: stated tolerance in
                                      tests/test_gpu_bf16_train.py (BASELINE configs[2]) */
    int32_t fm_kind;               /* VITA_FM_*: flow_matcher.name conditional / exact (0), target (1), schrodinger (2) */
    float fm_sigma;                /* flow_matcher.sigma (vita.yaml:24: 0.0); > 0 or kinds 1 / 2 need `fm_noise` */
    int32_t side_streams;          /* 0 (default): everything on `stream`.  1: with grads, B <= 1024 and fp32 GEMMs the decoder's weight
                                      gradients, the time path's and the encoders' backward are enqueued on three library-owned
                                      non-blocking side streams (created on a host thread's first such call on a device and kept),
                                      forked from and joined back into `stream` with events; grad_ready[0] is then recorded on a side
                                      stream.  Same bits as the single-stream order.  Measured on MI355X: the branches do not overlap
                                      the main chain's small kernels (B=128 step 3.34 -> 3.44 ms), hence off by default */
    int32_t grad_ready_layers;     /* 1: `grad_ready` holds 4 + L events, recorded in the order 0, 3, .., 3 + L - 1, 2, 1, 3 + L as the
                                      gradients they stand for become final: (0) the decoder's; (3 + l) flow_net.layers.l.mlp.{fc1,fc2}
                                      .{weight,bias} (the stacked weight-gradient GEMMs run layer by layer after the input-gradient
                                      chains); (2) the encoders' and obs_encoder's gradients and d_obs_features; (1) the time path's
                                      (time_embed.*, layers.l.time_modulator.*); (3 + L) the flow net's outer ones (input_proj, norm,
                                      out_proj).  A data-parallel caller starts each bucket's all-reduce at its event: the sections
                                      that make the most gradient bytes per unit of compute come first and 2 MB last (at H = 512), so
                                      that little is left to send when the call's kernels end, whatever the link's speed.  Same
                                      bits as without it */
    int32_t external_decoder;      /* 1: the action decoder is the caller's own module (any: the reference's CNNActionDecoder /
                                      CNNAttentionActionDecoder, action_vae.py:232-387 ...), vita_compute_loss_f32 with fp32 GEMMs
                                      only.  Of `dec` only D, OUT and in_w are read: in_w of the weights descriptor is a (2B, D)
                                      buffer that RECEIVES [pred; z1] (the sampler's result, zeros when that branch is off, and the
                                      action latents), in_w of the gradient descriptor is the (2B, D) buffer that SUPPLIES
                                      d (caller's terms) / d [pred; z1].  The reconstruction terms (vita_policy.py:174-187) are then
                                      the caller's: losses[FLOW_RECON], [ENC_RECON] stay 0 and TOTAL is the sum of the other terms.
                                      The step takes two calls on the same workspace: grads = NULL (forward; writes the latents),
                                      then — after the caller's decoder has produced its gradient — resume_backward = 1 with grads */
    int32_t resume_backward;       /* 1 (external_decoder only, grads != NULL): run only the backward pass of the forward the previous
                                      call on this workspace, stream and arguments enqueued; `losses` is left as that call wrote it */
} vita_loss_config;

enum { VITA_LOSS_TOTAL = 0, VITA_LOSS_FLOW = 1, VITA_LOSS_CONSISTENCY = 2, VITA_LOSS_FLOW_RECON = 3, VITA_LOSS_ENC_RECON = 4,
       VITA_LOSS_ENC_CONTRASTIVE = 5, VITA_LOSS_FLOW_CONTRASTIVE = 6, VITA_LOSS_ACTION_KL = 7, VITA_LOSS_COUNT = 8 };

/* vita_compute_loss_f32 — VitaPolicy.compute_loss from observer features on (vita_policy.py:112-189, defaults: plain
 * autoencoder, nothing frozen, contrastive weights 0) followed by its backward pass (loss.backward(),
 * policy_trainer.py:82), every kernel enqueued by this one call:
 *   z0 = obs_encoder(obs); z1 = action_encoder(actions);
 *   L_flow = mean((flow_net(xt, t) - ut)^2), (xt, ut) = CFM(z0[pair_i], z1[pair_j], t)       flow_matchers.py:30-36
 *   pred = Euler_K(z0);  L_c = mse(pred, z1);  L_fr = recon(dec(pred), actions);  L_er = recon(dec(z1), actions)
 *   loss = L_flow + cw*L_c + fw*L_fr + ew*L_er
 * `t` (B) and the OT pairing (NULL = identity) are inputs: the reference draws them on the host (torchcfm / POT).
 * fm_noise: (B, D) standard-normal draw of the matcher (randn_like(x0) in torchcfm's sample_location_and_conditional_flow);
 *   required unless cfg->fm_kind == VITA_FM_CONDITIONAL and cfg->fm_sigma == 0 (then it may be NULL).
 * enc_noise: variational encoder only (vita_encoder_weights.variational): the (B, D) standard-normal draw of
 *   DiagonalGaussianDistribution.sample (dists.py:20-21): z1 = mean + exp(0.5*logvar) * enc_noise, and with kl_weight > 0
 *   and an unfrozen encoder loss += kl_weight * mean_b(0.5 * sum_c(mean^2 + var - 1 - logvar)) (:146-150).  NULL = the
 *   deterministic posterior of eval mode (:121, dists.py:12-18,25-26): z1 = mean, KL = 0.  Must be NULL otherwise.
 * losses: device float[VITA_LOSS_COUNT], indexed by VITA_LOSS_*.
 * grads: NULL = forward only; else a descriptor of the same shape whose pointers are the gradient buffers: each
 *   is OVERWRITTEN with d loss / d weight (no zero_grad needed).  The K sampler steps and the flow-matching
 *   evaluation share one weight-gradient GEMM per layer (their rows are stacked in the workspace).
 * d_obs_features: NULL or (B,O) receiving d loss / d obs_features — what flows on into a trainable observer
 *   (the ResNet backbone is in the reference's single AdamW group, vita_policy.py:222-229).
 * grad_ready: NULL or 3 hipEvent_t recorded on `stream` when (0) the decoder's, (1) the flow net's, (2) every
 *   gradient is final — the points at which a data-parallel caller starts that bucket's all-reduce; with
 *   cfg->grad_ready_layers, 4 + L events in a different order and with narrower meanings (see vita_loss_config). */
size_t vita_compute_loss_workspace_bytes(const vita_policy_weights* w, const vita_loss_config* cfg, int32_t B);
int vita_compute_loss_f32(const vita_policy_weights* w, const vita_policy_weights* grads, const vita_loss_config* cfg,
                          const float* obs_features, int64_t ldobs, const float* actions, const float* t,
                          const int64_t* pair_i, const int64_t* pair_j, const float* fm_noise, const float* enc_noise,
                          int32_t B, float* losses,
                          float* d_obs_features, int64_t ld_dobs,
                          void* const* grad_ready, void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* vita_action_encoder_forward_f32 — CNNActionEncoder.forward / SimpleActionEncoder.forward alone:
 * actions (B,T,A) contiguous -> latents (B,D); the variational encoder writes its posterior's parameter rows
 * (B,2D) = [mean | logvar] (action_vae.py:224). */
size_t vita_action_encoder_workspace_bytes(const vita_encoder_weights* w, int32_t B);
int vita_action_encoder_forward_f32(const vita_encoder_weights* w, const float* actions, float* latents, int64_t ldl, int32_t B,
                                    void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * bf16-operand inference (north star (b); BASELINE configs[2]).  GEMM operands are bf16, everything else (accumulators,
 * bias, activations, LayerNorm, residual / Euler state, time path) fp32.  NOT bit-comparable with the fp32 reference:
 * the stated tolerance is 2e-2 normalised max error on the actions (tests/test_gpu_bf16.py).
 * --------------------------------------------------------------------------------------- */
/* fp32 (rows, cols) -> bf16 (rows, ld_dst), zero-filling columns cols..ld_dst-1.  bf16 GEMM operands need row strides
 * and K that are multiples of 8: pack with ld_dst = (cols + 7) & ~7. */
int vita_pack_bf16_f32(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int32_t cols, vita_stream_t stream);

/* vita_linear_f32 with bf16 X and W (args->X / args->W point at bf16 data; K, ldx, ldw count bf16 elements, multiples of
 * 8, 16-byte aligned rows).  Y (fp32) and / or y_bf16 receive the result; the epilogue options are those of
 * vita_linear_f32 except the LayerNorm prologue and transposed operands. */
int vita_linear_bf16(const vita_linear_args* args, uint16_t* y_bf16, int64_t ld_ybf, vita_stream_t stream);

/* C (M, N) fp32 = A^T . B for ROW-MAJOR bf16 A (R, lda >= M) and B (R, ldb >= N): the weight gradient dW = dY^T X of F.linear
 * (reference flow net, simple_flow_net.py:27-47, under torch.autograd) read from the bf16 copies of dY and X as they lie in
 * memory -- no transposed operand copies; the fragments are transposed on their way out of LDS (ds_read_b64_tr_b16).
 * M, N, lda, ldb multiples of 8, 16-byte aligned rows.  splitk_ws: NULL, or vita_splitk_workspace_bytes() of 256-byte aligned
 * scratch: the reduction over R is then split over workgroups and summed in a fixed order (deterministic).
 * fp32 accumulation of bf16 products: tests/test_gpu_bf16.py states the tolerance (1e-3 of the result's scale). */
int vita_linear_bf16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N, int32_t R,
                        void* splitk_ws, size_t splitk_ws_bytes, vita_stream_t stream);

typedef struct vita_bf16_weights { /* vita_pack_bf16_f32 copies of the GEMM weights, row stride = (K + 7) & ~7 */
    const uint16_t* obs_w;                                            /* obs_encoder.weight */
    const uint16_t *in_w, *out_w;                                     /* flow_net.input_proj / out_proj */
    const uint16_t *fc1_w[VITA_MAX_LAYERS], *fc2_w[VITA_MAX_LAYERS];  /* flow_net.layers.l.mlp.fc1 / fc2 */
    const uint16_t *dec_in_w, *dec_out_w;                             /* action_decoder.input_proj / output_proj */
    const uint16_t *dec_fc1_w[VITA_MAX_LAYERS], *dec_fc2_w[VITA_MAX_LAYERS];
} vita_bf16_weights;

/* VitaPolicy.generate_actions (vita_policy.py:191-208) with bf16 GEMM operands; biases, LayerNorm and time-path weights
 * come from the fp32 descriptors. */
size_t vita_generate_actions_bf16_workspace_bytes(const vita_flownet_weights* fw, const vita_decoder_weights* dw, int32_t O, int32_t B,
                                                  int32_t num_steps);
int vita_generate_actions_bf16(const float* obs_b, int32_t O, const vita_flownet_weights* fw, const vita_decoder_weights* dw,
                               const vita_bf16_weights* bw, const float* obs_features, int64_t ldobs, float* actions, int64_t lda,
                               float* latents_out, int32_t B, int32_t num_steps, void* workspace, size_t workspace_bytes,
                               vita_stream_t stream);
/* the status word of that call's in-launch sampler (vita_ode_sample_status); SYNCHRONISES */
int vita_generate_actions_bf16_status(const vita_flownet_weights* fw, int32_t O, int32_t B, int32_t num_steps, const void* workspace,
                                      int32_t* status);

/* ---------------------------------------------------------------------------------------
 * Optional loss terms (weights 0.0 / non-variational in vita.yaml:18-19,40-41).
 * --------------------------------------------------------------------------------------- */
/* compute_contrastive_loss (losses/contrastive_loss.py:5-22): symmetric InfoNCE between a (B,D) and b (B,D):
 * F.normalize both, logits = a_n.b_n^T / temperature, (CE(logits, arange) + CE(logits^T, arange)) / 2 -> loss (device float).
 * With gout (device float) the gradients gout[0] * d loss/d a, d b are written to grad_a / grad_b (either may be NULL). */
size_t vita_contrastive_loss_workspace_bytes(int32_t B, int32_t D);
int vita_contrastive_loss_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t B, int32_t D, float temperature, float* loss,
                              const float* gout, float* grad_a, int64_t ld_ga, float* grad_b, int64_t ld_gb, void* workspace,
                              size_t workspace_bytes, vita_stream_t stream);
/* DiagonalGaussianDistribution.kl(other=None).mean() (models/dists.py:23-31, vita_policy.py:148):
 * mean(0.5 * sum_c(mean^2 + exp(logvar) - 1 - logvar)).  row_scratch: B floats. */
int vita_gaussian_kl_fwd_f32(const float* mean, int64_t ld_mean, const float* logvar, int64_t ld_logvar, int32_t B, int32_t C, float* row_scratch,
                             float* out, vita_stream_t stream);
int vita_gaussian_kl_bwd_f32(const float* mean, int64_t ld_mean, const float* logvar, int64_t ld_logvar, int32_t B, int32_t C, const float* gout,
                             float* grad_mean, int64_t ld_gm, float* grad_logvar, int64_t ld_gv, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Alternate matchers, INFERENCE only.  Their training losses are not built.
 * --------------------------------------------------------------------------------------- */
/* ConsistencyFlowMatcher.sample (flow/consistency_flow_matcher.py:72-106) over SimpleFlowNet: num_steps steps at
 * t_i = i/K*(1-eps)+eps.  sigma_var == 0: z <- z + v*dt (:98).  sigma_var > 0: the update of :93-96 with
 * sigma_t = (1-t_i)*sigma_var and the caller's standard-normal `noise` (num_steps, B, D) in place of torch.randn_like.
 *   x0 (B,D) -> x_out (B,D); x_out may alias x0. */
size_t vita_consistency_sample_workspace_bytes(const vita_flownet_weights* w, int32_t B, int32_t num_steps);
int vita_consistency_sample_f32(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int32_t B,
                                int32_t num_steps, float eps, float sigma_var, float noise_scale, const float* noise, void* workspace,
                                size_t workspace_bytes, vita_stream_t stream);

typedef struct vita_meanflow_weights { /* SimpleMeanFlowNet, policies/vita/simple_mean_flow_net.py:5-61; checkpoint layout */
    int32_t D, H, FF, L;               /* input/output dim, hidden_dim, int(hidden_dim*mlp_ratio), num_layers */
    const float *t1_w, *t1_b, *t2_w, *t2_b;   /* t_embed.0 (H,1), t_embed.2 (H,H) */
    const float *h1_w, *h1_b, *h2_w, *h2_b;   /* h_embed.0, h_embed.2 */
    const float *in_w, *in_b;                 /* input_layer (H,D) */
    const float* fc1_w[VITA_MAX_LAYERS]; const float* fc1_b[VITA_MAX_LAYERS];   /* hidden_layers.{l}.0 (FF,H) */
    const float* fc2_w[VITA_MAX_LAYERS]; const float* fc2_b[VITA_MAX_LAYERS];   /* hidden_layers.{l}.3 (H,FF) */
    const float *u_w, *u_b;                   /* u_output_layer (D,H) */
    const float *v_w, *v_b;                   /* v_output_layer (D,H); may be NULL when v is not requested */
} vita_meanflow_weights;

/* SimpleMeanFlowNet.forward(x, timestep, h) (:63-81) with per-row t and h: u (B,D) and optionally v (B,D).
 * (internal_features, used only by the dispersive training loss, are not returned.) */
size_t vita_meanflow_forward_workspace_bytes(const vita_meanflow_weights* w, int32_t B);
int vita_meanflow_forward_f32(const vita_meanflow_weights* w, const float* x, int64_t ldx, const float* t, const float* h, float* u, int64_t ldu,
                              float* v, int64_t ldv, int32_t B, void* workspace, size_t workspace_bytes, vita_stream_t stream);
/* MeanFlowMatcher.sample (flow/mean_flow_matcher.py:201-233): for step k: t = 1 - k/K, h = 1/K, x <- x - h * u(x, t, h). */
size_t vita_meanflow_sample_workspace_bytes(const vita_meanflow_weights* w, int32_t B, int32_t num_steps);
int vita_meanflow_sample_f32(const vita_meanflow_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, int32_t B,
                             int32_t num_steps, void* workspace, size_t workspace_bytes, vita_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * MeanFlow TRAINING step (flow_matcher.name = "mean", flow_net.name = "simple_mean_flow_net"): VitaPolicy.compute_loss
 * (vita_policy.py:112-189) with MeanFlowMatcher.compute_loss (flow/mean_flow_matcher.py:109-199, use_imf = False,
 * dispersive_loss_weight = 0) and the MeanFlow sampler with grad (:201-233), forward + backward in one call.
 *   t, r (B): the (t, r) draw of sample_t_r (:87-107), made by the caller.
 *   losses: device float[VITA_MF_LOSS_COUNT], indexed by VITA_MF_LOSS_*.
 *   grads: NULL = forward only; else every gradient buffer is overwritten (without use_imf the v head's with zeros: it takes no part).
 *   d_obs_features: NULL or (B,O) receiving d loss / d obs_features (needs grads), as in vita_compute_loss_f32.
 *   grad_ready: NULL or 3 hipEvent_t, recorded as in vita_compute_loss_f32 (decoder / flow net / all gradients final).
 * use_imf = 1 is improved MeanFlow (flow_matcher.name = "improved_mean", :137-163).
 */
typedef struct vita_meanflow_policy_weights {
    int32_t O;
    const float *obs_w, *obs_b;
    vita_meanflow_weights flow;
    vita_encoder_weights enc;
    vita_decoder_weights dec;
} vita_meanflow_policy_weights;
typedef struct vita_meanflow_loss_config {
    int32_t num_steps, recon_l1, decode_flow_latents;
    float consistency_weight, enc_recon_weight, flow_recon_weight;
    float adaptive_loss_gamma;   /* MeanFlowMatcher(adaptive_loss_gamma=0.5) */
    float adaptive_loss_c;       /* adaptive_l2_loss(c=1e-3), mean_flow_matcher.py:10 */
    int32_t use_imf;             /* improved MeanFlow (:137-163): tangent dz = the v head's no-grad prediction at (z_t, t, h = 0),
                                    compound = u + h*stopgrad(dudt) against v, plus the auxiliary v-head loss */
    float norm_p, norm_eps;      /* adaptive_imf_loss (:21-24): per-sample sum of squares / stopgrad((that + norm_eps)^norm_p) */
    float aux_v_loss_weight;
    float dispersive_loss_weight; /* > 0: + weight * sum over the hidden layers' outputs of dispersive_loss (:27-39, 185-195): */
    float dispersive_loss_tau;    /*   log(mean(exp(-cdist(z, z)^2 / (max(cdist^2).detach() + 1e-8) / tau))) over the batch */
    int32_t freeze_encoder;       /* vita_policy.py:119: action latents without grad; with either flag the sampler block (:153) is skipped */
    int32_t freeze_decoder;       /* vita_policy.py:174, 183: no decoder terms; the frozen module's gradients are written as zeros */
    float enc_contrastive_weight; /* vita_policy.py:134-139: InfoNCE(obs_latents, action_latents) */
    float flow_contrastive_weight;/* vita_policy.py:167-172: InfoNCE(obs_latents, sampled latents), inside the sampler block */
    float temperature;            /* <= 0: 0.07, the default of compute_contrastive_loss (contrastive_loss.py:5) */
    float kl_weight;              /* variational encoder (w->enc.variational): action_kl_weight, vita_policy.py:146-150; `enc_noise` (B, D) is
                                     the posterior's standard-normal draw, NULL = deterministic posterior (z = mean, KL = 0) */
    int32_t grad_ready_layers;    /* 1: `grad_ready` holds 4 + L events (round 3), recorded in the order [0] decoder, [3 + l] hidden layer l's
                                     two Linears (l = 0 first), [1] the time / interval embeddings, [3 + L] input layer + u / v heads,
                                     [2] encoders + obs_encoder (+ d_obs_features); 0: three events, [0] decoder, [1] flow net, [2] everything */
} vita_meanflow_loss_config;
/* losses of the MeanFlow step: device float[VITA_MF_LOSS_COUNT] */
enum { VITA_MF_LOSS_TOTAL = 0, VITA_MF_LOSS_FLOW = 1 /* meanflow_loss, or imf_loss + aux_v_loss_weight * aux_v_loss */,
       VITA_MF_LOSS_CONSISTENCY = 2, VITA_MF_LOSS_FLOW_RECON = 3, VITA_MF_LOSS_ENC_RECON = 4, VITA_MF_LOSS_IMF = 5, VITA_MF_LOSS_AUX_V = 6,
       VITA_MF_LOSS_IMF_MSE = 7, VITA_MF_LOSS_AUX_V_MSE = 8, VITA_MF_LOSS_DISPERSIVE = 9 /* unweighted sum over layers */,
       VITA_MF_LOSS_FLOW_BASE = 10 /* the matcher's loss before + weight * dispersive (its 'meanflow_loss' metric); written only with the dispersive term */,
       VITA_MF_LOSS_ENC_CONTRASTIVE = 11, VITA_MF_LOSS_FLOW_CONTRASTIVE = 12, VITA_MF_LOSS_ACTION_KL = 13,
       VITA_MF_LOSS_COUNT = 16 };
size_t vita_compute_loss_meanflow_workspace_bytes(const vita_meanflow_policy_weights* w, const vita_meanflow_loss_config* cfg, int32_t B);
int vita_compute_loss_meanflow_f32(const vita_meanflow_policy_weights* w, const vita_meanflow_policy_weights* grads,
                                   const vita_meanflow_loss_config* cfg, const float* obs_features, int64_t ldobs, const float* actions,
                                   const float* t, const float* r, const float* enc_noise, int32_t B, float* losses,
                                   float* d_obs_features, int64_t ld_dobs, void* const* grad_ready, void* workspace,
                                   size_t workspace_bytes, vita_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITA_HIP_H */
