// TorchFlowMatcher.sample (flow_matchers.py:41-79) as ONE persistent launch per batch of latents: all K Euler steps of
// SimpleFlowNet (simple_flow_net.py:39-48, 99-108) — input_proj, L x (fc1, fc2), out_proj with the x += v*dt update —
// run inside one kernel; what used to be ~60 kernel boundaries become hand-offs between workgroups.
//
// Decomposition (B = 4096, D = H = 512, FF = 2048 fills the chip exactly):
//   * the batch is cut into ROW GROUPS of 256 rows; a TEAM of TS = FF/128 workgroups (16) owns one row group through every
//     layer of every step.  No data ever crosses teams, so the only synchronisation is a TEAM barrier (16 arrivals on one
//     counter), never a grid barrier.
//   * a phase = one GEMM of the chain on the team's 256 rows, cut into the same tiles the stand-alone kernels use
//     (fc1: one 256x128 tile per member, <256,128,32,3>; input_proj / fc2 / out_proj: 64x128 tiles with the in-workgroup
//     split-K, <64,128,64,3,KG=4>), with the same folded-LayerNorm epilogues (vita_api.hip: flow_eval_folded) — so a row's
//     arithmetic is the per-launch path's, bit for bit.
//   * teams are laid out so that a team's members have equal blockIdx % 8 (one XCD under the dispatcher's round-robin):
//     the activations a team hands over are then re-read by 16 workgroups through ONE L2.  That is a speed choice only.
//
// Hand-off (cdna_hip_programming.md Guideline 16; placement-independent, correct for any workgroup -> XCD map):
//   producer : every activation byte another workgroup will read (h*, u, x, row statistics) is stored WRITE-THROUGH
//              (`sc1` buffer stores in the epilogue); then every wave `s_waitcnt vmcnt(0)`, a workgroup barrier, and ONE
//              lane's agent-scope atomic add on the team counter.
//   consumer : the next tile's WEIGHT stages are prefetched first (they depend on nobody); ONE lane polls the counter with
//              relaxed agent-scope loads (+ s_sleep, bounded), then ONE agent-scope acquire (drops this CU's stale L1
//              lines), `s_waitcnt vmcnt(0)`, a workgroup barrier; only then are the A-operand DMAs, the row statistics
//              and the residual rows loaded (plain loads).
//   state    : one counter per team on its own 128-byte line + one timeout word, zeroed by a hipMemsetAsync ahead of the
//              launch; counters only grow within a launch (phase p may start at TS * p arrivals).
//   residency: the grid is at most 256 workgroups of 512 threads with ~150 KB of LDS each (one per CU); a team never waits
//              for another team, and every spin is bounded: on a timeout the workgroup sets the timeout word and leaves.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "vita_common.h"

#define VITA_EPI_STAMP(i)
#include "vita_gemm_dma_tile.h"
#include "vita_host.h"
#include "vita_ode_sync.h"

namespace vita {

struct OdeParams {
    vita_flownet_weights w;
    const float* x0; int64_t ldx0;       // step 0 reads it
    float* x_out; int64_t ldo;           // every step writes it (steps >= 1 in place) unless traj
    float* traj;                         // NULL or (K+1, B, D): step k reads slab k, writes slab k+1
    float *h, *hn, *u, *stat;
    const float *mod, *c1, *c2, *c1f, *c2f;
    int B, K;
    float dt;
    int TS;
    // bf16-operand variant (vita_api_bf16.hip): GEMM operands bf16, everything else fp32.  bf16 != 0 selects it.
    int bf16;
    const uint16_t *in_wb, *out_wb, *fc1_wb[VITA_MAX_LAYERS], *fc2_wb[VITA_MAX_LAYERS];
    const uint16_t* x0b;        // bf16 copy of x0 (step 0's A operand)
    uint16_t *xb_mid, *xb_final; // bf16 state written by steps 0..K-2 / by the last step
    uint16_t *hnb, *ub;          // bf16 h * mul and u
};

// the consumer half of the hand-off; run by all threads between the weight prefetch and the first dependent load
struct TeamSync {
    unsigned* cnt;
    unsigned* tmo;
    unsigned target;      // arrivals that must have been counted (0: nothing to wait for)
    unsigned limit;
    unsigned code;        // what a timeout writes into the timeout word (phase + 1)
    float* flag;          // one LDS word
    unsigned long long* stamp;   // diagnostic (NULL in normal use): receives the clock when the wait ended
    __device__ __forceinline__ bool operator()() const {
        if (target == 0) return true;
        if (threadIdx.x == 0) {
            unsigned ok = 1, spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > limit || ((spins & 255u) == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = 0;
                    break;
                }
            }
            if (!ok) {
                __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(tmo + (kStickyWord - kMaxTeams * kTeamStride), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (stamp) *stamp = clock64_real();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire after the poll has matched
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // ... completed before the barrier lets anyone load
            *reinterpret_cast<volatile unsigned*>(flag) = ok;
        }
        __syncthreads();
        const unsigned ok = *reinterpret_cast<volatile unsigned*>(flag);
        return ok != 0;
    }
};

constexpr int P_BM1 = 256, P_BN = 128, P_BK1 = 32;    // fc1 tile
constexpr int P_BM2 = 64, P_BK2 = 64;                 // input_proj / fc2 / out_proj tile (KG = 4)
constexpr int P_NS = 3;
constexpr size_t kStageFloats = (size_t)P_NS * (P_BM1 + P_BN) * P_BK1;   // == P_NS * (P_BM2 + P_BN) * P_BK2
static_assert(kStageFloats == (size_t)P_NS * (P_BM2 + P_BN) * P_BK2, "both tile shapes use the same stage footprint");
constexpr size_t kSmemFloats = kStageFloats + 2 * P_BM1 + 4;             // + folded-LayerNorm row statistics + the flag word

// One GEMM of the chain = one phase: everything dma_tile needs, plus how its tiles are dealt to the team.
struct PhaseDesc {
    GemmParamsDma g;
    int ntiles;
    int big;       // 1: 256 x 128 tiles (fc1), 0: 64 x 128 tiles
};

// Phase table for all K steps, built on the device by one small launch ahead of the persistent one (the entry points
// copy nothing from the host).  The persistent kernel then reads its phase's descriptor through the scalar cache instead
// of carrying ~100 uniform values through the whole loop (which spilled).
__global__ void ode_phase_table_kernel(OdeParams q, PhaseDesc* table) {
    const int D = q.w.D, H = q.w.H, FF = q.w.FF, L = q.w.L, B = q.B;
    const int PH = 2 + 2 * L;
    const int nseg = H / 128;
    const int phase = blockIdx.x * blockDim.x + threadIdx.x;
    if (phase >= q.K * PH) return;
    const int k = phase / PH, ph = phase - k * PH;
    const bool is_in = ph == 0, is_out = ph == PH - 1;
    const int l = is_in || is_out ? 0 : (ph - 1) >> 1;
    const bool is_fc1 = !is_in && !is_out && ((ph - 1) & 1) == 0;
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
    const float* mod = q.mod + (size_t)k * L * 3 * H;
    PhaseDesc d{};
    GemmParamsDma& g = d.g;
    g.M = B;
    g.epi.M = B; g.epi.vec = 1; g.epi.gate_scalar = 1.0f;
    if (is_in) {
        // ---- input_proj (simple_flow_net.py:100): h, h * (1 + scale_0), row statistics
        g.X = xs; g.ldx = lds_; g.W = q.w.in_w; g.ldw = D; g.N = H; g.K = D;
        g.epi.Y = q.h; g.epi.ldy = H; g.epi.N = H; g.epi.bias = q.w.in_b; g.epi.act = VITA_ACT_NONE;
        g.epi.aux_out = q.hn; g.epi.ld_aux = H; g.epi.aux_mul = mod + H; g.epi.aux_add_one = 1; g.epi.stat_out = q.stat;
        d.ntiles = 4 * (H / 128);
    } else if (is_out) {
        // ---- final LayerNorm (affine, folded) + out_proj + the Euler update x + v * dt (:106-107, flow_matchers.py:71)
        g.X = q.hn; g.ldx = H; g.W = q.w.out_w; g.ldw = H; g.N = D; g.K = H;
        g.epi.Y = xd; g.epi.ldy = ldd_; g.epi.N = D; g.epi.bias = q.c2f; g.epi.act = VITA_ACT_NONE;
        g.epi.use_gate_scalar = 1; g.epi.gate_scalar = q.dt; g.epi.resid = xs; g.epi.ld_resid = lds_;
        g.epi.ln_stats = q.stat; g.epi.ln_c1 = q.c1f; g.epi.ln_nseg = nseg; g.epi.ln_fold_eps = 1e-5f;
        d.ntiles = 4 * (D / 128);
    } else if (is_fc1) {
        // ---- fc1 + tanh-GELU on LN(h) * (1 + scale) + shift, the LayerNorm finished on the accumulators (:44-45, 19-24)
        g.X = q.hn; g.ldx = H; g.W = q.w.fc1_w[l]; g.ldw = H; g.N = FF; g.K = H;
        g.epi.Y = q.u; g.epi.ldy = FF; g.epi.N = FF; g.epi.bias = q.c2 + ((size_t)k * L + l) * FF; g.epi.act = VITA_ACT_GELU_TANH;
        g.epi.ln_stats = q.stat; g.epi.ln_c1 = q.c1 + ((size_t)k * L + l) * FF; g.epi.ln_nseg = nseg; g.epi.ln_fold_eps = 1e-6f;
        d.ntiles = FF / 128;
        d.big = 1;
    } else {
        // ---- fc2, gate, residual (:46): h += gamma * fc2(u); also the next consumer's h * mul and statistics
        g.X = q.u; g.ldx = FF; g.W = q.w.fc2_w[l]; g.ldw = FF; g.N = H; g.K = FF;
        g.epi.Y = q.h; g.epi.ldy = H; g.epi.N = H; g.epi.bias = q.w.fc2_b[l]; g.epi.act = VITA_ACT_NONE;
        g.epi.gate = mod + (size_t)l * 3 * H; g.epi.ld_gate = 0; g.epi.resid = q.h; g.epi.ld_resid = H;
        g.epi.aux_out = q.hn; g.epi.ld_aux = H; g.epi.stat_out = q.stat;
        if (l + 1 < L) { g.epi.aux_mul = mod + (size_t)(l + 1) * 3 * H + H; g.epi.aux_add_one = 1; }
        else { g.epi.aux_mul = q.w.norm_w; g.epi.aux_add_one = 0; }
        d.ntiles = 4 * (H / 128);
    }
    if (q.bf16) {
        // same chain on bf16 operands: A / W pointers and strides count 4-byte units (bf16 pairs); the fp32 residual stream h,
        // the Euler state x and the statistics stay; products a GEMM consumes are written as bf16 only
        const uint16_t* xbs = k == 0 ? q.x0b : q.xb_mid;
        uint16_t* xbd = k == q.K - 1 ? q.xb_final : q.xb_mid;
        auto f = [](const uint16_t* p_) { return reinterpret_cast<const float*>(p_); };
        g.epi.aux_out = nullptr;
        if (is_in) {
            g.X = f(xbs); g.ldx = D / 2; g.W = f(q.in_wb); g.ldw = D / 2; g.K = D / 2;
            g.epi.aux_bf = q.hnb; g.epi.ld_aux_bf = H;
        } else if (is_out) {
            g.X = f(q.hnb); g.ldx = H / 2; g.W = f(q.out_wb); g.ldw = H / 2; g.K = H / 2;
            g.epi.ybf = xbd; g.epi.ld_ybf = D;
        } else if (is_fc1) {
            g.X = f(q.hnb); g.ldx = H / 2; g.W = f(q.fc1_wb[l]); g.ldw = H / 2; g.K = H / 2;
            g.epi.Y = nullptr; g.epi.ybf = q.ub; g.epi.ld_ybf = FF;
        } else {
            g.X = f(q.ub); g.ldx = FF / 2; g.W = f(q.fc2_wb[l]); g.ldw = FF / 2; g.K = FF / 2;
            g.epi.aux_bf = q.hnb; g.epi.ld_aux_bf = H;
        }
    }
    table[phase] = d;
}

struct PersistArgs {
    const PhaseDesc* table;
    unsigned* sync;
    int nphases, TS, rg0;
    unsigned spin_limit;
    unsigned long long* stamps;   // diagnostic: NULL, or [workgroup][phase][6] clocks: 100 MHz (phase start, wait over, tiles done, published), shader (start, published)
};

template <bool BF16>
__global__ __launch_bounds__(DNT, 2) void ode_persistent_kernel(PersistArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* flag = smem + kStageFloats + 2 * P_BM1;
    const int tid = threadIdx.x;
    const int TS = q.TS;
    int team, member;
    {
        const int b = blockIdx.x, g = gridDim.x;
        if (g % (8 * TS) == 0) {      // members of a team share blockIdx % 8
            const int x = b & 7, slot = b >> 3, tpx = (g >> 3) / TS;
            team = x * tpx + slot / TS;
            member = slot % TS;
        } else {
            team = b / TS;
            member = b % TS;
        }
    }
    const int rg = q.rg0 + team;
    unsigned* cnt = q.sync + team * kTeamStride;
    unsigned* tmo = q.sync + kMaxTeams * kTeamStride;
    const PhaseDesc* __restrict__ table = q.table;

    for (int phase = 0; phase < q.nphases; ++phase) {     // phases this workgroup has completed == the team's, once all arrived
        const PhaseDesc& d = table[phase];
        // one phase: wait for the team, run this member's tiles, publish
        unsigned long long* st = q.stamps ? q.stamps + ((size_t)blockIdx.x * q.nphases + phase) * 6 : nullptr;
        if (st && tid == 0) { st[0] = clock64_real(); st[1] = st[0]; st[4] = clock64_shader(); }
        TeamSync sy{cnt, tmo, (unsigned)(TS * phase), q.spin_limit, (unsigned)phase + 1u, flag, st ? st + 1 : nullptr};
        bool first = true;
        const int ntiles = d.ntiles;
        for (int t = member; t < ntiles; t += TS) {
            if (!first) __syncthreads();          // the previous tile's epilogue is done with the stage buffers
            bool ok;
            if (d.big) {                          // 256 x 128 tiles
                ok = dma_tile<P_BM1, P_BN, P_BK1, P_NS, false, BF16, 1, false, true, VITA_ACT_GELU_TANH>(d.g, smem, rg, t, 0, sy);
            } else {                              // 64 x 128 tiles over the team's 256 x N block
                const int tn = d.g.N / P_BN;
                ok = dma_tile<P_BM2, P_BN, P_BK2, P_NS, false, BF16, 4, false, true, VITA_ACT_NONE>(d.g, smem, rg * 4 + t / tn, t % tn, 0, sy);
            }
            if (!ok) return;
            sy.target = 0;
            first = false;
        }
        if (first && !sy()) return;                        // no tile in this phase: still keep in step with the team
        if (st && tid == 0) st[2] = clock64_real();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have left
        __syncthreads();                                    // ... all of them; and the stage buffers are free again
        if (st && tid == 0) { st[3] = clock64_real(); st[5] = clock64_shader(); }
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// process-wide switch: VITA_NO_PERSISTENT=1 in the environment, or vita_ode_sample_set_persistent() at run time
static std::atomic<int> g_persistent{-1};
static bool persistent_disabled() {
    int v = g_persistent.load(std::memory_order_relaxed);
    if (v < 0) {
        v = (getenv("VITA_NO_PERSISTENT") != nullptr && atoi(getenv("VITA_NO_PERSISTENT")) != 0) ? 0 : 1;
        g_persistent.store(v, std::memory_order_relaxed);
    }
    return v == 0;
}
bool ode_persistent_enabled() { return !persistent_disabled(); }
static std::atomic<unsigned long long*> g_stamps{nullptr};
void set_ode_persistent_stamps(void* device_buf) { g_stamps.store(static_cast<unsigned long long*>(device_buf)); }
void* ode_persistent_stamps() { return g_stamps.load(); }
int set_ode_persistent(int on) {
    const bool was_off = persistent_disabled();
    g_persistent.store(on ? 1 : 0, std::memory_order_relaxed);
    return was_off ? 0 : 1;
}

// hand-off words, then the phase table of a solve with up to `K` steps and `L` layers
static size_t sync_words_bytes() { return ode_sync_words_bytes(); }
size_t ode_persistent_sync_bytes(int K, int L) { return ((sync_words_bytes() + 255) & ~size_t(255)) + (size_t)K * (2 + 2 * L) * sizeof(PhaseDesc); }
size_t ode_persistent_status_offset() { return (size_t)kStickyWord * sizeof(unsigned); }

// Shapes the persistent launch takes: everything the folded-LayerNorm path needs (interior 128-multiples, B % 256 == 0)
// and launches that fill the chip exactly with one workgroup per CU.
bool ode_persistent_eligible(const vita_flownet_weights* w, int B, const float* x0, int64_t ldx0, const float* x_out, int64_t ldo) {
    if (persistent_disabled()) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (w->L < 1 || w->H % 128 || w->FF % 128 || w->D % 128 || w->H > 2048 || B % 256) return false;
    const int TS = w->FF / 128;
    // teams of up to 16 workgroups (H <= 512 at mlp_ratio 4): measured at H = 1024 (teams of 32) the persistent launch is 4 %
    // slower than the per-launch path (profiles/r02_persistent_vs_per_launch.jsonl), at H = 512 equal
    if (TS > 16 || 256 % TS) return false;
    const int rg_per_launch = 256 / TS;
    if ((B / 256) % rg_per_launch) return false;
    if (!al(x0) || !al(x_out) || ldx0 % 4 || ldo % 4) return false;
    if ((size_t)B * (size_t)(w->FF > ldo ? w->FF : ldo) * 4 >= (size_t(1) << 31)) return false;   // 32-bit store offsets
    // every workgroup of a launch must be resident at once (a team waits for its members): one per CU, 256 per launch.  A device
    // that shows fewer CUs (a compute partition, a CU mask) keeps the per-launch path
    return device_cus() >= 256;
}

// Enqueues the phase-table build, the memset of the hand-off state and the persistent launch(es).
static int launch_persistent(OdeParams& q, int L, void* sync, hipStream_t s) {
    const int TS = q.TS;
    const int rg_per_launch = 256 / TS;
    const size_t lds = kSmemFloats * sizeof(float);
    const int nphases = q.K * (2 + 2 * L);
    PhaseDesc* table = reinterpret_cast<PhaseDesc*>(static_cast<char*>(sync) + ((sync_words_bytes() + 255) & ~size_t(255)));
    hipLaunchKernelGGL(ode_phase_table_kernel, dim3((unsigned)((nphases + 63) / 64)), dim3(64), 0, s, q, table);
    VITA_CHECK_HIP(hipGetLastError());
    PersistArgs a{};
    a.table = table; a.sync = static_cast<unsigned*>(sync); a.nphases = nphases; a.TS = TS;
    a.spin_limit = ode_spin_limit();
    a.stamps = g_stamps.load();
    for (int rg0 = 0; rg0 < q.B / 256; rg0 += rg_per_launch) {
        a.rg0 = rg0;
        // counters + timeout word before every launch; the sticky status word only with the call's first launch
        VITA_CHECK_HIP(hipMemsetAsync(sync, 0, rg0 == 0 ? sync_words_bytes() : (size_t)kSyncWords * sizeof(unsigned), s));
        if (q.bf16) hipLaunchKernelGGL(ode_persistent_kernel<true>, dim3((unsigned)(rg_per_launch * TS)), dim3(DNT), lds, s, a);
        else hipLaunchKernelGGL(ode_persistent_kernel<false>, dim3((unsigned)(rg_per_launch * TS)), dim3(DNT), lds, s, a);
        VITA_CHECK_HIP(hipGetLastError());
    }
    // a timed-out launch leaves NaN, not undefined values, in everything the caller reads (vita_ode_resident.hip)
    VITA_TRY(launch_poison_on_timeout(sync, q.x_out, q.ldo, q.B, q.w.D, s));
    if (q.traj) VITA_TRY(launch_poison_on_timeout(sync, q.traj, q.w.D, (int64_t)(q.K + 1) * q.B, q.w.D, s));
    if (q.bf16 && q.xb_final) VITA_TRY(launch_poison_on_timeout(sync, q.xb_final, ((q.w.D + 7) / 8 * 8) / 2, q.B, ((q.w.D + 7) / 8 * 8) / 2, s));
    return VITA_OK;
}

// The time-path tables (mod, c1, c2, c1f, c2f) must already be enqueued on `s`.
int launch_ode_persistent(const vita_flownet_weights* w, const float* x0, int64_t ldx0, float* x_out, int64_t ldo, float* traj, int B,
                          int K, float dt, float* h, float* hn, float* u, float* stat, const float* mod, const float* c1, const float* c2,
                          const float* c1f, const float* c2f, void* sync, hipStream_t s) {
    OdeParams q{};
    q.w = *w;
    q.x0 = x0; q.ldx0 = ldx0; q.x_out = x_out; q.ldo = ldo; q.traj = traj;
    q.h = h; q.hn = hn; q.u = u; q.stat = stat;
    q.mod = mod; q.c1 = c1; q.c2 = c2; q.c1f = c1f; q.c2f = c2f;
    q.B = B; q.K = K; q.dt = dt; q.TS = w->FF / 128;
    return launch_persistent(q, w->L, sync, s);
}

// bf16-operand variant: x0 / x_out are the fp32 Euler state (x_out may alias x0), x0b its bf16 copy, xb_final receives the
// bf16 copy of the result (may alias x0b), xb_mid the steps in between; hnb / ub are the bf16 activations.
int launch_ode_persistent_bf16(const vita_flownet_weights* w, const vita_bf16_weights* bw, const float* x0, int64_t ldx0, const uint16_t* x0b,
                               float* x_out, int64_t ldo, uint16_t* xb_mid, uint16_t* xb_final, int B, int K, float dt, float* h, uint16_t* hnb,
                               uint16_t* ub, float* stat, const float* mod, const float* c1, const float* c2, const float* c1f, const float* c2f,
                               void* sync, hipStream_t s) {
    OdeParams q{};
    q.w = *w;
    q.x0 = x0; q.ldx0 = ldx0; q.x_out = x_out; q.ldo = ldo; q.traj = nullptr;
    q.h = h; q.hn = nullptr; q.u = nullptr; q.stat = stat;
    q.mod = mod; q.c1 = c1; q.c2 = c2; q.c1f = c1f; q.c2f = c2f;
    q.B = B; q.K = K; q.dt = dt; q.TS = w->FF / 128;
    q.bf16 = 1;
    q.in_wb = bw->in_w; q.out_wb = bw->out_w;
    for (int l = 0; l < w->L; ++l) { q.fc1_wb[l] = bw->fc1_w[l]; q.fc2_wb[l] = bw->fc2_w[l]; }
    q.x0b = x0b; q.xb_mid = xb_mid; q.xb_final = xb_final ? xb_final : xb_mid; q.hnb = hnb; q.ub = ub;
    return launch_persistent(q, w->L, sync, s);
}

}  // namespace vita
