// Training-side elementwise / reduction kernels: activation and LayerNorm backward, gate
// backward, column sums (bias and broadcast-parameter gradients), conv1d im2col/col2im, CFM
// interpolation backward, and the fused clip + AdamW + EMA update.  All HBM-bound: 16-byte
// accesses where alignment allows, wave64 shuffle reductions, deterministic two-stage sums.
#include "vita_common.h"
#include "vita_host.h"

namespace vita {

constexpr int TT = 256;
static inline int tgrid(int64_t items, int cap = 4096) {
    int64_t b = (items + TT - 1) / TT;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ __launch_bounds__(TT) void act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy, float* __restrict__ dz,
                                                    int64_t n, int act) {
    const int64_t stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) dz[i] = dy[i] * act_grad(z[i], act);
}
__global__ __launch_bounds__(TT) void act_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, int64_t n, int act) {
    const int64_t stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) y[i] = apply_act(z[i], act);
}

// ---- gate backward: y = r + z*g ------------------------------------------------------------------
// dz = dy*g ; dg_rows = dy*z (optional).  g: vector (ld_g = 0), matrix (ld_g >= N) or scalar (g == NULL).
// dy, z, dz are contiguous (M,N); z and dz may be the same buffer (each element is read, then written, by one thread).
// dzb: optional bf16 copy of dz, row stride ld_b (the bf16-operand step: dz is the next GEMM's A operand).
__global__ __launch_bounds__(TT) void gate_bwd_kernel(const float* __restrict__ dy, const float* z, const float* __restrict__ g,
                                                     int64_t ld_g, float gs, float* dz, float* __restrict__ dg, int64_t ld_dg, int M, int N,
                                                     uint16_t* __restrict__ dzb, int64_t ld_b) {
    const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / N), c = (int)(i - (int64_t)r * N);
        const float d = dy[i];
        const float gv = g ? g[(int64_t)r * ld_g + c] : gs;
        if (dg) dg[(int64_t)r * ld_dg + c] = d * z[i];
        if (dz) dz[i] = d * gv;
        if (dzb) dzb[(int64_t)r * ld_b + c] = f32_to_bf16(d * gv);
    }
}

// ---- LayerNorm(+modulation) backward, one wave per row --------------------------------------------
// NC = 16-byte chunks per lane (H <= 256*NC).  Everything the row needs (x, dy, the modulation row, the old dx when
// accumulating) is requested before the first reduction: one memory round trip on the critical path instead of three.
// VEC: every pointer 16-byte aligned and every stride a multiple of 4.
// tail (LnGateTail, vita_host.h): the gate backward of the layer BELOW on the row's new dx -- dz = dx * g, dg = dx * z -- which in the
// flow net's backward chain is the next kernel and reads exactly what this one has in registers (one launch less per layer).
constexpr int LNB_MAXC = 8;
template <int NC, bool VEC>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t ldd,
                                                         const float* __restrict__ mul, int64_t ld_mul, int M, int H, float eps, int mod,
                                                         float* __restrict__ dx, int64_t ld_dx, int dx_accumulate, float* __restrict__ dmul,
                                                         float* __restrict__ dadd, int64_t ld_dma, LnGateTail tail) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * ldx;
    const float* dr = dy + (int64_t)row * ldd;
    const float* mr = mul + (int64_t)row * ld_mul;
    float* ox = dx ? dx + (int64_t)row * ld_dx : nullptr;
    const bool acc = dx != nullptr && dx_accumulate != 0;
    auto ld4 = [&](const float* p, int k) -> f32x4 {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (k < H) {
            if constexpr (VEC) r = *reinterpret_cast<const f32x4*>(p + k);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < H) r[j] = p[k + j];
            }
        }
        return r;
    };
    auto st4 = [&](float* p, int k, const f32x4& v) {
        if (k < H) {
            if constexpr (VEC) *reinterpret_cast<f32x4*>(p + k) = v;
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < H) p[k + j] = v[j];
            }
        }
    };
    f32x4 v[NC], g[NC], m[NC], old[NC], tz[NC], tg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = (c * 64 + lane) * 4;
        v[c] = ld4(xr, k);
        g[c] = ld4(dr, k);
        m[c] = ld4(mr, k);
        if (acc) old[c] = ld4(ox, k);
        if (tail.dz) {
            tz[c] = ld4(tail.z + (int64_t)row * H, k);
            tg[c] = ld4(tail.g + (int64_t)row * tail.ld_g, k);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[c][j];
    s = wave_sum(s);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = (c * 64 + lane) * 4 + j;
            const float t = k < H ? v[c][j] - mu : 0.f;
            q = fmaf(t, t, q);
        }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
    // n = (x-mu)*rstd ; dn = dy*m ; s1 = mean(dn) ; s2 = mean(dn*n) ; dx = rstd*(dn - s1 - n*s2)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k0 = (c * 64 + lane) * 4;
        f32x4 gn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < H) {
                const float nrm = (v[c][j] - mu) * rstd;
                const float mm = mod ? (m[c][j] + 1.0f) : m[c][j];
                const float dn = g[c][j] * mm;
                gn[j] = g[c][j] * nrm;
                v[c][j] = nrm;
                m[c][j] = dn;
                s1 += dn;
                s2 = fmaf(dn, nrm, s2);
            } else {
                gn[j] = 0.f; v[c][j] = 0.f; m[c][j] = 0.f;
            }
        }
        if (dmul) st4(dmul + (int64_t)row * ld_dma, k0, gn);
        if (dadd) st4(dadd + (int64_t)row * ld_dma, k0, g[c]);
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    if (dx) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k0 = (c * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gx = rstd * (m[c][j] - s1 - v[c][j] * s2);
                o[j] = acc ? old[c][j] + gx : gx;
            }
            st4(ox, k0, o);
            if (tail.dz) {
                f32x4 dgv, dzv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { dgv[j] = o[j] * tz[c][j]; dzv[j] = o[j] * tg[c][j]; }
                if (tail.dg) st4(tail.dg + (int64_t)row * tail.ld_dg, k0, dgv);
                st4(tail.dz + (int64_t)row * H, k0, dzv);
            }
        }
    }
}

// ---- column sums: out[n] (+)= sum_m x[m,n]; two deterministic stages --------------------------------
constexpr int CS_R = 64;   // row splits == partial rows; vita_colsum_partials_count(N) = CS_R * N
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ x, int64_t ldx, int M, int N, float* __restrict__ part) {
    // block: 64 columns x 4 row lanes; blockIdx.y = row split
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rows_per = (M + CS_R - 1) / CS_R;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    // eight independent row streams per thread (eight loads in flight; one was latency-bound at 1.6 TB/s), combined in a fixed order
    float s = 0.f;
    if (col < N) {
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += x[(int64_t)(r + 4 * u) * ldx + col];
        }
        for (int u = 0; r < r1; r += 4, ++u) a8[u] += x[(int64_t)r * ldx + col];
        s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    __shared__ float sh[4][64];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// The same first stage on 16-byte loads (N % 4 == 0, 16-byte aligned rows): a thread owns 4 adjacent columns, a wave 256, the
// block's 4 waves are row lanes, blockIdx.y one of CS_R4 row splits; eight float4 loads in flight per thread.  The scalar form
// above kept 32 bytes per lane in flight on 2 waves per SIMD: 3.7 TB/s; this one 128 bytes.
constexpr int CS_R4 = 256;
__global__ __launch_bounds__(256) void colsum_stage1_v4(const float* __restrict__ x, int64_t ldx, int M, int N, float* __restrict__ part) {
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int rl = threadIdx.x >> 6;
    const int rows_per = (M + CS_R4 - 1) / CS_R4;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        f32x4 a8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(x + (int64_t)(r + 4 * u) * ldx + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += t[u];
        }
        for (int u = 0; r < r1; r += 4, ++u) a8[u] += *reinterpret_cast<const f32x4*>(x + (int64_t)r * ldx + col);
        s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    __shared__ f32x4 sh[4][64];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N)
        *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.y * N + col) = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// second stage: 32 columns x 8 row lanes per block (lane j sums partial rows j, j + 8, ...; the 8 sums are added in lane order)
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int N, int R, float* __restrict__ out, int accumulate) {
    const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < N) {
#pragma unroll 8
        for (int r = rl; r < R; r += 8) s += part[(int64_t)r * N + col];
    }
    __shared__ float sh[8][33];
    sh[rl][c] = s;
    __syncthreads();
    if (rl == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sh[k][c];
        out[col] = accumulate ? out[col] + t : t;
    }
}

// few rows (M <= 2048: bias gradients of a B=128 step): ONE launch; 32 columns x 32 row lanes per block, fixed summation order
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ x, int64_t ldx, int M, int N, float* __restrict__ out,
                                                            int accumulate) {
    const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f;
    if (col < N) {
        int r = rl;
        for (; r + 32 < M; r += 64) { s0 += x[(int64_t)r * ldx + col]; s1 += x[(int64_t)(r + 32) * ldx + col]; }
        if (r < M) s0 += x[(int64_t)r * ldx + col];
    }
    __shared__ float sh[32][33];
    sh[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && col < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += sh[k][c];
        out[col] = accumulate ? out[col] + s : s;
    }
}

// ---- conv1d(k=5, s=2, p=2) as a GEMM: im2col / col2im on channels-last activations ------------------
// x (B, Lin, C) -> col (B*Lout, C*5), col[(b,lo), c*5+kk] = x[b, 2*lo+kk-2, c] (0 outside), so the
// checkpoint weight (Cout, Cin, 5) viewed as (Cout, Cin*5) is used unchanged (action_ae.py:303-305).
__global__ __launch_bounds__(TT) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int Lin, int Lout, int C) {
    const int64_t n = (int64_t)B * Lout * C * 5, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        const int kk = (int)(i % 5);
        const int c = (int)((i / 5) % C);
        const int64_t row = i / (5 * (int64_t)C);
        const int lo = (int)(row % Lout), b = (int)(row / Lout);
        const int t = 2 * lo + kk - 2;
        col[i] = (t >= 0 && t < Lin) ? x[((int64_t)b * Lin + t) * C + c] : 0.f;
    }
}
// dx[b,t,c] = sum over (lo,kk) with 2*lo+kk-2 == t of dcol[(b,lo), c*5+kk]   (a gather: deterministic)
__global__ __launch_bounds__(TT) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int Lin, int Lout, int C) {
    const int64_t n = (int64_t)B * Lin * C, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C);
        const int t = (int)((i / C) % Lin);
        const int b = (int)(i / ((int64_t)C * Lin));
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int u = t + 2 - kk;
            if (u >= 0 && (u & 1) == 0 && (u >> 1) < Lout) s += dcol[(((int64_t)b * Lout + (u >> 1)) * C + c) * 5 + kk];
        }
        dx[i] = s;
    }
}
// (B, L, C) <-> (B, C, L) flatten order (x.view(B,-1) on the (B,C,L) conv output, action_ae.py:323)
__global__ __launch_bounds__(TT) void nlc_ncl_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int L, int C, int to_ncl) {
    const int64_t n = (int64_t)B * L * C, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        // i indexes the OUTPUT
        if (to_ncl) {
            const int l = (int)(i % L), c = (int)((i / L) % C), b = (int)(i / ((int64_t)L * C));
            out[i] = in[((int64_t)b * L + l) * C + c];
        } else {
            const int c = (int)(i % C), l = (int)((i / C) % L), b = (int)(i / ((int64_t)L * C));
            out[i] = in[((int64_t)b * C + c) * L + l];
        }
    }
}

// (rows, cols) -> (cols, rows), both contiguous: 32 x 32 tiles through LDS, both sides coalesced (weight transposes of the
// large-batch training step, vita_host.h WtScope)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = src[(int64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[tx][ty + i];
    }
}

// ---- CFM interpolation backward --------------------------------------------------------------------
// xt = t*x1[j] + (1-t)*x0[i], ut = x1[j]-x0[i]  =>  dx0[i] += (1-t)*dxt - dut ; dx1[j] += t*dxt + dut
__global__ __launch_bounds__(TT) void cfm_bwd_kernel(const float* __restrict__ dxt, const float* __restrict__ dut, const float* __restrict__ t,
                                                    const int64_t* __restrict__ pi, const int64_t* __restrict__ pj, float* __restrict__ dx0,
                                                    float* __restrict__ dx1, int B, int D, int accumulate) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const float a = dxt ? dxt[i] : 0.f, u = dut ? dut[i] : 0.f, tt = t[r];
        const float g0 = (1.0f - tt) * a - u, g1 = tt * a + u;
        if (dx0) { if (pi) atomicAdd(&dx0[pi[r] * D + c], g0); else dx0[i] = accumulate ? dx0[i] + g0 : g0; }
        if (dx1) { if (pj) atomicAdd(&dx1[pj[r] * D + c], g1); else dx1[i] = accumulate ? dx1[i] + g1 : g1; }
    }
}

// backward of fm_kernel (vita_elementwise.hip).  conditional and schrodinger: d xt / d(x0, x1) = (1-t, t), d ut / d(x0, x1) =
// (-1, 1) (the bridge's extra term r*(xt - mu) = r*st*eps does not depend on x0, x1); target: x0 takes no part,
// d xt / d x1 = t, d ut / d x1 = 1/st and d ut / d xt = -(1-sigma)/st.
template <int KIND>
__global__ __launch_bounds__(TT) void fm_bwd_kernel(const float* __restrict__ dxt, const float* __restrict__ dut, const float* __restrict__ t,
                                                   const int64_t* __restrict__ pi, const int64_t* __restrict__ pj, float* __restrict__ dx0,
                                                   float* __restrict__ dx1, int B, int D, int accumulate, float oms) {
    const int64_t n = (int64_t)B * D, stride = (int64_t)gridDim.x * TT;
    for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
        const float a = dxt ? dxt[i] : 0.f, u = dut ? dut[i] : 0.f, tt = t[r];
        float g0, g1;
        if constexpr (KIND == VITA_FM_TARGET) {
            const float st = 1.0f - oms * tt;
            const float q = u / st;
            g0 = 0.f;
            g1 = q + tt * (a - oms * q);
        } else {
            g0 = (1.0f - tt) * a - u;
            g1 = tt * a + u;
        }
        if (dx0) { if (pi) atomicAdd(&dx0[pi[r] * D + c], g0); else dx0[i] = accumulate ? dx0[i] + g0 : g0; }
        if (dx1) { if (pj) atomicAdd(&dx1[pj[r] * D + c], g1); else dx1[i] = accumulate ? dx1[i] + g1 : g1; }
    }
}

// the same, up to kColsumJobs matrices in one launch (blockIdx.y = job): the bias gradients of a few-row training step are ~40
// launches of ~4 us each when taken one by one.  Same summation order per job as colsum_small_kernel: the same bits.
__global__ __launch_bounds__(1024) void colsum_small_multi_kernel(ColsumJobs jobs) {
    const ColsumJob& jb = jobs.j[blockIdx.y];
    const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    if ((int)blockIdx.x * 32 >= jb.N) return;
    const float* __restrict__ x = jb.x;
    const int64_t ldx = jb.ldx;
    const int M = jb.M;
    float s0 = 0.f, s1 = 0.f;
    if (col < jb.N) {
        int r = rl;
        for (; r + 32 < M; r += 64) { s0 += x[(int64_t)r * ldx + col]; s1 += x[(int64_t)(r + 32) * ldx + col]; }
        if (r < M) s0 += x[(int64_t)r * ldx + col];
    }
    __shared__ float sh[32][33];
    sh[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && col < jb.N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += sh[k][c];
        jb.out[col] = s;
    }
}

int ColsumBatch::add(const float* x, int64_t ldx, int M, int N, float* out) {
    if (M > 2048 || M <= 0) {      // many rows: the two-stage kernels, one call each
        return vita_colsum_f32(x, ldx, M, N, cs_part, out, 0, st);
    }
    ColsumJob& jb = jobs.j[n++];
    jb.x = x; jb.ldx = ldx; jb.M = M; jb.N = N; jb.out = out;
    if (N > max_n) max_n = N;
    return n == kColsumJobs ? flush() : VITA_OK;
}
int ColsumBatch::flush() {
    if (n == 0) return VITA_OK;
    hipLaunchKernelGGL(colsum_small_multi_kernel, dim3((max_n + 31) / 32, n), dim3(1024), 0, st, jobs);
    n = 0; max_n = 0;
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

// ---- sum of squares (gradient norm), two-stage ------------------------------------------------------
__global__ __launch_bounds__(TT) void sumsq_stage1(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    // float4 loads, four independent accumulators (16-byte aligned base: flat parameter buffers are), scalar tail
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * TT;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const int64_t n4 = n / 4;
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n4; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = fmaf(v[k], v[k], q[k]);
        }
        s = (q[0] + q[1]) + (q[2] + q[3]);
        for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) s = fmaf(x[i], x[i], s);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * TT + threadIdx.x; i < n; i += stride) s = fmaf(x[i], x[i], s);
    }
    s = wave_sum(s);
    __shared__ float ws[TT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(TT) void sumsq_stage2(const float* __restrict__ part, int count, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += TT) s += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[TT / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((ws[0] + ws[1]) + (ws[2] + ws[3]));
}

// ---- fused clip + AdamW + EMA over a flat parameter buffer -------------------------------------------
// torch.nn.utils.clip_grad_norm_ (policy_trainer.py:83-87): coef = min(1, max_norm / (norm + 1e-6));
// torch.optim.AdamW single-tensor update (vita_policy.py:222-229); diffusers EMAModel.step
// (base_policy.py:45-49): ema -= (1-decay)*(ema - p).
__global__ __launch_bounds__(TT) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, float* __restrict__ ema, int64_t n, const float* __restrict__ gnorm_sq,
                                                      float max_norm, float grad_scale, float decay_mul, float one_m_b1, float beta2,
                                                      float one_m_b2, float eps, float step_size, float bc2_sqrt, float one_m_decay) {
    float coef = grad_scale;
    if (gnorm_sq != nullptr && max_norm > 0.f) {
        const float norm = sqrtf(gnorm_sq[0]) * grad_scale;
        const float c = max_norm / (norm + 1e-6f);
        coef = grad_scale * (c < 1.0f ? c : 1.0f);
    }
    const int64_t stride = (int64_t)gridDim.x * TT;
    auto update = [&](float gi, float& pi, float& mi, float& vi, float& ei) {
        gi = gi * coef;
        pi = pi * decay_mul;                                             // p.mul_(1 - lr*wd)
        mi = mi + (gi - mi) * one_m_b1;                                  // exp_avg.lerp_(g, 1-beta1)
        vi = vi * beta2 + one_m_b2 * gi * gi;                            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);                              // p.addcdiv_(exp_avg, denom, -lr/bc1)
        ei = ei - one_m_decay * (ei - pi);
    };
    // The five streams are slices [a, b) of flat buffers with one offset: equally (mis)aligned.  A scalar head up to the first
    // 16-byte boundary, float4 body (9 streams of 16-byte accesses), scalar tail; element-wise, so the bits do not depend on it.
    const uintptr_t ap = reinterpret_cast<uintptr_t>(p);
    const bool same = ((ap ^ reinterpret_cast<uintptr_t>(g)) & 15u) == 0 && ((ap ^ reinterpret_cast<uintptr_t>(m)) & 15u) == 0 &&
                      ((ap ^ reinterpret_cast<uintptr_t>(v)) & 15u) == 0 && (ema == nullptr || ((ap ^ reinterpret_cast<uintptr_t>(ema)) & 15u) == 0) &&
                      (ap & 3u) == 0;
    int64_t head = same ? (int64_t)(((16u - (unsigned)(ap & 15u)) & 15u) >> 2) : n;
    if (head > n) head = n;
    const int64_t n4 = same ? (n - head) / 4 : 0;
    const int64_t tid = (int64_t)blockIdx.x * TT + threadIdx.x;
    for (int64_t i = tid; i < n4; i += stride) {
        const int64_t o = head + i * 4;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + o);
        f32x4 p4 = *reinterpret_cast<const f32x4*>(p + o), m4 = *reinterpret_cast<const f32x4*>(m + o), v4 = *reinterpret_cast<const f32x4*>(v + o);
        f32x4 e4 = {0.f, 0.f, 0.f, 0.f};
        if (ema) e4 = *reinterpret_cast<const f32x4*>(ema + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p4[k], mk = m4[k], vk = v4[k], ek = e4[k];
            update(g4[k], pk, mk, vk, ek);
            p4[k] = pk; m4[k] = mk; v4[k] = vk; e4[k] = ek;
        }
        *reinterpret_cast<f32x4*>(p + o) = p4; *reinterpret_cast<f32x4*>(m + o) = m4; *reinterpret_cast<f32x4*>(v + o) = v4;
        if (ema) *reinterpret_cast<f32x4*>(ema + o) = e4;
    }
    // head [0, head) and tail [head + 4 n4, n)
    const int64_t tail0 = head + n4 * 4, nrest = head + (n - tail0);
    for (int64_t j = tid; j < nrest; j += stride) {
        const int64_t i = j < head ? j : tail0 + (j - head);
        float pi = p[i], mi = m[i], vi = v[i], ei = ema ? ema[i] : 0.f;
        update(g[i], pi, mi, vi, ei);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) ema[i] = ei;
    }
}

// ---- up to 8 host scalars into device memory (loss weights seeding the backward pass) -----------------
struct Scalars8 { float v[8]; };
__global__ void set_scalars_kernel(float* __restrict__ out, Scalars8 s, int n) {
    if ((int)threadIdx.x < n) out[threadIdx.x] = s.v[threadIdx.x];
}
int launch_set_scalars(float* out, const float* host_vals, int n, hipStream_t st) {
    Scalars8 s{};
    for (int i = 0; i < n && i < 8; ++i) s.v[i] = host_vals[i];
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(64), 0, st, out, s, n);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
// out[0] = sum_i w[i] * vals[i]   (the composite loss, vita_policy.py:126-187)
__global__ void weighted_sum_kernel(const float* __restrict__ vals, Scalars8 w, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s = s + w.v[i] * vals[i];
        out[0] = s;
    }
}
int launch_weighted_sum(const float* vals, const float* host_w, int n, float* out, hipStream_t st) {
    Scalars8 s{};
    for (int i = 0; i < n && i < 8; ++i) s.v[i] = host_w[i];
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, st, vals, s, n, out);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

#define LAUNCH_OK() do { VITA_CHECK_HIP(hipGetLastError()); return VITA_OK; } while (0)

namespace vita {
int launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t s) {
    VITA_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, src, dst, rows, cols);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
}  // namespace vita

extern "C" {

int vita_act_fwd_f32(const float* z, float* y, int64_t n, int32_t act, vita_stream_t s) {
    VITA_REQUIRE(z && y && n >= 0 && act >= 0 && act <= VITA_ACT_MISH, "vita_act_fwd_f32: bad arguments");
    if (n == 0) return VITA_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(tgrid(n)), dim3(TT), 0, (hipStream_t)s, z, y, n, act);
    LAUNCH_OK();
}
int vita_act_bwd_f32(const float* z, const float* dy, float* dz, int64_t n, int32_t act, vita_stream_t s) {
    VITA_REQUIRE(z && dy && dz && n >= 0 && act >= 0 && act <= VITA_ACT_MISH, "vita_act_bwd_f32: bad arguments");
    if (n == 0) return VITA_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(tgrid(n)), dim3(TT), 0, (hipStream_t)s, z, dy, dz, n, act);
    LAUNCH_OK();
}
int vita_gate_bwd_f32(const float* dy, const float* z, const float* gate, int64_t ld_gate, float gate_scalar, float* dz, float* dgate_rows,
                      int64_t ld_dgate, int32_t M, int32_t N, vita_stream_t s) {
    return vita::launch_gate_bwd(dy, z, gate, ld_gate, gate_scalar, dz, dgate_rows, ld_dgate, M, N, (hipStream_t)s, nullptr, 0);
}
}  // extern "C"
namespace vita {
int launch_gate_bwd(const float* dy, const float* z, const float* gate, int64_t ld_gate, float gate_scalar, float* dz, float* dgate_rows,
                    int64_t ld_dgate, int M, int N, hipStream_t s, uint16_t* dz_bf16, int64_t ld_bf16) {
    VITA_REQUIRE(dy && M >= 0 && N > 0, "vita_gate_bwd_f32: bad arguments");
    VITA_REQUIRE(dz_bf16 == nullptr || (ld_bf16 >= N && (N % 8 == 0 || ld_bf16 == N)), "vita_gate_bwd_f32: bad bf16 output stride");
    VITA_REQUIRE(dgate_rows == nullptr || (z != nullptr && ld_dgate >= N), "vita_gate_bwd_f32: dgate needs z and ld_dgate >= N");
    VITA_REQUIRE(gate == nullptr || ld_gate == 0 || ld_gate >= N, "vita_gate_bwd_f32: ld_gate must be 0 or >= N");
    VITA_REQUIRE(dz == nullptr || dz != dy, "vita_gate_bwd_f32: dz must not alias dy");
    if (M == 0 || (!dz && !dgate_rows && !dz_bf16)) return VITA_OK;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(tgrid((int64_t)M * N)), dim3(TT), 0, (hipStream_t)s, dy, z, gate, ld_gate, gate_scalar, dz,
                       dgate_rows, ld_dgate, M, N, dz_bf16, ld_bf16);
    LAUNCH_OK();
}
}  // namespace vita
extern "C" {
int vita_layernorm_modulate_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* mul, int64_t ld_mul, int32_t M,
                                    int32_t H, float eps, int32_t mode, float* dx, int64_t ld_dx, int32_t dx_accumulate, float* dmul_rows,
                                    float* dadd_rows, int64_t ld_drows, vita_stream_t s) {
    return vita::launch_ln_mod_bwd(x, ldx, dy, ld_dy, mul, ld_mul, M, H, eps, mode, dx, ld_dx, dx_accumulate, dmul_rows, dadd_rows, ld_drows,
                                   (hipStream_t)s, nullptr);
}
}  // extern "C"
namespace vita {
int launch_ln_mod_bwd(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* mul, int64_t ld_mul, int M, int H, float eps,
                      int mode, float* dx, int64_t ld_dx, int dx_accumulate, float* dmul_rows, float* dadd_rows, int64_t ld_drows, hipStream_t s,
                      const LnGateTail* tail_in) {
    LnGateTail tail{};
    if (tail_in) tail = *tail_in;
    VITA_REQUIRE(tail.dz == nullptr || (dx != nullptr && tail.z != nullptr && tail.g != nullptr && (tail.ld_g == 0 || tail.ld_g >= H) &&
                                        (tail.dg == nullptr || tail.ld_dg >= H)),
                 "vita_layernorm_modulate_bwd_f32: the fused gate backward needs dx, z, g (row stride 0 or >= H) and ld_dg >= H");
    VITA_REQUIRE(x && dy && mul, "vita_layernorm_modulate_bwd_f32: null pointer");
    VITA_REQUIRE(M >= 0 && H > 0 && H <= 64 * 4 * LNB_MAXC, "vita_layernorm_modulate_bwd_f32: H=%d outside (0, %d]", H, 64 * 4 * LNB_MAXC);
    VITA_REQUIRE(ldx >= H && ld_dy >= H && (ld_mul == 0 || ld_mul >= H) && (dx == nullptr || ld_dx >= H), "vita_layernorm_modulate_bwd_f32: bad leading dimension");
    VITA_REQUIRE(mode == VITA_PRO_LN_MOD || mode == VITA_PRO_LN_AFFINE, "vita_layernorm_modulate_bwd_f32: bad mode");
    VITA_REQUIRE((dmul_rows == nullptr && dadd_rows == nullptr) || ld_drows >= H, "vita_layernorm_modulate_bwd_f32: ld_drows must be >= H");
    if (M == 0) return VITA_OK;
    auto al = [](const void* q, int64_t ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15u) == 0 && ld % 4 == 0); };
    const bool vec = H % 4 == 0 && al(x, ldx) && al(dy, ld_dy) && al(mul, ld_mul) && al(dx, ld_dx) && al(dmul_rows, ld_drows) && al(dadd_rows, ld_drows) &&
                     al(tail.z, H) && al(tail.dz, H) && al(tail.g, tail.ld_g) && al(tail.dg, tail.ld_dg);
#define VITA_LNB(NC, V)                                                                                                              \
    hipLaunchKernelGGL((ln_mod_bwd_kernel<NC, V>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)s, x, ldx, dy, ld_dy, mul, ld_mul, M, H, eps, \
                       mode == VITA_PRO_LN_MOD ? 1 : 0, dx, ld_dx, dx_accumulate, dmul_rows, dadd_rows, ld_drows, tail)
    if (H <= 512) { if (vec) VITA_LNB(2, true); else VITA_LNB(2, false); }
    else if (H <= 1024) { if (vec) VITA_LNB(4, true); else VITA_LNB(4, false); }
    else { if (vec) VITA_LNB(8, true); else VITA_LNB(8, false); }
#undef VITA_LNB
    LAUNCH_OK();
}
}  // namespace vita
extern "C" {
int64_t vita_colsum_partials_count(int32_t N) { return (int64_t)CS_R4 * (N > 0 ? N : 0); }
int vita_colsum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* partials, float* out, int32_t accumulate, vita_stream_t s) {
    VITA_REQUIRE(x && partials && out && M >= 0 && N > 0 && ldx >= N, "vita_colsum_f32: bad arguments");
    if (M <= 2048) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((N + 31) / 32), dim3(1024), 0, (hipStream_t)s, x, ldx, M, N, out, accumulate);
        LAUNCH_OK();
    }
    if (N % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(partials) & 15u) == 0 &&
        M >= 8 * CS_R4) {
        hipLaunchKernelGGL(colsum_stage1_v4, dim3((N + 255) / 256, CS_R4), dim3(256), 0, (hipStream_t)s, x, ldx, M, N, partials);
        hipLaunchKernelGGL(colsum_stage2, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)s, partials, N, CS_R4, out, accumulate);
        LAUNCH_OK();
    }
    hipLaunchKernelGGL(colsum_stage1, dim3((N + 63) / 64, CS_R), dim3(256), 0, (hipStream_t)s, x, ldx, M, N, partials);
    hipLaunchKernelGGL(colsum_stage2, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)s, partials, N, CS_R, out, accumulate);
    LAUNCH_OK();
}
int vita_im2col_k5s2_f32(const float* x, float* col, int32_t B, int32_t Lin, int32_t C, vita_stream_t s) {
    VITA_REQUIRE(x && col && B >= 0 && Lin > 0 && C > 0, "vita_im2col_k5s2_f32: bad arguments");
    const int Lout = (Lin - 1) / 2 + 1;
    if (B == 0) return VITA_OK;
    hipLaunchKernelGGL(im2col_kernel, dim3(tgrid((int64_t)B * Lout * C * 5)), dim3(TT), 0, (hipStream_t)s, x, col, B, Lin, Lout, C);
    LAUNCH_OK();
}
int vita_col2im_k5s2_f32(const float* dcol, float* dx, int32_t B, int32_t Lin, int32_t C, vita_stream_t s) {
    VITA_REQUIRE(dcol && dx && B >= 0 && Lin > 0 && C > 0, "vita_col2im_k5s2_f32: bad arguments");
    const int Lout = (Lin - 1) / 2 + 1;
    if (B == 0) return VITA_OK;
    hipLaunchKernelGGL(col2im_kernel, dim3(tgrid((int64_t)B * Lin * C)), dim3(TT), 0, (hipStream_t)s, dcol, dx, B, Lin, Lout, C);
    LAUNCH_OK();
}
int vita_nlc_ncl_f32(const float* in, float* out, int32_t B, int32_t L, int32_t C, int32_t to_ncl, vita_stream_t s) {
    VITA_REQUIRE(in && out && in != out && B >= 0 && L > 0 && C > 0, "vita_nlc_ncl_f32: bad arguments");
    if (B == 0) return VITA_OK;
    hipLaunchKernelGGL(nlc_ncl_kernel, dim3(tgrid((int64_t)B * L * C)), dim3(TT), 0, (hipStream_t)s, in, out, B, L, C, to_ncl);
    LAUNCH_OK();
}
int vita_cfm_interpolate_bwd_f32(const float* dxt, const float* dut, const float* t, const int64_t* pair_i, const int64_t* pair_j, float* dx0,
                                 float* dx1, int32_t B, int32_t D, int32_t accumulate, vita_stream_t s) {
    VITA_REQUIRE(t && (dxt || dut) && B >= 0 && D > 0, "vita_cfm_interpolate_bwd_f32: bad arguments");
    if (B == 0 || (!dx0 && !dx1)) return VITA_OK;
    hipStream_t st = (hipStream_t)s;
    if (!accumulate) {
        if (dx0 && pair_i) VITA_CHECK_HIP(hipMemsetAsync(dx0, 0, (size_t)B * D * 4, st));
        if (dx1 && pair_j) VITA_CHECK_HIP(hipMemsetAsync(dx1, 0, (size_t)B * D * 4, st));
    }
    hipLaunchKernelGGL(cfm_bwd_kernel, dim3(tgrid((int64_t)B * D)), dim3(TT), 0, st, dxt, dut, t, pair_i, pair_j, dx0, dx1, B, D, accumulate);
    LAUNCH_OK();
}
int vita_fm_interpolate_bwd_f32(int32_t kind, double sigma, const float* dxt, const float* dut, const float* t, const int64_t* pair_i,
                                const int64_t* pair_j, float* dx0, float* dx1, int32_t B, int32_t D, int32_t accumulate, vita_stream_t s) {
    const char* who = "vita_fm_interpolate_bwd_f32";
    VITA_REQUIRE(kind >= VITA_FM_CONDITIONAL && kind <= VITA_FM_SCHRODINGER, "%s: unknown matcher kind %d", who, kind);
    VITA_REQUIRE(t && (dxt || dut) && (dx0 || dx1) && B >= 0 && D > 0 && sigma >= 0.0, "%s: bad arguments", who);
    if (B == 0) return VITA_OK;
    if (!accumulate) {
        if (dx0 && pair_i) VITA_CHECK_HIP(hipMemsetAsync(dx0, 0, (size_t)B * D * 4, (hipStream_t)s));
        if (dx1 && pair_j) VITA_CHECK_HIP(hipMemsetAsync(dx1, 0, (size_t)B * D * 4, (hipStream_t)s));
    }
    const float oms = (float)(1.0 - sigma);
    if (kind == VITA_FM_TARGET)
        hipLaunchKernelGGL((fm_bwd_kernel<VITA_FM_TARGET>), dim3(tgrid((int64_t)B * D)), dim3(TT), 0, (hipStream_t)s, dxt, dut, t, pair_i, pair_j, dx0, dx1, B, D, accumulate, oms);
    else
        hipLaunchKernelGGL((fm_bwd_kernel<VITA_FM_CONDITIONAL>), dim3(tgrid((int64_t)B * D)), dim3(TT), 0, (hipStream_t)s, dxt, dut, t, pair_i, pair_j, dx0, dx1, B, D, accumulate, oms);
    LAUNCH_OK();
}
int vita_sumsq_f32(const float* x, int64_t n, float* partials, float* out, vita_stream_t s) {
    VITA_REQUIRE(x && partials && out && n > 0, "vita_sumsq_f32: bad arguments");
    const int blocks = tgrid(n, 1024);
    hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(TT), 0, (hipStream_t)s, x, n, partials);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(TT), 0, (hipStream_t)s, partials, blocks, out);
    LAUNCH_OK();
}
int vita_adamw_ema_step_f32(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* gnorm_sq, float max_norm,
                            float grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step,
                            double ema_decay, vita_stream_t s) {
    VITA_REQUIRE(p && g && m && v && n > 0 && step >= 1, "vita_adamw_ema_step_f32: bad arguments");
    // scalar hyper-parameter arithmetic in double, rounded to fp32 once, as torch does with python floats
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(tgrid(n)), dim3(TT), 0, (hipStream_t)s, p, g, m, v, ema, n, gnorm_sq, max_norm, grad_scale,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr / bc1), (float)sqrt(bc2), (float)(1.0 - ema_decay));
    LAUNCH_OK();
}

}  // extern "C"
