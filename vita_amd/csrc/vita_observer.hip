// The observer's elementwise tail: ResNet18 with frozen batch-norm statistics (resnet_observer.py:40-44) follows
// every convolution with  y = x * scale[c] + shift[c]  (torchvision FrozenBatchNorm2d),  optionally  + shortcut,  then ReLU.  As
// separate torch ops that is 3-4 passes over activations of up to 565 MB (128 frames, 64 x 112 x 154); here it is one pass forward
// (read x [+ shortcut], write y) and one pass backward (read dy and y, write dx [+ d shortcut]).  HBM-bound: 8 B (12 B with a
// shortcut) per element forward.  The convolutions themselves stay MIOpen's.
// Arithmetic order is torch's: (x * scale) + shift, then + shortcut, then the ReLU (no fused multiply-add: -ffp-contract=off).
#include "vita_common.h"
#include "vita_host.h"

namespace vita {

struct AffineArgs {
    const float* x; const float* res; const float* scale; const float* shift; float* y;
    unsigned n;      // elements (scalar kernel) or float4 groups (vector kernel)
    unsigned hw;     // row length in the same unit
    unsigned C;
};

__device__ __forceinline__ float relu_nan(float v) { return v < 0.0f ? 0.0f : v; }      // NaN stays NaN, as torch's clamp_min

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void channel_affine_v4_kernel(AffineArgs a) {
    const float4* x4 = reinterpret_cast<const float4*>(a.x);
    const float4* r4 = reinterpret_cast<const float4*>(a.res);
    float4* y4 = reinterpret_cast<float4*>(a.y);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned i = (blockIdx.x * 4u + u) * 256u + threadIdx.x;
        if (i >= a.n) return;
        const unsigned c = (i / a.hw) % a.C;
        const float s = a.scale[c], b = a.shift[c];
        float4 v = x4[i];
        v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
        if (RES) { const float4 r = r4[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (RELU) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
        y4[i] = v;
    }
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void channel_affine_kernel(AffineArgs a) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned i = (blockIdx.x * 4u + u) * 256u + threadIdx.x;
        if (i >= a.n) return;
        const unsigned c = (i / a.hw) % a.C;
        float v = a.x[i] * a.scale[c] + a.shift[c];
        if (RES) v += a.res[i];
        a.y[i] = RELU ? relu_nan(v) : v;
    }
}

struct AffineBwdArgs {
    const float* dy; const float* y; const float* scale; float* dx; float* dres;
    unsigned n, hw, C;
};

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void channel_affine_bwd_v4_kernel(AffineBwdArgs a) {
    const float4* g4 = reinterpret_cast<const float4*>(a.dy);
    const float4* y4 = reinterpret_cast<const float4*>(a.y);
    float4* dx4 = reinterpret_cast<float4*>(a.dx);
    float4* dr4 = reinterpret_cast<float4*>(a.dres);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned i = (blockIdx.x * 4u + u) * 256u + threadIdx.x;
        if (i >= a.n) return;
        const float s = a.scale[(i / a.hw) % a.C];
        float4 g = g4[i];
        if (RELU) {       // threshold_backward: the gradient is dropped where the output is <= 0 (a NaN output passes it)
            const float4 y = y4[i];
            g.x = y.x <= 0.0f ? 0.0f : g.x; g.y = y.y <= 0.0f ? 0.0f : g.y; g.z = y.z <= 0.0f ? 0.0f : g.z; g.w = y.w <= 0.0f ? 0.0f : g.w;
        }
        if (RES) dr4[i] = g;
        g.x *= s; g.y *= s; g.z *= s; g.w *= s;
        dx4[i] = g;
    }
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void channel_affine_bwd_kernel(AffineBwdArgs a) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned i = (blockIdx.x * 4u + u) * 256u + threadIdx.x;
        if (i >= a.n) return;
        float g = a.dy[i];
        if (RELU) g = a.y[i] <= 0.0f ? 0.0f : g;
        if (RES) a.dres[i] = g;
        a.dx[i] = g * a.scale[(i / a.hw) % a.C];
    }
}

// ---- the stem: frozen norm + ReLU + MaxPool2d(3, stride 2, padding 1) in one pass ------------------------------------------------
// Forward reads the convolution's output once (565 MB at 128 frames) and writes only the pooled map (a quarter of it) and one
// byte per pooled element: which of the window's 9 positions won (torch's rule: the first maximum in row-major order, NaN wins).
// A window whose maximum is <= 0 passes no gradient through the ReLU: its byte says so (kNoGrad).
// Backward therefore needs neither the convolution's output nor any activation: a pixel receives dy of every window it won.
// Sums run over windows in (row, column) order, as torch's max_pool2d backward does.
// Even W (the reference's crops: 308, 224, 84): a lane holds two adjacent columns (float2 loads and stores, whole cache lines) plus the
// column to their left (an L1 hit: the neighbour lane's); odd W takes the one-element-per-lane kernels.
constexpr unsigned char kNoGrad = 255;
#define VITA_POOL_UPDATE(v, k) do { const float v__ = (v); if (arg < 0 || v__ > best || v__ != v__) { best = v__; arg = (k); } } while (0)
struct StemArgs {
    const float* x; const float* scale; const float* shift; float* y; unsigned char* idx;
    unsigned planes, C, H, W, Ho, Wo;
};

__global__ __launch_bounds__(256) void stem_tail_kernel(StemArgs a) {
    const unsigned wo = blockIdx.x * 64u + (threadIdx.x & 63u);
    const unsigned ho = blockIdx.y * 4u + (threadIdx.x >> 6);
    const unsigned plane = blockIdx.z;
    if (wo >= a.Wo || ho >= a.Ho) return;
    const unsigned c = plane % a.C;
    const float s = a.scale[c], b = a.shift[c];
    const float* xp = a.x + (size_t)plane * a.H * a.W;
    float best = -INFINITY;
    int arg = -1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = (int)(2u * ho) - 1 + kh;
        if (h < 0 || h >= (int)a.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = (int)(2u * wo) - 1 + kw;
            if (w < 0 || w >= (int)a.W) continue;
            const float v = relu_nan(xp[(size_t)h * a.W + w] * s + b);
            if (arg < 0 || v > best || v != v) { best = v; arg = kh * 3 + kw; }
        }
    }
    const size_t o = ((size_t)plane * a.Ho + ho) * a.Wo + wo;
    a.y[o] = best;
    a.idx[o] = best <= 0.0f ? kNoGrad : (unsigned char)arg;
}

__global__ __launch_bounds__(256) void stem_tail_even_kernel(StemArgs a) {
    // lanes run over the plane's pooled outputs in row-major order: full waves whatever Wo is.  All six loads of a lane (a float2 and
    // its left neighbour on three rows) are unconditional on clamped addresses, so they are in flight together; validity is applied
    // when the window is scanned.
    const unsigned total = a.Ho * a.Wo;
    const unsigned p0 = blockIdx.x * 256u + threadIdx.x;
    const unsigned plane = blockIdx.y;
    const bool live = p0 < total;
    const unsigned p = live ? p0 : total - 1u;
    const unsigned ho = p / a.Wo, wo = p - ho * a.Wo;
    const bool has_left = wo > 0;
    const unsigned c = plane % a.C;
    const float s = a.scale[c], b = a.shift[c];
    const float* xp = a.x + (size_t)plane * a.H * a.W;
    float2 v[3];
    float l[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = (int)(2u * ho) - 1 + kh;
        const int hc = h < 0 ? 0 : (h >= (int)a.H ? (int)a.H - 1 : h);
        const float* row = xp + (size_t)hc * a.W + 2u * wo;
        v[kh] = *reinterpret_cast<const float2*>(row);
        l[kh] = *(row - (has_left ? 1 : 0));
    }
    float best = -INFINITY;
    int arg = -1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = (int)(2u * ho) - 1 + kh;
        if (h < 0 || h >= (int)a.H) continue;
        if (has_left) VITA_POOL_UPDATE(relu_nan(l[kh] * s + b), kh * 3 + 0);
        VITA_POOL_UPDATE(relu_nan(v[kh].x * s + b), kh * 3 + 1);
        VITA_POOL_UPDATE(relu_nan(v[kh].y * s + b), kh * 3 + 2);
    }
    if (!live) return;
    const size_t o = (size_t)plane * total + p;
    a.y[o] = best;
    a.idx[o] = best <= 0.0f ? kNoGrad : (unsigned char)arg;
}

struct StemBwdArgs {
    const float* dy; const unsigned char* idx; const float* scale; float* dx;
    unsigned planes, C, H, W, Ho, Wo;
};

__global__ __launch_bounds__(256) void stem_tail_bwd_kernel(StemBwdArgs a) {
    const unsigned w = blockIdx.x * 64u + (threadIdx.x & 63u);
    const unsigned h = blockIdx.y * 4u + (threadIdx.x >> 6);
    const unsigned plane = blockIdx.z;
    if (w >= a.W || h >= a.H) return;
    const size_t po = (size_t)plane * a.Ho * a.Wo;
    // windows ho with 2 ho - 1 <= h <= 2 ho + 1
    const int ho_lo = (int)(h / 2u), ho_hi = (int)((h + 1u) / 2u);
    const int wo_lo = (int)(w / 2u), wo_hi = (int)((w + 1u) / 2u);
    float g = 0.0f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= (int)a.Ho) break;
        const int kh = (int)h - (2 * ho - 1);
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            if (wo >= (int)a.Wo) break;
            const int kw = (int)w - (2 * wo - 1);
            const size_t o = po + (size_t)ho * a.Wo + wo;
            if (a.idx[o] == (unsigned char)(kh * 3 + kw)) g += a.dy[o];
        }
    }
    a.dx[((size_t)plane * a.H + h) * a.W + w] = g * a.scale[plane % a.C];
}

__global__ __launch_bounds__(256) void stem_tail_bwd_even_kernel(StemBwdArgs a) {
    // lanes run over (h, j) in row-major order, j = a column pair (2j, 2j + 1).  The (up to) two window rows x two windows a lane looks
    // at are loaded unconditionally on clamped addresses (eight loads in flight), validity is applied when they are summed.
    const unsigned total = a.H * a.Wo;
    const unsigned p0 = blockIdx.x * 256u + threadIdx.x;
    const unsigned plane = blockIdx.y;
    const bool live = p0 < total;
    const unsigned p = live ? p0 : total - 1u;
    const unsigned h = p / a.Wo, j = p - h * a.Wo;
    const bool has_right = j + 1u < a.Wo;
    const size_t po = (size_t)plane * a.Ho * a.Wo;
    const unsigned ho0 = h / 2u, nwin = (h & 1u) ? 2u : 1u;
    int i0[2], i1[2];
    float d0[2], d1[2];
#pragma unroll
    for (unsigned t = 0; t < 2u; ++t) {
        const unsigned ho = ho0 + t < a.Ho ? ho0 + t : a.Ho - 1u;
        const size_t o = po + (size_t)ho * a.Wo + j;
        const size_t o1 = o + (has_right ? 1u : 0u);
        i0[t] = a.idx[o]; d0[t] = a.dy[o];
        i1[t] = a.idx[o1]; d1[t] = a.dy[o1];
    }
    float ga = 0.0f, gb = 0.0f;
#pragma unroll
    for (unsigned t = 0; t < 2u; ++t) {
        const unsigned ho = ho0 + t;
        if (t >= nwin || ho >= a.Ho) continue;
        const int kh = (int)h - (2 * (int)ho - 1);
        if (i0[t] == kh * 3 + 1) ga += d0[t];           // column 2j: the middle of window j
        if (i0[t] == kh * 3 + 2) gb += d0[t];           // column 2j + 1: the right edge of window j, then the left edge of window j + 1
        if (has_right && i1[t] == kh * 3 + 0) gb += d1[t];
    }
    if (!live) return;
    const float s = a.scale[plane % a.C];
    *reinterpret_cast<float2*>(a.dx + ((size_t)plane * a.H + h) * a.W + 2u * j) = make_float2(ga * s, gb * s);
}

__global__ __launch_bounds__(256) void stem_tail_bwd_quad_kernel(StemBwdArgs a) {
    // W % 4 == 0: a lane owns four columns 4q .. 4q + 3 of row h (one float4 store) and looks at windows 2q, 2q + 1, 2q + 2 of the one
    // or two window rows over h: per window row one 2-byte and one 1-byte index load, one float2 and one float dy load.
    const unsigned Q = a.W / 4u;
    const unsigned total = a.H * Q;
    const unsigned p0 = blockIdx.x * 256u + threadIdx.x;
    const unsigned plane = blockIdx.y;
    const bool live = p0 < total;
    const unsigned p = live ? p0 : total - 1u;
    const unsigned h = p / Q, q = p - h * Q;
    const bool has_right = q + 1u < Q;                 // window 2q + 2 exists
    const size_t po = (size_t)plane * a.Ho * a.Wo;
    const unsigned ho0 = h / 2u, nwin = (h & 1u) ? 2u : 1u;
    int ia[2], ib[2], ic[2];
    float da[2], db[2], dc[2];
#pragma unroll
    for (unsigned t = 0; t < 2u; ++t) {
        const unsigned ho = ho0 + t < a.Ho ? ho0 + t : a.Ho - 1u;
        const size_t o = po + (size_t)ho * a.Wo + 2u * q;
        const size_t oc = o + (has_right ? 2u : 0u);
        const unsigned short two = *reinterpret_cast<const unsigned short*>(a.idx + o);
        const float2 d = *reinterpret_cast<const float2*>(a.dy + o);
        ia[t] = two & 0xff; ib[t] = two >> 8; ic[t] = a.idx[oc];
        da[t] = d.x; db[t] = d.y; dc[t] = a.dy[oc];
    }
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
#pragma unroll
    for (unsigned t = 0; t < 2u; ++t) {
        const unsigned ho = ho0 + t;
        if (t >= nwin || ho >= a.Ho) continue;
        const int k = ((int)h - (2 * (int)ho - 1)) * 3;
        if (ia[t] == k + 1) g0 += da[t];                      // column 4q: the middle of window 2q
        if (ia[t] == k + 2) g1 += da[t];                      // column 4q + 1: right edge of window 2q, then left edge of window 2q + 1
        if (ib[t] == k + 0) g1 += db[t];
        if (ib[t] == k + 1) g2 += db[t];                      // column 4q + 2: the middle of window 2q + 1
        if (ib[t] == k + 2) g3 += db[t];                      // column 4q + 3: right edge of window 2q + 1, then left edge of window 2q + 2
        if (has_right && ic[t] == k + 0) g3 += dc[t];
    }
    if (!live) return;
    const float s = a.scale[plane % a.C];
    *reinterpret_cast<float4*>(a.dx + ((size_t)plane * a.H + h) * a.W + 4u * q) = make_float4(g0 * s, g1 * s, g2 * s, g3 * s);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define VITA_AFFINE_DISPATCH(KERNEL, ARGS)                                                                                   \
    do {                                                                                                                       \
        const dim3 grid((ARGS.n + 1023u) / 1024u), block(256);                                                                 \
        if (has_res && relu) hipLaunchKernelGGL((KERNEL<true, true>), grid, block, 0, (hipStream_t)stream, ARGS);              \
        else if (has_res) hipLaunchKernelGGL((KERNEL<true, false>), grid, block, 0, (hipStream_t)stream, ARGS);                \
        else if (relu) hipLaunchKernelGGL((KERNEL<false, true>), grid, block, 0, (hipStream_t)stream, ARGS);                   \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, block, 0, (hipStream_t)stream, ARGS);                            \
    } while (0)

static int check_shape(const char* who, int64_t rows, int32_t C, int32_t HW) {
    VITA_REQUIRE(rows >= 0 && C >= 1 && HW >= 1 && rows % C == 0, "%s: rows (%lld) must be a multiple of C (%d), HW (%d) >= 1", who,
                 (long long)rows, C, HW);
    // the kernels index in 32-bit unsigned and round the grid up to whole 1024-element blocks: leave that block's headroom
    VITA_REQUIRE(rows * (int64_t)HW <= ((int64_t)1 << 32) - 1024, "%s: %lld x %d elements: at most 2^32 - 1024 per call", who, (long long)rows, HW);
    return VITA_OK;
}

}  // namespace vita

using namespace vita;

extern "C" int vita_channel_affine_relu_f32(const float* x, const float* res, const float* scale, const float* shift, float* y, int64_t rows,
                                            int32_t C, int32_t HW, int32_t relu, vita_stream_t stream) {
    const char* who = "vita_channel_affine_relu_f32";
    VITA_TRY(check_shape(who, rows, C, HW));
    if (rows == 0) return VITA_OK;
    VITA_REQUIRE(x && scale && shift && y, "%s: null pointer", who);
    const bool has_res = res != nullptr;
    const bool vec = HW % 4 == 0 && aligned16(x) && aligned16(y) && (!has_res || aligned16(res));
    const uint64_t n = (uint64_t)rows * (uint64_t)HW;
    AffineArgs a{x, res, scale, shift, y, (unsigned)(vec ? n / 4 : n), (unsigned)(vec ? HW / 4 : HW), (unsigned)C};
    if (vec) VITA_AFFINE_DISPATCH(channel_affine_v4_kernel, a);
    else VITA_AFFINE_DISPATCH(channel_affine_kernel, a);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

extern "C" int vita_channel_affine_relu_bwd_f32(const float* dy, const float* y, const float* scale, float* dx, float* dres, int64_t rows,
                                                int32_t C, int32_t HW, int32_t relu, vita_stream_t stream) {
    const char* who = "vita_channel_affine_relu_bwd_f32";
    VITA_TRY(check_shape(who, rows, C, HW));
    if (rows == 0) return VITA_OK;
    VITA_REQUIRE(dy && scale && dx && (y || !relu), "%s: null pointer", who);
    const bool has_res = dres != nullptr;
    const bool vec = HW % 4 == 0 && aligned16(dy) && aligned16(dx) && (!relu || aligned16(y)) && (!has_res || aligned16(dres));
    const uint64_t n = (uint64_t)rows * (uint64_t)HW;
    AffineBwdArgs a{dy, y, scale, dx, dres, (unsigned)(vec ? n / 4 : n), (unsigned)(vec ? HW / 4 : HW), (unsigned)C};
    if (vec) VITA_AFFINE_DISPATCH(channel_affine_bwd_v4_kernel, a);
    else VITA_AFFINE_DISPATCH(channel_affine_bwd_kernel, a);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

static int check_stem(const char* who, int64_t planes, int32_t C, int32_t H, int32_t W) {
    VITA_REQUIRE(planes >= 0 && C >= 1 && H >= 1 && W >= 1 && planes % C == 0 && planes <= 65535,
                 "%s: planes (%lld) must be a multiple of C (%d) and at most 65535 per call, H (%d), W (%d) >= 1", who, (long long)planes, C, H, W);
    return VITA_OK;
}

extern "C" int vita_affine_relu_maxpool3s2_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, int64_t planes,
                                               int32_t C, int32_t H, int32_t W, vita_stream_t stream) {
    const char* who = "vita_affine_relu_maxpool3s2_f32";
    VITA_TRY(check_stem(who, planes, C, H, W));
    if (planes == 0) return VITA_OK;
    VITA_REQUIRE(x && scale && shift && y && idx, "%s: null pointer", who);
    const unsigned Ho = (unsigned)(H - 1) / 2u + 1u, Wo = (unsigned)(W - 1) / 2u + 1u;
    StemArgs a{x, scale, shift, y, idx, (unsigned)planes, (unsigned)C, (unsigned)H, (unsigned)W, Ho, Wo};
    const dim3 grid((Wo + 63u) / 64u, (Ho + 3u) / 4u, (unsigned)planes);
    if (W % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0)
        hipLaunchKernelGGL(stem_tail_even_kernel, dim3((Ho * Wo + 255u) / 256u, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

extern "C" int vita_affine_relu_maxpool3s2_bwd_f32(const float* dy, const uint8_t* idx, const float* scale, float* dx, int64_t planes, int32_t C,
                                                   int32_t H, int32_t W, vita_stream_t stream) {
    const char* who = "vita_affine_relu_maxpool3s2_bwd_f32";
    VITA_TRY(check_stem(who, planes, C, H, W));
    if (planes == 0) return VITA_OK;
    VITA_REQUIRE(dy && idx && scale && dx, "%s: null pointer", who);
    const unsigned Ho = (unsigned)(H - 1) / 2u + 1u, Wo = (unsigned)(W - 1) / 2u + 1u;
    StemBwdArgs a{dy, idx, scale, dx, (unsigned)planes, (unsigned)C, (unsigned)H, (unsigned)W, Ho, Wo};
    if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(idx) & 1u) == 0)
        hipLaunchKernelGGL(stem_tail_bwd_quad_kernel, dim3(((unsigned)H * ((unsigned)W / 4u) + 255u) / 256u, (unsigned)planes), dim3(256), 0,
                           (hipStream_t)stream, a);
    else if (W % 2 == 0 && (reinterpret_cast<uintptr_t>(dx) & 7u) == 0)
        hipLaunchKernelGGL(stem_tail_bwd_even_kernel, dim3(((unsigned)H * Wo + 255u) / 256u, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(stem_tail_bwd_kernel, dim3(((unsigned)W + 63u) / 64u, ((unsigned)H + 3u) / 4u, (unsigned)planes), dim3(256), 0,
                           (hipStream_t)stream, a);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}
