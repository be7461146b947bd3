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
        if (RELU) {       // threshold_backward: the gradient passes where the output is > 0
            const float4 y = y4[i];
            g.x = y.x > 0.0f ? g.x : 0.0f; g.y = y.y > 0.0f ? g.y : 0.0f; g.z = y.z > 0.0f ? g.z : 0.0f; g.w = y.w > 0.0f ? g.w : 0.0f;
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
        if (RELU) g = a.y[i] > 0.0f ? g : 0.0f;
        if (RES) a.dres[i] = g;
        a.dx[i] = g * a.scale[(i / a.hw) % a.C];
    }
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
    VITA_REQUIRE(rows * (int64_t)HW < (int64_t)1 << 32, "%s: %lld x %d elements: at most 2^32 - 1 per call", who, (long long)rows, HW);
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
