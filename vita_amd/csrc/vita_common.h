// Shared device helpers for the VITA hot-path kernels (gfx950 / CDNA4 only).
// Compiled with -ffp-contract=off: every FMA in this tree is written explicitly so the op
// order of the reference (e.g. `x + v*dt`: multiply, round, add — flow_matchers.py:71) is kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vita_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

namespace vita {

void set_error(const char* fmt, ...);

#define VITA_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            vita::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return VITA_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define VITA_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            vita::set_error(__VA_ARGS__);                                                 \
            return VITA_ERR_INVALID;                                                      \
        }                                                                                 \
    } while (0)

// ---- activations (fp32, accurate to a few ulp of the output scale) ----------------------
// 0.5*(1+tanh(u)) == 1/(1+exp(-2u)); avoids the cancellation of the tanh form and costs one
// v_exp_f32 + one division.  PyTorch: GeluKernel "tanh" approximation constants.
__device__ __forceinline__ float act_gelu_tanh(float x) {
    // inner = sqrt(2/pi) * (x + 0.044715 x^3) = x * (beta + beta*kappa * x^2); exp(-2*inner) = exp2(x * (c0 + c1 * x^2)) with the
    // -2*log2(e) factor folded into the two constants: mul, fma, mul, v_exp_f32, add, v_rcp_f32, mul (v_exp_f32 and v_rcp_f32
    // are 1-ulp instructions).  Round 3: 7 VALU operations instead of 10 -- fc1's epilogue is bound by them.
    const float c0 = -2.8853900817779268f * 0.7978845608028654f;              // -2 log2(e) sqrt(2/pi)
    const float c1 = -2.8853900817779268f * 0.7978845608028654f * 0.044715f;
    const float p = fmaf(x * x, c1, c0);
    const float e = __builtin_amdgcn_exp2f(x * p);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float act_gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_mish(float x) {
    // x * tanh(softplus(x)), softplus threshold 20 as in torch.
    float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case VITA_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case VITA_ACT_GELU_ERF: return act_gelu_erf(x);
        case VITA_ACT_GELU_TANH: return act_gelu_tanh(x);
        case VITA_ACT_SILU: return act_silu(x);
        case VITA_ACT_MISH: return act_mish(x);
        default: return x;
    }
}

template <int ACT>
__device__ __forceinline__ float apply_act_t(float x) {
    if constexpr (ACT == VITA_ACT_RELU) return x > 0.0f ? x : 0.0f;
    else if constexpr (ACT == VITA_ACT_GELU_ERF) return act_gelu_erf(x);
    else if constexpr (ACT == VITA_ACT_GELU_TANH) return act_gelu_tanh(x);
    else if constexpr (ACT == VITA_ACT_SILU) return act_silu(x);
    else if constexpr (ACT == VITA_ACT_MISH) return act_mish(x);
    else return x;
}

// ---- activation derivatives (autograd of the activations above; torch's *Backward formulas) ----
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case VITA_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case VITA_ACT_GELU_ERF: {   // torch GeluBackward (none): cdf + z*pdf
            const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
            const float pdf = expf(-0.5f * z * z) * 0.3989422804014327f;
            return cdf + z * pdf;
        }
        case VITA_ACT_GELU_TANH: {  // torch GeluBackward (tanh)
            const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
            const float z2 = z * z;
            const float inner = kBeta * (z + kKappa * z2 * z);
            // tanh(u) = 1 - 2 / (1 + exp(2u)) on v_exp_f32 / v_rcp_f32 (1-ulp instructions; absolute error ~1e-7, the derivative
            // only uses 1 + th and 1 - th^2): tanhf() cost the input-gradient GEMMs' epilogue 10 % of the kernel
            const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(inner * 2.8853900817779268f));
            const float left = 0.5f * z, right = 1.0f + th;
            const float dth = 1.0f - th * th;
            const float dinner = kBeta * (1.0f + 3.0f * kKappa * z2);
            return 0.5f * right + left * (dth * dinner);
        }
        case VITA_ACT_SILU: {
            const float s = 1.0f / (1.0f + expf(-z));
            return s * (1.0f + z * (1.0f - s));
        }
        case VITA_ACT_MISH: {       // d/dz z*tanh(softplus(z)) = tanh(sp) + z*sigmoid(z)*(1-tanh(sp)^2)
            const float sp = z > 20.0f ? z : log1pf(expf(z));
            const float th = tanhf(sp);
            const float s = 1.0f / (1.0f + expf(-z));
            return th + z * s * (1.0f - th * th);
        }
        default: return 1.f;
    }
}

// fp32 -> bf16, round to nearest even (hipcc emits v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float x) {
    const __bf16 b = (__bf16)x;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace vita
