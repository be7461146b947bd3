// Dataset -> device batches on a replay buffer that lives in HBM.
//
// The reference assembles a training batch on the host: 16 DataLoader workers each run AVAlohaDataset.__getitem__
// (gym_av_aloha/datasets/av_aloha_dataset.py:307-372: clamp idx + delta to the frame's episode, flag what was clamped, read the
// rows from the zarr replay buffer, uint8 images -> float32 / 255 in channels-first order), default_collate stacks the items and
// the batch crosses PCIe.  With 288 GB of HBM the replay buffer (uint8 images included) stays on the device, and a batch is one
// launch per key: every output row is gathered straight from its clamped source row.  HBM-bound: bytes of the batch read + written.
#include "vita_common.h"
#include "vita_host.h"

namespace vita {

struct WindowArgs {
    const int64_t* idx; int B;           // frame indices of the batch
    const int64_t* delta; int T;         // delta indices of this key (get_delta_indices: round(delta_timestamp * fps))
    const int64_t* ep_index;             // (N) episode of every frame (replay_buffer['episode_index'])
    const int64_t* ep_from;              // (E) first frame of every episode
    const int64_t* ep_to;                // (E) one past the last frame
    unsigned char* is_pad;               // NULL or (B, T): 1 where idx + delta fell outside the episode (av_aloha_dataset.py:314-319)
    int64_t N; int64_t E;                // frames / episodes of the replay buffer: bounds of idx and of ep_index's values
    int* err;                            // NULL or one word: set to 1 when an index (after wrapping) or an episode id is out of range
};

// source frame of (b, j), _get_query_indices (:307-313): max(ep_start, min(ep_end - 1, idx + delta))
__device__ __forceinline__ int64_t window_source(const WindowArgs& a, int b, int j, bool* pad) {
    // the reference indexes numpy / zarr arrays: a negative index wraps once, anything else outside [0, N) raises IndexError.
    // Here an out-of-range index never leaves the buffer: it reads frame 0, flags the row as padding and sets *err (the host
    // wrapper raises IndexError for host-side index lists before the launch; device-side index tensors are checked through err)
    int64_t i = a.idx[b];
    if (i < 0) i += a.N;
    bool bad = i < 0 || i >= a.N;
    if (bad) i = 0;
    int64_t ep = a.ep_index[i];
    if (ep < 0 || ep >= a.E) { bad = true; ep = 0; }
    if (bad && a.err) *a.err = 1;
    int64_t lo = a.ep_from[ep], hi = a.ep_to[ep];
    if (lo < 0 || hi > a.N || lo >= hi) { lo = i; hi = i + 1; if (a.err) *a.err = 1; }
    const int64_t q = i + a.delta[j];
    *pad = (q < lo) | (q >= hi) | bad;
    return q < lo ? lo : (q > hi - 1 ? hi - 1 : q);
}

// float rows: out[(b, j), :] = src[q, :]; one 256-thread block per (b, j) row strides over the R floats (float4 when aligned)
__global__ __launch_bounds__(256) void gather_window_f32_kernel(const float* __restrict__ src, int64_t ld_src, int R, WindowArgs a,
                                                                float* __restrict__ out, int vec) {
    const int row = blockIdx.x;
    const int b = row / a.T, j = row - b * a.T;
    bool pad;
    const int64_t q = window_source(a, b, j, &pad);
    if (threadIdx.x == 0 && a.is_pad) a.is_pad[row] = pad ? 1 : 0;
    const float* s = src + q * ld_src;
    float* o = out + (int64_t)row * R;
    if (vec) {
        for (int k = threadIdx.x * 4; k < R; k += 1024) *reinterpret_cast<f32x4*>(o + k) = *reinterpret_cast<const f32x4*>(s + k);
    } else {
        for (int k = threadIdx.x; k < R; k += 256) o[k] = s[k];
    }
}

// uint8 images stored (N, H, W, C) -> out (B, T, C, H, W) float32 = value / 255 (:355-356: .type(float32).permute(0, 3, 1, 2) / 255.0)
// grid: (ceil(H*W / 256), B*T); a thread owns one pixel: C byte loads, C coalesced float stores (one per plane)
template <int C>
__global__ __launch_bounds__(256) void gather_window_u8_kernel(const unsigned char* __restrict__ src, int HW, WindowArgs a,
                                                               float* __restrict__ out) {
    const int row = blockIdx.y;
    const int b = row / a.T, j = row - b * a.T;
    bool pad;
    const int64_t q = window_source(a, b, j, &pad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.is_pad) a.is_pad[row] = pad ? 1 : 0;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const unsigned char* s = src + (q * HW + p) * C;
    float* o = out + (int64_t)row * C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) o[(int64_t)c * HW] = __fdiv_rn((float)s[c], 255.0f);
}

// the same with four pixels per thread (H * W % 4 == 0): C aligned 4-byte loads, one float4 store per plane
template <int C>
__global__ __launch_bounds__(256) void gather_window_u8_x4_kernel(const unsigned char* __restrict__ src, int HW, WindowArgs a,
                                                                  float* __restrict__ out) {
    const int row = blockIdx.y;
    const int b = row / a.T, j = row - b * a.T;
    bool pad;
    const int64_t q = window_source(a, b, j, &pad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.is_pad) a.is_pad[row] = pad ? 1 : 0;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const unsigned* s = reinterpret_cast<const unsigned*>(src + (q * HW + p) * C);     // (q * HW + p) * C is a multiple of 4
    unsigned wds[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wds[c] = s[c];
    float* o = out + (int64_t)row * C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int byte = k * C + c;                                   // pixel k, channel c of the 4 C bytes this thread read
            v[k] = __fdiv_rn((float)((wds[byte >> 2] >> ((byte & 3) * 8)) & 0xffu), 255.0f);
        }
        *reinterpret_cast<f32x4*>(o + (int64_t)c * HW) = v;
    }
}

}  // namespace vita

using namespace vita;

extern "C" {

static int check_window(const char* who, const void* src, const int64_t* idx, int32_t B, const int64_t* delta, int32_t T, const int64_t* ep_index,
                        const int64_t* ep_from, const int64_t* ep_to, const void* out, int64_t N, int64_t E) {
    VITA_REQUIRE(B >= 0 && T >= 1, "%s: bad batch / window (B=%d, T=%d)", who, B, T);
    VITA_REQUIRE(B == 0 || (N >= 1 && E >= 1), "%s: the replay buffer needs at least one frame and one episode (N=%lld, E=%lld)", who, (long long)N, (long long)E);
    VITA_REQUIRE(B == 0 || (src && idx && delta && ep_index && ep_from && ep_to && out), "%s: null pointer", who);
    return VITA_OK;
}

int vita_gather_window_f32(const float* src, int64_t ld_src, int32_t R, const int64_t* idx, int32_t B, const int64_t* delta, int32_t T,
                           const int64_t* ep_index, const int64_t* ep_from, const int64_t* ep_to, int64_t num_frames, int64_t num_episodes,
                           float* out, uint8_t* is_pad, int32_t* err, vita_stream_t stream) {
    const char* who = "vita_gather_window_f32";
    VITA_TRY(check_window(who, src, idx, B, delta, T, ep_index, ep_from, ep_to, out, num_frames, num_episodes));
    VITA_REQUIRE(R >= 1 && ld_src >= R, "%s: bad row length / stride (R=%d, ld=%lld)", who, R, (long long)ld_src);
    if (B == 0) return VITA_OK;
    WindowArgs a{idx, B, delta, T, ep_index, ep_from, ep_to, is_pad, num_frames, num_episodes, err};
    const int vec = (R % 4 == 0) && (ld_src % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    hipLaunchKernelGGL(gather_window_f32_kernel, dim3((unsigned)B * (unsigned)T), dim3(256), 0, (hipStream_t)stream, src, ld_src, R, a, out, vec);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

int vita_gather_window_u8_nchw_f32(const uint8_t* src, int32_t H, int32_t W, int32_t C, const int64_t* idx, int32_t B, const int64_t* delta,
                                   int32_t T, const int64_t* ep_index, const int64_t* ep_from, const int64_t* ep_to, int64_t num_frames,
                                   int64_t num_episodes, float* out, uint8_t* is_pad, int32_t* err, vita_stream_t stream) {
    const char* who = "vita_gather_window_u8_nchw_f32";
    VITA_TRY(check_window(who, src, idx, B, delta, T, ep_index, ep_from, ep_to, out, num_frames, num_episodes));
    VITA_REQUIRE(H >= 1 && W >= 1 && (C == 1 || C == 3 || C == 4) && (int64_t)H * W < (int64_t(1) << 30), "%s: bad image shape (%d, %d, %d)", who, H, W, C);
    VITA_REQUIRE((int64_t)B * T <= 65535, "%s: B * T = %lld images per launch exceed 65535", who, (long long)B * T);
    if (B == 0) return VITA_OK;
    WindowArgs a{idx, B, delta, T, ep_index, ep_from, ep_to, is_pad, num_frames, num_episodes, err};
    const int HW = H * W;
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        const dim3 grid4((unsigned)((HW / 4 + 255) / 256), (unsigned)B * (unsigned)T);
        if (C == 1) hipLaunchKernelGGL((gather_window_u8_x4_kernel<1>), grid4, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
        else if (C == 3) hipLaunchKernelGGL((gather_window_u8_x4_kernel<3>), grid4, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
        else hipLaunchKernelGGL((gather_window_u8_x4_kernel<4>), grid4, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
        VITA_CHECK_HIP(hipGetLastError());
        return VITA_OK;
    }
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B * (unsigned)T);
    if (C == 1) hipLaunchKernelGGL((gather_window_u8_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
    else if (C == 3) hipLaunchKernelGGL((gather_window_u8_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
    else hipLaunchKernelGGL((gather_window_u8_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, src, HW, a, out);
    VITA_CHECK_HIP(hipGetLastError());
    return VITA_OK;
}

}  // extern "C"
