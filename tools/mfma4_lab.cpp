// Layout check of v_mfma_f32_4x4x1_16B_f32 on gfx950: lane l supplies A[i = l % 4] and B[j = l % 4] of block l / 4;
// D: lane l, register v holds D[i = v][j = l % 4] of block l / 4 ?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = c[v];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const int blk = l / 4, j = l % 4, i = v;
            const float want = ha[blk * 4 + i] * hb[blk * 4 + j];
            if (hd[l * 4 + v] != want) { if (bad < 8) printf("lane %d v %d: got %g want %g\n", l, v, hd[l * 4 + v], want); ++bad; }
        }
    printf("layout %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed", bad);
    return bad != 0;
}
