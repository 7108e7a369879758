// Probe (not product): lane layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(float *out)
{
    const int l = threadIdx.x;
    // A[b][i] = 100*b + i ; B[b][j] = 1000 + j + 10*b  -> D[b][i][j] = A*B
    const float a = 100.0f * (l / 4) + (l % 4) + 1.0f;
    const float b = 1000.0f + (l % 4) + 10.0f * (l / 4);
    f32x4 c{0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void rate_kernel(float *out, int iters, long long *cycles)
{
    f32x4 c0{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

__global__ void rate16_kernel(float *out, int iters, long long *cycles)
{
    f32x4 c0{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main()
{
    float *d; long long *dc;
    hipMalloc(&d, 64 * 4 * sizeof(float)); hipMalloc(&dc, 8);
    layout_kernel<<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4, i = r;  // hypothesis: lane (blk, j), register = row i
            const float want = (100.0f * blk + i + 1.0f) * (1000.0f + j + 10.0f * blk);
            if (h[l * 4 + r] != want) ++bad;
        }
    printf("layout hypothesis D[lane=(blk,j)][reg=i] = A[blk][i]*B[blk][j]: %s (%d mismatches); lane5 regs %g %g %g %g\n",
           bad ? "WRONG" : "OK", bad, h[20], h[21], h[22], h[23]);
    long long c;
    rate_kernel<<<1, 64>>>(d, 10000, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("4x4x1_16b: %.2f cycles per MFMA (8 independent accumulators, one wave)\n", (double)c / (10000.0 * 8));
    rate16_kernel<<<1, 64>>>(d, 10000, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4:   %.2f cycles per MFMA (4 independent accumulators, one wave)\n", (double)c / (10000.0 * 4));
    return 0;
}
