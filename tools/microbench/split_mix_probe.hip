// Probe (not product, not a test): the v_fma_mixlo/hi_f16 form of split_f16 against the plain C++ split, bit for bit.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_f16(const f32x4 &a, const f32x4 &b, f16x8 &hi, f16x8 &lo)
{
    const f32x8 x = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    hi = __builtin_convertvector(x, f16x8);
    const u32x4 h = __builtin_bit_cast(u32x4, hi);
    u32x4 l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned r;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h[i]), "v"(x[2 * i]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h[i]), "v"(x[2 * i + 1]));
        l[i] = r;
    }
    lo = __builtin_bit_cast(f16x8, l);
}
__global__ void k(const f32x4 *in, f16x8 *out)
{
    f16x8 h, l;
    split_f16(in[2 * threadIdx.x], in[2 * threadIdx.x + 1], h, l);
    out[2 * threadIdx.x] = h;
    out[2 * threadIdx.x + 1] = l;
}
int main()
{
    const int n = 256;
    f32x4 *in; f16x8 *out;
    hipMallocManaged(&in, n * 2 * sizeof(f32x4)); hipMallocManaged(&out, n * 2 * sizeof(f16x8));
    float *f = (float *)in;
    unsigned s = 12345;
    for (int i = 0; i < n * 8; ++i) { s = s * 1664525u + 1013904223u; f[i] = ((int)(s >> 8) - (1 << 23)) * (i % 7 == 0 ? 1e-9f : (i % 5 == 0 ? 3e-3f : 1.3e-5f)); }
    f[0] = 0.f; f[1] = -0.f; f[2] = 65504.f; f[3] = 6e-8f; f[4] = 1e-10f; f[5] = 70000.f;
    k<<<1, n>>>(in, out);
    hipDeviceSynchronize();
    int bad = 0;
    const _Float16 *o = (const _Float16 *)out;
    for (int t = 0; t < n; ++t)
        for (int j = 0; j < 8; ++j) {
            const float x = f[t * 8 + j];
            const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
            const _Float16 gh = o[t * 16 + j], gl = o[t * 16 + 8 + j];
            if (__builtin_memcmp(&h, &gh, 2) || __builtin_memcmp(&l, &gl, 2)) { if (bad < 5) printf("x=%g h %g/%g l %g/%g\n", x, (float)h, (float)gh, (float)l, (float)gl); ++bad; }
        }
    printf("bad=%d\n", bad);
    return bad != 0;
}
