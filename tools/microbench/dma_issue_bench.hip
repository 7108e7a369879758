// What does it cost a wave to ASK for data on gfx950?  (round 4: the persistent LDS-resident kernels - dense_chain_kernel,
// vortex_branch_kernel - lose 20-35 % of every step to the waves that fetch the next window, see profiles/r04_pair_experiments.txt.)
// One block of 16 waves per CU (150 KB of LDS requested).  `nm` waves run an MFMA + ds_read_b128 loop (what the computing waves of
// those kernels do), `nl` waves each issue `nops` memory instructions of one kind back to back and stamp s_memtime before the
// first, after the last has been ISSUED, and after all have RETURNED:
//   kind 0  global_load_lds_dwordx4 (LDS-DMA, 1 KB per instruction)
//   kind 1  global_load_dwordx4 into registers
// from a footprint that stays in the L2 (hit = 1) or from 1 GB at pseudo-random 1-KB offsets (hit = 0), at wave priority `prio`.
// Prints cycles per instruction to issue and the time until everything returned (medians over the blocks).
//   hipcc --offload-arch=gfx950 -O3 -o dma_issue_bench.exe dma_issue_bench.hip ; ./dma_issue_bench.exe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kMaxOps = 16;

template <int KIND, int NOPS>
__global__ __launch_bounds__(1024) void issue_kernel(const f32x4 *src, size_t span_f4, int nm, int nl, int prio, int mfma_iters, long long *out, float *sink)
{
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 1024) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    if (wave < nm) {  // a computing wave: three weight reads + two operand reads + five MFMAs per round, like chain_mac_t
        f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        for (int it = 0; it < mfma_iters; ++it) {
            const int o = ((it * 5 + wave) * 64 + lane) & 4095;
            const f16x8 w0 = __builtin_bit_cast(f16x8, lds[o]), w1 = __builtin_bit_cast(f16x8, lds[o + 64]), w2 = __builtin_bit_cast(f16x8, lds[o + 128]);
            const f16x8 xa = __builtin_bit_cast(f16x8, lds[4096 + o]), xb = __builtin_bit_cast(f16x8, lds[4096 + o + 64]);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, xb, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, xa, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (acc[0][0] + acc[1][1] + acc[2][2] == 12345.f) sink[tid] = acc[0][0];
        return;
    }
    if (wave >= nm + nl) return;
    if (prio) __builtin_amdgcn_s_setprio(3);
    // let the computing waves get going
    __builtin_amdgcn_s_sleep(64);
    const int lw = wave - nm;
    unsigned h = (blockIdx.x * 16u + lw) * 2654435761u + 12345u;
    f32x4 stg[NOPS];
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < NOPS; ++k) {
        h = h * 1664525u + 1013904223u;
        const size_t off = ((size_t)(h >> 4) % (span_f4 / 64)) * 64 + lane;  // a 1-KB piece somewhere in the footprint
        if (KIND == 0)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + off),
                                             (void __attribute__((address_space(3))) *)(lds + 8192 + (lw * NOPS + k) * 64), 16, 0, 0);
        else
            stg[k] = src[off];
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = (long long)__builtin_amdgcn_s_memtime();
    if (KIND == 1) {
        f32x4 s = stg[0];
#pragma unroll
        for (int k = 1; k < NOPS; ++k) s += stg[k];
        if (s[0] == 12345.f) sink[tid] = s[1];
    }
    if (lane == 0 && lw == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = t2 - t0;
    }
}

template <int KIND, int NOPS>
static void run(const f32x4 *src, size_t span_f4, int nm, int nl, int prio, long long *out, float *sink, const char *what)
{
    const int blocks = 256, lds_bytes = 150 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&issue_kernel<KIND, NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    std::vector<long long> h(blocks * 2);
    std::vector<double> iss, ret;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((issue_kernel<KIND, NOPS>), dim3(blocks), dim3(1024), lds_bytes, 0, src, span_f4, nm, nl, prio, 4000, out, sink);
        CHECK(hipDeviceSynchronize());
        if (rep == 0) continue;
        CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int b = 0; b < blocks; ++b) { iss.push_back((double)h[2 * b] / NOPS); ret.push_back((double)h[2 * b + 1]); }
    }
    std::sort(iss.begin(), iss.end());
    std::sort(ret.begin(), ret.end());
    printf("%-34s ops/wave %2d  computing waves %2d  asking waves %2d  prio %d : issue %7.1f cycles per instruction (p90 %7.1f), all returned after %7.0f (p90 %7.0f)\n",
           what, NOPS, nm, nl, prio, iss[iss.size() / 2], iss[iss.size() * 9 / 10], ret[ret.size() / 2], ret[ret.size() * 9 / 10]);
}

int main()
{
    const size_t big = (size_t)1 << 26;  // float4: 1 GB
    f32x4 *src;
    long long *out;
    float *sink;
    CHECK(hipMalloc(reinterpret_cast<void **>(&src), big * sizeof(f32x4)));
    CHECK(hipMemset(src, 0, big * sizeof(f32x4)));
    CHECK(hipMalloc(reinterpret_cast<void **>(&out), 512 * sizeof(long long)));
    CHECK(hipMalloc(reinterpret_cast<void **>(&sink), 1024 * sizeof(float)));
    const size_t small = 1 << 14;  // 256 KB: stays in every L2
    for (int hit = 1; hit >= 0; --hit) {
        const size_t span = hit ? small : big;
        const char *d = hit ? "LDS-DMA, L2 hits" : "LDS-DMA, beyond L2";
        const char *r = hit ? "plain loads, L2 hits" : "plain loads, beyond L2";
        // nobody computes: the bare cost
        run<0, 2>(src, span, 0, 1, 0, out, sink, d);
        run<0, 4>(src, span, 0, 1, 0, out, sink, d);
        run<0, 8>(src, span, 0, 1, 0, out, sink, d);
        run<0, 16>(src, span, 0, 1, 0, out, sink, d);
        run<1, 4>(src, span, 0, 1, 0, out, sink, r);
        run<1, 16>(src, span, 0, 1, 0, out, sink, r);
        // sixteen waves ask at once (what the chain kernel does at the top of a step: 3-4 pieces per wave)
        run<0, 4>(src, span, 0, 16, 0, out, sink, d);
        run<1, 4>(src, span, 0, 16, 0, out, sink, r);
        // two loaders beside fourteen computing waves
        run<0, 8>(src, span, 14, 2, 0, out, sink, d);
        run<0, 8>(src, span, 14, 2, 1, out, sink, d);
        run<1, 8>(src, span, 14, 2, 0, out, sink, r);
        run<1, 8>(src, span, 14, 2, 1, out, sink, r);
        run<1, 16>(src, span, 14, 2, 1, out, sink, r);
    }
    return 0;
}
