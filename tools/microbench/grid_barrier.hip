// Microbenchmark (not part of the library): what does a device-wide barrier INSIDE a kernel cost on gfx950, against the
// ~4.7 us floor of a dependent launch in a replayed graph?  Gate for "one cooperative launch per bottleneck unit" of the 2-D
// network: every phase writes a layer-sized tensor that blocks on OTHER XCDs read in the next phase, so the barrier carries an
// agent-scope release (L2 write-back) and acquire (L2 invalidate), like a kernel boundary does.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// fence = 1: release / acquire on the counter operations themselves (every poll of the spin loop is an acquire = an L2 invalidate);
// fence = 2: one release fence, relaxed add + relaxed polls, one acquire fence (the cheapest correct form)
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *err, int fence)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (fence == 2) {
            __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope: the block's stores are written back first
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
        }
        int spins = 0;
        while ((fence == 2 ? __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __atomic_load_n(counter, __ATOMIC_ACQUIRE)) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; ok = false; break; }  // never hang the box
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return ok;
}

// phases x (write `per_block` float4 per block, barrier, read the slice of a block half the grid away)
__global__ __launch_bounds__(256) void probe(float4 *a, float4 *b, int per_thread, int phases, unsigned *counter, int *err, float *sink, int fence)
{
    const int G = gridDim.x;
    float4 acc{0, 0, 0, 0};
    for (int ph = 0; ph < phases; ++ph) {
        float4 *dst = (ph & 1) ? b : a;
        const float4 *src = (ph & 1) ? a : b;
        for (int i = 0; i < per_thread; ++i) {
            const size_t o = ((size_t)blockIdx.x * per_thread + i) * 256 + threadIdx.x;
            dst[o] = float4{(float)ph, acc.x, 1.f, 2.f};
        }
        if (fence) {
            if (!grid_barrier(counter, (unsigned)(ph + 1) * G, err, fence)) return;
        } else {
            __syncthreads();
        }
        const int other = (blockIdx.x + G / 2 + 1) % G;
        for (int i = 0; i < per_thread; ++i) {
            const float4 v = dst[((size_t)other * per_thread + i) * 256 + threadIdx.x];
            if (fence && v.x != (float)ph) *err = 2;  // stale data = missing coherence
            acc.x += v.x; acc.y += v.y;
        }
        (void)src;
    }
    if (acc.x == -1.f) sink[0] = acc.y;
}

int main()
{
    unsigned *counter; int *err; float *sink; float4 *a, *b;
    const size_t max_f4 = (size_t)1024 * 16 * 256;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&a, max_f4 * 16)); CHECK(hipMalloc(&b, max_f4 * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grids[] = {128, 256, 304, 512, 1024};
    const int pts[] = {1, 4, 16};  // float4 per thread per phase: 4 KB, 16 KB, 64 KB per block
    for (int fence = 2; fence >= 0; --fence)
        for (int pt : pts)
            for (int G : grids) {
                float ms[2];
                for (int k = 0; k < 2; ++k) {
                    const int phases = k ? 68 : 4;
                    float best = 1e9f;
                    for (int rep = 0; rep < 5; ++rep) {
                        CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(err, 0, 4));
                        CHECK(hipEventRecord(e0));
                        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, a, b, pt, phases, counter, err, sink, fence);
                        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                        float t; CHECK(hipEventElapsedTime(&t, e0, e1));
                        best = t < best ? t : best;
                    }
                    ms[k] = best;
                }
                int h_err; CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
                printf("%s grid %4d  %3d KB/block (%5.1f MB/phase): %.2f us per phase (68 vs 4 phases)%s\n", fence == 2 ? "barrier (fences)" : fence ? "barrier (rel/acq ops)" : "no barrier", G, pt * 4,
                       (double)G * pt * 4096 / 1e6, (ms[1] - ms[0]) * 1000.0 / 64, h_err ? (h_err == 1 ? "  TIMEOUT" : "  STALE DATA") : "");
            }
    return 0;
}
