// Microbenchmark (not product code): how many blocks with N bytes of static LDS does a CU of this device hold?  Asked of the runtime
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) and MEASURED: a kernel whose blocks record their start on the wall clock and then wait
// ~20 us - the number of blocks that start within the first microseconds is what was resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(unsigned long long *starts, int spin)
{
    __shared__ unsigned int lds[BYTES / 4];
    if (threadIdx.x == 0) starts[blockIdx.x] = wall_clock64();
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { lds[(threadIdx.x * 7) % (BYTES / 4)] += 1; }
    if (lds[threadIdx.x] == 0xffffffffu) starts[0] = 0;
}

template <int BYTES, int THREADS>
void run(const char *what)
{
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe<BYTES, THREADS>, THREADS, 0);
    const int blocks = 256 * 6;
    unsigned long long *d = nullptr;
    (void)hipMalloc(&d, blocks * 8);
    (void)hipMemset(d, 0, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<BYTES, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, 2000 /* 20 us at 100 MHz */);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto t : h) first += (t - t0) < 500;  // started within 5 us of the first block
    printf("%-26s LDS %6d B, %4d threads: runtime says %d blocks per CU; measured %d of %d blocks resident at once = %.2f per CU\n", what, BYTES, THREADS, nb, first, blocks,
           first / 256.0);
    (void)hipFree(d);
}

int main()
{
    run<47112, 512>("accumulate (kept)");
    run<51208, 512>("accumulate + LDS list");
    run<49152, 256>("48 KB");
    run<51200, 256>("vortex_tail (50 KB)");
    run<53248, 256>("entry1x1 (52 KB)");
    run<54272, 256>("53 KB");
    run<55296, 256>("54 KB");
    run<40960, 256>("40 KB");
    run<32768, 256>("32 KB");
    run<24576, 256>("conv_f16x3 (24 KB)");
    run<17024, 256>("pool_pyramid");
    run<65536, 256>("64 KB");
    return 0;
}
