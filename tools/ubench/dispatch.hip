// dispatch.hip - what does a round of workgroup dispatch cost when a CU holds ONE block (151 KB of LDS, 512 threads)?
// Launches N empty blocks (a store per block so that the wave has something to retire) and N blocks that spin for a fixed time.
// hipcc --offload-arch=gfx950 dispatch.hip -o dispatch && ./dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"

__global__ __launch_bounds__(512) void blk(float *out, int spin_us)
{
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 1.f;
    if (spin_us) {
        const unsigned long long t0 = wall_clock64();                          // constant 100 MHz counter
        while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) {}
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = smem[0];
}

int main()
{
    const int blocks = 5184;
    float *out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipFuncSetAttribute((const void *)blk, hipFuncAttributeMaxDynamicSharedMemorySize, 151552);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 5, 20, 80}) {
        for (int lds : {151552, 32768}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(blk, dim3(blocks), dim3(512), lds, 0, out, spin);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(blk, dim3(blocks), dim3(512), lds, 0, out, spin);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 100.0;           // per launch
            const double rounds = lds > 80000 ? blocks / 256.0 : blocks / (256.0 * 4);
            printf("spin %2d us, LDS %6d B: %8.1f us per launch = %6.2f us per round of blocks (%.2f rounds); overhead per round %.2f us\n",
                   spin, lds, us, us / rounds, rounds, us / rounds - spin);
        }
    }
    return 0;
}
