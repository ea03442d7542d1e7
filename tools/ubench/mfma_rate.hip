// mfma_rate.hip - sustained fp32 MFMA rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 (same flops per cycle on paper):
// 8 waves per CU, each a stream of MFMAs over NACC rotating accumulators.   hipcc --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k32(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float *out, int iters)
{
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
static void run(const char *name, K kern, double flops_per_mfma, int nacc, float *out)
{
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double fl = (double)blocks * 8 * iters * nacc * flops_per_mfma;
    printf("%-28s %2d accumulators: %7.3f ms  %6.1f TFLOP/s\n", name, nacc, ms, fl / ms / 1e9);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run("v_mfma_f32_32x32x2_f32", k32<1>, 4096.0, 1, out);
    run("v_mfma_f32_32x32x2_f32", k32<4>, 4096.0, 4, out);
    run("v_mfma_f32_16x16x4_f32", k16<1>, 2048.0, 1, out);
    run("v_mfma_f32_16x16x4_f32", k16<2>, 2048.0, 2, out);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 2048.0, 4, out);
    run("v_mfma_f32_16x16x4_f32", k16<8>, 2048.0, 8, out);
    return 0;
}
