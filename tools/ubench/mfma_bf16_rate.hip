// mfma_bf16_rate.hip - cycles per instruction of the bf16 MFMA forms a Winograd M phase could use (one wave per SIMD and two, s_memtime
// around 4000 back-to-back instructions on 4 rotating accumulators), and the same stream with VALU work interleaved in the SAME wave.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int FORM, int NV>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters)
{
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    union { bf16x8 v8; bf16x4 v4[2]; s16x4 s4[2]; unsigned short u[8]; } a, b;
    for (int e = 0; e < 8; ++e) { a.u[e] = 0x3f80 + threadIdx.x + e; b.u[e] = 0x3e00 + 3 * threadIdx.x + e; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (FORM == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v8, b.v8, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a.s4[0], b.s4[0], acc[q], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], 1.0001f, 0.5f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += acc[q][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K>
static void run(const char *name, K kern, int threads, float *out, long long *cyc)
{
    const int iters = 1000, blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; const int nw = threads / 64; for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) s += h[b * 8 + w];
    printf("%-52s %d waves/SIMD: %7.1f cycles per MFMA per wave\n", name, threads / 256, s / (blocks * nw) / (iters * 4));
}

int main()
{
    float *out; long long *cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
    run("v_mfma_f32_32x32x16_bf16", k<0, 0>, 256, out, cyc);
    run("v_mfma_f32_32x32x16_bf16", k<0, 0>, 512, out, cyc);
    run("v_mfma_f32_32x32x8_bf16_1k", k<1, 0>, 256, out, cyc);
    run("v_mfma_f32_32x32x8_bf16_1k", k<1, 0>, 512, out, cyc);
    run("32x32x16 + 4 v_fma per MFMA (same wave)", k<0, 4>, 256, out, cyc);
    run("32x32x16 + 8 v_fma per MFMA (same wave)", k<0, 8>, 256, out, cyc);
    run("32x32x16 + 8 v_fma per MFMA (same wave)", k<0, 8>, 512, out, cyc);
    run("32x32x16 + 16 v_fma per MFMA (same wave)", k<0, 16>, 512, out, cyc);
    run("32x32x8_1k + 4 v_fma per MFMA (same wave)", k<1, 4>, 256, out, cyc);
    run("32x32x8_1k + 8 v_fma per MFMA (same wave)", k<1, 8>, 512, out, cyc);
    return 0;
}
