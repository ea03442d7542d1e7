// Does a latency-bound VALU chain of one wave advance only at the MFMA boundaries of its SIMD partner?
// Block = 512 threads: waves 0-3 stream dependent v_mfma_f32_32x32x2_f32 (4 accumulators round robin), waves 4-7 run CH
// dependent fma chains of length LEN each (ILP = CH), at priority PRIO.  Reports the cycles the chain waves need alone /
// beside the MFMA waves, and the MFMA waves' cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CH, int PRIO, bool SWAP = false, int GAP = 0>
__global__ __launch_bounds__(512) void k(int nmfma, int len, float a, float b, float *out, long long *cyc)
{
    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = SWAP ? (wave0 ^ 4) : wave0;      // SWAP: the chain runs in the OLDER waves 0-3, the MFMAs in waves 4-7
    const long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < 4) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < nmfma; i += 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            if (GAP) __builtin_amdgcn_s_sleep(GAP);
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else {
        __builtin_amdgcn_s_setprio(PRIO);
        float v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = a + c;
        for (int i = 0; i < len; ++i) {
#pragma unroll
            for (int c = 0; c < CH; ++c) v[c] = __builtin_fmaf(v[c], a, b);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) s += v[c];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (s == 12345.678f) *out = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
static float *d_out; static long long *d_cyc;
template <int CH, int PRIO, bool SWAP = false, int GAP = 0> void run(int nmfma, int len)
{
    hipLaunchKernelGGL((k<CH, PRIO, SWAP, GAP>), dim3(256), dim3(512), 0, 0, nmfma, len, 1.0001f, 0.5f, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<CH, PRIO, SWAP, GAP>), dim3(256), dim3(512), 0, 0, nmfma, len, 1.0001f, 0.5f, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(256 * 8);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
    printf("%s gap %d ILP %2d prio %d  mfmas %5d chain %5d : MFMA waves %9.0f cyc (%.1f/mfma)   chain waves %9.0f cyc (%.1f per level, %.2f per instr)\n", SWAP ? "chain=older" : "chain=younger", GAP, CH, PRIO, nmfma, len,
           m / 1024, nmfma ? m / 1024 / nmfma : 0.0, v / 1024, v / 1024 / len, v / 1024 / len / CH);
}
int main()
{
    CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_cyc, 1 << 16));
    printf("chain alone:\n");
    run<1, 0>(0, 2000); run<4, 0>(0, 2000); run<12, 0>(0, 2000);
    printf("MFMA alone:\n");
    run<1, 0>(4000, 0);
    printf("beside a long MFMA stream (the chain finishes first):\n");
    run<1, 0>(8000, 2000); run<1, 3>(8000, 2000);
    run<4, 0>(8000, 2000); run<4, 3>(8000, 2000);
    run<12, 0>(8000, 2000); run<12, 3>(8000, 2000);
    printf("roles swapped (chain in waves 0-3):\n");
    run<4, 0, true>(8000, 2000); run<4, 3, true>(8000, 2000); run<12, 0, true>(8000, 2000);
    printf("s_sleep N after every 4 MFMAs (chain 8000 long so that it never finishes first):\n");
    run<4, 0, false, 1>(8000, 8000); run<4, 0, false, 2>(8000, 8000); run<4, 0, false, 3>(8000, 8000); run<12, 0, false, 2>(8000, 8000); run<12, 0, true, 2>(8000, 8000); run<12, 3, false, 2>(8000, 8000);
    return 0;
}
