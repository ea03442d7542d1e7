// mfma16.hip - in which order does v_mfma_f32_16x16x4_f32 add its four k-products to the accumulator?
// C[16x16] = sum_k A[16xK] B[Kx16] with K = 64 (16 instructions); compared bit for bit with candidate orders on the host.
// hipcc --offload-arch=gfx950 mfma16.hip -o mfma16 && ./mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k16(const float *A, const float *B, float *C, int K)
{
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;      // A: lane holds A[row r][k = 4 step + kq]; B: B[k = 4 step + kq][col r]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + 4 * s + kq], B[(4 * s + kq) * 16 + r], acc, 0, 0, 0);
    // C layout: lane (col = lane & 15, rows 4 (lane >> 4) + e)
    for (int e = 0; e < 4; ++e) C[(4 * (lane >> 4) + e) * 16 + (lane & 15)] = acc[e];
}

int main()
{
    const int K = 64;
    float hA[16 * K], hB[K * 16], hC[256];
    srand(1);
    for (auto &v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : hB) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_pair = 0, bad_tree = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float seq = 0.f, pr = 0.f, tr = 0.f;
            for (int s = 0; s < K / 4; ++s) {
                float p[4];
                for (int q = 0; q < 4; ++q) p[q] = hA[i * K + 4 * s + q] * hB[(4 * s + q) * 16 + j];
                for (int q = 0; q < 4; ++q) seq = fmaf(hA[i * K + 4 * s + q], hB[(4 * s + q) * 16 + j], seq);       // one fmaf chain, k ascending
                pr = pr + ((p[0] + p[1]) + (p[2] + p[3]));                                                         // rounded products, tree, then accumulate
                tr = fmaf(hA[i * K + 4 * s + 3], hB[(4 * s + 3) * 16 + j], fmaf(hA[i * K + 4 * s + 2], hB[(4 * s + 2) * 16 + j],
                     fmaf(hA[i * K + 4 * s + 1], hB[(4 * s + 1) * 16 + j], fmaf(hA[i * K + 4 * s], hB[(4 * s) * 16 + j], tr))));
            }
            const float c = hC[i * 16 + j];
            bad_seq += memcmp(&c, &seq, 4) != 0;
            bad_pair += memcmp(&c, &pr, 4) != 0;
            bad_tree += memcmp(&c, &tr, 4) != 0;
        }
    printf("v_mfma_f32_16x16x4_f32 vs host: fmaf chain k ascending: %d / 256 differ; product tree: %d; (same chain, nested form): %d\n", bad_seq, bad_pair, bad_tree);
    return 0;
}
