// mfma_bf16_probe.hip - how does v_mfma_f32_32x32x16_bf16 (and 16x16x32) add its K products and the C input?
// Reads cases (tools/ubench/gen_bf16_cases.py): N x { a[32] bf16, b[32] bf16, c f32 }, slot s = 8 (lane / 32) + e for the 32x32x16 form
// (s < 16), s = 8 (lane / 16) + e for 16x16x32; case n sits on the diagonal (n % 32, n % 32) of MFMA number n / 32.  Writes d[N] per form.
// hipcc --offload-arch=gfx950 -O2 mfma_bf16_probe.hip -o mfma_bf16_probe && ./mfma_bf16_probe cases.bin out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Case { uint16_t a[32], b[32]; float c; };
union Frag { bf16x8 v; uint16_t u[8]; };

__global__ void probe32(const Case *cs, int N, float *d)
{
    const int lane = threadIdx.x, m = blockIdx.x, i = lane & 31, h = lane >> 5;
    const int n = m * 32 + i;
    Frag fa, fb;
    for (int e = 0; e < 8; ++e) { fa.u[e] = n < N ? cs[n].a[8 * h + e] : 0; fb.u[e] = n < N ? cs[n].b[8 * h + e] : 0; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;       // this lane holds D[row][col = i]
        acc[r] = (row == i && n < N) ? cs[n].c : 0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row == i && n < N) d[n] = acc[r];
    }
}
__global__ void probe16(const Case *cs, int N, float *d)
{
    const int lane = threadIdx.x, m = blockIdx.x, i = lane & 15, q = lane >> 4;
    const int n = m * 16 + i;
    Frag fa, fb;
    for (int e = 0; e < 8; ++e) { fa.u[e] = n < N ? cs[n].a[8 * q + e] : 0; fb.u[e] = n < N ? cs[n].b[8 * q + e] : 0; }
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = (4 * q + r == i && n < N) ? cs[n].c : 0.f;      // lane holds D[4 q + r][col = i]
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) if (4 * q + r == i && n < N) d[n] = acc[r];
}
// chained: two MFMAs on one accumulator (cases n and n + N/2 share a diagonal slot): is the hand-over a plain fp32 value?
__global__ void probe32_chain(const Case *cs, int N2, float *d)
{
    const int lane = threadIdx.x, m = blockIdx.x, i = lane & 31, h = lane >> 5;
    const int n = m * 32 + i;
    Frag fa, fb, ga, gb;
    for (int e = 0; e < 8; ++e) {
        fa.u[e] = n < N2 ? cs[n].a[8 * h + e] : 0; fb.u[e] = n < N2 ? cs[n].b[8 * h + e] : 0;
        ga.u[e] = n < N2 ? cs[n + N2].a[8 * h + e] : 0; gb.u[e] = n < N2 ? cs[n + N2].b[8 * h + e] : 0;
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        acc[r] = (row == i && n < N2) ? cs[n].c : 0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.v, gb.v, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row == i && n < N2) d[n] = acc[r];
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s cases.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("cases"); return 2; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    const int N = (int)(sz / sizeof(Case));
    std::vector<Case> cs(N);
    if (fread(cs.data(), sizeof(Case), N, f) != (size_t)N) return 2;
    fclose(f);
    Case *dc; float *dd;
    hipMalloc(&dc, (size_t)N * sizeof(Case)); hipMalloc(&dd, (size_t)N * 3 * sizeof(float));
    hipMemcpy(dc, cs.data(), (size_t)N * sizeof(Case), hipMemcpyHostToDevice);
    hipMemset(dd, 0, (size_t)N * 3 * sizeof(float));
    hipLaunchKernelGGL(probe32, dim3((N + 31) / 32), dim3(64), 0, 0, dc, N, dd);
    hipLaunchKernelGGL(probe16, dim3((N + 15) / 16), dim3(64), 0, 0, dc, N, dd + N);
    hipLaunchKernelGGL(probe32_chain, dim3((N / 2 + 31) / 32), dim3(64), 0, 0, dc, N / 2, dd + 2 * (size_t)N);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<float> out((size_t)N * 3);
    hipMemcpy(out.data(), dd, out.size() * 4, hipMemcpyDeviceToHost);
    FILE *g = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), g);
    fclose(g);
    printf("mfma_bf16_probe: %d cases -> %s\n", N, argv[2]);
    return 0;
}
