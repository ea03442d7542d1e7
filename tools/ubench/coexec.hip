// Micro-benchmarks that decide the Winograd kernel structure (round 3):
//  (a) does VALU work of one wave co-execute with fp32 MFMA of another wave on the same SIMD (and inside one wave)?
//  (b) how many bytes per clock per CU does an L2-resident, chip-wide shared weight stream deliver (1 KiB wave loads)?
// Build: hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void mfma_loop(int iters, float a, float b, float *out)
{
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) *out = s;
}
__device__ __forceinline__ void valu_loop(int iters, float a, float b, float *out)
{
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v0 = __builtin_fmaf(v0, a, b); v1 = __builtin_fmaf(v1, a, b); v2 = __builtin_fmaf(v2, a, b); v3 = __builtin_fmaf(v3, a, b);
            v4 = __builtin_fmaf(v4, a, b); v5 = __builtin_fmaf(v5, a, b); v6 = __builtin_fmaf(v6, a, b); v7 = __builtin_fmaf(v7, a, b);
        }
    }
    const float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (s == 12345.678f) *out = s;
}
// mode 0: all waves MFMA.  mode 1: all waves VALU.  mode 2: waves 0-3 MFMA, 4-7 VALU (block of 512).
__global__ __launch_bounds__(512) void k_mix(int mode, int mi, int vi, float a, float b, float *out, long long *cyc)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0 || (mode == 2 && wave < 4)) mfma_loop(mi, a, b, out);
    else valu_loop(vi, a, b, out);
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
// one wave: per iteration 4 MFMAs and NV independent VALU fmas interleaved
template <int NV>
__global__ __launch_bounds__(256) void k_same(int iters, float a, float b, float *out, long long *cyc)
{
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], a, b);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], a, b);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], a, b);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) *out = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// (b) weight stream: every block walks the same `bytes`-sized buffer, wave w takes 1-KiB pieces w, w+NW, ...; optional per-block phase.
// MF = MFMAs issued per loaded piece (0: loads only).
template <int MF>
__global__ __launch_bounds__(1024) void k_stream(const float *w, size_t pieces, int rounds, int dephase, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc = {0, 0, 0, 0};
    f32x16 c0 = {0}, c1 = {0};
    const size_t start = dephase ? ((size_t)blockIdx.x * 977u) % pieces : 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        for (size_t p = wave; p < pieces; p += 4 * nw) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                size_t q = p + (size_t)u * nw + start;
                if (q >= pieces) q -= pieces;
                if (q >= pieces) q -= pieces;
                v[u] = *reinterpret_cast<const f32x4 *>(w + q * 256 + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MF == 0) acc += v[u];
                else {
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u][m & 3], v[u][(m + 1) & 3], c0, 0, 0, 0);
                        ++m;
                        if (m < MF) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u][m & 3], v[u][(m + 1) & 3], c1, 0, 0, 0);
                    }
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    if (s == 12345.678f) *out = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

static double run(void (*launch)(), int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
static float *d_out; static long long *d_cyc; static float *d_w;
static double avg_cyc(int n)
{
    std::vector<long long> h(n);
    CK(hipMemcpy(h.data(), d_cyc, n * sizeof(long long), hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += v; return s / n;
}
int main()
{
    CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_cyc, 1 << 20)); CK(hipMemset(d_cyc, 0, 1 << 20));
    const size_t WB = 8u << 20; CK(hipMalloc(&d_w, WB));
    { std::vector<float> h(WB / 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f; CK(hipMemcpy(d_w, h.data(), WB, hipMemcpyHostToDevice)); }
    const int MI = 4000, VI = 4000;   // 16000 MFMAs (1.02M cycles) ; 128000 VALU fmas (256k cycles at 2/clk... measured)
    printf("== (a) co-execution, 256 blocks x 512 threads (2 waves/SIMD)\n");
    static int g_mode, g_mi, g_vi;
    auto L = []() { hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, g_mode, g_mi, g_vi, 1.0001f, 0.5f, d_out, d_cyc); };
    struct { int mode, mi, vi; const char *name; } cs[] = {
        {0, MI, 0, "all 8 waves MFMA (2/SIMD)"}, {1, 0, VI, "all 8 waves VALU"}, {2, MI, VI, "4 MFMA + 4 VALU waves"},
        {2, MI, 0, "4 MFMA waves + 4 idle"}, {2, 0, VI, "4 idle + 4 VALU waves"}, {2, MI, 2 * VI, "4 MFMA + 4 VALU(2x)"}, {2, MI, 4 * VI, "4 MFMA + 4 VALU(4x)"} };
    for (auto &c : cs) {
        g_mode = c.mode; g_mi = c.mi; g_vi = c.vi;
        const double ms = run(L);
        printf("%-32s %8.3f ms   avg wave cycles %10.0f\n", c.name, ms, avg_cyc(256 * 8));
    }
    printf("== (a2) same wave, 4 MFMA + 4*NV VALU per iteration, 256 blocks x 256 threads (1 wave/SIMD), %d iters\n", MI);
    static int g_it = MI;
#define SAME(NV) { auto L2 = []() { hipLaunchKernelGGL(k_same<NV>, dim3(256), dim3(256), 0, 0, g_it, 1.0001f, 0.5f, d_out, d_cyc); }; \
        const double ms = run(L2); printf("NV=%2d  %8.3f ms  cycles/iter %8.1f\n", NV, ms, avg_cyc(256 * 4) / MI); }
    SAME(0) SAME(4) SAME(8) SAME(16) SAME(24) SAME(32)
    printf("== (b) shared L2-resident weight stream, 256 blocks; bytes/clk/CU from wave cycles\n");
    static size_t g_pieces; static int g_rounds, g_deph, g_nt;
#define STREAM(MF, NT, BYTES, DEPH) { g_pieces = (size_t)(BYTES) / 1024; g_rounds = (int)((64u << 20) / (BYTES)); g_deph = DEPH; g_nt = NT; \
        auto L3 = []() { hipLaunchKernelGGL(k_stream<MF>, dim3(256), dim3(g_nt), 0, 0, d_w, g_pieces, g_rounds, g_deph, d_out, d_cyc); }; \
        const double ms = run(L3, 3); const double cyc = avg_cyc(256 * 16) * 16.0 / (NT / 64); \
        const double bytes = (double)g_pieces * 1024 * g_rounds; \
        printf("MF=%d waves=%2d buf=%4zu KB dephase=%d : %7.3f ms  %6.1f B/clk/CU (cycles)  %6.2f TB/s chip\n", MF, NT / 64, (size_t)(BYTES) >> 10, DEPH, ms, bytes / cyc, bytes * 256 / ms * 1e-9); }
    STREAM(0, 256, 1u << 20, 0) STREAM(0, 512, 1u << 20, 0) STREAM(0, 1024, 1u << 20, 0)
    STREAM(0, 512, 1u << 20, 1) STREAM(0, 1024, 1u << 20, 1)
    STREAM(0, 512, 4u << 20, 0) STREAM(0, 512, 4u << 20, 1) STREAM(0, 512, 256u << 10, 0) STREAM(0, 512, 256u << 10, 1)
    STREAM(2, 512, 1u << 20, 0) STREAM(2, 512, 1u << 20, 1) STREAM(4, 512, 1u << 20, 0) STREAM(4, 512, 1u << 20, 1)
    STREAM(8, 512, 1u << 20, 0) STREAM(8, 512, 1u << 20, 1)
    return 0;
}
