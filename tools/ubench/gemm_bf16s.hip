// gemm_bf16s.hip - stand-alone driver of femasr_amd/csrc/kernels_gemm_bf16.hip (the split-bf16 fp32-grade GEMM of the Swin linears):
// timing at the network's shapes, and a bit-for-bit check of sampled outputs against the host restatement of the instruction's
// arithmetic (the model of tools/ubench/fit_bf16_model.py), with the fp64 error of this GEMM and of the fp32 fmaf chain beside it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt gemm_bf16s.hip -o gemm_bf16s && ./gemm_bf16s
#include "../../femasr_amd/csrc/kernels_gemm_bf16.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdarg>
#include <vector>
#include <random>

int femasr_set_error(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n");
    return code;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#include "mfma_bf16_model.h"
struct Split { std::vector<uint16_t> p[3]; };
static void split_all(const std::vector<float> &x, Split &s)
{
    for (int i = 0; i < 3; ++i) s.p[i].resize(x.size());
    for (size_t i = 0; i < x.size(); ++i) split3(x[i], s.p[0][i], s.p[1][i], s.p[2][i]);
}
// one output: rows of the split planes (K contiguous)
static float model_dot(const Split &A, size_t arow, const Split &W, size_t wrow, int K)
{
    float hi = 0.f, lo = 0.f;
    static const int TA[5] = {2, 0, 1, 1, 0}, TB[5] = {0, 2, 1, 0, 1};
    for (int k = 0; k < K; k += 16) {
        for (int t = 0; t < 5; ++t) {
            lo = mfma_dot8(lo, &A.p[TA[t]][arow + k], &W.p[TB[t]][wrow + k]);
            lo = mfma_dot8(lo, &A.p[TA[t]][arow + k + 8], &W.p[TB[t]][wrow + k + 8]);
        }
        hi = mfma_dot8(hi, &A.p[0][arow + k], &W.p[0][wrow + k]);
        hi = mfma_dot8(hi, &A.p[0][arow + k + 8], &W.p[0][wrow + k + 8]);
    }
    return hi + lo;
}
static float host_gelu(float x) { return (float)(0.5 * (double)x * (1.0 + erf((double)x * 0.70710678118654752440))); }

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    struct Shape { const char *name; int M, N, K, act, res; } shapes[] = {
        {"qkv   256->768 ", 82944, 768, 256, 0, 0}, {"proj  256->256 +res", 82944, 256, 256, 0, 1}, {"fc1   256->1024 gelu", 82944, 1024, 256, 1, 0},
        {"fc2  1024->256 +res", 82944, 256, 1024, 0, 1}, {"bq    256->512 ", 82944, 512, 256, 0, 0},
        {"qkv   M=31104", 31104, 768, 256, 0, 0}, {"fc2   M=31104", 31104, 256, 1024, 0, 1}, {"ragged M=5000 N=200", 5000, 200, 64, 0, 1},
    };
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hr((size_t)M * N);
        for (auto &v : hA) v = nd(rng);
        if (getenv("GS_ZERO")) { const float z = (float)atof(getenv("GS_ZERO")); for (auto &v : hA) v = z; }      // data-dependent power: constant operands
        const float ws = 1.f / sqrtf((float)K);
        for (auto &v : hW) v = nd(rng) * ws;
        for (auto &v : hb) v = nd(rng) * 0.1f;
        for (auto &v : hr) v = nd(rng);
        float *dA, *dW, *db, *dr, *dout; void *dWp;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
        CK(hipMalloc(&dr, hr.size() * 4)); CK(hipMalloc(&dout, hr.size() * 4));
        CK(hipMalloc(&dWp, femasr_packed_weight_bf16s_bytes(N, K)));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dout, 0xff, hr.size() * 4));
        if (femasr_repack_k1_bf16s(0, dW, N, K, dWp)) return 1;
        femasr_conv_args a{};
        a.in = dA; a.bias = db; a.out = dout; a.res1 = sh.res ? dr : nullptr; a.B = 1; a.H = M; a.W = 1; a.Ho = M; a.Wo = 1; a.Cin = K; a.Cout = N;
        a.ksz = 1; a.stride = 1; a.pad = 0; a.act = sh.act ? FEMASR_ACT_GELU : FEMASR_ACT_NONE; a.prologue = FEMASR_PRO_NONE;
        int variant; double flops;
        if (femasr_gemm_bf16s_launch(0, &a, dWp, &variant, &flops)) return 1;
        CK(hipDeviceSynchronize());
        { int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g_gsv[variant].kern, 256, GS_LDS_BYTES)); if (&sh == &shapes[0]) printf("resident blocks per CU: %d\n", nb); }
        const int warm = getenv("GS_WARM") ? atoi(getenv("GS_WARM")) : 0;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_of = [&](const femasr_conv_args &q) {
            for (int i = 0; i < warm; ++i) femasr_gemm_bf16s_launch(0, &q, dWp, nullptr, nullptr);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) femasr_gemm_bf16s_launch(0, &q, dWp, nullptr, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            return t / reps;
        };
        const float ms = time_of(a);
        if (getenv("GS_NOVERIFY")) printf("%-22s M=%6d  %8.1f us  %7.1f TF(fp32-equivalent)\n", sh.name, M, ms * 1e3, flops / ms * 1e-9);
        if (getenv("GS_NOVERIFY")) { hipFree(dA); hipFree(dW); hipFree(db); hipFree(dr); hipFree(dout); hipFree(dWp); continue; }
        std::vector<float> hout(hr.size());
        CK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
        // verification on sampled rows (all columns of 24 rows incl. the last ones)
        Split sW; split_all(hW, sW);
        int bad = 0, checked = 0; double e_split = 0, e_chain = 0, scale = 0; int bad_print = 0;
        std::vector<int> rows;
        for (int i = 0; i < 4; ++i) rows.push_back((int)(rng() % (unsigned)M));
        rows.push_back(0); rows.push_back(M - 1); rows.push_back(M - 2); rows.push_back(127 < M ? 127 : 0); rows.push_back(128 < M ? 128 : 0);
        for (int r : rows) {
            std::vector<float> xa(hA.begin() + (size_t)r * K, hA.begin() + (size_t)(r + 1) * K);
            Split sA; split_all(xa, sA);
            for (int n = 0; n < N; ++n) {
                float v = model_dot(sA, 0, sW, (size_t)n * K, K) + hb[n];
                double ref = 0; float ch = 0.f; double sabs = 0;
                for (int k = 0; k < K; ++k) { ref += (double)xa[k] * (double)hW[(size_t)n * K + k]; ch = fmaf(xa[k], hW[(size_t)n * K + k], ch); sabs += fabs((double)xa[k] * (double)hW[(size_t)n * K + k]); }
                const float got = hout[(size_t)r * N + n];
                if (!sh.act) {
                    if (sh.res) v = v + hr[(size_t)r * N + n];
                    if (f2u(v) != f2u(got)) { ++bad; if (bad_print++ < 5) printf("   mismatch row %d col %d: model %.9g gpu %.9g\n", r, n, v, got); }
                    const double pre = (double)got - (sh.res ? (double)hr[(size_t)r * N + n] : 0.0) - (double)hb[n];
                    e_split = fmax(e_split, fabs(pre - ref)); e_chain = fmax(e_chain, fabs((double)ch - ref)); scale = fmax(scale, sabs);
                } else {
                    const float g = host_gelu(v);             // (the device GELU is the polynomial of detmath.h: compare loosely)
                    if (fabsf(g - got) > 2e-6f * (1.f + fabsf(g))) { ++bad; if (bad_print++ < 5) printf("   gelu mismatch row %d col %d: %.9g vs %.9g\n", r, n, g, got); }
                }
                ++checked;
            }
        }
        printf("%-22s M=%6d  %8.1f us  %7.1f TF(fp32-equivalent)  blocks %d | %d / %d sampled outputs differ from the host restatement%s",
               sh.name, M, ms * 1e3, flops / ms * 1e-9, ((M + 127) / 128) * ((N + 127) / 128), bad, checked, sh.act ? " (GELU: tolerance)" : "");
        if (!sh.act) printf(" | max err vs fp64: split %.3g, fp32 chain %.3g (sum|ab| %.3g)", e_split, e_chain, scale);
        printf("\n");
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dr); hipFree(dout); hipFree(dWp);
    }
    return 0;
}
