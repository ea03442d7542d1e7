// mfma16_layout.hip - operand / result layout of v_mfma_f32_16x16x1_f32 (4 blocks), as kernels_wino_c128.hip assumes it:
//   A: lane l = row l % 16 of block l / 16;  B: lane l = column l % 16 of block l / 16;
//   D: register 4 b + r of lane l = D_b[row 4 (l / 16) + r][column l % 16].
// Run 1: a = lane + 1, b = 1 -> D = the A lane that fed the element; run 2: a = 1, b = lane + 1 -> the B lane.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma16_layout.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float *out)
{
    const int l = threadIdx.x;
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    const f32x16 da = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    const f32x16 db = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        out[(r * 64 + l) * 2] = da[r];
        out[(r * 64 + l) * 2 + 1] = db[r];
    }
}
int main()
{
    float *d, h[16 * 64 * 2];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r)
        for (int l = 0; l < 64; ++l) {
            const int b = r >> 2, rr = r & 3;
            const int ea = 16 * b + 4 * (l >> 4) + rr + 1, eb = 16 * b + (l & 15) + 1;
            if ((int)h[(r * 64 + l) * 2] != ea || (int)h[(r * 64 + l) * 2 + 1] != eb) ++bad;
        }
    printf("mfma_f32_16x16x1 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad)
        for (int r = 0; r < 16; ++r) {
            printf("reg %2d:", r);
            for (int l = 0; l < 64; l += 1) printf(" %d/%d", (int)h[(r * 64 + l) * 2] - 1, (int)h[(r * 64 + l) * 2 + 1] - 1);
            printf("\n");
        }
    return bad ? 1 : 0;
}
