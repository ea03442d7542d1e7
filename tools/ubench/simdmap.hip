// which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void k(unsigned *out)
{
    const unsigned id = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d; hipMalloc(&d, 4096 * 4); 
    for (int nt : {512, 768, 1024}) {
        hipMemset(d, 0xff, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(8), dim3(nt), 0, 0, d);
        unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("block of %d threads: SIMD of wave 0..%d:\n", nt, nt / 64 - 1);
        for (int b = 0; b < 8; ++b) { printf("  block %d:", b); for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("   (cu %u)\n", (h[b * 16] >> 8) & 15); }
    }
    return 0;
}
