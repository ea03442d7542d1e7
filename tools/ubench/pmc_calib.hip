// pmc_calib.hip - known-bytes kernels for calibrating the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -o w -- ./pmc_calib          (and the same with FETCH_SIZE; tools/gpu_run.sh pmc_calib)
//
// Every kernel moves exactly NB = 256 MiB in the direction under test (and nothing else worth counting), so the counter value per launch
// should read 262144 KiB if the counter means what the roofline arithmetic takes it to mean.  Store patterns:
//   write_x4            one coalesced 16-B store per lane: 1 KiB contiguous per wave instruction
//   write_b32           one coalesced 4-B store per lane: 256 B contiguous per wave instruction
//   write_wino          the output stage of the Winograd convs (kernels_wino.hip): buffer_store_b32, lanes 0-31 write 32 consecutive channels
//                       (128 B) of one pixel, lanes 32-63 the same channels of the pixel 4 to the right (two 128-B segments per instruction,
//                       4 Cout floats apart); a thread walks its 4 x 4 pixels row-major; the second half of a pixel's 256-B channel row comes
//                       from another wave a round later - exactly the order of the real epilogue for Cout = 64 (nt = the same with the
//                       non-temporal bit the kernels set)
//   write_half_line     64 B per wave-instruction segment (16 lanes) with the other half of every 128-B line never written: what a partial
//                       line costs in the counter
//   read_x4 / read_b32  the loads, for FETCH_SIZE (a lane's value is summed and stored only if it is NaN: no writes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const size_t NB = 256ull << 20;

__global__ void write_x4(float4 *out, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ void write_b32(float *out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
// image [H][W][64] floats; block = 512 threads = two 16x16-pixel sub-blocks; thread = (tile pair, channel lane & 31), stores the 16 pixels of
// its two 4x4 tiles; two rounds of 32 channels
template <int NT>
__global__ void write_wino(float *out, int H, int W)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c31 = lane & 31, hh = lane >> 5;
    const int sbx = W / 16, sb = 2 * blockIdx.x + (wave >> 2);
    const int by = sb / sbx, bx = sb - by * sbx;
    if (by * 16 >= H) return;
    const int oy = 16 * by + 4 * (wave & 3), ox = 16 * bx + 8 * hh;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0x7fffffff, 0x00020000);
    const unsigned ooff = (unsigned)(((size_t)oy * W + ox) * 64 + c31) * 4u;
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)(k + e)), rs, ooff, ((k >> 2) * W + 4 * e + (k & 3)) * 256 + 128 * r, NT);
        __syncthreads();
    }
}
__global__ void write_half_line(float *out, size_t nlines)
{
    // 16 lanes x 4 B = the first 64 B of every 128-B line
    const size_t l = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4;
    const int q = threadIdx.x & 15;
    for (size_t i = l; i < nlines; i += ((size_t)gridDim.x * blockDim.x) >> 4) out[i * 32 + q] = (float)i;
}
__global__ void read_x4(const float4 *in, size_t n4, float *sink)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s != s) sink[0] = s;
}
__global__ void read_b32(const float *in, size_t n, float *sink)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s != s) sink[0] = s;
}

int main()
{
    float *buf, *sink;
    CHECK(hipMalloc(&buf, 2 * NB));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, 2 * NB));
    const int H = 1024, W = 1024;        // 1024 x 1024 x 64 floats = 256 MiB
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(write_x4, dim3(4096), dim3(256), 0, 0, (float4 *)buf, NB / 16);
        hipLaunchKernelGGL(write_b32, dim3(4096), dim3(256), 0, 0, buf, NB / 4);
        hipLaunchKernelGGL((write_wino<0>), dim3((H / 16) * (W / 16) / 2), dim3(512), 0, 0, buf, H, W);
        hipLaunchKernelGGL((write_wino<2>), dim3((H / 16) * (W / 16) / 2), dim3(512), 0, 0, buf, H, W);
        hipLaunchKernelGGL(write_half_line, dim3(4096), dim3(256), 0, 0, buf, 2 * NB / 128);        // 256 MiB in 64-B halves of 512 MiB of lines
        hipLaunchKernelGGL(read_x4, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, NB / 16, sink);
        hipLaunchKernelGGL(read_b32, dim3(4096), dim3(256), 0, 0, (const float *)buf, NB / 4, sink);
        CHECK(hipDeviceSynchronize());
    }
    // check the Winograd pattern covered the image exactly once (host side, last launch order: half_line overwrote the first halves; redo)
    CHECK(hipMemset(buf, 0xff, NB));
    hipLaunchKernelGGL((write_wino<0>), dim3((H / 16) * (W / 16) / 2), dim3(512), 0, 0, buf, H, W);
    CHECK(hipDeviceSynchronize());
    float *h = (float *)malloc(NB);
    CHECK(hipMemcpy(h, buf, NB, hipMemcpyDeviceToHost));
    size_t missing = 0;
    for (size_t i = 0; i < NB / 4; ++i) missing += h[i] != h[i];
    printf("pmc_calib: every kernel moves %zu bytes (= %zu KiB) per launch; write_wino left %zu of %zu floats unwritten\n", NB, NB >> 10, missing, NB / 4);
    return 0;
}
