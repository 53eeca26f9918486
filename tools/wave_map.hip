// Where do the 8 waves of a 512-thread workgroup with 154 KB of LDS land?  Prints (CU, SIMD) per wave for a few workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/wave_map.hip -o tools/wave_map && tools/wave_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(unsigned *out)
{
    __shared__ char big[154 * 1024];
    big[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw | (xcc << 28) | (big[threadIdx.x] & 0);
    }
}
int main()
{
    unsigned *d, h[256 * 8];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 256; b += 37) {
        printf("wg %3d:", b);
        for (int w = 0; w < 8; ++w) {
            const unsigned v = h[b * 8 + w];
            printf("  w%d xcc%u se%u cu%2u simd%u slot%u", w, v >> 28, (v >> 13) & 7, (v >> 8) & 15, (v >> 4) & 3, v & 15);
        }
        printf("\n");
    }
    return 0;
}
