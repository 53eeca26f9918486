// xcc_map.hip -- which XCD does workgroup b of a dispatch land on, and is that stable from launch to launch?
// (HIP promises nothing; this measures what the hardware does.)  hipcc --offload-arch=gfx950 tools/xcc_map.hip -o tools/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_xcc(unsigned char *out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned char)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15);
}
__global__ void k_noise(int *p, int spin)
{
    int a = 0;
    for (int i = 0; i < spin; ++i) a += __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0 && a == -1) *p = a;
}

int main()
{
    const int B = 1024, L = 200;
    unsigned char *d;
    int *dn;
    hipMalloc(&d, (size_t)B * L);
    hipMalloc(&dn, 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<unsigned char> h((size_t)B * L);
    const char *names[4] = {"back-to-back launches, one stream", "a 3-block kernel between launches, same stream",
                            "3/5/7-block kernels running on ANOTHER stream meanwhile", "128-thread blocks, 1021-block noise kernels on another stream"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d, 0xff, (size_t)B * L);
        hipDeviceSynchronize();
        for (int l = 0; l < L; ++l) {
            if (mode == 1) hipLaunchKernelGGL(k_noise, dim3(3), dim3(64), 0, s1, dn, 10);
            if (mode == 2) hipLaunchKernelGGL(k_noise, dim3(3 + 2 * (l % 3)), dim3(256), 0, s2, dn, 2000);
            if (mode == 3) hipLaunchKernelGGL(k_noise, dim3(1021), dim3(128), 0, s2, dn, 500);
            hipLaunchKernelGGL(k_xcc, dim3(B), dim3(mode == 3 ? 128 : 256), 0, s1, d + (size_t)l * B);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, (size_t)B * L, hipMemcpyDeviceToHost);
        int changed = 0, not_mod8 = 0;
        for (int l = 0; l < L; ++l)
            for (int b = 0; b < B; ++b) {
                if (h[(size_t)l * B + b] != h[b]) ++changed;
                if (h[(size_t)l * B + b] != (unsigned char)((h[(size_t)l * B] + b) % 8)) ++not_mod8;
            }
        printf("%-62s: block 0 on XCD %d..., blocks whose XCD differs from launch 0: %d of %d; blocks off the (x0 + b) %% 8 pattern: %d\n",
               names[mode], h[0], changed, B * L, not_mod8);
        printf("    first launch, blocks 0-15: ");
        for (int b = 0; b < 16; ++b) printf("%d ", h[b]);
        printf("| launch 100: ");
        for (int b = 0; b < 16; ++b) printf("%d ", h[100 * B + b]);
        printf("\n");
    }
    return 0;
}
