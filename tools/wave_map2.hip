// (CU, SIMD) of the two waves of every 128-thread workgroup (7.5 KiB LDS) of a 1024-workgroup launch: which roles of k_env_split
// end up sharing a SIMD.   hipcc --offload-arch=gfx950 -O2 tools/wave_map2.hip -o tools/wave_map2 && tools/wave_map2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void k(unsigned *out, int spin)
{
    __shared__ char lds[7680];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;     // keep the wave resident while the others arrive
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        out[blockIdx.x * 2 + (threadIdx.x >> 6)] = (hw & 0x0fffffffu) | (xcc << 28) | ((x == 12345.0f) ? 1u : 0u) | (lds[1] & 0);
    }
}
int main()
{
    const int nb = 1024;
    unsigned *d; std::vector<unsigned> h(nb * 2);
    (void)hipMalloc(&d, h.size() * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(128), 0, 0, d, 20000);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu, simd): how many role-0 and role-1 waves
    std::map<unsigned, std::pair<int, int>> m;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 2; ++w) {
            const unsigned v = h[b * 2 + w];
            const unsigned key = ((v >> 28) << 16) | (((v >> 13) & 7) << 12) | (((v >> 8) & 15) << 4) | ((v >> 4) & 3);
            if (w == 0) m[key].first++; else m[key].second++;
        }
    int hist[5][5] = {};
    for (auto &kv : m) { int a = kv.second.first, b = kv.second.second; if (a < 5 && b < 5) hist[a][b]++; }
    printf("SIMDs used: %zu; count of SIMDs by (role-0 waves, role-1 waves):\n", m.size());
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) if (hist[a][b]) printf("  (%d chaser, %d target): %d SIMDs\n", a, b, hist[a][b]);
    for (int b = 0; b < 4; ++b) printf("wg %d: w0 simd%u slot%u  w1 simd%u slot%u\n", b, (h[2*b] >> 4) & 3, h[2*b] & 15, (h[2*b+1] >> 4) & 3, h[2*b+1] & 15);
    return 0;
}
