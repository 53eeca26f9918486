// Where do the two role waves of the step kernel's 128-thread workgroups land?  1024 workgroups (65 536 envs) x 2 waves, all
// resident at once (each wave spins ~20 us): per (XCC, CU, SIMD) the number of wave-0 ("chaser") and wave-1 ("target") waves.
//   hipcc --offload-arch=gfx950 -O2 tools/wave_map3.hip -o tools/wave_map3 && tools/wave_map3 [threads-per-block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
__global__ void k(unsigned *out, int spin)
{
    __shared__ char lds[7680];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = (hw & 0x0fffffffu) | (xcc << 28) | (lds[threadIdx.x] & 0);   // HW_ID's own top bits (STATE_ID / ME_ID) masked off
    }
}
int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 128, wpb = threads / 64, total = argc > 2 ? atoi(argv[2]) : 2048, blocks = total / wpb;
    unsigned *d, *h = (unsigned *)malloc(total * 4);
    hipMalloc(&d, total * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 2000);
    hipMemcpy(h, d, total * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::pair<int, int>> m;   // (xcc, se, cu, simd) -> (role0, role1)
    int same_simd = 0, same_cu = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned key[8][4];
        for (int w = 0; w < wpb; ++w) {
            const unsigned v = h[b * wpb + w];
            key[w][0] = v >> 28; key[w][1] = (v >> 12) & 15; key[w][2] = (v >> 8) & 15; key[w][3] = (v >> 4) & 3;   // SE_ID:SH_ID together
            auto &e = m[std::make_tuple(key[w][0], key[w][1], key[w][2], key[w][3])];
            if ((w & 1) == 0) ++e.first; else ++e.second;
        }
        for (int w = 0; w + 1 < wpb; w += 2) {
            if (key[w][0] == key[w + 1][0] && key[w][1] == key[w + 1][1] && key[w][2] == key[w + 1][2]) {
                ++same_cu;
                if (key[w][3] == key[w + 1][3]) ++same_simd;
            }
        }
        if (b < 6) {
            printf("wg %d:", b);
            for (int w = 0; w < wpb; ++w) printf("  w%d xcc%u se%u cu%u simd%u", w, key[w][0], key[w][1], key[w][2], key[w][3]);
            printf("\n");
        }
    }
    std::map<std::tuple<unsigned, unsigned, unsigned>, int> cus;
    for (auto &kv : m) cus[std::make_tuple(std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first))] += kv.second.first + kv.second.second;
    int wh[64] = {0};
    for (auto &kv : cus) ++wh[kv.second > 63 ? 63 : kv.second];
    printf("distinct CUs hosting waves: %zu; CUs by resident waves:", cus.size());
    for (int i = 0; i < 64; ++i) if (wh[i]) printf("  %d waves: %d CUs", i, wh[i]);
    printf("\n");
    int per_xcc[16] = {0};
    for (auto &kv : cus) ++per_xcc[std::get<0>(kv.first)];
    printf("CUs used per XCC:"); for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]); printf("\n");
    int hist[16][16] = {{0}};
    for (auto &kv : m) ++hist[kv.second.first > 15 ? 15 : kv.second.first][kv.second.second > 15 ? 15 : kv.second.second];
    printf("%d-thread workgroups: %d; role pairs on the same CU %d, on the same SIMD %d; SIMDs used %zu\n", threads, blocks, same_cu, same_simd, m.size());
    printf("SIMDs by (even-role waves, odd-role waves) resident:\n");
    for (int a = 0; a < 16; ++a)
        for (int c = 0; c < 16; ++c)
            if (hist[a][c]) printf("  (%d, %d): %d SIMDs\n", a, c, hist[a][c]);
    return 0;
}
