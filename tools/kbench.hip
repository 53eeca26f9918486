// tools/kbench.hip -- standalone dev harness (not shipped): launch-rate and phase experiments on the step kernel.
// build: hipcc -std=c++20 -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/kbench.hip -o tools/kbench
#include "../quadsim_amd/csrc/quadsim_hip.hip"
#include <chrono>
#include <vector>

__global__ void k_empty(StepArgs A) { if (A.n < 0) A.st[0] = 1.0f; }

// phase ablations of k_env<0,false,1>: 0 full, 1 no compute (load+store only), 2 load only, 3 compute only
template <int WHAT>
__global__ __launch_bounds__(kBlock) void k_ablate(StepArgs A, unsigned long long *stamps)
{
    const int lane = threadIdx.x & (kTile - 1);
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + (threadIdx.x >> 6);
    const int64_t env = tile * kTile + lane;
    if (env >= A.n) return;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    Env e;
    load_env(A.st, tile, lane, e);
    Par P = A.par_nom;
    const float4 av = reinterpret_cast<const float4 *>(A.actions)[env];
    float a[4] = {av.x, av.y, av.z, av.w};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float obs[12] = {0}, reward = 0; unsigned flags = 0; bool done = false;
    if (WHAT == 0 || WHAT == 3) step_and_maybe_reset<0, false, 1>(e, P, a, A, env, A.step_idx, obs, reward, flags, done, true);
    else { obs[0] = e.sc[0] + a[0]; reward = e.ls; }
    asm volatile("" :: "v"(obs[0]), "v"(obs[11]), "v"(reward));
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (WHAT != 2 && WHAT != 3) {
        store_env(A.st, tile, lane, e);
        store_obs(A.obs, env, obs);
        A.reward[env] = reward;
        A.done[env] = done ? 1 : 0;
        if (A.flags) A.flags[env] = (uint8_t)flags;
    } else if (obs[3] == 123.456f) A.reward[env] = reward + e.sc[5];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if (stamps && lane == 0) { stamps[tile * 4 + 0] = t0; stamps[tile * 4 + 1] = t1; stamps[tile * 4 + 2] = t2; stamps[tile * 4 + 3] = t3; }
}

// memory-only variants of the tile I/O (no compute): MODE 0 dword SoA (product), 1 LDS-staged dwordx4 plain,
// 2 LDS-staged dwordx4 nontemporal stores, 3 LDS-staged dwordx4 sc1 (write-through) stores
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_io(StepArgs A)
{
    __shared__ float4 lds4[(kBlock / kTile) * (kRecWords * kTile / 4)];   // 4 tiles x 10 KiB
    const int lane = threadIdx.x & (kTile - 1);
    const int w = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / kTile) + w;
    const int64_t env = tile * kTile + lane;
    if (env >= A.n) return;
    float v[kRecWords];
    const float4 av = reinterpret_cast<const float4 *>(A.actions)[env];
    float4 *t4 = lds4 + w * (kRecWords * kTile / 4);
    float *tf = reinterpret_cast<float *>(t4);
    float4 *g4 = reinterpret_cast<float4 *>(A.st + tile * (int64_t)(kRecWords * kTile));
    if (MODE == 0) {
        const float *b = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
#pragma unroll
        for (int f = 0; f < kRecWords; ++f) v[f] = b[f * kTile];
    } else {
#pragma unroll
        for (int j = 0; j < kRecWords / 4; ++j) t4[j * kTile + lane] = g4[j * kTile + lane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int f = 0; f < kRecWords; ++f) v[f] = tf[f * kTile + lane];
    }
    v[0] += av.x * 1e-9f; v[39] += av.w * 1e-9f;
    float obs[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) obs[i] = v[i] + v[13 + i];
    if (MODE == 0) {
        float *b = A.st + tile * (int64_t)(kRecWords * kTile) + lane;
#pragma unroll
        for (int f = 0; f < kRecWords; ++f) b[f * kTile] = v[f];
        store_obs(A.obs, env, obs);
    } else {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int f = 0; f < kRecWords; ++f) tf[f * kTile + lane] = v[f];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < kRecWords / 4; ++j) {
            float4 x = t4[j * kTile + lane];
            float4 *dst = g4 + j * kTile + lane;
            if (MODE == 1) *dst = x;
            else if (MODE == 2) { typedef float v4f __attribute__((ext_vector_type(4))); v4f xv = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(xv, reinterpret_cast<v4f *>(dst)); }
            else { typedef float v4f __attribute__((ext_vector_type(4))); v4f xv = {x.x, x.y, x.z, x.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(xv) : "memory"); }
        }
        // obs: stage [64][12] row-major through LDS -> 3 contiguous 1 KiB stores
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 12; ++i) tf[lane * 12 + i] = obs[i];
        __builtin_amdgcn_wave_barrier();
        float4 *o4 = reinterpret_cast<float4 *>(A.obs + tile * kTile * 12);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float4 x = t4[j * kTile + lane];
            float4 *dst = o4 + j * kTile + lane;
            if (MODE == 1) *dst = x;
            else if (MODE == 2) { typedef float v4f __attribute__((ext_vector_type(4))); v4f xv = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(xv, reinterpret_cast<v4f *>(dst)); }
            else { typedef float v4f __attribute__((ext_vector_type(4))); v4f xv = {x.x, x.y, x.z, x.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(xv) : "memory"); }
        }
    }
    A.reward[env] = v[38];
    A.done[env] = v[39] > 1e30f;
    if (A.flags) A.flags[env] = 0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <typename F> double period_us(F &&launch, int K, hipStream_t s)
{
    for (int i = 0; i < 50; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto h0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < K; ++i) launch(i);
    auto h1 = std::chrono::steady_clock::now();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double host = std::chrono::duration<double, std::micro>(h1 - h0).count() / K;
    printf("   [host enqueue %.2f us/launch] ", host);
    return ms * 1e3 / K;
}

int main(int argc, char **argv)
{
    int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    QsConfig cfg; qs_config_default(&cfg);
    cfg.num_envs = N; cfg.auto_reset = 1; cfg.randomise = 1; cfg.seed = 1;
    cfg.init_range[0] = 0.5f; cfg.init_range[1] = 0.1f; cfg.init_range[2] = 0.2f; cfg.init_range[3] = 0.1f;
    QsEnv *e; if (qs_create(&cfg, &e)) { printf("create: %s\n", qs_last_error()); return 1; }
    const int P = 64;
    float *act, *obs, *rew; uint8_t *done, *flags; unsigned long long *stamps;
    CK(hipMalloc(&act, P * N * 16)); CK(hipMalloc(&obs, N * 48)); CK(hipMalloc(&rew, N * 4)); CK(hipMalloc(&done, N)); CK(hipMalloc(&flags, N));
    CK(hipMalloc(&stamps, (N / 64 + 4) * 32));
    qs_reset(e, nullptr, nullptr);
    qs_fill_random_actions(e, P, 0, act);
    qs_sync(e);
    hipStream_t s = e->stream;
    const unsigned grid = grid_tiles(N);
    StepArgs A = make_args(e); A.actions = act; A.obs = obs; A.reward = rew; A.done = done; A.flags = flags;
    int K = 2000;
    printf("N=%lld grid=%u blocks\n", (long long)N, grid);
    printf("empty kernel (same kernarg)   period %.2f us\n", period_us([&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    printf("qs_step (C ABI)                period %.2f us\n", period_us([&](int i) { qs_step(e, act + (i % P) * N * 4, obs, rew, done, flags, nullptr); }, K, s));
    printf("k_env direct                   period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; A.step_idx = i; hipLaunchKernelGGL((k_env<0, false, 1>), dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    printf("ablate full                    period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_ablate<0>), dim3(grid), dim3(kBlock), 0, s, A, (unsigned long long *)nullptr); }, K, s));
    printf("ablate load+store (no compute) period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_ablate<1>), dim3(grid), dim3(kBlock), 0, s, A, (unsigned long long *)nullptr); }, K, s));
    printf("ablate load only               period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_ablate<2>), dim3(grid), dim3(kBlock), 0, s, A, (unsigned long long *)nullptr); }, K, s));
    printf("ablate load+compute (no store) period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_ablate<3>), dim3(grid), dim3(kBlock), 0, s, A, (unsigned long long *)nullptr); }, K, s));
    printf("io dword SoA                   period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_io<0>), dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    printf("io LDS x4 plain                period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_io<1>), dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    printf("io LDS x4 nt stores            period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_io<2>), dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    printf("io LDS x4 sc1 stores           period %.2f us\n", period_us([&](int i) { A.actions = act + (i % P) * N * 4; hipLaunchKernelGGL((k_io<3>), dim3(grid), dim3(kBlock), 0, s, A); }, K, s));
    // in-kernel stamps of one launch (100 MHz realtime counter)
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_ablate<0>), dim3(grid), dim3(kBlock), 0, s, A, stamps);
        CK(hipStreamSynchronize(s));
    }
    int64_t tiles = (N + 63) / 64;
    std::vector<unsigned long long> h(tiles * 4);
    CK(hipMemcpy(h.data(), stamps, tiles * 32, hipMemcpyDeviceToHost));
    unsigned long long t_first = ~0ull, t_last = 0; double ld = 0, cp = 0, st = 0;
    for (int64_t i = 0; i < tiles; ++i) { t_first = std::min(t_first, h[i * 4]); t_last = std::max(t_last, h[i * 4 + 3]); ld += h[i*4+1]-h[i*4]; cp += h[i*4+2]-h[i*4+1]; st += h[i*4+3]-h[i*4+2]; }
    unsigned long long start_spread = 0; for (int64_t i = 0; i < tiles; ++i) start_spread = std::max(start_spread, h[i * 4] - t_first);
    printf("stamps (10 ns ticks): first wave start -> last wave end %.2f us; wave start spread %.2f us; per-wave mean load %.2f us compute %.2f us store %.2f us\n",
           (t_last - t_first) * 0.01, start_spread * 0.01, ld / tiles * 0.01, cp / tiles * 0.01, st / tiles * 0.01);
    return 0;
}
