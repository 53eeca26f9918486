// rollout_ops.hpp -- the steps either side of env.step in the trainer's Runner:
// GAE(lambda) reverse scan (rl_baselines/ppo2/ppo2.py:507-520) and swap_and_flatten (:531-539) on [T,N,.]
// roll-out buffers resident in HBM.  Included by quadsim_hip.hip (one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qs {

constexpr int kGaeChunk = 64;   // steps per chunk of the two-pass scan

// GAE is the first-order linear recurrence A_t = delta_t + k_t A_{t+1}, k_t = gamma lam nonterminal_{t+1}.
// One lane per (env, chunk of 64 steps) so that the T-long dependent chain becomes T/64-long and the grid has
// T/64 times more waves to hide the load latency with.  Pass 1 reduces each chunk to (P = prod k, S = the chunk's
// advantage at its first step assuming A = 0 behind it); pass 2 rebuilds the incoming A of a chunk from the
// chunks behind it (<= T/64 fused multiply-adds) and re-walks the chunk writing advantages and returns.
struct GaeArgs {
    const float *rewards, *values, *last_values;
    const uint8_t *dones, *last_dones;
    float *advs, *returns, *ws;   // ws: [2][C][N]
    int64_t T, N, C;
    float gamma, lam;
};

__device__ __forceinline__ void gae_terms(const GaeArgs &G, int64_t t, int64_t i, float &delta, float &k)
{
    float nextv, nonterm;
    if (t == G.T - 1) {                                            // ppo2.py:512-514
        nonterm = G.last_dones[i] ? 0.0f : 1.0f;
        nextv = G.last_values[i];
    } else {                                                       // :516-517
        nonterm = G.dones[(t + 1) * G.N + i] ? 0.0f : 1.0f;
        nextv = G.values[(t + 1) * G.N + i];
    }
    delta = G.rewards[t * G.N + i] + (G.gamma * nextv) * nonterm - G.values[t * G.N + i];   // :518
    k = G.gamma * G.lam * nonterm;                                  // :519
}

__global__ __launch_bounds__(256) void k_gae_reduce(GaeArgs G)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (i >= G.N) return;
    const int64_t t0 = c * kGaeChunk, t1 = min(G.T, t0 + kGaeChunk);
    float P = 1.0f, S = 0.0f;
    for (int64_t t = t1 - 1; t >= t0; --t) {
        float d, k;
        gae_terms(G, t, i, d, k);
        S = fmaf(k, S, d);
        P *= k;
    }
    G.ws[c * G.N + i] = P;
    G.ws[(G.C + c) * G.N + i] = S;
}

__global__ __launch_bounds__(256) void k_gae_apply(GaeArgs G)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (i >= G.N) return;
    float A = 0.0f;                                                 // last_gae_lam = 0, :510
    for (int64_t cc = G.C - 1; cc > c; --cc) A = fmaf(G.ws[cc * G.N + i], A, G.ws[(G.C + cc) * G.N + i]);
    const int64_t t0 = c * kGaeChunk, t1 = min(G.T, t0 + kGaeChunk);
    for (int64_t t = t1 - 1; t >= t0; --t) {
        float d, k;
        gae_terms(G, t, i, d, k);
        A = fmaf(k, A, d);
        G.advs[t * G.N + i] = A;
        G.returns[t * G.N + i] = A + G.values[t * G.N + i];         // :520
    }
}

// Single-pass variant for wide batches: one lane walks one env's whole T-step chain, 16 steps at a time: the 48
// loads of a group are issued before its 16 dependent fmas, so every wave keeps ~9 KiB in flight and the data
// is read exactly once (17 B per (t, env) instead of 26 B for the two-pass scan).
constexpr int kGaeGroup = 16;
__global__ __launch_bounds__(256) void k_gae_serial(GaeArgs G)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= G.N) return;
    float A = 0.0f;
    float nextv = G.last_values[i];
    float nonterm = G.last_dones[i] ? 0.0f : 1.0f;
    int64_t t = G.T - 1;
    const float gl = G.gamma * G.lam;
    for (; t >= kGaeGroup - 1; t -= kGaeGroup) {
        float r[kGaeGroup], v[kGaeGroup];
        uint8_t d[kGaeGroup];
#pragma unroll
        for (int j = 0; j < kGaeGroup; ++j) {
            const int64_t o = (t - j) * G.N + i;
            r[j] = G.rewards[o]; v[j] = G.values[o]; d[j] = G.dones[o];
        }
#pragma unroll
        for (int j = 0; j < kGaeGroup; ++j) {
            const float delta = r[j] + (G.gamma * nextv) * nonterm - v[j];
            A = fmaf(gl * nonterm, A, delta);
            const int64_t o = (t - j) * G.N + i;
            G.advs[o] = A;
            G.returns[o] = A + v[j];
            nextv = v[j];
            nonterm = d[j] ? 0.0f : 1.0f;          // dones[t] gates the step before it (:516)
        }
    }
    for (; t >= 0; --t) {
        const int64_t o = t * G.N + i;
        const float r = G.rewards[o], v = G.values[o];
        const float delta = r + (G.gamma * nextv) * nonterm - v;
        A = fmaf(gl * nonterm, A, delta);
        G.advs[o] = A;
        G.returns[o] = A + v;
        nextv = v;
        nonterm = G.dones[o] ? 0.0f : 1.0f;
    }
}

// swap_and_flatten (ppo2.py:531-539): in [T][N][D] -> out [N][T][D].  32 x 32 tile of D-float rows staged in
// LDS so that both the global reads (rows of consecutive envs) and writes (rows of consecutive steps) are
// contiguous runs of 32*D floats.
template <int D>
__global__ __launch_bounds__(256) void k_swap_flatten(const float *__restrict__ in, float *__restrict__ out, int64_t T, int64_t N)
{
    __shared__ float tile[32][32 * D + 1];
    const int64_t i0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    for (int idx = threadIdx.x; idx < 32 * 32 * D; idx += 256) {
        const int tt = idx / (32 * D), r = idx - tt * (32 * D);      // r = ii*D + d
        const int64_t t = t0 + tt, i = i0 + r / D;
        if (t < T && i < N) tile[tt][r] = in[(t * N + i0) * D + r];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32 * D; idx += 256) {
        const int ii = idx / (32 * D), r = idx - ii * (32 * D);      // r = tt*D + d
        const int tt = r / D, d = r - tt * D;
        const int64_t i = i0 + ii, t = t0 + tt;
        if (t < T && i < N) out[(i * T + t0) * D + r] = tile[tt][ii * D + d];
    }
}

// float4 flavour for row widths that are multiples of 4 floats (actions D=4, observations D=12): the same
// 32 x 32 tile, moved 16 B per lane
template <int D4>
__global__ __launch_bounds__(256) void k_swap_flatten_v4(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t T, int64_t N)
{
    __shared__ float4 tile[32][32 * D4 + 1];
    const int64_t i0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    const int ni = (int)min((int64_t)32, N - i0), nt = (int)min((int64_t)32, T - t0);
    for (int idx = threadIdx.x; idx < 32 * 32 * D4; idx += 256) {
        const int tt = idx / (32 * D4), r = idx - tt * (32 * D4);      // r = ii*D4 + q
        if (tt < nt && r < ni * D4) tile[tt][r] = in[((t0 + tt) * N + i0) * D4 + r];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32 * D4; idx += 256) {
        const int ii = idx / (32 * D4), r = idx - ii * (32 * D4);      // r = tt*D4 + q
        const int tt = r / D4, q = r - tt * D4;
        if (ii < ni && tt < nt) out[((i0 + ii) * T + t0) * D4 + r] = tile[tt][ii * D4 + q];
    }
}

}  // namespace qs
