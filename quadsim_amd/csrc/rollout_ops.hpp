// rollout_ops.hpp -- the steps either side of env.step in the trainer's Runner:
// GAE(lambda) reverse scan (rl_baselines/ppo2/ppo2.py:507-520) and swap_and_flatten (:531-539) on [T,N,.]
// roll-out buffers resident in HBM.  Included by quadsim_hip.hip (one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef QS_ASSERT
#define QS_ASSERT(c) ((void)0)
#endif

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
            QS_ASSERT(o >= 0 && o < G.T * G.N);
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
template <int D, typename E = float>
__global__ __launch_bounds__(256) void k_swap_flatten(const E *__restrict__ in, E *__restrict__ out, int64_t T, int64_t N)
{
    __shared__ E tile[32][32 * D + 1];
    const int64_t i0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
    for (int idx = threadIdx.x; idx < 32 * 32 * D; idx += 256) {
        const int tt = idx / (32 * D), r = idx - tt * (32 * D);      // r = ii*D + d
        const int64_t t = t0 + tt, i = i0 + r / D;
        QS_ASSERT(!(t < T && i < N) || (t * N + i0) * D + r < T * N * D);
        if (t < T && i < N) tile[tt][r] = in[(t * N + i0) * D + r];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32 * D; idx += 256) {
        const int ii = idx / (32 * D), r = idx - ii * (32 * D);      // r = tt*D + d
        const int tt = r / D, d = r - tt * D;
        const int64_t i = i0 + ii, t = t0 + tt;
        QS_ASSERT(!(t < T && i < N) || (i * T + t0) * D + r < T * N * D);
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
        QS_ASSERT(!(tt < nt && r < ni * D4) || ((t0 + tt) * N + i0) * D4 + r < T * N * D4);
        if (tt < nt && r < ni * D4) tile[tt][r] = in[((t0 + tt) * N + i0) * D4 + r];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32 * D4; idx += 256) {
        const int ii = idx / (32 * D4), r = idx - ii * (32 * D4);      // r = tt*D4 + q
        const int tt = r / D4, q = r - tt * D4;
        QS_ASSERT(!(ii < ni && tt < nt) || ((i0 + ii) * T + t0) * D4 + r < T * N * D4);
        if (ii < ni && tt < nt) out[((i0 + ii) * T + t0) * D4 + r] = tile[tt][ii * D4 + q];
    }
}


// ---- GAE + the env-major flatten of everything that is one scalar per (t, env), in ONE pass -----------------------------
// Runner.run (rl_baselines/ppo2/ppo2.py:507-523) computes mb_returns from (rewards, values, dones) and then hands
// swap_and_flatten(mb_returns), (mb_dones), (mb_values), (mb_neglogpacs), (true_reward) to the trainer.  The reverse scan
// already holds r, v, d of 16 steps per lane in registers, so it can emit all of them env-major itself: each lane writes
// 64 contiguous bytes per array and group (four 16-B stores) instead of re-reading [T,N] arrays in five more launches.
// Reads 13 B, writes 17 B (+ 8 B for the optional time-major advs / returns) per (t, env).
struct GaeFlatArgs {
    const float *rewards, *values, *neglogp, *last_values;   // [T,N] x3 (neglogp nullable), [N]
    const uint8_t *dones, *last_dones;                        // [T,N], [N]
    float *f_returns, *f_values, *f_neglogp, *f_rewards;      // [N,T] env-major (f_neglogp nullable with neglogp)
    uint8_t *f_masks;                                         // [N,T] env-major mb_dones (0/1)
    float *advs, *returns;                                    // nullable [T,N] time-major (what qs_gae returns)
    int64_t T, N;
    float gamma, lam;
};

// per-wave LDS staging of one 64-env x 16-step block: written one env row per lane, read back so that FOUR lanes cover the
// 16 consecutive steps of one env -- a store instruction then writes 16 runs of 64 contiguous bytes instead of 64 runs of 16
constexpr int kGfStride = kGaeGroup + 1;

__device__ __forceinline__ void gf_emit_f32(float *__restrict__ stage, const float x[kGaeGroup], float *__restrict__ out,
                                            int64_t tile_env0, int64_t t_lo, int64_t T, int64_t N, int lane)
{
    // x[j] belongs to step t_lo + (kGaeGroup - 1 - j)
#pragma unroll
    for (int j = 0; j < kGaeGroup; ++j) stage[lane * kGfStride + (kGaeGroup - 1 - j)] = x[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int q = lane & 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (lane >> 2) + 16 * p;
        const float *src = stage + e * kGfStride + 4 * q;
        const float4 v = make_float4(src[0], src[1], src[2], src[3]);
        const int64_t env = tile_env0 + e;
        if (env < N) *reinterpret_cast<float4 *>(out + env * T + t_lo + 4 * q) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void k_gae_flatten(GaeFlatArgs G)
{
    __shared__ float s_stage[4][64 * kGfStride];
    __shared__ uint8_t s_mask[4][64 * (kGaeGroup + 4)];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t tile_env0 = i - lane;
    const bool active = i < G.N;
    const int64_t ii = active ? i : 0;           // idle lanes of the tail wave walk env 0 and store nothing
    float *stage = s_stage[w];
    uint8_t *mstage = s_mask[w];
    float A = 0.0f;
    float nextv = G.last_values[ii];
    float nonterm = G.last_dones[ii] ? 0.0f : 1.0f;
    int64_t t = G.T - 1;
    const float gl = G.gamma * G.lam;
    const bool vec = (G.T & 3) == 0;                         // env-major rows are 16-B aligned
    const int64_t row = ii * G.T;
    for (; t >= kGaeGroup - 1; t -= kGaeGroup) {
        float r[kGaeGroup], v[kGaeGroup], nl[kGaeGroup], ret[kGaeGroup];
        uint8_t d[kGaeGroup];
#pragma unroll
        for (int j = 0; j < kGaeGroup; ++j) {
            const int64_t o = (t - j) * G.N + ii;
            QS_ASSERT(o >= 0 && o < G.T * G.N);
            r[j] = G.rewards[o]; v[j] = G.values[o]; d[j] = G.dones[o];
            nl[j] = G.neglogp ? G.neglogp[o] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < kGaeGroup; ++j) {
            const float delta = r[j] + (G.gamma * nextv) * nonterm - v[j];
            A = fmaf(gl * nonterm, A, delta);
            ret[j] = A + v[j];
            if (G.advs && active) { const int64_t o = (t - j) * G.N + ii; G.advs[o] = A; G.returns[o] = ret[j]; }
            nextv = v[j];
            nonterm = d[j] ? 0.0f : 1.0f;
        }
        // element j belongs to step t - j: the group covers steps [t-15, t] -> env-major offsets row + t-15 .. row + t
        const int64_t t_lo = t - (kGaeGroup - 1);
        QS_ASSERT(row + t_lo >= 0 && row + t_lo + kGaeGroup <= G.N * G.T);
        if (vec) {
            gf_emit_f32(stage, ret, G.f_returns, tile_env0, t_lo, G.T, G.N, lane);
            gf_emit_f32(stage, v, G.f_values, tile_env0, t_lo, G.T, G.N, lane);
            gf_emit_f32(stage, r, G.f_rewards, tile_env0, t_lo, G.T, G.N, lane);
            if (G.f_neglogp) gf_emit_f32(stage, nl, G.f_neglogp, tile_env0, t_lo, G.T, G.N, lane);
#pragma unroll
            for (int j = 0; j < kGaeGroup; ++j) mstage[lane * (kGaeGroup + 4) + (kGaeGroup - 1 - j)] = d[j] ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int q = lane & 3;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int e = (lane >> 2) + 16 * p;
                const uchar4 m = *reinterpret_cast<const uchar4 *>(mstage + e * (kGaeGroup + 4) + 4 * q);
                const int64_t env = tile_env0 + e;
                if (env < G.N) *reinterpret_cast<uchar4 *>(G.f_masks + env * G.T + t_lo + 4 * q) = m;
            }
            __builtin_amdgcn_wave_barrier();
        } else if (active) {
#pragma unroll
            for (int j = 0; j < kGaeGroup; ++j) {
                const int64_t o = row + t - j;
                G.f_returns[o] = ret[j]; G.f_values[o] = v[j]; G.f_rewards[o] = r[j];
                if (G.f_neglogp) G.f_neglogp[o] = nl[j];
                G.f_masks[o] = d[j] ? 1 : 0;
            }
        }
    }
    if (!active) return;
    for (; t >= 0; --t) {
        const int64_t o = t * G.N + i;
        const float r = G.rewards[o], v = G.values[o];
        const uint8_t d = G.dones[o];
        const float delta = r + (G.gamma * nextv) * nonterm - v;
        A = fmaf(gl * nonterm, A, delta);
        if (G.advs) { G.advs[o] = A; G.returns[o] = A + v; }
        const int64_t f = row + t;
        G.f_returns[f] = A + v; G.f_values[f] = v; G.f_rewards[f] = r;
        if (G.f_neglogp) G.f_neglogp[f] = G.neglogp[o];
        G.f_masks[f] = d ? 1 : 0;
        nextv = v;
        nonterm = d ? 0.0f : 1.0f;
    }
}

// ---- episode accounting of one roll-out (what the Monitor wrapper of run_docking_ppo2.py:19-35 reports through
// info['episode'] and Runner._run collects into ep_infos, ppo2.py:486-489): for every episode that ENDS inside the roll-out
// its return and length.  One lane per env walks t forward carrying (return, length) across roll-outs in ep_ret / ep_len.
// done-after-step-t = dones[t+1] (t < T-1) / last_dones (t = T-1): mb_dones holds the flags BEFORE each step (:479).
// Output is a compact list (key = t*N + env, return, length) in wave-segment order: pass 1 counts the wave's episodes,
// ONE atomic add per wave reserves its segment, pass 2 fills it ordered by (t, lane).  Sorting by key gives the
// reference's (step, env) order.
struct EpisodeArgs {
    const float *rewards;              // [T,N]
    const uint8_t *dones, *last_dones; // [T,N] flags before each step, [N] flags after the last one
    float *ep_ret;                     // [N] in/out: return of the unfinished episode
    int32_t *ep_len;                   // [N] in/out
    unsigned long long *count;         // device counter (zeroed by qs_episode_stats before the launch): episodes appended
    int64_t *out_key;                  // [cap]
    float *out_ret;                    // [cap]
    int32_t *out_len;                  // [cap]
    int64_t T, N, cap;
};

__device__ __forceinline__ bool ep_done_after(const EpisodeArgs &E, int64_t t, int64_t i)
{
    return (t + 1 < E.T ? E.dones[(t + 1) * E.N + i] : E.last_dones[i]) != 0;
}

constexpr int kEpGroup = 16;   // steps whose loads are in flight together (a lone wave per SIMD is latency-bound otherwise)

__global__ __launch_bounds__(256) void k_episode_stats(EpisodeArgs E)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < E.N;
    const int64_t ii = active ? i : 0;
    const int lane = threadIdx.x & 63;
    // pass 1: episodes ending in this lane's env, then in this wave
    unsigned mine = 0;
    int64_t t = 0;
    for (; t + kEpGroup <= E.T; t += kEpGroup) {
        uint8_t d[kEpGroup];
#pragma unroll
        for (int j = 0; j < kEpGroup; ++j) d[j] = (t + j + 1 < E.T ? E.dones[(t + j + 1) * E.N + ii] : E.last_dones[ii]);
#pragma unroll
        for (int j = 0; j < kEpGroup; ++j) mine += d[j] ? 1u : 0u;
    }
    for (; t < E.T; ++t) mine += ep_done_after(E, t, ii) ? 1u : 0u;
    if (!active) mine = 0;
    unsigned total = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(E.count, (unsigned long long)total);
    base = __shfl(base, 0, 64);
    // pass 2: the running sums; a step's finished episodes go to consecutive slots in lane order
    float ret = active ? E.ep_ret[ii] : 0.0f;
    int32_t len = active ? E.ep_len[ii] : 0;
    unsigned long long next = base;
    auto one = [&](int64_t tt, float r, bool d) {
        ret += r;
        len += 1;
        d = d && active;
        const unsigned long long bal = __ballot(d);
        if (bal) {
            if (d) {
                const unsigned long long slot = next + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
                QS_ASSERT(slot < base + total);
                if ((int64_t)slot < E.cap) { E.out_key[slot] = tt * E.N + ii; E.out_ret[slot] = ret; E.out_len[slot] = len; }
                ret = 0.0f;
                len = 0;
            }
            next += (unsigned long long)__popcll(bal);
        }
    };
    t = 0;
    for (; t + kEpGroup <= E.T; t += kEpGroup) {
        uint8_t d[kEpGroup];
        float r[kEpGroup];
#pragma unroll
        for (int j = 0; j < kEpGroup; ++j) {
            d[j] = (t + j + 1 < E.T ? E.dones[(t + j + 1) * E.N + ii] : E.last_dones[ii]);
            r[j] = E.rewards[(t + j) * E.N + ii];
        }
#pragma unroll
        for (int j = 0; j < kEpGroup; ++j) one(t + j, r[j], d[j] != 0);
    }
    for (; t < E.T; ++t) one(t, E.rewards[t * E.N + ii], ep_done_after(E, t, ii));
    if (active) { E.ep_ret[ii] = ret; E.ep_len[ii] = len; }
}

}  // namespace qs
